"""World = 8 without an 8-GPU node: the real CLI under torchrun with EIGHT ranks sharing this one GPU over gloo
(RCCL refuses a device twice) — the README's shape of BASELINE configs[3] / [4]: `--N_rand 20` over 8 ranks = 3/3/3/3/2/2/2/2
shard files per rank and step (reference: main.py:472-479, 802-805), pseudo-data poses i % 8 with rank-disjoint index ranges
(utils/create_data.py:297-299), test frames / video poses sharded over 8 ranks.  Not a measurement of anything: a walk of the
host logic and the real kernels — at the real model, W256 D88 — at the rank count the scaling run uses.  The CPU twin (no GPU, oracle gradients through the real
R2LTrainer host code) is tests/test_driver_cpu.py::test_eight_rank_gloo_trainer_host_logic."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O
from tests.test_driver_cpu import ROOT, make_scene
from tests.test_forward_gpu import build_model  # noqa: E402

pytestmark = pytest.mark.gpu
WORLD = 8


def _env():
    env = {k: v for k, v in os.environ.items() if not k.startswith("R2L_")}
    env.update(MASTER_ADDR="127.0.0.1", R2L_DIST_BACKEND="gloo")
    return env


def _torchrun(port, script, args, tmp_path, env=None, timeout=1500):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(WORLD), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, script)] + args
    r = subprocess.run(cmd, env=env or _env(), cwd=str(tmp_path), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    return r.stdout + r.stderr


@pytest.fixture(scope="module")
def scene_and_teacher(tmp_path_factory):
    root = tmp_path_factory.mktemp("w8")
    scene = str(root / "scene")
    os.makedirs(scene)
    make_scene(scene, size=128)  # half_res -> 64x64 = 4096 rays per pose = one shard per pose
    csd, fsd = O.make_teacher_state_dicts(5, 2, alpha_bias=0.5)
    torch.save({"network_fn_state_dict": csd, "network_fine_state_dict": fsd}, str(root / "teacher.tar"))
    return root, scene


def test_create_data_then_train_eight_ranks(scene_and_teacher):
    """configs[4] then configs[3] at world = 8: 21 poses over 8 ranks (shares 3/3/3/3/3/2/2/2 — poses i % 8, every rank's shards
    inside its own index range, none overwritten), then 6 training iterations with --N_rand 20: 3/3/3/3/2/2/2/2 shard files per
    rank and step, gradients weighted by ray share, per-rank hard-ray pools, the staged backward with 4 buckets in flight; the
    replicas end bit-identical and rank 0 writes the checkpoint."""
    from r2l_amd.checkpoint import load_ckpt
    from r2l_amd.create_data import shard_index_base
    root, scene = scene_and_teacher
    kd = str(root / "pseudo8")
    n_pose, chunk = 21, 2
    args = ["--create_data", "rand", "--config", os.path.join(ROOT, "configs", "lego.txt"), "--datadir", scene, "--teacher_ckpt",
            str(root / "teacher.tar"), "--create_data_chunk", str(chunk), "--datadir_kd", scene + ":" + kd, "--experiment_name",
            "cd8", "--n_pose_kd", str(n_pose)]
    _torchrun(29651, os.path.join("utils", "create_data.py"), args, root)
    fpf = (chunk * 64 * 64) // 4096
    want = []
    for rank in range(WORLD):
        mine = [i for i in range(1, n_pose + 1) if i % WORLD == rank]
        base = shard_index_base(rank, WORLD, n_pose, chunk, fpf)
        want += list(range(base, base + len(mine)))  # one 4096-ray shard per 64x64 pose
    idx = sorted(int(f[5:-4]) for f in os.listdir(kd))
    assert idx == sorted(want) and len(idx) == n_pose, (idx, sorted(want))
    for i in idx:
        a = np.load(os.path.join(kd, "data_%d.npy" % i))
        assert a.shape == (4096, 9) and np.isfinite(a).all()
    env = _env()
    env["R2L_CHECK_SYNC"] = "1"
    out = _torchrun(29653, "main.py",
                    ["--model_name", "R2L", "--config", os.path.join(ROOT, "configs", "lego_noview.txt"), "--datadir", scene,
                     "--n_sample_per_ray", "16", "--netwidth", "256", "--netdepth", "88", "--use_residual", "--trial.ON",
                     "--trial.body_arch", "resmlp", "--testskip", "1", "--datadir_kd", kd, "--data_mode", "rays", "--N_rand", "20",
                     "--hard_ratio", "0.2", "--hard_mul", "2", "--warmup_lr", "0.0001,200", "--i_print", "2", "--i_testset", "100",
                     "--i_weights", "6", "--N_iters", "6", "--experiment_name", "dp8"], root, env=env)
    assert "[3, 3, 3, 3, 2, 2, 2, 2] shard files per rank and step" in out, out[-3000:]
    assert "replicas in sync after 6 iterations: True (skipped steps: 0)" in out, out[-3000:]
    ckpts = [os.path.join(dp, f) for dp, _, fs in os.walk(root) for f in fs if f == "ckpt.tar" and "dp8" in dp]
    assert len(ckpts) == 1 and load_ckpt(ckpts[0])["global_step"] == 6


def test_cli_render_eight_ranks(scene_and_teacher):
    """`main.py --render_only` over 8 ranks: the test views (metrics all-reduced) and the 5-pose video (three ranks hold no pose
    at all; frames gathered to rank 0 and re-interleaved) come out as from one process."""
    from r2l_amd import driver
    from r2l_amd.checkpoint import save_ckpt
    root, scene = scene_and_teacher
    sd = O.make_state_dict(n_block=43, seed=1)  # the real model: W256 D88
    ckpt = str(root / "SERVER-20260101-000000_iter7" / "weights" / "ckpt.tar")
    save_ckpt(ckpt, 7, build_model(sd, 43).cpu(), {"state": {}, "param_groups": []}, 0., 0)
    common = ["--model_name", "R2L", "--config", os.path.join(ROOT, "configs", "lego_noview.txt"), "--datadir", scene,
              "--n_sample_per_ray", "16", "--netwidth", "256", "--netdepth", "88", "--use_residual", "--trial.ON",
              "--trial.body_arch", "resmlp", "--testskip", "1", "--pretrained_ckpt", ckpt, "--render_only", "--n_pose_video", "5",
              # byte equality across rank counts needs ONE kernel family at every launch size: the single process renders the 5
              # video frames in one 20 480-ray launch, a rank of the eight renders one 4096-ray frame.  The exact-fp32 family serves
              # both with the 16-ray cooperative kernels (profiles/r05_dispatch_table.md); the default fp16x2 family would take
              # the one-wave-per-tile kernel for the former and the cooperative one for the latter — equal within rounding
              # (~2e-6 in rgb), not in every byte of a PNG
              "--r2l_precision", "fp32_mfma"]
    cwd = os.getcwd()
    os.chdir(root)
    try:
        one_test = driver.main(common + ["--render_test", "--experiment_name", "one_test8"])
        driver.main(common + ["--experiment_name", "one_video8"])
    finally:
        os.chdir(cwd)
    out = _torchrun(29655, "main.py", common + ["--render_test", "--experiment_name", "eight_test"], root)
    want = "[TEST] TestPSNR %.4f TestPSNRv2 %.4f TestSSIM %.4f" % (one_test["misc"]["test_psnr"].item(),
                                                                  one_test["misc"]["test_psnr_v2"].item(),
                                                                  one_test["misc"]["test_ssim"].item())
    assert want in out, (want, [l for l in out.splitlines() if "[TEST]" in l])
    _torchrun(29657, "main.py", common + ["--experiment_name", "eight_video"], root)
    avis = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(root) for f in fs if f.endswith(".avi") and "_video" in dp)
    assert len(avis) == 2, avis  # one from the single process, one from rank 0 of the eight
    a, b = (open(f, "rb").read() for f in avis)
    assert a == b and len(a) > 1000  # the same five frames in the same order


def test_cli_teacher_render_eight_ranks(scene_and_teacher):
    """The teacher through `main.py --model_name nerf --render_only` (main.py:275-282) over 8 ranks sharing this GPU: the test
    frames (2 views: six ranks hold none) and the 9-pose video come out byte for byte as from one process."""
    from r2l_amd import driver
    root, scene = scene_and_teacher
    common = ["--model_name", "nerf", "--config", os.path.join(ROOT, "configs", "lego.txt"), "--datadir", scene, "--pretrained_ckpt",
              str(root / "teacher.tar"), "--testskip", "1", "--render_only", "--n_pose_video", "9"]
    cwd = os.getcwd()
    os.chdir(root)
    try:
        one_test = driver.main(common + ["--render_test", "--experiment_name", "t1_test"])
        driver.main(common + ["--experiment_name", "t1_vid"])
    finally:
        os.chdir(cwd)
    out = _torchrun(29659, "main.py", common + ["--render_test", "--experiment_name", "t8_test"], root)
    want = "[TEST] TestPSNR %.4f TestPSNRv2 %.4f TestSSIM %.4f" % (one_test["misc"]["test_psnr"].item(),
                                                                  one_test["misc"]["test_psnr_v2"].item(),
                                                                  one_test["misc"]["test_ssim"].item())
    assert want in out, (want, [l for l in out.splitlines() if "[TEST]" in l])
    _torchrun(29661, "main.py", common + ["--experiment_name", "t8_vid"], root)

    def files(tag, ext):
        hits = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(root) for f in fs if f.endswith(ext) and os.sep + tag + "_" in dp)
        return {os.path.basename(f).split("_SERVER")[0] if ext == ".avi" else os.path.basename(f): open(f, "rb").read() for f in hits}
    for tag1, tag8, ext, n in (("t1_test", "t8_test", ".png", 6), ("t1_vid", "t8_vid", ".png", 9), ("t1_vid", "t8_vid", ".avi", 1),
                               ("t1_test", "t8_test", ".avi", 2)):
        a, b = files(tag1, ext), files(tag8, ext)
        assert len(a) == n and a.keys() == b.keys(), (tag1, ext, sorted(a), sorted(b))
        for k in a:
            assert a[k] == b[k] and len(a[k]) > 100, (tag1, ext, k)


def test_bench_eight_ranks_walk(tmp_path):
    """`bench.py --gpus 8` as the driver launches it, with the eight ranks sharing this GPU over gloo (R2L_BENCH_SHARED_GPU_TEST=1):
    every leg an 8-GPU node would time produces a well-formed record — frames sharded over 8 ranks, weak-scaling train legs with
    the bucketed all-reduce (timeline of 4 buckets + the head), the strong-scaling leg, the 12 288-ray leg (= 98 304 / 8, the per-rank
    share of the README step) on the two-tile cooperative chains, the 4096-ray legs, the pose-sharded teacher leg — so that the first contact with a real node has nothing at
    N = 8 that is new code (reference mechanism: main.py:472-479, utils/create_data.py:297-299).  Not a measurement."""
    import json
    env = _env()
    env.pop("R2L_DIST_BACKEND", None)
    env["R2L_BENCH_SHARED_GPU_TEST"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(WORLD), "--master-addr", "127.0.0.1",
           "--master-port", "29663", os.path.join(ROOT, "bench.py"), "--gpus", str(WORLD), "--steps", "2", "--warmup", "1",
           # (eight ranks share ONE GPU's memory here: the default 98 304-ray legs hold 36 GB of stash per rank — 288 GB, the whole
           # HBM, and the walk then spends minutes in allocator retries; a third of the rays walks the same code)
           "--train-rays", "32768"]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and "shared_gpu_test" in out and out["scaling"] == "weak"
    assert out["value"] > 0 and out["config"]["parallelism"].startswith("frames sharded across 8")
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    fast = out["fast_mode"]

    def well_formed(leg, n_rays):
        assert leg["value"] > 0 and leg["ms_per_step"] > 0 and leg["rays_per_step_per_gpu"] == n_rays, leg
        assert np.isfinite(leg["final_loss"]) and "over 8 rank(s)" in leg["workload"]
        rf = leg["roofline"]
        assert rf["grad_allreduce_alone_ms"] > 0 and rf["grad_allreduce_bytes"] == 5917187 * 4 and rf["allreduce_buckets"] == 4
        tl = rf["bucket_timeline_ms"]
        # (4 body / tail buckets + the head's range, in submission order: together the whole flat gradient, each float once)
        assert len(tl) == 5 and sum(b["floats"] for b in tl) == 5917187 and all(b["submit"] >= 0 for b in tl), tl
        assert rf["bucket_timeline_step_ms"] > 0

    for leg, n in ((out["train"], 32768), (out["train_4096"], 4096), (out["train_12288"], 12288), (fast["train"], 32768),
                   (fast["train_strong"], 4096), (fast["train_4096"], 4096), (fast["train_12288"], 12288)):
        well_formed(leg, n)
    assert fast["train_strong"]["scaling"] == "strong" and fast["train_strong"]["global_rays_per_step"] == 32768
    assert "2 tile(s) per workgroup" in fast["train_12288"]["roofline"]["matrix_path"]  # the per-GPU share of the README step at 8 GPUs
    for t in (out["teacher"], fast["teacher"]):
        assert t["value"] > 0 and t["precision"] in ("fp32_mfma", "fp16x2") and t["parallelism"].startswith("poses sharded across 8")
    assert list(out)[-1] == "summary" and out["summary"]["fast_train_strong"][0] > 0 and out["summary"]["graded_train_4096"][0] > 0
