"""End-to-end on the GPU through the real CLI surface: teacher pseudo-data generation (utils/create_data.py 'rand'),
then R2L distillation training on those shards with hard-ray mining, test-set evaluation, checkpoint save + resume."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O
from tests.test_driver_cpu import ROOT, make_scene

from tests.test_forward_gpu import build_model  # noqa: E402

pytestmark = pytest.mark.gpu


def test_create_data_then_train(tmp_path, monkeypatch):
    from r2l_amd import create_data, driver
    from r2l_amd.checkpoint import load_ckpt
    monkeypatch.chdir(tmp_path)
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene, size=128)  # half_res -> 64x64 = 4096 rays per pose = one shard per pose
    csd, fsd = O.make_teacher_state_dicts(5, 2, alpha_bias=0.5)
    torch.save({"network_fn_state_dict": csd, "network_fine_state_dict": fsd}, str(tmp_path / "teacher.tar"))
    kd = str(tmp_path / "pseudo")
    out = create_data.main(["--create_data", "rand", "--config", os.path.join(ROOT, "configs", "lego.txt"), "--datadir",
                            scene, "--teacher_ckpt", str(tmp_path / "teacher.tar"), "--n_pose_kd", "3",
                            "--create_data_chunk", "2", "--datadir_kd", scene + ":" + kd, "--experiment_name", "cd",
                            "--test_teacher", "--testskip", "1"])  # (create_data.py:723-741: the test views first, Loss / PSNR logged)
    files = sorted(os.listdir(kd))
    assert len(files) == 3 and out["n_rays"] == 3 * 4096
    log = open(os.path.join(out["logger"].log_path, "log.txt")).read()
    line = [l for l in log.splitlines() if "Teacher test: Loss" in l]
    assert len(line) == 1
    from r2l_amd import data
    from tests.test_driver_cpu import oracle_teacher_frame
    imgs, poses, _, hwf, i_split = data.load_blender_data(scene, True, 1)
    imgs = torch.as_tensor(imgs)
    gts = imgs[..., :3] * imgs[..., -1:] + (1. - imgs[..., -1:])
    mse = np.mean([O.img2mse(oracle_teacher_frame(csd, fsd, poses[i], 64, 64, float(hwf[2])), gts[i]).item() for i in i_split[2]])
    got = [float(v) for v in line[0].split("Teacher test: Loss ")[1].replace("PSNR", "").split()]
    assert abs(got[0] - mse) < 2e-4 and abs(got[1] + 10. * np.log10(mse)) < 2e-3, (line[0], mse)
    rows = np.load(os.path.join(kd, files[0]))
    assert rows.shape == (4096, 9) and rows.dtype == np.float32
    assert np.all(np.isfinite(rows)) and rows[:, 6:].min() >= -1e-4 and rows[:, 6:].max() <= 1.0 + 1e-4
    assert np.allclose(np.linalg.norm(rows[:, :3], axis=1), 4.0, atol=1e-4)  # origins on the radius-4 sphere

    common = ["--model_name", "R2L", "--config", os.path.join(ROOT, "configs", "lego_noview.txt"), "--datadir", scene,
              "--n_sample_per_ray", "16", "--netwidth", "256", "--netdepth", "6", "--use_residual", "--trial.ON",
              "--trial.body_arch", "resmlp", "--testskip", "1", "--datadir_kd", kd, "--data_mode", "rays", "--N_rand", "2",
              "--hard_ratio", "0.2", "--hard_mul", "2", "--warmup_lr", "0.0001,200", "--i_print", "2", "--i_testset", "4",
              "--i_weights", "6", "--experiment_name", "train"]
    res = driver.main(common + ["--N_iters", "6"])
    wdir = res["logger"].weights_path
    assert sorted(os.listdir(wdir)) == ["ckpt.tar", "ckpt_best.tar"]
    ck = load_ckpt(os.path.join(wdir, "ckpt.tar"))
    assert ck["global_step"] == 6 and ck["optimizer_state_dict"]["state"][0]["exp_avg"].shape == (256, 1008)
    loss = res["trainer"].loss_out[0].item()
    assert np.isfinite(loss) and 0 < loss < 1
    # resume from it and take two more steps; the step counter and Adam moments carry over
    res2 = driver.main(common + ["--N_iters", "8", "--pretrained_ckpt", os.path.join(wdir, "ckpt.tar"), "--resume"])
    assert res2["trainer"].step_count == 8
    # render_only with the trained checkpoint reports PSNR on the 2 test views
    res3 = driver.main(common + ["--pretrained_ckpt", os.path.join(wdir, "ckpt.tar"), "--render_only", "--render_test"])
    assert res3["rgbs"].shape == (2, 64, 64, 3) and np.isfinite(res3["misc"]["test_psnr"].item())


def test_cli_training_two_ranks_uneven_shards(tmp_path):
    """The CLI under torchrun with TWO ranks on this one GPU (gloo: RCCL refuses a device twice) and --N_rand 3: rank 0 takes two
    shard files per step, rank 1 one, gradients weighted by ray share (driver.train, dist_utils.split_shards), hard-ray pools
    per rank, small steps on the staged backward (the default at every world size); the replicas must end bit-identical and rank 0
    writes the checkpoint."""
    import subprocess
    import sys
    from r2l_amd import create_data
    from r2l_amd.checkpoint import load_ckpt
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene, size=128)
    csd, fsd = O.make_teacher_state_dicts(5, 2, alpha_bias=0.5)
    torch.save({"network_fn_state_dict": csd, "network_fine_state_dict": fsd}, str(tmp_path / "teacher.tar"))
    kd = str(tmp_path / "pseudo")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        create_data.main(["--create_data", "rand", "--config", os.path.join(ROOT, "configs", "lego.txt"), "--datadir", scene,
                          "--teacher_ckpt", str(tmp_path / "teacher.tar"), "--n_pose_kd", "6", "--create_data_chunk", "3",
                          "--datadir_kd", scene + ":" + kd, "--experiment_name", "cd"])
    finally:
        os.chdir(cwd)
    assert len(os.listdir(kd)) == 6
    env = {k: v for k, v in os.environ.items() if not k.startswith("R2L_")}
    env.update(MASTER_ADDR="127.0.0.1", R2L_DIST_BACKEND="gloo", R2L_CHECK_SYNC="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(ROOT, "main.py"), "--model_name", "R2L", "--config",
           os.path.join(ROOT, "configs", "lego_noview.txt"), "--datadir", scene, "--n_sample_per_ray", "16", "--netwidth", "256",
           "--netdepth", "6", "--use_residual", "--trial.ON", "--trial.body_arch", "resmlp", "--testskip", "1", "--datadir_kd", kd,
           "--data_mode", "rays", "--N_rand", "3", "--hard_ratio", "0.2", "--hard_mul", "2", "--warmup_lr", "0.0001,200",
           "--i_print", "2", "--i_testset", "100", "--i_weights", "8", "--N_iters", "8", "--experiment_name", "dp2"]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "[2, 1] shard files per rank and step" in out and "replicas in sync after 8 iterations: True (skipped steps: 0)" in out
    ckpts = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f == "ckpt.tar"]
    assert len(ckpts) == 1 and load_ckpt(ckpts[0])["global_step"] == 8


def test_cli_render_two_ranks(tmp_path):
    """`main.py --render_only` with two ranks on this GPU (gloo): the test views (--render_test: metrics all-reduced, every rank
    writes its own frames) and the novel-pose video (5 poses over 2 ranks: frames gathered to rank 0 and re-interleaved; with 3
    ranks' worth of poses one rank may hold none — ADVICE r2) come out as from one process."""
    import subprocess
    import sys
    from r2l_amd import driver
    from r2l_amd.checkpoint import save_ckpt
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene, size=128)
    sd = O.make_state_dict(n_block=2, seed=1)
    ckpt = str(tmp_path / "SERVER-20260101-000000_iter7" / "weights" / "ckpt.tar")
    save_ckpt(ckpt, 7, build_model(sd, 2).cpu(), {"state": {}, "param_groups": []}, 0., 0)
    common = ["--model_name", "R2L", "--config", os.path.join(ROOT, "configs", "lego_noview.txt"), "--datadir", scene,
              "--n_sample_per_ray", "16", "--netwidth", "256", "--netdepth", "6", "--use_residual", "--trial.ON",
              "--trial.body_arch", "resmlp", "--testskip", "1", "--pretrained_ckpt", ckpt, "--render_only", "--n_pose_video", "5"]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        one_test = driver.main(common + ["--render_test", "--experiment_name", "one_test"])
        one_video = driver.main(common + ["--experiment_name", "one_video"])
    finally:
        os.chdir(cwd)
    env = {k: v for k, v in os.environ.items() if not k.startswith("R2L_")}
    env.update(MASTER_ADDR="127.0.0.1", R2L_DIST_BACKEND="gloo")
    run = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29645", os.path.join(ROOT, "main.py")] + common
    r = subprocess.run(run + ["--render_test", "--experiment_name", "two_test"], env=env, cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    want = "[TEST] TestPSNR %.4f TestPSNRv2 %.4f TestSSIM %.4f" % (one_test["misc"]["test_psnr"].item(),
                                                                  one_test["misc"]["test_psnr_v2"].item(),
                                                                  one_test["misc"]["test_ssim"].item())
    assert want in out, (want, [l for l in out.splitlines() if "[TEST]" in l])
    r = subprocess.run(run + ["--experiment_name", "two_video"], env=env, cwd=str(tmp_path), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    # (the --render_test runs write a video of the test frames as well, main.py:1096-1097: not the ones compared here)
    avis = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f.endswith(".avi") and "_video_" in dp)
    assert len(avis) == 2  # one from the single process, one from rank 0 of the pair
    a, b = (open(f, "rb").read() for f in avis)
    assert a == b and len(a) > 1000  # the same five frames in the same order


def test_cli_teacher_render_test_vs_oracle(tmp_path):
    """README step 2's teacher test command (`main.py --model_name nerf --config configs/lego.txt --pretrained_ckpt <tar>
    --render_only --render_test --testskip 1`, /root/reference/README.md:72; main.py:275-282 `model_name in ['nerf']` branch of
    render_path, 107-186 render, 407-453 create_nerf) on the teacher kernels: every test frame against the oracle's render_rays
    of the same seeded pair (rgb < 1e-4, north_star's bar), PSNR / SSIM against the oracle's metrics of the oracle's frames,
    for each arithmetic the CLI can select; plus the novel-pose video of the teacher."""
    from r2l_amd import data, driver
    from tests.test_driver_cpu import oracle_teacher_frame
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene, size=64)  # half_res -> 32 x 32 frames, 2 test views
    csd, fsd = O.make_teacher_state_dicts(5, 2, alpha_bias=0.5)
    ck = str(tmp_path / "NeRF__lego_SERVER000-20260101-000000" / "weights" / "200000.tar")
    os.makedirs(os.path.dirname(ck))
    torch.save({"global_step": 200000, "network_fn_state_dict": csd, "network_fine_state_dict": fsd}, ck)
    common = ["--model_name", "nerf", "--config", os.path.join(ROOT, "configs", "lego.txt"), "--datadir", scene,
              "--pretrained_ckpt", ck, "--testskip", "1", "--render_only"]
    imgs, poses, _, hwf, i_split = data.load_blender_data(scene, True, 1)
    imgs = torch.as_tensor(imgs)
    gts = imgs[..., :3] * imgs[..., -1:] + (1. - imgs[..., -1:])
    refs = [oracle_teacher_frame(csd, fsd, poses[i], 32, 32, float(hwf[2])) for i in i_split[2]]
    want_psnr = np.mean([O.mse2psnr(O.img2mse(r, gts[i])).item() for r, i in zip(refs, i_split[2])])
    want_ssim = np.mean([O.ssim(r, gts[i]).item() for r, i in zip(refs, i_split[2])])
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        for prec in ("auto", "fp32_mfma", "bf16x3"):
            out = driver.main(common + ["--render_test", "--experiment_name", "Test__NeRF__" + prec, "--r2l_precision", prec])
            rgbs, misc = out["rgbs"].cpu(), out["misc"]
            assert rgbs.shape == (2, 32, 32, 3)
            for k in range(2):
                assert (rgbs[k] - refs[k]).abs().max().item() < 1e-4, (prec, k)
            assert abs(misc["test_psnr_v2"].item() - want_psnr) < 1e-3 and abs(misc["test_ssim"].item() - want_ssim) < 1e-4
            pngs = sorted(f for f in os.listdir(out["logger"].gen_img_path) if f.endswith(".png"))
            assert pngs == ["000.png", "000_error.png", "000_gt.png", "001.png", "001_error.png", "001_gt.png"]
            assert "_SERVER000-20260101-000000_iter200000_" in os.path.basename(out["video_path"])
        out = driver.main(common + ["--n_pose_video", "3", "--experiment_name", "Video__NeRF"])
        assert out["rgbs"].shape == (3, 32, 32, 3) and out["video_path"].endswith("_pose3.avi")
        with pytest.raises(NotImplementedError, match="TRAINING"):
            driver.main([a for a in common if a != "--render_only"])
    finally:
        os.chdir(cwd)


def test_cli_arithmetic_is_a_recorded_option(tmp_path):
    """--r2l_precision / --r2l_dw_mode (r2l_amd/options.py -> engine.set_config): `main.py ... --r2l_precision
    fp32_mfma --render_only --render_test` gives the frames of engine.set_config(precision='fp32_mfma') bit for bit (and not the
    default family's); a training run logs the effective r2l_config and stores it in the checkpoint it writes."""
    from r2l_amd import data, driver
    from r2l_amd.checkpoint import load_ckpt, save_ckpt
    from r2l_amd.nerf_raybased import PointSampler
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene, size=128)
    sd = O.make_state_dict(n_block=2, seed=1)
    ckpt = str(tmp_path / "ckpt.tar")
    save_ckpt(ckpt, 7, build_model(sd, 2).cpu(), {"state": {}, "param_groups": []}, 0., 0)
    common = ["--model_name", "R2L", "--config", os.path.join(ROOT, "configs", "lego_noview.txt"), "--datadir", scene,
              "--n_sample_per_ray", "16", "--netwidth", "256", "--netdepth", "6", "--use_residual", "--trial.ON",
              "--trial.body_arch", "resmlp", "--testskip", "1"]
    render = common + ["--pretrained_ckpt", ckpt, "--render_only", "--render_test"]
    _, poses, _, hwf, i_split = data.load_blender_data(scene, True, 1)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        frames = {}
        for prec in ("fp32_mfma", "bf16x3", "auto"):
            out = driver.main(render + ["--r2l_precision", prec, "--experiment_name", "arith_" + prec])
            assert out["r2l_config"]["precision"] == ("fp16x2" if prec == "auto" else prec)
            frames[prec] = out["rgbs"]
            net = build_model(sd, 2)
            net.engine().set_config(precision=prec)
            ps = PointSampler(64, 64, float(hwf[2]), 16, 2., 6., device="cuda")
            c2ws = torch.stack([torch.as_tensor(poses[i], dtype=torch.float32)[:3, :4] for i in i_split[2]], 0)
            with torch.no_grad():
                want = net.render_poses(c2ws, ps).view(len(i_split[2]), 64, 64, 3)
            assert torch.equal(out["rgbs"], want), prec
            log = open(os.path.join(out["logger"].log_path, "log.txt")).read()
            assert "r2l_config: precision %s" % out["r2l_config"]["precision"] in log
        assert not torch.equal(frames["fp32_mfma"], frames["auto"])  # (different arithmetic: the option is not a no-op)
        # training: the record goes into the checkpoint
        rng = np.random.RandomState(0)
        kd = str(tmp_path / "kd")
        os.makedirs(kd)
        for k in range(4):
            rows = np.concatenate([rng.randn(4096, 3) * 0.3 + [0, 0, 4], rng.randn(4096, 3), rng.rand(4096, 3)], 1).astype(np.float32)
            np.save(os.path.join(kd, "data_%d.npy" % k), rows)
        out = driver.main(common + ["--datadir_kd", kd, "--data_mode", "rays", "--N_rand", "1", "--N_iters", "3", "--i_weights", "3",
                                    "--i_print", "1", "--i_testset", "100", "--r2l_precision", "fp16x2", "--r2l_dw_mode", "exact",
                                    "--experiment_name", "arith_train"])
        ck = load_ckpt(os.path.join(out["logger"].weights_path, "ckpt.tar"))
        assert ck["r2l_config"] == {"precision": "fp16x2", "dw_mode": "exact", "requested": {"precision": "fp16x2", "dw_mode": "exact"}}
        assert out["trainer"].eng.effective_config().dw_mode == 2
        # resuming / rendering from it says what trained it
        out = driver.main(common + ["--pretrained_ckpt", os.path.join(out["logger"].weights_path, "ckpt.tar"), "--render_only",
                                    "--render_test", "--experiment_name", "arith_back"])
        assert "checkpoint was trained with r2l_config: precision fp16x2, dw_mode exact" in open(
            os.path.join(out["logger"].log_path, "log.txt")).read()
    finally:
        os.chdir(cwd)


def test_create_data_two_ranks_then_continue(tmp_path):
    """utils/create_data.py under torchrun with two ranks on this GPU (gloo): poses shard i % world, rank-disjoint shard index
    ranges with gaps between them; a second (single-process) run into the kept directory continues behind the LARGEST index
    (counting files would overwrite the last rank's shards)."""
    import subprocess
    import sys
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene, size=128)  # half_res -> 64x64 = 4096 rays per pose = one shard per pose
    csd, fsd = O.make_teacher_state_dicts(5, 2, alpha_bias=0.5)
    torch.save({"network_fn_state_dict": csd, "network_fine_state_dict": fsd}, str(tmp_path / "teacher.tar"))
    kd = str(tmp_path / "pseudo")
    env = {k: v for k, v in os.environ.items() if not k.startswith("R2L_")}
    env.update(MASTER_ADDR="127.0.0.1", R2L_DIST_BACKEND="gloo")
    args = ["--create_data", "rand", "--config", os.path.join(ROOT, "configs", "lego.txt"), "--datadir", scene, "--teacher_ckpt",
            str(tmp_path / "teacher.tar"), "--create_data_chunk", "2", "--datadir_kd", scene + ":" + kd, "--experiment_name", "cd"]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29643", os.path.join(ROOT, "utils", "create_data.py")] + args +
                       ["--n_pose_kd", "5"], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    idx = sorted(int(f[5:-4]) for f in os.listdir(kd))
    # rank 0: poses 2, 4 -> data_0, data_1; rank 1: poses 1, 3, 5 -> behind the range sized for the larger share (2 flushes x 2)
    assert idx == [0, 1, 4, 5, 6], idx
    first = {i: np.load(os.path.join(kd, "data_%d.npy" % i)) for i in idx}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "utils", "create_data.py")] + args + ["--n_pose_kd", "2"], env=env,
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    idx2 = sorted(int(f[5:-4]) for f in os.listdir(kd))
    assert idx2 == [0, 1, 4, 5, 6, 7, 8], idx2
    for i in idx:  # nothing of the first run was overwritten
        assert np.array_equal(first[i], np.load(os.path.join(kd, "data_%d.npy" % i)))


@pytest.fixture(scope="module")
def trained_student():
    """A W256 D88 student distilled for 1000 fused steps from a seeded teacher (no released checkpoint exists offline):
    weights that have LEFT the default-init distribution, for parity checks on a trained-weight distribution."""
    from model.nerf_raybased import NeRF, NeRF_v3_2, PointSampler
    from r2l_amd.options import parse_args
    from r2l_amd.render import get_rays, render
    from r2l_amd.train_step import R2LTrainer, lr_schedule
    from r2l_amd import data
    for k in ("R2L_FORCE_VARIANT", "R2L_NO_FWD3", "R2L_NO_FWD2", "R2L_NO_BWD2", "R2L_NO_DW2"):
        assert k not in os.environ  # the student is trained on the default kernels
    nets = []
    for sd in O.make_teacher_state_dicts(21, 2, alpha_bias=0.5):
        m = NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
        m.load_state_dict(sd)
        nets.append(m.cuda())
    H = W = 64
    focal = 80.
    kw = dict(network_fn=nets[0], network_query_fn=None, N_samples=64, N_importance=128, network_fine=nets[1],
              white_bkgd=True, perturb=0., ndc=False, near=2., far=6., use_viewdirs=True)
    rng = np.random.RandomState(0)
    rows = []
    with torch.no_grad():
        for _ in range(5):
            pose = data.get_rand_pose(rng).cuda()
            ro, rd = get_rays(H, W, focal, pose[:3, :4])
            rgb, *_ = render(H, W, focal, chunk=1 << 15, rays=torch.stack([ro, rd], 0), **kw)
            rows.append(torch.cat([ro.reshape(-1, 3), rd.reshape(-1, 3), rgb.reshape(-1, 3)], -1))
    train, held = torch.cat(rows[:4], 0), rows[4]
    args = parse_args(["--netdepth", "88", "--netwidth", "256", "--use_residual", "--trial.ON", "--trial.body_arch",
                       "resmlp", "--n_sample_per_ray", "16"])
    torch.manual_seed(0)
    net = NeRF_v3_2(args, 1008, 3).cuda()
    init = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ps = PointSampler(H, W, focal, 16, 2., 6.)
    tr = R2LTrainer(net, ps)

    def held_psnr():
        with torch.no_grad():
            out = net.forward_rays(held[:, :3], held[:, 3:6], ps)
        return -10 * np.log10(((out - held[:, 6:])**2).mean().item())

    p0 = held_psnr()
    losses, psnr150 = [], None
    g = torch.Generator(device="cuda").manual_seed(0)
    for it in range(1, 1001):
        idx = torch.randint(0, train.shape[0], (8192,), device="cuda", generator=g)
        b = train[idx]
        _, lo = tr.step(b[:, :3], b[:, 3:6], b[:, 6:], lr_schedule(it, 5e-4, 500, "0.0001,200"), perturb=1.)
        losses.append(lo[0].clone())  # loss_out is ONE device buffer the trainer rewrites every step
        if it == 150:
            psnr150 = held_psnr()
    losses = torch.stack(losses).cpu().numpy()
    return {"net": net, "ps": ps, "held": held, "train": train, "losses": losses, "psnr": (p0, psnr150, held_psnr()),
            "init": init}


def test_distillation_converges_on_teacher_data(trained_student):
    """Functional check of the whole loop at the README architecture (W256 D88): a student trained for 150 fused steps
    on rays rendered by a (seeded) teacher reduces its loss by > 3x and its held-out PSNR rises."""
    losses, (p0, p150, p1000) = trained_student["losses"], trained_student["psnr"]
    print("loss %.4f -> %.4f (150 steps) -> %.4f (1000), held-out psnr %.2f -> %.2f -> %.2f dB" %
          (losses[0], np.mean(losses[140:150]), np.mean(losses[-10:]), p0, p150, p1000))
    assert np.isfinite(losses).all()
    assert np.mean(losses[140:150]) < losses[0] / 3
    assert p150 > p0 + 3 and p1000 >= p150 - 0.5


# kernel families through r2l_config (tests/conftest.py use_family)
VARIANTS = {"main-fp16x2": dict(tiling="main"), "coopf-fp16x2": dict(tiling="coopf"),
            "main-bf16x3": dict(tiling="main", precision="bf16x3"),
            "main-f32mfma": dict(tiling="main", precision="fp32_mfma"),
            "coop16": dict(tiling="coop16")}


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_trained_weights_parity_vs_oracle(trained_student, variant, monkeypatch):
    """RGB within 1e-4 and PSNR within 0.01 dB of the fp32 CPU oracle on TRAINED weights (1000 Adam steps away from the
    nn.Linear init every other parity test uses), under every forward kernel family — the fp16x2 default included, whose
    operand range guard (|x| < 32768) and two-way splits were only ever exercised on |w| <= 1/16 before."""
    from tests.conftest import use_family
    use_family(monkeypatch, **VARIANTS[variant])
    net, ps, held = trained_student["net"], trained_student["ps"], trained_student["held"]
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    moved = max((sd[k] - trained_student["init"][k]).abs().max().item() for k in sd)
    assert moved > 0.02, moved  # the weights really moved (default init is |w| <= 1/16)
    o, d, tgt = held[:, :3], held[:, 3:6], held[:, 6:]
    with torch.no_grad():
        rgb = net.forward_rays(o, d, ps).cpu()
    ref = O.r2l_forward(sd, O.positional_embed(O.sample_train(o.cpu(), d.cpu(), O.z_vals(16, 2., 6.), 0.), 10))
    err = (rgb - ref).abs().max().item()
    psnr_hip = -10 * np.log10(((rgb - tgt.cpu())**2).mean().item())
    psnr_ref = -10 * np.log10(((ref - tgt.cpu())**2).mean().item())
    print("%s: max |rgb - oracle| = %.2e, psnr %.4f vs %.4f dB, max |w - w_init| = %.3f" % (variant, err, psnr_hip,
                                                                                             psnr_ref, moved))
    assert err < 1e-4 and abs(psnr_hip - psnr_ref) < 0.01


def test_trained_weights_gradient_parity_vs_oracle(trained_student, monkeypatch):
    """Gradients on the TRAINED weights, every tensor of the W256 D88 net, against fp64 autograd of the oracle ("truth").
    An 88-layer ReLU net's gradient is discontinuous in the forward's last bits: pre-activations within ~1e-6 of zero (257
    of 45 M on these inputs) flip their mask under ANY change of summation order, and every flip moves the gradient of all
    layers below it discretely.  Measured (tools/diag_grad.py, profiles/r02_grad_noise.txt): the kernel families differ from
    each other and from the truth by 2e-4 .. 9e-4 (median per-tensor relative L2) on these weights — the exact-fp32 MFMA
    kernels included — so that is the resolution any fp32 implementation can be held to.  The bars: the default fp16 trio
    (fp16-rounded operands in the weight-gradient GEMMs) (a) stays inside that band (< 1e-3; every family < 2e-3) and (b) is as
    far from the truth as its exact-dW variant (the chain's mask flips, not the dW operands); the tail (above every mask) agrees
    to 1e-4."""
    from r2l_amd.train_step import R2LTrainer
    net, ps, train = trained_student["net"], trained_student["ps"], trained_student["train"]
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    b = train[:4096].contiguous()
    b[:, 6:] = torch.rand(4096, 3, generator=g).cuda()  # random targets: residuals O(1), every ray contributes
    emb = O.positional_embed(O.sample_train(b[:, :3].cpu(), b[:, 3:6].cpu(), O.z_vals(16, 2., 6.), 0.), 10)
    loss, _, _ = O.r2l_loss_and_grads(sd, emb, b[:, 6:].cpu())
    _, _, g64 = O.r2l_loss_and_grads({k: v.double() for k, v in sd.items()}, emb.double(), b[:, 6:].cpu().double())
    med, worst, tail = {}, {}, {}
    from tests.conftest import use_family
    for fam, cfg in (("fp16 trio", {}), ("fp16 trio, exact dW", dict(dw_mode="exact")), ("bf16x3 trio", dict(precision="bf16x3")),
                     ("fp32 mfma", dict(precision="fp32_mfma"))):
        with monkeypatch.context() as mp:
            use_family(mp, tiling="main", **cfg)
            tr = R2LTrainer(net, ps)
            tr.forward_backward(b[:, :3].contiguous(), b[:, 3:6].contiguous(), b[:, 6:].contiguous())
            assert abs(tr.loss_out[0].item() - loss.item()) < 1e-6
            off, flat, l2, mx = 0, tr.grads.cpu().double(), {}, {}
            for k, v in sd.items():
                gk = flat[off:off + v.numel()].view(v.shape)
                off += v.numel()
                l2[k] = ((gk - g64[k]).norm() / g64[k].norm()).item()
                mx[k] = ((gk - g64[k]).abs().max() / g64[k].abs().max()).item()
            body = [k for k in sd if ".body." in k]
            med[fam], worst[fam] = float(np.median([l2[k] for k in body])), max(mx.values())
            tail[fam] = max(l2[k] for k in sd if k.startswith("tail"))
            print("%s: body tensors median rel L2 %.2e, worst tensor max-norm %.2e, tail rel L2 %.2e" %
                  (fam, med[fam], worst[fam], tail[fam]))
    for fam in med:
        assert med[fam] < 2e-3 and worst[fam] < 5e-2 and tail[fam] < 1e-4, fam
    # (b) the default trio's distance to the truth is inside the band the families span among themselves on trained weights
    # (2e-4 .. 9e-4: which family lands low is mask-flip luck — 1.4e-4 / 1.8e-4 for the two fp32-exact families in one run of round
    # 4, 4.0e-4 / 2.6e-4 in round 2 — so a bar RELATIVE to them is a coin toss), and it is the CHAIN's, not the fp16 operands of the
    # weight-gradient GEMMs': the exact-dW variant of the same chains sits at the same distance
    assert med["fp16 trio"] < 1.0e-3
    assert abs(med["fp16 trio"] - med["fp16 trio, exact dW"]) < 0.25 * med["fp16 trio, exact dW"]
    # exact-dW mode (hi + mid operands, three products in the weight-gradient GEMMs).  Measured (profiles/r03_summary.md):
    # 4.51e-4 against 4.55e-4 for the default trio on these weights — on TRAINED weights the distance to the fp64 truth is the
    # chain's (mask flips of near-zero pre-activations under the fp16x2 forward / dX chain), not the rounding of the
    # weight-gradient operands, so exact dW cannot pull the trio to the fp32 families' figure; what it buys shows where the
    # masks are stable: the strict 2e-5 Adam bar of tests/test_train_gpu.py::test_three_adam_steps_vs_oracle.
    assert med["fp16 trio, exact dW"] < 1.1 * med["fp16 trio"] and med["fp16 trio, exact dW"] < 1.0e-3


def test_hard_ray_pool_on_gpu(golden_dir):
    """HardRayPool on cuda tensors (main.py:1325-1347 augment, :1410-1425 sort / top-k / append / replace): the rows
    that enter are the reference's hard indices in the reference's order; append until B*hard_mul rows, then the rows
    handed out by augment() are exactly the ones the next update() replaces."""
    from r2l_amd.driver import HardRayPool
    g = np.load(os.path.join(golden_dir, "hard_rays.npz"))
    rgb, target = torch.from_numpy(g["rgb"]).cuda(), torch.from_numpy(g["target"]).cuda()
    o = torch.arange(256 * 3, dtype=torch.float32, device="cuda").view(256, 3)
    d = -o
    pool = HardRayPool(0.2, 2, seed=3)
    assert pool.sizes(256) == (51, 51)
    ro, _, _ = pool.augment(o, d, target)
    assert ro.shape[0] == 256 and ro.data_ptr() == o.data_ptr()  # pool not full yet: batch untouched
    pool.update(rgb, o, d, target, 256)
    hard = torch.from_numpy(g["hard_indices"]).cuda()
    assert pool.pool.is_cuda and torch.equal(pool.pool[:, :3], o[hard]) and torch.equal(pool.pool[:, 3:6], d[hard])
    assert torch.equal(pool.pool[:, 6:], target[hard])
    n_updates = 1
    while not pool.full:
        pool.update(rgb, o, d, target, 256)
        n_updates += 1
    assert n_updates == 11 and pool.pool.shape[0] == 11 * 51  # first multiple of 51 that reaches 256 * 2
    # full pool: augment appends 51 pool rows; update replaces exactly those rows with the new hard rays
    ro, rd, tg = pool.augment(o, d, target)
    assert ro.shape == (307, 3) and tg.shape == (307, 3)
    ix = pool._ix_out.clone()
    assert ix.numel() == 51 and ix.unique().numel() == 51 and int(ix.max()) < pool.pool.shape[0]
    assert torch.equal(ro[256:], pool.pool[ix, :3])
    before = pool.pool.clone()
    rgb2 = torch.rand(307, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    pool.update(rgb2, ro, rd, tg, 256)
    assert pool.pool.shape == before.shape  # replacement, not growth
    err = ((rgb2[:256] - tg[:256])**2).mean(1)
    new_hard = torch.sort(err)[1][-51:]
    assert torch.equal(pool.pool[ix, :3], ro[new_hard]) and torch.equal(pool.pool[ix, 6:], tg[new_hard])
    keep = torch.ones(before.shape[0], dtype=torch.bool, device="cuda")
    keep[ix] = False
    assert torch.equal(pool.pool[keep], before[keep])
    # two pools with the same seed draw the same rows; another seed draws others
    a, b, c = HardRayPool(0.2, 2, seed=9), HardRayPool(0.2, 2, seed=9), HardRayPool(0.2, 2, seed=10)
    for p in (a, b, c):
        while not p.full:
            p.update(rgb, o, d, target, 256)
        p.augment(o, d, target)
    assert torch.equal(a._ix_out, b._ix_out) and not torch.equal(a._ix_out, c._ix_out)


def test_pool_kernels_vs_torch_ops():
    """include/r2l_hip.h r2l_pool_pick / _augment / _store (the hard-ray pool's row choice and data movement, main.py:1325-1347,
    1410-1425): the picked rows are distinct, in range, reproducible per key, different per key and cover the pool evenly; the
    augmented batch and the stored rows are bit for bit what the torch ops of the CPU path give — on column slices of a [B,9]
    shard batch (row stride 9) as the training loop passes them."""
    import ctypes
    from r2l_amd import _lib
    from r2l_amd.driver import HardRayPool
    lib = _lib.load()
    p = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n_rows, n_out in ((1638400, 16384), (561, 51), (5, 5), (1, 1), (4097, 4097)):
        a = torch.empty(n_out, dtype=torch.int64, device="cuda")
        b, c = torch.empty_like(a), torch.empty_like(a)
        _lib.check(lib.r2l_pool_pick(p(a), n_out, n_rows, 12345, st), "pick")
        _lib.check(lib.r2l_pool_pick(p(b), n_out, n_rows, 12345, st), "pick")
        _lib.check(lib.r2l_pool_pick(p(c), n_out, n_rows, 12346, st), "pick")
        assert int(a.min()) >= 0 and int(a.max()) < n_rows and a.unique().numel() == n_out
        assert torch.equal(a, b) and (n_rows < 10 or not torch.equal(a, c))
        if n_out == n_rows:
            assert torch.equal(a.sort()[0], torch.arange(n_rows, device="cuda"))  # a permutation
    # every row equally likely: 200 draws of 16 384 of 1.6 M rows -> ~2 hits per row on average over 32 coarse bins
    hits = torch.zeros(32, device="cuda")
    ix = torch.empty(16384, dtype=torch.int64, device="cuda")
    for k in range(200):
        _lib.check(lib.r2l_pool_pick(p(ix), 16384, 1638400, 777 + k, st), "pick")
        hits += torch.bincount(ix // 51200, minlength=32).float()
    assert (hits / hits.mean() - 1).abs().max().item() < 0.02, hits
    assert lib.r2l_pool_pick(p(ix), 10, 5, 1, st) != 0  # more rows asked for than there are
    # augment / store against the torch ops
    g = torch.Generator(device="cuda").manual_seed(0)
    B, n_out, rows = 4096 * 3, 2457, 49140
    batch = torch.rand(B, 9, device="cuda", generator=g)
    pool = torch.rand(rows, 9, device="cuda", generator=g)
    o, d, t = batch[:, :3], batch[:, 3:6], batch[:, 6:]
    idx = torch.randperm(rows, device="cuda", generator=g)[:n_out]
    out = torch.empty(3, B + n_out, 3, device="cuda")
    _lib.check(lib.r2l_pool_augment(p(o), p(d), p(t), 9, 9, 9, p(pool), p(idx), B, n_out, p(out[0]), p(out[1]), p(out[2]), st), "augment")
    assert torch.equal(out[0], torch.cat([o, pool[idx, :3]])) and torch.equal(out[1], torch.cat([d, pool[idx, 3:6]]))
    assert torch.equal(out[2], torch.cat([t, pool[idx, 6:]]))
    hard = torch.randperm(B, device="cuda", generator=g)[:n_out]
    want = pool.clone()
    want[idx] = torch.cat([o[hard], d[hard], t[hard]], -1)
    _lib.check(lib.r2l_pool_store(p(o), p(d), p(t), 9, 9, 9, p(hard), p(pool), p(idx), 0, n_out, st), "store")
    assert torch.equal(pool, want)
    want[100:100 + n_out] = torch.cat([o[hard], d[hard], t[hard]], -1)
    _lib.check(lib.r2l_pool_store(p(o), p(d), p(t), 9, 9, 9, p(hard), p(pool), None, 100, n_out, st), "store append")
    assert torch.equal(pool, want)
    # the pool object on strided inputs == on contiguous copies
    pa, pb = HardRayPool(0.2, 2, seed=4), HardRayPool(0.2, 2, seed=4)
    rgb = torch.rand(B, 3, device="cuda", generator=g)
    for _ in range(12):
        xa = pa.augment(o, d, t)
        xb = pb.augment(o.contiguous(), d.contiguous(), t.contiguous())
        for u, v in zip(xa, xb):
            assert torch.equal(u, v)
        r2 = torch.rand(xa[0].shape[0], 3, device="cuda", generator=g)
        pa.update(r2, *xa, B)
        pb.update(r2, *xb, B)
    assert pa.full and torch.equal(pa.pool, pb.pool)
