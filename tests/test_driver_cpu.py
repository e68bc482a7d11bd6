"""BASELINE config 0 — `main.py --model_name R2L --render_only --render_test` on CPU (plumbing, no GPU): a tiny
synthetic Blender scene + a checkpoint written by our save_ckpt, driven through the real CLI surface; plus the
hard-ray pool and a 2-rank gloo run of the gradient all-reduce / shard partition logic."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_scene(root, size=12):
    from PIL import Image
    from r2l_amd import data
    rng = np.random.RandomState(0)
    for split, n in (("train", 2), ("val", 1), ("test", 2)):
        os.makedirs(os.path.join(root, split))
        frames = []
        for i in range(n):
            Image.fromarray((rng.rand(size, size, 4) * 255).astype(np.uint8)).save(
                os.path.join(root, split, "r_%d.png" % i))
            frames.append({"file_path": "./%s/r_%d" % (split, i),
                           "transform_matrix": data.pose_spherical(40. * i, -30., 4.).tolist()})
        with open(os.path.join(root, "transforms_%s.json" % split), "w") as f:
            json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, f)


def test_render_only_cpu_plumbing(tmp_path, monkeypatch):
    from r2l_amd import driver
    from r2l_amd.checkpoint import save_ckpt
    from r2l_amd.options import parse_args
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene)
    common = ["--model_name", "R2L", "--config", os.path.join(ROOT, "configs", "lego_noview.txt"), "--datadir", scene,
              "--n_sample_per_ray", "16", "--netwidth", "256", "--netdepth", "6", "--use_residual", "--trial.ON",
              "--trial.body_arch", "resmlp", "--testskip", "1", "--experiment_name", "cpu_plumbing"]
    # a checkpoint in the reference layout (pickled module + state dict), W256 D6
    args = parse_args(common)
    from model.nerf_raybased import NeRF_v3_2
    torch.manual_seed(0)
    net = NeRF_v3_2(args, 1008, 3)
    ck = save_ckpt(str(tmp_path / "ckpt.tar"), 123, net, {"state": {}, "param_groups": []}, 0., 0)
    out = driver.main(common + ["--pretrained_ckpt", ck, "--render_only", "--render_test"])
    rgbs, misc = out["rgbs"], out["misc"]
    assert rgbs.shape == (2, 6, 6, 3)  # half_res of the 12x12 test views
    assert np.isfinite(misc["test_psnr"].item()) and np.isfinite(misc["test_psnr_v2"].item())
    pngs = sorted(f for f in os.listdir(out["logger"].gen_img_path) if f.endswith(".png"))
    assert pngs == ["000.png", "000_error.png", "000_gt.png", "001.png", "001_error.png", "001_gt.png"]
    assert out["video_path"].endswith(".avi")  # the reference writes the video of the test frames too (main.py:1096-1097)
    # the frame equals the oracle's evaluation of the same weights on the same pose
    from r2l_amd import data
    imgs, poses, _, hwf, i_split = data.load_blender_data(scene, True, 1)
    dirs = O.pixel_dirs(6, 6, hwf[2])
    pts = O.sample_test(dirs, O.z_vals(16, 2., 6.), poses[i_split[2][0]][:3, :4])
    ref = O.r2l_forward(net.state_dict(), O.positional_embed(pts, 10)).view(6, 6, 3)
    assert (rgbs[0] - ref).abs().max().item() < 1e-5
    # --render_only without --render_test: the novel-pose video (main.py:1080-1099); Motion-JPEG AVI here (r2l_amd/video.py)
    out = driver.main(common + ["--pretrained_ckpt", ck, "--render_only", "--n_pose_video", "5"])
    assert out["rgbs"].shape == (5, 6, 6, 3)
    assert os.path.basename(out["video_path"]).startswith("video_") and out["video_path"].endswith("_pose5.avi")
    from r2l_amd.video import read_mjpeg_avi
    from r2l_amd.metrics import to8b
    frames, fps = read_mjpeg_avi(out["video_path"])
    assert frames.shape == (5, 6, 6, 3) and fps == 30
    assert np.abs(frames.astype(np.int32) - to8b(out["rgbs"]).astype(np.int32)).mean() < 12  # (JPEG of 6x6 noise-like frames)


def oracle_teacher_frame(csd, fsd, pose, H, W, focal, white_bkgd=True):
    """[H,W,3] of the seeded teacher pair through the oracle's render_rays (perturb 0, det u), rays as render() packs them
    (main.py:141-175): [o, d, near = 2, far = 6, d / |d|]."""
    o, d = O.rays_from_pose(O.pixel_dirs(H, W, focal), torch.as_tensor(np.asarray(pose), dtype=torch.float32)[:3, :4])
    ones = torch.ones_like(d[:, :1])
    rb = torch.cat([o, d, 2. * ones, 6. * ones, d / torch.norm(d, dim=-1, keepdim=True)], -1)
    with torch.no_grad():
        return O.render_rays(rb, csd, fsd, 64, 128, perturb=0., white_bkgd=white_bkgd)["rgb_map"].view(H, W, 3)


def test_teacher_render_only_cpu_plumbing(tmp_path, monkeypatch):
    """README step 2's teacher test command, `main.py --model_name nerf --config configs/lego.txt --pretrained_ckpt <tar>
    --render_only --render_test --testskip 1` (/root/reference/README.md:72; main.py:275-282, 407-453, 1063-1099), on CPU:
    every frame equals the oracle's render_rays of the same weights, metrics are the reference's, TRAINING stays rejected."""
    from r2l_amd import data, driver
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene)
    csd, fsd = O.make_teacher_state_dicts(5, 2, alpha_bias=0.5)
    ck = str(tmp_path / "teacher.tar")
    torch.save({"global_step": 200000, "network_fn_state_dict": csd, "network_fine_state_dict": fsd}, ck)
    common = ["--model_name", "nerf", "--config", os.path.join(ROOT, "configs", "lego.txt"), "--datadir", scene,
              "--pretrained_ckpt", ck, "--testskip", "1", "--experiment_name", "Test__NeRF__cpu"]
    out = driver.main(common + ["--render_only", "--render_test"])
    rgbs, misc = out["rgbs"], out["misc"]
    assert rgbs.shape == (2, 6, 6, 3)
    imgs, poses, _, hwf, i_split = data.load_blender_data(scene, True, 1)
    imgs = torch.as_tensor(imgs)
    gts = imgs[..., :3] * imgs[..., -1:] + (1. - imgs[..., -1:])
    psnrs = []
    for k, i in enumerate(i_split[2]):
        ref = oracle_teacher_frame(csd, fsd, poses[i], 6, 6, float(hwf[2]))
        assert (rgbs[k] - ref).abs().max().item() < 1e-5
        psnrs.append(O.mse2psnr(O.img2mse(ref, gts[i])).item())
    assert abs(misc["test_psnr_v2"].item() - np.mean(psnrs)) < 1e-3
    pngs = sorted(f for f in os.listdir(out["logger"].gen_img_path) if f.endswith(".png"))
    assert pngs == ["000.png", "000_error.png", "000_gt.png", "001.png", "001_error.png", "001_gt.png"]
    assert out["video_path"].endswith(".avi")  # the reference writes the video of the test frames too (main.py:1096-1097)
    # novel-pose video of the teacher
    out = driver.main(common + ["--render_only", "--n_pose_video", "3", "--experiment_name", "Video__NeRF__cpu"])
    assert out["rgbs"].shape == (3, 6, 6, 3) and out["video_path"].endswith("_pose3.avi")
    # --render_factor 2 ("render downsampled for speed", main.py:197-201): H, W, focal halved, the target CROPPED to the frame (:329-333)
    out = driver.main(common + ["--render_only", "--render_test", "--render_factor", "2", "--experiment_name", "Half__NeRF__cpu"])
    assert out["rgbs"].shape == (2, 3, 3, 3)
    ref = oracle_teacher_frame(csd, fsd, poses[i_split[2][0]], 3, 3, float(hwf[2]) / 2)
    assert (out["rgbs"][0] - ref).abs().max().item() < 1e-5
    assert abs(out["misc"]["test_psnr_v2"].item() - np.mean([O.mse2psnr(O.img2mse(oracle_teacher_frame(
        csd, fsd, poses[i], 3, 3, float(hwf[2]) / 2), gts[i][:3, :3])).item() for i in i_split[2]])) < 1e-3
    with pytest.raises(NotImplementedError, match="TRAINING"):
        driver.main(common)


def test_mjpeg_avi_round_trip(tmp_path):
    """RIFF structure of the video writer: header fields, index entries pointing at the frame chunks, frames decodable and
    close to the input at the reference's quality setting (imageio quality=8)."""
    import struct
    from r2l_amd.video import read_mjpeg_avi, write_mjpeg_avi
    yy, xx = np.mgrid[0:48, 0:64]
    frames = np.stack([np.stack([(xx * 3 + 7 * k) % 256, (yy * 4) % 256, (xx + yy + 9 * k) % 256], -1).astype(np.uint8)
                       for k in range(4)])
    path = str(tmp_path / "v.avi")
    n = write_mjpeg_avi(path, frames, fps=30, quality=8)
    data = open(path, "rb").read()
    assert n == len(data) and data[:4] == b"RIFF" and struct.unpack("<I", data[4:8])[0] == len(data) - 8
    assert data[8:12] == b"AVI " and data[12:16] == b"LIST" and data[20:24] == b"hdrl" and data[24:28] == b"avih"
    us, _, _, flags, total, _, streams, _, w, h = struct.unpack("<10I", data[32:72])
    assert (us, flags, total, streams, w, h) == (33333, 0x10, 4, 1, 64, 48)
    movi = data.index(b"movi")
    idx = data.index(b"idx1")
    assert struct.unpack("<I", data[idx + 4:idx + 8])[0] == 16 * 4
    for k in range(4):
        cc, fl, off, size = struct.unpack("<4sIII", data[idx + 8 + 16 * k:idx + 24 + 16 * k])
        assert cc == b"00dc" and fl == 0x10
        assert data[movi + off:movi + off + 4] == b"00dc" and struct.unpack("<I", data[movi + off + 4:movi + off + 8])[0] == size
        assert data[movi + off + 8:movi + off + 10] == b"\xff\xd8"  # a JPEG starts here
    out, fps = read_mjpeg_avi(path)
    assert out.shape == frames.shape and fps == 30
    mse = ((out.astype(np.float64) - frames) ** 2).mean()
    assert 10 * np.log10(255. ** 2 / mse) > 30
    with pytest.raises(ValueError):
        write_mjpeg_avi(path, frames.astype(np.float32))


def test_hard_ray_pool(golden_dir):
    from r2l_amd.driver import HardRayPool
    g = np.load(os.path.join(golden_dir, "hard_rays.npz"))
    rgb, target = torch.from_numpy(g["rgb"]), torch.from_numpy(g["target"])
    o = torch.arange(256 * 3, dtype=torch.float32).view(256, 3)
    d = -o
    pool = HardRayPool(0.2, 2, rng=np.random.RandomState(0))
    assert pool.sizes(256) == (51, 51)
    ro, rd, tg = pool.augment(o, d, target)
    assert ro.shape[0] == 256  # pool not full yet
    pool.update(rgb, o, d, target, 256)
    # the rows that entered are exactly the reference's hard indices (main.py:1411-1414), in the same order
    assert torch.equal(pool.pool[:, :3], o[torch.from_numpy(g["hard_indices"])])
    for _ in range(10):
        pool.update(rgb, o, d, target, 256)
    assert pool.full and pool.pool.shape[0] >= 512
    ro, rd, tg = pool.augment(o, d, target)
    assert ro.shape[0] == 256 + 51 and tg.shape == (307, 3)
    n_before = pool.pool.shape[0]
    pool.update(torch.cat([rgb, rgb[:51]]), ro, rd, tg, 256)
    assert pool.pool.shape[0] == n_before  # replacement, not growth


WORKER = r"""
import os, sys, torch, numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from oracle import r2l_oracle as O
from r2l_amd import data
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
# rank-disjoint shards
files = ["f%%d" %% i for i in range(7)]
mine = data.shard_for_rank(files, rank, world)
gathered = [None] * world
dist.all_gather_object(gathered, mine)
assert sorted(sum(gathered, [])) == files and len(set(sum(gathered, []))) == 7
# data-parallel gradient: each rank back-props its half of the batch, ONE sum all-reduce of the flat buffer, / world
sd = O.make_state_dict(n_block=1, W=32, input_dim=1008, seed=3)
g = torch.Generator().manual_seed(0)
n = 64
o, d, tgt = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g), torch.rand(n, 3, generator=g)
emb = O.positional_embed(O.sample_train(o, d, O.z_vals(16, 2., 6.), 0.), 10)
_, _, full = O.r2l_loss_and_grads(sd, emb, tgt)
sl = slice(rank * n // world, (rank + 1) * n // world)
_, _, part = O.r2l_loss_and_grads(sd, emb[sl], tgt[sl])
flat = torch.cat([part[k].reshape(-1) for k in sd])
from r2l_amd.dist_utils import GradAllReducer   # the trainer's collective (R2LTrainer.allreduce_grads)
red = GradAllReducer(bucket_floats=10000)
assert red.world() == world
red.allreduce(flat)
avg = flat * red.grad_scale()
ref = torch.cat([full[k].reshape(-1) for k in sd])
err = (avg - ref).abs().max().item() / ref.abs().max().item()
assert err < 1e-5, err
# overlapped form: buckets in backward order, submitted one by one, == ONE all-reduce of the whole buffer, bit for bit
from r2l_amd.dist_utils import bucket_plan, sync_parameters, parameters_in_sync, HEAD_FLOATS, LAYER_FLOATS, TAIL_FLOATS
nb = 5
total = HEAD_FLOATS + 2 * nb * LAYER_FLOATS + TAIL_FLOATS
gg = torch.Generator().manual_seed(100 + rank)
mine_g = torch.randn(total, generator=gg)
single = mine_g.clone()
dist.all_reduce(single)
for n_buckets in (1, 2, 4, 9):
    plan = bucket_plan(nb, n_buckets)
    # the plan tiles the flat buffer exactly once, from its end to its start, in whole blocks
    covered = sorted((lo, hi) for _, _, lo, hi in plan)
    assert covered[0][0] == 0 and covered[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert [p[2] for p in plan] == sorted((p[2] for p in plan), reverse=True)
    assert all(l0 %% 2 == 0 and l1 %% 2 == 0 for l0, l1, _, _ in plan)
    assert sorted(l for l0, l1, _, _ in plan for l in range(l0, l1)) == list(range(2 * nb))
    buf = mine_g.clone()
    for _, _, lo, hi in plan:
        red.submit(buf[lo:hi])
    assert red.pending() == len(plan)
    red.finish()
    assert red.pending() == 0 and torch.equal(buf, single), n_buckets
# replicas start identical: each rank builds its own random "parameters", rank 0's win
params = torch.randn(1000, generator=gg)
assert not parameters_in_sync(params)
assert sync_parameters(params)
assert parameters_in_sync(params)
# one experiment folder per job: the logger's ExpID is rank 0's on every rank
import argparse, time
from r2l_amd.logger import Logger
os.chdir(%(tmp)r)
time.sleep(1.1 * rank)  # ranks read the clock in different seconds
lg = Logger(argparse.Namespace(experiment_name="t", experiments_dir="Experiments", debug=False), rank)
ids = [None] * world
dist.all_gather_object(ids, lg.ExpID)
assert len(set(ids)) == 1, ids
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", err)
"""


def test_two_rank_gloo_allreduce_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "tmp": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


def test_cli_render_two_ranks_cpu(tmp_path):
    """`main.py --render_only --render_test` under torchrun with two gloo ranks on CPU (the rank-sharded test-set loop of the student
    AND of the teacher, main.py:189-398 with its `model_name in ['nerf']` branch): frames poses[rank::2], metrics all-reduced, the
    video gathered to rank 0 — the [TEST] line and the PNG / AVI bytes equal the single process's."""
    from r2l_amd import driver
    from r2l_amd.checkpoint import save_ckpt
    from r2l_amd.options import parse_args
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene)
    csd, fsd = O.make_teacher_state_dicts(5, 2, alpha_bias=0.5)
    torch.save({"network_fn_state_dict": csd, "network_fine_state_dict": fsd}, str(tmp_path / "teacher.tar"))
    stu = ["--model_name", "R2L", "--config", os.path.join(ROOT, "configs", "lego_noview.txt"), "--datadir", scene,
           "--n_sample_per_ray", "16", "--netwidth", "256", "--netdepth", "6", "--use_residual", "--trial.ON", "--trial.body_arch",
           "resmlp", "--testskip", "1"]
    from model.nerf_raybased import NeRF_v3_2
    torch.manual_seed(0)
    save_ckpt(str(tmp_path / "student.tar"), 1, NeRF_v3_2(parse_args(stu), 1008, 3), {"state": {}, "param_groups": []}, 0., 0)
    runs = {"student": stu + ["--pretrained_ckpt", str(tmp_path / "student.tar")],
            "teacher": ["--model_name", "nerf", "--config", os.path.join(ROOT, "configs", "lego.txt"), "--datadir", scene,
                        "--pretrained_ckpt", str(tmp_path / "teacher.tar"), "--testskip", "1"]}
    env = {k: v for k, v in os.environ.items() if not k.startswith("R2L_")}
    env.update(MASTER_ADDR="127.0.0.1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    cwd = os.getcwd()
    for k, (name, args) in enumerate(runs.items()):
        args = args + ["--render_only", "--render_test"]
        os.chdir(tmp_path)
        try:
            import unittest.mock as mock
            with mock.patch.object(torch.cuda, "is_available", lambda: False):
                one = driver.main(args + ["--experiment_name", "one_" + name])
        finally:
            os.chdir(cwd)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                            "127.0.0.1", "--master-port", str(29621 + k), os.path.join(ROOT, "main.py")] + args +
                           ["--experiment_name", "two_" + name], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
        out = r.stdout + r.stderr
        assert r.returncode == 0, out[-3000:]
        want = "[TEST] TestPSNR %.4f TestPSNRv2 %.4f TestSSIM %.4f" % (one["misc"]["test_psnr"].item(), one["misc"]["test_psnr_v2"].item(),
                                                                      one["misc"]["test_ssim"].item())
        assert want in out, (want, [l for l in out.splitlines() if "[TEST]" in l])

        def files(tag):
            hits = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs
                          if f.endswith((".png", ".avi")) and os.sep + tag + "_" + name + "_" in dp)
            return {os.path.basename(f): open(f, "rb").read() for f in hits}
        a, b = files("one"), files("two")
        assert len(a) == 8 and a.keys() == b.keys(), (sorted(a), sorted(b))  # 2 frames x (frame, target, error) + video + error video
        for f in a:
            assert a[f] == b[f], (name, f)


WORKER8 = r"""
import ctypes, os, sys, types
sys.path.insert(0, %(root)r)
os.environ["R2L_NO_DW_SLAB"] = "1"  # (134 MB of weight-gradient partials per rank: nothing here launches a kernel)
import torch
import torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 8
torch.set_num_threads(1)
from oracle import r2l_oracle as O            # test infrastructure: stands in for the HIP kernels below
from r2l_amd import _lib
from r2l_amd.train_step import R2LTrainer, lr_schedule
from r2l_amd.dist_utils import split_shards, bucket_plan, parameters_in_sync, HEAD_FLOATS, LAYER_FLOATS, TAIL_FLOATS

NB, RAYS_PER_SHARD, N_RAND = 4, 16, 20
Z = O.z_vals(16, 2., 6.)


class HostEngine:
    # the attribute surface of r2l_amd.engine.R2LEngine that R2LTrainer's HOST code touches; forward = the oracle on CPU tensors
    def __init__(self, sd):
        self.lib = _lib.load()  # the real library: its host-side queries (sizes, layouts) need no GPU
        self.keys = list(sd)
        self.shapes = [tuple(sd[k].shape) for k in self.keys]
        self.n_block = NB
        self.flat = torch.cat([sd[k].reshape(-1) for k in self.keys]).clone()
        self.n_param = self.flat.numel()
        assert self.n_param == self.lib.r2l_param_count(NB) == HEAD_FLOATS + 2 * NB * LAYER_FLOATS + TAIL_FLOATS
        self.device = torch.device("cpu")
        self.cfg = _lib.Config()
        self.dirty = 0

    def sd(self):
        out, off = {}, 0
        for k, shp in zip(self.keys, self.shapes):
            n = 1
            for q in shp:
                n *= q
            out[k] = self.flat[off:off + n].view(shp)
            off += n
        return out

    def set_config(self, **kw):
        for k, v in kw.items():
            setattr(self.cfg, k, int(v))

    def effective_config(self):
        return self.cfg

    def _cfg(self):
        return ctypes.byref(self.cfg)

    def ensure_packed(self, n=None, with_stash=True):
        pass

    def mark_dirty(self):
        self.dirty += 1

    def version(self):
        return self.dirty

    def layout_for(self, n, with_stash=True):
        return 32

    def range_info(self):
        return {}

    def ztab(self, z_vals, perturb):
        return torch.zeros(32)

    def forward_rays(self, o, d, z_vals, perturb=0., t_rand=None, save=None):
        return O.r2l_forward(self.sd(), O.positional_embed(O.sample_train(o, d, Z, 0.), 10))


class HostOnlyTrainer(R2LTrainer):
    # R2LTrainer with its four launches replaced by the oracle: what runs here is the trainer's own host logic — replica sync,
    # ray-share gradient weights, the staged backward's bucket order, submit / finish of the exchange, Adam's 1 / world
    def _stream(self):
        return None

    def _pack_bwd(self, n):
        pass

    def _launch_backward(self, args, parts, lo, hi, what=""):
        o, d, tgt, t_rand, grad_scale = self._step_inputs
        n = o.shape[0]
        if parts & _lib.BWD_CHAIN:
            emb = O.positional_embed(O.sample_train(o, d, Z, 0.), 10)
            loss, _, g = O.r2l_loss_and_grads(self.eng.sd(), emb, tgt)
            # the oracle differentiates mean((rgb - t)^2) over the n rays; the kernels seed dL/drgb = grad_scale * (rgb - t)
            self._g = torch.cat([g[k].reshape(-1) for k in self.eng.keys]) * (grad_scale / (2.0 / (3.0 * n)))
            self._loss = loss
            self.order = []
        total = self.grads.numel()
        if parts & _lib.BWD_TAIL:
            self.grads[total - TAIL_FLOATS:] = self._g[total - TAIL_FLOATS:]
            self.order.append("tail")
        if parts & _lib.BWD_BODY:
            a, b = HEAD_FLOATS + lo * LAYER_FLOATS, HEAD_FLOATS + hi * LAYER_FLOATS
            self.grads[a:b] = self._g[a:b]
            self.order.append((lo, hi))
        if parts & _lib.BWD_HEAD:
            self.grads[:HEAD_FLOATS] = self._g[:HEAD_FLOATS]
            self.order.append("head")

    def _launch_loss_finish(self, n):
        self.loss_out[0] = self._loss

    def _launch_adam(self, lr):
        p, m, v = O.adam_step(self.eng.flat, self.grads * self.reducer.grad_scale(), self.exp_avg, self.exp_avg_sq,
                              self.step_count, lr)
        self.eng.flat.copy_(p); self.exp_avg.copy_(m); self.exp_avg_sq.copy_(v)


# --N_rand 20 over 8 ranks: the README command's shape (reference main.py:802-805: shard files per step)
shares = split_shards(N_RAND, world)
assert shares == [3, 3, 3, 3, 2, 2, 2, 2]
first = sum(shares[:rank])
eng = HostEngine(O.make_state_dict(n_block=NB, seed=100 + rank))  # every rank starts from its OWN random weights ...
tr = HostOnlyTrainer(None, types.SimpleNamespace(z_vals=Z), engine=eng)
assert parameters_in_sync(eng.flat)                              # ... and continues with rank 0's (replica sync)
ref_sd = {k: v.clone() for k, v in O.make_state_dict(n_block=NB, seed=100).items()}
assert torch.equal(eng.flat, torch.cat([ref_sd[k].reshape(-1) for k in eng.keys]))
assert tr.world() == 8 and tr.n_buckets == 4 and eng.cfg.reserve_cus == 8  # CUs left to the collective beside the dW kernels
m = {k: torch.zeros_like(v) for k, v in ref_sd.items()}
v2 = {k: torch.zeros_like(v) for k, v in ref_sd.items()}
for it in range(1, 4):
    g = torch.Generator().manual_seed(1000 + it)  # the step's 20 shard files, the same on every rank
    o = torch.randn(N_RAND * RAYS_PER_SHARD, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])
    d = torch.nn.functional.normalize(torch.randn(N_RAND * RAYS_PER_SHARD, 3, generator=g), dim=-1)
    tgt = torch.rand(N_RAND * RAYS_PER_SHARD, 3, generator=g)
    sl = slice(first * RAYS_PER_SHARD, (first + shares[rank]) * RAYS_PER_SHARD)
    lr = lr_schedule(it, 5e-4, 500, "0.0001,200")
    tr.step(o[sl], d[sl], tgt[sl], lr, perturb=0., n_global=[q * RAYS_PER_SHARD for q in shares])
    # the staged backward completed and handed over its ranges in backward order: tail + last block first, the head last
    assert tr.order == ["tail", (6, 8), (4, 6), (2, 4), (0, 2), "head"], tr.order
    assert tr.reducer.pending() == 0
    # single-process oracle on the FULL batch: one global mean (main.py:1377), one Adam
    emb = O.positional_embed(O.sample_train(o, d, Z, 0.), 10)
    _, _, gr = O.r2l_loss_and_grads(ref_sd, emb, tgt)
    for k in ref_sd:
        ref_sd[k], m[k], v2[k] = O.adam_step(ref_sd[k], gr[k], m[k], v2[k], it, lr)
    # the all-reduced, share-weighted gradient IS the global mean's gradient
    flat_ref = torch.cat([gr[k].reshape(-1) for k in eng.keys])
    gerr = (tr.grads * tr.reducer.grad_scale() - flat_ref).abs().max().item() / flat_ref.abs().max().item()
    assert gerr < 2e-6, (it, gerr)
assert parameters_in_sync(eng.flat)
err = (eng.flat - torch.cat([ref_sd[k].reshape(-1) for k in eng.keys])).abs().max().item()
assert err < 2e-6, err
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok8", err)
"""


def test_eight_rank_gloo_trainer_host_logic(tmp_path):
    """World = 8 without a node: EIGHT gloo processes run the real R2LTrainer — its host logic: replica sync from
    rank 0, --N_rand 20 as 3/3/3/3/2/2/2/2 shard files with ray-share gradient weights, the staged backward's four body buckets +
    head handed to the collective in backward order, one exchange per step, Adam with 1 / world — with the four kernel launches
    of a step replaced by the CPU oracle (a test-side subclass: the product has no CPU path), on a 4-block net; three Adam steps
    end within 2e-6 of the single-process oracle trained on the full batch (reference: main.py:472-479, 1371-1406)."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", "29619", str(script)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok8") == 8


def test_bench_launch_plan_spawns_or_refuses():
    """`python bench.py --gpus N`: bare with N > 1 -> becomes the launcher of N ranks; fewer visible GPUs than asked for,
    or a WORLD_SIZE that contradicts --gpus -> non-zero exit instead of a mislabelled single-GPU number."""
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    assert bench.plan_launch(1, {}, 1) == ("run", 1, 0, 0)
    kind, cmd = bench.plan_launch(4, {}, 8, argv=["--gpus", "4", "--steps", "5"], free_port=29999)
    assert kind == "spawn"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "5"]
    env = {"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2", "LOCAL_WORLD_SIZE": "4"}
    assert bench.plan_launch(4, env, 4) == ("run", 4, 2, 2)
    for gpus, e, vis in ((2, {}, 1), (8, {}, 4), (1, {"WORLD_SIZE": "2"}, 2), (4, {"WORLD_SIZE": "2"}, 4),
                         (4, env, 2), (0, {}, 1)):
        with pytest.raises(SystemExit) as ex:
            bench.plan_launch(gpus, e, vis)
        assert ex.value.code not in (0, None)


def test_bench_gpus2_fails_loudly_without_two_gpus():
    """The real command line on this (GPU-less) box: must exit non-zero and say why, never print a JSON line."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr and "{" not in r.stdout


def test_create_data_rank_index_ranges_never_overlap():
    from r2l_amd.create_data import shard_index_base
    fpf = (100 * 400 * 400) // 4096
    for n_pose, world, chunk in ((301, 3, 100), (10000, 8, 100), (7, 4, 2), (1, 2, 100)):
        fpf_ = (chunk * 400 * 400) // 4096
        ranges = []
        for rank in range(world):
            mine = [i for i in range(1, n_pose + 1) if i % world == rank]
            flushes = (len(mine) + chunk - 1) // chunk
            base = shard_index_base(rank, world, n_pose, chunk, fpf_, n_existing=17)
            ranges.append((base, base + flushes * fpf_))
        assert min(r[0] for r in ranges) == 17  # numbering continues behind the files already there
        ranges.sort()
        assert all(a[1] <= b[0] for a, b in zip(ranges, ranges[1:])), (n_pose, world, chunk, ranges)
    assert shard_index_base(2, 3, 301, 100, fpf) == 2 * 2 * fpf  # the advisor's example: flush counts 1, 2, 1


def test_create_data_numbering_continues_behind_a_gapped_directory():
    """a 3-rank, 301-pose run leaves gaps between the rank ranges; the next run into the kept directory must start
    behind the LARGEST index there, not at the file count (which would overwrite the last rank's shards)."""
    from r2l_amd.create_data import next_free_shard_index, shard_index_base
    fpf = (100 * 400 * 400) // 4096
    names = []
    for rank in range(3):
        mine = [i for i in range(1, 302) if i % 3 == rank]
        base = shard_index_base(rank, 3, 301, 100, fpf)
        names += ["data_%d.npy" % k for k in range(base, base + (len(mine) * 400 * 400) // 4096)]
    used = sorted(int(n[5:-4]) for n in names)
    assert len(used) < used[-1] + 1  # there ARE gaps: counting files would land inside rank 2's range
    nxt = next_free_shard_index(names + ["notes.txt", "data_x.npy"])
    assert nxt == used[-1] + 1
    assert min(shard_index_base(r, 2, 10, 100, fpf, nxt) for r in range(2)) > used[-1]
    assert next_free_shard_index([]) == 0


def test_uneven_n_rand_split_and_pool_schedule():
    """--N_rand that does not divide by the rank count (README: --N_rand 20 on 8 GPUs): floor / ceil shards per rank, and the
    per-rank step size incl. the hard-ray pool is a pure function every rank can evaluate for every other rank."""
    from r2l_amd.dist_utils import split_shards
    from r2l_amd.driver import HardRayPool
    assert split_shards(20, 8) == [3, 3, 3, 3, 2, 2, 2, 2] and sum(split_shards(20, 8)) == 20
    assert split_shards(16, 8) == [2] * 8 and split_shards(7, 3) == [3, 2, 2]
    with pytest.raises(ValueError):
        split_shards(3, 8)
    for batch in (4096 * 3, 4096 * 2, 1000):
        pool = HardRayPool(0.2, 20, rng=np.random.RandomState(0))
        g = torch.Generator().manual_seed(batch)
        for updates_done in range(130):
            o, d, t = (torch.randn(batch, 3, generator=g) for _ in range(3))
            o2, d2, t2 = pool.augment(o, d, t)
            assert o2.shape[0] - batch == pool.extra_rays(batch, updates_done), (batch, updates_done)
            pool.update(torch.rand(o2.shape[0], 3, generator=g), o2, d2, t2, batch)
        assert pool.full and pool.extra_rays(batch, 130) == int(0.2 * batch)
    assert HardRayPool(0.0, 20).extra_rays(4096, 10 ** 6) == 0
