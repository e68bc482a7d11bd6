"""BASELINE config 0 — `main.py --model_name R2L --render_only --render_test` on CPU (plumbing, no GPU): a tiny
synthetic Blender scene + a checkpoint written by our save_ckpt, driven through the real CLI surface; plus the
hard-ray pool and a 2-rank gloo run of the gradient all-reduce / shard partition logic."""
import json
import os
import subprocess
import sys

import numpy as np
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
    pngs = sorted(os.listdir(out["logger"].gen_img_path))
    assert pngs == ["000.png", "000_gt.png", "001.png", "001_gt.png"]
    # the frame equals the oracle's evaluation of the same weights on the same pose
    from r2l_amd import data
    imgs, poses, _, hwf, i_split = data.load_blender_data(scene, True, 1)
    dirs = O.pixel_dirs(6, 6, hwf[2])
    pts = O.sample_test(dirs, O.z_vals(16, 2., 6.), poses[i_split[2][0]][:3, :4])
    ref = O.r2l_forward(net.state_dict(), O.positional_embed(pts, 10)).view(6, 6, 3)
    assert (rgbs[0] - ref).abs().max().item() < 1e-5


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
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", err)
"""


def test_two_rank_gloo_allreduce_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2
