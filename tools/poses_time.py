"""Render throughput: one frame per launch vs K frames per launch (r2l_forward_poses_cfg), 400x400, W256D88.
  python tools/poses_time.py [frames=72]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model
from model.nerf_raybased import PointSampler
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 72
sd = O.make_state_dict(n_block=43, seed=0)
m = build_model(sd, 43)
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6., device="cuda")
poses = torch.stack([torch.from_numpy(O.pose_spherical(5. * k, -30., 4.)[:3, :4]) for k in range(n_frames)], 0).cuda()
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
with torch.no_grad():
    m.render_pose(poses[0], ps); m.render_poses(poses[:2], ps)
    t1 = timed(lambda: [m.render_pose(p, ps) for p in poses])
    print("1 frame per launch : %.3f ms/frame" % (t1 / n_frames))
    for K in (2, 4, 8, 9, 18, 24, 36, 72):
        if n_frames % K: continue
        tk = timed(lambda: [m.render_poses(poses[i:i + K], ps) for i in range(0, n_frames, K)])
        print("%2d frames per launch: %.3f ms/frame (%.1f rounds of 256 workgroups per launch)" % (K, tk / n_frames, K * 1250 / 256.))
