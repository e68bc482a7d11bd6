"""fp16x2 forward (R2L_FWD2=1) against the default bf16x3 forward and the CPU oracle: max |dRGB| and frame time (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model
from model.nerf_raybased import PointSampler

sd = O.make_state_dict(43, seed=0)
m = build_model(sd, 43)
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
c2w = torch.from_numpy(O.pose_spherical(30., -30., 4.)[:3, :4])
out = {}
for mode in ("1", "0"):
    os.environ["R2L_NO_FWD2"] = mode
    with torch.no_grad():
        for _ in range(2):
            rgb = m.render_pose(c2w, ps)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10):
            rgb = m.render_pose(c2w, ps)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    out[mode] = rgb.reshape(-1, 3).cpu()
    print("R2L_NO_FWD2=%s: %.3f ms/frame  %.2f M rays/s" % (mode, dt * 1e3, 160000 / dt / 1e6))
print("max|fp16x2 - bf16x3| = %.3e" % (out["0"] - out["1"]).abs().max().item())
rows = torch.arange(0, 160000, 97)[:1024]
dirs = O.pixel_dirs(400, 400, 555.5555155968841)
emb = O.positional_embed(O.sample_test(dirs, O.z_vals(16, 2., 6.), c2w)[rows], 10)
ref = O.r2l_forward(sd, emb)
print("vs oracle (1024 px): bf16x3 %.3e  fp16x2 %.3e" % ((out["1"][rows] - ref).abs().max().item(), (out["0"][rows] - ref).abs().max().item()))
