"""Quick forward timing on the GPU box (dev aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model
from model.nerf_raybased import PointSampler

sd = O.make_state_dict(43, seed=0)
m = build_model(sd, 43)
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
c2w = torch.from_numpy(O.pose_spherical(30., -30., 4.)[:3, :4])
for name, n in (("frame400", 160000), ("32768", 32768), ("131072", 131072)):
    o = torch.randn(n, 3, device="cuda"); d = torch.randn(n, 3, device="cuda")
    with torch.no_grad():
        for _ in range(2):
            m.forward_rays(o, d, ps)
        torch.cuda.synchronize()
        t0 = time.time()
        K = 5
        for _ in range(K):
            m.forward_rays(o, d, ps)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / K
    print("%s: %.3f ms  %.3f Mrays/s  %.1f TFLOP/s (%.1f%% of 157.3)" % (name, dt * 1e3, n / dt / 1e6, n * 11789824 / dt / 1e12, n * 11789824 / dt / 157.3e12 * 100))
with torch.no_grad():
    for _ in range(2): m.render_pose(c2w, ps)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): m.render_pose(c2w, ps)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print("pose frame: %.3f ms  %.3f Mrays/s %.1f TF" % (dt * 1e3, 160000 / dt / 1e6, 160000 * 11789824 / dt / 1e12))
