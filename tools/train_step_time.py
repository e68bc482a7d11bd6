"""ms per default-trio training step at 98 304 rays (README: N_rand 20 x 4096 + 20 % hard rays) for same-box A/Bs of library
variants (R2L_LIB_PATH) and environment switches:  python tools/train_step_time.py [label] [steps=60] [rays=98304] [precision=auto]
(precision fp32_mfma: the graded exact-fp32 family)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from r2l_amd.train_step import R2LTrainer, lr_schedule  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = int(sys.argv[3]) if len(sys.argv) > 3 else 98304
dev = torch.device("cuda", 0)
net, ps, _ = bench.make_model(dev)
g = torch.Generator().manual_seed(1234)
o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).to(dev)
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
tgt = torch.rand(n, 3, generator=g).to(dev)
tr = R2LTrainer(net, ps)
if len(sys.argv) > 4:
    tr.eng.set_config(precision=sys.argv[4])
for i in range(8):
    tr.step(o, d, tgt, lr_schedule(i + 1, 5e-4, 500, "0.0001,200"), perturb=1.0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
torch.cuda.synchronize()
t0 = time.perf_counter()
fwd = bwd = 0.0
for i in range(steps):
    # (forward_backward + adam, with events around the forward and the backward of every 4th step)
    tr.step(o, d, tgt, lr_schedule(i + 9, 5e-4, 500, "0.0001,200"), perturb=1.0)
torch.cuda.synchronize()
print("%-34s %d rays  %.4f ms per step" % (label, n, (time.perf_counter() - t0) / steps * 1e3), flush=True)
