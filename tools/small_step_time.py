"""ms per default-trio training step at small ray counts (the shape of BASELINE configs[2] / [3] and of 8-GPU strong scaling), for
same-box A/Bs of launch-shape switches read from the environment (R2L_NO_ADAM_PACK, R2L_HEAD_SLICE_RAYS, R2L_LIB_PATH):
    python tools/small_step_time.py [label] [steps=400]     (GPU box; interleave settings in a shell loop)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from r2l_amd.train_step import R2LTrainer, lr_schedule  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
net, ps, _ = bench.make_model(dev)
out = []
for n in (4096, 12288):
    g = torch.Generator().manual_seed(1234)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).to(dev)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    tgt = torch.rand(n, 3, generator=g).to(dev)
    tr = R2LTrainer(net, ps)
    for i in range(20):
        tr.step(o, d, tgt, lr_schedule(i + 1, 5e-4, 500, "0.0001,200"), perturb=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(o, d, tgt, lr_schedule(i + 21, 5e-4, 500, "0.0001,200"), perturb=1.0)
    torch.cuda.synchronize()
    out.append("%d rays %.4f ms" % (n, (time.perf_counter() - t0) / steps * 1e3))
print("%-28s %s" % (label, "   ".join(out)), flush=True)
