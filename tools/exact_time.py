"""Training step time, default fp16 trio vs exact-dW mode (r2l_config.dw_mode), per kernel from device events.
  python tools/exact_time.py [rays=98304]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model
from model.nerf_raybased import PointSampler
from r2l_amd.train_step import R2LTrainer, lr_schedule
n = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
sd = O.make_state_dict(n_block=43, seed=0)
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6., device="cuda")
g = torch.Generator().manual_seed(1)
o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
tgt = torch.rand(n, 3, generator=g).cuda()
for mode in ("fp16", "exact", "fp16", "exact"):
    m = build_model(sd, 43)
    tr = R2LTrainer(m, ps, dw_mode=mode)
    for i in range(3):
        tr.step(o, d, tgt, lr_schedule(i + 1, 5e-4, 500, "0.0001,200"), perturb=1.0)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    for i in range(10):
        ev[i].record()
        tr.step(o, d, tgt, lr_schedule(i + 4, 5e-4, 500, "0.0001,200"), perturb=1.0)
    ev[10].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))
    print("%d rays, dw_mode %-5s: step median %.3f ms (min %.3f)  loss %.6f" % (n, mode, ts[5], ts[0], tr.loss_out[0].item()))
