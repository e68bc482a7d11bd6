"""Gradient error of the 3-product gradient GEMMs (R2L_GRAD_TERMS=3) against the default 6-product path, W256D88,
98 304 rays (dev aid, GPU box): per-tensor max|d|/max|g| and the step time of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model
from model.nerf_raybased import PointSampler
from r2l_amd.train_step import R2LTrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
sd = O.make_state_dict(43, seed=0)
m = build_model(sd, 43)
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
g = torch.Generator().manual_seed(1)
o = (torch.randn(n, 3, generator=g) * 1.5).cuda(); d = torch.randn(n, 3, generator=g).cuda(); t = torch.rand(n, 3, generator=g).cuda()
u = torch.rand(n, 16, generator=g).cuda()
res = {}
for terms in ("6", "3", "exact"):
    os.environ["R2L_GRAD_TERMS"] = "6" if terms == "exact" else terms
    for k in ("R2L_NO_FWD2", "R2L_NO_BWD2", "R2L_NO_DW2"):  # "exact": every GEMM with six bf16 products (fp32-exact products)
        if terms == "exact":
            os.environ[k] = "1"
        else:
            os.environ.pop(k, None)
    tr = R2LTrainer(m, ps)
    tr.forward_backward(o, d, t, perturb=1., t_rand=u)
    torch.cuda.synchronize()
    res[terms] = tr.grads.clone()
    t0 = time.time()
    for _ in range(5):
        tr.forward_backward(o, d, t, perturb=1., t_rand=u)
    torch.cuda.synchronize()
    print("terms", terms, "fwd+bwd %.3f ms" % ((time.time() - t0) / 5 * 1e3))
def cmp(a, b, label):
    off = 0
    worst = 0.
    for k, v in sd.items():
        ga, gb = a[off:off + v.numel()], b[off:off + v.numel()]
        off += v.numel()
        worst = max(worst, ((ga - gb).abs().max() / ga.abs().max().clamp_min(1e-30)).item())
    print("%s: worst per-tensor max|d|/max|g| = %.3e ; whole-gradient relative L2 = %.3e" % (label, worst, ((a - b).norm() / a.norm()).item()))


cmp(res["exact"], res["6"], "default (fp16x2 trio) vs six-bf16-product kernels")
cmp(res["exact"], res["3"], "R2L_GRAD_TERMS=3 vs six-bf16-product kernels")
a, b = res["6"], res["3"]
off = 0
worst = 0.
for k, v in sd.items():
    ga, gb = a[off:off + v.numel()], b[off:off + v.numel()]
    off += v.numel()
    e = ((ga - gb).abs().max() / ga.abs().max().clamp_min(1e-30)).item()
    worst = max(worst, e)
print("worst per-tensor max|d|/max|g| = %.3e ; whole-gradient relative L2 = %.3e" % (worst, ((a - b).norm() / a.norm()).item()))
