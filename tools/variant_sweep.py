"""Step / forward time of the chain variants over N (GPU box): which N should switch from the cooperative fp16x2 kernels
(one tile per workgroup, r2l_coopf) to the one-wave-per-tile ones (R2L_COOPF_MAX_RAYS, csrc/r2l_common.h); the 16-ray
fp32-MFMA cooperative family for comparison (R2L_FORCE_VARIANT=coop16 by hand); coopf/1, coopf/2 = ray tiles per workgroup."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model
from model.nerf_raybased import PointSampler
from r2l_amd.train_step import R2LTrainer, lr_schedule

sd = O.make_state_dict(43, seed=0)
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
for n in (1024, 4096, 8192, 8224, 12288, 16384, 24576, 32768, 49152, 65536, 98304):
    g = torch.Generator().manual_seed(1)
    o = (torch.randn(n, 3, generator=g) * 1.5).cuda(); d = torch.randn(n, 3, generator=g).cuda(); t = torch.rand(n, 3, generator=g).cuda()
    line = "N %6d:" % n
    for var in ("coopf/1", "coopf/2", "main", None):
        os.environ.pop("R2L_COOPF_TILES", None)
        if var is None:
            os.environ.pop("R2L_FORCE_VARIANT", None)
        else:
            os.environ["R2L_FORCE_VARIANT"] = var.split("/")[0]
            if "/" in var:
                os.environ["R2L_COOPF_TILES"] = var.split("/")[1]
        m = build_model(sd, 43)
        tr = R2LTrainer(m, ps)
        for i in range(3):
            tr.step(o, d, t, 1e-4, perturb=1.0)
        torch.cuda.synchronize(); t0 = time.time()
        for i in range(10):
            tr.step(o, d, t, 1e-4, perturb=1.0)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 10
        with torch.no_grad():
            for i in range(3):
                m.forward_rays(o, d, ps)
            torch.cuda.synchronize(); t0 = time.time()
            for i in range(10):
                m.forward_rays(o, d, ps)
            torch.cuda.synchronize(); df = (time.time() - t0) / 10
        line += "  %s step %.3f fwd %.3f ms |" % (var or "auto", dt * 1e3, df * 1e3)
    print(line)
