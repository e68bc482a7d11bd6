"""As tools/operand_entropy_render.py, for the default training step (98 304 rays, fp16 trio / bf16x3 trio / exact fp32) and the teacher
frame (fp16x2 / fp32): default weights against weights whose body is zero (student: body + tail zero — the weight operand of every
MFMA of the forward is zero and the gradients are zero, so both factors of the backward's products vanish; teacher: every layer zero).
Same launches, same instruction streams:  python tools/operand_entropy_train.py [steps=40]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402  (seeded teacher weights: test infrastructure, this is a diagnosis tool)
from model.nerf_raybased import NeRF  # noqa: E402
from r2l_amd.engine import get_engine  # noqa: E402
from r2l_amd import _lib  # noqa: E402
from r2l_amd.render import render, teacher_engine  # noqa: E402
from r2l_amd.train_step import R2LTrainer, lr_schedule  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
n = 98304
g = torch.Generator().manual_seed(1234)
o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).to(dev)
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
tgt = torch.rand(n, 3, generator=g).to(dev)
for fam in ("fp16x2", "bf16x3", "fp32_mfma"):
    for name in ("default", "zero body"):
        net, ps, _ = bench.make_model(dev)
        if name == "zero body":
            with torch.no_grad():
                for k, p in net.named_parameters():
                    if not k.startswith("head"):
                        p.zero_()
        get_engine(net).set_config(precision=fam)
        tr = R2LTrainer(net, ps)
        for i in range(6):
            tr.step(o, d, tgt, 0.0, perturb=1.0)  # lr 0: the weights stay what they are
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(o, d, tgt, 0.0, perturb=1.0)
        torch.cuda.synchronize()
        print("train   %-10s %-10s %8.3f ms per step   %s" % (fam, name, (time.perf_counter() - t0) / steps * 1e3,
                                                             {k: v for k, v in tr.range_info().items() if k in ("scale", "trips", "bwd_trips", "grad_scale")} if fam == "fp16x2" else ""), flush=True)
        del tr, net

csd, fsd = O.make_teacher_state_dicts(11, 2, alpha_bias=0.5)
c2w = torch.from_numpy(O.pose_spherical(30., -30., 4.)[:3, :4]).to(dev)
for fam in ("fp16x2", "fp32_mfma"):
    for name in ("default", "zero"):
        nets = []
        for sd in (csd, fsd):
            m = NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
            m.load_state_dict(sd)
            if name == "zero":
                with torch.no_grad():
                    for p in m.parameters():
                        p.zero_()
            nets.append(m.to(dev))
        for m in nets:
            teacher_engine(m).cfg = _lib.make_config(precision=fam)
        kw = dict(network_fn=nets[0], network_query_fn=None, N_samples=64, N_importance=128, network_fine=nets[1], white_bkgd=True,
                  perturb=1., ndc=False, near=2., far=6., use_viewdirs=True)
        with torch.no_grad():
            render(400, 400, 555.5555155968841, chunk=32768, c2w=c2w, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                render(400, 400, 555.5555155968841, chunk=32768, c2w=c2w, **kw)
            torch.cuda.synchronize()
        print("teacher %-10s %-10s %8.2f ms per 400x400 frame" % (fam, name, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
