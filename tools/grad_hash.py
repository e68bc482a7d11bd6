"""Hash of (rgb, loss, flat gradient) of one training step per (precision, ray count), for comparing two builds of the library bit for bit:
    R2L_LIB_PATH=<a>/libr2l_hip.so python tools/grad_hash.py ; python tools/grad_hash.py   (GPU box)"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from r2l_amd.train_step import R2LTrainer  # noqa: E402

dev = torch.device("cuda", 0)
for prec, dw in (("fp32_mfma", "auto"), ("fp16x2", "fp16"), ("fp16x2", "exact"), ("bf16x3", "auto")):
    for n in (4096, 12288 - 5, 40000, 98304):
        net, ps, _ = bench.make_model(dev)
        g = torch.Generator().manual_seed(n)
        o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).to(dev)
        d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        tgt = torch.rand(n, 3, generator=g).to(dev)
        u = torch.rand(n, 16, generator=g).to(dev)
        tr = R2LTrainer(net, ps)
        tr.eng.set_config(precision=prec, dw_mode=dw)
        rgb = tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
        torch.cuda.synchronize()
        h = hashlib.sha1()
        for t in (rgb, tr.loss_out, tr.grads):
            h.update(t.detach().cpu().numpy().tobytes())
        print("%-10s dw %-5s %6d rays  %s" % (prec, dw, n, h.hexdigest()[:16]), flush=True)
