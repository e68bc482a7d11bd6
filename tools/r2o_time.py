"""Eager launches of the teacher path's HBM kernels at the shapes of render_rays — r2l_raw2outputs16_kernel (S = 64 with weights, S = 192
without) and r2l_sample_pdf_sort16_kernel (64 coarse + 128 new depths) — on 32 768 rays, inputs cycled over several HBM-resident sets,
for rocprofv3 passes (tools/r05_profile.sh): python tools/r2o_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r2l_amd.render import raw2outputs, sample_pdf_sort  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
for S, need_w, n_sets in ((64, True, 8), (192, False, 4)):
    raws = [torch.randn(32768, S, 4, generator=g).to(dev) for _ in range(n_sets)]
    zs = [(torch.sort(torch.rand(32768, S, generator=g), -1)[0] * 4. + 2.).to(dev) for _ in range(n_sets)]
    d = torch.nn.functional.normalize(torch.randn(32768, 3, generator=g), dim=-1).to(dev)
    for i in range(24):
        raw2outputs(raws[i % n_sets], zs[i % n_sets], d, 0., True, need_weights=need_w)
    torch.cuda.synchronize()
zs = [(torch.sort(torch.rand(32768, 64, generator=g), -1)[0] * 4. + 2.).to(dev) for _ in range(8)]
ws = [(torch.rand(32768, 64, generator=g) ** 4).to(dev) for _ in range(8)]
us = [torch.rand(32768, 128, generator=g).to(dev) for _ in range(8)]
for i in range(24):
    sample_pdf_sort(zs[i % 8], ws[i % 8], 128, u=us[i % 8])
torch.cuda.synchronize()
print("done")
