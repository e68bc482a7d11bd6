"""The render kernels on operands of different ENTROPY — same launch, same instruction stream, different data (the 16-bit
kernels run against the power cap, and what the multipliers cost depends on how many mantissa bits toggle).  W256 D88 student, 9 frames
of 400x400 per launch (the bench workload), three weight sets:
    default      nn.Linear default init (the bench's weights)
    fp16-exact   the same weights rounded to fp16: the `mid` halves of all weights are exactly zero (one of the three fp16 products of
                 every fp32 product multiplies zeros)
    zero body    body and tail weights zero, head as default: the weight operand of every MFMA of 86 of the 88 layers is zero (the hidden
                 activations come out zero, the residual stream stays at the head's output); loads, LDS traffic, operand splits,
                 barriers — everything else — unchanged
for the fp16x2 (default), bf16x3 and fp32-MFMA families:  python tools/operand_entropy_render.py [launches=12]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from r2l_amd.data import pose_spherical  # noqa: E402
from r2l_amd.engine import get_engine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
poses = torch.stack([torch.as_tensor(pose_spherical(-180.0 + 9.0 * i, -30., 4.)[:3, :4], dtype=torch.float32) for i in range(9)], 0).to(dev)


def variant(name):
    net, ps, _ = bench.make_model(dev)
    with torch.no_grad():
        if name == "fp16-exact":
            for p in net.parameters():
                p.copy_(p.half().float())
        elif name == "zero body":
            for k, p in net.named_parameters():
                if not k.startswith("head"):
                    p.zero_()
    return net, ps


for name in ("default", "fp16-exact", "zero body"):
    net, ps = variant(name)
    eng = get_engine(net)
    for fam in ("fp16x2", "bf16x3", "fp32_mfma"):
        eng.set_config(precision=fam)
        with torch.no_grad():
            for _ in range(3):
                net.render_poses(poses, ps)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(K):
                rgb = net.render_poses(poses, ps)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        tf = 9 * 160000 * 11789824 / (ms * 1e-3) / 1e12
        print("%-11s %-10s %8.2f ms per 9-frame launch  %6.1f TF algorithmic  (rgb mean %.4f)" % (name, fam, ms, tf, rgb.mean().item()), flush=True)
