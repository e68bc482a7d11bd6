"""Golden vectors for the SSIM metric, produced by the REFERENCE's utils/ssim_torch.py (dev container only).

    python tests/golden/gen_golden_ssim.py       # needs /root/reference; writes tests/golden/ssim.npz

Called exactly as main.py:46 does: ssim_(img[None] in [N,C,H,W], ref[None])."""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/utils/ssim_torch.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    spec = importlib.util.spec_from_file_location("ref_ssim_torch", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(7)
    out = {}
    for tag, (H, W) in (("a", (40, 52)), ("b", (7, 9)), ("c", (33, 16))):
        yy, xx = torch.meshgrid(torch.linspace(0, 3, H), torch.linspace(0, 4, W), indexing="ij")
        gt = torch.stack([0.5 + 0.5 * torch.sin(2.1 * xx + yy), 0.5 + 0.5 * torch.cos(1.3 * yy * xx),
                          (xx / 4. + yy / 3.) / 2.], -1)
        gt = (gt + 0.1 * torch.rand(H, W, 3, generator=g)).clamp(0, 1)
        pred = (gt + 0.08 * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
        val = mod.ssim(pred.permute(2, 0, 1)[None], gt.permute(2, 0, 1)[None])
        same = mod.ssim(gt.permute(2, 0, 1)[None], gt.permute(2, 0, 1)[None])
        out["pred_" + tag], out["gt_" + tag] = pred.numpy(), gt.numpy()
        out["ssim_" + tag], out["ssim_same_" + tag] = np.float32(val.item()), np.float32(same.item())
    out["window"] = mod.create_window(11, 1)[0, 0].numpy()
    np.savez_compressed(os.path.join(OUT, "ssim.npz"), **out)
    print({k: float(v) for k, v in out.items() if k.startswith("ssim")})


if __name__ == "__main__":
    main()
