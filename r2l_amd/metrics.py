"""Small numeric helpers shared by the drivers (reference: utils/run_nerf_raybased_helpers.py:14-20) and the
test-set SSIM (utils/ssim_torch.py)."""
import ctypes
import math

import numpy as np
import torch


def to_tensor(x, device=None):
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    return x.to(device) if isinstance(x, torch.Tensor) else torch.Tensor(x).to(device)


def to_array(x):
    return x if isinstance(x, np.ndarray) else x.data.cpu().numpy()


def to8b(x):
    return (255 * np.clip(to_array(x), 0, 1)).astype(np.uint8)


def img2mse(x, y):
    return torch.mean((x - y)**2)


def mse2psnr(x):
    return -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device))


_WINDOW = None


def _ssim_window():
    """The reference's 11x11 window, built the way ssim_torch.py:11-25 builds it (fp32 normalise, fp32 outer product)."""
    global _WINDOW
    if _WINDOW is None:
        g = torch.tensor([math.exp(-(x - 5)**2 / float(2 * 1.5**2)) for x in range(11)])
        g = g / g.sum()
        _WINDOW = (g[:, None] @ g[None, :]).contiguous()
    return _WINDOW


def ssim(img, ref):
    """SSIM of two [H, W, C] images in [0,1] (main.py:46 `ssim`, minus the permutes: the kernel reads HWC directly).
    CUDA tensors run the fused HIP kernel (r2l_ssim); CPU tensors (the CPU plumbing config) use torch conv2d."""
    assert img.shape == ref.shape and img.dim() == 3
    win = _ssim_window()
    if img.is_cuda:
        from . import _lib
        L = _lib.load()
        a, b = img.detach().float().contiguous(), ref.detach().to(img.device).float().contiguous()
        H, W, C = a.shape
        partial = torch.empty(L.r2l_ssim_partial_count(H, W, C), device=a.device)
        out = torch.empty(1, device=a.device)
        _lib.check(L.r2l_ssim(a.data_ptr(), b.data_ptr(), H, W, C, win.data_ptr(), partial.data_ptr(), out.data_ptr(),
                              ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)), "r2l_ssim")
        return out[0]
    import torch.nn.functional as F
    a, b = img.float().permute(2, 0, 1)[None], ref.float().permute(2, 0, 1)[None]
    C = a.shape[1]
    w = win[None, None].expand(C, 1, 11, 11).contiguous()
    mu1, mu2 = F.conv2d(a, w, padding=5, groups=C), F.conv2d(b, w, padding=5, groups=C)
    s1 = F.conv2d(a * a, w, padding=5, groups=C) - mu1 * mu1
    s2 = F.conv2d(b * b, w, padding=5, groups=C) - mu2 * mu2
    s12 = F.conv2d(a * b, w, padding=5, groups=C) - mu1 * mu2
    C1, C2 = 0.01**2, 0.03**2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()
