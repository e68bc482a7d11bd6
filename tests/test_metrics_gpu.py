"""SSIM kernel (csrc/r2l_ssim.hip) through the C ABI against the reference's outputs and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


def test_ssim_golden(golden_dir):
    from r2l_amd import metrics
    g = np.load(os.path.join(golden_dir, "ssim.npz"))
    for tag in "abc":  # 40x52 (ragged tiles), 7x9 (smaller than the window), 33x16
        pred, gt = torch.from_numpy(g["pred_" + tag]).cuda(), torch.from_numpy(g["gt_" + tag]).cuda()
        got = metrics.ssim(pred, gt)
        assert got.is_cuda
        assert abs(got.item() - float(g["ssim_" + tag])) < 1e-5, (tag, got.item(), float(g["ssim_" + tag]))
        assert abs(metrics.ssim(gt, gt).item() - 1.0) < 1e-6


@pytest.mark.parametrize("shape", [(400, 400, 3), (401, 263, 3), (64, 64, 1), (16, 16, 4)])
def test_ssim_vs_oracle_fullsize(shape):
    from r2l_amd import metrics
    g = torch.Generator().manual_seed(shape[0] + shape[1])
    H, W, C = shape
    yy, xx = torch.meshgrid(torch.linspace(0, 9, H), torch.linspace(0, 7, W), indexing="ij")
    gt = (0.5 + 0.4 * torch.sin(xx * 1.7 + yy)[..., None] * torch.ones(C) + 0.1 * torch.rand(H, W, C, generator=g))
    gt = gt.clamp(0, 1)
    for noise in (0.0, 0.02, 0.3):
        pred = (gt + noise * torch.randn(H, W, C, generator=g)).clamp(0, 1)
        want = O.ssim(pred, gt).item()
        got = metrics.ssim(pred.cuda(), gt.cuda()).item()
        assert abs(got - want) < 1e-5, (shape, noise, got, want)
    # white frame vs itself and vs black: degenerate variances
    one, zero = torch.ones(H, W, C), torch.zeros(H, W, C)
    assert abs(metrics.ssim(one.cuda(), one.cuda()).item() - 1.0) < 1e-6
    assert abs(metrics.ssim(one.cuda(), zero.cuda()).item() - O.ssim(one, zero).item()) < 1e-6


def test_ssim_window_computed_by_the_library():
    """r2l_ssim(window_host = NULL): the library builds the 11x11 Gaussian itself (include/r2l_hip.h) — same value as with the
    window the Python layer passes (ssim_torch.py:11-25)."""
    import ctypes
    from r2l_amd import _lib, metrics
    L = _lib.load()
    g = torch.Generator().manual_seed(2)
    a = torch.rand(57, 83, 3, generator=g).cuda()
    b = (a + 0.05 * torch.randn(57, 83, 3, generator=g).cuda()).clamp(0, 1)
    want = metrics.ssim(a, b).item()
    partial = torch.empty(L.r2l_ssim_partial_count(57, 83, 3), device="cuda")
    out = torch.empty(1, device="cuda")
    _lib.check(L.r2l_ssim(a.data_ptr(), b.data_ptr(), 57, 83, 3, None, partial.data_ptr(), out.data_ptr(),
                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "r2l_ssim")
    assert abs(out.item() - want) < 1e-6, (out.item(), want)
