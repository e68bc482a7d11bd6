"""GPU parity of the teacher path (fused NeRF MLP, raw2outputs, sample_pdf+sort, render_rays) through the C ABI,
against reference-made goldens and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["fp16x2", "bf16x3", "f32mfma"])
def mlp_path(request, monkeypatch):
    """Every test runs on the three point-network kernels: r2l_teacher2.hip (3 fp16 products per fp32 product, the
    default), r2l_teacher3.hip (6 bf16 products: fp32-exact products, R2L_NO_FWD2=1) and r2l_teacher_mlp.hip (exact-fp32
    MFMA, R2L_NO_FWD3=1)."""
    from tests.conftest import use_family
    use_family(monkeypatch, precision={"fp16x2": "fp16x2", "bf16x3": "bf16x3", "f32mfma": "fp32_mfma"}[request.param])
    return request.param
T = torch.from_numpy


def make_teacher(sd):
    from model.nerf_raybased import NeRF
    m = NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
    m.load_state_dict(sd)
    return m.cuda()


@pytest.mark.parametrize("S", [64, 192])
@pytest.mark.parametrize("wb", [False, True])
def test_raw2outputs_golden(golden_dir, S, wb):
    from r2l_amd.render import raw2outputs
    g = np.load(os.path.join(golden_dir, "raw2outputs.npz"))
    outs = raw2outputs(T(g["S%d/raw" % S]).cuda(), T(g["S%d/z" % S]).cuda(), T(g["S%d/d" % S]).cuda(), 0, wb)
    for name, t in zip(("rgb", "disp", "acc", "weights", "depth"), outs):
        ref = g["S%d_wb%d/%s" % (S, int(wb), name)]
        got = t.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), name  # disp = NaN on the empty ray, nowhere else
        np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(ref), rtol=2e-5, atol=2e-6, err_msg=name)


# S = 64 / 128 / 192 / 256 take the quarter-wave-per-ray kernel (r2l_raw2outputs16_kernel: 16 rays per workgroup, so R = 37 / 21 / 5
# leave tail rows and tail waves), every other S the one-ray-per-wave kernel.  S=1 is degenerate in the reference (empty dists).
@pytest.mark.parametrize("S,R", [(2, 3), (16, 5), (65, 7), (200, 9), (256, 1000), (64, 37), (128, 21), (192, 5), (64, 4099)])
def test_raw2outputs_shapes_vs_oracle(S, R):
    from r2l_amd.render import raw2outputs
    g = torch.Generator().manual_seed(S)
    raw = torch.randn(R, S, 4, generator=g) * 3
    z = torch.sort(torch.rand(R, S, generator=g) * 4 + 2, -1)[0]
    d = torch.randn(R, 3, generator=g)
    noise = torch.randn(R, S, generator=g)
    ref = O.raw2outputs(raw, z, d, None, True)
    got = raw2outputs(raw.cuda(), z.cuda(), d.cuda(), 0, True)
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=3e-5, atol=3e-6)
    # raw_noise_std > 0 (create_data.py:368-378): the same draw added to sigma before the ReLU on both sides
    ref = O.raw2outputs(raw, z, d, noise, False)
    got = raw2outputs(raw.cuda(), z.cuda(), d.cuda(), 1.0, False, noise=noise.cuda())
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=3e-5, atol=3e-6)
    # the fine pass's form: no weights written, the four maps unchanged bit for bit
    lean = raw2outputs(raw.cuda(), z.cuda(), d.cuda(), 1.0, False, noise=noise.cuda(), need_weights=False)
    assert lean[3] is None
    for a, b in zip(lean[:3] + lean[4:], got[:3] + got[4:]):  # (disp is NaN on an empty ray: compare the bits)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def _sample_pdf_last_admissible(bins, weights, got_last):
    """Membership test for sample_pdf(det=True)'s last sample (u = 1.0): the answers the reference's op sequence
    (/root/reference/utils/run_nerf_raybased_helpers.py:315-328) can give.
      * its cdf[-1] rounds to <= 1: clamped index pair (last, last) -> bins[-1] exactly (candidate a);
      * its cdf[-1] rounds to >  1: pair (last - 1, last), interpolated (candidate b);
      * and when the last bin is EMPTY (weight 0 + 1e-5) on a ray whose weights sum to ~1 — every opaque ray — its pdf is
        1e-5 / (1 + 62e-5), within fp32's spacing at 1.0 (6e-8) of the `denom < 1e-5` clamp (helpers:325): cdf[-1] - cdf[-2]
        comes out as 167 or 168 ulps = 0.9954e-5 or 1.0014e-5 depending on the rounding of the running sum (torch's CPU cumsum
        accumulates in double, its CUDA cumsum is a parallel scan, the kernel sums left to right in fp32), so t is ~1e-5 or
        ~1: any value between the last two bin edges is an answer of the reference on SOME platform.  `ambiguous` marks those.
    got_last None -> (candidate a, candidate b, ambiguous)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    cand_a = bins[:, -1]
    denom = cdf[:, -1] - cdf[:, -2]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    cand_b = bins[:, -2] + (1.0 - cdf[:, -2]) / denom * (bins[:, -1] - bins[:, -2])
    ambiguous = (pdf[:, -1] - 1e-5).abs() <= 1.2e-7
    if got_last is None:
        return cand_a, cand_b, ambiguous
    g = torch.as_tensor(got_last)
    tol = 1e-5 + 1e-5 * cand_a.abs()
    between = (g >= bins[:, -2] - tol) & (g <= bins[:, -1] + tol)
    return (((g - cand_a).abs() <= tol) | ((g - cand_b).abs() <= tol) | (ambiguous & between)).numpy()


def test_sample_pdf_sort_golden(golden_dir):
    from r2l_amd.render import sample_pdf_sort
    g = np.load(os.path.join(golden_dir, "sample_pdf.npz"))
    bins, w = T(g["bins"]), T(g["weights"])
    R = bins.shape[0]
    # rebuild a z_vals / weights pair whose mid-points and inner weights are the golden's bins / weights
    # (sample_pdf_sort takes the coarse z and weights like render_rays does): use z with exact midpoints instead
    z = torch.zeros(R, 64)
    z[:, 0] = bins[:, 0] - 0.01
    for k in range(63):
        z[:, k + 1] = 2 * bins[:, k] - z[:, k]
    mids = .5 * (z[:, 1:] + z[:, :-1])
    wfull = torch.cat([torch.zeros(R, 1), w, torch.zeros(R, 1)], -1)
    for det, u, key in ((True, torch.linspace(0., 1., 128), "samples_det"), (False, T(g["u_pytest"]), "samples_pytest")):
        ref = O.sample_pdf(mids, w, 128, det=det, u=None if det else u)
        zs, z_all, z_std = sample_pdf_sort(z.cuda(), wfull.cuda(), 128, det=det, u=u)
        got = zs.cpu().numpy()
        bad = np.abs(got - ref.numpy()) > 1e-5 + 1e-5 * np.abs(ref.numpy())
        if det:
            # u == 1.0 (det's last sample) sits exactly on cdf[-1] ~= 1 +- 1 ulp, and which side it falls on depends on the
            # rounding of the reference's own vectorised torch.sum (not reproducible across CPUs).  Both outcomes are
            # spelled out by helpers:315-328 and the kernel must return ONE OF THEM:
            #   cdf[-1] <= 1: inds = len(cdf) -> below = above = last -> denom 0 -> 1 -> sample = bins[-1] exactly;
            #   cdf[-1] >  1: inds = last -> (below, above) = (last - 1, last), denom < 1e-5 -> 1 in a degenerate bin.
            ok_last = _sample_pdf_last_admissible(mids, w, got[:, -1])
            assert ok_last.all(), np.argwhere(~ok_last)[:10]
            bad[:, -1] = False
        assert not bad.any(), np.argwhere(bad)[:10]
        ref_all = torch.sort(torch.cat([z, zs.cpu()], -1), -1)[0]
        assert torch.equal(z_all.cpu(), ref_all)  # sorting is exact
        np.testing.assert_allclose(z_std.cpu().numpy(), torch.std(zs.cpu(), dim=-1, unbiased=False).numpy(), rtol=1e-4)


def test_sample_pdf_sort_three_paths_agree_bit_for_bit():
    """r2l_sample_pdf_sort16_kernel (round 6; reference: helpers:283-330 sample_pdf + the sort of create_data.py:513-517) sorts the
    128 samples and MERGES them with the 64 coarse depths instead of sorting all 256 values — legitimate only while the coarse
    depths ascend, which the kernel checks per wave (four rays).  All its paths must give torch.sort's bits:
      (a) random u, ascending z           -> sort of the samples + merge;
      (b) det u (a linspace), ascending z -> the samples already ascend: their sort is skipped, merge only;
      (c) rays whose z is NOT ascending (never produced by the render stack; the C ABI accepts any z) -> the full 256-element
          network, for that wave, while its neighbours stay on the merge;
    and z_samples / z_std carry the same bits as the generic one-ray-per-wave kernel (same summation association)."""
    import ctypes
    from r2l_amd import _lib
    from r2l_amd.render import sample_pdf_sort
    lib = _lib.load()
    R, S, NI = 203, 64, 128
    g = torch.Generator().manual_seed(77)
    z = torch.sort(torch.rand(R, S, generator=g) * 4 + 2, -1)[0]
    z[5, 10:20] = z[5, 10]           # ties
    w = torch.rand(R, S, generator=g) ** 3
    u = torch.rand(R, NI, generator=g)
    u[7] = torch.sort(u[7])[0]       # one ray of a random wave with ascending samples: its wave still sorts
    u[9, 3] = u[9, 4]                # equal uniforms
    bad = z.clone()
    for r in (0, 41, 42, 43, 130, 202):  # non-ascending coarse depths: a single swap, a reversed row, a late inversion
        bad[r] = z[r].flip(0) if r == 41 else z[r]
    bad[0, [3, 4]] = bad[0, [4, 3]]
    bad[42, [62, 63]] = bad[42, [63, 62]]
    bad[43, [7, 8]] = bad[43, [8, 7]]      # (an inversion across the boundary of two lanes' eight depths)
    bad[130, [31, 32]] = bad[130, [32, 31]]
    bad[202, [0, 63]] = bad[202, [63, 0]]
    assert (bad[:, 1:] < bad[:, :-1]).any(1).sum().item() == 6

    def generic(zz, uu):  # the one-ray-per-wave kernel: reached through an odd u stride (the 16-lane kernel needs 16-byte rows)
        upad = torch.zeros(R, NI + 1)
        upad[:, :NI] = uu
        ud, zd_in, wd = upad.cuda(), zz.cuda().contiguous(), w.cuda().contiguous()  # (kept alive until the results are read back)
        zs, za, zd = (torch.empty(R, NI, device="cuda"), torch.empty(R, S + NI, device="cuda"), torch.empty(R, device="cuda"))
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(lib.r2l_sample_pdf_sort(p(zd_in), p(wd), p(ud), NI + 1, p(zs), p(za), p(zd), R, S, NI,
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "r2l_sample_pdf_sort")
        return zs.cpu(), za.cpu(), zd.cpu()

    for zz, uu, det in ((z, u, False), (z, torch.linspace(0., 1., NI).expand(R, NI).contiguous(), True), (bad, u, False)):
        zs, z_all, z_std = sample_pdf_sort(zz.cuda(), w.cuda(), NI, det=det, u=uu if not det else torch.linspace(0., 1., NI))
        zs, z_all, z_std = zs.cpu(), z_all.cpu(), z_std.cpu()
        assert torch.equal(z_all, torch.sort(torch.cat([zz, zs], -1), -1)[0])
        gs, ga, gd = generic(zz, uu)
        assert torch.equal(zs, gs) and torch.equal(z_all, ga) and torch.equal(z_std, gd)
        if det:
            assert (zs[:, 1:] >= zs[:, :-1]).all()  # the monotone inverse cdf: what lets the kernel skip the sort


@pytest.mark.parametrize("S,NI,R", [(64, 128, 37), (64, 128, 4099), (64, 64, 9), (33, 77, 21), (16, 192, 5), (5, 3, 7)])
def test_sample_pdf_sort_shapes_vs_oracle(S, NI, R):
    """(64, 128): the quarter-wave-per-ray kernel (r2l_sample_pdf_sort16_kernel; 16 rays per workgroup: R = 37 / 4099 leave tail
    rows), every other shape the one-ray-per-wave kernel: per-ray random u, samples vs the oracle's sample_pdf (helpers:283-330),
    the merged depths exactly torch.sort's, z_std."""
    from r2l_amd.render import sample_pdf_sort
    g = torch.Generator().manual_seed(S * 1000 + NI)
    z = torch.sort(torch.rand(R, S, generator=g) * 4 + 2, -1)[0]
    w = torch.rand(R, S, generator=g) ** 3
    w[R // 2] = 0.  # an empty ray: uniform pdf from the 1e-5 floor
    u = torch.rand(R, NI, generator=g)
    mids = .5 * (z[:, 1:] + z[:, :-1])
    ref = O.sample_pdf(mids, w[:, 1:-1], NI, det=False, u=u)
    zs, z_all, z_std = sample_pdf_sort(z.cuda(), w.cuda(), NI, det=False, u=u)
    # Conditioning: sample = b0 + (u - c0) / (c1 - c0) * (b1 - b0).  torch's CPU cumsum accumulates in double, the kernels sum
    # left to right in fp32 (as torch's fp32 semantics say): c0 differs by up to 2 ulp at ~1 (2.4e-7), which a narrow cdf step
    # (a bin holding 1e-4 of the ray's weight) amplifies by 1 / step — an fp32 emulation of the kernel's order on the CPU
    # reproduces both the count (38 of 524 672 at R = 4099) and the size of the deviations.  Bar: 1e-5 + that term with margin.
    # helpers:325 also replaces a step below 1e-5 by 1 — a discontinuity of the reference's own formula: where the step sits
    # within rounding of 1e-5 the sample only has to stay in its bin.
    pdf = (w[:, 1:-1] + 1e-5) / torch.sum(w[:, 1:-1] + 1e-5, -1, keepdim=True)
    cdf = torch.cat([torch.zeros(R, 1), torch.cumsum(pdf, -1)], -1)
    inds = torch.searchsorted(cdf, u.contiguous(), right=True)
    below, above = (inds - 1).clamp(min=0), inds.clamp(max=cdf.shape[-1] - 1)
    step = (torch.gather(cdf, 1, above) - torch.gather(cdf, 1, below)).numpy()
    lo_edge, hi_edge = torch.gather(mids, 1, below).numpy(), torch.gather(mids, 1, above).numpy()
    knife = np.abs(step - 1e-5) < 3e-7
    assert knife.mean() < 1e-3
    got = zs.cpu().numpy()
    allowed = 1e-5 + 1e-5 * np.abs(ref.numpy()) + (hi_edge - lo_edge) * 4e-7 / np.maximum(step, 1e-5)
    close = np.abs(got - ref.numpy()) <= allowed
    assert (close | knife).all(), np.argwhere(~(close | knife))[:10]
    assert (np.abs(got - ref.numpy()) > 1e-5 + 1e-5 * np.abs(ref.numpy())).mean() < 3e-4  # (and the amplified ones are rare)
    assert ((got >= lo_edge - 1e-5) & (got <= hi_edge + 1e-5))[knife].all()
    assert torch.equal(z_all.cpu(), torch.sort(torch.cat([z, zs.cpu()], -1), -1)[0])  # sorting is exact
    np.testing.assert_allclose(z_std.cpu().numpy(), torch.std(zs.cpu(), dim=-1, unbiased=False).numpy(), rtol=1e-4, atol=1e-6)


def test_teacher_mlp_vs_oracle():
    from r2l_amd.render import teacher_engine
    coarse, _ = O.make_teacher_state_dicts(11, 2, alpha_bias=0.5)
    m = make_teacher(coarse)
    g = torch.Generator().manual_seed(0)
    R, S = 37, 64
    o = torch.randn(R, 3, generator=g)
    d = torch.randn(R, 3, generator=g)
    vd = d / d.norm(dim=-1, keepdim=True)
    z = torch.sort(torch.rand(R, S, generator=g) * 4 + 2, -1)[0]
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    with torch.no_grad():
        ref = O.run_network(coarse, pts, vd)
        raw = teacher_engine(m).mlp(o.cuda(), d.cuda(), vd.cuda(), z.cuda()).cpu()
    err = (raw - ref).abs().max().item()
    print("teacher raw max err", err, "ref scale", ref.abs().max().item())
    assert err < 2e-5


def test_render_rays_golden(golden_dir):
    """render_rays with the seeded teacher pair vs the reference's own dict (perturb=0 and the pytest=True path)."""
    from r2l_amd.render import render_rays
    g = np.load(os.path.join(golden_dir, "render_rays.npz"))
    csd, fsd = O.make_teacher_state_dicts(11, 2, alpha_bias=0.5)
    coarse, fine = make_teacher(csd), make_teacher(fsd)
    rb = T(g["ray_batch"]).cuda()
    with torch.no_grad():
        det = render_rays(rb, coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=0.)
        rnd = render_rays(rb, coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1.,
                          pytest=True)
    for tag, ret in (("det", det), ("pytest", rnd)):
        for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "rgb0", "disp0", "acc0", "z_std"):
            if tag + "/" + k not in g.files:
                continue
            np.testing.assert_allclose(ret[k].cpu().numpy(), g[tag + "/" + k], rtol=2e-4, atol=1e-4,
                                       err_msg=tag + "/" + k)
        assert np.abs(ret["rgb_map"].cpu().numpy() - g[tag + "/rgb_map"]).max() < 1e-4


def test_render_frame_chunks():
    """render() over a small frame with 2 chunks equals one chunk; shapes as the reference returns them."""
    from r2l_amd.render import render
    csd, fsd = O.make_teacher_state_dicts(3, 2, alpha_bias=0.5)
    coarse, fine = make_teacher(csd), make_teacher(fsd)
    c2w = T(O.pose_spherical(20., -40., 4.)[:3, :4]).cuda()
    kw = dict(network_fn=coarse, network_query_fn=None, N_samples=64, N_importance=128, network_fine=fine,
              white_bkgd=True, perturb=0., ndc=False, near=2., far=6., use_viewdirs=True)
    with torch.no_grad():
        a = render(20, 24, 30., chunk=1 << 15, c2w=c2w, **kw)
        b = render(20, 24, 30., chunk=200, c2w=c2w, **kw)
    assert a[0].shape == (20, 24, 3) and a[1].shape == (20, 24) and "rgb0" in a[3]
    assert torch.equal(a[0], b[0])


# ---------------------------------------------------------------------------------------------------------------------
# Teacher parity OFF the init distribution: a coarse / fine pair fitted to an analytic scene
# (tests/teacher_util.py) — sigma pre-activations -10^2 .. 10^3, hidden activations of a few 10^2, like a trained NeRF.
# ---------------------------------------------------------------------------------------------------------------------
_TRAINED = {}


def trained_like_pair():
    if not _TRAINED:
        from tests.teacher_util import fit_teacher, teacher_stats
        for name, seed in (("coarse", 21), ("fine", 22)):
            _TRAINED[name] = fit_teacher(seed, steps=800, n=4096, device="cuda")
        for name, sd in _TRAINED.items():
            hmax, smin, smax = teacher_stats(sd)
            print("trained-like %s: max |hidden| %.0f, sigma pre-activation %.0f .. %.0f" % (name, hmax, smin, smax))
            assert hmax > 50 and smax > 200 and smin < -20  # the fit really left the init distribution
    return _TRAINED["coarse"], _TRAINED["fine"]


def scene_rays(R, seed, H=181, W=181, focal=250.):
    """R rays of one camera on the r = 4 sphere looking at the analytic scene, as render() packs them (create_data.py:97-176)."""
    from r2l_amd.render import get_rays
    c2w = T(O.pose_spherical(35., -25., 4.)[:3, :4])
    ro, rd = get_rays(H, W, focal, c2w)
    ro, rd = ro.reshape(-1, 3)[:R], rd.reshape(-1, 3)[:R]
    vd = rd / rd.norm(dim=-1, keepdim=True)
    return torch.cat([ro, rd, 2. * torch.ones_like(rd[:, :1]), 6. * torch.ones_like(rd[:, :1]), vd], -1).float()


def test_teacher_mlp_trained_like_vs_oracle(mlp_path):
    """r2l_teacher*_mlp raw on trained-like weights: sigma reaches 10^3, so the bar is relative to the value (fp32 itself
    resolves 6e-5 at 10^3); the colour logits keep the absolute bar of the init-distribution test."""
    from r2l_amd.render import teacher_engine
    coarse, _ = trained_like_pair()
    m = make_teacher(coarse)
    rb = scene_rays(181 * 181, 0)[torch.randperm(181 * 181, generator=torch.Generator().manual_seed(1))[:96]]
    o, d, vd = rb[:, 0:3].contiguous(), rb[:, 3:6].contiguous(), rb[:, 8:11].contiguous()
    z = torch.sort(torch.rand(96, 64, generator=torch.Generator().manual_seed(2)) * 4 + 2, -1)[0]
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    with torch.no_grad():
        ref = O.run_network(coarse, pts, vd)
        ref64 = O.run_network({k: v.double() for k, v in coarse.items()}, pts.double(), vd.double())
        raw = teacher_engine(m).mlp(o.cuda(), d.cuda(), vd.cuda(), z.cuda()).cpu()
    assert ref[..., 3].max().item() > 200 and ref[..., 3].min().item() < -20  # the rays do cross the dense parts
    # Yardstick: the reference's own fp32 arithmetic against fp64.  With hidden activations of ~2e2 a sigma of 1e3 is a
    # cancelling sum of 256 terms of total magnitude ~1e4: ANY fp32 evaluation order is only good to ~1e-3 there.
    e_ref = (ref.double() - ref64).abs()
    e_hip = (raw.double() - ref64).abs()
    print("trained-like teacher raw (%s): |hip - fp64| rgb %.3g sigma %.3g;  |reference fp32 - fp64| rgb %.3g sigma %.3g;  "
          "|hip - reference fp32| %.3g at |sigma| up to %.0f" % (
              mlp_path, e_hip[..., :3].max().item(), e_hip[..., 3].max().item(), e_ref[..., :3].max().item(),
              e_ref[..., 3].max().item(), (raw - ref).abs().max().item(), ref[..., 3].abs().max().item()))
    for sl in (slice(0, 3), slice(3, 4)):
        assert e_hip[..., sl].max().item() <= 3.0 * e_ref[..., sl].max().item() + 2e-5
    assert (raw - ref)[..., :3].abs().max().item() < 1e-4  # colour logits: the absolute bar of the init-distribution test


def test_render_rays_trained_like_full_chunk(mlp_path):
    """render_rays (64 + 128 samples, perturb = 0, white background) on ONE FULL 32 768-ray chunk with the trained-like pair;
    every map of 1024 of its rays against the oracle's render_rays on the same inputs (rays are independent)."""
    from r2l_amd.render import render_rays
    csd, fsd = trained_like_pair()
    coarse, fine = make_teacher(csd), make_teacher(fsd)
    rb = scene_rays(32768, 0)
    with torch.no_grad():
        ret = render_rays(rb.cuda(), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=0.)
    pick = torch.randperm(32768, generator=torch.Generator().manual_seed(3))[:1024]
    with torch.no_grad():
        ref = O.render_rays(rb[pick], csd, fsd, 64, 128, perturb=0., white_bkgd=True)
    acc = ref["acc_map"]
    assert (acc > 0.99).float().mean().item() > 0.2 and (acc < 0.01).float().mean().item() > 0.02  # surfaces AND empty rays
    # Bars: colours at north_star's 1e-4.  Opacity / depth of this scene amplify the fp32 noise of sigma (3e-4 at 10^3, the
    # reference's own distance from fp64: test above) through exp(-sigma * dist) at the shell's flanks: the exact-fp32 MFMA
    # kernel, the bf16x3 and the fp16x2 one all sit at the same 1.2e-4 from the oracle, i.e. it is summation order, not the splits.
    errs = {}
    for k, bar in (("rgb_map", 1e-4), ("rgb0", 1e-4), ("acc_map", 3e-4), ("acc0", 3e-4), ("depth_map", 1.5e-3)):
        errs[k] = ((ret[k][pick.cuda()].cpu() - ref[k]).abs().max().item(), bar)
    # z_std = std of the 128 importance samples: its last one (u = 1.0) has two admissible values (test_sample_pdf_sort_golden),
    # so the oracle's samples are rebuilt here and z_std is accepted against either
    sub = rb[pick]
    zc = (2. * (1. - torch.linspace(0., 1., 64)) + 6. * torch.linspace(0., 1., 64)).expand(1024, 64)
    with torch.no_grad():
        raw0 = O.run_network(csd, sub[:, None, 0:3] + sub[:, None, 3:6] * zc[:, :, None], sub[:, 8:11])
        w0 = O.raw2outputs(raw0, zc, sub[:, 3:6], None, True)[3]
        mids = .5 * (zc[:, 1:] + zc[:, :-1])
        zs = O.sample_pdf(mids, w0[:, 1:-1], 128, det=True)
    got_std = ret["z_std"][pick.cuda()].cpu()
    cand_a, cand_b, ambiguous = _sample_pdf_last_admissible(mids, w0[:, 1:-1], None)
    e_std = torch.full_like(got_std, float("inf"))
    for cand in (cand_a, cand_b):
        e_std = torch.minimum(e_std, (got_std - torch.std(torch.cat([zs[:, :-1], cand[:, None]], -1), -1, unbiased=False)).abs())
    # ambiguous last sample (opaque rays, see _sample_pdf_last_admissible): anywhere between the last two bin edges, and one
    # of 128 samples moving by d changes the std by at most d / sqrt(128)
    slack = torch.where(ambiguous, (mids[:, -1] - mids[:, -2]) / 128**0.5, torch.zeros_like(e_std))
    assert ambiguous.float().mean().item() > 0.1  # the scene does have opaque rays
    errs["z_std"] = ((e_std - slack).clamp_min(0.).max().item(), 2e-4)
    print("trained-like render_rays (%s): " % mlp_path + ", ".join("%s %.3g" % (k, e) for k, (e, _) in errs.items()))
    for k, (e, bar) in errs.items():
        assert e < bar, (k, e)
    # disparity = 1 / max(1e-10, depth / acc): relative bar where it is defined
    dg, dr = ret["disp_map"][pick.cuda()].cpu(), ref["disp_map"]
    ok = torch.isfinite(dr) & (acc > 1e-3)
    assert ((dg[ok] - dr[ok]).abs() / dr[ok].abs().clamp_min(1e-3)).max().item() < 1e-3


def test_teacher_range_control(mlp_path, monkeypatch):
    """A teacher whose first hidden layer leaves fp16's range (|x| ~ 1e5 > R2L_F2_RANGE): r2l_teacher2 raises its flag and the
    bf16x3 kernel launched behind it redoes the launch — the result is BIT FOR BIT r2l_teacher3's (forced with R2L_NO_FWD2=1) —
    and the stream is re-packed for a power-of-two activation scale on the device (include/r2l_hip.h "range control"; the
    teacher's weights never change, a sticky guard would be for good): the launches after it run on the fp16 kernel again,
    flag 0, no further fallbacks, all within the oracle's bar.  Mirrors tests/test_forward_gpu.py::test_fp16_range_control."""
    if mlp_path != "fp16x2":
        pytest.skip("one comparison")
    from r2l_amd.render import teacher_engine
    sd = {k: v.clone() for k, v in O.make_teacher_state_dicts(5, 1, alpha_bias=0.5)[0].items()}
    sd["pts_linears.0.weight"] *= 1.0e5
    sd["pts_linears.0.bias"] *= 1.0e5
    sd["pts_linears.1.weight"] *= 1.0e-5  # back to the usual scale behind the out-of-range layer
    g = torch.Generator().manual_seed(0)
    R, S = 4099, 64  # several workgroups, ragged
    o = torch.randn(R, 3, generator=g)
    d = torch.randn(R, 3, generator=g)
    vd = d / d.norm(dim=-1, keepdim=True)
    z = torch.sort(torch.rand(R, S, generator=g) * 4 + 2, -1)[0]
    pts = o[:64, None, :] + d[:64, None, :] * z[:64, :, None]
    emb = torch.cat([O.nerf_embed(pts.reshape(-1, 3), 10)], -1)
    h0 = torch.relu(torch.nn.functional.linear(emb, sd["pts_linears.0.weight"], sd["pts_linears.0.bias"]))
    assert h0.max().item() > 4.0e4  # the case really leaves the guarded range
    args = (o.cuda(), d.cuda(), vd.cuda(), z.cuda())
    with torch.no_grad():
        ref = O.run_network(sd, pts, vd[:64])
        m = make_teacher(sd)
        eng = teacher_engine(m)
        raws, infos = [], []
        for _ in range(5):
            raws.append(eng.mlp(*args).cpu())
            infos.append(eng.range_info())
        from tests.conftest import use_family
        use_family(monkeypatch, precision="bf16x3")
        raw3 = teacher_engine(make_teacher(sd)).mlp(*args).cpu()
    assert torch.equal(raws[0], raw3)  # the launch that tripped: redone by the bf16x3 kernel
    assert infos[0]["trips"] == 1 and infos[0]["scale"] >= 4 and infos[0]["flag"] == 0, infos[0]
    assert all(i["trips"] == 1 and i["flag"] == 0 for i in infos[1:]), infos  # ... the later ones stay on the fp16 kernel
    assert infos[2]["scale"] == infos[4]["scale"] and torch.equal(raws[3], raws[4])
    assert 4.0 <= infos[4]["headroom"] <= 16.0 and infos[4]["amax"] > 4.0e4, infos[4]
    for r in raws:
        assert torch.isfinite(r).all()
        # with 1e5-sized activations the oracle's own fp32 rounding is ~1e-2 absolute before the 1e-5 layer behind it
        assert (r[:64] - ref).abs().max().item() < 2e-4
        assert (r - raw3).abs().max().item() < 2e-4


@pytest.mark.parametrize("variant", ["black_bkgd", "coarse_only", "one_net", "retraw", "jitter_fixed"])
def test_render_rays_option_variants_vs_cpu_mirror(mlp_path, variant):
    """The reference's render_rays options off the create_data defaults (create_data.py:405-544: white_bkgd=False,
    N_importance=0, network_fine=None -> the coarse net evaluates the fine pass, retraw=True, perturb=1 with given uniforms): the
    HIP kernels against the torch-op branch of the SAME mirror on CPU tensors, which tests/test_render_cpu.py pins to the
    reference's goldens."""
    from model.nerf_raybased import NeRF
    from r2l_amd.render import get_embedder, render_rays, run_network
    sds = O.make_teacher_state_dicts(11, 2, alpha_bias=0.5)
    cpu_nets = []
    for sd in sds:
        m = NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
        m.load_state_dict(sd)
        cpu_nets.append(m)
    gpu_nets = [make_teacher(sd) for sd in sds]
    e10, _ = get_embedder(10)
    e4, _ = get_embedder(4)
    qfn = lambda pts, vd, fn: run_network(pts, vd, fn, embed_fn=e10, embeddirs_fn=e4)
    rb = scene_rays(257, 0)  # (ragged: not a multiple of the kernels' tiles)
    kw = dict(N_importance=128, white_bkgd=True, perturb=0.)
    fine = [cpu_nets[1], gpu_nets[1]]
    if variant == "black_bkgd":
        kw["white_bkgd"] = False
    elif variant == "coarse_only":
        kw["N_importance"] = 0
    elif variant == "one_net":
        fine = [None, None]
    elif variant == "retraw":
        kw["retraw"] = True
    elif variant == "jitter_fixed":
        kw.update(perturb=1., pytest=True)  # np.random.seed(0) uniforms on both sides (create_data.py:477-480, helpers:301-309)
    with torch.no_grad():
        ref = render_rays(rb, cpu_nets[0], qfn, 64, network_fine=fine[0], **kw)
        out = render_rays(rb.cuda(), gpu_nets[0], None, 64, network_fine=fine[1], **kw)
        if variant == "black_bkgd":  # (lindisp sampling — LLFF scenes — is outside the accelerated path: loud on either device)
            for dev_rb, net, q in ((rb, cpu_nets[0], qfn), (rb.cuda(), gpu_nets[0], None)):
                with pytest.raises(NotImplementedError):
                    render_rays(dev_rb, net, q, 64, lindisp=True)
    assert set(out) == set(ref), (sorted(out), sorted(ref))
    for k in ref:
        a, b = out[k].cpu(), ref[k]
        assert a.shape == b.shape, k
        if k == "z_std" and kw["perturb"] == 0.:
            continue  # det=True: the last sample has two admissible values (test_sample_pdf_sort_golden checks membership)
        nan = torch.isnan(b)
        assert torch.equal(torch.isnan(a), nan), k
        bar = 2e-3 if k == "raw" else (5e-4 if k.startswith("disp") else 1e-4)  # raw sigma ~ 30: fp32 noise of a 256-term sum; disp = 1 / depth
        err = (a[~nan] - b[~nan]).abs().max().item() if (~nan).any() else 0.
        assert err < bar, (variant, k, err)
