"""GPU parity of the fused HIP forward (through the C ABI) against the CPU oracle and the reference-made goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


FAMILIES = {
    "main": dict(tiling="main"),
    "coopf": dict(tiling="coopf"),
    "coopf2": dict(tiling="coopf", coop_tiles=2),
    "main-bf16x3": dict(tiling="main", precision="bf16x3"),
    "main-f32mfma": dict(tiling="main", precision="fp32_mfma"),
    "coop16": dict(tiling="coop16"),
}


@pytest.fixture(autouse=True, params=list(FAMILIES))
def chain_variant(request, monkeypatch):
    """Every test runs under each forward kernel family, selected through r2l_config (tests/conftest.py use_family): one wave
    per tile on the fp16x2 matrix path (r2l_fwd2.hip, the default of forward-only launches), on the bf16x3 path (r2l_fwd3.hip)
    and on the fp32 MFMA (r2l_forward.hip), the cooperative fp16x2 kernels (r2l_coopf_fwd.hip: one tile per workgroup, the
    default of small launches; coopf2: two) and the cooperative fp32-MFMA small-batch family (16-ray tiles; the 32-ray one was
    retired in round 5: profiles/r05_dispatch_table.md)."""
    from tests.conftest import use_family
    use_family(monkeypatch, **FAMILIES[request.param])
    return request.param


T = torch.from_numpy
TOL = 1e-4  # north_star: RGB within 1e-4 abs of the reference PyTorch path


def build_model(sd, n_block):
    import argparse
    from model.nerf_raybased import NeRF_v3_2
    trial = argparse.Namespace(ON=True, body_arch="resmlp", inact="relu", outact="none", res_scale=1., n_learnable=2,
                               n_block=-1, near=-1, far=-1)
    args = argparse.Namespace(netdepth=2 * n_block + 2, netwidth=256, layerwise_netwidths="", act="relu",
                              linear_tail=False, use_residual=True, trial=trial)
    m = NeRF_v3_2(args, 1008, 3)
    m.load_state_dict(sd)
    return m.cuda()


@pytest.fixture(scope="module")
def model88():
    sd = O.make_state_dict(n_block=43, seed=0)
    return sd, build_model(sd, 43)


def test_golden_w256d88_rays(golden_dir, model88):
    """256 rays, seeded W256D88 weights: HIP rgb vs the REFERENCE's own output."""
    from model.nerf_raybased import PointSampler
    g = np.load(os.path.join(golden_dir, "r2l_w256d88.npz"))
    sd, m = model88
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    with torch.no_grad():
        rgb = m.forward_rays(T(g["rays_o"]).cuda(), T(g["rays_d"]).cuda(), ps, perturb=0.)
    err = np.abs(rgb.cpu().numpy() - g["rgb"]).max()
    print("max |rgb - reference| =", err)
    assert err < TOL


def test_emb_path_matches_oracle(model88, chain_variant):
    """`model(embedded)` — the reference's module-boundary idiom (model/nerf_raybased.py:539-544) — follows the engine's precision
    (include/r2l_hip.h r2l_forward_emb_cfg, round 5): AUTO / fp32_mfma = the exact-fp32 kernel with the encoding as its B operand;
    bf16x3 (fp16x2 alike: no range-guard fallback on this path) = head on the fp32 MFMA, the 86 body layers + tail on the bf16x3
    chain from X_0.  Both within the parity bar of the oracle, bit-wise different from each other; ragged and one-tile sizes."""
    if chain_variant != "main":
        pytest.skip("configs set explicitly (one comparison)")
    from r2l_amd.engine import get_engine
    sd, m = model88
    eng = get_engine(m)
    torch.manual_seed(3)
    z = O.z_vals(16, 2., 6.)
    for n in (1000, 31, 4097):
        o = torch.randn(n, 3) * 2
        d = torch.randn(n, 3)
        emb = O.positional_embed(O.sample_train(o, d, z, 0.), 10)
        ref = O.r2l_forward(sd, emb)
        outs = {}
        try:
            for prec in ("auto", "fp32_mfma", "bf16x3", "fp16x2"):
                eng.set_config(precision=prec, tiling="auto")
                with torch.no_grad():
                    outs[prec] = m(emb.cuda()).cpu()
                err = (outs[prec] - ref).abs().max().item()
                print("emb path, %d rays, precision %s: max err %.2e" % (n, prec, err))
                assert err < TOL, (n, prec, err)
        finally:
            eng.set_config(precision="auto", tiling="auto")
        assert torch.equal(outs["auto"], outs["fp32_mfma"]) and torch.equal(outs["bf16x3"], outs["fp16x2"])
        assert not torch.equal(outs["bf16x3"], outs["fp32_mfma"])  # a different kernel did the body


@pytest.mark.parametrize("n", [0, 1, 31, 32, 33, 127, 4097])
def test_ragged_sizes_and_perturb(model88, n):
    """empty and ragged launches (N not a multiple of 32/128) and stratified jitter with a given t_rand."""
    from model.nerf_raybased import PointSampler
    sd, m = model88
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    g = torch.Generator().manual_seed(n)
    o = torch.randn(n, 3, generator=g) * 1.5
    d = torch.randn(n, 3, generator=g)
    u = torch.rand(n, 16, generator=g)
    z = O.z_vals(16, 2., 6.)
    for perturb, tr in ((0., None), (1., u)):
        with torch.no_grad():
            out = m.forward_rays(o.cuda(), d.cuda(), ps, perturb=perturb, t_rand=None if tr is None else tr.cuda())
        assert out.shape == (n, 3)
        if n == 0:  # (the reference's sample_train cannot reshape an empty batch: nothing to compare with; no launch here)
            continue
        emb = O.positional_embed(O.sample_train(o, d, z, perturb, tr), 10)
        ref = O.r2l_forward(sd, emb)
        assert (out.cpu() - ref).abs().max().item() < TOL


def test_pose_frame(model88):
    """whole 400x400 frame from a pose vs the oracle on 4096 sampled pixels; PSNR-vs-oracle on those pixels."""
    from model.nerf_raybased import PointSampler
    sd, m = model88
    H = W = 400
    focal = 555.5555155968841
    ps = PointSampler(H, W, focal, 16, 2., 6.)
    c2w = T(O.pose_spherical(30., -30., 4.)[:3, :4])
    with torch.no_grad():
        rgb = m.render_pose(c2w, ps).cpu()
    assert rgb.shape == (H * W, 3)
    rows = torch.randperm(H * W, generator=torch.Generator().manual_seed(0))[:4096]
    dirs = O.pixel_dirs(H, W, focal)
    pts = O.sample_test(dirs, O.z_vals(16, 2., 6.), c2w)[rows]
    ref = O.r2l_forward(sd, O.positional_embed(pts, 10))
    err = (rgb[rows] - ref).abs().max().item()
    mse = ((rgb[rows] - ref)**2).mean().item()
    print("pose max err", err, "psnr(hip vs oracle)", -10 * np.log10(max(mse, 1e-20)))
    assert err < TOL


def test_small_depth_and_gain(golden_dir):
    """a 2-block net with 4x larger head gain (stress on the positional-encoding precision)."""
    from model.nerf_raybased import PointSampler
    sd = O.make_state_dict(n_block=2, seed=5)
    sd["head.0.weight"] = sd["head.0.weight"] * 4
    m = build_model(sd, 2)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    g = np.load(os.path.join(golden_dir, "r2l_w256d88.npz"))
    o, d = T(g["rays_o"]), T(g["rays_d"])
    emb = O.positional_embed(O.sample_train(o, d, O.z_vals(16, 2., 6.), 0.), 10)
    ref = O.r2l_forward(sd, emb)
    with torch.no_grad():
        out = m.forward_rays(o.cuda(), d.cuda(), ps)
    assert (out.cpu() - ref).abs().max().item() < TOL


def _big_activation_net(gain=3.0e4):
    """3-block net with a head scaled up until |x_0| reaches ~1e5 (tail scaled down to keep the sigmoid active)."""
    sd = O.make_state_dict(n_block=3, seed=4)
    sd = {k: v.clone() for k, v in sd.items()}
    sd["head.0.weight"] *= gain
    sd["head.0.bias"] *= gain
    for k in sd:
        if k.startswith("tail."):
            sd[k] = sd[k] * (0.3 / gain)
    return sd


def test_fp16_range_control(chain_variant):
    """Activations beyond fp16's range (|x| ~ 1e5): the FIRST fp16x2 launch raises its flag, the bf16x3 kernel behind it redoes
    that launch — and re-packs the stream for a power-of-two activation scale, on the device (include/r2l_hip.h "range
    control") — so every LATER launch runs on the fp16 kernels again (flag 0, no further fallbacks), all of them within the
    parity bar of the oracle.  Launches that stay in range never change the scale: the default nets run at s = 1."""
    from model.nerf_raybased import PointSampler
    from r2l_amd.engine import get_engine
    sd = _big_activation_net()
    m = build_model(sd, 3)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    g = torch.Generator().manual_seed(11)
    n = 70000  # enough rays for the one-wave-per-tile kernels to be the natural choice as well
    o = torch.randn(n, 3, generator=g) * 1.5
    d = torch.randn(n, 3, generator=g)
    emb = O.positional_embed(O.sample_train(o[:2048], d[:2048], O.z_vals(16, 2., 6.), 0.), 10)
    ref, xs, ts = O.r2l_forward(sd, emb, return_acts=True)
    amax_ref = max(x.abs().max().item() for x in xs + ts)
    assert amax_ref > 4.0e4  # the case really leaves the guarded range
    eng = get_engine(m)
    fp16 = chain_variant in ("main", "coopf", "coopf2")
    with torch.no_grad():
        outs, infos = [], []
        for _ in range(5):
            outs.append(m.forward_rays(o.cuda(), d.cuda(), ps, perturb=0.))
            infos.append(eng.range_info())
    for r in outs:
        assert torch.isfinite(r).all()
        # relative bar: with activations of 1e5 the oracle's own fp32 rounding is ~1e-2 absolute before the 1e-5 tail
        assert (r[:2048].cpu() - ref).abs().max().item() < TOL
    if fp16:
        i1, i3, i5 = infos[0], infos[2], infos[4]
        assert i1["trips"] == 1 and i1["scale"] >= 4 and i1["flag"] == 0, i1  # redone once, re-scaled, guard open again
        # ... and never again.  (A value beyond 65504 makes the recorded amax inf: the first re-scale is then a blind x 256 and
        # the kernel behind the SECOND launch refines it from that launch's amax: settled from the third launch on.)
        assert all(i["trips"] == 1 and i["flag"] == 0 for i in infos[1:]), infos
        assert i3["scale"] == i5["scale"] and torch.equal(outs[3], outs[4])
        # telemetry: the largest |activation| of the scaled launches, in the model's own units (fp32 vs 2048 oracle rays)
        assert 0.5 * amax_ref < i5["amax"] < 4.0 * amax_ref, (i5, amax_ref)
        assert 4.0 <= i5["headroom"] <= 16.0, i5
    else:
        assert infos[-1]["trips"] == 0 and infos[-1]["scale"] == 1.0
        assert torch.equal(outs[0], outs[4])


def _body_amplified_net(n_block=6, amp=2.6):
    """Default-size head (|W| <= 0.0315, |x_0| ~ 5), body weights scaled up so that the activations GROW THROUGH THE BODY to ~2e5;
    tail scaled so that the logits stay O(10)."""
    sd = {k: v.clone() for k, v in O.make_state_dict(n_block=n_block, seed=4).items()}
    for k in sd:
        if k.startswith("body.") and k.endswith("body.2.weight"):
            sd[k] *= 4 * amp
        if k.startswith("body.") and k.endswith("body.0.weight"):
            sd[k] *= amp
    for k in sd:
        if k.startswith("tail."):
            sd[k] = sd[k] * (90.0 / 2.28e5)
    return sd


def test_fp16_range_control_body_amplified(chain_variant):
    """range control with a head of DEFAULT size and activations that grow through the body (|x_0| ~ 5, |x_6| ~ 2e5:
    the stream settles on s = 32).  Round 4 divided the head's weights by s before their fp16 split — 0.03 / 32 has no mid half
    left in fp16 (a CPU model of that rounding moves this net's rgb by 2.9e-4; 1.2e-5 at s = 1) — now the head stages are packed
    unscaled and the kernels scale the head's fp32 accumulators: the settled launches sit within the parity bar of the oracle
    like every other net, and within 8e-5 of the fp32-exact products of the bf16x3 family on the same rays."""
    from model.nerf_raybased import PointSampler
    from r2l_amd.engine import get_engine
    sd = _body_amplified_net()
    m = build_model(sd, 6)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    g = torch.Generator().manual_seed(11)
    n = 40000 if chain_variant.startswith("main") else 4096
    o = torch.randn(n, 3, generator=g) * 1.5
    d = torch.randn(n, 3, generator=g)
    emb = O.positional_embed(O.sample_train(o[:2048], d[:2048], O.z_vals(16, 2., 6.), 0.), 10)
    ref, xs, ts = O.r2l_forward(sd, emb, return_acts=True)
    assert xs[0].abs().max().item() < 10 and max(x.abs().max().item() for x in xs + ts) > 1.0e5
    eng = get_engine(m)
    with torch.no_grad():
        outs = [m.forward_rays(o.cuda(), d.cuda(), ps, perturb=0.) for _ in range(4)]
        info = eng.range_info()
        eng.set_config(precision="bf16x3", tiling="main")
        exact = m.forward_rays(o.cuda(), d.cuda(), ps, perturb=0.)
    for r in outs:
        assert (r[:2048].cpu() - ref).abs().max().item() < TOL
    if chain_variant in ("main", "coopf", "coopf2"):
        assert info["scale"] >= 16 and info["flag"] == 0 and info["trips"] >= 1, info
        assert torch.equal(outs[2], outs[3])  # settled
        assert (outs[3] - exact).abs().max().item() < 8e-5


def test_fp16_range_control_rescales_at_pack(chain_variant):
    """The scale follows the weights at every pack: activations that grow towards the guard without crossing it get a larger
    scale with the NEXT pack (no launch is ever redone), activations that shrink again bring it back to 1 — and a net in
    range is bit-identical whatever history the stream has seen (powers of two: exact)."""
    if chain_variant not in ("main", "coopf"):
        pytest.skip("fp16 families")
    from model.nerf_raybased import PointSampler
    from r2l_amd.engine import get_engine
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    g = torch.Generator().manual_seed(3)
    n = 40000 if chain_variant == "main" else 4000
    o, d = (torch.randn(n, 3, generator=g) * 1.5).cuda(), torch.randn(n, 3, generator=g).cuda()
    base = O.make_state_dict(n_block=3, seed=4)
    m = build_model(base, 3)
    eng = get_engine(m)
    with torch.no_grad():
        ref = m.forward_rays(o, d, ps, perturb=0.).clone()
        assert eng.range_info()["scale"] == 1.0
        # |x| ~ 1.6e4: in range (no flag), above the 8192 mark
        big = _big_activation_net(3.2e3)
        m.load_state_dict(big)
        r1 = m.forward_rays(o, d, ps, perturb=0.)
        i1 = eng.range_info()
        assert i1["trips"] == 0 and i1["scale"] == 1.0 and 8192 < i1["amax"] < 32768, i1
        eng.mark_dirty()  # "an optimizer step": the stream is re-packed before the next launch
        r2 = m.forward_rays(o, d, ps, perturb=0.)
        i2 = eng.range_info()
        assert i2["trips"] == 0 and i2["scale"] in (2.0, 4.0) and i2["rescales"] == 1, i2
        assert (r1 - r2).abs().max().item() < 2e-5
        # back to the small net: first launch still at the old scale (precision to spare), then s = 1 again
        m.load_state_dict(base)
        r3 = m.forward_rays(o, d, ps, perturb=0.)
        assert (r3 - ref).abs().max().item() < 2e-5
        eng.mark_dirty()
        r4 = m.forward_rays(o, d, ps, perturb=0.)
        i4 = eng.range_info()
        assert i4["scale"] == 1.0 and i4["trips"] == 0, i4
        assert torch.equal(r4, ref)


def test_explicit_config_selects_the_family(chain_variant, monkeypatch):
    """Dispatch through ARGUMENTS (include/r2l_hip.h r2l_config, the *_cfg entry points): one process, one model, four
    kernel families selected call by call with no environment switch involved — each bit-identical to the same family
    selected the environment way, all within the parity bar of the oracle; the teacher's precision likewise."""
    if chain_variant != "main":
        pytest.skip("one comparison")
    from model.nerf_raybased import NeRF, PointSampler
    from r2l_amd.engine import get_engine
    from r2l_amd.render import teacher_engine
    sd = O.make_state_dict(n_block=43, seed=0)
    m = build_model(sd, 43)
    eng = get_engine(m)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    g = torch.Generator().manual_seed(21)
    n = 6000
    o = torch.randn(n, 3, generator=g) * 1.5
    d = torch.randn(n, 3, generator=g)
    ref = O.r2l_forward(sd, O.positional_embed(O.sample_train(o[:512], d[:512], O.z_vals(16, 2., 6.), 0.), 10))
    oc, dc = o.cuda(), d.cuda()
    families = [
        (dict(precision="fp16x2", tiling="main"), dict(R2L_FORCE_VARIANT="main")),
        (dict(precision="bf16x3", tiling="main"), dict(R2L_FORCE_VARIANT="main", R2L_NO_FWD2="1")),
        (dict(precision="fp32_mfma", tiling="coop16"), dict(R2L_FORCE_VARIANT="coop16", R2L_NO_FWD3="1")),
        (dict(precision="fp16x2", tiling="coopf", coop_tiles=2), dict(R2L_FORCE_VARIANT="coopf", R2L_COOPF_TILES="2")),
    ]
    from tests.conftest import use_family
    use_family(monkeypatch)  # no defaults, clean environment: only what this test passes / sets
    by_cfg = []
    with torch.no_grad():
        for cfg, _ in families:  # arguments only: the environment is clean
            eng.set_config(**cfg)
            by_cfg.append(m.forward_rays(oc, dc, ps, perturb=0.).clone())
        eng.set_config(precision="auto", tiling="auto", coop_tiles=0)
        for (cfg, env), rgb in zip(families, by_cfg):
            assert (rgb[:512].cpu() - ref).abs().max().item() < TOL, cfg
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            assert torch.equal(m.forward_rays(oc, dc, ps, perturb=0.), rgb), cfg
            for k in env:
                monkeypatch.delenv(k)
    assert not torch.equal(by_cfg[0], by_cfg[1]) and not torch.equal(by_cfg[1], by_cfg[2])  # they ARE different kernels
    # explicit fields beat the environment
    monkeypatch.setenv("R2L_NO_FWD3", "1")
    eng.set_config(precision="bf16x3", tiling="main")
    with torch.no_grad():
        assert torch.equal(m.forward_rays(oc, dc, ps, perturb=0.), by_cfg[1])
    monkeypatch.delenv("R2L_NO_FWD3")
    # teacher: precision through its config
    tsd = O.make_teacher_state_dicts(11, 1, alpha_bias=0.5)[0]
    t = NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
    t.load_state_dict(tsd)
    te = teacher_engine(t.cuda())
    R, S = 257, 64
    to, td = torch.randn(R, 3, generator=g), torch.randn(R, 3, generator=g)
    vd = td / td.norm(dim=-1, keepdim=True)
    z = torch.sort(torch.rand(R, S, generator=g) * 4 + 2, -1)[0]
    raws = []
    from r2l_amd import _lib
    for prec, env in (("fp16x2", {}), ("bf16x3", {"R2L_NO_FWD2": "1"}), ("fp32_mfma", {"R2L_NO_FWD3": "1"})):
        te.cfg = _lib.make_config(precision=prec)
        raws.append(te.mlp(to.cuda(), td.cuda(), vd.cuda(), z.cuda()).clone())
        te.cfg = _lib.Config()
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert torch.equal(te.mlp(to.cuda(), td.cuda(), vd.cuda(), z.cuda()), raws[-1]), prec
        for k in env:
            monkeypatch.delenv(k)
    assert not torch.equal(raws[0], raws[1]) and not torch.equal(raws[1], raws[2])


def test_multi_pose_launch_equals_per_pose(chain_variant, model88):
    """r2l_forward_poses_cfg: K frames in one launch are bit for bit the K single-pose launches (a frame size that is NOT a
    multiple of the 32-ray tile, so tiles straddle frames); pinned cooperative tilings go frame by frame in the host layer."""
    from model.nerf_raybased import PointSampler
    sd, m = model88
    H, W = 37, 41
    ps = PointSampler(H, W, 50., 16, 2., 6., device="cuda")
    poses = torch.stack([T(O.pose_spherical(30. * k, -20. - 3 * k, 4.)[:3, :4]) for k in range(5)], 0)
    with torch.no_grad():
        one = torch.stack([m.render_pose(p, ps) for p in poses], 0)
        many = m.render_poses(poses, ps)
    assert many.shape == (5, H * W, 3) and torch.equal(one, many)
    ref = O.r2l_forward(sd, O.positional_embed(O.sample_test(O.pixel_dirs(H, W, 50.), O.z_vals(16, 2., 6.), poses[3])[:200], 10))
    assert (many[3, :200].cpu() - ref).abs().max().item() < TOL
