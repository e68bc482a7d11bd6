"""GPU parity of the hand-written backward + fused Adam (through the C ABI) against the CPU oracle's autograd and
against the reference's own gradients frozen in tests/golden/r2l_w256d88.npz."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model

pytestmark = pytest.mark.gpu


FAMILIES = {
    "main": dict(tiling="main"),
    "coopf": dict(tiling="coopf"),
    "coopf2": dict(tiling="coopf", coop_tiles=2),
    "main-exact": dict(tiling="main", dw_mode="exact"),
    "coopf-exact": dict(tiling="coopf", dw_mode="exact"),
    "main-bf16x3-trio": dict(tiling="main", precision="bf16x3"),
    "main-f32mfma": dict(tiling="main", precision="fp32_mfma"),
    "coop16": dict(tiling="coop16"),
}


@pytest.fixture(autouse=True, params=list(FAMILIES))
def chain_variant(request, monkeypatch):
    """Every test runs under each kernel family of the training step, selected through r2l_config (tests/conftest.py
    use_family; the R2L_* switches named below are the equivalent environment overrides of AUTO fields):
      main             one wave per tile, the default trio: fp16x2 forward (r2l_fwd2.hip) and dX chain (r2l_bwd2.hip) stashing
                       fp16 stage pieces, fp16 weight-gradient GEMMs on them (r2l_dw16.hip); range-guarded, with the bf16x3
                       kernels launched behind them
      coopf            the same trio with the cooperative chains (r2l_coopf_fwd / _bwd.hip: one 32-ray tile per workgroup; the
                       default of steps up to 16 384 rays), same stash, same weight-gradient kernels
      coopf2           coopf with two ray tiles per workgroup forced (R2L_COOPF_TILES=2: what launches of more than one tile
                       per CU take)
      main-exact /     the fp16 trio with EXACT weight gradients (R2L_DW_EXACT=1 = r2l_config.dw_mode R2L_DW_EXACT): the chains
      coopf-exact      also stash the operands' mid halves and r2l_dw16 / r2l_dw_head16 take three products — held to the
                       strict bars of the fp32-exact families
      main-bf16x3-trio R2L_NO_FWD2 = R2L_NO_BWD2 = R2L_NO_DW2 = 1 (any one of them would do): the whole step on six bf16
                       products per fp32 product and the chunked fp32 stash — exactly the kernels the guards fall back to
      main-f32mfma     R2L_NO_FWD3=1: everything on the exact-fp32 MFMA
      coop16           the cooperative fp32-MFMA small-batch family (16-ray tiles)."""
    from tests.conftest import use_family
    use_family(monkeypatch, **FAMILIES[request.param])
    return request.param


T = torch.from_numpy


def split_flat(flat, sd):
    out, off = {}, 0
    for k, v in sd.items():
        out[k] = flat[off:off + v.numel()].view(v.shape)
        off += v.numel()
    return out


def rel_err(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def test_grads_vs_reference_golden(golden_dir):
    """W256D88, the 256 golden rays: loss, per-tensor grad norms and selected full grads recorded from the reference."""
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    g = np.load(os.path.join(golden_dir, "r2l_w256d88.npz"))
    sd = O.make_state_dict(n_block=43, seed=0)
    m = build_model(sd, 43)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    tr = R2LTrainer(m, ps)
    rgb = tr.forward_backward(T(g["rays_o"]).cuda(), T(g["rays_d"]).cuda(), T(g["target"]).cuda())
    assert np.abs(rgb.cpu().numpy() - g["rgb"]).max() < 1e-4
    assert abs(tr.loss_out[0].item() - float(g["loss"])) < 1e-6
    assert abs(tr.loss_out[1].item() - float(g["psnr"])) < 1e-3
    grads = split_flat(tr.grads.cpu(), sd)
    gn = np.array([v.norm().item() for v in grads.values()])
    np.testing.assert_allclose(gn, g["grad_norms"], rtol=1e-3)
    for key, name in (("grad_tail_w", "tail.0.weight"), ("grad_tail_b", "tail.0.bias"), ("grad_head_b", "head.0.bias"),
                      ("grad_body0_b0", "body.0.body.0.bias"), ("grad_body42_b2", "body.42.body.2.bias")):
        # 88 layers deep, a different fp32 summation order flips a few ReLU masks of near-zero pre-activations, which
        # moves individual gradient entries discretely: observed 4e-4 .. 1.1e-3 of the tensor's max across the variants
        assert rel_err(grads[name], T(g[key])) < 2e-3, name
    # single weight rows: one flipped mask of one of the 256 rays moves a whole row by that ray's share (~1/256)
    assert rel_err(grads["body.20.body.0.weight"][:4], T(g["grad_body20_w0_rows"])) < 1e-2
    assert rel_err(grads["head.0.weight"][:2], T(g["grad_head_w_rows"])) < 1e-2


@pytest.mark.parametrize("n,perturb", [(1, 0.), (33, 1.), (200, 0.), (1000, 1.)])
def test_full_grads_vs_oracle(n, perturb):
    """every gradient tensor of a 3-block net vs oracle autograd; ragged N, with and without stratified jitter."""
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    sd = O.make_state_dict(n_block=3, seed=2)
    m = build_model(sd, 3)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    gen = torch.Generator().manual_seed(n)
    o = torch.randn(n, 3, generator=gen) * 1.5
    d = torch.randn(n, 3, generator=gen)
    tgt = torch.rand(n, 3, generator=gen)
    u = torch.rand(n, 16, generator=gen)
    emb = O.positional_embed(O.sample_train(o, d, O.z_vals(16, 2., 6.), perturb, u), 10)
    loss, rgb_ref, gref = O.r2l_loss_and_grads(sd, emb, tgt)
    tr = R2LTrainer(m, ps)
    rgb = tr.forward_backward(o.cuda(), d.cuda(), tgt.cuda(), perturb=perturb, t_rand=u.cuda())
    assert (rgb.cpu() - rgb_ref).abs().max().item() < 1e-4
    assert abs(tr.loss_out[0].item() - loss.item()) < 1e-6
    grads = split_flat(tr.grads.cpu(), sd)
    for k in sd:
        assert rel_err(grads[k], gref[k]) < 2e-3, (k, rel_err(grads[k], gref[k]))


def test_three_adam_steps_vs_oracle(chain_variant):
    """3 fused steps (warm-up LR schedule, main.py:1181-1195) vs oracle autograd + Adam: losses and parameters."""
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer, lr_schedule
    sd = O.make_state_dict(n_block=2, seed=9)
    m = build_model(sd, 2)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    gen = torch.Generator().manual_seed(5)
    n = 512
    o = torch.randn(n, 3, generator=gen) * 1.5
    d = torch.randn(n, 3, generator=gen)
    tgt = torch.rand(n, 3, generator=gen)
    emb = O.positional_embed(O.sample_train(o, d, O.z_vals(16, 2., 6.), 0.), 10)
    ref = {k: v.clone() for k, v in sd.items()}
    mo = {k: torch.zeros_like(v) for k, v in sd.items()}
    vo = {k: torch.zeros_like(v) for k, v in sd.items()}
    tr = R2LTrainer(m, ps)
    for step in (1, 2, 3):
        lr = lr_schedule(step, 5e-4, 500, "0.0001,200")
        assert lr == O.lr_schedule(step, 5e-4, 500, "0.0001,200")
        loss, _, gr = O.r2l_loss_and_grads(ref, emb, tgt)
        for k in ref:
            ref[k], mo[k], vo[k] = O.adam_step(ref[k], gr[k], mo[k], vo[k], step, lr)
        _, lo = tr.step(o.cuda(), d.cuda(), tgt.cuda(), lr)
        assert abs(lo[0].item() - loss.item()) < 2e-6, step
    new = m.state_dict()
    travel = sum(lr_schedule(s, 5e-4, 500, "0.0001,200") for s in (1, 2, 3))  # how far Adam can move a weight in 3 steps
    for k in ref:
        # Adam's first steps move every weight by ~lr regardless of gradient size; compare on that scale
        diff = (new[k].cpu() - ref[k]).abs()
        if chain_variant in ("main", "coopf", "coopf2") and not k.startswith("tail"):
            # default trio: the weight-gradient GEMMs of head and body take fp16-rounded operands (r2l_dw16.hip,
            # r2l_dw_head16.hip).  Each gradient entry is a sum over the rays whose rounding errors average out (per-tensor
            # error ~1e-4 of its max here, 512 rays), but Adam normalises every entry by its own magnitude: the entries whose
            # true gradient nearly cancels (|g| below ~1e-3 of the tensor's typical entry; many of the head's, whose encoding
            # columns oscillate) can change sign and travel the other way.  A CPU model of the rounding (fp32 chain, dW
            # operands through .half()) reproduces this test's numbers (body.0.body.0.weight: 1.3e-3 of the entries beyond 2e-5
            # there, 1.28e-3 here): at most 1.3e-3 of a tensor's entries beyond 2e-5, the largest 2.9e-4, >= 94 % within
            # 1e-6; update direction cosine >= 0.99995.
            # Bars: those, with margin.  That this per-entry relaxation does not show in what training converges to is
            # measured, not argued: 32 seeds x 12 000 steps, held-out PSNR 25.579 +- 0.192 dB (this default) vs 25.565 +- 0.208 dB
            # (exact fp32 MFMA): +0.013 dB paired by seed, standard error 0.027 dB, against 0.2 dB between two seeds of one family
            # (profiles/r05_train_equivalence_seeds.txt, r06_train_equivalence_seeds.txt).  The strict 2e-5 bar below holds for every
            # fp32-exact family and for dw_mode="exact" (this test under the main-exact / coopf-exact variants).
            assert (diff > 2e-5).float().mean().item() < 1e-2, k
            assert (diff > 1e-6).float().mean().item() < 0.15, k
            assert diff.max().item() < 2 * travel, k
            a, b = (new[k].cpu() - sd[k]).flatten().double(), (ref[k] - sd[k]).flatten().double()
            assert (torch.dot(a, b) / (a.norm() * b.norm())).item() > 0.9998, k
        else:
            assert diff.max().item() < 2e-5, k
    # torch.optim.Adam-format state round trip
    osd = tr.optimizer_state_dict(lr)
    assert osd["state"][0]["exp_avg"].shape == sd["head.0.weight"].shape
    opt = torch.optim.Adam(list(m.parameters()), lr=5e-4)
    opt.load_state_dict(osd)


def test_gradient_accumulation_and_zeroing():
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    sd = O.make_state_dict(n_block=1, seed=4)
    m = build_model(sd, 1)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    gen = torch.Generator().manual_seed(8)
    o, d, tgt = torch.randn(64, 3, generator=gen), torch.randn(64, 3, generator=gen), torch.rand(64, 3, generator=gen)
    tr = R2LTrainer(m, ps)
    tr.forward_backward(o.cuda(), d.cuda(), tgt.cuda())
    g1 = tr.grads.clone()
    tr.forward_backward(o.cuda(), d.cuda(), tgt.cuda(), zero_grad=False)
    assert rel_err(tr.grads, 2 * g1) < 1e-5


def test_autograd_bridge_module_boundary():
    """rgb = model(embedded); loss.backward(); torch.optim.Adam.step() — the reference's idiom — on the HIP path."""
    from model.nerf_raybased import PointSampler, PositionalEmbedder
    sd = O.make_state_dict(n_block=2, seed=6)
    m = build_model(sd, 2)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    pe = PositionalEmbedder(10)
    gen = torch.Generator().manual_seed(2)
    n = 300
    o, d, tgt = torch.randn(n, 3, generator=gen), torch.randn(n, 3, generator=gen), torch.rand(n, 3, generator=gen)
    emb_ref = O.positional_embed(O.sample_train(o, d, O.z_vals(16, 2., 6.), 0.), 10)
    loss_ref, _, gref = O.r2l_loss_and_grads(sd, emb_ref, tgt)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    emb = pe(ps.sample_train(o.cuda(), d.cuda(), perturb=0.))
    rgb = m(emb)
    loss = torch.mean((rgb - tgt.cuda())**2)
    opt.zero_grad()
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-6
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu(), gref[k]) < 2e-3, k
    opt.step()
    with torch.no_grad():
        rgb2 = m(emb)  # parameters changed in place -> engine re-packs the weight stream
    assert (rgb2 - rgb).abs().max().item() > 1e-5


@pytest.mark.parametrize("n", [700, 5000])
def test_gradients_bit_reproducible_and_slab_matches_atomics(n):
    """With the partial-sum slab every weight gradient is reduced in a fixed order: two runs agree bit for bit.  The
    slab-less path (dw_slab = NULL, fp32 atomics) gives the same gradients up to summation order."""
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    sd = O.make_state_dict(n_block=43, seed=3)
    m = build_model(sd, 43)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    tr = R2LTrainer(m, ps)
    g = torch.Generator().manual_seed(n)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda()
    u = torch.rand(n, 16, generator=g).cuda()
    tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    g1 = tr.grads.clone()
    tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    assert torch.equal(g1, tr.grads)
    slab, tr.dw_slab = tr.dw_slab, None
    tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    tr.dw_slab = slab
    g3 = split_flat(tr.grads.cpu(), sd)
    for k, v in split_flat(g1.cpu(), sd).items():
        assert rel_err(g3[k], v) < 1e-5, k


@pytest.mark.parametrize("n,buckets", [(700, 4), (5000, 1), (5000, 5), (5000, 43)])
def test_staged_backward_equals_single_call(n, buckets, monkeypatch):
    """r2l_backward_part (dX chain + tail, body buckets from the last blocks to the first, head) — the form the
    overlapped gradient all-reduce drives — leaves the gradients of the one-call r2l_backward (bit-identical with one
    bucket; with several, equal up to the fp32 summation order of the per-workgroup partials)."""
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    sd = O.make_state_dict(n_block=43, seed=3)
    m = build_model(sd, 43)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    tr = R2LTrainer(m, ps)
    g = torch.Generator().manual_seed(n)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda()
    u = torch.rand(n, 16, generator=g).cuda()
    tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    g1, l1 = tr.grads.clone(), tr.loss_out.clone()
    tr.force_staged, tr.n_buckets = True, buckets
    if buckets == 5:  # as the data-parallel trainer runs it: 8 CUs kept free for the RCCL kernels (other work split)
        monkeypatch.setenv("R2L_RESERVE_CUS", "8")
    tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    assert tr.reducer.pending() == 0  # world == 1: nothing goes to a collective
    assert torch.equal(l1, tr.loss_out)
    if buckets == 1:  # same work list as the one-call form: same partial sums, same order
        assert torch.equal(g1, tr.grads)
    else:  # a bucket's (layer, ray-chunk) list is cut into other per-workgroup ranges: fp32 summation order only
        a, b = split_flat(g1.cpu(), sd), split_flat(tr.grads.cpu(), sd)
        for k in sd:
            assert rel_err(b[k], a[k]) < 1e-5, k
        staged = tr.grads.clone()
        tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
        assert torch.equal(staged, tr.grads)  # and the staged form is bit-reproducible run to run as well


@pytest.mark.parametrize("n", [4096, 12288 - 5])
def test_segmented_chain_equals_uncut(chain_variant, n, monkeypatch):
    """The dX chain of a small step cut into block segments (R2L_BWD_CHAIN with a layer range; weight gradients of a finished
    segment on a second stream beside the next segment, R2L_BWD_NOFALLBACK, guarded Adam): the chain's outputs — gx[0], the
    stash the weight-gradient kernels read, loss — are BIT-IDENTICAL to the uncut chain, hence the gradients equal the staged
    form with the same buckets bit for bit, and the one-call form up to the summation order of per-workgroup partials; three
    Adam steps end on the same parameters bit for bit; one tile (4096 rays) and two tiles (12 283) per workgroup."""
    if chain_variant not in ("coopf", "coopf-exact"):
        pytest.skip("segments are a property of the cooperative fp16 chains")
    from model.nerf_raybased import PointSampler
    from r2l_amd import _lib
    from r2l_amd.train_step import R2LTrainer, lr_schedule
    from tests.conftest import use_family
    use_family(monkeypatch)  # the default dispatch takes these sizes to the cooperative chains by itself
    sd = O.make_state_dict(n_block=43, seed=3)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    g = torch.Generator().manual_seed(n)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda()
    u = torch.rand(n, 16, generator=g).cuda()

    def run(segments, staged):
        m = build_model(sd, 43)
        tr = R2LTrainer(m, ps, chain_segments=segments)
        assert tr.lib.r2l_chain_segments_ok_cfg(n, 43, tr.eng._cfg()) == 1
        tr.force_staged, tr.n_buckets = staged, 4
        tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
        g1, gx0, l1 = tr.grads.clone(), tr.gx[:tr.lib.r2l_padded_rows(n) * 256].clone(), tr.loss_out.clone()
        for i in range(3):
            tr.step(o, d, tgt, lr_schedule(i + 1, 5e-4, 500, "0.0001,200"), perturb=1., t_rand=u)
        torch.cuda.synchronize()
        assert tr.drain() == 0 and not tr.segments_disabled
        return g1, gx0, l1, tr.eng.flat.clone()

    one = run(1, False)
    staged = run(1, True)
    seg = run(4, False)
    assert torch.equal(seg[1], one[1]) and torch.equal(seg[2], one[2])  # gx[0] (end of the chain), loss
    assert torch.equal(seg[0], staged[0]) and torch.equal(seg[3], staged[3])  # same buckets: same partial sums, same order
    a, b = split_flat(one[0].cpu(), sd), split_flat(seg[0].cpu(), sd)
    for k in sd:
        assert rel_err(b[k], a[k]) < 1e-5, k


def test_segmented_step_skips_itself_when_the_range_guard_trips(chain_variant, monkeypatch):
    """A segmented step runs without the bf16x3 fallback kernels: when the step belongs to them (here: a head scaled until
    |x_0| ~ 1e5 leaves fp16's range, so the forward falls back and every chain segment raises the status word) the update is
    skipped ON THE DEVICE — parameters and Adam moments untouched —, the trainer reads the step's validity word a fixed number
    of steps later and goes back to the uncut backward, whose fallback handles the same batch."""
    if chain_variant != "coopf":
        pytest.skip("one comparison")
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    sd = {k: v.clone() for k, v in O.make_state_dict(n_block=3, seed=4).items()}
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    n = 4096
    g = torch.Generator().manual_seed(1)
    o = (torch.randn(n, 3, generator=g) * 1.5).cuda()
    d = torch.randn(n, 3, generator=g).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda()
    tr = R2LTrainer(build_model(sd, 3), ps, chain_segments=3)
    tr.forward_backward(o, d, tgt)
    assert tr.gradients_valid()
    tr.step(o, d, tgt, 1e-4)
    torch.cuda.synchronize()
    assert not torch.equal(tr.eng.flat.cpu(), torch.cat([v.reshape(-1) for v in sd.values()]))  # a clean segmented step updates
    assert tr.skipped_steps == 0
    sd["head.0.weight"] *= 3.0e4
    sd["head.0.bias"] *= 3.0e4
    for k in sd:
        if k.startswith("tail."):
            sd[k] = sd[k] * 1.0e-5
    # (round 4: a trainer first calibrates the stream's activation scale on its first batch, include/r2l_hip.h "range control":
    # the same net then never skips — checked at the end; calibrate = False reproduces a range excursion in mid-training.  And
    # the guard is per launch: the fallback forward of the skipped step re-scales the stream, the NEXT step is clean.)
    tr = R2LTrainer(build_model(sd, 3), ps, chain_segments=3)
    tr.calibrate = False
    tr.forward_backward(o, d, tgt)
    assert not tr.gradients_valid()  # what a caller that reads tr.grads itself has to ask
    tr.forward_backward(o, d, tgt)
    assert tr.gradients_valid()      # the stream has been re-scaled: the same batch now stays on the fp16 kernels
    tr = R2LTrainer(build_model(sd, 3), ps, chain_segments=3)
    tr.calibrate = False
    p0, m0 = tr.eng.flat.clone(), tr.exp_avg.clone()
    tr.step(o, d, tgt, 1e-4)
    torch.cuda.synchronize()
    assert torch.equal(tr.eng.flat, p0) and torch.equal(tr.exp_avg, m0)  # skipped on the device
    assert tr.drain() == 1 and tr.segments_disabled
    tr.step(o, d, tgt, 1e-4)  # this one runs uncut
    torch.cuda.synchronize()
    assert not torch.equal(tr.eng.flat, p0) and torch.isfinite(tr.eng.flat).all()
    # the training loop never synchronises: the host runs steps ahead of the device, and the word of step i is read at the start
    # of step i + STATUS_LAG (the same step on every rank, so that all ranks leave the segmented form together)
    from r2l_amd.train_step import STATUS_LAG
    tr = R2LTrainer(build_model(sd, 3), ps, chain_segments=3)
    tr.calibrate = False
    p0 = tr.eng.flat.clone()
    tr.step(o, d, tgt, 1e-4)
    torch.cuda.synchronize()
    assert torch.equal(tr.eng.flat, p0)  # only the FIRST step is skipped ...
    for i in range(STATUS_LAG - 1):
        tr.step(o, d, tgt, 1e-4)
        assert not tr.segments_disabled
    torch.cuda.synchronize()
    assert not torch.equal(tr.eng.flat, p0)  # ... the later segmented ones are clean
    tr.step(o, d, tgt, 1e-4)  # reads step 1's word first: uncut from here on
    assert tr.segments_disabled and tr.skipped_steps == 1
    assert tr.drain() == 1
    # the default trainer calibrates on its first batch: the same net trains segmented without a single skipped step
    tr = R2LTrainer(build_model(sd, 3), ps, chain_segments=3)
    p0 = tr.eng.flat.clone()
    for i in range(STATUS_LAG + 2):
        tr.step(o, d, tgt, 1e-4)
    assert tr.drain() == 0 and not tr.segments_disabled and not torch.equal(tr.eng.flat, p0)
    info = tr.range_info()
    assert info["scale"] >= 4 and info["trips"] >= 1 and info.get("bwd_trips", 0) == 0, info


def test_generic_mode_backward_after_forward_rays():
    """C-ABI generic mode: r2l_forward_rays (with stash) followed by r2l_backward(target = NULL, drgb = caller's dL/drgb).  On
    the default trio the stash is fp16 and the dX chain needs a power-of-two scale it cannot derive from an MSE scale: it is
    chosen on the device from max |drgb| (r2l_gscale_kernel).  Same gradients as MSE mode fed the equivalent dL/drgb."""
    from model.nerf_raybased import PointSampler
    from r2l_amd import _lib
    from r2l_amd.engine import _ptr, _stream
    from r2l_amd.train_step import R2LTrainer
    sd = O.make_state_dict(n_block=43, seed=11)
    m = build_model(sd, 43)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    tr = R2LTrainer(m, ps)
    g = torch.Generator().manual_seed(12)
    n = 4500
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda()
    rgb = tr.forward_backward(o, d, tgt)
    g_mse = split_flat(tr.grads.clone().cpu(), sd)
    eng = tr.eng
    rgb2 = eng.forward_rays(o, d, ps.z_vals, 0., None, save=(tr.save_x, tr.save_t))
    assert torch.equal(rgb, rgb2)
    drgb = ((2.0 / (3.0 * n)) * (rgb2 - tgt)).contiguous()
    tr.grads.zero_()
    # (the config the forward was given: a stash is only readable by the family that wrote it, include/r2l_hip.h)
    _lib.check(tr.lib.r2l_backward_part_cfg(_ptr(o), _ptr(d), None, _ptr(eng.ztab(ps.z_vals, 0.)), None, _ptr(rgb2), None,
                                            _ptr(drgb), _ptr(tr.save_x), _ptr(tr.save_t), _ptr(tr.wstream_bwd), _ptr(eng.flat),
                                            eng.n_block, 0.0, _ptr(tr.dpre), _ptr(tr.gx), _ptr(tr.gt), None, _ptr(tr.grads),
                                            _ptr(tr.dw_slab), n, _stream(), _lib.BWD_ALL, 0, 2 * eng.n_block, eng._cfg()),
               "r2l_backward (generic)")
    g_gen = split_flat(tr.grads.cpu(), sd)
    for k in sd:
        assert torch.isfinite(g_gen[k]).all(), k
        # same masks, same chain; only the power-of-two scale (hence the fp16 rounding of the scaled gradients) may differ
        assert rel_err(g_gen[k], g_mse[k]) < 3e-4, (k, rel_err(g_gen[k], g_mse[k]))


@pytest.mark.parametrize("calibrate", [True, False])
def test_fp16_range_control_in_training(chain_variant, calibrate):
    """Activations beyond fp16's range (head scaled up until |x| ~ 1e5) and chain gradients far below it (tail scaled down by
    1e5).  calibrate = False: the step is redone by the bf16x3 kernels behind the fp16 ones (forward flag -> fp32 stash ->
    bf16x3 dX chain and weight gradients), as in rounds 2 - 3.  calibrate = True (the trainer's default): forward-only
    launches on the first batch re-scale the stream first (include/r2l_hip.h "range control"), so the STEP itself runs on the
    fp16 trio — stash of x / s, dW scaled back by s at the flush — and the gradient chain on a scale corrected for the tiny
    tail weights.  Either way rgb and every gradient match the oracle."""
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    sd = {k: v.clone() for k, v in O.make_state_dict(n_block=3, seed=4).items()}
    sd["head.0.weight"] *= 3.0e4
    sd["head.0.bias"] *= 3.0e4
    for k in sd:
        if k.startswith("tail."):
            sd[k] = sd[k] * 1.0e-5
    m = build_model(sd, 3)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    gen = torch.Generator().manual_seed(21)
    n = 600
    o = torch.randn(n, 3, generator=gen) * 1.5
    d = torch.randn(n, 3, generator=gen)
    tgt = torch.rand(n, 3, generator=gen)
    emb = O.positional_embed(O.sample_train(o, d, O.z_vals(16, 2., 6.), 0.), 10)
    loss, rgb_ref, gref = O.r2l_loss_and_grads(sd, emb, tgt)
    tr = R2LTrainer(m, ps)
    tr.calibrate = calibrate
    fp16 = chain_variant in ("main", "coopf", "coopf2", "main-exact", "coopf-exact")
    for it in range(2):  # the second pass: same step again, now certainly on the re-scaled stream
        rgb = tr.forward_backward(o.cuda(), d.cuda(), tgt.cuda())
        assert (rgb.cpu() - rgb_ref).abs().max().item() < 1e-4
        grads = split_flat(tr.grads.cpu(), sd)
        for k in sd:
            assert torch.isfinite(grads[k]).all(), k
            assert rel_err(grads[k], gref[k]) < 2e-3, (it, k, rel_err(grads[k], gref[k]))
        info = tr.range_info()
        if fp16:
            assert info["scale"] >= 4 and info["trips"] == 1, info
            # every step ran on the fp16 chains, except the uncalibrated first one (forward fell back: fp32 stash)
            assert info["bwd_trips"] == (0 if calibrate else 1), (it, info)
            assert info["grad_scale"] > 2.0 ** 20, info  # a-priori rule x 2^17 for the 1e-5 tail


def test_dx_chain_only_trip_expands_the_fp16_stash(chain_variant):
    """a step whose FORWARD stays on the fp16 kernels (calibrated: the stream sits at s = 8 for these 600 rays) while its
    dX CHAIN leaves fp16's range — default-size head, activations AND gradients growing ~4e4-fold through the body, so the
    a-priori gradient scale of the first step puts chain values at ~4e5 (a CPU model of it: u of block 0).  The bf16x3 chain and
    weight-gradient kernels behind the fp16 ones then meet an fp16 stage-piece stash of x / s: the fallback pack expands it in
    place to the chunked fp32 layout they read (csrc/r2l_bwd3.hip; round 4 read the pieces as fp32 and applied garbage body
    dW silently).  Every gradient of that step matches the oracle; the NEXT step re-centres the gradient scale on the tripped
    step's amax (r2l_bwd_prepare_kernel) and runs on the fp16 chains again — no step after the first falls back."""
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    from tests.test_forward_gpu import _body_amplified_net
    sd = _body_amplified_net()
    m = build_model(sd, 6)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    gen = torch.Generator().manual_seed(21)
    n = 600
    o = torch.randn(n, 3, generator=gen) * 1.5
    d = torch.randn(n, 3, generator=gen)
    tgt = torch.rand(n, 3, generator=gen)
    emb = O.positional_embed(O.sample_train(o, d, O.z_vals(16, 2., 6.), 0.), 10)
    loss, rgb_ref, gref = O.r2l_loss_and_grads(sd, emb, tgt)
    tr = R2LTrainer(m, ps)  # calibrate = True: forward-only launches settle the activation scale before the first step
    fp16 = chain_variant in ("main", "coopf", "coopf2", "main-exact", "coopf-exact")
    fwd_trips = None
    for it in range(3):
        rgb = tr.forward_backward(o.cuda(), d.cuda(), tgt.cuda())
        assert (rgb.cpu() - rgb_ref).abs().max().item() < 1e-4
        grads = split_flat(tr.grads.cpu(), sd)
        for k in sd:
            assert torch.isfinite(grads[k]).all(), (it, k)
            # (an amplifying net: the ReLU masks of near-zero pre-activations move the gradients of everything below them;
            # garbage operands would be O(1))
            assert rel_err(grads[k], gref[k]) < 1e-2, (it, k, rel_err(grads[k], gref[k]))
        flat, ref = tr.grads.cpu(), torch.cat([gref[k].reshape(-1) for k in sd])
        assert torch.nn.functional.cosine_similarity(flat, ref, dim=0).item() > 0.9999, it
        info = tr.range_info()
        if fp16:
            if fwd_trips is None:
                fwd_trips = info["trips"]
            assert info["scale"] >= 4 and info["trips"] == fwd_trips, (it, info)  # no FORWARD of a step fell back
            assert info["bwd_trips"] == 1, (it, info)  # the first step's chain, and only that one
            if it > 0:
                assert 2.0 ** 4 <= info["grad_amax"] * info["grad_scale"] <= 2.0 ** 10, (it, info)


def test_adam_packed_equals_adam_then_pack(chain_variant, monkeypatch):
    """r2l_adam_step_packed (round 5, opt-in R2L_ADAM_PACK=1: the optimizer kernel writes the body weights' stage pieces of both fp16x2 streams itself, a
    small kernel packs head / bias stages and commits the activation scale) against the separate launches it replaces
    (r2l_adam_step_guarded, then r2l_pack_forward_layout / r2l_pack_backward_layout at the next step): three training steps,
    parameters, moments and BOTH packed streams bit for bit, and the renders that read them."""
    if chain_variant not in ("main", "coopf", "main-exact"):
        pytest.skip("the default trio's step (fp16x2 layouts)")
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer, lr_schedule
    sd = O.make_state_dict(n_block=5, seed=3)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    gen = torch.Generator().manual_seed(9)
    n = 40000 if chain_variant != "coopf" else 3000
    o = (torch.randn(n, 3, generator=gen) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=gen).cuda()
    runs = []
    for fused in (True, False):
        if fused:
            monkeypatch.setenv("R2L_ADAM_PACK", "1")
        else:
            monkeypatch.delenv("R2L_ADAM_PACK", raising=False)
        m = build_model(sd, 5)
        tr = R2LTrainer(m, ps)
        for it in range(1, 4):
            tr.step(o, d, tgt, lr_schedule(it, 5e-4, 500, "0.0001,200"), perturb=0.)
            assert tr._fused_repack == fused
        tr.eng.ensure_packed(n)   # (separate form: the pack the NEXT launch would trigger)
        tr._pack_bwd(n)
        with torch.no_grad():
            rgb = m.forward_rays(o, d, ps, perturb=0.)
        torch.cuda.synchronize()
        runs.append((tr.eng.flat.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.eng.wstream.clone(), tr.wstream_bwd.clone(),
                     rgb.clone(), tr.range_info()))
    a, b = runs
    for k, name in enumerate(("params", "exp_avg", "exp_avg_sq")):
        assert torch.equal(a[k], b[k]), name
    # the packed streams as raw bits (fp16 pairs viewed through fp32 words: compare integers, NaN patterns included); the status
    # words behind them carry the same scale / epoch state
    assert torch.equal(a[3].view(torch.int32), b[3].view(torch.int32)), "forward stream"
    assert torch.equal(a[4].view(torch.int32), b[4].view(torch.int32)), "backward stream"
    assert torch.equal(a[5], b[5])
    assert a[6]["scale"] == b[6]["scale"] == 1.0 and a[6]["trips"] == b[6]["trips"] == 0


def test_gradient_scale_follows_the_gradients(chain_variant):
    """The power of two the fp16 dX chain runs on is chosen on the device, step to step, from the last clean step's largest
    chain value (r2l_bwd_prepare_kernel): kept while that lies in [2^-2, 2^13] scaled (so default nets run bit for bit on the
    a-priori scale of rounds 1 - 3), re-centred when the gradients drift out of the band — here by rescaling the TARGETS' error
    through lw_rgb between steps: x 2^-12 (would underflow fp16's mid halves) and x 2^+14 (would cross the guard)."""
    if chain_variant not in ("main", "coopf", "main-exact"):
        pytest.skip("fp16 trio")
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    sd = O.make_state_dict(n_block=5, seed=2)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    gen = torch.Generator().manual_seed(5)
    n = 40000 if chain_variant != "coopf" else 3000
    o = (torch.randn(n, 3, generator=gen) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=gen).cuda()
    tr = R2LTrainer(build_model(sd, 5), ps)
    tr.forward_backward(o, d, tgt)
    g0, i0 = tr.grads.clone(), tr.range_info()
    tr.forward_backward(o, d, tgt)
    i1 = tr.range_info()
    assert torch.equal(tr.grads, g0) and i1["grad_scale"] == i0["grad_scale"] and i1["bwd_trips"] == 0  # in band: kept
    assert 0.25 <= i1["grad_amax"] * i1["grad_scale"] <= 8192, i1
    for factor in (2.0 ** -12, 2.0 ** 14):
        tr.lw_rgb = factor  # the same step with every gradient x factor (a power of two: the truth is g0 * factor exactly)
        tr.forward_backward(o, d, tgt)  # still on the old scale's estimate x factor: out of band -> re-centred already
        ia = tr.range_info()
        assert ia["bwd_trips"] == 0, ia
        assert 2.0 ** 5 <= ia["grad_amax"] * ia["grad_scale"] <= 2.0 ** 9, ia
        assert rel_err(tr.grads / factor, g0) < 1e-3
        tr.forward_backward(o, d, tgt)
        ib = tr.range_info()
        assert ib["grad_scale"] == ia["grad_scale"] and ib["bwd_trips"] == 0
    tr.lw_rgb = 1.0


def test_coopf_two_tiles_bitwise(chain_variant, monkeypatch):
    """Two ray tiles per workgroup (r2l_coopf_*: NT = 2) share the weight loads and nothing else: rgb, loss and every gradient
    of a step are bit-for-bit those of one tile per workgroup — on an odd, ragged tile count of MORE tiles than CUs (the last
    workgroup's second tile re-does the last live tile; the one-tile launch has to keep its workgroups on separate CUs,
    csrc/r2l_coopf.h FC_SOLO_LDS_BYTES), several launches each — and within rounding of the one-wave-per-tile kernels."""
    if chain_variant != "coopf":
        pytest.skip("one comparison")
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    sd = O.make_state_dict(n_block=43, seed=0)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    n = 32 * 387 - 7
    g = torch.Generator().manual_seed(5)
    o = (torch.randn(n, 3, generator=g) * 1.5).cuda(); d = torch.randn(n, 3, generator=g).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda(); tr = torch.rand(n, 16, generator=g).cuda()

    def run():
        m = build_model(sd, 43)
        t = R2LTrainer(m, ps)
        rgb = t.forward_backward(o, d, tgt, perturb=1.0, t_rand=tr)
        return t.loss_out.clone(), rgb.clone(), t.grads.clone()

    monkeypatch.setenv("R2L_COOPF_TILES", "2")
    ref = run()
    assert ref[2].abs().max().item() > 0
    for tiles in ("1", "2", "1", "1"):
        monkeypatch.setenv("R2L_COOPF_TILES", tiles)
        out = run()
        for a, b in zip(out, ref):
            assert torch.equal(a, b), tiles
    from tests.conftest import use_family
    use_family(monkeypatch, tiling="main")
    out = run()
    assert (out[1] - ref[1]).abs().max().item() < 2e-6
    assert torch.nn.functional.cosine_similarity(out[2], ref[2], dim=0).item() > 0.9999


def test_coopf_one_tile_beside_other_kernels(chain_variant, monkeypatch):
    """The one-tile cooperative chains (BASELINE configs[2] / [3]: 4096 rays per step) stay bit-exact while OTHER kernels are
    resident on the same CUs: a second stream keeps every CU busy with (a) elementwise waves (small register footprint: they fit
    beside a 4 x 240-VGPR cooperative workgroup on every SIMD) and (b) fp16 matrix-pipe waves, for the whole duration of the
    step.  (Round 2's co-residency fault — two waves on a SIMD, a lost FMA term in the forward's tail — is removed at its
    instruction, csrc/r2l_coopf.h; a data-parallel step runs beside RCCL kernels, so neighbours are the normal case.)"""
    if chain_variant != "coopf":
        pytest.skip("one comparison")
    from model.nerf_raybased import PointSampler
    from r2l_amd import _lib
    from r2l_amd.train_step import R2LTrainer
    sd = O.make_state_dict(n_block=43, seed=0)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    n = 4096
    assert _lib.load().r2l_coop_tiles_for(n, 43) == 1
    g = torch.Generator().manual_seed(9)
    o = (torch.randn(n, 3, generator=g) * 1.5).cuda(); d = torch.randn(n, 3, generator=g).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda(); tr = torch.rand(n, 16, generator=g).cuda()
    m = build_model(sd, 43)
    t = R2LTrainer(m, ps)

    def run():
        rgb = t.forward_backward(o, d, tgt, perturb=1.0, t_rand=tr)
        return t.loss_out.clone(), rgb.clone(), t.grads.clone()

    ref = run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.rand(64 << 20, device="cuda")
    a16 = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
    for kind in ("elementwise", "matmul", "elementwise", "matmul"):
        with torch.cuda.stream(side):
            for _ in range(12):  # ~10 ms of neighbours: longer than the 0.8 ms step launched beside them
                if kind == "elementwise":
                    big.mul_(1.0000001).add_(1e-9)
                else:
                    a16 @ a16
        outs = [run() for _ in range(4)]
        side.synchronize()
        torch.cuda.synchronize()
        for out in outs:
            for x, y in zip(out, ref):
                assert torch.equal(x, y), kind


@pytest.mark.parametrize("n,dw_mode,segments", [(12288 - 5, "fp16", 1), (8192 + 32, "fp16", 1), (16384 - 64, "exact", 1),
                                                 (12288, "fp16", 3)])
def test_mixed_cooperative_launch_is_bit_identical(chain_variant, monkeypatch, n, dw_mode, segments):
    """Tile counts between one and two per CU (256 < tiles < 512; 12 288 rays = the per-GPU share of the README's step at 8
    GPUs, /root/reference/main.py:1371-1406) run as ONE grid of tiles - 256 two-tile and 512 - tiles one-tile cooperative
    workgroups (r2l_config.coop_tiles = 3 — opt-in: measured slower than the two-tile launch, profiles/r06_mixed_coopf_ab.txt;
    csrc/r2l_coopf_fwd.hip r2l_coopf_fwd_mixed_kernel).
    Every tile takes the path it takes in the one-tile and in the two-tile kernels: rgb, loss and the whole flat gradient are
    bit-identical to both forced forms — also with the mid halves stashed (exact dW) and with the dX chain cut into segments."""
    if chain_variant != "coopf":
        pytest.skip("one comparison")
    from model.nerf_raybased import PointSampler
    from r2l_amd import _lib
    from r2l_amd.train_step import R2LTrainer
    from tests.conftest import use_family
    use_family(monkeypatch)
    for k in ("R2L_COOPF_TILES", "R2L_FORCE_VARIANT"):
        monkeypatch.delenv(k, raising=False)
    nb = 43 if segments == 1 and dw_mode == "fp16" and n == 12288 - 5 else 6
    sd = O.make_state_dict(n_block=nb, seed=3)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    g = torch.Generator().manual_seed(n)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda(); tr = torch.rand(n, 16, generator=g).cuda()

    def run(tiles):
        m = build_model(sd, nb)
        t = R2LTrainer(m, ps, dw_mode=dw_mode, chain_segments=segments)
        t.calibrate = False
        t.eng.set_config(precision="fp16x2", tiling="coopf", coop_tiles=tiles)
        if segments > 1:
            t.force_staged = True
        assert _lib.load().r2l_coop_tiles_for_cfg(n, nb, t.eng._cfg()) == (tiles or 2)
        rgb = t.forward_backward(o, d, tgt, perturb=1.0, t_rand=tr)
        with torch.no_grad():
            plain = m.forward_rays(o, d, ps, perturb=1.0, t_rand=tr)
        return t.loss_out.clone(), rgb.clone(), t.grads.clone(), plain.clone()

    mixed, auto, one, two = run(3), run(0), run(1), run(2)
    assert torch.isfinite(mixed[2]).all() and mixed[2].abs().max().item() > 0
    for other, name in ((auto, "auto"), (one, "one tile"), (two, "two tiles")):
        for x, y, what in zip(mixed, other, ("loss", "rgb", "grads", "forward-only rgb")):
            assert torch.equal(x, y), (name, what)


def test_mid_size_step_on_cooperative_chains(chain_variant, monkeypatch):
    """Launches between one and one and a half rounds of the one-wave-per-tile kernels (32 769 .. 49 152 rays) take the
    two-tile cooperative chains by default (csrc/r2l_common.h r2l_use_coopf): same step within rounding, ragged size."""
    if chain_variant != "main":
        pytest.skip("one comparison")
    from model.nerf_raybased import PointSampler
    from r2l_amd import _lib
    from r2l_amd.train_step import R2LTrainer
    sd = O.make_state_dict(n_block=43, seed=0)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    n = 40000 - 3
    g = torch.Generator().manual_seed(6)
    o = (torch.randn(n, 3, generator=g) * 1.5).cuda(); d = torch.randn(n, 3, generator=g).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda(); tr = torch.rand(n, 16, generator=g).cuda()

    def run():
        m = build_model(sd, 43)
        t = R2LTrainer(m, ps)
        rgb = t.forward_backward(o, d, tgt, perturb=1.0, t_rand=tr)
        with torch.no_grad():
            plain = m.forward_rays(o, d, ps)
        return t.loss_out.clone(), rgb.clone(), t.grads.clone(), plain.clone()

    from r2l_amd import engine
    from tests.conftest import use_family
    main = run()
    assert _lib.load().r2l_coop_tiles_for_cfg(n, 43, _lib.make_config(**engine.DEFAULT_CONFIG)) == 0
    use_family(monkeypatch)  # AUTO: the library's own choice for this size
    assert _lib.load().r2l_coop_tiles_for_cfg(n, 43, _lib.make_config(**engine.DEFAULT_CONFIG)) == 2
    auto = run()
    assert abs(auto[0][0].item() - main[0][0].item()) < 1e-6
    assert (auto[1] - main[1]).abs().max().item() < 2e-6 and (auto[3] - main[3]).abs().max().item() < 2e-6
    assert torch.nn.functional.cosine_similarity(auto[2], main[2], dim=0).item() > 0.9999


def test_forward_backward_is_graph_capturable(chain_variant):
    """INTEGRATION.md "graph-capturable": forward + stash + backward enqueue kernels and async memsets only (no host sync, no
    allocation inside the library), so a hipGraph captured around them replays to bit-identical gradients — also after the
    inputs behind the captured pointers change."""
    if chain_variant not in ("main", "coopf", "main-exact"):
        pytest.skip("three families")
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    sd = O.make_state_dict(n_block=5, seed=9)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    n = 4096 if chain_variant == "coopf" else 40000
    g = torch.Generator().manual_seed(4)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda()
    u = torch.rand(n, 16, generator=g).cuda()
    tr = R2LTrainer(build_model(sd, 5), ps, dw_mode="exact" if chain_variant.endswith("exact") else None)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up on the capture stream: lazy packs, buffer growth, occupancy queries
        tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager, loss = tr.grads.clone(), tr.loss_out.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    tr.grads.fill_(7.)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(tr.grads, eager) and torch.equal(tr.loss_out, loss)
    tgt.copy_(torch.rand(n, 3, generator=g))  # new targets behind the same pointer
    graph.replay()
    torch.cuda.synchronize()
    replayed = tr.grads.clone()
    tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    torch.cuda.synchronize()
    assert torch.equal(tr.grads, replayed) and not torch.equal(replayed, eager)
