"""Launches far beyond BASELINE.json's sizes (SURVEY.md §8c "maximum sizes"): the 800x800 Blender frames of the reference's
full-resolution configs, nine of them per launch, and a training step of 393 216 rays whose stash is > 100 GB.  The oracle
cannot run these, so they are checked through size-independent properties: a ray's result does not depend on the launch it
rides in (bit for bit), and the gradient of a batch is the sum of its chunks' gradients."""
import pytest
import torch

from oracle import r2l_oracle as O
from tests.test_forward_gpu import T, build_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model88():
    sd = O.make_state_dict(n_block=43, seed=0)
    return sd, build_model(sd, 43)


def _need_free(gib):
    free, _ = torch.cuda.mem_get_info()
    if free < gib * 2 ** 30:
        pytest.skip("needs %d GiB of free HBM (%.0f free)" % (gib, free / 2 ** 30))


def test_nine_800x800_frames_in_one_launch(model88):
    """5.76 M rays in one launch (r2l_forward_poses_cfg) == nine 640 000-ray launches == 160 000-ray pieces of explicit rays."""
    from model.nerf_raybased import PointSampler
    sd, m = model88
    H = W = 800
    focal = 1111.111
    ps = PointSampler(H, W, focal, 16, 2., 6., device="cuda")
    poses = torch.stack([T(O.pose_spherical(40. * k, -30. + 2 * k, 4.)[:3, :4]) for k in range(9)], 0)
    with torch.no_grad():
        many = m.render_poses(poses, ps)
        assert many.shape == (9, H * W, 3) and torch.isfinite(many).all()
        for k in (0, 4, 8):
            assert torch.equal(m.render_pose(poses[k], ps), many[k]), k
        # the last frame again as explicit rays, in pieces the size of a 400x400 frame
        rays_o, rays_d = O.rays_from_pose(O.pixel_dirs(H, W, focal), poses[8])  # on the CPU, the reference's arithmetic
        rays_o, rays_d = rays_o.contiguous().cuda(), rays_d.contiguous().cuda()
        worst = 0.
        for lo in range(0, H * W, 160000):
            piece = m.forward_rays(rays_o[lo:lo + 160000], rays_d[lo:lo + 160000], ps, perturb=0.)
            worst = max(worst, (piece - many[8, lo:lo + 160000]).abs().max().item())
    # pose mode computes origin and direction in the kernel with the reference's rounding; the explicit rays above come from
    # the oracle's restatement of sample_test: the same numbers up to the last bit of the direction
    print("800x800 frame as explicit rays vs pose mode: max |d rgb|", worst)
    assert worst < 2e-5, worst


def test_two_million_rays_equal_their_pieces(model88):
    """forward_rays: 2 097 152 + 5 rays in one launch, bit for bit the same rays launched in 131 072-ray pieces."""
    from model.nerf_raybased import PointSampler
    sd, m = model88
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6., device="cuda")
    n = 2 ** 21 + 5
    g = torch.Generator(device="cuda").manual_seed(5)
    o = torch.randn(n, 3, generator=g, device="cuda") * 0.3 + torch.tensor([0., 0., 4.], device="cuda")
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g, device="cuda"), dim=-1)
    with torch.no_grad():
        whole = m.forward_rays(o, d, ps, perturb=0.)
        for lo in range(0, 2 ** 21, 131072):  # (the last piece takes the 5 odd rays along: a 5-ray launch of its own would
            hi = n if lo + 131072 == 2 ** 21 else lo + 131072  # go to the cooperative kernels — equal within TOL, not bitwise)
            assert torch.equal(m.forward_rays(o[lo:hi], d[lo:hi], ps, perturb=0.), whole[lo:hi]), lo
    ref = O.r2l_forward(sd, O.positional_embed(O.sample_train(o[-64:].cpu(), d[-64:].cpu(), O.z_vals(16, 2., 6.), 0.), 10))
    assert (whole[-64:].cpu() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("dw_mode", ["fp16", "exact"])
def test_training_step_of_393216_rays_is_the_sum_of_its_chunks(dw_mode):
    """One step on 4 x 98 304 rays (stash + gradient operands > 100 GB, slot offsets far beyond 32 bits): its gradient equals the
    accumulated gradients of the four 98 304-ray chunks (every ray's contribution is the same number either way; only the
    order of the fp32 sums over rays differs), its loss their mean."""
    _need_free(150)
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    from tests.test_train_gpu import rel_err, split_flat
    sd = O.make_state_dict(n_block=43, seed=2)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    chunk, k = 98304, 4
    n = chunk * k
    g = torch.Generator().manual_seed(11)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda()
    u = torch.rand(n, 16, generator=g).cuda()
    tr = R2LTrainer(build_model(sd, 43), ps, dw_mode=dw_mode)
    rgb = tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    whole, loss = tr.grads.clone(), tr.loss_out[0].item()
    del tr
    torch.cuda.empty_cache()
    tr = R2LTrainer(build_model(sd, 43), ps, dw_mode=dw_mode)
    losses = []
    for c in range(k):
        s = slice(c * chunk, (c + 1) * chunk)
        piece = tr.forward_backward(o[s], d[s], tgt[s], perturb=1., t_rand=u[s], zero_grad=(c == 0))
        assert torch.equal(piece, rgb[s]), c  # forward: bit for bit
        losses.append(tr.loss_out[0].item())
    parts = tr.grads / k  # each chunk's gradient carries 1 / chunk; the whole batch's 1 / (k * chunk)
    assert abs(sum(losses) / k - loss) < 1e-6 * max(1., abs(loss))
    a, b = split_flat(whole.cpu(), sd), split_flat(parts.cpu(), sd)
    worst = max(rel_err(a[name], b[name]) for name in sd)
    print("393 216 rays vs 4 chunks: worst per-tensor relative L2", worst)
    assert worst < 2e-5, worst


@pytest.mark.parametrize("n", [8192, 8193, 16384, 16385, 32768, 32769, 49152, 49153])
def test_dispatch_thresholds_forward_and_gradient(n, monkeypatch):
    """AUTO dispatch at the ray counts where the library changes kernel family (one tile per cooperative workgroup -> two at
    8192 / 8193 rays, cooperative -> one wave per tile at 16 384 / 16 385, the 32 769 .. 49 152 window of the two-tile kernels):
    the launch on either side of every threshold agrees with the oracle (first and last 128 rays) and its gradient with the
    one-wave-per-tile family pinned by r2l_config."""
    for k in ("R2L_FORCE_VARIANT", "R2L_COOPF_TILES", "R2L_NO_FWD3", "R2L_NO_FWD2", "R2L_NO_BWD2", "R2L_NO_DW2"):
        monkeypatch.delenv(k, raising=False)
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    from tests.test_train_gpu import rel_err, split_flat
    nb = 3
    sd = O.make_state_dict(n_block=nb, seed=6)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    g = torch.Generator().manual_seed(n)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda()
    u = torch.rand(n, 16, generator=g).cuda()
    m = build_model(sd, nb)
    tr = R2LTrainer(m, ps)
    tiles = tr.lib.r2l_coop_tiles_for_cfg(n, nb, tr.eng._cfg())
    assert tiles == {8192: 1, 8193: 2, 16384: 2, 16385: 0, 32768: 0, 32769: 2, 49152: 2, 49153: 0}[n]
    rgb = tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    auto = tr.grads.clone()
    pick = torch.cat([torch.arange(128), torch.arange(n - 128, n)])
    emb = O.positional_embed(O.sample_train(o[pick].cpu(), d[pick].cpu(), O.z_vals(16, 2., 6.), 1., t_rand=u[pick].cpu()), 10)
    assert (rgb[pick.cuda()].cpu() - O.r2l_forward(sd, emb)).abs().max().item() < 1e-4
    pinned = R2LTrainer(build_model(sd, nb), ps)
    pinned.eng.set_config(tiling="main")
    rgb2 = pinned.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    assert (rgb2 - rgb).abs().max().item() < 2e-6
    a, b = split_flat(auto.cpu(), sd), split_flat(pinned.grads.cpu(), sd)
    worst = max(rel_err(a[k], b[k]) for k in sd)
    assert worst < 2e-4, worst  # two fp16x2 families: same products, different summation order over rays
