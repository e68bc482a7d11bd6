"""Pin the CPU oracle (oracle/r2l_oracle.py) against outputs of the reference itself (tests/golden/*.npz, made by
tests/golden/gen_golden.py from /root/reference).  CPU only; runs in seconds."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

T = torch.from_numpy


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_point_sampler_and_get_rays(golden_dir):
    g = load(golden_dir, "sampler.npz")
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    dirs = O.pixel_dirs(H, W, focal)
    z = O.z_vals(16, float(g["near"]), float(g["far"]))
    assert np.array_equal(z.numpy(), g["z_vals"])
    assert np.array_equal(dirs[:2, :5].numpy(), g["dirs_corner"])
    rows = g["rows"]
    for p, c2w in enumerate(g["poses"]):
        pts = O.sample_test(dirs, z, T(c2w))
        assert np.array_equal(pts[rows].numpy(), g["pts"][p])  # bit-exact: same op sequence
        ro, rd = O.rays_from_pose(dirs, T(c2w))
        assert np.array_equal(ro[rows].numpy(), g["rays_o"][p])
        assert np.array_equal(rd[rows].numpy(), g["rays_d"][p])
        th, ph, r = g["pose_spherical_args"][p]
        np.testing.assert_allclose(O.pose_spherical(th, ph, r)[:3, :4], c2w, atol=1e-6)


def test_sample_train(golden_dir):
    g = load(golden_dir, "sample_train.npz")
    z = O.z_vals(16, 2., 6.)
    o, d = T(g["rays_o"]), T(g["rays_d"])
    assert np.array_equal(O.sample_train(o, d, z, 0.).numpy(), g["pts_perturb0"])
    assert np.array_equal(O.sample_train(o, d, z, 1., T(g["t_rand"])).numpy(), g["pts_perturb1"])


def test_embedders(golden_dir):
    g = load(golden_dir, "embed.npz")
    assert np.array_equal(O.positional_embed(T(g["pts"]), 10).numpy(), g["emb"])
    assert np.array_equal(O.nerf_embed(T(g["x3"]), 10).numpy(), g["nerf_emb10"])
    assert np.array_equal(O.nerf_embed(T(g["x3"]), 4).numpy(), g["nerf_emb4"])


def test_r2l_w256d88_forward_loss_grads(golden_dir):
    g = load(golden_dir, "r2l_w256d88.npz")
    sd = O.make_state_dict(n_block=43, seed=0)
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    # the replayed constructor order reproduces the reference's seeded weights exactly
    np.testing.assert_allclose([v.double().sum().item() for v in sd.values()], g["param_sums"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([v.double().abs().sum().item() for v in sd.values()], g["param_abs_sums"], rtol=1e-12)
    assert O.flatten_state_dict(sd).numel() == 5917187
    z = O.z_vals(16, 2., 6.)
    emb = O.positional_embed(O.sample_train(T(g["rays_o"]), T(g["rays_d"]), z, 0.), 10)
    loss, rgb, grads = O.r2l_loss_and_grads(sd, emb, T(g["target"]))
    np.testing.assert_allclose(rgb.numpy(), g["rgb"], atol=2e-6)
    assert abs(loss.item() - float(g["loss"])) < 1e-7
    assert abs(O.mse2psnr(loss).item() - float(g["psnr"])) < 1e-4
    gn = np.array([v.norm().item() for v in grads.values()])
    np.testing.assert_allclose(gn, g["grad_norms"], rtol=2e-4)
    np.testing.assert_allclose(grads["tail.0.weight"].numpy(), g["grad_tail_w"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(grads["head.0.bias"].numpy(), g["grad_head_b"], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(grads["body.20.body.0.weight"][:4].numpy(), g["grad_body20_w0_rows"], rtol=1e-3,
                               atol=1e-8)


def test_r2l_w32d6_full_grads_and_adam(golden_dir):
    g = load(golden_dir, "r2l_w32d6.npz")
    g256 = load(golden_dir, "r2l_w256d88.npz")
    sd = {k[3:]: T(g[k]) for k in g.files if k.startswith("p0/")}
    z = O.z_vals(16, 2., 6.)
    emb = O.positional_embed(O.sample_train(T(g256["rays_o"]), T(g256["rays_d"]), z, 0.), 10)
    target = T(g256["target"])
    loss, rgb, grads = O.r2l_loss_and_grads(sd, emb, target)
    np.testing.assert_allclose(rgb.numpy(), g["rgb"], atol=1e-6)
    for k, v in grads.items():
        np.testing.assert_allclose(v.numpy(), g["g0/" + k], rtol=1e-4, atol=1e-9, err_msg=k)
    # 3 Adam steps with the warm-up schedule (main.py:1181-1195, 465-467)
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v_ = {k: torch.zeros_like(v) for k, v in sd.items()}
    for step in (1, 2, 3):
        lr = O.lr_schedule(step, 5e-4, 500, "0.0001,200")
        assert abs(lr - g["lrs"][step - 1]) < 1e-12
        loss, _, grads = O.r2l_loss_and_grads(sd, emb, target)
        assert abs(loss.item() - g["adam_losses"][step - 1]) < 1e-6
        for k in sd:
            sd[k], m[k], v_[k] = O.adam_step(sd[k], grads[k], m[k], v_[k], step, lr)
    for k in sd:
        np.testing.assert_allclose(sd[k].numpy(), g["p3/" + k], rtol=0, atol=2e-6, err_msg=k)


def test_hard_rays(golden_dir):
    g = load(golden_dir, "hard_rays.npz")
    idx = O.hard_ray_indices(T(g["rgb"]), T(g["target"]), 51)
    assert np.array_equal(idx.numpy(), g["hard_indices"])


@pytest.mark.parametrize("S", [64, 192])
@pytest.mark.parametrize("wb", [False, True])
def test_raw2outputs(golden_dir, S, wb):
    g = load(golden_dir, "raw2outputs.npz")
    outs = O.raw2outputs(T(g["S%d/raw" % S]), T(g["S%d/z" % S]), T(g["S%d/d" % S]), None, wb)
    for name, t in zip(("rgb", "disp", "acc", "weights", "depth"), outs):
        ref = g["S%d_wb%d/%s" % (S, int(wb), name)]
        assert np.array_equal(t.numpy(), ref, equal_nan=True), name
    # the documented edge cases: empty ray -> acc 0, disp NaN, white background 1
    acc, disp, rgb = outs[2].numpy(), outs[1].numpy(), outs[0].numpy()
    assert acc[0] == 0 and np.isnan(disp[0])
    if wb:
        assert np.all(rgb[0] == 1.0)


def test_sample_pdf(golden_dir):
    g = load(golden_dir, "sample_pdf.npz")
    bins, w = T(g["bins"]), T(g["weights"])
    assert np.array_equal(O.sample_pdf(bins, w, 128, det=True).numpy(), g["samples_det"])
    assert np.array_equal(O.sample_pdf(bins, w, 128, det=False, u=T(g["u_pytest"])).numpy(), g["samples_pytest"])


def test_render_rays(golden_dir):
    g = load(golden_dir, "render_rays.npz")
    coarse, fine = O.make_teacher_state_dicts(11, 2, alpha_bias=0.5)
    sums = [v.double().sum().item() for sd in (coarse, fine) for v in sd.values()]
    np.testing.assert_allclose(sums, g["teacher_param_sums"], rtol=0, atol=1e-6)
    rb = T(g["ray_batch"])
    with torch.no_grad():
        det = O.render_rays(rb, coarse, fine, perturb=0.)
        rnd = O.render_rays(rb, coarse, fine, perturb=1., t_rand=T(g["pytest/t_rand"]), u=T(g["pytest/u"]))
    for tag, ret in (("det", det), ("pytest", rnd)):
        for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"):
            np.testing.assert_allclose(ret[k].numpy(), g[tag + "/" + k], rtol=2e-5, atol=2e-6, err_msg=tag + k)


def test_ssim_matches_reference(golden_dir):
    """oracle.ssim and the CPU branch of r2l_amd.metrics.ssim against utils/ssim_torch.py outputs (gen_golden_ssim.py)."""
    from r2l_amd import metrics
    g = np.load(os.path.join(golden_dir, "ssim.npz"))
    np.testing.assert_array_equal(O.ssim_window().numpy(), g["window"])
    np.testing.assert_array_equal(metrics._ssim_window().numpy(), g["window"])
    for tag in "abc":
        pred, gt = torch.from_numpy(g["pred_" + tag]), torch.from_numpy(g["gt_" + tag])
        for fn in (O.ssim, metrics.ssim):
            assert abs(fn(pred, gt).item() - float(g["ssim_" + tag])) < 2e-6
            assert abs(fn(gt, gt).item() - 1.0) < 2e-6
