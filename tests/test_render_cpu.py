"""CPU plumbing of the mirrored render layer (r2l_amd/render.py, torch-op branch) against the reference goldens."""
import os

import numpy as np
import torch

from oracle import r2l_oracle as O

T = torch.from_numpy


def test_raw2outputs_cpu(golden_dir):
    from r2l_amd.render import raw2outputs
    g = np.load(os.path.join(golden_dir, "raw2outputs.npz"))
    for S in (64, 192):
        for wb in (False, True):
            outs = raw2outputs(T(g["S%d/raw" % S]), T(g["S%d/z" % S]), T(g["S%d/d" % S]), 0, wb)
            for name, t in zip(("rgb", "disp", "acc", "weights", "depth"), outs):
                assert np.array_equal(t.numpy(), g["S%d_wb%d/%s" % (S, int(wb), name)], equal_nan=True), name


def test_sample_pdf_cpu(golden_dir):
    from r2l_amd.render import sample_pdf
    g = np.load(os.path.join(golden_dir, "sample_pdf.npz"))
    bins, w = T(g["bins"]), T(g["weights"])
    np.testing.assert_allclose(sample_pdf(bins, w, 128, det=True).numpy(), g["samples_det"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(sample_pdf(bins, w, 128, det=False, pytest=True).numpy(), g["samples_pytest"], rtol=0,
                               atol=1e-6)


def test_get_rays_and_embedder(golden_dir):
    from r2l_amd.render import get_embedder, get_rays
    g = np.load(os.path.join(golden_dir, "sampler.npz"))
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    for p, c2w in enumerate(g["poses"]):
        ro, rd = get_rays(H, W, focal, T(c2w))
        assert np.array_equal(ro.reshape(-1, 3)[g["rows"]].numpy(), g["rays_o"][p])
        assert np.array_equal(rd.reshape(-1, 3)[g["rows"]].numpy(), g["rays_d"][p])
    e = np.load(os.path.join(golden_dir, "embed.npz"))
    f10, d10 = get_embedder(10)
    f4, d4 = get_embedder(4)
    assert d10 == 63 and d4 == 27
    assert np.array_equal(f10(T(e["x3"])).numpy(), e["nerf_emb10"])
    assert np.array_equal(f4(T(e["x3"])).numpy(), e["nerf_emb4"])


def test_render_rays_cpu(golden_dir):
    from model.nerf_raybased import NeRF
    from r2l_amd.render import get_embedder, render_rays, run_network
    g = np.load(os.path.join(golden_dir, "render_rays.npz"))
    nets = []
    for sd in O.make_teacher_state_dicts(11, 2, alpha_bias=0.5):
        m = NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
        m.load_state_dict(sd)
        nets.append(m)
    e10, _ = get_embedder(10)
    e4, _ = get_embedder(4)
    qfn = lambda pts, vd, fn: run_network(pts, vd, fn, embed_fn=e10, embeddirs_fn=e4)
    with torch.no_grad():
        det = render_rays(T(g["ray_batch"]), nets[0], qfn, 64, N_importance=128, network_fine=nets[1], white_bkgd=True,
                          perturb=0.)
        rnd = render_rays(T(g["ray_batch"]), nets[0], qfn, 64, N_importance=128, network_fine=nets[1], white_bkgd=True,
                          perturb=1., pytest=True)
    for tag, ret in (("det", det), ("pytest", rnd)):
        for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "z_std"):
            np.testing.assert_allclose(ret[k].numpy(), g[tag + "/" + k], rtol=2e-5, atol=2e-6, err_msg=tag + k)
