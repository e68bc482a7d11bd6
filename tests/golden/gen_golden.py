"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself (dev container only).

    python tests/golden/gen_golden.py            # needs /root/reference; writes tests/golden/*.npz, ckpt_w32d6.tar

The reference never travels to the GPU box: only these small data files (inputs + the reference's outputs) are
committed.  The reference imports cleanly for model/nerf_raybased.py and utils/run_nerf_raybased_helpers.py;
main.py (render_rays / raw2outputs / save_ckpt) additionally needs stub modules for packages that are not
installed here (imageio, cv2, lpips, smilelogging, configargparse via `option`) — SURVEY.md §8(c).
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    assert os.path.isdir(REF), "reference not mounted"
    # the repo root also has `model`/`utils` packages (the drop-in mirror): make sure the reference wins here
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(OUT, "..", ".."))]
    sys.path.insert(0, REF)
    import model.nerf_raybased as rm
    import utils.run_nerf_raybased_helpers as rh
    assert rm.__file__.startswith(REF) and rh.__file__.startswith(REF)
    torch.autograd.set_detect_anomaly(False)

    class _Logger:
        def __init__(self, *a, **k):
            self.ExpID, self.log_path, self.weights_path, self.gen_img_path = "golden", "/tmp", "/tmp", "/tmp"
            self.log_printer = self

        def info(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            pass

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def to(self, *a, **k):
            return self

    _stub("imageio")
    _stub("cv2")
    _stub("lpips", LPIPS=_Dummy)
    _stub("smilelogging", Logger=_Logger)
    _stub("smilelogging.utils", Timer=_Dummy, LossLine=_Dummy, get_n_params_=lambda m: 0, get_n_flops_=lambda *a, **k: 0,
          AverageMeter=_Dummy, ProgressMeter=_Dummy, check_path=lambda p: p)
    trial = argparse.Namespace(ON=True, body_arch="resmlp", inact="relu", outact="none", res_scale=1., n_learnable=2,
                               n_block=-1, near=-1, far=-1)
    args = argparse.Namespace(netdepth=88, netwidth=256, layerwise_netwidths="", act="relu", linear_tail=False,
                              use_residual=True, trial=trial, lpips_net="alex", model_name="R2L",
                              given_render_path_rays="", N_importance=0)
    _stub("option", args=args)
    import main as rmain
    assert rmain.__file__.startswith(REF)
    return rm, rh, rmain, args


def r2l_args(args, D, W):
    a = argparse.Namespace(**vars(args))
    a.netdepth, a.netwidth = D, W
    return a


def main():
    rm, rh, rmain, args = import_reference()
    dev = torch.device("cpu")
    H = W = 400
    focal = 555.5555155968841
    near, far = 2., 6.
    from dataset.load_blender import pose_spherical  # reference pose utility (stubs satisfy its imports)

    # ---- 1. PointSampler / get_rays ---------------------------------------------------------------------------
    ps = rm.PointSampler(H, W, focal, 16, near, far)
    rng = np.random.RandomState(0)
    rows = np.sort(rng.choice(H * W, 64, replace=False))
    poses = torch.stack([pose_spherical(t, -30., 4.) for t in (-180., -63., 117.)])[:, :3, :4]
    pts_rows, o_rows, d_rows = [], [], []
    for c2w in poses:
        pts_rows.append(ps.sample_test(c2w)[rows].numpy())
        ro, rd = rh.get_rays(H, W, focal, c2w)
        o_rows.append(ro.reshape(-1, 3)[rows].numpy())
        d_rows.append(rd.reshape(-1, 3)[rows].numpy())
    np.savez(os.path.join(OUT, "sampler.npz"), H=H, W=W, focal=focal, near=near, far=far, z_vals=ps.z_vals.numpy(),
             poses=poses.numpy(), rows=rows, pts=np.stack(pts_rows), rays_o=np.stack(o_rows), rays_d=np.stack(d_rows),
             dirs_corner=ps.dirs[:2, :5].numpy(), pose_spherical_args=np.array([[-180., -30., 4.], [-63., -30., 4.],
                                                                                  [117., -30., 4.]]))

    # ---- 2. sample_train ---------------------------------------------------------------------------------------
    ro, rd = rh.get_rays(H, W, focal, poses[1])
    o64 = ro.reshape(-1, 3)[rows].contiguous()
    d64 = rd.reshape(-1, 3)[rows].contiguous()
    pts0 = ps.sample_train(o64, d64, perturb=0.)
    torch.manual_seed(123)
    U = torch.rand(64, 16)
    torch.manual_seed(123)
    pts1 = ps.sample_train(o64, d64, perturb=1.)
    np.savez(os.path.join(OUT, "sample_train.npz"), rays_o=o64.numpy(), rays_d=d64.numpy(), t_rand=U.numpy(),
             pts_perturb0=pts0.numpy(), pts_perturb1=pts1.numpy())

    # ---- 3. embedders ------------------------------------------------------------------------------------------
    pe = rm.PositionalEmbedder(L=10)
    emb32 = pe(pts0[:32])
    e10, d10 = rh.get_embedder(10, 0)
    e4, d4 = rh.get_embedder(4, 0)
    x3 = pts0[:64].reshape(-1, 3)[:64].contiguous()
    np.savez(os.path.join(OUT, "embed.npz"), pts=pts0[:32].numpy(), emb=emb32.numpy(), x3=x3.numpy(),
             nerf_emb10=e10(x3).numpy(), nerf_emb4=e4(x3).numpy())

    # ---- 4. NeRF_v3_2 W256 D88, seed 0 -------------------------------------------------------------------------
    torch.manual_seed(0)
    net = rm.NeRF_v3_2(r2l_args(args, 88, 256), 1008, 3)
    sd = net.state_dict()
    rng = np.random.RandomState(1)
    rows256 = np.sort(rng.choice(H * W, 256, replace=False))
    o256 = ro.reshape(-1, 3)[rows256].contiguous()
    d256 = rd.reshape(-1, 3)[rows256].contiguous()
    target = torch.from_numpy(rng.rand(256, 3).astype(np.float32))
    emb = pe(ps.sample_train(o256, d256, perturb=0.))
    rgb = net(emb)
    loss = rh.img2mse(rgb, target)
    net.zero_grad()
    loss.backward()
    gn = np.array([p.grad.norm().item() for p in net.parameters()], dtype=np.float64)
    named = dict(net.named_parameters())
    np.savez(os.path.join(OUT, "r2l_w256d88.npz"), rays_o=o256.numpy(), rays_d=d256.numpy(), target=target.numpy(),
             rgb=rgb.detach().numpy(), loss=loss.item(), psnr=rh.mse2psnr(loss.detach()).item(), grad_norms=gn,
             param_sums=np.array([v.double().sum().item() for v in sd.values()]),
             param_abs_sums=np.array([v.double().abs().sum().item() for v in sd.values()]),
             keys=np.array(list(sd.keys())),
             grad_tail_w=named["tail.0.weight"].grad.numpy(), grad_tail_b=named["tail.0.bias"].grad.numpy(),
             grad_head_b=named["head.0.bias"].grad.numpy(), grad_body0_b0=named["body.0.body.0.bias"].grad.numpy(),
             grad_body42_b2=named["body.42.body.2.bias"].grad.numpy(),
             grad_body20_w0_rows=named["body.20.body.0.weight"].grad[:4].numpy(),
             grad_head_w_rows=named["head.0.weight"].grad[:2].numpy())

    # ---- 5/6. NeRF_v3_2 W32 D6: full tensors, grads, 3 Adam steps with the warm-up LR schedule --------------------
    torch.manual_seed(7)
    small = rm.NeRF_v3_2(r2l_args(args, 6, 32), 1008, 3)
    sd_small0 = {k: v.clone() for k, v in small.state_dict().items()}
    rgb_s = small(emb)
    loss_s = rh.img2mse(rgb_s, target)
    small.zero_grad()
    loss_s.backward()
    grads_small = {k: p.grad.clone() for k, p in small.named_parameters()}
    opt = torch.optim.Adam(params=list(small.parameters()), lr=5e-4, betas=(0.9, 0.999))
    lrs, losses = [], []
    start_lr, end_iter, lrate, decay_steps = 0.0001, 200., 5e-4, 500 * 1000
    for step in (1, 2, 3):
        new_lr = (lrate - start_lr) / end_iter * step + start_lr if step < end_iter else lrate * (0.1**(
            (step - end_iter) / decay_steps))
        for g in opt.param_groups:
            g["lr"] = new_lr
        l_ = rh.img2mse(small(emb), target)
        opt.zero_grad()
        l_.backward()
        opt.step()
        lrs.append(new_lr)
        losses.append(l_.item())
    out = {"emb_rows": np.arange(256), "rgb": rgb_s.detach().numpy(), "loss": loss_s.item(), "lrs": np.array(lrs),
           "adam_losses": np.array(losses)}
    for k, v in sd_small0.items():
        out["p0/" + k] = v.numpy()
    for k, v in grads_small.items():
        out["g0/" + k] = v.numpy()
    for k, v in small.state_dict().items():
        out["p3/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "r2l_w32d6.npz"), **out)

    # ---- 9b. a reference-written checkpoint of the W32D6 net (main.py:1516-1542 save_ckpt) ----------------------
    rmain.global_step = 3
    rmain.args.model_name = "R2L"
    rmain.args.N_importance = 0

    class _L:
        weights_path = OUT

    rmain.logger.weights_path = OUT
    path = rmain.save_ckpt("ckpt_w32d6.tar", {"network_fn": small}, opt, 12.5, 2)
    assert os.path.exists(path)

    # ---- 10. hard-ray selection (main.py:1411-1414) ---------------------------------------------------------------
    _, indices = torch.sort(torch.mean((rgb.detach() - target)**2, dim=1))
    np.savez(os.path.join(OUT, "hard_rays.npz"), rgb=rgb.detach().numpy(), target=target.numpy(),
             hard_indices=indices[-51:].numpy())

    # ---- 7. raw2outputs (main.py:556-621; identical copies in create_data.py:335-402, nerf_raybased.py:226-295) ---
    r2o = {}
    rng = np.random.RandomState(2)
    for S in (64, 192):
        R = 48
        raw = (rng.randn(R, S, 4) * 2.0).astype(np.float32)
        raw[:, :, 3] *= 4.0  # wide sigma range incl. negatives
        raw[0, :, 3] = -1.0  # empty ray: acc 0, disp NaN
        raw[1, 0, 3] = 1e4  # opaque first sample
        raw[2, :, 3] = 50.0  # saturating alpha everywhere
        z = np.sort(rng.uniform(2., 6., size=(R, S)).astype(np.float32), axis=-1)
        z[3] = np.linspace(2., 6., S, dtype=np.float32)
        d = rng.randn(R, 3).astype(np.float32)
        for wb in (False, True):
            outs = rmain.raw2outputs(torch.from_numpy(raw), torch.from_numpy(z), torch.from_numpy(d), 0, wb,
                                     pytest=False)
            key = "S%d_wb%d" % (S, int(wb))
            for name, t in zip(("rgb", "disp", "acc", "weights", "depth"), outs):
                r2o[key + "/" + name] = t.numpy()
        r2o["S%d/raw" % S], r2o["S%d/z" % S], r2o["S%d/d" % S] = raw, z, d
    np.savez_compressed(os.path.join(OUT, "raw2outputs.npz"), **r2o)

    # ---- 8. sample_pdf (helpers:283-330) ---------------------------------------------------------------------------
    rng = np.random.RandomState(3)
    R = 40
    zc = np.sort(rng.uniform(2., 6., size=(R, 64)).astype(np.float32), axis=-1)
    bins = torch.from_numpy(.5 * (zc[:, 1:] + zc[:, :-1]))
    wts = rng.rand(R, 62).astype(np.float32)
    wts[0] = 0.0  # all-zero weights -> uniform pdf
    wts[1, :] = 0.0
    wts[1, 30] = 5.0  # single spike -> zero-width cdf steps elsewhere
    wts = torch.from_numpy(wts)
    s_det = rh.sample_pdf(bins, wts, 128, det=True, pytest=False)
    s_rnd = rh.sample_pdf(bins, wts, 128, det=False, pytest=True)
    np.random.seed(0)
    u_pytest = np.random.rand(R, 128)
    np.savez_compressed(os.path.join(OUT, "sample_pdf.npz"), bins=bins.numpy(), weights=wts.numpy(),
                        samples_det=s_det.numpy(), samples_pytest=s_rnd.numpy(),
                        u_pytest=torch.Tensor(u_pytest).numpy())

    # ---- 9. render_rays with a seeded teacher (main.py:624-756) ---------------------------------------------------
    torch.manual_seed(11)
    coarse = rm.NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
    fine = rm.NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
    with torch.no_grad():  # bias the density head so rays are neither empty nor opaque
        for n_ in (coarse, fine):
            n_.alpha_linear.bias.add_(0.5)
    embed_fn, _ = rh.get_embedder(10, 0)
    embeddirs_fn, _ = rh.get_embedder(4, 0)
    qfn = lambda inputs, viewdirs, network_fn: rmain.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                 embeddirs_fn=embeddirs_fn, netchunk=65536)
    R = 96
    rr_rows = np.sort(np.random.RandomState(4).choice(H * W, R, replace=False))
    o_ = ro.reshape(-1, 3)[rr_rows]
    d_ = rd.reshape(-1, 3)[rr_rows]
    vd = d_ / torch.norm(d_, dim=-1, keepdim=True)
    ray_batch = torch.cat([o_, d_, near * torch.ones_like(d_[..., :1]), far * torch.ones_like(d_[..., :1]), vd], -1)
    rr = {"ray_batch": ray_batch.numpy(), "teacher_param_sums": np.array(
        [v.double().sum().item() for n_ in (coarse, fine) for v in n_.state_dict().values()])}
    with torch.no_grad():
        for tag, kw in (("det", dict(perturb=0., pytest=False)), ("pytest", dict(perturb=1., pytest=True))):
            ret = rmain.render_rays(ray_batch, coarse, qfn, 64, retraw=True, N_importance=128, network_fine=fine,
                                    white_bkgd=True, raw_noise_std=0., **kw)
            for k, v in ret.items():
                if k != "raw":
                    rr[tag + "/" + k] = v.numpy()
    np.random.seed(0)
    rr["pytest/t_rand"] = torch.Tensor(np.random.rand(R, 64)).numpy()
    np.random.seed(0)
    rr["pytest/u"] = torch.Tensor(np.random.rand(R, 128)).numpy()
    np.savez_compressed(os.path.join(OUT, "render_rays.npz"), **rr)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print("  %-22s %8d B" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
