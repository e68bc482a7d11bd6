"""Diagnostic (run by hand: pytest tools/diag_grad.py -s -q): gradient error of every kernel family against fp64 autograd, on
default-init and on trained weights, next to the error of the reference's own fp32 path."""
import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O
from tests.test_driver_gpu import trained_student  # noqa: F401  (fixture)
from tests.test_forward_gpu import build_model

FAMILIES = {"fp16 trio": {"R2L_FORCE_VARIANT": "main"}, "bf16x3 trio": {"R2L_FORCE_VARIANT": "main", "R2L_NO_DW2": "1"},
            "fp32 mfma": {"R2L_FORCE_VARIANT": "main", "R2L_NO_FWD3": "1"}, "coop16": {"R2L_FORCE_VARIANT": "coop16"}}


def errs(g, t):
    return ((g - t).abs().max() / t.abs().max()).item(), ((g - t).norm() / t.norm()).item()


def report(tag, sd, net, ps, rays, monkeypatch):
    from r2l_amd.train_step import R2LTrainer
    emb = O.positional_embed(O.sample_train(rays[:, :3].cpu(), rays[:, 3:6].cpu(), O.z_vals(16, 2., 6.), 0.), 10)
    tgt = rays[:, 6:].cpu()
    _, _, g32 = O.r2l_loss_and_grads(sd, emb, tgt)
    _, rgb64, g64 = O.r2l_loss_and_grads({k: v.double() for k, v in sd.items()}, emb.double(), tgt.double())
    # activation statistics of the net
    _, xs, ts = O.r2l_forward(sd, emb, return_acts=True)
    print("%s: max |x| %.1f, max |t| %.1f, mean |rgb - target| %.2e" % (tag, max(x.abs().max().item() for x in xs),
                                                                         max(t.abs().max().item() for t in ts),
                                                                         (rgb64.float() - tgt).abs().mean().item()))
    rows = {"fp32 oracle": {k: v.double() for k, v in g32.items()}}
    for fam, env in FAMILIES.items():
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            tr = R2LTrainer(net, ps)
            tr.forward_backward(rays[:, :3].contiguous(), rays[:, 3:6].contiguous(), rays[:, 6:].contiguous())
            flat, off, d = tr.grads.cpu().double(), 0, {}
            for k, v in sd.items():
                d[k] = flat[off:off + v.numel()].view(v.shape)
                off += v.numel()
            rows[fam] = d
    # how often would a forward error of 1e-6 flip a ReLU mask?  and how many does fp32 flip against fp64?
    _, xs64, ts64 = O.r2l_forward({k: v.double() for k, v in sd.items()}, emb.double(), return_acts=True)
    pre = []  # pre-activations are not returned: recompute t_pre for block inputs
    import torch.nn.functional as F
    near, flips, tot = 0, 0, 0
    for b in range(len(ts)):
        t32 = F.linear(xs[b], sd["body.%d.body.0.weight" % b], sd["body.%d.body.0.bias" % b])
        t64 = F.linear(xs64[b], sd["body.%d.body.0.weight" % b].double(), sd["body.%d.body.0.bias" % b].double())
        near += (t64.abs() < 1e-6).sum().item()
        flips += ((t32 > 0) != (t64 > 0)).sum().item()
        tot += t64.numel()
    print("  pre-activations: %d of %d within 1e-6 of zero (%.2e); fp32 oracle flips %d masks against fp64" % (near, tot, near / tot, flips))
    fams = [f for f in rows if f != "fp32 oracle"]
    bw = [k for k in sd if ".body." in k and k.endswith("weight")]
    for i in range(len(fams)):
        for j in range(i + 1, len(fams)):
            e = np.median([((rows[fams[i]][k] - rows[fams[j]][k]).norm() / g64[k].norm()).item() for k in bw])
            print("  body W, %s vs %s: median rel L2 distance %.2e" % (fams[i], fams[j], e))
    prof = [((rows["fp32 mfma"]["body.%d.body.0.weight" % b] - g64["body.%d.body.0.weight" % b]).norm() /
             g64["body.%d.body.0.weight" % b].norm()).item() for b in range(0, len(ts), 6)]
    print("  fp32 mfma, body.b.body.0.weight rel L2 error for b = 0, 6, ..: " + " ".join("%.1e" % v for v in prof))
    for name, d in rows.items():
        for grp, sel in (("head", lambda k: k.startswith("head")), ("body W", lambda k: ".body." in k and k.endswith("weight")),
                         ("body b", lambda k: ".body." in k and k.endswith("bias")), ("tail", lambda k: k.startswith("tail"))):
            e = np.array([errs(d[k], g64[k]) for k in sd if sel(k)])
            print("  %-12s %-7s max-norm: median %.2e worst %.2e | rel L2: median %.2e worst %.2e" %
                  (name, grp, np.median(e[:, 0]), e[:, 0].max(), np.median(e[:, 1]), e[:, 1].max()))


@pytest.mark.gpu
def test_diag(trained_student, monkeypatch):
    from model.nerf_raybased import PointSampler
    net, ps, train = trained_student["net"], trained_student["ps"], trained_student["train"]
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    rays = train[:4096].contiguous()
    g = torch.Generator().manual_seed(0)
    rnd = rays.clone()
    rnd[:, 6:] = torch.rand(rays.shape[0], 3, generator=g).cuda()
    report("trained weights, random targets", sd, net, ps, rnd, monkeypatch)
    sd0 = O.make_state_dict(n_block=43, seed=0)
    net0 = build_model(sd0, 43)
    report("default init, teacher targets", sd0, net0, PointSampler(64, 64, 80., 16, 2., 6.), rays, monkeypatch)
