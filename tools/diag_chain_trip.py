"""Diagnosis of a dX-chain-only trip (tests/test_train_gpu.py::test_dx_chain_only_trip_expands_the_fp16_stash): per-tensor
gradient errors of the fallback step against the oracle and against the bf16x3 trio on the same GPU, and the expanded stash
against the oracle's activations.  GPU box: python tools/diag_chain_trip.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import r2l_oracle as O  # noqa: E402
from tests.test_forward_gpu import _body_amplified_net, build_model  # noqa: E402
from model.nerf_raybased import PointSampler  # noqa: E402
from r2l_amd import engine  # noqa: E402
from r2l_amd.train_step import R2LTrainer  # noqa: E402


def rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def split_flat(flat, sd):
    out, off = {}, 0
    for k, v in sd.items():
        out[k] = flat[off:off + v.numel()].view(v.shape)
        off += v.numel()
    return out


def unchunk(slot, n_tiles):
    """chunked fp32 slot [tile][chunk 32][ray 32][8] -> [rays, 256]"""
    return slot[:n_tiles * 8192].view(n_tiles, 32, 32, 8).permute(0, 2, 1, 3).reshape(n_tiles * 32, 256)


sd = _body_amplified_net()
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
gen = torch.Generator().manual_seed(21)
n = 600
o = torch.randn(n, 3, generator=gen) * 1.5
d = torch.randn(n, 3, generator=gen)
tgt = torch.rand(n, 3, generator=gen)
emb = O.positional_embed(O.sample_train(o, d, O.z_vals(16, 2., 6.), 0.), 10)
loss, rgb_ref, gref = O.r2l_loss_and_grads(sd, emb, tgt)
_, xs, ts = O.r2l_forward(sd, emb, return_acts=True)
results = {}
for fam, cfg in (("bf16x3", dict(tiling="main", precision="bf16x3")), ("main", dict(tiling="main")), ("coopf", dict(tiling="coopf"))):
    engine.DEFAULT_CONFIG = dict(cfg)
    m = build_model(sd, 6)
    tr = R2LTrainer(m, ps)
    for it in range(3):
        rgb = tr.forward_backward(o.cuda(), d.cuda(), tgt.cuda())
        torch.cuda.synchronize()
        g = split_flat(tr.grads.cpu(), sd)
        info = tr.range_info()
        print("\n%s step %d: rgb err %.2e | range %s" % (fam, it, (rgb.cpu() - rgb_ref).abs().max().item(),
              {k: (round(v, 4) if isinstance(v, float) else v) for k, v in info.items()}))
        print("   vs oracle: " + "  ".join("%s %.1e" % (k.replace("body.", "b").replace(".weight", ".w").replace(".bias", ".b"), rel(g[k], gref[k])) for k in sd))
        if it == 0:
            gx0 = tr.gx[:n * 256].view(n, 256).cpu().clone()  # row-major head gradient dL/d(head pre-activation)
            if fam == "bf16x3":
                results["gx0"] = gx0
            else:
                dgx = (gx0 - results["gx0"]).abs()
                print("   gx[0] vs bf16x3 trio: max |d| per 32-feature tile %s ; rays with any |d| > 1e-3 of max: %d of %d; features hit: %s"
                      % (["%.1e" % dgx[:, 32 * T:32 * T + 32].max().item() for T in range(8)],
                         int((dgx.max(1)[0] > 1e-3 * results["gx0"].abs().max()).sum()), n,
                         torch.nonzero(dgx.max(0)[0] > 1e-3 * results["gx0"].abs().max()).flatten().tolist()[:40]))
        if fam == "bf16x3" and it == 0:
            results["bf16x3"] = {k: v.clone() for k, v in g.items()}
        elif it == 0:
            print("   vs bf16x3 trio: " + "  ".join("%s %.1e" % (k.replace("body.", "b").replace(".weight", ".w").replace(".bias", ".b"), rel(g[k], results["bf16x3"][k])) for k in sd))
        if it == 0:
            slot = int(tr.lib.r2l_stash_slot_floats(n))
            nt = (n + 31) // 32
            fmt = tr.save_x.view(torch.int32)[6 * slot + nt * 32 * 256].item()
            print("   stash format word after the step: %d (1 = chunked fp32), act scale word %g" % (fmt, tr.save_x[6 * slot + nt * 32 * 256 + 1].item()))
            if fmt == 1:
                for b in range(6):
                    x = unchunk(tr.save_x[b * slot:(b + 1) * slot].cpu(), nt)[:n]
                    t = unchunk(tr.save_t[b * slot:(b + 1) * slot].cpu(), nt)[:n]
                    print("   slot %d: |x - oracle| / max %.2e   |relu(t) - oracle| / max %.2e" % (b, rel(x, xs[b]), rel(t, torch.relu(ts[b]))))
