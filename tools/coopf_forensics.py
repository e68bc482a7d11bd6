"""Forensics of the co-residency fault of the one-tile cooperative training forward (csrc/r2l_coopf_fwd.hip), WITHOUT
touching the kernel: run the faulty configuration (a reproducer build of the forward, see tools/coopf_coresidency.py:
R2L_LIB_PATH=tools/_bin/share_r2/libr2l_hip.so; one tile per workgroup, more tiles than CUs), and for every ray whose rgb differs from the two-tile kernel's reconstruct on the host, from the kernel's own y = x_n + x_0
rows (save_x slot n_block, bit-exact by earlier evidence) and the tail weights, what the tail's partial sums should have
been — then search for the single substitution (stale weight operand, lost FMA terms, ...) that explains the observed
logit.  GPU box only.   python tools/coopf_forensics.py [N=16384] [launches=6]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model
from model.nerf_raybased import PointSampler
from r2l_amd.train_step import R2LTrainer

os.environ["R2L_FORCE_VARIANT"] = "coopf"
NB = 43
sd = O.make_state_dict(n_block=NB, seed=0)
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = torch.Generator().manual_seed(5)
o = (torch.randn(n, 3, generator=g) * 1.5).cuda(); d = torch.randn(n, 3, generator=g).cuda()
tgt = torch.rand(n, 3, generator=g).cuda(); tr = torch.rand(n, 16, generator=g).cuda()


def run(tiles):
    os.environ["R2L_COOPF_TILES"] = tiles
    m = build_model(sd, NB)
    t = R2LTrainer(m, ps)
    rgb = t.forward_backward(o, d, tgt, perturb=1.0, t_rand=tr).clone()
    torch.cuda.synchronize()
    slot = int(t.lib.r2l_stash_slot_floats(n))
    npad = (n + 31) // 32 * 32
    y = t.save_x[NB * slot: NB * slot + npad * 256].view(npad, 256).clone()
    return rgb, y, t


ref_rgb, ref_y, t0 = run("2")
flat = t0.eng.flat.detach().cpu().numpy()
off_w = 1008 * 256 + 256 + 2 * NB * (256 * 256 + 256)
Wt = flat[off_w: off_w + 768].reshape(3, 256).astype(np.float32)
bt = flat[off_w + 768: off_w + 771].astype(np.float32)


def partials(yrow):
    """p[w][h][c] as the kernel accumulates them (fp32 FMA chain over tt, q, e), and the running values after every term."""
    p = np.zeros((4, 2, 3), np.float32)
    trace = {}
    for w in range(4):
        for h in range(2):
            acc = np.zeros(3, np.float64)
            steps = []
            for tt in range(2):
                for q in range(4):
                    for e in range(4):
                        f = 32 * (2 * w + tt) + 8 * q + 4 * h + e
                        for c in range(3):
                            acc[c] = np.float32(np.float64(Wt[c, f]) * np.float64(yrow[f]) + acc[c])  # fma: one rounding
                        steps.append((f, acc.copy()))
            p[w, h] = acc
            trace[(w, h)] = steps
    return p, trace


def logit(x):
    x = np.float64(x)
    return np.log(x) - np.log1p(-x)


hist_j, hist_c, hist_tilepar = collections.Counter(), collections.Counter(), collections.Counter()
shown = 0
for it in range(launches):
    rgb, y, _ = run("1")
    same_y = torch.equal(y, ref_y)
    dd = (rgb - ref_rgb).abs()
    bad = (dd.amax(1) > 0).nonzero().flatten().tolist()
    print("launch %d: %d rays differ; y rows bit-equal to the two-tile run: %s" % (it, len(bad), same_y))
    tiles = sorted(set(r // 32 for r in bad))
    print("   tiles (blockIdx) hit: %d  first %s" % (len(tiles), tiles[:12]))
    for r in bad:
        hist_j[r % 32] += 1
        hist_tilepar[(r // 32) % 2] += 1
        for c in range(3):
            if dd[r, c] > 0:
                hist_c[c] += 1
    for r in bad[:3] if shown < 12 else []:
        shown += 1
        yrow = y[r].cpu().numpy()
        p, trace = partials(yrow)
        tot = ((p[0].sum(0, dtype=np.float32) + p[1].sum(0, dtype=np.float32)) + (p[2].sum(0, dtype=np.float32) + p[3].sum(0, dtype=np.float32))) + bt
        got, want = rgb[r].cpu().numpy(), ref_rgb[r].cpu().numpy()
        lg, lw = logit(got), logit(want)
        print("   ray %d (tile %d, j %d): rgb got %s want %s; logit got %s want %s host %s" % (
            r, r // 32, r % 32, got, want, np.round(lg, 5), np.round(lw, 5), np.round(tot, 5)))
        for c in range(3):
            delta = lg[c] - lw[c]
            if abs(delta) < 1e-5:
                continue
            print("     channel %d: delta logit %.6f" % (c, delta))
            tol = max(2e-4 * abs(delta), 3e-6)
            # (a) a whole per-(wave, half) partial lost or doubled or replaced by another channel's
            for w in range(4):
                for h in range(2):
                    for c2 in range(3):
                        for name, val in (("lost", -p[w, h, c]), ("doubled", p[w, h, c]), ("replaced by channel %d" % c2, p[w, h, c2] - p[w, h, c])):
                            if abs(val - delta) < tol and (name[0] != "r" or c2 != c):
                                print("       (a) partial of wave %d half %d %s: %.6f" % (w, h, name, val))
            # (b) a contiguous run of FMA terms lost (accumulator not forwarded), per (wave, half)
            for (w, h), steps in trace.items():
                terms = [np.float64(Wt[c, f]) * np.float64(yrow[f]) for f, _ in steps]
                cs = np.concatenate([[0.], np.cumsum(terms)])
                for a in range(32):
                    for b in range(a + 1, 33):
                        if abs(-(cs[b] - cs[a]) - delta) < tol:
                            print("       (b) wave %d half %d: terms %d..%d lost (features %d..%d): %.6f" % (w, h, a, b - 1, steps[a][0], steps[b - 1][0], -(cs[b] - cs[a])))
            # (c) ONE term took another weight w' (stale operand): w' = w + delta / y_f; is w' one of the tail weights / 0 / y?
            allw = Wt.reshape(-1)
            for (w, h), steps in trace.items():
                for k, (f, _) in enumerate(steps):
                    if abs(yrow[f]) < 1e-12:
                        continue
                    wp = Wt[c, f] + delta / np.float64(yrow[f])
                    idx = np.nonzero(np.abs(allw - wp) < max(3e-4 * abs(wp), 1e-7))[0]
                    for i in idx[:4]:
                        print("       (c) wave %d half %d term %d (feature %d): weight %.6f read as Wt[%d][%d] = %.6f" % (
                            w, h, k, f, Wt[c, f], i // 256, i % 256, allw[i]))
                    if abs(wp) < 1e-7:
                        print("       (c) wave %d half %d term %d (feature %d): weight read as 0" % (w, h, k, f))
            # (d) a run of terms used the weights of another (wave', tile) position: shifted weight quad
            for (w, h), steps in trace.items():
                for q0 in range(8):
                    fs = [f for f, _ in steps[4 * q0: 4 * q0 + 4]]
                    base = sum(np.float64(Wt[c, f]) * np.float64(yrow[f]) for f in fs)
                    for c2 in range(3):
                        for f0 in range(0, 256, 4):
                            if c2 == c and f0 == fs[0]:
                                continue
                            alt = sum(np.float64(Wt[c2, f0 + e]) * np.float64(yrow[f]) for e, f in enumerate(fs))
                            if abs((alt - base) - delta) < tol:
                                print("       (d) wave %d half %d quad %d (features %d..): weights read from Wt[%d][%d..]: %.6f" % (w, h, q0, fs[0], c2, f0, alt - base))
print("rays j histogram:", sorted(hist_j.items()))
print("channel histogram:", sorted(hist_c.items()), " tile parity:", sorted(hist_tilepar.items()))
