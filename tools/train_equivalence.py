"""Does the default fp16 trio (fp16x2 chains + ONE-fp16-product weight-gradient GEMMs on fp16-rounded operands) train like
the fp32-exact families?  The same W256 D88 student (seed 0), the same seeded batches and jitter, the reference's schedule
(lr 5e-4, warm-up 1e-4 -> 5e-4 over 200 iterations, Adam), four kernel families:
  fp16 trio (default) | the same with exact weight gradients (R2L_DW_EXACT=1) | bf16x3 trio (R2L_NO_FWD2/BWD2/DW2=1: products exact to fp32) | fp32 MFMA (R2L_NO_FWD3=1)
on an analytic scene with silhouettes and shading (three lit spheres on white, rays from the r = 4 sphere like
create_data's poses).  Held-out PSNR every 250 iterations.  Trajectories of a chaotic optimisation separate whatever the
rounding (the two fp32-exact families differ only in summation order), so their mutual distance is the yardstick for the
fp16 trio's distance to either.  GPU box:  python tools/train_equivalence.py [iters=1500] [rays=16384]
R2L_EQ_FAMILIES=0,3 keeps a subset of the families; R2L_EQ_PARITY=4096 also compares each TRAINED student's forward on that many
held-out rays with the CPU restatement (fp32 torch ops).  R2L_EQ_SEEDS=0,1,2,3 (round 5): the whole comparison
once per seed — the seed moves the initial weights, the batches and the jitter; the held-out rays stay — and a table of
mean +- sample standard deviation of the final held-out PSNR per family, with each family's distance to the fp32-MFMA mean."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import r2l_oracle as O  # noqa: E402  (seeded W256 D88 state dict only)
from tests.test_forward_gpu import build_model  # noqa: E402
from model.nerf_raybased import PointSampler  # noqa: E402
from r2l_amd.train_step import R2LTrainer, lr_schedule  # noqa: E402

# kernel families through r2l_config (r2l_amd.engine.DEFAULT_CONFIG), not the environment
FAMILIES = {"fp16 trio (default)": {}, "fp16 trio, exact dW": {"dw_mode": "exact"},
            "bf16x3 trio": {"precision": "bf16x3"}, "fp32 MFMA": {"precision": "fp32_mfma"}}


def scene(o, d):
    """rgb[N,3] of rays against three diffuse spheres under one light, white background (first hit wins)."""
    d = torch.nn.functional.normalize(d, dim=-1)
    centers = torch.tensor([[0., 0., 0.], [0.9, 0.3, 0.4], [-0.7, -0.5, 0.6]], device=o.device)
    radii = torch.tensor([0.8, 0.45, 0.5], device=o.device)
    albedo = torch.tensor([[0.9, 0.3, 0.2], [0.2, 0.7, 0.9], [0.3, 0.8, 0.3]], device=o.device)
    light = torch.nn.functional.normalize(torch.tensor([0.5, 0.8, 0.6], device=o.device), dim=0)
    best = torch.full((o.shape[0],), 1e9, device=o.device)
    rgb = torch.ones(o.shape[0], 3, device=o.device)
    for c, r, a in zip(centers, radii, albedo):
        oc = o - c
        b = (oc * d).sum(-1)
        disc = b * b - ((oc * oc).sum(-1) - r * r)
        t = -b - torch.sqrt(disc.clamp_min(0.))
        hit = (disc > 0) & (t > 0) & (t < best)
        n = torch.nn.functional.normalize(o + t[:, None] * d - c, dim=-1)
        shade = 0.25 + 0.75 * (n * light).sum(-1).clamp_min(0.)
        rgb = torch.where(hit[:, None], a[None, :] * shade[:, None], rgb)
        best = torch.where(hit, t, best)
    return rgb


def rays(n, gen):
    """origins on the r = 4 sphere, directions towards a point near the centre (what a 400x400 view of the scene covers);
    drawn ON THE DEVICE (a host-side draw of 16 384 x 22 numbers per step made the loop host-bound: 27 ms per step)"""
    o = torch.randn(n, 3, generator=gen, device="cuda")
    o = 4. * o / o.norm(dim=-1, keepdim=True)
    tgt = (torch.rand(n, 3, generator=gen, device="cuda") - 0.5) * 2.4
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True) * (1. + 0.1 * torch.rand(n, 1, generator=gen, device="cuda"))  # un-normalised like get_rays'
    return o, d


def main(iters=1500, n=16384):
    if os.environ.get("R2L_EQ_FAMILIES"):  # e.g. "0,3": a subset of the families (long runs)
        keep = [int(v) for v in os.environ["R2L_EQ_FAMILIES"].split(",")]
        for k in [k for i, k in enumerate(list(FAMILIES)) if i not in keep]:
            del FAMILIES[k]
    seeds = [int(v) for v in os.environ.get("R2L_EQ_SEEDS", "0").split(",")]
    finals = {}
    for seed in seeds:
        if len(seeds) > 1:
            print("\n=== seed %d ===" % seed, flush=True)
        for name, psnr in run(iters, n, seed).items():
            finals.setdefault(name, []).append(psnr)
    if len(seeds) > 1:
        import statistics
        print("\n=== final held-out PSNR after %d steps of %d rays over seeds %s: mean +- sample std (min .. max) ===" % (iters, n, seeds))
        ref = "fp32 MFMA" if "fp32 MFMA" in finals else list(finals)[-1]
        mref = statistics.mean(finals[ref])
        for name, v in finals.items():
            sd_ = statistics.stdev(v)
            print("  %-22s %.3f +- %.3f dB  (%.3f .. %.3f)   mean - mean(%s) = %+.3f dB = %+.2f of this family's seed std; per seed: %s"
                  % (name, statistics.mean(v), sd_, min(v), max(v), ref, statistics.mean(v) - mref,
                     (statistics.mean(v) - mref) / max(sd_, 1e-9), " ".join("%.3f" % q for q in v)))
        pooled = statistics.mean([statistics.stdev(v) for v in finals.values()])
        print("  mean seed std over the families: %.3f dB; standard error of a %d-seed mean: %.3f dB" % (pooled, len(seeds), pooled / len(seeds) ** 0.5))


def run(iters, n, seed):
    """One comparison of the families at one seed; returns {family: final held-out PSNR}."""
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    sd = O.make_state_dict(43, seed=seed)
    gen = torch.Generator(device="cuda").manual_seed(123)
    test_o, test_d = rays(65536, gen)
    test_rgb = scene(test_o, test_d)
    curves, finals, weights = {}, {}, {}
    from r2l_amd import engine
    ranges = {}
    for name, cfg in FAMILIES.items():
        engine.DEFAULT_CONFIG = dict(cfg)
        m = build_model(sd, 43)
        tr = R2LTrainer(m, ps)
        ranges[name] = []
        g = torch.Generator(device="cuda").manual_seed(7 + 1000 * seed)       # batches
        gj = torch.Generator(device="cuda").manual_seed(8 + 1000 * seed)      # jitter
        curve = []
        torch.cuda.synchronize()
        t0 = time.time()
        for it in range(1, iters + 1):
            o, d = rays(n, g)
            tgt = scene(o, d)
            t_rand = torch.rand(n, 16, generator=gj, device="cuda")
            tr.step(o, d, tgt, lr_schedule(it, 5e-4, 500, "0.0001,200"), perturb=1.0, t_rand=t_rand)
            if it % max(250, iters // 8) == 0 or it == iters:
                with torch.no_grad():
                    out = m.forward_rays(test_o, test_d, ps)
                psnr = (-10. * torch.log10(((out - test_rgb) ** 2).mean())).item()
                curve.append((it, psnr))
                ri = tr.range_info()  # range control telemetry of the fp16 kernels (include/r2l_hip.h): how close does training get?
                ranges[name].append((it, ri["amax"], ri["scale"], ri["trips"], ri.get("grad_amax", 0.), ri.get("grad_scale", 0.),
                                     ri.get("bwd_trips", 0)))
        torch.cuda.synchronize()
        curves[name], finals[name] = curve, out.clone()
        if os.environ.get("R2L_EQ_PARITY"):  # the TRAINED student against the CPU restatement (checker only; same rays, fp32 torch CPU ops)
            k = int(os.environ["R2L_EQ_PARITY"])
            sd_t = {key: v.detach().cpu().clone() for key, v in m.state_dict().items()}
            ref = O.r2l_forward(sd_t, O.positional_embed(O.sample_train(test_o[:k].cpu(), test_d[:k].cpu(), O.z_vals(16, 2., 6.), 0.), 10))
            with torch.no_grad():
                got = m.forward_rays(test_o[:k], test_d[:k], ps, perturb=0.).cpu()
            ri = tr.range_info()
            print("%-22s trained weights after %d steps, %d held-out rays: max |rgb_hip - rgb_cpu_fp32| = %.3e (bar 1e-4); "
                  "largest |activation| %.3g, scale %g, launches redone %d" % (name, iters, k, (got - ref).abs().max().item(),
                                                                              ri["amax"], ri["scale"], ri["trips"]), flush=True)
        weights[name] = m.engine().flat.clone() if hasattr(m, "engine") else None
        print("%-22s %5.1f s  " % (name, time.time() - t0) + "  ".join("it %d: %.3f dB" % c for c in curve), flush=True)
    names = list(FAMILIES)
    print("\nheld-out frame distance between families after %d iterations (PSNR of one family's prediction against another's):" % iters)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            a, b = finals[names[i]], finals[names[j]]
            print("  %-22s vs %-22s  %.2f dB  (max |dRGB| %.4f)" % (names[i], names[j],
                  (-10. * torch.log10(((a - b) ** 2).mean())).item(), (a - b).abs().max().item()))
    print("final held-out PSNR: " + ", ".join("%s %.3f dB" % (k, v[-1][1]) for k, v in curves.items()))
    print("\nrange of the fp16 kernels during training (largest |activation| of the last step incl. the held-out frame, activation "
          "scale, forward launches redone | largest |chain gradient|, gradient scale, steps redone):")
    for name, rows in ranges.items():
        if rows and rows[-1][5]:
            print("  %-22s " % name + "  ".join("it %d: %.3g x%g (%d) | %.3g x%g (%d)" % r for r in rows))
    return {k: v[-1][1] for k, v in curves.items()}


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:3]])
