#!/usr/bin/env python
"""bench.py — rays/sec of the R2L W256D88 hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Started under torch.distributed.run (the driver's form) this process is a rank;
started bare it launches the N ranks itself (plan_launch) — and exits non-zero rather than run on fewer GPUs than
--gpus says.  The printed line carries n_gpus and rccl_ranks (the latter measured by an all-reduce).

Workload (config.workload): BASELINE.json configs[1] — render 400x400 test frames (160 000 rays each, 16 samples/ray,
L=10, W256 D88, seeded weights, synthetic pose_spherical poses).  One "step" = ONE LAUNCH of the fused HIP forward (ray
sampling + positional encoding + 88-layer ResMLP + RGB head) = FRAMES_PER_STEP (9) test frames per GPU, the way
driver.render_path walks the 200 test poses (r2l_forward_poses_cfg: 9 x 1250 workgroups = 43.95 rounds of the 256 CUs, no
launch gap and no partly filled last round per frame).  Inputs (poses, weights) are resident before the timed region.  Frames
shard across ranks with no collective -> "scaling": "weak".
The TOP-LEVEL record (value / dtype / roofline / ms_per_step) is the GRADED number: the exact-fp32 MFMA kernel family
(v_mfma_f32_32x32x2_f32, peak 157.3 TF) — the reference's arithmetic.  "train" (distillation step: forward + backward + Adam)
and "teacher" at the top level are the exact-fp32 families too.  The library's default family — fp16x2: every fp32 product as 3
fp16 MFMA products, ~2^-21 relative — is the FAST MODE and is reported under "fast_mode" with its own roofline, range telemetry
and parity figure: "fast_mode.train" (default trio; at N > 1 the bucketed RCCL all-reduce of the flat gradient overlapped with
the weight-gradient stages), "fast_mode.train_strong" (N > 1), "fast_mode.train_4096" / "train_12288", "fast_mode.teacher".
"fp32_grade_products" = the bf16x3 family (six bf16 products per fp32 product).  "raw2outputs" = the alpha-composite kernel
against the HBM roofline.  Every leg is timed by the same barrier-bracketed recipe with the full K / W.
Prints ONE JSON line on rank 0; its last key "summary" is a digest of every leg.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FWD_FLOP_PER_RAY = 2 * (1008 * 256 + 86 * 256 * 256 + 256 * 3)  # 11 789 824
TRAIN_FLOP_PER_RAY = 2 * (3 * 5894912 - 258048)  # 34 853 376: fwd + dX + dW, no dX for the head
PEAK_FP32_MFMA = 157.3  # TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md chip table (exact-fp32 MFMA)
PEAK_BF16_MFMA = 2500.0  # TFLOP/s dense bf16 MFMA (same guide; the sparsity-inflated headline figure is not used)
# What an MFMA-ONLY stream sustains on this chip with random operand mantissas (tools/mfma_power_probe.hip, every SIMD issuing
# nothing but v_mfma_f32_32x32x16 on register operands; profiles/r03_mfma_power_probe.txt): the datasheet rate is reached on
# zero operands only, the power cap holds random data to these product rates.  Reported beside `peak`, never instead of it.
SUSTAINED_FP16_MFMA_ONLY = 1770.0  # TFLOP/s of fp16 products
SUSTAINED_BF16_MFMA_ONLY = 1920.0
SUSTAINED_FP16_MFMA_LDS_FED = 1570.0  # the same fp16 stream with its A operands re-read from LDS at the render kernel's ratio
H = W = 400
FOCAL = 555.5555155968841
FRAMES_PER_STEP = 9  # driver.POSES_PER_LAUNCH: test frames per render launch


def make_model(device):
    """W256 D88 student with default nn.Linear init under torch.manual_seed(0) (no checkpoint exists offline)."""
    from model.nerf_raybased import NeRF_v3_2, PointSampler
    trial = argparse.Namespace(ON=True, body_arch="resmlp", inact="relu", outact="none", res_scale=1., n_learnable=2,
                               n_block=-1, near=-1, far=-1)
    args = argparse.Namespace(netdepth=88, netwidth=256, layerwise_netwidths="", act="relu", linear_tail=False,
                              use_residual=True, trial=trial)
    torch.manual_seed(0)
    net = NeRF_v3_2(args, 1008, 3)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}  # CPU copy for the cpu_baseline leg
    net = net.to(device)
    ps = PointSampler(H, W, FOCAL, 16, 2., 6., device=device)
    return net, ps, sd


def barrier_sync(distributed):
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()


LEG_WALL = {}  # leg name -> host wall seconds incl. warm-up and set-up (so that the driver's clock can be reconciled leg by leg)


class leg_clock:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        torch.cuda.synchronize()
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        LEG_WALL[self.name] = LEG_WALL.get(self.name, 0.) + time.perf_counter() - self.t0


def timed(fn, steps, warmup, distributed, device):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks.
    Also returns the mean device time per step from HIP events recorded on the launch stream."""
    for i in range(warmup):
        fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    barrier_sync(distributed)
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        fn(warmup + i)
        ev[i][1].record()
    barrier_sync(distributed)
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
    return dt, kernel_ms


def pmc_traffic(kernel_prefix, grid_threads=None):
    """HBM-side bytes per launch of a kernel from the newest committed rocprofv3 --pmc summary (profiles/
    rNN_bench_pmc_summary.json, made by tools/profile_gpu.sh in separate counter passes): FETCH_SIZE x 2 (gfx950 tallies
    the 128-B requests of 16 B/lane loads at 64 B, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both reported in KiB.
    PMC passes cannot run inside this process, so the figure is the recorded one, or None if no summary is there."""
    import glob
    import json as _json
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    files = sorted(glob.glob(os.path.join(root, "r*_bench_pmc_summary.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        table = _json.load(f)
    hits = [(name, c) for name, c in table.items()
            if name.startswith(kernel_prefix) and "FETCH_SIZE" in c and "WRITE_SIZE" in c]
    if grid_threads is not None:  # the launch of THIS workload (the summary keys end in " grid=<work-items>")
        hits = [h for h in hits if h[0].endswith(" grid=%d" % grid_threads)] or []
    for name, c in hits:
        return (2. * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024., os.path.relpath(files[-1], os.path.dirname(root))
    return None, None


def cpu_baseline(sd, n_rays=32768):
    """The oracle (CPU restatement of the reference op sequence) timed on the host cores: reported, not a target
    (SURVEY.md §8d / BASELINE.md §3: forward at 4096 / 32 768 rays / one frame, training step at 4096 rays, medians of 5,
    one training step with autograd anomaly mode on as the reference runs it, model/nerf_raybased.py:4).
    torch's intra-op pool scales badly past a few dozen threads on these 256x256 GEMMs (measured on the EPYC 9575F box:
    16 threads 32.6 k rays/s, 64 threads 12.5 k, 256 threads 0.4 k), so a few thread counts are tried and the best kept."""
    from oracle import r2l_oracle as Or
    g = torch.Generator().manual_seed(0)
    dirs = Or.pixel_dirs(H, W, FOCAL)
    c2w = torch.from_numpy(Or.pose_spherical(30., -30., 4.)[:3, :4])
    z = Or.z_vals(16, 2., 6.)
    rows = torch.randperm(H * W, generator=g)[:n_rays]
    ncpu = os.cpu_count() or 1
    best, best_threads, rgb = None, 1, None

    def fwd(idx):
        t0 = time.perf_counter()
        pts = Or.sample_test(dirs, z, c2w)
        out = Or.r2l_forward(sd, Or.positional_embed(pts if idx is None else pts[idx], 10))
        return time.perf_counter() - t0, out

    with torch.no_grad():
        for nt in sorted({min(ncpu, 16), min(ncpu, 32), min(ncpu, 64)}):
            torch.set_num_threads(nt)
            fwd(rows[:2048])  # warm-up
            dt, _ = fwd(rows)
            if best is None or dt < best:
                best, best_threads = dt, nt
        torch.set_num_threads(best_threads)
        med = {}
        for name, idx in (("4096", rows[:4096]), ("32768", rows)):
            ts = []
            for _ in range(5):
                dt, out = fwd(idx)
                ts.append(dt)
            med[name] = sorted(ts)[2]
            if name == "32768":
                rgb = out
        t_frame, _ = fwd(None)  # the whole 400x400 frame, once
    # training step on the host: sample_train + encode + forward + autograd backward + Adam, N = 4096
    n_tr = 4096
    o = torch.randn(n_tr, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])
    d = torch.nn.functional.normalize(torch.randn(n_tr, 3, generator=g), dim=-1)
    tgt = torch.rand(n_tr, 3, generator=g)
    p = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}

    def train_once(it):
        t0 = time.perf_counter()
        emb = Or.positional_embed(Or.sample_train(o, d, z, 1.0, t_rand=torch.rand(n_tr, 16, generator=g)), 10)
        grads = Or.r2l_loss_and_grads(p, emb, tgt)[2]
        for k in p:
            p[k], m[k], v2[k] = Or.adam_step(p[k], grads[k], m[k], v2[k], it, 5e-4)
        return time.perf_counter() - t0

    times = [train_once(it) for it in range(1, 7)]
    train_s = sorted(times[1:])[2]  # median of 5 after one warm-up
    with torch.autograd.set_detect_anomaly(True):  # the reference's setting (model/nerf_raybased.py:4)
        train_anomaly_s = train_once(7)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": n_rays / med["32768"], "unit": "rays/s", "cores": best_threads, "kind": "port",
            "forward": {"rays_4096": 4096 / med["4096"], "rays_32768": n_rays / med["32768"], "frame_160000": H * W / t_frame,
                        "unit": "rays/s", "sample": "medians of 5 (4096, 32 768 rays); one run of the whole 400x400 frame"},
            "train": {"value": n_tr / train_s, "unit": "rays/s", "rays_per_step": n_tr, "cores": best_threads,
                      "anomaly_mode_on": n_tr / train_anomaly_s,
                      "sample": "median of 5 oracle training steps (sample + encode + fwd + autograd bwd + Adam); one more step "
                                "with torch.autograd.set_detect_anomaly(True), the reference's setting"},
            "sample": "%d rays of one 400x400 frame: sample + encode + W256D88 forward, fp32 torch CPU ops, median of 5 with the "
                      "best of {16,32,64} threads on %d logical CPUs; %s" % (n_rays, ncpu, cpu_model)}, rgb, rows


def teacher_leg(device, world, rank, distributed, frames=2, precision="fp16x2"):
    """NeRF-teacher pseudo-data render (BASELINE configs[4]): `frames` 400x400 poses per GPU, 64 coarse + 128 fine
    samples, perturb=1 (create_data.py 'rand' settings), seeded D8W256 teacher pair; poses shard over ranks.
    precision: the r2l_config.precision handed to every r2l_teacher_mlp_cfg call (what the printed peak is derived from)."""
    from model.nerf_raybased import NeRF
    from r2l_amd import _lib
    from r2l_amd.data import pose_spherical
    from r2l_amd.render import render, teacher_engine
    nets = []
    torch.manual_seed(11)
    for _ in range(2):  # coarse + fine, default init; density bias so that rays are neither empty nor opaque
        m = NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
        with torch.no_grad():
            m.alpha_linear.bias.add_(0.5)
        nets.append(m.to(device))
        teacher_engine(nets[-1]).cfg = _lib.make_config(precision=precision)
    kw = dict(network_fn=nets[0], network_query_fn=None, N_samples=64, N_importance=128, network_fine=nets[1],
              white_bkgd=True, perturb=1., ndc=False, near=2., far=6., use_viewdirs=True)
    poses = [pose_spherical(17. * (i * world + rank), -35., 4.)[:3, :4].to(device) for i in range(frames + 1)]

    def step(i):
        with torch.no_grad():
            render(H, W, FOCAL, chunk=32768, c2w=poses[i % len(poses)], **kw)

    dt, step_ms = timed(step, frames, 1, distributed, device)
    flop_per_ray = 2 * 593408 * 256  # 303.82 MFLOP/ray
    achieved = H * W * flop_per_ray / (step_ms * 1e-3) / 1e12
    fwd2, fwd3 = precision == "fp16x2", precision in ("fp16x2", "bf16x3")  # 3 fp16 / 6 bf16 products per fp32 product, or fp32 MFMA
    peak = PEAK_BF16_MFMA / 3. if fwd2 else (PEAK_BF16_MFMA / 6. if fwd3 else PEAK_FP32_MFMA)
    return {"value": H * W * frames * world / dt, "unit": "rays/s", "ms_per_frame": dt / frames * 1e3,
            "precision": precision, "range": teacher_engine(nets[1]).range_info() if fwd2 else None,
            "parallelism": "poses sharded across %d rank(s), no collective" % world,
            "workload": "NeRF teacher render 400x400, 64+128 samples/ray, perturb=1, chunk 32768 (create_data rand)",
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "peak_fp32_mfma": PEAK_FP32_MFMA, "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA,
                         "kernel": "r2l_teacher2_kernel" if fwd2 else ("r2l_teacher3_kernel" if fwd3 else "r2l_teacher_mlp_kernel"),
                         "flop_per_ray": flop_per_ray}}


def graph_us(launch, k, warmup):
    """Device time per launch of `launch(i)`, i = 0 .. k-1: the k launches are captured into one hipGraph and the replay is timed with
    HIP events (a 10 - 30 us kernel behind output allocations and a ctypes call is HOST-bound when launched eagerly)."""
    for i in range(max(2, warmup)):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    method = "hipGraph replay of %d captured launches" % k
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(k):
                launch(i)
        graph.replay()
        torch.cuda.synchronize()
        reps = 5
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (k * reps) * 1e3
    except Exception as exc:  # noqa: BLE001
        method = "eager launches (graph capture failed: %s)" % type(exc).__name__
        torch.cuda.synchronize()
        e0.record()
        for i in range(k):
            launch(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / k * 1e3
    return us, method


def raw2outputs_leg(device, steps, warmup, n_rays=32768):
    """The teacher's alpha-composite kernel (r2l_raw2outputs16_kernel, create_data.py:335-402) against the HBM roofline, at the two
    shapes render_rays launches it with per 32 768-ray chunk: S = 64 (coarse pass: weights emitted for sample_pdf) and S = 192
    (fine pass: no weights).  ALGORITHMIC bytes per ray: S x (16 B raw + 4 B z) + 12 B rays_d read, 24 B of maps
    written (rgb 12, disp, acc, depth), + 4 S when the weights are emitted.  Device time per launch: graph_us; several input sets
    are cycled (one set, 42 / 126 MB, would sit in the 256 MB Infinity Cache from the second launch on and the "HBM" rate would be
    the cache's).  `at_262144_rays`: the same kernel on a launch eight times the reference's --chunk, where ramp-up and tail no
    longer weigh (what the kernel itself sustains).  `sample_pdf_sort`: the hierarchical-sampling kernel between the two passes
    (r2l_sample_pdf_sort16_kernel, helpers:283-330 + create_data.py:505-515; S = 64 coarse depths + weights and 128 uniforms in,
    128 new depths + the 192 merged and sorted depths + z_std out: 2308 B per ray), measured the same way."""
    from r2l_amd.render import raw2outputs, sample_pdf_sort
    g = torch.Generator(device="cpu").manual_seed(5)
    out = {"bound": "hbm", "peak": 8.0, "unit": "TB/s", "rays_per_launch": n_rays, "kernel": "r2l_raw2outputs16_kernel",
           "peak_note": "HBM3E 8 TB/s spec (6.3 TB/s is what a plain copy achieves: /opt/skills/guides/MI355X_MICROARCH.md)"}

    def measure(S, need_w, rays, n_sets, k):
        raws = [torch.randn(rays, S, 4, generator=g).to(device) for _ in range(n_sets)]
        zs = [(torch.sort(torch.rand(rays, S, generator=g), -1)[0] * 4. + 2.).to(device) for _ in range(n_sets)]
        d = torch.nn.functional.normalize(torch.randn(rays, 3, generator=g), dim=-1).to(device)
        us, method = graph_us(lambda i: raw2outputs(raws[i % n_sets], zs[i % n_sets], d, 0., True, need_weights=need_w), k, warmup)
        bpr = S * 20 + 12 + 24 + (4 * S if need_w else 0)
        tbs = rays * bpr / (us * 1e-6) / 1e12
        return {"bytes_per_ray": bpr, "weights_emitted": need_w, "us_per_launch": us, "timing": method, "input_sets_cycled": n_sets,
                "achieved": tbs, "frac": tbs / 8.0, "frac_of_achievable_6.3": tbs / 6.3}

    def measure_pdf(rays, n_sets, k, S=64, NI=128, det=False):
        zs = [(torch.sort(torch.rand(rays, S, generator=g), -1)[0] * 4. + 2.).to(device) for _ in range(n_sets)]
        ws = [(torch.rand(rays, S, generator=g) ** 4).to(device) for _ in range(n_sets)]
        if det:  # perturb = 0 (the test-set / video renders of the teacher): ONE linspace for every ray, 512 B instead of 512 B per ray
            lin = torch.linspace(0., 1., NI).to(device)
            us, method = graph_us(lambda i: sample_pdf_sort(zs[i % n_sets], ws[i % n_sets], NI, det=True, u=lin), k, warmup)
        else:
            us_ = [torch.rand(rays, NI, generator=g).to(device) for _ in range(n_sets)]
            us, method = graph_us(lambda i: sample_pdf_sort(zs[i % n_sets], ws[i % n_sets], NI, u=us_[i % n_sets]), k, warmup)
        bpr = 4 * (S + S + (0 if det else NI)) + 4 * (NI + S + NI) + 4
        tbs = rays * bpr / (us * 1e-6) / 1e12
        return {"bytes_per_ray": bpr, "us_per_launch": us, "timing": method, "input_sets_cycled": n_sets, "achieved": tbs,
                "frac": tbs / 8.0, "frac_of_achievable_6.3": tbs / 6.3, "kernel": "r2l_sample_pdf_sort16_kernel",
                "bound_note": "2308 B per ray would make it HBM-bound; round 6: the 128 samples are sorted and merged with the ascending "
                              "coarse depths (2816 compare-exchanges per ray instead of 4608; 1342 VALU instructions per wave of four rays "
                              "instead of 1581) — VALU-issue bound end to end: 1342 x 4 cycles x 8 waves per SIMD = 18 us, first-load "
                              "latency and the last stores around it; a staggered-start probe gained nothing (profiles/r06_sample_pdf_sort.txt)"}

    for S, need_w in ((64, True), (192, False)):
        r = measure(S, need_w, n_rays, 8 if S <= 64 else 4, max(20, steps))
        big = measure(S, need_w, 8 * n_rays, 2, 8)
        r["at_262144_rays"] = {k: big[k] for k in ("us_per_launch", "achieved", "frac", "input_sets_cycled")}
        # HBM-side bytes per launch from the committed PMC summary (2 x FETCH_SIZE + WRITE_SIZE; separate rocprofv3 --pmc passes of
        # tools/r2o_time.py): the quarter-wave-per-ray kernel is templated on <S / 16, weights emitted>
        r["traffic"], r["traffic_source"] = pmc_traffic("void r2l_raw2outputs16_kernel<%d, %s>" % (S // 16, "true" if need_w else "false"),
                                                        grid_threads=(n_rays + 15) // 16 * 256)
        r["algorithmic_bytes"] = n_rays * r["bytes_per_ray"]
        out["S%d" % S] = r
    r = measure_pdf(n_rays, 8, max(20, steps))
    big = measure_pdf(8 * n_rays, 2, 8)
    r["at_262144_rays"] = {k: big[k] for k in ("us_per_launch", "achieved", "frac", "input_sets_cycled")}
    r["traffic"], r["traffic_source"] = pmc_traffic("r2l_sample_pdf_sort16_kernel", grid_threads=(n_rays + 15) // 16 * 256)
    r["algorithmic_bytes"] = n_rays * r["bytes_per_ray"]
    out["sample_pdf_sort"] = r
    rd = measure_pdf(n_rays, 8, max(20, steps), det=True)  # perturb = 0: the samples come out ascending, their sort is skipped
    rd["note"] = "det u (perturb = 0: main.py --model_name nerf --render_test): merge only"
    out["sample_pdf_sort_det"] = rd
    return out


def summary_of(out):
    """A compact digest of the line, printed as its LAST key: leg -> [rays/s, ms per step, roofline frac]."""
    def row(d):
        if not isinstance(d, dict) or "value" not in d:
            return None
        r = d.get("roofline", {})
        return [round(d["value"]), round(d.get("ms_per_step", d.get("ms_per_frame", 0.)), 4), round(r.get("frac", 0.), 4)]
    fm = out.get("fast_mode", {})
    sm = {"graded_render_fp32_mfma": [round(out["value"]), round(out["ms_per_step"], 4), round(out["roofline"]["frac"], 4)],
          "graded_train_fp32_mfma": row(out.get("train")), "graded_train_4096": row(out.get("train_4096")),
          "graded_train_12288": row(out.get("train_12288")), "graded_teacher_fp32_mfma": row(out.get("teacher")),
          "bf16x3_render": row(out.get("fp32_grade_products")),
          "bf16x3_train": row(out.get("fp32_grade_products", {}).get("train")),
          "fast_render_fp16x2": row(fm), "fast_train": row(fm.get("train")), "fast_train_exact_dw": row(fm.get("train_exact_dw")),
          "fast_train_4096": row(fm.get("train_4096")), "fast_train_12288": row(fm.get("train_12288")),
          "fast_train_strong": row(fm.get("train_strong")), "fast_teacher": row(fm.get("teacher")),
          "fast_render_trained_like": row(fm.get("render_trained_like")),
          "parity_max_abs_err_vs_cpu": {"fp32_mfma": out.get("parity_max_abs_err_vs_cpu"), "fp16x2": fm.get("parity_max_abs_err_vs_cpu")},
          "columns": "[rays/s, ms per step (teacher: per frame), fraction of the leg's own matrix-pipe peak]"}
    r2o = out.get("raw2outputs")
    if r2o:
        sm["raw2outputs_frac_of_8TBs"] = {k: round(v["frac"], 4) for k, v in r2o.items() if isinstance(v, dict)}
        sm["raw2outputs_TBs"] = {k: round(v["achieved"], 3) for k, v in r2o.items() if isinstance(v, dict)}
        sm["raw2outputs_TBs_at_262144_rays"] = {k: round(v["at_262144_rays"]["achieved"], 3) for k, v in r2o.items()
                                                if isinstance(v, dict) and "at_262144_rays" in v}
    cb = out.get("cpu_baseline")
    if cb:
        sm["cpu_baseline_rays_per_s"] = {"forward": round(cb["value"]), "train": round(cb["train"]["value"]), "cores": cb["cores"]}
    return {k: v for k, v in sm.items() if v is not None}


def plan_launch(gpus, env, n_visible, argv=None, free_port=None):
    """What `python bench.py --gpus N` has to do in this process (reference: the DataParallel branch main.py:472-479 is
    replaced by one process per GPU).  Returns ("run", world, rank, local_rank) when this process is a rank (or the
    single-GPU job), or ("spawn", cmd) when it was started bare with N > 1 and must launch N ranks under
    torch.distributed.run itself.  Raises SystemExit (non-zero) instead of ever running fewer GPUs than asked for."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    world_env = env.get("WORLD_SIZE")
    if world_env is None:
        if n_visible < gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible; refusing to report a smaller job "
                             "under that label" % (gpus, n_visible))
        if gpus == 1:
            return ("run", 1, 0, 0)
        if free_port is None:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                free_port = sk.getsockname()[1]
        argv = list(sys.argv[1:]) if argv is None else list(argv)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port), os.path.abspath(__file__)] + argv
        return ("spawn", cmd)
    world = int(world_env)
    if world != gpus:
        raise SystemExit("bench.py: --gpus %d but launched with WORLD_SIZE=%d" % (gpus, world))
    local_world = int(env.get("LOCAL_WORLD_SIZE", world))
    if env.get("R2L_BENCH_SHARED_GPU_TEST") == "1":
        # test hook (tests/test_multirank_gpu.py): every rank on cuda:0 over gloo, to walk the N > 1 code of this file on a
        # one-GPU box; the line it prints says so and is not a measurement
        return ("run", world, int(env.get("RANK", "0")), 0)
    if n_visible < local_world:
        raise SystemExit("bench.py: %d local rank(s) but only %d GPU(s) are visible" % (local_world, n_visible))
    return ("run", world, int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--train-rays", type=int, default=98304,
                    help="rays per GPU per training step (README: N_rand 20 x 4096 + 20%% hard rays)")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-teacher", action="store_true")
    ap.add_argument("--segmented-leg", action="store_true",
                    help="N > 1: also time the 4096-ray step with the opt-in segmented dX chain (R2LTrainer(chain_segments=3))")
    ap.add_argument("--one-frame-leg", action="store_true",
                    help="also time the render kernel ONE frame per launch (the step of rounds 1 / 2), for comparison; off by "
                         "default so that the kernel-trace of this command holds one population of render launches")
    a = ap.parse_args()

    plan = plan_launch(a.gpus, os.environ, torch.cuda.device_count())
    if plan[0] == "spawn":  # `python bench.py --gpus N` run bare: become the launcher of N ranks (one per GPU)
        import subprocess
        raise SystemExit(subprocess.call(plan[1]))
    _, world, rank, local_rank = plan
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        shared_gpu_test = os.environ.get("R2L_BENCH_SHARED_GPU_TEST") == "1"
        if shared_gpu_test:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        # the rank count is MEASURED: a SUM all-reduce of ones over RCCL must return --gpus on every rank
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        if rccl_ranks != a.gpus or dist.get_world_size() != a.gpus:
            raise SystemExit("bench.py: --gpus %d but the RCCL all-reduce saw %d rank(s) (world_size %d)"
                             % (a.gpus, rccl_ranks, dist.get_world_size()))

    from r2l_amd import build as r2l_build
    if not os.path.exists(r2l_build.LIB):  # normally prebuilt in-tree by __graft_entry__.build(); never a CPU fallback
        if local_rank == 0:
            r2l_build.build(verbose=False)
        for _ in range(1200):
            if os.path.exists(r2l_build.LIB):
                break
            time.sleep(0.5)
    from r2l_amd.data import pose_spherical
    net, ps, sd = make_model(device)
    # synthetic test poses: pose_spherical(theta, -30, 4), theta = linspace(-180,180,41)[:-1]  (load_blender.py:84-86)
    thetas = [-180.0 + 9.0 * i for i in range(40)]
    poses = [pose_spherical(t, -30., 4.)[:3, :4] for t in thetas]
    frames = {}
    pose_t = torch.stack([torch.as_tensor(p, dtype=torch.float32) for p in poses], 0).to(device)  # resident before the clock

    def render_step(i):
        k0 = ((i * world + rank) * FRAMES_PER_STEP) % len(poses)
        idx = [(k0 + j) % len(poses) for j in range(FRAMES_PER_STEP)]
        with torch.no_grad():
            frames["rgb"] = net.render_poses(pose_t[idx], ps)

    from r2l_amd.engine import get_engine
    eng = get_engine(net)
    # Every leg selects its kernel family through r2l_config (include/r2l_hip.h; eng.set_config), and every printed peak /
    # kernel name is derived from the config that was passed — no environment switch is read or written in this file.
    PATHS = {
        "fp16x2": dict(peak=PEAK_BF16_MFMA / 3., products=3, sustained=SUSTAINED_FP16_MFMA_ONLY, kernel="r2l_fwd2_kernel<POSE>",
                       prof="void r2l_fwd2_kernel<true",
                       dtype="f32 (every product as 3 fp16 MFMA products of two-way fp16 operand splits hi + mid, ~2^-21 relative, "
                             "fp32 accumulate; range-controlled: power-of-two activation scale, bf16x3 fallback per launch)"),
        "bf16x3": dict(peak=PEAK_BF16_MFMA / 6., products=6, sustained=SUSTAINED_BF16_MFMA_ONLY, kernel="r2l_fwd3_kernel<POSE>",
                       prof="void r2l_fwd3_kernel<true",
                       dtype="f32 (products as 6 bf16 MFMA terms of exact bf16 hi/mid/lo splits: fp32-exact products, fp32 accumulate)"),
        "fp32_mfma": dict(peak=PEAK_FP32_MFMA, products=1, sustained=None, kernel="r2l_fwd_kernel<MODE_POSE>",
                          prof="void r2l_fwd_kernel<1, false>", dtype="f32 (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulate)"),
    }
    GRID = (FRAMES_PER_STEP * H * W + 127) // 128 * 256  # work-items of one render launch (the key of the PMC summary rows)

    def render_leg(precision, name):
        """K = --steps launches of FRAMES_PER_STEP frames on the kernel family `precision`, W = --warmup untimed ones."""
        eng.set_config(precision=precision)
        with leg_clock(name):
            dt_, kms = timed(render_step, a.steps, a.warmup, distributed, device)
        info = PATHS[precision]
        ach = FRAMES_PER_STEP * H * W * FWD_FLOP_PER_RAY / (kms * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(info["prof"], grid_threads=GRID)
        return {"path": precision, "value": FRAMES_PER_STEP * H * W * a.steps * world / dt_, "unit": "rays/s", "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": dt_ / a.steps * 1e3, "ms_per_frame": dt_ / a.steps * 1e3 / FRAMES_PER_STEP,
                "dtype": info["dtype"],
                "roofline": {"bound": "mfma", "achieved": ach, "peak": info["peak"], "unit": "TFLOP/s", "frac": ach / info["peak"],
                             "peak_fp32_mfma": PEAK_FP32_MFMA, "frac_of_fp32_mfma_peak": ach / PEAK_FP32_MFMA,
                             "traffic": traffic, "traffic_source": traffic_src,
                             "kernel": info["kernel"], "kernel_ms": kms, "rays_per_launch": FRAMES_PER_STEP * H * W,
                             "flop_per_ray": FWD_FLOP_PER_RAY}}

    # THE GRADED NUMBER = the top-level record (SURVEY.md §7 "Precision vs peak", §8(d)): the workload on arithmetic
    # equal to the reference's — exact fp32 products and accumulation on v_mfma_f32_32x32x2_f32 (r2l_forward.hip), priced against
    # the fp32 MFMA peak.  The library's DEFAULT family (fp16x2, 3 fp16 products per fp32 product, ~2^-21 relative) is the opt-in
    # sense of "fast mode" here and is reported under "fast_mode", never as `value`.
    head = render_leg("fp32_mfma", "render")
    traffic_note = ("HBM-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes; gfx950 tallies the "
                    "128-B requests of 16 B/lane loads at 64 B) recorded in %s; algorithmic bytes per launch = %d x 160000 rays x 12 "
                    "B out + one pass of the packed weight stream (24.3 MB fp32 / 25.1 MB of fp16 pairs / 37.6 MB of bf16 triples); "
                    "the excess is the per-workgroup re-stream of the weights missing the XCD L2s — at < 0.4 TB/s it bounds nothing"
                    % (head["roofline"]["traffic_source"], FRAMES_PER_STEP))
    out = {
        "metric": "rays/sec (train+render) W256D88 lego@400x400", "value": head["value"], "unit": "rays/s", "n_gpus": world,
        "rccl_ranks": rccl_ranks,  # measured: SUM all-reduce of ones over the nccl (= RCCL) process group
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": head["dtype"],
        "data": "synthetic",
        "frames_per_step": FRAMES_PER_STEP, "ms_per_frame": head["ms_per_frame"],
        "config": {"workload": "R2L W256D88 render_test 400x400 testskip=1: %d test frames (160000 rays each, 16 samples/ray, "
                               "L=10) per GPU per step in ONE launch of the fused sample+encode+ResMLP forward (as "
                               "driver.render_path walks the test poses); seeded weights, pose_spherical poses"
                               % FRAMES_PER_STEP,
                   "rays_per_step_per_gpu": FRAMES_PER_STEP * H * W,
                   "parallelism": "frames sharded across %d rank(s), no collective" % world,
                   "r2l_config": {"precision": "fp32_mfma (exact fp32: the graded number; the library's default family fp16x2 is "
                                               "reported under `fast_mode`)"}},
        "roofline": dict(head["roofline"], traffic_note=traffic_note,
                         peak_note="achieved counts ALGORITHMIC fp32 FLOPs (11 789 824 per ray) on the exact-fp32 MFMA "
                                   "(v_mfma_f32_32x32x2_f32, 157.3 TF dense: /opt/skills/guides/MI355X_MICROARCH.md)"),
    }

    def fast_roofline_extras(r):
        r.update({"frac_of_measured_mfma_only_rate": r["achieved"] * 3. / SUSTAINED_FP16_MFMA_ONLY,
                  "frac_of_measured_lds_fed_mfma_rate": r["achieved"] * 3. / SUSTAINED_FP16_MFMA_LDS_FED,
                  "peak_note": "3 fp16 MFMA products per fp32 product: matrix-pipe peak in algorithmic FLOP/s = dense 16-bit MFMA "
                               "peak 2500 TF / 3; an MFMA-only stream with random operand mantissas sustains 1.77 PF/s of fp16 "
                               "products on this chip under its 1.4 kW cap (1.57 with the A operands re-read from LDS at this "
                               "kernel's ratio): profiles/r03_mfma_power_probe.txt"})
        return r

    # FAST MODE (the library's default family): fp16x2 — every fp32 product as 3 fp16 MFMA products of (hi, mid) operand splits
    fast = render_leg("fp16x2", "render_fp16x2")
    fast_roofline_extras(fast["roofline"])
    fast["range"] = eng.range_info()  # the fp16 kernels' range control on these weights (scale 1, no launch redone)
    fast["note"] = ("the library's default kernel family; 22-bit operand products (~2^-21 relative) against the reference's 24-bit "
                    "fp32 — inside north_star's 1e-4 RGB tolerance (parity_max_abs_err_vs_cpu), but NOT the graded number")
    fast["speedup_vs_graded"] = fast["value"] / out["value"]
    out["fast_mode"] = fast
    # fp32-grade products on the 16-bit pipe: six bf16 products per fp32 product (exact hi/mid/lo splits; fp32 accumulate)
    out["fp32_grade_products"] = render_leg("bf16x3", "render_bf16x3")

    if rank == 0 and world == 1 and a.one_frame_leg:
        # the same kernel launched ONE frame at a time (round 1 / 2's step: 1250 workgroups = 4.88 rounds of the 256 CUs per
        # launch), for comparison across rounds; box-to-box spread of the 16-bit kernels is +-3 % (power-capped clocks)
        eng.set_config(precision="fp16x2")

        def one_frame_step(i):
            with torch.no_grad():
                frames["rgb"] = net.render_pose(poses[i % len(poses)], ps)
        with leg_clock("render_one_frame_per_launch"):
            dt1, k1 = timed(one_frame_step, a.steps, a.warmup, distributed, device)
        a1 = H * W * FWD_FLOP_PER_RAY / (k1 * 1e-3) / 1e12
        pk = PATHS["fp16x2"]["peak"]
        fast["render_one_frame_per_launch"] = {"value": H * W * a.steps / dt1, "unit": "rays/s", "ms_per_frame": dt1 / a.steps * 1e3,
                                               "roofline": {"bound": "mfma", "achieved": a1, "peak": pk, "unit": "TFLOP/s",
                                                            "frac": a1 / pk, "kernel_ms": k1}}
    if rank == 0 and world == 1:
        # the fast-mode kernel on "trained-like" weights: head scaled until the largest activation is ~1e5 (3x fp16's guard, tail
        # scaled back to keep the sigmoid active) — the range control of the fp16 kernels (include/r2l_hip.h) re-scales the
        # stream during the warm-up launches and the timed ones run on the SAME kernel at the same rate
        eng.set_config(precision="fp16x2")
        amax0 = fast["range"]["amax"]
        gain = 1.0e5 / max(amax0, 1e-3)
        with torch.no_grad():
            keep = {k: v.detach().clone() for k, v in net.state_dict().items()}
            net.head[0].weight.mul_(gain); net.head[0].bias.mul_(gain)
            net.tail[0].weight.mul_(1.0 / gain)
        tl = render_leg("fp16x2", "render_trained_like")
        tl["range"] = eng.range_info()
        tl["workload"] = ("as `fast_mode`, weights with |activation| up to %.3g (head x %.3g, tail / %.3g): warm-up launches re-scale "
                          "the stream (range.scale, range.trips), timed launches stay on r2l_fwd2_kernel" % (tl["range"]["amax"], gain, gain))
        tl["rate_vs_default_weights"] = tl["value"] / fast["value"]
        fast["render_trained_like"] = tl
        with torch.no_grad():
            net.load_state_dict(keep)
        eng.reset_range_history()

    rgb_gpu_check = {}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        with torch.no_grad():  # the frame the CPU baseline will be compared with (the training legs below update the weights)
            for prec in ("fp32_mfma", "fp16x2", "bf16x3"):
                eng.set_config(precision=prec)
                rgb_gpu_check[prec] = net.render_pose(pose_spherical(30., -30., 4.)[:3, :4], ps).cpu()
    eng.set_config(precision="fp16x2")
    train_mod = None
    if not a.no_train:
        try:
            from r2l_amd import train_step as train_mod
        except ImportError:
            train_mod = None
    if train_mod is not None:
        def train_leg(dest, key, **kw):
            with leg_clock(key if dest is out else "fast_mode." + key):
                dest[key] = train_mod.bench(net, ps, a, world, rank, distributed, device, timed, TRAIN_FLOP_PER_RAY,
                                            PEAK_FP32_MFMA, **kw)
        # GRADED training leg (top level): the distillation step with every GEMM on the exact-fp32 MFMA
        train_leg(out, "train", precision="fp32_mfma")
        out["train"]["note"] = "exact-fp32 MFMA in every GEMM (forward, dX chain, dW): the graded training number"
        # FAST MODE: the default trio (r2l_config precision fp16x2, dw_mode fp16)
        train_leg(fast, "train", precision="fp16x2", dw_mode="fp16")
        if distributed:
            # strong-scaling leg: the single-GPU batch (98 304 rays) split over the ranks, same global batch and the same
            # optimisation schedule as N = 1; the bucketed all-reduce has to hide under 1/N of the dW kernels here
            per = max(32, (a.train_rays // world + 31) // 32 * 32)
            train_leg(fast, "train_strong", precision="fp16x2", dw_mode="fp16", n_rays=per)
            fast["train_strong"]["scaling"] = "strong"
            fast["train_strong"]["global_rays_per_step"] = per * world
        # BASELINE configs[2] read literally ("N_rand=4096" as 4096 rays per step; at N GPUs configs[3]: 4096 rays per GPU), and
        # 12 288 rays = the per-GPU share of the 98 304-ray step at 8 GPUs (strong scaling)
        train_leg(fast, "train_4096", precision="fp16x2", dw_mode="fp16", n_rays=4096)
        train_leg(fast, "train_12288", precision="fp16x2", dw_mode="fp16", n_rays=12288)
        # the same two step sizes on the reference's arithmetic (exact-fp32 MFMA in every GEMM): graded, top level
        train_leg(out, "train_4096", precision="fp32_mfma", n_rays=4096)
        train_leg(out, "train_12288", precision="fp32_mfma", n_rays=12288)
        if distributed and a.segmented_leg:
            # the same steps with the dX chain cut into 3 segments (opt-in, R2LTrainer(chain_segments=3)): each segment's weight
            # gradients and all-reduce beside the next segment — what cutting the chain buys, once a node measures it.  Behind a
            # flag: the form has never run on more than one GPU, and an unattended scaling run must not be its first test
            train_leg(fast, "train_4096_segmented_chain", precision="fp16x2", dw_mode="fp16", n_rays=4096,
                      chain_segments=3)
        if rank == 0 and world == 1:
            # exact weight gradients (r2l_config.dw_mode = R2L_DW_EXACT): hi + mid operands, three products in the dW GEMMs
            train_leg(fast, "train_exact_dw", precision="fp16x2", dw_mode="exact")
            # fp32-grade products: every GEMM with six bf16 products per fp32 product
            train_leg(out, "train_bf16x3", precision="bf16x3")
            out["fp32_grade_products"]["train"] = out.pop("train_bf16x3")
        eng.set_config(precision="fp16x2", dw_mode="auto")

    if not a.no_teacher:
        with leg_clock("teacher"):  # graded: the exact-fp32 MFMA point network
            out["teacher"] = teacher_leg(device, world, rank, distributed, precision="fp32_mfma")
        with leg_clock("teacher_fp16x2"):
            fast["teacher"] = teacher_leg(device, world, rank, distributed, precision="fp16x2")
        if rank == 0:
            with leg_clock("raw2outputs"):
                out["raw2outputs"] = raw2outputs_leg(device, a.steps, a.warmup)
    # the CPU baseline goes LAST: torch's intra-op pool keeps its 16-64 threads spinning for a while after the oracle's GEMMs,
    # which slows the host thread that launches the (launch-bound, ~0.8 ms) 4096-ray steps: 0.83 -> 1.46 ms per step measured
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        t_cb = time.perf_counter()
        cb, rgb_cpu, rows = cpu_baseline(sd)
        LEG_WALL["cpu_baseline"] = time.perf_counter() - t_cb
        out["cpu_baseline"] = cb
        # parity spot check of the benchmarked kernels against the CPU baseline output (same pose, same seeded weights: the GPU
        # frames were rendered before the training legs moved them)
        out["parity_max_abs_err_vs_cpu"] = (rgb_gpu_check["fp32_mfma"][rows] - rgb_cpu).abs().max().item()
        fast["parity_max_abs_err_vs_cpu"] = (rgb_gpu_check["fp16x2"][rows] - rgb_cpu).abs().max().item()
        out["fp32_grade_products"]["parity_max_abs_err_vs_cpu"] = (rgb_gpu_check["bf16x3"][rows] - rgb_cpu).abs().max().item()
    if distributed and shared_gpu_test:
        out["shared_gpu_test"] = "ranks share ONE GPU over gloo (R2L_BENCH_SHARED_GPU_TEST=1): a walk through the N > 1 code, not a measurement"
    out["leg_wall_s"] = {k: round(v, 3) for k, v in LEG_WALL.items()}  # rank 0's host wall per leg, warm-up and set-up included
    out["summary"] = summary_of(out)  # LAST key: a compact digest of every leg (records that keep only the tail of this line still hold it)
    if rank == 0:
        def finite(o):  # strict JSON has no Infinity / NaN (an unused head-room reads inf): null instead
            if isinstance(o, float):
                return o if math.isfinite(o) else None
            if isinstance(o, dict):
                return {k: finite(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [finite(v) for v in o]
            return o
        print(json.dumps(finite(out), allow_nan=False))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
