"""Socket power + shader clock of the MI355X while the hot kernels run (keep a power / sclk trace next to
the GRBM-derived clocks).  A sampler thread polls amdsmi (gpu_metrics: socket power, per-XCD gfx clocks; power cap) every
~10 ms while the main thread drives one phase after another; per phase: mean / max power, mean gfx clock, kernel time.

    python tools/power_trace.py [seconds per phase]   ->  gpurun_out/power_trace.csv + a summary table on stdout
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self, period=0.01):
        super().__init__(daemon=True)
        import amdsmi
        self.smi = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[0]
        self.period, self.rows, self.phase, self.stop = period, [], "idle", False
        self.cap = None
        try:
            info = amdsmi.amdsmi_get_power_cap_info(self.h)
            self.cap = info.get("power_cap")
            print("power cap info:", info)
        except Exception as e:  # noqa: BLE001
            print("power cap: n/a (%s)" % e)

    def sample(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        power = m.get("current_socket_power")
        if power in (None, "N/A", 0xFFFF, 0xFFFFFFFF):
            power = m.get("average_socket_power")
        clks = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 60000]
        clk = sum(clks) / len(clks) if clks else m.get("current_gfxclk")
        return power, clk, m.get("temperature_hotspot"), m.get("average_gfx_activity")

    def run(self):
        t0 = time.perf_counter()
        while not self.stop:
            try:
                p, c, temp, act = self.sample()
                self.rows.append((time.perf_counter() - t0, self.phase, p, c, temp, act))
            except Exception as e:  # noqa: BLE001
                self.rows.append((time.perf_counter() - t0, self.phase, None, None, None, str(e)[:60]))
            time.sleep(self.period)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    dev = torch.device("cuda", 0)
    net, ps, _ = bench.make_model(dev)
    from r2l_amd.data import pose_spherical
    from r2l_amd.train_step import R2LTrainer, lr_schedule
    poses = [pose_spherical(-180. + 9. * i, -30., 4.)[:3, :4] for i in range(40)]
    n = 98304
    g = torch.Generator().manual_seed(1)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).to(dev)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    tgt = torch.rand(n, 3, generator=g).to(dev)
    tr = R2LTrainer(net, ps)

    def render(i):
        with torch.no_grad():
            net.render_pose(poses[i % 40], ps)

    def train(i):
        tr.step(o, d, tgt, lr_schedule(i + 1, 5e-4, 500, "0.0001,200"), perturb=1.0)

    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)

    def gemm(i):
        torch.matmul(a, b)

    phases = [("render fp16x2 (default)", render, {}), ("render bf16x3", render, {"R2L_NO_FWD2": "1"}),
              ("render fp32 MFMA", render, {"R2L_NO_FWD3": "1"}), ("train step 98304 rays (default trio)", train, {}),
              ("hipBLASLt bf16 GEMM 8192^3 (reference load)", gemm, {})]
    s = Sampler()
    s.start()
    time.sleep(1.0)
    out = []
    for name, fn, env in phases:
        os.environ.update(env)
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        s.phase = name
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0, it = time.perf_counter(), 0
        e0.record()
        while time.perf_counter() - t0 < secs:
            for _ in range(8):
                fn(it)
                it += 1
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        s.phase = "idle"
        out.append((name, e0.elapsed_time(e1) / it, it))
        for k in env:
            del os.environ[k]
        time.sleep(1.0)
    s.stop = True
    s.join()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/power_trace.csv", "w") as f:
        f.write("t_s,phase,socket_power_W,gfx_clk_MHz,hotspot_C,gfx_activity\n")
        for r in s.rows:
            f.write(",".join("" if v is None else str(v) for v in r) + "\n")
    print("\n| phase | ms per call | samples | power mean / max (W) | gfx clock mean / min (MHz) | cap (W) |")
    print("|---|---|---|---|---|---|")
    for name, ms, it in [("idle", 0., 0)] + out:
        rows = [r for r in s.rows if r[1] == name and isinstance(r[2], (int, float)) and isinstance(r[3], (int, float))]
        rows = rows[len(rows) // 5:] if name != "idle" else rows  # skip the ramp at the start of a phase
        if not rows:
            print("| %s | %.3f | 0 | n/a | n/a | %s |" % (name, ms, s.cap))
            continue
        p = [r[2] for r in rows]
        c = [r[3] for r in rows]
        print("| %s | %.3f | %d | %.0f / %.0f | %.0f / %.0f | %s |" % (name, ms, len(rows), sum(p) / len(p), max(p),
                                                                      sum(c) / len(c), min(c), s.cap))


if __name__ == "__main__":
    main()
