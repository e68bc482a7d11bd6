"""Socket power and gfx clock (amdsmi, 10 ms samples) while tools/_bin/mfma_power_probe runs ONE MFMA-only case for ~2 s each:
ties the probe's sustained product rates to the power cap and the clock the chip settles at."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from power_trace import Sampler  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("fp16, random, A re-read from LDS (2 ds_read_b128 / 3 MFMAs)", "0 2 1 10 lds"), ("fp16, zero operands", "0 0 1 10"), ("fp16, random mantissas", "0 2 1 10"), ("bf16, random mantissas", "1 2 1 10"),
         ("fp16, random, 4 of 10 mantissa bits", "0 2 1 4"), ("fp16, random, 0 of 10 mantissa bits", "0 2 1 0")]


def main():
    s = Sampler()
    s.start()
    time.sleep(0.3)
    print("%-66s %10s %10s %10s %10s" % ("case", "PF/s", "mean W", "max W", "gfx MHz"))
    for name, spec in CASES:
        env = dict(os.environ, PROBE_ONE=spec.replace(" lds", ""), PROBE_LONG="80")
        if spec.endswith("lds"):
            env["PROBE_LDS"] = "1"
        s.phase = name
        out = subprocess.run([os.path.join(ROOT, "tools", "_bin", "mfma_power_probe")], env=env, capture_output=True, text=True).stdout
        s.phase = "idle"
        time.sleep(0.5)
        rows = [r for r in s.rows if r[1] == name and isinstance(r[2], (int, float))]
        rows = rows[len(rows) // 3:]  # the settled part
        pw = [r[2] for r in rows]
        ck = [r[3] for r in rows if isinstance(r[3], (int, float))]
        print("%-66s %10s %10.0f %10.0f %10.0f" % (name, out.split()[2] if out else "?", sum(pw) / max(len(pw), 1), max(pw or [0]),
                                                   sum(ck) / max(len(ck), 1)))
    s.stop = True


if __name__ == "__main__":
    main()
