"""Where a kernel's spilled registers are touched: per loop of the device assembly, the number of MFMA, scratch (spill / fill),
global-load and LDS instructions.  A spill count alone does not say whether the scratch traffic sits in the hot loop or in a
prologue / flush that runs once per workgroup.  Runs without a GPU:
    python tools/spill_sites.py r2l_backward.hip r2l_dw_body_kernel > profiles/rNN_dw_body_spill_sites.txt
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from r2l_amd import build  # noqa: E402


def kernel_asm(src, kernel):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [build.HIPCC] + build.FLAGS + ["-I", os.path.join(build.HERE, "..", "include"), "-S", "--cuda-device-only",
                                             os.path.join(build.CSRC, src), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        lines = open(out).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and kernel in l]
    if not start:
        raise SystemExit("no kernel matching %r in %s" % (kernel, src))
    out = []
    for s in start:
        e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))  # a kernel may hold several s_endpgm
        out.append((lines[s].split(":")[0], lines[s:e]))
    return out


KINDS = (("mfma", "v_mfma"), ("scratch", "scratch_"), ("global/buffer load", r"(global|buffer)_load"),
         ("global/buffer store", r"(global|buffer)_store"), ("lds", r"ds_(read|write|load|store)"))


def count(body, lo, hi):
    return [sum(1 for l in body[lo:hi] if re.search(pat, l)) for _, pat in KINDS]


def report(name, body):
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print("%s   (%d lines)" % (demangled, len(body)))
    label = {}
    for i, l in enumerate(body):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            label[m.group(1)] = i
    loops = set()
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and label.get(m.group(1), len(body)) < i:
            loops.add((label[m.group(1)], i))
    # keep the widest back edge per loop header
    widest = {}
    for lo, hi in loops:
        widest[lo] = max(widest.get(lo, 0), hi)
    loops = sorted(widest.items())
    hdr = "%-22s" % "lines" + "".join("%22s" % k for k, _ in KINDS) + "   nesting"
    print(hdr)
    print("%-22s" % "whole kernel" + "".join("%22d" % c for c in count(body, 0, len(body))))
    for lo, hi in loops:
        depth = sum(1 for a, b in loops if a <= lo and hi <= b) - 1
        print("%-22s" % ("loop %5d..%5d" % (lo, hi)) + "".join("%22d" % c for c in count(body, lo, hi + 1)) + "   %d" % depth)
    inner = [(lo, hi) for lo, hi in loops if not any(a >= lo and b <= hi and (a, b) != (lo, hi) for a, b in loops)]
    tot = count(body, 0, len(body))
    ins = [sum(count(body, lo, hi + 1)[k] for lo, hi in inner) for k in range(len(KINDS))]
    print("innermost loops hold %d of %d MFMAs and %d of %d scratch instructions" % (ins[0], tot[0], ins[1], tot[1]))
    print()


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    for name, body in kernel_asm(sys.argv[1], sys.argv[2]):
        report(name, body)


if __name__ == "__main__":
    main()
