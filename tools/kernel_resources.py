"""Register / LDS / scratch use of every kernel of libr2l_hip.so, from the code-object metadata hipcc writes into the device
assembly (same flags as r2l_amd/build.py).  Runs without a GPU:  python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from r2l_amd import build  # noqa: E402


def one(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [build.HIPCC] + build.FLAGS + ["-I", os.path.join(build.HERE, "..", "include"), "-S", "--cuda-device-only",
                                             os.path.join(build.CSRC, src), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        text = open(out).read()
    rows = []
    for blk in text.split("  - .agpr_count:")[1:]:
        def f(key, blk=blk):
            m = re.search(r"\.%s:\s*(\S+)" % key, blk)
            return m.group(1) if m else "?"
        name = subprocess.run(["c++filt", f("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*\)$", "", name)
        rows.append((src.replace(".hip", ""), name, f("vgpr_count"), blk.split()[0], f("vgpr_spill_count"),
                     f("private_segment_fixed_size"), f("group_segment_fixed_size"), f("sgpr_count")))
    return rows


def main():
    srcs = build._sources()
    with ThreadPoolExecutor(max_workers=8) as ex:
        allrows = [r for rows in ex.map(one, srcs) for r in rows]
    print("Register / LDS / scratch use of every kernel of libr2l_hip.so (hipcc %s; from the code object metadata:" % " ".join(build.FLAGS))
    print("vgpr_count includes AGPRs, 512 = one wave per SIMD; spills = VGPRs spilled to scratch; LDS = static bytes).\n")
    print("%-22s %-62s %5s %5s %6s %9s %7s %5s" % ("file", "kernel", "VGPR", "AGPR", "spills", "scratch B", "LDS B", "SGPR"))
    for r in allrows:
        print("%-22s %-62s %5s %5s %6s %9s %7s %5s" % r)


if __name__ == "__main__":
    main()
