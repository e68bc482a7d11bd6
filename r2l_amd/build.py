"""Build libr2l_hip.so (hand-written gfx950 HIP kernels + C ABI) in-tree with hipcc.

The .so lands in r2l_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).  hipcc cross-compiles
gfx950 without a GPU, so this also serves as the CPU-side "does it build" check (__graft_entry__.build).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(LIBDIR, "libr2l_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the point/ray arithmetic must reproduce the reference's separately rounded mul/add
# (SURVEY.md §7 "Bit-exact point arithmetic"); FMAs are requested explicitly where wanted.
EXTRA = os.environ.get("R2L_EXTRA_FLAGS", "").split()
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path):
    h = hashlib.sha1()
    h.update(" ".join(FLAGS).encode())
    with open(path, "rb") as f:
        h.update(f.read())
    for hdr in sorted(os.listdir(CSRC)):
        if hdr.endswith(".h"):
            with open(os.path.join(CSRC, hdr), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    stamp = obj + ".sha1"
    dig = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-I", os.path.join(HERE, "..", "include"), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(verbose=True, force=False):
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results) or not os.path.exists(LIB)
    if rebuilt:
        tmp = "%s.%d.tmp" % (LIB, os.getpid())  # link aside, then rename: other ranks never see a half-written .so
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        os.replace(tmp, LIB)
    if verbose:
        print("[r2l_amd.build] %s (%s)" % (LIB, "rebuilt" if rebuilt else "up to date"))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
