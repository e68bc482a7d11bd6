"""Build libr2l_hip.so (hand-written gfx950 HIP kernels + C ABI) in-tree with hipcc.

The .so lands in r2l_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).  hipcc cross-compiles
gfx950 without a GPU, so this also serves as the CPU-side "does it build" check (__graft_entry__.build).
"""
import glob
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.environ.get("R2L_LIB_DIR") or os.path.join(HERE, "lib")  # env override: A/B builds (tools/)
OBJDIR = os.path.join(LIBDIR, "obj")
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
    # the C ABI header is compiled into every object (r2l_common.h includes it: r2l_config's layout), and a cached object was
    # audited by the rule in force when it was built
    with open(os.path.join(HERE, "..", "include", "r2l_hip.h"), "rb") as f:
        h.update(f.read())
    h.update(("audit:%s:%s" % (AUDIT_VERSION, _PK_F32.pattern)).encode())
    return h.hexdigest()


# ISA audit.  On gfx950 a packed-fp32 VALU op whose LOW lane takes the HIGH dword of src1 (`v_pk_fma_f32 ... op_sel:[0,1,0]`)
# was caught dropping its low-lane result in lanes 48-63 when a second wave shared the SIMD (csrc/r2l_coopf.h,
# profiles/r03_coresidency.md).  hipcc forms these from plain scalar code (SLP vectoriser), so every object's device assembly
# is checked and the build fails if one appears.  tools/pk_opsel_mfma_probe.hip classifies the forms beside MFMA-issuing
# neighbours: only the src1 select of the packed MULTIPLIES (v_pk_mul_f32, v_pk_fma_f32) goes wrong; src0 / src2 selects, the
# op_sel_hi forms and v_pk_add_f32 are clean.  The rule below is kept a little wider (src1 or src2, add included).
AUDIT_VERSION = 2
_PK_F32 = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\b.*\bop_sel:\[([01]),([01])(?:,([01]))?\]")


def audit_isa(text, what):
    kernel, bad = "?", []
    for line in text.splitlines():
        if re.match(r"^[A-Za-z_][\w$]*:", line):  # a function symbol (basic-block labels start with a dot)
            kernel = line.split(":")[0]
        m = _PK_F32.search(line)
        if m and (m.group(3) == "1" or m.group(4) == "1"):
            bad.append("%s: %s" % (kernel, line.strip()))
    if bad and not os.environ.get("R2L_ALLOW_PK_OPSEL"):
        raise RuntimeError("ISA audit of %s: packed fp32 op with a low-lane src1/src2 op_sel (unsafe on gfx950 with >1 wave per "
                           "SIMD, see r2l_amd/build.py):\n  %s" % (what, "\n  ".join(bad[:8])))
    return bad


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    stamp = obj + ".sha1"
    dig = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    # -save-temps=obj: the device assembly of this very compilation lands next to the object (audited, then removed)
    cmd = [HIPCC] + FLAGS + ["-save-temps=obj", "-I", os.path.join(HERE, "..", "include"), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    base = os.path.join(OBJDIR, src.replace(".hip", ""))
    temps = [t for t in glob.glob(base + "-hip-*") + glob.glob(base + "-host-*") + glob.glob(base + ".hip-hip-*")]
    try:
        asm = [t for t in temps if t.endswith(".s") and "-hip-amdgcn" in t]
        if not asm:
            raise RuntimeError("ISA audit: no device assembly produced for %s" % src)
        for t in asm:
            with open(t) as f:
                audit_isa(f.read(), src)
    finally:
        for t in temps:
            os.remove(t)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def _require_zlib():
    """csrc/r2l_png.hip (the native PNG encoder threads of the test-set loop) is part of the one library and needs zlib's
    development files; say so by name instead of failing somewhere inside 24 hipcc jobs."""
    probe = subprocess.run([HIPCC, "-x", "c++", "-E", "-"], input="#include <zlib.h>\n", capture_output=True, text=True)
    if probe.returncode != 0:
        raise RuntimeError("r2l_amd.build: <zlib.h> not found — libr2l_hip.so links zlib (-lz) for csrc/r2l_png.hip; install the "
                           "zlib development package (zlib1g-dev / zlib-devel)")


def build(verbose=True, force=False):
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    srcs = _sources()
    if any(not os.path.exists(os.path.join(OBJDIR, f.replace(".hip", ".o"))) for f in srcs):
        _require_zlib()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results) or not os.path.exists(LIB)
    if rebuilt:
        tmp = "%s.%d.tmp" % (LIB, os.getpid())  # link aside, then rename: other ranks never see a half-written .so
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-lz"]  # zlib: csrc/r2l_png.hip
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        os.replace(tmp, LIB)
    if verbose:
        print("[r2l_amd.build] %s (%s)" % (LIB, "rebuilt" if rebuilt else "up to date"))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
