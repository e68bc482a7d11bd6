"""r2l_amd — MI355X (gfx950) native implementation of the R2L hot path.

csrc/            hand-written HIP kernels + the C ABI (include/r2l_hip.h) -> lib/libr2l_hip.so
engine.py        flat parameters, packed MFMA weight streams, kernel dispatch
nerf_raybased.py host-side mirror of the reference's model layer (exported as model.nerf_raybased)
"""
__all__ = ["build", "engine", "nerf_raybased"]
