"""ctypes binding of libr2l_hip.so (C ABI declared in include/r2l_hip.h).

There is deliberately NO fallback: if the shared library is missing or an export is absent the import of the
HIP path fails loudly (RuntimeError), so a GPU run can never silently route through PyTorch ops or the oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("R2L_LIB_PATH") or os.path.join(_HERE, "lib", "libr2l_hip.so")  # env override: A/B builds

_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_int64
_f = ctypes.c_float



class Config(ctypes.Structure):
    """r2l_config of include/r2l_hip.h: explicit dispatch for the *_cfg entry points (all zero = AUTO)."""
    _fields_ = [("precision", _i), ("tiling", _i), ("coop_tiles", _i), ("reserve_cus", _i), ("dw_mode", _i),
                ("reserved", _i * 3)]


PRECISION = {"auto": 0, "fp16x2": 1, "bf16x3": 2, "fp32_mfma": 3}
TILING = {"auto": 0, "main": 1, "coop16": 3, "coopf": 4}  # (2 = "coop", the 32-ray fp32-MFMA cooperative family: retired in round 5)
DW_MODE = {"auto": 0, "fp16": 1, "exact": 2}
_cfgp = ctypes.POINTER(Config)


def make_config(precision="auto", tiling="auto", coop_tiles=0, reserve_cus=0, dw_mode="auto"):
    if tiling == "coop":
        raise ValueError("tiling 'coop' (the 32-ray fp32-MFMA cooperative kernels) was retired in round 5: every *_cfg call would "
                         "fail with it; 'coop16' serves those launches")
    return Config(PRECISION[precision], TILING[tiling], int(coop_tiles), int(reserve_cus), DW_MODE[dw_mode])


# name -> (restype, argtypes); one row per declaration in include/r2l_hip.h
SIGNATURES = {
    "r2l_last_error": (ctypes.c_char_p, []),
    "r2l_param_count": (_l, [_i]),
    "r2l_fwd_stream_floats": (_l, [_i]),
    "r2l_bwd_stream_floats": (_l, [_i]),
    "r2l_pack_forward": (_i, [_p, _i, _p, _p]),
    "r2l_pack_backward": (_i, [_p, _i, _p, _p]),
    "r2l_variant_for": (_i, [_l]),
    "r2l_coop_tiles_for": (_i, [_l, _i]),
    "r2l_forward_layout_for": (_i, [_l, _i]),
    "r2l_backward_layout_for": (_i, [_l]),
    "r2l_pack_forward_layout": (_i, [_p, _i, _p, _i, _p]),
    "r2l_pack_backward_layout": (_i, [_p, _i, _p, _i, _p]),
    "r2l_variant_for_cfg": (_i, [_l, _cfgp]),
    "r2l_coop_tiles_for_cfg": (_i, [_l, _i, _cfgp]),
    "r2l_forward_layout_for_cfg": (_i, [_l, _i, _cfgp]),
    "r2l_backward_layout_for_cfg": (_i, [_l, _cfgp]),
    "r2l_forward_rays": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _l, _p]),
    "r2l_forward_pose": (_i, [_p, _i, _i, _f, _p, _p, _p, _i, _p, _p]),
    "r2l_forward_rays_cfg": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _l, _p, _cfgp]),
    "r2l_forward_pose_cfg": (_i, [_p, _i, _i, _f, _p, _p, _p, _i, _p, _p, _cfgp]),
    "r2l_forward_poses_cfg": (_i, [_p, _i, _i, _i, _f, _p, _p, _p, _i, _p, _p, _cfgp]),
    "r2l_forward_emb": (_i, [_p, _p, _p, _i, _p, _p, _p, _l, _p]),
    "r2l_forward_emb_cfg": (_i, [_p, _p, _p, _i, _p, _p, _p, _l, _p, _p, _cfgp]),
    "r2l_num_tiles": (_l, [_l]),
    "r2l_padded_rows": (_l, [_l]),
    "r2l_stash_slot_floats": (_l, [_l]),
    "r2l_dw_slab_floats": (_l, []),
    "r2l_backward": (_i, [_p] * 12 + [_i, _f] + [_p] * 6 + [_l, _p]),
    "r2l_backward_part": (_i, [_p] * 12 + [_i, _f] + [_p] * 6 + [_l, _p, _i, _i, _i]),
    "r2l_backward_part_cfg": (_i, [_p] * 12 + [_i, _f] + [_p] * 6 + [_l, _p, _i, _i, _i, _cfgp]),
    "r2l_allreduce_unique_id": (_i, [_p]),
    "r2l_allreduce_init": (_i, [_p, _i, _i, _p]),
    "r2l_grad_allreduce": (_i, [_p, _p, _l, _p]),
    "r2l_allreduce_destroy": (_i, [_p]),
    "r2l_adam_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _i, _f, _p]),
    "r2l_adam_step_guarded": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _i, _f, _p, _p]),
    "r2l_adam_step_packed": (_i, [_p, _p, _p, _p, _i, _f, _f, _f, _f, _i, _f, _p, _p, _p, _p]),
    "r2l_chain_segments_ok_cfg": (_i, [_l, _i, _cfgp]),
    "r2l_backward_status_word": (_p, [_p, _i]),
    "r2l_forward_status_words": (_p, [_p, _i]),
    "r2l_backward_status_words": (_p, [_p, _i]),
    "r2l_teacher_status_words": (_p, [_p]),
    "r2l_loss_finish": (_i, [_p, _l, _f, _p, _p]),
    "r2l_teacher_param_count": (_l, []),
    "r2l_teacher_stream_floats": (_l, []),
    "r2l_pack_teacher": (_i, [_p, _p, _p]),
    "r2l_teacher_mlp": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _p]),
    "r2l_teacher_mlp_cfg": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _p, _cfgp]),
    "r2l_stratified_z": (_i, [_p, _p, _i, _p, _p, _p, _l, _i, _p]),
    "r2l_raw2outputs": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _l, _i, _p]),
    "r2l_sample_pdf_sort": (_i, [_p, _p, _p, _l, _p, _p, _p, _l, _i, _i, _p]),
    "r2l_ssim_partial_count": (_l, [_i, _i, _i]),
    "r2l_ssim": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "r2l_pool_pick": (_i, [_p, _l, _l, ctypes.c_uint64, _p]),
    "r2l_pool_augment": (_i, [_p, _p, _p, _l, _l, _l, _p, _p, _l, _l, _p, _p, _p, _p]),
    "r2l_pool_store": (_i, [_p, _p, _p, _l, _l, _l, _p, _p, _p, _l, _l, _p]),
    "r2l_png_writer_open": (_i, [_i, _i, _p]),
    "r2l_png_writer_submit": (_i, [_p, ctypes.c_char_p, _p, _i, _i, _i, _p, _p]),
    "r2l_png_writer_wait": (_i, [_p, _l]),
    "r2l_png_writer_close": (_i, [_p]),
    "r2l_npy_shape": (_i, [ctypes.c_char_p, _p, _p]),
    "r2l_reader_open": (_i, [_p, _l, _i, _i, ctypes.c_uint64, _p, _i, _p]),
    "r2l_reader_info": (_i, [_p, _p, _p, _p]),
    "r2l_reader_next": (_i, [_p, _p]),
    "r2l_reader_release": (_i, [_p, _i]),
    "r2l_reader_close": (_i, [_p]),
}

# range control of the fp16 kernels (include/r2l_hip.h "range control"; csrc/r2l_common.h F2S_* / B2S_*): the word that marks a
# status area as initialised, and the decoding of the forward / teacher area's 16 words
RANGE_MAGIC = 0x52324c34


def decode_range_words(w):
    """w: the 16 status words as an int32 CPU tensor -> {'amax', 'scale', 'headroom', 'trips', 'rescales', 'flag'}."""
    import torch
    f = w.view(torch.float32)
    scale = float(f[2]) if int(w[4]) == RANGE_MAGIC else 1.0
    live = float(f[1]) * scale
    amax = live if live > 0 else float(f[6])
    return {"amax": amax, "scale": scale, "headroom": (32768.0 * scale / amax) if amax > 0 else float("inf"),
            "trips": int(w[5]), "rescales": int(w[7]), "flag": int(w[0])}


# stage bits of r2l_backward_part (include/r2l_hip.h)
BWD_CHAIN, BWD_BODY, BWD_HEAD, BWD_TAIL, BWD_ALL, BWD_NOFALLBACK = 1, 2, 4, 8, 15, 16

_lib = None


def load():
    """Load the library once and bind every export; raises RuntimeError if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own libamdhip64 (same soname as /opt/rocm's).  It must be the one the process loads FIRST:
    # if libr2l_hip.so pulled in the system runtime before torch was imported, torch would bind to that copy and fail
    # with "no ROCm-capable device is detected".  One HIP runtime per process: torch's.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libr2l_hip.so not found at %s — build it with `python -m r2l_amd.build` "
            "(or __graft_entry__.build()); there is no non-HIP fallback for the R2L hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError("libr2l_hip.so lacks export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().r2l_last_error()
        raise RuntimeError("%s failed: hip error %d: %s" % (what, code, msg.decode() if msg else "?"))
