"""Host-side engine: owns the flat parameter buffer, the packed MFMA weight streams and the scratch tensors, and
dispatches the R2L student hot path to libr2l_hip.so (include/r2l_hip.h).  PyTorch is plumbing here: device memory,
streams, nn.Parameter views; every FLOP of the path runs in the hand-written HIP kernels.

The engine is attached lazily to a NeRF_v3_2 module (r2l_amd/nerf_raybased.py) — including modules restored by
un-pickling a reference checkpoint, whose __init__ never ran.
"""
import ctypes

import torch

from . import _lib

W = 256
N_SAMPLE = 16
L_PE = 10
INPUT_DIM = N_SAMPLE * 3 * (2 * L_PE + 1)  # 1008


# Process-wide defaults for the AUTO (0) fields of every engine's r2l_config (make_config keywords: precision, tiling,
# coop_tiles, reserve_cus, dw_mode).  Empty = the library decides.  A field an engine sets itself (set_config) wins.  This is how
# the test suites and bench.py select kernel families — through ARGUMENTS of the *_cfg entry points, not the environment.
DEFAULT_CONFIG = {}


def merged_config(cfg):
    """cfg with its AUTO fields filled from DEFAULT_CONFIG (read at call time)."""
    if not DEFAULT_CONFIG:
        return cfg
    base = _lib.make_config(**DEFAULT_CONFIG)
    return _lib.Config(cfg.precision or base.precision, cfg.tiling or base.tiling, cfg.coop_tiles or base.coop_tiles,
                       cfg.reserve_cus or base.reserve_cus, cfg.dw_mode or base.dw_mode)


def arithmetic(cfg=None):
    """{'precision': 'fp16x2'|'bf16x3'|'fp32_mfma', 'dw_mode': 'fp16'|'exact'}: the arithmetic the launches made with `cfg`
    (an r2l_config, AUTO fields filled from DEFAULT_CONFIG) run on, AUTO resolved the way the library resolves it
    (csrc/r2l_common.h r2l_use_fwd3 / r2l_use_fwd2 / r2l_use_trio16 / r2l_dw_exact: the R2L_NO_* / R2L_DW_EXACT environment
    switches an AUTO field falls through to).  What the drivers log and store in the checkpoints they write."""
    import os
    cfg = merged_config(cfg if cfg is not None else _lib.Config())
    on = lambda k: os.environ.get(k, "")[:1] not in ("", "0")  # r2l_env_on
    names = {v: k for k, v in _lib.PRECISION.items()}
    if cfg.precision:
        prec = names[cfg.precision]
    elif on("R2L_NO_FWD3"):
        prec = "fp32_mfma"
    elif on("R2L_NO_FWD2") or on("R2L_NO_BWD2") or on("R2L_NO_DW2"):
        prec = "bf16x3"  # (training steps; forward-only launches look at R2L_NO_FWD2 alone)
    else:
        prec = "fp16x2"
    if prec != "fp16x2":
        dw = "exact"  # the fp32-MFMA / bf16x3 weight-gradient GEMMs carry fp32-grade operands
    elif cfg.dw_mode:
        dw = {v: k for k, v in _lib.DW_MODE.items()}[cfg.dw_mode]
    else:
        dw = "exact" if on("R2L_DW_EXACT") else "fp16"
    return {"precision": prec, "dw_mode": dw}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def supported_reason(module):
    """None if the HIP chain kernels implement this NeRF_v3_2 configuration, else the reason they do not."""
    args = getattr(module, "args", None)
    try:
        head, tail = module.head[0], module.tail[0]
        blocks = list(module.body)
    except Exception as e:  # not the ResMLP architecture
        return "unexpected module structure: %r" % (e,)
    if head.in_features != INPUT_DIM or head.out_features != W:
        return "head must be Linear(%d,%d)" % (INPUT_DIM, W)
    if not isinstance(tail, torch.nn.Linear) or tail.in_features != W or tail.out_features != 3:
        return "tail must be Linear(256,3)+Sigmoid"
    if args is None or not getattr(args, "use_residual", False):
        return "--use_residual is required"
    if getattr(args, "act", "relu").lower() != "relu":
        return "only act=relu"
    for b in blocks:
        if type(b).__name__ != "ResMLP" or len(b.body) != 3 or b.outact is not None or float(b.res_scale) != 1.0:
            return "body must be ResMLP(256) blocks with n_learnable=2, res_scale=1, outact=none"
        if not isinstance(b.body[1], torch.nn.ReLU):
            return "only --trial.inact relu (the chain kernels apply a hard-coded ReLU inside a block)"
        if b.body[0].in_features != W or b.body[0].out_features != W:
            return "ResMLP width must be 256"
    return None


class R2LEngine:
    """Flat parameters + packed weight streams for one NeRF_v3_2 on one GPU."""

    def __init__(self, module):
        reason = supported_reason(module)
        if reason is not None:
            raise NotImplementedError("R2L HIP path does not implement this configuration: " + reason)
        self.lib = _lib.load()
        self.module = module
        self.n_block = len(module.body)
        self.params = [p for _, p in module.named_parameters()]
        self.n_param = sum(p.numel() for p in self.params)
        if self.n_param != self.lib.r2l_param_count(self.n_block):
            raise RuntimeError("parameter census mismatch: %d vs %d" % (self.n_param, self.lib.r2l_param_count(
                self.n_block)))
        self.device = None
        self.flat = None
        self.wstream = None
        self._packed_version = None
        self._status = None
        self._dirty = 0
        self._ztab_cache = {}
        # explicit dispatch handed to every call of this engine (include/r2l_hip.h r2l_config; all zero = AUTO: the library
        # picks by ray count, R2L_* environment switches still apply).  set_config() changes it.
        self.cfg = _lib.Config()

    def set_config(self, **kw):
        """precision='auto|fp16x2|bf16x3|fp32_mfma', tiling='auto|main|coop16|coopf', coop_tiles=0|1|2|3 (mixed),
        reserve_cus=n, dw_mode='auto|fp16|exact'.  Weight streams are layout-specific: they re-pack on the next call."""
        cur = dict(precision=self.cfg.precision, tiling=self.cfg.tiling, coop_tiles=self.cfg.coop_tiles,
                   reserve_cus=self.cfg.reserve_cus, dw_mode=self.cfg.dw_mode)
        names = {"precision": _lib.PRECISION, "tiling": _lib.TILING, "dw_mode": _lib.DW_MODE}
        for k, v in kw.items():
            if k not in cur:
                raise TypeError("unknown config field %r" % k)
            cur[k] = names[k][v] if k in names and isinstance(v, str) else int(v)
        self.cfg = _lib.Config(cur["precision"], cur["tiling"], cur["coop_tiles"], cur["reserve_cus"], cur["dw_mode"])
        self._packed_version = None
        return self.cfg

    def effective_config(self):
        """The r2l_config this engine's calls are made with: its own fields, AUTO ones filled from DEFAULT_CONFIG."""
        return merged_config(self.cfg)

    def _cfg(self):
        self._eff = self.effective_config()  # (kept alive for the duration of the call)
        return ctypes.byref(self._eff)

    # ---- parameter storage ------------------------------------------------------------------------------------
    def _aliased(self):
        if self.flat is None:
            return False
        off = self.flat.data_ptr()
        for p in self.params:
            if p.data_ptr() != off or p.dtype != torch.float32:
                return False
            off += p.numel() * 4
        return True

    def flatten(self, device=None):
        """Move every parameter into ONE flat fp32 buffer (state_dict order) and re-point p.data at views of it."""
        device = torch.device(device) if device is not None else self.params[0].device
        if device.type != "cuda":
            raise RuntimeError("the R2L HIP engine needs parameters on a cuda (ROCm) device, got %s" % device)
        flat = torch.empty(self.n_param, dtype=torch.float32, device=device)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                view = flat[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                off += n
        self.flat = flat
        self.device = device
        # zero-filled: the status words inside carry the range-control history (include/r2l_hip.h); recycled memory of a
        # previous engine would be taken for this one's
        self.wstream = torch.zeros(self.lib.r2l_fwd_stream_floats(self.n_block), dtype=torch.float32, device=device)
        self._packed_version = None
        self._status = None

    def mark_dirty(self):
        """Call after writing the parameters behind autograd's back (fused Adam through the C ABI)."""
        self._dirty += 1

    def version(self):
        """Changes whenever the parameters may have changed (autograd version counters + mark_dirty)."""
        return sum(p._version for p in self.params) + self._dirty

    def layout_for(self, n, with_stash=True):
        """Which part of the packed forward stream a launch with n rays reads: 16 (16-ray cooperative kernels), 32
        (32-ray cooperative / fp32-MFMA kernels), 3 (the bf16x3 kernels) or 2 (fp16x2 forward-only kernel, R2L_FWD2=1)."""
        return self.lib.r2l_forward_layout_for_cfg(int(n), 1 if with_stash else 0, self._cfg())

    def pack_now(self):
        """Unconditional re-pack of both layouts (used inside captured graphs, where the host-side version check does
        not replay)."""
        _lib.check(self.lib.r2l_pack_forward(_ptr(self.flat), self.n_block, _ptr(self.wstream), _stream()),
                   "r2l_pack_forward")
        self._packed_version = {16: self.version(), 32: self.version(), 3: self.version(), 2: self.version()}

    def ensure_packed(self, n=None, with_stash=True):
        """Re-pack the weight stream if the parameters changed since it was packed.  With n given only the layout that a
        launch with n rays reads (the other parts stay stale until someone asks for them)."""
        if not self._aliased():
            self.flatten(self.params[0].device)
        ver = self.version()
        if self._packed_version is None:
            self._packed_version = {16: None, 32: None, 3: None, 2: None}
        for layout in ((16, 32, 3, 2) if n is None else (self.layout_for(n, with_stash),)):
            if self._packed_version[layout] != ver:
                _lib.check(self.lib.r2l_pack_forward_layout(_ptr(self.flat), self.n_block, _ptr(self.wstream), layout,
                                                            _stream()), "r2l_pack_forward_layout")
                self._packed_version[layout] = ver

    # ---- range control telemetry (include/r2l_hip.h: "range control of the fp16 kernels") ---------------------------------
    def status_words(self):
        """The 16 status words of the fp16x2 forward stream as an int32 view of self.wstream (device; no sync)."""
        if self.wstream is None or not self._aliased():  # an engine that has not been flattened / packed yet: neutral telemetry
            self.ensure_packed()
        if self._status is None or self._status.data_ptr() < self.wstream.data_ptr():
            word = ctypes.cast(self.lib.r2l_forward_status_words(_ptr(self.wstream), self.n_block), ctypes.c_void_p).value
            off = (word - self.wstream.data_ptr()) // 4
            self._status = self.wstream[off:off + 16].view(torch.int32)
        return self._status

    def reset_range_history(self):
        """Forget the recorded amax / scale (as after allocation): the next pack starts again from s = 1."""
        self.status_words().zero_()
        self._packed_version = None

    def range_info(self):
        """Where the fp16 forward kernels stand with this model's activations (synchronises: a 64-byte copy).
        amax: largest |activation| the launches have seen (current scale epoch, else the previous one); scale: the power of
        two the stream is packed for; headroom: 32768 * scale / amax (x-fold growth the fp16 kernels still take before a
        launch has to be redone by the bf16x3 kernel); trips: launches that were redone; rescales: times the scale changed."""
        return _lib.decode_range_words(self.status_words().cpu())

    # ---- sampler tables ---------------------------------------------------------------------------------------
    def ztab(self, z_vals, perturb):
        """[32] = z_lower[16] ++ z_span[16] for PointSampler.z_vals (model/nerf_raybased.py:115-123), on device."""
        key = (z_vals.data_ptr(), bool(perturb > 0), z_vals._version)
        tab = self._ztab_cache.get(key)
        if tab is None:
            z = z_vals.detach().float().cpu()
            if z.numel() != N_SAMPLE:
                raise NotImplementedError("HIP path is compiled for n_sample_per_ray = 16")
            if perturb > 0:
                mids = .5 * (z[1:] + z[:-1])
                upper = torch.cat([mids, z[-1:]])
                lower = torch.cat([z[:1], mids])
                tab = torch.cat([lower, upper - lower])
            else:
                tab = torch.cat([z, torch.zeros_like(z)])
            tab = tab.to(self.device).contiguous()
            self._ztab_cache = {key: tab}
        return tab

    # ---- forward entry points ---------------------------------------------------------------------------------
    def forward_rays(self, rays_o, rays_d, z_vals, perturb=0., t_rand=None, save=None):
        """rgb[N,3] from rays; t_rand [N,16] U[0,1) is drawn here when perturb > 0 and none is given."""
        rays_o = rays_o.contiguous().float()
        rays_d = rays_d.contiguous().float()
        n = rays_o.shape[0]
        self.ensure_packed(n, with_stash=save is not None)
        if perturb > 0 and t_rand is None:
            t_rand = torch.rand(n, N_SAMPLE, device=self.device)
        if perturb <= 0:
            t_rand = None
        if t_rand is not None:
            t_rand = t_rand.contiguous().float()
        rgb = torch.empty(n, 3, dtype=torch.float32, device=self.device)
        sx, st = (save if save is not None else (None, None))
        _lib.check(
            self.lib.r2l_forward_rays_cfg(_ptr(rays_o), _ptr(rays_d), _ptr(t_rand), _ptr(self.ztab(z_vals, perturb)),
                                          _ptr(self.wstream), _ptr(self.flat), self.n_block, _ptr(rgb), _ptr(sx), _ptr(st),
                                          n, _stream(), self._cfg()), "r2l_forward_rays")
        return rgb

    def forward_pose(self, c2w, H, Wimg, focal, z_vals):
        """rgb[H*W,3] for the frame seen from c2w[3,4] (PointSampler.sample_test fused in front of the chain)."""
        self.ensure_packed(int(H) * int(Wimg), with_stash=False)
        c = torch.as_tensor(c2w, dtype=torch.float32).detach().cpu()[:3, :4].contiguous()
        host = (ctypes.c_float * 12)(*c.reshape(-1).tolist())
        rgb = torch.empty(H * Wimg, 3, dtype=torch.float32, device=self.device)
        _lib.check(
            self.lib.r2l_forward_pose_cfg(ctypes.cast(host, ctypes.c_void_p), int(H), int(Wimg), float(focal),
                                          _ptr(self.ztab(z_vals, 0.)), _ptr(self.wstream), _ptr(self.flat), self.n_block,
                                          _ptr(rgb), _stream(), self._cfg()), "r2l_forward_pose")
        return rgb

    def forward_poses(self, c2ws, H, Wimg, focal, z_vals):
        """rgb[K, H*W, 3] for the K frames seen from c2ws[K,3(+),4] in ONE launch (r2l_forward_poses_cfg): no launch gap and
        no partly filled last round of workgroups per frame.  Pinned cooperative tilings: frame by frame."""
        c = torch.as_tensor(c2ws, dtype=torch.float32)[:, :3, :4]
        K = int(c.shape[0])
        n = K * int(H) * int(Wimg)
        self.ensure_packed(n, with_stash=False)
        if K == 0:
            return torch.empty(0, H * Wimg, 3, dtype=torch.float32, device=self.device)
        if self.lib.r2l_variant_for_cfg(n, self._cfg()) != 0 or self.lib.r2l_coop_tiles_for_cfg(n, self.n_block, self._cfg()):
            return torch.stack([self.forward_pose(c[k], H, Wimg, focal, z_vals) for k in range(K)], 0)
        cd = c.to(self.device).contiguous()
        rgb = torch.empty(K, H * Wimg, 3, dtype=torch.float32, device=self.device)
        _lib.check(
            self.lib.r2l_forward_poses_cfg(_ptr(cd), K, int(H), int(Wimg), float(focal), _ptr(self.ztab(z_vals, 0.)),
                                           _ptr(self.wstream), _ptr(self.flat), self.n_block, _ptr(rgb), _stream(),
                                           self._cfg()), "r2l_forward_poses")
        return rgb

    def forward_emb(self, emb, save=None):
        """rgb[N,3] from the already embedded input [N,1008] (nn.Module-boundary compatibility path)."""
        self.ensure_packed()
        emb = emb.contiguous().float()
        if emb.shape[-1] != INPUT_DIM:
            raise ValueError("expected [N,%d] embedded input, got %s" % (INPUT_DIM, tuple(emb.shape)))
        n = emb.shape[0]
        rgb = torch.empty(n, 3, dtype=torch.float32, device=self.device)
        sx, st = (save if save is not None else (None, None))
        # forward-only launches follow the engine's precision: bf16x3 (fp16x2 alike) = head on the fp32 MFMA, body + tail on the
        # bf16x3 chain from X_0 in a scratch buffer (include/r2l_hip.h r2l_forward_emb_cfg); with the stash, or AUTO / fp32_mfma:
        # the exact-fp32 kernel throughout (its backward reads a row-major fp32 stash)
        x0 = None
        if save is None and n > 0 and self.effective_config().precision in (_lib.PRECISION["bf16x3"], _lib.PRECISION["fp16x2"]):
            x0 = torch.empty(int(self.lib.r2l_padded_rows(n)) * W, dtype=torch.float32, device=self.device)
        _lib.check(
            self.lib.r2l_forward_emb_cfg(_ptr(emb), _ptr(self.wstream), _ptr(self.flat), self.n_block, _ptr(rgb), _ptr(sx),
                                         _ptr(st), n, _ptr(x0), _stream(), self._cfg()), "r2l_forward_emb")
        return rgb


def get_engine(module):
    """The module's engine, created on first use (works for un-pickled modules: state lives outside __dict__ keys
    that pickle would try to serialise — see NeRF_v3_2.__getstate__)."""
    eng = module.__dict__.get("_r2l_engine")
    if eng is None:
        eng = R2LEngine(module)
        module.__dict__["_r2l_engine"] = eng
    return eng
