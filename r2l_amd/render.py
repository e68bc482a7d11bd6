"""Host-side mirror of the reference's render/ray layer for the NeRF teacher (pseudo-data generation):
get_rays / get_embedder / raw2outputs / sample_pdf / run_network / render_rays / batchify_rays / render, with the
reference's names, argument meaning and return layout (/root/reference/utils/create_data.py:41-177,335-544 and
utils/run_nerf_raybased_helpers.py:24-74,231-330).

On ROCm tensors every stage runs in libr2l_hip.so (fused embed+MLP, wave-scan alpha compositing, on-GPU inverse-CDF
sampling + sort: no CPU round trip, no netchunk loop); on CPU tensors the same functions run plain torch ops
(plumbing only).
"""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from . import engine as _engine
from .engine import _ptr, _stream

# ---------------------------------------------------------------------------------------------------------------
# rays + embedders
# ---------------------------------------------------------------------------------------------------------------


def get_rays(H, W, focal, c2w, trans_origin="", focal_scale=1):
    """rays_o, rays_d [H,W,3] of a pinhole camera (helpers:231-257; trans_origin variants are out of scope)."""
    if trans_origin:
        raise NotImplementedError("trans_origin is not part of the accelerated path")
    focal = focal * focal_scale
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    dev = c2w.device
    cols = torch.arange(W, dtype=torch.float32, device=dev).expand(H, W)
    rows = torch.arange(H, dtype=torch.float32, device=dev).unsqueeze(1).expand(H, W)
    dirs = torch.stack([(cols - W * .5) / focal, -(rows - H * .5) / focal, -torch.ones(H, W, device=dev)], -1)
    rays_d = (dirs.unsqueeze(-2) * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays_np(H, W, focal, c2w):
    o, d = get_rays(H, W, float(focal), torch.as_tensor(np.asarray(c2w), dtype=torch.float32))
    return o.numpy(), d.numpy()


class Embedder:
    """NeRF positional encoding [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] (helpers:24-56)."""

    def __init__(self, multires, input_dims=3):
        self.freqs = 2.**torch.linspace(0., multires - 1, steps=multires)
        self.out_dim = input_dims * (2 * multires + 1)

    def embed(self, x):
        out = [x]
        for f in self.freqs.tolist():
            out += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(out, -1)


def get_embedder(multires, i=0):
    if i == -1:
        return torch.nn.Identity(), 3
    e = Embedder(multires)
    return e.embed, e.out_dim


# ---------------------------------------------------------------------------------------------------------------
# teacher engine: flat parameters + packed stream of one NeRF module
# ---------------------------------------------------------------------------------------------------------------
class TeacherEngine:
    def __init__(self, module):
        self.lib = _lib.load()
        ok = (getattr(module, "use_viewdirs", False) and module.D == 8 and module.W == 256 and module.input_ch == 63
              and module.input_ch_views == 27 and list(module.skips) == [4])
        if not ok:
            raise NotImplementedError("teacher HIP path implements NeRF(D=8, W=256, 63+27, skips=[4], use_viewdirs)")
        self.module = module
        self.params = [p for _, p in module.named_parameters()]
        n = sum(p.numel() for p in self.params)
        if n != self.lib.r2l_teacher_param_count():
            raise RuntimeError("teacher parameter census mismatch: %d" % n)
        self.flat = None
        self._ver = None
        self.cfg = _lib.Config()  # r2l_config of this teacher's calls (only .precision matters); all zero = AUTO

    def set_config(self, precision="auto"):
        """Kernel family of this teacher's point-network launches: 'auto|fp16x2|bf16x3|fp32_mfma' (r2l_config.precision)."""
        self.cfg = _lib.make_config(precision=precision)
        return self.cfg

    def _aliased(self):
        if self.flat is None:
            return False
        off = self.flat.data_ptr()
        for p in self.params:
            if p.data_ptr() != off:
                return False
            off += p.numel() * 4
        return True

    def ensure_packed(self):
        if not self._aliased():
            dev = self.params[0].device
            flat = torch.empty(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for p in self.params:
                    n = p.numel()
                    v = flat[off:off + n].view(p.shape)
                    v.copy_(p.data)
                    p.data = v
                    off += n
            self.flat = flat
            # (zero-filled: the status words inside carry the range-control history, include/r2l_hip.h)
            self.wstream = torch.zeros(self.lib.r2l_teacher_stream_floats(), dtype=torch.float32, device=dev)
            self._ver = None
        ver = sum(p._version for p in self.params)
        if ver != self._ver:
            _lib.check(self.lib.r2l_pack_teacher(_ptr(self.flat), _ptr(self.wstream), _stream()), "r2l_pack_teacher")
            self._ver = ver

    def mlp(self, rays_o, rays_d, viewdirs, z):
        """raw[R,S,4] for the points o + d*z[R,S]."""
        self.ensure_packed()
        R, S = z.shape
        raw = torch.empty(R, S, 4, dtype=torch.float32, device=z.device)
        _lib.check(
            self.lib.r2l_teacher_mlp_cfg(_ptr(rays_o.contiguous()), _ptr(rays_d.contiguous()), _ptr(viewdirs.contiguous()),
                                         _ptr(z.contiguous()), _ptr(self.wstream), _ptr(self.flat), _ptr(raw), R, S,
                                         _stream(), ctypes.byref(_engine.merged_config(self.cfg))), "r2l_teacher_mlp")
        return raw


    def range_info(self):
        """Range control of the fp16 teacher kernel (include/r2l_hip.h; as R2LEngine.range_info; synchronises)."""
        self.ensure_packed()
        word = ctypes.cast(self.lib.r2l_teacher_status_words(_ptr(self.wstream)), ctypes.c_void_p).value
        off = (word - self.wstream.data_ptr()) // 4
        return _lib.decode_range_words(self.wstream[off:off + 16].view(torch.int32).cpu())


def teacher_engine(module):
    module = getattr(module, "module", module)  # tolerate a DataParallel-style wrapper
    eng = module.__dict__.get("_r2l_teacher_engine")
    if eng is None:
        eng = TeacherEngine(module)
        module.__dict__["_r2l_teacher_engine"] = eng
    return eng


def run_network(inputs, viewdirs, fn, embed_fn=None, embeddirs_fn=None, netchunk=1024 * 64):
    """raw = fn(cat[embed(pts), embed(dirs)])   (create_data.py:55-77).  CPU tensors only; the GPU path is fused in
    render_rays (points are never materialised)."""
    flat = inputs.reshape(-1, inputs.shape[-1])
    emb = embed_fn(flat)
    if viewdirs is not None:
        dirs = viewdirs[:, None].expand(inputs.shape).reshape(-1, inputs.shape[-1])
        emb = torch.cat([emb, embeddirs_fn(dirs)], -1)
    out = torch.cat([fn(emb[i:i + netchunk]) for i in range(0, emb.shape[0], netchunk)], 0)
    return out.reshape(list(inputs.shape[:-1]) + [out.shape[-1]])


# ---------------------------------------------------------------------------------------------------------------
# raw2outputs / sample_pdf
# ---------------------------------------------------------------------------------------------------------------
def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, noise=None, need_weights=True,
                **_ignored):
    """(rgb_map[R,3], disp_map[R], acc_map[R], weights[R,S], depth_map[R])   (create_data.py:335-402).
    `noise` [R,S] (already scaled) replaces the internal draw of the reference when given (tests: same draw on both sides).
    need_weights=False (GPU path): the [R,S] weights are not written to HBM and None is returned in their place — the fine
    pass of render_rays never reads them (4 S of the 24 S + 36 bytes per ray the kernel moves)."""
    if noise is not None:
        noise = noise.to(raw.device).float().contiguous()
    elif raw_noise_std > 0.:
        if pytest:
            np.random.seed(0)
            noise = torch.Tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std)
        else:
            noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std
        noise = noise.to(raw.device)
    if not raw.is_cuda:
        dists = z_vals[..., 1:] - z_vals[..., :-1]
        dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1) * torch.norm(rays_d[..., None, :], dim=-1)
        sigma = raw[..., 3] if noise is None else raw[..., 3] + noise
        alpha = 1. - torch.exp(-F.relu(sigma) * dists)
        trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
        weights = alpha * trans
        rgb_map = torch.sum(weights[..., None] * torch.sigmoid(raw[..., :3]), -2)
        depth_map = torch.sum(weights * z_vals, -1)
        acc_map = torch.sum(weights, -1)
        disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
        if white_bkgd:
            rgb_map = rgb_map + (1. - acc_map[..., None])
        return rgb_map, disp_map, acc_map, weights, depth_map
    lib = _lib.load()
    R, S = z_vals.shape
    f = dict(dtype=torch.float32, device=raw.device)
    rgb_map, disp, acc = torch.empty(R, 3, **f), torch.empty(R, **f), torch.empty(R, **f)
    weights, depth = (torch.empty(R, S, **f) if need_weights else None), torch.empty(R, **f)
    _lib.check(
        lib.r2l_raw2outputs(_ptr(raw.contiguous()), _ptr(z_vals.contiguous()), _ptr(rays_d.contiguous()), _ptr(noise),
                            int(bool(white_bkgd)), _ptr(rgb_map), _ptr(disp), _ptr(acc), _ptr(weights), _ptr(depth), R,
                            S, _stream()), "r2l_raw2outputs")
    return rgb_map, disp, acc, weights, depth


def _uniforms(shape, N_samples, det, pytest, device=None):
    """The u of sample_pdf (helpers:291-307): linspace if det, numpy's seeded draws if pytest, else torch.rand — drawn on
    `device` (the reference runs with a CUDA default tensor type, main.py; a host draw + pageable copy per 32 768-ray
    chunk would serialise the host with the GPU)."""
    if pytest:
        np.random.seed(0)
        if det:
            return torch.Tensor(np.broadcast_to(np.linspace(0., 1., N_samples), list(shape) + [N_samples]).copy())
        return torch.Tensor(np.random.rand(*(list(shape) + [N_samples])))
    if det:
        return torch.linspace(0., 1., steps=N_samples).expand(list(shape) + [N_samples])
    return torch.rand(list(shape) + [N_samples], device=device)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None):
    """Inverse-CDF samples [R,N_samples] from the piecewise-constant pdf `weights` over `bins` (helpers:283-330).
    torch-op implementation (any device); the teacher's GPU path uses the fused r2l_sample_pdf_sort instead."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        u = _uniforms(cdf.shape[:-1], N_samples, det, pytest, device=cdf.device)
    u = u.to(cdf.device).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bins_b, bins_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bins_b + (u - cdf_b) / denom * (bins_a - bins_b)


def sample_pdf_sort(z_vals, weights, N_importance, det=False, pytest=False, u=None):
    """(z_samples[R,NI], z_all[R,S+NI] sorted, z_std[R]) — the hierarchical-sampling step of render_rays
    (create_data.py:505-515) fused on the GPU."""
    R, S = z_vals.shape
    if u is None:
        u = torch.linspace(0., 1., steps=N_importance) if (det and not pytest) else _uniforms(
            (R,), N_importance, det, pytest, device=z_vals.device)
    if not z_vals.is_cuda:
        z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        uu = u if u.dim() == 2 else u.expand(R, N_importance)
        zs = sample_pdf(z_mid, weights[..., 1:-1], N_importance, u=uu).detach()
        z_all, _ = torch.sort(torch.cat([z_vals, zs], -1), -1)
        return zs, z_all, torch.std(zs, dim=-1, unbiased=False)
    lib = _lib.load()
    u = u.to(z_vals.device).float().contiguous()
    u_stride = N_importance if u.dim() == 2 else 0
    f = dict(dtype=torch.float32, device=z_vals.device)
    zs, z_all, z_std = torch.empty(R, N_importance, **f), torch.empty(R, S + N_importance, **f), torch.empty(R, **f)
    _lib.check(
        lib.r2l_sample_pdf_sort(_ptr(z_vals.contiguous()), _ptr(weights.contiguous()), _ptr(u), u_stride, _ptr(zs),
                                _ptr(z_all), _ptr(z_std), R, S, N_importance, _stream()), "r2l_sample_pdf_sort")
    return zs, z_all, z_std


# ---------------------------------------------------------------------------------------------------------------
# render_rays / batchify_rays / render
# ---------------------------------------------------------------------------------------------------------------
def _coarse_z(near, far, N_samples, lindisp, perturb, pytest, t_rand=None):
    """z_vals[R,S] (create_data.py:457-482): near,far [R,1]."""
    R = near.shape[0]
    dev = near.device
    if lindisp:
        raise NotImplementedError("lindisp sampling is not part of the accelerated (blender) path")
    t = torch.linspace(0., 1., steps=N_samples)
    if perturb > 0. and t_rand is None:
        if pytest:
            np.random.seed(0)
            t_rand = torch.Tensor(np.random.rand(R, N_samples))
        else:
            t_rand = torch.rand(R, N_samples, device=dev)  # on the rays' device (see _uniforms)
    if perturb <= 0.:
        t_rand = None
    if not near.is_cuda:
        z = near * (1. - t) + far * t
        if t_rand is not None:
            mids = .5 * (z[..., 1:] + z[..., :-1])
            upper = torch.cat([mids, z[..., -1:]], -1)
            lower = torch.cat([z[..., :1], mids], -1)
            z = lower + (upper - lower) * t_rand
        return z.expand(R, N_samples).contiguous()
    lib = _lib.load()
    ttab = torch.cat([t, 1. - t]).to(dev)
    z = torch.empty(R, N_samples, dtype=torch.float32, device=dev)
    tr = None if t_rand is None else t_rand.to(dev).float().contiguous()
    near_c, far_c = near.contiguous(), far.contiguous()
    _lib.check(lib.r2l_stratified_z(_ptr(near_c), _ptr(far_c), 1, _ptr(ttab), _ptr(tr), _ptr(z), R, N_samples,
                                    _stream()), "r2l_stratified_z")
    return z


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                t_rand=None, u=None):
    """Volumetric rendering of ray_batch[R, 8|11] = [o, d, near, far, (viewdirs)]  (create_data.py:405-544).
    Returns the reference's dict: rgb_map, disp_map, acc_map, depth_map, (raw), and with N_importance > 0 also
    rgb0, disp0, acc0, z_std.  t_rand / u override the random draws (tests)."""
    rays_o, rays_d = ray_batch[:, 0:3].contiguous(), ray_batch[:, 3:6].contiguous()
    viewdirs = ray_batch[:, -3:].contiguous() if ray_batch.shape[-1] > 8 else None
    near, far = ray_batch[:, 6:7].contiguous(), ray_batch[:, 7:8].contiguous()
    z_vals = _coarse_z(near, far, N_samples, lindisp, perturb, pytest, t_rand)
    on_gpu = ray_batch.is_cuda

    def query(z, net):
        if on_gpu:
            if viewdirs is None:
                raise NotImplementedError("the teacher HIP path needs use_viewdirs=True")
            return teacher_engine(net).mlp(rays_o, rays_d, viewdirs, z)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
        return network_query_fn(pts, viewdirs, net)

    raw = query(z_vals, network_fn)
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd,
                                                                 pytest=pytest)
    if N_importance > 0:
        rgb0, disp0, acc0 = rgb_map, disp_map, acc_map
        z_samples, z_vals, z_std = sample_pdf_sort(z_vals, weights, N_importance, det=(perturb == 0.), pytest=pytest,
                                                   u=u)
        raw = query(z_vals, network_fn if network_fine is None else network_fine)
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd,
                                                                     pytest=pytest, need_weights=False)
    ret = {"rgb_map": rgb_map, "disp_map": disp_map, "acc_map": acc_map, "depth_map": depth_map}
    if retraw:
        ret["raw"] = raw
    if N_importance > 0:
        ret.update(rgb0=rgb0, disp0=disp0, acc0=acc0, z_std=z_std)
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """render_rays over chunks of rays (create_data.py:80-94)."""
    parts = {}
    for i in range(0, rays_flat.shape[0], chunk):
        for k, v in render_rays(rays_flat[i:i + chunk], **kwargs).items():
            parts.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in parts.items()}


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """[rgb_map, disp_map, acc_map, extras] for explicit rays [2,...,3] or a full frame from c2w
    (create_data.py:97-176).  NDC (forward-facing LLFF scenes) is out of scope."""
    if ndc:
        raise NotImplementedError("ndc rays (LLFF) are not part of the accelerated blender path; pass ndc=False")
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, focal, c2w)
    else:
        rays_o, rays_d = rays
    viewdirs = None
    if use_viewdirs:
        viewdirs = rays_d
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, focal, c2w_staticcam)
        viewdirs = (viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)).reshape(-1, 3).float()
    sh = rays_d.shape
    rays_o = rays_o.reshape(-1, 3).float()
    rays_d = rays_d.reshape(-1, 3).float()
    ones = torch.ones_like(rays_d[..., :1])
    packed = [rays_o, rays_d, near * ones, far * ones]
    if use_viewdirs:
        packed.append(viewdirs)
    all_ret = batchify_rays(torch.cat(packed, -1), chunk, **kwargs)
    for k in all_ret:
        all_ret[k] = all_ret[k].reshape(list(sh[:-1]) + list(all_ret[k].shape[1:]))
    main = ["rgb_map", "disp_map", "acc_map"]
    return [all_ret[k] for k in main] + [{k: v for k, v in all_ret.items() if k not in main}]
