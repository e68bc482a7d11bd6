"""torch.autograd bridge for the HIP student path: keeps `rgb = model(embedded); loss.backward(); optimizer.step()`
(the reference's module-boundary training idiom, main.py:1374-1406) working with any loss and any torch optimizer.
Forward = r2l_forward_emb with the activation stash, backward = r2l_backward in generic mode (dL/drgb supplied by
autograd).  The build's own training loop does not go through autograd (r2l_amd/train_step.py)."""
import torch

from . import _lib
from .engine import _ptr, _stream, get_engine


class R2LEmbFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, emb, *params):
        eng = get_engine(module)
        eng.ensure_packed()
        emb2 = emb.reshape(-1, emb.shape[-1]).contiguous().float()
        n, nb = emb2.shape[0], eng.n_block
        f = dict(dtype=torch.float32, device=emb2.device)
        slot = int(eng.lib.r2l_stash_slot_floats(n))  # include/r2l_hip.h: (n_block + 1) / n_block slots of this many floats
        save_x = torch.empty((nb + 1) * slot, **f)
        save_t = torch.empty(max(nb, 1) * slot, **f)
        rgb = eng.forward_emb(emb2, save=(save_x, save_t))
        ctx.module, ctx.n = module, n
        ctx.lead = emb.shape[:-1]
        ctx.save_for_backward(emb2, rgb, save_x, save_t)
        return rgb.reshape(*emb.shape[:-1], 3)

    @staticmethod
    def backward(ctx, grad_rgb):
        emb2, rgb, save_x, save_t = ctx.saved_tensors
        eng = get_engine(ctx.module)
        eng.ensure_packed()
        lib, n, nb = eng.lib, ctx.n, eng.n_block
        f = dict(dtype=torch.float32, device=emb2.device)
        wbwd = torch.empty(lib.r2l_bwd_stream_floats(nb), **f)
        _lib.check(lib.r2l_pack_backward(_ptr(eng.flat), nb, _ptr(wbwd), _stream()), "r2l_pack_backward")
        grads = torch.zeros(eng.n_param, **f)
        slot = int(lib.r2l_stash_slot_floats(n))
        gx = torch.empty((nb + 1) * slot, **f)
        gt = torch.empty(max(nb, 1) * slot, **f)
        dpre = torch.empty(n * 3, **f)
        slab = torch.empty(int(lib.r2l_dw_slab_floats()), **f)
        drgb = grad_rgb.reshape(-1, 3).contiguous().float()
        # The pre-embedded path always runs on the exact-fp32 MFMA kernels with a row-major stash (r2l_forward_emb has one
        # kernel); of the engine's r2l_config only `tiling` applies — which fp32 chain walks the backward — and `reserve_cus`;
        # precision / dw_mode concern the fused (ray-input) path only
        _lib.check(
            lib.r2l_backward_part_cfg(None, None, None, None, _ptr(emb2), _ptr(rgb), None, _ptr(drgb), _ptr(save_x),
                                      _ptr(save_t), _ptr(wbwd), _ptr(eng.flat), nb, 0.0, _ptr(dpre), _ptr(gx), _ptr(gt), None,
                                      _ptr(grads), _ptr(slab), n, _stream(), _lib.BWD_ALL, 0, 2 * nb, eng._cfg()),
            "r2l_backward")
        out, off = [], 0
        for p in eng.params:
            k = p.numel()
            out.append(grads[off:off + k].view(p.shape))
            off += k
        return (None, None) + tuple(out)  # no gradient w.r.t. the embedded input (the reference never needs it)
