"""Distillation training step of the R2L student on one GPU per process (reference loop: main.py:1175-1425).

    rgb = model(embed(sample_train(o, d, perturb)));  loss = mean((rgb - target)^2) * lw_rgb
    optimizer.zero_grad(); loss.backward(); optimizer.step()

becomes four HIP stages on flat fp32 buffers (include/r2l_hip.h): fused forward with activation stash ->
r2l_backward (dX chain, dW GEMMs, head/tail gradients) -> [one RCCL all-reduce of the flat 23.67 MB gradient when
world_size > 1] -> fused Adam -> re-pack of the MFMA weight streams.  No autograd graph, no anomaly mode.
"""
import collections
import ctypes
import math

import os
import time

import torch
import torch.distributed as dist

from . import _lib
from .dist_utils import GradAllReducer, bucket_plan, sync_parameters
from .engine import N_SAMPLE, W, _ptr, _stream, get_engine

STATUS_LAG = 4  # a segmented step's validity word is looked at exactly this many steps later (on every rank alike)


class R2LTrainer:
    """Owns gradient / Adam-moment / activation-stash buffers for one NeRF_v3_2 and runs fused training steps.

    Data parallel: every rank holds identical parameters, calls step() on its own ray shard (equal sizes), and the
    flat gradient is summed across ranks with ONE all-reduce (RCCL over xGMI on GPUs; gloo on CPU tensors in tests),
    then averaged inside the Adam kernel — identical to the reference's single global mean (main.py:1377) that
    nn.DataParallel computed on GPU 0.
    """

    def __init__(self, module, point_sampler, betas=(0.9, 0.999), eps=1e-8, lw_rgb=1.0, process_group=None, dw_mode=None,
                 chain_segments=None, engine=None):
        """dw_mode: None (keep the engine's config: default 'auto' = fp16 weight-gradient operands unless R2L_DW_EXACT=1),
        'fp16' or 'exact' (weight-gradient GEMMs of the fp16 trio on hi + mid operands: fp32-grade dW, ~2x the stash
        traffic; README.md "Training modes")."""
        self.module = module
        self.ps = point_sampler
        # (engine: tests of the HOST logic at world sizes no box offers hand in a stand-in, tests/test_driver_cpu.py; the product
        # always takes the module's HIP engine — every launch below goes through the four _launch hooks of this class)
        self.eng = engine if engine is not None else get_engine(module)
        if dw_mode is not None:
            self.eng.set_config(dw_mode=dw_mode)
        self.lib = self.eng.lib
        self.betas, self.eps, self.lw_rgb = betas, eps, lw_rgb
        self.reducer = GradAllReducer(process_group)
        # world > 1: gradient buckets in backward order, each all-reduced (async, RCCL's own stream) while the gradient
        # kernels of the next bucket run; R2L_AR_BUCKETS=0 selects one blocking all-reduce after the whole backward
        self.n_buckets = int(os.environ.get("R2L_AR_BUCKETS", "4"))
        self.force_staged = False  # tests: run the staged backward on one GPU (nothing is submitted at world == 1)
        # Small steps (the cooperative chains leave CUs idle): the dX chain itself cut into `chain_segments` block segments,
        # the weight gradients of a finished segment — and their all-reduce — on a second stream beside the next segment
        # (include/r2l_hip.h R2L_BWD_CHAIN with a layer range).  Such steps run WITHOUT the bf16x3 fallback kernels: a step
        # whose chain raised the range-guard word is skipped on the device (r2l_adam_step_guarded, on every rank: MAX over
        # ranks), read here STATUS_LAG steps later (no stall; the same step on every rank), and the trainer goes back to the uncut form for good.
        # Gradients are bit-identical to the staged form.  On one GPU it only adds launches (4096 rays: 0.78 ms one call, 0.85
        # in 3 segments, 0.96 staged in 4 buckets; profiles/r03_staged_backward.txt).  OPT-IN (argument, or
        # R2L_CHAIN_SEGMENTS=n): the default is 1 (off) at every world size until the form has been measured on more than one
        # GPU — it drops a batch where the reference never does, and its gain is a projection so far.
        self.chain_segments = int(os.environ.get("R2L_CHAIN_SEGMENTS", "1")) if chain_segments is None else int(chain_segments)
        # range control (include/r2l_hip.h): the first step is preceded by forward-only launches on its batch until the weight
        # stream's activation scale fits this model, so that no TRAINING step has to fall back (or, segmented, be skipped)
        # just because a checkpoint's activations sit above fp16's range.  calibrate=False: tests of the fallback itself.
        self.calibrate = True
        self._calibrated = False
        self.segments_disabled = False
        self.skipped_steps = 0
        self._side = None
        self._head_side = None  # staged backward of small steps: the head gradient beside the body buckets (_launch_head_beside)
        self._status_dev = None
        self._status_host = None  # pinned ring of validity words, one slot per segmented step still in flight
        self._status_pending = collections.deque()  # (event behind the copy, ring slot), oldest first
        self._status_slot = 0
        self._guard = None  # device word handed to the guarded Adam of the current step, or None
        if self.reducer.world() > 1 and self.n_buckets > 0 and self.eng.effective_config().reserve_cus == 0:
            # the weight-gradient kernels are persistent workgroups that fill every CU: leave a few to the RCCL kernels that
            # run beside them (r2l_config.reserve_cus of this trainer's calls; R2L_RESERVE_CUS overrides the count)
            self.eng.set_config(reserve_cus=int(os.environ.get("R2L_RESERVE_CUS", "8")))
        self.step_count = 0
        self.cap = 0
        self._fused_repack = False
        self.dw_slab = None
        self._alloc_state()

    # ---- buffers --------------------------------------------------------------------------------------------------
    def _alloc_state(self):
        eng = self.eng
        eng.ensure_packed()
        if sync_parameters(eng.flat, self.reducer.pg):  # replicas start from rank 0's weights (per-rank RNG)
            eng.mark_dirty()
            eng.ensure_packed()
        dev, n = eng.device, eng.n_param
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        # (zero-filled: its status words carry the gradient-scale history, include/r2l_hip.h)
        self.wstream_bwd = torch.zeros(self.lib.r2l_bwd_stream_floats(eng.n_block), dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(2, dtype=torch.float32, device=dev)
        self._bwd_packed = None

    def _ensure_capacity(self, n):
        if n <= self.cap:
            return
        dev, nb = self.eng.device, self.eng.n_block
        self.cap = n
        f = dict(dtype=torch.float32, device=dev)
        # floats per stash slot: N rounded up to a 32-ray tile x 1.5 KiB (the bf16x3 chains stash bf16 triples; the other
        # chains use the first Np*256 floats of their slots, row-major fp32)
        slot = int(self.lib.r2l_stash_slot_floats(n))
        self.save_x = torch.empty((nb + 1) * slot, **f)
        self.save_t = torch.empty(max(nb, 1) * slot, **f)
        self.gx = torch.empty((nb + 1) * slot, **f)
        self.gt = torch.empty(max(nb, 1) * slot, **f)
        self.dpre = torch.empty(n * 3, **f)
        self.sqerr = torch.empty(int(self.lib.r2l_num_tiles(n)), **f)
        if getattr(self, "dw_slab", None) is None and not os.environ.get("R2L_NO_DW_SLAB"):  # env: A/B diagnostics only
            self.dw_slab = torch.empty(int(self.lib.r2l_dw_slab_floats()), **f)

    def _pack_bwd(self, n):
        """Transposed stream of the layout the n-ray launches read, if the parameters changed since it was packed."""
        eng = self.eng
        eff = eng.effective_config()
        ver, layout = (eng.version(), eff.precision, eff.tiling), self.lib.r2l_backward_layout_for_cfg(int(n), eng._cfg())  # 16 / 32 / 3 / 2: what r2l_backward will read
        if self._bwd_packed is None:  # (layout 2 = the fp16x2 stream; its bf16x3 fallback stream packs itself when it runs)
            self._bwd_packed = {16: None, 32: None, 3: None, 2: None}
        if self._bwd_packed[layout] != ver:
            _lib.check(self.lib.r2l_pack_backward_layout(_ptr(eng.flat), eng.n_block, _ptr(self.wstream_bwd), layout,
                                                         _stream()), "r2l_pack_backward_layout")
            self._bwd_packed[layout] = ver

    # ---- the launches of a step (everything else in this class is host logic) --------------------------------------------------
    def _stream(self):
        return _stream()

    def _launch_backward(self, args, parts, layer_lo, layer_hi, what="r2l_backward_part"):
        """One call of the staged backward (include/r2l_hip.h r2l_backward_part_cfg) on the step's argument tuple."""
        _lib.check(self.lib.r2l_backward_part_cfg(*args, parts, layer_lo, layer_hi, self.eng._cfg()), what)

    def _launch_loss_finish(self, n):
        _lib.check(self.lib.r2l_loss_finish(_ptr(self.sqerr), int(self.lib.r2l_num_tiles(n)), self.lw_rgb / (3.0 * n),
                                            _ptr(self.loss_out), self._stream()), "r2l_loss_finish")

    def _launch_adam(self, lr):
        eng = self.eng
        if self._fused_repack:
            # the default trio: Adam + the re-pack of both fp16x2 streams in two launches (include/r2l_hip.h r2l_adam_step_packed)
            _lib.check(
                self.lib.r2l_adam_step_packed(_ptr(eng.flat), _ptr(self.grads), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                              eng.n_block, float(lr), self.betas[0], self.betas[1], self.eps, self.step_count,
                                              self.reducer.grad_scale(), _ptr(self._guard), _ptr(eng.wstream),
                                              _ptr(self.wstream_bwd), self._stream()), "r2l_adam_step_packed")
            return
        _lib.check(
            self.lib.r2l_adam_step_guarded(_ptr(eng.flat), _ptr(self.grads), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                           eng.n_param, float(lr), self.betas[0], self.betas[1], self.eps, self.step_count,
                                           self.reducer.grad_scale(), _ptr(self._guard), self._stream()), "r2l_adam_step")

    # ---- one optimisation step --------------------------------------------------------------------------------------
    def forward_backward(self, rays_o, rays_d, target, perturb=0., t_rand=None, zero_grad=True, n_global=None):
        """Forward + backward on this rank's rays; leaves d(loss)/d(params) in self.grads. Returns rgb [N,3].
        n_global: when the ranks' shares differ (--N_rand not divisible by the world size), the ray counts of ALL ranks in this
        step (a sequence, the same on every rank; or just their sum): the gradient is then scaled so that the all-reduced sum,
        divided by world in Adam, is the reference's single global mean (main.py:1377); None = equal shares
        (n_global = world * N).
        Segmented form (chain_segments > 1, small steps): no fallback kernels run, so self.grads is meaningful only if the
        step's validity word is 0 — step() hands the word to the guarded Adam; a caller that reads self.grads itself asks
        gradients_valid() first."""
        eng = self.eng
        n = rays_o.shape[0]
        eng.ensure_packed(n)
        self._pack_bwd(n)
        self._ensure_capacity(n)
        # opt-in (R2L_ADAM_PACK=1): steps of the default trio (both streams in the fp16x2 layout) get their re-pack from the optimizer
        # kernel (adam()).  Bit-identical and 23 us less kernel time per step (r2l_adam_pack_kernel 37 us against 60 for adam and
        # the two pack kernels), but the dX chain then finds its stream cold — packed a whole step earlier instead of right in
        # front of it: +15 us at 4096 rays, +33 us at 12 288 — so the step does not get faster: off by default
        # (profiles/r05_small_step_ab.txt)
        self._fused_repack = (bool(os.environ.get("R2L_ADAM_PACK")) and eng.layout_for(n, True) == 2 and
                              self.lib.r2l_backward_layout_for_cfg(int(n), eng._cfg()) == 2)
        rays_o = rays_o.contiguous().float()
        rays_d = rays_d.contiguous().float()
        target = target.contiguous().float()
        if perturb > 0 and t_rand is None:
            t_rand = torch.rand(n, N_SAMPLE, device=eng.device)
        if perturb <= 0:
            t_rand = None
        ztab = eng.ztab(self.ps.z_vals, perturb)
        if self.calibrate and not self._calibrated:
            self._calibrate(rays_o, rays_d, perturb, t_rand)
        rgb = eng.forward_rays(rays_o, rays_d, self.ps.z_vals, perturb, t_rand, save=(self.save_x, self.save_t))
        if zero_grad:
            self.grads.zero_()
        shares = None
        if n_global is not None and not isinstance(n_global, int):
            shares = [int(q) for q in n_global]
            n_global = sum(shares)
        grad_scale = 2.0 * self.lw_rgb / (3.0 * n) if n_global is None else 2.0 * self.lw_rgb * self.world() / (3.0 * n_global)
        args = (_ptr(rays_o), _ptr(rays_d), _ptr(t_rand), _ptr(ztab), None, _ptr(rgb), _ptr(target), None,
                _ptr(self.save_x), _ptr(self.save_t), _ptr(self.wstream_bwd), _ptr(eng.flat), eng.n_block, grad_scale,
                _ptr(self.dpre), _ptr(self.gx), _ptr(self.gt), _ptr(self.sqerr), _ptr(self.grads), _ptr(self.dw_slab), n,
                self._stream())
        self._step_inputs = (rays_o, rays_d, target, t_rand, grad_scale)  # (what the launch hooks of this step work on)
        self._guard = None
        self._check_skipped()
        if (self.chain_segments > 1 and zero_grad and not self.segments_disabled and
                self._segments_ok(n, n_global is not None, shares)):
            self._segmented_backward(args)
        elif (self.world() > 1 or self.force_staged) and self.n_buckets > 0 and zero_grad:
            # staged backward (include/r2l_hip.h r2l_backward_part): dX chain + tail, then the body buckets from the last
            # blocks to the first, the head last; every finished range of the flat gradient goes to the collective at once
            self._launch_backward(args, _lib.BWD_CHAIN | _lib.BWD_TAIL, 0, 0, "r2l_backward_part(chain, tail)")
            # small steps: the head's weight gradient BESIDE the body buckets, on a second stream (round 6: what the one-call
            # form does inside the library since round 5 — the head kernels are VALU-bound on 64 workgroups and would run
            # alone on a mostly idle chip behind the last bucket; they share only the chain's outputs with the body kernels and
            # write their own regions of the flat gradient and of dw_slab, r2l_dw.h).  Same kernels, same arguments: bit-identical.
            head_done = self._launch_head_beside(args, n)
            for lo, hi, flat_lo, flat_hi in bucket_plan(eng.n_block, self.n_buckets):
                if flat_lo == 0:
                    if head_done is None:
                        self._launch_backward(args, _lib.BWD_HEAD, 0, 0, "r2l_backward_part(head)")
                    else:
                        torch.cuda.current_stream().wait_event(head_done)
                elif hi > lo:  # (a net without body blocks has one empty body bucket: only its tail range to exchange)
                    self._launch_backward(args, _lib.BWD_BODY, lo, hi, "r2l_backward_part(%d, %d)" % (lo, hi))
                self.reducer.submit(self.grads[flat_lo:flat_hi])
        else:
            self._launch_backward(args, _lib.BWD_ALL, 0, 2 * eng.n_block, "r2l_backward")
        self._launch_loss_finish(n)
        return rgb

    HEAD_BESIDE_MAX_RAYS = 8192  # one tile per cooperative workgroup = idle CUs beside the body kernels (measured: 4096 rays 0.950 vs 0.967 ms, 12 288 rays 1.565 vs 1.550: profiles/r06_staged_backward.txt)

    def _launch_head_beside(self, args, n):
        """Staged backward, small steps: launch the head weight gradient on the side stream behind the dX chain (already enqueued on
        the current stream) and return the event that marks it done — or None: the caller launches it in line (large steps, where
        the body kernels fill the chip; R2L_NO_DW_OVERLAP=1; a stand-in engine without a device, tests/test_driver_cpu.py)."""
        dev = getattr(self.eng, "device", None)
        if (n > int(os.environ.get("R2L_HEAD_BESIDE_MAX_RAYS", self.HEAD_BESIDE_MAX_RAYS)) or os.environ.get("R2L_NO_DW_OVERLAP", "")[:1] not in ("", "0") or self.dw_slab is None
                or not isinstance(dev, torch.device) or dev.type != "cuda"):
            return None
        if self._head_side is None:
            self._head_side = torch.cuda.Stream(device=dev)
        main, side = torch.cuda.current_stream(), self._head_side
        fork = torch.cuda.Event()
        fork.record(main)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            self._launch_backward(args[:-1] + (ctypes.c_void_p(side.cuda_stream),), _lib.BWD_HEAD, 0, 0, "r2l_backward_part(head)")
            done = torch.cuda.Event()
            done.record(side)
        return done

    # ---- range control ------------------------------------------------------------------------------------------------------
    def _calibrate(self, rays_o, rays_d, perturb, t_rand):
        """Once per trainer (a handful of forward-only launches and host reads before the first step): a launch that leaves
        fp16's range is redone by the bf16x3 kernel and re-scales the stream (library side, include/r2l_hip.h); repeat until
        a launch stays on the fp16 kernels.  Nets within range (s = 1: default init, every net measured so far): one launch."""
        self._calibrated = True
        eng = self.eng
        if eng.layout_for(rays_o.shape[0], True) != 2:  # not the fp16 trio: nothing to scale
            return
        trips = eng.range_info()["trips"]
        for _ in range(6):
            eng.forward_rays(rays_o, rays_d, self.ps.z_vals, perturb, t_rand)
            now = eng.range_info()
            if now["trips"] == trips:
                break
            trips = now["trips"]

    def range_info(self):
        """Telemetry of the fp16 kernels' range control (synchronises; the driver logs it every i_print):
        forward (engine.range_info) + 'grad_amax' / 'grad_scale' / 'grad_headroom' / 'bwd_trips' of the backward chain."""
        info = dict(self.eng.range_info())
        word = ctypes.cast(self.lib.r2l_backward_status_words(_ptr(self.wstream_bwd), self.eng.n_block), ctypes.c_void_p).value
        off = (word - self.wstream_bwd.data_ptr()) // 4
        w = self.wstream_bwd[off:off + 16].view(torch.int32).cpu()
        f = w.view(torch.float32)
        if int(w[9]) == _lib.RANGE_MAGIC:
            gscale, scaled = float(f[4]), float(f[8])
            info.update(grad_scale=gscale, grad_amax=(scaled / gscale if gscale > 0 else 0.0),
                        grad_headroom=(32768.0 / scaled if scaled > 0 else float("inf")), bwd_trips=int(w[10]) + int(w[0] != 0))
        return info

    # ---- segmented dX chain (small steps) -----------------------------------------------------------------------------------
    def _segments_ok(self, n, uneven, shares):
        """Does this step take the segmented form?  The answer must be the same on every rank (the two forms submit different
        collectives), so with uneven shares it is asked for every share size that occurs — and is "no" when only their sum
        is known."""
        ok, nb, cfg = self.lib.r2l_chain_segments_ok_cfg, self.eng.n_block, self.eng._cfg()
        if not uneven or self.world() == 1:
            return bool(ok(int(n), nb, cfg))
        return shares is not None and all(q > 0 and bool(ok(q, nb, cfg)) for q in set(shares))

    def _segmented_backward(self, args):
        eng, part, cfg, NF = self.eng, self.lib.r2l_backward_part_cfg, self.eng._cfg(), _lib.BWD_NOFALLBACK
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=eng.device)
            self._status_dev = torch.zeros(1, dtype=torch.int32, device=eng.device)
            self._status_host = torch.zeros(STATUS_LAG, dtype=torch.int32).pin_memory()
            word = ctypes.cast(self.lib.r2l_backward_status_word(_ptr(self.wstream_bwd), eng.n_block), ctypes.c_void_p).value
            off = (word - self.wstream_bwd.data_ptr()) // 4
            self._status_src = self.wstream_bwd[off:off + 1].view(torch.int32)  # the chain's range-guard word
        side = self._side
        side_args = args[:-1] + (ctypes.c_void_p(side.cuda_stream),)
        plan = bucket_plan(eng.n_block, self.chain_segments)
        first = True
        for lo, hi, flat_lo, flat_hi in plan:
            if flat_lo == 0 or hi <= lo:
                continue
            what = _lib.BWD_CHAIN | NF | (_lib.BWD_TAIL if first else 0)
            _lib.check(part(*args, what, lo, hi, cfg), "r2l_backward_part(chain segment %d..%d)" % (lo, hi))
            first = False
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                _lib.check(part(*side_args, _lib.BWD_BODY | NF, lo, hi, cfg), "r2l_backward_part(body %d..%d)" % (lo, hi))
                self.reducer.submit(self.grads[flat_lo:flat_hi])
        with torch.cuda.stream(side):  # the head needs gx[0] of the last segment (already waited for) and shares dw_slab with the body
            _lib.check(part(*side_args, _lib.BWD_HEAD | NF, 0, 0, cfg), "r2l_backward_part(head)")
            self.reducer.submit(self.grads[0:bucket_plan(eng.n_block, 1)[-1][3]])
            self._status_dev.copy_(self._status_src)  # valid step? (0) — the same answer on every rank: MAX
            self.reducer.submit(self._status_dev, op=dist.ReduceOp.MAX)
            done = torch.cuda.Event()
            done.record(side)
        main.wait_event(done)
        self._guard = self._status_dev

    def _check_skipped(self, wait=False):
        """Did an earlier segmented step turn out invalid (its update was skipped on the device)?  Every such step copies its
        validity word into its own slot of a pinned ring behind its Adam, and the word of step i is read at the start of step
        i + STATUS_LAG — by then it has long landed, so the host, which runs a few steps ahead of the device, does not stall —
        or in drain().  A fixed lag rather than "whenever the copy has landed": the word is the same on every rank (MAX
        all-reduce), and reading it at the same step makes every rank leave the segmented form at the same step, which the
        collectives need (the two forms submit different bucket sequences)."""
        pend = self._status_pending
        while pend and (wait or len(pend) >= STATUS_LAG):
            ev, slot = pend.popleft()
            ev.synchronize()
            if int(self._status_host[slot]) != 0:
                self.skipped_steps += 1
                if not self.segments_disabled:
                    self.segments_disabled = True
                    import logging
                    logging.getLogger("r2l_amd").warning(
                        "a segmented training step needed the bf16x3 fallback (fp16 range guard): its update was skipped on "
                        "every rank; continuing with the uncut backward (chain_segments off)")

    def gradients_valid(self):
        """After forward_backward: False iff this was a segmented step whose chain raised the fp16 range guard (self.grads is
        then not a gradient; the uncut form — chain_segments = 1 — recomputes such steps on the bf16x3 kernels).  Waits for
        the device."""
        return self._guard is None or int(self._guard.item()) == 0

    def drain(self):
        """Waits for the validity words of all segmented steps in flight; returns the number of skipped steps so far."""
        self._check_skipped(wait=True)
        return self.skipped_steps

    def world(self):
        return self.reducer.world()

    def allreduce_grads(self):
        """The step's one exchange: sum of the flat fp32 gradient over ranks.  When forward_backward
        already handed the buckets to the collective this only makes the stream wait for them."""
        if self.reducer.pending():
            self.reducer.finish()
        else:
            self.reducer.allreduce(self.grads)

    def adam(self, lr):
        self.step_count += 1
        eng = self.eng
        fused = self._fused_repack
        self._launch_adam(lr)
        if self._guard is not None:  # segmented step: its validity word goes to the host behind the update (checked next step)
            slot = self._status_slot
            self._status_slot = (slot + 1) % STATUS_LAG
            self._status_host[slot:slot + 1].copy_(self._guard, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._status_pending.append((ev, slot))
            self._guard = None
        eng.mark_dirty()
        if fused and eng._packed_version is not None and self._bwd_packed is not None:
            # the fp16x2 parts of both streams already hold the new weights: nothing to pack before the next launch of this layout
            eff = eng.effective_config()
            eng._packed_version[2] = eng.version()
            self._bwd_packed[2] = (eng.version(), eff.precision, eff.tiling)

    def step(self, rays_o, rays_d, target, lr, perturb=0., t_rand=None, n_global=None):
        """zero_grad + forward + backward + all-reduce + Adam.  Returns (rgb[N,3], loss_out[2] = [loss, psnr]) on
        the device (no host sync)."""
        rgb = self.forward_backward(rays_o, rays_d, target, perturb, t_rand, n_global=n_global)
        self.allreduce_grads()
        self.adam(lr)
        return rgb, self.loss_out

    # ---- torch.optim.Adam-compatible state (checkpoint surface: 'optimizer_state_dict', main.py:1528-1529) --------------
    def optimizer_state_dict(self, lr):
        state, off = {}, 0
        for i, p in enumerate(self.eng.params):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
            off += n
        group = {"lr": lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(self.eng.params)))}
        return {"state": state if self.step_count > 0 else {}, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        off = 0
        steps = [0]
        for i, p in enumerate(self.eng.params):
            n = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.append(int(float(st["step"])))
            off += n
        self.step_count = max(steps)


def lr_schedule(step, lrate, lrate_decay, warmup_lr=""):
    """Learning rate of iteration `step` (1-based): linear warm-up 'start_lr,end_iter' then
    lrate * 0.1^((step-end_iter)/(lrate_decay*1000))  — the reference's schedule, main.py:1181-1193."""
    decay_steps = lrate_decay * 1000
    if warmup_lr:
        start_lr, end_iter = [float(v) for v in warmup_lr.split(",")]
        if step < end_iter:
            return (lrate - start_lr) / end_iter * step + start_lr
        return lrate * (0.1**((step - end_iter) / decay_steps))
    return lrate * (0.1**(step / decay_steps))


# ---------------------------------------------------------------------------------------------------------------
# hook used by bench.py
# ---------------------------------------------------------------------------------------------------------------
def bench(net, ps, a, world, rank, distributed, device, timed, flop_per_ray, peak, n_rays=None, dw_mode=None, chain_segments=None,
          precision="fp16x2"):
    """Training leg of bench.py: K fused steps of `a.train_rays` rays per GPU (synthetic [o,d,rgb] rows as in the
    `.npy` shards, main.py:1305-1311), RCCL all-reduce when world > 1.  The kernel family is selected through the engine's
    r2l_config (precision, dw_mode) and everything printed about it — matrix path, peak — is derived from the layout / tiling the
    library reports for THAT config (r2l_*_for_cfg), never from the environment."""
    n = a.train_rays if n_rays is None else n_rays
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).to(device)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(device)
    tgt = torch.rand(n, 3, generator=g).to(device)
    get_engine(net).set_config(precision=precision, dw_mode=dw_mode or "auto")
    tr = R2LTrainer(net, ps, chain_segments=chain_segments)
    steps, warm = max(1, a.steps), max(1, a.warmup)  # the full K / W of the command line, every leg

    def train_step(i):
        tr.step(o, d, tgt, lr_schedule(i + 1, 5e-4, 500, "0.0001,200"), perturb=1.0)

    dt, step_ms = timed(train_step, steps, warm, distributed, device)
    achieved = n * flop_per_ray / (step_ms * 1e-3) / 1e12
    # Which matrix path the step's GEMM kernels take, and the matrix-pipe bound that goes with it in ALGORITHMIC FLOP/s: a
    # kernel that evaluates an fp32 product as p 16-bit MFMA products can at best reach (dense 16-bit MFMA peak) / p.
    #   fp16 trio (backward layout 2): forward 11.79 MFLOP/ray x 3 fp16 products, dX chain 11.27 x 3, dW body 11.27 x 1 (fp16 hi
    #   operands; x 3 with exact weight gradients), head dW 0.52
    #   bf16x3 trio (layout 3): 6 products everywhere; layouts 32 / 16: fp32-MFMA chains (their dW on the fp32 MFMA or bf16x3)
    cfg = tr.eng._cfg()
    layout = int(tr.lib.r2l_backward_layout_for_cfg(int(n), cfg))
    exact = tr.eng.effective_config().dw_mode == _lib.DW_MODE["exact"]
    peak_fp32 = peak
    path = "fp32 MFMA chains (layout %d)" % layout
    extra = {"r2l_config": {"precision": precision, "dw_mode": dw_mode or "auto", "backward_layout": layout}}
    fwd_f, dx_f, dw_f, head_f = 11789824. - 516096., 2 * 86 * 256 * 256., 2 * 86 * 256 * 256., 2 * 1008 * 256.
    if layout == 2 and exact:
        peak = 2500.0 / 3.
        path = ("fp16 trio with EXACT weight gradients: forward, dX chain, dW body and head dW with 3 fp16 MFMA products per fp32 "
                "product (hi + mid operands everywhere); range-controlled, bf16x3 trio behind it")
    elif layout == 2:
        mfma_flops = 3 * (fwd_f + head_f) + 3 * dx_f + 1 * dw_f + (2500.0 / peak_fp32) * head_f
        peak = 2500.0 * flop_per_ray / mfma_flops
        path = ("fp16 trio: forward and dX chain with 3 fp16 MFMA products per fp32 product (two-way operand splits), dW body "
                "with 1 (fp16 hi operands from the fp16 stash); range-controlled, bf16x3 trio behind it")
        extra.update({"peak_if_every_gemm_took_3_products": 2500.0 / 3., "frac_of_3_product_peak": achieved / (2500.0 / 3.)})
    elif layout == 3:
        peak = 2500.0 / 6.
        path = "bf16x3 trio: forward, dX chain and dW body with 6 bf16 products per fp32 product (fp32-exact products)"
    if layout == 2:
        nt = int(tr.lib.r2l_coop_tiles_for_cfg(int(n), tr.eng.n_block, cfg))
        if nt:
            # cooperative chains (r2l_coopf: one or two 32-ray tiles per workgroup), each workgroup streaming the 25 MB of
            # packed weights per chain from L2 — measured ~45 B/clk per CU, which is what bounds the small launches
            tiles = (n + 31) // 32
            if nt == 3:  # mixed grid (csrc/r2l_coopf.h r2l_coopf_policy): one workgroup per CU
                n_cu = torch.cuda.get_device_properties(device).multi_processor_count
                wgs = n_cu
                path += "; chains: cooperative kernels, MIXED grid (%d two-tile + %d one-tile workgroups = one per CU), L2 weight " \
                        "stream: %d workgroups x 25.1 MB per chain" % (tiles - n_cu, 2 * n_cu - tiles, wgs)
            else:
                wgs = (tiles + nt - 1) // nt
                path += "; chains: cooperative kernels (%d tile(s) per workgroup), L2 weight stream: %d workgroups x 25.1 MB per " \
                        "chain" % (nt, wgs)
            extra["weight_stream_bytes_per_step"] = 2 * wgs * 25.1e6
        extra["range"] = tr.range_info()
    if distributed:
        # the exchange alone (the step's whole flat gradient in one all-reduce, nothing to hide behind): what the bucketed
        # overlap has to cover, for reading the scaling numbers
        buf = torch.zeros_like(tr.grads)
        for _ in range(2):
            tr.reducer.allreduce(buf)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(5):
            tr.reducer.allreduce(buf)
        torch.cuda.synchronize(device)
        extra["grad_allreduce_alone_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        extra["grad_allreduce_bytes"] = buf.numel() * 4
        extra["allreduce_buckets"] = tr.n_buckets
        extra["chain_segments"] = tr.chain_segments if tr.lib.r2l_chain_segments_ok_cfg(int(n), tr.eng.n_block, tr.eng._cfg()) else 1
        # one more step with device timestamps per bucket: when its all-reduce was handed over and when the compute stream
        # got past the wait for it (exposed = what lies between the last gradient kernel and Adam), relative to the step start
        tr.reducer.enable_trace()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        train_step(steps + warm)
        e1.record()
        torch.cuda.synchronize(device)
        extra["bucket_timeline_ms"] = [{"floats": nf, "submit": round(ts, 4), "wait_passed": None if tw is None else round(tw, 4)}
                                       for nf, ts, tw in tr.reducer.trace_ms(e0)]
        extra["bucket_timeline_step_ms"] = e0.elapsed_time(e1)
        tr.reducer.trace = None
    return {"value": n * steps * world / dt, "unit": "rays/s", "steps": steps, "warmup": warm,
            "ms_per_step": dt / steps * 1e3, "rays_per_step_per_gpu": n,
            "workload": "distillation step (fwd + bwd + Adam + weight re-pack), %d rays/GPU/step, perturb=1; "
                        "grad all-reduce over %d rank(s)" % (n, world),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "peak_fp32_mfma": peak_fp32,
                         "frac_of_fp32_mfma_peak": achieved / peak_fp32,
                         "matrix_path": path,
                         "flop_per_ray": flop_per_ray, "step_ms_device": step_ms, **extra},
            "final_loss": tr.loss_out[0].item()}
