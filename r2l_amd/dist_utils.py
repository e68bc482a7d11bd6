"""Data-parallel plumbing of R2L training: one process per GPU, identical parameters on every rank, and ONE logical
exchange per step — the SUM of the flat fp32 gradient buffer (5 917 187 floats = 23.67 MB for W256 D88) over RCCL/xGMI
(`nccl` backend on ROCm; gloo on CPU tensors in tests).  The division by world_size is folded into the Adam kernel
(grad_scale).  Replaces nn.DataParallel's per-step parameter broadcast + input scatter + output gather +
ReduceAddCoalesced (reference main.py:37-42, 472-479, 1374, 1404).

The exchange is cut into contiguous BUCKETS of the flat buffer that are handed to the collective in the order the
backward finishes them (tail + the last body layers first, the head last: `bucket_plan`), each as an asynchronous
all-reduce: on RCCL the collective's stream waits for the kernels already enqueued on the compute stream and runs
beside the gradient kernels of the next bucket; `finish()` makes the compute stream (not the host) wait before Adam.
"""
import torch
import torch.distributed as dist

W = 256
IN_DIM = 1008
LAYER_FLOATS = W * W + W          # one body layer: weight [256,256] + bias [256]
HEAD_FLOATS = IN_DIM * W + W      # head.0.weight + head.0.bias, at the start of the flat buffer
TAIL_FLOATS = 3 * W + 3           # tail.0.weight + tail.0.bias, at its end


def bucket_plan(n_block, n_buckets):
    """Gradient buckets in backward order.  Returns a list of (layer_lo, layer_hi, flat_lo, flat_hi): the body layers
    [layer_lo, layer_hi) of the 2*n_block (layer 2b = body.b.body.0, 2b+1 = body.b.body.2) whose gradients
    r2l_backward_part(R2L_BWD_BODY, layer_lo, layer_hi) completes, and the range of the flat buffer to all-reduce then.
    The first bucket's range also covers the tail (computed before the body), and a final (0, 0, 0, HEAD_FLOATS) entry
    is the head.  Buckets hold whole blocks (layer pairs) so that a block's two GEMMs share a launch."""
    n_buckets = max(1, min(int(n_buckets), max(n_block, 1)))
    edges = [round(i * n_block / n_buckets) for i in range(n_buckets + 1)]
    plan = []
    end = HEAD_FLOATS + 2 * n_block * LAYER_FLOATS + TAIL_FLOATS
    for i in reversed(range(n_buckets)):
        lo, hi = 2 * edges[i], 2 * edges[i + 1]
        flat_lo = HEAD_FLOATS + lo * LAYER_FLOATS
        flat_hi = end if i == n_buckets - 1 else HEAD_FLOATS + hi * LAYER_FLOATS
        plan.append((lo, hi, flat_lo, flat_hi))
    plan.append((0, 0, 0, HEAD_FLOATS))
    return plan


def split_shards(n_rand, world):
    """--N_rand shard files per step (GLOBAL, as in the reference: one DataLoader batch that nn.DataParallel scatters,
    main.py:794-806,1374) over `world` ranks: the first N_rand % world ranks take one more.  Every rank needs at least one."""
    if n_rand < world:
        raise ValueError("--N_rand %d (shard files per step, global) is smaller than the %d ranks" % (n_rand, world))
    base, rem = divmod(int(n_rand), int(world))
    return [base + (1 if r < rem else 0) for r in range(world)]


class GradAllReducer:
    def __init__(self, process_group=None, bucket_floats=0):
        self.pg = process_group
        self.bucket = bucket_floats  # blocking form: 0 = one call on the whole buffer
        self._works = []
        self.trace = None  # enable_trace(): [(floats, submit event, wait-passed event)] of the buckets of the last step

    def world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.pg)
        return 1

    def grad_scale(self):
        return 1.0 / self.world()

    # ---- blocking form ----------------------------------------------------------------------------------------------
    def allreduce(self, flat, async_op=False):
        """In-place sum over ranks; returns the list of work handles when async_op."""
        if self.world() == 1:
            return []
        handles = []
        n = flat.numel()
        step = self.bucket if self.bucket and self.bucket < n else n
        for off in range(0, n, step):
            h = dist.all_reduce(flat[off:off + step], op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
            if async_op:
                handles.append(h)
        return handles

    # ---- overlapped form: submit finished buckets as the backward produces them ------------------------------------------
    def submit(self, bucket, op=None):
        """Start the all-reduce of a finished, contiguous range of the flat gradient (a view).  The kernels that wrote
        it must already be enqueued on the current stream.  (op: SUM; MAX for the step-validity word of segmented steps.)"""
        if self.world() > 1:
            if self.trace is not None and bucket.is_cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()  # on the submitting stream: the bucket's gradient kernels are enqueued in front of it
                self.trace.append([bucket.numel(), ev, None])
            self._works.append(dist.all_reduce(bucket, op=op or dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def enable_trace(self):
        """bench.py at N > 1: device timestamps of every bucket of the next step — when its all-reduce was handed over
        (submit, on the stream that produced the bucket) and when the compute stream got past the wait for it (finish)."""
        self.trace = []

    def trace_ms(self, t0):
        """[(floats, submit ms, wait-passed ms)] relative to the event t0 (call after a synchronize)."""
        return [(n, t0.elapsed_time(a), t0.elapsed_time(b) if b is not None else None) for n, a, b in (self.trace or [])]

    def pending(self):
        return len(self._works)

    def finish(self):
        """Everything submitted is summed once this returns (GPU: once the current stream gets there; no host block)."""
        for i, w in enumerate(self._works):
            w.wait()
            if self.trace is not None and i < len(self.trace) and torch.cuda.is_available():
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()  # the compute stream is past this bucket's exchange
                self.trace[len(self.trace) - len(self._works) + i][2] = ev
        self._works = []


def sync_parameters(flat, process_group=None, src=0):
    """Every rank continues with rank `src`'s parameters (flat buffer, in place).  nn.DataParallel re-broadcast the one
    module of GPU 0 every step (reference main.py:472-479); with one process per GPU the replicas are made identical
    once, at trainer construction, and stay identical because every rank applies the same Adam update to the same
    summed gradient."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(flat, src=src, group=process_group)
        return True
    return False


def parameters_in_sync(flat, process_group=None):
    """True on every rank iff all ranks hold bit-identical `flat` (a debugging / test aid: two all-reduces)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return True
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=process_group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=process_group)
    return bool(torch.equal(lo, hi))


class NativeGradAllReducer:
    """The C-ABI form of the exchange (include/r2l_hip.h r2l_allreduce_*: RCCL dlopen'ed by libr2l_hip.so, no
    torch.distributed) — what a non-PyTorch host binds.  Same submit/finish surface as GradAllReducer, and the same overlap:
    a bucket's all-reduce goes to the reducer's OWN stream behind an event recorded on the stream that produced the bucket
    (hipEventRecord / hipStreamWaitEvent in a C host), so it runs beside the gradient kernels enqueued next; `finish` makes
    the current stream wait for the events recorded behind every collective (no host block).  overlap=False: everything on
    the current stream (the serial form)."""

    def __init__(self, unique_id, world, rank, overlap=True):
        import ctypes
        from . import _lib
        self._lib, self._ctypes = _lib, ctypes
        self.lib = _lib.load()
        self._world, self.rank = int(world), int(rank)
        self._comm_stream = torch.cuda.Stream() if overlap else None
        self._done = []
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        _lib.check(self.lib.r2l_allreduce_init(ctypes.cast(buf, ctypes.c_void_p), self._world, self.rank,
                                               ctypes.cast(ctypes.byref(self._h), ctypes.c_void_p)), "r2l_allreduce_init")

    @staticmethod
    def make_unique_id():
        """128 opaque bytes, made on rank 0; the host distributes them to the other ranks."""
        import ctypes
        from . import _lib
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().r2l_allreduce_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "r2l_allreduce_unique_id")
        return buf.raw

    def world(self):
        return self._world

    def grad_scale(self):
        return 1.0 / self._world

    def submit(self, bucket, op=None):
        ct = self._ctypes
        if op is not None and op not in (torch.distributed.ReduceOp.SUM, torch.distributed.ReduceOp.MAX):
            raise NotImplementedError("the C-ABI exchange is a SUM of floats (include/r2l_hip.h r2l_grad_allreduce)")
        is_max = op is not None and op == torch.distributed.ReduceOp.MAX
        if not is_max and bucket.dtype != torch.float32:
            raise NotImplementedError("r2l_grad_allreduce sums fp32 buffers")
        stream = torch.cuda.current_stream()
        if self._comm_stream is not None:
            ready = torch.cuda.Event()
            ready.record(stream)                    # the bucket's gradient kernels are enqueued in front of this
            self._comm_stream.wait_event(ready)
            stream = self._comm_stream
        with torch.cuda.stream(stream):
            # MAX of a flag word (the segmented trainer's step-validity word, train_step.py): "any rank raised it" is the SUM
            # of the flags being > 0 — the exchange stays a float SUM
            buf = (bucket != 0).to(torch.float32) if is_max else bucket
            self._lib.check(self.lib.r2l_grad_allreduce(self._h, ct.c_void_p(buf.data_ptr()), buf.numel(),
                                                        ct.c_void_p(stream.cuda_stream)), "r2l_grad_allreduce")
            if is_max:
                bucket.copy_((buf > 0).to(bucket.dtype))
                buf.record_stream(stream)
        if self._comm_stream is not None:
            done = torch.cuda.Event()
            done.record(self._comm_stream)
            self._done.append(done)

    def allreduce(self, flat):
        self.submit(flat)
        self.finish()

    def pending(self):
        return len(self._done)

    def finish(self):
        for ev in self._done:
            torch.cuda.current_stream().wait_event(ev)
        self._done = []

    def close(self):
        if self._h:
            self._lib.check(self.lib.r2l_allreduce_destroy(self._h), "r2l_allreduce_destroy")
            self._h = self._ctypes.c_void_p()
