"""The one collective of data-parallel R2L training: a SUM all-reduce of the flat fp32 gradient buffer
(5 917 187 floats = 23.67 MB for W256 D88) over RCCL/xGMI (`nccl` backend on ROCm) — gloo on CPU tensors in tests.
The division by world_size is folded into the Adam kernel (grad_scale).  Replaces nn.DataParallel's per-step
parameter broadcast + input scatter + output gather + ReduceAddCoalesced (reference main.py:37-42, 472-479)."""
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, process_group=None, bucket_floats=0):
        self.pg = process_group
        self.bucket = bucket_floats  # 0: one call on the whole buffer

    def world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.pg)
        return 1

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

    def grad_scale(self):
        return 1.0 / self.world()
