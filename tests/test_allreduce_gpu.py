"""The C-ABI gradient all-reduce (include/r2l_hip.h r2l_allreduce_*, RCCL bound by dlopen inside libr2l_hip.so) on the
one GPU a test box has: a 1-rank communicator is a real ncclCommInitRank / ncclAllReduce / ncclCommDestroy round trip
(SUM over one rank = identity), and two processes sharing the GPU is not something RCCL supports, so N > 1 is covered by
the gloo tests of the torch.distributed form (tests/test_driver_cpu.py) and by bench.py's measured rccl_ranks."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_native_allreduce_single_rank_round_trip():
    from r2l_amd.dist_utils import NativeGradAllReducer, bucket_plan
    uid = NativeGradAllReducer.make_unique_id()
    assert len(uid) == 128 and any(uid)
    red = NativeGradAllReducer(uid, 1, 0)
    assert red.world() == 1 and red.grad_scale() == 1.0
    g = torch.randn(5917187, device="cuda")
    ref = g.clone()
    for _, _, lo, hi in bucket_plan(43, 4):  # bucket by bucket, on the current stream, like the staged backward
        red.submit(g[lo:hi])
    red.finish()
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        red.allreduce(g)
    s.synchronize()
    assert torch.equal(g, ref)
    red.close()


def test_native_allreduce_overlapped_with_the_staged_backward():
    """The C-ABI exchange in its OVERLAPPED form (own stream, event behind the producer, event behind the collective) driven by
    the real staged backward of the trainer: every bucket's all-reduce runs on the reducer's stream beside the weight-gradient
    kernels of the next bucket; gradients and the Adam update equal the serial one-call step bit for bit (one rank: SUM =
    identity, so any ordering mistake between the streams shows as a torn or stale bucket)."""
    from model.nerf_raybased import PointSampler
    from oracle import r2l_oracle as O
    from r2l_amd.dist_utils import NativeGradAllReducer
    from r2l_amd.train_step import R2LTrainer
    from tests.test_forward_gpu import build_model
    sd = O.make_state_dict(n_block=43, seed=3)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    n = 20000
    g = torch.Generator().manual_seed(2)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    tgt = torch.rand(n, 3, generator=g).cuda()

    def run(native):
        tr = R2LTrainer(build_model(sd, 43), ps)
        tr.force_staged, tr.n_buckets = True, 4
        if native:
            tr.reducer = NativeGradAllReducer(NativeGradAllReducer.make_unique_id(), 1, 0)
        for i in range(3):
            tr.forward_backward(o, d, tgt)
            if native:
                assert tr.reducer.pending() == 5  # 4 body buckets + the head, in flight on the reducer's stream
            tr.allreduce_grads()
            tr.adam(1e-4)
        torch.cuda.synchronize()
        out = tr.grads.clone(), tr.eng.flat.clone()
        if native:
            tr.reducer.close()
        return out

    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_native_allreduce_reports_errors():
    from r2l_amd import _lib
    lib = _lib.load()
    rc = lib.r2l_allreduce_init(None, 2, 0, None)
    assert rc >= 10000 and b"bad arguments" in lib.r2l_last_error()
    assert lib.r2l_grad_allreduce(None, None, 4, None) >= 10000
    assert lib.r2l_allreduce_destroy(None) == 0


TORCH_NCCL_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29631")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)       # the call bench.py and the driver make (nccl = RCCL on ROCm)
from r2l_amd.dist_utils import bucket_plan
flat = torch.arange(5917187, dtype=torch.float32, device=dev) * 1e-3
ref = flat.clone()
works = []
for lo, hi, a, b in bucket_plan(43, 4):               # the trainer's buckets: async all-reduces of views of one buffer
    works.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, async_op=True))
for w in works:
    w.wait()                                          # (stream wait, not a host block)
flat.mul_(1.0)
torch.cuda.synchronize()
assert torch.equal(flat, ref)                         # SUM over one rank
t = torch.tensor([3.5], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.broadcast(flat, src=0)
dist.barrier()
assert t.item() == 3.5 and torch.equal(flat, ref)
dist.destroy_process_group()
print("torch nccl single rank ok")
"""


def test_torch_nccl_backend_single_rank(tmp_path):
    """torch.distributed's nccl (= RCCL) backend as bench.py / the driver use it at world > 1 — process group with device_id,
    asynchronous all-reduces of the trainer's gradient buckets (views of one flat buffer), MAX reduce, broadcast, barrier —
    brought up on the one GPU of this box with a communicator of one rank (two ranks need two devices)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "nccl1.py"
    script.write_text(TORCH_NCCL_WORKER % {"root": root})
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "torch nccl single rank ok" in r.stdout
