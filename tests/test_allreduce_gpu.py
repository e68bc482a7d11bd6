"""The C-ABI gradient all-reduce (include/r2l_hip.h r2l_allreduce_*, RCCL bound by dlopen inside libr2l_hip.so) on the
one GPU a test box has: a 1-rank communicator is a real ncclCommInitRank / ncclAllReduce / ncclCommDestroy round trip
(SUM over one rank = identity), and two processes sharing the GPU is not something RCCL supports, so N > 1 is covered by
the gloo tests of the torch.distributed form (tests/test_driver_cpu.py) and by bench.py's measured rccl_ranks."""
import ctypes

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


def test_native_allreduce_reports_errors():
    from r2l_amd import _lib
    lib = _lib.load()
    rc = lib.r2l_allreduce_init(None, 2, 0, None)
    assert rc >= 10000 and b"bad arguments" in lib.r2l_last_error()
    assert lib.r2l_grad_allreduce(None, None, 4, None) >= 10000
    assert lib.r2l_allreduce_destroy(None) == 0
