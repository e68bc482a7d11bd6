"""Multi-GPU pre-flight.  The build's leases have ONE GPU, so nothing at N > 1 has touched RCCL / xGMI; these
tests skip there and fire, unattended, the first time a box with >= 2 GPUs runs the suite:
  * the C-ABI gradient exchange (include/r2l_hip.h r2l_allreduce_*: RCCL by dlopen, id made on rank 0 and handed over by the
    host) between two processes, one GPU each, bucket by bucket on its own stream — the worker also runs as ONE rank on every
    box, so its host logic (id hand-over through a file, streams, events) is exercised on the 1-GPU leases too;
  * `python bench.py --gpus 2`: the line carries the MEASURED RCCL rank count, the per-bucket all-reduce timeline and no
    shared-GPU label;
  * the 2-rank trainer on torch.distributed's nccl backend is tests/test_multirank_gpu.py, which picks nccl (one GPU per rank)
    by itself when the box has the devices.
Reference mechanism replaced: nn.DataParallel, /root/reference/main.py:37-42,472-479."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL refuses two ranks on one device)")

NATIVE_WORKER = r"""
import os, sys, time, torch
sys.path.insert(0, %(root)r)
from r2l_amd.dist_utils import NativeGradAllReducer, bucket_plan
rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
torch.cuda.set_device(rank)
if rank == 0:                                    # rank 0 makes the 128-byte id, the HOST hands it over (here: a file)
    uid = NativeGradAllReducer.make_unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 120, "no id from rank 0"
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
assert len(uid) == 128
red = NativeGradAllReducer(uid, world, rank)
n = 5917187                                      # the flat W256D88 gradient
g = (torch.arange(n, dtype=torch.float32, device="cuda") %% 1000) * (rank + 1)
want = (torch.arange(n, dtype=torch.float32, device="cuda") %% 1000) * sum(r + 1 for r in range(world))
for _, _, lo, hi in bucket_plan(43, 4):          # bucket by bucket, each on the reducer's stream behind an event
    red.submit(g[lo:hi])
assert red.pending() == 5
red.finish()
torch.cuda.synchronize()
assert torch.equal(g, want), (g[:4], want[:4])
flag = torch.tensor([1 if rank == world - 1 else 0], dtype=torch.int32, device="cuda")
import torch.distributed as dist
red.submit(flag, op=dist.ReduceOp.MAX)           # the segmented trainer's step-validity word
red.finish()
torch.cuda.synchronize()
assert int(flag.item()) == 1
assert abs(red.grad_scale() - 1.0 / world) < 1e-12
red.close()
print("native allreduce rank %%d of %%d ok" %% (rank, world))
"""


def _run_native(tmp_path, world):
    script = tmp_path / "native_worker.py"
    script.write_text(NATIVE_WORKER % {"root": ROOT})
    idfile = str(tmp_path / "rccl_id.bin")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), idfile], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d:\n%s" % (r, out[-3000:])
        assert "native allreduce rank %d of %d ok" % (r, world) in out


def test_native_allreduce_worker_one_rank(tmp_path):
    """The worker of the 2-GPU test below as a single rank: id through the file, buckets on the reducer's stream, MAX of a flag."""
    _run_native(tmp_path, 1)


@two_gpus
def test_native_allreduce_two_ranks(tmp_path):
    _run_native(tmp_path, 2)


@two_gpus
def test_bench_two_gpus_on_rccl():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env={k: v for k, v in os.environ.items() if not k.startswith("R2L_")}, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and "shared_gpu_test" not in out
    assert out["value"] > 0 and out["scaling"] == "weak"
    # top level = the graded exact-fp32 families; the default fp16 trio's legs live under fast_mode (bench.py docstring)
    assert out["dtype"].startswith("f32 (v_mfma_f32_32x32x2_f32") and out["roofline"]["peak"] == 157.3
    legs = {"train": out["train"], **{k: out["fast_mode"][k] for k in ("train", "train_strong", "train_4096", "train_12288")}}
    for leg, d in legs.items():
        tl = d["roofline"]["bucket_timeline_ms"]
        assert len(tl) >= 2 and all(b["submit"] >= 0 for b in tl), (leg, tl)
        assert d["roofline"]["grad_allreduce_alone_ms"] > 0
