"""Device time of the per-iteration host-side tensor work around the fused step (hard-ray pool) at the README sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r2l_amd.driver import HardRayPool

dev = torch.device("cuda")
B = 81920
pool = HardRayPool(0.2, 20)
g = torch.Generator(device=dev).manual_seed(0)
def batch():
    return (torch.randn(B, 3, device=dev), torch.randn(B, 3, device=dev), torch.rand(B, 3, device=dev))
o, d, t = batch()
while not pool.full:
    rgb = torch.rand(B, 3, device=dev)
    pool.update(rgb, o, d, t, B)
print("pool rows", pool.pool.shape[0])
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
state = {}
def aug():
    state["b"] = pool.augment(o, d, t)
def upd():
    oo, dd, tt = state["b"]
    pool.update(torch.rand(oo.shape[0], 3, device=dev), oo, dd, tt, B)
print("augment %.3f ms" % timeit(aug))
print("update  %.3f ms" % timeit(upd))
print("randperm(%d) %.3f ms" % (pool.pool.shape[0], timeit(lambda: torch.randperm(pool.pool.shape[0], device=dev))))
err = torch.rand(B, device=dev)
print("sort(%d) %.3f ms   topk %.3f ms" % (B, timeit(lambda: torch.sort(err)), timeit(lambda: torch.topk(err, 16384, sorted=False))))
