"""Per-block and head cost of the forward chain kernel (dev aid): time vs n_block at 131072 rays (4 full rounds)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model
from model.nerf_raybased import PointSampler
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
n = 131072
o = torch.randn(n, 3, device="cuda"); d = torch.randn(n, 3, device="cuda")
res = {}
for nb in (1, 11, 22, 43):
    m = build_model(O.make_state_dict(nb, seed=0), nb)
    with torch.no_grad():
        for _ in range(2): m.forward_rays(o, d, ps)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): m.forward_rays(o, d, ps)
        torch.cuda.synchronize(); res[nb] = (time.time() - t0) / 5
    print("n_block %2d: %.3f ms" % (nb, res[nb] * 1e3))
per_block = (res[43] - res[1]) / 42
head = res[1] - per_block
rounds = n / 32 / 1024
print("per block per round: %.2f us = %.0f cycles @2.38GHz (ideal 2048 MFMA x 64 = 131072)" % (per_block / rounds * 1e6, per_block / rounds * 2.38e9))
print("head+tail per round: %.2f us = %.0f cycles (ideal 4032 MFMA x 64 = 258048)" % (head / rounds * 1e6, head / rounds * 2.38e9))
