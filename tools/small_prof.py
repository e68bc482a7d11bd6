import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model
from model.nerf_raybased import PointSampler
from r2l_amd.train_step import R2LTrainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = build_model(O.make_state_dict(43, seed=0), 43)
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
tr = R2LTrainer(m, ps)
o = torch.randn(n, 3, device="cuda"); d = torch.randn(n, 3, device="cuda"); t = torch.rand(n, 3, device="cuda")
for _ in range(6): tr.step(o, d, t, 1e-4, perturb=1.)
torch.cuda.synchronize()
