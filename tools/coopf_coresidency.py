"""Reproducer of the co-residency fault of the cooperative training forward (csrc/r2l_coopf.h): one ray tile per
workgroup on launches of more tiles than CUs, compared bit for bit with the two-tile kernel (143 KiB of LDS: never two on a
CU).  The shipped library keeps one-tile workgroups on separate CUs (FC_SOLO_LDS_BYTES, checked at launch), so the fault only
shows in reproducer builds of the forward:
  tools/build_variant.sh share_r2  r2l_coopf_fwd.hip -DFC_ALLOW_SHARE_CU -DFC_TAIL_ASM=0   # round 2's tail, two per CU
  tools/build_variant.sh share_fix r2l_coopf_fwd.hip -DFC_ALLOW_SHARE_CU                   # the shipped tail, two per CU
  R2L_LIB_PATH=tools/_bin/share_r2/libr2l_hip.so python tools/coopf_coresidency.py [N=16384]
Prints, per mode, how many of 10 launches differ and where.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r2l_oracle as O
from tests.test_forward_gpu import build_model
from model.nerf_raybased import PointSampler
from r2l_amd.train_step import R2LTrainer

os.environ["R2L_FORCE_VARIANT"] = "coopf"
sd = O.make_state_dict(n_block=43, seed=0)
ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator().manual_seed(5)
o = (torch.randn(n, 3, generator=g) * 1.5).cuda(); d = torch.randn(n, 3, generator=g).cuda()
tgt = torch.rand(n, 3, generator=g).cuda(); tr = torch.rand(n, 16, generator=g).cuda()


def run(tiles, mode):
    os.environ["R2L_COOPF_TILES"] = tiles
    m = build_model(sd, 43)
    t = R2LTrainer(m, ps)
    if mode == "render":
        with torch.no_grad():
            return (m.forward_rays(o, d, ps).clone(),)
    rgb = t.forward_backward(o, d, tgt, perturb=1.0, t_rand=tr).clone()
    return rgb, t.grads.clone()


print("library:", os.environ.get("R2L_LIB_PATH", "(shipped)"))
for share in ("-",):
    for mode in ("render", "train"):
        ref = run("2", mode)
        bad, where = 0, None
        for i in range(10):
            r = run("1", mode)
            if not all(torch.equal(a, b) for a, b in zip(r, ref)):
                bad += 1
                if where is None:
                    dd = (r[0] - ref[0]).abs().amax(1)
                    idx = (dd > 0).nonzero().flatten()
                    tiles = torch.unique(idx // 32)
                    where = "rgb differs on %d rays (first %s; tiles < 256: %d, >= 256: %d), max %.3g, channels %s; grads equal: %s" % (
                        idx.numel(), idx[:4].tolist(), int((tiles < 256).sum()), int((tiles >= 256).sum()), dd.max().item(),
                        ((r[0] - ref[0]).abs().amax(0) > 0).tolist(), torch.equal(r[-1], ref[-1]) if mode == "train" else "-")
        print("N %d  %s %-6s one tile per workgroup vs two: %d of 10 launches differ%s" % (
            n, share, mode, bad, ("  [" + where + "]") if where else ""))
