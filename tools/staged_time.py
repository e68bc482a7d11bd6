"""Single-GPU cost of the staged backward that the overlapped gradient all-reduce drives (r2l_backward_part: dX chain + tail,
body buckets, head) against the one-call r2l_backward, with and without CUs reserved for the collective kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else ""
    sizes = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (98304, 12288, 4096)
    dev = torch.device("cuda", 0)
    net, ps, _ = bench.make_model(dev)
    from r2l_amd.train_step import R2LTrainer, lr_schedule
    for n in sizes:
        g = torch.Generator().manual_seed(1)
        o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).to(dev)
        d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        tgt = torch.rand(n, 3, generator=g).to(dev)
        tr = R2LTrainer(net, ps)
        cases = [("one call", False, 4, None, 1), ("staged, 4 buckets", True, 4, None, 1),
                 ("staged, 4 buckets, 8 CUs reserved", True, 4, "8", 1), ("staged, 8 buckets, 8 CUs reserved", True, 8, "8", 1)]
        if tr.lib.r2l_chain_segments_ok_cfg(n, tr.eng.n_block, tr.eng._cfg()):
            # the dX chain itself cut into segments, weight gradients of a finished segment on a second stream beside the next
            cases += [("chain in %d segments, dW beside it" % k, False, 4, None, k) for k in (2, 3, 4, 6)]
            cases += [("chain in 4 segments, 8 CUs reserved", False, 4, "8", 4)]
        for case, staged, buckets, reserve, segments in cases:
            tr.force_staged, tr.n_buckets, tr.chain_segments = staged, buckets, segments
            if reserve:
                os.environ["R2L_RESERVE_CUS"] = reserve
            else:
                os.environ.pop("R2L_RESERVE_CUS", None)
            for i in range(3):
                tr.step(o, d, tgt, lr_schedule(i + 1, 5e-4, 500, "0.0001,200"), perturb=1.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            k = 20
            for i in range(k):
                tr.step(o, d, tgt, lr_schedule(i + 4, 5e-4, 500, "0.0001,200"), perturb=1.0)
            e1.record()
            torch.cuda.synchronize()
            print("%-28s %6d rays  %-36s %.3f ms/step" % (label, n, case, e0.elapsed_time(e1) / k))
        os.environ.pop("R2L_RESERVE_CUS", None)


if __name__ == "__main__":
    main()
