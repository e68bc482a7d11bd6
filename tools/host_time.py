"""Host-side cost of one eager training step (Python + ctypes + launch calls) against its device time, and the same step
replayed from a hipGraph of forward_backward (Adam stays eager: its bias corrections are host scalars)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    net, ps, _ = bench.make_model(dev)
    from r2l_amd.train_step import R2LTrainer
    for n in (4096, 12288, 98304):
        g = torch.Generator().manual_seed(1)
        o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).to(dev)
        d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        tgt = torch.rand(n, 3, generator=g).to(dev)
        tr = R2LTrainer(net, ps)
        for i in range(5):
            tr.step(o, d, tgt, 5e-4, perturb=1.0)
        torch.cuda.synchronize()
        k = 200 if n < 50000 else 50
        t0 = time.perf_counter()
        for i in range(k):
            tr.step(o, d, tgt, 5e-4, perturb=1.0)
        t_host = (time.perf_counter() - t0) / k
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / k
        # graph of forward_backward (jitter drawn inside the graph by torch's graph-safe generator)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            tr.forward_backward(o, d, tgt, perturb=1.0)
            tr.adam(5e-4)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            tr.forward_backward(o, d, tgt, perturb=1.0)
        tr.adam(5e-4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            graph.replay()
            tr.adam(5e-4)  # (marks the streams dirty on the host; the re-pack itself is inside the graph)
        t_host_g = (time.perf_counter() - t0) / k
        torch.cuda.synchronize()
        t_all_g = (time.perf_counter() - t0) / k
        print("%6d rays: eager host %.3f ms/step, wall %.3f | graph(fwd+bwd)+eager Adam host %.3f, wall %.3f ms/step"
              % (n, 1e3 * t_host, 1e3 * t_all, 1e3 * t_host_g, 1e3 * t_all_g))


if __name__ == "__main__":
    main()
