"""Would the HBM-bound weight-gradient kernel hide under the matrix-bound dX chain at 98 304 rays?  Times r2l_backward_part's
dX chain (+ tail) and body weight gradients one after the other on one stream, and side by side on two streams with the
weight-gradient grid held to 256 - reserve workgroups (the operands of the body launch are the previous step's: same values,
so the timing is that of a real overlapped step; nothing is checked here)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    net, ps, _ = bench.make_model(dev)
    from r2l_amd import _lib
    from r2l_amd.engine import _ptr, _stream
    from r2l_amd.train_step import R2LTrainer
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
    g = torch.Generator().manual_seed(1)
    o = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).to(dev)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    tgt = torch.rand(n, 3, generator=g).to(dev)
    u = torch.rand(n, 16, generator=g).to(dev)
    tr = R2LTrainer(net, ps)
    eng = tr.eng
    rgb = tr.forward_backward(o, d, tgt, perturb=1., t_rand=u)
    ztab = eng.ztab(ps.z_vals, 1.)
    side = torch.cuda.Stream()
    nb = eng.n_block

    def args(stream):
        return (_ptr(o), _ptr(d), _ptr(u), _ptr(ztab), None, _ptr(rgb), _ptr(tgt), None, _ptr(tr.save_x), _ptr(tr.save_t),
                _ptr(tr.wstream_bwd), _ptr(eng.flat), nb, 2.0 / (3.0 * n), _ptr(tr.dpre), _ptr(tr.gx), _ptr(tr.gt),
                _ptr(tr.sqerr), _ptr(tr.grads), _ptr(tr.dw_slab), n, ctypes.c_void_p(stream))

    part = tr.lib.r2l_backward_part_cfg

    def chain(stream):
        _lib.check(part(*args(stream), _lib.BWD_CHAIN | _lib.BWD_TAIL, 0, 0, eng._cfg()), "chain")

    def body(stream):
        _lib.check(part(*args(stream), _lib.BWD_BODY, 0, 2 * nb, eng._cfg()), "body")

    def timed(fn, k=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k

    main_s = torch.cuda.current_stream()

    def serial():
        chain(main_s.cuda_stream)
        body(main_s.cuda_stream)

    def both(first):
        def run():
            ev = torch.cuda.Event()
            ev.record(main_s)
            side.wait_event(ev)
            if first == "body":
                body(side.cuda_stream)
                chain(main_s.cuda_stream)
            else:
                chain(main_s.cuda_stream)
                body(side.cuda_stream)
            done = torch.cuda.Event()
            done.record(side)
            main_s.wait_event(done)
        return run

    print("%d rays: chain alone %.3f ms, body alone %.3f ms" % (n, timed(lambda: chain(main_s.cuda_stream)),
                                                             timed(lambda: body(main_s.cuda_stream))))
    for reserve in (-1, 64, 128):
        eng.set_config(reserve_cus=reserve)
        print("  weight-gradient grid %3d workgroups: body alone %.3f, serial %.3f, side by side (chain first) %.3f, (body first) %.3f ms"
              % (256 - max(reserve, 0), timed(lambda: body(main_s.cuda_stream)), timed(serial), timed(both("chain")),
                 timed(both("body"))))


if __name__ == "__main__":
    main()
