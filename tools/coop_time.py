"""Main vs cooperative chain variants at small batch sizes (dev aid): R2L_FORCE_VARIANT is read once per process."""
import sys, os, time, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from oracle import r2l_oracle as O
    from tests.test_forward_gpu import build_model
    from model.nerf_raybased import PointSampler
    from r2l_amd.train_step import R2LTrainer
    m = build_model(O.make_state_dict(43, seed=0), 43)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6.)
    tr = R2LTrainer(m, ps)
    out = {}
    for n in (2048, 4096, 8192, 12288, 16384, 24576, 40000):
        o = torch.randn(n, 3, device="cuda"); d = torch.randn(n, 3, device="cuda"); t = torch.rand(n, 3, device="cuda")
        with torch.no_grad():
            for _ in range(2): m.forward_rays(o, d, ps)
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(5): m.forward_rays(o, d, ps)
            torch.cuda.synchronize(); f = (time.time() - t0) / 5
        for _ in range(2): tr.step(o, d, t, 1e-4, perturb=1.)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): tr.step(o, d, t, 1e-4, perturb=1.)
        torch.cuda.synchronize(); s = (time.time() - t0) / 5
        out[n] = (f * 1e3, s * 1e3)
    print(json.dumps(out))
else:
    res = {}
    for v in ("main", "coopf", "coop16"):  # ("coop", the 32-ray fp32-MFMA cooperative family, was retired in round 5)
        env = dict(os.environ, R2L_FORCE_VARIANT=v)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        res[v] = json.loads(r.stdout.strip().splitlines()[-1])
    print("%8s | %10s %10s %10s | %10s %10s %10s" % ("rays", "fwd main", "fwd coopf", "fwd c16", "step main", "step coopf",
                                                       "step c16"))
    for n in res["main"]:
        print("%8s | %8.3f ms %8.3f ms %8.3f ms | %8.3f ms %8.3f ms %8.3f ms" % (
            n, res["main"][n][0], res["coopf"][n][0], res["coop16"][n][0], res["main"][n][1], res["coopf"][n][1],
            res["coop16"][n][1]))
