"""Wall-clock of the whole training loop through the CLI surface (main.py flags of README step 3): native shard reader
-> H2D -> hard-ray pool -> fused step, W256 D88, N_rand 20 (81 920 rays) + 16 384 hard rays.  Two runs of different
length are timed and subtracted so that start-up (imports, packing, pool fill) drops out.  Extra command-line arguments are passed
through to main.py (e.g. `python tools/e2e_train.py --r2l_precision fp32_mfma`: the loop on the graded arithmetic)."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from r2l_amd import data, driver  # noqa: E402
from tests.test_driver_cpu import make_scene  # noqa: E402


def main(n_files=240, it_a=230, it_b=530):
    tmp = tempfile.mkdtemp(prefix="r2l_e2e_")
    os.chdir(tmp)
    scene = os.path.join(tmp, "scene")
    os.makedirs(scene)
    make_scene(scene, size=128)
    kd = os.path.join(tmp, "pseudo")
    os.makedirs(kd)
    rng = np.random.RandomState(0)
    for k in range(0, n_files, 40):  # synthetic [4096,9] shards: origins on the r=4 sphere, inward directions
        o = rng.randn(40 * 4096, 3).astype(np.float32)
        o *= 4. / np.linalg.norm(o, axis=1, keepdims=True)
        d = (-o / 4. + 0.2 * rng.randn(40 * 4096, 3)).astype(np.float32)
        rows = np.concatenate([o, d, rng.rand(40 * 4096, 3).astype(np.float32)], 1)
        data.write_ray_shards(rows, kd, k)
    common = ["--model_name", "R2L", "--config", os.path.join(ROOT, "configs", "lego_noview.txt"), "--datadir", scene,
              "--n_sample_per_ray", "16", "--netwidth", "256", "--netdepth", "88", "--use_residual", "--trial.ON",
              "--trial.body_arch", "resmlp", "--testskip", "1", "--datadir_kd", kd, "--data_mode", "rays",
              "--N_rand", "20", "--hard_ratio", "0.2", "--hard_mul", "20", "--warmup_lr", "0.0001,200",
              "--i_print", "100", "--i_testset", "100000", "--i_weights", "100000", "--num_workers", "8"] + sys.argv[1:]
    out = {}
    # (an untimed first run: one-off costs of the process — library load, first-touch of the big buffers — would otherwise
    # sit in the first timed run only and not cancel in the subtraction)
    driver.main(common + ["--N_iters", "130", "--experiment_name", "e2e_w"])
    for tag, iters in (("a", it_a), ("b", it_b)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        driver.main(common + ["--N_iters", str(iters), "--experiment_name", "e2e_" + tag])
        torch.cuda.synchronize()
        out[tag] = time.perf_counter() - t0
    per = (out["b"] - out["a"]) / (it_b - it_a)
    rays = 81920 + 16384
    print("e2e: run %d it %.2f s, run %d it %.2f s -> %.3f ms/iter with the hard pool full = %.3f M rays/s "
          "(%d rays/iter)" % (it_a, out["a"], it_b, out["b"], per * 1e3, rays / per / 1e6, rays))


if __name__ == "__main__":
    main()
