"""Wall-clock of pseudo-data generation through the CLI surface (utils/create_data.py --create_data rand): 12 random
poses at 400x400, 64+128 samples, seeded D8 W256 teacher pair; shards shuffled and written by the background writer."""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from model.nerf_raybased import NeRF  # noqa: E402
from r2l_amd import create_data  # noqa: E402


def main(n_pose=12):
    tmp = tempfile.mkdtemp(prefix="r2l_cd_")
    os.chdir(tmp)
    torch.manual_seed(3)
    sds = []
    for _ in range(2):
        m = NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
        with torch.no_grad():
            m.alpha_linear.bias.add_(0.5)
        sds.append(m.state_dict())
    torch.save({"network_fn_state_dict": sds[0], "network_fine_state_dict": sds[1]}, os.path.join(tmp, "teacher.tar"))
    kd = os.path.join(tmp, "pseudo")
    argv = ["--create_data", "rand", "--config", os.path.join(ROOT, "configs", "lego.txt"), "--datadir",
            os.path.join(tmp, "no_scene"), "--teacher_ckpt", os.path.join(tmp, "teacher.tar"), "--create_data_chunk", "6",
            "--datadir_kd", "x:" + kd, "--experiment_name", "cd"]
    create_data.main(argv + ["--n_pose_kd", "2"])  # warm-up (imports, packing, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = create_data.main(argv + ["--n_pose_kd", str(n_pose), "--rm_existing_data"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("create_data rand: %d poses, %d rays in %.2f s = %.1f ms/pose = %.3f M rays/s; %d shard files" %
          (n_pose, out["n_rays"], dt, dt * 1e3 / n_pose, out["n_rays"] / dt / 1e6, len(os.listdir(kd))))


if __name__ == "__main__":
    main()
