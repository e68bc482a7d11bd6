"""Wall-clock of the test-set evaluation loop (driver.render_path: render + PSNR + SSIM + PNG writing of prediction and
ground truth) on 40 synthetic 400x400 views, W256 D88."""
import argparse
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from model.nerf_raybased import NeRF_v3_2, PointSampler  # noqa: E402
from r2l_amd import data, driver  # noqa: E402


class _Log:
    def info(self, *a):
        if "frames in" in str(a[0]):
            print(*a)


def main(n=40):
    dev = torch.device("cuda")
    trial = argparse.Namespace(ON=True, body_arch="resmlp", inact="relu", outact="none", res_scale=1., n_learnable=2,
                               n_block=-1, near=-1, far=-1)
    args = argparse.Namespace(netdepth=88, netwidth=256, layerwise_netwidths="", act="relu", linear_tail=False,
                              use_residual=True, trial=trial)
    torch.manual_seed(0)
    net = NeRF_v3_2(args, 1008, 3).to(dev)
    ps = PointSampler(400, 400, 555.5555155968841, 16, 2., 6., device=dev)
    poses = torch.stack([data.pose_spherical(-180. + 9. * i, -30., 4.) for i in range(n)]).to(dev)
    # ground-truth frames: smooth images (bilinear upsampling of 25x25 noise; real test frames — an object on a white
    # background — compress at least as well) or, `noise`, incompressible ones: zlib's worst case, 10x the encode time
    if len(sys.argv) > 2 and sys.argv[2] == "noise":
        gts = torch.rand(n, 400, 400, 3)
    else:
        gts = torch.nn.functional.interpolate(torch.rand(n, 3, 25, 25), size=400, mode="bilinear").permute(0, 2, 3, 1).contiguous()
    out = tempfile.mkdtemp(prefix="r2l_frames_")
    for tag, sd in (("metrics only", None), ("metrics + PNGs", out)):
        driver.render_path(poses[:3], net, ps, dev, _Log(), gt_imgs=gts[:3], savedir=None)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, misc = driver.render_path(poses, net, ps, dev, _Log(), gt_imgs=gts, savedir=sd)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-15s: %.1f ms/frame (%d frames, psnr %.3f ssim %.4f)" % (tag, dt * 1e3 / n, n, misc["test_psnr"].item(),
                                                                      misc["test_ssim"].item()))
    print("files:", len(os.listdir(out)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
