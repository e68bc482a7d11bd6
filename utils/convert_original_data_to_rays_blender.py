#!/usr/bin/env python
"""Real Blender images -> `train_<k>.npy` ray shards for the fine-tuning stage (same flags as the reference script):

  python utils/convert_original_data_to_rays_blender.py --splits train --datadir data/nerf_synthetic/lego

Implementation: r2l_amd.data.convert_images_to_ray_shards."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r2l_amd.data import convert_images_to_ray_shards  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--splits", type=str, default="train")
    ap.add_argument("--datadir", type=str, required=True)
    ap.add_argument("--suffix", type=str, default="")
    ap.add_argument("--ignore", type=str, default="", help="comma-separated image indices to skip")
    ap.add_argument("--full_res", action="store_true")
    a = ap.parse_args()
    savedir, n = convert_images_to_ray_shards(a.datadir, a.splits.split(","), a.suffix,
                                              [i for i in a.ignore.split(",") if i], a.full_res)
    print('%d shards saved at "%s"' % (n, savedir))
