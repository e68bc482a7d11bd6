#!/usr/bin/env python
"""Teacher pseudo-data generation, same command line as the reference's utils/create_data.py ('rand' mode):

  python utils/create_data.py --create_data rand --config configs/lego.txt --teacher_ckpt <teacher.tar> \
      --n_pose_kd 10000 --datadir_kd data/nerf_synthetic/lego:data/nerf_synthetic/lego_pseudo_images10k

Implementation: r2l_amd/create_data.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r2l_amd.create_data import main  # noqa: E402

if __name__ == "__main__":
    main()
