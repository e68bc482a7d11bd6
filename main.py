#!/usr/bin/env python
"""R2L on MI355X — same command line as the reference's main.py for the accelerated path, e.g.

  python main.py --model_name R2L --config configs/lego_noview.txt --n_sample_per_ray 16 --netwidth 256 --netdepth 88 \
      --use_residual --trial.ON --trial.body_arch resmlp --pretrained_ckpt <ckpt.tar> --render_only --render_test --testskip 1
  torchrun --nproc-per-node 8 main.py ... --datadir_kd data/lego_pseudo_images10k --data_mode rays --N_rand 20 \
      --hard_ratio 0.2 --hard_mul 20 --warmup_lr 0.0001,200 --N_iters 1200000

Implementation: r2l_amd/driver.py (host logic) + r2l_amd/csrc (HIP kernels behind include/r2l_hip.h)."""
from r2l_amd.driver import main

if __name__ == "__main__":
    main()
