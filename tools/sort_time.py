"""bench.py's raw2outputs leg alone (hipGraph replay over cycled HBM-resident input sets): us per 32 768-ray launch of
r2l_raw2outputs16_kernel (S = 64 / 192) and r2l_sample_pdf_sort16_kernel (random u / det u).  python tools/sort_time.py [label]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

r = bench.raw2outputs_leg(torch.device("cuda", 0), 20, 3)
print("%-28s S64 %.2f us  S192 %.2f us  sample_pdf_sort %.2f us (262144 rays: %.1f us)  det %.2f us"
      % (sys.argv[1] if len(sys.argv) > 1 else "", r["S64"]["us_per_launch"], r["S192"]["us_per_launch"],
         r["sample_pdf_sort"]["us_per_launch"], r["sample_pdf_sort"]["at_262144_rays"]["us_per_launch"],
         r["sample_pdf_sort_det"]["us_per_launch"]), flush=True)
