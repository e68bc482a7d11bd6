export TMPDIR=/tmp
python tools/train_equivalence.py 12000 > gpurun_out/train_eq12k.txt 2>&1; tail -22 gpurun_out/train_eq12k.txt
