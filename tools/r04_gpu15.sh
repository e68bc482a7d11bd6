mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/t_all.log 2>&1; grep -n "passed\|failed" gpurun_out/t_all.log | tail -3
python __graft_entry__.py --smoke 2>&1 | grep smoke
bash tools/r04_profile.sh > gpurun_out/r04_profile.log 2>&1
python tools/e2e_create_data.py > gpurun_out/e2e_create_data.txt 2>&1; tail -1 gpurun_out/e2e_create_data.txt
R2L_EQ_FAMILIES=0 python tools/train_equivalence.py 100000 > gpurun_out/train_eq100k.txt 2>&1; tail -4 gpurun_out/train_eq100k.txt
