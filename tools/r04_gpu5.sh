mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/t_all.log 2>&1; tail -6 gpurun_out/t_all.log
python __graft_entry__.py --smoke 2>&1 | grep smoke
R2L_EQ_FAMILIES=0 python tools/train_equivalence.py 30000 > gpurun_out/train_eq.txt 2>&1; tail -8 gpurun_out/train_eq.txt
for a in "40" "200" "200 noise"; do python tools/e2e_render.py $a 2>&1 | grep "metrics\|files" ; done > gpurun_out/e2e_render.txt; cat gpurun_out/e2e_render.txt
python tools/e2e_train.py > gpurun_out/e2e_train.txt 2>&1; tail -1 gpurun_out/e2e_train.txt
bash tools/small_prof.sh 4096 > gpurun_out/small4096.txt 2>&1; head -30 gpurun_out/small4096.txt
bash tools/r04_profile.sh > gpurun_out/r04_profile.log 2>&1; tail -3 gpurun_out/r04_profile.log | cut -c1-300
