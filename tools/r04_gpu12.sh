mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/t_all.log 2>&1; grep -n "passed\|failed" gpurun_out/t_all.log | tail -3
python __graft_entry__.py --smoke 2>&1 | grep smoke
bash tools/r04_profile.sh > gpurun_out/r04_profile.log 2>&1; tail -2 gpurun_out/r04_profile.log | cut -c1-200
