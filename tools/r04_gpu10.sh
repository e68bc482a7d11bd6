export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_teacher_gpu.py -m gpu -q -x 2>&1 | tail -1
for r in 1 2 3; do
  echo "select-both: $(R2L_LIB_PATH=$R/tools/_bin/t2both/libr2l_hip.so python tools/teacher_time.py | tail -1)"
  echo "branch:      $(python tools/teacher_time.py | tail -1)"
done
