export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_teacher_gpu.py -m gpu -q -x 2>&1 | tail -3
for r in 1 2; do
  echo "old:   $(R2L_LIB_PATH=$R/tools/_bin/t2old/libr2l_hip.so python tools/teacher_time.py | tail -1)"
  for reps in 1 2 4 8 16; do echo "reps $reps: $(R2L_T2_REPS=$reps python tools/teacher_time.py | tail -1)"; done
done
