export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in 1 2; do
  for v in "" t2nosin t2noheads t2nobar t2noamax; do
    if [ -z "$v" ]; then L=$R/r2l_amd/lib/libr2l_hip.so; else L=$R/tools/_bin/$v/libr2l_hip.so; fi
    echo "${v:-shipped}: $(R2L_LIB_PATH=$L python tools/teacher_time.py | tail -1)"
  done
done
