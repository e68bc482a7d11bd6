#!/bin/bash
# same-box A/B of library variants on the teacher frame: tools/ab_teacher_time.sh <dirA> <dirB> ...   (directories holding a
# libr2l_hip.so, e.g. r2l_amd/lib tools/_bin/t2old; three interleaved rounds of tools/teacher_time.py)
R=$(pwd)
for r in 1 2 3; do for L in "$@"; do
  echo "$L: $(R2L_LIB_PATH=$R/$L/libr2l_hip.so python tools/teacher_time.py | tail -1)"
done; done
