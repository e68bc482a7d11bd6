#!/bin/bash
# same-box A/B of library variants on the teacher leg: tools/ab_teacher.sh libA.so libB.so
for r in 1 2; do for L in "$@"; do
  R2L_LIB_PATH=$(pwd)/r2l_amd/lib/$L python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'teacher %.2f ms/frame' % d['teacher']['ms_per_frame'])"
done; done
