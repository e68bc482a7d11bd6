#!/bin/bash
# same-box A/B of library variants on the render leg only: tools/ab_render.sh libA.so libB.so ...
for r in 1 2; do for L in "$@"; do
  R2L_LIB_PATH=$(pwd)/r2l_amd/lib/$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-teacher --no-train 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'render %.3f ms' % d['ms_per_step'], '%.2f M rays/s' % (d['value']/1e6))"
done; done
