#!/bin/bash
# same-box A/B of library variants: tools/ab.sh libA.so libB.so  (interleaved, 2 rounds)
for r in 1 2; do for L in "$@"; do
  R2L_LIB_PATH=$(pwd)/r2l_amd/lib/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-teacher 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'render %.3f ms' % d['fast_mode']['ms_per_step'], 'train %.3f ms' % d['fast_mode']['train']['ms_per_step'], 'train4096 %.3f ms' % d['fast_mode']['train_4096']['ms_per_step'])"
done; done
