export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in 1 2 3; do for L in r2l_amd/lib tools/_bin/f2nosin; do
  R2L_LIB_PATH=$R/$L/libr2l_hip.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-teacher --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'render %.3f ms/launch kernel %.3f ms frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
