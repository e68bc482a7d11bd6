export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_forward_gpu.py tests/test_fullsize_gpu.py tests/test_train_gpu.py -m gpu -q 2>&1 | tail -4
for r in 1 2 3; do for L in tools/_bin/trigboth r2l_amd/lib; do
  R2L_LIB_PATH=$R/$L/libr2l_hip.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-teacher --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'render %.3f ms/launch kernel %.3f ms frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
for r in 1 2; do for L in tools/_bin/trigboth r2l_amd/lib; do echo "$L $(R2L_LIB_PATH=$R/$L/libr2l_hip.so python tools/exact_time.py 2>&1 | grep 'dw_mode fp16' | head -1)"; done; done
