mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_driver_gpu.py tests/test_metrics_gpu.py -m gpu -q -x 2>&1 | tail -5
for L in r2l_amd/lib tools/_bin/midlate tools/_bin/both r2l_amd/lib tools/_bin/midlate tools/_bin/both; do echo "== $L"; R2L_LIB_PATH=$R/$L/libr2l_hip.so python tools/exact_time.py 2>&1 | grep "dw_mode" ; done > gpurun_out/exact_ab.txt; cat gpurun_out/exact_ab.txt
for a in "40" "200" "200 noise"; do python tools/e2e_render.py $a 2>&1 | grep "metrics\|files" ; done > gpurun_out/e2e_render.txt; cat gpurun_out/e2e_render.txt
python tools/e2e_train.py > gpurun_out/e2e_train.txt 2>&1; tail -1 gpurun_out/e2e_train.txt
