mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 tools/_bin/cu_exchange_probe 2000 > gpurun_out/cu_exchange.txt 2>&1; cat gpurun_out/cu_exchange.txt
for a in "40" "200" "40 noise" "200 noise"; do python tools/e2e_render.py $a 2>&1 | grep "metrics\|files" ; done > gpurun_out/e2e_render.txt; cat gpurun_out/e2e_render.txt
for L in r2l_amd/lib tools/_bin/midlate tools/_bin/dwnt tools/_bin/both r2l_amd/lib tools/_bin/both; do echo "== $L"; R2L_LIB_PATH=$R/$L/libr2l_hip.so python tools/exact_time.py 2>&1 | grep "dw_mode exact" ; done > gpurun_out/exact_ab.txt; cat gpurun_out/exact_ab.txt
cd /tmp; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_e2e -o e2e -- python $R/tools/e2e_train.py > $R/gpurun_out/e2e_train_prof.txt 2>&1; cd $R; ls gpurun_out/prof_e2e/ gpurun_out/prof_e2e/* | head
