#!/bin/bash
# Round-5 GPU call 2: chain-trip diagnosis, new tests, layer-pipeline probe, quick bench (run from the repo root via gpurun).
export TMPDIR=/tmp
OUT=gpurun_out/r05b
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 300 python tools/diag_chain_trip.py > $OUT/diag_chain_trip.txt 2>&1
cat $OUT/diag_chain_trip.txt | cut -c1-900
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_forward_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "adam_packed or dx_chain or fp16_range or gradient_scale or explicit_config or three_adam or two_tiles" > $OUT/tests.log 2>&1
tail -8 $OUT/tests.log
for cfg in "10 128 4" "10 128 8" "10 512 4" "12 128 4"; do
  timeout 120 tools/_bin/layer_pipeline_probe $cfg >> $OUT/layer_pipeline_probe.txt 2>&1
done
cat $OUT/layer_pipeline_probe.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05b/bench.json'))
print(json.dumps(d['summary'])[:3000])
print(json.dumps(d['raw2outputs'])[:1500])
PY
