#!/bin/bash
# Same-box interleaved A/B of environment / library settings under ONE timer script (replaces the per-session r05_runN.sh files):
#
#   tools/run_ab.sh <out-name> <rounds> "<timer command>" "<label>|<ENV=VAL ...>" ["<label>|<ENV=VAL ...>" ...]
#
# builds the library, then runs the timer once per setting, `rounds` times over (settings interleaved: boxes of the pool differ
# by +-3 % on the power-capped kernels and drift while warming up), appending every line the timer prints to
# gpurun_out/<out-name>/ab.txt.  The timer gets the label as its first argument (or wherever the timer command says LABEL).  With PROF=1 every setting is also run once under
# `rocprofv3 --kernel-trace --stats` (kernel averages -> gpurun_out/<out-name>/prof_<k>_kernel_stats.csv).  Examples:
#   tools/run_ab.sh r06_mixed 3 "python tools/small_step_time.py" "two-tile|R2L_COOPF_TILES=2" "mixed|R2L_COOPF_TILES=3"
#   tools/run_ab.sh r06_lib 3 "python tools/train_step_time.py LABEL 40 98304 fp32_mfma" "base|A=1" "variant|R2L_LIB_PATH=tools/_bin/v/libr2l_hip.so"
export TMPDIR=/tmp
name=$1; rounds=$2; timer=$3; shift 3
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$name
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1 || { tail -20 "$OUT/build.log"; exit 1; }
run_timer() {  # env assignments as arguments; the label of the current setting in $label
  case "$timer" in
    *LABEL*) env "$@" bash -c "${timer//LABEL/\"$label\"}" ;;
    *) env "$@" $timer "$label" ;;
  esac
}
for r in $(seq 1 "$rounds"); do
  for setting in "$@"; do
    label=${setting%%|*}; envs=${setting#*|}
    # shellcheck disable=SC2086
    run_timer $envs 2>&1 | grep -v "amdgpu.ids" >> "$OUT/ab.txt"
  done
done
if [ -n "$PROF" ]; then
  k=0
  for setting in "$@"; do
    label=${setting%%|*}; envs=${setting#*|}; k=$((k + 1))
    # shellcheck disable=SC2086
    (cd /tmp && env $envs rocprofv3 --kernel-trace --stats -d "$OUT/prof_$k" -o p --output-format csv -- bash -c "cd $ROOT && $(case "$timer" in *LABEL*) echo "${timer//LABEL/\"$label\"}" ;; *) echo "$timer '$label'" ;; esac)" > "$OUT/prof_$k.log" 2>&1)
    f=$(find "$OUT/prof_$k" -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && { echo "# $label ($envs)" > "$OUT/prof_${k}_kernel_stats.csv"; head -25 "$f" >> "$OUT/prof_${k}_kernel_stats.csv"; }
    rm -rf "$OUT/prof_$k"
  done
fi
cat "$OUT/ab.txt"
