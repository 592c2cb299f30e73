#!/bin/bash
# Every workload of profiles/rNN_*: the headline and the BASELINE configs under rocprofv3 (tools/profile_gpu.sh each), summaries
# gathered in gpurun_out/profiles_new/ (copy what is to be judged into profiles/).   bash tools/profile_all.sh [round tag]
set -o pipefail
TAG=${1:-r05}
DST=gpurun_out/profiles_new; mkdir -p $DST
run() {   # name, bench args
  local name=$1; shift
  OUT=gpurun_out/prof_$name STEPS=${STEPS:-200} BENCH_ARGS="--no-configs $*" bash tools/profile_gpu.sh > gpurun_out/prof_$name.log 2>&1 || { tail -5 gpurun_out/prof_$name.log; return 1; }
  cp gpurun_out/prof_$name/summary.txt $DST/${TAG}_${name}_summary.txt
  cp gpurun_out/prof_$name/summary.json $DST/${TAG}_${name}_summary.json
  local st=$(find gpurun_out/prof_$name/trace -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp $st $DST/${TAG}_${name}_kernel_stats.csv
  rm -rf gpurun_out/prof_$name
  echo "== $name done" >&2
}
run bench_headline && run classic_headline --classic && run il_onelayer --workload il_onelayer && run il_twolayer_cg --workload il_twolayer --solver cg \
  && run il_onelayer_pppm --workload il_onelayer --pppm 40 45 180 && run headline_slab --workload headline_slab \
  && run headline_rough --workload headline_rough
ls -la $DST
