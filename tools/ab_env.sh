#!/bin/bash
# A/B of comparison switches on ONE box: bash tools/ab_env.sh [workload] "NAME=VALUE" ...   ("-" = product defaults); two rounds;
# AB_ARGS: more bench.py flags, e.g. AB_ARGS="--pppm 40 45 180"
set -o pipefail
W=headline; case "$1" in headline|big|headline_slab|headline_rough|il_onelayer|il_twolayer|dilute|cond2) W=$1; shift;; esac
for round in 1 2; do for e in "$@"; do
  E="$e"; [ "$e" = "-" ] && E="CONP_X=0"
  env $E python bench.py --workload $W --steps 60 --no-cpu-baseline --no-configs $AB_ARGS > gpurun_out/ab_env.json 2> gpurun_out/ab_env.err || exit 1
  python3 -c "
import json
r=json.loads(open('gpurun_out/ab_env.json').read().strip().splitlines()[-1]); print('$e', $round, round(r['value'],1), round(r['ms_per_step'],4), round(r['roofline']['frac'],4) if r.get('roofline') else '', r['kernels_ms'])"
done; done
