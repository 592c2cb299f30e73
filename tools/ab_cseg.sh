#!/bin/bash
# stream-K cost model: the per-segment constant (CONP_SK_CSEG, comparison switch) on ONE box; two rounds
set -o pipefail
W=${1:-headline}
for round in 1 2; do for v in ${CSEGS:-5.74 9 12 15}; do
  CONP_SK_CSEG=$v python bench.py --workload $W --steps 60 --no-cpu-baseline --no-configs > gpurun_out/ab_cseg_$v.json 2> gpurun_out/ab_cseg_$v.err || exit 1
  python3 -c "
import json
r=json.loads(open('gpurun_out/ab_cseg_$v.json').read().strip().splitlines()[-1]); print('cseg $v', $round, round(r['value'],1), round(r['ms_per_step'],4), round(r['roofline']['frac'],4), r['kernels_ms']['sk_gemm'])"
done; done
