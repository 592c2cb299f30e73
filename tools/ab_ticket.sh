#!/bin/bash
# who adds a row tile's projected pieces, on ONE box: its last segment inside sk_gemm (product at the headline size) vs hc_sum
# (CONP_HC_PRESUM=1) vs the dot kernel (CONP_HC_PRESUM=0)
set -o pipefail
W=${1:-headline}
for round in 1 2; do for v in ticket p1 p0; do
  E="CONP_X=0"; [ $v = p0 ] && E="CONP_HC_PRESUM=0"; [ $v = p1 ] && E="CONP_HC_PRESUM=1"
  env $E python bench.py --workload $W --steps 60 --no-cpu-baseline --no-configs > gpurun_out/ab_ticket_$v.json 2> gpurun_out/ab_ticket_$v.err || exit 1
  python3 -c "
import json
r=json.loads(open('gpurun_out/ab_ticket_$v.json').read().strip().splitlines()[-1]); print('$v', $round, round(r['value'],1), round(r['ms_per_step'],4), round(r['roofline']['frac'],4), r['kernels_ms'])"
done; done
