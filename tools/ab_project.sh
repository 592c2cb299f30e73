#!/bin/bash
# A/B on ONE box: sk_gemm projecting its tiles (product, planar electrodes) vs partial tiles + reducing launch (CONP_SK_PARTIALS=1);
# optional: the stamped build's epilogue share first.  Output: gpurun_out/ab_project_*.json
set -o pipefail
if [ "$1" = stamp ]; then python3 tools/sk_stamp.py headline 5 > gpurun_out/stamp_proj.txt 2>&1 && grep -E "epilogue" gpurun_out/stamp_proj.txt; fi
for v in base partials base2 partials2; do
  E="CONP_X=0"; case $v in partials*) E="CONP_SK_PARTIALS=1";; esac
  env $E python bench.py --steps 100 --no-cpu-baseline --no-configs > gpurun_out/ab_project_$v.json 2> gpurun_out/ab_project_$v.err || exit 1
done
python3 - <<'PY'
import json
for v in ("base", "partials", "base2", "partials2"):
    r = json.loads(open("gpurun_out/ab_project_%s.json" % v).read().strip().splitlines()[-1])
    print(v, round(r["value"], 1), round(r["ms_per_step"], 4), round(r["roofline"]["frac"], 4), r["kernels_ms"])
PY
