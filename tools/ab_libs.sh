#!/bin/bash
# A/B of library builds on ONE box (boxes differ by 1-2 %): bash tools/ab_libs.sh [workload] name1 name2 ...   (name = "product" or
# the NAME of lammps-user-conp2_amd/conp_amd/libconp_hip_NAME.so, e.g. a `make variant_NAME VDEF=...` build); two rounds each;
# AB_ARGS: more bench.py flags, e.g. AB_ARGS='--solver cg'
set -o pipefail
W=headline; case "$1" in headline|big|headline_slab|headline_rough|il_onelayer|il_twolayer|dilute|cond2) W=$1; shift;; esac
D=lammps-user-conp2_amd/conp_amd
for round in 1 2; do for n in "$@"; do
  L=$D/libconp_hip_$n.so; [ $n = product ] && L=$D/libconp_hip.so
  CONP_LIB=$PWD/$L python bench.py --workload $W --steps 60 --no-cpu-baseline --no-configs $AB_ARGS > gpurun_out/ab_$n.$round.json 2> gpurun_out/ab_$n.$round.err || exit 1
  python3 -c "
import json,sys
r=json.loads(open('gpurun_out/ab_$n.$round.json').read().strip().splitlines()[-1]); print('$n', $round, round(r['value'],1), round(r['ms_per_step'],4), round(r['roofline']['frac'],4) if r.get('roofline') else '', r['kernels_ms'])"
done; done
