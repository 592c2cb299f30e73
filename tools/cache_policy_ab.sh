#!/bin/bash
# A/B of cache policies on the headline update: which streams are non-temporal (GEMV matrix, partial tiles, phase tables).
# Comparison libraries: make -C lammps-user-conp2_amd/csrc policy_p policy_t policy_pt.  Prints ms per update and the kernels.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
D=$PWD/lammps-user-conp2_amd/conp_amd
run() {   # name, lib, GEMV_NT
  CONP_LIB=$2 CONP_GEMV_NT=$3 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs > gpurun_out/pol_$1.json 2> gpurun_out/pol_$1.err
  python3 - "$1" <<'PY'
import json, sys
n = sys.argv[1]
r = json.loads(open(f"gpurun_out/pol_{n}.json").read().strip().splitlines()[-1])
print(f"{n:28s} ms/update {r['ms_per_step']:.4f}  {r['kernels_ms']}", flush=True)
PY
}
run product_gemv_nt        $D/libconp_hip.so 1
run gemv_plain             $D/libconp_hip.so 0
run gemv_plain_part_nt     $D/libconp_hip_policy_p.so 0
run gemv_plain_tables_nt   $D/libconp_hip_policy_t.so 0
run gemv_plain_both_nt     $D/libconp_hip_policy_pt.so 0
run gemv_nt_part_nt        $D/libconp_hip_policy_p.so 1
