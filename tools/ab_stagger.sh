#!/bin/bash
# sk_gemm with and without the stagger of the SIMD partner waves (diagnostic library, CONP_SK_DBG bit 8), headline and big box
cd "$(dirname "$0")/.."
export CONP_LIB=$PWD/lammps-user-conp2_amd/conp_amd/libconp_hip_diag.so
for w in headline big; do
  st=100; [ $w = big ] && st=10
  for d in 0 8 0 8; do
    CONP_SK_DBG=$d python3 bench.py --workload $w --steps $st --warmup 5 --no-cpu-baseline --no-configs > gpurun_out/stg_${w}_$d.json 2> gpurun_out/stg_${w}_$d.err
    python3 - "$w" "$d" <<'PY'
import json, sys
w, d = sys.argv[1], sys.argv[2]
r = json.loads(open(f"gpurun_out/stg_{w}_{d}.json").read().strip().splitlines()[-1])
print(f"{w:9s} dbg {d}: ms/update {r['ms_per_step']:.4f}  sk_gemm {1e3 * r['kernels_ms']['sk_gemm']:.1f} us  frac {r['roofline']['frac']:.4f}", flush=True)
PY
  done
done
