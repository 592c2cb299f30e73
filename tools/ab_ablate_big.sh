#!/bin/bash
# phases of sk_gemm switched off (timing only) at the 16384 / 262144 size: current diagnostic library vs the round-2 library
cd "$(dirname "$0")/.."
D=$PWD/lammps-user-conp2_amd/conp_amd
for lib in diag r2; do
  for d in 0 1 4 5 2 16; do
    CONP_LIB=$D/libconp_hip_$lib.so CONP_SK_DBG=$d python3 bench.py --workload ${1:-big} --steps ${2:-10} --warmup 3 --no-cpu-baseline --no-configs > gpurun_out/abl2_${lib}_$d.json 2> gpurun_out/abl2_${lib}_$d.err
    python3 - "$lib" "$d" <<'PY'
import json, sys
l, d = sys.argv[1], sys.argv[2]
r = json.loads(open(f"gpurun_out/abl2_{l}_{d}.json").read().strip().splitlines()[-1])
print(f"{l:5s} dbg {d:>2s}: sk_gemm {1e3 * r['kernels_ms']['sk_gemm']:9.1f} us", flush=True)
PY
  done
done
