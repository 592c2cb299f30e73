#!/bin/bash
# sk_gemm with phases switched off (CONP_SK_DBG bits: 1 no panel build, 2 no MFMA, 4 no global loads, 8 no stagger, 16 no epilogue
# stores): TIMING ONLY -- the charges are garbage.  Prints the kernel's average duration per variant.
# The switch exists in the DIAGNOSTIC library only (make -C lammps-user-conp2_amd/csrc diag), loaded here through CONP_LIB.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export CONP_LIB="$PWD/lammps-user-conp2_amd/conp_amd/libconp_hip_diag.so"
[ -f "$CONP_LIB" ] || { echo "build the diagnostic library first: make -C lammps-user-conp2_amd/csrc diag" >&2; exit 1; }
for d in ${SK_ABLATE_SET:-0 1 4 5 2 16 8 21 23}; do
  CONP_SK_DBG=$d python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-configs > gpurun_out/abl_$d.json 2> gpurun_out/abl_$d.err
  python3 - "$d" <<'PY'
import json, sys
d = sys.argv[1]
r = json.loads(open(f"gpurun_out/abl_{d}.json").read().strip().splitlines()[-1])
print(f"dbg {d:>3}: sk_gemm {1e3 * r['kernels_ms']['sk_gemm']:7.1f} us   ms_per_step {r['ms_per_step']:.4f}", flush=True)
PY
done
