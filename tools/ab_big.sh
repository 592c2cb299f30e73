#!/bin/bash
# A/B of sk_gemm builds on one box: product vs round-2 library vs loop-form MFMA phase, headline and 16384 / 262144 box.
cd "$(dirname "$0")/.."
D=$PWD/lammps-user-conp2_amd/conp_amd
run() {   # name lib workload steps
  CONP_LIB=$2 python3 bench.py --workload $3 --steps $4 --warmup 5 --no-cpu-baseline --no-configs > gpurun_out/ab_$1_$3.json 2> gpurun_out/ab_$1_$3.err
  python3 - "$1" "$3" <<'PY'
import json, sys
n, w = sys.argv[1], sys.argv[2]
try:
    r = json.loads(open(f"gpurun_out/ab_{n}_{w}.json").read().strip().splitlines()[-1])
    print(f"{n:10s} {w:9s} ms/update {r['ms_per_step']:.4f}  sk_gemm frac {r['roofline']['frac']:.4f}  {r['kernels_ms']}", flush=True)
except Exception as e:
    print(n, w, "failed", e, flush=True)
PY
}
for w in headline big; do
  st=100; [ $w = big ] && st=10
  run product $D/libconp_hip.so $w $st
  run r2 $D/libconp_hip_r2.so $w $st
  run loopform $D/libconp_hip_loopform.so $w $st
  run product2 $D/libconp_hip.so $w $st
done
