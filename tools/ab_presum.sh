set -o pipefail
for v in auto p0 p1; do
  E="CONP_X=0"; [ $v = p0 ] && E="CONP_HC_PRESUM=0"; [ $v = p1 ] && E="CONP_HC_PRESUM=1"
  env $E python bench.py --steps 50 --no-cpu-baseline > gpurun_out/cfg_$v.json 2> gpurun_out/cfg_$v.err || exit 1
  python3 -c "
import json
r=json.loads(open('gpurun_out/cfg_$v.json').read().strip().splitlines()[-1])
print('$v', round(r['ms_per_step'],4), r['kernels_ms'])
for k,c in r['configs'].items(): print('   ', k, round(c['ms_per_update'],4), c['kernels_ms'])"
done
