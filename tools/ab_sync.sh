#!/bin/bash
# host-buffer update (conp_fix_pre_force with LAMMPS' host arrays) on ONE box: product vs comparison switches
for e in ${ENVS:-CONP_X=0 CONP_RESULTS_COPY=1 CONP_SYNC_BLOCK=1 CONP_X=0 CONP_RESULTS_COPY=1}; do
env $e CONP_TIME_HOST=1 python bench.py --steps 50 --no-cpu-baseline --no-configs > gpurun_out/th.json 2> gpurun_out/th.err || exit 1
echo $e; grep -i "host-buffer" gpurun_out/th.err | sed 's/.*calls: //'; python3 -c "
import json
r=json.loads(open('gpurun_out/th.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['ms_per_step_host_buffers_pcie'])"
done
