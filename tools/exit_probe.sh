#!/bin/bash
# Runs on the GPU box: tools/exit_probe.py under `rocprofv3 --kernel-trace --stats`, one variant after the other, recording the
# exit status of each and keeping the logs + /proc/self/maps under gpurun_out/exit_probe/ (see DESIGN.md "exit-time fault").
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/exit_probe
mkdir -p $OUT
for v in ${VARIANTS:-torch load create project invert invert_single setup setup_rows_host linalg update host}; do
  export PROBE_TAG=_prof
  [ $v = invert_single ] && export CONP_PANEL_SINGLE=1
  [ $v = setup_rows_host ] && export CONP_ROWS_HOST=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -- python3 tools/exit_probe.py $v $OUT > $OUT/$v.log 2>&1
  echo "variant $v exit=$?" | tee -a $OUT/status.txt
  unset CONP_PANEL_SINGLE CONP_ROWS_HOST
done
unset PROBE_TAG
# the same without the profiler (must be clean too)
for v in ${PLAIN:-update host}; do
  python3 tools/exit_probe.py $v $OUT > $OUT/plain_$v.log 2>&1
  echo "plain $v exit=$?" | tee -a $OUT/status.txt
done
exit 0
