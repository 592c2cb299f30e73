#!/bin/bash
# Runs the given steps one after the other on the GPU box; every step is "name|timeout seconds|command".  A step that fails by an
# assertion lets the next one run; a step that is killed at its limit (124 / 137) ends the call: no further GPU work behind a hang.
mkdir -p gpurun_out
for step in "$@"; do
  name="${step%%|*}"; rest="${step#*|}"; lim="${rest%%|*}"; cmd="${rest#*|}"
  echo "== $name (limit ${lim}s): $cmd"
  timeout -k 10 "$lim" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"; tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name was killed at its limit: stopping here"; exit $rc; fi
done
exit 0
