#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace statistics and PMC passes for bench.py.
# Outputs under gpurun_out/prof/; the summaries judged are copied into profiles/ by tools/collect_profiles.py.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/prof}
mkdir -p $OUT
STEPS=${STEPS:-300}
ARGS="bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-profile ${BENCH_ARGS:-}"
echo "== kernel trace" >&2
# Every pass must EXIT CLEANLY.  (Round 1 masked a SIGSEGV inside exit() here; its cause was the HIP runtime's cooperative-launch
# queue, created by hipLaunchCooperativeKernel in the inverse, whose teardown faults under rocprofv3 -- tools/exit_probe.sh.)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
echo "== pmc SQ pass" >&2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; exit 1; }
echo "== pmc SQ pass 2" >&2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || { tail -5 $OUT/pmc_sq2.log; exit 1; }
echo "== pmc FETCH pass" >&2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
echo "== pmc WRITE pass" >&2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
find $OUT -name "*.csv" | head -40 >&2
python3 tools/collect_profiles.py $OUT $OUT/summary.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep what is judged (summary, kernel statistics) and drop the per-dispatch CSVs: gpurun merges at most 64 MiB back
find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
find $OUT -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
du -sh $OUT >&2
