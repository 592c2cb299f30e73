python -m pytest tests/test_gpu_parity.py tests/test_gpu_decks.py -x -q 2>&1 | tail -2
for wl in headline il_onelayer il_twolayer dilute cond2; do
echo -n "$wl: "; python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print(d['value'], d['ms_per_step'], k)"
done
