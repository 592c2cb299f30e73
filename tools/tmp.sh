python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'])"
