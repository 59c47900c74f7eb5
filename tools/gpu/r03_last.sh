python -m pytest tests/test_engine.py tests/test_fullsize_gpu.py tests/test_dp_equivalence.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['ms_per_step'], d['value'], d['build_id'])"
done
