python -m pytest tests/test_kernels.py tests/test_engine.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-wgrad-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['ms_per_step'], d['value'], [(c['kernel'][:26], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'] if 's2' in c['kernel']])"
done
