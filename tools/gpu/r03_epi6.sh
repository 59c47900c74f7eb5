KB=misc python tools/kbench.py new 2>&1 | grep -v amdgpu.ids | grep -v bf16
python -m pytest tests/test_kernels.py tests/test_engine.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-infer --no-bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['ms_per_step'], d['value'], [(c['kernel'][:28], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'][:6]])"
done
