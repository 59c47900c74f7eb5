python tools/tile_fit.py 2>&1 | grep "^frames [24]:"
KB_SHAPES=0,1,3,5 python tools/k3bench.py new 2>&1 | grep -v amdgpu.ids | grep "src=bf16"
python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-wgrad-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['ms_per_step'], d['value'], [(c['kernel'][:24], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'][:5]])"
done
