python tools/tile_fit.py 2>&1 | grep "^frames [24]:"
KB_SHAPES=0,1,3,5 python tools/k3bench.py new 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_kernels.py tests/test_engine.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-wgrad-overlap --by-shape gpurun_out/r03_epi_bf16_by_shape.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['ms_per_step'], d['value'])"
done
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-infer --no-bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['ms_per_step'], d['value'])"
