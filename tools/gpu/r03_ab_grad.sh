python -m pytest tests/test_engine.py tests/test_fullsize_gpu.py -m gpu -x -q -k "stored_as_bf16 or bf16 or config5 or T8" 2>&1 | tail -3
for i in 1 2; do
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-wgrad-overlap --by-shape gpurun_out/r03x_bf16_by_shape.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 new', d['ms_per_step'], d['value'])"
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-wgrad-overlap --ab-f32-grad --by-shape gpurun_out/r03x_bf16_by_shape_old.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 f32-grad', d['ms_per_step'], d['value'])"
done
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 new overlap', d['ms_per_step'], d['value'])"
