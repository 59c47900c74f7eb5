export KB=tape16
for i in 1 2; do
KB_LIB=abl_tmp/lib_old.so python tools/kbench.py old 2>&1 | grep -v amdgpu.ids | grep "step"
python tools/kbench.py new 2>&1 | grep -v amdgpu.ids | grep "step"
done
python -m pytest tests/test_kernels.py tests/test_engine.py -m gpu -x -q -k "lstm or gates or fused or tape or bf16 or lazily" 2>&1 | tail -3
for i in 1 2; do
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --by-shape gpurun_out/r03w_bf16_by_shape.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['ms_per_step'], d['value'])"
done
