python -m pytest tests/test_kernels.py -q -m gpu -x -k "wgrad" 2>&1 | tail -1
python tools/wgbench.py lean 2>&1 | grep -v amdgpu.ids
python tools/wgbench.py lean 2>&1 | grep -v amdgpu.ids
for net in params lstm3; do
for i in 1 2; do
python bench.py --net $net --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-variants --no-wgrad-overlap 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$net', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:26], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'][:5]])"
done
done
