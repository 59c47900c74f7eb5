tag=${1:-r04n}
python -m pytest tests/test_kernels.py -q -m gpu -x 2>&1 | tail -1
for i in 1 2; do
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-infer 2>/dev/null > gpurun_out/${tag}_f32_$i.json
python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_f32_$i.json'))
print('fp32', d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], (d.get('bf16_mode') or {}).get('ms_per_step'))
print([(r['kernel'][:30], r['frac'], r['ms_per_step']) for r in d['roofline']['all_mfma_kernels'][:6]])
PY
done
