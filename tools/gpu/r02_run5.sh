python -m pytest tests -x -q -m gpu > gpurun_out/r02_t5.log 2>&1; tail -8 gpurun_out/r02_t5.log
python bench.py --steps 4 --warmup 2 > gpurun_out/r02c_f32_bench.json 2> gpurun_out/r02c_f32_bench.err; tail -3 gpurun_out/r02c_f32_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/r02c_f32_bench.json'))
print(d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d['inference'], d['bf16_mode'])
print(d['cpu_baseline'])
for r in d['roofline']['all_mfma_kernels']: print('%-100s %8.1f TF  %6.2f ms/step  n=%d' % (r['kernel'][:100], r['achieved'], r['ms_per_step'], r['launches_per_step']))
PY
