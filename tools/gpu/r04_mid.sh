tag=${1:-r04m}
python -m pytest tests/test_kernels.py -q -m gpu -x 2>&1 | tail -2
python tools/inf_hostprof.py bf16 300 2>&1 | grep "frames, host"
python tools/inf_hostprof.py fp32 100 2>&1 | grep "frames, host"
python bench.py --net default5 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-infer --no-variants --no-wgrad-overlap 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('default5 bf16', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:40], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'][:8]])"
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants 2>/dev/null > gpurun_out/${tag}_f32.json
python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_f32.json'))
print('fp32', d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d.get('inference'), (d.get('bf16_mode') or {}))
print([(r['kernel'][:30], r['frac'], r.get('clock_mhz'), r.get('frac_of_clocked_peak')) for r in d['roofline']['all_mfma_kernels'][:5]])
PY
