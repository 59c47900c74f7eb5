# quick regression + speed check after a kernel / dispatch change (usage: bash tools/gpu/r02_check.sh <tag> [pytest args])
tag=${1:-chk}
python -m pytest tests/test_kernels.py tests/test_engine.py tests/test_fullsize_gpu.py tests/test_drivers.py -q -m gpu -x > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log
python tools/inf_try.py 2>&1 | grep -v amdgpu | grep "None cap 32\|None cap 16"
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_f32.json 2> gpurun_out/${tag}_f32.err; tail -1 gpurun_out/${tag}_f32.err
python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_f32.json'))
print('fp32', d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d.get('inference'), (d.get('bf16_mode') or {}))
PY
