# quick regression + speed check (usage: bash tools/gpu/r03_check.sh <tag> [all])
tag=${1:-chk}
if [ "$2" = "all" ]; then
  python -m pytest tests -q -m gpu -s > gpurun_out/${tag}_tests.log 2>&1
else
  python -m pytest tests/test_kernels.py tests/test_engine.py tests/test_fullsize_gpu.py tests/test_drivers.py -q -m gpu -x > gpurun_out/${tag}_tests.log 2>&1
fi
tail -3 gpurun_out/${tag}_tests.log; grep "^T=8\|^dp2\|^bench --gpus" gpurun_out/${tag}_tests.log | cut -c1-330
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_f32.json 2> gpurun_out/${tag}_f32.err; tail -1 gpurun_out/${tag}_f32.err
python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_f32.json'))
print('fp32', d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d.get('inference'), (d.get('bf16_mode') or {}))
PY
