# same-box A/B of the fp32 step: bench.py with / without extra flags, alternating (usage: bash tools/gpu/r03_ab_f32.sh <tag> "<flags A>" "<flags B>")
tag=${1:-abf}; fa="$2"; fb="$3"
for rep in 1 2; do
  for v in A B; do
    fl="$fa"; [ $v = B ] && fl="$fb"
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-infer --no-bf16 $fl > gpurun_out/${tag}_${v}${rep}.json 2>/dev/null
    python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_${v}${rep}.json'))
r = d['roofline']
print('$v$rep [$fl]', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(k['kernel'][:22], k['frac']) for k in r['all_mfma_kernels'][:3]])
PY
  done
done
