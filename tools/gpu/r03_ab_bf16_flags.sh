# same-box A/B of the bf16 step with extra bench flags (usage: bash tools/gpu/r03_ab_bf16_flags.sh <tag> "<flags A>" "<flags B>")
tag=${1:-abb}; fa="$2"; fb="$3"
for rep in 1 2; do
  for v in A B; do
    fl="$fa"; [ $v = B ] && fl="$fb"
    python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-wgrad-overlap $fl > gpurun_out/${tag}_${v}${rep}.json 2>/dev/null
    python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_${v}${rep}.json'))
r = d['roofline']
print('$v$rep [$fl]', d['ms_per_step'], [(k['kernel'][:34], k['ms_per_step']) for k in r['all_mfma_kernels'][:3]])
PY
  done
done
