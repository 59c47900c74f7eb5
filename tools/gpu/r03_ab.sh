# A/B of the bf16 step on ONE box (the part's power state moves bf16 numbers by ~2 % between boxes): old decoder tail vs new,
# alternating.  usage: bash tools/gpu/r03_ab.sh <tag> [extra bench args]
tag=${1:-ab}; shift
for rep in 1 2; do
  for v in old new; do
    fl=""; [ $v = old ] && fl="--ab-old-tail"
    python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer $fl "$@" > gpurun_out/${tag}_${v}${rep}.json 2>/dev/null
    python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_${v}${rep}.json'))
print('$v$rep', d['ms_per_step'], d['step_tflops_achieved_per_gpu'])
PY
  done
done
