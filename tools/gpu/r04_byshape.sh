# per-layer-shape HIP-event tables of the bf16 step (bench.py --by-shape), weight gradients in line
tag=${1:-r04s}; shift
for net in ${@:-params lstm3}; do
python bench.py --net $net --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-infer --no-variants --no-wgrad-overlap --by-shape gpurun_out/${tag}_${net}_bf16_by_shape.json $EXTRA > gpurun_out/${tag}_${net}_bf16.json 2>/dev/null
python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_${net}_bf16_by_shape.json'))
print('$net', json.load(open('gpurun_out/${tag}_${net}_bf16.json'))['ms_per_step'])
for r in d['rows'][:${ROWS:-40}]:
    print('  %-44s %9.3g  x%-3d %8.4f ms  %7.3f ms/step  frac %.3f' % (r['kernel'][:44], r['work_per_launch'], r['launches'], r['avg_launch_ms'], r['ms_per_step'], r['frac']))
PY
done
