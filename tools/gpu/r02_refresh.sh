# refresh the PMC tables and the two config-2 bench lines from the current binary (usage: bash tools/gpu/r02_refresh.sh <tag>)
tag=${1:-r02z}
bash tools/gpu/pmc_traffic.sh $tag | tail -14
python - <<PY
import json
for sfx in ('', '_bf16'):
    p = 'gpurun_out/${tag}_pmc_traffic%s.json' % sfx
    d = json.load(open(p))
    d['collected'] = 'round 2 final binary, $tag'
    json.dump(d, open(p, 'w'), indent=1)
    json.dump(d, open('profiles/r02_pmc_traffic%s.json' % sfx, 'w'), indent=1)
PY
python bench.py --steps 8 --warmup 3 > gpurun_out/${tag}_f32_bench_line.json 2>/dev/null
python bench.py --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bf16_bench_line.json 2>/dev/null
python - <<PY
import json
for n in ('f32_bench_line','bf16_bench_line'):
    d=json.load(open('gpurun_out/${tag}_%s.json' % n)); r=d['roofline']
    print(n, d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d['inference']['frames_per_s'], d['inference']['frames_per_s_with_postprocess'], r['kernel'][:40], r['achieved'], r['frac'], r['traffic'])
PY
