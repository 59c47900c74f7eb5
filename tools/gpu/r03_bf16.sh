# bf16 bench line (per-class HIP-event table, weight gradients in line) + a kernel trace of the same command
# usage: bash tools/gpu/r03_bf16.sh <tag>
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
python bench.py --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bf16_bench_line.json 2> gpurun_out/${tag}_bf16.err; tail -2 gpurun_out/${tag}_bf16.err
python bench.py --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-wgrad-overlap > gpurun_out/${tag}_bf16_inline_bench_line.json 2>> gpurun_out/${tag}_bf16.err
python - <<PY
import json
for n in ('bf16_bench_line', 'bf16_inline_bench_line'):
    d = json.load(open('gpurun_out/${tag}_%s.json' % n))
    print(n, d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d.get('inference'))
r = d['roofline']
for k in r['all_mfma_kernels'] + r['hbm_kernels']:
    print('  %-62s n=%5.1f avg=%8.3f ms=%7.3f frac=%.3f' % (k['kernel'][:62], k['launches_per_step'], k['avg_launch_ms'], k['ms_per_step'], k['frac']))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_bf16 -- python $R/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-wgrad-overlap > /dev/null 2>&1
cd $R && python tools/prof_summary.py gpurun_out/${tag}_prof_bf16 gpurun_out/${tag}_bf16_kernel_stats | head -2; rm -rf gpurun_out/${tag}_prof_bf16
