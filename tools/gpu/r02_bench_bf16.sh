# bf16-mode bench line + rocprofv3 kernel stats (usage: bash tools/gpu/r02_bench_bf16.sh <tag>)
tag=${1:-r02x}
python bench.py --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-infer > gpurun_out/${tag}_bf16_bench.json 2> gpurun_out/${tag}_bf16_bench.err; tail -3 gpurun_out/${tag}_bf16_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bf16_bench.json'))
print(d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'])
for r in d['roofline']['all_mfma_kernels']: print('%-100s %8.1f TF  %6.2f ms/step  n=%d' % (r['kernel'][:100], r['achieved'], r['ms_per_step'], r['launches_per_step']))
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -- python $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-infer > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py gpurun_out/${tag}_prof gpurun_out/${tag}_bf16_kernel_stats | head -3; rm -rf gpurun_out/${tag}_prof
