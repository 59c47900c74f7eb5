# round 5, sixth GPU call: precision 'bf16x3' (fp32 arithmetic on the bf16 MFMA: exact three-way bf16 split of the ConvLSTM operands) --
# kernel / engine tests on the MI355X, accuracy against the fp32 engine at config-2 geometry, step time next to the fp32 step (same box)
tag=${1:-r05f}
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine.py -q -x -m gpu -k "split6 or bf16x3" -s 2>&1 | grep -v "^$" | tail -8
timeout 600 python tools/x3_compare.py > gpurun_out/${tag}_x3_compare.json 2> gpurun_out/${tag}_x3_compare.err; tail -3 gpurun_out/${tag}_x3_compare.err; head -c 3000 gpurun_out/${tag}_x3_compare.json
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$1', d['ms_per_step'], d['value'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], [(c['kernel'][:40], c['frac'], c['ms_per_step'], c['launches_per_step']) for c in r['all_mfma_kernels'][:6]], [(c['kernel'][:28], c['frac'], c['ms_per_step']) for c in r['hbm_kernels'][:4]])"; }
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-infer --no-variants --no-bf16 --no-x3"
for i in 1 2; do
$B 2>gpurun_out/${tag}_err1.log | line "fp32  "
$B --precision bf16x3 2>gpurun_out/${tag}_err2.log | line "bf16x3"
done 2>&1 | tee gpurun_out/${tag}_ab.log
tail -3 gpurun_out/${tag}_err2.log
$B --precision bf16x3 --by-shape gpurun_out/${tag}_x3_by_shape.json > gpurun_out/${tag}_x3_bench_line.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_x3_by_shape.json'))
for r in d['rows'][:40]: print('  %-46s n=%3d avg=%8.3f ms/step=%7.2f frac=%.3f work=%.3g' % (r['kernel'][:46], r['launches'], r['avg_launch_ms'], r['ms_per_step'], r['frac'], r['work_per_launch']))
print('sum', sum(r['ms_per_step'] for r in d['rows']))
PY
