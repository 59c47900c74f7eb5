# round 5, first GPU call: (1) config-4 ALONE and FIRST on the fresh box ("find the 0.8 s"): bench line
# with allocator counters, then a rocprofv3 kernel trace of two steps and its gap analysis; (2) the production-geometry
# gradient tests; (3) the baseline lines of this binary on this box (fp32 / bf16) for later same-box comparisons.
tag=${1:-r05a}
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --hw 832 992 --batch 2 --unroll 16 --steps 2 --warmup 1 --no-bf16 --no-infer --no-cpu-baseline --no-variants > gpurun_out/${tag}_f32_c4_bench_line.json 2> gpurun_out/${tag}_c4.err
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_f32_c4_bench_line.json'))
print('c4', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], d.get('allocator'))
r=d['roofline']
tot=0
for k in r['all_mfma_kernels']+r['hbm_kernels']:
    tot+=k['ms_per_step']; print('   %8.2f ms %5.3f %4d  %s' % (k['ms_per_step'], k['frac'], k['launches_per_step'], k['kernel'][:70]))
print('   listed classes sum %.1f ms' % tot)
PY
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_c4 -- python $R/bench.py --hw 832 992 --batch 2 --unroll 16 --steps 1 --warmup 1 --no-bf16 --no-infer --no-cpu-baseline --no-variants > /dev/null 2> $R/gpurun_out/${tag}_c4_prof.err)
python tools/prof_summary.py gpurun_out/${tag}_prof_c4 gpurun_out/${tag}_c4_kernel_stats 60 | head -30
python tools/trace_gaps.py gpurun_out/${tag}_prof_c4 --tail 0.34 --out gpurun_out/${tag}_c4_gaps.json | tee gpurun_out/${tag}_c4_gaps.txt
rm -rf gpurun_out/${tag}_prof_c4
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -x -s -k "every_gradient or ragged_gradients" > gpurun_out/${tag}_grad_tests.log 2>&1; tail -5 gpurun_out/${tag}_grad_tests.log
grep -E "^(w64|c2|c4)|torch-fp32 vs|HIP fp32|bf16 mode|worst max" gpurun_out/${tag}_grad_tests.log
python bench.py --no-variants > gpurun_out/${tag}_f32_bench_line.json 2> gpurun_out/${tag}_f32.err
python bench.py --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-variants --no-infer > gpurun_out/${tag}_bf16_bench_line.json 2>> gpurun_out/${tag}_f32.err
python - <<PY
import json
for n in ('f32','bf16'):
    d=json.load(open('gpurun_out/${tag}_%s_bench_line.json' % n))
    print(n, d['ms_per_step'], d['value'], d['step_tflops_achieved_per_gpu'], d.get('allocator'), (d.get('bf16_mode') or {}).get('ms_per_step'))
    for k in d['roofline']['all_mfma_kernels']: print('   %8.2f ms %5.3f %4d  %s' % (k['ms_per_step'], k['frac'], k['launches_per_step'], k['kernel'][:70]))
PY
