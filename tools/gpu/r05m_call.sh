# round 5, GPU call behind the final set: the engine-only additions to precision 'bf16x3' (wide Conv2D units on split operands, weight gradients of
# W % 32 != 0 layers on padded copies) -- tests, same-box A/B, config-4, and the bf16x3 lines / kernel trace of the final set again (same binary)
tag=${1:-r05m}
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine.py -q -x -m gpu -k "split or bf16x3" 2>&1 | tail -2
timeout 1800 python -m pytest tests/test_fullsize_gpu.py -q -x -m gpu -k "bf16x3" -s 2>&1 | grep -v "^$\|amdgpu" > gpurun_out/${tag}_fullsize_x3.log; tail -4 gpurun_out/${tag}_fullsize_x3.log | cut -c1-250
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$1', d['ms_per_step'], d['value'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], d.get('inference') and (d['inference']['frames_per_s'], d['inference']['frames_per_s_with_postprocess']), [(c['kernel'][:30], c['frac'], c['ms_per_step'], c['launches_per_step']) for c in r['all_mfma_kernels'][:6]], [(c['kernel'][:22], c['ms_per_step']) for c in r['hbm_kernels'][:3]])"; }
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-bf16 --no-x3 --no-infer"
for i in 1 2; do
$B --precision bf16x3 2>gpurun_out/${tag}_err1.log | line "x3 + conv units"
$B --precision bf16x3 --ab-x3-lstm-only 2>/dev/null | line "x3 lstm only   "
$B 2>/dev/null | line "fp32           "
done 2>&1 | tee gpurun_out/${tag}_ab.log
tail -2 gpurun_out/${tag}_err1.log
C4="--hw 832 992 --batch 2 --unroll 16 --steps 3 --warmup 2 --no-bf16 --no-x3 --no-infer --no-cpu-baseline --no-variants"
timeout 900 python bench.py $C4 --precision bf16x3 2>gpurun_out/${tag}_c4_x3.err | tee gpurun_out/${tag}_x3_c4_bench_line.json | line "c4 x3  "
tail -2 gpurun_out/${tag}_c4_x3.err
timeout 900 python bench.py $C4 2>/dev/null | tee gpurun_out/${tag}_f32_c4_bench_line.json | line "c4 fp32"
python bench.py --precision bf16x3 --steps 8 --warmup 3 --no-cpu-baseline --no-variants --by-shape gpurun_out/${tag}_x3_by_shape.json 2>/dev/null | tee gpurun_out/${tag}_x3_bench_line.json | line "x3 line"
python bench.py --precision bf16x3 --size 512 --batch 2 --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-infer 2>/dev/null | tee gpurun_out/${tag}_x3_c5shape_bench_line.json | line "x3 512 "
python tools/x3_compare.py > gpurun_out/${tag}_x3_compare_vs_fp32.json 2>/dev/null
python bench.py > gpurun_out/${tag}_f32_bench_line.json 2> gpurun_out/${tag}_f32_bench.err; tail -1 gpurun_out/${tag}_f32_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_f32_bench_line.json'))
print('default line:', d['value'], d['ms_per_step'], 'bf16x3_mode', {k: v for k, v in d['bf16x3_mode'].items() if k not in ('what','mfma_kernels')}, 'bf16', d['bf16_mode']['ms_per_step'])
print('variants', {k: {p: (v[p]['ms_per_step'], v[p].get('frac_of_peak')) for p in ('fp32','bf16','bf16x3')} for k, v in d['variants'].items()})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_x3 -- python $R/bench.py --precision bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-variants > /dev/null 2>&1
cd $R && python tools/prof_summary.py gpurun_out/${tag}_prof_x3 gpurun_out/${tag}_x3_kernel_stats 60 | head -14; rm -rf gpurun_out/${tag}_prof_x3
