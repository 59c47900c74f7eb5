# full measurement set of round 3 -- run as the LAST GPU call of the round so that every table is from the final binary
# (usage: bash tools/gpu/r03_final.sh <tag>); everything lands in gpurun_out/<tag>_*, the caller copies it into profiles/r03_*
# order: GPU tests, PMC passes (stamped with the library's build id; the bench lines below quote them as roofline.traffic and
# say traffic_stale = false only if the ids match), bench lines, kernel-trace profiles (weight gradients in line so that the
# traced averages are the ones the bench's HIP events see)
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -s > gpurun_out/${tag}_gpu_tests.log 2>&1; tail -3 gpurun_out/${tag}_gpu_tests.log
grep "^T=8\|^dp2\|^dp[0-9] over\|^bench --gpus\|config-" gpurun_out/${tag}_gpu_tests.log | cut -c1-300
bash tools/gpu/pmc_traffic.sh $tag | tail -14
python - <<PY
import json
for sfx in ('', '_bf16'):
    p = 'gpurun_out/${tag}_pmc_traffic%s.json' % sfx
    d = json.load(open(p))
    d['collected'] = 'round 3 final binary %s, $tag' % d.get('build_id')
    json.dump(d, open(p, 'w'), indent=1)
    json.dump(d, open('profiles/r03_pmc_traffic%s.json' % sfx, 'w'), indent=1)      # (the box's copy: read by bench.py below)
PY
python bench.py --steps 8 --warmup 3 --by-shape gpurun_out/${tag}_f32_by_shape.json > gpurun_out/${tag}_f32_bench_line.json 2> gpurun_out/${tag}_f32_bench.err; tail -2 gpurun_out/${tag}_f32_bench.err
python bench.py --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bf16_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
python bench.py --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-wgrad-overlap --by-shape gpurun_out/${tag}_bf16_by_shape.json > gpurun_out/${tag}_bf16_inline_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
python bench.py --precision bf16 --size 512 --batch 2 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bf16_c5shape_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
python bench.py --hw 832 992 --batch 2 --unroll 16 --steps 2 --warmup 1 --no-bf16 --no-infer --no-cpu-baseline > gpurun_out/${tag}_f32_c4_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
python - <<PY
import json
for n in ('f32_bench_line','bf16_bench_line','bf16_inline_bench_line','bf16_c5shape_bench_line','f32_c4_bench_line'):
    try:
        d=json.load(open('gpurun_out/${tag}_%s.json' % n))
        print(n, d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], d.get('build_id'), (d.get('roofline') or {}).get('traffic_stale'), d.get('inference'), (d.get('bf16_mode') or {}).get('frames_per_s'))
    except Exception as e: print(n, 'FAILED', e)
PY
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16; do
  short=f32; [ $mode = bf16 ] && short=bf16
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_$mode -- python $R/bench.py --precision $mode --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-bf16 --no-wgrad-overlap > /dev/null 2>&1
  (cd $R && python tools/prof_summary.py gpurun_out/${tag}_prof_$mode gpurun_out/${tag}_${short}_kernel_stats | head -2; rm -rf gpurun_out/${tag}_prof_$mode)
done
cd $R && bash tools/gpu/r02_inf_prof.sh ${tag}_inf 2>&1 | grep "launches/frame"
python tools/post_ab.py bf16 gpurun_out/${tag}_post_832x992.json 2>&1 | grep -v amdgpu | tail -9
# bf16 vs fp32 training on the same stream (the bf16 accuracy contract: loss curves, held-out IoU, argmax agreement)
python tools/train_compare.py 300 > gpurun_out/${tag}_bf16_vs_fp32_training.json 2> gpurun_out/${tag}_train_compare.err; tail -c 400 gpurun_out/${tag}_bf16_vs_fp32_training.json
