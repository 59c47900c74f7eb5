# full measurement set of round 6 -- run as the LAST GPU call of the round so that every table is from the final binary
# (usage: bash tools/gpu/r06_final.sh <tag>); everything lands in gpurun_out/<tag>_*, the caller copies it into profiles/r06_*
# config-4: two warm-up steps (the caching allocator's pool stops growing in step 2: profiles/r06_c4_trace_per_step_and_gaps.txt)
# order: PMC passes (stamped with the library's build id; the bench lines below quote them as roofline.traffic / clock_mhz and say
# *_stale = false only if the ids match), bench lines (the driver's default command first), kernel-trace profiles (weight gradients
# in line so that the traced averages are the ones the bench's HIP events see), DP-over-gloo line, streaming repeatability
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_smoke.log
# the whole GPU suite on the final tree first (its log is the round's test evidence)
[ -n "$RUN_SUITE" ] && { timeout 2400 python -m pytest tests -q -m gpu -s --durations=20 > gpurun_out/${tag}_gpu_tests.log 2>&1; tail -3 gpurun_out/${tag}_gpu_tests.log; }
bash tools/gpu/pmc_traffic.sh $tag | tail -14
# ... and (round 6) for the other workloads the bench lines below report: config-4, the config-5 per-GPU shape, the two net variants
PMC_EXTRA="--hw 832 992 --batch 2 --unroll 16" PMC_SFX=_c4 PMC_MODES=fp32 bash tools/gpu/pmc_traffic.sh $tag | tail -4
PMC_EXTRA="--size 512 --batch 2" PMC_SFX=_c5shape PMC_MODES=bf16 bash tools/gpu/pmc_traffic.sh $tag | tail -4
for net in lstm3 default5; do PMC_EXTRA="--net $net" PMC_SFX=_$net PMC_MODES="fp32 bf16" bash tools/gpu/pmc_traffic.sh $tag | tail -6; done
python - <<PY
import json, glob, os
for p in sorted(glob.glob('gpurun_out/${tag}_pmc_traffic*.json')):
    d = json.load(open(p))
    d['collected'] = 'round 6 final binary %s, $tag' % d.get('build_id')
    json.dump(d, open(p, 'w'), indent=1)
    json.dump(d, open('profiles/r06_' + os.path.basename(p)[len('${tag}_'):], 'w'), indent=1)      # (the box's copy: read by bench.py below)
for p in sorted(glob.glob('gpurun_out/${tag}_pmc_mfma_util*.json')):
    json.dump(json.load(open(p)), open('profiles/r06_' + os.path.basename(p)[len('${tag}_'):], 'w'), indent=1)
PY
python bench.py > gpurun_out/${tag}_f32_bench_line.json 2> gpurun_out/${tag}_f32_bench.err; tail -2 gpurun_out/${tag}_f32_bench.err
python bench.py --steps 8 --warmup 3 --no-variants --no-cpu-baseline --by-shape gpurun_out/${tag}_f32_by_shape.json > gpurun_out/${tag}_f32_long_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
python bench.py --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/${tag}_bf16_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
# precision 'bf16x3' (fp32 arithmetic on the bf16 MFMA): its own line (with its streaming inference), its by-shape table, and the step next to the fp32 engine's
python bench.py --precision bf16x3 --steps 8 --warmup 3 --no-cpu-baseline --no-variants --by-shape gpurun_out/${tag}_x3_by_shape.json > gpurun_out/${tag}_x3_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
for fl in 0 8; do python bench.py --precision bf16x3 --conv-flags $fl --steps 8 --warmup 3 --no-cpu-baseline --no-variants --no-infer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16x3 step, conv flags $fl (8 = LU_CONV_F_XCD_BY_N):', d['ms_per_step'])" | tee -a gpurun_out/${tag}_x3_xcd_by_n_ab.log; done
python tools/x3_compare.py > gpurun_out/${tag}_x3_compare_vs_fp32.json 2>> gpurun_out/${tag}_f32_bench.err
python bench.py --precision bf16x3 --size 512 --batch 2 --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-infer > gpurun_out/${tag}_x3_c5shape_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
python bench.py --size 512 --batch 2 --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-infer --no-bf16 --no-x3 > gpurun_out/${tag}_f32_c5shape_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
python bench.py --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-variants --no-infer --no-wgrad-overlap --by-shape gpurun_out/${tag}_bf16_by_shape.json > gpurun_out/${tag}_bf16_inline_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
python bench.py --precision bf16 --size 512 --batch 2 --steps 6 --warmup 2 --no-cpu-baseline --no-variants > gpurun_out/${tag}_bf16_c5shape_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
python bench.py --hw 832 992 --batch 2 --unroll 16 --steps 3 --warmup 2 --no-bf16 --no-x3 --no-infer --no-cpu-baseline --no-variants > gpurun_out/${tag}_f32_c4_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
for net in lstm3 default5; do for prec in fp32 bf16 bf16x3; do
python bench.py --net $net --precision $prec --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-infer --no-bf16 --no-x3 --by-shape gpurun_out/${tag}_${net}_${prec}_by_shape.json > gpurun_out/${tag}_${net}_${prec}_bench_line.json 2>> gpurun_out/${tag}_f32_bench.err
done; done
python bench.py --force-collectives --sync-bn --steps 5 --warmup 2 --no-variants --no-cpu-baseline --no-infer --no-bf16 --no-x3 > gpurun_out/${tag}_forced_rccl_bench_line.json 2> gpurun_out/${tag}_forced_rccl.err
LU_DP_BACKEND=gloo python bench.py --gpus 2 --check --sync-bn --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_dp2_gloo_bench_line.json 2> gpurun_out/${tag}_dp2_gloo.err; grep "check" gpurun_out/${tag}_dp2_gloo.err | tail -1
python - <<PY
import json
for n in ('f32_bench_line','f32_long_bench_line','x3_bench_line','x3_c5shape_bench_line','f32_c5shape_bench_line','bf16_bench_line','bf16_inline_bench_line','bf16_c5shape_bench_line','f32_c4_bench_line',
          'lstm3_fp32_bench_line','lstm3_bf16_bench_line','lstm3_bf16x3_bench_line','default5_fp32_bench_line','default5_bf16_bench_line','default5_bf16x3_bench_line','dp2_gloo_bench_line'):
    try:
        d=json.load(open('gpurun_out/${tag}_%s.json' % n))
        r=d.get('roofline') or {}
        print(n, d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], d.get('build_id'), r.get('traffic_stale'), r.get('clock_stale'), d.get('inference'), (d.get('bf16_mode') or {}).get('ms_per_step'))
        if n == 'dp2_gloo_bench_line': print('   dp:', {k: v for k, v in d['dp'].items() if k in ('bucket_trace','exposed_allreduce_ms','allreduce_ms_per_step','self_check')})
        if n == 'f32_bench_line': print('   variants:', {k: {p: (v[p]['ms_per_step'], v[p].get('frac_of_peak', v[p].get('frac_of_bf16_peak'))) for p in ('fp32','bf16','bf16x3')} for k, v in d['variants'].items()}); print('   cpu:', d['cpu_baseline']['value'], d['cpu_baseline']['tensorflow_probe']); print('   bf16x3_mode:', {k: v for k, v in d['bf16x3_mode'].items() if k not in ('what','mfma_kernels')})
    except Exception as e: print(n, 'FAILED', e)
PY
# streaming repeatability: three processes per precision (the bench's own inference block)
for prec in fp32 bf16 bf16x3; do for i in 1 2 3; do
python bench.py --precision $prec --steps 1 --warmup 1 --no-cpu-baseline --no-variants --no-bf16 --no-x3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream $prec run $i', d['inference']['frames_per_s'], d['inference']['frames_per_s_with_postprocess'], d['inference']['frames_per_s_with_postprocess_runs'])"
done; done | tee gpurun_out/${tag}_streaming_5runs.log
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16 bf16x3; do
  short=f32; [ $mode = bf16 ] && short=bf16; [ $mode = bf16x3 ] && short=x3
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_$mode -- python $R/bench.py --precision $mode --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-bf16 --no-x3 --no-variants --no-wgrad-overlap > /dev/null 2>&1
  (cd $R && python tools/prof_summary.py gpurun_out/${tag}_prof_$mode gpurun_out/${tag}_${short}_kernel_stats | head -2; rm -rf gpurun_out/${tag}_prof_$mode)
done
cd $R && bash tools/gpu/r02_inf_prof.sh ${tag}_inf 2>&1 | grep "launches/frame"
python tools/post_ab.py bf16 gpurun_out/${tag}_post_832x992.json 2>&1 | grep -v amdgpu | tail -9
# instruction / wait counters per kernel (SQ_INSTS_* per MFMA, wave-cycle shares): profiles/r06_pmc_sq.json
bash tools/gpu/pmc_sq.sh $tag 2>&1 | tail -12
# config-4: kernel trace of one step (per-kernel table) -- the allocator-pool finding of this round is in profiles/r06_c4_trace_per_step_and_gaps.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_c4 -- python $R/bench.py --hw 832 992 --batch 2 --unroll 16 --steps 1 --warmup 2 --no-bf16 --no-x3 --no-infer --no-cpu-baseline --no-variants > /dev/null 2>&1)
python tools/prof_summary.py gpurun_out/${tag}_prof_c4 gpurun_out/${tag}_c4_kernel_stats 60 | head -4; python tools/trace_gaps.py gpurun_out/${tag}_prof_c4 --step-marker adam_kernel | head -6; rm -rf gpurun_out/${tag}_prof_c4
# bf16 vs fp32 training on the same stream (the bf16 accuracy contract: loss curves, held-out IoU, argmax agreement)
python tools/train_compare.py 300 > gpurun_out/${tag}_bf16_vs_fp32_training.json 2> gpurun_out/${tag}_train_compare.err; tail -c 400 gpurun_out/${tag}_bf16_vs_fp32_training.json
