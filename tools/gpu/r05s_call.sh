# round 5: kernel trace of one config-4 step in precision 'bf16x3' (step-by-step route, padded weight gradients) next to its bench line
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05s_prof_c4x3 -- python $R/bench.py --precision bf16x3 --hw 832 992 --batch 2 --unroll 16 --steps 1 --warmup 2 --no-bf16 --no-x3 --no-infer --no-cpu-baseline --no-variants > $R/gpurun_out/r05s_x3_c4_bench_line.json 2>/dev/null
cd $R && python tools/prof_summary.py gpurun_out/r05s_prof_c4x3 gpurun_out/r05s_x3_c4_kernel_stats 60 | head -14; python tools/trace_gaps.py gpurun_out/r05s_prof_c4x3 --step-marker adam_kernel | head -8; rm -rf gpurun_out/r05s_prof_c4x3
python -c "
import json; d=json.load(open('gpurun_out/r05s_x3_c4_bench_line.json')); print(d['value'], d['ms_per_step'], d['peak_hbm_gb'], d['allocator'])"
