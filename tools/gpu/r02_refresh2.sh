# refresh the kernel-trace CSVs (config-2, both precisions) and the config-5-shape line from the current binary
tag=${1:-r03e}
R=$GRAFT_REPO_ROOT
python bench.py --precision bf16 --size 512 --batch 2 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bf16_c5shape_bench_line.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/${tag}_bf16_c5shape_bench_line.json')); print('c5', d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'])"
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16; do
  short=f32; [ $mode = bf16 ] && short=bf16
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_$mode -- python $R/bench.py --precision $mode --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-bf16 > /dev/null 2>&1
  (cd $R && python tools/prof_summary.py gpurun_out/${tag}_prof_$mode gpurun_out/${tag}_${short}_kernel_stats | head -5; rm -rf gpurun_out/${tag}_prof_$mode)
done
