# kernel profile of the bf16-mode training step (usage: bash tools/gpu/r02_prof_bf16.sh <tag>)
tag=${1:-pb}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof -- python $R/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-bf16 > $R/gpurun_out/${tag}_line.json 2> /dev/null
cd $R && python tools/prof_summary.py gpurun_out/${tag}_prof gpurun_out/${tag}_bf16_kernel_stats | head -16; python tools/prof_overlap.py gpurun_out/${tag}_prof 0.5; rm -rf gpurun_out/${tag}_prof
python -c "
import json; d=json.load(open('gpurun_out/${tag}_line.json')); print(d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'])"
