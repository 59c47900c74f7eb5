R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pg -- python $R/bench.py --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-infer --no-bf16 --no-wgrad-overlap > /dev/null 2>&1
cd $R && python tools/prof_dispatches.py gpurun_out/pg conv_gather_bf16 19; rm -rf gpurun_out/pg
