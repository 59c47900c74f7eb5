set -x
python -m pytest tests/test_kernels.py tests/test_engine.py -x -q -m gpu > gpurun_out/r02_t2.log 2>&1; tail -15 gpurun_out/r02_t2.log
python bench.py --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-infer > gpurun_out/r02a_bf16_bench.json 2> gpurun_out/r02a_bf16_bench.err; tail -c 3000 gpurun_out/r02a_bf16_bench.json; tail -5 gpurun_out/r02a_bf16_bench.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02a_prof -- python $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-infer > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py gpurun_out/r02a_prof gpurun_out/r02a_bf16_kernel_stats; rm -rf gpurun_out/r02a_prof
