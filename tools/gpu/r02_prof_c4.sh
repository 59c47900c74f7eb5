# kernel profile of one config-4 fp32 step and one config-5-shape bf16 step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pc4 -- python $R/bench.py --hw 832 992 --batch 2 --unroll 16 --steps 1 --warmup 1 --no-bf16 --no-infer --no-cpu-baseline > /dev/null 2>&1
cd $R && python tools/prof_summary.py gpurun_out/pc4 gpurun_out/r02_c4_kernel_stats | head -26; rm -rf gpurun_out/pc4
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pc5 -- python $R/bench.py --precision bf16 --size 512 --batch 2 --steps 1 --warmup 1 --no-infer --no-cpu-baseline > /dev/null 2>&1
cd $R && python tools/prof_summary.py gpurun_out/pc5 gpurun_out/r02_c5_kernel_stats | head -22; rm -rf gpurun_out/pc5
