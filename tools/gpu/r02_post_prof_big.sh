R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pp -- python $R/tools/post_prof.py 832 992 2>&1 | grep "postprocess alone"
cd $R && python tools/prof_summary.py gpurun_out/pp gpurun_out/r02_post832_kernel_stats | head -14; rm -rf gpurun_out/pp
