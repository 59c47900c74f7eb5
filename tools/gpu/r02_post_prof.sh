R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pp -- python $R/tools/post_prof.py 2>&1 | grep "postprocess alone"
cd $R && python tools/prof_summary.py gpurun_out/pp gpurun_out/r02_post_kernel_stats | head -24; python tools/prof_overlap.py gpurun_out/pp 0.8; rm -rf gpurun_out/pp
python tools/post_prof.py 2>&1 | grep "postprocess alone"
