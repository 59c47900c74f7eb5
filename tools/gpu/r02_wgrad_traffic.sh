R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcw
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmcw -- python $R/tools/wgrad_traffic.py > $R/gpurun_out/wgrad_traffic.log 2>&1
grep "splits\|Error\|error" $R/gpurun_out/wgrad_traffic.log | head -20
python $R/tools/pmc_dispatch.py $R/gpurun_out/pmcw FETCH_SIZE "wgrad_row_bf16_kernel<5" | grep "^{" | python -c "
import sys
for l in sys.stdin:
    d = eval(l)
    print(d['dispatch_id'], d['grid_size'] // 512, 'fetch %.0f MB' % (d['value'] * 2 * 1024 / 1e6))
"
rm -rf $R/gpurun_out/pmcw
