R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
rm -rf $R/gpurun_out/pmcf
rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/pmcf -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-infer --no-bf16 > /dev/null 2>&1
python $R/tools/pmc_dispatch.py $R/gpurun_out/pmcf $ctr "wgrad_row_kernel<5>" | grep "^{" | python -c "
import sys
for l in list(sys.stdin)[-7:]:
    d = eval(l); print('$ctr', d['dispatch_id'], '%.0f MB' % (d['value'] * (2 if '$ctr' == 'FETCH_SIZE' else 1) * 1024 / 1e6))
"
rm -rf $R/gpurun_out/pmcf
done
cd $R && python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-infer --no-bf16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel'][:30], r['achieved'], r['avg_launch_ms'])"
