# per-dispatch FETCH_SIZE / WRITE_SIZE of the bf16 5x5 weight gradients (usage: bash tools/gpu/r02_pmc_wgrad.sh <tag>)
tag=${1:-pw}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcw_$ctr
  rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/pmcw_$ctr -- python $R/bench.py --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-infer --no-bf16 > /dev/null 2>&1
  python $R/tools/pmc_dispatch.py $R/gpurun_out/pmcw_$ctr $ctr "wgrad_row_bf16_kernel<5" $R/gpurun_out/${tag}_wgrad5_$ctr.json | tail -32
  python $R/tools/pmc_dispatch.py $R/gpurun_out/pmcw_$ctr $ctr "conv_halo_frag2_kernel<5, 1, 8" $R/gpurun_out/${tag}_fused_$ctr.json | tail -12
  rm -rf $R/gpurun_out/pmcw_$ctr
done
