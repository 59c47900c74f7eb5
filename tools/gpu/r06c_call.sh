# (HISTORICAL: see r06d_call.sh)
# round 6, third GPU call: the piece-aware 'bf16x3' weight gradient (wgrad_row_x3_kernel) against the terms-as-frames form -- micro-benchmark, the
# kernel tests, the bf16x3 step both ways --, the forced one-rank RCCL bench line (with a fault handler: the first attempt printed no line), then the
# whole GPU suite with per-test durations
tag=${1:-r06c}
R=$GRAFT_REPO_ROOT
python tools/wgbench_x3.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_wgbench_x3.log
timeout 900 python -m pytest tests/test_kernels.py -q -m gpu -s -k "split6 or window_copy or wgrad_bf16" 2>&1 | tail -25 | tee gpurun_out/${tag}_kernel_tests.log
for rep in 1 2; do for pieces in 0 1; do
  LU_X3_PIECES=$pieces python bench.py --precision bf16x3 --steps 8 --warmup 3 --no-cpu-baseline --no-variants --no-infer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step bf16x3 pieces=$pieces', d['ms_per_step'], d['value'], d['build_id'], d['bf16x3_summary'], [(r['kernel'][:28], r['ms_per_step'], r['frac']) for r in d['roofline']['all_mfma_kernels'][:5]])" | tee -a gpurun_out/${tag}_x3_step_ab.log
done; done
python -X faulthandler bench.py --force-collectives --sync-bn --steps 5 --warmup 2 --no-variants --no-cpu-baseline --no-infer --no-bf16 --no-x3 > gpurun_out/${tag}_forced_rccl_bench_line.json 2> gpurun_out/${tag}_forced_rccl.err; echo "forced rccl bench rc=$?"; grep -v amdgpu.ids gpurun_out/${tag}_forced_rccl.err | tail -25
timeout 2400 python -m pytest tests -q -m gpu -s --durations=40 > gpurun_out/${tag}_gpu_tests.log 2>&1; tail -60 gpurun_out/${tag}_gpu_tests.log | cut -c1-220
