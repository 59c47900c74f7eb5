# (HISTORICAL: ran on binary 345e144325b810a5, when ops.X3_PIECES still read LU_X3_PIECES from the environment; the switch is a module attribute now)
# round 6, fourth GPU call: the bf16x3 step with the piece-aware weight gradient against the terms-as-frames form (LU_X3_PIECES=0), same box, alternating;
# config-4 and the 512x512 shape in the mode; then the whole GPU suite (oracle farm + multi-rank jobs started in front of the first test) with durations
tag=${1:-r06d}
R=$GRAFT_REPO_ROOT
: > gpurun_out/${tag}_x3_step_ab.log
for rep in 1 2; do for pieces in 0 1; do
  LU_X3_PIECES=$pieces python bench.py --precision bf16x3 --steps 8 --warmup 3 --no-cpu-baseline --no-variants --no-infer 2> gpurun_out/${tag}_x3.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step bf16x3 pieces=$pieces', d['ms_per_step'], d['value'], d['build_id'], d['bf16x3_summary'], [(r['kernel'][:28], r['ms_per_step'], r['frac']) for r in d['roofline']['all_mfma_kernels'][:5]])" | tee -a gpurun_out/${tag}_x3_step_ab.log
done; done
tail -3 gpurun_out/${tag}_x3.err
for pieces in 0 1; do
  LU_X3_PIECES=$pieces python bench.py --precision bf16x3 --size 512 --batch 2 --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-infer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512x512 bf16x3 pieces=$pieces', d['ms_per_step'], d['value'], d['peak_hbm_gb'])" | tee -a gpurun_out/${tag}_x3_step_ab.log
  LU_X3_PIECES=$pieces python bench.py --precision bf16x3 --hw 832 992 --batch 2 --unroll 16 --steps 2 --warmup 2 --no-infer --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config-4 bf16x3 pieces=$pieces', d['ms_per_step'], d['value'], d['peak_hbm_gb'])" | tee -a gpurun_out/${tag}_x3_step_ab.log
done
timeout 2400 python -m pytest tests -q -m gpu -s --durations=25 > gpurun_out/${tag}_gpu_tests.log 2>&1; tail -45 gpurun_out/${tag}_gpu_tests.log | cut -c1-200
