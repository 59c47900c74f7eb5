# round 5: the counter tables of precision 'bf16x3' again with the final engine (Conv2D units on split operands: new 3x3 launches), and the driver's
# command once more at its own step counts on the final tree
cd $GRAFT_REPO_ROOT
PMC_MODES=bf16x3 bash tools/gpu/pmc_traffic.sh r05r 2>&1 | tail -8
python - <<PY
import json
d=json.load(open('gpurun_out/r05r_pmc_traffic_bf16x3.json')); d['collected']='round 5, final engine (Conv2D units on split operands), binary %s, r05r' % d.get('build_id')
json.dump(d, open('gpurun_out/r05r_pmc_traffic_bf16x3.json','w'), indent=1)
json.dump(d, open('profiles/r05_pmc_traffic_bf16x3.json','w'), indent=1)
u=json.load(open('profiles/r05_pmc_mfma_util.json')); n=json.load(open('gpurun_out/r05r_pmc_mfma_util.json'))
assert u['build_id'] == n['build_id'], (u['build_id'], n['build_id'])
u['bf16x3']=n['bf16x3']; json.dump(u, open('profiles/r05_pmc_mfma_util.json','w'), indent=1); json.dump(u, open('gpurun_out/r05r_pmc_mfma_util_merged.json','w'), indent=1)
PY
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05r_driver_like_bench_line.json 2> gpurun_out/r05r_driver_like.err ) 2>&1 | tail -3
python bench.py --precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-variants > gpurun_out/r05r_x3_20steps_bench_line.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/r05r_driver_like_bench_line.json'))
print('driver-like:', d['value'], d['ms_per_step'], d['steps'], 'x3 mode', d['bf16x3_mode']['ms_per_step'], d['bf16x3_mode']['frames_per_s'], 'bf16', d['bf16_mode']['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['traffic_stale'], d['roofline']['clock_stale'])
x=json.load(open('gpurun_out/r05r_x3_20steps_bench_line.json'))
print('x3 20 steps:', x['value'], x['ms_per_step'], x['inference'] and x['inference']['frames_per_s'], [(c['kernel'][:30], c['frac'], c.get('traffic'), c.get('clock_mhz'), c.get('mfma_busy')) for c in x['roofline']['all_mfma_kernels'][:6]], x['roofline']['traffic_stale'])
PY
