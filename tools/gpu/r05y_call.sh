# round 5: counter tables of precision 'bf16x3' for the 512x512 shape and config-4 (so that those bench lines carry traffic / clock / MFMA-busy),
# then the two lines again
cd $GRAFT_REPO_ROOT
PMC_EXTRA="--size 512 --batch 2" PMC_SFX=_c5shape PMC_MODES=bf16x3 bash tools/gpu/pmc_traffic.sh r05y 2>&1 | tail -5
PMC_EXTRA="--hw 832 992 --batch 2 --unroll 16" PMC_SFX=_c4 PMC_MODES=bf16x3 bash tools/gpu/pmc_traffic.sh r05y 2>&1 | tail -5
python - <<PY
import json
for sfx in ('_c5shape', '_c4'):
    d=json.load(open('gpurun_out/r05y_pmc_traffic_bf16x3%s.json' % sfx)); d['collected']='round 5, final engine, binary %s, r05y' % d.get('build_id')
    json.dump(d, open('gpurun_out/r05y_pmc_traffic_bf16x3%s.json' % sfx,'w'), indent=1); json.dump(d, open('profiles/r05_pmc_traffic_bf16x3%s.json' % sfx,'w'), indent=1)
    u=json.load(open('profiles/r05_pmc_mfma_util%s.json' % sfx)); n=json.load(open('gpurun_out/r05y_pmc_mfma_util%s.json' % sfx))
    assert u['build_id'] == n['build_id']
    u['bf16x3']=n['bf16x3']; json.dump(u, open('profiles/r05_pmc_mfma_util%s.json' % sfx,'w'), indent=1); json.dump(u, open('gpurun_out/r05y_pmc_mfma_util%s_merged.json' % sfx,'w'), indent=1)
PY
python bench.py --precision bf16x3 --size 512 --batch 2 --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-infer > gpurun_out/r05y_x3_c5shape_bench_line.json 2>/dev/null
python bench.py --precision bf16x3 --hw 832 992 --batch 2 --unroll 16 --steps 3 --warmup 2 --no-infer --no-cpu-baseline --no-variants > gpurun_out/r05y_x3_c4_bench_line.json 2>/dev/null
python - <<PY
import json
for n in ('x3_c5shape','x3_c4'):
    d=json.load(open('gpurun_out/r05y_%s_bench_line.json' % n)); r=d['roofline']
    print(n, d['value'], d['ms_per_step'], d['peak_hbm_gb'], r['traffic_stale'], r['clock_stale'], [(c['kernel'][:28], c['frac'], c.get('traffic'), c.get('clock_mhz'), c.get('mfma_busy')) for c in r['all_mfma_kernels'][:4]])
PY
