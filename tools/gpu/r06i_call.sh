# round 6, last evidence call on the final library: precision 'bf16x3' with the piece-aware weight gradient -- 300 training steps against the fp32 engine on the
# same stream, config-4 and the 512 x 512 shape with their counter tables (FETCH_SIZE / WRITE_SIZE / MFMA-busy, each in its own rocprofv3 --pmc pass)
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
python tools/train_compare.py 300 bf16x3 > gpurun_out/${tag}_bf16x3_vs_fp32_training.json 2> gpurun_out/${tag}_train_compare_x3.err; tail -c 500 gpurun_out/${tag}_bf16x3_vs_fp32_training.json
PMC_EXTRA="--size 512 --batch 2" PMC_SFX=_c5shape PMC_MODES=bf16x3 bash tools/gpu/pmc_traffic.sh $tag | tail -4
PMC_EXTRA="--hw 832 992 --batch 2 --unroll 16" PMC_SFX=_c4 PMC_MODES=bf16x3 bash tools/gpu/pmc_traffic.sh $tag | tail -4
for f in gpurun_out/${tag}_pmc_traffic_bf16x3_c5shape.json gpurun_out/${tag}_pmc_traffic_bf16x3_c4.json; do [ -f $f ] && cp $f profiles/; done
python - <<PY
import json
for sfx in ('_c5shape', '_c4'):      # the MFMA-busy tables of this call hold only the bf16x3 block: merge it into the round's table of that workload
    try:
        new = json.load(open('gpurun_out/${tag}_pmc_mfma_util%s.json' % sfx)); old = json.load(open('profiles/r06_pmc_mfma_util%s.json' % sfx))
        old['bf16x3'] = new.get('bf16x3', {}); json.dump(old, open('profiles/r06_pmc_mfma_util%s.json' % sfx, 'w'), indent=1); json.dump(old, open('gpurun_out/${tag}_pmc_mfma_util%s_merged.json' % sfx, 'w'), indent=1)
    except Exception as e: print('merge', sfx, e)
PY
python bench.py --precision bf16x3 --size 512 --batch 2 --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-infer > gpurun_out/${tag}_x3_c5shape_bench_line.json 2> gpurun_out/${tag}_x3_extra.err
python bench.py --precision bf16x3 --hw 832 992 --batch 2 --unroll 16 --steps 3 --warmup 2 --no-infer --no-cpu-baseline --no-variants > gpurun_out/${tag}_x3_c4_bench_line.json 2>> gpurun_out/${tag}_x3_extra.err
python - <<PY
import json
for n in ('x3_c5shape_bench_line', 'x3_c4_bench_line'):
    d = [json.loads(l) for l in open('gpurun_out/${tag}_%s.json' % n).read().splitlines() if l.startswith('{')][-1]
    r = d['roofline']
    print(n, d['value'], d['ms_per_step'], d['peak_hbm_gb'], d['build_id'], r.get('traffic_stale'), r.get('clock_stale'), d['bf16x3_summary'], [(k['kernel'][:24], k['ms_per_step'], k['frac'], k.get('traffic'), k.get('clock_mhz')) for k in r['all_mfma_kernels'][:3]])
PY
