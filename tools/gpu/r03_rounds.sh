for i in 1 2; do
for r in 5 3 4 6 8; do
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-wgrad-overlap --wgrad-rounds $r 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rounds $r', d['ms_per_step'], [(c['kernel'][:24], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'] if 'wgrad_row_bf16' in c['kernel']][:2], [(c['kernel'][:20], c['ms_per_step']) for c in d['roofline'].get('hbm_kernels',[]) if 'reduce' in c['kernel']])"
done
done
