# same-box A/B of the batched weight preparation + lazy recurrent state (alternating runs)
tag=${1:-r03v}
for i in 1 2; do
  python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer > gpurun_out/${tag}_bf16_new$i.json 2>> gpurun_out/${tag}.err
  python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --ab-no-prep > gpurun_out/${tag}_bf16_old$i.json 2>> gpurun_out/${tag}.err
done
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-infer --no-bf16 > gpurun_out/${tag}_f32_new.json 2>> gpurun_out/${tag}.err
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-infer --no-bf16 --ab-no-prep > gpurun_out/${tag}_f32_old.json 2>> gpurun_out/${tag}.err
python - <<'PY'
import json,glob,sys
tag=sys.argv[1] if len(sys.argv)>1 else 'r03v'
for f in sorted(glob.glob('gpurun_out/%s_*.json' % tag)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'])
    except Exception as ex: print(f, 'ERR', ex)
PY
