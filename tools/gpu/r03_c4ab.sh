python -m pytest tests/test_dp_equivalence.py -m gpu -x -q -s 2>&1 | grep "dp2 vs\|passed\|failed"
for v in "" "--ab-no-prep"; do
python bench.py --hw 832 992 --batch 2 --unroll 16 --steps 2 --warmup 2 --no-bf16 --no-infer --no-cpu-baseline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 $v', d['ms_per_step'], d['value'], d['peak_hbm_gb'])"
done
python bench.py --hw 832 992 --batch 2 --unroll 16 --steps 2 --warmup 1 --no-bf16 --no-infer --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 warmup1', d['ms_per_step'], d['value'], d['peak_hbm_gb'])"
