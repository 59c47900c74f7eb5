python -c "import torch; print(torch.cuda.Stream.priority_range())"
for i in 1 2 3; do
for fl in "" "--infer-hi-prio"; do
python bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 $fl', d['inference']['frames_per_s'], d['inference']['frames_per_s_with_postprocess'])"
done
done
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['inference']['frames_per_s'], d['inference']['frames_per_s_with_postprocess'])"
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --infer-hi-prio 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 hi', d['inference']['frames_per_s'], d['inference']['frames_per_s_with_postprocess'])"
