for i in 1 2; do
KB_SHAPES=0,1,2,3 python tools/k3bench.py th8 2>&1 | grep -v amdgpu.ids | grep "src=bf16"
KB_SHAPES=0,1,2,3 KB_FLAGS=2 python tools/k3bench.py th16 2>&1 | grep -v amdgpu.ids | grep "src=bf16"
done
python -m pytest tests/test_kernels.py -m gpu -x -q -k "bf16" 2>&1 | tail -2
