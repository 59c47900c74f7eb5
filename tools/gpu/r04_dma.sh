python -m pytest tests/test_kernels.py -q -m gpu -x -k "wgrad" 2>&1 | tail -1
for i in 1 2; do
WG_FLAGS=0 python tools/wgbench.py regs 2>&1 | grep -v amdgpu.ids
WG_FLAGS=4096 python tools/wgbench.py dma 2>&1 | grep -v amdgpu.ids
done
