export KB_SHAPES=0,1,3
python tools/k3bench.py product 2>&1 | grep -v amdgpu.ids
for b in 32; do KB_LIB=abl_tmp/liblstmunet_abl$b.so python tools/k3bench.py abl$b 2>&1 | grep -v amdgpu.ids; done
