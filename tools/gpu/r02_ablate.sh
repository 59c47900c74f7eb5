# what bounds the bf16 fragment kernel?  compile-time ablation builds: python -m lu_native.build --ablation 0 1 2 4 8 3 7 15
export KB=tape16
for lvl in L1 L0 L3; do
  export KB_LEVEL=$lvl
  KB_LIB= python tools/kbench.py product 2>&1 | grep -v amdgpu.ids
  for bits in 0 1 2 4 8 3 7 15; do
    KB_LIB=lstm-unet_amd/csrc/liblstmunet_abl$bits.so python tools/kbench.py abl$bits 2>&1 | grep -v amdgpu.ids
  done
done
