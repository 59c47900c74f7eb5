# what bounds the bf16 fragment kernel?  (ablation build: python -m lu_native.build --ablation)
export KB=tape16
for lvl in L1 L0; do
  export KB_LEVEL=$lvl
  KB_LIB= python tools/kbench.py product 2>&1 | grep -v amdgpu.ids
  for dbg in 0 1 2 4 8 16 3 6 7 15 31; do
    KB_LIB=lstm-unet_amd/csrc/liblstmunet_abl.so KB_DBG=$dbg python tools/kbench.py abl$dbg 2>&1 | grep -v amdgpu.ids
  done
done
