for i in 1 2; do
for lib in "" abl_tmp/lib_prio.so; do
KB_LIB=$lib python tools/kbench.py "${lib:-product}" 2>&1 | grep -v amdgpu.ids | grep "L0\|L1"
KB=tape16 KB_LIB=$lib python tools/kbench.py "${lib:-product}" 2>&1 | grep -v amdgpu.ids | grep "L0\|L1"
done
done
