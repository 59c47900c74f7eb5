cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "conv or wgrad or convlstm" 2>&1 | tail -4 | tee gpurun_out/r04l_ktests.log
for i in 1 2; do
KB=wgrad KB_LIB=tools/ab/liblstmunet_old.so timeout 300 python tools/kbench.py old 2>&1 | grep -v amdgpu.ids
KB=wgrad timeout 300 python tools/kbench.py new 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04l_kbench.log
bash tools/gpu/r04_ab_lib.sh r04l_ab fp32 2
