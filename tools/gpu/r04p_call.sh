cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
KB=wgrad KB_LIB=tools/ab/liblstmunet_old.so timeout 300 python tools/kbench.py old 2>&1 | grep -v amdgpu.ids
KB=wgrad timeout 300 python tools/kbench.py new 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04p_kbench.log
