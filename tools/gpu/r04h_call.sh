cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do
KB=fwd,dgrad timeout 300 python tools/kbench.py new 2>&1 | grep -v amdgpu.ids
for bits in 1 2 4 32 3 7 39; do
KB=fwd,dgrad KB_LIB=tools/ab/libabl$bits.so timeout 300 python tools/kbench.py abl$bits 2>&1 | grep -v amdgpu.ids
done
done | tee gpurun_out/r04h_abl.log
