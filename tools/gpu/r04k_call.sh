cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do
for fl in 0 0x400000 0x800000 0xC00000; do
KB=fwd,dgrad KB_CONV_FLAGS=$fl timeout 300 python tools/kbench.py prio$fl 2>&1 | grep -v amdgpu.ids
done
done | tee gpurun_out/r04k_prio.log
