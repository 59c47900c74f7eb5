# Negative result: the two blocks resident on a CU at different issue priorities.  The run-time knob (ConvArgs.dbg bits
# 64 / 128 / 192 -> s_setprio by HW_REG_HW_ID's workgroup-slot / wave-slot bits) lived in conv_halo_kernel for this measurement only.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do
for fl in 0 0x400000 0x800000 0xC00000; do
KB=fwd,dgrad KB_CONV_FLAGS=$fl timeout 300 python tools/kbench.py prio$fl 2>&1 | grep -v amdgpu.ids
done
done | tee gpurun_out/r04k_prio.log
