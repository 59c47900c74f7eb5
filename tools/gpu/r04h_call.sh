# What bounds the fp32 halo loop?  Compile-time ablations of conv_halo_kernel.  Build the variants HERE first (scratch, never shipped):
#   for bits in 1 2 4 32 3 7 39; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLU_ABLATION=$bits -c -x hip lstm-unet_amd/csrc/lu_conv.hip -o /tmp/abl_conv_$bits.o; done
#   ... link each with the other three objects of the product build into tools/ab/libabl$bits.so (git-ignored; they travel with gpurun)
# bits: 1 no global loads / transfers in the loop, 2 no halo stores, 4 no stage barrier, 32 no LDS operand reads
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do
KB=fwd,dgrad timeout 300 python tools/kbench.py new 2>&1 | grep -v amdgpu.ids
for bits in 1 2 4 32 3 7 39; do
KB=fwd,dgrad KB_LIB=tools/ab/libabl$bits.so timeout 300 python tools/kbench.py abl$bits 2>&1 | grep -v amdgpu.ids
done
done | tee gpurun_out/r04h_abl.log
