# what bounds the bf16 weight-gradient loop?  compile-time ablations of lu_wgrad.hip built ON the GPU box (scratch, never shipped)
mkdir -p gpurun_out/abl
cd lstm-unet_amd/csrc
for f in lu_conv lu_pointwise lu_postprocess; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -x hip $f.hip -o ../../gpurun_out/abl/$f.o & done
for bits in 1 2 4 8 16 3 31; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLU_WG_ABL=$bits -c -x hip lu_wgrad.hip -o ../../gpurun_out/abl/wg$bits.o & done
wait
cd ../..
python tools/wgbench.py product 2>&1 | grep -v amdgpu.ids
for bits in 1 2 4 8 16 3 31; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gpurun_out/abl/lu_conv.o gpurun_out/abl/lu_pointwise.o gpurun_out/abl/lu_postprocess.o gpurun_out/abl/wg$bits.o -o gpurun_out/abl/libwg$bits.so
  KB_LIB=gpurun_out/abl/libwg$bits.so python tools/wgbench.py abl$bits 2>&1 | grep -v amdgpu.ids
done
rm -rf gpurun_out/abl
