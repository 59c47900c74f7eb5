"""TEST INFRASTRUCTURE: compile the kernel sources for the host SIMT emulator (clang++ -DLU_EMU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'lstm-unet_amd', 'csrc')
OUT = os.path.join(HERE, '_build')
LIB = os.path.join(OUT, 'liblstmunet_emu.so')
SOURCES = ['lu_conv.hip', 'lu_wgrad.hip', 'lu_pointwise.hip', 'lu_postprocess.hip']


def build():
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'lu_device.h'),
            os.path.join(HERE, 'emu_runtime.h'), os.path.join(ROOT, 'include', 'lstm_unet_hip.h')]
    if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    cxx = '/opt/rocm/lib/llvm/bin/clang++'
    if not os.path.exists(cxx):
        cxx = shutil.which('clang++')
    if cxx is None:
        raise RuntimeError('clang++ not found (needed for ext_vector_type)')
    cmd = [cxx, '-DLU_EMU', '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'c++', '-I', HERE,
           '-Wno-unused-value', '-U_FORTIFY_SOURCE', '-D_FORTIFY_SOURCE=0'] + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB]
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build())
