"""Build the gfx950 kernel library in-tree with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'csrc')
SOURCES = ['lu_conv.hip', 'lu_wgrad.hip', 'lu_pointwise.hip', 'lu_postprocess.hip']
LIB = os.path.join(CSRC, 'liblstmunet_hip.so')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'lu_device.h'),
            os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'lstm_unet_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-x', 'hip'] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
