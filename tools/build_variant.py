"""TOOLS ONLY: build a variant of the kernel library with extra -D defines into abl_tmp/ (git-ignored, but shipped to the GPU box
with the snapshot), for same-box A/Bs through `bench.py --lib abl_tmp/<name>.so`.  The product library never carries these switches'
non-default values.   usage: python tools/build_variant.py <name> <source.hip> -DLU_G3_LA=2 [...]"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
from lu_native import build as b      # noqa: E402


def main(name, source, *defines):
    out_dir = os.path.join(ROOT, 'abl_tmp')
    os.makedirs(out_dir, exist_ok=True)
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    objs = []
    for src in b.SOURCES:
        var = src == source
        obj = os.path.join(out_dir, ('%s_' % name if var else 'base_') + src.replace('.hip', '.o'))
        path = os.path.join(b.CSRC, src)
        if var or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), os.path.getmtime(os.path.join(b.CSRC, 'lu_device.h'))):
            subprocess.check_call([hipcc] + b.FLAGS + (list(defines) if var else []) + ['-c', '-x', 'hip', path, '-o', obj])
        objs.append(obj)
    out = os.path.join(out_dir, 'liblstmunet_%s.so' % name)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out])
    print(out)


if __name__ == '__main__':
    main(*sys.argv[1:])
