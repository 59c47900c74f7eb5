"""TOOLS ONLY: the kernel library with tools/probe/mfma_f16_probe.h force-included (every bf16 MFMA becomes the fp16 MFMA on the same
bits: a timing probe, results are garbage) -> abl_tmp/liblstmunet_f16probe.so, for `bench.py --precision bf16 --lib ...` next to the
product library on the same box.  The sources are untouched, so the product's build id does not move."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
from lu_native import build as b      # noqa: E402


def main():
    out_dir = os.path.join(ROOT, 'abl_tmp')
    os.makedirs(out_dir, exist_ok=True)
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    probe = os.path.join(ROOT, 'tools', 'probe', 'mfma_f16_probe.h')

    def one(src):
        obj = os.path.join(out_dir, 'f16probe_' + src.replace('.hip', '.o'))
        subprocess.check_call([hipcc] + b.FLAGS + ['-include', probe, '-c', '-x', 'hip', os.path.join(b.CSRC, src), '-o', obj])
        return obj
    with ThreadPoolExecutor(4) as pool:
        objs = list(pool.map(one, b.SOURCES))
    out = os.path.join(out_dir, 'liblstmunet_f16probe.so')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out])
    print(out)


if __name__ == '__main__':
    main()
