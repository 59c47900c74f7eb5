"""Build the gfx950 kernel library in-tree with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'csrc')
SOURCES = ['lu_conv.hip', 'lu_wgrad.hip', 'lu_pointwise.hip', 'lu_postprocess.hip']
LIB = os.path.join(CSRC, 'liblstmunet_hip.so')


FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']
STAMP = LIB + '.srchash'      # sha256 of (flags + every source byte) the library was built from; ships beside the .so


def source_hash():
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'lu_device.h'),
            os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'lstm_unet_hip.h')]
    for d in deps:
        with open(d, 'rb') as fh:
            h.update(b'\0' + os.path.basename(d).encode() + b'\0' + fh.read())
    return h.hexdigest()


def build_id():
    """sha256 of the library file itself (first 16 hex digits): stamped on bench lines and on every counter table under
    profiles/ so that a table collected from another binary is recognisable as stale."""
    import hashlib
    with open(LIB, 'rb') as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def needs_build():
    """Content-addressed, not mtime-based: the library is reused only if it was built from exactly these source bytes and
    flags (a checkout, a copy to the GPU box or a touched file do not matter; an edited kernel always does)."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    digest = source_hash()
    # objects never land in the tree (nothing extra ships).  The directory NAME is a function of the sources: hipcc leaves the
    # object paths in the library, and a random name made two builds of the same sources differ in their build id
    tmp = os.path.join(tempfile.gettempdir(), 'lu_build_' + digest[:16])
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    try:
        def compile_one(src):
            obj = os.path.join(tmp, src.replace('.hip', '.o'))
            cmd = [hipcc] + FLAGS + ['-c', '-x', 'hip', os.path.join(CSRC, src), '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj
        with ThreadPoolExecutor(len(SOURCES)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB + '.tmp']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    os.replace(LIB + '.tmp', LIB)
    with open(STAMP, 'w') as fh:
        fh.write(digest + '\n')
    return LIB


def build_ablation(bits, verbose=True):
    """TOOLS ONLY (tools/kbench.py): lu_conv.hip with -DLU_ABLATION=<bits>, which compiles parts of the fragment kernel's loop
    out (1 weight loads, 2 LDS fragment reads, 4 halo prefetch, 8 epilogue, 16 chunk barrier) to see what bounds it.
    Built under gpurun_out/abl/ (scratch, never shipped with a snapshot): the product library never contains these switches."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    out_dir = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'gpurun_out', 'abl')
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(out_dir, ('abl%d_' % bits if src == 'lu_conv.hip' else 'abl_') + src.replace('.hip', '.o'))
        if src == 'lu_conv.hip' or not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(os.path.join(CSRC, src)):
            cmd = [hipcc] + FLAGS + (['-DLU_ABLATION=%d' % bits] if src == 'lu_conv.hip' else []) + \
                  ['-c', '-x', 'hip', os.path.join(CSRC, src), '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    out = os.path.join(out_dir, 'liblstmunet_abl%d.so' % bits)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out])
    return out


if __name__ == '__main__':
    if '--ablation' in sys.argv:
        for b_ in sys.argv[sys.argv.index('--ablation') + 1:]:
            print(build_ablation(int(b_)))
    else:
        build(force='--force' in sys.argv)
