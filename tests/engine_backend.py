"""TEST INFRASTRUCTURE: run the product's host code (lu_native.engine / Networks / losses / train2D)
either on the real HIP library ('hip', needs a GPU) or -- to debug host-side logic in the GPU-less
build container -- against the host-emulated kernel build ('emu') by monkeypatching the three
device hooks of lu_native.ops.  The product itself has no such switch: outside this fixture
ops.lib() only ever loads the gfx950 .so and rejects non-device tensors."""
import contextlib

import torch


@contextlib.contextmanager
def engine_backend(name):
    import Networks
    from lu_native import ops
    import kernel_harness as KH
    if name == 'hip':
        yield torch.device('cuda', 0)
        return
    emu = KH.backend('emu')
    saved = (ops.lib, ops._stream, ops._chk, Networks._device)
    ops.lib = lambda: emu.lib
    ops._stream = lambda: None
    ops._chk = lambda *a: None
    Networks._device = lambda: torch.device('cpu')
    try:
        yield torch.device('cpu')
    finally:
        ops.lib, ops._stream, ops._chk, Networks._device = saved
