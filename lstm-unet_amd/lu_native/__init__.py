"""Host-side plumbing of the MI355X ConvLSTM-UNet kernels.

torch MUST be imported before the kernel library is dlopen'ed: the PyTorch-ROCm wheel bundles its own
libamdhip64, and device pointers / streams are only meaningful inside ONE HIP runtime instance.  With
torch loaded first, liblstmunet_hip.so's DT_NEEDED libamdhip64 resolves to the already-loaded copy.
(Loading our .so first gives it a second, device-less runtime: "no ROCm-capable device is detected".)
"""
import torch  # noqa: F401  (see above -- keep this the first import)
