"""What does the VENDOR library reach on this box under the same socket power limit?  torch.matmul in bf16 (hipBLASLt / rocBLAS) on
large GEMMs, incl. one with the weight gradient's aspect (short M, N, very deep K), timed with HIP events -- a practical reference point
next to the nominal 2.5 PFLOP/s for the hand-written bf16 kernels (tools only; nothing of the product calls a library GEMM)."""
import sys
import torch
dev = torch.device('cuda', 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for (M, N, K) in [(8192, 8192, 8192), (16384, 16384, 4096), (640, 512, 2097152), (1280, 1024, 524288), (4096, 4096, 65536)]:
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(K, N, device=dev) * 0.5).to(torch.bfloat16)
    for lay in ('nn', 'tn'):
        aa = a if lay == 'nn' else a.t().contiguous().t()      # 'tn': A stored k-major (pixel-major, as the weight gradient's operands are)
        torch.matmul(aa, b); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.matmul(aa, b)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('gemm_ref bf16 %s M=%d N=%d K=%d  %8.3f ms  %7.1f TFLOP/s = %.3f of 2.5 PF' % (lay, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 2500), flush=True)
    del a, b
