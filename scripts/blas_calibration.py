"""Calibration only (not a product path): what the vendor GEMM (torch.mm -> hipBLASLt/rocBLAS) achieves on the bf16 shapes of the
hot path, to know the practical speed of light for plain (no prologue / epilogue) GEMMs of these skinny shapes."""
import torch
shapes = [('G1 s0', 54400, 648, 128), ('G2 s0', 54400, 128, 256), ('G3 s0', 54400, 128, 128), ('G4 s0', 54400, 256, 384),
          ('conv1', 41344, 256, 768), ('G1 s1', 41344, 1288, 256), ('G4 s1', 41344, 512, 768), ('G2 s1', 41344, 256, 512),
          ('G3 s1', 41344, 256, 256), ('conv2', 2176, 512, 1536), ('G4 s2', 2176, 1024, 1536),
          ('wgrad G4 s1 (N=512,K=768 over M)', 512, 768, 41344), ('wgrad G1 s1', 1288, 256, 41344), ('wgrad G4 s0', 256, 384, 54400)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    W = torch.randn(N, K, device='cuda', dtype=torch.bfloat16)
    for _ in range(3):
        C = A @ W.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        C = A @ W.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print('%-36s M=%6d N=%5d K=%6d  %7.1f us  %7.1f TF/s  %5.2f TB/s' % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6, (M * K + N * K + M * N) * 2 / us / 1e6))
