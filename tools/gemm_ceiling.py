"""Practical MFMA ceiling on this box: the vendor library's bf16 GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the im2col-equivalent
shapes of the hot convs (an explicit-GEMM conv would ALSO have to write and read the 9x larger im2col matrix, which is not counted here)
and on large square problems.  Not part of the product; a yardstick for DESIGN.md."""
import torch, time
def run(M, N, K, reps=10, tag=""):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): c = a @ b
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"GEMM M={M} N={N} K={K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s  {tag}")
run(262144, 192, 1728, tag="enc 64x64 192->192 3x3 (k3)")
run(262144, 192, 3456, tag="dec 64x64 384->192 3x3")
run(262144, 384, 3456, tag="64x64 384->384 3x3")
run(65536, 384, 3456, tag="32x32 384->384 3x3")
run(65536, 384, 6912, tag="32x32 768->384 3x3")
run(16384, 576, 5184, tag="16x16 576->576 3x3")
run(16384, 576, 12096, tag="16x16 1344->576 3x3")
run(4096, 768, 6912, tag="8x8 768->768 3x3")
run(4096, 768, 13824, tag="8x8 1536->768 3x3")
run(8192, 8192, 8192); run(16384, 16384, 8192, reps=5)
