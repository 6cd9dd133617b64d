"""Practical MFMA ceiling on this box: the vendor library's bf16 GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the im2col-equivalent
shapes of the hot convs and on a large square problem.  Not part of the product; a yardstick for DESIGN.md."""
import torch, time
def run(M, N, K, reps=10):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): c = a @ b
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"GEMM M={M} N={N} K={K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
run(262144, 384, 3456); run(262144, 192, 1728); run(65536, 576, 5184); run(16384, 768, 6912); run(8192, 8192, 8192); run(16384, 16384, 8192, reps=5)
