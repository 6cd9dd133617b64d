// Per-CU memory-path microbenchmark (round 3): how many bytes per clock can ONE CU move to / from HBM, as a function of the access pattern
// and of how many CUs are doing it?  Decides whether the conv epilogue (stores of 16 B per lane, 32 pixels x 32 B per instruction) is limited
// by the CU's own memory path or by the chip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mem_path_bench.hip -o tools/mem_path_bench.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// mode 0: store, conv-epilogue pattern: lane l -> row (l & 31) of a 32-row block, 16-byte half (l >> 5) of a 32-byte run, row pitch `pitch` bytes
// mode 1: store, full lines: 8 consecutive lanes cover one 128-byte run of a row (8 rows per instruction)
// mode 2: load, epilogue pattern      mode 3: load, full lines
// each workgroup (256 threads, 4 waves) owns `bytes_per_wg` of distinct memory; grid = nwg
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned char* base, size_t bytes_per_wg, int pitch, int iters, unsigned* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* wg = base + (size_t)blockIdx.x * bytes_per_wg;
    // a wave-instruction covers 1 KiB: epilogue pattern = 32 rows x 32 B at column offset c; full lines = 8 rows x 128 B
    const int rows_total = (int)(bytes_per_wg / pitch);
    u32x4 acc = {0, 0, 0, 0};
    const u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    for (int it = 0; it < iters; ++it) {
        // walk: row blocks, and inside a row the column offset advances so that every byte of the wg's region is touched once per sweep
        if (MODE == 0 || MODE == 2) {
            const int per_row = pitch / 32;                       // instructions to cover a 32-row block
            const int nblk = rows_total / 32;
            for (int b = wave; b < nblk; b += 4)
                for (int c = 0; c < per_row; ++c) {
                    unsigned char* a = wg + (size_t)(b * 32 + (lane & 31)) * pitch + c * 32 + (lane >> 5) * 16;
                    if (MODE == 0) *(u32x4*)a = v; else { u32x4 t = *(const u32x4*)a; acc += t; }
                }
        } else {
            const int per_row = pitch / 128;
            const int nblk = rows_total / 8;
            for (int b = wave; b < nblk; b += 4)
                for (int c = 0; c < per_row; ++c) {
                    unsigned char* a = wg + (size_t)(b * 8 + (lane >> 3)) * pitch + c * 128 + (lane & 7) * 16;
                    if (MODE == 1) *(u32x4*)a = v; else { u32x4 t = *(const u32x4*)a; acc += t; }
                }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) sink[0] = acc.x;
}

int main(int argc, char** argv) {
    const int pitch = 384;                 // 192 bf16 channels per pixel
    const size_t per_wg = (size_t)384 * 4096;   // 1.5 MiB per workgroup (4096 pixels): far beyond L2 share when many workgroups run
    const int maxwg = 2048;
    unsigned char* buf; unsigned* sink;
    CK(hipMalloc(&buf, per_wg * maxwg)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, per_wg * maxwg));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"store 32x32B (conv epilogue pattern)", "store full 128-B lines", "load  32x32B (residual pattern)", "load  full 128-B lines"};
    for (int mode = 0; mode < 4; ++mode)
        for (int nwg : {32, 64, 128, 256, 512, 1024, 2048}) {
            auto launch = [&]() {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 0, 0, buf, per_wg, pitch, 1, sink); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 0, 0, buf, per_wg, pitch, 1, sink); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), 0, 0, buf, per_wg, pitch, 1, sink); break;
                    default: hipLaunchKernelGGL(k<3>, dim3(nwg), dim3(256), 0, 0, buf, per_wg, pitch, 1, sink); break;
                }
            };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            const int reps = 5;
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            const double bytes = (double)per_wg * nwg;
            const int cus = nwg < 256 ? nwg : 256;
            printf("%-40s wgs %4d: %8.1f us  %7.1f GB/s total  %6.2f GB/s per busy CU  (~%.1f B/clk/CU at 2.0 GHz)\n", names[mode], nwg, ms * 1e3, bytes / ms / 1e6,
                   bytes / ms / 1e6 / cus, bytes / ms / 1e6 / cus / 2.0);
        }
    return 0;
}
