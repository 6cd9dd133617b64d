// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts in the access patterns of the conv kernels
// (MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths and
// WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel below touches each byte it counts
// exactly once, over a buffer far larger than the 256 MiB memory-side cache.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pmc_calib.hip -o tools/pmc_calib.out
//   rocprofv3 --pmc FETCH_SIZE -- tools/pmc_calib.out ; rocprofv3 --pmc WRITE_SIZE -- tools/pmc_calib.out      (tools/pmc_calib.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// R1: wide coalesced stream: a wave reads 1 KiB per instruction, consecutive
__global__ __launch_bounds__(256) void calib_read_stream(const u32x4* p, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc ^= p[i].x;
    if (acc == 0x12345678u) *sink = acc;
}
// R2: the conv's patch staging: 8 lanes read one 128-byte run (64 channels) of a pixel, pixels `pitch` bytes apart, ONE run per pixel (K-chunk `chunk`)
__global__ __launch_bounds__(256) void calib_read_patch(const unsigned char* p, size_t npix, int pitch, int chunk, unsigned* sink) {
    unsigned acc = 0;
    const int l8 = threadIdx.x & 7;
    for (size_t px = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); px < npix; px += (size_t)gridDim.x * 32) acc ^= ((const u32x4*)(p + px * pitch + chunk * 128 + l8 * 16))->x;
    if (acc == 0x12345678u) *sink = acc;
}
// W1: wide coalesced stream store
__global__ __launch_bounds__(256) void calib_write_stream(u32x4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = u32x4{1u, 2u, 3u, (unsigned)i};
}
// W2: the conv epilogue's store: lane l -> pixel row (l & 31) of a 32-pixel block, 16-byte half (l >> 5) of a 32-byte run (16 couts), pixel pitch
// `pitch` bytes; a wave covers the `runs` 32-byte runs of its 32 pixels one after the other (all couts of the pixel: the whole row is written)
__global__ __launch_bounds__(256) void calib_write_epilogue(unsigned char* p, size_t npix, int pitch, unsigned* dummy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int runs = pitch / 32;
    for (size_t blk = (size_t)blockIdx.x * 4 + wave; blk * 32 < npix; blk += (size_t)gridDim.x * 4) {
        unsigned char* row = p + (blk * 32 + (lane & 31)) * pitch + (lane >> 5) * 16;
        for (int r = 0; r < runs; ++r) *(u32x4*)(row + r * 32) = u32x4{1u, 2u, (unsigned)r, (unsigned)blk};
    }
}

int main() {
    const size_t bytes = (size_t)3 << 30;   // 3 GiB: 12x the memory-side cache
    unsigned char* buf; unsigned* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, bytes)); CK(hipDeviceSynchronize());
    const int pitch = 384;                  // 192 channels x 2 bytes
    const size_t npix = bytes / pitch;
    hipLaunchKernelGGL(calib_read_stream, dim3(4096), dim3(256), 0, 0, (const u32x4*)buf, bytes / 16, sink);
    for (int c = 0; c < 3; ++c) hipLaunchKernelGGL(calib_read_patch, dim3(4096), dim3(256), 0, 0, buf, npix, pitch, c, sink);
    hipLaunchKernelGGL(calib_write_stream, dim3(4096), dim3(256), 0, 0, (u32x4*)buf, bytes / 16);
    hipLaunchKernelGGL(calib_write_epilogue, dim3(4096), dim3(256), 0, 0, buf, npix, pitch, sink);
    CK(hipDeviceSynchronize());
    printf("known bytes: calib_read_stream %zu | calib_read_patch %zu per launch (one 128-byte run of every %d-byte pixel; x3 launches = every byte once) | calib_write_stream %zu | calib_write_epilogue %zu\n",
           bytes, npix * 128, pitch, bytes, npix * (size_t)pitch);
    return 0;
}
