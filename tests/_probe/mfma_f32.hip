#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(const float* A, const float* B, float* D, int K) {   // A[16][K], B[K][16]
  int l = threadIdx.x; int r = l & 15, g = l >> 4;
  f32x4 acc = {0,0,0,0};
  for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r*K + k + g], B[(k+g)*16 + r], acc, 0,0,0);
  for (int i = 0; i < 4; ++i) D[(g*4+i)*16 + r] = acc[i];
}
int main() {
  for (int K : {64, 576, 1728, 13824}) {
    std::mt19937 rng(1); std::normal_distribution<float> nd;
    std::vector<float> A(16*K), B(K*16), D(256);
    for (auto& v : A) v = nd(rng); for (auto& v : B) v = nd(rng);
    float *dA, *dB, *dD; hipMalloc(&dA, A.size()*4); hipMalloc(&dB, B.size()*4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size()*4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size()*4, hipMemcpyHostToDevice);
    k_mfma<<<1,64>>>(dA, dB, dD, K); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double e2=0, r2=0, c2=0; 
    for (int i=0;i<16;++i) for (int j=0;j<16;++j) { double ref=0; float ch=0; for (int k=0;k<K;++k){ ref += (double)A[i*K+k]*B[k*16+j]; ch = fmaf(A[i*K+k],B[k*16+j],ch);} 
       e2 += (D[i*16+j]-ref)*(D[i*16+j]-ref); r2 += ref*ref; c2 += (ch-ref)*(ch-ref);} 
    printf("K=%5d mfma rel-rms %.3e   fp32 fma-chain rel-rms %.3e\n", K, sqrt(e2/r2), sqrt(c2/r2));
  }
}
