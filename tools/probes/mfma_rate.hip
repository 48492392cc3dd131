// MX-fp8 MFMA issue-rate probe (gfx950): 8 independent 32x32x64 accumulators per wave, random-ish operands, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int WAVES_PER_SIMD, int MODE>
__global__ void __launch_bounds__(256 * WAVES_PER_SIMD) probe(const int* __restrict__ in, float* __restrict__ out, int iters) {
  v8i fa[4], fw[2];
  const int t = threadIdx.x;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) fa[i][j] = in[(t * 61 + i * 8 + j) & 4095];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 8; ++j) fw[i][j] = in[(t * 37 + 64 + i * 8 + j) & 4095];
  v16f acc[4][2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (MODE == 0)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[j], fa[i], acc[i][j], 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        else
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[j], fa[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    if (MODE == 2) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * blockDim.x + t] = s;
}

template <int W, int MODE> void run(const char* name, const int* din, float* dout, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256;
  probe<W, MODE><<<blocks, 256 * W>>>(din, dout, 16);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    probe<W, MODE><<<blocks, 256 * W>>>(din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double mfma = (double)blocks * 4 * W * iters * 8;  // wave-level MFMAs
  const double flops = mfma * 2.0 * 32 * 32 * 64;
  printf("%-44s %7.1f TF/s   %6.1f ns per MFMA per SIMD\n", name, flops / (best * 1e-3) / 1e12, best * 1e6 / (iters * 8.0 * W));
}

int main() {
  int* h = (int*)malloc(4096 * 4);
  srand(1);
  int* din; float* dout;
  hipMalloc(&din, 4096 * 4); hipMalloc(&dout, 256 * 512 * 4);
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 4096; ++i) {
      unsigned v = 0;
      for (int b = 0; b < 4; ++b) v |= (pass == 0 ? ((unsigned)(rand() % 120 + 4 + (rand() & 1) * 128)) : 0u) << (8 * b);  // finite fp8 bytes / zeros
      h[i] = (int)v;
    }
    hipMemcpy(din, h, 4096 * 4, hipMemcpyHostToDevice);
    printf(pass == 0 ? "# random fp8 operands\n" : "# zero operands\n");
    run<1, 0>("1 wave/SIMD, e4m3 x e5m2", din, dout, 20000);
    run<2, 0>("2 waves/SIMD, e4m3 x e5m2", din, dout, 20000);
    run<2, 1>("2 waves/SIMD, e4m3 x e4m3", din, dout, 20000);
    run<2, 2>("2 waves/SIMD + s_barrier per 8 MFMAs", din, dout, 20000);
  }
  return 0;
}
