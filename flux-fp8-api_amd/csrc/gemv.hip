// fluxmi -- batched skinny linear (M = batch <= 8): the 76 Modulation layers + LastLayer.adaLN +
// the MLPEmbedders of one denoise step in ONE launch (they all depend only on `vec`).
//
//   out[b, n] = bf16( (x_q[b,:] . W[n,:]) * sa*sb + bias[n] )     Modulation.forward flux_model.py:251-257
//   x_q = fp8-quantised bf16(silu(x))                              F8Linear.forward   float8_quantize.py:272-296
//
// Weight-stream bound (3.2 GB fp8 per step at Flux-dev): each wave streams whole weight rows with
// 16 B/lane loads straight to VGPRs (no LDS round trip: a row is read exactly once), the <= 8
// activation vectors sit dequantised in LDS as fp32 and are re-read per row, wave64 shuffles finish
// the dot products.  fp8 x fp8 products are exact in fp32; accumulation is fp32.
#include "common.h"
#include "fluxmi_internal.h"

namespace {

constexpr int GEMV_ROWS = 64;  // output rows per block (16 per wave)
constexpr int GEMV_MAXB = 8;

template <int FMT> __device__ __forceinline__ float quant_dequant(float x, float scale) {
  const unsigned w = cvt2_fp8<FMT>(q_prepare<FMT>(x, scale), 0.f);
  return fp8_to_f32<FMT>(w, 0);
}

__global__ void __launch_bounds__(256) gemv_kernel(const FluxmiGemvLayer* __restrict__ layers, int n_layers,
                                                   const FluxmiGemvLayer single, int B, int xrow0) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [B][K]
  // ---- locate layer ------------------------------------------------------------------------
  FluxmiGemvLayer L = single;
  if (layers) {
    int li = 0;
    for (int i = 1; i < n_layers; ++i) li = ((int)blockIdx.x >= layers[i].blk_start) ? i : li;
    L = layers[li];
  }
  // xrow0: first of the B activation / output rows this launch handles (step-ahead batches of the modulation table)
  L.x = (const u16*)L.x + (long long)xrow0 * L.ldx;
  L.out = (u16*)L.out + (long long)xrow0 * L.ld_out;
  const int K = L.K;
  const int row0 = ((int)blockIdx.x - L.blk_start) * GEMV_ROWS;
  // ---- stage activations: silu -> (quantise -> dequantise) -> fp32 in LDS ---------------------
  const float in_scale = (L.w_fp8 && L.in_scale) ? *L.in_scale : 1.f;
  for (int i = threadIdx.x; i < B * K; i += 256) {
    const int b = i / K, k = i % K;
    float v = bf2f(((const u16*)L.x)[(long long)b * L.ldx + k]);
    if (L.pre_silu) v = rbf(silu_f(v));
    if (L.w_fp8) v = (L.act_fmt == FLUXMI_FMT_E5M2) ? quant_dequant<FLUXMI_FMT_E5M2>(v, in_scale)
                                                   : quant_dequant<FLUXMI_FMT_E4M3>(v, in_scale);
    xs[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float s = L.w_fp8 ? ((L.sa_recip ? *L.sa_recip : 1.f) * (L.sb_recip ? *L.sb_recip : 1.f)) : 1.f;
  for (int rr = 0; rr < GEMV_ROWS / 4; ++rr) {
    const int n = row0 + wave * (GEMV_ROWS / 4) + rr;
    if (n >= L.N) break;
    float acc[GEMV_MAXB];
#pragma unroll
    for (int b = 0; b < GEMV_MAXB; ++b) acc[b] = 0.f;
    for (int k0 = lane * 16; k0 < K; k0 += 1024) {
      float w[16];
      if (L.w_fp8) {
        const uint4 v = *(const uint4*)((const unsigned char*)L.W + (long long)n * K + k0);
        const unsigned ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) w[q * 4 + e] = fp8_to_f32<FLUXMI_FMT_E4M3>(ww[q], e);
      } else {
        const u16* wp = (const u16*)L.W + (long long)n * K + k0;
        unpack8(*(const uint4*)wp, w);
        unpack8(*(const uint4*)(wp + 8), w + 8);
      }
#pragma unroll
      for (int b = 0; b < GEMV_MAXB; ++b) {
        if (b < B) {
          const float4* xp = (const float4*)(xs + b * K + k0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 xv = xp[q];
            acc[b] = fmaf(w[q * 4 + 0], xv.x, acc[b]);
            acc[b] = fmaf(w[q * 4 + 1], xv.y, acc[b]);
            acc[b] = fmaf(w[q * 4 + 2], xv.z, acc[b]);
            acc[b] = fmaf(w[q * 4 + 3], xv.w, acc[b]);
          }
        }
      }
    }
    const float bias = L.bias ? bf2f(((const u16*)L.bias)[n]) : 0.f;
#pragma unroll
    for (int b = 0; b < GEMV_MAXB; ++b) {
      if (b < B) {
        const float t = wave_sum(acc[b]);
        if (lane == 0) ((u16*)L.out)[(long long)b * L.ld_out + n] = f2bf(fmaf(t, s, bias));
      }
    }
  }
}

}  // namespace

// layers_host: the same descriptors on the host (to size the grid); layers_dev may be nullptr when
// n_layers == 1 (the single descriptor then travels as a kernel argument).
int fluxmi_launch_gemv(const FluxmiGemvLayer* layers_dev, FluxmiGemvLayer* layers_host, int n_layers, int B, int total_blocks,
                       int max_K, hipStream_t s, int row0) {
  FLUXMI_REQUIRE(B >= 1 && B <= GEMV_MAXB, "gemv: batch %d unsupported (1..8)", B);
  FLUXMI_REQUIRE(n_layers >= 1, "gemv: no layers");
  if (layers_host) {
    int blk = 0;
    max_K = 0;
    for (int i = 0; i < n_layers; ++i) {
      FLUXMI_REQUIRE(layers_host[i].K % 16 == 0, "gemv: K=%d must be a multiple of 16", layers_host[i].K);
      layers_host[i].blk_start = blk;
      blk += (layers_host[i].N + GEMV_ROWS - 1) / GEMV_ROWS;
      max_K = layers_host[i].K > max_K ? layers_host[i].K : max_K;
    }
    total_blocks = blk;
    if (layers_dev && n_layers >= 1)  // caller keeps host/dev copies in sync: upload blk_start
      FLUXMI_CHECK_HIP(hipMemcpyAsync((void*)layers_dev, layers_host, sizeof(FluxmiGemvLayer) * n_layers, hipMemcpyHostToDevice, s));
  }
  if (total_blocks == 0) return 0;
  const size_t smem = (size_t)B * max_K * sizeof(float);
  FLUXMI_REQUIRE(smem <= 160 * 1024, "gemv: B*K too large for LDS (%zu bytes)", smem);
  static size_t attr = 0;
  if (smem > attr) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    attr = 160 * 1024;
  }
  FluxmiGemvLayer single = layers_host ? layers_host[0] : FluxmiGemvLayer{};
  hipLaunchKernelGGL(gemv_kernel, dim3(total_blocks), dim3(256), smem, s, layers_dev, n_layers, single, B, row0);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_gemv_blocks(const FluxmiGemvLayer* layers_host, int n_layers) {
  int blk = 0;
  for (int i = 0; i < n_layers; ++i) blk += (layers_host[i].N + GEMV_ROWS - 1) / GEMV_ROWS;
  return blk;
}
