// fluxmi -- GEMM epilogue helpers shared by the tile kernels (gemm.hip, gemm_ring.hip).
#pragma once
#include "common.h"
#include "fluxmi_internal.h"

namespace {

template <int CNT> __device__ __forceinline__ void load_bf(const void* base, long long idx, float* out) {
  const u16* p = (const u16*)base + idx;
  if constexpr (CNT == 4) {
    uint2 v = *(const uint2*)p;
    out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xffff0000u);
    out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xffff0000u);
  } else {
    out[0] = bf2f(p[0]);
  }
}
template <int CNT> __device__ __forceinline__ void store_bf(void* base, long long idx, const float* v) {
  u16* p = (u16*)base + idx;
  if constexpr (CNT == 4) {
    uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
    *(uint2*)p = o;
  } else {
    p[0] = f2bf(v[0]);
  }
}
template <int FMT, int CNT> __device__ __forceinline__ void store_q(void* base, long long idx, const float* v, float qs) {
  unsigned char* p = (unsigned char*)base + idx;
  if constexpr (CNT == 4) {
    *(unsigned*)p = cvt4_fp8<FMT>(q_prepare<FMT>(v[0], qs), q_prepare<FMT>(v[1], qs),
                                  q_prepare<FMT>(v[2], qs), q_prepare<FMT>(v[3], qs));
  } else {
    p[0] = (unsigned char)(cvt2_fp8<FMT>(q_prepare<FMT>(v[0], qs), 0.f) & 0xff);
  }
}

// h = bf16(acc*s + bias) already applied by the caller; h holds CNT consecutive columns n..n+CNT-1 of row m
template <int EPI, int FMT, int CNT>
__device__ __forceinline__ void epilogue(const FluxmiGemmGroup& G, float qs, int m, int n,
                                         const float* h, const float* gate) {
  if constexpr (EPI == FLUXMI_EPI_SPLIT) {
    if (n < G.split_n) {
      store_bf<CNT>(G.C, (long long)m * G.ldc + n, h);
    } else {
      float g[CNT];
#pragma unroll
      for (int j = 0; j < CNT; ++j) g[j] = rbf(gelu_tanh_f(h[j]));
      store_q<FMT, CNT>(G.C2, (long long)m * G.ldc2 + G.c2_col0 + (n - G.split_n), g, qs);
    }
  } else if constexpr (EPI == FLUXMI_EPI_BF16) {
    store_bf<CNT>(G.C, (long long)m * G.ldc + n, h);
  } else if constexpr (EPI == FLUXMI_EPI_GELU_QUANT) {
    float g[CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) g[j] = rbf(gelu_tanh_f(h[j]));
    store_q<FMT, CNT>(G.C, (long long)m * G.ldc + n, g, qs);
  } else if constexpr (EPI == FLUXMI_EPI_SILU_QUANT) {
    float g[CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) g[j] = rbf(silu_f(h[j]));
    store_q<FMT, CNT>(G.C, (long long)m * G.ldc + n, g, qs);
  } else if constexpr (EPI == FLUXMI_EPI_QUANT) {
    store_q<FMT, CNT>(G.C, (long long)m * G.ldc + n, h, qs);
  } else if constexpr (EPI == FLUXMI_EPI_GATE_RESID) {
    float r[CNT], o[CNT];
    load_bf<CNT>(G.resid, (long long)m * G.ldr + n, r);
#pragma unroll
    for (int j = 0; j < CNT; ++j) o[j] = r[j] + rbf(gate[j] * h[j]);
    store_bf<CNT>(G.C, (long long)m * G.ldc + n, o);
  }
}

__device__ __forceinline__ float load_scale(const float* p) { return p ? *p : 1.0f; }


}  // namespace
