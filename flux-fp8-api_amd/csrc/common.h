// fluxmi -- shared device helpers (gfx950 only).
//
// Numerics contract (DESIGN.md "rounding points"): the reference runs eager PyTorch, i.e. every
// elementwise op rounds its result to the flow dtype (bf16).  The fused kernels here keep values in
// fp32 registers but re-apply a bf16 rounding (rbf) at every point where the reference
// materialises a bf16 tensor, so fused == unfused == oracle up to reduction order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

typedef __bf16 bf16;
typedef unsigned short u16;
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define FLUXMI_FMT_E4M3 0  // MFMA cbsz/blgp code for OCP e4m3fn ("fp8")
#define FLUXMI_FMT_E5M2 1  // MFMA cbsz/blgp code for OCP e5m2   ("bf8")

__device__ __forceinline__ float bf2f(u16 u) { return __uint_as_float(((unsigned)u) << 16); }
// f32 -> bf16 bits, round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ u16 f2bf(float f) {
  bf16 h = (bf16)f;
  return __builtin_bit_cast(u16, h);
}
// round an fp32 value through bf16 (the reference materialised a bf16 tensor here)
__device__ __forceinline__ float rbf(float f) { return (float)((bf16)f); }

// Byte offset of (row m, byte column col) in an fp8 activation buffer with rows of `ld` bytes (ld % 64 == 0): plain row-major, or the ROW-PAIR
// layout [M/2][ld/64][2][64] of fluxmi_gemm_group_t.a_pairs / c8_pairs (the 64-byte K-steps of rows 2r and 2r + 1 share one 128-byte line).
// A store / load of <= 16 bytes that is aligned to its own size never crosses a 64-byte chunk, so every access site just swaps its offset.
__device__ __forceinline__ long long f8_act_off(long long m, long long ld, long long col, int pairs) {
  return pairs ? (m >> 1) * 2 * ld + (m & 1) * 64 + (col >> 6) * 128 + (col & 63) : m * ld + col;
}
// two floats -> packed bf16 pair (RNE).  Written as ONE vector fptrunc so that hipcc selects a single v_cvt_pk_bf16_f32 for the pair;
// two scalar casts + shift + or compile to two half-used v_cvt_pk_bf16_f32 plus two integer ops (4 instructions, same bits).
typedef float v2f_pk_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const v2f_pk_t x = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
  v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
  return v;
}

// eight floats -> eight fp16 (RNE), packed like pack8
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_h2(float lo, float hi) {
  const v2f_pk_t x = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2_t));
}
__device__ __forceinline__ uint4 pack8_f16(const float* f) {
  uint4 v;
  v.x = pack_h2(f[0], f[1]); v.y = pack_h2(f[2], f[3]);
  v.z = pack_h2(f[4], f[5]); v.w = pack_h2(f[6], f[7]);
  return v;
}

// ---- fp8 (OCP) conversion --------------------------------------------------------------------
// float8_quantize.py:217-218 then `.to(fp8)`:  t = bf16(x * scale); clamp(t, +-max); RNE cast.
// The clamp guarantees the hardware convert never sees an out-of-range value.
template <int FMT> __device__ __forceinline__ float fp8_max() { return FMT == FLUXMI_FMT_E5M2 ? 57344.0f : 448.0f; }

// clamp to +-mx that propagates NaN like torch.clamp: IEEE 754-2019 maximum / minimum (v_maximum3_f32 / v_minimum3_f32 on gfx950), two
// instructions -- the compare + select form costs four
__device__ __forceinline__ float clamp_nan(float t, float mx) {
  return __builtin_elementwise_minimum(__builtin_elementwise_maximum(t, -mx), mx);
}
template <int FMT> __device__ __forceinline__ float q_prepare(float x, float scale) { return clamp_nan(rbf(x * scale), fp8_max<FMT>()); }
// two prepared floats -> two fp8 bytes in the low half of the result
template <int FMT> __device__ __forceinline__ unsigned cvt2_fp8(float a, float b) {
  if (FMT == FLUXMI_FMT_E5M2) return (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xffffu;
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}
template <int FMT> __device__ __forceinline__ unsigned cvt4_fp8(float a, float b, float c, float d) {
  int lo = (FMT == FLUXMI_FMT_E5M2) ? __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false)
                                     : __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  int r = (FMT == FLUXMI_FMT_E5M2) ? __builtin_amdgcn_cvt_pk_bf8_f32(c, d, lo, true)
                                    : __builtin_amdgcn_cvt_pk_fp8_f32(c, d, lo, true);
  return (unsigned)r;
}
template <int FMT> __device__ __forceinline__ float fp8_to_f32(unsigned word, int byte);
template <> __device__ __forceinline__ float fp8_to_f32<FLUXMI_FMT_E4M3>(unsigned w, int b) {
  switch (b) {
    case 0: return __builtin_amdgcn_cvt_f32_fp8((int)w, 0);
    case 1: return __builtin_amdgcn_cvt_f32_fp8((int)w, 1);
    case 2: return __builtin_amdgcn_cvt_f32_fp8((int)w, 2);
    default: return __builtin_amdgcn_cvt_f32_fp8((int)w, 3);
  }
}
template <> __device__ __forceinline__ float fp8_to_f32<FLUXMI_FMT_E5M2>(unsigned w, int b) {
  switch (b) {
    case 0: return __builtin_amdgcn_cvt_f32_bf8((int)w, 0);
    case 1: return __builtin_amdgcn_cvt_f32_bf8((int)w, 1);
    case 2: return __builtin_amdgcn_cvt_f32_bf8((int)w, 2);
    default: return __builtin_amdgcn_cvt_f32_bf8((int)w, 3);
  }
}

// ---- activations (fp32 internals exactly as ATen evaluates them on a bf16 tensor) -------------
__device__ __forceinline__ float gelu_tanh_f(float x) {
  // aten GeluKernel (approximate="tanh"): (0.5*x)*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715*x^3), all in fp32.
  // tanh(u) = 2*sigmoid(2u)-1 with sigmoid(2u) = 1/(1 + 2^(x*(c1 + c2*x^2))), c1 = -2*log2(e)*sqrt(2/pi), c2 = 0.044715*c1:
  // 7 plain VALU ops + v_exp_f32 + v_rcp_f32 instead of ~20 with an IEEE division.  tanh is rounded to fp32 BEFORE the
  // "1 +" exactly as aten does, so the catastrophic cancellation of the negative tail (x < -3, result -> -0 below -5.2)
  // is reproduced rather than "fixed"; every caller then rounds to bf16 (>= 99.9 % bit-equal to F.gelu over all bf16 inputs).
  const float c1 = -2.3022081981f, c2 = -0.10294323958f;
  const float z = x * fmaf(x * x, c2, c1);
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
  const float t = fmaf(2.0f, s, -1.0f);
  return (0.5f * x) * (1.0f + t);
}
// The same GELU on a pair of values with packed f32 VALU ops (v_pk_mul/fma/add: two lanes of work per 4-cycle instruction; the
// epilogues are VALU-issue bound).  Bit-identical to gelu_tanh_f per element: identical operations in identical order.
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_t gelu_tanh_f2(v2f_t x) {
  const v2f_t c1 = {-2.3022081981f, -2.3022081981f}, c2 = {-0.10294323958f, -0.10294323958f};
  const v2f_t one = {1.0f, 1.0f}, two = {2.0f, 2.0f}, mone = {-1.0f, -1.0f}, half = {0.5f, 0.5f};
  const v2f_t z = x * __builtin_elementwise_fma(x * x, c2, c1);
  const v2f_t d = one + (v2f_t){__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
  const v2f_t sg = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const v2f_t t = __builtin_elementwise_fma(two, sg, mone);
  return (half * x) * (one + t);
}
// round a pair through bf16: one v_cvt_pk_bf16_f32 + two unpacks
__device__ __forceinline__ v2f_t rbf2(v2f_t x) {
  const unsigned u = pack_bf2(x[0], x[1]);
  return (v2f_t){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// ---- wave-level reductions (wave = 64) ---------------------------------------------------------
// xor butterfly 32, 16, 8, 4, 2, 1 on the VALU only: v_permlane32_swap / v_permlane16_swap (gfx950) for the two widest steps, DPP for the
// rest (row_ror:8 = lane ^ 8 inside a 16-lane row; row_ror:4 = lane ^ 4 once lanes i and i ^ 8 agree; quad_perm for ^2 and ^1).  Same
// additions in the same order as the __shfl_xor loop it replaces (bit-identical results) -- that one compiles to six ds_bpermute round trips
// through the LDS pipeline, each followed by s_waitcnt lgkmcnt(0).
template <class Op>
__device__ __forceinline__ float wave_butterfly(float v, Op op) {
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  auto dpp = [](float x, auto CTRL) {
    const int xi = __float_as_int(x);
    return __int_as_float(__builtin_amdgcn_update_dpp(xi, xi, decltype(CTRL)::value, 0xf, 0xf, false));
  };
  v = op(v, dpp(v, std::integral_constant<int, 0x128>{}));  // row_ror:8
  v = op(v, dpp(v, std::integral_constant<int, 0x124>{}));  // row_ror:4
  v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // quad_perm:[2,3,0,1]
  v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // quad_perm:[1,0,3,2]
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  return wave_butterfly(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
  return wave_butterfly(v, [](float a, float b) { return fmaxf(a, b); });
}

// async global -> LDS copy of 16 bytes per lane (LDS destination = wave-uniform base + lane*16)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware, bijective remap of a 1-D block id: blocks that run on one XCD (bid % 8) get a
// contiguous range of logical ids so that neighbouring tiles share that XCD's private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---- weight prefetch (FluxmiPrefetch, fluxmi_internal.h): workgroup `wg` of `nwg` reads its contiguous share of every range, 16 B per
// lane, eight loads in flight per lane, results discarded.  Default cache policy on purpose: the lines are to stay in the memory-side
// cache (and may stay in this XCD's L2).
template <class PF>
__device__ __forceinline__ void fluxmi_prefetch_ranges(const PF& pf, int wg, int nwg, int tid, int nthreads) {
  typedef int pf_v4i __attribute__((ext_vector_type(4)));
  for (int r = 0; r < pf.n; ++r) {
    const long long n16 = pf.bytes[r] >> 4;
    const long long per = (n16 + nwg - 1) / nwg;
    const long long lo = per * wg, hi = lo + per < n16 ? lo + per : n16;
    const pf_v4i* base = (const pf_v4i*)pf.ptr[r];
    for (long long i = lo + tid; i < hi; i += (long long)nthreads * 8) {
      pf_v4i x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long j = i + (long long)u * nthreads;
        x[u] = base[j < hi ? j : lo];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" ::"v"(x[u]));
    }
  }
}
