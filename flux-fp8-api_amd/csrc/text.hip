// fluxmi -- kernels of the text-conditioning encoders (T5-v1.1-XXL encoder + CLIP-L text model), gfx950.
//
// Reference: modules/conditioner.py:74-117 runs transformers' T5EncoderModel / CLIPTextModel in bf16 (attention_mask=None) once per
// request, before the denoise loop.  The Linear layers run on the bf16 MFMA GEMM of gemm*.hip (residual adds in its gate*y+x
// epilogue); this file holds what is left:
//   * row_norm       T5LayerNorm (RMS, no mean / bias; weight applied after a bf16 rounding, as the HF module does) and LayerNorm;
//   * act_mul        gelu_new(wi_0 x) * wi_1 x on the fused [wi_0 | wi_1] GEMM output (T5 v1.1 gated FF), quick_gelu (CLIP MLP);
//   * text_attention self-attention for head_dim 64, L <= 1024: bf16 MFMA, fp32 softmax, additive relative-position bias (T5: a
//                    function of key - query only, so the [H, L, L] tensor is never built) or causal mask + scale (CLIP).
// The attention kernel needs no LDS: with the "swapped" product S^T = K . Q^T a lane owns one query and 16 of a tile's 32 keys, the
// bf16 P values it produces are exactly a B operand of O^T += V^T . P^T once the contraction index of that MFMA is taken in the
// lane's own key order -- and V^T (which the v-projection GEMM writes directly by swapping its operands) then supplies the matching
// A operand as two 8-byte loads per k-step.  Two passes over the keys (row max / sum first, then P and P V): QK^T is recomputed
// rather than staged; at L = 512, d = 64 the whole T5-XXL encoder spends ~0.1 GFLOP per head here, noise next to its 4.7 TFLOP of
// Linear layers.
#include "common.h"
#include "fluxmi_internal.h"

namespace {

// ---- row norm: one block per row ------------------------------------------------------------------------------------------------
// mode 0 (T5LayerNorm): y = bf16( w * bf16( x * rsqrt(mean(x^2) + eps) ) )          mode 1 (LayerNorm): y = bf16( (x - mean) * rstd * w + b )
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(256) row_norm_kernel(const u16* __restrict__ x, const u16* __restrict__ w, const u16* __restrict__ b,
                                                       u16* __restrict__ y, int D, long long ldx, long long ldy, float eps, int mode) {
  __shared__ float red[4];
  const u16* xr = x + (long long)blockIdx.x * ldx;
  u16* yr = y + (long long)blockIdx.x * ldy;
  float s = 0.f;
  if (mode == 1) {
    for (int c = threadIdx.x * 8; c < D; c += 2048) {
      float v[8];
      unpack8(*(const uint4*)(xr + c), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
  }
  const float mean = mode == 1 ? block_sum(s, red) / (float)D : 0.f;
  float q = 0.f;
  for (int c = threadIdx.x * 8; c < D; c += 2048) {
    float v[8];
    unpack8(*(const uint4*)(xr + c), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) q += (v[j] - mean) * (v[j] - mean);
  }
  const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)D + eps);
  for (int c = threadIdx.x * 8; c < D; c += 2048) {
    float v[8], ww[8], bb[8], o[8];
    unpack8(*(const uint4*)(xr + c), v);
    unpack8(*(const uint4*)(w + c), ww);
    if (mode == 1) unpack8(*(const uint4*)(b + c), bb);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = mode == 1 ? (v[j] - mean) * rstd * ww[j] + bb[j] : ww[j] * rbf(v[j] * rstd);
    *(uint4*)(yr + c) = pack8(o);
  }
}

// ---- activations -----------------------------------------------------------------------------------------------------------------
// mode 0: in [R, 2F] = [a | b] -> out [R, F] = bf16( bf16(gelu_new(a)) * b )      mode 1: in [R, F] -> out = bf16( a * sigmoid(1.702 a) )
__global__ void __launch_bounds__(256) act_mul_kernel(const u16* __restrict__ in, u16* __restrict__ out, int R, int F, long long ld_in,
                                                      long long ld_out, int mode) {
  const int f8 = F >> 3;
  const long long total = (long long)R * f8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / f8;
    const int c = (int)(i % f8) * 8;
    float a[8], o[8];
    unpack8(*(const uint4*)(in + r * ld_in + c), a);
    if (mode == 0) {
      float g[8];
      unpack8(*(const uint4*)(in + r * ld_in + F + c), g);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rbf(gelu_tanh_f(a[j])) * g[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = a[j] / (1.0f + __expf(-1.702f * a[j]));
    }
    *(uint4*)(out + r * ld_out + c) = pack8(o);
  }
}

// ---- self-attention, head_dim 64 ---------------------------------------------------------------------------------------------------
struct TextAttnArgs {
  const u16* q; const u16* k; long long ld_qk;   // [Lp, ld_qk], head h at columns h*64 ..
  const u16* vt; long long ld_vt;                // [H*64, ld_vt]: V transposed (row h*64 + d, column = key)
  u16* out; long long ld_out;                    // [Lp, ld_out], head h at columns h*64 ..
  const float* rel_bias; int bias_ld;            // [H, bias_ld] indexed by key - query + bias_ld/2, or nullptr
  const u16* v_bias;                             // [H*64] added to the output (rows of P sum to 1), or nullptr
  float scale; int causal; int L, Lp;
};

// a lane's 16 keys of a 32-key tile, in accumulator order r = 0..15: key = 8*(r/4) + 4*half + r%4
__device__ __forceinline__ void score_tile(const TextAttnArgs& a, const v8bf* qf, int h, int kt, int l31, int half, int qrow, const float* brow,
                                           float* s) {
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const u16* kp = a.k + (long long)(kt * 32 + l31) * a.ld_qk + h * 64 + half * 8;
#pragma unroll
  for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const v8bf*)(kp + c * 16), qf[c], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = kt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
    float v = acc[r] * a.scale;
    if (brow) v += brow[key - qrow];
    if (key >= a.L || (a.causal && key > qrow)) v = -3.0e38f;
    s[r] = v;
  }
}

__global__ void __launch_bounds__(256) text_attention_kernel(const TextAttnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int h = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  if (q0 >= a.Lp) return;
  const int qrow = q0 + l31;
  v8bf qf[4];
  {
    const u16* qp = a.q + (long long)qrow * a.ld_qk + h * 64 + half * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *(const v8bf*)(qp + c * 16);
  }
  const float* brow = a.rel_bias ? a.rel_bias + (long long)h * a.bias_ld + a.bias_ld / 2 : nullptr;
  const int nkt = a.causal ? min(a.Lp / 32, q0 / 32 + 1) : a.Lp / 32;  // causal: tiles past the diagonal are fully masked
  const float LOG2E = 1.4426950408889634f;
  // pass 1: row max and sum over this lane's keys, then across the two lane halves
  float m = -3.0e38f, l = 0.f;
  for (int kt = 0; kt < nkt; ++kt) {
    float s[16];
    score_tile(a, qf, h, kt, l31, half, qrow, brow, s);
    float tm = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, s[r]);
    const float mn = fmaxf(m, tm);
    float add = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f((s[r] - mn) * LOG2E);
    l = l * __builtin_amdgcn_exp2f((m - mn) * LOG2E) + add;
    m = mn;
  }
  {
    const float mo = __shfl_xor(m, 32), lo = __shfl_xor(l, 32);
    const float mt = fmaxf(m, mo);
    l = l * __builtin_amdgcn_exp2f((m - mt) * LOG2E) + lo * __builtin_amdgcn_exp2f((mo - mt) * LOG2E);
    m = mt;
  }
  const float inv = 1.0f / l;
  // pass 2: P = exp(S - m) / l in bf16 (the reference rounds the softmax output to bf16 before P V), O^T += V^T P^T
  v16f o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
  const u16* v0 = a.vt + (long long)(h * 64 + l31) * a.ld_vt + half * 4;
  const u16* v1 = v0 + 32 * a.ld_vt;
  for (int kt = 0; kt < nkt; ++kt) {
    float s[16];
    score_tile(a, qf, h, kt, l31, half, qrow, brow, s);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // contraction slots of k-step t = this lane's accumulator entries r = 8t .. 8t+7  <->  keys kt*32 + 16t + 4*half + {0..3, 8..11}
      v8bf pf;
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[e] = (bf16)(__builtin_amdgcn_exp2f((s[8 * t + e] - m) * LOG2E) * inv);
      const int kb = kt * 32 + 16 * t;
      uint2 a0 = *(const uint2*)(v0 + kb), a1 = *(const uint2*)(v0 + kb + 8);
      uint2 b0 = *(const uint2*)(v1 + kb), b1 = *(const uint2*)(v1 + kb + 8);
      const uint4 va = make_uint4(a0.x, a0.y, a1.x, a1.y), vb = make_uint4(b0.x, b0.y, b1.x, b1.y);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, va), pf, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, vb), pf, o1, 0, 0, 0);
    }
  }
  if (qrow >= a.L) return;
  // O^T[d][q]: this lane holds, for its query, d = 32*dt + 8*(r/4) + 4*half + r%4
  u16* op = a.out + (long long)qrow * a.ld_out + h * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * dt + 8 * g + 4 * half;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = dt == 0 ? o0[4 * g + e] : o1[4 * g + e];
        if (a.v_bias) v[e] = rbf(v[e]) + bf2f(a.v_bias[h * 64 + d + e]);
      }
      *(uint2*)(op + d) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    }
}

int grid1d(long long n) { return (int)((n + 255) / 256 > 65535 * 16 ? 65535 * 16 : (n + 255) / 256); }

}  // namespace

int fluxmi_k_row_norm(const void* x, const void* w, const void* b, void* y, int rows, int D, long long ldx, long long ldy, float eps, int mode,
                      hipStream_t s) {
  FLUXMI_REQUIRE(D % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "row_norm: D and the row strides must be multiples of 8");
  FLUXMI_REQUIRE(mode == 0 || (mode == 1 && b != nullptr), "row_norm: mode 0 (RMS) or 1 (LayerNorm, needs a bias)");
  if (rows == 0 || D == 0) return 0;
  hipLaunchKernelGGL(row_norm_kernel, dim3(rows), dim3(256), 0, s, (const u16*)x, (const u16*)w, (const u16*)b, (u16*)y, D, ldx, ldy, eps, mode);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_act_mul(const void* in, void* out, int rows, int F, long long ld_in, long long ld_out, int mode, hipStream_t s) {
  FLUXMI_REQUIRE(F % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && (mode == 0 || mode == 1), "act_mul: F / strides %% 8 == 0, mode in {0, 1}");
  if ((long long)rows * F == 0) return 0;
  hipLaunchKernelGGL(act_mul_kernel, dim3(grid1d((long long)rows * (F / 8))), dim3(256), 0, s, (const u16*)in, (u16*)out, rows, F, ld_in, ld_out, mode);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_text_attention(const void* q, const void* k, long long ld_qk, const void* vt, long long ld_vt, void* out, long long ld_out,
                            const float* rel_bias, int bias_ld, const void* v_bias, float scale, int causal, int L, int Lp, int H, hipStream_t s) {
  FLUXMI_REQUIRE(Lp % 32 == 0 && L <= Lp && L > 0, "text_attention: Lp=%d must be a multiple of 32 and >= L=%d > 0", Lp, L);
  FLUXMI_REQUIRE(ld_qk % 8 == 0 && ld_vt % 4 == 0 && ld_vt >= Lp && ld_out % 4 == 0, "text_attention: strides (ld_qk %% 8, ld_vt %% 4 and >= Lp, ld_out %% 4)");
  FLUXMI_REQUIRE(rel_bias == nullptr || bias_ld >= 2 * Lp, "text_attention: bias_ld=%d must be >= 2*Lp", bias_ld);
  TextAttnArgs a;
  a.q = (const u16*)q; a.k = (const u16*)k; a.ld_qk = ld_qk; a.vt = (const u16*)vt; a.ld_vt = ld_vt; a.out = (u16*)out; a.ld_out = ld_out;
  a.rel_bias = rel_bias; a.bias_ld = bias_ld; a.v_bias = (const u16*)v_bias; a.scale = scale; a.causal = causal; a.L = L; a.Lp = Lp;
  hipLaunchKernelGGL(text_attention_kernel, dim3((Lp + 127) / 128, H), dim3(256), 0, s, a);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
