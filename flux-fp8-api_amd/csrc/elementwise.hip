// fluxmi -- HBM-bound row / elementwise kernels of the Flux block (gfx950).
//
// Every kernel moves 16 B per lane per access (8 bf16), reduces with wave64 shuffles and applies the
// reference's bf16 rounding points (common.h).  Reference sites are cited per kernel.
#include "common.h"
#include "fluxmi_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// activation quantise: q = fp8( clamp( bf16(x * scale) ) )            float8_quantize.py:217-218,274-276
// x bf16 [rows, cols] (row stride ld_in) -> fp8 bytes [rows, cols] (row stride ld_out)
// ---------------------------------------------------------------------------------------------
template <int FMT>
__global__ void __launch_bounds__(256) quantize_act_kernel(const u16* __restrict__ x, unsigned char* __restrict__ q,
                                                           const float* __restrict__ scale_p, int rows, int cols,
                                                           long long ld_in, long long ld_out) {
  const float scale = *scale_p;
  const int cpr = cols >> 3;  // 8-element chunks per row
  const long long total = (long long)rows * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cpr), c = (int)(i % cpr) * 8;
    float f[8];
    unpack8(*(const uint4*)(x + r * ld_in + c), f);
    uint2 o;
    o.x = cvt4_fp8<FMT>(q_prepare<FMT>(f[0], scale), q_prepare<FMT>(f[1], scale), q_prepare<FMT>(f[2], scale), q_prepare<FMT>(f[3], scale));
    o.y = cvt4_fp8<FMT>(q_prepare<FMT>(f[4], scale), q_prepare<FMT>(f[5], scale), q_prepare<FMT>(f[6], scale), q_prepare<FMT>(f[7], scale));
    *(uint2*)(q + r * ld_out + c) = o;
  }
}

// dequantise fp8 -> fp32: w32 = float(fp8) * scale_recip                       lora_loading.py:615-631
template <int FMT>
__global__ void __launch_bounds__(256) dequant_kernel(const unsigned char* __restrict__ q, float* __restrict__ out,
                                                      const float* __restrict__ recip_p, long long n) {
  const float r = *recip_p;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const unsigned w = *(const unsigned*)(q + i);
    float4 o;
    o.x = fp8_to_f32<FMT>(w, 0) * r; o.y = fp8_to_f32<FMT>(w, 1) * r;
    o.z = fp8_to_f32<FMT>(w, 2) * r; o.w = fp8_to_f32<FMT>(w, 3) * r;
    *(float4*)(out + i) = o;
  }
}

// wave shuffle -> LDS -> ONE atomic per block (16 K same-address atomics from per-wave updates cost ~150 us)
__device__ __forceinline__ void block_atomic_max(float m, float* amax) {
  __shared__ float part[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (m > 0.f) atomicMax((unsigned*)amax, __float_as_uint(m));
  }
}

// ---------------------------------------------------------------------------------------------
// amax = max(|x|) accumulated into *amax (float bits, atomic max valid for non-negative floats)
//                                                                        float8_quantize.py:198,227
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) amax_kernel(const u16* __restrict__ x, float* __restrict__ amax, int rows, int cols,
                                                   long long ld) {
  const int cpr = cols >> 3;
  const long long total = (long long)rows * cpr;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cpr), c = (int)(i % cpr) * 8;
    float f[8];
    unpack8(*(const uint4*)(x + r * ld + c), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(f[j]));
  }
  block_atomic_max(m, amax);
}

// amax of an fp32 tensor after rounding each element to bf16 (LoRA-fused weights: weight.type(dtype))
__global__ void __launch_bounds__(256) amax_f32_as_bf16_kernel(const float* __restrict__ x, float* __restrict__ amax, long long n) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(rbf(x[i])));
  block_atomic_max(m, amax);
}

// scale = clamp(max_val / max(amax, 1e-12), max = max_val); recip = 1/scale      float8_quantize.py:214-215
// `python_float / tensor` is Tensor.__rtruediv__ = reciprocal(tensor) * scalar in PyTorch: TWO roundings, reproduced
// here.  The reciprocals go through fp64 (fp64 division of two fp32 values rounds to the correctly rounded fp32 quotient).
__device__ __forceinline__ float recip_rn(float x) { return (float)(1.0 / (double)x); }
__device__ __forceinline__ float amax_to_scale(float amax, float max_val) {
  return fminf(recip_rn(fmaxf(amax, 1e-12f)) * max_val, max_val);
}

// Calibration state machine of F8Linear.quantize_input (float8_quantize.py:220-246), for `n_layers`
// layers that all saw the tensor whose amax is *amax_p.  trial_index < num_trials: record + running max;
// trial_index == num_trials: freeze from all recorded trials (no new amax is recorded).
__global__ void calib_update_kernel(const float* __restrict__ amax_p, const FluxmiCalibLayer* __restrict__ layers, int n_layers,
                                    int trial_index, int num_trials, float max_val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_layers) return;
  const FluxmiCalibLayer L = layers[i];
  int upto = num_trials;
  if (trial_index < num_trials) {
    L.trials[trial_index] = *amax_p;
    upto = trial_index + 1;
  }
  float m = L.trials[0];
  for (int t = 1; t < upto; ++t) m = fmaxf(m, L.trials[t]);
  const float s = amax_to_scale(m, max_val);
  *L.scale = s;
  *L.recip = recip_rn(s);
}
__global__ void calib_update_single_kernel(const float* amax_p, float* trials, float* scale, float* recip, int trial_index,
                                           int num_trials, float max_val) {
  int upto = num_trials;
  if (trial_index < num_trials) { trials[trial_index] = *amax_p; upto = trial_index + 1; }
  float m = trials[0];
  for (int t = 1; t < upto; ++t) m = fmaxf(m, trials[t]);
  const float s = amax_to_scale(m, max_val);
  *scale = s;
  *recip = recip_rn(s);
}
// weight scale from amax                                                float8_quantize.py:198-203
__global__ void weight_scale_kernel(const float* amax_p, float* scale, float* recip, float max_val) {
  const float s = amax_to_scale(*amax_p, max_val);
  *scale = s;
  *recip = recip_rn(s);
}
// fp32 -> (round to bf16) -> fp8 with scale (re-quantise a LoRA-fused weight)   float8_quantize.py:199-202
template <int FMT>
__global__ void __launch_bounds__(256) quantize_f32_kernel(const float* __restrict__ x, unsigned char* __restrict__ q,
                                                           const float* __restrict__ scale_p, long long n) {
  const float scale = *scale_p;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const float4 v = *(const float4*)(x + i);
    *(unsigned*)(q + i) = cvt4_fp8<FMT>(q_prepare<FMT>(rbf(v.x), scale), q_prepare<FMT>(rbf(v.y), scale),
                                        q_prepare<FMT>(rbf(v.z), scale), q_prepare<FMT>(rbf(v.w), scale));
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm (no affine, eps 1e-6) + modulate (+ optional fp8 quantise), one wave per row
//   y = bf16( bf16( bf16(1 + scale) * LN(x) ) + shift )                 flux_model.py:367-368,374-375,389,395,469-470,501
// rows are laid out [B][L]; rows l < split use stream 0's (shift, scale, q_scale), others stream 1's.
// ---------------------------------------------------------------------------------------------
struct LnModArgs {
  const u16* x; long long ldx, x_bstride;      // row (b,l) at x + b*x_bstride + l*ldx
  void* out; long long ldo, out_bstride;       // bf16 or fp8, same addressing
  const u16* shift[2]; const u16* scale[2]; long long mod_bstride;  // [B][H] each, batch stride in elements
  const float* q_scale[2];
  int B, L, split, H;
  int out_pairs;  // fp8 output in the row-pair layout over the B * L rows (fluxmi_gemm_group_t.a_pairs: dense rows, ldo == H, out_bstride == L * ldo)
};
// One wave per row, four rows per workgroup.  Round 1 kept x, scale and shift of the row in registers (72 packed + ~40 unpacked =
// 124 VGPRs -> 4 waves per SIMD -> the 4608 rows of a 1024x1024 step need 1.125 rounds of the chip, i.e. the kernel took two wave
// lifetimes: 18 us for 42 MB).  Now the workgroup stages bf16(1 + scale) and shift of its stream in LDS once (12 KiB at H = 3072;
// a quarter of the L2 traffic) and the wave holds only its packed row: <= 80 VGPRs -> 6 waves per SIMD -> every row is resident at
// once (single round), and the modulation vectors cost ds_read_b128 instead of global loads.
template <int NCH, bool OUT_FP8, int FMT>
__global__ void __launch_bounds__(256, 5) ln_modulate_kernel(const LnModArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lnm_smem[];  // [m1 = bf16(1+scale) | shift], H bf16 each
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 4;
  const int rows = a.B * a.L;
  const int row = row0 + wave;
  const int rowc = min(row, rows - 1);
  const int b = rowc / a.L, l = rowc % a.L, st = (l < a.split) ? 0 : 1;
  // the workgroup's table belongs to the (batch, stream) of its first row; a wave of another (batch, stream) -- only when the
  // split or L is not a multiple of 4 -- reads its vectors from global memory instead
  const int b0 = row0 / a.L, st0 = ((row0 % a.L) < a.split) ? 0 : 1;
  const bool from_lds = (b == b0) && (st == st0);
  u16* m1_lds = (u16*)lnm_smem;
  u16* sh_lds = m1_lds + a.H;
  {
    const u16* sc0 = a.scale[st0] + (long long)b0 * a.mod_bstride;
    const u16* sh0 = a.shift[st0] + (long long)b0 * a.mod_bstride;
    for (int c = threadIdx.x * 8; c < a.H; c += 256 * 8) {
      float fs[8], m1[8];
      unpack8(*(const uint4*)(sc0 + c), fs);
#pragma unroll
      for (int j = 0; j < 8; ++j) m1[j] = 1.0f + fs[j];
      *(uint4*)(m1_lds + c) = pack8(m1);                       // bf16(1 + scale): the reference materialises it (flux_model.py:367)
      *(uint4*)(sh_lds + c) = *(const uint4*)(sh0 + c);
    }
  }
  const u16* xr = a.x + (long long)b * a.x_bstride + (long long)l * a.ldx;
  const long long orow = (long long)b * a.out_bstride + (long long)l * a.ldo;
  uint4 xr4[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = (lane + 64 * k) * 8;
    xr4[k] = (c < a.H) ? *(const uint4*)(xr + c) : make_uint4(0, 0, 0, 0);
  }
  float qs = 1.f;
  if (OUT_FP8) qs = *a.q_scale[st];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    float v[8];
    unpack8(xr4[k], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
  }
  const float mean = wave_sum(sum) / (float)a.H;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = (lane + 64 * k) * 8;
    if (c < a.H) {
      float v[8];
      unpack8(xr4[k], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; sq += d * d; }
    }
  }
  const float var = wave_sum(sq) / (float)a.H;
  const float rstd = 1.0f / sqrtf(var + 1e-6f);
  __syncthreads();  // the table is complete (every wave reaches this: no early exit above)
  if (row >= rows) return;
  const u16* scg = a.scale[st] + (long long)b * a.mod_bstride;
  const u16* shg = a.shift[st] + (long long)b * a.mod_bstride;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = (lane + 64 * k) * 8;
    if (c < a.H) {
      float v[8], m1[8], fh[8], y[8];
      unpack8(xr4[k], v);
      if (from_lds) {
        unpack8(*(const uint4*)(m1_lds + c), m1);
        unpack8(*(const uint4*)(sh_lds + c), fh);
      } else {
        float fs[8];
        unpack8(*(const uint4*)(scg + c), fs);
        unpack8(*(const uint4*)(shg + c), fh);
#pragma unroll
        for (int j = 0; j < 8; ++j) m1[j] = rbf(1.0f + fs[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float n = rbf((v[j] - mean) * rstd);
        y[j] = rbf(rbf(m1[j] * n) + fh[j]);
      }
      if (OUT_FP8) {
        uint2 o;
        o.x = cvt4_fp8<FMT>(q_prepare<FMT>(y[0], qs), q_prepare<FMT>(y[1], qs), q_prepare<FMT>(y[2], qs), q_prepare<FMT>(y[3], qs));
        o.y = cvt4_fp8<FMT>(q_prepare<FMT>(y[4], qs), q_prepare<FMT>(y[5], qs), q_prepare<FMT>(y[6], qs), q_prepare<FMT>(y[7], qs));
        *(uint2*)((unsigned char*)a.out + orow + c) = o;
      } else {
        *(uint4*)((u16*)a.out + orow + c) = pack8(y);
      }
    }
  }
}

// Streaming variant (round 2, FLUXMI_LN_V=2).  The one-wave-per-row kernel above puts EVERY row of a step on the chip at once, so all
// waves move through the same phases together -- load (HBM busy, VALU idle), moments + modulate + quantise (~20 VALU ops per element:
// VALU busy, HBM idle), store -- and the launch costs the SUM of its HBM and VALU time (18.5 us for 42 MB at L = 4608; either alone
// is ~9 us).  Here one 8-wave workgroup per CU walks a contiguous block of rows; a wave owns every 8th row of it and issues the loads
// of its NEXT row before it starts the arithmetic of the current one, so HBM and VALU time overlap inside each wave; both streams'
// bf16(1 + scale) | shift vectors of the workgroup's batch element sit in LDS (4 x H bf16).
// FULL: H == NCH * 512, i.e. no column guard anywhere (a guarded load sits in its own basic block and hipcc then waits vmcnt(0) at the
// join -- the prefetch would be waited for right after its issue).  The next-row load is unconditional for the same reason (the
// last iteration re-loads the wave's last row).
template <int NCH, bool OUT_FP8, int FMT, bool FULL>
__global__ void __launch_bounds__(512) ln_modulate_stream_kernel(const LnModArgs a, int rows_per_wg, int wgs_per_b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lnm_smem[];  // [stream][m1 | shift][H] as fp32 (of the bf16 values: no unpack per use)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row and stream live in SGPRs
  // a workgroup's rows lie inside ONE batch element (grid = B x wgs_per_b): its modulation vectors always come from the LDS table.  (With
  // workgroups allowed to straddle two batch elements, the table-or-global choice sat as a branch inside every 512-column chunk: 12 basic
  // blocks per row, an LDS round trip exposed in each.)
  const int b0 = blockIdx.x / wgs_per_b;
  const int r_begin = (blockIdx.x - b0 * wgs_per_b) * rows_per_wg, r_end = min(a.L, r_begin + rows_per_wg);
  const u16* xb = a.x + (long long)b0 * a.x_bstride;
  auto load_row = [&](int l, uint4 (&dst)[NCH]) {
    const u16* xr = xb + (long long)l * a.ldx;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = (lane + 64 * k) * 8;
      if (FULL) dst[k] = *(const uint4*)(xr + c);
      else dst[k] = (c < a.H) ? *(const uint4*)(xr + c) : make_uint4(0, 0, 0, 0);
    }
  };
  int row = r_begin + wave;
  uint4 cur[NCH];
  load_row(min(row, r_end - 1), cur);
  float* tab = (float*)lnm_smem;
  for (int st = 0; st < 2; ++st) {
    const u16* sc0 = a.scale[st] + (long long)b0 * a.mod_bstride;
    const u16* sh0 = a.shift[st] + (long long)b0 * a.mod_bstride;
    for (int c = threadIdx.x * 8; c < a.H; c += 512 * 8) {
      float fs[8], sh[8];
      unpack8(*(const uint4*)(sc0 + c), fs);
      unpack8(*(const uint4*)(sh0 + c), sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) fs[j] = rbf(1.0f + fs[j]);  // bf16(1 + scale): the reference materialises it (flux_model.py:367)
      float* tm0 = tab + (st * 2) * a.H + c;
      *(float4*)tm0 = make_float4(fs[0], fs[1], fs[2], fs[3]); *(float4*)(tm0 + 4) = make_float4(fs[4], fs[5], fs[6], fs[7]);
      *(float4*)(tm0 + a.H) = make_float4(sh[0], sh[1], sh[2], sh[3]); *(float4*)(tm0 + a.H + 4) = make_float4(sh[4], sh[5], sh[6], sh[7]);
    }
  }
  float qs2[2] = {1.f, 1.f};
  if (OUT_FP8) { qs2[0] = *a.q_scale[0]; qs2[1] = *a.q_scale[1]; }
  __syncthreads();
  while (row < r_end) {
    const int nrow = row + 8;
    uint4 nxt[NCH];
    load_row(min(nrow, r_end - 1), nxt);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch on top (hipcc sinks it below the first reduction otherwise)
    const int l = row, st = (l < a.split) ? 0 : 1;
    const float qs = st ? qs2[1] : qs2[0];
    const long long grow = (long long)b0 * a.L + l;  // row-pair layout: rows 2r, 2r + 1 interleaved in 64-byte chunks (common.h f8_act_off)
    const long long orow = (OUT_FP8 && a.out_pairs) ? (grow >> 1) * 2 * a.ldo + (grow & 1) * 64 : (long long)b0 * a.out_bstride + (long long)l * a.ldo;
    // pairs of elements in packed f32 VALU ops (v_pk_add / v_pk_mul / v_pk_fma: two elements per instruction) -- the kernel is VALU-bound
    v2f_t sum2 = {0.f, 0.f};
    float xv[NCH][8];  // the row in fp32, unpacked once for the three passes
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      unpack8(cur[k], xv[k]);
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(xv[k][j]));  // opaque: hipcc would re-unpack the packed row in every pass
#pragma unroll
      for (int j = 0; j < 8; j += 2) sum2 += (v2f_t){xv[k][j], xv[k][j + 1]};
    }
    // (the order of the fp32 additions differs from the one-wave-per-row kernel's; both are within fp32 rounding of the exact
    // mean, and LayerNorm's output is rounded to bf16 -- tests/test_ops_gpu.py::test_ln_modulate holds both to the same gate)
    const float mean = wave_sum(sum2[0] + sum2[1]) / (float)a.H;
    const v2f_t mean2 = {mean, mean};
    v2f_t sq2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = (lane + 64 * k) * 8;
      if (FULL || c < a.H) {
        const float* v = xv[k];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const v2f_t d = (v2f_t){v[j], v[j + 1]} - mean2;
          sq2 = __builtin_elementwise_fma(d, d, sq2);
        }
      }
    }
    const float sq = sq2[0] + sq2[1];
    const float var = wave_sum(sq) / (float)a.H;
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    // the chunk's bf16(1 + scale) | shift words come from the LDS table one chunk ahead of their use
    const float* tm = tab + (st * 2) * a.H + lane * 8;
    const float* tsft = tm + a.H;
    auto tab_ok = [&](int k) { return FULL || (lane + 64 * k) * 8 < a.H; };
    struct Tab { float4 m[2], s[2]; };
    auto tab_read = [&](int k) {
      Tab t;
      t.m[0] = *(const float4*)(tm + 512 * k); t.m[1] = *(const float4*)(tm + 512 * k + 4);
      t.s[0] = *(const float4*)(tsft + 512 * k); t.s[1] = *(const float4*)(tsft + 512 * k + 4);
      return t;
    };
    Tab tc = tab_read(0);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = (lane + 64 * k) * 8;
      Tab tn = tc;
      if (k + 1 < NCH && tab_ok(k + 1)) tn = tab_read(k + 1);
      if (FULL || c < a.H) {
        float y[8];
        const float* v = xv[k];
        const float m1[8] = {tc.m[0].x, tc.m[0].y, tc.m[0].z, tc.m[0].w, tc.m[1].x, tc.m[1].y, tc.m[1].z, tc.m[1].w};
        const float fh[8] = {tc.s[0].x, tc.s[0].y, tc.s[0].z, tc.s[0].w, tc.s[1].x, tc.s[1].y, tc.s[1].z, tc.s[1].w};
        const v2f_t rstd2 = {rstd, rstd}, qs2v = {qs, qs};
        float q[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          // one v_cvt_pk_bf16_f32 per rounded PAIR (rbf2); same operations in the same order as the scalar form
          const v2f_t n = rbf2(((v2f_t){v[j], v[j + 1]} - mean2) * rstd2);
          const v2f_t yy = rbf2(rbf2((v2f_t){m1[j], m1[j + 1]} * n) + (v2f_t){fh[j], fh[j + 1]});
          y[j] = yy[0]; y[j + 1] = yy[1];
          if (OUT_FP8) {
            const v2f_t t = rbf2(yy * qs2v);
            const float mx = fp8_max<FMT>();
            q[j] = clamp_nan(t[0], mx);  // propagates NaN like torch.clamp; +-inf saturate
            q[j + 1] = clamp_nan(t[1], mx);
          }
        }
        if (OUT_FP8) {
          uint2 o;
          o.x = cvt4_fp8<FMT>(q[0], q[1], q[2], q[3]);
          o.y = cvt4_fp8<FMT>(q[4], q[5], q[6], q[7]);
          *(uint2*)((unsigned char*)a.out + orow + (a.out_pairs ? ((c >> 6) * 128 + (c & 63)) : c)) = o;
        } else {
          *(uint4*)((u16*)a.out + orow + c) = pack8(y);
        }
      }
      tc = tn;
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) cur[k] = nxt[k];
    row = nrow;
  }
}

// ---------------------------------------------------------------------------------------------
// unfused elementwise ops (calibration / mixed-precision path)
// ---------------------------------------------------------------------------------------------
// mode 0: gelu(tanh)  flux_model.py:301,335,455,480     mode 1: silu  flux_model.py:139,249,496
__global__ void __launch_bounds__(256) act_kernel(const u16* x, u16* y, int rows, int cols,
                                                  long long ld_in, long long ld_out, int mode) {
  const int cpr = cols >> 3;
  const long long total = (long long)rows * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cpr), c = (int)(i % cpr) * 8;
    float f[8];
    unpack8(*(const uint4*)(x + r * ld_in + c), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = mode == 0 ? gelu_tanh_f(f[j]) : silu_f(f[j]);
    *(uint4*)(y + r * ld_out + c) = pack8(f);
  }
}

// out = bf16( x + bf16(gate[b] * y) ),  rows [B][L], gate [B][H] (batch stride gate_bstride)   flux_model.py:387-396,484
__global__ void __launch_bounds__(256) gate_residual_kernel(const u16* __restrict__ x, const u16* __restrict__ y,
                                                            const u16* __restrict__ gate, u16* __restrict__ out, int B, int L,
                                                            int H, long long ldx, long long ldy, long long ldo,
                                                            long long gate_bstride) {
  const int cpr = H >> 3;
  const long long total = (long long)B * L * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cpr;
    const int c = (int)(i % cpr) * 8, b = (int)(r / L);
    float fx[8], fy[8], fg[8];
    unpack8(*(const uint4*)(x + r * ldx + c), fx);
    unpack8(*(const uint4*)(y + r * ldy + c), fy);
    unpack8(*(const uint4*)(gate + b * gate_bstride + c), fg);
#pragma unroll
    for (int j = 0; j < 8; ++j) fx[j] = fx[j] + rbf(fg[j] * fy[j]);
    *(uint4*)(out + r * ldo + c) = pack8(fx);
  }
}

// z = bf16(a + b) elementwise over n (multiple of 8) elements           flux_model.py:694,697
__global__ void __launch_bounds__(256) add_kernel(const u16* __restrict__ a, const u16* __restrict__ b, u16* __restrict__ z, long long n) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long long)gridDim.x * blockDim.x * 8) {
    float fa[8], fb[8];
    unpack8(*(const uint4*)(a + i), fa);
    unpack8(*(const uint4*)(b + i), fb);
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] += fb[j];
    *(uint4*)(z + i) = pack8(fa);
  }
}

// z[r, :] = a[r, :] + b[r % nb, :]  (rows of n8*8 bf16): the step-invariant parts of `vec` added to every step's time embedding
__global__ void __launch_bounds__(256) add_bcast_kernel(const u16* __restrict__ a, const u16* __restrict__ b, u16* __restrict__ z,
                                                        long long rows, int nb, int cols) {
  const int cpr = cols >> 3;
  const long long total = rows * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cpr;
    const int c = (int)(i % cpr) * 8;
    float fa[8], fb[8];
    unpack8(*(const uint4*)(a + r * cols + c), fa);
    unpack8(*(const uint4*)(b + (r % nb) * cols + c), fb);
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] += fb[j];
    *(uint4*)(z + r * cols + c) = pack8(fa);
  }
}
// dst[0..n) = table[(*step - step0) * n ..]: the modulation vectors of the current denoise step out of the per-request table
__global__ void __launch_bounds__(256) select_step_kernel(const uint4* __restrict__ table, const int* __restrict__ step,
                                                          const int* __restrict__ step0, uint4* __restrict__ dst, long long n16) {
  const uint4* src = table + (long long)(*step - *step0) * n16;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

// lut[b] = fp8( clamp( bf16( bf16(act(bf16 b)) * scale ) ) ) for all 65536 bf16 bit patterns b: the quantising GEMM epilogues as a
// table (same helpers as gemm_epilogue.h -> bit-identical).  act: 0 none, 1 gelu-tanh, 2 silu.
template <int FMT>
__global__ void __launch_bounds__(256) build_qlut_kernel(const float* __restrict__ scale, int act, unsigned char* __restrict__ lut) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float h = bf2f((u16)i);
  const float g = act == 1 ? rbf(gelu_tanh_f(h)) : act == 2 ? rbf(silu_f(h)) : h;
  lut[i] = (unsigned char)(cvt2_fp8<FMT>(q_prepare<FMT>(g, *scale), 0.f) & 0xff);
}

// ---------------------------------------------------------------------------------------------
// sinusoidal timestep embedding                                           flux_model.py:95-116
//   t' = bf16(1000 * t);  out[b, i] = bf16(cos(t' * f_i)),  out[b, half+i] = bf16(sin(t' * f_i))
// freqs[half] is the fp32 table exp(-ln(1e4) * i / half) computed once by the host.
// ---------------------------------------------------------------------------------------------
__global__ void timestep_embedding_kernel(const u16* __restrict__ t, const float* __restrict__ freqs, u16* __restrict__ out,
                                          int B, int half, float time_factor) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i % half;
  const float tt = rbf(time_factor * bf2f(t[b]));
  const float arg = tt * freqs[k];
  out[(long long)b * 2 * half + k] = f2bf(cosf(arg));
  out[(long long)b * 2 * half + half + k] = f2bf(sinf(arg));
}

// RoPE table: pe[b, l, p] = (bf16 cos, bf16 sin) of  float(ids[b,l,axis(p)]) * omega[p]     flux_model.py:49-57,82-92
__global__ void rope_table_kernel(const u16* __restrict__ ids, const float* __restrict__ omega, const int* __restrict__ axis,
                                  u16* __restrict__ pe, long long rows, int n_axes, int pairs) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * pairs) return;
  const long long r = i / pairs;
  const int p = (int)(i % pairs);
  const float ang = bf2f(ids[r * n_axes + axis[p]]) * omega[p];
  pe[i * 2] = f2bf(cosf(ang));
  pe[i * 2 + 1] = f2bf(sinf(ang));
}

// Euler step of the flow ODE:  img = bf16( img + bf16(dt * pred) ),  dt = dts[*step]       flux_pipeline.py:651
__global__ void __launch_bounds__(256) euler_kernel(u16* __restrict__ img, const u16* __restrict__ pred, const float* __restrict__ dts,
                                                    const int* __restrict__ step, long long n) {
  const float dt = dts[step ? *step : 0];
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long long)gridDim.x * blockDim.x * 8) {
    float fi[8], fp[8];
    unpack8(*(const uint4*)(img + i), fi);
    unpack8(*(const uint4*)(pred + i), fp);
#pragma unroll
    for (int j = 0; j < 8; ++j) fi[j] += rbf(dt * fp[j]);
    *(uint4*)(img + i) = pack8(fi);
  }
}
// per-step scalars kept on the device so that one captured graph serves every step:
//   t_vec[b] = bf16(ts[*step]),  then ++*step happens in advance_step_kernel at the end of the step.
__global__ void set_timestep_kernel(u16* __restrict__ t_vec, const float* __restrict__ ts, const int* __restrict__ step, int B) {
  const int b = threadIdx.x;
  if (b < B) t_vec[b] = f2bf(ts[*step]);
}
__global__ void advance_step_kernel(int* step) { *step += 1; }
// rows of the step-ahead modulation table: t_rows[r] = bf16(ts[step0 + r / B]) (the value Flux.forward receives, flux_pipeline.py:636-640)
__global__ void __launch_bounds__(256) timestep_rows_kernel(u16* __restrict__ t_rows, const float* __restrict__ ts, int step0, int B, int R) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < R) t_rows[r] = f2bf(ts[step0 + r / B]);
}
__global__ void fill_bf16_kernel(u16* __restrict__ dst, float v, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < n) dst[i] = f2bf(v);
}

// w += delta (fp32), n multiple of 4                                       lora_loading.py:564-577
__global__ void __launch_bounds__(256) axpy_f32_kernel(float* __restrict__ w, const float* __restrict__ d, float alpha, long long n) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    float4 a = *(float4*)(w + i);
    const float4 b = *(const float4*)(d + i);
    a.x += alpha * b.x; a.y += alpha * b.y; a.z += alpha * b.z; a.w += alpha * b.w;
    *(float4*)(w + i) = a;
  }
}
// fp32 rank-r update: delta[n,k] = sum_r B[n,r] * A[r,k]   (LoRA, r <= 128)      lora_loading.py:509-544
__global__ void __launch_bounds__(256) lora_delta_kernel(const float* __restrict__ Bm, const float* __restrict__ A,
                                                         float* __restrict__ delta, int N, int K, int R, float scale, int accumulate) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (k >= K) return;
  float acc = 0.f;
  for (int r = 0; r < R; ++r) acc = fmaf(Bm[(long long)n * R + r], A[(long long)r * K + k], acc);
  acc *= scale;
  float* d = delta + (long long)n * K + k;
  *d = accumulate ? (*d + acc) : acc;
}

inline int grid_for(long long work_items, int block = 256, int cap = 256 * 16) {
  long long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace

// ================================== launchers (internal C++ API) ==================================
int fluxmi_k_quantize_act(const void* x, void* q, const float* scale, int rows, int cols, long long ld_in, long long ld_out,
                          int fmt, hipStream_t s) {
  FLUXMI_REQUIRE(cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0, "quantize_act: cols/ld must be multiples of 8");
  if (rows == 0 || cols == 0) return 0;
  const int g = grid_for((long long)rows * (cols / 8));
  if (fmt == FLUXMI_FMT_E5M2)
    hipLaunchKernelGGL(quantize_act_kernel<FLUXMI_FMT_E5M2>, dim3(g), dim3(256), 0, s, (const u16*)x, (unsigned char*)q, scale, rows, cols, ld_in, ld_out);
  else
    hipLaunchKernelGGL(quantize_act_kernel<FLUXMI_FMT_E4M3>, dim3(g), dim3(256), 0, s, (const u16*)x, (unsigned char*)q, scale, rows, cols, ld_in, ld_out);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_amax(const void* x, float* amax, int rows, int cols, long long ld, hipStream_t s) {
  FLUXMI_REQUIRE(cols % 8 == 0 && ld % 8 == 0, "amax: cols/ld must be multiples of 8");
  if (rows == 0 || cols == 0) return 0;
  hipLaunchKernelGGL(amax_kernel, dim3(grid_for((long long)rows * (cols / 8), 256, 1024)), dim3(256), 0, s, (const u16*)x, amax, rows, cols, ld);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_calib_update(const float* amax, float* trials, float* scale, float* recip, int trial_index, int num_trials,
                          float max_val, hipStream_t s) {
  hipLaunchKernelGGL(calib_update_single_kernel, dim3(1), dim3(1), 0, s, amax, trials, scale, recip, trial_index, num_trials, max_val);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
int fluxmi_k_calib_update_many(const float* amax, const void* layers_dev, int n_layers, int trial_index, int num_trials,
                               float max_val, hipStream_t s) {
  if (n_layers == 0) return 0;
  hipLaunchKernelGGL(calib_update_kernel, dim3((n_layers + 63) / 64), dim3(64), 0, s, amax, (const FluxmiCalibLayer*)layers_dev, n_layers,
                     trial_index, num_trials, max_val);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_quantize_weight(const void* w_bf16, void* q, float* amax_tmp, float* scale, float* recip, int N, int K, int fmt,
                             hipStream_t s) {
  FLUXMI_REQUIRE(K % 8 == 0, "quantize_weight: K must be a multiple of 8");
  FLUXMI_CHECK_HIP(hipMemsetAsync(amax_tmp, 0, sizeof(float), s));
  int rc = fluxmi_k_amax(w_bf16, amax_tmp, N, K, K, s);
  if (rc) return rc;
  const float mx = fmt == FLUXMI_FMT_E5M2 ? 57344.f : 448.f;
  hipLaunchKernelGGL(weight_scale_kernel, dim3(1), dim3(1), 0, s, amax_tmp, scale, recip, mx);
  FLUXMI_LAUNCH_CHECK();
  return fluxmi_k_quantize_act(w_bf16, q, scale, N, K, K, K, fmt, s);
}

int fluxmi_k_dequant(const void* q, float* out, const float* recip, long long n, int fmt, hipStream_t s) {
  FLUXMI_REQUIRE(n % 4 == 0, "dequant: n must be a multiple of 4");
  if (n == 0) return 0;
  const int g = grid_for(n / 4);
  if (fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL(dequant_kernel<FLUXMI_FMT_E5M2>, dim3(g), dim3(256), 0, s, (const unsigned char*)q, out, recip, n);
  else hipLaunchKernelGGL(dequant_kernel<FLUXMI_FMT_E4M3>, dim3(g), dim3(256), 0, s, (const unsigned char*)q, out, recip, n);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_requantize_f32(const float* w32, void* q, float* amax_tmp, float* scale, float* recip, long long n, int fmt,
                            hipStream_t s) {
  FLUXMI_REQUIRE(n % 4 == 0, "requantize: n must be a multiple of 4");
  FLUXMI_CHECK_HIP(hipMemsetAsync(amax_tmp, 0, sizeof(float), s));
  hipLaunchKernelGGL(amax_f32_as_bf16_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, w32, amax_tmp, n);
  const float mx = fmt == FLUXMI_FMT_E5M2 ? 57344.f : 448.f;
  hipLaunchKernelGGL(weight_scale_kernel, dim3(1), dim3(1), 0, s, amax_tmp, scale, recip, mx);
  if (fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL(quantize_f32_kernel<FLUXMI_FMT_E5M2>, dim3(grid_for(n / 4)), dim3(256), 0, s, w32, (unsigned char*)q, scale, n);
  else hipLaunchKernelGGL(quantize_f32_kernel<FLUXMI_FMT_E4M3>, dim3(grid_for(n / 4)), dim3(256), 0, s, w32, (unsigned char*)q, scale, n);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_lora_delta(const float* Bm, const float* A, float* delta, int N, int K, int R, float scale, int accumulate, hipStream_t s) {
  if (N == 0 || K == 0) return 0;
  hipLaunchKernelGGL(lora_delta_kernel, dim3((K + 255) / 256, N), dim3(256), 0, s, Bm, A, delta, N, K, R, scale, accumulate);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
int fluxmi_k_axpy_f32(float* w, const float* d, float alpha, long long n, hipStream_t s) {
  FLUXMI_REQUIRE(n % 4 == 0, "axpy: n must be a multiple of 4");
  if (n == 0) return 0;
  hipLaunchKernelGGL(axpy_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, w, d, alpha, n);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_ln_modulate(const void* x, long long ldx, long long x_bstride, void* out, long long ldo, long long out_bstride,
                         const void* shift0, const void* scale0, const void* shift1, const void* scale1, long long mod_bstride,
                         const float* q0, const float* q1, int B, int L, int split, int H, int out_fp8, int fmt, hipStream_t s, int out_pairs) {
  FLUXMI_REQUIRE(H % 8 == 0 && H <= 4096, "ln_modulate: hidden size %d unsupported (need %%8==0, <=4096)", H);
  FLUXMI_REQUIRE(!out_pairs || (out_fp8 && ldo == H && out_bstride == (long long)L * ldo && H % 64 == 0 && ((long long)B * L) % 2 == 0 &&
                                fluxmi_tuning().ln_variant >= 2 && (size_t)H * 16 <= 49152),
                 "ln_modulate: the row-pair output layout needs fp8 output, dense rows, an even row count and the streaming kernel");
  FLUXMI_REQUIRE(!out_fp8 || (q0 && q1), "ln_modulate: fp8 output needs q_scale pointers");
  if (B * L == 0) return 0;
  LnModArgs a;
  a.x = (const u16*)x; a.ldx = ldx; a.x_bstride = x_bstride; a.out = out; a.ldo = ldo; a.out_bstride = out_bstride;
  a.shift[0] = (const u16*)shift0; a.scale[0] = (const u16*)scale0;
  a.shift[1] = (const u16*)shift1; a.scale[1] = (const u16*)scale1;
  a.mod_bstride = mod_bstride; a.q_scale[0] = q0; a.q_scale[1] = q1;
  a.B = B; a.L = L; a.split = split; a.H = H; a.out_pairs = out_pairs;
  const int nch = (H + 511) / 512;
  // fluxmi_tuning_t.ln_variant: 2 = streaming kernel (one 8-wave workgroup per CU, next row's loads under this row's arithmetic),
  // 1 = one wave per row, every row resident at once
  const int lnv = fluxmi_tuning().ln_variant;
  if (lnv >= 2 && (size_t)H * 16 <= 49152) {  // the streaming kernel keeps four fp32 vectors of H in LDS (48 KiB at H = 3072)
    // one workgroup per CU overall, each inside one batch element
    const int n_wg_b = std::max(1, std::min((lnv == 3 ? 512 : 256) / std::max(B, 1), (L + 7) / 8));
    const int rows_per_wg = (L + n_wg_b - 1) / n_wg_b;
    const int wgs_per_b = (L + rows_per_wg - 1) / rows_per_wg;
    const dim3 grid(B * wgs_per_b), block(512);
    const size_t lds = (size_t)H * 16;  // [stream][1 + scale | shift][H] fp32
#define LNS2(N_, F_)                                                                                                             \
  do {                                                                                                                           \
    if (!out_fp8) hipLaunchKernelGGL((ln_modulate_stream_kernel<N_, false, FLUXMI_FMT_E5M2, F_>), grid, block, lds, s, a, rows_per_wg, wgs_per_b);           \
    else if (fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL((ln_modulate_stream_kernel<N_, true, FLUXMI_FMT_E5M2, F_>), grid, block, lds, s, a, rows_per_wg, wgs_per_b); \
    else hipLaunchKernelGGL((ln_modulate_stream_kernel<N_, true, FLUXMI_FMT_E4M3, F_>), grid, block, lds, s, a, rows_per_wg, wgs_per_b);                     \
  } while (0)
#define LNS(N_) do { if (H == N_ * 512) LNS2(N_, true); else LNS2(N_, false); } while (0)
    if (nch <= 1) LNS(1);
    else if (nch <= 2) LNS(2);
    else if (nch <= 4) LNS(4);
    else if (nch <= 6) LNS(6);
    else LNS(8);
#undef LNS2
#undef LNS
    FLUXMI_LAUNCH_CHECK();
    return 0;
  }
  const dim3 grid((B * L + 3) / 4), block(256);
  const size_t lds = (size_t)H * 4;  // bf16(1 + scale) and shift of the workgroup's stream
#define LNM(N_)                                                                                                       \
  do {                                                                                                                \
    if (!out_fp8) hipLaunchKernelGGL((ln_modulate_kernel<N_, false, FLUXMI_FMT_E5M2>), grid, block, lds, s, a);           \
    else if (fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL((ln_modulate_kernel<N_, true, FLUXMI_FMT_E5M2>), grid, block, lds, s, a); \
    else hipLaunchKernelGGL((ln_modulate_kernel<N_, true, FLUXMI_FMT_E4M3>), grid, block, lds, s, a);                     \
  } while (0)
  if (nch <= 1) LNM(1);
  else if (nch <= 2) LNM(2);
  else if (nch <= 4) LNM(4);
  else if (nch <= 6) LNM(6);
  else LNM(8);
#undef LNM
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

// rows x row_bytes -> [rows/2][row_bytes/64][2][64]: the 64-byte K-steps of a row pair share one 128-byte line (fluxmi_gemm_group_t.W_pairs).
// One 16-byte chunk per thread; reads and writes are both 64-byte runs.
// INVERSE: the same mapping read the other way (row-pair layout -> plain rows; the engine's test hook fluxmi_engine_copy_buffer).
template <bool INVERSE>
__global__ void __launch_bounds__(256) pair_rows_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long long chunks, int cpr) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cpr;
    const int c = (int)(i - r * cpr);                   // 16-byte chunk inside the row
    const long long line = (r >> 1) * (cpr >> 2) + (c >> 2);  // 128-byte line of (row pair, K-step)
    if (INVERSE) out[i] = in[line * 8 + (r & 1) * 4 + (c & 3)];
    else out[line * 8 + (r & 1) * 4 + (c & 3)] = in[i];
  }
}
int fluxmi_k_unpair_rows(const void* in, void* out, int rows, long long row_bytes, hipStream_t s) {
  FLUXMI_REQUIRE(in && out && in != out, "unpair_rows: NULL or aliased buffers");
  FLUXMI_REQUIRE(rows >= 0 && rows % 2 == 0 && row_bytes > 0 && row_bytes % 64 == 0, "unpair_rows: rows %d (even), row_bytes %lld (multiple of 64)", rows, row_bytes);
  if (rows == 0) return 0;
  const long long chunks = (long long)rows * (row_bytes / 16);
  hipLaunchKernelGGL(pair_rows_kernel<true>, dim3(grid_for(chunks)), dim3(256), 0, s, (const uint4*)in, (uint4*)out, chunks, (int)(row_bytes / 16));
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
int fluxmi_k_pair_rows(const void* in, void* out, int rows, long long row_bytes, hipStream_t s) {
  FLUXMI_REQUIRE(in && out && in != out, "pair_rows: NULL or aliased buffers");
  FLUXMI_REQUIRE(rows >= 0 && rows % 2 == 0 && row_bytes > 0 && row_bytes % 64 == 0, "pair_rows: rows %d (even), row_bytes %lld (multiple of 64)", rows, row_bytes);
  if (rows == 0) return 0;
  const long long chunks = (long long)rows * (row_bytes / 16);
  hipLaunchKernelGGL(pair_rows_kernel<false>, dim3(grid_for(chunks)), dim3(256), 0, s, (const uint4*)in, (uint4*)out, chunks, (int)(row_bytes / 16));
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_act(const void* x, void* y, int rows, int cols, long long ld_in, long long ld_out, int mode, hipStream_t s) {
  FLUXMI_REQUIRE(cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0, "act: cols/ld must be multiples of 8");
  if (rows == 0 || cols == 0) return 0;
  hipLaunchKernelGGL(act_kernel, dim3(grid_for((long long)rows * (cols / 8))), dim3(256), 0, s, (const u16*)x, (u16*)y, rows, cols, ld_in, ld_out, mode);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_gate_residual(const void* x, const void* y, const void* gate, void* out, int B, int L, int H, long long ldx,
                           long long ldy, long long ldo, long long gate_bstride, hipStream_t s) {
  FLUXMI_REQUIRE(H % 8 == 0, "gate_residual: H must be a multiple of 8");
  if ((long long)B * L * H == 0) return 0;
  hipLaunchKernelGGL(gate_residual_kernel, dim3(grid_for((long long)B * L * (H / 8))), dim3(256), 0, s, (const u16*)x, (const u16*)y,
                     (const u16*)gate, (u16*)out, B, L, H, ldx, ldy, ldo, gate_bstride);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_add(const void* a, const void* b, void* z, long long n, hipStream_t s) {
  FLUXMI_REQUIRE(n % 8 == 0, "add: n must be a multiple of 8");
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, (const u16*)a, (const u16*)b, (u16*)z, n);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_build_qlut(const float* scale, int fmt, int act, void* lut, hipStream_t s) {
  FLUXMI_REQUIRE(scale && lut && act >= 0 && act <= 2, "build_quant_lut: bad arguments");
  if (fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL(build_qlut_kernel<FLUXMI_FMT_E5M2>, dim3(256), dim3(256), 0, s, scale, act, (unsigned char*)lut);
  else hipLaunchKernelGGL(build_qlut_kernel<FLUXMI_FMT_E4M3>, dim3(256), dim3(256), 0, s, scale, act, (unsigned char*)lut);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
int fluxmi_k_add_bcast(const void* a, const void* b, void* z, long long rows, int nb, int cols, hipStream_t s) {
  FLUXMI_REQUIRE(cols % 8 == 0 && nb >= 1, "add_bcast: cols must be a multiple of 8");
  if (rows * cols == 0) return 0;
  hipLaunchKernelGGL(add_bcast_kernel, dim3(grid_for(rows * (cols / 8))), dim3(256), 0, s, (const u16*)a, (const u16*)b, (u16*)z, rows, nb, cols);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
int fluxmi_k_select_step(const void* table, const int* step, const int* step0, void* dst, long long bytes, hipStream_t s) {
  FLUXMI_REQUIRE(bytes % 16 == 0, "select_step: size must be a multiple of 16 bytes");
  if (bytes == 0) return 0;
  hipLaunchKernelGGL(select_step_kernel, dim3(grid_for(bytes / 16)), dim3(256), 0, s, (const uint4*)table, step, step0, (uint4*)dst, bytes / 16);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_timestep_embedding(const void* t, const float* freqs, void* out, int B, int half, float time_factor, hipStream_t s) {
  if (B * half == 0) return 0;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((B * half + 127) / 128), dim3(128), 0, s, (const u16*)t, freqs, (u16*)out, B, half, time_factor);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_rope_table(const void* ids, const float* omega, const int* axis, void* pe, long long rows, int n_axes, int pairs, hipStream_t s) {
  if (rows * pairs == 0) return 0;
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((rows * pairs + 255) / 256)), dim3(256), 0, s, (const u16*)ids, omega, axis, (u16*)pe, rows, n_axes, pairs);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_euler(void* img, const void* pred, const float* dts, const int* step, long long n, hipStream_t s) {
  FLUXMI_REQUIRE(n % 8 == 0, "euler: n must be a multiple of 8");
  if (n == 0) return 0;
  hipLaunchKernelGGL(euler_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, (u16*)img, (const u16*)pred, dts, step, n);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
int fluxmi_k_set_timestep(void* t_vec, const float* ts, const int* step, int B, hipStream_t s) {
  FLUXMI_REQUIRE(B <= 64, "set_timestep: batch %d > 64", B);
  hipLaunchKernelGGL(set_timestep_kernel, dim3(1), dim3(64), 0, s, (u16*)t_vec, ts, step, B);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
// eight one-lane workgroups (the dispatcher puts block b on XCD b % 8): out[3 b .. 3 b + 2] = {XCC id, shader-clock counter, 100 MHz
// real-time counter}.  Two samples around a region, paired by XCC id (the shader-clock counters of different XCDs are not aligned),
// give the average shader clock the chip sustained over it.
__global__ void clock_sample_kernel(unsigned long long* out) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;  // HW_REG_XCC_ID, bits [3:0]
  out[3 * blockIdx.x + 0] = xcc;
  out[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
  out[3 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
}
int fluxmi_k_clock_sample(unsigned long long* out, hipStream_t s) {
  hipLaunchKernelGGL(clock_sample_kernel, dim3(8), dim3(1), 0, s, out);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
int fluxmi_k_advance_step(int* step, hipStream_t s) {
  hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, s, step);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
int fluxmi_k_timestep_rows(void* t_rows, const float* ts, int step0, int B, int R, hipStream_t s) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(timestep_rows_kernel, dim3((R + 255) / 256), dim3(256), 0, s, (u16*)t_rows, ts, step0, B, R);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
int fluxmi_k_fill_bf16(void* dst, float v, int n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(fill_bf16_kernel, dim3((n + 63) / 64), dim3(64), 0, s, (u16*)dst, v, n);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
