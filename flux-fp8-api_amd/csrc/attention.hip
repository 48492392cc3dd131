// fluxmi -- flash-attention forward for Flux joint attention (bf16, head_dim 128, non-causal), gfx950.
//
// Replaces F.scaled_dot_product_attention + transpose/reshape (reference flux_model.py:41-45) and,
// in fused mode, the fp8 quantise of the consumer F8Linear (float8_quantize.py:274-276).
//
// Structure (wave64 / MFMA-first):
//  * workgroup = 4 waves x 32 query rows; KV tiles of 64 keys; K tile [64][128] and V^T tile [128][64]
//    arrive by LDS-DMA into a 2-deep ring (one barrier per tile), XOR-swizzled via the source address.
//  * "swapped" QK^T: S^T = K . Q^T on v_mfma_f32_32x32x16_bf16, so a lane owns ONE query row
//    (col = lane&31) and 16 keys per 32x32 tile -> row max/sum are in-lane plus one lane^32 exchange.
//  * the P^T operand of O^T += V^T . P^T is the S^T accumulator itself, converted to bf16 in place:
//    the k-slot <-> key mapping that results (key = (e&3) + 8*(e>>2) + 4*hi inside each 16-key group)
//    is pre-applied to V by qkv_rope.hip, so no cross-lane traffic and no transpose read is needed.
//  * online softmax in the exp2 domain, fp32 accumulators, P rounded to bf16 for the PV MFMA.
#include "common.h"
#include "fluxmi_internal.h"

namespace {

struct AttnArgs {
  const u16* Q; const u16* K; const u16* VT;
  void* out; long long ld_out; int col_off; int out_fp8;
  const float* q_scale[2]; int split;
  int B, L, Lp, H;
  float scale_log2;
};

constexpr int KT = 64;                 // keys per tile
constexpr int K_BYTES = KT * 256;      // 16 KiB
constexpr int V_BYTES = 128 * KT * 2;  // 16 KiB
constexpr int A_STAGE = K_BYTES + V_BYTES;

template <int FMT>
__global__ void __launch_bounds__(256, 2) attention_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = q0 + l31;
  const int qld = min(qrow, a.L - 1);
  const long long bh = (long long)b * a.H + h;

  // ---- Q fragments (MFMA B operand): 8 x (8 bf16) ------------------------------------------------
  v8bf qf[8];
  {
    const u16* qp = a.Q + (bh * a.L + qld) * 128 + hi * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) qf[c] = *(const v8bf*)(qp + c * 16);
  }
  // ---- LDS-DMA source addresses -------------------------------------------------------------------
  const unsigned char* kbase = (const unsigned char*)(a.K + bh * a.L * 128);
  const unsigned char* vbase = (const unsigned char*)(a.VT + bh * 128 * a.Lp);
  int k_row[4], k_slot[4];
  const unsigned char* v_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + 256 * i;
    k_row[i] = p >> 4; k_slot[i] = ((p & 15) ^ ((p >> 4) & 15)) * 16;
    const int d = p >> 3, vs = (p & 7) ^ ((d >> 1) & 7);
    v_src[i] = vbase + (long long)d * a.Lp * 2 + vs * 16;
  }
  auto stage = [&](int buf, int kv0) {
    unsigned char* dk = smem + buf * A_STAGE + wave * 1024;
    unsigned char* dv = dk + K_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = min(kv0 + k_row[i], a.L - 1);
      glds16(kbase + (long long)kr * 256 + k_slot[i], dk + 4096 * i);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(v_src[i] + (long long)kv0 * 2, dv + 4096 * i);
  };

  v16f o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_part = 0.f;

  const int ntiles = (a.L + KT - 1) / KT;
  stage(0, 0);
  for (int kt = 0; kt < ntiles; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < ntiles) stage((kt + 1) & 1, (kt + 1) * KT);
    const unsigned char* sk = smem + (kt & 1) * A_STAGE;
    const unsigned char* sv = sk + K_BYTES;

    // ---- S^T = K . Q^T : two 32-key tiles, two independent accumulator chains -----------------------
    // All 8 K fragments of tile 0 are in flight before the first MFMA; tile 1's fragments are fetched
    // under tile 0's MFMAs (the MFMA chain on one accumulator needs no wait between links).
    v16f st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
    {
      const unsigned char* row0 = sk + l31 * 256;
      const unsigned char* row1 = row0 + 32 * 256;  // (32 + l31) & 15 == l31 & 15: same swizzle key
      const int sw = l31 & 15;
      v8bf ka[8], kb[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) ka[c] = *(const v8bf*)(row0 + (((c * 2 + hi) ^ sw) << 4));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        kb[c] = *(const v8bf*)(row1 + (((c * 2 + hi) ^ sw) << 4));
        st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[c], qf[c], st[0], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[c], qf[c], st[1], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    // ---- online softmax (log2 domain) ------------------------------------------------------------
    const int kv0 = kt * KT;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] *= a.scale_log2;
    if (kv0 + KT > a.L) {  // ragged last tile only: mask keys >= L (kept a real branch, not 64 selects per tile)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          st[t][r] = key < a.L ? st[t][r] : -1e30f;
        }
    }
    float mx = st[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run)) {  // wave-uniform; once the running max has settled the O rescale is skipped exactly
      asm volatile("" ::: "memory");
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_part *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      m_run = m_new;
    }
    float psum = 0.f;
    v8bf pf[4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float p = __builtin_amdgcn_exp2f(st[t][u * 8 + e] - m_run);
          psum += p;
          pf[t * 2 + u][e] = (bf16)p;
        }
    l_part += psum;
    // ---- O^T += V^T . P^T : four independent accumulators, fragments fetched two key-chunks ahead -----
    {
      const unsigned char* vrow[4];
      int vsw[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int d = db * 32 + l31;
        vrow[db] = sv + d * 128;
        vsw[db] = (d >> 1) & 7;
      }
      v8bf va[8], vb[8];
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int db = 0; db < 4; ++db) va[ch * 4 + db] = *(const v8bf*)(vrow[db] + (((ch * 2 + hi) ^ vsw[db]) << 4));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          vb[ch * 4 + db] = *(const v8bf*)(vrow[db] + ((((ch + 2) * 2 + hi) ^ vsw[db]) << 4));
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[ch * 4 + db], pf[ch], o[db], 0, 0, 0);
        }
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int db = 0; db < 4; ++db)
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[ch * 4 + db], pf[ch + 2], o[db], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------------
  const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qrow < a.L) {
    const long long orow = ((long long)b * a.L + qrow) * a.ld_out + a.col_off + h * 128;
    if (a.out_fp8) {
      const float qs = *a.q_scale[qrow < a.split ? 0 : 1];
      unsigned char* op = (unsigned char*)a.out + orow;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + g * 8 + hi * 4;
          *(unsigned*)(op + d) = cvt4_fp8<FMT>(q_prepare<FMT>(rbf(o[db][g * 4 + 0] * inv), qs), q_prepare<FMT>(rbf(o[db][g * 4 + 1] * inv), qs),
                                               q_prepare<FMT>(rbf(o[db][g * 4 + 2] * inv), qs), q_prepare<FMT>(rbf(o[db][g * 4 + 3] * inv), qs));
        }
    } else {
      u16* op = (u16*)a.out + orow;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + g * 8 + hi * 4;
          uint2 v;
          v.x = pack_bf2(o[db][g * 4 + 0] * inv, o[db][g * 4 + 1] * inv);
          v.y = pack_bf2(o[db][g * 4 + 2] * inv, o[db][g * 4 + 3] * inv);
          *(uint2*)(op + d) = v;
        }
    }
  }
}

}  // namespace

int fluxmi_k_attention(const void* Q, const void* K, const void* VT, void* out, long long ld_out, int col_off, int out_fp8,
                       const float* q_scale0, const float* q_scale1, int split, int B, int L, int Lp, int H, int fmt,
                       hipStream_t s) {
  FLUXMI_REQUIRE(Lp % 64 == 0 && Lp >= L, "attention: Lp=%d must be a multiple of 64 and >= L=%d", Lp, L);
  FLUXMI_REQUIRE(!out_fp8 || (q_scale0 && q_scale1), "attention: fp8 output needs q_scale pointers");
  if (B * L * H == 0) return 0;
  AttnArgs a;
  a.Q = (const u16*)Q; a.K = (const u16*)K; a.VT = (const u16*)VT;
  a.out = out; a.ld_out = ld_out; a.col_off = col_off; a.out_fp8 = out_fp8;
  a.q_scale[0] = q_scale0; a.q_scale[1] = q_scale1; a.split = split;
  a.B = B; a.L = L; a.Lp = Lp; a.H = H;
  a.scale_log2 = 0.08838834764831845f * 1.4426950408889634f;  // 128^-0.5 * log2(e)
  constexpr int SMEM = 2 * A_STAGE;
  static bool attr = false;
  if (!attr) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)attention_kernel<FLUXMI_FMT_E5M2>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)attention_kernel<FLUXMI_FMT_E4M3>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr = true;
  }
  const dim3 grid((L + 127) / 128, H, B);
  if (fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL(attention_kernel<FLUXMI_FMT_E5M2>, grid, dim3(256), SMEM, s, a);
  else hipLaunchKernelGGL(attention_kernel<FLUXMI_FMT_E4M3>, grid, dim3(256), SMEM, s, a);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
