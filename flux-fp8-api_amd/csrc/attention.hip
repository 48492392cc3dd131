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
#include "attention_common.h"

namespace {

// NW waves x 32 query rows per workgroup, RD-deep K and V^T rings.  <4, 2>: 64 KiB LDS, two workgroups per CU (the two waves of a
// SIMD belong to different workgroups and drift freely); <8, 4>: 128 KiB, one workgroup per CU, every K/V tile is fetched once per
// CU instead of twice (the LDS-DMA refill is 17 % of the <4, 2> kernel, FLUXMI_ATTN_ABL=1) and two tiles stay in flight across
// each barrier (counted vmcnt).
template <int FMT, int NW, int RD>
__global__ void __launch_bounds__(NW * 64, 2) attention_kernel(const AttnArgs a) {
  constexpr int QB = NW * 32;            // query rows per workgroup
  constexpr int LPW = 16 / NW;           // 1 KiB LDS-DMA pieces per wave per 16 KiB tile
  constexpr int VRING = RD * K_BYTES;    // byte offset of the V^T ring
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  // 1-D grid, XCD-aware: the blocks that run on one XCD (bid % 8) get a contiguous range of (batch, head, q-block) ids, i.e.
  // whole heads, so the K / V^T tiles of a head (2.4 MB at L = 4608) are fetched into that XCD's 4 MiB L2 once and shared by
  // its 36 q-blocks.  With the plain (q-block, head) grid every XCD touched ~14 heads at a time and each block streamed its
  // K/V from the fabric: 864 x 2.4 MB = 2 GB per launch = the HBM roofline, not the MFMA one.
  const int nqb = (a.L + QB - 1) / QB;
  const int lid = xcd_remap(blockIdx.x, nqb * a.H * a.B);
  const int bhid = lid / nqb;
  const int h = bhid % a.H, b = bhid / a.H;
  const int q0 = (lid - bhid * nqb) * QB + wave * 32;
  const int qrow = q0 + l31;
  const int qld = min(qrow, a.L - 1);
  const long long bh = (long long)b * a.H + h;

  v8bf qf[8];
  load_q_frags(a, b, h, qld, hi, qf);
  // ---- LDS-DMA sources: one buffer descriptor per operand (head base, SGPRs), a fixed 32-bit per-lane offset and the tile offset
  // in an SGPR -> `buffer_load_dwordx4 v, s[rsrc], s_off offen lds`: no address arithmetic in the tile loop, and rows past L of a
  // ragged last tile read as zero (hardware bounds check; their scores are masked anyway).
  const auto krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.K + bh * a.L * 128), 0, a.L * 256, 0x00020000);
  const auto vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.VT + bh * 128 * a.Lp), 0, 128 * a.Lp * 2, 0x00020000);
  unsigned k_off[LPW], v_off[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int p = tid + NW * 64 * i;
    k_off[i] = (unsigned)((p >> 4) * 256 + ((p & 15) ^ ((p >> 4) & 15)) * 16);
    const int d = p >> 3, vs = (p & 7) ^ ((d >> 1) & 7);
    v_off[i] = (unsigned)(d * a.Lp * 2 + vs * 16);
  }
  // LDS: K ring (RD x 16 KiB) then V^T ring (RD x 16 KiB).  Tile t lives in slot t % RD of its ring.
  auto stage_k = [&](int slot, int kv0) {
    unsigned char* dk = smem + slot * K_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < LPW; ++i)
      dma16(krsrc, dk + NW * 1024 * i, k_off[i], kv0 * 256);
  };
  auto stage_v = [&](int slot, int kv0) {
    unsigned char* dv = smem + VRING + slot * V_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < LPW; ++i)
      dma16(vrsrc, dv + NW * 1024 * i, v_off[i], kv0 * 2);
  };

  v16f o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f;  // running max of the RAW scores (the softmax scale is folded into the exp2 fma)
  v2f l2 = {0.f, 0.f};   // row sum, two partial sums (packed adds)
  const float c = a.scale_log2;

  // Fragment addresses are loop-invariant VGPRs (the XOR swizzle is per lane); ring slot, 32-row half and d-block go into the
  // ds_read immediate offset, so the tile loop carries no address arithmetic at all.
  unsigned kx[8], vx[4];
  {
    const int sw = l31 & 15, vsw = (l31 >> 1) & 7;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) kx[cc] = (unsigned)(l31 * 256 + (((cc * 2 + hi) ^ sw) << 4));
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) vx[ch] = (unsigned)(VRING + l31 * 128 + (((ch * 2 + hi) ^ vsw) << 4));
  }
  // S^T = K . Q^T for one 64-key tile.  The two 32-key halves ALTERNATE on the matrix pipe: consecutive MFMAs never share an
  // accumulator, so none waits for the previous one's result.
  // K fragments are fetched QK_PF chunks ahead into a rotating register set and the order is pinned (sched_barrier): left alone, the
  // compiler sinks every ds_read next to its MFMA to save registers and each MFMA pair then waits a full LDS round trip.
  constexpr int QK_PF = 2;
  auto qk = [&](int slot, v16f (&st)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
    const unsigned char* kbase = smem + slot * K_BYTES;
    v8bf ka[QK_PF + 1], kb[QK_PF + 1];
#pragma unroll
    for (int cc = 0; cc < QK_PF; ++cc) {
      ka[cc] = *(const v8bf*)(kbase + kx[cc]);
      kb[cc] = *(const v8bf*)(kbase + kx[cc] + 32 * 256);  // (32 + l31) & 15 == l31 & 15: same swizzle key
    }
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      if (cc + QK_PF < 8) {
        ka[(cc + QK_PF) % (QK_PF + 1)] = *(const v8bf*)(kbase + kx[cc + QK_PF]);
        kb[(cc + QK_PF) % (QK_PF + 1)] = *(const v8bf*)(kbase + kx[cc + QK_PF] + 32 * 256);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[cc % (QK_PF + 1)], qf[cc], st[0], 0, 0, 0);
      st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[cc % (QK_PF + 1)], qf[cc], st[1], 0, 0, 0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ragged last tile only: keys >= L get a score that exponentiates to 0
  auto mask_tile = [&](v16f (&st)[2], int kv0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        st[t][r] = key < a.L ? st[t][r] : -1e30f;
      }
  };
  // row max over the 64 keys of a tile: 32 in-lane values, then one exchange with lane ^ 32 (v_permlane32_swap, no LDS)
  auto row_max = [&](const v16f (&st)[2]) -> float {
    float mx = st[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
    const unsigned u = __float_as_uint(mx);
    const auto sw2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(sw2[0]), __uint_as_float(sw2[1]));
  };

  const int ntiles = (a.L + KT - 1) / KT;
  const bool ragged = (a.L % KT) != 0;

  // ---- prologue: K_0 .. K_{RD-1} and V_0 .. V_{RD-2} in flight (issue order K0 V0 K1 V1 ...); S_0 and its row max -----------------
  int n_issued = 0;
  // (written out per ring slot so that slot and offsets are compile-time constants)
#define ATTN_PROLOGUE_TILE(T)                                                         \
  if constexpr (RD > (T)) {                                                           \
    if ((T) < ntiles) { stage_k((T), (T) * KT); n_issued += LPW; }                    \
    if ((T) < RD - 1 && (T) < ntiles) { stage_v((T), (T) * KT); n_issued += LPW; }    \
  }
  ATTN_PROLOGUE_TILE(0) ATTN_PROLOGUE_TILE(1) ATTN_PROLOGUE_TILE(2) ATTN_PROLOGUE_TILE(3)
#undef ATTN_PROLOGUE_TILE
  // K_0 is the oldest piece: wait until at most (n_issued - LPW) loads remain (n_issued is wave-uniform and takes RD values)
  if (n_issued == (2 * RD - 1) * LPW) wait_vm<(2 * RD - 2) * LPW>();
  else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  // FLUXMI_ATTN_ABL & 4: static priority for the second-dispatched half of the workgroup (it loses VALU arbitration by age otherwise)
  if ((a.abl & 4) && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  v16f sa[2], sb[2];
  qk(0, sa);
  if (ragged && ntiles == 1) mask_tile(sa, 0);
  float mx = row_max(sa);

  // ---- main loop, software-pipelined inside the wave: while the matrix pipe runs S_{j+1} = K_{j+1} Q^T the VALU turns S_j into
  // P_j (exp2 / row sum / bf16), and while it runs O += V_j P_j the VALU reduces the row max of S_{j+1}.  The step is written for
  // a compile-time ring slot (the loop is unrolled by RD, S_j / S_{j+1} swap roles instead of being copied), packed f32 math
  // halves the VALU instruction count of the softmax: per tile the VALU issue time (4 cycles per wave64 instruction) was longer
  // than the 32 MFMAs (profiles/r01_attention_pmc.txt).
  auto step = [&](auto PAR, v16f (&cur)[2], v16f (&nxt)[2], int j) {
    constexpr int par = decltype(PAR)::value;      // = j % RD
    constexpr int nslot = (par + 1) % RD;          // slot of K_{j+1}
    // K_{j+1} and V_j must have landed (own pieces; the barrier extends that to every wave).  The RD-2 younger tile pairs stay in
    // flight; near the end fewer are outstanding, so drain.
    if (RD > 2 && j + RD - 1 < ntiles) wait_vm<(RD - 2) * 2 * LPW>(); else wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own LDS reads of step j-1 done before anyone refills those slots
    if (!(a.abl & 2)) __builtin_amdgcn_s_barrier();
    // The refills of this step (K_{j+RD} into the slot of K_j, V_{j+RD-1} into the slot of V_{j-1}; both were last read in step j-1)
    // are NOT issued here: an LDS-DMA piece costs 100-185 issue cycles next to a ds_read-dense segment and 25-60 in a VALU-only
    // gap (MI355X_MICROARCH.md), so K goes behind the QK^T block and V behind the softmax.
    // -- A: running max; once it has settled the O rescale is skipped exactly (wave-uniform branch)
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run)) {
      asm volatile("" ::: "memory");
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      l2 *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      m_run = m_new;
    }
    const float nmc = -m_run * c;
    const v2f c2 = {c, c}, nmc2 = {nmc, nmc};
    // -- B: S_{j+1} (matrix pipe)  ||  P_j = 2^(S_j*c - m*c), row sum, bf16 (VALU).  On the last step S_{j+1} is computed from
    //       a stale slot and never used (keeps the step branch-free).
    qk(nslot, nxt);
    if (!(a.abl & 1) && j + RD < ntiles) stage_k(par, (j + RD) * KT);
    v8bf pf[4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          v2f x = {cur[t][u * 8 + e], cur[t][u * 8 + e + 1]};
          x = __builtin_elementwise_fma(x, c2, nmc2);
          v2f p = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
          l2 += p;
          pf[t * 2 + u][e] = (bf16)p[0];
          pf[t * 2 + u][e + 1] = (bf16)p[1];
        }
    // (measured: both refills right behind the QK^T block -- where the compiler hoists this one too -- beat V between the PV halves or
    // after them by 1.5-2.5 %, and the old top-of-step placement by 4 %: there the matrix pipe was empty while the pieces issued)
    if (!(a.abl & 1) && j + RD - 1 < ntiles) stage_v((par + RD - 1) % RD, (j + RD - 1) * KT);
    // -- C: O^T += V_j^T . P_j^T (matrix pipe, four independent accumulators)  ||  row max of S_{j+1} (VALU)
    if (ragged && j + 2 == ntiles) mask_tile(nxt, (j + 1) * KT);  // rare wave-uniform branch, kept ahead of the overlapped region
    mx = row_max(nxt);
    {
      v8bf va[8], vb[8];
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int db = 0; db < 4; ++db) va[ch * 4 + db] = *(const v8bf*)(smem + vx[ch] + par * V_BYTES + db * 4096);
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          vb[ch * 4 + db] = *(const v8bf*)(smem + vx[ch + 2] + par * V_BYTES + db * 4096);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[ch * 4 + db], pf[ch], o[db], 0, 0, 0);
        }
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int db = 0; db < 4; ++db)
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[ch * 4 + db], pf[ch + 2], o[db], 0, 0, 0);
    }
  };
  {
    int j = 0;
    for (; j + RD <= ntiles; j += RD) {
      step(std::integral_constant<int, 0>{}, sa, sb, j);
      step(std::integral_constant<int, 1>{}, sb, sa, j + 1);
      if constexpr (RD == 4) {
        step(std::integral_constant<int, 2>{}, sa, sb, j + 2);
        step(std::integral_constant<int, 3>{}, sb, sa, j + 3);
      }
    }
    if (j < ntiles) { step(std::integral_constant<int, 0>{}, sa, sb, j); ++j; }
    if (RD == 4 && j < ntiles) { step(std::integral_constant<int, 1>{}, sb, sa, j); ++j; }
    if (RD == 4 && j < ntiles) { step(std::integral_constant<int, 2>{}, sa, sb, j); ++j; }
  }
  const float l_part = l2[0] + l2[1];

  // ---- epilogue ---------------------------------------------------------------------------------------
  const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
  const float inv = 1.0f / l_tot;
  store_o<FMT>(a, o, inv, b, h, qrow, hi);
}

}  // namespace

int fluxmi_k_attention(const void* Q, const void* K, const void* VT, void* out, long long ld_out, int col_off, int out_fp8,
                       const float* q_scale0, const float* q_scale1, int split, int B, int L, int Lp, int H, int fmt,
                       hipStream_t s, const void* qraw, long long ldq, const void* pe, const void* qn0, const void* qn1, int k_f16) {
  FLUXMI_REQUIRE(Q || (qraw && pe && qn0 && qn1 && ldq % 8 == 0), "attention: need Q, or raw q + pe + both q-norm scales (ld %% 8 == 0)");
  FLUXMI_REQUIRE(Lp % 64 == 0 && Lp >= L, "attention: Lp=%d must be a multiple of 64 and >= L=%d", Lp, L);
  FLUXMI_REQUIRE(!out_fp8 || (q_scale0 && q_scale1), "attention: fp8 output needs q_scale pointers");
  if (B * L * H == 0) return 0;
  AttnArgs a;
  a.Q = (const u16*)Q; a.K = (const u16*)K; a.VT = (const u16*)VT;
  a.qraw = (const u16*)qraw; a.ldq = ldq; a.pe = (const u16*)pe; a.qn[0] = (const u16*)qn0; a.qn[1] = (const u16*)qn1;
  a.out = out; a.ld_out = ld_out; a.col_off = col_off; a.out_fp8 = out_fp8;
  a.q_scale[0] = q_scale0; a.q_scale[1] = q_scale1; a.split = split;
  a.B = B; a.L = L; a.Lp = Lp; a.H = H;
  a.scale_log2 = 0.08838834764831845f * 1.4426950408889634f;  // 128^-0.5 * log2(e)
  a.k_f16 = k_f16;
  {
    const char* e = getenv("FLUXMI_ATTN_ABL");  // read per call (A/B probes flip it inside one process)
    a.abl = e ? atoi(e) : 0;
    // the regrouped fp8 epilogue stores 16 B per lane: rows that are not 16-byte aligned keep the 4-byte stores
    if (out_fp8 && ((((uintptr_t)out) | (uintptr_t)ld_out | (uintptr_t)col_off) & 15)) a.abl |= 8;
  }
  // round-2 pipeline (attention2.hip) by default; FLUXMI_ATTN_V=1 selects the round-1 kernel below (kept for A/B and as the
  // independently written cross-check of tests/test_ops_gpu.py::test_attention_v1_v2_agree)
  {
    const char* e = getenv("FLUXMI_ATTN_V");  // read per call (the tests compare the two kernels in one process)
    if (!(e && atoi(e) == 1)) return fluxmi_launch_attention2(a, fmt, s);
  }
  FLUXMI_REQUIRE(!k_f16, "attention: the round-1 kernel (FLUXMI_ATTN_V=1) takes bf16 K only");
  // 8-wave workgroups (256 query rows, K/V tiles shared by twice as many rows) by default; FLUXMI_ATTN_NW=4 selects the 4-wave
  // kernel with two workgroups per CU
  static int nw = 0;
  if (!nw) {
    const char* e = getenv("FLUXMI_ATTN_NW");
    nw = (e && atoi(e) == 4) ? 4 : 8;
  }
  static bool attr = false;
  if (!attr) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)attention_kernel<FLUXMI_FMT_E5M2, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * A_STAGE));
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)attention_kernel<FLUXMI_FMT_E4M3, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * A_STAGE));
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)attention_kernel<FLUXMI_FMT_E5M2, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * A_STAGE));
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)attention_kernel<FLUXMI_FMT_E4M3, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * A_STAGE));
    attr = true;
  }
  const int qb = nw * 32;
  const dim3 grid(((L + qb - 1) / qb) * H * B);
  if (nw == 4) {
    if (fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL((attention_kernel<FLUXMI_FMT_E5M2, 4, 2>), grid, dim3(256), 2 * A_STAGE, s, a);
    else hipLaunchKernelGGL((attention_kernel<FLUXMI_FMT_E4M3, 4, 2>), grid, dim3(256), 2 * A_STAGE, s, a);
  } else {
    if (fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL((attention_kernel<FLUXMI_FMT_E5M2, 8, 4>), grid, dim3(512), 4 * A_STAGE, s, a);
    else hipLaunchKernelGGL((attention_kernel<FLUXMI_FMT_E4M3, 8, 4>), grid, dim3(512), 4 * A_STAGE, s, a);
  }
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
