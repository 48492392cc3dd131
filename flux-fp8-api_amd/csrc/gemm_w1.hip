// fluxmi -- "w1" GEMM: 256x256 tile, FOUR waves, one per SIMD, each owning a 128x128 accumulator block (gfx950).
//
// Same math, operands, grouping and epilogues as gemm.hip / gemm_pp.hip.  Why another shape: on real (non-zero) data the fp8
// matrix pipe of MI355X is power-limited (tools/probes/mfma_rate.hip: 3.7 PF/s of MFMA-only work on random operands, 5.0 on
// zeros), so what a GEMM reaches is set by how much energy it spends NOT doing MFMAs.  A 128x128 wave tile reads 8 fragments
// per 16 MFMAs from LDS instead of 6 per 8 (-33 % LDS read traffic for the same flops), needs no partner wave (one wave per SIMD
// issues the MX MFMA back to back at the full rate, same probe), and its 256 accumulator + 128 fragment registers fit the
// 512-register budget of a single wave per SIMD.
//  * 4-slot LDS ring of 64-byte K-steps (4 x 32 KiB), refilled three steps ahead by LDS-DMA through buffer descriptors
//    (`buffer_load_dwordx4 ... offen lds`: fixed per-lane offset in a VGPR, tile / K offset in an SGPR -> no address VALU, rows
//    past M read as zero);
//  * one barrier per K-step, counted vmcnt (the refill of step k+2 stays in flight across it);
//  * the step is fully software-pipelined inside the wave: MFMA slot s of step k is followed by the LDS read of fragment half s of
//    step k+1 (second register set) and, every other slot, by one LDS-DMA piece of step k+3.  The loop is unrolled by four so
//    that ring slots are immediates;
//  * epilogue through LDS as in the ring kernels (the idle ring is the transposition scratch, 32 KiB per wave).
// Requires K*bytes % 256 == 0 (four K-steps per unrolled iteration).
//
// Round 5: TM_ = 3 instantiates the same kernel on 192 x 256 tiles (wave tiles 96 x 128; tile config 17) for the gate*y+x launches whose
// 256-row tiling leaves half the chip idle -- Flux-dev 768^2: M = 2816 gives 11 x 12 = 132 tiles for 256 CUs on mlp.2 / linear2 (9.7 of the
// step's 30 ms at 48 % occupancy); 15 x 12 = 180 tiles of three quarters the work run in one round as well.  Same K loop, same MFMAs in
// the same order: every output bit equals config 16's.
#include <stdlib.h>

#include <algorithm>

#include "gemm_epilogue.h"

namespace {

template <int TN, int TM> struct W1FragsT { v8i fw[TN]; v8i fa[TM]; };

// ESEL >= 0: the kernel is compiled for ONE epilogue (the hot ones get their own instantiation: with the run-time switch over six inlined
// epilogues hipcc allocates registers for all of them at once and spilled ~110 ACCUMULATOR pairs to scratch right after the K loop -- on every
// path, each reload behind an s_waitcnt vmcnt(0)); ESEL = -1 keeps the switch (cold epilogues)
// WM_ (round 6): waves along M.  2 = the 2 x 2 wave grid above (wave tile TM*32 x 128).  1 = the four waves side by side along N, each owning
// ALL TM*32 rows of the tile and 64 columns: tiles of 32 * TM rows for ANY TM -- tile config 20 = 224 x 256 (TM = 7) is the exact fit of the
// step's linear2 launch (M = 4608: 21 row tiles x 12 = 252 tiles = ONE round of the 256 CUs at 7/8 of a 256-row tile's work, instead of
// 216 tiles on 256 CUs); 14 MFMAs per K-step read 18 fragment halves (9/7 of the 2 x 2 grid's LDS bytes per MFMA).  Same K loop, same
// MFMAs in the same order per output: every bit equals config 16's.
template <bool FP8, int ACT_FMT, int ESEL, int TM_ = 4, int WM_ = 2>
__global__ void __launch_bounds__(256, 1) gemm_w1_kernel(const FluxmiGemmParams P) {
  constexpr int TM = TM_, TN = WM_ == 2 ? 4 : 2, BM = WM_ * TM * 32, BN = 256, NT = 256, NS = 4;
  constexpr int A_ROWS = ((BM + 63) / 64) * 64;  // LDS-DMA moves 64 rows per piece: a 224-row tile stages 256 (the last 32 are never read)
  constexpr int A_BYTES = A_ROWS * 64, W_BYTES = BN * 64, STAGE = A_BYTES + W_BYTES;
  constexpr int NA = A_ROWS / 64;  // LDS-DMA pieces of A per wave per K-step (64 rows x 4 chunks = 256 lanes each), 4 of W
  constexpr int LPT = NA + 4;   // pieces per wave per K-step
  constexpr int NFR = TN + TM;  // fragments per K-step: TN W tiles + TM A tiles, two 16-byte halves each
  constexpr int EB = FP8 ? 1 : 2;
  using W1Frags = W1FragsT<TN, TM>;
  static_assert(STAGE + (TM > TN ? TM : TN) * 2048 <= 65536, "ds_read immediates are 16 bits");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WM_ == 2 ? wave >> 1 : 0, wn = WM_ == 2 ? wave & 1 : wave;
  const int l31 = lane & 31, hi = lane >> 5;

  const int tiles_n = P.N / BN;
  const int nblk = P.tiles_m_total * tiles_n;
  if ((int)blockIdx.x >= nblk) {  // extra workgroups behind the tiles: weight prefetch for the launches that follow (FluxmiPrefetch)
    fluxmi_prefetch_ranges(P.pf, (int)blockIdx.x - nblk, (int)gridDim.x - nblk, tid, NT);
    return;
  }
  const int lid = xcd_remap(blockIdx.x, nblk);
  const int width = P.group_m * tiles_n;
  const int first_m = (lid / width) * P.group_m;
  const int gsz = min(P.tiles_m_total - first_m, P.group_m);
  const int tm = first_m + (lid % width) % gsz;
  const int tn = (lid % width) / gsz;
  int gi = 0;
  for (int i = 1; i < P.n_groups; ++i) gi = (tm >= P.g[i].m_tile_start) ? i : gi;
  const FluxmiGemmGroup& G = P.g[gi];
  const int M = G.M;
  const int m0 = (tm - G.m_tile_start) * BM;
  const int n0 = tn * BN;
  const int nk = (P.K * EB) / 64;

  // ---- LDS-DMA: descriptors (SGPRs), per-lane offsets (VGPRs, loop-invariant), tile offsets (SGPRs) ----------------------------
  const long long a_row_b = (long long)G.lda * EB, w_row_b = (long long)P.K * EB;
  const __amdgpu_buffer_rsrc_t ars = make_rsrc(G.A, (unsigned)min((long long)M * a_row_b, 0xffffffffLL));
  // W in the row-pair layout when the caller has one (fluxmi_gemm_group_t.W_pairs: the 64-byte K-steps of rows 2r, 2r + 1 share a 128-byte line,
  // consecutive K-steps of a pair are 128 bytes apart -- every L2 line of W crosses to the CU once per tile instead of twice)
  const bool w_pairs = uni_ptr((const u16*)G.W_pairs) != nullptr;
  const unsigned w_kstep = w_pairs ? 128u : 64u;
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(w_pairs ? G.W_pairs : G.W, (unsigned)min((long long)P.N * w_row_b, 0xffffffffLL));
  // A in the row-pair layout as well (fluxmi_gemm_group_t.a_pairs: the engine's fp8 activation buffers in fused mode; fp8 only, lda == K, M even)
  const bool a_pairs = FP8 && uni_u32((unsigned)G.a_pairs) != 0;
  const unsigned a_kstep = a_pairs ? 128u : 64u;
  unsigned a_voff[4], w_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + NT * i, row = p >> 2, slot = (p & 3) ^ ((row >> 2) & 3);
    a_voff[i] = a_pairs ? (unsigned)((row >> 1) * 2 * a_row_b + (row & 1) * 64 + slot * 16) : (unsigned)(row * a_row_b + slot * 16);  // (i < NA is used)
    w_voff[i] = w_pairs ? (unsigned)((row >> 1) * 2 * w_row_b + (row & 1) * 64 + slot * 16) : (unsigned)(row * w_row_b + slot * 16);
  }
  const unsigned a_soff0 = uni_u32((unsigned)(m0 * a_row_b)), w_soff0 = uni_u32((unsigned)(n0 * w_row_b));
  // piece q (0..LPT-1: NA of A, then 4 of W) of K-step kt into ring slot `slot`
  auto dma_piece = [&](int q, int slot, int kt) {
    unsigned char* d = smem + slot * STAGE + wave * 1024;
    if (q < NA) dma16_buf(ars, d + NT * 16 * q, a_voff[q], a_soff0 + kt * a_kstep);
    else dma16_buf(wrs, d + A_BYTES + NT * 16 * (q - NA), w_voff[q - NA], w_soff0 + kt * w_kstep);
  };

  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addresses: (row * 64 + swizzled 16-B slot) per operand half; ring slot and 32-row tile index are immediates.
  // ds_read immediates are 16 bits, so slots 2 and 3 use a second base (+64 KiB).
  unsigned a_lo[2], a_hi[2], w_lo[2], w_hi[2];
  {
    const int ra = wm * (TM * 32) + l31, ka = (ra >> 2) & 3;
    const int rw = wn * (TN * 32) + l31, kw = (rw >> 2) & 3;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      a_lo[h2] = (unsigned)(h2 * 2 * STAGE + ra * 64 + (((hi * 2) ^ ka) << 4));
      a_hi[h2] = (unsigned)(h2 * 2 * STAGE + ra * 64 + (((hi * 2 + 1) ^ ka) << 4));
      w_lo[h2] = (unsigned)(h2 * 2 * STAGE + A_BYTES + rw * 64 + (((hi * 2) ^ kw) << 4));
      w_hi[h2] = (unsigned)(h2 * 2 * STAGE + A_BYTES + rw * 64 + (((hi * 2 + 1) ^ kw) << 4));
    }
  }
  // half `hf` (0 = low 16 B, 1 = high 16 B) of fragment `f` (0..TN-1 = W tiles, TN..TN+TM-1 = A tiles) of ring slot `slot`
  auto read_half = [&](W1Frags& F, int slot, int f, int hf) {
    const int h2 = slot >> 1, imm = (slot & 1) * STAGE + (f < TN ? f : f - TN) * 2048;
    const unsigned base = f < TN ? (hf ? w_hi[h2] : w_lo[h2]) : (hf ? a_hi[h2] : a_lo[h2]);
    const v4i v = *(const v4i*)(smem + base + imm);
    v8i& dst = f < TN ? F.fw[f] : F.fa[f - TN];
    dst[hf * 4 + 0] = v[0]; dst[hf * 4 + 1] = v[1]; dst[hf * 4 + 2] = v[2]; dst[hf * 4 + 3] = v[3];
  };
  auto mma = [&](const W1Frags& F, int i, int j) {
    if constexpr (FP8) {
      acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F.fw[j], F.fa[i], acc[i][j], FLUXMI_FMT_E4M3, ACT_FMT, 0, 0x7f7f7f7f, 0,
                                                                0x7f7f7f7f);
    } else {
      const v4i alo = (v4i){F.fa[i][0], F.fa[i][1], F.fa[i][2], F.fa[i][3]}, ahi = (v4i){F.fa[i][4], F.fa[i][5], F.fa[i][6], F.fa[i][7]};
      const v4i wlo = (v4i){F.fw[j][0], F.fw[j][1], F.fw[j][2], F.fw[j][3]}, whi = (v4i){F.fw[j][4], F.fw[j][5], F.fw[j][6], F.fw[j][7]};
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, wlo), __builtin_bit_cast(v8bf, alo), acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, whi), __builtin_bit_cast(v8bf, ahi), acc[i][j], 0, 0, 0);
    }
  };
  auto fence = []() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: K-steps 0..2 in flight, fragments of step 0 in registers ---------------------------------------------------------
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int q = 0; q < LPT; ++q) dma_piece(q, t, t);  // (K offsets past the row end fetch harmless bytes; nk >= 4 anyway)
  wait_vmcnt<2 * LPT>();
  __builtin_amdgcn_s_barrier();
  W1Frags fa_, fb_;
#pragma unroll
  for (int f = 0; f < NFR; ++f) {
    read_half(fa_, 0, f, 0);
    read_half(fa_, 0, f, 1);
  }

  // One K-step with compile-time ring slot S.  Entry: `cur` = fragments of step kt (LDS reads possibly still in flight), LDS-DMA of
  // steps kt+1, kt+2 in flight.  TM x 4 MFMA slots; behind MFMA slot s: LDS read s of the NEXT step's fragments, and behind every even
  // slot one LDS-DMA piece of step kt+3 (into the slot step kt-1 vacated).  TM = 3 has 12 slots for 14 fragment halves and 7 pieces: the
  // first two slots read two halves, the last slot issues the seventh piece.  Past the end of K the refill fetches don't-care bytes
  // into a slot nobody reads any more and the "next" fragments are re-read from a stale slot: no branches in the body.
  auto step = [&](auto SLOT, W1Frags& cur, W1Frags& nxt, int kt) {
    constexpr int S = decltype(SLOT)::value, SN = (S + 1) & 3, SR = (S + 3) & 3;
    wait_vmcnt<LPT>();  // step kt+1 landed (own pieces); step kt+2 stays in flight
    __builtin_amdgcn_s_barrier();
    fence();
    constexpr int NSLOT = TM * TN, NRD = 2 * NFR;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      mma(cur, s / TN, s % TN);
      fence();
      // next fragments: W halves first (needed by every MFMA row), then A
      read_half(nxt, SN, s >> 1, s & 1);
      if (s + NSLOT < NRD) read_half(nxt, SN, (s + NSLOT) >> 1, (s + NSLOT) & 1);
      if ((s & 1) == 0) dma_piece(s >> 1, SR, kt + 3);
      if (s == NSLOT - 1) {  // the pieces the even slots did not cover (TM = 3: the seventh; the 1 x 4 wave grid with TM = 5: the sixth and seventh)
#pragma unroll
        for (int q = (NSLOT + 1) >> 1; q < LPT; ++q) dma_piece(q, SR, kt + 3);
      }
      fence();
    }
  };
  for (int kt = 0; kt < nk; kt += 4) {
    step(std::integral_constant<int, 0>{}, fa_, fb_, kt);
    step(std::integral_constant<int, 1>{}, fb_, fa_, kt + 1);
    step(std::integral_constant<int, 2>{}, fa_, fb_, kt + 2);
    step(std::integral_constant<int, 3>{}, fb_, fa_, kt + 3);
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------------
  const float s = load_scale_u(G.sa_recip) * load_scale_u(G.sb_recip);
  const float qs = load_scale_u(G.q_scale);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the trailing don't-care refills must land before the ring is reused
  __builtin_amdgcn_s_barrier();
  unsigned char* wbuf = smem + wave * (TM * 32 * TN * 32 * 2);
  const int mw = m0 + wm * (TM * 32), nw = n0 + wn * (TN * 32);
  if constexpr (ESEL >= 0) {
    lds_epilogue<ESEL, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane);
  } else {
    switch (P.epi) {
      case FLUXMI_EPI_BF16: lds_epilogue<FLUXMI_EPI_BF16, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane); break;
      case FLUXMI_EPI_GELU_QUANT: lds_epilogue<FLUXMI_EPI_GELU_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane); break;
      case FLUXMI_EPI_GATE_RESID: lds_epilogue<FLUXMI_EPI_GATE_RESID, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane); break;
      case FLUXMI_EPI_SPLIT: lds_epilogue<FLUXMI_EPI_SPLIT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane); break;
      case FLUXMI_EPI_QUANT: lds_epilogue<FLUXMI_EPI_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane); break;
      case FLUXMI_EPI_SILU_QUANT: lds_epilogue<FLUXMI_EPI_SILU_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane); break;
      default: break;
    }
  }
}

template <bool FP8, int ACT, int ESEL = -1, int TM_ = 4, int WM_ = 2>
int launch_w1(FluxmiGemmParams& p, hipStream_t s) {
  constexpr int BM = 32 * WM_ * TM_, BN = 256, A_ROWS = ((BM + 63) / 64) * 64;
  int t = 0;
  for (int i = 0; i < p.n_groups; ++i) {
    p.g[i].m_tile_start = t;
    t += (p.g[i].M + BM - 1) / BM;
  }
  p.tiles_m_total = t;
  // bands of six row tiles: the step's launches of this kernel have 18 (three equal bands instead of 8 + 8 + 2); in-step A/B of 8 / 4 / 6:
  // 40.93 / 40.85 / 40.77 ms per step (profiles/r04_gemm_persist.txt section 11)
#ifndef W1_GROUP_M
#define W1_GROUP_M 6
#endif
  p.group_m = W1_GROUP_M;
  constexpr int SMEM = 4 * (A_ROWS + BN) * 64;
  auto kern = gemm_w1_kernel<FP8, ACT, ESEL, TM_, WM_>;
  static bool attr_set = false;
  if (!attr_set) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set = true;
  }
  const int nblk = t * (p.N / BN);
  if (nblk == 0) return 0;
  // a launch that leaves CUs idle in its last round carries the pending weight prefetch on them (fluxmi_internal.h, FluxmiPrefetch)
  // (the engine sizes pf.wgs for the 256-row tiling; a lower tile that fills the round leaves fewer idle CUs -- a handful of workgroups must not
  // be left with the whole prefetch: 4 workgroups reading 66 MB took 340 us and ended the launch.  Then the prefetch is dropped.)
  const int idle = (256 - nblk % 256) % 256;
  const int extra = (p.pf.n > 0 && idle >= p.pf.wgs) ? p.pf.wgs : 0;
  if (!extra) p.pf.n = 0;
  hipLaunchKernelGGL(kern, dim3(nblk + extra), dim3(256), SMEM, s, p);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// config 17 = the same kernel on 192 x 256 tiles, for the launches it exists for: fp8 x e5m2 operands with the gate*y+x epilogue (Flux-dev
// 768^2 mlp.2 / linear2) and bf16 operands with the plain or the gate*y+x epilogue (M = 512: Flux-schnell 256^2 linear1 runs 3 x 84 = 252
// tiles instead of 2 x 84 = 168 for 256 CUs; the text encoders)
int fluxmi_launch_gemm_w1_192(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s) {
  const bool f8_ok = is_fp8 && act_fmt == FLUXMI_FMT_E5M2 && p.epi == FLUXMI_EPI_GATE_RESID;
  const bool bf_ok = !is_fp8 && (p.epi == FLUXMI_EPI_BF16 || p.epi == FLUXMI_EPI_GATE_RESID);
  FLUXMI_REQUIRE(f8_ok || bf_ok, "gemm tile config 17 (192x256 one-wave-per-SIMD tiles): fp8 x e5m2 operands with the gate*y+x epilogue, or bf16 operands "
                 "with the plain / gate*y+x epilogue (fp8 %d, epi %d)", is_fp8, p.epi);
  for (int i = 0; i < p.n_groups; ++i)
    FLUXMI_REQUIRE(!p.g[i].vt_out && !p.g[i].k_out, "gemm tile config 17: no fused K / V^T outputs");
  p.pf = fluxmi_take_prefetch();
  if (!fluxmi_tuning().prefetch) p.pf.n = 0;
  const int eb = is_fp8 ? 1 : 2;
  for (int i = 0; i < p.n_groups; ++i)
    FLUXMI_REQUIRE((long long)p.g[i].M * p.g[i].lda * eb < (1LL << 32) && (long long)p.N * p.K * eb < (1LL << 32), "gemm_w1: operand larger than 4 GiB");
  if (is_fp8) return launch_w1<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_GATE_RESID, 3>(p, s);
  if (p.epi == FLUXMI_EPI_GATE_RESID) return launch_w1<false, FLUXMI_FMT_E5M2, FLUXMI_EPI_GATE_RESID, 3>(p, s);
  return launch_w1<false, FLUXMI_FMT_E5M2, FLUXMI_EPI_BF16, 3>(p, s);
}

// config 20 (round 6) = 224 x 256 tiles, the four waves side by side along N (wave tile 224 x 64): fp8 x e5m2 operands with the gate*y+x epilogue
// -- the exact fit of Flux-dev 1024^2 linear2 (21 x 12 = 252 tiles for 256 CUs)
int fluxmi_launch_gemm_w1_224(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s) {
  FLUXMI_REQUIRE(is_fp8 && act_fmt == FLUXMI_FMT_E5M2 && p.epi == FLUXMI_EPI_GATE_RESID,
                 "gemm tile config 20 (224x256 one-wave-per-SIMD tiles): fp8 x e5m2 operands with the gate*y+x epilogue (fp8 %d, epi %d)", is_fp8, p.epi);
  for (int i = 0; i < p.n_groups; ++i) {
    FLUXMI_REQUIRE(!p.g[i].vt_out && !p.g[i].k_out, "gemm tile config 20: no fused K / V^T outputs");
    FLUXMI_REQUIRE((long long)p.g[i].M * p.g[i].lda < (1LL << 32) && (long long)p.N * p.K < (1LL << 32), "gemm_w1: operand larger than 4 GiB");
  }
  p.pf = fluxmi_take_prefetch();
  if (!fluxmi_tuning().prefetch) p.pf.n = 0;
  return launch_w1<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_GATE_RESID, 7, 1>(p, s);
}
// config 21 = the same wave layout on 160 x 256 tiles (wave tile 160 x 64): Flux-dev 768^2 mlp.2 / linear2 (M = 2816: 18 x 12 = 216 tiles of 5/8 of the
// work instead of config 17's 180 tiles of 6/8)
int fluxmi_launch_gemm_w1_160(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s) {
  FLUXMI_REQUIRE(is_fp8 && act_fmt == FLUXMI_FMT_E5M2 && p.epi == FLUXMI_EPI_GATE_RESID,
                 "gemm tile config 21 (160x256 one-wave-per-SIMD tiles): fp8 x e5m2 operands with the gate*y+x epilogue (fp8 %d, epi %d)", is_fp8, p.epi);
  for (int i = 0; i < p.n_groups; ++i) {
    FLUXMI_REQUIRE(!p.g[i].vt_out && !p.g[i].k_out, "gemm tile config 21: no fused K / V^T outputs");
    FLUXMI_REQUIRE((long long)p.g[i].M * p.g[i].lda < (1LL << 32) && (long long)p.N * p.K < (1LL << 32), "gemm_w1: operand larger than 4 GiB");
  }
  p.pf = fluxmi_take_prefetch();
  if (!fluxmi_tuning().prefetch) p.pf.n = 0;
  return launch_w1<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_GATE_RESID, 5, 1>(p, s);
}

// config 16 = 256x256, one wave per SIMD
int fluxmi_launch_gemm_w1(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s) {
  p.pf = fluxmi_take_prefetch();
  if (!fluxmi_tuning().prefetch) p.pf.n = 0;
  // buffer descriptors address 4 GiB per operand
  for (int i = 0; i < p.n_groups; ++i)
    FLUXMI_REQUIRE((long long)p.g[i].M * p.g[i].lda * (is_fp8 ? 1 : 2) < (1LL << 32) && (long long)p.N * p.K * (is_fp8 ? 1 : 2) < (1LL << 32),
                   "gemm_w1: operand larger than 4 GiB");
  if (is_fp8) {
    if (act_fmt == FLUXMI_FMT_E5M2) {
      // the step's K >= 8192 launches (mlp.2, linear2) all end in gate*y + x
      const int esel = fluxmi_tuning().gemm_esel;  // 0: the run-time-switch kernel for every epilogue (A/B)
      if (esel && p.epi == FLUXMI_EPI_GATE_RESID) return launch_w1<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_GATE_RESID>(p, s);
      return launch_w1<true, FLUXMI_FMT_E5M2>(p, s);
    }
    return launch_w1<true, FLUXMI_FMT_E4M3>(p, s);
  }
  if (act_fmt == FLUXMI_FMT_E5M2) {  // bf16 operands (VAE convolutions, text encoders, bf16 flow): plain and residual epilogues specialised
    if (p.epi == FLUXMI_EPI_GATE_RESID) return launch_w1<false, FLUXMI_FMT_E5M2, FLUXMI_EPI_GATE_RESID>(p, s);
    if (p.epi == FLUXMI_EPI_BF16) return launch_w1<false, FLUXMI_FMT_E5M2, FLUXMI_EPI_BF16>(p, s);
    return launch_w1<false, FLUXMI_FMT_E5M2>(p, s);
  }
  return launch_w1<false, FLUXMI_FMT_E4M3>(p, s);
}
