// fluxmi -- "ping-pong" GEMM: the deep-pipelined 256x256 tile kernel for the F8Linear GEMMs with K < 8192 bytes (gfx950).
//
// Same math, operands, grouping and epilogues as gemm.hip; different pipeline:
//  * K-step = 64 BYTES of every row (one v_mfma_scale_f32_32x32x64_f8f6f4 deep), 4-stage LDS ring (4 x 32 KiB = 128 KiB), filled by
//    LDS-DMA three K-steps ahead;
//  * counted `s_waitcnt vmcnt(N)` + raw s_barrier: the loads of the NEXT-next tile stay in flight across every barrier (hipcc's
//    __syncthreads would drain them), nothing but the tile needed next is waited for;
//  * epilogue through LDS: each wave parks its bf16 tile in the (now idle) ring, then every lane handles 8 consecutive columns of one
//    row: residual / gate / bias vectors are 16 B coalesced loads and the output leaves as full 128 B lines (the direct accumulator
//    layout gives 8 B pieces at a 6 KB stride).
// LDS rows are 64 B (4 x 16 B slots); bank-conflict-free ds_read_b128 needs slot ^= (row >> 2) & 3, applied on the DMA source address
// and on the read address.  (Rounds 1-2 also carried a lock-step ring kernel in five tile shapes, a two-workgroups-per-CU variant and
// timing-only ablations of it: measured slower everywhere -- profiles/r01_gemm_ablation*.txt, r02_gemm_ab.txt -- and removed in round 3.)
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#include "gemm_epilogue.h"

namespace {

// -------------------------------------------------------------------------------------------------------------------
// "ping-pong" kernel: 256x256 tile, 8 waves (2 x 4), 4-slot ring, ONE barrier per 64-byte K-step -- and the two waves
// that share a SIMD (wave w and w+4 = the two M-halves) run the K-step in OPPOSITE order:
//     group 0:  barrier | 8 MFMA (fragments already in registers) | LDS-DMA refill | read fragments of step k+1
//     group 1:  barrier | LDS-DMA refill | read fragments of step k | 8 MFMA
// so that on every SIMD one wave feeds the matrix pipe while the other sits in VMEM/LDS issue (an LDS-DMA instruction
// costs 60-180 cycles of issue, a K-step carries 4 per wave): measured on the lock-step ring the DMA issue alone was 19 %
// of the kernel and the LDS reads 8 % (profiles/r01_gemm_ablation.txt).  No double-buffered fragments, no register moves.
// LDS hazards with a single barrier: before barrier k every wave has waited for its own DMA of step k+1, so group 1 may
// read step k and group 0 step k+1 after it; the slot refilled after barrier k held step k-1, last read (by group 1)
// before that barrier.
// -------------------------------------------------------------------------------------------------------------------
// ESEL >= 0: compiled for ONE epilogue (see gemm_w1.hip), -1: run-time switch
// SPLITK: blockIdx.x = split * tiles + tile; the workgroup accumulates K-steps [split * nk / S, (split + 1) * nk / S) of its tile and
// stores the raw fp32 accumulators (no scale, no bias) into P.partial -- the small-M launches (M <= 512: schnell 256x256, the text
// encoders) have 24-96 tiles for 256 CUs and are weight-stream bound: a tile per workgroup leaves the stream to a tenth of the chip.
template <bool FP8, int ACT_FMT, int VAR, int ESEL, bool SPLITK = false>
__global__ void __launch_bounds__(512, 2) gemm_pp_kernel(const FluxmiGemmParams P) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NT = 512, TM = 4, TN = 2, NS = 4, D = 3;
  constexpr int WTM = 128, WTN = 64;
  constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64, STAGE = A_BYTES + W_BYTES;
  constexpr int IA = 2, IW = 2, LPT = 4;
  constexpr int EB = FP8 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;

  const int tiles_n = P.N / BN;
  const int nblk = P.tiles_m_total * tiles_n;
  if constexpr (!SPLITK) {
    if ((int)blockIdx.x >= nblk) {  // extra workgroups behind the tiles: weight prefetch for the launches that follow (FluxmiPrefetch)
      fluxmi_prefetch_ranges(P.pf, (int)blockIdx.x - nblk, (int)gridDim.x - nblk, tid, NT);
      return;
    }
  }
  const int split = SPLITK ? (int)blockIdx.x / nblk : 0;
  const int lid = xcd_remap(SPLITK ? (int)blockIdx.x % nblk : (int)blockIdx.x, nblk);
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
  const int nk_all = (P.K * EB) / 64;
  const int k_begin = SPLITK ? (int)(((long long)split * nk_all) / P.split_k) : 0;
  const int nk = SPLITK ? (int)(((long long)(split + 1) * nk_all) / P.split_k) - k_begin : nk_all;

  // VAR & 4: LDS-DMA through buffer descriptors (SGPR tile / K offsets, rows past M read as zero) instead of per-lane 64-bit
  // addresses (4 v_lshl_add_u64 per K-step and a clamped row index)
  constexpr bool BUF = (VAR & 4) != 0;
  const unsigned char* srcA[IA];
  const unsigned char* srcW[IW];
  unsigned a_voff[IA], w_voff[IW];
  const long long a_row_b = (long long)G.lda * EB, w_row_b = (long long)P.K * EB;
  const bool a_pairs = FP8 && !SPLITK && G.a_pairs != 0;
  const int a_kstep = a_pairs ? 128 : 64;
  // W in the row-pair layout when the caller has one (fluxmi_gemm_group_t.W_pairs, as in the persistent and the one-wave kernels)
  const bool w_pairs = G.W_pairs != nullptr;  // fp8 or bf16 (row bytes K * EB), split-K or not (this launch uses per-lane pointers: VAR 2)
  const int w_kstep = w_pairs ? 128 : 64;
  const unsigned char* w_base = (const unsigned char*)(w_pairs ? G.W_pairs : G.W);
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int p = tid + NT * i, row = p >> 2, slot = (p & 3) ^ ((row >> 2) & 3);
    const int gr = min(m0 + row, M - 1);
    // a_pairs: fp8 activations in the row-pair layout (fluxmi_gemm_group_t; never with split-K): K-steps of a row are 128 bytes apart
    srcA[i] = a_pairs ? (const unsigned char*)G.A + f8_act_off(gr, G.lda, slot * 16, 1)
                      : (const unsigned char*)G.A + ((long long)gr * G.lda) * EB + slot * 16 + (long long)k_begin * 64;
    a_voff[i] = a_pairs ? (unsigned)((row >> 1) * 2 * a_row_b + (row & 1) * 64 + slot * 16) : (unsigned)(row * a_row_b + slot * 16);
  }
#pragma unroll
  for (int i = 0; i < IW; ++i) {
    const int p = tid + NT * i, row = p >> 2, slot = (p & 3) ^ ((row >> 2) & 3);
    srcW[i] = w_pairs ? w_base + f8_act_off(n0 + row, w_row_b, slot * 16, 1) + (long long)k_begin * 128
                      : w_base + ((long long)(n0 + row) * P.K) * EB + slot * 16 + (long long)k_begin * 64;
    w_voff[i] = w_pairs ? (unsigned)((row >> 1) * 2 * w_row_b + (row & 1) * 64 + slot * 16) : (unsigned)(row * w_row_b + slot * 16);
  }
  const __amdgpu_buffer_rsrc_t ars = make_rsrc(G.A, (unsigned)min((long long)M * a_row_b, 0xffffffffLL));
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(w_base, (unsigned)min((long long)P.N * w_row_b, 0xffffffffLL));
  const unsigned a_soff0 = uni_u32((unsigned)(m0 * a_row_b)), w_soff0 = uni_u32((unsigned)(n0 * w_row_b));
  auto stage = [&](int kt, int slot) {
    unsigned char* dA = smem + slot * STAGE + wave * 1024;
    unsigned char* dW = dA + A_BYTES;
    if constexpr (BUF) {
#pragma unroll
      for (int i = 0; i < IA; ++i) dma16_buf(ars, dA + NT * 16 * i, a_voff[i], a_soff0 + kt * a_kstep);
#pragma unroll
      for (int i = 0; i < IW; ++i) dma16_buf(wrs, dW + NT * 16 * i, w_voff[i], w_soff0 + kt * w_kstep);
    } else {
      const long long koff = (long long)kt * 64;
#pragma unroll
      for (int i = 0; i < IA; ++i) glds16(srcA[i] + (long long)kt * a_kstep, dA + NT * 16 * i);
#pragma unroll
      for (int i = 0; i < IW; ++i) glds16(srcW[i] + (long long)kt * w_kstep, dW + NT * 16 * i);
    }
  };

  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int a_lo, a_hi, w_lo, w_hi;
  {
    const int ra = wm * WTM + l31, ka = (ra >> 2) & 3;
    a_lo = ra * 64 + (((hi * 2) ^ ka) << 4);
    a_hi = ra * 64 + (((hi * 2 + 1) ^ ka) << 4);
    const int rw = wn * WTN + l31, kw = (rw >> 2) & 3;
    w_lo = A_BYTES + rw * 64 + (((hi * 2) ^ kw) << 4);
    w_hi = A_BYTES + rw * 64 + (((hi * 2 + 1) ^ kw) << 4);
  }
  v8i fa[TM], fw[TN];
  auto read_frags = [&](int slot) {
    const unsigned char* sb = smem + slot * STAGE;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const v4i lo = *(const v4i*)(sb + j * 2048 + w_lo), h4 = *(const v4i*)(sb + j * 2048 + w_hi);
      fw[j] = (v8i){lo[0], lo[1], lo[2], lo[3], h4[0], h4[1], h4[2], h4[3]};
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const v4i lo = *(const v4i*)(sb + i * 2048 + a_lo), h4 = *(const v4i*)(sb + i * 2048 + a_hi);
      fa[i] = (v8i){lo[0], lo[1], lo[2], lo[3], h4[0], h4[1], h4[2], h4[3]};
    }
  };
  auto mma_all = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (FP8) {
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[j], fa[i], acc[i][j], FLUXMI_FMT_E4M3, ACT_FMT, 0, 0x7f7f7f7f, 0,
                                                                    0x7f7f7f7f);
        } else {
          const v4i alo = (v4i){fa[i][0], fa[i][1], fa[i][2], fa[i][3]}, ahi = (v4i){fa[i][4], fa[i][5], fa[i][6], fa[i][7]};
          const v4i wlo = (v4i){fw[j][0], fw[j][1], fw[j][2], fw[j][3]}, whi = (v4i){fw[j][4], fw[j][5], fw[j][6], fw[j][7]};
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, wlo), __builtin_bit_cast(v8bf, alo), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, whi), __builtin_bit_cast(v8bf, ahi), acc[i][j], 0, 0, 0);
        }
      }
  };
  auto fence = []() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: three K-steps in flight ---------------------------------------------------------------------------
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  if (nk > 2) stage(2, 2);
  if (nk > 2) wait_vmcnt<2 * LPT>(); else if (nk > 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  if (wm == 0) {
    read_frags(0);
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 2 < nk) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      fence();
      if (VAR & 1) __builtin_amdgcn_s_setprio(1);
      mma_all();
      if (VAR & 1) __builtin_amdgcn_s_setprio(0);
      fence();
      const int rslot = slot == 0 ? NS - 1 : slot - 1;
      if (kt + D < nk) stage(kt + D, rslot);
      fence();
      slot = slot + 1 == NS ? 0 : slot + 1;
      read_frags(kt + 1 < nk ? slot : (slot == 0 ? NS - 1 : slot - 1));  // last step: harmless re-read of the same slot
      fence();
    }
  } else {
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 2 < nk) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      fence();
      const int rslot = slot == 0 ? NS - 1 : slot - 1;
      if (VAR & 8) {
        read_frags(slot);
      } else if (VAR & 2) {
        read_frags(slot);
        fence();
        if (kt + D < nk) stage(kt + D, rslot);
      } else {
        if (kt + D < nk) stage(kt + D, rslot);
        fence();
        read_frags(slot);
      }
      fence();
      if (VAR & 1) __builtin_amdgcn_s_setprio(1);
      mma_all();
      if (VAR & 1) __builtin_amdgcn_s_setprio(0);
      fence();
      if (VAR & 8) {  // refill behind this wave's own MFMA block (an LDS-DMA piece issues in 60 cycles there, 100-185 beside ds_reads)
        if (kt + D < nk) stage(kt + D, rslot);
        fence();
      }
      slot = slot + 1 == NS ? 0 : slot + 1;
    }
  }

  if constexpr (SPLITK) {
    // raw fp32 partial tile: lane = one row, 4 consecutive columns per register group (16-byte stores)
    float* base = P.partial + ((size_t)split * P.tiles_m_total * BM + (size_t)tm * BM) * P.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ml = wm * WTM + i * 32 + l31;
      if (m0 + ml < M) {
        float* rowp = base + (size_t)ml * P.N + n0 + wn * WTN + hi * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            *(v4f*)(rowp + j * 32 + g4 * 8) = (v4f){acc[i][j][g4 * 4 + 0], acc[i][j][g4 * 4 + 1], acc[i][j][g4 * 4 + 2], acc[i][j][g4 * 4 + 3]};
      }
    }
    return;
  }
  // ---- epilogue ----------------------------------------------------------------------------------------------------
  const float s = load_scale_u(G.sa_recip) * load_scale_u(G.sb_recip);
  const float qs = load_scale_u(G.q_scale);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave is done reading the ring before it is reused as epilogue scratch
  unsigned char* wbuf = smem + wave * (WTM * WTN * 2);
  const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;
  float* xchg = (float*)(smem + NS * STAGE);
  if constexpr (ESEL >= 0) {
    lds_epilogue<ESEL, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane, xchg, wave, smem);
  } else {
    switch (P.epi) {
      case FLUXMI_EPI_BF16: lds_epilogue<FLUXMI_EPI_BF16, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane, xchg, wave, smem); break;
      case FLUXMI_EPI_GELU_QUANT: lds_epilogue<FLUXMI_EPI_GELU_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane, xchg, wave, smem); break;
      case FLUXMI_EPI_GATE_RESID: lds_epilogue<FLUXMI_EPI_GATE_RESID, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane, xchg, wave, smem); break;
      case FLUXMI_EPI_SPLIT: lds_epilogue<FLUXMI_EPI_SPLIT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane, xchg, wave, smem); break;
      case FLUXMI_EPI_QUANT: lds_epilogue<FLUXMI_EPI_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane, xchg, wave, smem); break;
      case FLUXMI_EPI_SILU_QUANT: lds_epilogue<FLUXMI_EPI_SILU_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane, xchg, wave, smem); break;
      default: break;
    }
  }
}

template <bool FP8, int ACT, int VAR, int ESEL = -1>
int launch_pp(FluxmiGemmParams& p, hipStream_t s) {
  constexpr int BM = 256, BN = 256;
  int t = 0;
  for (int i = 0; i < p.n_groups; ++i) {
    p.g[i].m_tile_start = t;
    t += (p.g[i].M + BM - 1) / BM;
  }
  p.tiles_m_total = t;
  p.group_m = 8;
  constexpr int SMEM = 4 * (BM + BN) * 64 + 8 * 128 * 4;  // ring + the K-epilogue's row-sum exchange (8 waves x 128 rows)
  auto kern = gemm_pp_kernel<FP8, ACT, VAR, ESEL>;
  static bool attr_set = false;
  if (!attr_set) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set = true;
  }
  const int nblk = t * (p.N / BN);
  if (nblk == 0) return 0;
  const int idle = (256 - nblk % 256) % 256;
  const int extra = p.pf.n > 0 ? std::min(p.pf.wgs, idle) : 0;
  if (!extra) p.pf.n = 0;
  hipLaunchKernelGGL(kern, dim3(nblk + extra), dim3(512), SMEM, s, p);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

// ---- split-K: reduce + epilogue pass ---------------------------------------------------------------------------------
// one thread = 8 consecutive columns of one row: sum of the splits in ascending order (deterministic), h = bf16(acc * sa*sb + bias), then
// C = h (FLUXMI_EPI_BF16) or resid + bf16(gate * h) (FLUXMI_EPI_GATE_RESID) -- the same rounding points as lds_epilogue.
template <int EPI>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const FluxmiGemmParams P) {
  const int prow = blockIdx.x;  // padded row: tile row * 256 + local row
  const int tm = prow >> 8;
  int gi = 0;
  for (int i = 1; i < P.n_groups; ++i) gi = (tm >= P.g[i].m_tile_start) ? i : gi;
  const FluxmiGemmGroup& G = P.g[gi];
  const int m = prow - G.m_tile_start * 256;
  if (m >= G.M) return;
  const float s = load_scale(G.sa_recip) * load_scale(G.sb_recip);
  const size_t split_stride = (size_t)P.tiles_m_total * 256 * P.N;
  for (int c = threadIdx.x; c < P.N / 8; c += 256) {
    const int n = c * 8;
    const float* src = P.partial + (size_t)prow * P.N + n;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < P.split_k; ++sp) {
      const v4f a = *(const v4f*)(src + sp * split_stride), b = *(const v4f*)(src + sp * split_stride + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[e] += a[e]; acc[4 + e] += b[e]; }
    }
    float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, h[8];
    if (G.bias) unpack8(*(const uint4*)((const u16*)G.bias + n), bias);
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = rbf(fmaf(acc[e], s, bias[e]));
    if constexpr (EPI == FLUXMI_EPI_GATE_RESID) {
      float r[8], g[8], o[8];
      unpack8(*(const uint4*)((const u16*)G.resid + (long long)m * G.ldr + n), r);
      unpack8(*(const uint4*)((const u16*)G.gate + n), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = r[e] + rbf(g[e] * h[e]);
      *(uint4*)((u16*)G.C + (long long)m * G.ldc + n) = pack8(o);
    } else {
      *(uint4*)((u16*)G.C + (long long)m * G.ldc + n) = pack8(h);
    }
  }
}

// Split-K scratch: one 256 MiB fp32 buffer per (device, stream) that ever launches a split-K GEMM -- the partial tiles live from the
// GEMM pass to the reduce pass of the SAME stream, so two streams must never share one -- created under a mutex on the first use, and
// never under stream capture (a hipMalloc there would invalidate the capture: the launch is refused instead; run the shape once
// eagerly first, as fluxmi_engine_denoise does with its warm step).
constexpr size_t SPLITK_WS_BYTES = FLUXMI_SPLITK_WS_BYTES;
thread_local float* t_splitk_override = nullptr;
std::mutex g_splitk_mu;
std::map<std::pair<int, hipStream_t>, float*> g_splitk_ws;
int splitk_workspace(hipStream_t s, float** out) {
  if (t_splitk_override) { *out = t_splitk_override; return 0; }
  int dev = 0;
  FLUXMI_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_splitk_mu);
  auto it = g_splitk_ws.find({dev, s});
  if (it == g_splitk_ws.end()) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s) FLUXMI_CHECK_HIP(hipStreamIsCapturing(s, &cap));
    FLUXMI_REQUIRE(cap == hipStreamCaptureStatusNone, "gemm split-K: first use on this stream happens under stream capture (no scratch yet): "
                   "run the launch once eagerly before capturing");
    float* p = nullptr;
    FLUXMI_CHECK_HIP(hipMalloc((void**)&p, SPLITK_WS_BYTES));
    it = g_splitk_ws.emplace(std::make_pair(dev, s), p).first;
  }
  *out = it->second;
  return 0;
}

template <bool FP8, int ACT>
int launch_pp_splitk(FluxmiGemmParams& p, int split_k, hipStream_t s) {
  constexpr int BM = 256, BN = 256;
  int t = 0;
  for (int i = 0; i < p.n_groups; ++i) {
    p.g[i].m_tile_start = t;
    t += (p.g[i].M + BM - 1) / BM;
  }
  p.tiles_m_total = t;
  p.group_m = 8;
  const int nblk = t * (p.N / BN);
  if (nblk == 0) return 0;
  FLUXMI_REQUIRE(p.epi == FLUXMI_EPI_BF16 || p.epi == FLUXMI_EPI_GATE_RESID, "gemm split-K: bf16 / gate*y+x epilogues only (got %d)", p.epi);
  const size_t need = (size_t)split_k * t * BM * p.N * sizeof(float);
  FLUXMI_REQUIRE(split_k >= 2 && need <= SPLITK_WS_BYTES, "gemm split-K: %d splits of %d x %d need %zu bytes of scratch (have %zu)", split_k, t * BM, p.N,
                 need, SPLITK_WS_BYTES);
  FLUXMI_REQUIRE(split_k <= (p.K * (FP8 ? 1 : 2)) / 64, "gemm split-K: %d splits for %d K-steps", split_k, (p.K * (FP8 ? 1 : 2)) / 64);
  float* ws = nullptr;
  FLUXMI_TRY(splitk_workspace(s, &ws));
  p.split_k = split_k;
  p.partial = ws;
  p.pf.n = 0;
  constexpr int SMEM = 4 * (BM + BN) * 64 + 8 * 128 * 4;
  auto kern = gemm_pp_kernel<FP8, ACT, 2, -1, true>;
  static bool attr_set = false;
  if (!attr_set) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nblk * split_k), dim3(512), SMEM, s, p);
  FLUXMI_LAUNCH_CHECK();
  if (p.epi == FLUXMI_EPI_GATE_RESID) hipLaunchKernelGGL(splitk_reduce_kernel<FLUXMI_EPI_GATE_RESID>, dim3(t * BM), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(splitk_reduce_kernel<FLUXMI_EPI_BF16>, dim3(t * BM), dim3(256), 0, s, p);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

template <bool FP8, int ACT>
int launch_pp_cfg(FluxmiGemmParams& p, hipStream_t s) {
  // the hot epilogues get a kernel compiled for them alone (fp8 x e5m2: the calibrated step; bf16: VAE / text encoders / bf16 flow)
  if constexpr (ACT == FLUXMI_FMT_E5M2) {
    const int esel = fluxmi_tuning().gemm_esel;  // 0: the run-time-switch kernel for every epilogue (A/B)
    if (esel) switch (p.epi) {
      case FLUXMI_EPI_BF16: return launch_pp<FP8, ACT, 2, FLUXMI_EPI_BF16>(p, s);
      case FLUXMI_EPI_GATE_RESID: return launch_pp<FP8, ACT, 2, FLUXMI_EPI_GATE_RESID>(p, s);
      case FLUXMI_EPI_SPLIT: if constexpr (FP8) return launch_pp<FP8, ACT, 2, FLUXMI_EPI_SPLIT>(p, s); else break;
      case FLUXMI_EPI_GELU_QUANT: if constexpr (FP8) return launch_pp<FP8, ACT, 2, FLUXMI_EPI_GELU_QUANT>(p, s); else break;
      default: break;
    }
  }
  return launch_pp<FP8, ACT, 2>(p, s);
}

}  // namespace

void fluxmi_set_splitk_scratch(float* p) { t_splitk_override = p; }

int fluxmi_launch_gemm_splitk(FluxmiGemmParams& p, int is_fp8, int act_fmt, int split_k, hipStream_t s) {
  FLUXMI_REQUIRE(fluxmi_gemm_tile_ok(p.N, p.K, is_fp8, 13), "gemm split-K: shape N=%d K=%d not tileable", p.N, p.K);
  for (int i = 0; i < p.n_groups; ++i)
    FLUXMI_REQUIRE(!p.g[i].vt_out && !p.g[i].k_out && !p.g[i].a_pairs && !p.g[i].c8_pairs, "gemm split-K: no fused K / V^T outputs, no row-pair activation layout");
  if (is_fp8) {
    if (act_fmt == FLUXMI_FMT_E5M2) return launch_pp_splitk<true, FLUXMI_FMT_E5M2>(p, split_k, s);
    return launch_pp_splitk<true, FLUXMI_FMT_E4M3>(p, split_k, s);
  }
  return launch_pp_splitk<false, FLUXMI_FMT_E5M2>(p, split_k, s);
}

// config 13 = 256x256 ping-pong ring (8 waves)
int fluxmi_launch_gemm_pp(FluxmiGemmParams& p, int is_fp8, int act_fmt, int cfg, hipStream_t s) {
  p.pf = fluxmi_take_prefetch();
  if (!fluxmi_tuning().prefetch) p.pf.n = 0;
  FLUXMI_REQUIRE(cfg == 13, "gemm_pp: unknown tile config %d", cfg);
  if (is_fp8) {
    if (act_fmt == FLUXMI_FMT_E5M2) return launch_pp_cfg<true, FLUXMI_FMT_E5M2>(p, s);
    return launch_pp_cfg<true, FLUXMI_FMT_E4M3>(p, s);
  }
  if (act_fmt == FLUXMI_FMT_E5M2) return launch_pp_cfg<false, FLUXMI_FMT_E5M2>(p, s);
  return launch_pp_cfg<false, FLUXMI_FMT_E4M3>(p, s);
}
