// fluxmi -- grouped F8Linear / Linear GEMM for gfx950 (CDNA4).
//
// Replaces torch._scaled_mm as called by F8Linear.forward (reference float8_quantize.py:284-292):
//     out[M,N] = bf16( (A_e5m2[M,K] . W_e4m3[N,K]^T) * (1/in_scale)*(1/w_scale) + bias )
// and the surrounding eager elementwise ops (GELU / gate*y+x / next-layer quantise) as epilogues.
//
// Design (MI355X-first):
//  * fp8 runs on the MX-scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4 with all E8M0 block scales = 2^0
//    (0x7F): the only opcode family that reaches the ~5 PF dense fp8 rate on gfx950; the real
//    per-tensor scales are applied in the epilogue.  bf16 layers use v_mfma_f32_32x32x16_bf16 through
//    the same byte-level pipeline (a K-step is 128 BYTES of every row for both dtypes).
//  * operands are swapped: MFMA "A" = weight rows (n), MFMA "B" = activation rows (m), so that a lane
//    ends up with 4 consecutive output columns n of one row m -> packed 8 B (bf16) / 4 B (fp8) stores
//    and per-lane bias/gate vectors.
//  * global -> LDS by LDS-DMA (global_load_lds, 16 B/lane), double buffered, one barrier per K-step.
//    The LDS image is lane-linear, so the bank-conflict XOR swizzle (16-B slot ^= (row>>1)&7) is
//    applied on the per-lane SOURCE address and again on the ds_read_b128 address.
//  * 1-D grid, XCD-aware bijective remap + grouped (GROUP_M) rasterisation so the 32 tiles resident
//    on one XCD share A/W panels through that XCD's private 4 MiB L2.
//  * grouped launch: up to 16 independent problems (txt/img streams x batch) sharing N,K in one grid.
#include "gemm_epilogue.h"

namespace {

// Per-wave epilogue over its TM x TN grid of 32x32 accumulator tiles.  Lane (l31, hi) owns row
// m = mrow0 + 32*i and columns n = ncol0 + 32*j + 8*g4 + [0,4)  (C/D layout of the 32x32 MFMA with
// weights as the A operand: i_row = (reg&3) + 8*(reg>>2) + 4*hi -> n, column = lane&31 -> m).
template <int EPI, int FMT, int TM, int TN>
__device__ __forceinline__ void tile_epilogue(const FluxmiGemmGroup& G, v16f (&acc)[TM][TN], float s, float qs,
                                              int mrow0, int ncol0, int M) {
  // bias (and gate) words of all TN*4 column groups in ONE batch of loads: inside the loops each load sat behind its own null check, in
  // its own basic block, and was waited for on the spot (TN*4 exposed L2 round trips per tile)
  uint2 braw[TN][4], graw[TN][4];
  const bool has_bias = G.bias != nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int n = ncol0 + j * 32 + g4 * 8;
      braw[j][g4] = has_bias ? *(const uint2*)((const u16*)G.bias + n) : make_uint2(0, 0);
      graw[j][g4] = EPI == FLUXMI_EPI_GATE_RESID ? *(const uint2*)((const u16*)G.gate + n) : make_uint2(0, 0);
    }
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      asm volatile("" : "+v"(braw[j][g4].x), "+v"(braw[j][g4].y));
      if constexpr (EPI == FLUXMI_EPI_GATE_RESID) asm volatile("" : "+v"(graw[j][g4].x), "+v"(graw[j][g4].y));
    }
  auto un4 = [](uint2 v, float* o) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  };
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int n = ncol0 + j * 32 + g4 * 8;
      float bias[4], gate[4];
      un4(braw[j][g4], bias);
      un4(graw[j][g4], gate);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = mrow0 + i * 32;
        if (m < M) {
          float h[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = rbf(fmaf(acc[i][j][g4 * 4 + e], s, bias[e]));
          epilogue<EPI, FMT, 4>(G, qs, m, n, h, gate);
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// tiled MFMA kernel
// -------------------------------------------------------------------------------------------------
// CONV (round 6): implicit 3x3 convolution -- the A rows are output pixels and each 128-byte K-step (64 bf16 channels of ONE tap) is gathered from
// the NHWC input by the LDS-DMA's own address computation (FluxmiGemmParams.conv), zero-padded through a zero page: the [pixels, 9 C] patch
// matrix of fluxmi_im2col3x3 (2.4 - 4.8 GB per 1024^2 convolution of the FLUX VAE, written and read back) never exists.  Same K order and the
// same MFMAs as the GEMM on the explicit patch matrix: identical bits.                     reference modules/autoencoder.py:55-120 (Conv2d 3x3)
template <int BM, int BN, int WM, int WN, bool FP8, int ACT_FMT, bool CONV = false>
__global__ void __launch_bounds__(WM* WN * 64) gemm_tile_kernel(const FluxmiGemmParams P) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 MFMA tiles per wave
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
  constexpr int IA = (BM * 8) / NT, IW = (BN * 8) / NT;  // 16-B slots per thread per K-step
  constexpr int EB = FP8 ? 1 : 2;
  static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile/threads mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;

  // ---- block -> (problem, m-tile, n-tile) ------------------------------------------------------
  const int tiles_n = P.N / BN;
  const int nblk = P.tiles_m_total * tiles_n;
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
  const int nk = (P.K * EB) / 128;

  // ---- per-thread LDS-DMA source pointers (swizzle on the source side) -------------------------
  const unsigned char* srcA[IA];
  const unsigned char* srcW[IW];
  int cv_y[IA], cv_x[IA], cv_b[IA], cv_s[IA];  // CONV: top-left source coordinate (before the tap offset), batch row base, 16-byte slot
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int p = tid + NT * i, row = p >> 3, slot = (p & 7) ^ ((row >> 1) & 7);
    const int gr = min(m0 + row, M - 1);
    if constexpr (CONV) {
      const int hw = P.conv.Ho * P.conv.Wo;
      const int b = gr / hw, rem = gr - b * hw, yo = rem / P.conv.Wo, xo = rem - yo * P.conv.Wo;
      cv_y[i] = yo * P.conv.stride - P.conv.pad; cv_x[i] = xo * P.conv.stride - P.conv.pad; cv_b[i] = b * P.conv.Hi; cv_s[i] = slot * 16;
    }
    // a_pairs (fp8 activations in the row-pair layout, fluxmi_gemm_group_t): the two 64-byte halves of this kernel's 128-byte K-step sit 128
    // bytes apart, consecutive K-steps 256
    srcA[i] = (FP8 && G.a_pairs) ? (const unsigned char*)G.A + f8_act_off(gr, G.lda, (slot >> 2) * 64 + (slot & 3) * 16, 1)
                                 : (const unsigned char*)G.A + ((long long)gr * G.lda) * EB + slot * 16;
  }
  const long long a_kstep = (FP8 && G.a_pairs) ? 256 : 128;
#pragma unroll
  for (int i = 0; i < IW; ++i) {
    const int p = tid + NT * i, row = p >> 3, slot = (p & 7) ^ ((row >> 1) & 7);
    srcW[i] = (const unsigned char*)G.W + ((long long)(n0 + row) * P.K) * EB + slot * 16;
  }
  auto stage = [&](int buf, int kt) {
    unsigned char* dA = smem + buf * STAGE + wave * 1024;
    unsigned char* dW = dA + A_BYTES;
    const long long koff = (long long)kt * 128;
    if constexpr (CONV) {
      // K-step kt = channels [cb * 64, cb * 64 + 64) of tap (dy, dx): wave-uniform; the source pixel and its validity are per row
      const int cpk = P.conv.C >> 6, tap = kt / cpk, cb = kt - tap * cpk, dy = tap / 3, dx = tap - dy * 3;
      const int Hv = P.conv.Hi << P.conv.rshift, Wv = P.conv.Wi << P.conv.rshift;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const int yy = cv_y[i] + dy, xx = cv_x[i] + dx;
        const bool ok = (unsigned)yy < (unsigned)Hv && (unsigned)xx < (unsigned)Wv;
        const long long pix = (long long)(cv_b[i] + (yy >> P.conv.rshift)) * P.conv.Wi + (xx >> P.conv.rshift);
        const unsigned char* src = ok ? (const unsigned char*)G.A + (pix * P.conv.C + cb * 64) * 2 + cv_s[i] : (const unsigned char*)P.conv.zeros + cv_s[i];
        glds16(src, dA + NT * 16 * i);
      }
    } else {
#pragma unroll
      for (int i = 0; i < IA; ++i) glds16(srcA[i] + kt * a_kstep, dA + NT * 16 * i);
    }
#pragma unroll
    for (int i = 0; i < IW; ++i) glds16(srcW[i] + koff, dW + NT * 16 * i);
  };

  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment row offsets (bytes) and swizzle keys
  int a_off[TM], a_sw[TM], w_off[TN], w_sw[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * WTM + i * 32 + l31;
    a_off[i] = r * 128; a_sw[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = wn * WTN + j * 32 + l31;
    w_off[j] = A_BYTES + r * 128; w_sw[j] = (r >> 1) & 7;
  }

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const unsigned char* sb = smem + (kt & 1) * STAGE;
    if constexpr (FP8) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int s0 = kk * 4 + hi * 2;
        v8i fa[TM], fw[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const v4i lo = *(const v4i*)(sb + a_off[i] + (((s0) ^ a_sw[i]) << 4));
          const v4i hi4 = *(const v4i*)(sb + a_off[i] + (((s0 + 1) ^ a_sw[i]) << 4));
          fa[i] = (v8i){lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const v4i lo = *(const v4i*)(sb + w_off[j] + (((s0) ^ w_sw[j]) << 4));
          const v4i hi4 = *(const v4i*)(sb + w_off[j] + (((s0 + 1) ^ w_sw[j]) << 4));
          fw[j] = (v8i){lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                fw[j], fa[i], acc[i][j], FLUXMI_FMT_E4M3 /*A = weights*/, ACT_FMT /*B = activations*/,
                0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // the K association of the 256x256 kernels (gemm_pp / gemm_w1 / gemm_persist: 64-byte K-steps, one MFMA over the 16-byte chunks
        // {0, 2} of a step, one over {1, 3}): MFMA kk covers chunks (kk >> 1) * 4 + (kk & 1) + {0, 2}.  With the natural order (chunks
        // 2 kk, 2 kk + 1) ~3e-4 of the bf16 outputs differed from theirs in the last bit, so a bf16 result followed the tile choice --
        // and through it the row count, i.e. the batch a sample rode in (profiles/r05_batch_invariance.txt).  Every bf16 tile config
        // now sums K in ONE order (tests/test_ops_gpu.py::test_bf16_tile_configs_are_bit_identical)
        const int s0 = (kk >> 1) * 4 + (kk & 1) + hi * 2;
        v8bf fa[TM], fw[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *(const v8bf*)(sb + a_off[i] + ((s0 ^ a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) fw[j] = *(const v8bf*)(sb + w_off[j] + ((s0 ^ w_sw[j]) << 4));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  const float s = load_scale(G.sa_recip) * load_scale(G.sb_recip);
  const float qs = G.q_scale ? *G.q_scale : 1.0f;
  const int mrow0 = m0 + wm * WTM + l31, ncol0 = n0 + wn * WTN + hi * 4;
  switch (P.epi) {
    case FLUXMI_EPI_BF16: tile_epilogue<FLUXMI_EPI_BF16, ACT_FMT, TM, TN>(G, acc, s, qs, mrow0, ncol0, M); break;
    case FLUXMI_EPI_GELU_QUANT: tile_epilogue<FLUXMI_EPI_GELU_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, mrow0, ncol0, M); break;
    case FLUXMI_EPI_GATE_RESID: tile_epilogue<FLUXMI_EPI_GATE_RESID, ACT_FMT, TM, TN>(G, acc, s, qs, mrow0, ncol0, M); break;
    case FLUXMI_EPI_SPLIT: tile_epilogue<FLUXMI_EPI_SPLIT, ACT_FMT, TM, TN>(G, acc, s, qs, mrow0, ncol0, M); break;
    case FLUXMI_EPI_QUANT: tile_epilogue<FLUXMI_EPI_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, mrow0, ncol0, M); break;
    case FLUXMI_EPI_SILU_QUANT: tile_epilogue<FLUXMI_EPI_SILU_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, mrow0, ncol0, M); break;
    default: break;
  }
}

// -------------------------------------------------------------------------------------------------
// generic kernel: any M,N,K (K % 16 == 0 for fp8, K % 8 == 0 for bf16).  One thread per output.
// Used for odd shapes (e.g. img_in K=64 when quantize_flow_embedder_layers) and as an on-device
// cross-check of the tiled kernel at full size.
// -------------------------------------------------------------------------------------------------
template <bool FP8, int ACT_FMT>
__global__ void __launch_bounds__(256) gemm_generic_kernel(const FluxmiGemmParams P) {
  const int gi = blockIdx.z;
  const FluxmiGemmGroup& G = P.g[gi];
  const int n = blockIdx.x * 16 + (threadIdx.x & 15);
  const int m = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (m >= G.M || n >= P.N) return;
  float acc = 0.f;
  if constexpr (FP8) {
    const unsigned char* a = (const unsigned char*)G.A + (long long)m * G.lda;
    const unsigned char* w = (const unsigned char*)G.W + (long long)n * P.K;
    for (int k = 0; k < P.K; k += 16) {
      const uint4 av = *(const uint4*)(a + k), wv = *(const uint4*)(w + k);
      const unsigned aw[4] = {av.x, av.y, av.z, av.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc = fmaf(fp8_to_f32<ACT_FMT>(aw[q], b), fp8_to_f32<FLUXMI_FMT_E4M3>(ww[q], b), acc);
    }
  } else {
    const u16* a = (const u16*)G.A + (long long)m * G.lda;
    const u16* w = (const u16*)G.W + (long long)n * P.K;
    for (int k = 0; k < P.K; k += 8) {
      float fa[8], fw[8];
      unpack8(*(const uint4*)(a + k), fa);
      unpack8(*(const uint4*)(w + k), fw);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc = fmaf(fa[q], fw[q], acc);
    }
  }
  const float s = load_scale(G.sa_recip) * load_scale(G.sb_recip);
  const float qs = G.q_scale ? *G.q_scale : 1.0f;
  float b = 0.f, g = 0.f;
  if (G.bias) load_bf<1>(G.bias, n, &b);
  if (P.epi == FLUXMI_EPI_GATE_RESID) load_bf<1>(G.gate, n, &g);
  float h = rbf(fmaf(acc, s, b));
  switch (P.epi) {
    case FLUXMI_EPI_BF16: epilogue<FLUXMI_EPI_BF16, ACT_FMT, 1>(G, qs, m, n, &h, &g); break;
    case FLUXMI_EPI_GELU_QUANT: epilogue<FLUXMI_EPI_GELU_QUANT, ACT_FMT, 1>(G, qs, m, n, &h, &g); break;
    case FLUXMI_EPI_GATE_RESID: epilogue<FLUXMI_EPI_GATE_RESID, ACT_FMT, 1>(G, qs, m, n, &h, &g); break;
    case FLUXMI_EPI_SPLIT: epilogue<FLUXMI_EPI_SPLIT, ACT_FMT, 1>(G, qs, m, n, &h, &g); break;
    case FLUXMI_EPI_QUANT: epilogue<FLUXMI_EPI_QUANT, ACT_FMT, 1>(G, qs, m, n, &h, &g); break;
    case FLUXMI_EPI_SILU_QUANT: epilogue<FLUXMI_EPI_SILU_QUANT, ACT_FMT, 1>(G, qs, m, n, &h, &g); break;
    default: break;
  }
}

template <int BM, int BN, int WM, int WN, bool FP8, int ACT>
int launch_tile(FluxmiGemmParams& p, hipStream_t s) {
  int t = 0;
  for (int i = 0; i < p.n_groups; ++i) {
    p.g[i].m_tile_start = t;
    t += (p.g[i].M + BM - 1) / BM;
  }
  p.tiles_m_total = t;
  p.group_m = 8;
  constexpr int SMEM = 2 * (BM + BN) * 128;
  auto kern = gemm_tile_kernel<BM, BN, WM, WN, FP8, ACT>;
  static bool attr_set = false;
  if (!attr_set) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set = true;
  }
  const int nblk = t * (p.N / BN);
  if (nblk == 0) return 0;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(WM * WN * 64), SMEM, s, p);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

template <bool FP8, int ACT>
int launch_cfg(FluxmiGemmParams& p, int cfg, hipStream_t s);

int launch_conv_tile(FluxmiGemmParams& p, hipStream_t s) {
  constexpr int BM = 128, BN = 128;
  p.g[0].m_tile_start = 0;
  p.tiles_m_total = (p.g[0].M + BM - 1) / BM;
  p.group_m = 8;
  constexpr int SMEM = 2 * (BM + BN) * 128;
  auto kern = gemm_tile_kernel<BM, BN, 2, 2, false, FLUXMI_FMT_E5M2, true>;
  static bool attr_set = false;
  if (!attr_set) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set = true;
  }
  const int nblk = p.tiles_m_total * (p.N / BN);
  if (nblk == 0) return 0;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), SMEM, s, p);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

template <bool FP8, int ACT>
int launch_cfg(FluxmiGemmParams& p, int cfg, hipStream_t s) {
  switch (cfg) {
    case 2: return launch_tile<128, 128, 2, 2, FP8, ACT>(p, s);  // two workgroups per CU: thin launches, N % 256 != 0
    case 15: return launch_tile<128, 64, 4, 1, FP8, ACT>(p, s);  // narrow outputs (LastLayer.linear: N = 64)
    default: fluxmi_set_error("gemm: unknown tile config %d", cfg); return 1;
  }
}

}  // namespace

// tile configs (the numbers are part of the C ABI, fluxmi_gemm_grouped): 2 = 128x128 and 15 = 128x64 double-buffered kernels of this
// file, 13 = 256x256 ping-pong ring (gemm_pp.hip), 16 = 256x256 one wave per SIMD (gemm_w1.hip), 18 = persistent ping-pong
// (gemm_persist.hip; 19 = its timing build), 100 = generic.  The other numbers
// belonged to kernel generations that were measured slower and removed in round 3 (profiles/r01_kernel_sweep.txt, r02_gemm_ab.txt).
int fluxmi_launch_gemm_pp(FluxmiGemmParams& p, int is_fp8, int act_fmt, int cfg, hipStream_t s);
int fluxmi_launch_gemm_w1(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s);

int fluxmi_gemm_tile_bn(int cfg) { return cfg == 2 ? 128 : cfg == 15 ? 64 : (cfg == 13 || cfg == 16 || cfg == 17 || cfg == 18 || cfg == 19 || cfg == 20 || cfg == 21) ? 256 : 0; }
int fluxmi_gemm_tile_bm(int cfg) { return (cfg == 2 || cfg == 15) ? 128 : cfg == 17 ? 192 : cfg == 20 ? 224 : cfg == 21 ? 160 : (cfg == 13 || cfg == 16 || cfg == 18 || cfg == 19) ? 256 : 0; }

int fluxmi_gemm_tile_ok(int N, int K, int is_fp8, int cfg) {
  const int bn = fluxmi_gemm_tile_bn(cfg);
  if (!bn) return 0;
  const int kb = K * (is_fp8 ? 1 : 2);
  const int kstep = (cfg == 16 || cfg == 17 || cfg == 18 || cfg == 19 || cfg == 20 || cfg == 21) ? 256 : cfg == 13 ? 64 : 128;
  if ((cfg == 18 || cfg == 19) && kb < 512) return 0;
  return (N % bn == 0) && (kb % kstep == 0) && kb >= kstep;
}

int fluxmi_launch_gemm(FluxmiGemmParams& p, int is_fp8, int act_fmt, int cfg, hipStream_t s) {
  FLUXMI_REQUIRE(p.n_groups >= 1 && p.n_groups <= FLUXMI_MAX_GROUPS, "gemm: n_groups=%d out of range", p.n_groups);
  FLUXMI_REQUIRE(fluxmi_gemm_tile_ok(p.N, p.K, is_fp8, cfg), "gemm: shape N=%d K=%d not tileable with cfg %d", p.N, p.K, cfg);
  if (p.epi == FLUXMI_EPI_SPLIT)
    FLUXMI_REQUIRE(p.g[0].split_n % fluxmi_gemm_tile_bn(cfg) == 0, "gemm: split_n=%d must be a multiple of the N tile", p.g[0].split_n);
  for (int i = 0; i < p.n_groups; ++i)
    if (p.g[i].vt_out || p.g[i].k_out)
      FLUXMI_REQUIRE(cfg == 13 || cfg == 16 || cfg == 18 || cfg == 19, "gemm: fused K / V^T outputs exist only in the 256x256 tile configs (got %d)", cfg);
  // activations in the row-pair layout (fluxmi_gemm_group_t.a_pairs / c8_pairs): every group of a launch or none; fp8, dense even rows
  for (int i = 0; i < p.n_groups; ++i) {
    const FluxmiGemmGroup& g = p.g[i];
    FLUXMI_REQUIRE((g.a_pairs != 0) == (p.g[0].a_pairs != 0) && (g.c8_pairs != 0) == (p.g[0].c8_pairs != 0),
                   "gemm: a_pairs / c8_pairs must be the same for every group of a launch");
    if (g.a_pairs)
      FLUXMI_REQUIRE(is_fp8 && g.lda == p.K && g.M % 2 == 0 && p.K % 64 == 0,
                     "gemm: a_pairs needs fp8 operands, dense rows (lda == K, K %% 64 == 0) and an even M (group %d: M=%d lda=%lld K=%d cfg=%d)", i, g.M, g.lda, p.K, cfg);
    if (g.c8_pairs) {
      const bool split = p.epi == FLUXMI_EPI_SPLIT;
      FLUXMI_REQUIRE(p.epi == FLUXMI_EPI_GELU_QUANT || p.epi == FLUXMI_EPI_QUANT || p.epi == FLUXMI_EPI_SILU_QUANT || split,
                     "gemm: c8_pairs applies to the quantising epilogues only (epilogue %d)", p.epi);
      FLUXMI_REQUIRE(g.M % 2 == 0 && (split ? (g.ldc2 % 64 == 0 && g.c2_col0 % 64 == 0 && g.split_n % 64 == 0) : g.ldc % 64 == 0),
                     "gemm: c8_pairs needs an even M and 64-byte aligned rows / column offsets (group %d)", i);
    }
  }
  if (cfg == 18 || cfg == 19) return fluxmi_launch_gemm_persist(p, is_fp8, act_fmt, cfg == 19, s);
  if (cfg == 16) return fluxmi_launch_gemm_w1(p, is_fp8, act_fmt, s);
  if (cfg == 17) return fluxmi_launch_gemm_w1_192(p, is_fp8, act_fmt, s);
  if (cfg == 20) return fluxmi_launch_gemm_w1_224(p, is_fp8, act_fmt, s);
  if (cfg == 21) return fluxmi_launch_gemm_w1_160(p, is_fp8, act_fmt, s);
  if (cfg == 13) return fluxmi_launch_gemm_pp(p, is_fp8, act_fmt, cfg, s);
  if (is_fp8) {
    if (act_fmt == FLUXMI_FMT_E5M2) return launch_cfg<true, FLUXMI_FMT_E5M2>(p, cfg, s);
    return launch_cfg<true, FLUXMI_FMT_E4M3>(p, cfg, s);
  }
  // bf16 operands; act_fmt only selects the fp8 format of quantising epilogues
  if (act_fmt == FLUXMI_FMT_E5M2) return launch_cfg<false, FLUXMI_FMT_E5M2>(p, cfg, s);
  return launch_cfg<false, FLUXMI_FMT_E4M3>(p, cfg, s);
}

int fluxmi_launch_gemm_conv(FluxmiGemmParams& p, hipStream_t s) {
  FLUXMI_REQUIRE(p.n_groups == 1 && p.conv.C > 0 && p.conv.C % 64 == 0 && p.K == 9 * p.conv.C && p.N % 128 == 0 && p.conv.zeros,
                 "conv3x3 (implicit): one group, C %% 64 == 0, K == 9 C, N %% 128 == 0 (C=%d N=%d K=%d)", p.conv.C, p.N, p.K);
  FLUXMI_REQUIRE(p.epi == FLUXMI_EPI_BF16 || p.epi == FLUXMI_EPI_GATE_RESID, "conv3x3 (implicit): plain or residual epilogue");
  FLUXMI_REQUIRE(!p.g[0].a_pairs && !p.g[0].c8_pairs && !p.g[0].vt_out && !p.g[0].k_out, "conv3x3 (implicit): no fused layouts");
  return launch_conv_tile(p, s);
}

int fluxmi_launch_gemm_generic(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s) {
  FLUXMI_REQUIRE(p.n_groups >= 1 && p.n_groups <= FLUXMI_MAX_GROUPS, "gemm: n_groups=%d out of range", p.n_groups);
  FLUXMI_REQUIRE(p.K % (is_fp8 ? 16 : 8) == 0, "gemm: K=%d must be a multiple of %d", p.K, is_fp8 ? 16 : 8);
  for (int i = 0; i < p.n_groups; ++i)
    FLUXMI_REQUIRE(!p.g[i].a_pairs && !p.g[i].c8_pairs, "gemm: the generic kernel does not read or write the row-pair activation layout (a_pairs / c8_pairs)");
  int maxM = 0;
  for (int i = 0; i < p.n_groups; ++i) maxM = p.g[i].M > maxM ? p.g[i].M : maxM;
  if (maxM == 0 || p.N == 0) return 0;
  dim3 grid((p.N + 15) / 16, (maxM + 15) / 16, p.n_groups);
  if (is_fp8) {
    if (act_fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL((gemm_generic_kernel<true, FLUXMI_FMT_E5M2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_generic_kernel<true, FLUXMI_FMT_E4M3>), grid, dim3(256), 0, s, p);
  } else {
    if (act_fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL((gemm_generic_kernel<false, FLUXMI_FMT_E5M2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_generic_kernel<false, FLUXMI_FMT_E4M3>), grid, dim3(256), 0, s, p);
  }
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
