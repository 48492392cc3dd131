// fluxmi -- GEMM epilogue helpers shared by the tile kernels (gemm.hip, gemm_pp.hip).
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
      store_q<FMT, CNT>(G.C2, f8_act_off(m, G.ldc2, G.c2_col0 + (n - G.split_n), G.c8_pairs), g, qs);
    }
  } else if constexpr (EPI == FLUXMI_EPI_BF16) {
    store_bf<CNT>(G.C, (long long)m * G.ldc + n, h);
  } else if constexpr (EPI == FLUXMI_EPI_GELU_QUANT) {
    float g[CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) g[j] = rbf(gelu_tanh_f(h[j]));
    store_q<FMT, CNT>(G.C, f8_act_off(m, G.ldc, n, G.c8_pairs), g, qs);
  } else if constexpr (EPI == FLUXMI_EPI_SILU_QUANT) {
    float g[CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) g[j] = rbf(silu_f(h[j]));
    store_q<FMT, CNT>(G.C, f8_act_off(m, G.ldc, n, G.c8_pairs), g, qs);
  } else if constexpr (EPI == FLUXMI_EPI_QUANT) {
    store_q<FMT, CNT>(G.C, f8_act_off(m, G.ldc, n, G.c8_pairs), h, qs);
  } else if constexpr (EPI == FLUXMI_EPI_GATE_RESID) {
    float r[CNT], o[CNT];
    load_bf<CNT>(G.resid, (long long)m * G.ldr + n, r);
#pragma unroll
    for (int j = 0; j < CNT; ++j) o[j] = r[j] + rbf(gate[j] * h[j]);
    store_bf<CNT>(G.C, (long long)m * G.ldc + n, o);
  }
}

__device__ __forceinline__ float load_scale(const float* p) { return p ? *p : 1.0f; }

// ---- LDS-DMA through buffer descriptors ------------------------------------------------------------------------------------
// `buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds`: fixed per-lane byte offset in a VGPR, tile / K offset in an SGPR, no address
// VALU in the K loop, bytes past `bytes` read as zero.  Kept out of the kernel templates (a device-only builtin in
// template-dependent code makes hipcc's host pass drop the kernel stub).  The descriptor inputs are wave-uniform in fact but reach
// the kernels through a dynamically indexed kernel-argument struct, i.e. in VGPRs: without readfirstlane hipcc wraps EVERY buffer
// op in a waterfall loop.
typedef __attribute__((address_space(3))) void* fluxmi_lds_ptr_t;
__device__ __forceinline__ unsigned uni_u32(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  const unsigned long long ub = ((unsigned long long)uni_u32((unsigned)(b >> 32)) << 32) | uni_u32((unsigned)b);
  return __builtin_amdgcn_make_buffer_rsrc((void*)ub, 0, uni_u32(bytes), 0x00020000);
}
// wave-uniform copies (SGPRs) of group fields that reach the kernel in VGPRs (see above): 64-bit pointers and strides held in VGPR pairs
// across the epilogue were what hipcc spilled to scratch and reloaded, one s_waitcnt vmcnt(0) each
// (returned as a GLOBAL address-space pointer: an integer round trip otherwise leaves a generic pointer, hipcc emits flat_load, and a
// flat load both counts in lgkmcnt -- every LDS wait then waits for it -- and forces vmcnt(0) before it is issued)
template <class T> using fluxmi_gptr = __attribute__((address_space(1))) T*;
typedef int fluxmi_v2i __attribute__((ext_vector_type(2)));  // HIP's uint2 / uint4 classes cannot be copied out of a qualified address space
template <class T> __device__ __forceinline__ fluxmi_gptr<T> uni_ptr(T* p) {
  const unsigned long long b = (unsigned long long)p;
  return (fluxmi_gptr<T>)(((unsigned long long)uni_u32((unsigned)(b >> 32)) << 32) | uni_u32((unsigned)b));
}
__device__ __forceinline__ long long uni_i64(long long v) {
  const unsigned long long b = (unsigned long long)v;
  return (long long)(((unsigned long long)uni_u32((unsigned)(b >> 32)) << 32) | uni_u32((unsigned)b));
}
// per-tensor scale through the scalar unit (the pointer is wave-uniform, the value constant for the kernel: s_load_dword instead of a
// vector load whose round trip the epilogue waited for)
__device__ __forceinline__ float load_scale_u(const float* p) {
  const unsigned long long b = (unsigned long long)p;
  const unsigned long long ub = ((unsigned long long)uni_u32((unsigned)(b >> 32)) << 32) | uni_u32((unsigned)b);
  if (!ub) return 1.0f;
  return *(__attribute__((address_space(4))) const float*)ub;
}
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (fluxmi_lds_ptr_t)lds, 16, voff, soff, 0, 0);
}

// ---- LDS-transposed epilogue shared by the ring / ping-pong / one-wave-per-SIMD kernels ---------------------------------------
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// phase 2 of the epilogue: lane owns 8 consecutive columns n..n+7 of row m, h[] = bf16(acc*s+bias) values
template <int EPI, int FMT>
__device__ __forceinline__ void row_epilogue(const FluxmiGemmGroup& G, float qs, int m, int n, const float* h) {
  if constexpr (EPI == FLUXMI_EPI_BF16) {
    *(uint4*)((u16*)G.C + (long long)m * G.ldc + n) = pack8(h);
  } else if constexpr (EPI == FLUXMI_EPI_GATE_RESID) {
    float r[8], g[8], o[8];
    unpack8(*(const uint4*)((const u16*)G.resid + (long long)m * G.ldr + n), r);
    unpack8(*(const uint4*)((const u16*)G.gate + n), g);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = r[j] + rbf(g[j] * h[j]);
    *(uint4*)((u16*)G.C + (long long)m * G.ldc + n) = pack8(o);
  } else {
    // quantising epilogues (optionally through GELU / SiLU); SPLIT routes by column range
    unsigned char* dst;
    bool plain = false;
    if constexpr (EPI == FLUXMI_EPI_SPLIT) {
      if (n < G.split_n) {
        plain = true;
        dst = nullptr;
      } else {
        dst = (unsigned char*)G.C2 + f8_act_off(m, G.ldc2, G.c2_col0 + (n - G.split_n), G.c8_pairs);
      }
    } else {
      dst = (unsigned char*)G.C + f8_act_off(m, G.ldc, n, G.c8_pairs);
    }
    if (plain) {
      *(uint4*)((u16*)G.C + (long long)m * G.ldc + n) = pack8(h);
      return;
    }
    // pairs of values through packed f32 math (the epilogue is VALU-issue bound: 4 cycles per wave64 instruction whatever it does)
    float t[8];
    const v2f_t qs2 = {qs, qs};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      v2f_t g = {h[j], h[j + 1]};
      if constexpr (EPI == FLUXMI_EPI_GELU_QUANT || EPI == FLUXMI_EPI_SPLIT) g = rbf2(gelu_tanh_f2(g));
      else if constexpr (EPI == FLUXMI_EPI_SILU_QUANT) g = (v2f_t){rbf(silu_f(g[0])), rbf(silu_f(g[1]))};
      // q_prepare: bf16(x * scale), clamp (NaN-propagating), then the RNE fp8 convert              float8_quantize.py:217-218
      const v2f_t p = rbf2(g * qs2);
      const float mx = fp8_max<FMT>();
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        t[j + e] = clamp_nan(p[e], mx);
      }
    }
    uint2 o;
    o.x = cvt4_fp8<FMT>(t[0], t[1], t[2], t[3]);
    o.y = cvt4_fp8<FMT>(t[4], t[5], t[6], t[7]);
    *(uint2*)dst = o;
  }
}

template <int EPI, int FMT, int TM, int TN>
__device__ __forceinline__ void lds_epilogue(const FluxmiGemmGroup& G, v16f (&acc)[TM][TN], float s, float qs, unsigned char* wbuf,
                                             int m_wave0, int n_wave0, int M, int lane, float* xchg = nullptr, int wave = 0,
                                             unsigned char* ring = nullptr) {
  // phase 1: h = bf16(acc*s + bias) -> per-wave LDS tile [TM*32 rows][TN*32 cols] bf16, 16-B chunks XOR-swizzled by row
  constexpr int ROW_B = TN * 64;       // bytes per row (TN*32 bf16)
  constexpr int CH = ROW_B / 16;       // 16-B chunks per row
  const int l31 = lane & 31, hi = lane >> 5;
  // ---- every global load of the epilogue is ISSUED HERE, in one batch.  Round 1 loaded the bias words inside the (j, g4) loops behind
  // a per-load null check and the residual rows inside the per-row `m < M` guard of phase 2: each load sat in its own basic block, hipcc
  // waited vmcnt(0) right behind it, and a tile paid TN*4 + (TM*32)/RPP exposed L2 / HBM round trips (8 + 16 on the ping-pong kernel,
  // 16 + 32 on the one-wave-per-SIMD kernel: 10-35 us of a 60-200 us launch whose tiles run in a single round).
  // bias words one 32-column block (j) ahead: braw[j & 1][g4] holds block j while block j+1 is in flight
  const fluxmi_gptr<const u16> bias_p = uni_ptr((const u16*)G.bias);
  const fluxmi_gptr<const u16> resid_p = uni_ptr((const u16*)G.resid);
  const fluxmi_gptr<const u16> gate_p = uni_ptr((const u16*)G.gate);
  const fluxmi_gptr<u16> c_p = uni_ptr((u16*)G.C);
  const long long ldc_u = uni_i64(G.ldc), ldr_u = uni_i64(G.ldr);
  uint2 braw[2][4];
  const bool has_bias = bias_p != nullptr;
  // no branch around the loads (an if / else whose else-arm zeroes the same registers makes hipcc wait vmcnt(0) before the branch): a
  // group without bias reads the same offsets of its weight matrix instead (always mapped: n < N <= N*K elements) and the words are
  // zeroed by a select once they are pinned
  const fluxmi_gptr<const u16> bias_src = has_bias ? bias_p : uni_ptr((const u16*)G.W);
  auto load_bias = [&](int j, uint2 (&dst)[4]) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      dst[g4] = __builtin_bit_cast(uint2, *(fluxmi_gptr<const fluxmi_v2i>)(bias_src + n_wave0 + j * 32 + g4 * 8 + hi * 4));
  };
  // called at the top of block j: issue block j+1, then pin block j (the empty asm keeps hipcc from sinking each load back into the
  // basic block that uses it and waiting for it there)
  auto bias_step = [&](int j) {
    if (j + 1 < TN) load_bias(j + 1, braw[(j + 1) & 1]);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      asm volatile("" : "+v"(braw[j & 1][g4].x), "+v"(braw[j & 1][g4].y));
      if (!has_bias) braw[j & 1][g4] = make_uint2(0, 0);
    }
  };
  auto bias_of = [&](int j, int g4, float* b) {
    const uint2 v = braw[j & 1][g4];
    b[0] = __uint_as_float(v.x << 16); b[1] = __uint_as_float(v.x & 0xffff0000u);
    b[2] = __uint_as_float(v.y << 16); b[3] = __uint_as_float(v.y & 0xffff0000u);
  };
  load_bias(0, braw[0]);
  // gate*y + x: the residual rows of phase 2 in batches of RB rows (row clamped instead of guarded: the load is unconditional, the store
  // is not); batch 0 is issued here and flies under phase 1, batch b+1 is issued before batch b is consumed
  constexpr int RPP0 = 64 / CH, NIT = (TM * 32) / RPP0;
  constexpr int RB = NIT < 8 ? NIT : NIT % 8 == 0 ? 8 : NIT % 7 == 0 ? 7 : NIT % 6 == 0 ? 6 : NIT % 5 == 0 ? 5 : 4;  // batches must tile the NIT passes
  static_assert(NIT % RB == 0, "residual batches must cover every pass");
  uint4 rres[2][EPI == FLUXMI_EPI_GATE_RESID ? RB : 1];
  uint4 graw = make_uint4(0, 0, 0, 0);
  auto load_resid = [&](int b, uint4 (&dst)[EPI == FLUXMI_EPI_GATE_RESID ? RB : 1]) {
    if constexpr (EPI == FLUXMI_EPI_GATE_RESID) {
      const int c = lane % CH, n = n_wave0 + c * 8;
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int m = min(m_wave0 + (b * RB + q) * RPP0 + lane / CH, M - 1);
        dst[q] = __builtin_bit_cast(uint4, *(fluxmi_gptr<const v4i>)(resid_p + (long long)m * ldr_u + n));
      }
    }
  };
  if constexpr (EPI == FLUXMI_EPI_GATE_RESID) {
    graw = __builtin_bit_cast(uint4, *(fluxmi_gptr<const v4i>)(gate_p + n_wave0 + (lane % CH) * 8));
    load_resid(0, rres[0]);
  }
  if constexpr ((EPI == FLUXMI_EPI_GELU_QUANT || EPI == FLUXMI_EPI_SPLIT) && TM == 4 && TN == 2) {
    // ---- table-driven quantising epilogue (8-wave 256x256 kernels; block-uniform branch: every wave of the workgroup takes it).
    // bf16(acc*s+bias) -> GELU -> bf16 -> x scale -> bf16 -> clamp -> fp8 is a pure function of the 16 bits of its input once the
    // consumer's input scale is frozen: G.q_lut holds it for all 65536 patterns (64 KiB, built by the same device code).  The
    // idle ring takes the table (first 64 KiB, LDS-DMA) and a [128 rows][64 B] fp8 transposition tile per wave (second 64 KiB);
    // ~25 VALU instructions per element become one ds_read_u8 gather.
    bool use_lut = G.q_lut != nullptr && ring != nullptr;
    if constexpr (EPI == FLUXMI_EPI_SPLIT) use_lut = use_lut && n_wave0 >= G.split_n;
    if (use_lut) {
      // (1) table -> LDS: 64 pieces of 1 KiB, 8 per wave
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int piece = wave * 8 + q;
        glds16((const unsigned char*)G.q_lut + piece * 1024 + lane * 16, ring + piece * 1024);
      }
      // (2) meanwhile: h = bf16(acc*s + bias), packed two per register
      unsigned hp[TM][TN][4][2];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bias_step(j);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float bias[4];
          bias_of(j, g4, bias);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            hp[i][j][g4][0] = pack_bf2(fmaf(acc[i][j][g4 * 4 + 0], s, bias[0]), fmaf(acc[i][j][g4 * 4 + 1], s, bias[1]));
            hp[i][j][g4][1] = pack_bf2(fmaf(acc[i][j][g4 * 4 + 2], s, bias[2]), fmaf(acc[i][j][g4 * 4 + 3], s, bias[3]));
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // (3) gather + transpose: lane owns 4 consecutive columns of row ml -> one dword of the wave's fp8 tile
      unsigned char* tile = ring + 65536 + wave * 8192;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ml = i * 32 + l31;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const unsigned a = hp[i][j][g4][0], b = hp[i][j][g4][1];
            const unsigned q0 = ring[a & 0xffffu], q1 = ring[a >> 16], q2 = ring[b & 0xffffu], q3 = ring[b >> 16];
            const int nl = j * 32 + g4 * 8 + hi * 4;
            const int chunk = (nl >> 4) ^ ((ml >> 1) & 3);
            *(unsigned*)(tile + ml * 64 + chunk * 16 + (nl & 12)) = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // (4) rows leave as 64-byte runs: lane -> (row, 16-byte chunk)
#pragma unroll
      for (int it = 0; it < (TM * 32) / 16; ++it) {
        const int ml = it * 16 + (lane >> 2), c = lane & 3;
        const int m = m_wave0 + ml, n = n_wave0 + c * 16;
        const uint4 raw = *(const uint4*)(tile + ml * 64 + ((c ^ ((ml >> 1) & 3)) * 16));
        if (m < M) {
          unsigned char* dst;
          if constexpr (EPI == FLUXMI_EPI_SPLIT) dst = (unsigned char*)G.C2 + f8_act_off(m, G.ldc2, G.c2_col0 + (n - G.split_n), G.c8_pairs);
          else dst = (unsigned char*)G.C + f8_act_off(m, G.ldc, n, G.c8_pairs);
          *(uint4*)dst = raw;
        }
      }
      return;
    }
  }
  if constexpr (EPI == FLUXMI_EPI_BF16 || EPI == FLUXMI_EPI_SPLIT) {
    // ---- fused V^T: this wave's tile is [128 keys][TN*32 d] of ONE head's V; it goes to LDS transposed ([d][128 keys], keys in
    // the PV MFMA's k-slot order) and leaves as 256-byte runs of one d-row of vt_out                       (wave-uniform branch)
    const int vcol0 = G.kv_col0 + G.heads * 128;
    if (G.vt_out && n_wave0 >= vcol0 && n_wave0 < vcol0 + G.heads * 128) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bias_step(j);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int dl = j * 32 + g4 * 8 + hi * 4;  // local d of this lane's 4 values
          float bias[4];
          bias_of(j, g4, bias);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int ml = i * 32 + l31;
            // storage position of key ml: bits 2 and 3 swapped inside its 16-key group
            const int pos = (ml & ~12) | ((ml & 4) << 1) | ((ml & 8) >> 1);
            const bool live = m_wave0 + ml < M;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const u16 v = live ? f2bf(fmaf(acc[i][j][g4 * 4 + e], s, bias[e])) : (u16)0;
              *(u16*)(wbuf + (dl + e) * (TM * 64) + pos * 2) = v;
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int KROW_B = TM * 64;        // bytes per d-row (TM*32 keys)
      constexpr int KCH = KROW_B / 16;       // 16-B chunks (8 keys) per d-row
      constexpr int DPP = 64 / KCH;          // d-rows per pass
      const int d0 = n_wave0 - vcol0;        // first d-row of the tile in vt_out (= head*128 + d)
#pragma unroll
      for (int it = 0; it < (TN * 32) / DPP; ++it) {
        const int dl = it * DPP + lane / KCH, c = lane % KCH;
        const uint4 raw = *(const uint4*)(wbuf + dl * KROW_B + c * 16);
        const int key = m_wave0 + c * 8;  // group-relative position of the 8 keys
        if (key < G.vt_rows) *(uint4*)((u16*)G.vt_out + (long long)(d0 + dl) * G.vt_ld + G.tok0 + key) = raw;
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bias_step(j);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int nl = j * 32 + g4 * 8 + hi * 4;  // local column of this lane's 4 values
      float bias[4];
      bias_of(j, g4, bias);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ml = i * 32 + l31;
        uint2 v;
        v.x = pack_bf2(fmaf(acc[i][j][g4 * 4 + 0], s, bias[0]), fmaf(acc[i][j][g4 * 4 + 1], s, bias[1]));
        v.y = pack_bf2(fmaf(acc[i][j][g4 * 4 + 2], s, bias[2]), fmaf(acc[i][j][g4 * 4 + 3], s, bias[3]));
        const int chunk = (nl >> 3) ^ (ml & (CH - 1));
        *(uint2*)(wbuf + ml * ROW_B + chunk * 16 + (nl & 4) * 2) = v;
      }
    }
  }
  // the tile is private to this wave: no barrier, only the LDS write -> read ordering of one wave
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // phase 2: lane -> (row, 16-B chunk); 64 lanes cover (64/CH) rows x ROW_B bytes per pass
  constexpr int RPP = 64 / CH;  // rows per pass
  if constexpr (EPI == FLUXMI_EPI_BF16 || EPI == FLUXMI_EPI_SPLIT) {
    // ---- fused K: QKNorm (fp32 rms over the head's 128 columns, learnable scale) + RoPE (bf16 arithmetic) + head-major store.
    // A 64-column wave tile holds half a head: the two waves of a head swap their per-row sums of squares through `xchg`
    // (block-uniform branch: a 256-column tile lies entirely inside the K range).            flux_model.py:158-176,60-65
    if (G.k_out && n_wave0 >= G.kv_col0 && n_wave0 < G.kv_col0 + G.heads * 128) {
      constexpr int NP = (TM * 32) / RPP;
      uint4 raw[NP];
      float ssr[NP];
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int ml = it * RPP + lane / CH, c = lane % CH;
        raw[it] = *(const uint4*)(wbuf + ml * ROW_B + ((c ^ (ml & (CH - 1))) * 16));
        float x[8];
        unpack8(raw[it], x);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
#pragma unroll
        for (int o = CH / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, CH);
        ssr[it] = ss;
      }
      if constexpr (TN * 32 < 128) {
        float* mine = xchg + wave * (TM * 32);
#pragma unroll
        for (int it = 0; it < NP; ++it)
          if (lane % CH == 0) mine[it * RPP + lane / CH] = ssr[it];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float* other = xchg + (wave ^ 1) * (TM * 32);
#pragma unroll
        for (int it = 0; it < NP; ++it) ssr[it] += other[it * RPP + lane / CH];
      }
      const int hcol = n_wave0 - G.kv_col0;           // column inside the K block of the row
      const int head = hcol >> 7, d0 = hcol & 127;
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int ml = it * RPP + lane / CH, c = lane % CH;
        const int m = m_wave0 + ml;
        if (m < M) {
          const int d = d0 + c * 8;
          const long long tok = (long long)G.tok0 + m;
          float x[8], w[8], cs[8], y[8];
          unpack8(raw[it], x);
          unpack8(*(const uint4*)((const u16*)G.k_norm + d), w);
          unpack8(*(const uint4*)((const u16*)G.pe + (tok * 64 + (d >> 1)) * 2), cs);
          const float rinv = 1.0f / sqrtf(ssr[it] * (1.0f / 128.0f) + 1e-6f);
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = rbf((x[j] * rinv) * w[j]);
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float cc = cs[2 * p], sn = cs[2 * p + 1];
            y[2 * p] = rbf(rbf(cc * x[2 * p]) + rbf((-sn) * x[2 * p + 1]));
            y[2 * p + 1] = rbf(rbf(sn * x[2 * p]) + rbf(cc * x[2 * p + 1]));
          }
          *(uint4*)((u16*)G.k_out + ((long long)head * G.k_rows + tok) * 128 + d) = G.k_f16 ? pack8_f16(y) : pack8(y);
        }
      }
      return;
    }
  }
  if constexpr (EPI == FLUXMI_EPI_GATE_RESID) {
    // x + bf16(gate * h), residual rows prefetched a batch ahead                                   flux_model.py:389-397,484
    float g[8];
    unpack8(graw, g);
    const int c = lane % CH;
#pragma unroll
    for (int b = 0; b < NIT / RB; ++b) {
      if (b + 1 < NIT / RB) load_resid(b + 1, rres[(b + 1) & 1]);
      // pin batch b's loads above this point (hipcc otherwise sinks each into the guarded store block and waits for it there)
#pragma unroll
      for (int q = 0; q < RB; ++q) asm volatile("" : "+v"(rres[b & 1][q].x), "+v"(rres[b & 1][q].y), "+v"(rres[b & 1][q].z), "+v"(rres[b & 1][q].w));
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int ml = (b * RB + q) * RPP + lane / CH;
        const int m = m_wave0 + ml;
        const uint4 raw = *(const uint4*)(wbuf + ml * ROW_B + ((c ^ (ml & (CH - 1))) * 16));
        float h[8], r[8], o[8];
        unpack8(raw, h);
        unpack8(rres[b & 1][q], r);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = r[j] + rbf(g[j] * h[j]);
        if (m < M) *(fluxmi_gptr<v4i>)(c_p + (long long)m * ldc_u + n_wave0 + c * 8) = __builtin_bit_cast(v4i, pack8(o));
      }
    }
  } else {
#pragma unroll
    for (int it = 0; it < (TM * 32) / RPP; ++it) {
      const int ml = it * RPP + lane / CH, c = lane % CH;
      const int m = m_wave0 + ml;
      const uint4 raw = *(const uint4*)(wbuf + ml * ROW_B + ((c ^ (ml & (CH - 1))) * 16));
      if (m < M) {
        float h[8];
        unpack8(raw, h);
        row_epilogue<EPI, FMT>(G, qs, m, n_wave0 + c * 8, h);
      }
    }
  }
}



}  // namespace
