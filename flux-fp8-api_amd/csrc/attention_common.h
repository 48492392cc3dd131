// fluxmi -- pieces shared by the attention kernel (attention2.hip: 8 waves x 32 query rows) and attention.hip.
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "fluxmi_internal.h"

// ---- balanced grid (round 5) ------------------------------------------------------------------------------------------------------
// One workgroup = one TASK = 256 query rows of one head over all key tiles; the chip runs 256 of them at a time (128 KiB of LDS each: one
// per CU), so 432 tasks (Flux-dev 1024^2: 24 heads x 18 row blocks) take TWO rounds with 80 CUs idle in the second, and 264 (768^2) take
// two rounds for 1.03 rounds of work.  With `on`, every XCD (32 CUs; whole heads per XCD as before) runs its first `full_per_x` tasks
// whole and cuts the key range of the remaining `rem` (< 32) into PIECES, one workgroup each: the concatenated tile sequence rem x ntiles
// is divided into `nb` contiguous bins of equal length -- the work of one CU -- and a bin that straddles a task boundary is two pieces.
// The pieces are launched longest first (first pieces of the bins, then the second pieces, each group by descending length): the CU that
// finishes the shortest first piece picks up the longest second piece, so every CU ends up with one bin's worth of tiles without any
// workgroup looping over pieces (a loop around the kernel body keeps the next piece's arguments alive across the step loop, which sits
// at exactly 256 VGPRs: 121 spilled registers, measured in round 5).  A piece that is not a whole task writes (O, m, l) in fp32 to its
// slot of `part`; the piece that arrives LAST at the task's counter merges all of them in piece order (log-sum-exp in the exp2 domain:
// the result does not depend on which piece arrives last) and stores the rows.
// Grid = 8 x (full_per_x + npieces) workgroups, workgroup b runs on XCD b % 8.
constexpr int ATTN_MAX_PIECES = 64;                                   // per XCD: at most two per CU
struct AttnPiece { unsigned char tloc, pidx, np, base; unsigned short tb, len; };  // task (leftover index), piece of np, canonical index of the task's first piece, tiles [tb, tb + len)
struct AttnSplit {
  int on;
  int n_per_x;     // tasks per XCD (= tasks / 8)
  int full_per_x;  // of which run whole (a multiple of 32)
  int npieces;     // pieces per XCD (launch order = order of `pieces`)
  int thin;        // 1: the binned tasks are a thin last round folded into the round in front of it, or a single partial round (what attn_split = 1 takes)
  int pad;
  float* part;     // [8][ATTN_MAX_PIECES] slots of ATTN_PART_FLOATS floats (slot = canonical piece index)
  unsigned* cnt;   // [8][ATTN_SPLIT_MAXT] arrival counters, zero between launches (the merging piece resets its task's)
  AttnPiece pieces[ATTN_MAX_PIECES];
};
static_assert(sizeof(AttnPiece) == 8, "AttnPiece is read as one 8-byte scalar load");
constexpr int ATTN_PART_FLOATS = 8 * 17 * 64 * 4;                     // per slot: [wave][i < 17][lane] float4 = 16 x four accumulator floats of every lane, then (m, l, -, -)
constexpr int ATTN_SPLIT_SNAP = 3;                                    // bin edges within 3 tiles of a task edge move onto it (no 1..3-tile pieces)
constexpr int ATTN_SPLIT_MAXP = 8;                                    // pieces per task the plan accepts
constexpr int ATTN_SPLIT_MAXT = 64;                                   // binned tasks per XCD (32 + a thin last round of <= 8, or <= 26)
constexpr size_t ATTN_SPLIT_WS_BYTES = (size_t)8 * ATTN_MAX_PIECES * ATTN_PART_FLOATS * 4 + 8 * ATTN_SPLIT_MAXT * 4;
static_assert(ATTN_SPLIT_WS_BYTES == FLUXMI_ATTN_SPLIT_WS_BYTES, "scratch size of the balanced grid (fluxmi_internal.h)");
AttnSplit fluxmi_attn_plan(int tasks, int ntiles, int cus);

// launch arguments of both attention kernels
struct AttnArgs {
  const u16* Q; const u16* K; const u16* VT;
  // raw-Q mode (Q == nullptr): query rows come straight from the qkv GEMM output; QKNorm + RoPE are applied while the Q fragments
  // are loaded (flux_model.py:158-176,60-65), so the normalised/rotated Q tensor is never written to or re-read from HBM
  const u16* qraw; long long ldq; const u16* pe; const u16* qn[2];
  void* out; long long ld_out; int col_off; int out_fp8;
  const float* q_scale[2]; int split;
  int B, L, Lp, H;
  float scale_log2;
  float defer_log2;  // the running max is updated (O, l, pending P rescaled) only when a row max grew by more than 2^defer_log2 (exp2 domain)
  int k_f16;  // K holds fp16: the folded kernel (scale * log2 e in Q, -max in the accumulator init), f16 MFMAs for QK^T
  FluxmiPrefetch pf;  // weights of the following GEMMs, read by pf.wgs extra workgroups behind the attention grid (fluxmi_internal.h)
  AttnSplit sp;       // balanced grid: the last, partial round of workgroups split along the keys (see AttnSplit)
  unsigned long long* dbg;  // probes (fluxmi_attention_debug_buffer): [workgroup][8] = {blockIdx, XCC id | HW_ID << 8, start, end, Q built, prologue landed, loop done, drain done} in 100 MHz ticks; null = off
  int out_pairs;  // fp8 output rows in the row-pair layout (fluxmi_gemm_group_t.a_pairs: the next F8Linear's A operand; dense rows, even B * L)
  int abl;  // A/B knobs (FLUXMI_ATTN_ABL, read per call): 2 = no barrier in the 8-wave kernel (timing only), 8 = fp8 output through 16 x 4 B
            // stores per lane (also taken when the output rows are not 16-byte aligned)
};

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one 1 KiB LDS-DMA piece: 16 B per lane from buffer `rsrc` at (per-lane voff + wave-uniform soff) to LDS at lds + lane*16.
// Kept out of the kernel template: a device-only builtin inside template-dependent code makes hipcc's HOST pass drop the kernel stub.
typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)lds, 16, voff, soff, 0, 0);
}


constexpr int KT = 64;                 // keys per tile
constexpr int K_BYTES = KT * 256;      // 16 KiB
constexpr int V_BYTES = 128 * KT * 2;  // 16 KiB
constexpr int A_STAGE = K_BYTES + V_BYTES;


// Q fragments (MFMA B operand): 8 x (8 bf16); lane (l31, hi) holds d = c*16 + hi*8 + [0,8) of query row `qld` of (b, h).
// raw-Q mode (a.Q == nullptr): the row comes straight from the qkv GEMM output and QKNorm + RoPE are applied here
// (flux_model.py:158-176,60-65), so the normalised / rotated Q tensor never exists in HBM.
// F16: the fragments are multiplied by `fold` (softmax scale * log2 e, attention2.hip) and stored as fp16 -- the bf16 Q values times
// the scale carry 2^-11 relative rounding error in fp16 (a bf16 re-rounding would put 2^-8 on every score; measured: it doubles the
// attention error against the oracle) -- and are bit-cast into the v8bf slots.
template <bool F16>
__device__ __forceinline__ void load_q_frags(const AttnArgs& a, int b, int h, int qld, int hi, float fold, v8bf (&qf)[8]) {
  const long long bh = (long long)b * a.H + h;
  if (a.Q) {
    const u16* qp = a.Q + (bh * a.L + qld) * 128 + hi * 8;
    uint4 raw[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) raw[c] = *(const uint4*)(qp + c * 16);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (F16) {
        float y[8];
        unpack8(raw[c], y);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] *= fold;
        raw[c] = pack8_f16(y);
      }
      qf[c] = __builtin_bit_cast(v8bf, raw[c]);
    }
  } else {
    const long long tok = (long long)b * a.L + qld;
    const u16* qp = a.qraw + tok * a.ldq + (long long)h * 128 + hi * 8;
    const u16* pp = a.pe + (tok * 64 + hi * 4) * 2;  // (cos, sin) of pairs d/2
    const u16* wn = a.qn[qld < a.split ? 0 : 1] + hi * 8;
    uint4 raw[8], rw[8], rcs[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      raw[c] = *(const uint4*)(qp + c * 16);
      rw[c] = *(const uint4*)(wn + c * 16);
      rcs[c] = *(const uint4*)(pp + c * 16);
    }
    float x[8][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      unpack8(raw[c], x[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += x[c][j] * x[c][j];
    }
    {  // the other 64 values of the row sit in lane ^ 32
      const unsigned u = __float_as_uint(ss);
      const auto sw2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      ss = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
    }
    const float rinv = 1.0f / sqrtf(ss * (1.0f / 128.0f) + 1e-6f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float w[8], cs[8], y[8];
      unpack8(rw[c], w);
      unpack8(rcs[c], cs);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[c][j] = rbf((x[c][j] * rinv) * w[j]);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float cc = cs[2 * p], sn = cs[2 * p + 1];
        y[2 * p] = rbf(rbf(cc * x[c][2 * p]) + rbf((-sn) * x[c][2 * p + 1]));
        y[2 * p + 1] = rbf(rbf(sn * x[c][2 * p]) + rbf(cc * x[c][2 * p + 1]));
      }
      if (F16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] *= fold;
      }
      const uint4 pk = F16 ? pack8_f16(y) : pack8(y);
      qf[c] = __builtin_bit_cast(v8bf, pk);
    }
  }
}
__device__ __forceinline__ void load_q_frags(const AttnArgs& a, int b, int h, int qld, int hi, v8bf (&qf)[8]) {
  load_q_frags<false>(a, b, h, qld, hi, 1.0f, qf);
}

// normalise the O^T accumulators by the row sum and store one query row per lane (bf16, or fp8 with the consumer's input scale)
template <int FMT>
__device__ __forceinline__ void store_o(const AttnArgs& a, const v16f (&o)[4], float inv, int b, int h, int qrow, int hi) {
  if (qrow >= a.L) return;
  const long long orow = ((long long)b * a.L + qrow) * a.ld_out + a.col_off + h * 128;
  if (a.out_fp8) {
    const float qs = *a.q_scale[qrow < a.split ? 0 : 1];
    // out_pairs: rows 2r, 2r + 1 interleaved in 64-byte chunks (common.h f8_act_off); a head's 128 columns are two chunks, 128 bytes apart
    const long long grow = (long long)b * a.L + qrow;
    const int pr = a.out_pairs;
    unsigned char* op = (unsigned char*)a.out + (pr ? f8_act_off(grow, a.ld_out, a.col_off + h * 128, 1) : orow);
    unsigned w[4][4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        w[db][g] = cvt4_fp8<FMT>(q_prepare<FMT>(rbf(o[db][g * 4 + 0] * inv), qs), q_prepare<FMT>(rbf(o[db][g * 4 + 1] * inv), qs),
                                 q_prepare<FMT>(rbf(o[db][g * 4 + 2] * inv), qs), q_prepare<FMT>(rbf(o[db][g * 4 + 3] * inv), qs));
    if (a.abl & 8) {  // 16 x 4 B per lane (unaligned rows; A/B)
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) *(unsigned*)(op + (pr ? (db >> 1) * 128 + (db & 1) * 32 : db * 32) + g * 8 + hi * 4) = w[db][g];
      return;
    }
    // word (db, g) of lane half `hi` covers d = db*32 + g*8 + hi*4 + [0,4).  One v_permlane32_swap of words g and g+2 leaves the
    // lower lane with (g, hi 0), (g, hi 1) and the upper lane with (g+2, hi 0), (g+2, hi 1): after two swaps per d-block each lane
    // owns 16 contiguous bytes -> 4 x global_store_dwordx4 per lane instead of 16 x dword (the store tail is issue-bound)
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const auto s0 = __builtin_amdgcn_permlane32_swap(w[db][0], w[db][2], false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(w[db][1], w[db][3], false, false);
      uint4 v;
      v.x = s0[0]; v.y = s0[1]; v.z = s1[0]; v.w = s1[1];
      *(uint4*)(op + (pr ? (db >> 1) * 128 + (db & 1) * 32 : db * 32) + hi * 16) = v;
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

}  // namespace

int fluxmi_launch_attention2(const AttnArgs& a, int fmt, hipStream_t s, bool exact);
