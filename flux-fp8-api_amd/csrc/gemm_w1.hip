// fluxmi -- "w1" GEMM: 256x256 tile, FOUR waves, one per SIMD, each owning a 128x128 accumulator block (gfx950).
//
// Same math, operands, grouping and epilogues as gemm.hip / gemm_ring.hip.  Why another shape: on real (non-zero) data the fp8
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
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "gemm_epilogue.h"

namespace {

struct W1Frags { v8i fw[4]; v8i fa[4]; };
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

// ---- stream-K (SK = true) for a launch of T <= G tiles on G CUs (mlp.2 / linear2: 216 tiles, 256 CUs: whole tiles keep 84 % of the chip
// busy).  The grid is G persistent workgroups, one per CU.  On every XCD (block b runs on XCD b % 8: used for locality only) the first
// nt_x workgroups are MAINS: each owns one tile and computes its K range [0, km_x) -- all mains walk K from 0 in lockstep, exactly like
// the plain launch, so tiles that share an A or W panel still hit it in their XCD's L2 at the same time (a first version that cut the
// launch's K-steps into equal CONTIGUOUS ranges put every workgroup at a different K position and ran at HBM speed: 308 vs 182 us).
// The other nh_x workgroups of the XCD are HELPERS: they share the tails [km_x, nku) of the XCD's own tiles (tile-major, cut into nh_x
// equal contiguous unit ranges, so a tail has one or two pieces), park each piece's fp32 partial accumulators in `partial` and publish
// a flag; km_x ~ nku * nt_x / (nt_x + nh_x) balances mains and helpers.  A main adds its tile's pieces in ascending K order, then runs the
// ordinary epilogue.  Hand-off = the write-through recipe of the CDNA guide (Guideline 16, R1): sc1 payload stores, every storing wave
// drains vmcnt, barrier, ONE relaxed agent-scope flag store; the main polls that word relaxed (bounded), ONE agent-scope acquire,
// barrier, plain loads.  Helpers never wait for anything, so there is no cycle and no dependence on dispatch order or placement; a wrong
// guess about the XCD of a block costs speed only.  Flags are cleared by their reader.
struct FluxmiSkArgs {
  float* partial;     // [2 * T][256 x 256] fp32 (tile t, piece j), layout [wave][quad q = 0..63][lane] x 16 B: every access is a 1 KiB line run
  unsigned* flags;    // [2 * T] 1 = piece published
  unsigned* err;      // != 0: a main gave up waiting (results of that tile are wrong; fluxmi_gemm_sk_status)
  int abl;            // timing-only ablations (FLUXMI_SK_ABL): 1 helpers store nothing, 2 mains add nothing, 4 mains do not wait
  unsigned long long* dbg;  // FLUXMI_SK_ABL & 16: per-workgroup timestamps (s_memrealtime, 10 ns): start, first K loop done, end, role
  int bias_units;     // added to every km_x (helpers pay a prologue + drain + 256 KiB store per piece): tuning knob
};
// per-XCD split of a stream-K launch (T tiles, G workgroups, nku K-units per tile): wave-uniform scalar arithmetic
struct SkXcd { int t0, nt, nh, km, tail; };
__device__ __forceinline__ SkXcd sk_xcd(int x, int T, int G, int nku, int bias) {
  SkXcd r;
  const int q = T >> 3, rem = T & 7;
  r.nt = q + (x < rem ? 1 : 0);
  r.t0 = x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q;
  r.nh = (G >> 3) - r.nt;
  // a helper's range must be at least one tail long (a tail then has at most two pieces): needs nt >= nh, else no split on this XCD
  const int km = (r.nh > 0 && r.nt >= r.nh) ? (nku * r.nt + (r.nt + r.nh) / 2) / (r.nt + r.nh) + bias : nku;
  r.km = min(max(km, 1), nku);
  r.tail = nku - r.km;
  return r;
}

template <bool FP8, int ACT_FMT, int ABL, int ESEL, bool SK>
__global__ void __launch_bounds__(256, 1) gemm_w1_kernel(const FluxmiGemmParams P, const FluxmiSkArgs SKA) {
  constexpr int BM = 256, BN = 256, NT = 256, TM = 4, TN = 4, NS = 4;
  constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64, STAGE = A_BYTES + W_BYTES;
  constexpr int LPT = 8;  // LDS-DMA pieces per wave per K-step (4 of A, 4 of W)
  constexpr int EB = FP8 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  const int tiles_n = P.N / BN;
  const int nblk = P.tiles_m_total * tiles_n;
  const int nk = (P.K * EB) / 64;
  constexpr int abl = ABL;  // timing-only ablations (1 = no LDS-DMA refill, 2 = no LDS reads, 4 = no barrier); 0 in production

  // fragment read addresses: (row * 64 + swizzled 16-B slot) per operand half; ring slot and 32-row tile index are immediates.
  // ds_read immediates are 16 bits, so slots 2 and 3 use a second base (+64 KiB).
  unsigned a_lo[2], a_hi[2], w_lo[2], w_hi[2];
  {
    const int ra = wm * 128 + l31, ka = (ra >> 2) & 3;
    const int rw = wn * 128 + l31, kw = (rw >> 2) & 3;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      a_lo[h2] = (unsigned)(h2 * 2 * STAGE + ra * 64 + (((hi * 2) ^ ka) << 4));
      a_hi[h2] = (unsigned)(h2 * 2 * STAGE + ra * 64 + (((hi * 2 + 1) ^ ka) << 4));
      w_lo[h2] = (unsigned)(h2 * 2 * STAGE + A_BYTES + rw * 64 + (((hi * 2) ^ kw) << 4));
      w_hi[h2] = (unsigned)(h2 * 2 * STAGE + A_BYTES + rw * 64 + (((hi * 2 + 1) ^ kw) << 4));
    }
  }
  auto fence = []() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- the work of this workgroup: one whole tile (SK = false); a main's K range of its tile or a helper's tail pieces (SK = true) ----
  const int nku = nk >> 2;  // units of four K-steps per tile
  int seg_lid = 0, seg_u0 = 0, seg_u1 = nku;  // current segment: K units [u0, u1) of tile lid
  int role = 0;                               // 0 = whole tile, 1 = main (first K range, collects the tail pieces), 2 = helper
  long long hpos = 0, hend = 0;               // helper: position in the XCD's tail space (tile-major, units)
  SkXcd X{0, 0, 0, 0, 0};
  bool first_seg = true;
  if constexpr (SK) {
    if ((SKA.abl & 16) && tid == 0) { SKA.dbg[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime(); SKA.dbg[1024 * 4 - 1024 + blockIdx.x] = __builtin_amdgcn_s_memtime(); }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    X = sk_xcd(xcd, nblk, gridDim.x, nku, SKA.bias_units);
    if (slot < X.nt) {
      role = X.tail > 0 ? 1 : 0;
      seg_lid = X.t0 + slot; seg_u0 = 0; seg_u1 = X.km;
    } else {
      role = 2;
      const int h = slot - X.nt;
      const long long U = (long long)X.nt * X.tail;
      hpos = ((long long)h * U) / X.nh; hend = ((long long)(h + 1) * U) / X.nh;
      if (hpos >= hend || (SKA.abl & 8)) return;  // (block-uniform) nothing to do
      const int tl = (int)(hpos / X.tail), off = (int)(hpos - (long long)tl * X.tail);
      seg_lid = X.t0 + tl; seg_u0 = X.km + off; seg_u1 = X.km + (int)min((long long)X.tail, off + (hend - hpos));
    }
  } else {
    seg_lid = xcd_remap(blockIdx.x, nblk);
  }

  do {
    const int lid = seg_lid, u0 = seg_u0, u1 = seg_u1;
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
    const int k0 = u0 * 4, k1 = u1 * 4;  // K-steps [k0, k1) of this tile

    // ---- LDS-DMA: descriptors (SGPRs), per-lane offsets (VGPRs, loop-invariant), tile offsets (SGPRs) ----------------------------
    const long long a_row_b = (long long)G.lda * EB, w_row_b = (long long)P.K * EB;
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(G.A, (unsigned)min((long long)M * a_row_b, 0xffffffffLL));
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(G.W, (unsigned)min((long long)P.N * w_row_b, 0xffffffffLL));
    unsigned a_voff[4], w_voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = tid + NT * i, row = p >> 2, slot = (p & 3) ^ ((row >> 2) & 3);
      a_voff[i] = (unsigned)(row * a_row_b + slot * 16);
      w_voff[i] = (unsigned)(row * w_row_b + slot * 16);
    }
    const unsigned a_soff0 = uni_u32((unsigned)(m0 * a_row_b)), w_soff0 = uni_u32((unsigned)(n0 * w_row_b));
    // piece q (0..7) of K-step kt into ring slot `slot`
    auto dma_piece = [&](int q, int slot, int kt) {
      unsigned char* d = smem + slot * STAGE + wave * 1024;
      if (q < 4) dma16_buf(ars, d + NT * 16 * q, a_voff[q], a_soff0 + kt * 64);
      else dma16_buf(wrs, d + A_BYTES + NT * 16 * (q - 4), w_voff[q - 4], w_soff0 + kt * 64);
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // half `hf` (0 = low 16 B, 1 = high 16 B) of fragment `f` (0..3 = W tiles, 4..7 = A tiles) of ring slot `slot`
    auto read_half = [&](W1Frags& F, int slot, int f, int hf) {
      const int h2 = slot >> 1, imm = (slot & 1) * STAGE + (f & 3) * 2048;
      const unsigned base = f < 4 ? (hf ? w_hi[h2] : w_lo[h2]) : (hf ? a_hi[h2] : a_lo[h2]);
      const v4i v = *(const v4i*)(smem + base + imm);
      v8i& dst = f < 4 ? F.fw[f] : F.fa[f - 4];
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

    // ---- prologue: K-steps k0..k0+2 in flight (ring slot = K-step mod 4; k0 is a multiple of 4), fragments of step k0 in registers ----
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int q = 0; q < LPT; ++q) dma_piece(q, t, k0 + t);  // (K offsets past the row end fetch harmless bytes; nk >= 4 anyway)
    wait_vmcnt<2 * LPT>();
    __builtin_amdgcn_s_barrier();
    W1Frags fa_, fb_;
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      read_half(fa_, 0, f, 0);
      read_half(fa_, 0, f, 1);
    }

    // One K-step with compile-time ring slot S.  Entry: `cur` = fragments of step kt (LDS reads possibly still in flight), LDS-DMA of
    // steps kt+1, kt+2 in flight.  16 MFMA slots; behind MFMA slot s: LDS read s of the NEXT step's fragments, and behind every even
    // slot one LDS-DMA piece of step kt+3 (into the slot step kt-1 vacated).  Past the end of the K range the refill fetches don't-care
    // bytes into a slot nobody reads any more and the "next" fragments are re-read from a stale slot: no branches in the body.
    auto step = [&](auto SLOT, W1Frags& cur, W1Frags& nxt, int kt) {
      constexpr int S = decltype(SLOT)::value, SN = (S + 1) & 3, SR = (S + 3) & 3;
      wait_vmcnt<LPT>();  // step kt+1 landed (own pieces); step kt+2 stays in flight
      if (!(abl & 4)) __builtin_amdgcn_s_barrier();
      fence();
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        mma(cur, s >> 2, s & 3);
        fence();
        // next fragments: W halves first (needed by every MFMA row), then A
        if (!(abl & 2)) read_half(nxt, SN, s >> 1, s & 1);
        if ((s & 1) == 0 && !(abl & 1)) dma_piece(s >> 1, SR, kt + 3);
        fence();
      }
    };
    for (int kt = k0; kt < k1; kt += 4) {
      step(std::integral_constant<int, 0>{}, fa_, fb_, kt);
      step(std::integral_constant<int, 1>{}, fb_, fa_, kt + 1);
      step(std::integral_constant<int, 2>{}, fa_, fb_, kt + 2);
      step(std::integral_constant<int, 3>{}, fb_, fa_, kt + 3);
    }

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the trailing don't-care refills must land before the ring is reused
    __builtin_amdgcn_s_barrier();

    if constexpr (SK) {
      if ((SKA.abl & 16) && tid == 0 && first_seg) SKA.dbg[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
      first_seg = false;
    }
    bool run_epilogue = true;
    // (SK: the tile loop invites hipcc to hoist every lane-derived address of the fix-up and the epilogue out of it, i.e. ABOVE the K
    // loop, where 100+ of them are spilled to scratch and reloaded one s_waitcnt at a time; an opaque copy of the lane / wave index keeps
    // that arithmetic below the K loop, where the fragment registers are free)
    int lane_e = lane, wave_e = wave;
    if constexpr (SK) {
      asm volatile("" : "+v"(lane_e));
      asm volatile("" : "+s"(wave_e));
    }
    if constexpr (SK) {
      // [wave][quad q][lane] x 16 B: quad q = (i * 4 + j) * 4 + r4 of the accumulator block
      if (role == 2) {
        // ---- helper: park this piece of the tile's tail, publish, move on (never waits)
        run_epilogue = false;
        const int pslot = lid * 2 + (u0 > X.km ? 1 : 0);
        float* mine = SKA.partial + (size_t)pslot * (BM * BN);
        const __amdgpu_buffer_rsrc_t prs = make_rsrc(mine, (unsigned)(BM * BN * 4));
        const unsigned lane_off = (unsigned)(wave_e * 65536 + lane_e * 16);
        if (!(SKA.abl & 1))
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const v4f v = {acc[i][j][r4 * 4 + 0], acc[i][j][r4 * 4 + 1], acc[i][j][r4 * 4 + 2], acc[i][j][r4 * 4 + 3]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), prs, lane_off + (unsigned)(((i * 4 + j) * 4 + r4) * 1024), 0,
                                                     /*aux: sc1 = write-through*/ 16);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains
        __builtin_amdgcn_s_barrier();
        if (tid == 0) __hip_atomic_store(SKA.flags + pslot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (role == 1) {
        // ---- main: add the one or two pieces of this tile's tail (two when a helper boundary falls strictly inside it)
        const long long t_start = (long long)(lid - X.t0) * X.tail, t_end = t_start + X.tail, U = (long long)X.nt * X.tail;
        int n_piece = 1;
        for (int j = 1; j < X.nh; ++j) {
          const long long bj = ((long long)j * U) / X.nh;
          n_piece = (bj > t_start && bj < t_end) ? 2 : n_piece;
        }
        if (wave_e == 0) {
          // one wave polls (lane p watches piece p), relaxed loads, bounded
          if (lane_e < n_piece && !(SKA.abl & 4)) {
            unsigned ok = 0;
            for (int spin = 0; spin < (1 << 18); ++spin) {
              if (__hip_atomic_load(SKA.flags + lid * 2 + lane_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 1; break; }
              __builtin_amdgcn_s_sleep(8);
            }
            if (!ok) __hip_atomic_store(SKA.err, 1u + (unsigned)lid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (lane_e == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // The pieces come in through the idle ring by LDS-DMA (no landing registers: the 256 accumulators + the epilogue's own state
        // leave none), 128 KiB = half a piece at a time; every wave moves and reads only its own 32 quads, so the only ordering needed
        // is its own vmcnt / lgkmcnt.
        for (int pr = 0; pr < ((SKA.abl & 2) ? 0 : n_piece); ++pr) {  // ascending K order: the sum does not depend on arrival order
          const __amdgpu_buffer_rsrc_t prs = make_rsrc(SKA.partial + (size_t)(lid * 2 + pr) * (BM * BN), (unsigned)(BM * BN * 4));
          unsigned char* mylds = smem + wave_e * 32768;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int qq = 0; qq < 32; ++qq) dma16_buf(prs, mylds + qq * 1024, (unsigned)(lane_e * 16), (unsigned)(wave_e * 65536 + (half * 32 + qq) * 1024));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int qq = 0; qq < 32; ++qq) {
              const int q = half * 32 + qq, i = q >> 4, j = (q >> 2) & 3, r4 = q & 3;
              const v4f t = *(const v4f*)(mylds + qq * 1024 + lane_e * 16);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[i][j][r4 * 4 + e] += t[e];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done before the next half's DMA rewrites the slots
          }
        }
        __syncthreads();  // every wave has consumed the pieces before their flags are cleared for the next launch
        if (tid < n_piece) __hip_atomic_store(SKA.flags + lid * 2 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------
    if (run_epilogue) {
      const float s = load_scale_u(G.sa_recip) * load_scale_u(G.sb_recip);
      const float qs = load_scale_u(G.q_scale);
      unsigned char* wbuf = smem + wave_e * (128 * 128 * 2);
      const int mw = m0 + (wave_e >> 1) * 128, nw = n0 + (wave_e & 1) * 128;
      if constexpr (ESEL >= 0) {
        lds_epilogue<ESEL, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane_e);
      } else {
        switch (P.epi) {
          case FLUXMI_EPI_BF16: lds_epilogue<FLUXMI_EPI_BF16, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane_e); break;
          case FLUXMI_EPI_GELU_QUANT: lds_epilogue<FLUXMI_EPI_GELU_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane_e); break;
          case FLUXMI_EPI_GATE_RESID: lds_epilogue<FLUXMI_EPI_GATE_RESID, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane_e); break;
          case FLUXMI_EPI_SPLIT: lds_epilogue<FLUXMI_EPI_SPLIT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane_e); break;
          case FLUXMI_EPI_QUANT: lds_epilogue<FLUXMI_EPI_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane_e); break;
          case FLUXMI_EPI_SILU_QUANT: lds_epilogue<FLUXMI_EPI_SILU_QUANT, ACT_FMT, TM, TN>(G, acc, s, qs, wbuf, mw, nw, M, lane_e); break;
          default: break;
        }
      }
    }
    if constexpr (SK) {
      if ((SKA.abl & 16) && tid == 0) { SKA.dbg[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime(); SKA.dbg[blockIdx.x * 4 + 3] = (unsigned long long)role | ((__builtin_amdgcn_s_memtime() - SKA.dbg[1024 * 4 - 1024 + blockIdx.x]) << 8); }
      if (role != 2) break;
      hpos += u1 - u0;
      if (hpos >= hend) break;
      const int tl = (int)(hpos / X.tail), off = (int)(hpos - (long long)tl * X.tail);
      seg_lid = X.t0 + tl; seg_u0 = X.km + off; seg_u1 = X.km + (int)min((long long)X.tail, off + (hend - hpos));
      // (every wave passed the barrier behind the partial stores: the ring is free for the next piece's LDS-DMA)
    }
  } while (SK);
}

// stream-K workspace: one per device (2 G x 256 KiB of partial sums + 2 G flags + 1 error word), allocated on first use -- the engine calls
// fluxmi_gemm_sk_prepare() from engine_prepare so that no allocation happens inside a forward pass or a graph capture.  Launches that
// use it must be stream-ordered with each other (they are: one engine, one stream).
struct SkWorkspace { FluxmiSkArgs a{nullptr, nullptr, nullptr, 0, nullptr, 0}; int n_wg = 0; int n_cu = 0; };
SkWorkspace g_sk[16];
int sk_workspace(SkWorkspace** out) {
  int dev = 0;
  FLUXMI_CHECK_HIP(hipGetDevice(&dev));
  FLUXMI_REQUIRE(dev >= 0 && dev < 16, "gemm stream-K: device ordinal %d out of range", dev);
  SkWorkspace& w = g_sk[dev];
  if (!w.a.partial) {
    hipDeviceProp_t prop;
    FLUXMI_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    w.n_cu = prop.multiProcessorCount;
    w.n_wg = (w.n_cu / 8) * 8;  // one 128-KiB-LDS workgroup per CU: every workgroup of the grid is resident
    char* base = nullptr;
    const size_t slots = 2 * (size_t)w.n_wg;  // (tile, piece), tiles <= workgroups
    const size_t bytes = slots * 256 * 256 * 4 + (slots + 64) * 4 + 1024 * 4 * 8;
    FLUXMI_CHECK_HIP(hipMalloc((void**)&base, bytes));
    FLUXMI_CHECK_HIP(hipMemset(base, 0, bytes));
    w.a.partial = (float*)base;
    w.a.flags = (unsigned*)(base + slots * 256 * 256 * 4);
    w.a.err = w.a.flags + slots;
    w.a.dbg = (unsigned long long*)(w.a.flags + slots + 64);
    const char* e = getenv("FLUXMI_SK_BIAS");
    w.a.bias_units = e ? atoi(e) : 2;
  }
  *out = &w;
  return 0;
}

template <bool FP8, int ACT, int ABL = 0, int ESEL = -1, bool SK = false>
int launch_w1(FluxmiGemmParams& p, hipStream_t s) {
  constexpr int BM = 256, BN = 256;
  int t = 0;
  for (int i = 0; i < p.n_groups; ++i) {
    p.g[i].m_tile_start = t;
    t += (p.g[i].M + BM - 1) / BM;
  }
  p.tiles_m_total = t;
  p.group_m = 8;
  constexpr int SMEM = 4 * (BM + BN) * 64;
  auto kern = gemm_w1_kernel<FP8, ACT, ABL, ESEL, SK>;
  static bool attr_set = false;
  if (!attr_set) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set = true;
  }
  const int nblk = t * (p.N / BN);
  if (nblk == 0) return 0;
  FluxmiSkArgs ska{nullptr, nullptr, nullptr, 0, nullptr, 0};
  int grid = nblk;
  if constexpr (SK) {
    SkWorkspace* w = nullptr;
    FLUXMI_TRY(sk_workspace(&w));
    ska = w->a;
    FLUXMI_REQUIRE(nblk <= w->n_wg, "gemm stream-K: %d tiles exceed the %d workgroups of one round (use config 16)", nblk, w->n_wg);
    grid = w->n_wg;
    { const char* e = getenv("FLUXMI_SK_BIAS"); if (e) ska.bias_units = atoi(e); }  // read per launch: tools/gemm_probe.py sweeps it
    { const char* e = getenv("FLUXMI_SK_GRID"); if (e) grid = atoi(e); }
    { const char* e = getenv("FLUXMI_SK_ABL"); ska.abl = e ? atoi(e) : 0; }
    if (getenv("FLUXMI_SK_DEBUG")) fprintf(stderr, "sk launch: tiles %d grid %d (n_cu %d) bias %d nku %d\n", nblk, grid, w->n_cu, ska.bias_units, (p.K * (FP8 ? 1 : 2)) / 256);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), SMEM, s, p, ska);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int fluxmi_gemm_sk_prepare(void) {
  SkWorkspace* w = nullptr;
  return sk_workspace(&w);
}
// stream-K launches leave a non-zero word behind when an owner gave up waiting for a partial sum (never observed; the spin is bounded so
// that a lost workgroup cannot hang the device).  Returns the word and clears it.
extern "C" int fluxmi_gemm_sk_status(unsigned* out) {
  FLUXMI_REQUIRE(out, "gemm_sk_status: NULL argument");
  SkWorkspace* w = nullptr;
  FLUXMI_TRY(sk_workspace(&w));
  FLUXMI_CHECK_HIP(hipDeviceSynchronize());
  FLUXMI_CHECK_HIP(hipMemcpy(out, w->a.err, 4, hipMemcpyDeviceToHost));
  FLUXMI_CHECK_HIP(hipMemset(w->a.err, 0, 4));
  return 0;
}

// timing-only diagnostics (FLUXMI_SK_ABL & 16): copies the per-workgroup timestamps of the LAST stream-K launch, 4 u64 per workgroup
extern "C" int fluxmi_gemm_sk_debug(unsigned long long* out, int n_wg) {
  SkWorkspace* w = nullptr;
  FLUXMI_TRY(sk_workspace(&w));
  FLUXMI_CHECK_HIP(hipDeviceSynchronize());
  FLUXMI_CHECK_HIP(hipMemcpy(out, w->a.dbg, (size_t)std::min(n_wg, 1024) * 32, hipMemcpyDeviceToHost));
  return 0;
}

// config 17 = config 16 as a stream-K launch (persistent grid, equal K ranges; see FluxmiSkArgs)
int fluxmi_launch_gemm_w1_sk(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s) {
  for (int i = 0; i < p.n_groups; ++i)
    FLUXMI_REQUIRE((long long)p.g[i].M * p.g[i].lda * (is_fp8 ? 1 : 2) < (1LL << 32) && (long long)p.N * p.K * (is_fp8 ? 1 : 2) < (1LL << 32),
                   "gemm_w1: operand larger than 4 GiB");
  for (int i = 0; i < p.n_groups; ++i)
    FLUXMI_REQUIRE(!p.g[i].vt_out && !p.g[i].k_out, "gemm stream-K: no fused K / V^T outputs");
  if (is_fp8 && act_fmt == FLUXMI_FMT_E5M2) {
    if (p.epi == FLUXMI_EPI_GATE_RESID) return launch_w1<true, FLUXMI_FMT_E5M2, 0, FLUXMI_EPI_GATE_RESID, true>(p, s);
    if (p.epi == FLUXMI_EPI_BF16) return launch_w1<true, FLUXMI_FMT_E5M2, 0, FLUXMI_EPI_BF16, true>(p, s);
  }
  fluxmi_set_error("gemm stream-K: built for e4m3 x e5m2 operands with the bf16 / gate*y+x epilogues only (epi %d)", p.epi);
  return 1;
}

// config 16 = 256x256, one wave per SIMD
int fluxmi_launch_gemm_w1(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s) {
  // buffer descriptors address 4 GiB per operand
  for (int i = 0; i < p.n_groups; ++i)
    FLUXMI_REQUIRE((long long)p.g[i].M * p.g[i].lda * (is_fp8 ? 1 : 2) < (1LL << 32) && (long long)p.N * p.K * (is_fp8 ? 1 : 2) < (1LL << 32),
                   "gemm_w1: operand larger than 4 GiB");
  if (is_fp8) {
    if (act_fmt == FLUXMI_FMT_E5M2) {
#ifdef FLUXMI_EXPERIMENTS
      static int abl = -1;  // FLUXMI_GEMM_ABL: timing-only ablations for tools/gemm_probe.py
      if (abl < 0) { const char* e = getenv("FLUXMI_GEMM_ABL"); abl = e ? atoi(e) : 0; }
      switch (abl) {
        case 1: return launch_w1<true, FLUXMI_FMT_E5M2, 1>(p, s);
        case 2: return launch_w1<true, FLUXMI_FMT_E5M2, 2>(p, s);
        case 3: return launch_w1<true, FLUXMI_FMT_E5M2, 3>(p, s);
        case 4: return launch_w1<true, FLUXMI_FMT_E5M2, 4>(p, s);
        case 7: return launch_w1<true, FLUXMI_FMT_E5M2, 7>(p, s);
        default: break;
      }
#endif
      // the step's K >= 8192 launches (mlp.2, linear2) all end in gate*y + x
      static int esel = -1;  // FLUXMI_GEMM_ESEL=0: the run-time-switch kernel for every epilogue (A/B)
      if (esel < 0) { const char* e = getenv("FLUXMI_GEMM_ESEL"); esel = e ? atoi(e) : 1; }
      if (esel && p.epi == FLUXMI_EPI_GATE_RESID) return launch_w1<true, FLUXMI_FMT_E5M2, 0, FLUXMI_EPI_GATE_RESID>(p, s);
      return launch_w1<true, FLUXMI_FMT_E5M2>(p, s);
    }
    return launch_w1<true, FLUXMI_FMT_E4M3>(p, s);
  }
  if (act_fmt == FLUXMI_FMT_E5M2) {  // bf16 operands (VAE convolutions, text encoders, bf16 flow): plain and residual epilogues specialised
    if (p.epi == FLUXMI_EPI_GATE_RESID) return launch_w1<false, FLUXMI_FMT_E5M2, 0, FLUXMI_EPI_GATE_RESID>(p, s);
    if (p.epi == FLUXMI_EPI_BF16) return launch_w1<false, FLUXMI_FMT_E5M2, 0, FLUXMI_EPI_BF16>(p, s);
    return launch_w1<false, FLUXMI_FMT_E5M2>(p, s);
  }
  return launch_w1<false, FLUXMI_FMT_E4M3>(p, s);
}
