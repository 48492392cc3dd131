// fluxmi -- flash-attention forward, 8-wave kernel (bf16, head_dim 128, non-causal), gfx950: the engine's kernel (fp16 K: the folded
// arithmetic with the per-tile barrier between the two MFMA groups; bf16 K: the unfolded one, barrier at the step start).  The 4-wave
// kernel of round 3 (one wave per SIMD, 64 query rows per wave) measured equal inside the step and slower on peaky score
// distributions (profiles/r03_attention4.txt) and left the tree in round 4 (git history: attention4.hip).
//
// 8 waves x 32 query rows, KV tiles of 64, swapped QK^T, P fed to the PV MFMA straight from the accumulator, K / V^T tiles by LDS-DMA
// into 4-deep rings (attention.hip).  The schedule inside the wave answers what the ISA of the round-1 kernel with the same geometry
// (profiles/r02_attention_isa_notes.txt) showed:
//   * hipcc sank the whole softmax of tile j (33 v_exp, 21 fma, 16 cvt) plus the 16 v_max3 of tile j+1 into the 16 PV MFMAs
//     (5.4 VALU per MFMA gap, more than the <= 5 the matrix pipe hides) while the 16 QK^T MFMAs ran with an idle VALU, and the
//     16 row-sum adds formed ONE dependent chain of v_pk_add behind the barrier with the matrix pipe empty;
//   * with exact running-max tracking the O rescale (33 v_pk_mul on the PV accumulators, i.e. a wait for the matrix pipe) fired
//     in ~65 % of the tiles on random scores: P(any of a wave's 32 rows sees a new max in tile j) = 1 - (1 - 1/(j+1))^32.
// This kernel therefore
//   1. skews the pipeline by one tile: step j runs S_{j+1} = K_{j+1} Q^T, P_j = softmax numerators of S_j, and O += V_{j-1} P_{j-1}.
//      P_{j-1} is complete before the step starts, so the softmax VALU work of P_j can be spread EVENLY over all 32 MFMA gaps of
//      the step (one score per gap: fma, exp2, row-sum add, half a cvt_pk; the row max of S_{j+1} rides in the PV gaps), each
//      gap pinned with sched_barrier: ~4 VALU + 1 ds_read per MFMA in both halves;
//   2. defers the running max (guide T13): O, l and the pending P_{j-1} are rescaled only when a row's max grew by more than
//      2^8 since the last rescale, so P is bounded by 256 (exact in bf16's exponent range, fp32 sums to 1.2e6 at L = 4608) and
//      the rescale leaves the steady state;
//   3. keeps four independent row-sum accumulators (no dependent chain) and scalar f32 VALU ops (packed f32 VALU beside MFMAs is
//      an anti-lever on CDNA4, MI355X_MICROARCH.md).
//   4. (FOLD, when K arrives as fp16: AttnArgs.k_f16) folds the softmax scale and the running max into the MATRIX pipe: Q is
//      multiplied by scale*log2(e) while its fragments are built and kept as fp16 (2^-11 relative rounding on the scaled values; the
//      same fold in bf16 puts 2^-8 on every score and doubled the attention error against the oracle, profiles/r02_attention_ab.txt),
//      QK^T runs on the f16 MFMA (same rate; K's bf16 values are exact in fp16), and the first MFMA of every S accumulator starts
//      from C = -M (M = the deferred running max, one 16-register block per lane, rewritten only in the rescale branch), so S arrives
//      as s - M and the per-score fma disappears: exp2, row-sum add, half a cvt_pk and half a max3 per MFMA gap.  The rescale
//      decision becomes `row max of the shifted tile > 2^8`.
// Correctness of the deferred rescale with a pending tile: when the branch fires with f = 2^((m_old - m_new) c), everything still at
// the old max is scaled exactly once -- O (tiles <= j-2), l (tiles <= j-1) and the bf16 fragments of P_{j-1} (re-rounded) -- and
// P_j is exponentiated after the decision.  tests/test_ops_gpu.py forces the branch (spiked key rows) and sweeps THR.
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "attention_common.h"

namespace {

template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

__device__ __forceinline__ void fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int NW2 = 8, RD2 = 4, LPW2 = 16 / NW2;
constexpr int VRING2 = RD2 * K_BYTES;
// EXACT (tests + A/B): exact max tracking, i.e. rescale whenever any row max grows.
// Measured and dropped in round 2 (profiles/r02_attention_ab.txt): refills in the PV half (-0.4 %), softmax work skewed by one gap so that
// nothing inside a gap depends on anything in it (-4 %: +17 register moves), row sums by an all-ones MFMA (-4 %), V fragments 2 instead of
// 4 MFMAs ahead (-0.5 %), Q arithmetic under the prologue DMA (-0.7 %), s_setprio 1 for the younger half (-0.3 %), no barrier at all
// (+0.4 ... 1.8 %, timing only: the barrier is not what the waves wait for), the consumers of a score's exp2 (row-sum add, cvt_pk) run one
// gap later so that nothing in a gap reads that gap's v_exp_f32 (270 fewer s_nop per four tiles, +-0 %: instruction issue is not the limit)
// MIDBAR (FLUXMI_ATTN_V=3, round 3, selectable): the per-tile barrier sits BETWEEN the two MFMA groups of a step instead of in front of it,
// so the first K fragments of step j + 1 are read under the last PV MFMAs of step j and the first V^T fragments under the last QK^T
// MFMAs (with the barrier in front, all eight waves leave it in lockstep and both waves of every SIMD wait out the LDS latency of their
// first fragments with the matrix pipe idle).  Refill order K0 K1 K2 V0 K3 | step j: V_{j+1} (QK^T half), K_{j+4} (PV half, behind
// the barrier of step j, where K_j is dead); the barrier of step j waits for K_{j+2} and V_j (two younger groups stay in flight).
template <int FMT, bool FOLD, bool EXACT, bool MIDBAR = false>
__global__ void __launch_bounds__(NW2 * 64, 2) attention2_kernel(const AttnArgs a) {
  constexpr int QB = NW2 * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqb = (a.L + QB - 1) / QB;
  const int ntiles_all = (a.L + KT - 1) / KT;
  // ---- which task, which of its key tiles: a whole task, or (balanced grid, AttnSplit) one piece of a leftover task ---------------------
  const bool split = FOLD && a.sp.on;
  const int main_wgs = split ? 8 * (a.sp.full_per_x + a.sp.npieces) : nqb * a.H * a.B;
  if ((int)blockIdx.x >= main_wgs) {  // extra workgroups behind the grid: weight prefetch for the launches that follow
    fluxmi_prefetch_ranges(a.pf, (int)blockIdx.x - main_wgs, a.pf.wgs, tid, NW2 * 64);
    return;
  }
  const unsigned long long t_dbg0 = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;  // probes only (tools/attn_timeline.py)
  auto stamp = [&](int i) {  // phase stamps 4..7 of this workgroup's record: Q fragments built, prologue tiles landed, step loop done, drain done
    if (a.dbg && tid == 0) ((volatile unsigned long long*)a.dbg)[(size_t)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memrealtime();
  };
  int lid, tb = 0, ntiles = ntiles_all;
  AttnPiece pc = {0, 0, 1, 0, 0, 0};
  if (split) {
    const int x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
    lid = x * a.sp.n_per_x + i;
    if (i >= a.sp.full_per_x) {
      // entry i - full_per_x of the launch-ordered piece table, one 8-byte scalar load from the kernarg segment (indexing `a` itself
      // with a run-time index would make hipcc copy the whole argument struct to scratch)
      typedef const __attribute__((address_space(4))) unsigned long long* kern_u64;
      const unsigned long long raw = ((kern_u64)((const __attribute__((address_space(4))) unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                                 __builtin_offsetof(AttnArgs, sp.pieces)))[i - a.sp.full_per_x];
      pc = __builtin_bit_cast(AttnPiece, raw);
      lid = x * a.sp.n_per_x + a.sp.full_per_x + pc.tloc;
      tb = pc.tb;
      ntiles = pc.len;
    }
  } else {
    lid = xcd_remap(blockIdx.x, nqb * a.H * a.B);  // whole heads per XCD: a head's K / V^T (2.4 MB at L = 4608) is fetched into one 4 MiB L2 once and shared by its q-blocks
  }
  const int L_rel = a.L - tb * KT;  // keys from this piece's first tile to the end of the sequence (masking of a ragged last tile)
  const int bhid = lid / nqb;
  const int h = bhid % a.H, b = bhid / a.H;
  const int q0 = (lid - bhid * nqb) * QB + wave * 32;
  const int qrow = q0 + l31;
  const int qld = min(qrow, a.L - 1);
  const long long bh = (long long)b * a.H + h;

  const float c = a.scale_log2;
  v8bf qf[8];  // FOLD: eight fp16 per fragment in the same registers
  load_q_frags<FOLD>(a, b, h, qld, hi, c, qf);
  if (a.dbg) { asm volatile("" ::"v"(qf[0]), "v"(qf[7])); stamp(4); }
  // S^T tile += K fragment . Q fragment (f16 MFMA for the folded kernel)
  auto mfma_qk = [](v8bf kfr, v8bf qfr, v16f acc) -> v16f {
    if constexpr (FOLD) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, kfr), __builtin_bit_cast(v8h, qfr), acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qfr, acc, 0, 0, 0);
  };

  const auto krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.K + bh * a.L * 128), 0, a.L * 256, 0x00020000);
  const auto vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.VT + bh * 128 * a.Lp), 0, 128 * a.Lp * 2, 0x00020000);
  unsigned k_off[LPW2], v_off[LPW2];
#pragma unroll
  for (int i = 0; i < LPW2; ++i) {
    const int p = tid + NW2 * 64 * i;
    k_off[i] = (unsigned)((p >> 4) * 256 + ((p & 15) ^ ((p >> 4) & 15)) * 16);
    const int d = p >> 3, vs = (p & 7) ^ ((d >> 1) & 7);
    v_off[i] = (unsigned)(d * a.Lp * 2 + vs * 16);
  }
  auto dma_k = [&](int slot, int kv0, int i) { dma16(krsrc, smem + slot * K_BYTES + wave * 1024 + NW2 * 1024 * i, k_off[i], kv0 * 256); };
  auto dma_v = [&](int slot, int kv0, int i) { dma16(vrsrc, smem + VRING2 + slot * V_BYTES + wave * 1024 + NW2 * 1024 * i, v_off[i], kv0 * 2); };

  v16f o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f;
  float l4[4] = {0.f, 0.f, 0.f, 0.f};
  // FOLD: M = deferred running max in the exp2 domain; ninit = sixteen copies of -M, the C operand of the first QK^T MFMAs
  float m_cur = 0.f;
  v16f ninit;
#pragma unroll
  for (int r = 0; r < 16; ++r) ninit[r] = 0.f;

  unsigned kx[8], vx[4];
  {
    const int sw = l31 & 15, vsw = (l31 >> 1) & 7;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) kx[cc] = (unsigned)(l31 * 256 + (((cc * 2 + hi) ^ sw) << 4));
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) vx[ch] = (unsigned)(VRING2 + l31 * 128 + (((ch * 2 + hi) ^ vsw) << 4));
  }
  auto k_frag = [&](int slot, int cc, int t) -> v8bf { return *(const v8bf*)(smem + slot * K_BYTES + kx[cc] + t * (32 * 256)); };
  auto v_frag = [&](int slot, int ch4, int db) -> v8bf { return *(const v8bf*)(smem + vx[ch4] + slot * V_BYTES + db * 4096); };

  auto mask_tile = [&](v16f (&st)[2], int kv0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        st[t][r] = key < L_rel ? st[t][r] : -1e30f;
      }
  };
  auto finish_max = [&](float mx) -> float {  // the other 32 keys of the row sit in lane ^ 32
    const unsigned u = __float_as_uint(mx);
    const auto sw2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(sw2[0]), __uint_as_float(sw2[1]));
  };

  const bool ragged = (a.L % KT) != 0 && tb + ntiles == ntiles_all;  // this piece ends with the sequence's partial tile

  // ---- prologue: K0 K1 V0 K2 V1 K3 in flight (step j issues K_{j+4} then V_{j+2}); S_0 and its row max ---------------------------
  // Every issue is unconditional: a tile past the end of the sequence reads zeros (descriptor bounds check) or harmless bytes of
  // the next V^T row into a ring slot nobody reads any more, so the vmcnt arithmetic is the same in every step and the step body
  // has no branch (a wave-uniform branch per refill split the step into basic blocks and let MachineSink drag the pinned VALU
  // work out of its MFMA gaps).
  {
    auto iss_k = [&](int t) { dma_k(t, (tb + t) * KT, 0); dma_k(t, (tb + t) * KT, 1); };
    auto iss_v = [&](int t) { dma_v(t, (tb + t) * KT, 0); dma_v(t, (tb + t) * KT, 1); };
    if constexpr (MIDBAR) { iss_k(0); iss_k(1); iss_k(2); iss_v(0); iss_k(3); }
    else { iss_k(0); iss_k(1); iss_v(0); iss_k(2); iss_v(1); iss_k(3); }
  }
  if constexpr (MIDBAR) wait_vm<3 * LPW2>();  // K0 and K1 (step 0 reads K1 before its barrier)
  else wait_vm<5 * LPW2>();
  __builtin_amdgcn_s_barrier();
  stamp(5);
  v16f sa[2], sb[2];
  v4i pfa[4], pfb[4];  // bf16 P fragments as packed words (two tiles: the one the PV MFMAs consume and the one being produced)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { pfa[i][e] = 0; pfb[i][e] = 0; }
  {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[t][r] = 0.f;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      sa[0] = mfma_qk(k_frag(0, cc, 0), qf[cc], sa[0]);
      sa[1] = mfma_qk(k_frag(0, cc, 1), qf[cc], sa[1]);
    }
  }
  if (ragged && ntiles == 1) {
    asm volatile("" ::: "memory");
    mask_tile(sa, 0);
  }
  float mx;
  {
    float m0 = sa[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) m0 = fmaxf(m0, sa[t][r]);
    mx = finish_max(m0);
  }
  if constexpr (FOLD) {  // M = exact max of tile 0; S_0 -> S_0 - M; nothing to rescale in step 0
    m_cur = mx;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[t][r] -= m_cur;
#pragma unroll
    for (int r = 0; r < 16; ++r) ninit[r] = -m_cur;
    asm volatile("" : "+v"(ninit));  // sixteen live registers, not a splat hipcc re-materialises in front of every MFMA pair
    mx = 0.f;
  }

  v8bf kpre[2][2], vpre[4];  // MIDBAR: first fragments of the next MFMA group, read one group ahead
  if constexpr (MIDBAR) {
#pragma unroll
    for (int q = 0; q < 4; ++q) kpre[q >> 1][q & 1] = k_frag(1, q >> 1, q & 1);
  }
  // ---- one step, compile-time ring slot PAR = j % 4; FIRST: no pending tile (j == 0) ------------------------------------------------
  // cur = S_j (raw scores, turned into P_j in place), nxt = S_{j+1}, pp = bf16 fragments of P_{j-1} (consumed), pc = of P_j (produced)
  auto step = [&](auto PARC, auto FIRSTC, v16f (&cur)[2], v16f (&nxt)[2], v4i (&pp)[4], v4i (&pc)[4], int j) {
    constexpr int PAR = decltype(PARC)::value;
    constexpr bool FIRST = decltype(FIRSTC)::value;
    constexpr int KS = (PAR + 1) & 3;  // slot of K_{j+1}
    constexpr int VS = (PAR + 3) & 3;  // slot of V_{j-1}
    constexpr int VR = (PAR + 2) & 3;  // slot V_{j+2} refills (held V_{j-2})
    // K_{j+1} and V_{j-1} have landed (own pieces; the barrier extends that to every wave); the two younger tile pairs stay in flight
    if constexpr (!MIDBAR) {
      wait_vm<4 * LPW2>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(a.abl & 2)) __builtin_amdgcn_s_barrier();
    }
    // -- A: running max with deferred rescale (wave-uniform branch, out of the steady state)
    auto rescale_state = [&](float alpha) {
#pragma unroll
      for (int i = 0; i < 4; ++i) l4[i] *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      if constexpr (!FIRST) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned w = (unsigned)pp[i][e];
            pp[i][e] = (int)pack_bf2(__uint_as_float(w << 16) * alpha, __uint_as_float(w & 0xffff0000u) * alpha);
          }
      }
    };
    float nmc = 0.f;
    if constexpr (FOLD) {
      // mx = this lane's part (32 of the row's 64 keys) of the row max of S_j - M: some row exceeds the threshold iff some lane does, so
      // the decision needs no cross-lane step; the other half of the row (lane ^ 32) is fetched inside the rare branch.  Rows that grew
      // are moved to their new max (delta = max(row max, 0)); S_j itself was produced with the old M
      if (__builtin_expect(__any(mx > (EXACT ? 0.0f : a.defer_log2)), 0)) {
        const float delta = fmaxf(finish_max(mx), 0.f);
        rescale_state(__builtin_amdgcn_exp2f(-delta));
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) cur[t][r] -= delta;
        m_cur += delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) ninit[r] = -m_cur;
        asm volatile("" : "+v"(ninit));
      }
    } else {
      if (__builtin_expect(__any((mx - m_run) * c > (EXACT ? 0.0f : a.defer_log2)), 0)) {  // per-lane part of the row max, as above
        const float m_new = fmaxf(m_run, finish_max(mx));
        rescale_state(__builtin_amdgcn_exp2f((m_run - m_new) * c));
        m_run = m_new;
      }
      nmc = -m_run * c;
    }
    // score e (0..31) of S_j -> P_j in place; every second one packs a bf16 pair of the P fragment
    // Pure VALU nodes carry no chain: left alone, the DAG scheduler floats every fma / exp of the step above the first MFMA and
    // MachineSink drags results down to their users.  Passing the input and the outputs of a gap's work through (empty) asm
    // volatile statements ties them between the two sched_barriers of that gap; hipcc still schedules and pads inside the gap.
    auto soft = [&](auto EC) {
      constexpr int e = decltype(EC)::value, t = e >> 4, r = e & 15;
      float sc = cur[t][r];
      asm volatile("" : "+v"(sc));
      float p = FOLD ? __builtin_amdgcn_exp2f(sc) : __builtin_amdgcn_exp2f(__builtin_fmaf(sc, c, nmc));
      float ls = l4[e & 3] + p;
      if constexpr (e & 1) {
        constexpr int f = t * 2 + (r >> 3), q = (r & 7) >> 1;
        // one v_cvt_pk_bf16_f32 for the pair (through pack_bf2 hipcc converts each half separately and ORs them: 4 instructions).
        // The s_nop is the trans-op -> VALU wait state hipcc would pad itself (p comes straight from v_exp_f32).
        int w;
        asm volatile("s_nop 0\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(cur[t][r - 1]), "v"(p));
        asm volatile("" : "+v"(p), "+v"(ls));
        pc[f][q] = w;
      } else {
        asm volatile("" : "+v"(p), "+v"(ls));
      }
      cur[t][r] = p;
      l4[e & 3] = ls;
    };
    auto gapwork = [&](auto GC) { soft(GC); };
    // -- B: S_{j+1} = K_{j+1} Q^T, two alternating accumulators; K fragments two chunks ahead; one score of P_j per gap
    {
      if constexpr (FOLD) {
        nxt[0] = ninit;
        nxt[1] = ninit;
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) nxt[t][r] = 0.f;
      }
      v8bf kf[3][2];
      if constexpr (MIDBAR) {
        kf[0][0] = kpre[0][0]; kf[0][1] = kpre[0][1];
        kf[1][0] = kpre[1][0]; kf[1][1] = kpre[1][1];
      } else {
        kf[0][0] = k_frag(KS, 0, 0); kf[0][1] = k_frag(KS, 0, 1);
        kf[1][0] = k_frag(KS, 1, 0); kf[1][1] = k_frag(KS, 1, 1);
      }
      fence();
      static_for<16>([&](auto SC) {
        constexpr int s = decltype(SC)::value, cc = s >> 1, t = s & 1;
        nxt[t] = mfma_qk(kf[cc % 3][t], qf[cc], nxt[t]);
        fence();
        if constexpr (cc + 2 < 8) kf[(cc + 2) % 3][t] = k_frag(KS, cc + 2, t);
        gapwork(std::integral_constant<int, s>{});
        // refills ride in the QK^T half (in the PV half: -0.4 %, profiles/r02_attention_ab.txt)
        if constexpr (MIDBAR) {
          if constexpr (s == 3) dma_v(KS, (tb + j + 1) * KT, 0);  // V_{j+1} into slot (j + 1) % 4 (held V_{j-3})
          if constexpr (s == 7) dma_v(KS, (tb + j + 1) * KT, 1);
          if constexpr (!FIRST && s >= 12) vpre[s - 12] = v_frag(VS, 0, s - 12);
        } else {
          if constexpr (s == 3) dma_k(PAR, (tb + j + 4) * KT, 0);
          if constexpr (s == 7) dma_k(PAR, (tb + j + 4) * KT, 1);
          if constexpr (s == 11) dma_v(VR, (tb + j + 2) * KT, 0);
          if constexpr (s == 15) dma_v(VR, (tb + j + 2) * KT, 1);
        }
        fence();
      });
    }
    // Rare wave-uniform branch between the two halves.  The empty asm statement keeps it a BRANCH: without a side effect hipcc
    // if-converts it into 32 x (v_add, v_cmp, v_cndmask) executed on every tile (96 VALU instructions per wave and tile, found in round 3)
    if (__builtin_expect(ragged && j + 2 == ntiles, 0)) {
      asm volatile("" ::: "memory");
      mask_tile(nxt, (j + 1) * KT);
    }
    if constexpr (MIDBAR) {
      // K_{j+2} and V_j have landed (own pieces; the barrier extends that to every wave).  Behind it every wave is done with K_j (read by
      // step j - 1) and K_{j+1} (read above): K_{j+4} goes into K_j's slot below; V_{j-1}, read below, is refilled two steps from now.
      wait_vm<2 * LPW2>();
      if (!(a.abl & 2)) __builtin_amdgcn_s_barrier();
    }
    constexpr int KSN = (PAR + 2) & 3;  // MIDBAR: slot of K_{j+2}, whose first fragments are read at the end of this step
    auto midbar_pv_gap = [&](auto SC) {
      constexpr int s = decltype(SC)::value;
      if constexpr (MIDBAR) {
        if constexpr (s == 3) dma_k(PAR, (tb + j + 4) * KT, 0);
        if constexpr (s == 7) dma_k(PAR, (tb + j + 4) * KT, 1);
        if constexpr (s >= 12) kpre[(s - 12) >> 1][(s - 12) & 1] = k_frag(KSN, (s - 12) >> 1, (s - 12) & 1);
      }
    };
    // -- C: O^T += V_{j-1}^T P_{j-1}^T, four independent accumulators; V fragments two ahead; the other 16 scores of P_j and the
    //       row max of S_{j+1} (two scores per gap, v_max3)
    float m0 = nxt[0][0];
    auto rmax = [&](auto SC) {  // two scores of S_{j+1} per gap (v_max3), pinned like the softmax work
      constexpr int s = decltype(SC)::value;
      asm volatile("" : "+v"(m0));
      m0 = fmaxf(fmaxf(m0, nxt[s >> 3][(2 * s) & 15]), nxt[s >> 3][((2 * s) & 15) + 1]);
      asm volatile("" : "+v"(m0));
    };
    if constexpr (!FIRST) {
      // V fragments VPF MFMAs ahead (PMC of the 2-ahead version: the waves sat a third of their cycles in s_waitcnt / s_barrier)
      constexpr int VPF = 4;
      v8bf vf[VPF + 1];
#pragma unroll
      for (int q = 0; q < VPF; ++q) vf[q] = MIDBAR ? vpre[q] : v_frag(VS, q >> 2, q & 3);
      fence();
      static_for<16>([&](auto SC) {
        constexpr int s = decltype(SC)::value, ch4 = s >> 2, db = s & 3;
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s % (VPF + 1)], __builtin_bit_cast(v8bf, pp[ch4]), o[db], 0, 0, 0);
        fence();
        if constexpr (s + VPF < 16) vf[(s + VPF) % (VPF + 1)] = v_frag(VS, (s + VPF) >> 2, (s + VPF) & 3);
        gapwork(std::integral_constant<int, 16 + s>{});
        rmax(SC);
        midbar_pv_gap(SC);
        fence();
      });
    } else {
      static_for<16>([&](auto SC) {
        constexpr int s = decltype(SC)::value;
        gapwork(std::integral_constant<int, 16 + s>{});
        rmax(SC);
        midbar_pv_gap(SC);
      });
    }
    mx = m0;
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using TT = std::true_type; using FF = std::false_type;
  // even j: cur = sa, nxt = sb, consumes pfa, produces pfb; odd j: the mirror image
  step(I0{}, TT{}, sa, sb, pfa, pfb, 0);
  int j = 1;
  for (; j + 4 <= ntiles; j += 4) {
    step(I1{}, FF{}, sb, sa, pfb, pfa, j);
    step(I2{}, FF{}, sa, sb, pfa, pfb, j + 1);
    step(I3{}, FF{}, sb, sa, pfb, pfa, j + 2);
    step(I0{}, FF{}, sa, sb, pfa, pfb, j + 3);
  }
  if (j < ntiles) { step(I1{}, FF{}, sb, sa, pfb, pfa, j); ++j; }
  if (j < ntiles) { step(I2{}, FF{}, sa, sb, pfa, pfb, j); ++j; }
  if (j < ntiles) { step(I3{}, FF{}, sb, sa, pfb, pfa, j); ++j; }

  // ---- drain: O^T += V_{n-1}^T P_{n-1}^T (the tile the last step produced) ---------------------------------------------------------
  stamp(6);
  wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    const int vs = (ntiles - 1) & 3;
    const bool odd_last = ((ntiles - 1) & 1) != 0;  // P of an odd tile was produced into pfa, of an even tile into pfb
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const v8bf vfr = *(const v8bf*)(smem + vx[s >> 2] + vs * V_BYTES + (s & 3) * 4096);
      const v4i pf = odd_last ? pfa[s >> 2] : pfb[s >> 2];
      o[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, __builtin_bit_cast(v8bf, pf), o[s & 3], 0, 0, 0);
    }
  }
  if (a.dbg) { asm volatile("" ::"v"(o[0][0]), "v"(o[3][15])); stamp(7); }
  const float l_part = (l4[0] + l4[1]) + (l4[2] + l4[3]);
  const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
  if (ntiles == ntiles_all) {  // the piece is the task
    const float inv = 1.0f / l_tot;
    store_o<FMT>(a, o, inv, b, h, qrow, hi);
  } else if constexpr (FOLD) {
    // ---- a piece of a task: (O, m, l) -> its slot; the piece that arrives last merges all pieces of the task in piece order -------------
    // Coherence.  The eight XCDs have private, mutually non-coherent L2s.  The textbook pattern (store, agent-scope release fence, counter,
    // acquire fence, load) costs a buffer_wbl2 + buffer_inv sc1 per piece, and the invalidate empties the L2 under the 31 other workgroups
    // of the XCD that are streaming K / V^T through it: 347 us per launch instead of 225.  Agent-scope atomic accesses (sc1: every word to
    // the memory side and back) are correct everywhere but put ~27 us of round trips behind the last tile (234 us: no gain left).  What is
    // used instead is the XCD's own L2: ALL pieces of a task run on ONE XCD (workgroup b runs on XCD b % 8 and the task index is derived
    // from b & 7; the host verifies that mapping once per device before it enables the balanced grid, fluxmi_xcd_mapping_ok), and within
    // an XCD the L2 is the single point of coherence of its 32 CUs -- the vector L1 is write-through, a store is acknowledged (vmcnt) when
    // the L2 has it, and a CU's L1 cannot hold a stale copy of a slot: it is invalidated at every kernel start and each slot line is read
    // exactly once per launch, by the merging workgroup (tests/test_ops_gpu.py alternates inputs between launches to catch a stale read).
    // So: plain 16-byte stores, s_waitcnt vmcnt(0), workgroup barrier, ONE agent-scope atomic add per piece, plain loads.
    // (profiles/r05_attention_split.txt has the three variants measured.)
    const int x = (int)blockIdx.x & 7;
    float* slot0 = a.sp.part + (size_t)(x * ATTN_MAX_PIECES + pc.base) * ATTN_PART_FLOATS;  // the task's first piece; its others follow
    {
      v4f* po = (v4f*)(slot0 + (size_t)pc.pidx * ATTN_PART_FLOATS) + (wave * 17) * 64 + lane;  // [wave][i < 16: four O floats | 16: (m, l, -, -)][lane]
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v4f v;
        v[0] = o[i >> 2][(i & 3) * 4 + 0]; v[1] = o[i >> 2][(i & 3) * 4 + 1]; v[2] = o[i >> 2][(i & 3) * 4 + 2]; v[3] = o[i >> 2][(i & 3) * 4 + 3];
        po[i * 64] = v;
      }
      v4f ml;
      ml[0] = m_cur; ml[1] = l_tot; ml[2] = 0.f; ml[3] = 0.f;
      po[16 * 64] = ml;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store of this wave has reached the L2 ...
    __syncthreads();                                   // ... of every wave, before the arrival is counted
    unsigned* cnt = a.sp.cnt + x * ATTN_SPLIT_MAXT + pc.tloc;
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *(volatile int*)smem = old == (unsigned)(pc.np - 1) ? 1 : 0;
    }
    __syncthreads();
    if (*(volatile int*)smem) {
      asm volatile("" ::: "memory");
      const int np = pc.np;
      constexpr size_t SLOT_V4 = ATTN_PART_FLOATS / 4;
      const v4f* p0 = (const v4f*)slot0 + (wave * 17) * 64 + lane;
      float M = -3.0e38f;
      for (int pi = 0; pi < np; ++pi) M = fmaxf(M, p0[pi * SLOT_V4 + 16 * 64][0]);
      v16f acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      float lsum = 0.f;
      // two pieces in flight: the loads of piece pi + 1 are issued before the multiply-adds of piece pi (fixed summation order)
      v4f wa[17], wb[17];
      auto ld = [&](v4f (&w)[17], int pi) {
#pragma unroll
        for (int i = 0; i < 17; ++i) w[i] = p0[pi * SLOT_V4 + i * 64];
      };
      auto fm = [&](const v4f (&w)[17]) {
        const float f = __builtin_amdgcn_exp2f(w[16][0] - M);
        lsum = __builtin_fmaf(w[16][1], f, lsum);
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i >> 2][(i & 3) * 4 + e] = __builtin_fmaf(w[i][e], f, acc[i >> 2][(i & 3) * 4 + e]);
      };
      ld(wa, 0);
      for (int pi = 0; pi < np; pi += 2) {
        if (pi + 1 < np) ld(wb, pi + 1);
        fm(wa);
        if (pi + 1 < np) {
          if (pi + 2 < np) ld(wa, pi + 2);
          fm(wb);
        }
      }
      if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero between launches
      store_o<FMT>(a, acc, 1.0f / lsum, b, h, qrow, hi);
    }
  }
  if (a.dbg && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores of this workgroup's first wave are out
    unsigned long long* d = a.dbg + (size_t)blockIdx.x * 8;
    d[0] = blockIdx.x;
    d[1] = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) << 8);
    d[2] = t_dbg0;
    d[3] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace

// ---- balanced grid: plan + scratch -----------------------------------------------------------------------------------------------------
// first tile (in one XCD's leftover tile sequence [0, T)) of bin k: equal shares, edges within ATTN_SPLIT_SNAP tiles of a task edge moved onto it
static int attn_bin_start(int k, int T, int nb, int ntiles) {
  int s = (int)((long long)k * T / nb);
  const int r = s % ntiles;
  if (r <= ATTN_SPLIT_SNAP) s -= r;
  else if (ntiles - r <= ATTN_SPLIT_SNAP) s += ntiles - r;
  return s;
}
// Pure arithmetic on (tasks, tiles per task).  Off unless the leftover round is worth splitting: tasks fill the XCDs evenly, 1 <= rem <= 26
// of an XCD's 32 CUs (a fuller last round gains < 10 %), >= 16 key tiles per task, bins of >= 8 tiles, no task in more than
// ATTN_SPLIT_MAXP pieces.
AttnSplit fluxmi_attn_plan(int tasks, int ntiles, int cus) {
  AttnSplit sp;
  memset(&sp, 0, sizeof(sp));
  if (cus != 256 || tasks % 8 || ntiles < 16 || ntiles > 65535) return sp;
  const int n = tasks / 8, last = n % 32;  // tasks per XCD; tasks in the last round of its 32 CUs
  if (last == 0 || last > 26) return sp;
  // which tasks are binned: a THIN last round (<= 8 of 32 CUs) is folded into the full round in front of it -- 32 + last tasks over 32 bins, every
  // CU gets (32 + last) / 32 tasks' worth of key tiles in one go (Flux-dev 768^2: 33 tasks of 44 tiles -> 45.4 tiles per CU instead of 44 + 11);
  // a single partial round (n < 32: 512^2 runs 18 tasks on 32 CUs) is spread over all CUs; a fuller last round is binned on its own
  // (a single partial round does NOT pay: 512^2, 18 tasks on 32 CUs, 40 -> 51 us per launch, +7 % per step -- the second prologue, the
  // hand-over and the merge cost a piece ~13 us, a third of such a task; it stays available under attn_split = 2)
  const bool thin = n >= 32 && last <= 8;
  const int rem = n < 32 ? n : (last <= 8 ? 32 + last : last);
  if (rem > ATTN_SPLIT_MAXT) return sp;
  const int T = rem * ntiles;
  int nb = std::min(32, rem * 4);
  while (nb > 1 && T / nb < 8) --nb;
  if (T / nb < 8 || nb >= rem * 4 + 1) return sp;
  if (nb == rem && n >= 32 && !(last <= 8)) return sp;  // nothing to balance: one bin per task
  // canonical pieces (ascending tile position) and, per piece, whether it is the first of its bin
  struct P { int tloc, tb, len, first; };
  std::vector<P> ps;
  for (int k = 0; k < nb; ++k) {
    int pos = attn_bin_start(k, T, nb, ntiles);
    const int end = k + 1 == nb ? T : attn_bin_start(k + 1, T, nb, ntiles);
    for (int first = 1; pos < end; first = 0) {
      const int t = pos / ntiles, e = std::min(end, (t + 1) * ntiles);
      ps.push_back(P{t, pos - t * ntiles, e - pos, first});
      pos = e;
    }
  }
  if ((int)ps.size() > ATTN_MAX_PIECES) return sp;
  std::vector<AttnPiece> canon(ps.size());
  for (size_t c = 0; c < ps.size(); ++c) {
    size_t b = c, e = c;
    while (b > 0 && ps[b - 1].tloc == ps[c].tloc) --b;
    while (e + 1 < ps.size() && ps[e + 1].tloc == ps[c].tloc) ++e;
    if (e - b + 1 > (size_t)ATTN_SPLIT_MAXP) return sp;
    canon[c] = AttnPiece{(unsigned char)ps[c].tloc, (unsigned char)(c - b), (unsigned char)(e - b + 1), (unsigned char)b, (unsigned short)ps[c].tb,
                         (unsigned short)ps[c].len};
  }
  // launch order: the bins' first pieces, longest first, then the others, longest first (stable: equal lengths keep their tile order)
  std::vector<int> order(ps.size());
  for (size_t c = 0; c < ps.size(); ++c) order[c] = (int)c;
  std::stable_sort(order.begin(), order.end(), [&](int u, int v) {
    if (ps[u].first != ps[v].first) return ps[u].first > ps[v].first;
    return ps[u].len > ps[v].len;
  });
  for (size_t c = 0; c < ps.size(); ++c) sp.pieces[c] = canon[order[c]];
  bool any_split = false;
  for (const AttnPiece& c : canon) any_split |= c.np > 1;
  if (!any_split) return sp;
  sp.on = 1; sp.thin = thin; sp.n_per_x = n; sp.full_per_x = n - rem; sp.npieces = (int)ps.size();
  return sp;
}

namespace {
// scratch of the balanced grid: partial (O, m, l) slots + the arrival counters (zeroed once, kept zero by the kernel).  Per (device, stream)
// like the split-K scratch (gemm_pp.hip), never allocated under stream capture -- the launch then simply runs the unsplit grid.
thread_local void* t_attn_override = nullptr;
std::mutex g_attn_mu;
std::map<std::pair<int, hipStream_t>, void*> g_attn_ws;
void* attn_workspace(hipStream_t s) {
  if (t_attn_override) return t_attn_override;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_attn_mu);
  auto it = g_attn_ws.find({dev, s});
  if (it == g_attn_ws.end()) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &cap) != hipSuccess) return nullptr;
    if (cap != hipStreamCaptureStatusNone) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, ATTN_SPLIT_WS_BYTES) != hipSuccess || hipMemset(p, 0, ATTN_SPLIT_WS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    it = g_attn_ws.emplace(std::make_pair(dev, s), p).first;
  }
  return it->second;
}
}  // namespace
void fluxmi_set_attn_scratch(void* p) { t_attn_override = p; }
static unsigned long long* g_attn_dbg = nullptr;  // fluxmi_attention_debug_buffer (probes)
int fluxmi_attn_debug_buffer(void* p) { g_attn_dbg = (unsigned long long*)p; return 0; }

// The merge of the balanced grid goes through ONE XCD's L2 (see the kernel): it needs workgroup b of a 1-D grid to run on XCD b % 8, the
// same for every launch.  Checked once per device on the hardware itself (512 one-wave workgroups record HW_REG_XCC_ID): 1 = holds, 0 = not
// (the balanced grid stays off on that device), -1 = not known yet and the calling stream is capturing (this launch runs unsplit).
namespace {
__global__ void xcc_probe_kernel(unsigned* out) { out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }
std::mutex g_xcc_mu;
std::map<int, int> g_xcc_ok;
}  // namespace
int fluxmi_xcd_mapping_ok(hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lk(g_xcc_mu);
  auto it = g_xcc_ok.find(dev);
  if (it != g_xcc_ok.end()) return it->second;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (s && (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) return -1;
  constexpr int N = 512;
  unsigned* d = nullptr;
  std::vector<unsigned> h(N, 99u);
  int ok = 0;
  if (hipMalloc((void**)&d, N * sizeof(unsigned)) == hipSuccess) {
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(N), dim3(64), 0, 0, d);
    if (hipMemcpy(h.data(), d, N * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess) {
      ok = 1;
      for (int b = 0; b < N; ++b) ok = ok && h[b] == h[b & 7];                                   // a function of b % 8 ...
      for (int u = 0; u < 8; ++u)
        for (int v = u + 1; v < 8; ++v) ok = ok && h[u] != h[v];                                 // ... onto eight distinct XCDs
    }
    hipFree(d);
  }
  (void)hipGetLastError();
  g_xcc_ok[dev] = ok;
  return ok;
}
// The plan is made PER SAMPLE (tasks of one batch element: row blocks x heads) and a batch is launched sample by sample when it is on, so the
// pieces a (sample, head, row block) is cut into -- and with them the bits of its output -- do not depend on the batch the sample rides in
// (round 5 planned over B x H x row blocks: at 768^2 a sample's latents depended on its batch by <= 2.5e-3; reference flux_model.py:672-716
// has no cross-sample operation).  B only has to be valid.
int fluxmi_attn_plan_any(int B, int L, int H) {
  return B >= 1 && fluxmi_attn_plan(((L + 255) / 256) * H, (L + KT - 1) / KT, 256).on && fluxmi_xcd_mapping_ok(nullptr) == 1;
}
int fluxmi_attn_plan_export(int B, int L, int H, int* n_per_x, int* full_per_x, int* npieces, unsigned long long* pieces) {
  if (B < 1 || L < 1 || H < 1) return 0;
  const AttnSplit sp = fluxmi_attn_plan(((L + 255) / 256) * H, (L + KT - 1) / KT, 256);
  if (!sp.on) return 0;
  if (n_per_x) *n_per_x = sp.n_per_x;
  if (full_per_x) *full_per_x = sp.full_per_x;
  if (npieces) *npieces = sp.npieces;
  if (pieces) memcpy(pieces, sp.pieces, sizeof(AttnPiece) * sp.npieces);
  return sp.thin ? 1 : 2;
}

template <bool FOLD, bool EXACT, bool MIDBAR = false> static int launch2(AttnArgs a, int fmt, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)attention2_kernel<FLUXMI_FMT_E5M2, FOLD, EXACT, MIDBAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * A_STAGE));
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)attention2_kernel<FLUXMI_FMT_E4M3, FOLD, EXACT, MIDBAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * A_STAGE));
    attr = true;
  }
  memset(&a.sp, 0, sizeof(a.sp));
  a.dbg = g_attn_dbg;
  const int tasks1 = ((a.L + 255) / 256) * a.H;  // of one sample
  const fluxmi_tuning_t tun = fluxmi_tuning();
  auto launch = [&](const AttnArgs& aa, int wgs) {
    const dim3 grid(wgs);
    if (fmt == FLUXMI_FMT_E5M2) hipLaunchKernelGGL((attention2_kernel<FLUXMI_FMT_E5M2, FOLD, EXACT, MIDBAR>), grid, dim3(512), 4 * A_STAGE, s, aa);
    else hipLaunchKernelGGL((attention2_kernel<FLUXMI_FMT_E4M3, FOLD, EXACT, MIDBAR>), grid, dim3(512), 4 * A_STAGE, s, aa);
  };
  if (FOLD && tun.attn_split && fluxmi_xcd_mapping_ok(s) == 1) {
    AttnSplit sp = fluxmi_attn_plan(tasks1, (a.L + KT - 1) / KT, 256);  // per SAMPLE: see fluxmi_attn_plan_any
    // attn_split = 1: only THIN last rounds (at most 8 of an XCD's 32 CUs busy, folded into the round in front of them; or a single partial
    // round).  Fuller ones were measured not to pay on this chip -- Flux-dev 1024^2, 22 of 32: 236 vs 223 - 235 us isolated, +3.4 % per
    // step (profiles/r05_attention_split.txt); 2 forces the balanced grid wherever a plan exists (tests, probes)
    if (sp.on && tun.attn_split == 1 && !sp.thin) sp.on = 0;
    void* ws = sp.on ? attn_workspace(s) : nullptr;
    if (sp.on && ws) {
      sp.part = (float*)ws;
      sp.cnt = (unsigned*)((char*)ws + (size_t)8 * ATTN_MAX_PIECES * ATTN_PART_FLOATS * 4);  // [8][ATTN_SPLIT_MAXT]
      // one launch per sample, each exactly the B = 1 launch of that sample (stream order keeps the scratch slots private to a launch)
      for (int b = 0; b < a.B; ++b) {
        AttnArgs ab = a;
        ab.B = 1;
        ab.sp = sp;
        ab.pf.n = 0;  // no CU idles in the last round any more: nothing for the weight prefetch to ride on
        ab.pf.wgs = 0;
        const long long bl = (long long)b * a.L;
        if (a.Q) ab.Q = a.Q + bl * a.H * 128;
        ab.K = a.K + bl * a.H * 128;
        ab.VT = a.VT + (long long)b * a.H * 128 * a.Lp;
        if (a.qraw) { ab.qraw = a.qraw + bl * a.ldq; ab.pe = a.pe + bl * 128; }
        ab.out = (char*)a.out + bl * a.ld_out * (a.out_fp8 ? 1 : 2);
        launch(ab, 8 * (sp.full_per_x + sp.npieces));
      }
      FLUXMI_LAUNCH_CHECK();
      return 0;
    }
  }
  launch(a, tasks1 * a.B + (a.pf.n > 0 ? a.pf.wgs : 0));
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

// The kernel follows the K operand: fp16 K (AttnArgs.k_f16, produced by fluxmi_qkv_rope(k_f16 = 1)) -> the folded kernel with its
// per-tile barrier between the two MFMA groups (round 3's "mid-barrier" variant: fragments of the next group are read under the last
// MFMAs of the current one; bit-identical to the barrier-at-step-start build it replaced, -1.6 %), bf16 K -> the unfolded kernel.
// `exact` (fluxmi_tuning_t.attn_var bit 1) selects exact instead of deferred max tracking (tests).
int fluxmi_launch_attention2(const AttnArgs& a, int fmt, hipStream_t s, bool exact) {
  if (a.k_f16) return exact ? launch2<true, true, true>(a, fmt, s) : launch2<true, false, true>(a, fmt, s);
  return exact ? launch2<false, true>(a, fmt, s) : launch2<false, false>(a, fmt, s);
}
