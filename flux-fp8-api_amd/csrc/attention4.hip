// fluxmi -- flash-attention forward, round-3 kernel: FOUR waves per workgroup, one per SIMD, 64 query rows per wave (head_dim 128,
// non-causal), gfx950.   Reference: `attention` flux_model.py:60-65 (F.scaled_dot_product_attention on bf16 q / k / v).
//
// STATUS: selectable (FLUXMI_ATTN_V=4), parity-tested, NOT the engine's default.  Measured on MI355X (profiles/r03_attention4.txt): at
// B = 1, H = 24, L = 4608 it ran 226 us against 233 us for the 8-wave kernel of that day on scores of unit spread (+3.5 %; the 8-wave
// kernel has since taken over its per-lane rescale decision and runs 225-229 us), but every rescale of the deferred running max costs
// it ~2000 cycles (128 O registers per wave behind v_accvgpr_read / write, and nothing else on the SIMD to hide them): 237 vs 229 us at
// a score spread of 3.2 (exp2 domain), 328 vs 255 us at 9; inside the denoise step the two are within 0.2 % (44.41 vs 44.32 ms/step).
// This kernel is bound by VALU / LDS / LDS-DMA ISSUE as much as by the matrix pipe: per 32 MFMAs (1024 cycles) its lone wave issues
// 32 v_exp (8.6 cycles each: tools/probes/issue_probe.hip), 32 v_add, 16 v_max3, 16 v_cvt_pk, 16 ds_read_b128 (16 cycles each) and
// 4 LDS-DMA (~39) = ~1000 cycles, and the chip runs the loop at ~1.67 GHz (power); MFMA-busy is 71 % of a workgroup's life (8-wave
// kernel: 74 %, held there by its per-tile rendezvous, not by instruction count: profiles/r03_attention4.txt).
//
// Why: the 8-wave kernel (attention2.hip, 32 rows per wave) reads every K and V^T fragment from LDS once per 32 query rows: 8 waves x
// 32 KiB = 256 KiB of ds_read_b128 traffic per 64-key tile and CU, half of the LDS read rate for the 2 x 1024 MFMA cycles the two waves
// of a SIMD need for that tile (MFMA busy 57 %, a third of the wave cycles in s_waitcnt: profiles/r02_attention_pmc.txt).  Here a wave
// owns TWO 32-row query blocks and the whole 512-register file of its SIMD: every K / V^T fragment it reads feeds two MFMAs (the same
// fragment against both query blocks), so the LDS traffic per MFMA halves, and the wave's own instruction stream keeps the matrix pipe
// fed -- one independent MFMA after another, <= 5 other issues per gap (MI355X_MICROARCH.md, "one wave per SIMD").
//
// Pipeline unit = HALF a K/V tile (32 keys), so that the skewed state of two query blocks fits the register file without spills:
// half-step h runs  S_{h+1} = K_{h+1} Q^T (16 MFMAs: 8 head-dim chunks x 2 query blocks),  P_h = softmax numerators of S_h (32 scores
// per lane, one per MFMA gap)  and  O^T += V_{h-1}^T P_{h-1}^T (16 MFMAs: 2 key slices x 4 head-dim blocks x 2 query blocks); the row
// max of S_{h+1} (v_max3, two scores each) rides in the PV gaps.  Live per lane: O 128 + Q 64 (AGPRs), S 2 x 32, P 2 x 16, -M 32, K /
// V^T fragments 28 (arch VGPRs).  The LDS image (XOR-swizzled K rows / V^T rows, 4-deep LDS-DMA rings of 64-key tiles), the deferred
// running max and the folded arithmetic (softmax scale * log2 e in the fp16 Q fragments, -max as the C operand of the first MFMA of
// every score block, f16 MFMAs for QK^T; K arrives as fp16: AttnArgs.k_f16) are attention2.hip's; one barrier per 64-key tile.
//
// The MFMAs and the softmax VALU work of the loop are inline asm: (1) the register FILE of every operand is chosen here, not by the
// allocator (left to hipcc, Q fragments went to scratch and every score was copied out of an AGPR); (2) every instruction of a gap is
// issued in the order written, and nothing in a gap reads what the gap's v_exp_f32 wrote (a trans result needs one wait state before a
// VALU read; hipcc pads with s_nop, which a lone wave cannot hide): see gapwork4.  The compiler sees no MFMA there, so the MFMA -> VALU
// read distance is kept by construction (a score block is read >= 16 MFMAs after its last accumulation; the rare rescale / mask
// branches pad with s_nop).  Nothing of a half-step waits on LDS latency with the matrix pipe idle: the first K fragments of half-step
// h + 1 are read under the last PV MFMAs of half-step h, the first V^T fragments under the last QK^T MFMAs, the rescale decision (on the
// per-lane part of the row max; the cross-lane finish lives in the cold branch) is taken two gaps before the branch that reads it; the
// one barrier per 64-key tile sits between the two MFMA groups of the even half-step, where the next tiles' visibility is needed neither
// by the fragments already in registers nor by the refills issued next.
#include "attention_common.h"

namespace {

typedef _Float16 v8h4 __attribute__((ext_vector_type(8)));

template <int N, class F> __device__ __forceinline__ void static_for4(F&& f) {
  if constexpr (N > 0) {
    static_for4<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

__device__ __forceinline__ void fence4() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// S block = K fragment . Q fragment + C (C = sixteen copies of -M: the running max leaves through the accumulator init)
__device__ __forceinline__ void mfma_qk0(v16f& acc, const v8bf& kf, const v8bf& qf, const v16f& ninit) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(kf), "a"(qf), "v"(ninit));
}
__device__ __forceinline__ void mfma_qk(v16f& acc, const v8bf& kf, const v8bf& qf) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(kf), "a"(qf));
}
// O^T block += V^T fragment . P fragment
__device__ __forceinline__ void mfma_pv(v16f& acc, const v8bf& vf, const v4i& pf) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(vf), "v"(pf));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }  // > 11 wait states after an 8-pass MFMA

// Softmax work of MFMA gap n = 0..31 of a half-step: score m is register m >> 1 of query block m & 1 (S_h, turned into P_h in place).
//   exp2:      gap 0: scores 0, 2; gap 1: 1, 3; gap n <= 29: score n + 2  (so that the last two gaps exponentiate nothing and the
//              half-step has no tail: everything of P_h is packed when its last MFMA issues)
//   row sums:  gap 2: scores 0, 2; gap 3: 1, 3; gap n >= 4: score n        (two gaps behind its exp2)
//   cvt_pk:    gap n with (n >> 1) odd: the bf16 pair (n - 2, n) = registers (r - 1, r), r = n >> 1, of block n & 1
//   row max of S_{h+1}: gaps 16..29, one v_max3 each (two in the last pair), per lane = over the 16 keys a lane holds; the rescale
//              DECISION needs no more than that (some row exceeds the threshold iff some lane does) and is taken in gap 30, two
//              gaps before the branch that reads it; the cross-lane finish (v_permlane32_swap) happens inside the rare branch
#define SREG4(arr, n) arr[(n) & 1][(n) >> 1]
template <int n>
__device__ __forceinline__ void gapwork4(v16f (&cur)[2], v16f (&nxt)[2], v4i (&pc)[2][2], float (&l2)[2][2], float (&m0)[2]) {
  constexpr int x = n & 1;
  if constexpr (n >= 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l2[x][(n >> 1) & 1]) : "v"(SREG4(cur, n)));
  if constexpr (n == 2 || n == 3) {
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(l2[x][0]) : "v"(SREG4(cur, n - 2)));
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(l2[x][1]) : "v"(SREG4(cur, n)));
  }
  if constexpr (n <= 1) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(SREG4(cur, n)), "+v"(SREG4(cur, n + 2)));
  else if constexpr (n <= 29) asm volatile("v_exp_f32 %0, %0" : "+v"(SREG4(cur, n + 2)));
  if constexpr (n >= 16 && n < 30) {  // row max of S_{h+1} (per-lane part): 16 x v_max3 over gaps 16..29 (two in gaps 28, 29), block g & 1
    constexpr int g = n - 16, xb = g & 1, s2 = g >> 1;  // registers 2 s2, 2 s2 + 1; gaps 28 / 29 also take registers 14, 15
    if constexpr (s2 == 0) asm volatile("v_max_f32 %0, %1, %2" : "=v"(m0[xb]) : "v"(nxt[xb][0]), "v"(nxt[xb][1]));
    else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0[xb]) : "v"(nxt[xb][2 * s2]), "v"(nxt[xb][2 * s2 + 1]));
    if constexpr (s2 == 6) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0[xb]) : "v"(nxt[xb][14]), "v"(nxt[xb][15]));
  }
  if constexpr (n >= 2 && ((n >> 1) & 1)) {
    constexpr int r = n >> 1;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pc[x][r >> 3][(r & 7) >> 1]) : "v"(SREG4(cur, n - 2)), "v"(SREG4(cur, n)));
  }
}

constexpr int NW4 = 4, RD4 = 4, LPW4 = 16 / NW4;
constexpr int VRING4 = RD4 * K_BYTES;

// ABL (timing-only ablations, compiled only with -DFLUXMI_ATTN4_ABLATIONS and selected by FLUXMI_ATTN4_ABL): 1 no softmax VALU work, 2 no LDS-DMA refills, 4 no barrier / vmcnt wait, 8 no fragment
// reads in the loop, 16 no cross-lane max finish / rescale decision, 32 no MFMAs
template <int FMT, bool EXACT, int ABL = 0>
__global__ void __launch_bounds__(NW4 * 64, 1) attention4_kernel(const AttnArgs a) {
  constexpr int QB = NW4 * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqb = (a.L + QB - 1) / QB;
  const int lid = xcd_remap(blockIdx.x, nqb * a.H * a.B);  // whole heads per XCD
  const int bhid = lid / nqb;
  const int h = bhid % a.H, b = bhid / a.H;
  const int q0 = (lid - bhid * nqb) * QB + wave * 64;
  const long long bh = (long long)b * a.H + h;

  // ---- Q fragments of both query blocks (x = 0: rows q0 + l31, x = 1: rows q0 + 32 + l31): fp16, times scale * log2 e -----------------
  v8bf qf[2][8];
#pragma unroll
  for (int x = 0; x < 2; ++x) load_q_frags<true>(a, b, h, min(q0 + x * 32 + l31, a.L - 1), hi, a.scale_log2, qf[x]);

  // ---- LDS-DMA: 16 pieces of 1 KiB per K tile and per V^T tile, four of each per wave.  Piece i is rows 16 i .. of the K tile (32 i ..
  // of the V^T tile): the per-lane offset is the same for every piece (the swizzle looks at the row modulo 16), the piece goes into the
  // wave-uniform offset.
  const auto krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.K + bh * a.L * 128), 0, a.L * 256, 0x00020000);
  const auto vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.VT + bh * 128 * a.Lp), 0, 128 * a.Lp * 2, 0x00020000);
  const unsigned k_off = (unsigned)((tid >> 4) * 256 + ((tid & 15) ^ ((tid >> 4) & 15)) * 16);
  const unsigned v_off = (unsigned)((tid >> 3) * a.Lp * 2 + ((tid & 7) ^ ((tid >> 4) & 7)) * 16);
  const int v_piece = 32 * a.Lp * 2;
  auto dma_k = [&](int slot, int tile, int i) __attribute__((always_inline)) { dma16(krsrc, smem + slot * K_BYTES + wave * 1024 + NW4 * 1024 * i, k_off, tile * (KT * 256) + i * 4096); };
  auto dma_v = [&](int slot, int tile, int i) __attribute__((always_inline)) { dma16(vrsrc, smem + VRING4 + slot * V_BYTES + wave * 1024 + NW4 * 1024 * i, v_off, tile * (KT * 2) + i * v_piece); };
  // prologue: K0 | K1 | K2 V0 | K3 V1 in flight; tile step j then issues K_{j+4} V_{j+2} behind its barrier.  Every issue is unconditional
  // (a tile past the end reads zeros through the descriptor's bounds check, or bytes nobody uses), so the vmcnt arithmetic is the same
  // in every step.
  {
    auto iss_k = [&](int t) {
#pragma unroll
      for (int i = 0; i < LPW4; ++i) dma_k(t, t, i);
    };
    auto iss_v = [&](int t) {
#pragma unroll
      for (int i = 0; i < LPW4; ++i) dma_v(t, t, i);
    };
    iss_k(0); iss_k(1); iss_k(2); iss_v(0); iss_k(3); iss_v(1);
  }

  v16f o[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[x][i][r] = 0.f;
  // The loop's MFMAs are invisible to the compiler: it would materialise these zeros (and the AGPR copies of the Q fragments) lazily, in
  // front of their first use, and pad no wait states between a v_accvgpr_write and the MFMA reading it.  Pin them here, far ahead.
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(o[x][i]));
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) asm volatile("" : "+a"(qf[x][cc]));
  }
  float l2[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  float m_cur[2];  // deferred running max M of each block's rows (exp2 domain)
  v16f ninit[2];   // sixteen copies of -M: the C operand of the first QK^T MFMA of every score block

  // fragment addresses as 32-bit LDS offsets: register = lane part, everything else goes into the ds_read offset field
  typedef __attribute__((address_space(3))) const v8bf* lds_frag_p;
  const unsigned smem_base = (unsigned)(size_t)(lds_ptr)smem;
  unsigned kx[8], vx[4];
  {
    const int sw = l31 & 15, vsw = (l31 >> 1) & 7;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) kx[cc] = smem_base + (unsigned)(l31 * 256 + (((cc * 2 + hi) ^ sw) << 4));
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) vx[ch] = smem_base + (unsigned)(VRING4 + l31 * 128 + (((ch * 2 + hi) ^ vsw) << 4));
  }
  // K fragment: 32 keys (half t of the tile in `slot`) x head-dim chunk cc; V^T fragment: head-dim block db x key slice ch4 (16 keys)
  auto k_frag = [&](int slot, int cc, int t) __attribute__((always_inline)) -> v8bf { return *(lds_frag_p)(size_t)(kx[cc] + (unsigned)(slot * K_BYTES + t * (32 * 256))); };
  auto v_frag = [&](int slot, int ch4, int db) __attribute__((always_inline)) -> v8bf { return *(lds_frag_p)(size_t)(vx[ch4] + (unsigned)(slot * V_BYTES + db * 4096)); };

  auto mask_half = [&](v16f (&st)[2], int key0) __attribute__((always_inline)) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        st[x][r] = key < a.L ? st[x][r] : -1e30f;
      }
  };
  auto finish_max = [&](float mx) __attribute__((always_inline)) -> float {  // the other 16 keys of the row sit in lane ^ 32
    const unsigned u = __float_as_uint(mx);
    const auto sw2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(sw2[0]), __uint_as_float(sw2[1]));
  };

  const int ntiles = (a.L + KT - 1) / KT;
  const int mask_from = a.L >> 5;  // first half-tile with keys >= L

  // ---- S_0 (keys 0..31), its row max, M = that max -------------------------------------------------------------------------------------
  wait_vm<5 * LPW4>();
  __builtin_amdgcn_s_barrier();
  v16f sa[2], sb[2];   // score blocks of the two query blocks: even half-steps turn sa into P and fill sb, odd ones the reverse
  v4i pa[2][2], pb[2][2];  // bf16 P fragments (two 16-key slices per block): produced by even / odd half-steps
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { pa[x][i][e] = 0; pb[x][i][e] = 0; }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[x][r] = 0.f;
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const v8h4 kf = __builtin_bit_cast(v8h4, k_frag(0, cc, 0));
    sa[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, __builtin_bit_cast(v8h4, qf[0][cc]), sa[0], 0, 0, 0);
    sa[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, __builtin_bit_cast(v8h4, qf[1][cc]), sa[1], 0, 0, 0);
  }
  if (mask_from == 0) mask_half(sa, 0);
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    float mi = sa[x][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mi = fmaxf(mi, sa[x][r]);
    m_cur[x] = finish_max(mi);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sa[x][r] -= m_cur[x];
      ninit[x][r] = -m_cur[x];
    }
  }
  asm volatile("" : "+v"(ninit[0]), "+v"(ninit[1]));  // thirty-two live registers, not a splat hipcc re-materialises

  // ---- half-step h = 2 j + T of tile step j (PAR = j % 4, compile time); FIRST: no pending P (h == 0) ---------------------------------
  // cur = S_h (turned into P_h in place), nxt = S_{h+1}, pp = fragments of P_{h-1} (consumed), pc = of P_h (produced)
  v8bf kpre[3], vpre[3];  // first fragments of the next MFMA group, read one group ahead
  bool need = false;      // some row max of S_h exceeds the deferred threshold (decided at the end of half-step h - 1)
  float m0[2] = {0.f, 0.f};  // per-lane part of the row max of S_h - M (the other 16 keys of the row sit in lane ^ 32)
  kpre[0] = k_frag(0, 0, 1);
  kpre[1] = k_frag(0, 1, 1);
  kpre[2] = k_frag(0, 2, 1);
  auto half_step = [&](auto PARC, auto TC, auto FIRSTC, auto MASKC, v16f (&cur)[2], v16f (&nxt)[2], v4i (&pp)[2][2], v4i (&pc)[2][2], int j) __attribute__((always_inline)) {
    constexpr int PAR = decltype(PARC)::value, T = decltype(TC)::value;
    constexpr bool FIRST = decltype(FIRSTC)::value, MASKCHK = decltype(MASKC)::value;
    // T = 0: S_{h+1} is key half 1 of tile j; P_{h-1} is key half 1 of tile j - 1.   T = 1: key half 0 of tile j + 1; key half 0 of tile j
    constexpr int KS = T ? (PAR + 1) & 3 : PAR, KH = T ? 0 : 1;
    constexpr int VS = T ? PAR : (PAR + 3) & 3, VH = T ? 0 : 1;
    constexpr int KSN = (PAR + 1) & 3, KHN = T ? 1 : 0;  // K fragments of half-step h + 1: half 0 (after T = 0) / half 1 of tile j + 1
    // -- A: deferred running max (wave-uniform branch, out of the steady state): mx = row max of S_h - M
    if (__builtin_expect(need, 0)) {
      mfma_drain();  // O was last written by the asm MFMAs of the previous half-step
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const float delta = fmaxf(finish_max(m0[x]), 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l2[x][0] *= alpha;
        l2[x][1] *= alpha;
        // one register at a time (read, multiply, write back, scheduling fence): the allocator sizes the loop's register budget for this
        // branch too, and with 128 temporaries here it parked the loop's LDS addresses in AGPRs (one v_accvgpr_read per ds_read)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float t = o[x][i][r];
            asm volatile("" : "+v"(t));
            t *= alpha;
            asm volatile("" : "+v"(t));
            o[x][i][r] = t;
            fence4();
          }
        if constexpr (!FIRST) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned w = (unsigned)pp[x][i][e];
              pp[x][i][e] = (int)pack_bf2(__uint_as_float(w << 16) * alpha, __uint_as_float(w & 0xffff0000u) * alpha);
            }
        }
        m_cur[x] += delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          cur[x][r] -= delta;  // S_h was produced with the old M
          ninit[x][r] = -m_cur[x];
        }
      }
      asm volatile("" : "+v"(ninit[0]), "+v"(ninit[1]));
    }
    // -- B: S_{h+1} = K_{h+1} Q^T: gap g = (chunk cc, block x); every K fragment feeds both blocks; fragments three chunks ahead
    {
      v8bf kf[4];
      kf[0] = kpre[0];
      kf[1] = kpre[1];
      kf[2] = kpre[2];
      fence4();
      static_for4<16>([&](auto GC) {
        constexpr int g = decltype(GC)::value, x = g & 1, cc = g >> 1;
        if constexpr (!(ABL & 32)) {
          if constexpr (cc == 0) mfma_qk0(nxt[x], kf[0], qf[x][0], ninit[x]);
          else mfma_qk(nxt[x], kf[cc % 4], qf[x][cc]);
        }
        fence4();
        if constexpr (!(ABL & 8)) {
          if constexpr (x == 0 && cc + 3 < 8) kf[(cc + 3) % 4] = k_frag(KS, cc + 3, KH);
          if constexpr (!FIRST && g >= 11 && x == 1) vpre[(g - 11) >> 1] = v_frag(VS, 2 * VH, (g - 11) >> 1);  // gaps 11, 13, 15
        }
        if constexpr (!(ABL & 1)) gapwork4<g>(cur, nxt, pc, l2, m0);
        if constexpr (!(ABL & 2) && T == 1 && (g & 3) == 3) dma_v((PAR + 2) & 3, j + 2, g >> 2);
        fence4();
      });
    }
    if constexpr (MASKCHK) {  // only the last tiles can hold keys >= L: the main loop runs without this (wave-uniform) branch
      if (__builtin_expect((2 * j + T + 1) >= mask_from, 0)) {
        mfma_drain();
        mask_half(nxt, (2 * j + T + 1) * 32);
      }
    }
    if constexpr (T == 0) {
      // K_{j+1} and V_j have landed (own pieces; the barrier extends that to every wave), one younger tile pair stays in flight.  Behind
      // the barrier every wave is done with K_j (its last reader is the MFMA group above) and with V_{j-2}: their slots take K_{j+4}
      // (issued below) and V_{j+2} (issued by the odd half-step).  The fragments already read ahead belong to V_{j-1}.
      if constexpr (!(ABL & 4)) {
        wait_vm<2 * LPW4>();
        __builtin_amdgcn_s_barrier();
      }
    }
    // -- C: O^T += V_{h-1}^T P_{h-1}^T: gap g = (key slice c, head-dim block db, block x); V^T fragments three ahead
    {
      constexpr int VPF = 3;
      v8bf vf[VPF + 1];
      if constexpr (!FIRST) {
        vf[0] = vpre[0];
        vf[1] = vpre[1];
        vf[2] = vpre[2];
      }
      fence4();
      static_for4<16>([&](auto GC) {
        constexpr int g = decltype(GC)::value, x = g & 1, s = g >> 1, cs = s >> 2, db = s & 3;
        if constexpr (!FIRST) {
          if constexpr (!(ABL & 32)) mfma_pv(o[x][db], vf[s % (VPF + 1)], pp[x][cs]);
          fence4();
          if constexpr (!(ABL & 8) && x == 0 && s + VPF < 8) vf[(s + VPF) % (VPF + 1)] = v_frag(VS, 2 * VH + ((s + VPF) >> 2), (s + VPF) & 3);
        }
        if constexpr (!(ABL & 8)) {
          if constexpr (g == 11) kpre[0] = k_frag(KSN, 0, KHN);
          if constexpr (g == 13) kpre[1] = k_frag(KSN, 1, KHN);
          if constexpr (g == 15) kpre[2] = k_frag(KSN, 2, KHN);
        }
        if constexpr (!(ABL & 1)) gapwork4<16 + g>(cur, nxt, pc, l2, m0);

        if constexpr (!(ABL & 17) && g == 14) need = __any(fmaxf(m0[0], m0[1]) > (EXACT ? 0.0f : a.defer_log2));
        if constexpr (!(ABL & 2) && T == 0 && (g & 3) == 3) dma_k(PAR, j + 4, g >> 2);
        fence4();
      });
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using TT = std::true_type; using FF = std::false_type;
  auto tile_step = [&](auto PARC, auto MASKC, int j) __attribute__((always_inline)) {
    half_step(PARC, I0{}, FF{}, MASKC, sa, sb, pb, pa, j);
    half_step(PARC, I1{}, FF{}, MASKC, sb, sa, pa, pb, j);
  };
  half_step(I0{}, I0{}, TT{}, TT{}, sa, sb, pb, pa, 0);
  half_step(I0{}, I1{}, FF{}, TT{}, sb, sa, pa, pb, 0);
  int j = 1;
  // tiles up to ntiles - 3 produce score blocks of keys < L only (half-tile 2 j + 2 <= 2 ntiles - 4 < L / 32)
  for (; j + 4 <= ntiles - 2; j += 4) {
    tile_step(I1{}, FF{}, j);
    tile_step(I2{}, FF{}, j + 1);
    tile_step(I3{}, FF{}, j + 2);
    tile_step(I0{}, FF{}, j + 3);
  }
  // at most five tiles are left; a loop around a switch over j % 4 made the allocator spill, so the tail is written out
  if (j < ntiles) { tile_step(I1{}, TT{}, j); ++j; }
  if (j < ntiles) { tile_step(I2{}, TT{}, j); ++j; }
  if (j < ntiles) { tile_step(I3{}, TT{}, j); ++j; }
  if (j < ntiles) { tile_step(I0{}, TT{}, j); ++j; }
  if (j < ntiles) { tile_step(I1{}, TT{}, j); ++j; }

  // ---- drain: O^T += V^T P^T of the last half-tile (key half 1 of tile ntiles - 1, produced into pb) --------------------------------
  wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  mfma_drain();
  {
    const int vs = (ntiles - 1) & 3;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const v8bf vfr = *(lds_frag_p)(size_t)(vx[2 + (s >> 2)] + (unsigned)(vs * V_BYTES + (s & 3) * 4096));
#pragma unroll
      for (int x = 0; x < 2; ++x)
        o[x][s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, __builtin_bit_cast(v8bf, pb[x][s >> 2]), o[x][s & 3], 0, 0, 0);
    }
  }
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const float l_part = l2[x][0] + l2[x][1];
    const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
    store_o<FMT>(a, o[x], 1.0f / l_tot, b, h, q0 + x * 32 + l31, hi);
  }
}

template <bool EXACT, int ABL = 0> int launch4(const AttnArgs& a, int fmt, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)attention4_kernel<FLUXMI_FMT_E5M2, EXACT, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, RD4 * A_STAGE));
    if constexpr (ABL == 0)
      FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)attention4_kernel<FLUXMI_FMT_E4M3, EXACT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, RD4 * A_STAGE));
    attr = true;
  }
  const dim3 grid(((a.L + 255) / 256) * a.H * a.B);
  if (fmt == FLUXMI_FMT_E5M2 || ABL != 0) hipLaunchKernelGGL((attention4_kernel<FLUXMI_FMT_E5M2, EXACT, ABL>), grid, dim3(NW4 * 64), RD4 * A_STAGE, s, a);
  else if constexpr (ABL == 0) hipLaunchKernelGGL((attention4_kernel<FLUXMI_FMT_E4M3, EXACT, 0>), grid, dim3(NW4 * 64), RD4 * A_STAGE, s, a);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// fp16 K only (the folded arithmetic).  FLUXMI_ATTN_VAR=2 (read per call: the tests sweep it) = exact instead of deferred max tracking.
int fluxmi_launch_attention4(const AttnArgs& a, int fmt, hipStream_t s) {
  FLUXMI_REQUIRE(a.k_f16, "attention: the 4-wave kernel takes fp16 K");
#ifdef FLUXMI_ATTN4_ABLATIONS  // make EXTRA=-DFLUXMI_ATTN4_ABLATIONS: the timing-only variants behind tools/attn4_abl.py (results are garbage)
  if (const char* ab = getenv("FLUXMI_ATTN4_ABL")) {
    switch (atoi(ab)) {
      case 1: return launch4<false, 1>(a, fmt, s);
      case 2: return launch4<false, 2>(a, fmt, s);
      case 4: return launch4<false, 4>(a, fmt, s);
      case 8: return launch4<false, 8>(a, fmt, s);
      case 16: return launch4<false, 16>(a, fmt, s);
      case 32: return launch4<false, 32>(a, fmt, s);
      case 27: return launch4<false, 27>(a, fmt, s);
      default: break;
    }
  }
#endif
  const char* e = getenv("FLUXMI_ATTN_VAR");
  return (e && (atoi(e) & 2)) ? launch4<true>(a, fmt, s) : launch4<false>(a, fmt, s);
}
