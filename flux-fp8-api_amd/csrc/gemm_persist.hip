// fluxmi -- PERSISTENT ping-pong GEMM (tile config 18): the 256x256 / 8-wave body of gemm_pp.hip, one workgroup per CU walking a static
// list of tiles, for the multi-round F8Linear launches of the step (single-block linear1: 5.9 rounds of the 256 CUs, double-block
// mlp.0: 3.0, qkv: 2.25).
//
// What the one-tile-per-workgroup kernel pays PER TILE and this one does not (profiles/r03_gemm_noepilogue.txt, r03_streamk.txt):
//  * workgroup launch, descriptor set-up and the cold prologue (three K-steps of LDS-DMA from L2 with the matrix pipes idle);
//  * the drain: the epilogue's stores have to reach L2 before the CU takes the next workgroup.
// Here the LDS ring never drains: the refill slots of a tile's last three K-steps take the FIRST three K-steps of the workgroup's next
// tile, so when the epilogue ends the next tile's operands are already in LDS, and the epilogue's stores retire under the next K loop.
// That needs most of the ring to stay untouched by the epilogue: the bf16 transposition goes through 4 KiB per wave of the ONE slot that
// is dead when the K loop ends (the last step's), one 32-row block of the wave's 128 x 64 tile at a time.  The table-driven
// GELU -> fp8 epilogue (64 KiB table, gemm_epilogue.h) needs two more slots -- a table tile therefore prefetches only two K-steps of its
// successor and issues the third and fourth right after its table reads.
//
// K loop = gemm_pp.hip's (ring of 64-byte K-steps filled by `buffer_load ... lds`, ONE raw s_barrier per step, counted vmcnt, the two
// waves of a SIMD in opposite phase) with three changes.  (1) FIVE slots, filled FOUR steps ahead -- all 160 KiB of LDS (with 2 / 3 / 4
// K-steps in flight a tile's loop takes 85.9 / 62.5 / 62.0 K cycles: beyond three the loop no longer waits for latency).  Ring slots are
// run-time values (48 K-steps per tile do not divide by five).  (2) The refill pieces and group 0's next fragments are interleaved with
// the MFMAs instead of following them as a block.  (3) W may arrive in the ROW-PAIR layout (fluxmi_gemm_group_t.W_pairs): what bounded the
// loop at depth 3+ was L2 LINE traffic -- a 64-byte K-step of a row is half a 128-byte line, the other half is the next K-step's and is
// fetched again (the 32 KiB L1 does not keep it), so the operand panels crossed L2 -> CU twice per tile and the data movement alone took
// longer than the MFMAs (ablation builds PS_ABL below; profiles/r04_gemm_persist.txt sections 9 - 10).  With the K-steps of two rows
// sharing a line every W line is fetched once: 62.4 K -> 56.5 K cycles per tile.
// vmcnt protocol across tiles: every path of the epilogue converts its accumulators (VALU only), then waits vmcnt(0) ONCE -- as the
// builtin, so that hipcc's own wait-count pass knows it: with LDS-DMA pending in its model it puts vmcnt(0) in front of every scratch
// access it cannot prove disjoint, i.e. drains the epilogue's stores block by block -- and only then touches LDS and stores.  The
// successor's K-steps were issued before that point, so steps 0 and 1 of the next K loop open without a wait and the stores retire under
// it; from step 2 on the counted vmcnt(4) is exact again (VMEM operations of a wave retire in order on gfx9).
// Tile order = the XCD-aware order of the one-tile-per-workgroup kernels: workgroup b sits on XCD b % 8 and walks that XCD's contiguous
// range of logical tile ids 32 at a time (bands of four row tiles, column by column inside a band: 4 x 8 tiles share A / W panels through
// the XCD's L2; launch_ps).
#include <type_traits>

#include "gemm_epilogue.h"

// Timing-only ablations of the K loop (wrong results; never defined in the product build -- tools/probes, profiles/r04_gemm_persist.txt §9):
// bit 0: no LDS-DMA refills of the current tile's K-steps, bit 1: no fragment reads after a tile's first, bit 2: no MFMAs,
// bit 3 / bit 4: the A / W refills address their operand as if it were stored in row pairs ([R/2][K/64][2][64]: a K-step of a row pair is ONE
// 128-byte line) -- same byte count, every L2 line fetched once per tile instead of twice (a 64-byte K-step is half a line).
#ifndef PS_ABL
#define PS_ABL 0
#endif

namespace {

// wave-uniform description of one 256 x 256 output tile: descriptors + tile offsets of its A / W panels, group index, tile origin.
// Plain scalars on purpose: as a struct (copied cur <- nxt once per tile) hipcc kept the ints in scratch memory, and every scratch access
// is a VMEM operation that waits vmcnt(1) -- i.e. for the epilogue's stores and the LDS-DMA in flight.
__device__ __forceinline__ void ps_setup(const FluxmiGemmParams& P, int lid, int eb, __amdgpu_buffer_rsrc_t& ars, __amdgpu_buffer_rsrc_t& wrs,
                                         unsigned& a_soff0, unsigned& w_soff0, int& gi_out, int& m0, int& n0) {
  const int tiles_n = P.N >> 8;
  const int width = P.group_m * tiles_n;
  const int first_m = (lid / width) * P.group_m;
  const int gsz = min(P.tiles_m_total - first_m, P.group_m);
  const int tm = first_m + (lid % width) % gsz;
  const int tn = (lid % width) / gsz;
  int gi = 0;
  for (int i = 1; i < P.n_groups; ++i) gi = (tm >= P.g[i].m_tile_start) ? i : gi;
  const FluxmiGemmGroup& G = P.g[gi];
  gi_out = gi;
  m0 = (tm - G.m_tile_start) * 256;
  n0 = tn * 256;
  const long long a_row_b = (long long)G.lda * eb, w_row_b = (long long)P.K * eb;
  ars = make_rsrc(G.A, (unsigned)min((long long)G.M * a_row_b, 0xffffffffLL));
  wrs = make_rsrc(G.W_pairs ? G.W_pairs : G.W, (unsigned)min((long long)P.N * w_row_b, 0xffffffffLL));
  a_soff0 = uni_u32((unsigned)(m0 * a_row_b));
  w_soff0 = uni_u32((unsigned)(n0 * w_row_b));
}

// s_waitcnt vmcnt(0) as the BUILTIN (expcnt / lgkmcnt fields = no wait): unlike an asm statement, hipcc's own wait-count pass sees it and
// knows that nothing older is pending -- with the asm form it kept every epilogue load "in flight" in its model and drained vmcnt (i.e.
// the epilogue's stores) at the first register reuse of the next tile
__device__ __forceinline__ void ps_wait_all_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// 16-byte store through a buffer descriptor: per-lane byte offset in a VGPR and NO guard -- an offset at or past the descriptor's size
// (a row past M, or 0xffffffff for a lane that must not store) is dropped by the hardware.  With the row guard as a branch hipcc put
// every store of the epilogue into its own basic block: ds_read -> lgkmcnt(0) -> 64-bit address multiply -> store, sixteen LDS round
// trips in a row.
// The wave's tile origin goes into the descriptor BASE, not into the instruction's SGPR offset: with a register in the soffset field
// hipcc (ROCm 7.2) assumes that a 128-bit MUBUF store has no "data VGPR overwritten before the store has read it" hazard and emits
//     buffer_store_dwordx4 v[138:141], v152, s[4:7], s17 offen ; v_add_u32 v138, s16, v152
// back to back -- on gfx950 the store then picks up the NEW v138 now and then (measured: 300 - 35000 wrong bytes per launch).  With the
// field hard-wired to 0 the compiler pads the hazard itself.
__device__ __forceinline__ void ps_store16(__amdgpu_buffer_rsrc_t rsrc, v4i data, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voff, 0, 0);
}
// descriptor of the wave's part of an output: `base + origin` .. `base + bytes` (empty if the origin lies past the end)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ps_out_rsrc(void* base, long long bytes, long long origin) {
  const long long left = bytes - origin;
  return make_rsrc((char*)base + origin, (unsigned)(left > 0 ? min(left, 0xffffffffLL) : 0));
}
__device__ __forceinline__ unsigned long long ps_clock() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned long long ps_realtime() { return __builtin_amdgcn_s_memrealtime(); }

// ---- epilogue of one wave's 128 x 64 block through its 4 KiB scratch ------------------------------------------------------------
// EPI: BF16 (plain rows or the fused V^T layout), GATE_RESID, SPLIT (columns < split_n as BF16, the rest through the table),
// GELU_QUANT (table).  `lut` selects the table path (tile-uniform).  Ends with its stores in flight.  `after_table` is called once
// (table path only) when every wave of the workgroup is done reading the table.
// the lane's bias words (4 consecutive columns per 8-column group) of its wave's 64 columns; issued by the kernel four K-steps before the
// end of the K loop, i.e. AHEAD of the last LDS-DMA refills: loads retire in order, so the epilogue's first use of a bias word then
// waits for nothing but the bias itself (issued behind the refills it would wait out their full L2 latency with the matrix pipes idle)
__device__ __forceinline__ void ps_load_bias(const FluxmiGemmGroup& G, int n_wave0, int hi, uint2 (&braw)[2][4]) {
  const fluxmi_gptr<const u16> bias_p = uni_ptr((const u16*)G.bias);
  const fluxmi_gptr<const u16> bias_src = bias_p != nullptr ? bias_p : uni_ptr((const u16*)G.W);  // never a branch around the loads (gemm_epilogue.h)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      braw[j][g4] = __builtin_bit_cast(uint2, *(fluxmi_gptr<const fluxmi_v2i>)(bias_src + n_wave0 + j * 32 + g4 * 8 + hi * 4));
}

// KIND (compile time, tile-uniform): 0 = plain tile, 1 = table tile, 2 = fused-K tile
template <int EPI, int FMT, int KIND, class AfterTable>
__device__ __forceinline__ void ps_epilogue(const FluxmiGemmGroup& G, v16f (&acc)[4][2], uint2 (&braw)[2][4], float s, unsigned char* wbuf,
                                            unsigned char* table, int m_wave0, int n_wave0, int M, int lane, int wave,
                                            AfterTable after_table, unsigned long long* stamps = nullptr) {
  // timing build only: phase stamps of the table path (stamps[0..3]) by the workgroup's first lane
  auto stamp = [&](int k) { if (stamps) stamps[k] = ps_clock(); };
  constexpr int TM = 4, TN = 2;
  const int l31 = lane & 31, hi = lane >> 5;
  const bool has_bias = uni_ptr((const u16*)G.bias) != nullptr;
  auto pin_bias = [&]() {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        asm volatile("" : "+v"(braw[j][g4].x), "+v"(braw[j][g4].y));
        if (!has_bias) braw[j][g4] = make_uint2(0, 0);
      }
  };
  auto bias_of = [&](int j, int g4, float* b) {
    const uint2 v = braw[j][g4];
    b[0] = __uint_as_float(v.x << 16); b[1] = __uint_as_float(v.x & 0xffff0000u);
    b[2] = __uint_as_float(v.y << 16); b[3] = __uint_as_float(v.y & 0xffff0000u);
  };

  // phase 0, common to every path and VALU only: h = bf16(acc * s + bias), two per register, block-major (an accumulator block is dead
  // once its 16 words are packed).  It runs BEFORE the one vmcnt(0) of the epilogue, i.e. under the L2 latency of the successor's last
  // K-step (issued in the last K-step of this tile) and of the first residual rows.
  unsigned hp[TM][TN][4][2];
  auto convert_all = [&]() {
    pin_bias();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float bias[4];
          bias_of(j, g4, bias);
          hp[i][j][g4][0] = pack_bf2(fmaf(acc[i][j][g4 * 4 + 0], s, bias[0]), fmaf(acc[i][j][g4 * 4 + 1], s, bias[1]));
          hp[i][j][g4][1] = pack_bf2(fmaf(acc[i][j][g4 * 4 + 2], s, bias[2]), fmaf(acc[i][j][g4 * 4 + 3], s, bias[3]));
        }
  };

  if constexpr (EPI == FLUXMI_EPI_GELU_QUANT || EPI == FLUXMI_EPI_SPLIT) {
    if constexpr (KIND == 1) {
      // ---- quantising path: fp8 = quantise(gelu(bf16(acc * s + bias))) -------------------------------------------------------------
      // The 64 KiB table (gemm_epilogue.h) sits in two adjacent dead ring slots; its LDS-DMA was issued inside the last K-step (kernel:
      // behind the barrier in the middle of that step, when the slots were dead), so it lands under the step's last MFMAs and the conversion.
      // The byte gather runs at the speed of its LDS bank conflicts (64 random addresses per ds_read_u8: ~10 K cycles per tile for the
      // 1024 gather instructions of the eight waves); quantising half of the blocks on the VALU instead, the two waves of a SIMD in
      // opposite order, was measured SLOWER (the exact GELU chain costs ~14 VALU instructions per element, two of them
      // transcendental: profiles/r04_gemm_persist.txt) and is not in the tree.
      convert_all();
      stamp(1);
      ps_wait_all_vmem();  // the table (and the next tile's two K-steps) landed
      __builtin_amdgcn_s_barrier();
      stamp(2);
      auto gather_block = [&](int i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const unsigned a = hp[i][j][g4][0], b = hp[i][j][g4][1];
            const unsigned q0 = table[a & 0xffffu], q1 = table[a >> 16], q2 = table[b & 0xffffu], q3 = table[b >> 16];
            hp[i][j][g4][0] = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);  // in place: the packed pair is dead
          }
      };
#pragma unroll
      for (int i = 0; i < TM; ++i) gather_block(i);
      // 32 rows x 64 B of fp8 per block through the wave's scratch: lane owns 4 consecutive columns of row l31 -> one dword, 16-B chunks
      // XOR-swizzled by row; read back as (row, 16-B chunk) per lane.  ALL of it happens before the successor's third K-step is issued:
      // hipcc orders every LDS read behind a pending LDS-DMA it cannot prove disjoint (vmcnt(0) in front of each read, and with it a full
      // drain of the stores issued so far)
      uint4 raw[TM][2];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int nl = j * 32 + g4 * 8 + hi * 4;
            const int chunk = (nl >> 4) ^ ((l31 >> 1) & 3);
            *(unsigned*)(wbuf + l31 * 64 + chunk * 16 + (nl & 12)) = hp[i][j][g4][0];
          }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int ml = it * 16 + (lane >> 2), c = lane & 3;
          raw[i][it] = *(const uint4*)(wbuf + ml * 64 + ((c ^ ((ml >> 1) & 3)) * 16));
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int it = 0; it < 2; ++it) asm volatile("" : "+v"(raw[i][it].x), "+v"(raw[i][it].y), "+v"(raw[i][it].z), "+v"(raw[i][it].w));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(3);
      __builtin_amdgcn_s_barrier();  // every wave is done with the table: its slots go back to the ring
      after_table();
      const unsigned ld8 = EPI == FLUXMI_EPI_SPLIT ? uni_u32((unsigned)G.ldc2) : uni_u32((unsigned)G.ldc);
      const int col0 = EPI == FLUXMI_EPI_SPLIT ? (int)uni_u32((unsigned)(G.c2_col0 - G.split_n)) : 0;
      // c8_pairs: the fp8 rows leave in the row-pair layout the next F8Linear reads with a_pairs (fluxmi_gemm_group_t; the wave's 64 columns
      // are ONE 64-byte chunk of every row, m_wave0 is even): chunk (row r, col) -> (r / 2) * 2 * ld + (r % 2) * 64 + (col / 64) * 128
      const bool c8p = uni_u32((unsigned)G.c8_pairs) != 0;
      const long long org = c8p ? (long long)m_wave0 * ld8 + (long long)((col0 + n_wave0) >> 6) * 128 : (long long)m_wave0 * ld8 + (col0 + n_wave0);
      const __amdgpu_buffer_rsrc_t c8 = ps_out_rsrc(EPI == FLUXMI_EPI_SPLIT ? G.C2 : G.C, (long long)M * ld8, org);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int ml = it * 16 + (lane >> 2), c = lane & 3, r = i * 32 + ml;
          ps_store16(c8, __builtin_bit_cast(v4i, raw[i][it]), c8p ? (unsigned)(r >> 1) * 2 * ld8 + (r & 1) * 64 + c * 16 : (unsigned)r * ld8 + c * 16);
        }
      return;
    }
  }

  const unsigned ldc_b = uni_u32((unsigned)G.ldc) * 2;  // row stride in bytes
  const __amdgpu_buffer_rsrc_t c_rs = ps_out_rsrc(G.C, (long long)M * ldc_b, (long long)m_wave0 * ldc_b + (long long)n_wave0 * 2);
  if constexpr (EPI == FLUXMI_EPI_BF16 || EPI == FLUXMI_EPI_SPLIT) {
    // ---- fused V^T (wave-uniform): [32 keys][64 d] of one head's V per block -> LDS as [64 d][32 keys] (keys in the PV MFMA's k-slot
    // order: bits 2 and 3 swapped inside a 16-key group), leaves as 64-byte runs of one d-row of vt_out
    const int vcol0 = G.kv_col0 + G.heads * 128;
    if (KIND == 0 && G.vt_out && n_wave0 >= vcol0 && n_wave0 < vcol0 + G.heads * 128) {
      const unsigned vt_ld_b = uni_u32((unsigned)G.vt_ld) * 2;
      const int d0 = n_wave0 - vcol0, tok0 = (int)uni_u32((unsigned)G.tok0), vt_rows = (int)uni_u32((unsigned)G.vt_rows);
      const __amdgpu_buffer_rsrc_t vt_rs = ps_out_rsrc(G.vt_out, (long long)(G.heads * 128) * vt_ld_b, (long long)d0 * vt_ld_b + (long long)(tok0 + m_wave0) * 2);
      const int posl = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
      convert_all();
      ps_wait_all_vmem();  // the ONE vmcnt(0) of this path: from here on hipcc's wait-count model holds no LDS-DMA a scratch access could alias
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const bool live = m_wave0 + i * 32 + l31 < M;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int dl = j * 32 + g4 * 8 + hi * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned w = hp[i][j][g4][e >> 1];
              const u16 v = live ? (u16)((e & 1) ? (w >> 16) : (w & 0xffffu)) : (u16)0;
              *(u16*)(wbuf + (dl + e) * 64 + posl * 2) = v;
            }
          }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int dl = it * 16 + (lane >> 2), c = lane & 3;
          const uint4 raw = *(const uint4*)(wbuf + dl * 64 + c * 16);
          const int key = m_wave0 + i * 32 + c * 8;  // group-relative position of the 8 keys
          ps_store16(vt_rs, __builtin_bit_cast(v4i, raw), key < vt_rows ? (unsigned)dl * vt_ld_b + (unsigned)(i * 32 + c * 8) * 2 : 0xffffffffu);
        }
      }
      return;
    }
    // ---- fused K (TILE-uniform: the q | k | v column blocks are multiples of 256, so all eight waves are here together and may meet at
    // barriers): QKNorm = fp32 rms over the head's 128 columns with the learnable scale, RoPE in bf16 arithmetic, head-major store --
    // what qkv_rope_kernel does to the K columns, with the same operations in the same order (bit-identical K), but on the accumulators:
    // the K columns never go to the qkv buffer and back (28 MB written + 28 MB read per launch at L = 4608) and the launch disappears.
    //   sums of squares: in the accumulator layout, where a row's 64 columns of this wave sit in lanes l31 and l31 + 32 as eight groups of
    //     8 columns (4 + 4): per group the sequential chain of qkv_rope's lane, continued across the two lanes by v_permlane32_swap; the two
    //     waves of a head (wave ^ 1: same rows, the other 64 columns) swap their 8 group sums per row through their 4 KiB scratches, and the
    //     16 sums are added in the order of qkv_rope's 16-lane xor tree (8, 4, 2, 1);
    //   normalisation: still in the accumulator layout (the row's 1 / rms lives in the lane that owns the row);
    //   RoPE + store: after the transposition, a lane holds 8 consecutive columns of a row = 4 (cos, sin) pairs = 16 B of `pe`, loaded one
    //     32-row block ahead like the residual rows of the gate path.                                       flux_model.py:158-176,60-65
    const int kcol0 = (int)uni_u32((unsigned)G.kv_col0);
    if constexpr (KIND == 2) {
      const int hcol = n_wave0 - kcol0, head = hcol >> 7, d0 = hcol & 127;
      const int tok0 = (int)uni_u32((unsigned)G.tok0), k_rows = (int)uni_u32((unsigned)G.k_rows);
      const bool f16 = uni_u32((unsigned)G.k_f16) != 0;
      const int c = lane & 7, rl = lane >> 3;
      // scale vector, pe rows and K rows through buffer descriptors (pe / K: the range ends with the group's row M - 1): one 32-bit offset
      // per lane plus constants, no clamps, no guards (a pe row past M reads 0, a K row past M is dropped) -- with 64-bit addresses and
      // row guards hipcc computed all 32 of them up front and spilled them
      const __amdgpu_buffer_rsrc_t kn_rs = make_rsrc((const char*)G.k_norm + d0 * 2, (unsigned)(128 - d0) * 2);
      const __amdgpu_buffer_rsrc_t pe_rs = ps_out_rsrc((void*)G.pe, (long long)(tok0 + M) * 256, (long long)(tok0 + m_wave0) * 256 + d0 * 2);
      const __amdgpu_buffer_rsrc_t k_rs = ps_out_rsrc((char*)G.k_out + ((long long)head * k_rows + tok0) * 256, (long long)M * 256,
                                                      (long long)m_wave0 * 256 + d0 * 2);
      const unsigned row_voff = (unsigned)rl * 256 + c * 16;
      uint4 per[4];
      auto load_pe = [&](int i, int it) {
        per[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(pe_rs, row_voff + (unsigned)(i * 32 + it * 8) * 256, 0, 0));
      };
      auto unpack4 = [&](int i, int j, int g4, float* x) {
        const unsigned a = hp[i][j][g4][0], b = hp[i][j][g4][1];
        x[0] = __uint_as_float(a << 16); x[1] = __uint_as_float(a & 0xffff0000u);
        x[2] = __uint_as_float(b << 16); x[3] = __uint_as_float(b & 0xffff0000u);
      };
      // every phase below is fenced (sched_barrier) and re-unpacks its inputs from an opaque copy: left alone, hipcc interleaves the
      // phases, keeps the 256 unpacked floats of one pass alive for the next and spills ~300 registers -- into the K loop's prologue too
      auto opaque_block = [&](int i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) asm volatile("" : "+v"(hp[i][j][g4][0]), "+v"(hp[i][j][g4][1]));
        __builtin_amdgcn_sched_barrier(0);
      };
      convert_all();
#pragma unroll
      for (int i = 0; i < TM; ++i) opaque_block(i);  // pins the whole conversion HERE: hipcc sinks the later blocks' into the code below otherwise
      ps_wait_all_vmem();  // the ONE vmcnt(0) of this path (see the V^T path)
      // issued after the conversion (128 accumulators + 64 packed words + bias leave no room); the L2 latency passes under the sums
      uint2 wraw[TN][4];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          wraw[j][g4] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(kn_rs, (unsigned)(j * 32 + g4 * 8 + hi * 4) * 2, 0, 0));
#pragma unroll
      for (int it = 0; it < 4; ++it) load_pe(0, it);
      __builtin_amdgcn_sched_barrier(0);
      float* mine = (float*)wbuf;
      const float* other = (const float*)(wbuf + ((wave ^ 1) - wave) * 4096);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float gsum[8];
        // pass 1: the lane's 4 columns of each group, left to right; pass 2: lanes 32..63 continue the chain of lanes 0..31
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            float x[4];
            unpack4(i, j, g4, x);
            float p = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) p += x[e] * x[e];
            gsum[j * 4 + g4] = p;
          }
        opaque_block(i);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            float x[4];
            unpack4(i, j, g4, x);
            const unsigned u = __float_as_uint(gsum[j * 4 + g4]);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // [0]: lanes 32..63 receive lanes 0..31
            float q = hi ? __uint_as_float(sw[0]) : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) q += x[e] * x[e];
            // lanes 32..63 now hold the group's 8 columns summed left to right; lanes 0..31 take a copy, so that the store below needs no
            // branch (both lanes of a row write the same words: a basic-block boundary here lets hipcc sink code across the fences)
            const unsigned uq = __float_as_uint(q);
            const auto sq = __builtin_amdgcn_permlane32_swap(uq, uq, false, false);  // [1]: lanes 0..31 receive lanes 32..63
            gsum[j * 4 + g4] = hi ? q : __uint_as_float(sq[1]);
          }
        *(float4*)(mine + (i * 32 + l31) * 8) = make_float4(gsum[0], gsum[1], gsum[2], gsum[3]);
        *(float4*)(mine + (i * 32 + l31) * 8 + 4) = make_float4(gsum[4], gsum[5], gsum[6], gsum[7]);
        opaque_block(i);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      float rinv[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float4 a0 = *(const float4*)(mine + (i * 32 + l31) * 8), a1 = *(const float4*)(mine + (i * 32 + l31) * 8 + 4);
        const float4 b0 = *(const float4*)(other + (i * 32 + l31) * 8), b1 = *(const float4*)(other + (i * 32 + l31) * 8 + 4);
        const float t0 = a0.x + b0.x, t1 = a0.y + b0.y, t2 = a0.z + b0.z, t3 = a0.w + b0.w;
        const float t4 = a1.x + b1.x, t5 = a1.y + b1.y, t6 = a1.z + b1.z, t7 = a1.w + b1.w;
        const float ss = ((t0 + t4) + (t2 + t6)) + ((t1 + t5) + (t3 + t7));  // the order of qkv_rope's xor tree: 8, 4, 2, 1
        rinv[i] = 1.0f / sqrtf(ss * (1.0f / 128.0f) + 1e-6f);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float w[4];
          const uint2 wv = wraw[j][g4];
          w[0] = __uint_as_float(wv.x << 16); w[1] = __uint_as_float(wv.x & 0xffff0000u);
          w[2] = __uint_as_float(wv.y << 16); w[3] = __uint_as_float(wv.y & 0xffff0000u);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            float x[4];
            unpack4(i, j, g4, x);
            hp[i][j][g4][0] = pack_bf2((x[0] * rinv[i]) * w[0], (x[1] * rinv[i]) * w[1]);
            hp[i][j][g4][1] = pack_bf2((x[2] * rinv[i]) * w[2], (x[3] * rinv[i]) * w[3]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // the partner has read this wave's sums: the scratch is free for the transposition
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int nl = j * 32 + g4 * 8 + hi * 4;
            uint2 v;
            v.x = hp[i][j][g4][0];
            v.y = hp[i][j][g4][1];
            const int chunk = (nl >> 3) ^ (l31 & 7);
            *(uint2*)(wbuf + l31 * 128 + chunk * 16 + (nl & 4) * 2) = v;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const unsigned f16m = f16 ? 0xffffffffu : 0u;  // blended, not branched: a basic-block boundary lets hipcc move code across the fences
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int ml = it * 8 + rl;
          const uint4 raw = *(const uint4*)(wbuf + ml * 128 + ((c ^ (ml & 7)) * 16));
          float x[8], cs[8], y[8];
          unpack8(raw, x);
          unpack8(per[it], cs);
          if (i + 1 < TM) load_pe(i + 1, it);  // the row's pe registers are free again: the same row of the next block, a block ahead
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float cc = cs[2 * p], sn = cs[2 * p + 1];
            y[2 * p] = rbf(rbf(cc * x[2 * p]) + rbf((-sn) * x[2 * p + 1]));
            y[2 * p + 1] = rbf(rbf(sn * x[2 * p]) + rbf(cc * x[2 * p + 1]));
          }
          const uint4 ph = pack8_f16(y), pb = pack8(y);
          uint4 pk;
          pk.x = (ph.x & f16m) | (pb.x & ~f16m); pk.y = (ph.y & f16m) | (pb.y & ~f16m);
          pk.z = (ph.z & f16m) | (pb.z & ~f16m); pk.w = (ph.w & f16m) | (pb.w & ~f16m);
          ps_store16(k_rs, __builtin_bit_cast(v4i, pk), row_voff + (unsigned)(i * 32 + it * 8) * 256);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      return;
    }
  }

  // ---- bf16 rows: plain store, or x + bf16(gate * h) with the residual rows one 32-row block ahead ------------------------------
  const fluxmi_gptr<const u16> resid_p = uni_ptr((const u16*)G.resid);
  const fluxmi_gptr<const u16> gate_p = uni_ptr((const u16*)G.gate);
  const long long ldr_u = uni_i64(G.ldr);
  constexpr bool GR = EPI == FLUXMI_EPI_GATE_RESID;
  const int c = lane & 7, rl = lane >> 3;  // phase 2: lane -> (row rl + 8 * it of the block, 16-B chunk c)
  uint4 rres[2][GR ? 4 : 1];
  uint4 graw = make_uint4(0, 0, 0, 0);
  auto load_resid = [&](int i, uint4 (&dst)[GR ? 4 : 1]) {
    if constexpr (GR) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = min(m_wave0 + i * 32 + it * 8 + rl, M - 1);  // clamped, not guarded: the load is unconditional, the store is not
        dst[it] = __builtin_bit_cast(uint4, *(fluxmi_gptr<const v4i>)(resid_p + (long long)m * ldr_u + n_wave0 + c * 8));
      }
    }
  };
  if constexpr (GR) {
    graw = __builtin_bit_cast(uint4, *(fluxmi_gptr<const v4i>)(gate_p + n_wave0 + c * 8));
    load_resid(0, rres[0]);
  }
  convert_all();
  ps_wait_all_vmem();  // the ONE vmcnt(0) of this path (see the V^T path); the first residual rows and the gate have landed as well
  float g[8];
  if constexpr (GR) unpack8(graw, g);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int nl = j * 32 + g4 * 8 + hi * 4;
        uint2 v;
        v.x = hp[i][j][g4][0];
        v.y = hp[i][j][g4][1];
        const int chunk = (nl >> 3) ^ (l31 & 7);
        *(uint2*)(wbuf + l31 * 128 + chunk * 16 + (nl & 4) * 2) = v;
      }
    if constexpr (GR) {
      if (i + 1 < TM) load_resid(i + 1, rres[(i + 1) & 1]);
#pragma unroll
      for (int it = 0; it < 4; ++it)
        asm volatile("" : "+v"(rres[i & 1][it].x), "+v"(rres[i & 1][it].y), "+v"(rres[i & 1][it].z), "+v"(rres[i & 1][it].w));
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ml = it * 8 + rl;
      const uint4 raw = *(const uint4*)(wbuf + ml * 128 + ((c ^ (ml & 7)) * 16));
      if constexpr (GR) {
        float h[8], r[8], o[8];
        unpack8(raw, h);
        unpack8(rres[i & 1][it], r);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = r[e] + rbf(g[e] * h[e]);
        ps_store16(c_rs, __builtin_bit_cast(v4i, pack8(o)), (unsigned)(i * 32 + ml) * ldc_b + c * 16);
      } else {
        ps_store16(c_rs, __builtin_bit_cast(v4i, raw), (unsigned)(i * 32 + ml) * ldc_b + c * 16);
      }
    }
  }
}

// compile-time switches of a K-step
struct PsNone {}; struct PsCur {}; struct PsNxt {};   // what the step's refill takes: nothing / a K-step of this tile / of the next tile

template <bool FP8, int ACT_FMT, int ESEL, bool TIMING>
__global__ void __launch_bounds__(512, 2) gemm_ps_kernel(const FluxmiGemmParams P) {
  constexpr int NT = 512, TM = 4, TN = 2, STAGE = 32768, A_BYTES = 16384, LPT = 4, NS = 5, D = 4;
  constexpr int EB = FP8 ? 1 : 2;
  constexpr bool MAY_LUT = ESEL == FLUXMI_EPI_GELU_QUANT || ESEL == FLUXMI_EPI_SPLIT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using T = std::true_type;
  using F = std::false_type;

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- static tile list of this workgroup ---------------------------------------------------------------------------------------
  const int nblk = P.tiles_m_total * (P.N >> 8);
  const int xcd = blockIdx.x & 7, wg_in_x = blockIdx.x >> 3;
  const int xq = nblk >> 3, xr = nblk & 7;
  const int base = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int cnt = xq + (xcd < xr ? 1 : 0);
  const int stride = ((int)gridDim.x - xcd + 7) >> 3;  // workgroups of this launch on this XCD
  if (wg_in_x >= cnt) return;
  const int nk = (P.K * EB) / 64;
  const unsigned a_row_b = (unsigned)(P.g[0].lda * EB), w_row_b = (unsigned)(P.K * EB);  // one lda for every group (host check)
  const bool w_pairs = (PS_ABL & 16) != 0 || uni_ptr((const u16*)P.g[0].W_pairs) != nullptr;  // every group or none (host check)
  const unsigned w_kstep = w_pairs ? 128u : 64u;                                               // bytes between consecutive K-steps of a W row
  // A in the row-pair layout (fluxmi_gemm_group_t.a_pairs: the engine's fp8 activation buffers in fused mode; every group or none, host check)
  const bool a_pairs = (PS_ABL & 8) != 0 || uni_u32((unsigned)P.g[0].a_pairs) != 0;
  const unsigned a_kstep = a_pairs ? 128u : 64u;

  // ring slot arithmetic (slots 0 .. NS-1; d <= NS)
  auto nslot = [](int sl, int d) { const int t = sl + d; return t >= NS ? t - NS : t; };
  // per-lane LDS-DMA source offsets of the lane's two A and two W pieces of a K-step (XOR swizzle on the source side)
  auto lane_voffs = [&](int tid, unsigned (&a_voff)[2], unsigned (&w_voff)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = tid + NT * i, row = p >> 2, slot = (p & 3) ^ ((row >> 2) & 3);
      a_voff[i] = a_pairs ? (unsigned)(row >> 1) * 2 * a_row_b + (row & 1) * 64 + slot * 16 : (unsigned)row * a_row_b + slot * 16;
      // W in the row-pair layout (W_pairs: the K-steps of rows 2r, 2r + 1 share a 128-byte line, consecutive K-steps of a pair are 128 bytes apart)
      w_voff[i] = w_pairs ? (unsigned)(row >> 1) * 2 * w_row_b + (row & 1) * 64 + slot * 16 : (unsigned)row * w_row_b + slot * 16;
    }
  };
  auto fence = []() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  __amdgpu_buffer_rsrc_t c_ars, c_wrs, n_ars, n_wrs;
  unsigned c_asoff, c_wsoff, n_asoff, n_wsoff;
  int c_gi, c_m0, c_n0, n_gi, n_m0, n_n0;
  ps_setup(P, base + wg_in_x, EB, c_ars, c_wrs, c_asoff, c_wsoff, c_gi, c_m0, c_n0);
  {
    unsigned a_voff[2], w_voff[2];
    lane_voffs(threadIdx.x, a_voff, w_voff);
#pragma unroll
    for (int st = 0; st < D; ++st) {
      unsigned char* dA = smem + st * STAGE + wave * 1024;
#pragma unroll
      for (int i = 0; i < 2; ++i) dma16_buf(c_ars, dA + NT * 16 * i, a_voff[i], c_asoff + st * a_kstep);
#pragma unroll
      for (int i = 0; i < 2; ++i) dma16_buf(c_wrs, dA + A_BYTES + NT * 16 * i, w_voff[i], c_wsoff + st * w_kstep);
    }
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  int slot0 = 0;  // ring slot of the current tile's K-step 0
  for (int jt = 0;; ++jt) {
    unsigned long long t_start = 0, t_kend = 0;
    if constexpr (TIMING) t_start = ps_clock();
    const int next_idx = wg_in_x + (jt + 1) * stride;
    const bool has_next = next_idx < cnt;
    // the last tile "prefetches" its own first K-steps again: valid, L2-hot addresses, and the vmcnt bookkeeping stays uniform
    if (has_next) {
      ps_setup(P, base + next_idx, EB, n_ars, n_wrs, n_asoff, n_wsoff, n_gi, n_m0, n_n0);
    } else {
      n_ars = c_ars; n_wrs = c_wrs; n_asoff = c_asoff; n_wsoff = c_wsoff; n_gi = c_gi; n_m0 = c_m0; n_n0 = c_n0;
    }
    const FluxmiGemmGroup& G = P.g[c_gi];
    // slots of the tile's last three K-steps: dead when the K loop ends.  A plain tile keeps its successor's first FOUR K-steps in the
    // other four slots and transposes through the slot of step nk - 1; a table tile keeps TWO, puts the 64 KiB table into two adjacent
    // dead slots and transposes through the third.
    const int s_m3 = nslot(slot0, (nk - 3) % NS), s_m1 = nslot(s_m3, 2);
    const int tbl_slot = s_m3 == NS - 1 ? 0 : s_m3, lut_scratch = s_m3 == NS - 1 ? NS - 1 : s_m1;

    // K loop + epilogue of one tile, compiled ONCE PER KIND of tile (table tile or not): as a run-time flag inside one body the table
    // tile's extra work in the last K-step made hipcc spill ~230 VGPRs -- with reloads inside the K loop, each a VMEM operation that
    // drains vmcnt (linear1: 268 -> 414 us)
    auto tile_body = [&](auto KIND_C) {
    constexpr int tile_kind = decltype(KIND_C)::value;
    constexpr bool lut_tile = tile_kind == 1;

    // everything derived from the lane index is recomputed per tile from an opaque copy: values that stay live across the epilogue (the
    // point of highest register pressure) are what hipcc spills, and a reload inside the K loop is a VMEM operation that drains vmcnt
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    unsigned a_voff[2], w_voff[2];
    lane_voffs(tid, a_voff, w_voff);
    int a_lo, a_hi, w_lo, w_hi;
    {
      const int ra = wm * 128 + l31, ka = (ra >> 2) & 3;
      a_lo = ra * 64 + (((hi * 2) ^ ka) << 4);
      a_hi = ra * 64 + (((hi * 2 + 1) ^ ka) << 4);
      const int rw = wn * 64 + l31, kw = (rw >> 2) & 3;
      w_lo = A_BYTES + rw * 64 + (((hi * 2) ^ kw) << 4);
      w_hi = A_BYTES + rw * 64 + (((hi * 2 + 1) ^ kw) << 4);
    }
    // piece q of a K-step's refill into ring slot sl: 0, 1 = the lane's two A pieces, 2, 3 = its two W pieces
    auto dma_piece = [&](__amdgpu_buffer_rsrc_t ars, __amdgpu_buffer_rsrc_t wrs, unsigned asoff, unsigned wsoff, int kt, int sl, int q) {
      unsigned char* dA = smem + sl * STAGE + wave * 1024;
      if (q < 2) dma16_buf(ars, dA + NT * 16 * q, a_voff[q], asoff + kt * a_kstep);
      else dma16_buf(wrs, dA + A_BYTES + NT * 16 * (q - 2), w_voff[q - 2], wsoff + kt * w_kstep);
    };
    auto dma_stage = [&](__amdgpu_buffer_rsrc_t ars, __amdgpu_buffer_rsrc_t wrs, unsigned asoff, unsigned wsoff, int kt, int sl) {
#pragma unroll
      for (int q = 0; q < 4; ++q) dma_piece(ars, wrs, asoff, wsoff, kt, sl, q);
    };
    v8i fa[TM], fw[TN];
    auto read_fa = [&](int sl, int i) {
      const unsigned char* sb = smem + sl * STAGE;
      const v4i lo = *(const v4i*)(sb + i * 2048 + a_lo), h4 = *(const v4i*)(sb + i * 2048 + a_hi);
      fa[i] = (v8i){lo[0], lo[1], lo[2], lo[3], h4[0], h4[1], h4[2], h4[3]};
    };
    auto read_fw = [&](int sl, int j) {
      const unsigned char* sb = smem + sl * STAGE;
      const v4i lo = *(const v4i*)(sb + j * 2048 + w_lo), h4 = *(const v4i*)(sb + j * 2048 + w_hi);
      fw[j] = (v8i){lo[0], lo[1], lo[2], lo[3], h4[0], h4[1], h4[2], h4[3]};
    };
    auto read_frags = [&](int sl) {
#pragma unroll
      for (int j = 0; j < TN; ++j) read_fw(sl, j);
#pragma unroll
      for (int i = 0; i < TM; ++i) read_fa(sl, i);
    };
    v16f acc[TM][TN];
    auto mma_row = [&](auto ZERO, int i) {  // the two MFMAs that take activation fragment i
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        v16f c0;
        if constexpr (decltype(ZERO)::value) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c0[r] = 0.f;
        } else {
          c0 = acc[i][j];
        }
        if constexpr ((PS_ABL & 4) != 0) {
          acc[i][j] = c0;
          acc[i][j][0] += __int_as_float(fw[j][0] ^ fa[i][0]);  // keeps the fragments alive
        } else if constexpr (FP8) {
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[j], fa[i], c0, FLUXMI_FMT_E4M3, ACT_FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
          const v4i alo = (v4i){fa[i][0], fa[i][1], fa[i][2], fa[i][3]}, ahi = (v4i){fa[i][4], fa[i][5], fa[i][6], fa[i][7]};
          const v4i wlo = (v4i){fw[j][0], fw[j][1], fw[j][2], fw[j][3]}, whi = (v4i){fw[j][4], fw[j][5], fw[j][6], fw[j][7]};
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, wlo), __builtin_bit_cast(v8bf, alo), c0, 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, whi), __builtin_bit_cast(v8bf, ahi), c0, 0, 0, 0);
        }
      }
    };
    // table tiles: the 64 pieces of the table into the two dead slots (through a descriptor: one VGPR of per-lane offset instead of eight
    // 64-bit addresses at the point of highest register pressure)
    auto table_dma = [&]() {
      const __amdgpu_buffer_rsrc_t trs = make_rsrc(G.q_lut, 65536u);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int piece = wave * 8 + q;
        dma16_buf(trs, smem + tbl_slot * STAGE + piece * 1024, (unsigned)lane * 16, (unsigned)piece * 1024);
      }
    };
    auto refill_piece = [&](auto SRC, int kt_src, int sl, int q) {
      if constexpr (std::is_same<decltype(SRC), PsNxt>::value) dma_piece(n_ars, n_wrs, n_asoff, n_wsoff, kt_src, sl, q);
      if constexpr (std::is_same<decltype(SRC), PsCur>::value && !(PS_ABL & 1)) dma_piece(c_ars, c_wrs, c_asoff, c_wsoff, kt_src, sl, q);
    };
    // One K-step in ring slot sl.  Behind the two MFMAs of activation fragment p: one piece of the refill of slot sl + 4 (K-step kt_src
    // of the current / the next tile) and, group 0, the reads of fragment p of the NEXT step (its registers are free: both MFMAs that
    // take it are issued) -- an LDS-DMA piece issues in ~60 cycles in the shadow of an MFMA and in 100 - 185 behind the block.
    // WAIT: counted wait for this wave's pieces of the NEXT step (two younger steps may stay in flight).  LAST (last step of a tile):
    // group 0 skips the next fragments; both groups meet at one more barrier once every fragment of the step is in registers, after
    // which the slots of the tile's last three steps are dead (epilogue scratch, table).
    // Group 0: barrier | 8 MFMA + refill + next fragments;  group 1: barrier | fragments | 8 MFMA + refill.
    auto step_g0 = [&](auto ZERO, auto WAIT, auto SRC, auto LAST, int kt_src, int sl) {
      if constexpr (decltype(WAIT)::value) wait_vmcnt<LPT * (D - 2)>();
      __builtin_amdgcn_s_barrier();
      fence();
      const int rs = nslot(sl, D), ns = nslot(sl, 1);
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        mma_row(ZERO, pp);
        fence();
        refill_piece(SRC, kt_src, rs, pp);
        if constexpr (!decltype(LAST)::value && !(PS_ABL & 2)) read_fa(ns, pp);
        fence();
      }
      if constexpr (!decltype(LAST)::value) {
        if constexpr (!(PS_ABL & 2)) {
          read_fw(ns, 0);
          read_fw(ns, 1);
        }
      } else {
        __builtin_amdgcn_s_barrier();
        if constexpr (lut_tile) table_dma();
      }
      fence();
    };
    auto step_g1 = [&](auto ZERO, auto WAIT, auto SRC, auto LAST, int kt_src, int sl) {
      if constexpr (decltype(WAIT)::value) wait_vmcnt<LPT * (D - 2)>();
      __builtin_amdgcn_s_barrier();
      fence();
      if constexpr (!(PS_ABL & 2) || decltype(ZERO)::value) read_frags(sl);
      fence();
      if constexpr (decltype(LAST)::value) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragments of the last K-step are in registers
        __builtin_amdgcn_s_barrier();
      }
      const int rs = nslot(sl, D);
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        mma_row(ZERO, pp);
        fence();
        refill_piece(SRC, kt_src, rs, pp);
        fence();
      }
      if constexpr (decltype(LAST)::value && lut_tile) table_dma();
    };
    // steps 0 .. 2 of a tile: their operands landed before the epilogue's vmcnt(0) (the first tile: before the prologue's), and all that
    // is in flight are the epilogue's stores -- no wait, they retire under the K loop.  A table tile issues its successor's third and
    // fourth K-step AFTER that vmcnt(0): where a predecessor may have been one, steps 1 and 2 keep the counted wait.
    using WE = std::integral_constant<bool, MAY_LUT>;
    using SRC23 = typename std::conditional<lut_tile, PsNone, PsNxt>::type;
    uint2 braw[2][4];
    int sl = slot0;
    auto run = [&](auto step) {
      step(T{}, F{}, PsCur{}, F{}, D, sl); sl = nslot(sl, 1);
      step(F{}, WE{}, PsCur{}, F{}, D + 1, sl); sl = nslot(sl, 1);
      step(F{}, WE{}, PsCur{}, F{}, D + 2, sl); sl = nslot(sl, 1);
      for (int kt = 3; kt < nk - 4; ++kt) {
        step(F{}, T{}, PsCur{}, F{}, kt + D, sl);
        sl = nslot(sl, 1);
      }
      ps_load_bias(G, c_n0 + wn * 64, hi, braw);
      step(F{}, T{}, PsNxt{}, F{}, 0, sl); sl = nslot(sl, 1);
      step(F{}, T{}, PsNxt{}, F{}, 1, sl); sl = nslot(sl, 1);
      step(F{}, T{}, SRC23{}, F{}, 2, sl); sl = nslot(sl, 1);
      step(F{}, T{}, SRC23{}, T{}, 3, sl);
    };
    if (wm == 0) {
      read_frags(sl);
      run(step_g0);
    } else {
      run(step_g1);
    }
    if constexpr (TIMING) t_kend = ps_clock();

    // ---- epilogue (through the dead slot(s) of the ring) -------------------------------------------------------------------------
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const float s = load_scale_u(G.sa_recip) * load_scale_u(G.sb_recip);
    const int M = (int)uni_u32((unsigned)G.M);
    unsigned char* wbuf = smem + (lut_tile ? lut_scratch : s_m1) * STAGE + wave * 4096;
    unsigned long long* stamps = nullptr;
    if constexpr (TIMING) {
      if (threadIdx.x == 0 && P.dbg) stamps = P.dbg + ((size_t)blockIdx.x * 8 + (jt < 8 ? jt : 7)) * 8 + 4;
    }
    ps_epilogue<ESEL, ACT_FMT, tile_kind>(G, acc, braw, s, wbuf, smem + tbl_slot * STAGE, c_m0 + wm * 128, c_n0 + wn * 64, M, lane_e, wave,
                               [&]() {  // table tiles: the successor's third and fourth K-step, into the slots the table vacates
                                 dma_stage(n_ars, n_wrs, n_asoff, n_wsoff, 2, s_m3);
                                 dma_stage(n_ars, n_wrs, n_asoff, n_wsoff, 3, nslot(s_m3, 1));
                               }, stamps);
    if constexpr (TIMING) {
      if (threadIdx.x == 0 && P.dbg) {
        unsigned long long* d = P.dbg + ((size_t)blockIdx.x * 8 + (jt < 8 ? jt : 7)) * 8;
        d[0] = t_start; d[1] = t_kend; d[2] = ps_clock(); d[3] = ps_realtime();
      }
    }
    };  // tile_body
    // fused-K tiles (whole 256-column tiles of the K columns, host check) get their own body like the table tiles: in one body with the
    // other paths their register pressure spilled the BIAS registers at the common entry of the epilogue -- a scratch reload + vmcnt(0) in
    // front of every tile's conversion
    using KPlain = std::integral_constant<int, 0>;
    using KTable = std::integral_constant<int, 1>;
    using KFused = std::integral_constant<int, 2>;
    const int kc0 = (int)uni_u32((unsigned)G.kv_col0);
    const bool k_tile = uni_ptr((const u16*)G.k_out) != nullptr && c_n0 >= kc0 && c_n0 < kc0 + (int)uni_u32((unsigned)G.heads) * 128;
    if constexpr (ESEL == FLUXMI_EPI_GELU_QUANT) {
      tile_body(KTable{});
    } else if constexpr (ESEL == FLUXMI_EPI_SPLIT) {
      if (c_n0 >= (int)uni_u32((unsigned)G.split_n)) tile_body(KTable{});
      else if (k_tile) tile_body(KFused{});
      else tile_body(KPlain{});
    } else if constexpr (ESEL == FLUXMI_EPI_BF16) {
      if (k_tile) tile_body(KFused{}); else tile_body(KPlain{});
    } else {
      tile_body(KPlain{});
    }
    if (!has_next) break;
    slot0 = nslot(slot0, nk % NS);
    c_ars = n_ars; c_wrs = n_wrs; c_asoff = n_asoff; c_wsoff = n_wsoff; c_gi = n_gi; c_m0 = n_m0; c_n0 = n_n0;
  }
  wait_vmcnt<0>();  // the trailing refills must land before the wave ends
}

template <bool FP8, int ACT, int ESEL, bool TIMING>
int launch_ps(FluxmiGemmParams& p, hipStream_t s) {
  constexpr int BM = 256, BN = 256;
  int t = 0;
  for (int i = 0; i < p.n_groups; ++i) {
    p.g[i].m_tile_start = t;
    t += (p.g[i].M + BM - 1) / BM;
  }
  p.tiles_m_total = t;
  // bands of FOUR row tiles (the one-tile-per-workgroup kernels use 8): an XCD's 32 concurrent tiles are 4 rows x 8 columns, the four A
  // panels of a band (3.1 MB at K = 3072) stay in its 4 MiB L2 while the band's columns stream by, and the fused-K / V^T tiles of a
  // launch spread more evenly over the workgroups.  Measured: qkv 145.2 -> 136.5 us, linear1 296.9 -> 292.7 us, step 41.64 -> 41.30 ms
  // (bands of 2: 307 us and +33 % fetch traffic, of 16: 316 us; profiles/r04_gemm_persist.txt section 11)
#ifndef PS_GROUP_M
#define PS_GROUP_M 4
#endif
  p.group_m = PS_GROUP_M;
  constexpr int SMEM = 5 * (BM + BN) * 64;  // five ring slots = all 160 KiB (the epilogue borrows the dead ones)
  auto kern = gemm_ps_kernel<FP8, ACT, ESEL, TIMING>;
  static bool attr_set = false;
  if (!attr_set) {
    FLUXMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set = true;
  }
  const int nblk = t * (p.N / BN);
  if (nblk == 0) return 0;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    FLUXMI_CHECK_HIP(hipGetDevice(&dev));
    FLUXMI_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    if (n_cu <= 0) n_cu = 256;
  }
  hipLaunchKernelGGL(kern, dim3(nblk < n_cu ? nblk : n_cu), dim3(512), SMEM, s, p);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

unsigned long long* g_ps_dbg = nullptr;

}  // namespace

// timing probe (tools/gemm_probe.py --timeline): device buffer of [workgroup][tile < 8][t_start, t_k_end, t_end (shader clock), realtime (100 MHz),
// table path: table DMA issued, accumulators converted, table landed (barrier), gathers done]
extern "C" int fluxmi_gemm_debug_buffer(void* dev_u64) {
  g_ps_dbg = (unsigned long long*)dev_u64;
  return 0;
}

// What the persistent kernel is built for: fp8 x e5m2 operands, the step's four hot epilogues (with the fused V^T / K outputs), K bytes
// % 256 == 0 and >= 512, one lda for every group; the quantising epilogues only through the table.
int fluxmi_gemm_persist_ok(const FluxmiGemmParams& p, int is_fp8, int act_fmt) {
  if (!is_fp8 || act_fmt != FLUXMI_FMT_E5M2) return 0;
  if (p.N % 256 != 0 || p.K % 256 != 0 || p.K < 512) return 0;
  if (p.epi != FLUXMI_EPI_BF16 && p.epi != FLUXMI_EPI_GATE_RESID && p.epi != FLUXMI_EPI_SPLIT && p.epi != FLUXMI_EPI_GELU_QUANT) return 0;
  for (int i = 0; i < p.n_groups; ++i) {
    const FluxmiGemmGroup& g = p.g[i];
    if (g.lda != p.g[0].lda || (g.W_pairs != nullptr) != (p.g[0].W_pairs != nullptr) || (g.a_pairs != 0) != (p.g[0].a_pairs != 0)) return 0;
    // fused K: whole 256-column tiles inside the K columns, head pairs (the dispatcher requires the same of configs 13 / 16)
    if (g.k_out && (g.kv_col0 % 256 != 0 || (g.heads * 128) % 256 != 0 || !g.pe || !g.k_norm || g.k_rows <= 0 ||
                    (long long)g.heads * g.k_rows * 256 >= (1LL << 32) || (p.epi != FLUXMI_EPI_BF16 && p.epi != FLUXMI_EPI_SPLIT)))
      return 0;
    if ((long long)g.M * g.lda >= (1LL << 32) || (long long)p.N * p.K >= (1LL << 32)) return 0;
    if ((p.epi == FLUXMI_EPI_SPLIT || p.epi == FLUXMI_EPI_GELU_QUANT) && !g.q_lut) return 0;
    if (p.epi == FLUXMI_EPI_SPLIT && (g.split_n % 256 != 0 || g.split_n != p.g[0].split_n)) return 0;
  }
  return 1;
}

// config 18 = persistent 256x256 ping-pong; 19 = the same with per-tile timestamps into fluxmi_gemm_debug_buffer
int fluxmi_launch_gemm_persist(FluxmiGemmParams& p, int is_fp8, int act_fmt, int timing, hipStream_t s) {
  FLUXMI_REQUIRE(fluxmi_gemm_persist_ok(p, is_fp8, act_fmt),
                 "gemm_persist: needs fp8 x e5m2 operands, N %% 256 == 0, K %% 256 == 0, K >= 512, a bf16 / gate_resid / split / gelu_quant "
                 "epilogue (the quantising ones with a table), one lda (N=%d K=%d epi=%d)", p.N, p.K, p.epi);
  p.dbg = timing ? g_ps_dbg : nullptr;
  if (timing) {
    switch (p.epi) {
      case FLUXMI_EPI_BF16: return launch_ps<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_BF16, true>(p, s);
      case FLUXMI_EPI_GATE_RESID: return launch_ps<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_GATE_RESID, true>(p, s);
      case FLUXMI_EPI_SPLIT: return launch_ps<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_SPLIT, true>(p, s);
      default: return launch_ps<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_GELU_QUANT, true>(p, s);
    }
  }
  switch (p.epi) {
    case FLUXMI_EPI_BF16: return launch_ps<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_BF16, false>(p, s);
    case FLUXMI_EPI_GATE_RESID: return launch_ps<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_GATE_RESID, false>(p, s);
    case FLUXMI_EPI_SPLIT: return launch_ps<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_SPLIT, false>(p, s);
    default: return launch_ps<true, FLUXMI_FMT_E5M2, FLUXMI_EPI_GELU_QUANT, false>(p, s);
  }
}
