// fluxmi -- C-ABI entry points for the individual operators (include/fluxmi.h).
// Compiled with hipcc (host-only translation unit; kernels live in the .hip files).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#include "fluxmi_internal.h"

static thread_local char g_err[1024] = "";

void fluxmi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Tile choice: minimise (#waves of tiles over the 256 CUs) x (per-tile cost).  Relative per-tile
// efficiencies were measured on MI355X (profiles/r01_kernel_sweep.txt); fluxmi_tuning_t.gemm_cfg overrides.
int fluxmi_gemm_auto_cfg(const FluxmiGemmParams& p, int is_fp8) {
  // a forced tile config (FLUXMI_GEMM_CFG) is taken where it applies; the persistent kernel (18 / its timing build 19) has conditions on
  // the operand format and the epilogue that this function does not see: run_gemm_chunk applies it after fluxmi_gemm_persist_ok, and a
  // launch it does not fit falls back to the cost model below instead of failing in the launcher
  const int forced = fluxmi_tuning().gemm_cfg;
  if (forced >= 0 && forced != 18 && forced != 19 && fluxmi_gemm_tile_ok(p.N, p.K, is_fp8, forced)) return forced;
  // candidates, in order of preference at equal cost; eff = measured relative rate per flop on MI355X at full occupancy
  // (profiles/r01_kernel_sweep.txt): 13 = 256x256 ping-pong ring (1 block/CU), 2 = 128x128 double-buffered (2 blocks/CU), 15 = 128x64
  // (narrow N).  Cost = (number of block waves) x (time of one wave of blocks).
  // 16 = 256x256 with ONE wave per SIMD (128x128 wave tiles): the leaner main loop wins once K is long enough to amortise its
  // twice-as-long per-wave epilogue (measured +6 % at K = 15360, -5 % at K = 3072)
  const bool long_k = (long long)p.K * (is_fp8 ? 1 : 2) >= 8192;
  constexpr int NC = 4;
  static const int cand[NC] = {13, 16, 2, 15};
  const double eff[NC] = {1.00, long_k ? 1.06 : 0.94, 0.84, 0.30};
  static const int occ[NC] = {1, 1, 2, 3};
  int best = -1;
  double best_cost = 1e300;
  for (int ci = 0; ci < NC; ++ci) {
    const int c = cand[ci];
    if (!fluxmi_gemm_tile_ok(p.N, p.K, is_fp8, c)) continue;
    const int bm = fluxmi_gemm_tile_bm(c), bn = fluxmi_gemm_tile_bn(c);
    long long tiles = 0;
    for (int i = 0; i < p.n_groups; ++i) tiles += (p.g[i].M + bm - 1) / bm;
    tiles *= p.N / bn;
    const long long slots = 256LL * occ[ci];
    const long long waves = (tiles + slots - 1) / slots;
    // a last wave that fills less than half of a 2-blocks/CU machine runs its blocks alone on their CUs (~1.6x faster)
    double w = (double)waves;
    if (occ[ci] == 2 && tiles - (waves - 1) * slots <= 256) w -= 0.4;
    const double cost = w * occ[ci] * (double)bm * bn / eff[ci];
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// mods_gemm (engine.hip) needs results that do not depend on how many rows share a launch (the step-ahead table of R = steps x B rows
// must equal the per-step R = B launches bit for bit): it switches the M-dependent split-K choice off around its launches
static thread_local int g_splitk_block = 0;
void fluxmi_gemm_block_splitk(int on) { g_splitk_block += on ? 1 : -1; }

// The split-K choice of a bf16 launch (below) depends on how many tiles the launch has, and a split-K sum associates K differently from the
// one-pass kernels (whose tile configs all give the same bits): decided on ALL rows of a batched launch, a sample's result would follow the
// batch it rides in (Flux-schnell 256^2, bf16 flow: split-K at B = 1, none from B = 4 on).  An engine therefore announces its batch
// (fluxmi_gemm_set_batch, thread-local like the scratch pointers) and the dispatcher decides on ONE sample's share of the groups;
// reference flux_model.py:672-716 has no cross-sample operation.
static thread_local int g_gemm_batch = 1;
void fluxmi_gemm_set_batch(int B) { g_gemm_batch = B < 1 ? 1 : B; }

// split-K slices for a bf16 launch of these groups (0 = one pass); `tiles`, `rows`: 256-row tiles x N / 256 and padded rows of the groups
static int splitk_slices(long long tiles, long long rows, int N, int K, int is_fp8, int epi, bool fused_out, int force_cfg) {
  const fluxmi_tuning_t tun = fluxmi_tuning();
  const int nk = K * (is_fp8 ? 1 : 2) / 64;
  const bool can = !fused_out && (epi == FLUXMI_EPI_BF16 || epi == FLUXMI_EPI_GATE_RESID) && fluxmi_gemm_tile_ok(N, K, is_fp8, 13) && tiles > 0;
  int S = 0;
  // measured on M = 512 bf16 launches (tools/bf16_gemm_probe.py, profiles/r03_small_m.txt): below ~190 K-steps per tile the 128x128 tiles at two
  // workgroups per CU are as fast as any split; above, ~40-50 K-steps per workgroup is the sweet spot (K = 15360: 72 us vs 158 us unsplit)
  if (force_cfg < 0 && tun.gemm_splitk && !g_splitk_block && can && !is_fp8 && tun.gemm_cfg < 0 && tiles <= 128 && nk >= 192)
    S = (int)std::max<long long>(2, std::min<long long>(std::min<long long>(256 / tiles, (nk + 24) / 48), 16));
  while (S >= 2 && (size_t)S * rows * N * 4 > ((size_t)256 << 20)) --S;
  return S >= 2 ? S : 0;
}

// s_hint: -1 = decide the split-K slices on this chunk's own groups; >= 0 = decided by the caller on one sample's groups (0: one pass)
static int run_gemm_chunk(const FluxmiGemmGroup* gs, int n, int N, int K, int is_fp8, int act_fmt, int epi, int force_cfg, hipStream_t s,
                          int s_hint = -1) {
  FluxmiGemmParams p;
  memset(&p, 0, sizeof(p));
  p.n_groups = n;
  for (int i = 0; i < n; ++i) p.g[i] = gs[i];
  p.N = N; p.K = K; p.epi = epi;
  int cfg = force_cfg >= 0 && fluxmi_gemm_tile_ok(N, K, is_fp8, force_cfg) ? force_cfg : fluxmi_gemm_auto_cfg(p, is_fp8);
  {
    const int forced = fluxmi_tuning().gemm_cfg;
    if (force_cfg < 0 && (forced == 18 || forced == 19) && fluxmi_gemm_tile_ok(N, K, is_fp8, 13) && fluxmi_gemm_persist_ok(p, is_fp8, act_fmt)) cfg = forced;
  }
  // small-M launches (M <= 512: schnell 256x256, the text encoders, the modulation GEMMs): 24-96 tiles of 256x256 for 256 CUs, weight-stream
  // bound -> split K over several workgroups per tile (fp32 partials + a reduce / epilogue pass).  fluxmi_tuning_t.gemm_splitk = 0 turns it off;
  // fluxmi_gemm_grouped(tile_cfg = 113 + S) forces S splits (tests).
  {
    bool fused_out = false;
    long long tiles = 0, rows = 0;
    for (int i = 0; i < n; ++i) { tiles += (gs[i].M + 255) / 256; rows += (long long)((gs[i].M + 255) / 256) * 256; fused_out |= (gs[i].vt_out || gs[i].k_out); }
    tiles *= N / 256;
    const int S = s_hint >= 0 ? s_hint : splitk_slices(tiles, rows, N, K, is_fp8, epi, fused_out, force_cfg);
    if (S >= 2) {
      // the fp32 partial tiles of S slices must fit the scratch: a batched launch whose S was fixed on one sample goes in as many pieces as it takes
      const size_t cap = (size_t)256 << 20;
      if ((size_t)S * rows * N * 4 <= cap) return fluxmi_launch_gemm_splitk(p, is_fp8, act_fmt, S, s);
      FLUXMI_REQUIRE(s_hint >= 0, "gemm: split-K scratch too small for %d slices", S);
      for (int i0 = 0; i0 < n;) {
        FluxmiGemmParams q = p;
        q.n_groups = 0;
        size_t r = 0;
        while (i0 < n) {
          const size_t gr = (size_t)((gs[i0].M + 255) / 256) * 256;
          if (q.n_groups > 0 && (size_t)S * (r + gr) * N * 4 > cap) break;
          FLUXMI_REQUIRE((size_t)S * gr * N * 4 <= cap, "gemm: one group of %d rows does not fit the split-K scratch at %d slices", gs[i0].M, S);
          q.g[q.n_groups++] = gs[i0++];
          r += gr;
        }
        FLUXMI_TRY(fluxmi_launch_gemm_splitk(q, is_fp8, act_fmt, S, s));
      }
      return 0;
    }
  }
  // bf16 operands, one thin round of 256x256 tiles (Flux-schnell linear1 at M = 512: 168 tiles): the one-wave-per-SIMD kernel runs the
  // single tile per CU fastest (74 us vs 95 / 98 us for configs 13 / 2, profiles/r03_small_m.txt)
  if (!is_fp8 && force_cfg < 0 && fluxmi_tuning().gemm_cfg < 0 && fluxmi_gemm_tile_ok(N, K, is_fp8, 16)) {
    long long t256 = 0;
    bool fused_out = false;
    for (int i = 0; i < n; ++i) { t256 += (gs[i].M + 255) / 256; fused_out |= (gs[i].vt_out || gs[i].k_out); }
    t256 *= N / 256;
    if (t256 > 128 && t256 <= 256 && (!fused_out || cfg == 13)) cfg = 16;
    // ... and on 192-row tiles when those still fit one round (M = 512, N = 21504: 3 x 84 = 252 tiles of three quarters the work instead of 168)
    if (cfg == 16 && !fused_out && (epi == FLUXMI_EPI_BF16 || epi == FLUXMI_EPI_GATE_RESID) && fluxmi_tuning().gemm_tile192) {
      long long t192 = 0;
      for (int i = 0; i < n; ++i) t192 += (gs[i].M + 191) / 192;
      t192 *= N / 256;
      if (t192 <= 256 && t192 > t256) cfg = 17;
    }
  }
  // multi-round fp8 launches of the step (single-block linear1: 5.9 rounds of the 256 CUs, double-block mlp.0: 3.0, qkv: 2.25): one
  // persistent workgroup per CU walks the tiles -- no workgroup relaunch, cold prologue or store drain per tile (gemm_persist.hip).
  // Single-round launches gain nothing from it and keep the one-tile-per-workgroup kernel.
  if (cfg == 13 && fluxmi_tuning().gemm_persist && fluxmi_tuning().gemm_cfg < 0 && fluxmi_gemm_persist_ok(p, is_fp8, act_fmt)) {
    long long t256 = 0;
    for (int i = 0; i < n; ++i) t256 += (gs[i].M + 255) / 256;
    t256 *= N / 256;
    if (t256 > 256) cfg = fluxmi_tuning().gemm_persist == 2 ? 19 : 18;  // 2: the timing build (probes: fluxmi_gemm_debug_buffer)
  }
  // gate*y+x launches of the one-wave-per-SIMD kernel whose 256-row tiling fills less than one round of the 256 CUs: lower tiles of the same
  // kernel (same bits).  Flux-dev 768^2 (M = 2816 -> 11 x 12 = 132 tiles on mlp.2 / linear2): 192-row tiles (config 17) are three quarters of
  // the work each and 15 x 12 = 180 of them still run in one round (round 5, -13.6 % per launch).  Flux-dev 1024^2 linear2 (M = 4608 -> 216
  // tiles): 224-row tiles (config 20, round 6: the four waves side by side along N) give 21 x 12 = 252 tiles = one round at 7/8 of the work
  // (isolated, cold operands: 203.7 -> 181.0 us).  Cost = rounds x tile height x a per-height factor for the fragment bytes per MFMA (the
  // 192-row 2 x 2 grid reads 8 % more, the 224-row 1 x 4 grid 29 % more but keeps all four SIMDs equally loaded); taken when it drops by more
  // than 5 %.  fluxmi_tuning_t.gemm_tile192 = 0 turns both off.
  if (cfg == 16 && force_cfg < 0 && is_fp8 && act_fmt == FLUXMI_E5M2 && epi == FLUXMI_EPI_GATE_RESID && fluxmi_tuning().gemm_cfg < 0 &&
      fluxmi_tuning().gemm_tile192) {
    auto tiles_of = [&](int bm) {
      long long t = 0;
      for (int i = 0; i < n; ++i) t += (gs[i].M + bm - 1) / bm;
      return t * (N / 256);
    };
    const double c256 = (double)((tiles_of(256) + 255) / 256);
    double best = 0.95 * c256;
    if (fluxmi_gemm_tile_ok(N, K, is_fp8, 17)) {
      const double c = (double)((tiles_of(192) + 255) / 256) * 0.75 * 1.04;
      if (c < best) { best = c; cfg = 17; }
    }
    const bool lower = fluxmi_tuning().gemm_tile192 == 1;  // 2 = config 17 only (A/B of the round-6 heights)
    if (lower && fluxmi_gemm_tile_ok(N, K, is_fp8, 20)) {
      const double c = (double)((tiles_of(224) + 255) / 256) * 0.875 * 1.03;
      if (c < best) { best = c; cfg = 20; }
    }
    if (lower && fluxmi_gemm_tile_ok(N, K, is_fp8, 21)) {  // 160-row tiles (768^2: 18 x 12 = 216 tiles)
      const double c = (double)((tiles_of(160) + 255) / 256) * 0.625 * 1.08;
      if (c < best) { best = c; cfg = 21; }
    }
  }
  const bool split_ok = epi != FLUXMI_EPI_SPLIT || cfg < 0 || (p.g[0].split_n % fluxmi_gemm_tile_bn(cfg) == 0);
  if (cfg < 0 || !split_ok) return fluxmi_launch_gemm_generic(p, is_fp8, act_fmt, s);
  return fluxmi_launch_gemm(p, is_fp8, act_fmt, cfg, s);
}

// One grouped GEMM.  When the 256x256 tiling leaves a thin last round of tiles (e.g. double-block mlp.0: img 768 tiles = 3.0 rounds
// of the 256 CUs, txt 96 more tiles -> a 4th round at 37 % occupancy), the smallest groups are peeled off into a second launch
// with 128x128 tiles at two workgroups per CU: 3 + ~0.6 rounds instead of 4.  Results do not depend on the tile shape.
int fluxmi_gemm_dispatch(const FluxmiGemmGroup* gs_in, int n_in, int N, int K, int is_fp8, int act_fmt, int epi, hipStream_t s) {
  std::vector<FluxmiGemmGroup> gs(gs_in, gs_in + n_in);
  const fluxmi_tuning_t tun = fluxmi_tuning();
  const int hybrid = tun.gemm_hybrid;
  bool fused_attn = false;
  for (auto& g : gs) fused_attn |= (g.vt_out != nullptr || g.k_out != nullptr);
  if (fused_attn) {
    // the attention-layout epilogue lives in the LDS-transposed epilogue of the 256x256 kernels only
    const bool long_k = (long long)K * (is_fp8 ? 1 : 2) >= 8192;
    const int cfg = (long_k && fluxmi_gemm_tile_ok(N, K, is_fp8, 16)) ? 16 : 13;
    FLUXMI_REQUIRE(fluxmi_gemm_tile_ok(N, K, is_fp8, cfg), "gemm: fused K / V^T outputs need N %% 256 == 0 and K*bytes %% 64 == 0 (N=%d K=%d)", N, K);
    for (auto& g : gs)
      FLUXMI_REQUIRE(g.heads > 0 && g.kv_col0 % 128 == 0 && g.tok0 % 16 == 0 && g.vt_rows % 8 == 0 && g.vt_ld % 8 == 0 &&
                         (!g.k_out || (g.kv_col0 % 256 == 0 && (g.heads * 128) % 256 == 0 && g.pe && g.k_norm && g.k_rows > 0)),
                     "gemm: fused K / V^T outputs need tok0 %% 16 == 0, vt_rows %% 8 == 0, vt_ld %% 8 == 0 (K: 256-aligned q|k|v blocks, pe, k_norm)");
    for (size_t off = 0; off < gs.size(); off += FLUXMI_MAX_GROUPS)
      FLUXMI_TRY(run_gemm_chunk(gs.data() + off, (int)std::min<size_t>(FLUXMI_MAX_GROUPS, gs.size() - off), N, K, is_fp8, act_fmt, epi, cfg, s));
    return 0;
  }
  const bool peel_ok = hybrid && tun.gemm_cfg < 0 && fluxmi_gemm_tile_ok(N, K, is_fp8, 13) && fluxmi_gemm_tile_ok(N, K, is_fp8, 2) &&
                       (epi != FLUXMI_EPI_SPLIT || gs[0].split_n % 256 == 0);
  // how many of the smallest groups (row counts `ms`, ascending) go to the 128x128 launch
  auto peel_count = [&](const std::vector<int>& ms) -> int {
    if (!peel_ok || ms.size() < 2 || ms.size() > FLUXMI_MAX_GROUPS) return 0;
    const long long tn = N / 256;
    long long T = 0;
    for (int m : ms) T += (long long)((m + 255) / 256) * tn;
    // multi-round fp8 launches run on the PERSISTENT kernel, whose last, partial round costs what its tiles cost: peeling only pays when that
    // round is thin.  Measured in-step after the row-pair activations (profiles/r06_act_pairs.txt section 5): Flux-dev 1024^2 mlp.0, 864 tiles
    // = 3 rounds + 96 tiles (37 % of a round): the peel LOSES 0.4 % per step; 768^2, 528 tiles = 2 rounds + 16 tiles: it gains 1.9 %
    if (is_fp8 && act_fmt == FLUXMI_E5M2 && tun.gemm_persist && T > 256 && T % 256 > 64) return 0;
    double best = (double)((T + 255) / 256) - 0.15;
    int best_k = 0;
    long long peeled = 0;
    for (size_t k = 1; k < ms.size(); ++k) {  // peel the k smallest groups
      peeled += (long long)((ms[k - 1] + 255) / 256) * tn;
      long long small_tiles = 0;
      for (size_t q = 0; q < k; ++q) small_tiles += (long long)((ms[q] + 127) / 128) * (N / 128);
      const double cost = (double)((T - peeled + 255) / 256) + 0.58 * (double)((small_tiles + 511) / 512);
      if (cost < best) { best = cost; best_k = (int)k; }
    }
    return best_k;
  };
  auto chunks = [&](std::vector<FluxmiGemmGroup>& v, int force_cfg, int s_hint) -> int {
    for (size_t off = 0; off < v.size(); off += FLUXMI_MAX_GROUPS)
      FLUXMI_TRY(run_gemm_chunk(v.data() + off, (int)std::min<size_t>(FLUXMI_MAX_GROUPS, v.size() - off), N, K, is_fp8, act_fmt, epi, force_cfg, s, s_hint));
    return 0;
  };
  // bf16 launches of a batched engine (fluxmi_gemm_set_batch): replay the decisions of ONE sample's launch -- which groups are peeled, how many
  // split-K slices the others get -- and apply them to every sample's groups.  One sample's share = 1 / batch of the groups of every row
  // count (the engine pushes one group per (sample, stream)), or 1 / batch of the rows of a single group that carries the whole batch.  The
  // one-pass tile configs all give the same bits, so only the split-K slices have to follow the sample; a launch whose groups do not divide
  // by the batch keeps the whole-launch decision.
  if (!is_fp8 && g_gemm_batch > 1) {
    const int B = g_gemm_batch;
    bool ok = true;
    std::vector<int> all, sub;  // row counts of the launch / of one sample's share, ascending
    std::map<int, int> per_sample;  // rows of a group of the launch -> rows of it that belong to one sample
    for (auto& g : gs) all.push_back(g.M);
    std::sort(all.begin(), all.end());
    for (size_t i = 0; i < all.size() && ok;) {
      size_t j = i;
      while (j < all.size() && all[j] == all[i]) ++j;
      const size_t cnt = j - i;
      if (cnt % B == 0) { sub.insert(sub.end(), cnt / B, all[i]); per_sample[all[i]] = all[i]; }
      else if (cnt == 1 && all[i] % B == 0) { sub.push_back(all[i] / B); per_sample[all[i]] = all[i] / B; }
      else ok = false;
      i = j;
    }
    std::sort(sub.begin(), sub.end());
    const int k = ok ? peel_count(sub) : 0;
    if (ok && k > 0 && k < (int)sub.size() && sub[k - 1] == sub[k]) ok = false;  // the peel would cut through groups of one row count
    if (ok) {
      const int m_small = k > 0 ? sub[k - 1] : -1;  // groups of at most this many rows (per sample) are peeled
      long long tiles = 0, rows = 0;
      for (size_t q = (size_t)k; q < sub.size(); ++q) { tiles += (sub[q] + 255) / 256; rows += (long long)((sub[q] + 255) / 256) * 256; }
      // one sample's launch decides per chunk of FLUXMI_MAX_GROUPS groups; a sample has a handful, so its launch is one chunk
      ok = sub.size() <= FLUXMI_MAX_GROUPS;
      if (ok) {
        const int S = splitk_slices(tiles * (N / 256), rows, N, K, is_fp8, epi, false, -1);
        std::vector<FluxmiGemmGroup> big, small;
        for (auto& g : gs) (per_sample[g.M] <= m_small ? small : big).push_back(g);
        FLUXMI_TRY(chunks(big, -1, S));
        return chunks(small, 2, 0);
      }
    }
  }
  if (peel_ok && gs.size() >= 2 && gs.size() <= FLUXMI_MAX_GROUPS) {
    std::vector<int> order(gs.size());
    for (size_t i = 0; i < gs.size(); ++i) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return gs[a].M < gs[b].M; });
    std::vector<int> ms;
    for (int i : order) ms.push_back(gs[i].M);
    const int best_k = peel_count(ms);
    if (best_k > 0) {
      std::vector<FluxmiGemmGroup> big, small;
      for (size_t q = 0; q < gs.size(); ++q) (q < (size_t)best_k ? small : big).push_back(gs[order[q]]);
      // (running the thin launch on a side stream BESIDE the big one, fork / join through events, was measured slower: 45.47 vs
      // 45.13 ms per step, profiles/r02_gemm_ab.txt -- its workgroups take CUs from the big grid's first rounds, not its last)
      FLUXMI_TRY(run_gemm_chunk(big.data(), (int)big.size(), N, K, is_fp8, act_fmt, epi, -1, s));
      return run_gemm_chunk(small.data(), (int)small.size(), N, K, is_fp8, act_fmt, epi, 2, s);
    }
  }
  return chunks(gs, -1, -1);
}

extern "C" {

const char* fluxmi_last_error(void) { return g_err; }
int fluxmi_abi_version(void) { return FLUXMI_ABI_VERSION; }

int fluxmi_gemm_grouped(const fluxmi_gemm_group_t* groups, int n_groups, int N, int K, int is_fp8, int act_fmt, int epilogue,
                        int tile_cfg, void* stream) {
  FLUXMI_REQUIRE(groups && n_groups >= 1 && n_groups <= FLUXMI_MAX_GROUPS, "gemm_grouped: n_groups=%d (1..%d)", n_groups, FLUXMI_MAX_GROUPS);
  FLUXMI_REQUIRE(N > 0 && K > 0, "gemm_grouped: bad N=%d K=%d", N, K);
  FluxmiGemmParams p;
  memset(&p, 0, sizeof(p));
  for (int i = 0; i < n_groups; ++i) {
    p.g[i] = groups[i];
    FLUXMI_REQUIRE(groups[i].M >= 0, "gemm_grouped: group %d has M=%d", i, groups[i].M);
    FLUXMI_REQUIRE(groups[i].M == 0 || (groups[i].A && groups[i].W && groups[i].C), "gemm_grouped: group %d has NULL A/W/C", i);
  }
  p.n_groups = n_groups; p.N = N; p.K = K; p.epi = epilogue;
  if (tile_cfg == 100) return fluxmi_launch_gemm_generic(p, is_fp8, act_fmt, (hipStream_t)stream);
  if (tile_cfg >= 115 && tile_cfg <= 113 + 32) return fluxmi_launch_gemm_splitk(p, is_fp8, act_fmt, tile_cfg - 113, (hipStream_t)stream);
  if (tile_cfg < 0) return fluxmi_gemm_dispatch(p.g, p.n_groups, N, K, is_fp8, act_fmt, epilogue, (hipStream_t)stream);
  return fluxmi_launch_gemm(p, is_fp8, act_fmt, tile_cfg, (hipStream_t)stream);
}

int fluxmi_f8_gemm(const void* a_fp8, const void* w_e4m3, const float* sa_recip, const float* sb_recip, const void* bias, void* out,
                   int M, int N, int K, int act_fmt, int epilogue, const void* gate, const void* resid, const float* q_scale,
                   int tile_cfg, void* stream) {
  fluxmi_gemm_group_t g;
  memset(&g, 0, sizeof(g));
  g.A = a_fp8; g.W = w_e4m3; g.bias = bias; g.sa_recip = sa_recip; g.sb_recip = sb_recip;
  g.C = out; g.gate = gate; g.resid = resid; g.q_scale = q_scale;
  g.lda = K; g.ldc = N; g.ldr = N; g.M = M;
  return fluxmi_gemm_grouped(&g, 1, N, K, 1, act_fmt, epilogue, tile_cfg, stream);
}

int fluxmi_gemv(const void* x, long long ldx, const void* W, const void* bias, const float* in_scale, const float* sa_recip,
                const float* sb_recip, void* out, long long ld_out, int B, int N, int K, int w_fp8, int act_fmt, int pre_silu,
                void* stream) {
  FluxmiGemvLayer L;
  memset(&L, 0, sizeof(L));
  L.W = W; L.bias = bias; L.in_scale = in_scale; L.sa_recip = sa_recip; L.sb_recip = sb_recip;
  L.out = out; L.x = x; L.ld_out = ld_out; L.ldx = ldx; L.N = N; L.K = K;
  L.w_fp8 = w_fp8; L.pre_silu = pre_silu; L.act_fmt = act_fmt;
  return fluxmi_launch_gemv(nullptr, &L, 1, B, 0, 0, (hipStream_t)stream);
}

int fluxmi_quantize_act(const void* x, void* q, const float* scale, int rows, int cols, long long ld_in, long long ld_out, int fmt,
                        void* stream) {
  return fluxmi_k_quantize_act(x, q, scale, rows, cols, ld_in, ld_out, fmt, (hipStream_t)stream);
}
int fluxmi_amax(const void* x, float* amax, int rows, int cols, long long ld, void* stream) {
  return fluxmi_k_amax(x, amax, rows, cols, ld, (hipStream_t)stream);
}
int fluxmi_calib_update(const float* amax, float* trials, float* scale, float* scale_recip, int trial_index, int num_trials,
                        float max_val, void* stream) {
  FLUXMI_REQUIRE(trial_index >= 0 && trial_index <= num_trials, "calib_update: trial_index %d out of range", trial_index);
  return fluxmi_k_calib_update(amax, trials, scale, scale_recip, trial_index, num_trials, max_val, (hipStream_t)stream);
}
int fluxmi_quantize_weight(const void* w_bf16, void* q, float* amax_tmp, float* scale, float* scale_recip, int N, int K, int fmt,
                           void* stream) {
  return fluxmi_k_quantize_weight(w_bf16, q, amax_tmp, scale, scale_recip, N, K, fmt, (hipStream_t)stream);
}
int fluxmi_dequant(const void* q, float* out, const float* scale_recip, long long n, int fmt, void* stream) {
  return fluxmi_k_dequant(q, out, scale_recip, n, fmt, (hipStream_t)stream);
}

// W' = requant( bf16( dequant(W) + sum_c lora_scale * B @ A_chunk_c ) ).  lora_A is [n_chunks*R, K] (already
// multiplied by alpha/rank on the host as the reference does, lora_loading.py:529-530), lora_B is [N, R].
int fluxmi_lora_fuse_f8(void* w_fp8, float* w_scale, float* w_scale_recip, const float* lora_B, const float* lora_A, int N, int K,
                        int R, int n_chunks, float lora_scale, float* work_f32, float* amax_tmp, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)N * K;
  FLUXMI_REQUIRE(n % 4 == 0 && n_chunks >= 1, "lora_fuse: bad shape N=%d K=%d chunks=%d", N, K, n_chunks);
  // delta accumulates in work_f32[n .. 2n), dequantised weight in work_f32[0 .. n)
  float* w32 = work_f32;
  float* delta = work_f32 + n;
  FLUXMI_TRY(fluxmi_k_dequant(w_fp8, w32, w_scale_recip, n, FLUXMI_E4M3, s));
  for (int c = 0; c < n_chunks; ++c)
    FLUXMI_TRY(fluxmi_k_lora_delta(lora_B, lora_A + (long long)c * R * K, delta, N, K, R, lora_scale, c > 0, s));
  FLUXMI_TRY(fluxmi_k_axpy_f32(w32, delta, 1.0f, n, s));
  return fluxmi_k_requantize_f32(w32, w_fp8, amax_tmp, w_scale, w_scale_recip, n, FLUXMI_E4M3, s);
}

int fluxmi_ln_modulate(const void* x, long long ldx, void* out, long long ldo, const void* shift0, const void* scale0,
                       const void* shift1, const void* scale1, long long mod_bstride, const float* q_scale0, const float* q_scale1,
                       int B, int L, int split, int H, int out_fp8, int fmt, void* stream) {
  return fluxmi_k_ln_modulate(x, ldx, (long long)L * ldx, out, ldo, (long long)L * ldo, shift0, scale0, shift1, scale1, mod_bstride,
                              q_scale0, q_scale1, B, L, split, H, out_fp8, fmt, (hipStream_t)stream);
}
int fluxmi_act(const void* x, void* y, int rows, int cols, long long ld_in, long long ld_out, int mode, void* stream) {
  return fluxmi_k_act(x, y, rows, cols, ld_in, ld_out, mode, (hipStream_t)stream);
}
int fluxmi_gate_residual(const void* x, const void* y, const void* gate, void* out, int B, int L, int H, long long ldx, long long ldy,
                         long long ldo, long long gate_bstride, void* stream) {
  return fluxmi_k_gate_residual(x, y, gate, out, B, L, H, ldx, ldy, ldo, gate_bstride, (hipStream_t)stream);
}
int fluxmi_add(const void* a, const void* b, void* z, long long n, void* stream) {
  return fluxmi_k_add(a, b, z, n, (hipStream_t)stream);
}
int fluxmi_rope_table(const void* ids, const float* omega, const int* axis, void* pe, long long rows, int n_axes, int pairs,
                      void* stream) {
  return fluxmi_k_rope_table(ids, omega, axis, pe, rows, n_axes, pairs, (hipStream_t)stream);
}
int fluxmi_qkv_rope(const void* qkv, long long ld, const void* pe, const void* q_scale0, const void* k_scale0, const void* q_scale1,
                    const void* k_scale1, void* Q, void* K, void* VT, int B, int L, int Lp, int H, int split, int k_f16, void* stream) {
  return fluxmi_k_qkv_rope(qkv, ld, pe, q_scale0, k_scale0, q_scale1, k_scale1, Q, K, VT, B, L, Lp, H, split, k_f16, (hipStream_t)stream);
}
int fluxmi_attention(const void* Q, const void* K, const void* VT, void* out, long long ld_out, int col_off, int out_fp8,
                     const float* q_scale0, const float* q_scale1, int split, int B, int L, int Lp, int H, int fmt, int k_f16, void* stream) {
  return fluxmi_k_attention(Q, K, VT, out, ld_out, col_off, out_fp8, q_scale0, q_scale1, split, B, L, Lp, H, fmt, (hipStream_t)stream,
                            nullptr, 0, nullptr, nullptr, nullptr, k_f16);
}
int fluxmi_im2col3x3(const void* x, void* col, int B, int H, int W, int C, int upsample, void* stream) {
  return fluxmi_k_im2col3x3(x, col, B, H, W, C, upsample, (hipStream_t)stream);
}
int fluxmi_conv3x3(const void* x, const void* w2, const void* bias, const void* gate, const void* resid, void* out, int B, int H, int W, int C,
                   int Cout, int upsample, void* stream) {
  FLUXMI_REQUIRE(x && w2 && out && B >= 1 && H >= 1 && W >= 1, "conv3x3: NULL tensor or empty grid");
  FLUXMI_REQUIRE(upsample == 1 || upsample == 2 || upsample == -2, "conv3x3: mode in {1, 2, -2}");
  FLUXMI_REQUIRE(upsample != 2 || (H % 2 == 0 && W % 2 == 0), "conv3x3: upsampled output dims must be even");
  FLUXMI_REQUIRE((long long)B * H * W < (1LL << 31), "conv3x3: more than 2^31 output pixels");
  FLUXMI_REQUIRE(!resid || gate, "conv3x3: a residual needs the gate vector (ones for a plain add)");
  FluxmiGemmParams p;
  memset(&p, 0, sizeof(p));
  FluxmiGemmGroup& g = p.g[0];
  g.A = x; g.W = w2; g.bias = bias; g.C = out; g.gate = gate; g.resid = resid;
  g.lda = C; g.ldc = Cout; g.ldr = Cout; g.M = B * H * W;
  p.n_groups = 1; p.N = Cout; p.K = 9 * C; p.epi = resid ? FLUXMI_EPI_GATE_RESID : FLUXMI_EPI_BF16;
  p.conv.zeros = fluxmi_zero_page();
  p.conv.Hi = upsample == 2 ? H / 2 : upsample == -2 ? H * 2 : H;
  p.conv.Wi = upsample == 2 ? W / 2 : upsample == -2 ? W * 2 : W;
  p.conv.C = C; p.conv.Ho = H; p.conv.Wo = W;
  p.conv.stride = upsample == -2 ? 2 : 1; p.conv.pad = upsample == -2 ? 0 : 1; p.conv.rshift = upsample == 2 ? 1 : 0;
  return fluxmi_launch_gemm_conv(p, (hipStream_t)stream);
}
int fluxmi_groupnorm(const void* x, const void* gamma, const void* beta, void* y, float* work, int B, int P, int C, int swish, float eps,
                     void* stream) {
  return fluxmi_k_groupnorm(x, gamma, beta, y, work, B, P, C, swish, eps, (hipStream_t)stream);
}
int fluxmi_softmax_rows(const void* S, void* P, int rows, int cols, long long ld, float scale, void* stream) {
  return fluxmi_k_softmax_rows(S, P, rows, cols, ld, scale, (hipStream_t)stream);
}
int fluxmi_row_norm(const void* x, const void* weight, const void* bias, void* y, int rows, int D, long long ldx, long long ldy, float eps, int mode,
                    void* stream) {
  return fluxmi_k_row_norm(x, weight, bias, y, rows, D, ldx, ldy, eps, mode, (hipStream_t)stream);
}
int fluxmi_act_mul(const void* in, void* out, int rows, int F, long long ld_in, long long ld_out, int mode, void* stream) {
  return fluxmi_k_act_mul(in, out, rows, F, ld_in, ld_out, mode, (hipStream_t)stream);
}
int fluxmi_text_attention(const void* q, const void* k, long long ld_qk, const void* vt, long long ld_vt, void* out, long long ld_out,
                          const float* rel_bias, int bias_ld, const void* v_bias, float scale, int causal, int L, int Lp, int H, void* stream) {
  return fluxmi_k_text_attention(q, k, ld_qk, vt, ld_vt, out, ld_out, rel_bias, bias_ld, v_bias, scale, causal, L, Lp, H, (hipStream_t)stream);
}
int fluxmi_build_quant_lut(const float* scale, int fmt, int act, void* lut, void* stream) {
  return fluxmi_k_build_qlut(scale, fmt, act, lut, (hipStream_t)stream);
}
int fluxmi_pair_rows(const void* in, void* out, int rows, long long row_bytes, void* stream) {
  return fluxmi_k_pair_rows(in, out, rows, row_bytes, (hipStream_t)stream);
}
int fluxmi_unpair_rows(const void* in, void* out, int rows, long long row_bytes, void* stream) {
  return fluxmi_k_unpair_rows(in, out, rows, row_bytes, (hipStream_t)stream);
}
int fluxmi_attention_debug_buffer(void* dev_u64) { return fluxmi_attn_debug_buffer(dev_u64); }

int fluxmi_attention_plan(int B, int L, int H, int* n_per_x, int* full_per_x, int* npieces, unsigned long long* pieces) {
  return fluxmi_attn_plan_export(B, L, H, n_per_x, full_per_x, npieces, pieces);
}

int fluxmi_attention_rawq(const void* qkv, long long ld_qkv, const void* pe, const void* qn_scale0, const void* qn_scale1, const void* K,
                          const void* VT, void* out, long long ld_out, int col_off, int out_fp8, const float* q_scale0,
                          const float* q_scale1, int split, int B, int L, int Lp, int H, int fmt, int k_f16, void* stream) {
  return fluxmi_k_attention(nullptr, K, VT, out, ld_out, col_off, out_fp8, q_scale0, q_scale1, split, B, L, Lp, H, fmt, (hipStream_t)stream,
                            qkv, ld_qkv, pe, qn_scale0, qn_scale1, k_f16);
}
int fluxmi_timestep_embedding(const void* t, const float* freqs, void* out, int B, int half, float time_factor, void* stream) {
  return fluxmi_k_timestep_embedding(t, freqs, out, B, half, time_factor, (hipStream_t)stream);
}
int fluxmi_clock_sample(void* out2_dev_u64, void* stream) {
  FLUXMI_REQUIRE(out2_dev_u64, "clock_sample: NULL argument");
  return fluxmi_k_clock_sample((unsigned long long*)out2_dev_u64, (hipStream_t)stream);
}
int fluxmi_euler(void* img, const void* pred, const float* dts, const int* step, long long n, void* stream) {
  return fluxmi_k_euler(img, pred, dts, step, n, (hipStream_t)stream);
}

}  // extern "C"
