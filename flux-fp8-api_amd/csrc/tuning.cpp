// fluxmi -- kernel-selection knobs, resolved ONCE (include/fluxmi.h, fluxmi_tuning_t).
//
// Every FLUXMI_* environment variable that chooses a kernel or a fusion level is read here, at the first call that needs a knob, into one
// process-wide struct; nothing else in the library calls getenv.  fluxmi_set_tuning replaces the struct at run time (A/B probes, the op
// tests) and bumps a generation counter: an engine remembers the generation its step graph was captured under and re-captures when it
// changed, so a replayed graph never runs with choices other than the ones in force.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "fluxmi_internal.h"

namespace {

std::mutex g_mu;
fluxmi_tuning_t g_tuning;
std::atomic<unsigned> g_generation{0};
std::once_flag g_once;

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

int validate(const fluxmi_tuning_t& t) {
  FLUXMI_REQUIRE(t.struct_size == (int)sizeof(fluxmi_tuning_t), "tuning: struct_size %d != %d (ABI mismatch)", t.struct_size, (int)sizeof(fluxmi_tuning_t));
  FLUXMI_REQUIRE(t.gemm_cfg >= -1 && t.gemm_cfg <= 200, "tuning: gemm_cfg %d out of range", t.gemm_cfg);
  // P is bounded by 2^defer_log2, O and l carry the same factor: beyond 2^16 the fp32 sums of 4608 keys lose their headroom, and a
  // negative or non-finite threshold makes the rescale test meaningless (inf / NaN images with no error otherwise)
  FLUXMI_REQUIRE(isfinite(t.attn_defer_log2) && t.attn_defer_log2 >= 0.f && t.attn_defer_log2 <= 16.f,
                 "tuning: attn_defer_log2 %g outside [0, 16]", (double)t.attn_defer_log2);
  FLUXMI_REQUIRE(t.fuse_kv >= 0 && t.fuse_kv <= 2, "tuning: fuse_kv %d (0..2)", t.fuse_kv);
  FLUXMI_REQUIRE(t.ln_variant >= 1 && t.ln_variant <= 3, "tuning: ln_variant %d (1 = wave per row, 2 = streaming, 3 = streaming, two workgroups per CU)", t.ln_variant);
  // the remaining switches: a value outside the documented set would be read as "on" by some call sites and as a variant number by others
  struct { const char* name; int v, lo, hi; } sw[] = {
      {"gemm_splitk", t.gemm_splitk, 0, 1}, {"gemm_hybrid", t.gemm_hybrid, 0, 1}, {"gemm_esel", t.gemm_esel, 0, 1}, {"gemm_persist", t.gemm_persist, 0, 2},
      {"attn_var", t.attn_var, 0, 3},       {"attn_abl", t.attn_abl, 0, 15},      {"attn_f16k", t.attn_f16k, 0, 1}, {"qlut", t.qlut, 0, 1},
      {"roctx", t.roctx, 0, 1},             {"prefetch", t.prefetch, 0, 3},       {"w_pairs", t.w_pairs, 0, 1},     {"log", t.log, 0, 1},
      {"attn_split", t.attn_split, 0, 2}, {"gemm_tile192", t.gemm_tile192, 0, 2}, {"a_pairs", t.a_pairs, 0, 1}};
  for (const auto& k : sw) FLUXMI_REQUIRE(k.v >= k.lo && k.v <= k.hi, "tuning: %s %d outside [%d, %d]", k.name, k.v, k.lo, k.hi);
  return 0;
}

void log_tuning(const fluxmi_tuning_t& t, const char* why) {
  fprintf(stderr,
          "fluxmi tuning (%s): gemm_cfg=%d splitk=%d hybrid=%d esel=%d persist=%d | attn var=%d abl=%d defer_log2=%g f16k=%d | "
          "fuse_kv=%d qlut=%d ln=%d roctx=%d prefetch=%d w_pairs=%d attn_split=%d tile192=%d a_pairs=%d\n",
          why, t.gemm_cfg, t.gemm_splitk, t.gemm_hybrid, t.gemm_esel, t.gemm_persist, t.attn_var, t.attn_abl,
          (double)t.attn_defer_log2, t.attn_f16k, t.fuse_kv, t.qlut, t.ln_variant, t.roctx, t.prefetch, t.w_pairs, t.attn_split, t.gemm_tile192, t.a_pairs);
}

void init_from_env() {
  fluxmi_tuning_t t;
  memset(&t, 0, sizeof(t));
  t.struct_size = (int)sizeof(t);
  t.gemm_cfg = env_int("FLUXMI_GEMM_CFG", -1);
  t.gemm_splitk = env_int("FLUXMI_GEMM_SPLITK", 1);
  t.gemm_hybrid = env_int("FLUXMI_GEMM_HYBRID", 1);
  t.gemm_esel = env_int("FLUXMI_GEMM_ESEL", 1);
  t.gemm_persist = env_int("FLUXMI_GEMM_PERSIST", 1);
  t.prefetch = env_int("FLUXMI_PREFETCH", 1);
  t.w_pairs = env_int("FLUXMI_W_PAIRS", 1);
  t.attn_var = env_int("FLUXMI_ATTN_VAR", 0);
  t.attn_abl = env_int("FLUXMI_ATTN_ABL", 0);
  {
    const char* e = getenv("FLUXMI_ATTN_THR");
    t.attn_defer_log2 = (e && *e) ? (float)atof(e) : 8.0f;
  }
  t.attn_f16k = env_int("FLUXMI_ATTN_F16K", 1);
  t.fuse_kv = env_int("FLUXMI_FUSE_KV", 2);
  t.qlut = env_int("FLUXMI_QLUT", 1);
  t.ln_variant = env_int("FLUXMI_LN_V", 2);
  t.roctx = env_int("FLUXMI_ROCTX", 0);
  t.log = env_int("FLUXMI_LOG", 0);
  t.attn_split = env_int("FLUXMI_ATTN_SPLIT", 1);
  t.gemm_tile192 = env_int("FLUXMI_GEMM_TILE192", 1);
  t.a_pairs = env_int("FLUXMI_A_PAIRS", 1);
  if (validate(t) != 0) {  // a bad environment must not silently change the arithmetic: say so and keep the compiled defaults for that knob
    fprintf(stderr, "fluxmi: ignoring invalid FLUXMI_* environment (%s)\n", fluxmi_last_error());
    if (!(isfinite(t.attn_defer_log2) && t.attn_defer_log2 >= 0.f && t.attn_defer_log2 <= 16.f)) t.attn_defer_log2 = 8.0f;
    if (t.fuse_kv < 0 || t.fuse_kv > 2) t.fuse_kv = 2;
    if (t.ln_variant < 1 || t.ln_variant > 3) t.ln_variant = 2;
    if (t.gemm_cfg < -1 || t.gemm_cfg > 200) t.gemm_cfg = -1;
    auto fix = [](int& v, int lo, int hi, int dflt) { if (v < lo || v > hi) v = dflt; };
    fix(t.gemm_splitk, 0, 1, 1); fix(t.gemm_hybrid, 0, 1, 1); fix(t.gemm_esel, 0, 1, 1); fix(t.gemm_persist, 0, 2, 1);
    fix(t.attn_var, 0, 3, 0); fix(t.attn_abl, 0, 15, 0); fix(t.attn_f16k, 0, 1, 1); fix(t.qlut, 0, 1, 1); fix(t.roctx, 0, 1, 0);
    fix(t.prefetch, 0, 3, 1); fix(t.w_pairs, 0, 1, 1); fix(t.log, 0, 1, 0); fix(t.attn_split, 0, 2, 1); fix(t.gemm_tile192, 0, 2, 1); fix(t.a_pairs, 0, 1, 1);
  }
  g_tuning = t;
  if (t.log) log_tuning(t, "environment");
}

}  // namespace

static thread_local FluxmiPrefetch g_pending_pf = {};
void fluxmi_set_prefetch(const FluxmiPrefetch* pf) {
  if (pf) g_pending_pf = *pf; else memset(&g_pending_pf, 0, sizeof(g_pending_pf));
}
FluxmiPrefetch fluxmi_take_prefetch() {
  FluxmiPrefetch r = g_pending_pf;
  memset(&g_pending_pf, 0, sizeof(g_pending_pf));
  return r;
}

fluxmi_tuning_t fluxmi_tuning() {
  std::call_once(g_once, init_from_env);
  std::lock_guard<std::mutex> lk(g_mu);
  return g_tuning;
}
unsigned fluxmi_tuning_generation() { return g_generation.load(); }
void fluxmi_log_tuning(const char* why) {
  const fluxmi_tuning_t t = fluxmi_tuning();
  if (t.log) log_tuning(t, why);
}

extern "C" {

int fluxmi_get_tuning(fluxmi_tuning_t* out) {
  FLUXMI_REQUIRE(out, "get_tuning: NULL argument");
  *out = fluxmi_tuning();
  return 0;
}

int fluxmi_set_tuning(const fluxmi_tuning_t* in) {
  FLUXMI_REQUIRE(in, "set_tuning: NULL argument");
  FLUXMI_TRY(validate(*in));
  std::call_once(g_once, init_from_env);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_tuning = *in;
  }
  g_generation.fetch_add(1);
  if (in->log) log_tuning(*in, "set_tuning");
  return 0;
}

}  // extern "C"
