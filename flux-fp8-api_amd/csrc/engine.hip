// fluxmi -- the Flux denoise engine: Flux.forward (reference modules/flux_model.py:672-716) and the Euler
// loop of FluxPipeline.generate (flux_pipeline.py:619-651) sequenced natively over the HIP kernels.
//
// Two execution modes share every kernel:
//   unfused (mode 0 calibrating / mode 2 frozen): producer -> bf16 -> [amax -> scale update] -> quantise -> GEMM,
//       i.e. the reference's eager op order, advancing F8Linear's 12-trial input-scale state machine
//       (float8_quantize.py:220-246) with device-resident amax/scale (no host sync);
//   fused (mode 1, frozen scales): LN+modulate+quantise, GEMM epilogues (GELU+quantise, gate*y+x, qkv|mlp split),
//       attention writing fp8 directly; ~8 launches per double block, 5 per single block, hipGraph-captured.
// Because every fused kernel re-applies the reference's bf16 rounding points, both modes produce the same bits
// for the same scales (tests/test_engine.py checks this on the GPU).
//
// HBM layout (per request shape B, Li, Lt; L = Lt + Li, txt rows first so torch.cat is free):
//   x      bf16 [B, L, H]        residual stream (img = rows Lt.., txt = rows ..Lt)
//   a8     fp8  [B, L, H]        quantised LN+modulate output (GEMM A operand)
//   qkv    bf16 [B, L, 3H]       qkv GEMM output
//   Q,K    bf16 [B, heads, L, 128];  VT bf16 [B, heads, 128, Lp]  (attention operands)
//   attn8  fp8  [B, L, H];  h8 fp8 [B, L, 4H];  cat8 fp8 [B, L, 5H]  (single block: attn | gelu(mlp))
//   mod    bf16 [B, 12H*depth + 3H*single + 2H]   all modulation vectors of the step
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "fluxmi_internal.h"

namespace {
constexpr int MAX_STEPS = 1024;
// samples per engine pass: bounds the workspace (1.1 GB per 1024x1024 sample) and the step-ahead modulation table (2.1 MB per step and
// sample); the host wrapper runs larger batches as equal consecutive passes
constexpr int FLUXMI_ENGINE_MAX_BATCH = 32;
struct Buf { void* p; size_t n; };
}  // namespace

struct fluxmi_engine {
  fluxmi_model_desc_t d;
  std::vector<fluxmi_linear_t> lin;
  std::vector<const void*> norm;
  int i_img_in, i_time_in, i_vec_in, i_guid_in, i_txt_in, i_double0, i_single0, i_final_mod, i_final_lin;
  int B = 0, Li = 0, Lt = 0, L = 0, Lp = 0;
  long long mod_cols = 0;
  char* ws = nullptr;
  size_t ws_bytes = 0;
  std::map<std::string, Buf> bufs;
  // persistent small device state
  char* consts = nullptr;
  float *d_freqs, *d_omega, *d_ts, *d_dts, *d_amax, *d_amax_own;
  int *d_axis, *d_step;
  FluxmiCalibLayer* d_calib;
  FluxmiGemvLayer* d_gemv = nullptr;
  std::vector<FluxmiGemvLayer> h_gemv;
  std::vector<FluxmiCalibLayer> h_calib_mod;  // modulation layers that share silu(vec)
  FluxmiCalibLayer* d_calib_mod = nullptr;
  int gemv_blocks = 0, gemv_maxK = 0;
  // step-ahead modulation table (frozen scales): every modulation vector of every remaining step of a request, computed
  // before the loop in batches of 8 steps so that the 3.2 GB of modulation weights are streamed once per 8 steps, not per step
  char* mods_all = nullptr;
  size_t mods_all_bytes = 0;
  int mods_step0 = 0;
  bool mods_table = false;  // the step being sequenced takes its modulations from the table
  bool qlut_valid = false;  // the quantising-epilogue tables reflect the current input scales
  hipGraphExec_t exec = nullptr;
  bool graph_ok = false;
  unsigned graph_gen = 0;          // fluxmi_tuning_generation() the step graph was captured under: a changed tuning struct re-captures
  bool warmed = false;             // one frozen step of this shape has run eagerly (lazy one-time inits done): later calls may capture at once
  bool txt_emb_valid = false;
  // row-pair copies of the F8Linear weights the persistent GEMM launches read (fluxmi_gemm_group_t.W_pairs): one allocation, offsets per linear
  // (-1 = none); rebuilt on the first launch after create / rebind (the weights may have been rewritten: LoRA fuse)
  char* pairs = nullptr;
  size_t pairs_bytes = 0;
  std::vector<long long> pairs_off;
  bool pairs_dirty = true;
  bool pairs_skipped = false;      // the copies were wanted and did not fit (ensure_pairs): retried at the next prepare
  unsigned pairs_gen = 0;          // fluxmi_tuning_generation() the copies were built under (fluxmi_tuning_t.w_pairs may have changed)
  int* d_step0 = nullptr;          // first step of the modulation table (device scalar: the captured graph reads it)
  int mods_rows_cap = 0;           // rows (steps x B) the table holds; sized in engine_prepare, never inside denoise
  // pinned host staging for the per-request schedule (ts | dts), guarded by an event so that engine_denoise never waits on the stream
  float* h_sched = nullptr;
  hipEvent_t ev_sched = nullptr;
  bool sched_pending = false;
  // hipEvent timing of the frozen (graph-replayed) part of the last denoise call, read back by fluxmi_engine_last_timing
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
  int timed_steps = 0;
  // multi-GPU calibration: caller-owned amax array + host hook called between the amax reduction of a layer and its scale update
  float* amax_ext = nullptr;
  fluxmi_amax_hook_t amax_hook = nullptr;
  void* amax_user = nullptr;
};

namespace {

typedef fluxmi_engine E;

template <class T> T* buf(E* e, const char* name) {
  auto it = e->bufs.find(name);
  return it == e->bufs.end() ? nullptr : (T*)it->second.p;
}

static u16 host_f2bf(double v) {
  float f = (float)v;
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

// fluxmi_tuning_t.fuse_kv (FLUXMI_FUSE_KV): 0 = K / V^T by the relayout kernel, 1 = V^T from the qkv GEMM epilogue, 2 (default) = K and V^T
// from the epilogue: no relayout launch at all.  Round 1 (profiles/r01_fuse_kv_ab.txt: 52.23 / 51.50 / 51.70 ms per step) had K's
// norm + RoPE behind guarded stores with the pe load inside each guard; the persistent kernel's K path (gemm_persist.hip) prefetches pe,
// normalises on the accumulators and stores unguarded (profiles/r04_fused_k.txt).
int fuse_kv_level() { return fluxmi_tuning().fuse_kv; }

// fluxmi_tuning_t.attn_f16k (FLUXMI_ATTN_F16K, default 1): K is stored as fp16 (by the relayout kernel or the fused-K GEMM epilogue) and attention runs the
// folded arithmetic (softmax scale in Q, running max in the accumulator init; include/fluxmi.h, fluxmi_attention).  0 = bf16 K, the unfolded kernel.
int attn_f16k() { return fluxmi_tuning().attn_f16k; }

// weight prefetch riding on launches with idle CUs (fluxmi_internal.h, FluxmiPrefetch): the fp8 weights of up to six linears, for the next
// launch that supports it; `wgs` = the CUs that launch leaves idle in its last round
void set_pf(fluxmi_engine* e, std::initializer_list<int> lins, int wgs);
const void* pairs_of(fluxmi_engine* e, int li);

int lin_count(const fluxmi_model_desc_t& d) { return 6 + (d.guidance_embed ? 2 : 0) + d.depth * 10 + d.depth_single * 3 + 2; }

// double-block linear slots / single-block linear slots
enum { D_IMG_MOD = 0, D_IMG_QKV, D_IMG_PROJ, D_IMG_MLP0, D_IMG_MLP2, D_TXT_MOD, D_TXT_QKV, D_TXT_PROJ, D_TXT_MLP0, D_TXT_MLP2 };
enum { S_MOD = 0, S_LIN1, S_LIN2 };

const fluxmi_linear_t& DL(E* e, int blk, int slot) { return e->lin[e->i_double0 + blk * 10 + slot]; }
const fluxmi_linear_t& SL(E* e, int blk, int slot) { return e->lin[e->i_single0 + blk * 3 + slot]; }
int DLi(E* e, int blk, int slot) { return e->i_double0 + blk * 10 + slot; }
int SLi(E* e, int blk, int slot) { return e->i_single0 + blk * 3 + slot; }

void set_pf(fluxmi_engine* e, std::initializer_list<int> lins, int wgs) {
  FluxmiPrefetch pf;
  memset(&pf, 0, sizeof(pf));
  if (fluxmi_tuning().prefetch && wgs > 0)
    for (int li : lins) {
      if (li < 0 || li >= (int)e->lin.size() || pf.n >= 6) continue;
      const fluxmi_linear_t& l = e->lin[li];
      if (!l.kind || !l.weight) continue;  // fp8 weights only (N * K bytes)
      const void* wp = pairs_of(e, li);  // what the launch will actually read
      pf.ptr[pf.n] = wp ? wp : l.weight;
      pf.bytes[pf.n] = ((long long)l.N * l.K) & ~15LL;
      ++pf.n;
    }
  pf.wgs = wgs;
  fluxmi_set_prefetch(pf.n ? &pf : nullptr);
}
int idle_cus(long long wgs) { return (int)((256 - wgs % 256) % 256); }

// the row-pair copy of linear li's weight, or nullptr
const void* pairs_of(E* e, int li) {
  if (!e->pairs || e->pairs_dirty || !fluxmi_tuning().w_pairs || li < 0 || li >= (int)e->pairs_off.size() || e->pairs_off[li] < 0) return nullptr;
  return e->pairs + e->pairs_off[li];
}
// ACTIVATIONS in the row-pair layout (round 6; fluxmi_gemm_group_t.a_pairs / c8_pairs, fluxmi_tuning_t.a_pairs): in FUSED mode the fp8
// activation buffers a8 / attn8 / h8 / cat8 -- written by LayerNorm, attention and the quantising GEMM epilogues, read as the A operand of
// the block linears -- keep the 64-byte K-steps of rows 2r and 2r + 1 in one 128-byte line, so an A panel's lines cross L2 -> CU once per
// tile instead of twice (what W_pairs does for the weights).  Needs every row offset of a group to be even: L and Lt even, hidden % 64 == 0.
// The unfused / calibrating modes keep plain rows (their producers are the standalone quantise kernels).
bool act_pairs(const E* e, bool fused) {
  return fused && fluxmi_tuning().a_pairs && e->L % 2 == 0 && e->Lt % 2 == 0 && e->d.hidden % 64 == 0 && e->d.mlp_hidden % 64 == 0;
}
// Linears whose launches go through the kernels that honour W_pairs at Flux geometry: the persistent kernel (double blocks' qkv and mlp.0,
// single blocks' linear1) and the one-wave-per-SIMD kernel (mlp.2, linear2).  +8 GB at Flux-dev.  Built lazily on the caller's stream:
// create / rebind have none, and a rebind follows weight surgery.
int ensure_pairs(E* e, hipStream_t s) {
  if (!e->pairs_dirty && e->pairs_gen == fluxmi_tuning_generation()) return 0;
  e->pairs_gen = fluxmi_tuning_generation();
  const int n = (int)e->lin.size();
  std::vector<long long> off(n, -1);
  size_t total = 0;
  if (fluxmi_tuning().w_pairs) {
    auto want = [&](int li) {
      const fluxmi_linear_t& l = e->lin[li];
      // fp8 weights only.  (The kernels read row-pair copies of BF16 weights as well since round 6 -- tests/test_ops_gpu.py::
      // test_gemm_bf16_row_pair_weights -- but for the bf16 flow they measured +1 %: Flux-schnell 256^2, 15.06 / 14.99 ms per step with copies,
      // 15.21 / 15.17 without, for 23.8 GB of copies; its M = 512 launches are bound by the A side and by latency)
      const size_t eb = 1;
      if (l.kind != 1 || !l.weight || l.N % 2 || (l.K * eb) % 64) return;
      off[li] = (long long)total;
      total += ((size_t)l.N * l.K * eb + 255) & ~(size_t)255;
    };
    // (proj: the ping-pong kernel, tile config 13, honours W_pairs since round 6, but copies of the proj weights measured nothing in the step --
    // 38.71 / 38.74 / 38.76 ms with, 38.71 / 38.75 / 38.71 without, profiles/r06_act_pairs.txt -- so they are not made)
    for (int i = 0; i < e->d.depth; ++i)
      for (int sl : {D_IMG_QKV, D_TXT_QKV, D_IMG_MLP0, D_TXT_MLP0, D_IMG_MLP2, D_TXT_MLP2}) want(DLi(e, i, sl));
    for (int i = 0; i < e->d.depth_single; ++i) { want(SLi(e, i, S_LIN1)); want(SLi(e, i, S_LIN2)); }
  }
  if (total == 0 && e->pairs) {  // switched off (fluxmi_tuning_t.w_pairs = 0): give the memory back
    FLUXMI_CHECK_HIP(hipStreamSynchronize(s));
    hipFree(e->pairs);
    e->pairs = nullptr; e->pairs_bytes = 0;
  }
  if (total > e->pairs_bytes) {
    if (e->pairs) hipFree(e->pairs);
    e->pairs = nullptr; e->pairs_bytes = 0;
    // The copies are an optimisation (same results without them): they must not take the memory the caller still needs -- the torch
    // allocator's next block, the VAE's 5.9 GB of patch matrices -- so they are only made while at least as much again (and 4 GiB) stays free
    size_t free_b = 0, total_b = 0;
    const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= 2 * total + ((size_t)4 << 30);
    if (!room || hipMalloc((void**)&e->pairs, total) != hipSuccess) {
      // no room for the copies: run without them (same results, ~3 % slower steps) -- said once per engine, and tried again at the next
      // fluxmi_engine_prepare (the caller's allocator may have given memory back by then); hipMemGetInfo does not see what the torch
      // allocator holds cached, so this depends on the allocator's state at the first launch
      (void)hipGetLastError();
      if (!e->pairs_skipped)
        fprintf(stderr, "fluxmi: row-pair weight copies skipped (%.1f GB wanted, %.1f GB free of %.1f): the GEMMs read the plain weights\n",
                total / 1e9, free_b / 1e9, total_b / 1e9);
      e->pairs_skipped = true;
      e->pairs_off.assign(n, -1);
      e->pairs_dirty = false;
      return 0;
    }
    e->pairs_skipped = false;
    e->pairs_bytes = total;
  }
  for (int li = 0; li < n; ++li)
    if (off[li] >= 0) FLUXMI_TRY(fluxmi_k_pair_rows(e->lin[li].weight, e->pairs + off[li], e->lin[li].N, (long long)e->lin[li].K * (e->lin[li].kind == 1 ? 1 : 2), s));
  e->pairs_off = off;
  e->pairs_dirty = false;
  return 0;
}

FluxmiGemmGroup mk_group(const fluxmi_linear_t& l, const void* A, long long lda, void* C, long long ldc, int M) {
  FluxmiGemmGroup g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.W = l.weight; g.bias = l.bias;
  g.sa_recip = l.kind ? l.in_scale_recip : nullptr;
  g.sb_recip = l.kind ? l.w_scale_recip : nullptr;
  g.C = C; g.lda = lda; g.ldc = ldc; g.M = M;
  return g;
}

int run_gemm(std::vector<FluxmiGemmGroup>& gs, int N, int K, int is_fp8, int act_fmt, int epi, hipStream_t s) {
  return fluxmi_gemm_dispatch(gs.data(), (int)gs.size(), N, K, is_fp8, act_fmt, epi, s);
}

// The bf16-operand linears around the blocks of an fp8 model (img_in, txt_in, final_layer.linear): ONE tile config whatever the batch size.  The
// fp8 tile configs give the same bits (one MFMA shape, one K order), the bf16 ones do not: configs 2 / 15 and 13 / 16 associate the K sum
// differently (~3e-4 of the outputs differ in their last bit, tools/probes/bf16_cfg_bits_probe.py), and the automatic choice follows the row
// count -- through img_in a sample's latents depended on the batch it rode in from B = 4 on (B x 4096 rows: config 13 instead of 2;
// tests/test_engine_gpu.py::test_maximum_batch_at_real_width).  Config 2 (128x128 tiles; 15 = its 128x64 form) is the choice at B = 1 for all
// three and costs nothing at larger B (K = 64 / one launch per request / N = 64: output-bound launches of tens of microseconds).
// Round 6: every bf16 tile config sums K in one order now (gemm.hip) and the split-K slices follow ONE sample's groups (api.cpp,
// fluxmi_gemm_set_batch), so the bf16 FLOW is batch-invariant as well; the pin stays (it is the B = 1 choice and keeps these three launches
// off the split-K path whatever the tuning says).
int run_gemm_fixed_cfg(std::vector<FluxmiGemmGroup>& gs, int N, int K, int is_fp8, int act_fmt, int epi, hipStream_t s) {
  const int cfg = fluxmi_gemm_tile_ok(N, K, 0, 2) ? 2 : (fluxmi_gemm_tile_ok(N, K, 0, 15) ? 15 : -1);
  if (is_fp8 || cfg < 0 || fluxmi_tuning().gemm_cfg >= 0) return run_gemm(gs, N, K, is_fp8, act_fmt, epi, s);
  for (size_t off = 0; off < gs.size(); off += FLUXMI_MAX_GROUPS) {
    FluxmiGemmParams p;
    memset(&p, 0, sizeof(p));
    p.n_groups = (int)std::min<size_t>(FLUXMI_MAX_GROUPS, gs.size() - off);
    for (int i = 0; i < p.n_groups; ++i) p.g[i] = gs[off + i];
    p.N = N; p.K = K; p.epi = epi;
    FLUXMI_TRY(fluxmi_launch_gemm(p, 0, act_fmt, cfg, s));
  }
  return 0;
}

// ---- calibration helpers (unfused path) ------------------------------------------------------------
int calib_begin(E* e, int li, hipStream_t s) { return hipMemsetAsync(e->d_amax + li, 0, sizeof(float), s) == hipSuccess ? 0 : 2; }
int calib_amax(E* e, int li, const void* x, int rows, int cols, long long ld, hipStream_t s) {
  return fluxmi_k_amax(x, e->d_amax + li, rows, cols, ld, s);
}
int calib_commit(E* e, int li, int trial, hipStream_t s) {
  const fluxmi_linear_t& l = e->lin[li];
  // float8_quantize.py:227 takes the max over the WHOLE batch: batch-sharded ranks exchange it here (all-reduce MAX of one float)
  if (e->amax_hook) FLUXMI_REQUIRE(e->amax_hook(e->amax_user, li, 1, (void*)s) == 0, "amax exchange hook failed (layer %d)", li);
  const float mx = l.in_fmt == FLUXMI_E5M2 ? 57344.f : 448.f;
  return fluxmi_k_calib_update(e->d_amax + li, l.amax_trials, l.in_scale, l.in_scale_recip, trial, e->d.num_trials, mx, s);
}

// Quantise the bf16 input of linear `li` (rows given as nb blocks of `rows` rows, block stride bstride) into dst8.
int stage_input(E* e, int li, bool calib, int trial, const u16* src, long long ld_src, long long src_bstride, uint8_t* dst,
                long long ld_dst, long long dst_bstride, int nb, int rows, int cols, hipStream_t s) {
  const fluxmi_linear_t& l = e->lin[li];
  if (!l.kind) return 0;
  if (calib) {
    FLUXMI_TRY(calib_begin(e, li, s));
    for (int b = 0; b < nb; ++b) FLUXMI_TRY(calib_amax(e, li, src + b * src_bstride, rows, cols, ld_src, s));
    FLUXMI_TRY(calib_commit(e, li, trial, s));
  }
  for (int b = 0; b < nb; ++b)
    FLUXMI_TRY(fluxmi_k_quantize_act(src + b * src_bstride, dst + b * dst_bstride, l.in_scale, rows, cols, ld_src, ld_dst, l.in_fmt, s));
  return 0;
}

// skinny linear through the GEMV kernel (M = B).  In calibrating mode the (possibly SiLU'd) input is materialised first.
int small_linear(E* e, int li, const u16* x, long long ldx, u16* out, long long ld_out, int pre_silu, bool calib, int trial,
                 u16* scratch, hipStream_t s) {
  const fluxmi_linear_t& l = e->lin[li];
  const u16* xin = x;
  if (l.kind && calib) {
    if (pre_silu) {
      FLUXMI_TRY(fluxmi_k_act(x, scratch, e->B, l.K, ldx, l.K, 1, s));
      xin = scratch; ldx = l.K; pre_silu = 0;
    }
    FLUXMI_TRY(calib_begin(e, li, s));
    FLUXMI_TRY(calib_amax(e, li, xin, e->B, l.K, ldx, s));
    FLUXMI_TRY(calib_commit(e, li, trial, s));
  }
  FluxmiGemvLayer g;
  memset(&g, 0, sizeof(g));
  g.W = l.weight; g.bias = l.bias; g.in_scale = l.in_scale; g.sa_recip = l.in_scale_recip; g.sb_recip = l.w_scale_recip;
  g.out = out; g.x = xin; g.ld_out = ld_out; g.ldx = ldx; g.N = l.N; g.K = l.K; g.w_fp8 = l.kind; g.pre_silu = pre_silu;
  g.act_fmt = l.in_fmt;
  // the GEMV kernel stages at most 8 activation rows in LDS: larger batches go in row chunks (same per-row arithmetic)
  for (int r0 = 0; r0 < e->B; r0 += 8) FLUXMI_TRY(fluxmi_launch_gemv(nullptr, &g, 1, std::min(8, e->B - r0), 0, 0, s, r0));
  return 0;
}

int build_gemv_table(E* e, hipStream_t s) {
  const int H = e->d.hidden;
  e->h_gemv.clear();
  e->h_calib_mod.clear();
  u16* mod = buf<u16>(e, "mod");
  u16* svec = buf<u16>(e, "svec");
  auto add = [&](int li, long long off) {
    const fluxmi_linear_t& l = e->lin[li];
    FluxmiGemvLayer g;
    memset(&g, 0, sizeof(g));
    g.W = l.weight; g.bias = l.bias; g.in_scale = l.in_scale; g.sa_recip = l.in_scale_recip; g.sb_recip = l.w_scale_recip;
    g.out = mod + off; g.x = svec; g.ld_out = e->mod_cols; g.ldx = H; g.N = l.N; g.K = l.K; g.w_fp8 = l.kind;
    g.pre_silu = 0; g.act_fmt = l.in_fmt;
    e->h_gemv.push_back(g);
    if (l.kind) e->h_calib_mod.push_back(FluxmiCalibLayer{l.amax_trials, l.in_scale, l.in_scale_recip});
  };
  for (int i = 0; i < e->d.depth; ++i) {
    add(DLi(e, i, D_IMG_MOD), (long long)i * 12 * H);
    add(DLi(e, i, D_TXT_MOD), (long long)i * 12 * H + 6 * H);
  }
  for (int i = 0; i < e->d.depth_single; ++i) add(SLi(e, i, S_MOD), (long long)e->d.depth * 12 * H + (long long)i * 3 * H);
  add(e->i_final_mod, (long long)e->d.depth * 12 * H + (long long)e->d.depth_single * 3 * H);
  int blk = 0, maxK = 0;
  for (auto& g : e->h_gemv) {
    g.blk_start = blk;
    blk += (g.N + 63) / 64;
    maxK = g.K > maxK ? g.K : maxK;
  }
  e->gemv_blocks = blk; e->gemv_maxK = maxK;
  FLUXMI_CHECK_HIP(hipMemcpyAsync(e->d_gemv, e->h_gemv.data(), sizeof(FluxmiGemvLayer) * e->h_gemv.size(), hipMemcpyHostToDevice, s));
  if (!e->h_calib_mod.empty())
    FLUXMI_CHECK_HIP(hipMemcpyAsync(e->d_calib_mod, e->h_calib_mod.data(), sizeof(FluxmiCalibLayer) * e->h_calib_mod.size(), hipMemcpyHostToDevice, s));
  FLUXMI_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Every Modulation.lin + LastLayer.adaLN_modulation of the model applied to R rows of silu(vec) (row stride H) -> out (row stride
// mod_cols): grouped MFMA GEMMs, up to 16 layers per launch, M = R padded to one tile.  Weight-stream bound (3.2 GB fp8 at
// Flux-dev).  Used both per step (R = B) and for the step-ahead table (R = steps x B): an output element's K-order of
// accumulation does not depend on M or on the tile shape, so the two are bit-identical.        flux_model.py:251-257,499-500
// a8: scratch for the per-layer quantised activations, FLUXMI_MAX_GROUPS slices of a8_stride bytes.
// ---------------------------------------------------------------------------------------------------------
int mods_gemm(E* e, const u16* sv, int R, u16* out, uint8_t* a8, size_t a8_stride, hipStream_t s) {
  const int H = e->d.hidden;
  const long long MC = e->mod_cols;
  u16* mod0 = buf<u16>(e, "mod");
  const size_t n = e->h_gemv.size();
  size_t i = 0;
  while (i < n) {
    const FluxmiGemvLayer& g0 = e->h_gemv[i];
    std::vector<FluxmiGemmGroup> gs;
    size_t j = i;
    for (; j < n && gs.size() < FLUXMI_MAX_GROUPS; ++j) {
      const FluxmiGemvLayer& g = e->h_gemv[j];
      if (g.N != g0.N || g.K != g0.K || g.w_fp8 != g0.w_fp8 || g.act_fmt != g0.act_fmt) break;
      FluxmiGemmGroup gg;
      memset(&gg, 0, sizeof(gg));
      if (g.w_fp8) {
        uint8_t* aq = a8 + gs.size() * a8_stride;
        FLUXMI_TRY(fluxmi_k_quantize_act(sv, aq, g.in_scale, R, g.K, H, g.K, g.act_fmt, s));
        gg.A = aq; gg.sa_recip = g.sa_recip; gg.sb_recip = g.sb_recip;
      } else {
        gg.A = sv;
      }
      gg.W = g.W; gg.bias = g.bias; gg.lda = g.K;
      gg.C = out + ((u16*)g.out - mod0); gg.ldc = MC; gg.M = R;
      gs.push_back(gg);
    }
    fluxmi_gemm_block_splitk(1);  // row-count-independent K order: table rows == per-step rows, bit for bit
    const int rc = run_gemm(gs, g0.N, g0.K, g0.w_fp8, g0.act_fmt, FLUXMI_EPI_BF16, s);
    fluxmi_gemm_block_splitk(0);
    FLUXMI_TRY(rc);
    i = j;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// vec = time_in(temb(t)) + guidance_in(temb(g)) + vector_in(y)  and all modulations      flux_model.py:687-697, 251-257
// ---------------------------------------------------------------------------------------------------------
int compute_vec_and_mods(E* e, const u16* t_vec, const u16* g_vec, const u16* y, bool calib, int trial, hipStream_t s) {
  const int H = e->d.hidden, B = e->B;
  u16 *temb = buf<u16>(e, "temb"), *emb_h = buf<u16>(e, "emb_h"), *emb_s = buf<u16>(e, "emb_s");
  u16 *vec_t = buf<u16>(e, "vec_t"), *vec_g = buf<u16>(e, "vec_g"), *vec_y = buf<u16>(e, "vec_y");
  u16 *vec = buf<u16>(e, "vec"), *svec = buf<u16>(e, "svec");
  FLUXMI_TRY(fluxmi_k_timestep_embedding(t_vec, e->d_freqs, temb, B, 128, 1000.0f, s));
  FLUXMI_TRY(small_linear(e, e->i_time_in, temb, 256, emb_h, H, 0, calib, trial, emb_s, s));
  FLUXMI_TRY(small_linear(e, e->i_time_in + 1, emb_h, H, vec_t, H, 1, calib, trial, emb_s, s));
  const u16* acc = vec_t;
  if (e->d.guidance_embed) {
    FLUXMI_REQUIRE(g_vec, "Didn't get guidance strength for guidance distilled model.");
    FLUXMI_TRY(fluxmi_k_timestep_embedding(g_vec, e->d_freqs, temb, B, 128, 1000.0f, s));
    FLUXMI_TRY(small_linear(e, e->i_guid_in, temb, 256, emb_h, H, 0, calib, trial, emb_s, s));
    FLUXMI_TRY(small_linear(e, e->i_guid_in + 1, emb_h, H, vec_g, H, 1, calib, trial, emb_s, s));
    FLUXMI_TRY(fluxmi_k_add(vec_t, vec_g, vec, (long long)B * H, s));
    acc = vec;
  }
  FLUXMI_TRY(small_linear(e, e->i_vec_in, y, e->d.vec_in, emb_h, H, 0, calib, trial, emb_s, s));
  FLUXMI_TRY(small_linear(e, e->i_vec_in + 1, emb_h, H, vec_y, H, 1, calib, trial, emb_s, s));
  FLUXMI_TRY(fluxmi_k_add(acc, vec_y, vec, (long long)B * H, s));
  // silu(vec) feeds every Modulation.lin and LastLayer.adaLN_modulation
  FLUXMI_TRY(fluxmi_k_act(vec, svec, B, H, H, H, 1, s));
  if (calib && !e->h_calib_mod.empty()) {
    FLUXMI_CHECK_HIP(hipMemsetAsync(e->d_amax, 0, sizeof(float), s));
    FLUXMI_TRY(fluxmi_k_amax(svec, e->d_amax, B, H, H, s));
    if (e->amax_hook) FLUXMI_REQUIRE(e->amax_hook(e->amax_user, 0, 1, (void*)s) == 0, "amax exchange hook failed (modulations)");
    FLUXMI_TRY(fluxmi_k_calib_update_many(e->d_amax, e->d_calib_mod, (int)e->h_calib_mod.size(), trial, e->d.num_trials, 57344.f, s));
  }
  const size_t a8_stride = ((size_t)B * e->gemv_maxK + 255) & ~(size_t)255;
  return mods_gemm(e, svec, B, buf<u16>(e, "mod"), buf<uint8_t>(e, "mods_a8"), a8_stride, s);
}

int embed_txt(E* e, const u16* txt, bool calib, int trial, u16* dst, long long dst_bstride, hipStream_t s) {
  const int H = e->d.hidden, B = e->B, Lt = e->Lt, C = e->d.ctx_in;
  const fluxmi_linear_t& l = e->lin[e->i_txt_in];
  uint8_t* in8 = buf<uint8_t>(e, "in8");
  FLUXMI_TRY(stage_input(e, e->i_txt_in, calib, trial, txt, C, 0, in8, C, 0, 1, B * Lt, C, s));
  std::vector<FluxmiGemmGroup> gs;
  for (int b = 0; b < B; ++b)
    gs.push_back(mk_group(l, l.kind ? (const void*)(in8 + (long long)b * Lt * C) : (const void*)(txt + (long long)b * Lt * C), C,
                          dst + b * dst_bstride, H, Lt));
  return run_gemm_fixed_cfg(gs, H, C, l.kind, l.in_fmt, FLUXMI_EPI_BF16, s);
}

// ---------------------------------------------------------------------------------------------------------
// Step-ahead modulations (frozen scales only).  `vec` depends on the timestep, the guidance and y -- never on the latents -- so
// the rows of every remaining step r = (step - step0) * B + b are known before the loop:
//   vec[r] = (time_in(temb(t_step)) + guidance_in(temb(g))[b]) + vector_in(y)[b]            flux_model.py:687-697
//   mods[r] = Modulation.lin(silu(vec[r])) for the 76 modulation layers + LastLayer.adaLN    flux_model.py:251-257,499-500
// computed with the SAME kernels as the per-step path (each row's dot products do not depend on how many rows share a launch,
// so the table is bit-identical to what compute_vec_and_mods produces step by step), 8 rows per weight pass.
// ---------------------------------------------------------------------------------------------------------
// table geometry for `rows` rows: [table | t | temb | hidden | vec_t | vec | silu(vec) | per-group quantised activations]
constexpr int MODS_STEPS = 64;  // steps per table window; a longer request rebuilds the table every MODS_STEPS steps
size_t mods_table_bytes(E* e, size_t rows) {
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t H = e->d.hidden;
  return al(rows * e->mod_cols * 2) + al(rows * 2) + al(rows * 512) + 4 * al(rows * H * 2) + FLUXMI_MAX_GROUPS * al(rows * (size_t)e->gemv_maxK);
}
// rows [step0, step_end) x B of the table; the allocation was made by engine_prepare (nothing is allocated or synchronised here)
int precompute_mods(E* e, int step0, int step_end, const u16* g_vec, const u16* y, hipStream_t s) {
  const int H = e->d.hidden, B = e->B;
  const long long MC = e->mod_cols;
  const int R = (step_end - step0) * B;
  if (R <= 0) return 0;
  FLUXMI_REQUIRE(R <= e->mods_rows_cap && e->mods_all, "precompute_mods: %d rows exceed the table of %d rows", R, e->mods_rows_cap);
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t Rc = (size_t)e->mods_rows_cap;
  const size_t sz_mod = al(Rc * MC * 2), sz_t = al(Rc * 2), sz_temb = al(Rc * 512), sz_h = al(Rc * H * 2);
  char* base = e->mods_all;
  u16* mod_all = (u16*)base;
  u16* tv = (u16*)(base + sz_mod);
  u16* temb = (u16*)(base + sz_mod + sz_t);
  u16* hbuf = (u16*)(base + sz_mod + sz_t + sz_temb);
  u16* vt = (u16*)((char*)hbuf + sz_h);
  u16* vec = (u16*)((char*)vt + sz_h);
  u16* sv = (u16*)((char*)vec + sz_h);
  FLUXMI_TRY(fluxmi_k_timestep_rows(tv, e->d_ts, step0, B, R, s));  // bf16(t) per row, from the schedule already on the device
  FLUXMI_TRY(fluxmi_k_timestep_embedding(tv, e->d_freqs, temb, R, 128, 1000.0f, s));
  auto rows_linear = [&](int li, const u16* x, long long ldx, u16* out, int pre_silu, int r0, int nr) -> int {
    const fluxmi_linear_t& l = e->lin[li];
    FluxmiGemvLayer g;
    memset(&g, 0, sizeof(g));
    g.W = l.weight; g.bias = l.bias; g.in_scale = l.in_scale; g.sa_recip = l.in_scale_recip; g.sb_recip = l.w_scale_recip;
    g.out = out; g.x = x; g.ld_out = H; g.ldx = ldx; g.N = l.N; g.K = l.K; g.w_fp8 = l.kind; g.pre_silu = pre_silu;
    g.act_fmt = l.in_fmt;
    return fluxmi_launch_gemv(nullptr, &g, 1, nr, 0, 0, s, r0);
  };
  for (int r0 = 0; r0 < R; r0 += 8) {
    const int nr = std::min(8, R - r0);
    FLUXMI_TRY(rows_linear(e->i_time_in, temb, 256, hbuf, 0, r0, nr));
    FLUXMI_TRY(rows_linear(e->i_time_in + 1, hbuf, H, vt, 1, r0, nr));
  }
  // step-invariant parts, B rows (same launches as compute_vec_and_mods)
  u16 *temb_b = buf<u16>(e, "temb"), *emb_h = buf<u16>(e, "emb_h"), *emb_s = buf<u16>(e, "emb_s");
  u16 *vec_g = buf<u16>(e, "vec_g"), *vec_y = buf<u16>(e, "vec_y");
  const u16* acc = vt;
  if (e->d.guidance_embed) {
    FLUXMI_REQUIRE(g_vec, "Didn't get guidance strength for guidance distilled model.");
    FLUXMI_TRY(fluxmi_k_timestep_embedding(g_vec, e->d_freqs, temb_b, B, 128, 1000.0f, s));
    FLUXMI_TRY(small_linear(e, e->i_guid_in, temb_b, 256, emb_h, H, 0, false, 0, emb_s, s));
    FLUXMI_TRY(small_linear(e, e->i_guid_in + 1, emb_h, H, vec_g, H, 1, false, 0, emb_s, s));
    FLUXMI_TRY(fluxmi_k_add_bcast(vt, vec_g, vec, R, B, H, s));
    acc = vec;
  }
  FLUXMI_TRY(small_linear(e, e->i_vec_in, y, e->d.vec_in, emb_h, H, 0, false, 0, emb_s, s));
  FLUXMI_TRY(small_linear(e, e->i_vec_in + 1, emb_h, H, vec_y, H, 1, false, 0, emb_s, s));
  FLUXMI_TRY(fluxmi_k_add_bcast(acc, vec_y, vec, R, B, H, s));
  FLUXMI_TRY(fluxmi_k_act(vec, sv, R, H, H, H, 1, s));
  FLUXMI_TRY(mods_gemm(e, sv, R, mod_all, (uint8_t*)((char*)sv + sz_h), al(Rc * (size_t)e->gemv_maxK), s));
  e->mods_step0 = step0;
  FLUXMI_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)e->d_step0, step0, 1, s));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Quantising-epilogue tables (frozen scales): one 64 KiB bf16 -> fp8 table per GELU -> F8Linear hand-off (double-block mlp.0 -> mlp.2
// per stream, single-block linear1 -> linear2), see fluxmi_gemm_group_t.q_lut.  Rebuilt (77 tiny launches) whenever a fused
// sequence starts, so they always reflect the current input scales.  fluxmi_tuning_t.qlut = 0 turns them off.
// ---------------------------------------------------------------------------------------------------------
bool qlut_enabled() { return fluxmi_tuning().qlut != 0; }
int build_qluts(E* e, hipStream_t s) {
  uint8_t* lut = buf<uint8_t>(e, "qlut");
  if (!lut || !qlut_enabled() || e->qlut_valid) return 0;
  e->qlut_valid = true;
  for (int i = 0; i < e->d.depth; ++i)
    for (int st = 0; st < 2; ++st) {
      const fluxmi_linear_t& l = e->lin[DLi(e, i, st == 0 ? D_TXT_MLP2 : D_IMG_MLP2)];
      if (l.kind) FLUXMI_TRY(fluxmi_k_build_qlut(l.in_scale, l.in_fmt, 1, lut + (size_t)(i * 2 + st) * 65536, s));
    }
  for (int i = 0; i < e->d.depth_single; ++i) {
    const fluxmi_linear_t& l = SL(e, i, S_LIN2);
    if (l.kind) FLUXMI_TRY(fluxmi_k_build_qlut(l.in_scale, l.in_fmt, 1, lut + (size_t)(e->d.depth * 2 + i) * 65536, s));
  }
  return 0;
}

// Workspace pointers of one forward pass (looked up once).
struct Ctx {
  int H, Hm, B, L, Lt, Li, heads;
  long long XB, MC;
  u16 *x, *mod, *qkv, *K, *VT, *pe, *abf, *catbf, *hbf, *attnbf, *lin1;
  uint8_t *a8, *attn8, *h8, *cat8, *qlut;
};
Ctx make_ctx(E* e) {
  Ctx c;
  c.H = e->d.hidden; c.Hm = e->d.mlp_hidden; c.B = e->B; c.L = e->L; c.Lt = e->Lt; c.Li = e->Li; c.heads = e->d.heads;
  c.XB = (long long)c.L * c.H; c.MC = e->mod_cols;
  c.x = buf<u16>(e, "x"); c.mod = buf<u16>(e, "mod"); c.qkv = buf<u16>(e, "qkv"); c.K = buf<u16>(e, "K"); c.VT = buf<u16>(e, "VT");
  c.pe = buf<u16>(e, "pe"); c.abf = buf<u16>(e, "abf"); c.catbf = buf<u16>(e, "catbf"); c.hbf = buf<u16>(e, "hbf");
  c.attnbf = buf<u16>(e, "attnbf"); c.lin1 = buf<u16>(e, "lin1");
  c.a8 = buf<uint8_t>(e, "a8"); c.attn8 = buf<uint8_t>(e, "attn8"); c.h8 = buf<uint8_t>(e, "h8"); c.cat8 = buf<uint8_t>(e, "cat8");
  c.qlut = buf<uint8_t>(e, "qlut");
  return c;
}

// ---------------------------------------------------------------------------------------------------------
// DoubleStreamBlock.forward (flux_model.py:356-400) as eight stages; [s0, s1] selects a sub-range (fluxmi_engine_run_block: the
// teacher-forced parity tests overwrite a stage's input buffer with the oracle's tensor and run that stage alone).
//   0 LN + modulate (+quantise) -> a8 | 1 qkv GEMM -> qkv (+ V^T) | 2 K (and V^T) relayout | 3 attention -> attn8
//   4 proj + gate1*y + x -> x | 5 LN + modulate (+quantise) -> a8 | 6 mlp.0 + GELU (+quantise) -> h8 | 7 mlp.2 + gate2*y + x -> x
// ---------------------------------------------------------------------------------------------------------
constexpr int DOUBLE_STAGES = 8, SINGLE_STAGES = 5;
int double_block(E* e, const Ctx& c, int i, int mode, int trial, int s0, int s1, hipStream_t s) {
  const bool fused = mode == 1, calib = mode == 0;
  const int H = c.H, Hm = c.Hm, B = c.B, L = c.L, Lt = c.Lt, Li = c.Li, heads = c.heads;
  const long long XB = c.XB, MC = c.MC;
  u16 *x = c.x, *qkv = c.qkv, *K = c.K, *VT = c.VT, *pe = c.pe, *abf = c.abf, *hbf = c.hbf, *attnbf = c.attnbf;
  uint8_t *a8 = c.a8, *attn8 = c.attn8, *h8 = c.h8;
  auto on = [&](int st) { return st >= s0 && st <= s1; };
  const u16* mi = c.mod + (long long)i * 12 * H;  // img: shift1 scale1 gate1 shift2 scale2 gate2
  const u16* mt = mi + 6 * H;                     // txt
  const int li_q[2] = {DLi(e, i, D_TXT_QKV), DLi(e, i, D_IMG_QKV)};
  const int li_p[2] = {DLi(e, i, D_TXT_PROJ), DLi(e, i, D_IMG_PROJ)};
  const int li_m0[2] = {DLi(e, i, D_TXT_MLP0), DLi(e, i, D_IMG_MLP0)};
  const int li_m2[2] = {DLi(e, i, D_TXT_MLP2), DLi(e, i, D_IMG_MLP2)};
  const u16* mods[2] = {mt, mi};
  const int roff[2] = {0, Lt}, rows[2] = {Lt, Li};
  const void* const* ns = &e->norm[i * 4];  // img q, img k, txt q, txt k
  // V^T leaves the qkv GEMM's epilogue directly in the attention kernel's layout when the 256x256 kernels apply
  // (short sequences -- Flux-schnell 256x256: 72 tiles -- leave V^T to the relayout kernel: the fused output exists only in the 256x256
  // kernels, and a launch that small runs 1.7x faster on 128x128 tiles at two workgroups per CU, profiles/r03_small_m.txt).  The threshold
  // is on ONE sample's rows, not on B x L: the fused K of tile config 13 and the relayout kernel's K agree on 99.9 % of the elements, not on
  // all, so a choice that followed the batch made a sample's bits follow it (round 6: schnell 256^2 at B = 4 crossed 2048 rows)
  const bool fuse_v = fuse_kv_level() >= 1 && fluxmi_gemm_tile_ok(3 * H, H, e->lin[li_q[0]].kind, 13) && Lt % 16 == 0 && L >= 2048;
  const bool fuse_k = fuse_kv_level() >= 2 && fuse_v && H % 256 == 0;  // a 256-column tile must not straddle the q|k|v boundaries
  const int ap = act_pairs(e, fused) ? 1 : 0;  // fp8 activation buffers in the row-pair layout (see act_pairs)

  for (int half = 0; half < 2; ++half) {
    const int so = half * 3;  // offset of (shift, scale, gate) triple inside the 6H chunk
    const int* li_in = half == 0 ? li_q : li_m0;
    if (on(half == 0 ? 0 : 5)) {
      if (fused) {
        FLUXMI_TRY(fluxmi_k_ln_modulate(x, H, XB, a8, H, XB, mt + so * H, mt + (so + 1) * H, mi + so * H, mi + (so + 1) * H, MC,
                                        e->lin[li_in[0]].in_scale, e->lin[li_in[1]].in_scale, B, L, Lt, H, 1, e->lin[li_in[0]].in_fmt, s, ap));
      } else {
        FLUXMI_TRY(fluxmi_k_ln_modulate(x, H, XB, abf, H, XB, mt + so * H, mt + (so + 1) * H, mi + so * H, mi + (so + 1) * H, MC, nullptr,
                                        nullptr, B, L, Lt, H, 0, 0, s));
        for (int st = 0; st < 2; ++st)
          FLUXMI_TRY(stage_input(e, li_in[st], calib, trial, abf + (long long)roff[st] * H, H, XB, a8 + (long long)roff[st] * H, H, XB, B,
                                 rows[st], H, s));
      }
    }
    if (half == 0) {
      if (on(1)) {  // qkv GEMM (both streams, all batch elements in one grouped launch)
        std::vector<FluxmiGemmGroup> gs;
        for (int b = 0; b < B; ++b)
          for (int st = 0; st < 2; ++st) {
            const fluxmi_linear_t& l = e->lin[li_q[st]];
            const long long r0 = (long long)b * L + roff[st];
            FluxmiGemmGroup g = mk_group(l, l.kind ? (const void*)(a8 + r0 * H) : (const void*)(abf + r0 * H), H, qkv + r0 * 3 * H, 3 * H, rows[st]);
            g.W_pairs = pairs_of(e, li_q[st]);
            g.a_pairs = ap;
            if (fuse_v) {
              g.vt_out = VT + (long long)b * H * e->Lp; g.vt_ld = e->Lp; g.tok0 = roff[st];
              g.vt_rows = st == 0 ? Lt : e->Lp - Lt; g.kv_col0 = H; g.heads = heads;
              if (fuse_k) {  // K: QKNorm (this stream's key scale) + RoPE in the epilogue as well -> no relayout kernel at all
                g.k_out = K + (long long)b * H * L; g.k_rows = L; g.pe = pe + (long long)b * L * 128; g.k_norm = ns[st == 0 ? 3 : 1];
                g.k_f16 = attn_f16k();
              }
            }
            gs.push_back(g);
          }
        FLUXMI_TRY(run_gemm(gs, 3 * H, H, e->lin[li_q[0]].kind, e->lin[li_q[0]].in_fmt, FLUXMI_EPI_BF16, s));
      }
      // K and V^T are relaid out once (every query block re-reads them); Q is normalised + rotated inside the attention kernel
      if (on(2) && !fuse_k)
        FLUXMI_TRY(fluxmi_k_qkv_rope(qkv, 3 * H, pe, ns[2], ns[3], ns[0], ns[1], nullptr, K, fuse_v ? nullptr : VT, B, L, e->Lp, heads, Lt, attn_f16k(), s));
      if (on(3)) {
        if (fused) {
          // attention's last round leaves CUs idle (432 workgroups = 1.69 rounds at L = 4608): they pull in the weights this block needs
          // next -- proj and mlp.0 (94 MB); fluxmi_tuning_t.prefetch = 2: mlp.2 as well (170 MB, as much as the idle CUs read in that time)
          if (fluxmi_tuning().prefetch >= 2)
            set_pf(e, {li_p[0], li_p[1], li_m0[0], li_m0[1], li_m2[0], li_m2[1]}, idle_cus((long long)B * heads * ((L + 255) / 256)));
          else set_pf(e, {li_p[0], li_p[1], li_m0[0], li_m0[1]}, idle_cus((long long)B * heads * ((L + 255) / 256)));
          FLUXMI_TRY(fluxmi_k_attention(nullptr, K, VT, attn8, H, 0, 1, e->lin[li_p[0]].in_scale, e->lin[li_p[1]].in_scale, Lt, B, L, e->Lp,
                                        heads, e->lin[li_p[0]].in_fmt, s, qkv, 3 * H, pe, ns[2], ns[0], attn_f16k(), ap));
          fluxmi_set_prefetch(nullptr);
        } else {
          FLUXMI_TRY(fluxmi_k_attention(nullptr, K, VT, attnbf, H, 0, 0, nullptr, nullptr, Lt, B, L, e->Lp, heads, 0, s, qkv, 3 * H, pe,
                                        ns[2], ns[0], attn_f16k()));
          for (int st = 0; st < 2; ++st)
            FLUXMI_TRY(stage_input(e, li_p[st], calib, trial, attnbf + (long long)roff[st] * H, H, XB, attn8 + (long long)roff[st] * H, H,
                                   XB, B, rows[st], H, s));
        }
      }
      if (on(4)) {  // proj GEMM + gate1 * y + x
        std::vector<FluxmiGemmGroup> gs;
        for (int b = 0; b < B; ++b)
          for (int st = 0; st < 2; ++st) {
            const fluxmi_linear_t& l = e->lin[li_p[st]];
            const long long r0 = (long long)b * L + roff[st];
            FluxmiGemmGroup g = mk_group(l, l.kind ? (const void*)(attn8 + r0 * H) : (const void*)(attnbf + r0 * H), H, x + r0 * H, H, rows[st]);
            g.resid = x + r0 * H; g.ldr = H; g.gate = mods[st] + (long long)b * MC + 2 * H;
            g.a_pairs = ap;
            g.W_pairs = pairs_of(e, li_p[st]);
            gs.push_back(g);
          }
        FLUXMI_TRY(run_gemm(gs, H, H, e->lin[li_p[0]].kind, e->lin[li_p[0]].in_fmt, FLUXMI_EPI_GATE_RESID, s));
      }
    } else {
      if (on(6)) {  // mlp.0 (+GELU, + quantise for mlp.2)
        std::vector<FluxmiGemmGroup> gs;
        for (int b = 0; b < B; ++b)
          for (int st = 0; st < 2; ++st) {
            const fluxmi_linear_t& l = e->lin[li_m0[st]];
            const long long r0 = (long long)b * L + roff[st];
            FluxmiGemmGroup g = mk_group(l, l.kind ? (const void*)(a8 + r0 * H) : (const void*)(abf + r0 * H), H,
                                         fused ? (void*)(h8 + r0 * Hm) : (void*)(hbf + r0 * Hm), Hm, rows[st]);
            g.q_scale = e->lin[li_m2[st]].in_scale;
            g.W_pairs = pairs_of(e, li_m0[st]);
            g.a_pairs = ap; g.c8_pairs = ap;  // reads a8 and writes h8 in the row-pair layout
            if (fused && qlut_enabled()) g.q_lut = c.qlut + (size_t)(i * 2 + st) * 65536;
            gs.push_back(g);
          }
        FLUXMI_TRY(run_gemm(gs, Hm, H, e->lin[li_m0[0]].kind, e->lin[li_m2[0]].in_fmt, fused ? FLUXMI_EPI_GELU_QUANT : FLUXMI_EPI_BF16, s));
        if (!fused) {
          FLUXMI_TRY(fluxmi_k_act(hbf, hbf, B * L, Hm, Hm, Hm, 0, s));
          for (int st = 0; st < 2; ++st)
            FLUXMI_TRY(stage_input(e, li_m2[st], calib, trial, hbf + (long long)roff[st] * Hm, Hm, (long long)L * Hm,
                                   h8 + (long long)roff[st] * Hm, Hm, (long long)L * Hm, B, rows[st], Hm, s));
        }
      }
      if (on(7)) {  // mlp.2 + gate2 * y + x
        std::vector<FluxmiGemmGroup> gs;
        for (int b = 0; b < B; ++b)
          for (int st = 0; st < 2; ++st) {
            const fluxmi_linear_t& l = e->lin[li_m2[st]];
            const long long r0 = (long long)b * L + roff[st];
            FluxmiGemmGroup g = mk_group(l, l.kind ? (const void*)(h8 + r0 * Hm) : (const void*)(hbf + r0 * Hm), Hm, x + r0 * H, H, rows[st]);
            g.resid = x + r0 * H; g.ldr = H; g.gate = mods[st] + (long long)b * MC + 5 * H;
            g.W_pairs = pairs_of(e, li_m2[st]);
            g.a_pairs = ap;
            gs.push_back(g);
          }
        if (fused) {  // the 216-tile launch leaves 40 CUs idle: they pull in the first weights of the NEXT block
          const long long tiles = (long long)B * (((Lt + 255) / 256) + ((Li + 255) / 256)) * (H / 256);
          if (i + 1 < e->d.depth) set_pf(e, {DLi(e, i + 1, D_TXT_QKV), DLi(e, i + 1, D_IMG_QKV)}, idle_cus(tiles));
          else set_pf(e, {SLi(e, 0, S_LIN1)}, idle_cus(tiles));
        }
        FLUXMI_TRY(run_gemm(gs, H, Hm, e->lin[li_m2[0]].kind, e->lin[li_m2[0]].in_fmt, FLUXMI_EPI_GATE_RESID, s));
        fluxmi_set_prefetch(nullptr);
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// SingleStreamBlock.forward (flux_model.py:467-485) as five stages:
//   0 LN + modulate (+quantise) -> a8 | 1 linear1 -> qkv (+ V^T) and gelu(mlp) -> cat8[:, H:] | 2 K (and V^T) relayout
//   3 attention -> cat8[:, :H] | 4 linear2 + gate*y + x -> x
// ---------------------------------------------------------------------------------------------------------
int single_block(E* e, const Ctx& c, int i, int mode, int trial, int s0, int s1, hipStream_t s) {
  const bool fused = mode == 1, calib = mode == 0;
  const int H = c.H, Hm = c.Hm, B = c.B, L = c.L, heads = c.heads;
  const long long XB = c.XB, MC = c.MC;
  u16 *x = c.x, *qkv = c.qkv, *K = c.K, *VT = c.VT, *pe = c.pe, *abf = c.abf, *catbf = c.catbf, *lin1 = c.lin1;
  uint8_t *a8 = c.a8, *cat8 = c.cat8;
  auto on = [&](int st) { return st >= s0 && st <= s1; };
  const int HC = H + Hm;
  const u16* ms = c.mod + (long long)e->d.depth * 12 * H + (long long)i * 3 * H;  // shift scale gate
  const int ap = act_pairs(e, fused) ? 1 : 0;  // fp8 activation buffers in the row-pair layout (see act_pairs)
  const int l1 = SLi(e, i, S_LIN1), l2 = SLi(e, i, S_LIN2);
  const fluxmi_linear_t &L1 = e->lin[l1], &L2 = e->lin[l2];
  const void* const* ns = &e->norm[e->d.depth * 4 + i * 2];
  if (fused) {
    const bool fuse_v = fuse_kv_level() >= 1 && fluxmi_gemm_tile_ok(3 * H + Hm, H, 1, 13);
    const bool fuse_k = fuse_kv_level() >= 2 && fuse_v && H % 256 == 0 && L >= 2048;  // short sequences: one launch of the relayout kernel is cheaper
    if (on(0))
      FLUXMI_TRY(fluxmi_k_ln_modulate(x, H, XB, a8, H, XB, ms, ms + H, ms, ms + H, MC, L1.in_scale, L1.in_scale, B, L, L, H, 1, L1.in_fmt, s, ap));
    if (on(1)) {
      std::vector<FluxmiGemmGroup> gs;
      for (int b = 0; b < B; ++b) {  // one group per batch element: the fused V^T output is per sequence
        const long long r0 = (long long)b * L;
        FluxmiGemmGroup g = mk_group(L1, a8 + r0 * H, H, qkv + r0 * 3 * H, 3 * H, L);
        g.C2 = cat8 + r0 * HC; g.ldc2 = HC; g.split_n = 3 * H; g.c2_col0 = H; g.q_scale = L2.in_scale;
        g.W_pairs = pairs_of(e, l1);
        g.a_pairs = ap; g.c8_pairs = ap;  // reads a8, writes gelu(mlp) into cat8[:, H:], both in the row-pair layout
        if (qlut_enabled()) g.q_lut = c.qlut + (size_t)(e->d.depth * 2 + i) * 65536;
        if (fuse_v) {
          g.vt_out = VT + (long long)b * H * e->Lp; g.vt_ld = e->Lp; g.tok0 = 0; g.vt_rows = e->Lp; g.kv_col0 = H; g.heads = heads;
          if (fuse_k) {
            g.k_out = K + (long long)b * H * L; g.k_rows = L; g.pe = pe + (long long)b * L * 128; g.k_norm = ns[1]; g.k_f16 = attn_f16k();
          }
        }
        gs.push_back(g);
      }
      FLUXMI_TRY(run_gemm(gs, 3 * H + Hm, H, 1, L2.in_fmt, FLUXMI_EPI_SPLIT, s));
    }
    if (on(2) && !fuse_k)
      FLUXMI_TRY(fluxmi_k_qkv_rope(qkv, 3 * H, pe, ns[0], ns[1], ns[0], ns[1], nullptr, K, fuse_v ? nullptr : VT, B, L, e->Lp, heads, L, attn_f16k(), s));
    if (on(3)) {
      // attention's idle CUs pull in linear2's weights; fluxmi_tuning_t.prefetch = 3: the NEXT block's linear1 as well (113 MB in all)
      if (fluxmi_tuning().prefetch >= 3 && i + 1 < e->d.depth_single) set_pf(e, {l2, SLi(e, i + 1, S_LIN1)}, idle_cus((long long)B * heads * ((L + 255) / 256)));
      else set_pf(e, {l2}, idle_cus((long long)B * heads * ((L + 255) / 256)));
      FLUXMI_TRY(fluxmi_k_attention(nullptr, K, VT, cat8, HC, 0, 1, L2.in_scale, L2.in_scale, L, B, L, e->Lp, heads, L2.in_fmt, s, qkv,
                                    3 * H, pe, ns[0], ns[0], attn_f16k(), ap));
      fluxmi_set_prefetch(nullptr);
    }
  } else {
    if (on(0)) {
      FLUXMI_TRY(fluxmi_k_ln_modulate(x, H, XB, abf, H, XB, ms, ms + H, ms, ms + H, MC, nullptr, nullptr, B, L, L, H, 0, 0, s));
      FLUXMI_TRY(stage_input(e, l1, calib, trial, abf, H, 0, a8, H, 0, 1, B * L, H, s));
    }
    if (on(1)) {
      std::vector<FluxmiGemmGroup> gs;
      gs.push_back(mk_group(L1, L1.kind ? (const void*)a8 : (const void*)abf, H, lin1, 3 * H + Hm, B * L));
      gs.back().W_pairs = pairs_of(e, l1);
      FLUXMI_TRY(run_gemm(gs, 3 * H + Hm, H, L1.kind, L1.in_fmt, FLUXMI_EPI_BF16, s));
    }
    if (on(2)) FLUXMI_TRY(fluxmi_k_qkv_rope(lin1, 3 * H + Hm, pe, ns[0], ns[1], ns[0], ns[1], nullptr, K, VT, B, L, e->Lp, heads, L, attn_f16k(), s));
    if (on(3)) {
      FLUXMI_TRY(fluxmi_k_attention(nullptr, K, VT, catbf, HC, 0, 0, nullptr, nullptr, L, B, L, e->Lp, heads, 0, s, lin1, 3 * H + Hm, pe,
                                    ns[0], ns[0], attn_f16k()));
      FLUXMI_TRY(fluxmi_k_act(lin1 + 3 * H, catbf + H, B * L, Hm, 3 * H + Hm, HC, 0, s));
      FLUXMI_TRY(stage_input(e, l2, calib, trial, catbf, HC, 0, cat8, HC, 0, 1, B * L, HC, s));
    }
  }
  if (on(4)) {
    std::vector<FluxmiGemmGroup> gs;
    for (int b = 0; b < B; ++b) {
      const long long r0 = (long long)b * L;
      FluxmiGemmGroup g = mk_group(L2, L2.kind ? (const void*)(cat8 + r0 * HC) : (const void*)(catbf + r0 * HC), HC, x + r0 * H, H, L);
      g.resid = x + r0 * H; g.ldr = H; g.gate = ms + (long long)b * MC + 2 * H;
      g.W_pairs = pairs_of(e, l2);
      g.a_pairs = ap;
      gs.push_back(g);
    }
    if (fused) {  // linear2's 216 tiles leave 40 CUs idle: they pull in the next block's linear1 (after the last block: the next step's first qkv)
      const long long tiles = (long long)B * ((L + 255) / 256) * (H / 256);
      if (i + 1 < e->d.depth_single) set_pf(e, {SLi(e, i + 1, S_LIN1)}, idle_cus(tiles));
      else set_pf(e, {DLi(e, 0, D_TXT_QKV), DLi(e, 0, D_IMG_QKV)}, idle_cus(tiles));
    }
    FLUXMI_TRY(run_gemm(gs, H, HC, L2.kind, L2.in_fmt, FLUXMI_EPI_GATE_RESID, s));
    fluxmi_set_prefetch(nullptr);
  }
  return 0;
}

// LastLayer.forward (flux_model.py:499-503) on the img rows of x: stage 0 = (1 + scale) * LayerNorm(x) + shift -> fin (bf16; the adaLN
// vectors are the last 2H entries of `mod`, shift first), stage 1 = the bf16 Linear hidden -> patch channels -> pred.  Never fp8
// (float8_quantize.py:476).
int final_layer(E* e, u16* pred, int s0, int s1, hipStream_t s) {
  const int H = e->d.hidden, B = e->B, L = e->L, Lt = e->Lt, Li = e->Li;
  const long long XB = (long long)L * H, MC = e->mod_cols;
  u16 *x = buf<u16>(e, "x"), *mod = buf<u16>(e, "mod"), *fin = buf<u16>(e, "fin");
  const u16* mf = mod + (long long)e->d.depth * 12 * H + (long long)e->d.depth_single * 3 * H;  // shift | scale
  if (s0 <= 0 && s1 >= 0)
    FLUXMI_TRY(fluxmi_k_ln_modulate(x + (long long)Lt * H, H, XB, fin, H, (long long)Li * H, mf, mf + H, mf, mf + H, MC, nullptr, nullptr, B,
                                    Li, Li, H, 0, 0, s));
  if (s0 <= 1 && s1 >= 1) {
    const fluxmi_linear_t& l = e->lin[e->i_final_lin];
    std::vector<FluxmiGemmGroup> gs;
    gs.push_back(mk_group(l, fin, H, pred, l.N, B * Li));
    FLUXMI_TRY(run_gemm_fixed_cfg(gs, l.N, H, 0, 0, FLUXMI_EPI_BF16, s));
  }
  return 0;
}

int require_all_f8(E* e) {
  for (int i = 0; i < e->d.depth; ++i)
    for (int sl : {D_IMG_QKV, D_IMG_PROJ, D_IMG_MLP0, D_IMG_MLP2, D_TXT_QKV, D_TXT_PROJ, D_TXT_MLP0, D_TXT_MLP2})
      FLUXMI_REQUIRE(DL(e, i, sl).kind == 1, "fused mode needs every block linear to be F8Linear (double block %d)", i);
  for (int i = 0; i < e->d.depth_single; ++i)
    FLUXMI_REQUIRE(SL(e, i, S_LIN1).kind == 1 && SL(e, i, S_LIN2).kind == 1, "fused mode needs F8Linear in single block %d", i);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
int forward_impl(E* e, const u16* img, const u16* txt, const u16* y, const u16* t_vec, const u16* g_vec, u16* pred, int mode,
                 int trial, bool txt_cached, hipStream_t s) {
  const int H = e->d.hidden, Hm = e->d.mlp_hidden, B = e->B, L = e->L, Lt = e->Lt, Li = e->Li, heads = e->d.heads;
  const bool fused = mode == 1, calib = mode == 0;
  const long long XB = (long long)L * H;  // batch stride of x
  u16* x = buf<u16>(e, "x");
  u16 *mod = buf<u16>(e, "mod"), *qkv = buf<u16>(e, "qkv"), *Q = buf<u16>(e, "Q"), *K = buf<u16>(e, "K"), *VT = buf<u16>(e, "VT");
  u16 *pe = buf<u16>(e, "pe"), *abf = buf<u16>(e, "abf"), *catbf = buf<u16>(e, "catbf"), *hbf = buf<u16>(e, "hbf");
  u16 *attnbf = buf<u16>(e, "attnbf"), *fin = buf<u16>(e, "fin");
  uint8_t *a8 = buf<uint8_t>(e, "a8"), *attn8 = buf<uint8_t>(e, "attn8"), *h8 = buf<uint8_t>(e, "h8"), *cat8 = buf<uint8_t>(e, "cat8");
  uint8_t* in8 = buf<uint8_t>(e, "in8");
  const long long MC = e->mod_cols;

  if (fused) FLUXMI_TRY(require_all_f8(e));
  const Ctx ctx = make_ctx(e);

  // ---- img_in / txt_in                                                             flux_model.py:686,699
  {
    const fluxmi_linear_t& l = e->lin[e->i_img_in];
    const int C = e->d.in_channels;
    FLUXMI_TRY(stage_input(e, e->i_img_in, calib, trial, img, C, 0, in8, C, 0, 1, B * Li, C, s));
    std::vector<FluxmiGemmGroup> gs;
    for (int b = 0; b < B; ++b)
      gs.push_back(mk_group(l, l.kind ? (const void*)(in8 + (long long)b * Li * C) : (const void*)(img + (long long)b * Li * C), C,
                            x + b * XB + (long long)Lt * H, H, Li));
    FLUXMI_TRY(run_gemm_fixed_cfg(gs, H, C, l.kind, l.in_fmt, FLUXMI_EPI_BF16, s));
  }
  if (txt_cached) {
    FLUXMI_CHECK_HIP(hipMemcpy2DAsync(x, XB * 2, buf<u16>(e, "txt_emb"), (size_t)Lt * H * 2, (size_t)Lt * H * 2, B, hipMemcpyDeviceToDevice, s));
  } else {
    FLUXMI_TRY(embed_txt(e, txt, calib, trial, x, XB, s));
  }
  if (e->mods_table) {
    FLUXMI_TRY(fluxmi_k_select_step(e->mods_all, e->d_step, e->d_step0, mod, (long long)B * MC * 2, s));
  } else {
    FLUXMI_TRY(compute_vec_and_mods(e, t_vec, g_vec, y, calib, trial, s));
  }

  for (int i = 0; i < e->d.depth; ++i) FLUXMI_TRY(double_block(e, ctx, i, mode, trial, 0, DOUBLE_STAGES - 1, s));
  for (int i = 0; i < e->d.depth_single; ++i) FLUXMI_TRY(single_block(e, ctx, i, mode, trial, 0, SINGLE_STAGES - 1, s));

  // ---- final layer                                                                flux_model.py:499-503, 714-715
  FLUXMI_TRY(final_layer(e, pred, 0, 1, s));
  return 0;
}

bool needs_splitk(E* e) {
  for (const fluxmi_linear_t& l : e->lin)
    if (!l.kind && (long long)l.K * 2 / 64 >= 192) return true;
  return false;
}
// the engine's split-K scratch for every launch of this thread while an engine entry point runs
// ... and its scratch for attention's balanced grid; a prefetch left pending by an error return between set_pf and the launch that
// would have consumed it is dropped here, on entry and on exit (it points into weights the caller may free afterwards)
struct SplitkScope {
  explicit SplitkScope(E* e) {
    auto it = e->bufs.find("splitk");
    fluxmi_set_splitk_scratch(it != e->bufs.end() && it->second.n >= FLUXMI_SPLITK_WS_BYTES ? (float*)it->second.p : nullptr);
    auto ia = e->bufs.find("attn_part");
    fluxmi_set_attn_scratch(ia != e->bufs.end() && ia->second.n >= FLUXMI_ATTN_SPLIT_WS_BYTES ? ia->second.p : nullptr);
    fluxmi_set_prefetch(nullptr);
    fluxmi_gemm_set_batch(e->B);
  }
  ~SplitkScope() {
    fluxmi_gemm_set_batch(1);
    fluxmi_set_splitk_scratch(nullptr);
    fluxmi_set_attn_scratch(nullptr);
    fluxmi_set_prefetch(nullptr);
  }
};

void free_ws(E* e) {
  if (e->exec) { hipGraphExecDestroy(e->exec); e->exec = nullptr; }
  e->graph_ok = false;
  e->warmed = false;
  e->qlut_valid = false;
  if (e->ws) { hipFree(e->ws); e->ws = nullptr; }
  e->bufs.clear();
  e->ws_bytes = 0;
}

}  // namespace

// roctx ranges (rocprofv3 --marker-trace) around the phases of a denoise call, resolved at run time so that the library has no
// hard dependency on the profiler: fluxmi_tuning_t.roctx (FLUXMI_ROCTX=1) turns them on.
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (!fluxmi_tuning().roctx) return;
    void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
Roctx& roctx() { static Roctx r; return r; }
struct Range {
  Range(const char* name) { if (roctx().push) roctx().push(name); }
  ~Range() { if (roctx().pop) roctx().pop(); }
};
}  // namespace

extern "C" {

int fluxmi_engine_num_linears(const fluxmi_model_desc_t* desc) { return desc ? lin_count(*desc) : -1; }

int fluxmi_engine_create(const fluxmi_model_desc_t* desc, const fluxmi_linear_t* linears, int n_linears,
                         const void* const* norm_scales, int n_norm_scales, fluxmi_engine_t** out) {
  FLUXMI_REQUIRE(desc && linears && norm_scales && out, "engine_create: NULL argument");
  fluxmi_log_tuning("engine_create");  // the kernel choices this engine will run with (FLUXMI_LOG=1)
  FLUXMI_REQUIRE(desc->hidden == desc->heads * 128, "engine_create: head_dim must be 128 (hidden %d, heads %d)", desc->hidden, desc->heads);
  FLUXMI_REQUIRE(desc->axes_dim[0] + desc->axes_dim[1] + desc->axes_dim[2] == 128, "engine_create: sum(axes_dim) must be 128");
  FLUXMI_REQUIRE(n_linears == lin_count(*desc), "engine_create: expected %d linears, got %d", lin_count(*desc), n_linears);
  FLUXMI_REQUIRE(n_norm_scales == desc->depth * 4 + desc->depth_single * 2, "engine_create: expected %d norm scales, got %d",
                 desc->depth * 4 + desc->depth_single * 2, n_norm_scales);
  FLUXMI_REQUIRE(desc->num_trials >= 1 && desc->num_trials <= 64, "engine_create: num_trials out of range");
  E* e = new E();
  e->d = *desc;
  e->lin.assign(linears, linears + n_linears);
  e->norm.assign(norm_scales, norm_scales + n_norm_scales);
  int i = 0;
  e->i_img_in = i++; e->i_time_in = i; i += 2; e->i_vec_in = i; i += 2;
  e->i_guid_in = -1;
  if (desc->guidance_embed) { e->i_guid_in = i; i += 2; }
  e->i_txt_in = i++;
  e->i_double0 = i; i += desc->depth * 10;
  e->i_single0 = i; i += desc->depth_single * 3;
  e->i_final_mod = i++; e->i_final_lin = i++;
  const int H = desc->hidden;
  e->mod_cols = (long long)desc->depth * 12 * H + (long long)desc->depth_single * 3 * H + 2 * H;
  // constants block
  const int n_mod = desc->depth * 2 + desc->depth_single + 1;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_freqs = carve(128 * 4), o_omega = carve(64 * 4), o_axis = carve(64 * 4), o_ts = carve((MAX_STEPS + 1) * 4),
               o_dts = carve((MAX_STEPS + 1) * 4), o_step = carve(4), o_step0 = carve(4), o_amax = carve((size_t)n_linears * 4),
               o_gemv = carve(sizeof(FluxmiGemvLayer) * n_mod), o_cm = carve(sizeof(FluxmiCalibLayer) * n_mod);
  if (hipMalloc((void**)&e->consts, off) != hipSuccess) { delete e; fluxmi_set_error("engine_create: hipMalloc(%zu) failed", off); return 2; }
  e->d_freqs = (float*)(e->consts + o_freqs); e->d_omega = (float*)(e->consts + o_omega); e->d_axis = (int*)(e->consts + o_axis);
  e->d_ts = (float*)(e->consts + o_ts); e->d_dts = (float*)(e->consts + o_dts); e->d_step = (int*)(e->consts + o_step);
  e->d_amax = e->d_amax_own = (float*)(e->consts + o_amax); e->d_gemv = (FluxmiGemvLayer*)(e->consts + o_gemv);
  e->d_calib_mod = (FluxmiCalibLayer*)(e->consts + o_cm);
  e->d_step0 = (int*)(e->consts + o_step0);
  hipMemset(e->consts, 0, off);
  // pinned staging for the schedule + the events (guard of the staging buffer, timing of the frozen steps)
  if (hipHostMalloc((void**)&e->h_sched, 2 * (MAX_STEPS + 1) * sizeof(float), hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_sched, hipEventDisableTiming) != hipSuccess || hipEventCreate(&e->ev_t0) != hipSuccess ||
      hipEventCreate(&e->ev_t1) != hipSuccess) {
    fluxmi_engine_destroy(e);
    fluxmi_set_error("engine_create: pinned staging / event allocation failed");
    return 2;
  }
  *out = e;
  return 0;
}

int fluxmi_engine_destroy(fluxmi_engine_t* e) {
  if (!e) return 0;
  free_ws(e);
  if (e->mods_all) hipFree(e->mods_all);
  if (e->pairs) hipFree(e->pairs);
  if (e->consts) hipFree(e->consts);
  if (e->h_sched) hipHostFree(e->h_sched);
  if (e->ev_sched) hipEventDestroy(e->ev_sched);
  if (e->ev_t0) hipEventDestroy(e->ev_t0);
  if (e->ev_t1) hipEventDestroy(e->ev_t1);
  delete e;
  return 0;
}

int fluxmi_engine_rebind(fluxmi_engine_t* e, const fluxmi_linear_t* linears, int n_linears) {
  FLUXMI_REQUIRE(e && linears && n_linears == (int)e->lin.size(), "engine_rebind: bad arguments");
  e->lin.assign(linears, linears + n_linears);
  e->graph_ok = false;
  e->warmed = false;
  e->txt_emb_valid = false;
  e->pairs_dirty = true;
  e->qlut_valid = false;
  if (e->ws) {
    // the split-K scratch is sized by the linears' kinds (bf16 linears with long K): a rebind that introduces such linears (an fp8
    // model re-bound to bf16 weights) drops the workspace, and the next fluxmi_engine_prepare allocates one with the scratch
    auto it = e->bufs.find("splitk");
    const bool have = it != e->bufs.end() && it->second.n >= FLUXMI_SPLITK_WS_BYTES;
    if (needs_splitk(e) && !have) {
      FLUXMI_CHECK_HIP(hipDeviceSynchronize());
      free_ws(e);
      return 0;
    }
    return build_gemv_table(e, 0);
  }
  return 0;
}

// host_consts: [0,128) timestep freqs, [128,192) rope omega per pair, then 64 ints (axis per pair) -- computed by the host
// with the reference's own torch expressions so that the tables are bit-identical (flux_model.py:50-51,106-110).
int fluxmi_engine_set_tables(fluxmi_engine_t* e, const float* freqs128, const float* omega64, const int* axis64) {
  FLUXMI_REQUIRE(e && freqs128 && omega64 && axis64, "engine_set_tables: NULL argument");
  FLUXMI_CHECK_HIP(hipMemcpy(e->d_freqs, freqs128, 128 * 4, hipMemcpyHostToDevice));
  FLUXMI_CHECK_HIP(hipMemcpy(e->d_omega, omega64, 64 * 4, hipMemcpyHostToDevice));
  FLUXMI_CHECK_HIP(hipMemcpy(e->d_axis, axis64, 64 * 4, hipMemcpyHostToDevice));
  return 0;
}

int fluxmi_engine_prepare(fluxmi_engine_t* e, int B, int Li, int Lt, const void* img_ids, const void* txt_ids, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  FLUXMI_REQUIRE(e && B >= 1 && B <= FLUXMI_ENGINE_MAX_BATCH && Li >= 1 && Lt >= 0, "engine_prepare: bad shape B=%d Li=%d Lt=%d (B must be 1..%d)", B, Li,
                 Lt, FLUXMI_ENGINE_MAX_BATCH);
  FLUXMI_REQUIRE(img_ids && (Lt == 0 || txt_ids), "engine_prepare: NULL ids");
  const int H = e->d.hidden, Hm = e->d.mlp_hidden, L = Li + Lt, Lp = ((L + 63) / 64) * 64;
  if (e->pairs_skipped) e->pairs_dirty = true;  // the row-pair copies did not fit last time: try again with this request
  if (B != e->B || Li != e->Li || Lt != e->Lt || !e->ws) {
    FLUXMI_CHECK_HIP(hipStreamSynchronize(s));
    free_ws(e);
    e->B = B; e->Li = Li; e->Lt = Lt; e->L = L; e->Lp = Lp;
    struct Item { const char* name; size_t bytes; };
    const size_t BL = (size_t)B * L;
    const size_t in8 = std::max((size_t)B * Li * e->d.in_channels, (size_t)B * Lt * e->d.ctx_in);
    std::vector<Item> items = {
        {"x", BL * H * 2}, {"a8", BL * H}, {"attn8", BL * H}, {"qkv", BL * 3 * H * 2}, {"Q", BL * H * 2}, {"K", BL * H * 2},
        {"VT", (size_t)B * H * Lp * 2}, {"h8", BL * Hm}, {"cat8", BL * (H + Hm)}, {"pe", BL * 64 * 2 * 2},
        {"mod", (size_t)B * e->mod_cols * 2}, {"fin", (size_t)B * Li * H * 2},
        // unfused-path temporaries
        {"abf", BL * H * 2}, {"attnbf", BL * H * 2}, {"hbf", BL * Hm * 2}, {"catbf", BL * (H + Hm) * 2}, {"lin1", BL * (3 * H + Hm) * 2},
        {"in8", in8},
        // small
        {"vec", (size_t)B * H * 2}, {"svec", (size_t)B * H * 2}, {"temb", (size_t)B * 256 * 2}, {"emb_h", (size_t)B * H * 2},
        {"emb_s", (size_t)B * std::max(H, 1024) * 2}, {"vec_t", (size_t)B * H * 2}, {"vec_g", (size_t)B * H * 2}, {"vec_y", (size_t)B * H * 2},
        {"tvec", 256}, {"gvec", 256}, {"ids", BL * 3 * 2},
        {"qlut", (size_t)(e->d.depth * 2 + e->d.depth_single) * 65536},
        {"mods_a8", (size_t)FLUXMI_MAX_GROUPS * (((size_t)B * std::max(H, 4096) + 255) & ~(size_t)255)},
        // static request buffers (make the captured graph independent of caller pointers)
        {"img_s", (size_t)B * Li * e->d.in_channels * 2}, {"txt_s", (size_t)B * Lt * e->d.ctx_in * 2}, {"y_s", (size_t)B * e->d.vec_in * 2},
        {"pred_s", (size_t)B * Li * e->d.in_channels * 2}, {"txt_emb", (size_t)B * Lt * H * 2},
        // split-K partial tiles of the bf16 small-M launches (api.cpp: bf16 operands, >= 192 K-steps): owned by the engine, because its step
        // graph is captured on a private stream and replayed on the caller's -- a scratch keyed by stream would be nobody's
        {"splitk", needs_splitk(e) ? FLUXMI_SPLITK_WS_BYTES : 256},
        // partial softmax states + arrival counters of attention's balanced grid (attention2.hip, AttnSplit; 69 MB), when this shape uses it.
        // Zeroed with the rest of the workspace below; the kernel leaves the counters at zero.
        {"attn_part", fluxmi_attn_plan_any(B, L, e->d.heads) ? FLUXMI_ATTN_SPLIT_WS_BYTES : 256},
    };
    size_t total = 0;
    for (auto& it : items) total += (it.bytes + 255) & ~(size_t)255;
    if (hipMalloc((void**)&e->ws, total) != hipSuccess) {
      fluxmi_set_error("engine_prepare: hipMalloc(%zu bytes) failed", total);
      return 2;
    }
    e->ws_bytes = total;
    FLUXMI_CHECK_HIP(hipMemsetAsync(e->ws, 0, total, s));
    size_t off = 0;
    for (auto& it : items) {
      e->bufs[it.name] = Buf{e->ws + off, it.bytes};
      off += (it.bytes + 255) & ~(size_t)255;
    }
    FLUXMI_TRY(build_gemv_table(e, s));
    // step-ahead modulation table: MODS_STEPS steps x B rows (59 MB per 28 steps at Flux-dev); engine_denoise never allocates
    const int rows_cap = MODS_STEPS * B;
    const size_t need = mods_table_bytes(e, (size_t)rows_cap);
    if (need > e->mods_all_bytes || rows_cap != e->mods_rows_cap) {
      if (e->mods_all) hipFree(e->mods_all);
      e->mods_all = nullptr; e->mods_all_bytes = 0;
      if (hipMalloc((void**)&e->mods_all, need) != hipSuccess) { fluxmi_set_error("engine_prepare: hipMalloc(%zu) failed (modulation table)", need); return 2; }
      e->mods_all_bytes = need;
    }
    e->mods_rows_cap = rows_cap;
  }
  // ids = cat(txt_ids, img_ids) per batch element; pe table                              flux_model.py:701-702
  u16* ids = buf<u16>(e, "ids");
  if (Lt > 0)
    FLUXMI_CHECK_HIP(hipMemcpy2DAsync(ids, (size_t)L * 6, txt_ids, (size_t)Lt * 6, (size_t)Lt * 6, B, hipMemcpyDeviceToDevice, s));
  FLUXMI_CHECK_HIP(hipMemcpy2DAsync(ids + (size_t)Lt * 3, (size_t)L * 6, img_ids, (size_t)Li * 6, (size_t)Li * 6, B, hipMemcpyDeviceToDevice, s));
  FLUXMI_TRY(fluxmi_k_rope_table(ids, e->d_omega, e->d_axis, buf<u16>(e, "pe"), (long long)B * L, 3, 64, s));
  e->txt_emb_valid = false;
  return 0;
}

int fluxmi_engine_forward(fluxmi_engine_t* e, const void* img, const void* txt, const void* y, const void* timesteps,
                          const void* guidance, void* pred, int mode, int trial_index, void* stream) {
  FLUXMI_REQUIRE(e && e->ws, "engine_forward: call fluxmi_engine_prepare first");
  FLUXMI_REQUIRE(img && txt && y && timesteps && pred, "engine_forward: NULL tensor");
  FLUXMI_REQUIRE(mode >= 0 && mode <= 2, "engine_forward: bad mode %d", mode);
  if (mode == 0) FLUXMI_REQUIRE(trial_index >= 0 && trial_index <= e->d.num_trials, "engine_forward: trial_index %d out of range", trial_index);
  if (mode == 0) e->qlut_valid = false;  // input scales move during calibration
  if (mode == 1) FLUXMI_TRY(build_qluts(e, (hipStream_t)stream));
  SplitkScope splitk(e);
  FLUXMI_TRY(ensure_pairs(e, (hipStream_t)stream));
  return forward_impl(e, (const u16*)img, (const u16*)txt, (const u16*)y, (const u16*)timesteps, (const u16*)guidance, (u16*)pred,
                      mode, trial_index, false, (hipStream_t)stream);
}


int fluxmi_engine_denoise(fluxmi_engine_t* e, void* img, const void* txt, const void* y, float guidance,
                          const double* timesteps_host, int n_steps, int* trial_index_inout, int use_graph, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  FLUXMI_REQUIRE(e && e->ws, "engine_denoise: call fluxmi_engine_prepare first");
  FLUXMI_REQUIRE(img && txt && y && timesteps_host && trial_index_inout, "engine_denoise: NULL argument");
  FLUXMI_REQUIRE(n_steps >= 0 && n_steps <= MAX_STEPS, "engine_denoise: n_steps=%d out of range", n_steps);
  Range whole("fluxmi_engine_denoise");
  SplitkScope splitk(e);
  FLUXMI_TRY(ensure_pairs(e, s));
  const int B = e->B, Li = e->Li, Lt = e->Lt, C = e->d.in_channels;
  // any bf16 block linear -> the fused path is unavailable, run unfused-frozen (mode 2)
  bool all_f8 = true;
  for (int i = e->i_double0; i < e->i_final_mod; ++i) {
    const int rel = i < e->i_single0 ? (i - e->i_double0) % 10 : -1;
    const bool is_mod = (i < e->i_single0) ? (rel == D_IMG_MOD || rel == D_TXT_MOD) : ((i - e->i_single0) % 3 == S_MOD);
    if (!is_mod && !e->lin[i].kind) all_f8 = false;
  }
  bool any_f8 = false;
  for (auto& l : e->lin) any_f8 |= (l.kind != 0);

  // schedule -> device through the engine's pinned staging buffer.  The buffer may still be the source of the previous request's
  // (long finished) copy: wait on that copy's event, never on the stream.
  if (e->sched_pending) FLUXMI_CHECK_HIP(hipEventSynchronize(e->ev_sched));
  float *h_ts = e->h_sched, *h_dts = e->h_sched + (MAX_STEPS + 1);
  for (int i = 0; i <= n_steps; ++i) h_ts[i] = (float)timesteps_host[i];
  for (int i = 0; i < n_steps; ++i) h_dts[i] = (float)(timesteps_host[i + 1] - timesteps_host[i]);
  h_dts[n_steps] = 0.f;
  FLUXMI_CHECK_HIP(hipMemcpyAsync(e->d_ts, h_ts, (n_steps + 1) * 4, hipMemcpyHostToDevice, s));
  FLUXMI_CHECK_HIP(hipMemcpyAsync(e->d_dts, h_dts, (n_steps + 1) * 4, hipMemcpyHostToDevice, s));
  FLUXMI_CHECK_HIP(hipEventRecord(e->ev_sched, s));
  e->sched_pending = true;
  u16 *gvec = buf<u16>(e, "gvec"), *tvec = buf<u16>(e, "tvec");
  FLUXMI_TRY(fluxmi_k_fill_bf16(gvec, guidance, B, s));  // guidance arrives in the flow dtype (flux_pipeline.py:619-623)
  FLUXMI_CHECK_HIP(hipMemsetAsync(e->d_step, 0, 4, s));
  u16 *img_s = buf<u16>(e, "img_s"), *txt_s = buf<u16>(e, "txt_s"), *y_s = buf<u16>(e, "y_s"), *pred_s = buf<u16>(e, "pred_s");
  const long long n_img = (long long)B * Li * C;
  FLUXMI_CHECK_HIP(hipMemcpyAsync(img_s, img, n_img * 2, hipMemcpyDeviceToDevice, s));
  FLUXMI_CHECK_HIP(hipMemcpyAsync(txt_s, txt, (size_t)B * Lt * e->d.ctx_in * 2, hipMemcpyDeviceToDevice, s));
  FLUXMI_CHECK_HIP(hipMemcpyAsync(y_s, y, (size_t)B * e->d.vec_in * 2, hipMemcpyDeviceToDevice, s));

  int trial = *trial_index_inout;
  int step = 0;
  const u16* g_arg = e->d.guidance_embed ? gvec : nullptr;
  // -- calibrating steps: the reference's first num_trials+1 calls of every F8Linear ----------------------
  {
    Range r("calibrating steps (unfused)");
    while (step < n_steps && any_f8 && trial <= e->d.num_trials) {
      FLUXMI_TRY(fluxmi_k_set_timestep(tvec, e->d_ts, e->d_step, B, s));
      e->qlut_valid = false;
      FLUXMI_TRY(forward_impl(e, img_s, txt_s, y_s, tvec, g_arg, pred_s, 0, trial, false, s));
      FLUXMI_TRY(fluxmi_k_euler(img_s, pred_s, e->d_dts, e->d_step, n_img, s));
      FLUXMI_TRY(fluxmi_k_advance_step(e->d_step, s));
      ++trial; ++step;
    }
  }
  // -- frozen steps -------------------------------------------------------------------------------------------
  e->timed_steps = 0;
  if (step < n_steps) {
    const int mode = all_f8 ? 1 : 2;
    if (mode == 1) {
      FLUXMI_TRY(embed_txt(e, txt_s, false, 0, buf<u16>(e, "txt_emb"), (long long)Lt * e->d.hidden, s));
      e->txt_emb_valid = true;
      FLUXMI_TRY(build_qluts(e, s));
    }
    auto one_step = [&](hipStream_t st) -> int {
      e->mods_table = true;
      int rc = forward_impl(e, img_s, txt_s, y_s, tvec, g_arg, pred_s, mode, 0, mode == 1, st);
      e->mods_table = false;
      FLUXMI_TRY(rc);
      FLUXMI_TRY(fluxmi_k_euler(img_s, pred_s, e->d_dts, e->d_step, n_img, st));
      return fluxmi_k_advance_step(e->d_step, st);
    };
    // ev_t0 .. ev_t1 (fluxmi_engine_last_timing) brackets frozen STEPS only: recorded behind the first window's modulation table, the eager
    // warm step and the graph capture of a new shape (a request longer than MODS_STEPS steps includes its later table builds)
    bool t0_recorded = false;
    int first_timed = step;
    // the modulation vectors of up to MODS_STEPS steps are produced ahead (one pass over the 3.2 GB of modulation weights per window);
    // the table address and the device-side window origin never change, so ONE captured graph serves every step of every request
    while (step < n_steps) {
      const int win_end = std::min(n_steps, step + MODS_STEPS);
      {
        Range r("step-ahead modulation table");
        FLUXMI_TRY(precompute_mods(e, step, win_end, g_arg, y_s, s));
      }
      if (e->graph_ok && e->graph_gen != fluxmi_tuning_generation()) {
        e->graph_ok = false;  // kernel choices are baked into a captured graph: never replay one captured under other knobs
        e->warmed = false;
        e->qlut_valid = false;
      }
      if (use_graph && !e->graph_ok) {
        // the first frozen step of a shape runs eagerly so that every lazy one-time init (function attributes) happens outside capture
        if (!e->warmed) {
          FLUXMI_TRY(one_step(s));
          ++step;
          e->warmed = true;
        }
        if (step < win_end) {
          hipStream_t cs;
          FLUXMI_CHECK_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
          FLUXMI_CHECK_HIP(hipStreamSynchronize(s));  // one-time, at graph capture only
          hipGraph_t graph = nullptr;
          FLUXMI_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
          int rc = one_step(cs);
          hipError_t ce = hipStreamEndCapture(cs, &graph);
          if (rc || ce != hipSuccess) {
            if (graph) hipGraphDestroy(graph);
            hipStreamDestroy(cs);
            if (!rc) fluxmi_set_error("engine_denoise: hipStreamEndCapture failed: %s", hipGetErrorString(ce));
            return rc ? rc : 2;
          }
          if (e->exec) { hipGraphExecDestroy(e->exec); e->exec = nullptr; }
          hipError_t ie = hipGraphInstantiate(&e->exec, graph, nullptr, nullptr, 0);
          hipGraphDestroy(graph);
          hipStreamDestroy(cs);
          if (ie != hipSuccess) { fluxmi_set_error("engine_denoise: hipGraphInstantiate failed: %s", hipGetErrorString(ie)); return 2; }
          e->graph_ok = true;
          e->graph_gen = fluxmi_tuning_generation();
        }
      }
      Range r("frozen steps (hipGraph replay)");
      if (!t0_recorded) {
        FLUXMI_CHECK_HIP(hipEventRecord(e->ev_t0, s));
        first_timed = step;
        t0_recorded = true;
      }
      if (use_graph && e->graph_ok) {
        for (; step < win_end; ++step) FLUXMI_CHECK_HIP(hipGraphLaunch(e->exec, s));
      } else {
        for (; step < win_end; ++step) FLUXMI_TRY(one_step(s));
      }
    }
    FLUXMI_CHECK_HIP(hipEventRecord(e->ev_t1, s));
    e->timed_steps = t0_recorded ? n_steps - first_timed : 0;
  }
  FLUXMI_CHECK_HIP(hipMemcpyAsync(img, img_s, n_img * 2, hipMemcpyDeviceToDevice, s));
  *trial_index_inout = trial;
  return 0;
}

int fluxmi_engine_last_timing(fluxmi_engine_t* e, float* ms, int* steps) {
  FLUXMI_REQUIRE(e && ms && steps, "engine_last_timing: NULL argument");
  *ms = 0.f; *steps = e->timed_steps;
  if (e->timed_steps > 0) {
    FLUXMI_CHECK_HIP(hipEventSynchronize(e->ev_t1));
    FLUXMI_CHECK_HIP(hipEventElapsedTime(ms, e->ev_t0, e->ev_t1));
  }
  return 0;
}

int fluxmi_engine_set_amax_exchange(fluxmi_engine_t* e, float* amax_dev, int n, fluxmi_amax_hook_t hook, void* user) {
  FLUXMI_REQUIRE(e, "engine_set_amax_exchange: NULL engine");
  if (!hook) {
    e->amax_hook = nullptr; e->amax_user = nullptr; e->amax_ext = nullptr;
    e->d_amax = e->d_amax_own;
    return 0;
  }
  FLUXMI_REQUIRE(amax_dev && n >= (int)e->lin.size(), "engine_set_amax_exchange: need a device array of >= %d floats", (int)e->lin.size());
  e->amax_hook = hook; e->amax_user = user; e->amax_ext = amax_dev; e->d_amax = amax_dev;
  return 0;
}

// One block (kind 0 = DoubleStreamBlock `index`, 1 = SingleStreamBlock `index`, 2 = LastLayer), stages [stage_from, stage_to], on the engine's own
// buffers: x (and the stage's input buffer) hold whatever the caller put there with fluxmi_engine_copy_buffer, the modulation vectors
// are read from `mod`.  Test hook for teacher-forced per-layer parity; mode as in fluxmi_engine_forward (1 fused, 2 unfused-frozen).
int fluxmi_engine_run_block(fluxmi_engine_t* e, int kind, int index, int mode, int stage_from, int stage_to, void* stream) {
  FLUXMI_REQUIRE(e && e->ws, "engine_run_block: call fluxmi_engine_prepare first");
  FLUXMI_REQUIRE(mode == 1 || mode == 2, "engine_run_block: mode must be 1 (fused) or 2 (unfused, frozen scales)");
  SplitkScope splitk(e);
  FLUXMI_TRY(ensure_pairs(e, (hipStream_t)stream));
  if (kind == 2) {  // LastLayer: x (img rows) + the last 2H entries of `mod` -> the engine's own `pred_s` buffer
    FLUXMI_REQUIRE(index == 0 && stage_from >= 0 && stage_to <= 1 && stage_from <= stage_to, "engine_run_block: LastLayer has stages 0..1, index 0");
    return final_layer(e, buf<u16>(e, "pred_s"), stage_from, stage_to, (hipStream_t)stream);
  }
  const int nst = kind == 0 ? DOUBLE_STAGES : SINGLE_STAGES, nb = kind == 0 ? e->d.depth : e->d.depth_single;
  FLUXMI_REQUIRE((kind == 0 || kind == 1) && index >= 0 && index < nb, "engine_run_block: no block %d of kind %d", index, kind);
  FLUXMI_REQUIRE(stage_from >= 0 && stage_to < nst && stage_from <= stage_to, "engine_run_block: stages [%d, %d] out of range (0..%d)",
                 stage_from, stage_to, nst - 1);
  if (mode == 1) {
    FLUXMI_TRY(require_all_f8(e));
    FLUXMI_TRY(build_qluts(e, (hipStream_t)stream));
  }
  const Ctx ctx = make_ctx(e);
  return kind == 0 ? double_block(e, ctx, index, mode, 0, stage_from, stage_to, (hipStream_t)stream)
                   : single_block(e, ctx, index, mode, 0, stage_from, stage_to, (hipStream_t)stream);
}

// copy `bytes` between a named workspace buffer (at byte `offset`) and a caller DEVICE buffer; to_engine != 0 writes the workspace
int fluxmi_engine_copy_buffer(fluxmi_engine_t* e, const char* name, long long offset, void* dev_ptr, long long bytes, int to_engine,
                              void* stream) {
  FLUXMI_REQUIRE(e && name && dev_ptr, "engine_copy_buffer: NULL argument");
  auto it = e->bufs.find(name);
  FLUXMI_REQUIRE(it != e->bufs.end(), "engine_copy_buffer: no buffer named '%s'", name);
  FLUXMI_REQUIRE(offset >= 0 && bytes >= 0 && (size_t)(offset + bytes) <= it->second.n, "engine_copy_buffer: [%lld, +%lld) outside '%s' (%zu bytes)",
                 offset, bytes, name, it->second.n);
  char* p = (char*)it->second.p + offset;
  // the fp8 activation buffers are exchanged as PLAIN rows; the FUSED path keeps them in the row-pair layout (act_pairs): convert on the way
  // (whole row pairs only).  The unfused / calibrating modes use plain rows -- this hook serves the fused teacher-forced tests.
  long long ld = 0;
  const std::string nm(name);
  if (nm == "a8" || nm == "attn8") ld = e->d.hidden;
  else if (nm == "h8") ld = e->d.mlp_hidden;
  else if (nm == "cat8") ld = (long long)e->d.hidden + e->d.mlp_hidden;
  if (ld > 0 && act_pairs(e, true)) {
    FLUXMI_REQUIRE(offset % (2 * ld) == 0 && bytes % (2 * ld) == 0, "engine_copy_buffer: '%s' is kept in row pairs: offset / size must cover whole pairs of %lld-byte rows",
                   name, ld);
    return to_engine ? fluxmi_k_pair_rows(dev_ptr, p, (int)(bytes / ld), ld, (hipStream_t)stream)
                     : fluxmi_k_unpair_rows(p, dev_ptr, (int)(bytes / ld), ld, (hipStream_t)stream);
  }
  FLUXMI_CHECK_HIP(hipMemcpyAsync(to_engine ? (void*)p : dev_ptr, to_engine ? (const void*)dev_ptr : (const void*)p, (size_t)bytes,
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int fluxmi_engine_workspace_bytes(fluxmi_engine_t* e, long long* bytes) {
  FLUXMI_REQUIRE(e && bytes, "engine_workspace_bytes: NULL argument");
  *bytes = (long long)(e->ws_bytes + e->pairs_bytes + e->mods_all_bytes);  // workspace + row-pair weight copies + modulation table
  return 0;
}

int fluxmi_engine_get_buffer(fluxmi_engine_t* e, const char* name, void** ptr, long long* bytes) {
  FLUXMI_REQUIRE(e && name && ptr, "engine_get_buffer: NULL argument");
  auto it = e->bufs.find(name);
  FLUXMI_REQUIRE(it != e->bufs.end(), "engine_get_buffer: no buffer named '%s'", name);
  *ptr = it->second.p;
  if (bytes) *bytes = (long long)it->second.n;
  return 0;
}

}  // extern "C"
