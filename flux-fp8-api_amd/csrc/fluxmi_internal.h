// fluxmi -- internal launch descriptors shared by the kernels, the C-ABI layer and the engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fluxmi.h"

typedef fluxmi_gemm_group_t FluxmiGemmGroup;

// ---- weight prefetch riding on another launch -------------------------------------------------------------------------------------
// Inside the denoise step every F8Linear weight is read ONCE per step: 11.9 GB stream through the 256 MB Infinity Cache, so a GEMM's
// first tiles find their W panels in HBM and run their K loops ~20 % longer than the later tiles, whose panels another XCD has
// already pulled on chip (profiles/r04_gemm_persist.txt: in-step timeline 78 K -> 60 K cycles per tile; the back-to-back probe never
// shows it).  Launches that leave CUs idle -- attention's 432 workgroups = 1.69 rounds of the 256 CUs, the 216-tile GEMMs -- therefore
// carry a few EXTRA workgroups behind their own that do nothing but read the weights of the launches that follow (16 B per lane,
// results discarded): the lines land in the memory-side cache, where every XCD finds them.  Purely a hint: no output depends on it.
struct FluxmiPrefetch {
  const void* ptr[6];
  long long bytes[6];  // multiples of 16
  int n;               // ranges in use
  int wgs;             // extra workgroups the launch appends for it (0 = off)
};
// engine -> next launch that supports it (attention, the 256x256 one-tile-per-workgroup GEMMs); consumed by that launch.  Thread-local.
void fluxmi_set_prefetch(const FluxmiPrefetch* pf);
FluxmiPrefetch fluxmi_take_prefetch();

#define FLUXMI_MAX_GROUPS 16
struct FluxmiGemmParams {
  FluxmiGemmGroup g[FLUXMI_MAX_GROUPS];
  int n_groups, N, K, epi, tiles_m_total, group_m;
  // split-K (small-M launches, gemm_pp.hip): `split_k` workgroups per tile, each over its own K range, fp32 partial tiles in
  // partial[split][tiles_m_total * 256][N]; fluxmi_launch_splitk_reduce sums them in split order and applies the epilogue
  int split_k;
  float* partial;
  unsigned long long* dbg;  // per-tile timestamps of the persistent kernel's timing build (tile config 19), else null
  FluxmiPrefetch pf;        // weights of later launches, read by pf.wgs extra workgroups behind the tiles (see FluxmiPrefetch)
  // implicit 3x3 convolution (round 6, fluxmi_conv3x3 -> the 128x128 tile kernel only; C == 0: off): the A operand is never materialised --
  // row r of group 0 is output pixel (b, yo, xo) of a [B, Ho, Wo] grid and K index (tap, c) reads x[b, (yo*stride + dy - pad) >> rshift,
  // (xo*stride + dx - pad) >> rshift, c] of the NHWC input g[0].A (zero outside [0, Hi << rshift)): what fluxmi_im2col3x3 would have written
  struct { const void* zeros; int Hi, Wi, C, Ho, Wo, stride, pad, rshift; } conv;
};

// ---- batched skinny GEMV (modulations + embedders, M = batch <= 8) ----------------------------
struct FluxmiGemvLayer {
  const void* W;           // [N,K] fp8 or bf16
  const void* bias;        // bf16 [N] or nullptr
  const float* in_scale;   // fp8: device scalar input_scale
  const float* sa_recip;   // fp8
  const float* sb_recip;   // fp8
  void* out;               // bf16, row b at out + b*ld_out
  const void* x;           // bf16 [B,K] (row stride ldx)
  long long ld_out, ldx;
  int N, K;
  int w_fp8;               // 1: fp8 weights (x quantised on the fly), 0: bf16 weights
  int pre_silu;            // apply SiLU (rounded to bf16) to x first        flux_model.py:252,155,496
  int blk_start;           // first block id of this layer (filled by launcher)
  int act_fmt;
};

struct FluxmiCalibLayer { float* trials; float* scale; float* recip; };

// ---- misc ------------------------------------------------------------------------------------
void fluxmi_set_error(const char* fmt, ...);
#define FLUXMI_CHECK_HIP(expr)                                                            \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      fluxmi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                           \
    }                                                                                     \
  } while (0)
#define FLUXMI_REQUIRE(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      fluxmi_set_error(__VA_ARGS__);         \
      return 1;                              \
    }                                        \
  } while (0)
#define FLUXMI_LAUNCH_CHECK() FLUXMI_CHECK_HIP(hipGetLastError())
#define FLUXMI_TRY(expr)        \
  do {                          \
    int _rc = (expr);           \
    if (_rc) return _rc;        \
  } while (0)

// ---- kernel-selection knobs (tuning.cpp): a copy of the process-wide struct, and the counter fluxmi_set_tuning bumps
fluxmi_tuning_t fluxmi_tuning();
unsigned fluxmi_tuning_generation();
void fluxmi_log_tuning(const char* why);

// ---- internal launchers (defined in the .hip files) --------------------------------------------
int fluxmi_gemm_tile_ok(int N, int K, int is_fp8, int cfg);
int fluxmi_launch_gemm_w1_192(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s);  // tile config 17 (gemm_w1.hip)
int fluxmi_launch_gemm_w1_224(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s);  // tile config 20 (gemm_w1.hip)
int fluxmi_launch_gemm_w1_160(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s);  // tile config 21 (gemm_w1.hip)
int fluxmi_gemm_tile_bn(int cfg);
int fluxmi_gemm_tile_bm(int cfg);
int fluxmi_launch_gemm(FluxmiGemmParams& p, int is_fp8, int act_fmt, int tile_cfg, hipStream_t s);
int fluxmi_launch_gemm_generic(FluxmiGemmParams& p, int is_fp8, int act_fmt, hipStream_t s);
const void* fluxmi_zero_page();  // 256 zero bytes on the current device (vae.hip; allocated once per device, never under stream capture)
int fluxmi_launch_gemm_conv(FluxmiGemmParams& p, hipStream_t s);  // bf16, tile config 2 with the implicit 3x3 gather (p.conv filled by the caller)
int fluxmi_gemm_auto_cfg(const FluxmiGemmParams& p, int is_fp8);
// persistent 256x256 ping-pong kernel (gemm_persist.hip, tile config 18; 19 = with per-tile timestamps)
int fluxmi_gemm_persist_ok(const FluxmiGemmParams& p, int is_fp8, int act_fmt);
int fluxmi_launch_gemm_persist(FluxmiGemmParams& p, int is_fp8, int act_fmt, int timing, hipStream_t s);
// 256x256 ping-pong tiles with `split_k` K ranges per tile + the reduce / epilogue pass (BF16 and GATE_RESID epilogues)
void fluxmi_gemm_block_splitk(int on);  // +1 / -1: no M-dependent split-K choice while > 0 (thread-local)
int fluxmi_launch_gemm_splitk(FluxmiGemmParams& p, int is_fp8, int act_fmt, int split_k, hipStream_t s);
// split-K partial-tile scratch of the calling thread's launches (thread-local; nullptr = the library's own per-(device, stream) buffer, which is
// allocated at the first EAGER use on that stream and refuses to appear under stream capture).  An engine owns one and sets it around its launches:
// its step graph is captured on a private stream and replayed on the caller's, so the scratch must belong to the engine, not to a stream.
constexpr size_t FLUXMI_SPLITK_WS_BYTES = (size_t)256 << 20;
void fluxmi_set_splitk_scratch(float* p);
// the batch of the calling thread's engine (1 = none): bf16 launches take the split-K decision of ONE sample's groups, so a sample's bits do not
// follow the batch it rides in (api.cpp)
void fluxmi_gemm_set_batch(int B);
// scratch of attention's balanced grid (attention2.hip, AttnSplit: partial softmax states of the key bins + arrival counters, which must be
// ZERO when handed over and are left zero by every launch): thread-local like the split-K scratch, an engine owns one; nullptr = the library's own
// per-(device, stream) buffer.  fluxmi_attn_plan_any: does a balanced-grid plan exist for one sample of this shape (any tuning)?
constexpr size_t FLUXMI_ATTN_SPLIT_WS_BYTES = (size_t)8 * 64 * (8 * 17 * 64 * 4) * 4 + 8 * 64 * 4;
void fluxmi_set_attn_scratch(void* p);
int fluxmi_attn_plan_any(int B, int L, int H);
int fluxmi_attn_debug_buffer(void* dev_u64);  // fluxmi_attention_debug_buffer
int fluxmi_attn_plan_export(int B, int L, int H, int* n_per_x, int* full_per_x, int* npieces, unsigned long long* pieces);  // fluxmi_attention_plan  // the same for ANY tuning (what an engine sizes its workspace by: the knob may change later)
// tile choice + (when it pays) the split of a grouped launch into a 256x256 and a 128x128 launch; any number of groups
int fluxmi_gemm_dispatch(const FluxmiGemmGroup* gs, int n, int N, int K, int is_fp8, int act_fmt, int epi, hipStream_t s);
int fluxmi_launch_gemv(const FluxmiGemvLayer* layers_dev, FluxmiGemvLayer* layers_host, int n_layers, int B, int total_blocks,
                       int max_K, hipStream_t s, int row0 = 0);
int fluxmi_gemv_blocks(const FluxmiGemvLayer* layers_host, int n_layers);

int fluxmi_k_quantize_act(const void* x, void* q, const float* scale, int rows, int cols, long long ld_in, long long ld_out, int fmt, hipStream_t s);
int fluxmi_k_amax(const void* x, float* amax, int rows, int cols, long long ld, hipStream_t s);
int fluxmi_k_calib_update(const float* amax, float* trials, float* scale, float* recip, int trial_index, int num_trials, float max_val, hipStream_t s);
int fluxmi_k_calib_update_many(const float* amax, const void* layers_dev, int n_layers, int trial_index, int num_trials, float max_val, hipStream_t s);
int fluxmi_k_quantize_weight(const void* w_bf16, void* q, float* amax_tmp, float* scale, float* recip, int N, int K, int fmt, hipStream_t s);
int fluxmi_k_dequant(const void* q, float* out, const float* recip, long long n, int fmt, hipStream_t s);
int fluxmi_k_requantize_f32(const float* w32, void* q, float* amax_tmp, float* scale, float* recip, long long n, int fmt, hipStream_t s);
int fluxmi_k_lora_delta(const float* Bm, const float* A, float* delta, int N, int K, int R, float scale, int accumulate, hipStream_t s);
int fluxmi_k_axpy_f32(float* w, const float* d, float alpha, long long n, hipStream_t s);
int fluxmi_k_ln_modulate(const void* x, long long ldx, long long x_bstride, void* out, long long ldo, long long out_bstride,
                         const void* shift0, const void* scale0, const void* shift1, const void* scale1, long long mod_bstride,
                         const float* q0, const float* q1, int B, int L, int split, int H, int out_fp8, int fmt, hipStream_t s, int out_pairs = 0);
int fluxmi_k_act(const void* x, void* y, int rows, int cols, long long ld_in, long long ld_out, int mode, hipStream_t s);
int fluxmi_k_gate_residual(const void* x, const void* y, const void* gate, void* out, int B, int L, int H, long long ldx,
                           long long ldy, long long ldo, long long gate_bstride, hipStream_t s);
int fluxmi_k_add(const void* a, const void* b, void* z, long long n, hipStream_t s);
int fluxmi_k_build_qlut(const float* scale, int fmt, int act, void* lut, hipStream_t s);
int fluxmi_k_pair_rows(const void* in, void* out, int rows, long long row_bytes, hipStream_t s);
int fluxmi_k_unpair_rows(const void* in, void* out, int rows, long long row_bytes, hipStream_t s);  // the inverse
int fluxmi_k_add_bcast(const void* a, const void* b, void* z, long long rows, int nb, int cols, hipStream_t s);
int fluxmi_k_select_step(const void* table, const int* step, const int* step0_dev, void* dst, long long bytes, hipStream_t s);
int fluxmi_k_timestep_rows(void* t_rows, const float* ts, int step0, int B, int R, hipStream_t s);
int fluxmi_k_fill_bf16(void* dst, float v, int n, hipStream_t s);
int fluxmi_k_timestep_embedding(const void* t, const float* freqs, void* out, int B, int half, float time_factor, hipStream_t s);
int fluxmi_k_rope_table(const void* ids, const float* omega, const int* axis, void* pe, long long rows, int n_axes, int pairs, hipStream_t s);
int fluxmi_k_euler(void* img, const void* pred, const float* dts, const int* step, long long n, hipStream_t s);
int fluxmi_k_set_timestep(void* t_vec, const float* ts, const int* step, int B, hipStream_t s);
int fluxmi_k_advance_step(int* step, hipStream_t s);
int fluxmi_k_clock_sample(unsigned long long* out2, hipStream_t s);
int fluxmi_k_im2col3x3(const void* x, void* col, int B, int H, int W, int C, int up, hipStream_t s);
int fluxmi_k_groupnorm(const void* x, const void* gamma, const void* beta, void* y, float* work, int B, int P, int C, int swish, float eps,
                       hipStream_t s);
int fluxmi_k_softmax_rows(const void* S, void* P, int rows, int cols, long long ld, float scale, hipStream_t s);
int fluxmi_k_row_norm(const void* x, const void* w, const void* b, void* y, int rows, int D, long long ldx, long long ldy, float eps, int mode,
                      hipStream_t s);
int fluxmi_k_act_mul(const void* in, void* out, int rows, int F, long long ld_in, long long ld_out, int mode, hipStream_t s);
int fluxmi_k_text_attention(const void* q, const void* k, long long ld_qk, const void* vt, long long ld_vt, void* out, long long ld_out,
                            const float* rel_bias, int bias_ld, const void* v_bias, float scale, int causal, int L, int Lp, int H, hipStream_t s);
int fluxmi_k_qkv_rope(const void* qkv, long long ld, const void* pe, const void* q_scale0, const void* k_scale0,
                      const void* q_scale1, const void* k_scale1, void* Q, void* K, void* VT, int B, int L, int Lp, int H,
                      int split, int k_f16, hipStream_t s);
int fluxmi_k_attention(const void* Q, const void* K, const void* VT, void* out, long long ld_out, int col_off, int out_fp8,
                       const float* q_scale0, const float* q_scale1, int split, int B, int L, int Lp, int H, int fmt, hipStream_t s,
                       const void* qraw = nullptr, long long ldq = 0, const void* pe = nullptr, const void* qn0 = nullptr,
                       const void* qn1 = nullptr, int k_f16 = 0, int out_pairs = 0);
