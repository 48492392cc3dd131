/* fluxmi.h -- C ABI of libfluxmi.so, the MI355X (gfx950) implementation of the Flux denoise hot path
 * of aredden/flux-fp8-api.
 *
 * The reference has no FFI of its own: its only operator-swap mechanism is Python module replacement
 * (float8_quantize.py:320-392 swaps nn.Linear -> F8Linear / the external cublas_ops.CublasLinear).
 * This header is therefore the boundary a maintainer binds from the reference's Python operator
 * classes (ctypes stub in INTEGRATION.md).  Each entry point cites the reference code it replaces
 * (paths relative to the reference repo root).
 *
 * Conventions
 *   - plain C types only: raw DEVICE pointers, sizes, a hipStream_t passed as void* (0 = null stream)
 *   - every function returns 0 on success; non-zero -> call fluxmi_last_error() (thread-local string)
 *   - nothing throws across the ABI; no allocation and no host synchronisation inside any op call, fluxmi_engine_forward or
 *     fluxmi_engine_run_block (all are hipGraph-capturable).  Allocation happens in fluxmi_engine_create (constants, pinned
 *     staging, events) and fluxmi_engine_prepare (workspace + the step-ahead modulation table) only.  fluxmi_engine_denoise
 *     allocates nothing and never waits on the stream; its only host waits are (a) on the event of the PREVIOUS request's schedule
 *     upload before the pinned staging buffer is rewritten (complete long before, in practice never blocks) and (b) one
 *     hipStreamSynchronize the first time a shape is seen, immediately before the hipGraph capture
 *   - multi-GPU: the library holds no communicator.  The path shards by batch with no data-path collective, so the
 *     three small exchanges (embeddings + noise broadcast, per-layer amax MAX during calibration, latent gather) are made by the
 *     host with torch.distributed over RCCL (flux-fp8-api_amd/fluxmi/dist.py); the engine exposes the one hook they need that
 *     lies INSIDE a forward pass: fluxmi_engine_set_amax_exchange
 *   - caller owns every buffer passed in; row strides ("ld") are in ELEMENTS of that buffer
 *   - tensors are bf16 (uint16 storage) unless stated; fp8 tensors are OCP e4m3fn / e5m2 bytes
 *   - an engine handle is not re-entrant: the host wrapper serialises calls per engine
 */
#ifndef FLUXMI_H
#define FLUXMI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FLUXMI_ABI_VERSION 5

/* fp8 format codes (torch.float8_e4m3fn / torch.float8_e5m2, float8_quantize.py:39,43) */
#define FLUXMI_E4M3 0
#define FLUXMI_E5M2 1

/* GEMM epilogues: what happens to h = bf16(acc * sa * sb + bias) */
enum {
  FLUXMI_EPI_BF16 = 0,       /* C(bf16) = h                                   float8_quantize.py:284-292 */
  FLUXMI_EPI_GELU_QUANT = 1, /* C(fp8)  = q(bf16(gelu_tanh(h)), q_scale)      flux_model.py:301,335 + float8_quantize.py:274-276 */
  FLUXMI_EPI_GATE_RESID = 2, /* C(bf16) = bf16(resid + bf16(gate[n] * h))     flux_model.py:387-396,484 */
  FLUXMI_EPI_SPLIT = 3,      /* n < split_n: C(bf16) = h ; else C2(fp8) = q(gelu) at column c2_col0 + n - split_n
                                                                              flux_model.py:471-480 */
  FLUXMI_EPI_QUANT = 4,      /* C(fp8)  = q(h, q_scale) */
  FLUXMI_EPI_SILU_QUANT = 5  /* C(fp8)  = q(bf16(silu(h)), q_scale)           flux_model.py:154-155 */
};

/* one problem of a (grouped) linear: out[M,N] = epilogue(A[M,K] . W[N,K]^T) */
typedef struct fluxmi_gemm_group {
  const void* A;          /* activations [M,K]: fp8 bytes (F8Linear) or bf16 (nn.Linear) */
  const void* W;          /* weight [N,K] row-major = F8Linear.float8_data / nn.Linear.weight */
  const void* bias;       /* bf16 [N] or NULL */
  const float* sa_recip;  /* device scalar F8Linear.input_scale_reciprocal (NULL = 1) */
  const float* sb_recip;  /* device scalar F8Linear.scale_reciprocal       (NULL = 1) */
  void* C;                /* primary output */
  void* C2;               /* secondary output (FLUXMI_EPI_SPLIT) */
  const void* gate;       /* bf16 [N] */
  const void* resid;      /* bf16 [M, ldr] (may alias C) */
  const float* q_scale;   /* device scalar: input_scale of the consuming F8Linear */
  long long lda, ldc, ldc2, ldr;
  int M;
  int m_tile_start;       /* internal, filled by the launcher */
  int split_n, c2_col0;
  /* Optional fused attention-layout outputs (FLUXMI_EPI_BF16 / FLUXMI_EPI_SPLIT, 256x256 LDS-epilogue tile configs 13 and 16 only;
   * all NULL / 0 = off).  The N columns are [q | k | v | ...] with heads*128 columns each (the qkv Linear of a DoubleStreamBlock,
   * or the first 3*hidden columns of SingleStreamBlock.linear1).  V columns are written TRANSPOSED into vt_out
   * [heads*128][vt_ld] (key position = tok0 + row, key order inside every 16-key group bit2<->bit3 swapped = the k-slot order of
   * the attention kernel's PV MFMA; positions tok0+M .. tok0+vt_rows-1 are zero filled) instead of into C; K columns are
   * RMS-normalised (k_norm), rotated (pe) and written to k_out [heads][k_rows][128] instead of into C (tile configs 13, 16 and the
   * persistent config 18, which does it at a fraction of the cost: csrc/gemm_persist.hip).
   * Replaces the V / K halves of fluxmi_qkv_rope.                       flux_model.py:351-354,158-176,60-65,380-382 */
  void* vt_out;
  void* k_out;
  const void* pe;         /* (cos, sin) bf16 [k_rows][64][2] of this batch element */
  const void* k_norm;     /* bf16 [128] */
  long long vt_ld;
  int k_rows, tok0, vt_rows, kv_col0, heads;
  int k_f16;              /* k_out holds fp16 instead of bf16 (exact for bf16 values in fp16's range: fluxmi_attention's folded QK^T) */
  /* Optional 64 KiB table for the quantising epilogues (FLUXMI_EPI_GELU_QUANT and the mlp columns of FLUXMI_EPI_SPLIT, tile config
   * 13): q_lut[b] = the fp8 byte the epilogue would compute for the bf16 GEMM output with bit pattern b -- bf16 -> GELU -> bf16 ->
   * x input_scale -> bf16 -> clamp -> fp8 is a pure function of those 16 bits once the scale is frozen.  Built by
   * fluxmi_build_quant_lut from the same helpers; measured over all 65536 inputs (profiles/r06_qlut_vs_computed.txt) the table equals the
   * oracle's torch chain on every finite input outside a 4-pattern fp32-tanh cliff, and the epilogue the kernels COMPUTE when no table is
   * given differs from the table on ONE pattern (hipcc contracts the GELU polynomial differently per kernel) -- the table is the default
   * and the more faithful of the two; latents of a qlut = 0 run drift from the default's like two fp8 runs do.  It replaces ~25 VALU
   * instructions per element by one LDS gather while the matrix pipe is idle.  NULL = compute (tile configs 2, 13, 16; the persistent
   * config 18 has no computed quantising epilogue).                       flux_model.py:301,480 + float8_quantize.py:217-218,274-276 */
  const void* q_lut;
  /* Optional copy of W in the ROW-PAIR layout [N/2][K_bytes/64][2][64] (fluxmi_pair_rows): the 64-byte K-steps of rows 2r and 2r+1 share one
   * 128-byte line.  The tiled kernels fetch an operand one 64-byte K-step at a time, i.e. HALF an L2 line per row and step -- each line crosses the
   * L2 -> CU path twice per tile; with the weight stored in row pairs every line of W is fetched once (profiles/r04_gemm_persist.txt section 9: the
   * K loop of the persistent kernel 62.4 K -> 56.5 K cycles per tile).  Same values, same results; NULL = read W.  Honoured by tile configs 18
   * (all groups of a launch with or all without) and 16; the engine keeps such a copy of the weights those launches read (+8 GB at Flux-dev). */
  const void* W_pairs;
  /* ACTIVATIONS in the row-pair layout (ABI 5, round 6).  a_pairs != 0: A (fp8, dense rows: lda == K, M % 2 == 0) is stored as
   * [M/2][K/64][2][64] like W_pairs -- the 64-byte K-steps of rows 2r and 2r + 1 share one 128-byte line, so every L2 line of the A panel
   * crosses to the CU once per tile instead of twice (in-step, timing-only ablation of the persistent kernel alone: -1.3 % per step,
   * profiles/r06_act_pairs.txt).  c8_pairs != 0: the fp8 output of the QUANTISING epilogues (C for FLUXMI_EPI_GELU_QUANT / _QUANT /
   * _SILU_QUANT, C2 for FLUXMI_EPI_SPLIT) is written in that layout over rows of ldc (ldc2) bytes: byte (m, col) of the row-major buffer
   * lives at (m / 2) * 2 * ld + (m % 2) * 64 + (col / 64) * 128 + col % 64 (ld % 64 == 0; col = the column inside the FULL row, i.e.
   * c2_col0 + n - split_n for SPLIT) -- the next F8Linear reads it with a_pairs.  Same values, same results; honoured by tile configs 2,
   * 13, 16, 17, 18 (every group of a launch with the same flags); the generic and split-K kernels refuse a flagged group.  The engine
   * keeps its fp8 activation buffers this way in fused mode (fluxmi_tuning_t.a_pairs). */
  int a_pairs;
  int c8_pairs;
} fluxmi_gemm_group_t;

const char* fluxmi_last_error(void);
int fluxmi_abi_version(void);

/* ---- kernel-selection knobs (ABI 3; attn_split: ABI 4; a_pairs: ABI 5) --------------------------------------------------------------------------------------------
 * The reference has no counterpart (its only switches are the ModelSpec flags of util.py:40-77); these choose between kernels /
 * fusion levels that compute the same results.  They are resolved ONCE: the first call that needs a knob parses the FLUXMI_*
 * environment variables named below into this struct (csrc/tuning.cpp -- the only getenv site of the library);
 * fluxmi_set_tuning replaces it at run time (A/B probes, tests).  An engine re-captures its step graph when the struct changed
 * since the capture, and fluxmi_engine_create logs it once under FLUXMI_LOG=1. */
typedef struct fluxmi_tuning {
  int struct_size;       /* sizeof(fluxmi_tuning_t): filled by fluxmi_get_tuning, checked by fluxmi_set_tuning */
  int gemm_cfg;          /* FLUXMI_GEMM_CFG     -1 = cost model (default), else force this tile config where it applies */
  int gemm_splitk;       /* FLUXMI_GEMM_SPLITK   1: small-M bf16 launches split K over several workgroups per tile */
  int gemm_hybrid;       /* FLUXMI_GEMM_HYBRID   1: peel the thin groups of a grouped launch into a 128x128 launch */
  int gemm_esel;         /* FLUXMI_GEMM_ESEL     1: one kernel instantiation per hot epilogue (0 = run-time switch, A/B) */
  int gemm_persist;      /* FLUXMI_GEMM_PERSIST  1: multi-round fp8 launches on the persistent kernel (tile config 18); 2 = its timing build (probes) */
  int attn_var;          /* FLUXMI_ATTN_VAR      bit 1: exact instead of deferred running max */
  int attn_abl;          /* FLUXMI_ATTN_ABL      ablation bits of the 8-wave kernel (probes) */
  float attn_defer_log2; /* FLUXMI_ATTN_THR      rescale threshold of the deferred running max, log2; [0, 16], default 8 */
  int attn_f16k;         /* FLUXMI_ATTN_F16K     1: the engine stores K as fp16 and runs the folded attention arithmetic */
  int fuse_kv;           /* FLUXMI_FUSE_KV       0 / 1 / 2 (default): K, V^T by the relayout kernel / V^T from the qkv GEMM epilogue / both */
  int qlut;              /* FLUXMI_QLUT          1: table-driven GELU -> fp8 epilogues */
  int ln_variant;        /* FLUXMI_LN_V          2 = streaming LayerNorm kernel (default), 3 = the same at two workgroups per CU, 1 = one wave per row */
  int roctx;             /* FLUXMI_ROCTX         1: roctx ranges around the phases of a denoise call */
  int prefetch;          /* FLUXMI_PREFETCH      1: launches with idle CUs (attention, the 216-tile GEMMs) carry extra workgroups that read the
                                                 weights of the following launches into the memory-side cache (engine, fused mode); 2: a double
                                                 block's attention also pulls in mlp.2, 3: a single block's attention also the next block's linear1
                                                 (in-step -0.1 .. -0.7 % by lease for 2 and 3, +0.3 % once: inside the noise, default stays 1) */
  int w_pairs;           /* FLUXMI_W_PAIRS       1: the engine keeps a row-pair copy of the F8Linear weights its persistent GEMM launches read
                                                 (fluxmi_gemm_group_t.W_pairs: every L2 line of W fetched once per tile instead of twice) */
  int log;               /* FLUXMI_LOG           1: print the struct to stderr when it is resolved / set / an engine is created */
  int gemm_tile192;      /* FLUXMI_GEMM_TILE192  1 (default): gate*y+x launches of the one-wave-per-SIMD kernel (K >= 8192) whose 256-row tiles fill less
                                                 than one round of the CUs run on 192 x 256 tiles (tile config 17; same bits, 36 % more tiles of
                                                 three quarters the work: Flux-dev 768^2 mlp.2 / linear2, 132 -> 180 tiles); round 6: also 224-row
                                                 (config 20: 1024^2 linear2, 216 -> 252 tiles) and 160-row tiles (config 21: 768^2, 216 tiles), the
                                                 four waves side by side along N; 2 = config 17 only (A/B), 0 = 256-row tiles only */
  int attn_split;        /* FLUXMI_ATTN_SPLIT    1 (default): attention launches whose last round of workgroups is THIN (at most 8 of an XCD's
                                                 32 CUs busy: 264 tasks on 256 CUs at Flux-dev 768^2) run that round's tasks as pieces of
                                                 their key range and merge the partial softmax states (fp32 log-sum-exp in a fixed order:
                                                 deterministic); see fluxmi_attention_plan.  2 = wherever such a plan exists (fuller last
                                                 rounds: measured not to pay, Flux-dev 1024^2 +3.4 % per step), 0 = one workgroup per task */
  int a_pairs;           /* FLUXMI_A_PAIRS       1 (default): in fused mode the engine keeps its fp8 ACTIVATION buffers (LayerNorm / attention /
                                                 GELU outputs = the A operands of the block linears) in the row-pair layout
                                                 (fluxmi_gemm_group_t.a_pairs / c8_pairs; even L and Lt only) -- same bits, fewer L2 lines */
} fluxmi_tuning_t;
int fluxmi_get_tuning(fluxmi_tuning_t* out);
int fluxmi_set_tuning(const fluxmi_tuning_t* in); /* validates every field (non-zero + fluxmi_last_error on a bad value) */

/* Probes (tools/): a device buffer of [workgroup][tile < 8][8] uint64 that the timing build of the persistent GEMM (tile config 19)
 * fills with {tile start, K loop end, epilogue end} shader-clock stamps, the 100 MHz real-time counter and four phase stamps of the
 * table epilogue; NULL switches it off.
 * fluxmi_clock_sample writes {XCC id, s_memtime, s_memrealtime} of one wave per XCD to out[0..23] (24 uint64) on `stream`: two samples
 * around a timed region, paired by XCC id, give the average shader clock the chip sustained over it (bench.py). */
int fluxmi_gemm_debug_buffer(void* dev_u64);
int fluxmi_clock_sample(void* out24_dev_u64, void* stream);

/* ---- F8Linear / Linear ------------------------------------------------------------------------- */
/* Grouped linear.  is_fp8=1: A is `act_fmt` fp8, W is e4m3fn (torch._scaled_mm, float8_quantize.py:284-292);
 * is_fp8=0: A, W bf16 (F.linear).  tile_cfg: -1 auto (cost model + split of a thin last round, what the engine uses);
 * 13 = 256x256 ping-pong LDS ring (K*bytes % 64 == 0); 16 = 256x256 with one wave per SIMD (K*bytes % 256 == 0), 17 = the same kernel on 192x256 tiles
 * (fp8 x e5m2 or bf16 operands, plain bf16 or gate*y+x epilogue: launches with a thin single round of 256-row tiles); 2 = 128x128 and 15 = 128x64
 * double-buffered tiles (K*bytes % 128 == 0); 100 = generic any-shape kernel (scalar fma: <= 1 bf16 ulp of the others on bf16 operands).  The
 * tiled configs compute the same bits for fp8 AND, since round 6, for bf16 operands (one K association: test_bf16_tile_configs_are_bit_identical).  (Other numbers
 * named kernel generations that were removed: they are rejected.)  18 = config 13 as a PERSISTENT kernel (one workgroup per CU walks the tiles;
 * fp8 x e5m2, N % 256 == 0, K % 256 == 0, K >= 512; the auto dispatch takes it for launches of more than 256 tiles), 19 = its timing build.
 * 113 + S (S = 2..32, S <= K-steps): config 13 with SPLIT-K -- S workgroups per tile, each over its own K range, fp32 partial tiles in a 256 MiB
 * scratch summed in ascending K order by a second pass that applies the epilogue (FLUXMI_EPI_BF16 / FLUXMI_EPI_GATE_RESID only).  The scratch
 * belongs to the (device, stream) pair of the launch: allocated, under a lock, by the first EAGER split-K launch on that stream -- a first use
 * under stream capture is refused -- so launches on different streams never share partial tiles; a fluxmi_engine owns one of its own (its step
 * graph is captured on a private stream).  The auto dispatch uses split-K for bf16 launches of <= 128 tiles with >= 192 K-steps (M <= 512:
 * Flux-schnell at 256x256, the text encoders): deterministic, <= 1 bf16 ulp of fp64 like the others, but not bit-identical to the unsplit
 * kernels (the fp32 sum is associated differently) -- i.e. the bits of a bf16 auto-dispatched GEMM called through THIS entry depend on how many
 * rows share the launch.  A fluxmi_engine announces its batch to the dispatcher, which then takes the slice count of ONE sample's groups, so a
 * sample's bits do not follow the batch it rides in (round 6; test_a_sample_does_not_depend_on_its_batch).  Other callers that need
 * batch-invariant bits launch a fixed number of rows (the native text encoders run one prompt per launch; the engine's modulation-table GEMM
 * blocks the choice) or set fluxmi_tuning_t.gemm_splitk = 0. */
int fluxmi_gemm_grouped(const fluxmi_gemm_group_t* groups, int n_groups, int N, int K, int is_fp8, int act_fmt,
                        int epilogue, int tile_cfg, void* stream);
/* single-problem convenience form of the above (F8Linear.forward after quantisation) */
int fluxmi_f8_gemm(const void* a_fp8, const void* w_e4m3, const float* sa_recip, const float* sb_recip, const void* bias,
                   void* out, int M, int N, int K, int act_fmt, int epilogue, const void* gate, const void* resid,
                   const float* q_scale, int tile_cfg, void* stream);
/* skinny linear (M = batch <= 8): out[b,n] = bf16((x_q[b,:].W[n,:]) * sa*sb + bias), optional SiLU on x first
 * (Modulation.forward flux_model.py:251-257, MLPEmbedder flux_model.py:154-155, LastLayer.adaLN flux_model.py:495-500) */
int fluxmi_gemv(const void* x, long long ldx, const void* W, const void* bias, const float* in_scale, const float* sa_recip,
                const float* sb_recip, void* out, long long ld_out, int B, int N, int K, int w_fp8, int act_fmt, int pre_silu,
                void* stream);

/* ---- quantisation state machine (F8Linear) ------------------------------------------------------- */
/* q = fp8(clamp(bf16(x*scale)))                                                float8_quantize.py:217-218,274-276 */
int fluxmi_quantize_act(const void* x, void* q, const float* scale, int rows, int cols, long long ld_in, long long ld_out,
                        int fmt, void* stream);
/* *amax = max(*amax, max|x|)  (caller zeroes *amax first)                      float8_quantize.py:227 */
int fluxmi_amax(const void* x, float* amax, int rows, int cols, long long ld, void* stream);
/* one call of F8Linear.quantize_input's scale logic for trial `trial_index`     float8_quantize.py:220-246 */
int fluxmi_calib_update(const float* amax, float* trials, float* scale, float* scale_recip, int trial_index, int num_trials,
                        float max_val, void* stream);
/* F8Linear.quantize_weight: amax -> scale -> e4m3 data + reciprocal             float8_quantize.py:195-207 */
int fluxmi_quantize_weight(const void* w_bf16, void* q, float* amax_tmp, float* scale, float* scale_recip, int N, int K, int fmt,
                           void* stream);
/* LoRA fuse into an fp8 weight: W' = requant(bf16(dequant(W) + scale * B@A))     lora_loading.py:509-577,615-631,686-687 */
int fluxmi_lora_fuse_f8(void* w_fp8, float* w_scale, float* w_scale_recip, const float* lora_B, const float* lora_A, int N, int K,
                        int R, int n_chunks, float lora_scale, float* work_f32, float* amax_tmp, void* stream);
int fluxmi_dequant(const void* q, float* out, const float* scale_recip, long long n, int fmt, void* stream);

/* ---- block elementwise ----------------------------------------------------------------------------- */
/* y = (1+scale)*LayerNorm(x)+shift (+fp8 quantise); rows [B][L], rows l<split use set 0   flux_model.py:367-368,389,469-470,501 */
int fluxmi_ln_modulate(const void* x, long long ldx, void* out, long long ldo, const void* shift0, const void* scale0,
                       const void* shift1, const void* scale1, long long mod_bstride, const float* q_scale0,
                       const float* q_scale1, int B, int L, int split, int H, int out_fp8, int fmt, void* stream);
/* mode 0: GELU(tanh), 1: SiLU (bf16 -> bf16)                                     flux_model.py:301,139 */
int fluxmi_act(const void* x, void* y, int rows, int cols, long long ld_in, long long ld_out, int mode, void* stream);
/* out = x + gate[b] * y                                                           flux_model.py:387-396,484 */
int fluxmi_gate_residual(const void* x, const void* y, const void* gate, void* out, int B, int L, int H, long long ldx,
                         long long ldy, long long ldo, long long gate_bstride, void* stream);
int fluxmi_add(const void* a, const void* b, void* z, long long n, void* stream);

/* lut[b] for every bf16 bit pattern b: the fp8 byte of  to_fp8_saturated(act(bf16 b), *scale)  (act: 0 none, 1 gelu-tanh, 2 silu),
 * rounded at the same points as the fused epilogues.  65536 bytes.                      float8_quantize.py:217-218 */
int fluxmi_build_quant_lut(const float* scale, int fmt, int act, void* lut, void* stream);
/* rows x row_bytes (row-major, rows % 2 == 0, row_bytes % 64 == 0) -> the row-pair layout of fluxmi_gemm_group_t.W_pairs; out != in */
int fluxmi_pair_rows(const void* in, void* out, int rows, long long row_bytes, void* stream);
/* the inverse: the row-pair layout back to plain rows (ABI 5; tests, fluxmi_engine_copy_buffer) */
int fluxmi_unpair_rows(const void* in, void* out, int rows, long long row_bytes, void* stream);

/* ---- attention path --------------------------------------------------------------------------------- */
/* pe[rows, pairs, (cos,sin)] from position ids                                    flux_model.py:49-57,82-92 */
int fluxmi_rope_table(const void* ids, const float* omega, const int* axis, void* pe, long long rows, int n_axes, int pairs,
                      void* stream);
/* qkv split + QKNorm + RoPE + head-major relayout (V transposed); Q may be NULL      flux_model.py:351-354,158-176,60-65,380-382
 * k_f16 != 0: K is stored as fp16 instead of bf16 (same bytes per element; the normalised + rotated bf16 values are exact in fp16
 * unless |k| < 6.1e-5).  That is the operand format of the attention kernel's folded schedule, see fluxmi_attention. */
int fluxmi_qkv_rope(const void* qkv, long long ld, const void* pe, const void* q_scale0, const void* k_scale0,
                    const void* q_scale1, const void* k_scale1, void* Q, void* K, void* VT, int B, int L, int Lp, int H, int split,
                    int k_f16, void* stream);
/* softmax(QK^T/sqrt(128))V -> [B,L,H*128] (bf16, or fp8 with the consumer's input scale)   flux_model.py:41-45
 * k_f16 != 0: K holds fp16 (fluxmi_qkv_rope(k_f16 = 1)).  The kernel then multiplies Q by 128^-0.5 * log2(e) while it builds its
 * fragments (fp16, 2^-11 relative rounding), runs QK^T on the f16 MFMA and starts every score accumulator from minus the running
 * maximum, so a score costs one exp2 instead of fma + exp2 (+4.7 % at L = 4608).  Q and V^T are bf16 either way.  Knobs
 * (fluxmi_tuning_t): attn_defer_log2 = log2 of the deferred-rescale threshold (default 8), attn_var bit 1 = exact running max,
 * attn_split = the balanced grid of fluxmi_attention_plan (fp16-K calls). */
int fluxmi_attention(const void* Q, const void* K, const void* VT, void* out, long long ld_out, int col_off, int out_fp8,
                     const float* q_scale0, const float* q_scale1, int split, int B, int L, int Lp, int H, int fmt, int k_f16,
                     void* stream);

/* The launch plan of the kernel above for ONE SAMPLE's H heads of L keys on a 256-CU part (host arithmetic only, no GPU needed; tests + bench
 * notes).  Since round 6 the plan is per sample and a batch is launched sample by sample when it is on: B only has to be >= 1 and the pieces a
 * (head, row block) is cut into -- hence its bits -- do not depend on the batch (test_attention_is_batch_invariant_at_thin_last_rounds).
 * The kernel runs one workgroup of 256 query rows per (head, row block) -- a TASK -- and one workgroup per CU at a time, so 264 tasks take two
 * rounds for 1.03 rounds of work.  Under fluxmi_tuning_t.attn_split (fp16-K calls only) every XCD runs full_per_x of its n_per_x tasks whole
 * and the remaining ones as `npieces` PIECES of their key range, launched longest first, so that every CU ends up with the same number of key
 * tiles: a THIN last round (at most 8 of an XCD's 32 CUs) is folded into the full round in front of it (768^2: 33 tasks over 32 bins), a
 * single partial round (fewer than 32 tasks per XCD) is spread over all CUs, a fuller last round is binned on its own.  A piece writes its
 * softmax state (O, m, l; fp32) to a scratch slot and the piece that arrives last at the task's counter merges them in piece order -- the
 * result does not depend on the arrival order.  Returns 0 when the launch runs one workgroup per task, 1 when a plan exists and
 * attn_split = 1 takes it (a thin last round behind at least one full one), 2 when only attn_split = 2 does (fuller last rounds and single
 * partial rounds: measured not to pay).
 * pieces[i] (i < npieces <= 64), in launch order: bits 0-7 index of the binned task, 8-15 piece index within the task, 16-23 pieces of
 * the task, 24-31 canonical index of the task's first piece (= its scratch slot), 32-47 first key tile, 48-63 key tiles. */
int fluxmi_attention_plan(int B, int L, int H, int* n_per_x, int* full_per_x, int* npieces, unsigned long long* pieces);
/* Probes (tools/attn_timeline.py): a device buffer of [workgroups][8] uint64 that every attention workgroup fills with {blockIdx, XCC id | HW_ID << 8,
 * start, end, Q fragments built, prologue tiles landed, step loop done, drain done} (100 MHz real-time counter): which CU ran which task / piece
 * when, and where a workgroup's time outside its key tiles goes.  NULL switches it off. */
int fluxmi_attention_debug_buffer(void* dev_u64);

/* The same with Q taken RAW from the qkv GEMM output (q at column 0 of `qkv`, row stride ld_qkv): QKNorm (qn_scale0 for rows
 * < split, qn_scale1 otherwise) + RoPE (pe) are applied while the query fragments are loaded, so Q never round-trips through HBM.
 * Pair with fluxmi_qkv_rope(..., Q = NULL, ...) which then produces K and V^T only.     flux_model.py:41-45,60-65,158-176 */
int fluxmi_attention_rawq(const void* qkv, long long ld_qkv, const void* pe, const void* qn_scale0, const void* qn_scale1, const void* K,
                          const void* VT, void* out, long long ld_out, int col_off, int out_fp8, const float* q_scale0,
                          const float* q_scale1, int split, int B, int L, int Lp, int H, int fmt, int k_f16, void* stream);

/* ---- VAE pieces (NHWC bf16) -------------------------------------------------------------------- */
/* 3x3 patch matrix: x [B, Hi, Wi, C] -> col [B*H*W, 9*C], column (dy*3+dx)*C + c, (H, W) = the OUTPUT grid.  `upsample`:
 *   1  stride 1 / pad 1 (Hi = H);   2  nearest 2x upsample folded into the gather (Hi = H/2; Upsample.forward);
 *  -2  stride 2 with zero pad on the right/bottom only (Hi = 2H; Downsample.forward).
 * The convolution is then fluxmi_gemm_grouped(is_fp8=0) with the weight reordered to [Cout][dy][dx][Cin].
 *                                                                            modules/autoencoder.py:65-72,95-120,228,259 */
int fluxmi_im2col3x3(const void* x, void* col, int B, int H, int W, int C, int upsample, void* stream);
/* The same convolution WITHOUT the patch matrix (ABI 5, round 6): out [B*H*W, Cout] = conv3x3(x) (+ bias) (+ resid when given: out = resid +
 * gate * y with gate [Cout] bf16, normally ones) -- the 128x128 bf16 tile kernel gathers each 128-byte K-step (64 channels of one tap) from the
 * NHWC input inside its LDS-DMA address computation and pads through a zero page; w2 [Cout][3][3][C] bf16.  Same K order, same MFMAs as the GEMM
 * on fluxmi_im2col3x3's matrix: identical bits (tests/test_ops_gpu.py::test_conv3x3_implicit_equals_im2col_gemm).  Needs C %% 64 == 0 and
 * Cout %% 128 == 0 (FLUX VAE: every 3x3 convolution except conv_in / conv_out).  `upsample` as above; (H, W) = the OUTPUT grid. */
int fluxmi_conv3x3(const void* x, const void* w2, const void* bias, const void* gate, const void* resid, void* out, int B, int H, int W, int C,
                   int Cout, int upsample, void* stream);
/* GroupNorm(32 groups, affine) in fp32 + optional swish, rounded to bf16 once; x, y [B, P, C]; work: float[(B*ceil(P/512)+B)*64] (ABI 5: was ceil(P/4096)).
 *                                                                                modules/autoencoder.py:19-20,28-30,62-70,256 */
int fluxmi_groupnorm(const void* x, const void* gamma, const void* beta, void* y, float* work, int B, int P, int C, int swish, float eps,
                     void* stream);
/* P[r,:] = softmax(scale * S[r,:]) (fp32 inside), bf16 [rows, cols], row stride ld            modules/autoencoder.py:47 */
int fluxmi_softmax_rows(const void* S, void* P, int rows, int cols, long long ld, float scale, void* stream);

/* ---- text-conditioning encoder pieces (bf16) ------------------------------------------------------------
 * The reference runs transformers' T5EncoderModel / CLIPTextModel (modules/conditioner.py:74-117, flux_emphasis.py:420-429); their
 * Linear layers map to fluxmi_gemm_grouped(is_fp8=0), the rest to the three entry points below. */
/* mode 0: T5LayerNorm  y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps)))  (bias ignored);  mode 1: LayerNorm  y = bf16((x - mean) * rstd * w + b).
 * x, y [rows, D] bf16 with row strides ldx, ldy; fp32 statistics. */
int fluxmi_row_norm(const void* x, const void* weight, const void* bias, void* y, int rows, int D, long long ldx, long long ldy, float eps, int mode,
                    void* stream);
/* mode 0: in [rows, 2F] = [a | b] -> out [rows, F] = bf16(bf16(gelu_new(a)) * b)   (T5 v1.1 gated FF: wi_0 / wi_1 fused into one GEMM)
 * mode 1: in [rows, F] -> out = bf16(a * sigmoid(1.702 a))                          (CLIP quick_gelu) */
int fluxmi_act_mul(const void* in, void* out, int rows, int F, long long ld_in, long long ld_out, int mode, void* stream);
/* Self-attention with head_dim 64 for one sequence: q, k [Lp, ld_qk] (head h at columns h*64..), vt [H*64, ld_vt] = V transposed
 * (row h*64+d, column = key; keys >= L are ignored), out [Lp, ld_out] (rows < L written).  logits = scale * q.k
 * + rel_bias[h][key - query + bias_ld/2] (fp32 [H, bias_ld], NULL = none; T5's bucketed relative-position bias) with keys > query
 * masked when causal (CLIP); fp32 softmax, P rounded to bf16, out = P V + v_bias (bf16 [H*64] or NULL).  Lp %% 32 == 0. */
int fluxmi_text_attention(const void* q, const void* k, long long ld_qk, const void* vt, long long ld_vt, void* out, long long ld_out,
                          const float* rel_bias, int bias_ld, const void* v_bias, float scale, int causal, int L, int Lp, int H, void* stream);

/* ---- step scalars -------------------------------------------------------------------------------------- */
/* timestep_embedding(t, 2*half) with host-provided frequency table                 flux_model.py:95-116 */
int fluxmi_timestep_embedding(const void* t, const float* freqs, void* out, int B, int half, float time_factor, void* stream);
/* img += dts[*step] * pred                                                          flux_pipeline.py:651 */
int fluxmi_euler(void* img, const void* pred, const float* dts, const int* step, long long n, void* stream);

/* ---- whole-model engine ------------------------------------------------------------------------------- */
typedef struct fluxmi_linear {
  const void* weight;      /* fp8 float8_data [N,K] (kind 1) or bf16 weight [N,K] (kind 0) */
  const void* bias;        /* bf16 [N] or NULL */
  float* w_scale_recip;    /* F8Linear.scale_reciprocal        (device scalar) */
  float* in_scale;         /* F8Linear.input_scale             (device scalar) */
  float* in_scale_recip;   /* F8Linear.input_scale_reciprocal  (device scalar) */
  float* amax_trials;      /* F8Linear.input_amax_trials [num_trials] (device) */
  int kind;                /* 0 = nn.Linear (bf16), 1 = F8Linear */
  int N, K;
  int in_fmt;              /* FLUXMI_E5M2 / FLUXMI_E4M3 */
} fluxmi_linear_t;

typedef struct fluxmi_model_desc {
  int hidden, heads, mlp_hidden, depth, depth_single, in_channels, vec_in, ctx_in, guidance_embed;
  int axes_dim[3];
  int theta;
  int num_trials;
} fluxmi_model_desc_t;

/* Layer order in `linears` (count = fluxmi_engine_num_linears(desc)):
 *   img_in, time_in.in, time_in.out, vector_in.in, vector_in.out, [guidance_in.in, guidance_in.out], txt_in,
 *   per double block i: img_mod.lin, img_attn.qkv, img_attn.proj, img_mlp.0, img_mlp.2,
 *                       txt_mod.lin, txt_attn.qkv, txt_attn.proj, txt_mlp.0, txt_mlp.2
 *   per single block i: modulation.lin, linear1, linear2
 *   final_layer.adaLN_modulation.1, final_layer.linear
 * `norm_scales`: bf16 [128] pointers, per double block: img q, img k, txt q, txt k; per single block: q, k. */
typedef struct fluxmi_engine fluxmi_engine_t;
int fluxmi_engine_num_linears(const fluxmi_model_desc_t* desc);
int fluxmi_engine_create(const fluxmi_model_desc_t* desc, const fluxmi_linear_t* linears, int n_linears,
                         const void* const* norm_scales, int n_norm_scales, fluxmi_engine_t** out);
int fluxmi_engine_destroy(fluxmi_engine_t* e);
/* constant tables computed by the host with the reference's own expressions: timestep frequencies
 * exp(-ln(1e4)*i/128) (flux_model.py:106-110), RoPE omega per pair and the id axis each pair uses (flux_model.py:50-51,84-90) */
int fluxmi_engine_set_tables(fluxmi_engine_t* e, const float* freqs128, const float* omega64, const int* axis64);
/* re-read weight pointers / kinds after a LoRA fuse or dtype swap (float8_quantize.py:209-212).  MANDATORY after ANY write to a weight
 * the engine was created / last re-bound with, even an in-place one that keeps the pointer: the engine SNAPSHOTS weights -- its GEMMs read a
 * row-pair copy of the block linears' fp8 weights (fluxmi_tuning_t.w_pairs, rebuilt on the first launch after a rebind), the 64 KiB
 * quantising-epilogue tables and the captured step graph are derived from the scales.  A write without a rebind is silently ignored.
 * The copies are made only while the device has at least twice their size + 4 GiB free (else the GEMMs read the caller's weights: same
 * results) and are freed when w_pairs is switched off.  A rebind that changes which linears are bf16 may drop the workspace: call
 * fluxmi_engine_prepare again before the next forward (the host wrapper does, per call). */
int fluxmi_engine_rebind(fluxmi_engine_t* e, const fluxmi_linear_t* linears, int n_linears);
/* per-request setup: (re)allocates the workspace for (B, Li, Lt), builds the RoPE table from the position ids
 * (step-invariant: flux_model.py:701-702) */
int fluxmi_engine_prepare(fluxmi_engine_t* e, int B, int Li, int Lt, const void* img_ids, const void* txt_ids, void* stream);
/* one Flux.forward (flux_model.py:672-716).  mode 0 = calibrating/unfused (advances the F8Linear trial state
 * machine exactly like the reference's first 13 calls), 1 = frozen/fused.  pred: bf16 [B,Li,in_channels]. */
int fluxmi_engine_forward(fluxmi_engine_t* e, const void* img, const void* txt, const void* y, const void* timesteps,
                          const void* guidance, void* pred, int mode, int trial_index, void* stream);
/* the denoise loop (flux_pipeline.py:619-651): timesteps_host[n_steps+1]; img updated in place.
 * Steps with trial_index <= num_trials run unfused; the remainder replays ONE captured hipGraph per step. */
int fluxmi_engine_denoise(fluxmi_engine_t* e, void* img, const void* txt, const void* y, float guidance,
                          const double* timesteps_host, int n_steps, int* trial_index_inout, int use_graph, void* stream);
/* hipEvent timing of the frozen STEPS of the last fluxmi_engine_denoise call: the first event is recorded on the caller's stream behind
 * the step-ahead modulation table, the eager warm step and the graph capture of a new shape (none of them is inside the window), the
 * second behind the last step (the reference's only meter is tqdm's it/s, flux_pipeline.py:628-630).  A request longer than 64 steps
 * includes the table builds of its later windows.  Blocks until the second event has completed.  steps = number of steps between the
 * events (0: nothing was timed, ms = 0).
 * FLUXMI_ROCTX=1 additionally emits roctx ranges (calibrating steps / modulation table / graph replays) for rocprofv3 --marker-trace. */
int fluxmi_engine_last_timing(fluxmi_engine_t* e, float* ms, int* steps);
/* Batch-sharded calibration: F8Linear.quantize_input takes amax over the WHOLE batch (float8_quantize.py:227).
 * With a hook installed the engine keeps the per-layer running amax in the caller's device array amax_dev[n >= n_linears] and
 * calls hook(user, first, count, stream) on the host after the amax of layers [first, first+count) is enqueued on `stream` and
 * before their scale update is: the host enqueues an all-reduce(MAX) of amax_dev[first..first+count) on that stream (RCCL via
 * torch.distributed) and returns 0.  Calibrating steps only; never called from the captured graph.  hook = NULL uninstalls. */
typedef int (*fluxmi_amax_hook_t)(void* user, int first, int count, void* stream);
int fluxmi_engine_set_amax_exchange(fluxmi_engine_t* e, float* amax_dev, int n, fluxmi_amax_hook_t hook, void* user);
/* introspection for tests / bench; workspace_bytes = every device byte the engine owns beside the caller's weights: the per-shape workspace,
 * the step-ahead modulation table and the row-pair weight copies (fluxmi_gemm_group_t.W_pairs; 8 GB at Flux-dev, 0 with fluxmi_tuning_t.w_pairs = 0) */
int fluxmi_engine_workspace_bytes(fluxmi_engine_t* e, long long* bytes);
int fluxmi_engine_get_buffer(fluxmi_engine_t* e, const char* name, void** ptr, long long* bytes);
/* Teacher-forced parity hooks (tests only).  run_block: stages [stage_from, stage_to] of DoubleStreamBlock (kind 0; stages
 * 0 LN+modulate->a8, 1 qkv GEMM->qkv(+V^T), 2 K relayout, 3 attention->attn8, 4 proj+gate+resid->x, 5 LN+modulate->a8,
 * 6 mlp.0+GELU->h8, 7 mlp.2+gate+resid->x; flux_model.py:356-400) or SingleStreamBlock (kind 1; 0 LN+modulate->a8,
 * 1 linear1->qkv(+V^T)|cat8[:,H:], 2 K relayout, 3 attention->cat8[:,:H], 4 linear2+gate+resid->x; flux_model.py:467-485) `index`
 * on the engine's own workspace: the residual stream is buffer "x" ([B, Lt+Li, H], txt rows first), the modulation vectors are read
 * from buffer "mod" (per batch row: double block i at [i*12H, +12H) = img shift1|scale1|gate1|shift2|scale2|gate2 then txt, single
 * block i at depth*12H + i*3H = shift|scale|gate, LastLayer.adaLN at depth*12H + single*3H = shift|scale).  kind 2 = LastLayer (index 0;
 * stages 0 LN+modulate of the img rows of x -> "fin", 1 bf16 Linear -> buffer "pred_s" [B, Li, in_channels]; flux_model.py:499-503).
 * mode 1 = fused kernels, 2 = unfused with frozen scales.
 * copy_buffer: device-to-device copy between a named workspace buffer and a caller buffer (to_engine != 0 writes the workspace).  The fp8
 * activation buffers "a8", "attn8", "h8", "cat8" are exchanged as PLAIN rows: while the engine keeps them in row pairs (fused mode,
 * fluxmi_tuning_t.a_pairs, even L and Lt) the copy converts, and offset / bytes must then cover whole pairs of rows.  (The unfused modes
 * stage plain rows into the same buffers: this hook serves mode-1 teacher forcing.) */
int fluxmi_engine_run_block(fluxmi_engine_t* e, int kind, int index, int mode, int stage_from, int stage_to, void* stream);
int fluxmi_engine_copy_buffer(fluxmi_engine_t* e, const char* name, long long offset, void* dev_ptr, long long bytes, int to_engine,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLUXMI_H */
