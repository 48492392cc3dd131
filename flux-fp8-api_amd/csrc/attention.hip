// fluxmi -- flash-attention forward for Flux joint attention (head_dim 128, non-causal), gfx950: argument checks and kernel selection.
//
// Replaces F.scaled_dot_product_attention + transpose/reshape (reference flux_model.py:41-45) and, in fused mode, the fp8 quantise of
// the consumer F8Linear (float8_quantize.py:274-276).  The kernel (attention2.hip):
//  * KV tiles of 64 keys; K tile [64][128] and V^T tile [128][64] arrive by LDS-DMA into 4-deep rings, XOR-swizzled via the source
//    address; one workgroup = 256 query rows, whole heads per XCD (xcd_remap);
//  * "swapped" QK^T: S^T = K . Q^T on the 32x32x16 MFMA, so a lane owns ONE query row (col = lane & 31) and 16 keys per 32x32 tile ->
//    row max / sum are in-lane plus one lane ^ 32 exchange;
//  * the P^T operand of O^T += V^T . P^T is the S^T accumulator itself, converted to bf16 in place: the k-slot <-> key mapping that
//    results (key = (e & 3) + 8 (e >> 2) + 4 hi inside each 16-key group) is pre-applied to V by qkv_rope.hip, so no cross-lane traffic
//    and no transpose read is needed;
//  * online softmax in the exp2 domain with a deferred running max, fp32 accumulators, P rounded to bf16 for the PV MFMA.
#include "attention_common.h"

int fluxmi_k_attention(const void* Q, const void* K, const void* VT, void* out, long long ld_out, int col_off, int out_fp8,
                       const float* q_scale0, const float* q_scale1, int split, int B, int L, int Lp, int H, int fmt,
                       hipStream_t s, const void* qraw, long long ldq, const void* pe, const void* qn0, const void* qn1, int k_f16, int out_pairs) {
  FLUXMI_REQUIRE(Q || (qraw && pe && qn0 && qn1 && ldq % 8 == 0), "attention: need Q, or raw q + pe + both q-norm scales (ld %% 8 == 0)");
  FLUXMI_REQUIRE(Lp % 64 == 0 && Lp >= L, "attention: Lp=%d must be a multiple of 64 and >= L=%d", Lp, L);
  FLUXMI_REQUIRE(!out_fp8 || (q_scale0 && q_scale1), "attention: fp8 output needs q_scale pointers");
  FLUXMI_REQUIRE(!out_pairs || (out_fp8 && ld_out % 64 == 0 && col_off % 64 == 0 && ((long long)B * L) % 2 == 0 && L % 2 == 0),
                 "attention: the row-pair output layout needs fp8 output, 64-byte aligned rows and an even L");
  if (B * L * H == 0) return 0;
  AttnArgs a;
  a.out_pairs = out_pairs;
  a.Q = (const u16*)Q; a.K = (const u16*)K; a.VT = (const u16*)VT;
  a.qraw = (const u16*)qraw; a.ldq = ldq; a.pe = (const u16*)pe; a.qn[0] = (const u16*)qn0; a.qn[1] = (const u16*)qn1;
  a.out = out; a.ld_out = ld_out; a.col_off = col_off; a.out_fp8 = out_fp8;
  a.q_scale[0] = q_scale0; a.q_scale[1] = q_scale1; a.split = split;
  a.B = B; a.L = L; a.Lp = Lp; a.H = H;
  a.scale_log2 = 0.08838834764831845f * 1.4426950408889634f;  // 128^-0.5 * log2(e)
  a.k_f16 = k_f16;
  a.dbg = nullptr;
  const fluxmi_tuning_t tun = fluxmi_tuning();
  a.pf = fluxmi_take_prefetch();
  if (!tun.prefetch) a.pf.n = 0;
  // Deferred running max (guide T13): P is bounded by 2^defer_log2 instead of 1, O and l carry the same factor, so the quotient is
  // unchanged; fp32 holds 4608 keys x 2^24 x |V| with a hundred binades to spare.  fluxmi_tuning_t.attn_defer_log2, validated to [0, 16].
  a.defer_log2 = tun.attn_defer_log2;
  a.abl = tun.attn_abl;
  // the regrouped fp8 epilogue stores 16 B per lane: rows that are not 16-byte aligned keep the 4-byte stores
  if (out_fp8 && ((((uintptr_t)out) | (uintptr_t)ld_out | (uintptr_t)col_off) & 15)) a.abl |= 8;
  return fluxmi_launch_attention2(a, fmt, s, (tun.attn_var & 2) != 0);
}
