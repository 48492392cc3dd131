// fluxmi -- fused "qkv split + QK-RMSNorm + RoPE + head-major relayout" (gfx950).
//
// Replaces, per block of the reference:  reshape/permute (flux_model.py:351-354, 476-477),
// QKNorm = fp32 rms_norm over head_dim with learnable scale (flux_model.py:158-176),
// torch.cat((txt,img)) of q/k/v (flux_model.py:380-382) and apply_rope in bf16 arithmetic
// (flux_model.py:60-65).  One pass over the bf16 qkv GEMM output writes
//   Q [B,H,L,128]  K [B,H,L,128]      (normalised, rotated, bf16)
//   VT[B,H,128,Lp]                    V transposed; inside every 16-key group the key order is
//                                      bit2<->bit3 swapped, which is exactly the k-slot order the PV
//                                      MFMA of the attention kernels consumes -> no transpose in the hot loop.
// Rows l < split use norm-scale set 0 (txt stream), the rest set 1 (img stream).  Q == nullptr / VT == nullptr: that output is skipped (produced elsewhere).
#include "common.h"
#include "fluxmi_internal.h"

namespace {

struct QkvRopeArgs {
  const u16* qkv; long long ld;          // [B*L, ld]; q at col 0, k at H*128, v at 2*H*128
  const u16* pe;                         // [B, L, 64, 2] (cos, sin) bf16
  const u16* qs[2]; const u16* ks[2];    // RMSNorm scales [128] per stream
  u16* Q; u16* K; u16* VT;
  int B, L, Lp, H, split;
  int k_f16;  // K is stored as fp16 (exact for bf16 values in fp16's range): operand format of the folded QK^T of attention2.hip
};

__global__ void __launch_bounds__(256) qkv_rope_kernel(const QkvRopeArgs a) {
  __shared__ __attribute__((aligned(16))) u16 vt[64][136];
  const int l0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x, sub = t & 15, rgrp = t >> 4;
  const long long HD = (long long)a.H * 128;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = pass * 16 + rgrp, l = l0 + r;
    const bool ok = l < a.L;
    const int lc = ok ? l : a.L - 1;
    const u16* row = a.qkv + ((long long)b * a.L + lc) * a.ld + (long long)h * 128 + sub * 8;
    const int st = (lc < a.split) ? 0 : 1;
    // pe: 4 (cos,sin) pairs for d = sub*8 .. sub*8+7
    const uint4 pev = *(const uint4*)(a.pe + (((long long)b * a.L + lc) * 64 + sub * 4) * 2);
    float cs[8];
    unpack8(pev, cs);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      if (part == 0 && a.Q == nullptr) continue;  // Q is normalised + rotated inside the attention kernel (raw-Q mode)
      float x[8], w[8];
      unpack8(*(const uint4*)(row + part * HD), x);
      unpack8(*(const uint4*)((part == 0 ? a.qs[st] : a.ks[st]) + sub * 8), w);
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 16);
      const float rinv = 1.0f / sqrtf(ss * (1.0f / 128.0f) + 1e-6f);
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = rbf((x[j] * rinv) * w[j]);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float c = cs[2 * p], s = cs[2 * p + 1];
        y[2 * p] = rbf(rbf(c * x[2 * p]) + rbf((-s) * x[2 * p + 1]));
        y[2 * p + 1] = rbf(rbf(s * x[2 * p]) + rbf(c * x[2 * p + 1]));
      }
      if (ok) {
        u16* dst = (part == 0 ? a.Q : a.K) + (((long long)b * a.H + h) * a.L + l) * 128 + sub * 8;
        *(uint4*)dst = (part == 1 && a.k_f16) ? pack8_f16(y) : pack8(y);
      }
    }
    if (a.VT) {
      uint4 vv = *(const uint4*)(row + 2 * HD);
      if (!ok) vv = make_uint4(0, 0, 0, 0);
      *(uint4*)(&vt[r][sub * 8]) = vv;
    }
  }
  if (a.VT == nullptr) return;  // V^T was produced by the GEMM epilogue (fluxmi_gemm_group_t.vt_out)
  __syncthreads();
  // transposed, key-permuted write of V: item = (d, 8-position group)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int item = it * 256 + t;
    const int d = item & 127, pg = item >> 7;  // pg in [0,8): positions pg*8 .. pg*8+7
    const int g16 = pg >> 1, half = pg & 1;     // 16-key group, low/high half of storage positions
    u16 e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // storage position within the group: jj = half*8 + j  ->  key offset = swap bits 2,3 of jj
      const int koff = (j & 3) | ((j >> 2) << 3) | (half << 2);
      e[j] = vt[g16 * 16 + koff][d];
    }
    uint4 o;
    o.x = e[0] | ((unsigned)e[1] << 16); o.y = e[2] | ((unsigned)e[3] << 16);
    o.z = e[4] | ((unsigned)e[5] << 16); o.w = e[6] | ((unsigned)e[7] << 16);
    *(uint4*)(a.VT + (((long long)b * a.H + h) * 128 + d) * a.Lp + l0 + pg * 8) = o;
  }
}

}  // namespace

int fluxmi_k_qkv_rope(const void* qkv, long long ld, const void* pe, const void* q_scale0, const void* k_scale0,
                      const void* q_scale1, const void* k_scale1, void* Q, void* K, void* VT, int B, int L, int Lp, int H,
                      int split, int k_f16, hipStream_t s) {
  FLUXMI_REQUIRE(Lp % 64 == 0 && Lp >= L, "qkv_rope: Lp=%d must be a multiple of 64 and >= L=%d", Lp, L);
  FLUXMI_REQUIRE(ld % 8 == 0, "qkv_rope: ld must be a multiple of 8");
  if (B * L * H == 0) return 0;
  QkvRopeArgs a;
  a.qkv = (const u16*)qkv; a.ld = ld; a.pe = (const u16*)pe;
  a.qs[0] = (const u16*)q_scale0; a.ks[0] = (const u16*)k_scale0;
  a.qs[1] = (const u16*)q_scale1; a.ks[1] = (const u16*)k_scale1;
  a.Q = (u16*)Q; a.K = (u16*)K; a.VT = (u16*)VT;
  a.B = B; a.L = L; a.Lp = Lp; a.H = H; a.split = split; a.k_f16 = k_f16;
  hipLaunchKernelGGL(qkv_rope_kernel, dim3((L + 63) / 64, H, B), dim3(256), 0, s, a);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
