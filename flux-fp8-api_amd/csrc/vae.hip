// fluxmi -- kernels of the VAE (decoder = the step right after the denoise loop; encoder = img2img), gfx950.
//
// Reference: modules/autoencoder.py:203-283 (Decoder), :123-200 (Encoder), :55-93 (ResnetBlock), :23-52 (AttnBlock), :95-120 (Down/Upsample), run under
// torch.autocast(bf16) by flux_pipeline.py:423-437.  Layout here is NHWC (channels innermost) so that
//   * a 3x3 convolution is  im2col (this file) + the bf16 MFMA GEMM of gemm*.hip  with K = 9*Cin ordered (dy, dx, c), the 2x
//     nearest-neighbour upsample of Upsample.forward / the stride-2, right-bottom-padded window of Downsample.forward folded into the gather;
//   * 1x1 convolutions (nin_shortcut, q/k/v/proj_out) are plain GEMMs on the [pixels, C] matrix;
//   * residual adds ride in the GEMM's gate*y+x epilogue with gate = 1.
// GroupNorm(32 groups, eps 1e-6, affine) runs in fp32 like autocast does and is fused with the swish that always follows it in a
// ResnetBlock; the result is rounded to bf16 once -- exactly the cast autocast applies at the next convolution's input.
// Softmax of the single 512-wide attention head: fp32 rows of  scale * S,  S = Q K^T from the GEMM in bf16.
#include <map>
#include <mutex>

#include "common.h"
#include "fluxmi_internal.h"

namespace {

// ---- im2col for 3x3 convolutions on NHWC -------------------------------------------------------------------------------------
// x [B, Hi, Wi, C] -> col [B*H*W, 9*C], column (dy*3+dx)*C + c; (H, W) is the OUTPUT grid.  One thread moves 8 channels (16 B).
//   up = 1 : stride 1, pad 1                                   (Hi = H)
//   up = 2 : nearest 2x upsample, then stride 1, pad 1          (Hi = H/2;  Upsample.forward, reference :117-119)
//   up = -2: stride 2, zero pad (0,1,0,1) = right/bottom only   (Hi = 2H;   Downsample.forward, reference :103-107)
// i.e. source row = (yo*stride + dy - pad) / rep, zero outside [0, Hi*rep).
__global__ void __launch_bounds__(256) im2col3x3_kernel(const u16* __restrict__ x, u16* __restrict__ col, int B, int H, int W, int C, int Hi, int Wi,
                                                        int stride, int pad, int rep) {
  const int c8 = C >> 3;
  const long long total = (long long)B * H * W * 9 * c8;
  const int Hv = Hi * rep, Wv = Wi * rep;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    long long r = i / c8;
    const int tap = (int)(r % 9);
    r /= 9;
    const int xo = (int)(r % W);
    r /= W;
    const int yo = (int)(r % H);
    const int b = (int)(r / H);
    const int yy = yo * stride + tap / 3 - pad, xx = xo * stride + tap % 3 - pad;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (yy >= 0 && yy < Hv && xx >= 0 && xx < Wv) v = *(const uint4*)(x + (((long long)b * Hi + yy / rep) * Wi + xx / rep) * C + cc * 8);
    *(uint4*)(col + i * 8) = v;
  }
}

// ---- GroupNorm statistics: partial (sum, sum of squares) per (batch, pixel chunk, group) -----------------------------------------
// x [B, P, C] bf16 (P pixels), groups of C/32 consecutive channels.  Block = one chunk of `ppc` pixels of one image; thread t walks
// 8-channel vectors.  Deterministic: fixed partition, fixed tree.
__global__ void __launch_bounds__(256) gn_partial_kernel(const u16* __restrict__ x, float* __restrict__ part, int P, int C, int ppc, int nchunks) {
  __shared__ float ssum[256], ssq[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int c8 = C >> 3, cpg = C / 32;  // channels per group (>= 1); an 8-vector spans 8/cpg groups if cpg < 8
  const int p0 = chunk * ppc, p1 = min(P, p0 + ppc);
  // thread t owns vector column (t % c8) and pixel lane (t / c8); needs c8 <= 256
  const int vc = threadIdx.x % c8, pl = threadIdx.x / c8, npl = 256 / c8;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  if (pl < npl) {
    // two loads in flight per thread (round 6: with one load per iteration and 4096-pixel chunks -- 256 blocks for a 1024^2 image -- this kernel ran at
    // 0.9 TB/s and was a third of the VAE decode); the accumulation order per thread is unchanged (pixel p before pixel p + npl)
    const u16* xb = x + (long long)b * P * C + vc * 8;
    int p = p0 + pl;
    for (; p + npl < p1; p += 2 * npl) {
      const uint4 r0 = *(const uint4*)(xb + (long long)p * C), r1 = *(const uint4*)(xb + (long long)(p + npl) * C);
      float v[8], w[8];
      unpack8(r0, v);
      unpack8(r1, w);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += w[j]; q[j] += w[j] * w[j]; }
    }
    if (p < p1) {
      float v[8];
      unpack8(*(const uint4*)(xb + (long long)p * C), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
    }
  }
  // fold the 8 channels of the vector into their groups, then reduce over threads through LDS: slot = group (32)
  for (int g = 0; g < 32; ++g) {
    float a = 0.f, c = 0.f;
    if (pl < npl) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if ((vc * 8 + j) / cpg == g) { a += s[j]; c += q[j]; }
    }
    ssum[threadIdx.x] = a; ssq[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) { ssum[threadIdx.x] += ssum[threadIdx.x + o]; ssq[threadIdx.x] += ssq[threadIdx.x + o]; }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      float* dst = part + (((long long)b * nchunks + chunk) * 32 + g) * 2;
      dst[0] = ssum[0]; dst[1] = ssq[0];
    }
    __syncthreads();
  }
}
// stats[b, g] = (mean, rstd) from the partials, combined in double
// 256 threads per image: group g = t % 32, eight lanes per group each summing every eighth chunk, then the eight in lane order (fixed tree)
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, int nchunks, double count, float eps) {
  __shared__ double ls[8][32], lq[8][32];
  const int b = blockIdx.x, g = threadIdx.x & 31, k = threadIdx.x >> 5;
  double s = 0.0, q = 0.0;
  for (int c = k; c < nchunks; c += 8) {
    const float* p = part + (((long long)b * nchunks + c) * 32 + g) * 2;
    s += p[0]; q += p[1];
  }
  ls[k][g] = s; lq[k][g] = q;
  __syncthreads();
  if (k != 0) return;
  s = 0.0; q = 0.0;
  for (int i = 0; i < 8; ++i) { s += ls[i][g]; q += lq[i][g]; }
  const double mean = s / count, var = q / count - mean * mean;
  stats[(b * 32 + g) * 2] = (float)mean;
  stats[(b * 32 + g) * 2 + 1] = (float)(1.0 / sqrt((var < 0.0 ? 0.0 : var) + (double)eps));
}
// y = bf16( act( (x - mean) * rstd * gamma + beta ) ), act = swish (x * sigmoid(x)) or identity; all in fp32
__global__ void __launch_bounds__(256) gn_apply_kernel(const u16* __restrict__ x, const float* __restrict__ stats, const u16* __restrict__ gamma,
                                                       const u16* __restrict__ beta, u16* __restrict__ y, int B, int P, int C, int swish) {
  const int c8 = C >> 3, cpg = C / 32;
  const long long total = (long long)B * P * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int vc = (int)(i % c8);
    const int b = (int)(i / ((long long)P * c8));
    float v[8], ga[8], be[8], o[8];
    unpack8(*(const uint4*)(x + i * 8), v);
    unpack8(*(const uint4*)(gamma + vc * 8), ga);
    unpack8(*(const uint4*)(beta + vc * 8), be);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (vc * 8 + j) / cpg;
      const float mean = stats[(b * 32 + g) * 2], rstd = stats[(b * 32 + g) * 2 + 1];
      float t = (v[j] - mean) * rstd * ga[j] + be[j];
      if (swish) t = t / (1.0f + __expf(-t));
      o[j] = t;
    }
    *(uint4*)(y + i * 8) = pack8(o);
  }
}

// ---- row softmax: P[r, :] = softmax(scale * S[r, :]) in fp32, bf16 in / bf16 out; one block per row, cols % 8 == 0 ----------------
__global__ void __launch_bounds__(256) softmax_rows_kernel(const u16* __restrict__ S, u16* __restrict__ Pm, int cols, long long ld, float scale) {
  __shared__ float red[4];
  const u16* row = S + (long long)blockIdx.x * ld;
  u16* out = Pm + (long long)blockIdx.x * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = -3.0e38f;
  for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
    float v[8];
    unpack8(*(const uint4*)(row + c), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, v[j]);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  const float k = scale * 1.4426950408889634f;
  float sum = 0.f;
  for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
    float v[8];
    unpack8(*(const uint4*)(row + c), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += __builtin_amdgcn_exp2f((v[j] - mx) * k);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
    float v[8];
    unpack8(*(const uint4*)(row + c), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_exp2f((v[j] - mx) * k) * inv;
    *(uint4*)(out + c) = pack8(v);
  }
}

int grid1d(long long n) { return (int)((n + 255) / 256 > 65535 * 16 ? 65535 * 16 : (n + 255) / 256); }

}  // namespace

// 256 zero bytes per device: the source of an implicit convolution's out-of-image taps (gemm.hip, CONV)
const void* fluxmi_zero_page() {
  static std::mutex mu;
  static std::map<int, void*> pages;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  auto it = pages.find(dev);
  if (it != pages.end()) return it->second;
  void* p = nullptr;
  if (hipMalloc(&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  pages[dev] = p;
  return p;
}

int fluxmi_k_im2col3x3(const void* x, void* col, int B, int H, int W, int C, int up, hipStream_t s) {
  FLUXMI_REQUIRE(C % 8 == 0 && (up == 1 || up == 2 || up == -2), "im2col3x3: C %% 8 == 0, mode in {1, 2, -2}");
  FLUXMI_REQUIRE(up != 2 || (H % 2 == 0 && W % 2 == 0), "im2col3x3: upsampled output dims must be even");
  const long long total = (long long)B * H * W * 9 * (C / 8);
  if (total == 0) return 0;
  const int Hi = up == 2 ? H / 2 : up == -2 ? H * 2 : H, Wi = up == 2 ? W / 2 : up == -2 ? W * 2 : W;
  const int stride = up == -2 ? 2 : 1, pad = up == -2 ? 0 : 1, rep = up == 2 ? 2 : 1;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid1d(total)), dim3(256), 0, s, (const u16*)x, (u16*)col, B, H, W, C, Hi, Wi, stride, pad, rep);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

// workspace: float[B * nchunks * 64 + B * 64], nchunks = ceil(P / 512)
int fluxmi_k_groupnorm(const void* x, const void* gamma, const void* beta, void* y, float* work, int B, int P, int C, int swish, float eps,
                       hipStream_t s) {
  FLUXMI_REQUIRE(C % 32 == 0 && C % 8 == 0 && C <= 2048, "groupnorm: C must be a multiple of 32 (<= 2048)");
  if ((long long)B * P * C == 0) return 0;
  const int ppc = 512, nchunks = (P + ppc - 1) / ppc;  // 2048 blocks for a 1024^2 image (round 6; was 4096 pixels per block: 256 blocks)
  float* part = work;
  float* stats = work + (long long)B * nchunks * 64;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunks, B), dim3(256), 0, s, (const u16*)x, part, P, C, ppc, nchunks);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, part, stats, nchunks, (double)P * (C / 32), eps);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(grid1d((long long)B * P * (C / 8))), dim3(256), 0, s, (const u16*)x, stats, (const u16*)gamma,
                     (const u16*)beta, (u16*)y, B, P, C, swish);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}

int fluxmi_k_softmax_rows(const void* S, void* P, int rows, int cols, long long ld, float scale, hipStream_t s) {
  FLUXMI_REQUIRE(cols % 8 == 0 && ld % 8 == 0, "softmax_rows: cols and ld must be multiples of 8");
  if (rows == 0 || cols == 0) return 0;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, s, (const u16*)S, (u16*)P, cols, ld, scale);
  FLUXMI_LAUNCH_CHECK();
  return 0;
}
