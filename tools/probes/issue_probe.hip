// Issue cost of the attention loop's instruction kinds for ONE wave per SIMD (256-thread workgroups, one per CU), gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/issue_probe.hip -o tools/probes/issue_probe.bin && tools/probes/issue_probe.bin
// Each test is a fully unrolled block of REP copies of a pattern inside a loop of ITER iterations, timed with s_memtime (100 MHz
// constant clock -> converted with the measured wall clock) -- reported as GPU cycles per pattern at the clock the test ran at, taken
// from a calibration loop of dependent v_add (4 cycles... unknown) so also as ns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int TEST> __global__ void __launch_bounds__(256, 1) probe(float* out, int iters, long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int lane = threadIdx.x;
  float a0 = lane * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float m = 0.f;
  v16f acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  v8bf fa = {}, fb = {};
  for (int i = threadIdx.x; i < 16384; i += 256) ((float*)lds)[i] = i;
  __syncthreads();
  unsigned addr = (threadIdx.x & 63) * 16;
  int pk = 0;
  uint4 d0, d1;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (TEST == 0) {  // 8 independent v_add per pattern
      REP32(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if constexpr (TEST == 1) {  // 8 dependent v_add
      REP32(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(m));)
    } else if constexpr (TEST == 2) {  // 8 independent v_exp
      REP32(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (TEST == 3) {  // 4 x (exp, add) independent pairs
      REP32(asm volatile("v_exp_f32 %0, %0\n v_add_f32 %4, %4, %1\n v_exp_f32 %1, %1\n v_add_f32 %5, %5, %2\n v_exp_f32 %2, %2\n v_add_f32 %6, %6, %3\n v_exp_f32 %3, %3\n v_add_f32 %7, %7, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (TEST == 4) {  // 8 v_max3 (2 chains)
      REP32(asm volatile("v_max3_f32 %0, %0, %2, %3\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %0, %0, %6, %7\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %6, %7\n v_max3_f32 %0, %0, %2, %3\n v_max3_f32 %1, %1, %4, %5"
                         : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
    } else if constexpr (TEST == 5) {  // 8 v_cvt_pk
      REP32(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %0, %3, %4\n v_cvt_pk_bf16_f32 %0, %5, %6\n v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %0, %3, %4\n v_cvt_pk_bf16_f32 %0, %5, %6\n v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %0, %3, %4"
                         : "=v"(pk) : "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6));)
    } else if constexpr (TEST == 6) {  // 8 MFMA bf16 32x32x16 on 4 accumulators
      REP32(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n"
                         "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                         : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3) : "v"(fa), "v"(fb));)
    } else if constexpr (TEST == 7) {  // 4 x (MFMA + add, exp, max3, cvt)
      REP32(asm volatile("v_mfma_f32_32x32x16_bf16 %[c0], %[fa], %[fb], %[c0]\n v_add_f32 %[x0], %[x0], %[x1]\n v_exp_f32 %[x2], %[x2]\n v_max3_f32 %[x3], %[x3], %[x4], %[x5]\n v_cvt_pk_bf16_f32 %[pk], %[x4], %[x5]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c1], %[fa], %[fb], %[c1]\n v_add_f32 %[x1], %[x1], %[x0]\n v_exp_f32 %[x4], %[x4]\n v_max3_f32 %[x3], %[x3], %[x2], %[x5]\n v_cvt_pk_bf16_f32 %[pk], %[x4], %[x5]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c2], %[fa], %[fb], %[c2]\n v_add_f32 %[x0], %[x0], %[x1]\n v_exp_f32 %[x5], %[x5]\n v_max3_f32 %[x3], %[x3], %[x4], %[x2]\n v_cvt_pk_bf16_f32 %[pk], %[x4], %[x5]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c3], %[fa], %[fb], %[c3]\n v_add_f32 %[x1], %[x1], %[x0]\n v_exp_f32 %[x2], %[x2]\n v_max3_f32 %[x3], %[x3], %[x4], %[x5]\n v_cvt_pk_bf16_f32 %[pk], %[x4], %[x5]"
                         : [c0] "+a"(acc0), [c1] "+a"(acc1), [c2] "+a"(acc2), [c3] "+a"(acc3), [x0] "+v"(a0), [x1] "+v"(a1), [x2] "+v"(a2), [x3] "+v"(a3), [x4] "+v"(a4), [x5] "+v"(a5), [pk] "+v"(pk) : [fa] "v"(fa), [fb] "v"(fb));)
    } else if constexpr (TEST == 8) {  // 4 x (MFMA + add, exp)
      REP32(asm volatile("v_mfma_f32_32x32x16_bf16 %[c0], %[fa], %[fb], %[c0]\n v_add_f32 %[x0], %[x0], %[x1]\n v_exp_f32 %[x2], %[x2]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c1], %[fa], %[fb], %[c1]\n v_add_f32 %[x1], %[x1], %[x0]\n v_exp_f32 %[x4], %[x4]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c2], %[fa], %[fb], %[c2]\n v_add_f32 %[x0], %[x0], %[x1]\n v_exp_f32 %[x5], %[x5]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c3], %[fa], %[fb], %[c3]\n v_add_f32 %[x1], %[x1], %[x0]\n v_exp_f32 %[x2], %[x2]"
                         : [c0] "+a"(acc0), [c1] "+a"(acc1), [c2] "+a"(acc2), [c3] "+a"(acc3), [x0] "+v"(a0), [x1] "+v"(a1), [x2] "+v"(a2), [x3] "+v"(a3), [x4] "+v"(a4), [x5] "+v"(a5), [pk] "+v"(pk) : [fa] "v"(fa), [fb] "v"(fb));)
    } else if constexpr (TEST == 9) {  // 4 x (MFMA + 4 v_add)
      REP32(asm volatile("v_mfma_f32_32x32x16_bf16 %[c0], %[fa], %[fb], %[c0]\n v_add_f32 %[x0], %[x0], %[x1]\n v_add_f32 %[x2], %[x2], %[x1]\n v_add_f32 %[x3], %[x3], %[x1]\n v_add_f32 %[x4], %[x4], %[x1]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c1], %[fa], %[fb], %[c1]\n v_add_f32 %[x0], %[x0], %[x1]\n v_add_f32 %[x2], %[x2], %[x1]\n v_add_f32 %[x3], %[x3], %[x1]\n v_add_f32 %[x4], %[x4], %[x1]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c2], %[fa], %[fb], %[c2]\n v_add_f32 %[x0], %[x0], %[x1]\n v_add_f32 %[x2], %[x2], %[x1]\n v_add_f32 %[x3], %[x3], %[x1]\n v_add_f32 %[x4], %[x4], %[x1]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c3], %[fa], %[fb], %[c3]\n v_add_f32 %[x0], %[x0], %[x1]\n v_add_f32 %[x2], %[x2], %[x1]\n v_add_f32 %[x3], %[x3], %[x1]\n v_add_f32 %[x4], %[x4], %[x1]"
                         : [c0] "+a"(acc0), [c1] "+a"(acc1), [c2] "+a"(acc2), [c3] "+a"(acc3), [x0] "+v"(a0), [x1] "+v"(a1), [x2] "+v"(a2), [x3] "+v"(a3), [x4] "+v"(a4), [x5] "+v"(a5), [pk] "+v"(pk) : [fa] "v"(fa), [fb] "v"(fb));)
    } else if constexpr (TEST == 10) {  // 4 x (MFMA + 6 v_add)
      REP32(asm volatile("v_mfma_f32_32x32x16_bf16 %[c0], %[fa], %[fb], %[c0]\n v_add_f32 %[x0], %[x0], %[x1]\n v_add_f32 %[x2], %[x2], %[x1]\n v_add_f32 %[x3], %[x3], %[x1]\n v_add_f32 %[x4], %[x4], %[x1]\n v_add_f32 %[x5], %[x5], %[x1]\n v_add_f32 %[x0], %[x0], %[x1]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c1], %[fa], %[fb], %[c1]\n v_add_f32 %[x0], %[x0], %[x1]\n v_add_f32 %[x2], %[x2], %[x1]\n v_add_f32 %[x3], %[x3], %[x1]\n v_add_f32 %[x4], %[x4], %[x1]\n v_add_f32 %[x5], %[x5], %[x1]\n v_add_f32 %[x0], %[x0], %[x1]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c2], %[fa], %[fb], %[c2]\n v_add_f32 %[x0], %[x0], %[x1]\n v_add_f32 %[x2], %[x2], %[x1]\n v_add_f32 %[x3], %[x3], %[x1]\n v_add_f32 %[x4], %[x4], %[x1]\n v_add_f32 %[x5], %[x5], %[x1]\n v_add_f32 %[x0], %[x0], %[x1]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c3], %[fa], %[fb], %[c3]\n v_add_f32 %[x0], %[x0], %[x1]\n v_add_f32 %[x2], %[x2], %[x1]\n v_add_f32 %[x3], %[x3], %[x1]\n v_add_f32 %[x4], %[x4], %[x1]\n v_add_f32 %[x5], %[x5], %[x1]\n v_add_f32 %[x0], %[x0], %[x1]"
                         : [c0] "+a"(acc0), [c1] "+a"(acc1), [c2] "+a"(acc2), [c3] "+a"(acc3), [x0] "+v"(a0), [x1] "+v"(a1), [x2] "+v"(a2), [x3] "+v"(a3), [x4] "+v"(a4), [x5] "+v"(a5), [pk] "+v"(pk) : [fa] "v"(fa), [fb] "v"(fb));)
    } else if constexpr (TEST == 11) {  // 4 x (MFMA with VGPR C/D f16 + 2 add + exp): the QK form
      REP32(asm volatile("v_mfma_f32_32x32x16_f16 %[c0], %[fa], %[fb], %[c0]\n v_add_f32 %[x0], %[x0], %[x1]\n v_exp_f32 %[x2], %[x2]\n"
                         "v_mfma_f32_32x32x16_f16 %[c1], %[fa], %[fb], %[c1]\n v_add_f32 %[x1], %[x1], %[x0]\n v_exp_f32 %[x4], %[x4]\n"
                         "v_mfma_f32_32x32x16_f16 %[c2], %[fa], %[fb], %[c2]\n v_add_f32 %[x0], %[x0], %[x1]\n v_exp_f32 %[x5], %[x5]\n"
                         "v_mfma_f32_32x32x16_f16 %[c3], %[fa], %[fb], %[c3]\n v_add_f32 %[x1], %[x1], %[x0]\n v_exp_f32 %[x2], %[x2]"
                         : [c0] "+v"(acc0), [c1] "+v"(acc1), [c2] "+v"(acc2), [c3] "+v"(acc3), [x0] "+v"(a0), [x1] "+v"(a1), [x2] "+v"(a2), [x3] "+v"(a3), [x4] "+v"(a4), [x5] "+v"(a5), [pk] "+v"(pk) : [fa] "v"(fa), [fb] "a"(fb));)
    } else if constexpr (TEST == 13) {  // 1 chain
      REP32(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(fa), "v"(fb));)
    } else if constexpr (TEST == 14) {  // 2 chains
      REP32(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+a"(acc0), "+a"(acc1) : "v"(fa), "v"(fb));)
    } else if constexpr (TEST == 15) {  // 3 chains (6 MFMAs)
      REP32(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0\n v_mfma_f32_32x32x16_bf16 %1, %3, %4, %1\n v_mfma_f32_32x32x16_bf16 %2, %3, %4, %2\n v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0\n v_mfma_f32_32x32x16_bf16 %1, %3, %4, %1\n v_mfma_f32_32x32x16_bf16 %2, %3, %4, %2" : "+a"(acc0), "+a"(acc1), "+a"(acc2) : "v"(fa), "v"(fb));)
    } else if constexpr (TEST == 16) {  // 2 chains, VGPR accumulators, f16 (the QK form)
      REP32(asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1" : "+v"(acc0), "+v"(acc1) : "v"(fa), "a"(fb));)
    } else if constexpr (TEST == 17) {  // 2 chains VGPR f16 with 2 fillers per gap
      REP32(asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_add_f32 %4, %4, %5\n v_exp_f32 %6, %6\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_add_f32 %5, %5, %4\n v_exp_f32 %7, %7\n v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_add_f32 %4, %4, %5\n v_exp_f32 %6, %6\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_add_f32 %5, %5, %4\n v_exp_f32 %7, %7"
                         : "+v"(acc0), "+v"(acc1) : "v"(fa), "a"(fb), "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if constexpr (TEST == 18) {  // alternate: QK-form chain pair interleaved with independent PV-form MFMAs (distance 4)
      REP32(asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %2, %4, %4, %2\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n v_mfma_f32_32x32x16_bf16 %3, %4, %4, %3" : "+v"(acc0), "+v"(acc1), "+a"(acc2), "+a"(acc3) : "v"(fa), "a"(fb));)
    } else if constexpr (TEST == 12) {  // 8 ds_read_b128 then wait
      REP32(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n ds_read_b128 %0, %2 offset:8192\n ds_read_b128 %1, %2 offset:12288\n"
                         "ds_read_b128 %0, %2 offset:16384\n ds_read_b128 %1, %2 offset:20480\n ds_read_b128 %0, %2 offset:24576\n ds_read_b128 %1, %2 offset:28672\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(d0), "=&v"(d1) : "v"(addr));)
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m + acc0[0] + acc1[1] + acc2[2] + acc3[3] + pk + d0.x + d1.y;
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int T> void run(const char* name, int per_pattern, int blocks) {
  float* out; long long* cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("[%s] ", name); fflush(stdout);
  probe<T><<<blocks, 256>>>(out, 10, cyc);
  if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return; }
  hipEventRecord(e0);
  probe<T><<<blocks, 256>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 32;  // patterns
  printf("%-44s blocks=%3d: %8.2f ns per pattern (%d instr) = %6.2f ns/instr; s_memtime ticks/pattern %.2f\n", name, blocks, ms * 1e6 / n, per_pattern, ms * 1e6 / n / per_pattern,
         (double)h[0] / n);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {256}) {
    run<0>("8 independent v_add_f32", 8, blocks);
    run<1>("8 dependent v_add_f32", 8, blocks);
    run<2>("8 independent v_exp_f32", 8, blocks);
    run<3>("4 x (v_exp, v_add)", 8, blocks);
    run<4>("8 v_max3_f32 (2 chains)", 8, blocks);
    run<5>("8 v_cvt_pk_bf16_f32", 8, blocks);
    run<6>("8 MFMA 32x32x16 bf16 (AGPR acc)", 8, blocks);
    run<7>("4 x (MFMA + add exp max3 cvt)", 20, blocks);
    run<8>("4 x (MFMA + add exp)", 12, blocks);
    run<9>("4 x (MFMA + 4 add)", 20, blocks);
    run<10>("4 x (MFMA + 6 add)", 28, blocks);
    run<11>("4 x (MFMA f16 VGPR acc + add exp)", 12, blocks);
    run<12>("8 ds_read_b128 + wait", 9, blocks);
    run<13>("4 MFMA, 1 chain", 4, blocks);
    run<14>("4 MFMA, 2 chains", 4, blocks);
    run<15>("6 MFMA, 3 chains", 6, blocks);
    run<16>("4 MFMA f16 VGPR acc, 2 chains", 4, blocks);
    run<17>("4 x (MFMA f16 VGPR 2 chains + add exp)", 12, blocks);
    run<18>("4 MFMA: 2 VGPR f16 chains x 2 AGPR chains", 4, blocks);
  }
  return 0;
}
