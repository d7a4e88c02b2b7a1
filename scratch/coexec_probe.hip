// Diagnostic: do f32 MFMA and f32 VALU (v_pk_fma_f32) execute concurrently on one SIMD?  Waves 0-3 of a 512-thread
// block run an MFMA loop, waves 4-7 a packed-FMA loop (modes: both, MFMA only, VALU only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  float x = threadIdx.x * 0.001f + 0.5f, y = 0.25f + threadIdx.x * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float res = 0.f;
  if (wave < 4) {
    if (mode == 5 || mode == 6) {   // round 3: only TWO (mode 5) / ONE (mode 6) accumulator chains, as a conv tap with NI = 2 / 1 issues them
      f32x16 a0 = {0}, a1 = {0};
      for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        if (mode == 5) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0); else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a0, 0, 0, 0);
        if (mode == 5) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a1, 0, 0, 0); else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a0, 0, 0, 0);
      }
      res = a0[0] + a1[1];
    } else if (mode >= 3) {          // round 3: the same FLOPs as 16x16x4 instructions (32-cycle issue): does a partner's VALU stream get twice the slots?
      f32x4 c[8];
      for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(i & 1 ? x : y, i & 2 ? x : y, c[i], 0, 0, 0);
      }
      for (int i = 0; i < 8; ++i) res += c[i][i & 3];
    } else if (mode != 2) {
      f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
      for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
      }
      res = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else {
    if (mode != 1 && mode < 4) {
      f32x2 v[8];
      for (int i = 0; i < 8; ++i) v[i] = f32x2{x + i, y - i};
      const f32x2 m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
      for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * m + c;     // v_pk_fma_f32: 4 FLOP per lane
      }
      for (int i = 0; i < 8; ++i) res += v[i][0] + v[i][1];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = res;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 20000;
  const char* names[7] = {"MFMA 32x32x2 + VALU waves", "MFMA 32x32x2 waves only", "VALU waves only", "MFMA 16x16x4 + VALU waves", "MFMA 16x16x4 waves only",
                          "MFMA 32x32x2, 2 acc chains", "MFMA 32x32x2, 1 acc chain"};
  for (int mode = 0; mode < 7; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode, cyc); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256 * 8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double cm = 0, cv = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? cm : cv) += h[b * 8 + w];
    cm /= 1024; cv /= 1024;
    const double n_mfma = (mode == 3 || mode == 4) ? 8.0 : 4.0, fl = (mode == 3 || mode == 4) ? 2048.0 : 4096.0;
    double mf = mode != 2 ? 256.0 * 4 * iters * n_mfma * fl : 0, vf = (mode != 1 && mode < 4) ? 256.0 * 4 * 64 * (double)iters * 4 * 8 * 4 : 0;
    printf("%-28s %.2f ms | MFMA %.1f TFLOP/s (%.0f cycles/MFMA) | VALU %.1f TFLOP/s (%.1f cycles/pk_fma)\n", names[mode], ms, mf / ms / 1e9,
           mode != 2 ? cm / (iters * n_mfma) : 0.0, vf / ms / 1e9, (mode != 1 && mode < 4) ? cv / (iters * 32.0) : 0.0);
  }
  return 0;
}
