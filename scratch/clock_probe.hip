// Diagnostic: shader clock held under a bare fp32 MFMA loop vs MFMA + LDS reads, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* st, int iters, float seed, const f32x4* gbuf, f32x4* gout, size_t gmask) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) { unsigned u = (i + blockIdx.x * 8192) * 2654435761u; u ^= u >> 13; u *= 2246822519u; u ^= u >> 16; lds[i] = ((int)(u & 0xffffff) - 8388608) * (1.0f / 8388608.f); }
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = lds[threadIdx.x], y = lds[threadIdx.x + 256];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    f32x4 f0, f1;
    if (MODE == 2 || MODE == 3) {     // streaming global loads feed the MFMAs (16 B/lane per 16 MFMAs)
      size_t idx = ((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)it * 65536 * 4) & gmask;
      f0 = gbuf[idx];
      f1 = f32x4{f0[1], f0[0], f0[3], f0[2]};
      if (MODE == 3) gout[idx] = f1;
    } else if (MODE == 4) {           // VALU-heavy: 32 extra v_fma per 16 MFMAs
      f0 = f32x4{x, y, x, y}; f1 = f32x4{y, x, y, x};
#pragma unroll
      for (int q = 0; q < 16; ++q) { x = __builtin_fmaf(x, 1.0001f, y * 1e-6f); y = __builtin_fmaf(y, 0.9999f, x * 1e-6f); }
    } else if (MODE >= 5) {   // 6 ds_read_b128 per 16 MFMAs (the conv kernel's ratio), operands mixed from all of them
      f32x4 t[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) t[q] = *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 36 + it * 8 + q * 1336) & 8188));
      f0 = t[0] + t[2] * t[4]; f1 = t[1] + t[3] * t[5];
      if (MODE == 6) __syncthreads();
      if (MODE == 7 && (it & 3) == 0) gout[((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)it * 65536 * 2) & gmask] = t[5];   // ~0.6 TB/s of streaming stores
      if (MODE == 8 && (it & 3) == 0) { f32x4 u = gbuf[((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)it * 65536 * 2) & gmask]; x += u[0] * 1e-30f; }
    } else if (MODE >= 1) {
      f0 = *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 36 + it * 8) & 8188));
      f1 = *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 36 + it * 8 + 4096) & 8188));
    } else { f0 = f32x4{x, y, x, y}; f1 = f32x4{y, x, y, x}; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[j], f1[j], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f1[j], f0[j], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[j], f0[j], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(f1[j], f1[j], a3, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (threadIdx.x == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = r1 - r0; }
}
static f32x4 *gbuf, *gout; static size_t gmask;
template <int MODE> void run(const char* name, int blocks, int iters) {
  float* out; unsigned long long* st; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, st, iters, 0.37f, gbuf, gout, gmask); hipEventRecord(e1); hipEventSynchronize(e1); }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = new unsigned long long[blocks * 2]; hipMemcpy(h, st, blocks * 16, hipMemcpyDeviceToHost);
  double c = 0, r = 0; for (int i = 0; i < blocks; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
  double flops = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("%-28s blocks %5d: %.2f ms, %.1f TFLOP/s, clock %.2f GHz, cycles/MFMA/wave %.1f\n", name, blocks, ms, flops / ms / 1e9, c / r * 0.1, c / blocks / (iters * 16.0));
  hipFree(out); hipFree(st);
}
template <int MODE> void run_short(const char* name, int blocks, int iters, int reps, bool sync_each) {
  float* out; unsigned long long* st; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < reps; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, st, iters, 0.37f, gbuf, gout, gmask); if (sync_each) hipDeviceSynchronize(); }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = new unsigned long long[blocks * 2]; hipMemcpy(h, st, blocks * 16, hipMemcpyDeviceToHost);
  double c = 0, r = 0; for (int i = 0; i < blocks; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
  printf("%-34s %d x %.3f ms (sync_each=%d): last-kernel clock %.2f GHz, %.1f TFLOP/s overall\n", name, reps, ms / reps, (int)sync_each, c / r * 0.1,
         (double)blocks * 4 * iters * 16 * 4096.0 * reps / ms / 1e9);
  hipFree(out); hipFree(st);
}
int main() {
  size_t n = (size_t)1 << 26; gmask = n - 1; hipMalloc(&gbuf, n * 16); hipMalloc(&gout, n * 16); hipMemset(gbuf, 0x3c, n * 16);
  run<0>("bare MFMA, 1 wave/SIMD", 256, 20000);
  run<0>("bare MFMA, 2 waves/SIMD", 512, 20000);
  run<1>("MFMA + 2 ds_read_b128/16, 1w", 256, 20000);
  run<1>("MFMA + 2 ds_read_b128/16, 2w", 512, 20000);
  run<2>("MFMA + global loads, 2w", 512, 5000);
  run<3>("MFMA + global ld+st, 2w", 512, 5000);
  run<4>("MFMA + 32 v_fma/16, 2w", 512, 20000);
  run<5>("MFMA + 6 ds_read/16, 2w", 512, 20000);
  run<6>("MFMA + 6 ds_read/16 + barrier, 2w", 512, 20000);
  run<7>("MFMA + LDS + 0.6TB/s stores, 2w", 512, 20000);
  run<8>("MFMA + LDS + 0.6TB/s loads, 2w", 512, 20000);
  run_short<5>("short MFMA+LDS kernels", 512, 700, 20, false);
  run_short<5>("short MFMA+LDS kernels", 512, 700, 20, true);
  run_short<5>("short MFMA+LDS kernels", 512, 100, 50, false);
  return 0;
}
