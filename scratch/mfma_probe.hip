// Probe: verify the fp32 MFMA 32x32x2 / 16x16x4 lane maps on gfx950 and that a hipcc-7.2 code object
// runs under the HIP runtime torch has loaded.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C[32][32] = A[32][K] * B[K][32], row-major, one wave.
__global__ void probe32(const float* A, const float* B, float* C, int K) {
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) {
    float a = A[r * K + k + h];
    float b = B[(k + h) * 32 + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[row * 32 + r] = acc[i];
  }
}
// C[16][16] = A[16][K]*B[K][16]
__global__ void probe16(const float* A, const float* B, float* C, int K) {
  int l = threadIdx.x; int r = l & 15, q = l >> 4;
  f32x4 acc = {0};
  for (int k = 0; k < K; k += 4) {
    float a = A[r * K + k + q];
    float b = B[(k + q) * 16 + r];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) C[(q * 4 + i) * 16 + r] = acc[i];
}
extern "C" int probe_run(const float* A, const float* B, float* C, int K, int which, void* stream) {
  if (which == 32) hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C, K);
  else hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C, K);
  return (int)hipGetLastError();
}
