// Shared pieces of the fp32 matrix-core kernels: vector types, the fused-epilogue helpers (LeakyReLU tile, bias tile by one matrix
// instruction), raw-buffer addressing, per-device launcher state and the argument block of the convolution / GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef BSR_PAIR_S2
#define BSR_PAIR_S2 1
#endif

namespace bsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kLeakyAlpha = 0.3f;   // tf.keras.layers.LeakyReLU default (/root/reference/model.py:130,161)

// LeakyReLU(0.3) = max(x, 0.3 x).  fmaxf() costs three VALU instructions per element under the default IEEE mode (multiply,
// a canonicalising v_max x,x, the v_max); as the median of {x, 0.3x, FLT_MAX} it is one v_med3_f32, and the multiply of a
// register pair is one v_pk_mul_f32.  (No inline asm: the hazard recogniser must see these instructions next to MFMAs.)
// VALU time in an epilogue is not hidden: it runs beside a co-resident wave's MFMA stream and is starved by it.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 leaky_relu2(f32x2 x) {
  const f32x2 t = x * kLeakyAlpha;
  return f32x2{__builtin_amdgcn_fmed3f(x[0], t[0], 3.4028234664e38f), __builtin_amdgcn_fmed3f(x[1], t[1], 3.4028234664e38f)};
}
__device__ __forceinline__ float leaky_relu(float x) { return __builtin_amdgcn_fmed3f(x, x * kLeakyAlpha, 3.4028234664e38f); }
// A whole accumulator tile: the eight packed multiplies first, then the sixteen medians (a median right behind the multiply it reads
// costs a wait state — an s_nop, i.e. one more issue slot — each time).  alpha = 1 turns the activation off without a branch
// (median of {x, x, FLT_MAX} = x).
__device__ __forceinline__ void leaky_relu_tile(f32x16& v, float alpha) {
  f32x2 t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = f32x2{v[2 * i], v[2 * i + 1]} * alpha;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[2 * i] = __builtin_amdgcn_fmed3f(v[2 * i], t[i][0], 3.4028234664e38f);
    v[2 * i + 1] = __builtin_amdgcn_fmed3f(v[2 * i + 1], t[i][1], 3.4028234664e38f);
  }
}
// A 32x32 accumulator tile whose rows all hold the bias of its 32 output channels (register i of lane l = bias[l & 31]), produced by
// ONE matrix instruction instead of 16 v_mov_b32: A = [1 | 0] (the k = 0 column all ones: lanes 0-31 supply 1, lanes 32-63 supply 0),
// B row 0 = bias, row 1 = 0, C = 0 -> acc[m][n] = 1 * bias[n] + 0 * 0 + 0 = bias[n] exactly.  Beside a co-resident wave's MFMA stream
// a VALU instruction waits for a matrix instruction to drain every time; the 128 moves of a transposed-conv tile were 2-3 k cycles of
// every workgroup's prologue (and of every channel group of gemm_nloop, there at priority 0), eight matrix instructions are ~0.5 k.
__device__ __forceinline__ f32x16 bias_tile(int h, float bias_col) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float one = h ? 0.f : 1.f;
  asm volatile("" : "+v"(one));        // an empty statement, only to keep the compiler from computing ONE tile and copying it (v_mov) to the others
  return __builtin_amdgcn_mfma_f32_32x32x2f32(one, h ? 0.f : bias_col, zero, 0, 0, 0);
}
// Epilogue addressing through a raw buffer resource: buffer_store_dword v_data, v_lane_off, s[rsrc], s_uniform_off offen.
// The wave-uniform part of an element's address is a 32-bit SGPR byte offset from a per-workgroup base, the per-lane part one
// constant VGPR: no vector address arithmetic at all per element, and a lane whose offset has bit 31 set falls outside
// num_records and is dropped by the hardware (masking without touching exec).
constexpr unsigned kLaneOff = 0x80000000u;          // voffset of a lane that must not store / load
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);   // raw, stride 0, 2 GiB window
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are PER DEVICE: launchers keep their one-time state per
// device ordinal, so one process may hold handles on several GPUs (the deployment rule stays one process per GPU).
constexpr int kMaxDevices = 64;
struct PerDeviceOnce {
  bool done[kMaxDevices] = {};
  int value[kMaxDevices] = {};
  // returns the current device ordinal, or -1 (never cached) when it is out of range / unknown
  static int current() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return -1;
    return d;
  }
};

// compute units of the current device (cached per device ordinal; 256 on MI355X) — launchers that pick a tile size by the grid it gives
inline int device_cu_count() {
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev >= 0 && once.done[dev]) return once.value[dev];
  int cur = 0, cus = 0;
  if (hipGetDevice(&cur) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cur) != hipSuccess || cus <= 0) return 256;
  if (dev >= 0) { once.value[dev] = cus; once.done[dev] = true; }
  return cus;
}

struct ConvArgs {
  const float* in;      // NHWC activations, channel stride in_cs, first channel in_coff
  int in_cs, in_coff;
  int H, W;             // input spatial size per image
  float* out;           // NHWC, channel stride out_cs, first channel out_coff
  int out_cs, out_coff;
  int Ho, Wo;           // output spatial size per image
  const float* w;       // packed [nchunk][T][n_pad][CC+4]
  const float* bias;    // [n_pad]
  int nchunk, n_pad;
  int n_store;          // channels [0, n_store) are written
  int pad_t, pad_l;     // TF SAME pad-before (rows, cols); unused for transposed
  int act;              // 1: LeakyReLU(0.3)
  // --- used by gemm_nloop_kernel only ---
  float* out2;          // optional second destination: channels [n_split, n_store) go to out2[.., n - n_split]
  int out2_cs, n_split; // (n_split is a multiple of 32; channels [n_store1, n_split) of the first range are dropped)
  int n_store1;         // with out2: channels [0, n_store1) go to `out`
  const float* res1;    // optional residual, NHWC at the output resolution, added before the activation
  int res1_cs, res1_c;  // channel stride; channels [0,res1_c) are read
  int tiles_x, tiles_y; // M tiles per image
  int n_blocks;         // igemm_h16_kernel: > 0 = 1-D grid with the N block as the FASTEST index (the N blocks of a tile share its input through L2)
  unsigned* range_flag; // 16-bit kernels: set to 1 when a staged activation does not fit fp16 (|x| >= 65520); may be null
#ifdef BSR_STAMPS
  unsigned long long* stamps;   // diagnostic build only: [block][wave][4] s_memtime stamps
  unsigned long long* stamps3;  // diagnostic build only: where the prologue's cycles go, [block][wave][6] = entry, addresses set up, loads issued, loads landed + LDS written, barrier passed, accumulators + first fragments ready
  unsigned long long* stamps2;  // diagnostic build only: per-step timeline of 64 mid-kernel workgroups (1024 .. 1087): [block][wave][step][3] = step start, matrix work done, barrier passed
#endif
};

}  // namespace bsr
