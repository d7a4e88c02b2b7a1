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

// ---- 16-bit matrix-core modes: operand split and range guard (used by igemm_h16.h, gemm_nloop.h, gemm_tail.h, attention_x3.h, conv_n16.h, stem7.h) ----
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));


// hi = fp16(x) (round to nearest even), lo = fp16(x - hi).  Written on 2-wide vectors so that gfx950 selects its packed
// conversions: v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32 = 5 VALU instructions per pair of values
// (9 element by element).  The split runs beside the fp16 matrix stream and is what the 16-bit kernels are bound by once the
// matrix work has shrunk by 16/3, so it is kept as lean as the ISA allows.
__device__ __forceinline__ void split2(const f32x2 x, f16x2& hi, f16x2& lo) {
  hi = __builtin_convertvector(x, f16x2);
  lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x2), f16x2);
}
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo) {
  f16x2 h[4], l[4];
  split2(f32x2{a[0], a[1]}, h[0], l[0]);
  split2(f32x2{a[2], a[3]}, h[1], l[1]);
  split2(f32x2{b[0], b[1]}, h[2], l[2]);
  split2(f32x2{b[2], b[3]}, h[3], l[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[2 * i] = h[i][0]; hi[2 * i + 1] = h[i][1];
    lo[2 * i] = l[i][0]; lo[2 * i + 1] = l[i][1];
  }
}

// Range guard of the 16-bit modes.  fp16(x) is inf for |x| >= 65520 (round to nearest even; 65504 is the largest finite value), where
// the fp32 path stays finite.  Every kernel that converts fp32 activations keeps the running maximum of their magnitudes — one
// v_max3_f32 with |.| source modifiers per PAIR of values, beside the 5 instructions per pair of the split itself — and a thread that
// saw an overflowing value stores 1 to the handle's sticky flag (host-visible memory: bsr_check_range / the next bsr_forward report
// BSR_ERR_RANGE).  NaN inputs are not flagged (v_max drops them); they propagate to the outputs visibly.
constexpr float kF16Overflow = 65520.f;
__device__ __forceinline__ float amax8(const f32x4& a, const f32x4& b, float m) {
  m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a[0]), __builtin_fabsf(a[1])), m);
  m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a[2]), __builtin_fabsf(a[3])), m);
  m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(b[0]), __builtin_fabsf(b[1])), m);
  m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(b[2]), __builtin_fabsf(b[3])), m);
  return m;
}
__device__ __forceinline__ float amax4(const f32x4& a, float m) {
  m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a[0]), __builtin_fabsf(a[1])), m);
  return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a[2]), __builtin_fabsf(a[3])), m);
}
__device__ __forceinline__ void range_report(float m, unsigned* flag) {
  if (m >= kF16Overflow && flag != nullptr) *flag = 1u;
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
  int out2_split;       // gemm_nloop_kernel<.., H = 2>: out2 is the SPLIT qkv buffer of attention_h16.h — per pixel 3 x [128 hi halves | 128 lo halves], theta x log2 e
#ifdef BSR_STAMPS
  unsigned long long* stamps;   // diagnostic build only: [block][wave][4] s_memtime stamps
  unsigned long long* stamps3;  // diagnostic build only: where the prologue's cycles go, [block][wave][6] = entry, addresses set up, loads issued, loads landed + LDS written, barrier passed, accumulators + first fragments ready
  unsigned long long* stamps2;  // diagnostic build only: per-step timeline of 64 mid-kernel workgroups (1024 .. 1087): [block][wave][step][3] = step start, matrix work done, barrier passed
#endif
};

}  // namespace bsr
