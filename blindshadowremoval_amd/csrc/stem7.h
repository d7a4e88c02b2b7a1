// Stem: conv1 = Conv(32, 7x7) + BN + LeakyReLU over the 3-channel image (/root/reference/model.py:203,230), Cin = 3.
//
// K = 7 x 7 x 3 = 147 is too thin per tap for the generic channel-chunk kernel, but in NHWC a pixel's 7 horizontal taps x 3
// channels are 21 CONTIGUOUS floats of the image row.  The raw (TH+6) x (TW+6) x 3 tile is staged once in LDS and the
// A fragment of pixel tx, K group g is simply the 4 floats at row + 3*tx + 8g + 4h (+j): the "im2row" is only an
// address pattern (stride 3 floats per lane: conflict-free ds_read_b32), no 24-channel expansion buffer exists.
// GEMM per vertical tap ky: M = pixels, N = 32, K = 21 -> 24 (the 3 pad columns read the next pixel's data and meet
// zero weights).  All 7 taps' weights (25 KB) are staged once, so the MFMA stream has no barrier inside.
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_conv.h"
#include "igemm_h16.h"

namespace bsr {

struct StemArgs {
  const float* in;     // [B,H,W,3]
  float* out;          // [B,H,W,32]
  const float* w;      // packed [1][7][32][28]  (taps = ky, K index = kx*3 + c, 21 real of 24)
  const float* bias;   // [32]
  int H, W;
  int tiles_x, tiles_y;
  unsigned* range_flag;   // H = 2: set when an image value does not fit fp16 (igemm_h16.h); OUT16: when an output does not; may be null
};

// H = 2 (split precision, igemm_h16.h): every image value is split ONCE at staging time and kept in LDS as one 32-bit word
// (hi fp16 | lo fp16 << 16); the A fragment of pixel tx, K step s is the 8 words at row + 3 tx + 16 s + 8 h (same stride-3
// conflict-free ds_read_b32 pattern), unzipped into a hi and a lo f16x8 by four v_perm_b32 each.  K = 21 -> 32 (two
// 32x32x16 steps per vertical tap), weights = pack_taps_h16's [7][32][36-word] image.
template <int RW, int H = 0>   // image rows per wave
struct StemCfg {
  static constexpr int TH = 4 * RW, TW = 32, IH = TH + 6, ROWF = 120;   // 38 pixels x 3 floats = 114 used, reads reach 116 (H = 2: 124)
  static constexpr int IN_FLOATS = IH * ROWF + (H ? 8 : 0);
  static constexpr int W_FLOATS = H ? 7 * 32 * 36 : 7 * 32 * 28;
  static constexpr int SMEM_BYTES = (IN_FLOATS + W_FLOATS) * 4;
};

// OUT16 (f16 mode): x1 is written as fp16 (its consumer, down1, rounds it to fp16 anyway): half the bytes of the largest early tensor.
template <int RW, int H = 0, bool OUT16 = false>
__global__ __launch_bounds__(256, 2) void stem7_kernel(StemArgs p) {
  using C = StemCfg<RW, H>;
  constexpr int ROWF = C::ROWF;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w = smem;                    // weights first: 16-byte aligned rows for ds_read_b128
  float* s_in = smem + C::W_FLOATS;

  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, r = lane & 31;
  int bid = blockIdx.x;
  const int tile_x = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int tile_y = bid % p.tiles_y;
  const int img = bid / p.tiles_y;
  const int y0 = tile_y * C::TH, x0 = tile_x * C::TW;
  const float* in_img = p.in + (size_t)img * p.H * p.W * 3;

  // stage the weights (linear copy) and the raw image tile (zero outside the image: TF SAME padding 3/3)
  for (int i = tid; i < C::W_FLOATS / 4; i += 256) reinterpret_cast<f32x4*>(s_w)[i] = reinterpret_cast<const f32x4*>(p.w)[i];
  float amax = 0.f;
  for (int i = tid; i < C::IN_FLOATS; i += 256) {
    const int row = i / ROWF, f = i % ROWF;
    const int px = f / 3, c = f % 3;
    const int iy = y0 - 3 + row, ix = x0 - 3 + px;
    float v = 0.f;
    if (row < C::IH && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = in_img[((size_t)iy * p.W + ix) * 3 + c];
    if constexpr (H == 0) {
      s_in[i] = v;
    } else {
      amax = __builtin_fmaxf(__builtin_fabsf(v), amax);
      const _Float16 vh = (_Float16)v, vl = (_Float16)(v - (float)vh);
      reinterpret_cast<unsigned*>(s_in)[i] = (unsigned)__builtin_bit_cast(unsigned short, vh) | ((unsigned)__builtin_bit_cast(unsigned short, vl) << 16);
    }
  }
  if constexpr (H != 0) range_report(amax, p.range_flag);
  const float bias = p.bias[r];
  f32x16 acc[RW];
#pragma unroll
  for (int mi = 0; mi < RW; ++mi) acc[mi] = bias_tile(h, bias);      // one matrix instruction per tile (igemm_conv.h)
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);

  if constexpr (H == 0) {
  const int a_base = (wave * RW) * ROWF + 3 * r + 4 * h;      // + (mi + ky) * ROWF + 8g + j
  const int b_base = r * 28 + 4 * h;                            // + ky * 32 * 28 + 8g
#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const f32x4 bf = *reinterpret_cast<const f32x4*>(s_w + b_base + ky * 32 * 28 + g * 8);
      float af[RW][4];
#pragma unroll
      for (int mi = 0; mi < RW; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) af[mi][j] = s_in[a_base + (mi + ky) * ROWF + g * 8 + j];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < RW; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[j], acc[mi], 0, 0, 0);
    }
  }
  } else {
  const int a_base = (wave * RW) * ROWF + 3 * r + 8 * h;      // + (mi + ky) * ROWF + 16 s + j
  const int b_base = r * 36 + 4 * h;                            // + ky * 32 * 36 + 8 s (hi), + 16 (lo)
  const unsigned* s_pk = reinterpret_cast<const unsigned*>(s_in);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const f16x8 bh = *reinterpret_cast<const f16x8*>(s_w + b_base + ky * 32 * 36 + ks * 8);
      const f16x8 bl = *reinterpret_cast<const f16x8*>(s_w + b_base + ky * 32 * 36 + ks * 8 + 16);
#pragma unroll
      for (int mi = 0; mi < RW; ++mi) {
        unsigned wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = s_pk[a_base + (mi + ky) * ROWF + ks * 16 + j];
        u32x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hi[j] = __builtin_amdgcn_perm(wv[2 * j + 1], wv[2 * j], 0x05040100u);      // low halves of two words
          lo[j] = __builtin_amdgcn_perm(wv[2 * j + 1], wv[2 * j], 0x07060302u);      // high halves
        }
        const f16x8 ah = __builtin_bit_cast(f16x8, hi), al = __builtin_bit_cast(f16x8, lo);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mi], 0, 0, 0);
      }
    }
  }
  }

  __builtin_amdgcn_s_setprio(3);
  // raw-buffer stores (igemm_conv.h): SGPR offset per element, one constant lane offset
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  constexpr unsigned OB = OUT16 ? 2u : 4u;
  const unsigned lane_out = ((unsigned)(4 * h) * 32u + (unsigned)r) * OB;
  const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(reinterpret_cast<const float*>(
      reinterpret_cast<const char*>(p.out) + (((size_t)img * p.H + y0 + wave_u * RW) * p.W + x0) * 32 * OB));
#pragma unroll
  for (int mi = 0; mi < RW; ++mi) {
    f32x16 v = acc[mi];
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const f32x2 y = leaky_relu2(f32x2{v[i], v[i + 1]});
      v[i] = y[0];
      v[i + 1] = y[1];
    }
    if constexpr (OUT16) {
      float om = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i += 4) om = amax4(f32x4{v[i], v[i + 1], v[i + 2], v[i + 3]}, om);
      range_report(om, p.range_flag);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = (i & 3) + 8 * (i >> 2);
      if constexpr (OUT16)
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (_Float16)v[i]), orsrc, lane_out, ((unsigned)(mi * p.W + k) * 32u) * OB, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i]), orsrc, lane_out, ((unsigned)(mi * p.W + k) * 32u) * OB, 0);
    }
  }
}

template <int RW, int H = 0, bool OUT16 = false>
inline hipError_t launch_stem7(StemArgs a, int batch, hipStream_t stream) {
  using C = StemCfg<RW, H>;
  a.tiles_x = a.W / C::TW;
  a.tiles_y = a.H / C::TH;
  hipLaunchKernelGGL((stem7_kernel<RW, H, OUT16>), dim3(a.tiles_x * a.tiles_y * batch), dim3(256), C::SMEM_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace bsr
