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
  int n_tiles;            // tiles_x * tiles_y * batch (set by launch_stem7)
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
  static constexpr int SMEM_BYTES = (2 * IN_FLOATS + W_FLOATS) * 4;      // two tile buffers (persistent workgroups)
};

// OUT16 (f16 mode): x1 is written as fp16 (its consumer, down1, rounds it to fp16 anyway): half the bytes of the largest early tensor.
//
// Round 5: PERSISTENT workgroups (BSR_STEM_PERSIST, default 1; 0 = one workgroup per tile as before — the same kernel with a grid of
// one tile each).  The 7 taps' weights are 25 KB (fp32) / 32 KB (split) per workgroup against 10.6 KB of image tile: with one tile per
// workgroup three quarters of what a workgroup staged was the same weights again, 4 096 times per launch at B = 32.  A workgroup now
// stages them once and walks tiles b, b + grid, ...: the NEXT tile's 11 image words per thread are requested before the current tile's
// matrix loop and written (split) into the other of two LDS tile buffers after its stores — one LDS-only barrier per tile, no drain
// of the outstanding stores.  Same fragments, same accumulation order: bit-identical outputs.
#ifndef BSR_STEM_PERSIST
#define BSR_STEM_PERSIST 1
#endif

template <int RW, int H = 0, bool OUT16 = false>
__global__ __launch_bounds__(256, 2) void stem7_kernel(StemArgs p) {
  using C = StemCfg<RW, H>;
  constexpr int ROWF = C::ROWF;
  constexpr int NE = (C::IN_FLOATS + 255) / 256;              // image words per thread and tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w = smem;                    // weights first: 16-byte aligned rows for ds_read_b128
  float* s_in0 = smem + C::W_FLOATS;    // two tile buffers

  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, r = lane & 31;
  const int per_img = p.tiles_x * p.tiles_y;
  const int total = p.n_tiles;

  // this thread's words of a tile: word i = tid + 256 e -> tile row i / ROWF, float f = i % ROWF = 3 px + c
  int e_goff[NE];                       // offset in floats from the tile's origin pixel (y0, x0), channel 0
  int e_rp[NE];                         // row | px << 8 | in-range << 16
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int i = tid + e * 256;
    const int row = i / ROWF, f = i % ROWF, px = f / 3, c = f % 3;
    e_goff[e] = ((row - 3) * p.W + (px - 3)) * 3 + c;
    e_rp[e] = row | (px << 8) | ((i < C::IN_FLOATS && row < C::IH) ? (1 << 16) : 0);
  }
  auto tile_origin = [&](int t, int& img, int& y0, int& x0) {
    const int tile_x = t % p.tiles_x, rest = t / p.tiles_x;
    x0 = tile_x * C::TW;
    y0 = (rest % p.tiles_y) * C::TH;
    img = rest / p.tiles_y;
  };
  auto load_tile = [&](int t, float (&v)[NE]) {               // zero outside the image: TF SAME padding 3/3
    int img, y0, x0;
    tile_origin(t, img, y0, x0);
    const float* org = p.in + ((size_t)img * p.H * p.W + (size_t)y0 * p.W + x0) * 3;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int row = e_rp[e] & 0xFF, px = (e_rp[e] >> 8) & 0xFF;
      const int iy = y0 - 3 + row, ix = x0 - 3 + px;
      const bool ok = (e_rp[e] >> 16) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      v[e] = ok ? org[e_goff[e]] : 0.f;
    }
  };
  auto write_tile = [&](float* s_dst, const float (&v)[NE]) {
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int i = tid + e * 256;
      if (i < C::IN_FLOATS) {
        if constexpr (H == 0) {
          s_dst[i] = v[e];
        } else {
          amax = __builtin_fmaxf(__builtin_fabsf(v[e]), amax);
          const _Float16 vh = (_Float16)v[e], vl = (_Float16)(v[e] - (float)vh);
          reinterpret_cast<unsigned*>(s_dst)[i] = (unsigned)__builtin_bit_cast(unsigned short, vh) | ((unsigned)__builtin_bit_cast(unsigned short, vl) << 16);
        }
      }
    }
    if constexpr (H != 0) range_report(amax, p.range_flag);
  };

  int t = blockIdx.x;
  if (t >= total) return;
  // stage the weights (linear copy, once per workgroup) and the first tile
  float stg[NE];
  load_tile(t, stg);
  for (int i = tid; i < C::W_FLOATS / 4; i += 256) reinterpret_cast<f32x4*>(s_w)[i] = reinterpret_cast<const f32x4*>(p.w)[i];
  write_tile(s_in0, stg);
  const float bias = p.bias[r];
  __syncthreads();
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int cur = 0;
  for (; t < total; t += gridDim.x) {
  const float* s_in = s_in0 + cur * C::IN_FLOATS;
  const int tn = t + (int)gridDim.x;
  const bool has_next = tn < total;                              // uniform over the workgroup
  if (has_next) load_tile(tn, stg);                              // in flight during this tile's matrix loop
  int img, y0, x0;
  tile_origin(t, img, y0, x0);
  f32x16 acc[RW];
#pragma unroll
  for (int mi = 0; mi < RW; ++mi) acc[mi] = bias_tile(h, bias);      // one matrix instruction per tile (igemm_conv.h)
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
    if constexpr (OUT16 && BSR_H16_PACK_STORES) {      // two channels per lane (igemm_h16.h: pack_pair_f16): the odd lane writes the next pixel of a pair
      const bool odd = (r & 1) != 0;
      const unsigned lane_pk = ((unsigned)(4 * h + (odd ? 1 : 0)) * 32u + (unsigned)(r & ~1)) * 2u;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const int k = (i & 3) + 8 * (i >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(pack_pair_f16(v[i], v[i + 1], odd), orsrc, lane_pk, ((unsigned)(mi * p.W + k) * 32u) * 2u, 0);
      }
    } else {
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
  if (!has_next) break;
  write_tile(s_in0 + (cur ^ 1) * C::IN_FLOATS, stg);             // that buffer was last read one tile ago, behind the previous barrier
  // LDS-only barrier: __syncthreads() would also wait for this tile's stores to be acknowledged (vmcnt 0) — a memory round trip per tile
  __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));
  __builtin_amdgcn_s_barrier();
  cur ^= 1;
  __builtin_amdgcn_s_setprio(0);
  }
}

template <int RW, int H = 0, bool OUT16 = false>
inline hipError_t launch_stem7(StemArgs a, int batch, hipStream_t stream) {
  using C = StemCfg<RW, H>;
  a.tiles_x = a.W / C::TW;
  a.tiles_y = a.H / C::TH;
  a.n_tiles = a.tiles_x * a.tiles_y * batch;
  const int resident = 2 * device_cu_count();                    // __launch_bounds__(256, 2): two workgroups per CU
  const int grid = BSR_STEM_PERSIST ? (a.n_tiles < resident ? a.n_tiles : resident) : a.n_tiles;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (C::SMEM_BYTES > 48 * 1024 && (dev < 0 || !once.done[dev])) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem7_kernel<RW, H, OUT16>), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  hipLaunchKernelGGL((stem7_kernel<RW, H, OUT16>), dim3(grid), dim3(256), C::SMEM_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace bsr
