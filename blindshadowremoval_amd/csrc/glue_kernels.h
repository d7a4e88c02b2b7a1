// Small HBM-bound kernels around the MFMA convolutions: the glue of Generator.call
// (/root/reference/model.py:237,238,246-252,256-259,267-269,288).  fp contraction is disabled so the
// grayscale / gs / dif arithmetic rounds like the reference's separate TF ops (matters at the hard
// 0.1 threshold of model.py:256).
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_conv.h"

#pragma clang fp contract(off)

namespace bsr {

__device__ __forceinline__ float gray3(float r, float g, float b) {
  // tf.image.rgb_to_grayscale: sum(x * [0.2989, 0.5870, 0.1140]) (/root/reference/model.py:250)
  return (r * 0.2989f + g * 0.5870f) + b * 0.1140f;
}

// tf.image.resize(uv, [H/8, W/8]) bilinear, half-pixel centres (model.py:237): for an exact 8x reduction
// the sample point is 8o+3.5, i.e. the mean of the centre 2x2 block.  Writes the 3 channels into two
// concat slots: dst_a[..., coff_a..] (model.py:238) and dst_b[..., coff_b..] (model.py:259).
__global__ void uv_resize8_kernel(const float* __restrict__ uv, int H, int W, float* __restrict__ dst_a, int cs_a, int coff_a,
                                  float* __restrict__ dst_b, int cs_b, int coff_b, size_t ncell) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncell * 3) return;
  const size_t cell = gid / 3;
  const int c = (int)(gid % 3);
  const int h8 = H / 8, w8 = W / 8;
  const int p = (int)(cell % w8), o = (int)((cell / w8) % h8);
  const size_t img = cell / ((size_t)w8 * h8);
  const float* src = uv + ((img * H + 8 * o + 3) * W + 8 * p + 3) * 3 + c;
  const float a = src[0], b = src[3], cc = src[(size_t)W * 3], d = src[(size_t)W * 3 + 3];
  const float v = 0.5f * (0.5f * a + 0.5f * b) + 0.5f * (0.5f * cc + 0.5f * d);
  dst_a[cell * cs_a + coff_a + c] = v;
  dst_b[cell * cs_b + coff_b + c] = v;
}

// Heads epilogue (model.py:246-252).  q [B,H,W,16]: q[y][x'][kx*2+o] = sum_{ky,c} y[y+ky-3][x'][c] * w_o[ky][kx][c]
// (the 7x1 MFMA conv); here the 7 horizontal taps are summed, then
//   mask = tanh(. + b2), con = . + b3, gs = gray(inputs)*(1+mask)+con, mask22 = [relu(mask), 0, relu(-mask)].
__global__ void heads_post_kernel(const float* __restrict__ q, const float* __restrict__ inputs, float b_mask, float b_con,
                                  float* __restrict__ gs, float* __restrict__ mask22, int W, size_t npix) {
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int x = (int)(pix % W);
  float m = 0.f, cn = 0.f;
#pragma unroll
  for (int kx = 0; kx < 7; ++kx) {
    const int sx = x + kx - 3;
    if (sx >= 0 && sx < W) {
      const float2 v = *reinterpret_cast<const float2*>(q + (pix + (size_t)(kx - 3)) * 16 + kx * 2);
      m += v.x;
      cn += v.y;
    }
  }
  const float mask = tanhf(m + b_mask);
  const float con = cn + b_con;
  const float g0 = gray3(inputs[pix * 3], inputs[pix * 3 + 1], inputs[pix * 3 + 2]);
  const float g = g0 * (1.f + mask) + con;
  gs[pix] = g;
  mask22[pix * 3 + 0] = fmaxf(mask, 0.f);
  mask22[pix * 3 + 1] = mask * 0.f;
  mask22[pix * 3 + 2] = fmaxf(-mask, 0.f);
}

// model.py:251,256-259: dif = gs - gray(inputs); d32 = resize(dif, /8) (centre 2x2 mean);
// bmask = d32 > 0.1 (strict); x_hole = x*(1-bmask); xh = cat[x_hole, bmask, uv] (uv slot filled by uv_resize8).
// One workgroup of 64 threads per 32x32 cell; r [B,cells,r_cs] -> xh [B,cells,xh_cs]; probe [cells][2] = {d32, bmask}.
__global__ void bmask_xhole_kernel(const float* __restrict__ gs, const float* __restrict__ inputs, int H, int W,
                                   const float* __restrict__ r, int r_cs, int r_c, float* __restrict__ xh, int xh_cs,
                                   float* __restrict__ probe) {
  const size_t cell = blockIdx.x;
  const int h8 = H / 8, w8 = W / 8;
  const int p = (int)(cell % w8), o = (int)((cell / w8) % h8);
  const size_t img = cell / ((size_t)w8 * h8);
  const size_t p00 = (img * H + 8 * o + 3) * W + 8 * p + 3;
  float d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t px = p00 + (size_t)(k >> 1) * W + (k & 1);
    const float g0 = gray3(inputs[px * 3], inputs[px * 3 + 1], inputs[px * 3 + 2]);
    d[k] = gs[px] - g0;
  }
  const float d32 = 0.5f * (0.5f * d[0] + 0.5f * d[1]) + 0.5f * (0.5f * d[2] + 0.5f * d[3]);
  const float bm = d32 > 0.1f ? 1.f : 0.f;
  const float keep = 1.f - bm;
  for (int c = threadIdx.x; c < r_c; c += blockDim.x) xh[cell * xh_cs + c] = r[cell * r_cs + c] * keep;
  if (threadIdx.x == 0) {
    xh[cell * xh_cs + r_c] = bm;
    probe[cell * 2] = d32;
    probe[cell * 2 + 1] = bm;
  }
}

// ---- TSM ShareLayer (/root/reference/model_with_TSM.py:199-229) = offset warp -> group max|mean -> tile -> inverse warp ----
// tf_batch_map_offsets (/root/reference/warp.py:134-165): offsets = resize(reg, [S,S]) * S (centre-2x2 mean for the exact
// 8x reduction), channels 0:2; coords = offsets + (i, j); clamp to [0, S-1]; corners floor / ceil; lerp along axis 0
// first, then axis 1 (warp.py:111-113).

// reg [B,H,W,6] = reg_in(3) | reg_out(3) -> reg32 [B*cells][4] = (in0, in1, out0, out1) * S
__global__ void reg_resize8_kernel(const float* __restrict__ reg, int H, int W, float* __restrict__ reg32, size_t ncell) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncell * 4) return;
  const size_t cell = gid / 4;
  const int k = (int)(gid % 4);
  const int c = (k & 1) + 3 * (k >> 1);                    // channels 0,1 of reg_in and 3,4 (= 0,1 of reg_out)
  const int h8 = H / 8, w8 = W / 8;
  const int p = (int)(cell % w8), o = (int)((cell / w8) % h8);
  const size_t img = cell / ((size_t)w8 * h8);
  const float* src = reg + ((img * H + 8 * o + 3) * W + 8 * p + 3) * 6 + c;
  const float a = src[0], b = src[6], cc = src[(size_t)W * 6], d = src[(size_t)W * 6 + 6];
  const float v = 0.5f * (0.5f * a + 0.5f * b) + 0.5f * (0.5f * cc + 0.5f * d);
  reg32[gid] = v * (float)h8;
}

struct WarpTaps {
  int i0, i1, j0, j1;
  float o0, o1;
};
__device__ __forceinline__ WarpTaps warp_taps(float c0, float c1, int S) {
  c0 = fminf(fmaxf(c0, 0.f), (float)(S - 1));
  c1 = fminf(fmaxf(c1, 0.f), (float)(S - 1));
  const float f0 = floorf(c0), f1 = floorf(c1);
  WarpTaps t;
  t.i0 = (int)f0; t.i1 = (int)ceilf(c0); t.j0 = (int)f1; t.j1 = (int)ceilf(c1);
  t.o0 = c0 - f0; t.o1 = c1 - f1;
  return t;
}
__device__ __forceinline__ float warp_lerp(float lt, float rt, float lb, float rb, float o0, float o1) {
  const float vt = lt + (rt - lt) * o0;      // (i0,j0) -> (i1,j0)
  const float vb = lb + (rb - lb) * o0;      // (i0,j1) -> (i1,j1)
  return vt + (vb - vt) * o1;
}

// share[g][cell][0..C) = max_f warp(x[g*frame+f], reg_in), [C..2C) = mean_f.  One workgroup per (group, cell).
__global__ void share_reduce_kernel(const float* __restrict__ x, int x_cs, int C, const float* __restrict__ reg32, int S, int frame,
                                    float* __restrict__ share) {
  const int cells = S * S;
  const int cell = blockIdx.x % cells, g = blockIdx.x / cells;
  const int i = cell / S, j = cell % S;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float mx = -INFINITY, sum = 0.f;
    for (int f = 0; f < frame; ++f) {
      const size_t b = (size_t)g * frame + f;
      const float* r = reg32 + (b * cells + cell) * 4;
      const WarpTaps t = warp_taps(r[0] + (float)i, r[1] + (float)j, S);
      const float* xb = x + b * cells * x_cs + c;
      const float v = warp_lerp(xb[(size_t)(t.i0 * S + t.j0) * x_cs], xb[(size_t)(t.i1 * S + t.j0) * x_cs],
                                xb[(size_t)(t.i0 * S + t.j1) * x_cs], xb[(size_t)(t.i1 * S + t.j1) * x_cs], t.o0, t.o1);
      mx = fmaxf(mx, v);
      sum += v;
    }
    float* o = share + ((size_t)g * cells + cell) * 2 * C;
    o[c] = mx;
    o[C + c] = sum / (float)frame;
  }
}

// out[b][cell][coff .. coff+2C) = warp(share[b / frame], reg_out[b])   (share != 0), or cat[x, x] (share == 0)
__global__ void share_unwarp_kernel(const float* __restrict__ share, const float* __restrict__ x, int x_cs, int C,
                                    const float* __restrict__ reg32, int S, int frame, int do_share, float* __restrict__ out, int out_cs,
                                    int out_coff) {
  const int cells = S * S;
  const int cell = blockIdx.x % cells;
  const size_t b = blockIdx.x / cells;
  float* o = out + (b * cells + cell) * out_cs + out_coff;
  if (!do_share) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float v = x[(b * cells + cell) * x_cs + c];
      o[c] = v;
      o[C + c] = v;
    }
    return;
  }
  const int i = cell / S, j = cell % S;
  const float* r = reg32 + (b * cells + cell) * 4;
  const WarpTaps t = warp_taps(r[2] + (float)i, r[3] + (float)j, S);
  const float* sb = share + (b / frame) * cells * 2 * C;
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
    o[c] = warp_lerp(sb[(size_t)(t.i0 * S + t.j0) * 2 * C + c], sb[(size_t)(t.i1 * S + t.j0) * 2 * C + c],
                     sb[(size_t)(t.i0 * S + t.j1) * 2 * C + c], sb[(size_t)(t.i1 * S + t.j1) * 2 * C + c], t.o0, t.o1);
  }
}

// out[px][c] = LeakyReLU(x[px][c]) for c in [c0, c1): the channels of a ResBottleneck output beyond the 288 the `w` GEMM
// covers (the wider of x / y is kept, the narrower zero-padded: model.py:105-113; only the TSM widths 291 / 877 get here)
__global__ void lrelu_copy_kernel(const float* __restrict__ x, int x_cs, float* __restrict__ out, int out_cs, int c0, int c1, size_t npix) {
  const int nc = c1 - c0;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= npix * nc) return;
  const size_t px = gid / nc;
  const int c = c0 + (int)(gid % nc);
  const float v = x[px * x_cs + c];
  out[px * out_cs + c] = fmaxf(v, v * kLeakyAlpha);
}

// dense copy of a channel slice of an NHWC buffer (debug probes only)
__global__ void slice_copy_kernel(const float* __restrict__ src, int cs, int coff, int c, float* __restrict__ dst, size_t npix, int src_is_f16) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= npix * c) return;
  const size_t pix = gid / c;
  const int ch = (int)(gid % c);
  dst[gid] = src_is_f16 ? (float)reinterpret_cast<const _Float16*>(src)[pix * cs + coff + ch] : src[pix * cs + coff + ch];
}

// The clock the chip holds while something else runs (bench.py `sustained.clock_ghz`): ONE wave on a side stream samples the shader-cycle
// counter and the constant 100-MHz real-time counter every `spin` x ~3.4 us until `stop` becomes non-zero (or `samples` are taken); the
// host turns consecutive pairs into GHz = d(cycles) / d(ticks) x 0.1.  One wave on one CU: it takes nothing measurable from the forwards
// it runs beside (MI355X_MICROARCH.md, DVFS: the clock under a matrix-dense load is well under the 2.4 GHz the idle chip shows).
__global__ void clock_trace_kernel(unsigned long long* __restrict__ out, int samples, int spin, const int* stop, int* taken) {
  if (threadIdx.x != 0) return;
  int i = 0;
  for (; i < samples; ++i) {
    out[2 * i] = __builtin_amdgcn_s_memtime();
    out[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
    if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ++i; break; }
    for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(127);
  }
  *taken = i;
}

}  // namespace bsr
