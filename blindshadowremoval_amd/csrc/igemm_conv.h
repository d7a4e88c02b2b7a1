// Implicit-GEMM convolution on the CDNA4 fp32 matrix cores (v_mfma_f32_32x32x2_f32), NHWC.
//
// Replaces, for inference, TensorFlow's Conv2D / Conv2DTranspose + FusedBatchNorm + LeakyRelu op chain
// behind the reference's Conv / ConvT / ResBottleneck / NonLocalBlock layers
// (/root/reference/model.py:115-177, 81-113, 6-61).  BatchNorm(training=False) is folded into the
// packed weights/bias offline (blindshadowremoval_amd/pack.py), so the fused epilogue is
//   out = LeakyReLU_0.3( acc + bias [+ residual1 + residual2] ).
//
// GEMM view: M = pixels of one spatial tile (TH x TW), N = output channels, K = taps x input channels.
//  * the input tile + halo of one CC-channel chunk is staged once in LDS ([pixel][CC+4] floats) and every
//    tap reads its A fragments from it at a shifted address — no im2col duplication;
//  * weights are pre-packed [chunk][tap][N][CC+4] (the exact LDS image incl. the bank pad), copied per
//    tap through registers into a double-buffered LDS slot while the previous tap's MFMAs run;
//  * a wave computes 32*MI pixels x 32*NI channels; lanes 0-31 / 32-63 read channels 8g+0..3 / 8g+4..7 of
//    a K-group with one ds_read_b128 each and issue 4 MFMAs (the K order inside a group is permuted
//    identically for A and B, which a sum over K does not care about);
//  * Conv2DTranspose(3, stride 2, SAME) is computed as 4 output-parity phases that share one
//    (TH+1)x(TW+1) input tile: tap (a,b) of the 3x3 kernel feeds phase (a==1, b==1) from input pixel
//    (i - (a==2), j - (b==2)) — 9 taps in total, no zero-insertion waste.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kLeakyAlpha = 0.3f;   // tf.keras.layers.LeakyReLU default (/root/reference/model.py:130,161)

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
  const float* res1;    // optional residuals, NHWC at the OUTPUT resolution, added before the activation
  int res1_cs, res1_c;  // channel stride; channels [0,res1_c) are read
  const float* res2;
  int res2_cs, res2_c;
  int tiles_x, tiles_y; // M tiles per image
};

template <int KH, int KW, int S, bool TR, int TH, int TW, int WM, int WN, int MI, int NI, int CC, bool PF_IN>
struct ConvCfg {
  static constexpr int T = KH * KW;
  static constexpr int IH = TR ? TH + 1 : (TH - 1) * S + KH;
  static constexpr int IW = TR ? TW + 1 : (TW - 1) * S + KW;
  static constexpr int LDP = CC + 4;
  static constexpr int BN = WN * NI * 32;
  static constexpr int BM = WM * MI * 32;
  static constexpr int NPH = TR ? 4 : 1;
  static constexpr int IN_FLOATS = IH * IW * LDP;
  static constexpr int W_FLOATS = BN * LDP;
  static constexpr int SMEM_BYTES = (IN_FLOATS + 2 * W_FLOATS) * 4;
  static constexpr int IN_V4 = IH * IW * (CC / 4);               // float4 loads per input-tile chunk
  static constexpr int IN_PER_THREAD = (IN_V4 + 255) / 256;
  static constexpr int W_V4 = W_FLOATS / 4;
  static constexpr int W_PER_THREAD = (W_V4 + 255) / 256;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(BM == TH * TW, "M tile must equal the spatial tile");
  static_assert(CC % 8 == 0, "channel chunk must be a multiple of the 8-wide K group");
  static_assert(!TR || (KH == 3 && KW == 3 && S == 1), "transposed path is ConvT(3, stride 2)");
};

template <int KH, int KW, int S, bool TR, int TH, int TW, int WM, int WN, int MI, int NI, int CC, bool PF_IN>
__global__ __launch_bounds__(256, 2) void igemm_conv_kernel(ConvArgs p) {
  using C = ConvCfg<KH, KW, S, TR, TH, TW, WM, WN, MI, NI, CC, PF_IN>;
  constexpr int T = C::T, IW = C::IW, LDP = C::LDP, BN = C::BN, NPH = C::NPH;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + C::IN_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, r = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

  int bid = blockIdx.x;
  const int tile_x = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int tile_y = bid % p.tiles_y;
  const int img = bid / p.tiles_y;
  const int n0 = blockIdx.y * BN;
  const int y0 = tile_y * TH, x0 = tile_x * TW;
  const int iy0 = TR ? y0 - 1 : y0 * S - p.pad_t;
  const int ix0 = TR ? x0 - 1 : x0 * S - p.pad_l;
  const float* in_img = p.in + (size_t)img * p.H * p.W * p.in_cs + p.in_coff;

  int a_base[MI], b_base[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int px = (wm * MI + mi) * 32 + r;
    const int ty = px / TW, tx = px % TW;
    a_base[mi] = ((ty * S) * IW + tx * S) * LDP + 4 * h;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b_base[ni] = ((wn * NI + ni) * 32 + r) * LDP + 4 * h;

  f32x16 acc[NPH][MI][NI];
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ph][mi][ni][i] = 0.f;

  // global -> register fetch of one input-tile chunk.  Loads are UNCONDITIONAL (clamped address) so hipcc
  // keeps them in flight across the MFMAs; out-of-image pixels (TF SAME zero padding) are zeroed at store time
  // from a per-thread validity bitmask.
  auto fetch_in = [&](int ch, f32x4 (&regs)[C::IN_PER_THREAD], unsigned& okmask) {
    okmask = 0u;
#pragma unroll
    for (int i = 0; i < C::IN_PER_THREAD; ++i) {
      int idx = tid + i * 256;
      idx = idx < C::IN_V4 ? idx : C::IN_V4 - 1;
      const int pix = idx / (CC / 4), q = idx % (CC / 4);
      const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
      const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
      regs[i] = *reinterpret_cast<const f32x4*>(in_img + ((size_t)iyc * p.W + ixc) * p.in_cs + ch * CC + q * 4);
      okmask |= (ok ? 1u : 0u) << i;
    }
  };
  auto store_in = [&](const f32x4 (&regs)[C::IN_PER_THREAD], unsigned okmask) {
#pragma unroll
    for (int i = 0; i < C::IN_PER_THREAD; ++i) {
      const int idx = tid + i * 256;
      if (idx < C::IN_V4) {
        const int pix = idx / (CC / 4), q = idx % (CC / 4);
        f32x4 v = regs[i];
        if (!((okmask >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(s_in + pix * LDP + q * 4) = v;
      }
    }
  };
  auto fetch_w = [&](int step, f32x4 (&regs)[C::W_PER_THREAD]) {
    const float* src = p.w + ((size_t)step * p.n_pad + n0) * LDP;
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i) {
      int idx = tid + i * 256;
      idx = idx < C::W_V4 ? idx : C::W_V4 - 1;
      regs[i] = *reinterpret_cast<const f32x4*>(src + idx * 4);
    }
  };
  auto store_w = [&](int buf, const f32x4 (&regs)[C::W_PER_THREAD]) {
    float* dst = s_w + buf * C::W_FLOATS;
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i) {
      const int idx = tid + i * 256;
      if (idx < C::W_V4) *reinterpret_cast<f32x4*>(dst + idx * 4) = regs[i];
    }
  };

  f32x4 in_regs[C::IN_PER_THREAD];
  f32x4 w_regs[C::W_PER_THREAD];
  unsigned in_ok = 0u;

  // prologue: chunk 0 input tile + tap 0 weights
  fetch_in(0, in_regs, in_ok);
  fetch_w(0, w_regs);
  store_in(in_regs, in_ok);
  store_w(0, w_regs);
  __syncthreads();

  for (int ch = 0; ch < p.nchunk; ++ch) {
    const bool more = ch + 1 < p.nchunk;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int buf = t & 1;
      const bool last_tap = (t == T - 1);
      // prefetch what the next step needs while this step's MFMAs run
      if (!last_tap) {
        fetch_w(ch * T + t + 1, w_regs);
      } else if (more) {
        fetch_w((ch + 1) * T, w_regs);
        if (PF_IN) fetch_in(ch + 1, in_regs, in_ok);
      }

      // ---- MFMAs of tap t ----
      int tap_off, ph;
      if (TR) {
        const int a = t / 3, b = t % 3;
        tap_off = (((a == 2) ? 0 : 1) * IW + ((b == 2) ? 0 : 1)) * LDP;
        ph = ((a == 1) ? 2 : 0) + ((b == 1) ? 1 : 0);
      } else {
        tap_off = ((t / KW) * IW + (t % KW)) * LDP;
        ph = 0;
      }
      const float* wb = s_w + buf * C::W_FLOATS;
      // software-pipelined fragments: group g+1's ds_read_b128s are issued before group g's MFMAs
      f32x4 af[2][MI], bf[2][NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[0][mi] = *reinterpret_cast<const f32x4*>(s_in + a_base[mi] + tap_off);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[0][ni] = *reinterpret_cast<const f32x4*>(wb + b_base[ni]);
#pragma unroll
      for (int g = 0; g < CC / 8; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g + 1 < CC / 8) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) af[nxt][mi] = *reinterpret_cast<const f32x4*>(s_in + a_base[mi] + tap_off + (g + 1) * 8);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) bf[nxt][ni] = *reinterpret_cast<const f32x4*>(wb + b_base[ni] + (g + 1) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[ph][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][mi][j], bf[cur][ni][j], acc[ph][mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }

      // ---- publish the prefetched data ----
      if (!last_tap) {
        store_w(buf ^ 1, w_regs);
        __syncthreads();
      } else if (more) {
        __syncthreads();                 // every wave is done reading s_in / both weight slots
        if (!PF_IN) fetch_in(ch + 1, in_regs, in_ok);
        store_in(in_regs, in_ok);
        store_w(0, w_regs);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: bias (+ residuals) + LeakyReLU, NHWC store (32 consecutive channels per half-wave) ----
  const size_t out_img = (size_t)img * p.Ho * p.Wo;
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + (wn * NI + ni) * 32 + r;
        const bool n_ok = n < p.n_store;
        const float bias = n_ok ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
          const int px = (wm * MI + mi) * 32 + row;
          const int ty = px / TW, tx = px % TW;
          int oy, ox;
          if (TR) {
            oy = 2 * (y0 + ty) + (ph >> 1);
            ox = 2 * (x0 + tx) + (ph & 1);
          } else {
            oy = y0 + ty;
            ox = x0 + tx;
          }
          const size_t opix = out_img + (size_t)oy * p.Wo + ox;
          float v = acc[ph][mi][ni][i] + bias;
          if (p.res1 != nullptr && n < p.res1_c) v += p.res1[opix * p.res1_cs + n];
          if (p.res2 != nullptr && n < p.res2_c) v += p.res2[opix * p.res2_cs + n];
          if (p.act) v = v >= 0.f ? v : v * kLeakyAlpha;
          if (n_ok) p.out[opix * p.out_cs + p.out_coff + n] = v;
        }
      }
}

template <int KH, int KW, int S, bool TR, int TH, int TW, int WM, int WN, int MI, int NI, int CC, bool PF_IN>
inline hipError_t launch_igemm_conv(ConvArgs a, int batch, hipStream_t stream) {
  using C = ConvCfg<KH, KW, S, TR, TH, TW, WM, WN, MI, NI, CC, PF_IN>;
  auto kern = igemm_conv_kernel<KH, KW, S, TR, TH, TW, WM, WN, MI, NI, CC, PF_IN>;
  static bool attr_set = false;
  if (!attr_set && C::SMEM_BYTES > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int mh = TR ? a.H : a.Ho, mw = TR ? a.W : a.Wo;   // the M grid: input pixels for transposed, output pixels otherwise
  a.tiles_x = mw / TW;
  a.tiles_y = mh / TH;
  dim3 grid(a.tiles_x * a.tiles_y * batch, (a.n_store + C::BN - 1) / C::BN);
  hipLaunchKernelGGL(kern, grid, dim3(256), C::SMEM_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace bsr
