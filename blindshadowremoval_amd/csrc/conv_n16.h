// 16-output-channel convolutions on v_mfma_f32_16x16x4_f32 — the two thin, full-resolution layers of the
// generator, where a 32-wide MFMA tile would waste half (or more) of the matrix core:
//   * heads  = conv2 | conv3, 7x7, 64 -> 1 each (/root/reference/model.py:204-205,246-247), run as a 7x1 conv
//     with N = (kx, head) = 14 (-> 16); the 7 horizontal taps are summed, and tanh / gs / mask22 (model.py:246-252) applied, either by
//     heads_post_kernel from a [B,H,W,16] scratch tensor or — FUSE, round 3 — by this kernel itself: a persistent workgroup then
//     walks a whole ROW STRIP of tiles left to right, parks the tile's 14 partial planes in LDS (over the input tile it has
//     finished with), and each output pixel adds its 7 taps in heads_post_kernel's order, the 6 pixels that straddle a tile
//     boundary carrying their partial sums to the next tile — same bits, no 134 MB write + 171 MB re-read + second launch;
//   * clr_conv1 = Conv(16, 3x3) over cat[gs, f] (model.py:217,267), optionally fused with clr_conv2 (1x1 16->16
//     + BN + LeakyReLU), clr_conv3 (1x1 16->3) and dif = gray(con_rgb) - gray(inputs) (model.py:268-269,288).
//
// Transposed GEMM: A = weights (M = 16 output channels), B = pixels (N = 16 pixels), so the accumulator holds
// 4 consecutive channels of one pixel per lane (one 16-byte NHWC store) and is directly the B operand of the
// next 1x1 layer (k index = lane>>4, step = register) — the fused tail needs no LDS round trip.
// K = 64 input channels in two 32-channel chunks; all taps of a chunk are staged per barrier.  The extra
// `gs` input channel of clr_conv1 is one more K group built by an im2col gather from a (TH+2)x(TW+2) LDS tile.
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_conv.h"
#include "igemm_h16.h"

namespace bsr {

struct ConvN16Args {
  const float* in;      // NHWC, 64 channels used, channel stride in_cs
  int in_cs;
  int H, W;
  const float* w;       // packed [2][T][16][36]
  const float* bias;    // [16]
  float* out;           // NHWC, channel stride out_cs (unused when TAIL)
  int out_cs;
  int act;
  int pad_t, pad_l;
  const float* gs;      // GS: [B,H,W,1]
  const float* w_gs;    // GS: [16 n][16 k] (k < 9 = 3x3 taps of the gs channel, rest 0)
  const float* tail_w;  // TAIL: w2[16 k][16 n] | b2[16] | w3[16 k][3 n] | b3[3]
  const float* inputs;  // TAIL: [B,H,W,3]
  float* con_rgb;       // TAIL: [B,H,W,3]
  float* dif;           // TAIL: [B,H,W,1]
  float* packed;        // TAIL: when not null, con_rgb | dif are written as ONE [B,H,W,4] tensor (16-byte store per pixel) instead — the
                        //       payload of the multi-GPU output all-gather (bench.py / dist.py); con_rgb / dif are then not written
  int tiles_x, tiles_y, batch;   // filled by the launcher
  unsigned* range_flag; // H = 2: set when a staged activation does not fit fp16 (igemm_h16.h); may be null
  float* gs_out;        // FUSE: [B,H,W,1] gs = gray(inputs) * (1 + mask) + con   (inputs = the `inputs` field)
  float* mask22;        // FUSE: [B,H,W,3] = [relu(mask), 0, relu(-mask)]
  float b_mask, b_con;  // FUSE: conv2 / conv3 biases
#ifdef BSR_STAMPS
  unsigned long long* stamps;
#endif
};

template <int KH, int KW, bool GS, bool TAIL, int RW, bool FUSE = false>   // RW = tile rows per wave (MFMA work per staged byte)
struct ConvN16Cfg {
  static constexpr int T = KH * KW, TH = 4 * RW, TW = 32, MT = 2 * RW, CC = 32, LDP = 36, G = 2;
  static constexpr int IH = TH + KH - 1, IW = TW + KW - 1;
  static constexpr int IN_FLOATS = IH * IW * LDP;
  static constexpr int W_FLOATS = T * 16 * LDP;
  static constexpr int GS_FLOATS = GS ? (TH + 2) * (TW + 2) + 2 : 0;
  static constexpr int TAIL_FLOATS = TAIL ? 352 + TH * TW * 3 : 0;   // for the variant that keeps them in LDS: the fused tail's weights (337 floats) + the tile's input pixels
  static constexpr int QLD = 20;                                     // FUSE: floats per pixel of the parked partial planes (16 + 4: conflict-free 16-byte stores)
  static constexpr int CARRY_FLOATS = FUSE ? 2 * TH * 6 * 2 : 0;     // FUSE: partial sums of the 6 boundary pixels x 2 heads, double-buffered by tile parity
  static constexpr int SMEM_BYTES = (IN_FLOATS + W_FLOATS + GS_FLOATS + TAIL_FLOATS + CARRY_FLOATS) * 4;
  static_assert(!FUSE || (KH == 7 && KW == 1 && !GS && !TAIL), "FUSE is the heads epilogue");
  static_assert(!FUSE || TH * TW * QLD <= IN_FLOATS, "the parked planes reuse the input tile's LDS");
  static_assert(!FUSE || TH * TW == 256, "FUSE: one output pixel per thread");
  static constexpr int IN_V4 = IH * IW * (CC / 4);
  static constexpr int IN_PER_THREAD = IH + 1;                   // one float4 per tile row + one halo-column load
  static constexpr int W_V4 = W_FLOATS / 4;
  static constexpr int W_PER_THREAD = (W_V4 + 255) / 256;
};

// FUSE: one pixel of the heads' outputs from its two 7-tap sums — the arithmetic of heads_post_kernel (glue_kernels.h), same
// operation order, fp contraction off (gs feeds the hard 0.1 threshold of model.py:256).
__device__ __forceinline__ void heads_emit(const ConvN16Args& p, size_t pix, float m, float cn, const float (&in3)[3]) {
#pragma clang fp contract(off)
  const float mask = tanhf(m + p.b_mask);
  const float con = cn + p.b_con;
  const float g0 = (in3[0] * 0.2989f + in3[1] * 0.5870f) + in3[2] * 0.1140f;      // tf.image.rgb_to_grayscale (model.py:250)
  const float g = g0 * (1.f + mask) + con;
  p.gs_out[pix] = g;
  p.mask22[pix * 3 + 0] = fmaxf(mask, 0.f);
  p.mask22[pix * 3 + 1] = mask * 0.f;
  p.mask22[pix * 3 + 2] = fmaxf(-mask, 0.f);
}

// H = 0: fp32 matrix cores (v_mfma_f32_16x16x4_f32).  H = 2: split precision on v_mfma_f32_16x16x32_f16 (igemm_h16.h): the
// LDS row of a pixel / output channel keeps its 36 words but holds [32 hi halves | 32 lo halves | pad]; activations are split
// when the staged registers are written to LDS, the weight image is pack_taps_h16's, and one tap of a 32-channel chunk is ONE
// K = 32 step of three instructions (hi.hi + hi.lo + lo.hi) instead of eight fp32 ones.  Lane (r = l & 15, q = l >> 4) holds
// k = 8q .. 8q+7 of A and B; C/D is the fp32 instruction's layout, so the gs K group and the fused 1x1 tail stay as they are.
// IN16 (with H = 2, the f16 mode): p.in is an fp16 tensor (in_cs counts halves).  Its values ARE the hi plane — the lo plane and the
// weight-hi x input-lo instruction disappear: 2 instead of 3 matrix instructions per tap, 8-byte loads instead of 16.
template <int KH, int KW, bool GS, bool TAIL, int RW, int H = 0, bool IN16 = false, bool FUSE = false>
__global__ __launch_bounds__(256, 2) void conv_n16_kernel(ConvN16Args p) {
  static_assert(!IN16 || H == 2, "fp16 input belongs to the 16-bit kernel");
  using C = ConvN16Cfg<KH, KW, GS, TAIL, RW, FUSE>;
  constexpr int T = C::T, IW = C::IW, LDP = C::LDP, TW = C::TW, TH = C::TH, MT = C::MT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + C::IN_FLOATS;
  float* s_gs = s_w + C::W_FLOATS;
  float* s_tail = s_gs + C::GS_FLOATS;
  float* s_carry = s_tail + C::TAIL_FLOATS;      // FUSE
  float* s_q = s_in;                             // FUSE: [TH][TW][QLD], valid between the tile's MFMA loop and the next tile's staging
  // The split-precision variant needs ~290 VGPRs with the fused tail's 15 per-lane weights held across the tile loop and spilled them
  // (reloaded per tile from scratch: +45 % HBM traffic); it reads them from a 1.4-KB LDS copy at each epilogue instead.
  constexpr bool TAIL_LDS = TAIL && H == 2 && !IN16;

#ifdef BSR_STAMPS
  unsigned long long st0 = __builtin_amdgcn_s_memtime(), st1 = 0, st_epi = 0, rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  // PERSISTENT workgroups: the grid is two workgroups per CU and each walks tiles blockIdx.x, +gridDim.x, ...  With N = 16
  // the layer moves ~1 staged byte per 300 flop, so a one-tile workgroup spent as long waiting for its first tile as
  // computing (measured: prologue 19-24k cycles vs 25k of MFMA loop); here the NEXT (tile, chunk) is always in flight in
  // registers while the current one is multiplied, across tile boundaries too, and only the first tile pays the latency.
  struct Tile { int img, y0, x0; };
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int ntiles = tiles_per_img * p.batch;
  auto decode = [&](int t) {
    Tile d;
    d.x0 = (t % p.tiles_x) * TW;
    t /= p.tiles_x;
    d.y0 = (t % p.tiles_y) * TH;
    d.img = t / p.tiles_y;
    return d;
  };
  // Work sequence of this workgroup.  Plain: tiles blockIdx.x, + gridDim.x, ...  FUSE: whole row strips (tiles_x consecutive tile
  // indices, x fastest) blockIdx.x, + gridDim.x, ..., each walked left to right, because a tile hands its right-hand boundary sums
  // to its right neighbour through LDS.
  // XCD placement of the strips: workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8) and each XCD has its own L2.
  // A strip re-reads 6 of its 14 input rows from its vertical neighbours' range, so neighbouring strips must run on ONE XCD to share
  // them: virtual strip v -> strip (v % 8) * (nstrips / 8) + v / 8 gives every XCD a contiguous eighth of the strips.  (Dealt in plain
  // order — neighbours on different XCDs — the kernel moved 998 MB per forward instead of the unfused kernel's 272 MB: PMC, round 3.)
  const int nstrips = p.tiles_y * p.batch;
  auto seq = [&](int k) -> int {        // k-th tile of this workgroup, or >= ntiles when there is none
    if constexpr (FUSE) {
      const int v = blockIdx.x + (k / p.tiles_x) * (int)gridDim.x;
      if (v >= nstrips) return ntiles;
      const int strip = (nstrips % 8 == 0) ? (v % 8) * (nstrips / 8) + v / 8 : v;
      return strip * p.tiles_x + k % p.tiles_x;
    } else {
      return blockIdx.x + k * (int)gridDim.x;
    }
  };

  // Input-tile staging with NO per-load vector arithmetic (it would run beside the co-resident workgroup's MFMA stream, where
  // VALU instructions starve): thread t owns float4 column c4 = t & 7 of pixel column t >> 3 and walks the IH tile rows; the
  // row part of the address is a wave-uniform SGPR offset of a raw buffer load, out-of-image columns carry the kLaneOff
  // lane offset (the load returns 0 = TF SAME zero padding), out-of-image rows are a uniform branch.  The KW-1 halo columns
  // right of the 32 are one more load for the first HALO_V4 threads.
  constexpr int HALO_W = IW - 32, HALO_V4 = C::IH * HALO_W * 8;
  static_assert(HALO_V4 <= 256 && (HALO_W == 0 || HALO_W == 2), "halo pass: one load per thread, power-of-two index math");
  const int c4 = tid & 7, pxm = tid >> 3;
  const unsigned lds_m = (unsigned)(pxm * LDP + c4 * 4) * 4u;
  const int hrow = HALO_W ? (tid >> 4) : 0, hcol = 32 + ((tid >> 3) & 1);
  const unsigned lds_h = (unsigned)((hrow * IW + hcol) * LDP + c4 * 4) * 4u;
  auto fetch_in = [&](const Tile& d, int ch, f32x4 (&regs)[C::IN_PER_THREAD]) {
    const int iy0 = d.y0 - p.pad_t, ix0 = d.x0 - p.pad_l;
    constexpr unsigned IB = IN16 ? 2u : 4u;                     // bytes per input element
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(reinterpret_cast<const float*>(
        reinterpret_cast<const char*>(p.in) + (((ptrdiff_t)d.img * p.H + iy0) * p.W + ix0) * p.in_cs * (ptrdiff_t)IB));
    const bool colm_ok = ix0 + pxm >= 0 && ix0 + pxm < p.W;
    const unsigned voff_m = colm_ok ? (unsigned)(pxm * p.in_cs + c4 * 4) * IB : kLaneOff;
#pragma unroll
    for (int row = 0; row < C::IH; ++row) {
      const int iy = iy0 + row;
      if (iy >= 0 && iy < p.H) {
        if constexpr (IN16) {
          typedef unsigned u32x2v __attribute__((__vector_size__(8)));
          const u32x2v t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_m, (unsigned)((row * p.W) * p.in_cs + ch * 32) * IB, 0);
          regs[row] = f32x4{__uint_as_float(t[0]), __uint_as_float(t[1]), 0.f, 0.f};       // 4 halves in the first two words
        } else {
          regs[row] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_m, (unsigned)((row * p.W) * p.in_cs + ch * 32) * IB, 0));
        }
      } else {
        regs[row] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (HALO_W) {
      const bool halo_ok = tid < HALO_V4 && iy0 + hrow >= 0 && iy0 + hrow < p.H && ix0 + hcol >= 0 && ix0 + hcol < p.W;
      const unsigned voff_h = halo_ok ? (unsigned)((hrow * p.W + hcol) * p.in_cs + c4 * 4) * IB : kLaneOff;
      if constexpr (IN16) {
        typedef unsigned u32x2v __attribute__((__vector_size__(8)));
        const u32x2v t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_h, (unsigned)(ch * 32) * IB, 0);
        regs[C::IH] = f32x4{__uint_as_float(t[0]), __uint_as_float(t[1]), 0.f, 0.f};
      } else {
        regs[C::IH] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_h, (unsigned)(ch * 32) * IB, 0));
      }
    }
  };
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  float in_amax = 0.f;                                                  // range guard of the 16-bit modes (igemm_h16.h)
  auto put = [&](char* dst_f32, char* dst_h16, const f32x4& v) {      // 4 channels of one pixel: fp32, or hi | lo fp16 planes
    if constexpr (H == 0) {
      *reinterpret_cast<f32x4*>(dst_f32) = v;
    } else if constexpr (IN16) {
      *reinterpret_cast<f32x2*>(dst_h16) = f32x2{v[0], v[1]};      // already fp16: straight into the hi plane
    } else {
      f16x2 h0, l0, h1, l1;
      split2(f32x2{v[0], v[1]}, h0, l0);
      split2(f32x2{v[2], v[3]}, h1, l1);
      const f16x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
      in_amax = amax4(v, in_amax);
      *reinterpret_cast<f16x4*>(dst_h16) = hi;
      *reinterpret_cast<f16x4*>(dst_h16 + 64) = lo;
    }
  };
  const unsigned lds_m16 = (unsigned)(pxm * LDP * 4 + c4 * 8), lds_h16 = (unsigned)((hrow * IW + hcol) * LDP * 4 + c4 * 8);
  auto store_in = [&](const f32x4 (&regs)[C::IN_PER_THREAD]) {
    char* base = reinterpret_cast<char*>(s_in);
#pragma unroll
    for (int row = 0; row < C::IH; ++row) put(base + lds_m + row * IW * LDP * 4, base + lds_m16 + row * IW * LDP * 4, regs[row]);
    if (HALO_W && tid < HALO_V4) put(base + lds_h, base + lds_h16, regs[C::IH]);
    if constexpr (H == 2 && !IN16) {
      range_report(in_amax, p.range_flag);
      in_amax = 0.f;
    }
  };
  auto fetch_w = [&](int ch, f32x4 (&regs)[C::W_PER_THREAD]) {
    const float* src = p.w + (size_t)ch * C::W_FLOATS;
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i) {
      int idx = tid + i * 256;
      idx = idx < C::W_V4 ? idx : C::W_V4 - 1;
      regs[i] = *reinterpret_cast<const f32x4*>(src + idx * 4);
    }
  };
  auto store_w = [&](const f32x4 (&regs)[C::W_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i) {
      const int idx = tid + i * 256;
      if (idx < C::W_V4) *reinterpret_cast<f32x4*>(s_w + idx * 4) = regs[i];
    }
  };
  // (TH+2) x (TW+2) halo tile of the gs channel, zero outside the image: shift-only index math, masked buffer loads.
  // Three elements per thread: rows 0..7 x columns 0..31, rows 8..TH+1 x columns 0..31, columns 32..33.
  static_assert(!GS || TH + 2 <= 16, "gs tile: two 8-row passes");
  const int g_row[3] = {tid >> 5, 8 + (tid >> 5), tid >> 1};
  const int g_col[3] = {tid & 31, tid & 31, 32 + (tid & 1)};
  const bool g_act[3] = {true, 8 + (tid >> 5) < TH + 2, (tid >> 1) < TH + 2};
  auto fetch_gs = [&](const Tile& d, float (&regs)[3]) {
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(p.gs + ((ptrdiff_t)d.img * p.H + (d.y0 - 1)) * p.W + (d.x0 - 1));
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int gy = d.y0 - 1 + g_row[e], gx = d.x0 - 1 + g_col[e];
      const bool ok = g_act[e] && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      regs[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? (unsigned)(g_row[e] * p.W + g_col[e]) * 4u : kLaneOff, 0, 0));
    }
  };
  auto store_gs = [&](const float (&regs)[3]) {
#pragma unroll
    for (int e = 0; e < 3; ++e)
      if (g_act[e]) s_gs[g_row[e] * (TW + 2) + g_col[e]] = regs[e];
  };

  // ---- prologue: first tile's chunk 0 ----
  int kseq = 0;
  int tile = seq(0);
  Tile cur = decode(tile);
  f32x4 in_regs[C::IN_PER_THREAD];
  f32x4 w_regs[C::W_PER_THREAD];
  float gs_regs[3] = {0.f, 0.f, 0.f};
  fetch_in(cur, 0, in_regs);
  fetch_w(0, w_regs);
  if (GS) fetch_gs(cur, gs_regs);
  // fused-tail weights: clr_conv2 A operand W2^T[c2 = r][c = 4q + e], clr_conv3 A operand W3^T[c3 = r < 3][c = 4q + e], biases
  float tw2[4], tw3[4];
  f32x4 tb2 = {0.f, 0.f, 0.f, 0.f};
  float tb3[3] = {0.f, 0.f, 0.f};
  if constexpr (TAIL_LDS) {
    for (int i = tid; i < 337; i += 256) s_tail[i] = p.tail_w[i];
  } else if (TAIL) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      tw2[e] = p.tail_w[(4 * q + e) * 16 + r];
      tw3[e] = p.tail_w[272 + (4 * q + e) * 3 + (r < 3 ? r : 0)];      // masked to rows r < 3 at use (no early wait)
    }
    tb2 = *reinterpret_cast<const f32x4*>(p.tail_w + 256 + 4 * q);
#pragma unroll
    for (int c = 0; c < 3; ++c) tb3[c] = p.tail_w[320 + c];
  }
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + 4 * q);
  f32x4 wg = {0.f, 0.f, 0.f, 0.f};
  if (GS) wg = *reinterpret_cast<const f32x4*>(p.w_gs + r * 16 + 4 * q);
  store_in(in_regs);
  store_w(w_regs);
  if (GS) store_gs(gs_regs);
  __syncthreads();
#ifdef BSR_STAMPS
  st1 = __builtin_amdgcn_s_memtime();
#endif

  // this wave: tile rows wave*RW .. wave*RW+RW-1, two 16-pixel MFMA tiles per row
  int x_base[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) x_base[mt] = ((wave * RW + mt / 2) * IW + (mt % 2) * 16 + r) * LDP + 4 * q;
  const int w_base = r * LDP + 4 * q;

  for (; tile < ntiles; tile = seq(++kseq)) {
    const int tile_n = seq(kseq + 1);
    const bool has_next = tile_n < ntiles;
    const Tile nxt = decode(has_next ? tile_n : tile);
    // FUSE: the input pixels of this thread's output pixel (row tid >> 5, column x0 - 3 + (tid & 31)) for gs = gray(inputs) * (1 + mask) + con,
    // requested now, used after the MFMA loops; at the strip's last tile threads 0..2 of a row also finish columns W-3 .. W-1
    float fin[3] = {0.f, 0.f, 0.f}, fin_e[3] = {0.f, 0.f, 0.f};
    if constexpr (FUSE) {
      const int frow = tid >> 5, fj = tid & 31;
      const int fx = cur.x0 - 3 + fj;
      const size_t rowpix = ((size_t)cur.img * p.H + cur.y0 + frow) * p.W;
      if (fx >= 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) fin[c] = p.inputs[(rowpix + fx) * 3 + c];
      }
      if (cur.x0 + TW == p.W && fj < 3) {
#pragma unroll
        for (int c = 0; c < 3; ++c) fin_e[c] = p.inputs[(rowpix + p.W - 3 + fj) * 3 + c];
      }
      if (cur.x0 == 0 && tid < C::TH * 12) s_carry[tid] = 0.f;      // strip start: nothing to the left (parity-0 buffer; read after two barriers)
    }
    float tin[MT][3];
    float tin_st[3] = {0.f, 0.f, 0.f};
    if constexpr (TAIL_LDS) {   // the same pixels through LDS (3 registers instead of 12 across the MFMA loops): row-contiguous loads now, ds_write at the chunk-0 barrier
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int f = tid + e * 256, row = f / (TW * 3), col = f % (TW * 3);
        tin_st[e] = p.inputs[(((size_t)cur.img * p.H + cur.y0 + row) * p.W + cur.x0) * 3 + col];
      }
    } else if (TAIL) {   // the input pixels for the final grayscale difference (lanes q == 0 use them), latency hidden by the MFMA loop
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const size_t pix = ((size_t)cur.img * p.H + cur.y0 + wave * RW + mt / 2) * p.W + cur.x0 + (mt % 2) * 16 + r;
#pragma unroll
        for (int c = 0; c < 3; ++c) tin[mt][c] = p.inputs[pix * 3 + c];
      }
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      // the next (tile, chunk) goes into registers now and into LDS after this chunk's MFMAs
      if (ch == 0) {
        fetch_in(cur, 1, in_regs);
        fetch_w(1, w_regs);
      } else if (has_next) {
        fetch_in(nxt, 0, in_regs);
        fetch_w(0, w_regs);
        if (GS) fetch_gs(nxt, gs_regs);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (H == 0) {
      f32x4 wf[2], xf[2][MT];
      wf[0] = *reinterpret_cast<const f32x4*>(s_w + w_base);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[0][mt] = *reinterpret_cast<const f32x4*>(s_in + x_base[mt]);
#pragma unroll
      for (int i = 0; i < T * 2; ++i) {        // (tap, 16-channel group) steps, fragments read one step ahead
        const int cu = i & 1, nx = cu ^ 1;
        if (i + 1 < T * 2) {
          const int t = (i + 1) / 2, g = (i + 1) % 2;
          wf[nx] = *reinterpret_cast<const f32x4*>(s_w + w_base + t * 16 * LDP + g * 16);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            xf[nx][mt] = *reinterpret_cast<const f32x4*>(s_in + x_base[mt] + ((t / KW) * IW + (t % KW)) * LDP + g * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cu][j], xf[cu][mt][j], acc[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      } else {
      // one K = 32 step per tap: A = weight row r, k = 8q..8q+7 (16 bytes of the hi plane, lo plane 64 bytes further), B = pixel r
      // hi planes are read one tap ahead (two register sets); the lo planes of the split-precision form are read at the start of
      // their own tap into ONE set and consumed by the last instructions of the tap (register budget: see TAIL_LDS above)
      f16x8 wh[2], wl, xh[2][MT], xl[MT];
      wh[0] = *reinterpret_cast<const f16x8*>(s_w + w_base);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xh[0][mt] = *reinterpret_cast<const f16x8*>(s_in + x_base[mt]);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int cu = t & 1, nx = cu ^ 1;
        const int toff = ((t / KW) * IW + (t % KW)) * LDP;
        wl = *reinterpret_cast<const f16x8*>(s_w + w_base + t * 16 * LDP + 16);
        if constexpr (!IN16) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) xl[mt] = *reinterpret_cast<const f16x8*>(s_in + x_base[mt] + toff + 16);
        }
        if (t + 1 < T) {
          const int tn = t + 1;
          wh[nx] = *reinterpret_cast<const f16x8*>(s_w + w_base + tn * 16 * LDP);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            xh[nx][mt] = *reinterpret_cast<const f16x8*>(s_in + x_base[mt] + ((tn / KW) * IW + (tn % KW)) * LDP);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cu], xh[cu][mt], acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[cu][mt], acc[mt], 0, 0, 0);
        if constexpr (!IN16) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cu], xl[mt], acc[mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      if (ch == 1 && GS) {   // the gs channel: K group k = 4q + j <-> tap (k/3, k%3), k < 9
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 4 * q + j;
          const int kk = k < 9 ? k : 0;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float xv = s_gs[(wave * RW + mt / 2 + kk / 3) * (TW + 2) + (mt % 2) * 16 + r + kk % 3];
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[j], k < 9 ? xv : 0.f, acc[mt], 0, 0, 0);
          }
        }
      }
      if constexpr (FUSE) {
        if (ch == 1) {
          // ---- fused heads epilogue.  q[row][col][kx*2 + head] (this tile's 7x1 partial planes) -> LDS over the input tile ----
          __syncthreads();                                   // every wave has finished reading the input tile
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = acc[mt] + b4;
            *reinterpret_cast<f32x4*>(s_q + ((wave * RW + mt / 2) * TW + (mt % 2) * 16 + r) * C::QLD + 4 * q) = v;
          }
          __syncthreads();
          {
            // thread = output pixel (row frow, image column x0 - 3 + fj): out[x] = sum_kx q[x + kx - 3][kx], kx ascending as in
            // heads_post_kernel; columns left of this tile were summed by the previous tile (carry), columns right of it go to the next
            const int frow = tid >> 5, fj = tid & 31;
            const int par = (cur.x0 / TW) & 1;
            const float* cin = s_carry + par * (TH * 12) + frow * 12;
            float* cout = s_carry + (par ^ 1) * (TH * 12) + frow * 12;
            float m = 0.f, cn = 0.f;
            if (fj < 6) { m = cin[fj * 2]; cn = cin[fj * 2 + 1]; }
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
              const int c = fj + kx - 6;
              if (c >= 0) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(s_q + (frow * TW + c) * C::QLD + 2 * kx);
                m += v[0];
                cn += v[1];
              }
            }
            const size_t rowpix = ((size_t)cur.img * p.H + cur.y0 + frow) * p.W;
            const int fx = cur.x0 - 3 + fj;
            if (fx >= 0) heads_emit(p, rowpix + fx, m, cn, fin);
            if (fj < 6) {                                    // pixels x0 + 29 + fj: the taps that lie in this tile
              float cm = 0.f, ccn = 0.f;
#pragma unroll
              for (int kx = 0; kx < 6; ++kx) {
                if (kx <= 5 - fj) {
                  const f32x2 v = *reinterpret_cast<const f32x2*>(s_q + (frow * TW + 26 + fj + kx) * C::QLD + 2 * kx);
                  cm += v[0];
                  ccn += v[1];
                }
              }
              cout[fj * 2] = cm;
              cout[fj * 2 + 1] = ccn;
              if (cur.x0 + TW == p.W && fj < 3) heads_emit(p, rowpix + p.W - 3 + fj, cm, ccn, fin_e);      // right image edge: nothing follows
            }
          }
          if (has_next) {
            __syncthreads();                                 // the parked planes have been read: the next tile may overwrite them
            store_in(in_regs);
            store_w(w_regs);
            __syncthreads();
          }
        } else {
          __syncthreads();
          store_in(in_regs);
          store_w(w_regs);
          __syncthreads();
        }
      } else
      if (ch == 0 || has_next) {
        __syncthreads();
        store_in(in_regs);
        store_w(w_regs);
        if (ch == 1 && GS) store_gs(gs_regs);
        if constexpr (TAIL_LDS) {
          if (ch == 0) {
#pragma unroll
            for (int e = 0; e < 3; ++e) s_tail[352 + tid + e * 256] = tin_st[e];
          }
        }
        __syncthreads();
      }
    }
    if constexpr (FUSE) { cur = nxt; continue; }

#ifdef BSR_STAMPS
    const unsigned long long se0 = __builtin_amdgcn_s_memtime();
#endif
    // epilogue: lane (pixel r of tile mt, q) holds channels 4q .. 4q+3
    if constexpr (TAIL_LDS) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tw2[e] = s_tail[(4 * q + e) * 16 + r];
        tw3[e] = s_tail[272 + (4 * q + e) * 3 + (r < 3 ? r : 0)];
      }
      tb2 = *reinterpret_cast<const f32x4*>(s_tail + 256 + 4 * q);
#pragma unroll
      for (int c = 0; c < 3; ++c) tb3[c] = s_tail[320 + c];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < 3; ++c) tin[mt][c] = s_tail[352 + ((wave * RW + mt / 2) * TW + (mt % 2) * 16 + r) * 3 + c];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const size_t row_pix = ((size_t)cur.img * p.H + cur.y0 + wave * RW + mt / 2) * p.W + cur.x0;
      f32x4 v = acc[mt] + b4;
      if (p.act) {
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const f32x2 y = leaky_relu2(f32x2{v[e], v[e + 1]});
          v[e] = y[0];
          v[e + 1] = y[1];
        }
      }
      const size_t pix = row_pix + (mt % 2) * 16 + r;
      if (!TAIL) {
        *reinterpret_cast<f32x4*>(p.out + pix * p.out_cs + 4 * q) = v;
      } else {
        // clr_conv2: y2^T[c2][px] = sum_c W2^T[c2][c] * y1^T[c][px]; register e of v is channel c = 4q + e (k index q)
        f32x4 a2 = tb2;
#pragma unroll
        for (int e = 0; e < 4; ++e) a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(tw2[e], v[e], a2, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const f32x2 y = leaky_relu2(f32x2{a2[e], a2[e + 1]});
          a2[e] = y[0];
          a2[e + 1] = y[1];
        }
        // clr_conv3: rows c3 = 0..2 (lanes r < 3 carry the weights, the rest multiply by 0)
        f32x4 a3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(r < 3 ? tw3[e] : 0.f, a2[e], a3, 0, 0, 0);
        if (q == 0) {   // rows 0..2 = R,G,B of pixel r
#pragma clang fp contract(off)
          const float cr = a3[0] + tb3[0], cg = a3[1] + tb3[1], cb = a3[2] + tb3[2];
          const float g1 = (cr * 0.2989f + cg * 0.5870f) + cb * 0.1140f;
          const float g0 = (tin[mt][0] * 0.2989f + tin[mt][1] * 0.5870f) + tin[mt][2] * 0.1140f;
          if (p.packed != nullptr) {
            *reinterpret_cast<f32x4*>(p.packed + pix * 4) = f32x4{cr, cg, cb, g1 - g0};
          } else {
            p.con_rgb[pix * 3 + 0] = cr;
            p.con_rgb[pix * 3 + 1] = cg;
            p.con_rgb[pix * 3 + 2] = cb;
            p.dif[pix] = g1 - g0;
          }
        }
      }
    }
#ifdef BSR_STAMPS
    st_epi += __builtin_amdgcn_s_memtime() - se0;
#endif
    cur = nxt;
  }
#ifdef BSR_STAMPS
  if (p.stamps != nullptr && lane == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long st3 = __builtin_amdgcn_s_memtime(), rt3 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* d = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
    d[0] = st1 - st0; d[1] = st3 - st1 - st_epi; d[2] = rt3 - rt0; d[3] = st_epi;
  }
#endif
}

// FUSE grids are one workgroup per row strip (at most `resident`): conv_n16_fuse_pays() tells the caller when that fills the chip.
inline bool conv_n16_fuse_pays(int batch, int H, int TH, int resident) { return (long long)batch * (H / TH) >= resident; }

template <int KH, int KW, bool GS, bool TAIL, int RW, int H = 0, bool IN16 = false, bool FUSE = false>
inline hipError_t launch_conv_n16(ConvN16Args a, int batch, hipStream_t stream, int* resident_out = nullptr) {
  using C = ConvN16Cfg<KH, KW, GS, TAIL, RW, FUSE>;
  auto kern = conv_n16_kernel<KH, KW, GS, TAIL, RW, H, IN16, FUSE>;
  static PerDeviceOnce once;               // .value = workgroups the device holds at once (2 per CU: LDS-bound)
  const int dev = PerDeviceOnce::current();
  int resident = dev >= 0 && once.done[dev] ? once.value[dev] : 0;
  if (resident == 0) {
    if (C::SMEM_BYTES > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
      if (e != hipSuccess) return e;
    }
    int cur = 0, cus = 0;
    hipError_t e = hipGetDevice(&cur);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cur);
    if (e != hipSuccess) return e;
    resident = 2 * cus;
    if (dev >= 0) { once.value[dev] = resident; once.done[dev] = true; }
  }
  if (resident_out != nullptr) { *resident_out = resident; return hipSuccess; }      // query only
  a.tiles_x = a.W / C::TW;
  a.tiles_y = a.H / C::TH;
  a.batch = batch;
  // FUSE: the boundary carry is double-buffered by TILE PARITY and a strip start clears the parity-0 half, which is only right when
  // the previous strip ended on parity 1 — an even number of tiles per strip (bsr_forward's W % 256 == 0 guarantees it; refuse
  // anything else here rather than corrupt gs / mask22 at strip boundaries)
  if (FUSE && (a.tiles_x & 1)) return hipErrorInvalidValue;
  const int ntiles = a.tiles_x * a.tiles_y * batch;
  const int nunits = FUSE ? a.tiles_y * batch : ntiles;        // FUSE: a workgroup's unit of work is a row strip
  hipLaunchKernelGGL(kern, dim3(nunits < resident ? nunits : resident), dim3(256), C::SMEM_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace bsr
