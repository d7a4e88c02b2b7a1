// PERSISTENT variant of the stride-2 3x3 convolution (down1 / down2: igemm_conv_kernel<3,3,2,false,4,32,4,1,1,2,16,1>) — round 4,
// the review's item (ii).  The plain kernel pays a prologue of ~11 k cycles per 4x32-pixel tile (20 input-tile loads per thread plus
// their address set-up, two weight steps, LDS writes, barrier) against 38 k cycles of matrix loop: 22 % of a workgroup's life.  Here
// a workgroup walks tiles b, b + grid, b + 2 grid, ... and
//   * requests the NEXT tile's input (both 16-channel chunks of the first pair: 20 loads per thread into the staging registers the
//     current tile no longer needs) during the last chunk's taps of the current tile,
//   * keeps the weight ring running across the tile boundary (the step two ahead wraps to step 0 / 1 of the next tile),
// so that a tile change costs the epilogue, one LDS write of the staged tile and one barrier — no load latency, no pipeline drain.
// Same implicit-GEMM scheme, fragment layout and accumulation order as igemm_conv_kernel (bit-identical outputs: tested).
// Whether it pays is a measurement (profiles/HISTORY.md, round 4): the instructions of a prologue cost the same issue slots wherever
// they stand; only the latency and the workgroup turn-over can be hidden.
#pragma once
#include <hip/hip_runtime.h>
#include "mfma_common.h"

namespace bsr {

template <int NI>
struct S2PCfg {
  static constexpr int TH = 4, TW = 32, IH = 9, IW = 65, CC = 16, LDP = CC + 4, G = CC / 8, T = 9, BN = NI * 32, NT = 256;
  static constexpr int IN_FLOATS = IH * IW * LDP, W_FLOATS = BN * LDP;
  static constexpr int SMEM_BYTES = (IN_FLOATS + 3 * W_FLOATS) * 4;
  static constexpr int Q = CC / 4, CP = NT / Q, RC = IW - CP;      // 4 float4 per pixel, 64 columns per pass, 1 remainder column
  static constexpr int IN_PER_THREAD = IH + 1;
  static constexpr int W_V4 = W_FLOATS / 4, W_PER_THREAD = (W_V4 + NT - 1) / NT;
  static_assert(RC == 1 && IH * RC * Q <= NT, "row-wise staging plan");
};

// ConvArgs as for igemm_conv_kernel<3,3,2,...>: tiles_x / tiles_y = output tiles per image, n_blocks = total tiles (batch x tiles); nchunk even
template <int NI>
__global__ __launch_bounds__(256, 2) void igemm_s2p_kernel(ConvArgs p) {
  using C = S2PCfg<NI>;
  constexpr int IW = C::IW, LDP = C::LDP, G = C::G, T = C::T, IH = C::IH;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + C::IN_FLOATS;
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int total = p.n_blocks;
  const int nsteps = p.nchunk * T;

  const int a_base = ((wave * 2) * IW + r * 2) * LDP + 4 * h;      // output pixel (row wave, column r) of the tile -> input pixel (2 row, 2 col)
  int b_base[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b_base[ni] = (ni * 32 + r) * LDP + 4 * h;
  float bias_n[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bias_n[ni] = p.bias[ni * 32 + r < p.n_pad ? ni * 32 + r : 0];

  // ---- per-tile staging plan (igemm_conv_kernel's ROWWISE plan): load rr of a thread is tile row rr at the thread's own column / float4
  const int scol = tid / C::Q, sq = tid % C::Q;
  const int rem_e = tid < IH * C::Q ? tid : 0;                       // the one remainder column: row rem_e / Q, float4 rem_e % Q
  const bool rem_act = tid < IH * C::Q;
  const int in_loff0 = scol * LDP + sq * 4;
  const int in_loff1 = rem_act ? ((rem_e / C::Q) * IW + C::CP) * LDP + (rem_e % C::Q) * 4 : -1;
  unsigned in_goff[C::IN_PER_THREAD];
  __amdgpu_buffer_rsrc_t in_rsrc;
  int t_img = 0, t_y0 = 0, t_x0 = 0;                                 // the tile the staging plan / registers currently describe
  auto tile_setup = [&](int bid) {
    const int tile_x = bid % p.tiles_x;
    const int b2 = bid / p.tiles_x;
    const int tile_y = b2 % p.tiles_y;
    t_img = b2 / p.tiles_y;
    t_y0 = tile_y * C::TH;
    t_x0 = tile_x * C::TW;
    const int iy0 = t_y0 * 2 - p.pad_t, ix0 = t_x0 * 2 - p.pad_l;
    const int ix = ix0 + scol;
    const bool colok = ix >= 0 && ix < p.W;
    const unsigned colpart = (unsigned)((ix * p.in_cs + sq * 4) * 4);
    const unsigned rowstride = (unsigned)(p.W * p.in_cs * 4);
#pragma unroll
    for (int rr = 0; rr < IH; ++rr) {
      const int iy = iy0 + rr;
      in_goff[rr] = (iy >= 0 && iy < p.H && colok) ? colpart + (unsigned)iy * rowstride : kLaneOff;
    }
    {
      const int iy = iy0 + rem_e / C::Q, ixr = ix0 + C::CP;
      const bool ok = rem_act && iy >= 0 && iy < p.H && ixr >= 0 && ixr < p.W;
      in_goff[IH] = ok ? (unsigned)(((iy * p.W + ixr) * p.in_cs + (rem_e % C::Q) * 4) * 4) : kLaneOff;
    }
    in_rsrc = make_rsrc(p.in + (size_t)t_img * p.H * p.W * p.in_cs + p.in_coff);
  };
  auto fetch_in = [&](int ch, f32x4 (&regs)[C::IN_PER_THREAD]) {
    const unsigned soff = (unsigned)(ch * C::CC * 4);
#pragma unroll
    for (int i = 0; i < C::IN_PER_THREAD; ++i)
      regs[i] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(in_rsrc, in_goff[i], soff, 0));
  };
  auto store_in = [&](const f32x4 (&regs)[C::IN_PER_THREAD]) {
#pragma unroll
    for (int rr = 0; rr < IH; ++rr) *reinterpret_cast<f32x4*>(s_in + in_loff0 + rr * IW * LDP) = regs[rr];
    if (in_loff1 >= 0) *reinterpret_cast<f32x4*>(s_in + in_loff1) = regs[IH];
  };
  unsigned w_off[C::W_PER_THREAD];
#pragma unroll
  for (int i = 0; i < C::W_PER_THREAD; ++i) w_off[i] = (unsigned)(((tid + i * C::NT) % C::W_V4) * 16);
  const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.w);
  const unsigned w_step_bytes = (unsigned)(p.n_pad * LDP * 4);
  auto fetch_w = [&](int step, f32x4 (&regs)[C::W_PER_THREAD]) {
    const unsigned soff = (unsigned)step * w_step_bytes;
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i)
      regs[i] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off[i], soff, 0));
  };
  auto store_w = [&](int off, const f32x4 (&regs)[C::W_PER_THREAD]) {
    char* dst = reinterpret_cast<char*>(s_w + off);
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i) *reinterpret_cast<f32x4*>(dst + w_off[i]) = regs[i];
  };

  f32x4 in_regs[C::IN_PER_THREAD], in_regs2[C::IN_PER_THREAD], w_regs[C::W_PER_THREAD];
  int w_cur = 0, w_n1 = C::W_FLOATS, w_n2 = 2 * C::W_FLOATS;
  int bid = blockIdx.x;
  // ---- prologue of the FIRST tile only: everything requested before anything is waited for
  tile_setup(bid);
  {
    f32x4 w_regs1[C::W_PER_THREAD];
    fetch_in(0, in_regs);
    fetch_in(1, in_regs2);
    fetch_w(0, w_regs);
    fetch_w(1, w_regs1);
    store_in(in_regs);
    store_w(0, w_regs);
    store_w(w_n1, w_regs1);
  }
  __syncthreads();

  f32x4 af[2], bf[2][NI];
  auto read_frags = [&](int slot, int a_off, int b_off) {
    af[slot] = *reinterpret_cast<const f32x4*>(s_in + a_base + a_off);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bf[slot][ni] = *reinterpret_cast<const f32x4*>(s_w + b_base[ni] + b_off);
  };
  auto tap_offset = [&](int t) -> int { return ((t / 3) * IW + (t % 3)) * LDP; };
  const float act_alpha = p.act ? kLeakyAlpha : 1.f;
  const unsigned cs4 = (unsigned)p.out_cs * 4u;
  const unsigned lane_out = (unsigned)(4 * h) * cs4 + (unsigned)r * 4u;

  for (;;) {
    const int cur_img = t_img, cur_y0 = t_y0, cur_x0 = t_x0;          // the epilogue's tile (the plan moves on to the next one before it)
    const int next_bid = bid + (int)gridDim.x;
    const bool has_next = next_bid < total;
    f32x16 acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[ni] = bias_tile(h, bias_n[ni]);
    read_frags(0, tap_offset(0), w_cur);
    __builtin_amdgcn_s_setprio(0);

    for (int ch = 0; ch < p.nchunk; ++ch) {
      const bool more = ch + 1 < p.nchunk;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int s = ch * T + t;
        // the weight step two ahead: of this tile, or — across the tile boundary — step 0 / 1 of the next one
        const int s2 = s + 2;
        const bool has2 = s2 < nsteps || has_next;
        if (has2) fetch_w(s2 < nsteps ? s2 : s2 - nsteps, w_regs);
        if (t == T - 4) {
          if (more) {
            if (ch & 1) {                           // the next pair of chunks of THIS tile (igemm_conv_kernel's PAIR fetch)
              fetch_in(ch + 1, in_regs);
              if (ch + 2 < p.nchunk) fetch_in(ch + 2, in_regs2);
            }
          } else if (has_next) {                    // last chunk: both staging register sets are free — the NEXT tile's first pair
            tile_setup(next_bid);
            fetch_in(0, in_regs);
            fetch_in(1, in_regs2);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int tap_off = tap_offset(t);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int cur = (t * G + g) & 1, nxt = cur ^ 1;
          if (g + 1 < G) {
            read_frags(nxt, tap_off + (g + 1) * 8, w_cur + (g + 1) * 8);
          } else if (t + 1 < T) {
            read_frags(nxt, tap_offset(t + 1 < T ? t + 1 : 0), w_n1);
          }
          if (g == G - 1 && has2) store_w(w_n2, w_regs);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][j], bf[cur][ni][j], acc[ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (t == T - 1 && more) {                   // single input buffer: swap it between chunks (2 barriers)
          if (ch & 1) store_in(in_regs); else store_in(in_regs2);
          __syncthreads();
          read_frags((T * G) & 1, tap_offset(0), w_n1);
        }
        { const int tw = w_cur; w_cur = w_n1; w_n1 = w_n2; w_n2 = tw; }
      }
    }

    // ---- epilogue of the current tile: LeakyReLU + NHWC stores (igemm_conv_kernel's addressing)
    __builtin_amdgcn_s_setprio(3);
    {
      const size_t blk_pix = (size_t)cur_img * p.Ho * p.Wo + (size_t)cur_y0 * p.Wo + cur_x0;
      const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(p.out + blk_pix * p.out_cs + p.out_coff);
      const unsigned quad_step = 8u * cs4;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const bool n_ok = ni * 32 + r < p.n_store;
        unsigned voff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) voff[j] = n_ok ? lane_out + (unsigned)j * cs4 : kLaneOff;
        unsigned soff = ((unsigned)(wave * p.Wo) * (unsigned)p.out_cs + (unsigned)(ni * 32)) * 4u;
        f32x16 v = acc[ni];
        leaky_relu_tile(v, act_alpha);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[4 * q + j]), orsrc, voff[j], soff, 0);
          soff += quad_step;
        }
      }
    }
    if (!has_next) break;
    // ---- tile change: the next tile's first chunk is in registers, its weight steps 0 / 1 are in the ring
    store_in(in_regs);
    __syncthreads();
    bid = next_bid;
  }
}

template <int NI>
inline hipError_t launch_igemm_s2p(ConvArgs a, int batch, hipStream_t stream) {
  using C = S2PCfg<NI>;
  auto kern = igemm_s2p_kernel<NI>;
  if (a.nchunk < 2 || (a.nchunk & 1) || a.n_store > C::BN || a.Ho % C::TH != 0 || a.Wo % C::TW != 0) return hipErrorInvalidValue;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  a.tiles_x = a.Wo / C::TW;
  a.tiles_y = a.Ho / C::TH;
  a.n_blocks = a.tiles_x * a.tiles_y * batch;
  const int resident = 2 * device_cu_count();
  hipLaunchKernelGGL(kern, dim3(a.n_blocks < resident ? a.n_blocks : resident), dim3(C::NT), C::SMEM_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace bsr
