// 1x1-convolution GEMM with a RESIDENT activation tile: out[px][n] = act(sum_k x[px][k] * W[k][n] + b[n] (+ residuals)).
//
// The bottleneck blocks run three K = 128 GEMMs per block with wide N (conv3 | theta|phi|g : N = 672, w : N = 288;
// /root/reference/model.py:86,10-13,101-102,56-59).  With one workgroup per (128-pixel, 96-channel) tile the K loop is only four
// steps long and prologue + epilogue cost as much as the MFMAs.  Here a workgroup owns 128 pixels for a long run of
// N tiles (blockIdx.y splits the tiles in NSPLIT ranges so two workgroups share a CU): each wave keeps the A fragments
// of its 32 pixels x K in REGISTERS for the whole kernel (K = 128 -> 64 VGPRs, no activation LDS at all), the weights
// stream through the same 3-slot LDS ring as igemm_conv_kernel, continuously across the N groups (no pipeline drain
// between groups), and each group of NI tiles ends with its own fused epilogue.  Same fragment scheme (v_mfma_f32_32x32x2_f32, ds_read_b128, K permuted
// identically in A and B).
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_conv.h"
#include "igemm_h16.h"

namespace bsr {

// H = 0: fp32 matrix cores (v_mfma_f32_32x32x2_f32).  H = 2 / 1: 16-bit matrix cores as in igemm_h16.h — the A fragments are split
// into hi / lo fp16 planes once, when they are loaded into registers; the weight image is the fp16 one of pack_taps_h16 (for
// H = 2 it has the same 36-word rows as the fp32 image), and a 16-channel K group costs 3 (f32x3) or 1 (f16) v_mfma_f32_32x32x16_f16.
// CCF: channels per K chunk of the fp32 form (32; 24 for res*.conv1, whose K = 120 | 264 is packed in 24-channel chunks)
template <int NI, int NCH, int H = 0, int CCF = 32>   // NI 32-wide channel tiles per group, NCH = K / CC
struct GemmNLoopCfg {
  static_assert(H == 0 || CCF == 32, "the 16-bit forms use 32-channel chunks");
  static constexpr int CC = CCF, LDP = (H == 1 ? 20 : CCF + 4), G = (H ? 2 : CCF / 8), LO = 16, BM = 128, BN = NI * 32;
  static constexpr int W_FLOATS = BN * LDP;
  static constexpr int MAX_TILES = 24;                        // tiles per blockIdx.y range (bias staged in LDS)
  static constexpr int SMEM_BYTES = (3 * W_FLOATS + MAX_TILES * 32) * 4;
  static constexpr int W_V4 = W_FLOATS / 4;
  static constexpr int W_PER_THREAD = (W_V4 + 255) / 256;
};

// ConvArgs use: in/in_cs/in_coff (K = NCH*32 channels), pixels flattened (multiple of 128),
// w packed [NCH][1][n_pad][36] with n_pad >= 32 * (tiles + NI - 1), bias[n_pad], out/out_cs/out_coff/n_store
// (+ out2/n_split/n_store1), act, res1 (one residual, channels [0, res1_c)).  tiles_x = tiles per blockIdx.y range.
// MINW = waves per SIMD the register budget is sized for.  2 (default): two workgroups share a CU.  1 (round 5, res*.conv1 in the 16-bit
// modes: K = 288, N = 128): ONE workgroup per CU keeps 32 pixels x 288 channels per wave resident — 36 16-byte loads per lane, all in
// flight at once (147 KB per CU: the input is read exactly once, at full memory-level parallelism) — and computes all of N from them.
// The implicit-GEMM form of that layer took its input through a 3-slot LDS ring two 0.16-us steps ahead of an L2 / HBM round trip, in
// two N blocks that each fetched and split the tile: 27 us against ~11 of HBM time.
template <int NI, int NCH, int H = 0, int MINW = 2, int CCF = 32>
__global__ __launch_bounds__(256, MINW) void gemm_nloop_kernel(ConvArgs p) {
  using C = GemmNLoopCfg<NI, NCH, H, CCF>;
  constexpr int LDP = C::LDP, G = C::G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w = smem;
  float* s_bias = smem + 3 * C::W_FLOATS;

#ifdef BSR_STAMPS
  unsigned long long st0 = __builtin_amdgcn_s_memtime(), st1 = 0, st_epi = 0, rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, r = lane & 31;
  const size_t pix0 = (size_t)blockIdx.x * C::BM;            // first pixel of this block (flattened B*H*W index)
  const int tiles_total = (p.n_store + 31) / 32;
  const int t0 = blockIdx.y * p.tiles_x;                      // this block's tile range [t0, t1)
  const int t1 = min(t0 + p.tiles_x, tiles_total);
  // Group schedule.  The grid is exactly one wave of workgroups (2 per CU, one of each blockIdx.y range), which all
  // start together: with identical schedules the two workgroups of a CU stay in lockstep and reach their epilogues (VALU +
  // stores + residual latency, no MFMA) at the same time, leaving the matrix pipe idle.  Odd ranges therefore run their
  // SHORT group first ((ntiles-1) % NI + 1 tiles), which shifts their epilogues into the partner's MFMA phases.
  const int ntiles = t1 - t0;
  const int first = (blockIdx.y & 1) ? (ntiles - 1) % NI + 1 : NI;      // tiles in group 0
  const int ngroups = ntiles <= first ? 1 : 1 + (ntiles - first + NI - 1) / NI;
  const int nsteps = ngroups * NCH;
  auto group_tile0 = [&](int ng) { return t0 + (ng == 0 ? 0 : first + (ng - 1) * NI); };

  int b_base[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b_base[ni] = (ni * 32 + r) * LDP + 4 * h;
  unsigned w_off[C::W_PER_THREAD];
#pragma unroll
  for (int i = 0; i < C::W_PER_THREAD; ++i) {
    const int idx0 = tid + i * 256;
    w_off[i] = (unsigned)((idx0 % C::W_V4) * 16);                  // surplus threads of the last pass copy an element twice (same data): no guard at the LDS write
  }
  // raw buffer over the packed weights, the step's image as the wave-uniform SGPR offset (igemm_conv.h: no 64-bit address arithmetic,
  // no exec-mask branches inside the matrix loop)
  const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.w);
  auto fetch_w = [&](int s, f32x4 (&regs)[C::W_PER_THREAD]) {      // step s = (group, chunk)
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const int ng = s / NCH, ch = s % NCH;
    const unsigned soff = (unsigned)((ch * p.n_pad + group_tile0(ng) * 32) * LDP * 4);
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i)
      regs[i] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off[i], soff, 0));
  };
  auto store_w = [&](int off, const f32x4 (&regs)[C::W_PER_THREAD]) {
    char* dst = reinterpret_cast<char*>(s_w + off);
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i) *reinterpret_cast<f32x4*>(dst + w_off[i]) = regs[i];
  };

  // ---- prologue: everything is issued before anything is waited for: weight steps 0 and 1, the bias slice, and this
  // lane's A fragments (pixel r of the wave's 32, channels 8g + 4h .. +3) ----
  f32x4 w_regs[C::W_PER_THREAD], w_regs1[C::W_PER_THREAD];
  fetch_w(0, w_regs);
  if (nsteps > 1) fetch_w(1, w_regs1);
  // bias slice -> LDS (keeps bias loads out of the MFMA loop's vmcnt queue).  Issued BEFORE the A fragments: loads return in
  // order, and behind them this copy would hold the first barrier until the whole activation tile has arrived.
  for (int i = tid; i < (t1 - t0) * 32; i += 256) s_bias[i] = p.bias[t0 * 32 + i];
  f32x4 afr[H ? 1 : NCH * G];
  f16x8 ahi[H ? NCH * G : 1], alo[H == 2 ? NCH * G : 1];
  if constexpr (H == 0) {
    const float* row = p.in + (pix0 + wave * 32 + r) * p.in_cs + p.in_coff + 4 * h;
#pragma unroll
    for (int g = 0; g < NCH * G; ++g) afr[g] = *reinterpret_cast<const f32x4*>(row + g * 8);
  } else {          // channels 16g + 8h .. +7 of this lane's pixel: the A operand map of v_mfma_f32_32x32x16_f16
    const float* row = p.in + (pix0 + wave * 32 + r) * p.in_cs + p.in_coff + 8 * h;
    f32x4 raw[2 * NCH * G];
#pragma unroll
    for (int g = 0; g < NCH * G; ++g) {
      raw[2 * g] = *reinterpret_cast<const f32x4*>(row + g * 16);
      raw[2 * g + 1] = *reinterpret_cast<const f32x4*>(row + g * 16 + 4);
    }
    float amax = 0.f;
#pragma unroll
    for (int g = 0; g < NCH * G; ++g) {
      f16x8 hi, lo;
      split8(raw[2 * g], raw[2 * g + 1], hi, lo);
      amax = amax8(raw[2 * g], raw[2 * g + 1], amax);
      ahi[g] = hi;
      if (H == 2) alo[g] = lo;
    }
    range_report(amax, p.range_flag);            // range guard of the 16-bit modes (igemm_h16.h)
  }
  store_w(0, w_regs);
  if (nsteps > 1) store_w(C::W_FLOATS, w_regs1);
  __syncthreads();

  int w_cur = 0, w_n1 = C::W_FLOATS, w_n2 = 2 * C::W_FLOATS;
  f32x4 bf[2][H ? 1 : NI];
  f16x8 bh[2][H ? NI : 1], bl[2][H == 2 ? NI : 1];
  auto read_frags = [&](int slot, int b_off) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if constexpr (H == 0) {
        bf[slot][ni] = *reinterpret_cast<const f32x4*>(s_w + b_base[ni] + b_off);
      } else {
        bh[slot][ni] = *reinterpret_cast<const f16x8*>(s_w + b_base[ni] + b_off);
        if constexpr (H == 2) bl[slot][ni] = *reinterpret_cast<const f16x8*>(s_w + b_base[ni] + b_off + C::LO);
      }
    }
  };
  read_frags(0, w_cur);
  __builtin_amdgcn_s_setprio(0);
#ifdef BSR_STAMPS
  st1 = __builtin_amdgcn_s_memtime();
#endif

  // Epilogue addressing through raw buffer resources (igemm_conv.h): per element one 32-bit SGPR offset from this wave's
  // pixel-0 row and one constant per-lane VGPR offset.
  const bool has_res = p.res1 != nullptr;
  const float act_alpha = p.act ? kLeakyAlpha : 1.f;
  const size_t tile_pix = pix0 + (size_t)__builtin_amdgcn_readfirstlane(wave) * 32;      // this wave's 32 consecutive pixels
  const unsigned lane_out = ((unsigned)(4 * h) * (unsigned)p.out_cs + (unsigned)r) * 4u;
  const unsigned lane_out2 = ((unsigned)(4 * h) * (unsigned)p.out2_cs + (unsigned)r) * 4u;
  const unsigned lane_res = (unsigned)(4 * h) * (unsigned)p.res1_cs * 4u;
  const __amdgpu_buffer_rsrc_t rsrc_out = make_rsrc(p.out + tile_pix * p.out_cs + p.out_coff);
  const __amdgpu_buffer_rsrc_t rsrc_out2 = make_rsrc(p.out2 != nullptr ? p.out2 + tile_pix * p.out2_cs : p.out);
  const __amdgpu_buffer_rsrc_t rsrc_res = make_rsrc(has_res ? p.res1 + tile_pix * p.res1_cs : p.in);

  for (int ng = 0; ng < ngroups; ++ng) {
    const int tg = group_tile0(ng);                           // first tile of this group
    const int nvalid = ng == 0 ? min(first, ntiles) : min(NI, t1 - tg);     // tiles of this group that exist (uniform)
    f32x16 acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      acc[ni] = bias_tile(h, s_bias[min(tg - t0 + ni, ntiles - 1) * 32 + r]);      // one matrix instruction per tile (igemm_conv.h), not 16 moves at priority 0
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int s = ng * NCH + ch;
      const bool has1 = s + 1 < nsteps, has2 = s + 2 < nsteps;
#if defined(BSR_NL_DIAG) && (BSR_NL_DIAG == 6 || BSR_NL_DIAG == 7)
      if (has2 && p.act == 77) fetch_w(s + 2, w_regs);       // diagnostic: the loop without its weight traffic (stale LDS contents)
#else
      if (has2) fetch_w(s + 2, w_regs);
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int cur = (ch * G + g) & 1, nxt = cur ^ 1;      // fragment slots alternate across steps too (G may be odd: 24-channel chunks); every channel group has NCH * G pieces
        if (g + 1 < G) {
          read_frags(nxt, w_cur + (g + 1) * 8);
        } else if (has1) {
          read_frags(nxt, w_n1);
        }
#if defined(BSR_NL_DIAG) && BSR_NL_DIAG == 7
        if (g == G - 1 && has2 && p.act == 77) store_w(w_n2, w_regs);
#else
        if (g == G - 1 && has2) store_w(w_n2, w_regs);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (H == 0) {
          const f32x4 a = afr[ch * G + g];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if (ni < nvalid) {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bf[cur][ni][j], acc[ni], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if (ni < nvalid) {
              if constexpr (H == 2) {
                acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ch * G + g], bh[cur][ni], acc[ni], 0, 0, 0);
                acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ch * G + g], bl[cur][ni], acc[ni], 0, 0, 0);
              }
              acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ch * G + g], bh[cur][ni], acc[ni], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      const int tw = w_cur; w_cur = w_n1; w_n1 = w_n2; w_n2 = tw;
    }
    if constexpr (((NCH * G) & 1) != 0) {      // an odd number of pieces per group: the next group's first fragments were prefetched into slot 1
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if constexpr (H == 0) bf[0][ni] = bf[1][ni];
        else { bh[0][ni] = bh[1][ni]; if constexpr (H == 2) bl[0][ni] = bl[1][ni]; }
      }
    }

    // ---- epilogue of this channel group (same form as igemm_conv_kernel's) ----
    __builtin_amdgcn_s_setprio(3);
#ifdef BSR_STAMPS
    const unsigned long long se0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if (ni >= nvalid) continue;
      const int nt = (tg + ni) * 32;
      const int n = nt + r;
      const bool second = p.out2 != nullptr && nt >= p.n_split;
      const bool n_ok = second ? n < p.n_store : (p.out2 != nullptr ? n < p.n_store1 : n < p.n_store);
      f32x16 v = acc[ni];
      // Addressing as in igemm_conv_kernel's epilogue: the four pixel rows of a register quad are four per-lane offsets, the quad's
      // base one SGPR that moves by eight pixels — one scalar add per four loads / stores instead of three per element.
#if defined(BSR_NL_DIAG) && (BSR_NL_DIAG == 1 || BSR_NL_DIAG == 4 || BSR_NL_DIAG == 6 || BSR_NL_DIAG == 7)
      if (has_res && nt < p.res1_c && p.act == 77) {
#else
      if (has_res && nt < p.res1_c) {                        // ONE residual (res1, channels [0, res1_c)); uniform per tile
#endif
        const unsigned rcs4 = (unsigned)p.res1_cs * 4u;
        const unsigned l1 = n < p.res1_c ? lane_res + (unsigned)r * 4u : kLaneOff;      // out-of-range lanes read 0
        const unsigned lj[4] = {l1, l1 + rcs4, l1 + 2u * rcs4, l1 + 3u * rcs4};          // (0x80000000 + a few KB is still outside the buffer)
        float r1[16];
        unsigned so = (unsigned)nt * 4u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int j = 0; j < 4; ++j) r1[4 * q + j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_res, lj[j], so, 0));
          so += 8u * rcs4;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += r1[i];
      }
      leaky_relu_tile(v, act_alpha);
      if constexpr (H == 2) {
        if (second && p.out2_split) {
          // theta | phi | g for attention_h16.h: the operand split happens HERE, once per value, instead of in every query block's key loop.
          // Channel c of group g3 (0 theta, 1 phi, 2 g) -> halves at byte g3 * 512 + 2c (hi) and + 256 (lo) of the pixel's 1536-byte
          // record; theta is pre-scaled by log2 e (the kernel's softmax is base 2), in fp32, before the split — the same values the
          // attention kernel of rounds 2-5 computed from the fp32 buffer.  The range guard of what is converted moves here with it.
          const int n2 = nt - p.n_split;                            // multiple of 32, uniform
          const float pre = n2 < 128 ? 1.4426950408889634f : 1.f;
          const unsigned vb2 = n_ok ? ((unsigned)(4 * h) * 1536u + (unsigned)r * 2u) : kLaneOff;
          unsigned so = (unsigned)((n2 >> 7) * 512 + (n2 & 127) * 2);
          float amax = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float x = v[4 * q + j] * pre;
              amax = __builtin_fmaxf(__builtin_fabsf(x), amax);
              const _Float16 xh = (_Float16)x;
              const _Float16 xl = (_Float16)(x - (float)xh);
#if defined(BSR_NL_DIAG) && (BSR_NL_DIAG == 2 || BSR_NL_DIAG >= 4)
              if (x == 12345.678f)
#endif
              {
              __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, xh), rsrc_out2, vb2 + (unsigned)j * 1536u, so, 0);
              __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, xl), rsrc_out2, vb2 + (unsigned)j * 1536u + 256u, so, 0);
              }
            }
            so += 8u * 1536u;
          }
          range_report(amax, p.range_flag);
          continue;
        }
      }
      const unsigned vb = n_ok ? (second ? lane_out2 : lane_out) : kLaneOff;
      const unsigned cs4 = (second ? (unsigned)p.out2_cs : (unsigned)p.out_cs) * 4u;
      const unsigned vj[4] = {vb, vb + cs4, vb + 2u * cs4, vb + 3u * cs4};
      unsigned so = (second ? (unsigned)(nt - p.n_split) : (unsigned)nt) * 4u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#if defined(BSR_EPI_SKIP) || (defined(BSR_NL_DIAG) && (BSR_NL_DIAG == 3 || BSR_NL_DIAG >= 4))
          if (v[4 * q + j] == 12345.678f)        // diagnostic build only (scratch/bench_igemm.hip): the kernel without its output traffic
#endif
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[4 * q + j]), second ? rsrc_out2 : rsrc_out, vj[j], so, 0);
        }
        so += 8u * cs4;
      }
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef BSR_STAMPS
    st_epi += __builtin_amdgcn_s_memtime() - se0;
#endif
  }
#ifdef BSR_STAMPS
  if (p.stamps != nullptr && lane == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long st3 = __builtin_amdgcn_s_memtime(), rt3 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* d = p.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 4;
    d[0] = st1 - st0; d[1] = st3 - st1 - st_epi; d[2] = ((rt3 - rt0) << 32) | st_epi; d[3] = st_epi;
  }
#endif
}

template <int NI, int NCH, int H = 0, int MINW = 2, int CCF = 32>
inline hipError_t launch_gemm_nloop(ConvArgs a, size_t total_pixels, int nsplit, hipStream_t stream) {
  using C = GemmNLoopCfg<NI, NCH, H, CCF>;
  auto kern = gemm_nloop_kernel<NI, NCH, H, MINW, CCF>;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  const int tiles_total = (a.n_store + 31) / 32;
  a.tiles_x = (tiles_total + nsplit - 1) / nsplit;            // tiles per blockIdx.y range
  if (a.tiles_x > C::MAX_TILES) return hipErrorInvalidValue;
  nsplit = (tiles_total + a.tiles_x - 1) / a.tiles_x;         // no empty range
  hipLaunchKernelGGL(kern, dim3((unsigned)(total_pixels / C::BM), nsplit), dim3(256), C::SMEM_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace bsr
