// A K = 128 1x1-conv GEMM run as the TAIL of an 8-wave workgroup that has just produced its own 128 x 128 activation tile
// (round 4).  Two producers use it:
//   * nonlocal_attention_kernel<4, FUSEW>: tile = the normalised attention output of the workgroup's 128 queries, GEMM = the
//     NonLocalBlock's `w` conv + BN + block residual + LeakyReLU (/root/reference/model.py:56-59, 105-113), N = 288;
//   * igemm_conv_kernel<3,3,1,...,WN=2,NI=2, FUSE_TAIL> (res*.conv2): tile = conv2 + BN + LeakyReLU of a 4x32-pixel tile, all 128
//     channels, GEMM = conv3 + BN (+ block skip) | theta|phi|g composed (model.py:86,101,10-13), N = 288 + 384.
// The tile goes through LDS once ([pixel][128 + 4] floats) into the A-fragment registers gemm_nloop_kernel loads from HBM (64 VGPRs
// per wave: lane = pixel, channels 8g + 4h .. +3); the workgroup's two wave groups (waves 0-3 / 4-7: the same 4 x 32 pixels) take
// the channel tiles [0, TA) / [TA, TA + TB) of N, and the packed weight images of BOTH stream through one 3-slot LDS ring staged by
// all 512 threads.  Per output element the MFMA sequence is gemm_nloop_kernel<3, 4>'s (bias tile, then chunk 0..3 x K group 0..3 x 4
// matrix instructions): the fused launches are bit-identical to the separate ones.  What is saved is the separate launch's prologue
// (A-fragment and first weight loads from HBM), its one-round straggling, and the activation tile's HBM round trip.
#pragma once
#include <hip/hip_runtime.h>
#include "mfma_common.h"

namespace bsr {

struct GemmTailArgs {
  const float* w;        // packed [4][1][n_pad][36] (pack.py), n_pad >= 32 * (TA + TB + 2)
  const float* bias;     // [n_pad]
  int n_pad;
  const float* res;      // optional residual, NHWC at the tile's resolution, channels [0, res_c) added before the activation (null: none)
  int res_cs, res_c;
  float* out;            // channels [0, n_store1) of tiles below n_split (all tiles when out2 is null)
  int out_cs, n_store1;
  float* out2;           // optional second destination: channels [n_split, n_store) go to out2[.., n - n_split]
  int out2_cs, n_split, n_store;
  int act;               // 1: LeakyReLU(0.3)
  int stagger;           // 1: wave group 1 runs its SHORT channel group first, so that its epilogues fall into group 0's matrix phases
  unsigned* range_flag;  // H = 2 (16-bit modes): the handle's sticky range flag (mfma_common.h range_report); may be null
};

constexpr int kTailLdA = 128 + 4;                   // floats per pixel row of the activation tile in LDS
constexpr int kTailAFloats = 128 * kTailLdA;
constexpr int kTailSlot = 2 * 96 * 36;              // floats per ring slot: the (3 tiles x 36-word rows) images of BOTH wave groups
constexpr int kTailWPT = 4;                         // float4 per thread and ring slot: 2 x 864 over 512 threads (the surplus re-copies an element)

template <int TA, int TB>
struct GemmTailCfg {
  static constexpr int NI = 3, NCH = 4, G = 4, LDP = 36;
  static constexpr int NG = ((TA > TB ? TA : TB) + NI - 1) / NI;       // channel groups per wave group (both run the same number of steps)
  static constexpr int NSTEPS = NG * NCH;
  static constexpr int BIAS_FLOATS = (TA + NG * NI) * 32;               // group 1's last channel group may touch tiles past TA + TB (zero rows)
  static constexpr int SMEM_FLOATS = kTailAFloats + 3 * kTailSlot + BIAS_FLOATS;
  static_assert(TA >= TB && TA - TB <= NI, "wave group 0 takes the longer range");
};

template <int TA, int TB>
struct GemmTailState {
  unsigned voff[kTailWPT], loff[kTailWPT];
  f32x4 r0[kTailWPT], r1[kTailWPT];
  __amdgpu_buffer_rsrc_t rsrc;
  int sub;                                          // the wave group whose images this thread stages (tid >> 8)
};

// Channel-group schedule of a wave group: tiles [t0, t0 + nt) in groups of NI; with `short_first` the remainder group comes first
// (gemm_nloop_kernel's odd-range trick: the two wave groups of a workgroup share every SIMD, and with the same schedule they reach
// their epilogues — loads, VALU, stores, no matrix work — together).
template <int NI>
__device__ __forceinline__ int tail_group_tile0(int t0, int nt, bool short_first, int ng) {
  const int first = short_first ? (nt - 1) % NI + 1 : NI;
  return t0 + (ng == 0 ? 0 : first + (ng - 1) * NI);
}
template <int NI>
__device__ __forceinline__ int tail_group_valid(int t0, int nt, bool short_first, int ng) {
  const int first = short_first ? (nt - 1) % NI + 1 : NI;
  const int tg = ng == 0 ? 0 : first + (ng - 1) * NI;
  return ng == 0 ? min(first, nt) : min(NI, nt - tg);
}

// Each wave group stages ITS OWN weight images (waves 0-3 the images of group 0, waves 4-7 those of group 1: 864 float4 per image over
// 256 threads, the surplus re-copies an element), so the two groups may walk their channel groups in different orders.
template <int TA, int TB>
__device__ __forceinline__ void gemm_tail_fetch(GemmTailState<TA, TB>& st, const GemmTailArgs& a, int s, f32x4 (&regs)[kTailWPT]) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int ng = s >> 2, ch = s & 3;                            // step = (channel group, K chunk)
  const int tile0 = tail_group_tile0<3>(st.sub ? TA : 0, st.sub ? TB : TA, st.sub && a.stagger, ng);      // wave-uniform
  const unsigned soff = (unsigned)((ch * a.n_pad + tile0 * 32) * 36 * 4);
#pragma unroll
  for (int i = 0; i < kTailWPT; ++i)
    regs[i] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(st.rsrc, st.voff[i], soff, 0));
}

__device__ __forceinline__ void gemm_tail_store(const unsigned (&loff)[kTailWPT], float* s_ring, int slot_floats, const f32x4 (&regs)[kTailWPT]) {
  char* dst = reinterpret_cast<char*>(s_ring + slot_floats);
#pragma unroll
  for (int i = 0; i < kTailWPT; ++i) *reinterpret_cast<f32x4*>(dst + loff[i]) = regs[i];
}

// Request the weight images of steps 0 and 1 and copy the bias to LDS — as early as the caller's LDS plan allows (the ring and the
// bias live above the activation tile), so that their latency hides behind the caller's last phase.  512 threads.
template <int TA, int TB>
__device__ __forceinline__ void gemm_tail_prefetch(GemmTailState<TA, TB>& st, const GemmTailArgs& a, float* s_bias, int tid) {
  using C = GemmTailCfg<TA, TB>;
  st.rsrc = make_rsrc(a.w);
  st.sub = __builtin_amdgcn_readfirstlane(tid >> 8);
#pragma unroll
  for (int i = 0; i < kTailWPT; ++i) {
    const int e = ((tid & 255) + i * 256) % 864;                // float4 index inside this wave group's image
    st.voff[i] = (unsigned)(e * 16);
    st.loff[i] = (unsigned)((st.sub * 864 + e) * 16);           // slot layout: image of group 0 | image of group 1
  }
  gemm_tail_fetch(st, a, 0, st.r0);
  gemm_tail_fetch(st, a, 1, st.r1);
  for (int i = tid; i < C::BIAS_FLOATS; i += 512) s_bias[i] = a.bias[i];
}

// The GEMM itself.  Preconditions: the activation tile is complete in s_a as far as THIS thread's own writes go (the function's
// first barrier publishes it together with ring steps 0 and 1); gemm_tail_prefetch has run.  grp = wave group (0 / 1), wq = the
// wave's 32-pixel row of the tile, tile_pix = flattened NHWC pixel index of that row's first pixel (its 32 pixels are consecutive).
// H = 0: fp32 matrix cores.  H = 2 (round 5): the split-precision form of gemm_nloop_kernel<3, 4, 2> — the weight image is the fp16 one
// of pack_taps_h16 (the same 36-word rows: 32 halves hi | 32 halves lo | pad, so the ring machinery above is shared), the A fragments
// are read from the fp32 tile and split into hi / lo planes exactly as that kernel splits what it loads from HBM, and a 16-channel K
// group costs three v_mfma_f32_32x32x16_f16 in its order (lo.hi, hi.lo, hi.hi): bit-identical to the separate launch.
template <int TA, int TB, int H = 0>
__device__ __forceinline__ void gemm_tail_run(GemmTailState<TA, TB>& st, const GemmTailArgs& a, const float* s_a, float* s_ring, const float* s_bias,
                                              int grp, int wq, size_t tile_pix, int lane) {
  using C = GemmTailCfg<TA, TB>;
  constexpr int NI = C::NI, NCH = C::NCH, G = H ? 2 : C::G, LDP = C::LDP, NSTEPS = C::NSTEPS;
  static_assert(H == 0 || H == 2, "fp32, or split precision on the fp16 matrix cores");
  const int h = lane >> 5, r = lane & 31;
  gemm_tail_store(st.loff, s_ring, 0, st.r0);
  gemm_tail_store(st.loff, s_ring, kTailSlot, st.r1);
  __syncthreads();
  f32x4 afr[H ? 1 : NCH * G];                                  // this lane's pixel, channels 8g + 4h .. +3
  f16x8 ahi[H ? NCH * G : 1], alo[H ? NCH * G : 1];            // H = 2: channels 16g + 8h .. +7, split
  if constexpr (H == 0) {
#pragma unroll
    for (int g = 0; g < NCH * G; ++g) afr[g] = *reinterpret_cast<const f32x4*>(s_a + (wq * 32 + r) * kTailLdA + g * 8 + 4 * h);
  } else {
    float amax = 0.f;
#pragma unroll
    for (int g = 0; g < NCH * G; ++g) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(s_a + (wq * 32 + r) * kTailLdA + g * 16 + 8 * h);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(s_a + (wq * 32 + r) * kTailLdA + g * 16 + 8 * h + 4);
      split8(x0, x1, ahi[g], alo[g]);
      amax = amax8(x0, x1, amax);
    }
    range_report(amax, a.range_flag);
  }
  const int t0 = grp ? TA : 0, nt = grp ? TB : TA;             // this wave group's channel tiles [t0, t0 + nt)
  const bool short_first = grp && a.stagger;
  int b_base[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b_base[ni] = grp * (96 * LDP) + (ni * 32 + r) * LDP + 4 * h;
  int w_cur = 0, w_n1 = kTailSlot, w_n2 = 2 * kTailSlot;
  f32x4 bf[2][H ? 1 : NI];
  f16x8 bh[2][H ? NI : 1], bl[2][H ? NI : 1];
  auto read_frags = [&](int slot, int b_off) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if constexpr (H == 0) {
        bf[slot][ni] = *reinterpret_cast<const f32x4*>(s_ring + b_base[ni] + b_off);
      } else {
        bh[slot][ni] = *reinterpret_cast<const f16x8*>(s_ring + b_base[ni] + b_off);
        bl[slot][ni] = *reinterpret_cast<const f16x8*>(s_ring + b_base[ni] + b_off + 16);      // lo plane: 32 halves further
      }
    }
  };
  read_frags(0, w_cur);
  const bool has_res = a.res != nullptr;
  const float act_alpha = a.act ? kLeakyAlpha : 1.f;
  const unsigned lane_out = ((unsigned)(4 * h) * (unsigned)a.out_cs + (unsigned)r) * 4u;
  const unsigned lane_out2 = ((unsigned)(4 * h) * (unsigned)a.out2_cs + (unsigned)r) * 4u;
  const unsigned lane_res = (unsigned)(4 * h) * (unsigned)a.res_cs * 4u;
  const __amdgpu_buffer_rsrc_t rsrc_out = make_rsrc(a.out + tile_pix * a.out_cs);
  const __amdgpu_buffer_rsrc_t rsrc_out2 = make_rsrc(a.out2 != nullptr ? a.out2 + tile_pix * a.out2_cs : a.out);
  const __amdgpu_buffer_rsrc_t rsrc_res = make_rsrc(has_res ? a.res + tile_pix * a.res_cs : a.w);
#pragma unroll 1
  for (int ng = 0; ng < C::NG; ++ng) {
    const int tg = tail_group_tile0<NI>(t0, nt, short_first, ng);
    const int nvalid = tail_group_valid<NI>(t0, nt, short_first, ng);      // tiles of this group that exist (wave-uniform)
    f32x16 acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[ni] = bias_tile(h, s_bias[(tg + ni) * 32 + r]);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int s_ = ng * NCH + ch;
      const bool has1 = s_ + 1 < NSTEPS, has2 = s_ + 2 < NSTEPS;
      if (has2) gemm_tail_fetch(st, a, s_ + 2, st.r0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g + 1 < G) {
          read_frags(nxt, w_cur + (g + 1) * 8);
        } else if (has1) {
          read_frags(nxt, w_n1);
        }
        if (g == G - 1 && has2) gemm_tail_store(st.loff, s_ring, w_n2, st.r0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (H == 0) {
          const f32x4 av = afr[ch * G + g];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if (ni < nvalid) {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bf[cur][ni][j], acc[ni], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if (ni < nvalid) {
              acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ch * G + g], bh[cur][ni], acc[ni], 0, 0, 0);
              acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ch * G + g], bl[cur][ni], acc[ni], 0, 0, 0);
              acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ch * G + g], bh[cur][ni], acc[ni], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      const int tw = w_cur; w_cur = w_n1; w_n1 = w_n2; w_n2 = tw;
    }
    // epilogue of this channel group: gemm_nloop_kernel's (one residual, LeakyReLU, NHWC stores through raw buffer resources)
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if (ni >= nvalid) continue;
      const int nt = (tg + ni) * 32;
      const int n = nt + r;
      const bool second = a.out2 != nullptr && nt >= a.n_split;
      const bool n_ok = second ? n < a.n_store : (a.out2 != nullptr ? n < a.n_store1 : n < a.n_store);
      f32x16 v = acc[ni];
      if (has_res && nt < a.res_c) {
        const unsigned rcs4 = (unsigned)a.res_cs * 4u;
        const unsigned l1 = n < a.res_c ? lane_res + (unsigned)r * 4u : kLaneOff;
        const unsigned lj[4] = {l1, l1 + rcs4, l1 + 2u * rcs4, l1 + 3u * rcs4};
        float r1[16];
        unsigned so = (unsigned)nt * 4u;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) r1[4 * q4 + j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_res, lj[j], so, 0));
          so += 8u * rcs4;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += r1[i];
      }
      leaky_relu_tile(v, act_alpha);
      const unsigned vb = n_ok ? (second ? lane_out2 : lane_out) : kLaneOff;
      const unsigned cs4 = (second ? (unsigned)a.out2_cs : (unsigned)a.out_cs) * 4u;
      const unsigned vj[4] = {vb, vb + cs4, vb + 2u * cs4, vb + 3u * cs4};
      unsigned so = (second ? (unsigned)(nt - a.n_split) : (unsigned)nt) * 4u;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[4 * q4 + j]), second ? rsrc_out2 : rsrc_out, vj[j], so, 0);
        so += 8u * cs4;
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
}

}  // namespace bsr
