// Fused single-head attention for NonLocalBlock (/root/reference/model.py:51-53) in SPLIT PRECISION on the fp16 matrix cores:
// the same flash-style algorithm as attention.h (S^T = phi . theta^T per 32-key tile, online softmax in registers with the
// query on the lane, O^T += g^T . P^T with exp2(S^T) used directly as the B operand), but every fp32 operand — theta, phi, g
// and the probabilities P — is split into hi + lo fp16 halves and each contraction issues hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation (igemm_h16.h explains the numerics: ~2^-22 per product).
//
// Operand maps of v_mfma_f32_32x32x16_f16 (cdna_hip_programming.md §3): lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j]
// and B[k = 8h + j][col r], j = 0..7; C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4h.
//  * S^T: A = phi tile in LDS, [key][128 hi | 128 lo halves], one ds_read_b128 per plane and 16-channel K step; B = theta of the
//    lane's query, split once into registers.
//  * O^T: B = P^T straight from the S^T accumulator: registers 8t..8t+7 of lane half h are keys 16t + 8(j >> 2) + 4h + (j & 3).
//    A = g^T must present the same key order along k.  The g tile is staged exactly like phi — row-major [key][128 hi | 128 lo]
//    with 16-byte stores — and TRANSPOSED ON READ by ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per 16-lane group the
//    instruction reads a block of 4 keys x 16 channels and hands lane i the 4 keys of channel c0 + i, i.e. four consecutive k of
//    the A operand; two such reads (keys 16t + 4h .. +3 and 16t + 8 + 4h .. +3) make the 8-element fragment in P's key order.
//    (A first version transposed at staging time — one channel x 16 keys per thread, 16 dword loads — and was bound by those
//    narrow loads.)  Row stride 576 bytes puts the 4 keys of a block on disjoint bank quarters: conflict-free.
//
// Workgroup = 8 waves = 128 queries x TWO key streams: waves 0-3 take the even 32-key tiles, waves 4-7 the odd ones, each with
// its own running (max, sum, O^T); the two partial results are merged through LDS at the end (O = O0 2^(m0-m) + O1 2^(m1-m), same
// for the sums).  With one wave per SIMD the softmax / operand-split VALU work and the staging of a tile cannot overlap that wave's
// own matrix instructions (measured: 4 800 cycles per tile for 1 536 cycles of MFMA); two waves per SIMD on different tiles do
// overlap (fp16 matrix instructions and VALU co-issue), and a tile pair is staged by 512 threads, half the per-thread work.
#pragma once
#include <hip/hip_runtime.h>
#include "attention.h"
#include "igemm_h16.h"

#ifndef BSR_AX3_STAGGER
#define BSR_AX3_STAGGER 0
#endif
#ifndef BSR_AX3_PRIO
#define BSR_AX3_PRIO 0      // 0: equal priorities; 1 / 2: wave group 0 / 1 at s_setprio 2 (measured: profiles/HISTORY.md round 5)
#endif

namespace bsr {

constexpr int kAx3LdK = 132;                                   // words per phi row: 64 (hi) + 64 (lo) + 4 pad
constexpr int kAx3LdV = 144;                                   // words per g row: 64 (hi) + 64 (lo) + 16 pad (576 B: bank offset 16 words per key)
constexpr int kAx3StageWords = kAttKT * kAx3LdK + kAttKT * kAx3LdV;    // one 32-key tile: phi rows + g rows
constexpr int kAx3SmemBytes = 4 * kAx3StageWords * 4;                  // two tile PAIRS (double buffer)
static_assert(kAx3SmemBytes <= 160 * 1024, "LDS budget");
static_assert(66 * 64 * 4 <= 4 * kAx3StageWords, "merge scratch fits the staging buffers");
// FUSEW (round 5): the NonLocalBlock's `w` conv + BN + block residual + LeakyReLU as the TAIL of this kernel, as in attention.h — the
// workgroup's 128 normalised query rows go through LDS (fp32, over the merge scratch) into the split A fragments of
// gemm_tail_run<5, 4, 2>, the two key-stream wave groups take the channel tiles [0,5) / [5,9) of N = 288.  Same split, same
// matrix-instruction order per output element as gemm_nloop_kernel<3, 4, 2> reading the attention output from HBM: bit-identical.
constexpr int kAx3WSmemBytes = AttWCfg::SMEM_FLOATS * 4;
static_assert(kAx3WSmemBytes <= 160 * 1024 && kAx3WSmemBytes >= kAx3SmemBytes, "LDS budget of the fused tail");

template <bool FUSEW = false>
__global__ __launch_bounds__(512, 2) void nonlocal_attention_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out, int tokens,
                                                                       unsigned* __restrict__ range_flag, AttWArgs wa) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;                    // key stream (even / odd tiles), query block of 32
  const int h = lane >> 5, r = lane & 31;
  const int qblocks = tokens / 128;
  int img, qb;
  {   // query blocks of one image share an XCD's L2 copy of K/V (attention.h)
    const int nblk = gridDim.x, b = blockIdx.x;
    const int per_round = 8 * qblocks;
    if (nblk % per_round == 0) {
      const int round = b / per_round, within = b % per_round;
      img = round * 8 + (within % 8);
      qb = within / 8;
    } else {
      img = b / qblocks;
      qb = b % qblocks;
    }
  }
  const float* base = qkv + (size_t)img * tokens * (3 * kAttD);
  const int q = qb * 128 + wq * 32 + r;
#if BSR_AX3_PRIO
  // The two key-stream wave groups share every SIMD and run the same program between the same barriers: left alone they stay in
  // lock-step — both in their matrix phase, then both in their softmax / split / staging phase — and nothing overlaps.  A static
  // priority makes one group win every arbitration: it runs ahead until it needs the other pipe, and the groups settle half a phase
  // apart (MI355X_MICROARCH.md, "Two waves per SIMD", items 4 and 9).
  if (grp == (BSR_AX3_PRIO - 1)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif

  // theta of this lane's query, pre-scaled by log2(e) (softmax in base 2), split: K step s covers channels 16s + 8h .. +7
  f16x8 qh[kAttD / 16], ql[kAttD / 16];
  float amax = 0.f;                                            // range guard of the 16-bit modes (igemm_h16.h): theta here, phi / g at staging
#pragma unroll
  for (int s = 0; s < kAttD / 16; ++s) {
    const float* src = base + (size_t)q * (3 * kAttD) + 16 * s + 8 * h;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src) * 1.4426950408889634f;
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + 4) * 1.4426950408889634f;
    split8(a, b, qh[s], ql[s]);
    amax = amax8(a, b, amax);
  }
  range_report(amax, range_flag);

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging of a tile PAIR (2p, 2p+1) by 512 threads: phi and g as 8-channel pieces (2 x 512 each; piece i of a thread belongs to tile 2p+i)
  f32x4 kreg[4], vreg[4];
  const int kkey = tid >> 4, kc8 = tid & 15;
  auto fetch = [&](int pr) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* row = base + (size_t)((2 * pr + i) * kAttKT + kkey) * (3 * kAttD) + kAttD + kc8 * 8;
      kreg[2 * i] = *reinterpret_cast<const f32x4*>(row);
      kreg[2 * i + 1] = *reinterpret_cast<const f32x4*>(row + 4);
      vreg[2 * i] = *reinterpret_cast<const f32x4*>(row + kAttD);
      vreg[2 * i + 1] = *reinterpret_cast<const f32x4*>(row + kAttD + 4);
    }
  };
  auto publish = [&](int pbuf) {                               // pbuf = 0/1: which pair buffer (2 tiles each)
    range_report(amax8(vreg[0], vreg[1], amax8(vreg[2], vreg[3], amax8(kreg[0], kreg[1], amax8(kreg[2], kreg[3], 0.f)))), range_flag);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* sk = smem + (2 * pbuf + i) * kAx3StageWords;
      f16x8 hi, lo;
      split8(kreg[2 * i], kreg[2 * i + 1], hi, lo);
      *reinterpret_cast<f16x8*>(sk + kkey * kAx3LdK + kc8 * 4) = hi;
      *reinterpret_cast<f16x8*>(sk + kkey * kAx3LdK + 64 + kc8 * 4) = lo;
    }
  };
  auto publish_v = [&](int pbuf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* sv = smem + (2 * pbuf + i) * kAx3StageWords + kAttKT * kAx3LdK;
      f16x8 hi, lo;
      split8(vreg[2 * i], vreg[2 * i + 1], hi, lo);
      *reinterpret_cast<f16x8*>(sv + kkey * kAx3LdV + kc8 * 4) = hi;
      *reinterpret_cast<f16x8*>(sv + kkey * kAx3LdV + 64 + kc8 * 4) = lo;
    }
  };
  // transposed reads of the g tile: lane i of 16-lane group gq addresses key 4 (gq >> 1) + (i >> 2), channels 16 (gq & 1) + 4 (i & 3) .. +3
  typedef short s16x4v __attribute__((__vector_size__(8)));
  const int vbase = ((4 * (lane >> 5) + ((lane & 15) >> 2)) * kAx3LdV) + 8 * ((lane >> 4) & 1) + 2 * (lane & 3);
  auto read_vt = [&](const float* sv, int word_off) -> f16x8 {     // keys {0..3} and {8..11} relative to the addressed row
    const s16x4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(sv + vbase + word_off));
    const s16x4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(sv + vbase + word_off + 8 * kAx3LdV));
    typedef short s16x8v __attribute__((__vector_size__(16)));
    const s16x8v ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, ab);
  };

  const int npair = tokens / (2 * kAttKT);
  fetch(0);
  publish(0);
  publish_v(0);
  __syncthreads();

#if BSR_AX3_STAGGER
  // The two key-stream wave groups share every SIMD, run the same program and meet at one barrier per tile pair: left alone they stay in
  // lock-step — both in their matrix phases (S^T, then O^T), both in their VALU phases (softmax + split of P, split + LDS staging of the
  // next pair) — and neither pipe is busy half the time (measured: 9 400 cycles per pair and SIMD for 3 072 cycles of matrix work).
  // Staggered: group 1 does its share of the NEXT pair's staging at the START of a step (the buffer is free since the last barrier)
  // instead of at its end, so its phases run half a step out of phase with group 0's: stage | S | softmax | O against S | softmax | O |
  // stage.  It therefore fetches one pair further ahead.
  if (grp == 1 && npair > 1) fetch(1);
#endif
  for (int pr = 0; pr < npair; ++pr) {
    const int pbuf = pr & 1;
#if BSR_AX3_STAGGER
    if (grp == 1) {
      if (pr + 1 < npair) {
        publish(pbuf ^ 1);
        publish_v(pbuf ^ 1);
        if (pr + 2 < npair) fetch(pr + 2);
      }
    } else if (pr + 1 < npair) {
      fetch(pr + 1);
    }
#else
    if (pr + 1 < npair) fetch(pr + 1);
#endif
    const float* sk = smem + (2 * pbuf + grp) * kAx3StageWords;        // this wave's tile of the pair
    const float* sv = sk + kAttKT * kAx3LdK;

    f32x16 s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < kAttD / 16; ++ks) {
      const f16x8 kh = *reinterpret_cast<const f16x8*>(sk + r * kAx3LdK + ks * 8 + 4 * h);
      const f16x8 kl = *reinterpret_cast<const f16x8*>(sk + r * kAx3LdK + 64 + ks * 8 + 4 * h);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
    }

    // online softmax, exactly as attention.h: the decision covers this tile's P before any of it is exponentiated
    float mx = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__any(mx > m_run + kRescaleThreshold)) {
      const float m_new = fmaxf(m_run, mx);
      const float scale = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= scale;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= scale;
    }
    float psum = 0.f;
    f16x8 ph[2], pl[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[i] = __builtin_amdgcn_exp2f(s[i] - m_run);                 // <= 2^8: inside the fp16 range
      psum += s[i];
    }
    l_run += psum;
    split8(f32x4{s[0], s[1], s[2], s[3]}, f32x4{s[4], s[5], s[6], s[7]}, ph[0], pl[0]);
    split8(f32x4{s[8], s[9], s[10], s[11]}, f32x4{s[12], s[13], s[14], s[15]}, ph[1], pl[1]);

#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f16x8 vh = read_vt(sv, 16 * t * kAx3LdV + 16 * dt);            // channels 32 dt .., keys 16 t ..
        const f16x8 vl = read_vt(sv, 16 * t * kAx3LdV + 16 * dt + 64);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[t], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[t], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[t], o[dt], 0, 0, 0);
      }

    if (pr + 1 < npair) {
#if BSR_AX3_STAGGER
      if (grp == 0) {
        publish(pbuf ^ 1);
        publish_v(pbuf ^ 1);
      }
#else
      publish(pbuf ^ 1);
      publish_v(pbuf ^ 1);
#endif
      __syncthreads();
    }
  }

  // merge the two key streams: waves 4-7 hand (m, l, O^T) to waves 0-3 through LDS ([wq][66 values][64 lanes])
  __syncthreads();
  // FUSEW: the staging buffers are dead from here on; the weight images of GEMM steps 0 and 1 and the bias are requested now, by all
  // eight waves, so that their latency hides behind the merge (ring and bias live ABOVE the merge scratch / attention tile)
  [[maybe_unused]] float* s_ring = smem + kTailAFloats;
  [[maybe_unused]] float* s_bias = s_ring + 3 * kTailSlot;
  [[maybe_unused]] GemmTailState<5, 4> tail;
  if constexpr (FUSEW) gemm_tail_prefetch(tail, wa, s_bias, tid);
  float* sx = smem + (size_t)wq * 66 * 64 + lane;
  if (grp == 1) {
    sx[0] = m_run;
    sx[64] = l_run;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) sx[(2 + dt * 16 + i) * 64] = o[dt][i];
  }
  __syncthreads();
  if constexpr (!FUSEW) {
    if (grp == 1) return;
  }
  float inv = 0.f;
  if (grp == 0) {
    const float m1 = sx[0], l1 = sx[64];
    const float m = fmaxf(m_run, m1);
    const float s0 = __builtin_amdgcn_exp2f(m_run - m), s1 = __builtin_amdgcn_exp2f(m1 - m);
    l_run = l_run * s0 + l1 * s1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[dt][i] = o[dt][i] * s0 + sx[(2 + dt * 16 + i) * 64] * s1;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    inv = 1.f / l_tot;
  }

  // y[q][d], d = 32 dt + (i & 3) + 8 (i >> 2) + 4h: four consecutive channels per register quad
  if constexpr (!FUSEW) {
    float* orow = out + ((size_t)img * tokens + q) * kAttD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 v = {o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv, o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g4 + 4 * h) = v;
      }
  } else {
    // ---- the `w` GEMM tail (gemm_tail.h, H = 2: gemm_nloop_kernel<3, 4, 2> with the activation tile coming through LDS instead of HBM) ----
    __syncthreads();                                           // every read of the merge scratch is done: the attention tile may overwrite it
    float* s_att = smem;                                       // [128 queries][kTailLdA] fp32: the values the unfused kernel stores as att
    if (grp == 0) {
      float* arow = s_att + (wq * 32 + r) * kTailLdA;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 v = {o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv, o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv};
          *reinterpret_cast<f32x4*>(arow + 32 * dt + 8 * g4 + 4 * h) = v;
        }
    }
    const size_t tile_pix = (size_t)img * tokens + (size_t)qb * 128 + (size_t)__builtin_amdgcn_readfirstlane(wq) * 32;
    gemm_tail_run<5, 4, 2>(tail, wa, s_att, s_ring, s_bias, grp, wq, tile_pix, lane);
  }
}

inline hipError_t launch_nonlocal_attention_x3(const float* qkv, float* out, int batch, int tokens, hipStream_t stream, unsigned* range_flag = nullptr) {
  if (tokens % (2 * kAttKT) != 0) return hipErrorInvalidValue;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nonlocal_attention_x3_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kAx3SmemBytes);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  hipLaunchKernelGGL(nonlocal_attention_x3_kernel<false>, dim3(batch * (tokens / 128)), dim3(512), kAx3SmemBytes, stream, qkv, out, tokens, range_flag, AttWArgs{});
  return hipGetLastError();
}

// attention + `w` GEMM tail in one launch, split precision (the 16-bit modes at any batch: this kernel has one workgroup shape)
inline hipError_t launch_nonlocal_attention_x3_w(const float* qkv, int batch, int tokens, const AttWArgs& wa, hipStream_t stream, unsigned* range_flag = nullptr) {
  if (tokens % (2 * kAttKT) != 0 || tokens % 128 != 0 || wa.n_pad < 12 * 32 || wa.n_store > 288 || wa.res_c > 288 || wa.out2 != nullptr) return hipErrorInvalidValue;
  auto kern = nonlocal_attention_x3_kernel<true>;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kAx3WSmemBytes);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  AttWArgs w2 = wa;
  w2.range_flag = range_flag;
  hipLaunchKernelGGL(kern, dim3(batch * (tokens / 128)), dim3(512), kAx3WSmemBytes, stream, qkv, static_cast<float*>(nullptr), tokens, range_flag, w2);
  return hipGetLastError();
}

}  // namespace bsr
