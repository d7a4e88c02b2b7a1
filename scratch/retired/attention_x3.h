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

#ifndef BSR_AX3_ROLES
#define BSR_AX3_ROLES 0     // 1: S-waves / PV-waves (nonlocal_attention_x3r_kernel: built, correct, 60.2-60.8 us against 58.3 — profiles/HISTORY.md); 0: two key streams of identical waves
#endif
#ifndef BSR_AX3R_UPFRONT
#define BSR_AX3R_UPFRONT 0      // K steps the fragment reads of a matrix phase run ahead (0: as the compiler schedules them; 1-4 spill at the 256-VGPR cap)
#endif
#ifndef BSR_AX3R_PRIO
#define BSR_AX3R_PRIO 1
#endif
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

// ---- Round 5 EXPERIMENT (opt-in: -DBSR_AX3_ROLES=1; correct, not faster): the two waves of a SIMD in DIFFERENT ROLES ------------------
// Counters of the kernel above (profiles/HISTORY.md, round 5): matrix pipe 34 % busy, vector and matrix instructions executing together in
// a fifth of the matrix-busy cycles, a wave 45 % of its life issue-stalled — each wave runs matrix phase, vector phase, matrix phase, vector
// phase in sequence, its SIMD partner runs the same program, and neither de-phasing experiment moved it.  Here the overlap is structural:
//   * S-waves (waves 0-3, one per SIMD, 32 queries each): S^T = phi . theta^T of tile t (24 matrix instructions), the online softmax and
//     the hi / lo split of P (vector), and hand P^T (already in B-fragment layout: the PV wave's lane holds the same registers) plus the
//     rescale factor of the step to LDS;
//   * PV-waves (waves 4-7, the SIMD partners of the same 32 queries): O^T += g^T . P^T of tile t - 1 (24 matrix instructions) and all of
//     the staging (global loads -> hi / lo split -> LDS) of tile t + 1.
// While one wave of a SIMD is in its vector phase the other is in its matrix phase.  One tile (32 keys) per barrier; g tiles live one
// step longer than phi tiles (three g buffers, two phi buffers: 89 KB) + the P hand-off (35 KB).  One key stream per query: no merge.
constexpr int kAx3rK = kAttKT * kAx3LdK, kAx3rV = kAttKT * kAx3LdV;          // words of one phi / g tile
constexpr int kAx3rP = 64 * 16 + 64;                                          // words of one P hand-off: 4 fragments x 64 lanes x 16 B + 64 scale factors
constexpr int kAx3rPOff = 2 * kAx3rK + 3 * kAx3rV;
constexpr int kAx3rWords = kAx3rPOff + 4 * 2 * kAx3rP;
constexpr int kAx3rSmemBytes = kAx3rWords * 4;
static_assert(kAx3rSmemBytes <= 160 * 1024, "LDS budget");

// The step barrier: every LDS access of this wave done (lgkmcnt 0), then s_barrier — NOT __syncthreads(), in front of which hipcc also
// drains vmcnt to 0: the PV-waves' loads of tile t + 2 are issued in step t and consumed in step t + 1, and a full drain at the end of
// step t would put one HBM round trip (3-5 k cycles) into every 1.5-k-cycle step (measured: 60.9 us per launch with __syncthreads()).
__device__ __forceinline__ void ax3r_barrier() {
  __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));
  __builtin_amdgcn_s_barrier();
}

// Diagnostic build (-DBSR_AX3_STAMPS, attention-only launches): waves 0 (S) and 4 (PV) of workgroup 0 record s_memtime at four points of steps
// 8..15 into the 512 bytes BEHIND `out` (the caller allocates them) — where a step's cycles go, per role
#ifdef BSR_AX3_STAMPS
#define AX3R_STAMP(k) do { if (!FUSEW && blockIdx.x == 0 && wq == 0 && lane == 0 && t >= 8 && t < 16) { \
    __builtin_amdgcn_s_waitcnt(0); \
    reinterpret_cast<unsigned long long*>(out + (size_t)gridDim.x * 128 * kAttD)[(role * 8 + (t - 8)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define AX3R_STAMP(k) do { } while (0)
#endif

template <bool FUSEW = false>
__global__ __launch_bounds__(512, 2) void nonlocal_attention_x3r_kernel(const float* __restrict__ qkv, float* __restrict__ out, int tokens,
                                                                        unsigned* __restrict__ range_flag, AttWArgs wa) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, wq = wave & 3;                   // 0 = S-wave, 1 = PV-wave; query block of 32 (waves w and w + 4 share SIMD w)
  const int h = lane >> 5, r = lane & 31;
  const int qblocks = tokens / 128;
  int img, qb;
  {
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
  const int ntile = tokens / kAttKT;
  float* s_p = smem + kAx3rPOff + wq * (2 * kAx3rP);           // this query block's two hand-off buffers
  typedef short s16x4v __attribute__((__vector_size__(8)));
  typedef short s16x8v __attribute__((__vector_size__(16)));

  // Staging is split by operand: the S-waves stage phi (which they alone read), the PV-waves g.  256 threads per operand: two 8-channel
  // pieces per thread and tile, requested TWO steps before they are split and written (two register sets, by tile parity), so that a
  // request has two whole steps to land (stamps of a first version with one set and all staging on the PV-waves: 2 500 cycles per step
  // at the head of the PV-wave's step, most of them the wait for the loads it had issued one step earlier, the S-wave 2 000 cycles at the barrier).
  const int p_ = tid & 255;
  const int pkey0 = p_ >> 4, pc8 = p_ & 15;                    // piece i of this thread: key pkey0 + 16 i, channels 8 pc8 .. +7
  const float* prow = base + (size_t)pkey0 * (3 * kAttD) + kAttD + role * kAttD + pc8 * 8;      // phi (role 0) or g (role 1)
  f32x4 stg[2][4];
  auto fetch = [&](int t, f32x4 (&rg)[4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* row = prow + (size_t)(t * kAttKT + 16 * i) * (3 * kAttD);
      rg[2 * i] = *reinterpret_cast<const f32x4*>(row);
      rg[2 * i + 1] = *reinterpret_cast<const f32x4*>(row + 4);
    }
  };
  auto publish = [&](int t, const f32x4 (&rg)[4]) {             // tile t: phi -> buffer t & 1 (rows of kAx3LdK words), g -> buffer t % 3 (kAx3LdV)
    float* dst = role == 0 ? smem + (t & 1) * kAx3rK : smem + 2 * kAx3rK + (t % 3) * kAx3rV;
    const int ld = role == 0 ? kAx3LdK : kAx3LdV;
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f16x8 hi, lo;
      split8(rg[2 * i], rg[2 * i + 1], hi, lo);
      am = amax8(rg[2 * i], rg[2 * i + 1], am);
      *reinterpret_cast<f16x8*>(dst + (pkey0 + 16 * i) * ld + pc8 * 4) = hi;
      *reinterpret_cast<f16x8*>(dst + (pkey0 + 16 * i) * ld + 64 + pc8 * 4) = lo;
    }
    range_report(am, range_flag);
  };
  fetch(0, stg[0]);
  publish(0, stg[0]);
  if (ntile > 1) fetch(1, stg[1]);
  if (ntile > 2) fetch(2, stg[0]);

#if BSR_AX3R_PRIO
  if (role == 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);      // the S-wave's matrix phase first, the PV-wave's beside the softmax
#endif
  if (role == 0) {
    // ---------------- S-wave ----------------
    f16x8 qh[kAttD / 16], ql[kAttD / 16];
    float amax = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < kAttD / 16; ++s_) {
      const float* src = base + (size_t)q * (3 * kAttD) + 16 * s_ + 8 * h;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src) * 1.4426950408889634f;
      const f32x4 b = *reinterpret_cast<const f32x4*>(src + 4) * 1.4426950408889634f;
      split8(a, b, qh[s_], ql[s_]);
      amax = amax8(a, b, amax);
    }
    range_report(amax, range_flag);
    float m_run = -INFINITY, l_run = 0.f;
    __syncthreads();                                           // tile 0 staged
    auto s_step = [&](int t, f32x4 (&rg)[4]) {                  // rg = the register set of tile t + 1
      AX3R_STAMP(0);
      if (t < ntile) {
        const float* sk = smem + (t & 1) * kAx3rK;
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#if BSR_AX3R_UPFRONT
        // fragment reads run BSR_AX3R_UPFRONT K steps ahead of the matrix instructions that use them: with reads issued one step ahead
        // (what the compiler schedules) every group of three 32-cycle instructions waited out an LDS round trip — 44 cycles per
        // instruction by the stamps, 75 in the PV-wave.  (All sixteen up front: 256 VGPRs and 148 bytes of scratch — 86 us.)
        constexpr int DEPTH = BSR_AX3R_UPFRONT;
        f16x8 kh[DEPTH], kl[DEPTH];
#pragma unroll
        for (int ks = 0; ks < DEPTH; ++ks) {
          kh[ks] = *reinterpret_cast<const f16x8*>(sk + r * kAx3LdK + ks * 8 + 4 * h);
          kl[ks] = *reinterpret_cast<const f16x8*>(sk + r * kAx3LdK + 64 + ks * 8 + 4 * h);
        }
#pragma unroll
        for (int ks = 0; ks < kAttD / 16; ++ks) {
          __builtin_amdgcn_sched_barrier(0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[ks % DEPTH], qh[ks], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[ks % DEPTH], ql[ks], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[ks % DEPTH], qh[ks], s, 0, 0, 0);
          if (ks + DEPTH < kAttD / 16) {
            kh[ks % DEPTH] = *reinterpret_cast<const f16x8*>(sk + r * kAx3LdK + (ks + DEPTH) * 8 + 4 * h);
            kl[ks % DEPTH] = *reinterpret_cast<const f16x8*>(sk + r * kAx3LdK + 64 + (ks + DEPTH) * 8 + 4 * h);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
        for (int ks = 0; ks < kAttD / 16; ++ks) {
          const f16x8 kh = *reinterpret_cast<const f16x8*>(sk + r * kAx3LdK + ks * 8 + 4 * h);
          const f16x8 kl = *reinterpret_cast<const f16x8*>(sk + r * kAx3LdK + 64 + ks * 8 + 4 * h);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
        }
#endif
        AX3R_STAMP(1);
        float mx = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float scale = 1.f;
        if (__any(mx > m_run + kRescaleThreshold)) {
          const float m_new = fmaxf(m_run, mx);
          scale = __builtin_amdgcn_exp2f(m_run - m_new);       // exp2(-inf) = 0 on the first tile; 1 for lanes whose max did not move
          l_run *= scale;
          m_run = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[i] = __builtin_amdgcn_exp2f(s[i] - m_run);         // <= 2^8: inside the fp16 range
          psum += s[i];
        }
        l_run += psum;
        f16x8 ph[2], pl[2];
        split8(f32x4{s[0], s[1], s[2], s[3]}, f32x4{s[4], s[5], s[6], s[7]}, ph[0], pl[0]);
        split8(f32x4{s[8], s[9], s[10], s[11]}, f32x4{s[12], s[13], s[14], s[15]}, ph[1], pl[1]);
        float* sp = s_p + (t & 1) * kAx3rP;
        *reinterpret_cast<f16x8*>(sp + 0 * 256 + lane * 4) = ph[0];
        *reinterpret_cast<f16x8*>(sp + 1 * 256 + lane * 4) = pl[0];
        *reinterpret_cast<f16x8*>(sp + 2 * 256 + lane * 4) = ph[1];
        *reinterpret_cast<f16x8*>(sp + 3 * 256 + lane * 4) = pl[1];
        sp[1024 + lane] = scale;
        if (t + 1 < ntile) {                                   // phi of tile t + 1 (requested two steps ago) -> the buffer tile t - 1 has left
          publish(t + 1, rg);
          if (t + 3 < ntile) fetch(t + 3, rg);
        }
      } else {
        // after the last tile: the normalisation factor for the PV-wave (the two lane halves summed 16 keys of every tile each)
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        s_p[(t & 1) * kAx3rP + 1024 + lane] = 1.f / l_tot;
      }
      AX3R_STAMP(2);
      ax3r_barrier();
      AX3R_STAMP(3);
    };
    for (int t = 0; t <= ntile; t += 2) {
      s_step(t, stg[1]);
      if (t + 1 <= ntile) s_step(t + 1, stg[0]);
    }
  } else {
    // ---------------- PV-wave: O^T of the previous tile, then the g tile of the next ----------------
    const int vbase = ((4 * (lane >> 5) + ((lane & 15) >> 2)) * kAx3LdV) + 8 * ((lane >> 4) & 1) + 2 * (lane & 3);
    auto read_vt = [&](const float* sv, int word_off) -> f16x8 {
      const s16x4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(sv + vbase + word_off));
      const s16x4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(sv + vbase + word_off + 8 * kAx3LdV));
      const s16x8v ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(f16x8, ab);
    };
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    __syncthreads();
    auto pv_step = [&](int t, f32x4 (&rg)[4]) {                 // rg = the register set of tile t + 1
      AX3R_STAMP(0);
      if (t >= 1) {                                            // O^T += g^T . P^T of tile t - 1
        const float* sp = s_p + ((t - 1) & 1) * kAx3rP;
        const float* sv = smem + 2 * kAx3rK + ((t - 1) % 3) * kAx3rV;
        const float scale = sp[1024 + lane];
        if (__any(scale != 1.f)) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[dt][i] *= scale;
        }
        f16x8 ph[2], pl[2];
        ph[0] = *reinterpret_cast<const f16x8*>(sp + 0 * 256 + lane * 4);
        pl[0] = *reinterpret_cast<const f16x8*>(sp + 1 * 256 + lane * 4);
        ph[1] = *reinterpret_cast<const f16x8*>(sp + 2 * 256 + lane * 4);
        pl[1] = *reinterpret_cast<const f16x8*>(sp + 3 * 256 + lane * 4);
#if BSR_AX3R_UPFRONT
        constexpr int DEPTH = BSR_AX3R_UPFRONT;
        f16x8 vh[DEPTH], vl[DEPTH];                            // group g = (tt, dt): g = 4 tt + dt — four independent accumulators in turn
#pragma unroll
        for (int g = 0; g < DEPTH; ++g) {
          vh[g] = read_vt(sv, 16 * (g >> 2) * kAx3LdV + 16 * (g & 3));
          vl[g] = read_vt(sv, 16 * (g >> 2) * kAx3LdV + 16 * (g & 3) + 64);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_barrier(0);
          const int tt = g >> 2, dt = g & 3;
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[g % DEPTH], ph[tt], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[g % DEPTH], pl[tt], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[g % DEPTH], ph[tt], o[dt], 0, 0, 0);
          if (g + DEPTH < 8) {
            vh[g % DEPTH] = read_vt(sv, 16 * ((g + DEPTH) >> 2) * kAx3LdV + 16 * ((g + DEPTH) & 3));
            vl[g % DEPTH] = read_vt(sv, 16 * ((g + DEPTH) >> 2) * kAx3LdV + 16 * ((g + DEPTH) & 3) + 64);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const f16x8 vh = read_vt(sv, 16 * tt * kAx3LdV + 16 * dt);
            const f16x8 vl = read_vt(sv, 16 * tt * kAx3LdV + 16 * dt + 64);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[tt], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[tt], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[tt], o[dt], 0, 0, 0);
          }
#endif
      }
      AX3R_STAMP(1);
      if (t + 1 < ntile) {                                     // g of tile t + 1 -> the buffer tile t - 2 has left
        publish(t + 1, rg);
        if (t + 3 < ntile) fetch(t + 3, rg);
      }
      AX3R_STAMP(2);
      ax3r_barrier();
      AX3R_STAMP(3);
    };
    for (int t = 0; t <= ntile; t += 2) {
      pv_step(t, stg[1]);
      if (t + 1 <= ntile) pv_step(t + 1, stg[0]);
    }
    const float inv = s_p[(ntile & 1) * kAx3rP + 1024 + lane];  // written by the S-wave in its last step, published by that step's barrier
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
      __syncthreads();                                         // every wave is past its last LDS read: the attention tile may overwrite the buffers
      float* arow = smem + (wq * 32 + r) * kTailLdA;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 v = {o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv, o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv};
          *reinterpret_cast<f32x4*>(arow + 32 * dt + 8 * g4 + 4 * h) = v;
        }
    }
  }
  if constexpr (FUSEW) {
    if (role == 0) __syncthreads();                            // pairs with the PV-waves' barrier above
    // ---- the `w` GEMM tail: S-waves = wave group 0 (channel tiles [0,5)), PV-waves = group 1 ([5,9)), as in the kernel above ----
    float* s_ring = smem + kTailAFloats;
    float* s_bias = s_ring + 3 * kTailSlot;
    GemmTailState<5, 4> tail;
    gemm_tail_prefetch(tail, wa, s_bias, tid);
    const size_t tile_pix = (size_t)img * tokens + (size_t)qb * 128 + (size_t)__builtin_amdgcn_readfirstlane(wq) * 32;
    gemm_tail_run<5, 4, 2>(tail, wa, smem, s_ring, s_bias, role, wq, tile_pix, lane);
  }
}

inline hipError_t launch_nonlocal_attention_x3(const float* qkv, float* out, int batch, int tokens, hipStream_t stream, unsigned* range_flag = nullptr) {
  if (tokens % (2 * kAttKT) != 0) return hipErrorInvalidValue;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
#if BSR_AX3_ROLES
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nonlocal_attention_x3r_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kAx3rSmemBytes);
#else
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nonlocal_attention_x3_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kAx3SmemBytes);
#endif
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
#if BSR_AX3_ROLES
  hipLaunchKernelGGL(nonlocal_attention_x3r_kernel<false>, dim3(batch * (tokens / 128)), dim3(512), kAx3rSmemBytes, stream, qkv, out, tokens, range_flag, AttWArgs{});
#else
  hipLaunchKernelGGL(nonlocal_attention_x3_kernel<false>, dim3(batch * (tokens / 128)), dim3(512), kAx3SmemBytes, stream, qkv, out, tokens, range_flag, AttWArgs{});
#endif
  return hipGetLastError();
}

// attention + `w` GEMM tail in one launch, split precision (the 16-bit modes at any batch: this kernel has one workgroup shape)
inline hipError_t launch_nonlocal_attention_x3_w(const float* qkv, int batch, int tokens, const AttWArgs& wa, hipStream_t stream, unsigned* range_flag = nullptr) {
  if (tokens % (2 * kAttKT) != 0 || tokens % 128 != 0 || wa.n_pad < 12 * 32 || wa.n_store > 288 || wa.res_c > 288 || wa.out2 != nullptr) return hipErrorInvalidValue;
#if BSR_AX3_ROLES
  auto kern = nonlocal_attention_x3r_kernel<true>;
#else
  auto kern = nonlocal_attention_x3_kernel<true>;
#endif
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
