// Fused single-head attention for NonLocalBlock (/root/reference/model.py:51-53):
//   f = theta . phi^T  [HW x HW]   (NO 1/sqrt(d) scaling),  P = softmax(f, -1),  y = P . g
// on the fp32 matrix cores, flash-style (online softmax) so the 4 MB/image score matrix never leaves
// the CU.  Layout of the qkv buffer: [B][HW][3*D] rows = tokens (t = h*W + w, A.8), channels
// [0,D) = theta, [D,2D) = phi, [2D,3D) = g.  Output y: [B][HW][D].
//
// One workgroup = 4 waves = 128 query tokens of one image; each wave owns 32 queries.  Per 32-key tile:
//   S^T[key][q]  = sum_c phi[key][c] * theta[q][c]      A = phi tile (LDS, ds_read_b128), B = theta (registers)
//   -> the accumulator puts the QUERY on the lane (column) and 16 keys in the registers, so the row
//      max / sum is in-lane plus one exchange with lane^32, and exp(S^T) is already the B operand of
//   O^T[d][q]   += sum_key g[key][d] * P^T[key][q]      A = g tile (LDS), B = P (registers), no LDS round trip.
// The d index of the four O^T tiles is interleaved (tile dt, row i  <->  d = 4*i + dt) so one ds_read_b128
// of g[key][4i..4i+3] feeds four MFMAs.
//
// Round 2: a workgroup is 8 waves = 128 queries x TWO key streams (waves 0-3 take the even 32-key tiles, waves 4-7 the odd ones,
// each with its own running max / sum / O^T, merged through LDS at the end).  With one wave per SIMD the softmax VALU work, the
// LDS publish and the barrier of every tile sat between that wave's two MFMA phases with nothing to fill the matrix pipe; with two
// waves per SIMD on different tiles the partner's MFMAs run meanwhile (fp32 MFMAs starve a co-resident wave's VALU to one issue
// per ~25 cycles, but a tile's ~65 VALU instructions still fit inside the partner's 8 192-cycle MFMA phase).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "igemm_conv.h"
#include "gemm_tail.h"

#ifndef BSR_ATT_PRIO
#define BSR_ATT_PRIO 0      // 0: equal priorities; 1 / 2: key-stream wave group 0 / 1 at s_setprio 2
#endif

namespace bsr {

constexpr int kAttD = 128;        // C/2 of the 257-channel NonLocalBlock (/root/reference/model.py:10-12)
constexpr int kAttKT = 32;        // keys per LDS stage
constexpr int kAttLdK = kAttD + 4;
constexpr float kRescaleThreshold = 8.f;   // log2 units
constexpr int kAttStageFloats = kAttKT * kAttLdK + kAttKT * kAttD;     // one 32-key tile: phi rows (padded) + g rows
constexpr int kAttSmemBytes = 4 * kAttStageFloats * 4;                  // two tile PAIRS (double buffer)
static_assert(kAttSmemBytes <= 160 * 1024, "LDS budget");
static_assert(66 * 64 * 4 <= 4 * kAttStageFloats, "merge scratch fits the staging buffers");

// QW = query waves per workgroup (x 2 key streams): 4 = the 8-wave, 128-query workgroup the B = 32 forward runs (one round of
// B x 8 workgroups on 256 CUs).  Round 4: QW = 2 / 1 are the same kernel with 64 / 32 queries per workgroup for SMALL batches —
// LDS (two staged tile pairs, 133 KB) allows one workgroup per CU whatever its size, so with B x 8 < 256 workgroups most CUs idle;
// halving the query block doubles the grid, and a wave's own work (32 queries x half the keys) is unchanged.
// FUSEW (round 4, QW = 4 only): the NonLocalBlock's `w` 1x1 conv + BN, the block's residual and its LeakyReLU
// (/root/reference/model.py:56-59, 105-113: out = LeakyReLU(y3x + BN(w(att))), the launch that used to follow as gemm_nloop_kernel) run
// as the TAIL of this kernel: a workgroup's 128 queries are 128 pixels of the K = 128 GEMM, their normalised attention output goes
// through LDS into the A-fragment layout (never to HBM), and the two key-stream wave groups take the channel tiles [0,5) / [5,9) of
// N = 288 with the weight images of both streaming through one 3-slot LDS ring — the same MFMA order per output element as
// gemm_nloop_kernel (bit-identical results), one launch, one prologue and 17 MB of HBM round trip less per block.
typedef GemmTailArgs AttWArgs;                               // the fused `w` GEMM: gemm_tail.h, channel tiles [0,5) | [5,9) of N = 288
typedef GemmTailCfg<5, 4> AttWCfg;
constexpr int kAttWSmemBytes = AttWCfg::SMEM_FLOATS * 4;
static_assert(kAttWSmemBytes <= 160 * 1024 && kAttWSmemBytes >= kAttSmemBytes, "LDS budget of the fused tail");
static_assert(4 * 66 * 64 <= kTailAFloats && kTailLdA == kAttLdK, "the merge scratch sits under the attention tile, clear of the weight ring");

template <int QW, bool FUSEW = false>
__global__ __launch_bounds__(QW * 128, QW == 4 ? 2 : 1) void nonlocal_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int tokens, AttWArgs wa) {
  static_assert(!FUSEW || QW == 4, "the fused w tail is the 8-wave shape's");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = QW * 128;                                 // threads
  constexpr int SV = 2048 / NT;                                // float4 per operand and thread that stage one tile pair
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave / QW, wq = wave % QW;                   // key stream (even / odd tiles), query block of 32
  const int h = lane >> 5, r = lane & 31;
  // Workgroup -> (image, query block).  Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2;
  // all query blocks of an image re-read the same K/V (1 MB), so they are given ids that are congruent mod 8 and thus share
  // one XCD's L2 (measured before: 285 MB fetched per launch for 50 MB of qkv, every XCD pulling its own copy of every K/V).
  const int qblocks = tokens / (QW * 32);
  int img, qb;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int per_round = 8 * qblocks;                        // 8 images in flight per round, one per XCD
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
  const int q = qb * (QW * 32) + wq * 32 + r;
#if BSR_ATT_PRIO
  if (grp == (BSR_ATT_PRIO - 1)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);      // see attention_x3.h
#endif

  // theta fragment of this lane's query: element j of group g is channel 8g + 4h + j
  f32x4 qf[kAttD / 8];
#pragma unroll
  for (int g = 0; g < kAttD / 8; ++g)
    qf[g] = *reinterpret_cast<const f32x4*>(base + (size_t)q * (3 * kAttD) + g * 8 + 4 * h) * 1.4426950408889634f;   // log2(e): softmax in base 2

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging of a tile PAIR (2p, 2p+1) by NT threads: 2 x 1024 float4 per operand -> SV + SV per thread (4 + 4 at 512 threads); float4 i
  // of a thread belongs to tile 2p + ((tid + i * NT) >> 10)
  constexpr int V4_PER_TILE = kAttKT * kAttD / 4;       // 1024
  constexpr int NV = V4_PER_TILE / NT;                  // distinct (key, channel) positions per thread: float4 i and i + NV are the same position of the pair's two tiles
  f32x4 kreg[SV], vreg[SV];
  // K / V rows through a raw buffer over this image's qkv: the thread's part of an address is one of NV constant VGPR offsets, the
  // tile is the SGPR offset, phi / g are the instruction's immediate offsets — no 64-bit multiply-add per row inside the key loop (VALU
  // instructions there stand between this wave's MFMAs and take fp32 lanes from its partner's)
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t kv_rsrc = make_rsrc(base);
  unsigned kv_voff[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = (tid + i * NT) & (V4_PER_TILE - 1);
    kv_voff[i] = (unsigned)(((idx / (kAttD / 4)) * (3 * kAttD) + (idx % (kAttD / 4)) * 4) * 4);
  }
  auto fetch = [&](int pr) {
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      const unsigned soff = (unsigned)((2 * pr + (i / NV)) * kAttKT * (3 * kAttD) * 4);      // (tid + i * NT) >> 10 == i / NV for tid < NT
      kreg[i] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(kv_rsrc, kv_voff[i % NV] + (unsigned)(kAttD * 4), soff, 0));
      vreg[i] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(kv_rsrc, kv_voff[i % NV] + (unsigned)(2 * kAttD * 4), soff, 0));
    }
  };
  auto publish = [&](int pbuf) {
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      const int idx = (tid + i * NT) & (V4_PER_TILE - 1), sel = (tid + i * NT) >> 10;
      const int key = idx / (kAttD / 4), c4 = idx % (kAttD / 4);
      float* sk = smem + (2 * pbuf + sel) * kAttStageFloats;
      float* sv = sk + kAttKT * kAttLdK;
      *reinterpret_cast<f32x4*>(sk + key * kAttLdK + c4 * 4) = kreg[i];
      *reinterpret_cast<f32x4*>(sv + key * kAttD + c4 * 4) = vreg[i];
    }
  };

  const int npair = tokens / (2 * kAttKT);
  fetch(0);
  publish(0);
  __syncthreads();

  // two pairs per trip: the staging buffer of a pair (pr & 1) is then a compile-time constant and every LDS address of the loop a
  // constant offset from one per-thread register (tokens % 128 == 0, so the pair count is even)
  auto pair_step = [&](int pr, auto pbuf_const) {
    constexpr int pbuf = decltype(pbuf_const)::value;
    if (pr + 1 < npair) fetch(pr + 1);
    const float* sk = smem + (2 * pbuf + grp) * kAttStageFloats;      // this wave's tile of the pair
    const float* sv = sk + kAttKT * kAttLdK;

    // S^T tile: rows = keys (A from LDS), cols = queries (B from registers)
    f32x16 s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
    for (int g = 0; g < kAttD / 8; ++g) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(sk + r * kAttLdK + g * 8 + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[g][j], s, 0, 0, 0);
    }

    // online softmax for this lane's query: 16 keys here + 16 in lane^32.  Logits are in the log2 domain (theta was
    // pre-scaled by log2 e), so P = exp2(s - m) is one v_exp_f32 per element.  The running maximum is only raised — and O^T, l
    // rescaled — when some query's tile maximum exceeds it by more than kRescaleThreshold (P <= 2^8 stays far inside
    // fp32 range).
    float mx = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__any(mx > m_run + kRescaleThreshold)) {          // wave-uniform branch
      const float m_new = fmaxf(m_run, mx);
      const float scale = __builtin_amdgcn_exp2f(m_run - m_new);           // exp2(-inf) = 0 on the first tile; 1 for lanes whose max did not move
      l_run *= scale;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= scale;
    }
    f32x2 ps2 = {0.f, 0.f};                                  // packed subtraction and row sum: 16 instead of 32 VALU instructions
    const f32x2 m2 = {m_run, m_run};
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const f32x2 d = f32x2{s[i], s[i + 1]} - m2;
      s[i] = __builtin_amdgcn_exp2f(d[0]);
      s[i + 1] = __builtin_amdgcn_exp2f(d[1]);
      ps2 += f32x2{s[i], s[i + 1]};
    }
    l_run += ps2[0] + ps2[1];

    // O^T += g^T . P^T : register i of s holds key (i&3) + 8*(i>>2) + 4h
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = (i & 3) + 8 * (i >> 2) + 4 * h;
      const f32x4 vf = *reinterpret_cast<const f32x4*>(sv + key * kAttD + 4 * r);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[dt], s[i], o[dt], 0, 0, 0);
    }

    if (pr + 1 < npair) {
      publish(pbuf ^ 1);
      __syncthreads();
    }
  };
  for (int pr = 0; pr < npair; pr += 2) {
    pair_step(pr, std::integral_constant<int, 0>{});
    pair_step(pr + 1, std::integral_constant<int, 1>{});
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
    // combine the two key halves' partial sums, normalise
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    inv = 1.f / l_tot;
  }

  if constexpr (!FUSEW) {
    // store y[q][d], d = 4*row + dt
    float* orow = out + ((size_t)img * tokens + q) * kAttD;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      f32x4 v = {o[0][i] * inv, o[1][i] * inv, o[2][i] * inv, o[3][i] * inv};
      *reinterpret_cast<f32x4*>(orow + 4 * row) = v;
    }
  } else {
    // ---- the `w` GEMM tail (gemm_tail.h: gemm_nloop_kernel<3, 4> with the activation tile coming through LDS instead of HBM) ----
    __syncthreads();                                           // every read of the merge scratch is done: the attention tile may overwrite it
    float* s_att = smem;                                       // [128 queries][kTailLdA]
    if (grp == 0) {
      float* arow = s_att + (wq * 32 + r) * kTailLdA;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        f32x4 v = {o[0][i] * inv, o[1][i] * inv, o[2][i] * inv, o[3][i] * inv};      // the values the unfused kernel stores as att
        *reinterpret_cast<f32x4*>(arow + 4 * row) = v;
      }
    }
    const size_t tile_pix = (size_t)img * tokens + (size_t)qb * 128 + (size_t)__builtin_amdgcn_readfirstlane(wq) * 32;
    gemm_tail_run(tail, wa, s_att, s_ring, s_bias, grp, wq, tile_pix, lane);
  }
}

template <int QW>
inline hipError_t launch_nonlocal_attention_qw(const float* qkv, float* out, int batch, int tokens, hipStream_t stream) {
  auto kern = nonlocal_attention_kernel<QW, false>;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kAttSmemBytes);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(batch * (tokens / (QW * 32))), dim3(QW * 128), kAttSmemBytes, stream, qkv, out, tokens, AttWArgs{});
  return hipGetLastError();
}

// The workgroup shape launch_nonlocal_attention picks for a batch (see there).
inline int attention_auto_qw(int batch, int tokens) {
  const long long cus = device_cu_count();
  const long long blocks128 = (long long)batch * (tokens / 128);
  long long best = -1;
  int qw = 4;
  for (int cand : {4, 2, 1}) {                                    // ties go to the larger workgroup
    const long long rounds = (blocks128 * (4 / cand) + cus - 1) / cus;
    const long long cost = rounds * (cand == 4 ? 17 : 10);
    if (best < 0 || cost < best) { best = cost; qw = cand; }
  }
  return qw;
}

// attention + `w` GEMM tail in one launch (the 8-wave shape; callers use it when attention_auto_qw() == 4)
inline hipError_t launch_nonlocal_attention_w(const float* qkv, int batch, int tokens, const AttWArgs& wa, hipStream_t stream) {
  if (tokens % (4 * kAttKT) != 0 || wa.n_pad * 32 < AttWCfg::BIAS_FLOATS * 32 || wa.n_pad < 12 * 32 || wa.n_store > 288 || wa.res_c > 288 || wa.out2 != nullptr) return hipErrorInvalidValue;
  auto kern = nonlocal_attention_kernel<4, true>;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kAttWSmemBytes);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(batch * (tokens / 128)), dim3(512), kAttWSmemBytes, stream, qkv, static_cast<float*>(nullptr), tokens, wa);
  return hipGetLastError();
}

// qw = 0: pick the query block by the grid it gives.  LDS allows ONE workgroup per CU whatever its size, and a wave's work is the same
// in every shape, so a launch costs (rounds of workgroups over the CUs) x (time of one workgroup): measured 139 us for the 8-wave
// shape (two waves share each SIMD), ~80 us for the 4- and 2-wave shapes (one wave per SIMD) — B = 32: 256 x 8 waves, one round;
// B = 16: 256 x 4 waves; B = 10: 160 x 4 waves (not 320 x 2: two rounds); B <= 8: x 2 waves.  Every variant computes the same
// arithmetic in the same order per query, so the choice does not change a single bit of the output.
inline hipError_t launch_nonlocal_attention(const float* qkv, float* out, int batch, int tokens, hipStream_t stream, int qw = 0) {
  if (tokens % (4 * kAttKT) != 0) return hipErrorInvalidValue;      // 128-query blocks; the key loop takes two 64-key pairs per trip
  if (qw == 0) qw = attention_auto_qw(batch, tokens);
  if (qw == 4) return launch_nonlocal_attention_qw<4>(qkv, out, batch, tokens, stream);
  if (qw == 2) return launch_nonlocal_attention_qw<2>(qkv, out, batch, tokens, stream);
  if (qw == 1) return launch_nonlocal_attention_qw<1>(qkv, out, batch, tokens, stream);
  return hipErrorInvalidValue;
}

}  // namespace bsr
