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
struct AttWArgs {
  const float* w;       // packed [4][1][n_pad][36] (pack.py: res{i}.w), n_pad >= 12 * 32
  const float* bias;    // [n_pad]
  int n_pad;
  const float* res;     // y3x: conv3 output + block input, NHWC at the trunk resolution, channels [0, res_c)
  int res_cs, res_c;
  float* out;           // block output, channel stride out_cs, channels [0, n_store) written
  int out_cs, n_store;
  int act;              // 1: LeakyReLU(0.3)
};
constexpr int kAttWSlot = 2 * 96 * 36;                         // floats per ring slot: the (3 tiles x 36-word rows) images of BOTH wave groups
constexpr int kAttWAttFloats = 128 * kAttLdK;                  // normalised attention output of the workgroup, [query][D + 4]
constexpr int kAttWSmemFloats = kAttWAttFloats + 3 * kAttWSlot + 12 * 32;
constexpr int kAttWSmemBytes = kAttWSmemFloats * 4;
static_assert(kAttWSmemBytes <= 160 * 1024 && kAttWSmemBytes >= kAttSmemBytes, "LDS budget of the fused tail");
static_assert(4 * 66 * 64 <= kAttWAttFloats, "the merge scratch sits under the attention tile, clear of the weight ring");

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
  [[maybe_unused]] float* s_ring = smem + kAttWAttFloats;
  [[maybe_unused]] float* s_bias = s_ring + 3 * kAttWSlot;
  constexpr int WPT = 4;                                       // float4 per thread and ring slot: 2 x 864 images over 512 threads (the surplus re-copies)
  [[maybe_unused]] unsigned w_voff[WPT], w_loff[WPT];
  [[maybe_unused]] f32x4 w_regs[WPT], w_regs1[WPT];
  [[maybe_unused]] __amdgpu_buffer_rsrc_t w_rsrc;
  auto fetch_w = [&](int s, f32x4 (&regs)[WPT]) {              // GEMM step s = (channel group, K chunk); both wave groups' images
    const int ng = s >> 2, ch = s & 3;
    const unsigned soff = (unsigned)((ch * wa.n_pad + ng * 96) * 36 * 4);
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      regs[i] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff[i], soff, 0));
  };
  auto store_w = [&](int slot_floats, const f32x4 (&regs)[WPT]) {
    char* dst = reinterpret_cast<char*>(s_ring + slot_floats);
#pragma unroll
    for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(dst + w_loff[i]) = regs[i];
  };
  if constexpr (FUSEW) {
    w_rsrc = make_rsrc(wa.w);
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = (tid + i * NT) % (2 * 864);                // float4 index inside a slot: image of group 0 | image of group 1
      const int sub = e / 864, within = e % 864;
      w_voff[i] = (unsigned)(within * 16 + sub * (5 * 32 * 36 * 4));      // group 1's tiles start 5 tiles (160 channel rows) after group 0's
      w_loff[i] = (unsigned)(e * 16);
    }
    fetch_w(0, w_regs);
    fetch_w(1, w_regs1);
    for (int i = tid; i < 12 * 32; i += NT) s_bias[i] = wa.bias[i];
  }
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
    // ---- the `w` GEMM tail (gemm_nloop_kernel<3, 4> with the activation tile coming from registers instead of HBM) ----
    __syncthreads();                                           // every read of the merge scratch is done: the attention tile may overwrite it
    float* s_att = smem;                                       // [128 queries][kAttLdK]
    if (grp == 0) {
      float* arow = s_att + (wq * 32 + r) * kAttLdK;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        f32x4 v = {o[0][i] * inv, o[1][i] * inv, o[2][i] * inv, o[3][i] * inv};      // the values the unfused kernel stores as att
        *reinterpret_cast<f32x4*>(arow + 4 * row) = v;
      }
    }
    store_w(0, w_regs);
    store_w(kAttWSlot, w_regs1);
    __syncthreads();
    constexpr int NI = 3, NCH = 4, G = 4, LDP = 36, NSTEPS = 2 * NCH;
    f32x4 afr[NCH * G];                                        // this lane's pixel (query wq*32 + r), channels 8g + 4h .. +3
#pragma unroll
    for (int g = 0; g < NCH * G; ++g) afr[g] = *reinterpret_cast<const f32x4*>(s_att + (wq * 32 + r) * kAttLdK + g * 8 + 4 * h);
    const int t0 = grp ? 5 : 0, t1 = grp ? 9 : 5;              // this wave group's channel tiles of N = 288
    int b_base[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b_base[ni] = grp * (96 * LDP) + (ni * 32 + r) * LDP + 4 * h;
    int w_cur = 0, w_n1 = kAttWSlot, w_n2 = 2 * kAttWSlot;
    f32x4 bf[2][NI];
    auto read_frags = [&](int slot, int b_off) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[slot][ni] = *reinterpret_cast<const f32x4*>(s_ring + b_base[ni] + b_off);
    };
    read_frags(0, w_cur);
    const float act_alpha = wa.act ? kLeakyAlpha : 1.f;
    const size_t tile_pix = (size_t)img * tokens + (size_t)qb * 128 + (size_t)__builtin_amdgcn_readfirstlane(wq) * 32;
    const unsigned lane_out = ((unsigned)(4 * h) * (unsigned)wa.out_cs + (unsigned)r) * 4u;
    const unsigned lane_res = (unsigned)(4 * h) * (unsigned)wa.res_cs * 4u;
    const __amdgpu_buffer_rsrc_t rsrc_out = make_rsrc(wa.out + tile_pix * wa.out_cs);
    const __amdgpu_buffer_rsrc_t rsrc_res = make_rsrc(wa.res + tile_pix * wa.res_cs);
    for (int ng = 0; ng < 2; ++ng) {
      const int tg = t0 + ng * NI;
      const int nvalid = min(NI, t1 - tg);                     // 3 | 2 (group 0), 3 | 1 (group 1)
      f32x16 acc[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ni] = bias_tile(h, s_bias[(tg + ni) * 32 + r]);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int s_ = ng * NCH + ch;
        const bool has1 = s_ + 1 < NSTEPS, has2 = s_ + 2 < NSTEPS;
        if (has2) fetch_w(s_ + 2, w_regs);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int cur = g & 1, nxt = cur ^ 1;
          if (g + 1 < G) {
            read_frags(nxt, w_cur + (g + 1) * 8);
          } else if (has1) {
            read_frags(nxt, w_n1);
          }
          if (g == G - 1 && has2) store_w(w_n2, w_regs);
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 a = afr[ch * G + g];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if (ni < nvalid) {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bf[cur][ni][j], acc[ni], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        const int tw = w_cur; w_cur = w_n1; w_n1 = w_n2; w_n2 = tw;
      }
      // epilogue of this channel group: residual (y3x) + LeakyReLU, NHWC store — gemm_nloop_kernel's, one destination
      __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (ni >= nvalid) continue;
        const int nt = (tg + ni) * 32;
        const int n = nt + r;
        f32x16 v = acc[ni];
        if (nt < wa.res_c) {
          const unsigned rcs4 = (unsigned)wa.res_cs * 4u;
          const unsigned l1 = n < wa.res_c ? lane_res + (unsigned)r * 4u : kLaneOff;
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
        const unsigned vb = n < wa.n_store ? lane_out : kLaneOff;
        const unsigned cs4 = (unsigned)wa.out_cs * 4u;
        const unsigned vj[4] = {vb, vb + cs4, vb + 2u * cs4, vb + 3u * cs4};
        unsigned so = (unsigned)nt * 4u;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[4 * q4 + j]), rsrc_out, vj[j], so, 0);
          so += 8u * cs4;
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
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
  if (tokens % (4 * kAttKT) != 0 || wa.n_pad < 12 * 32 || wa.n_store > 288 || wa.res_c > 288) return hipErrorInvalidValue;
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
