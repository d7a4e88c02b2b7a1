// NonLocalBlock attention (/root/reference/model.py:51-53) in split precision on the fp16 matrix cores, ONE WAVE PER SIMD (round 6).
//
// Same arithmetic as the kernel it replaces (round 2-5: nonlocal_attention_x3_kernel, 8 waves = two key streams, register-staged
// tiles split by every workgroup; its counters said: matrix pipe 34 % busy, a wave 45 % issue-stalled + 30 % parked): S^T = phi.theta^T
// per 32-key tile with the query on the lane, base-2 online softmax in registers, O^T += g^T.P^T with exp2(S^T) as the B operand, every
// product as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16.  What changed is where the cycles went:
//
//  * the OPERAND SPLIT LEFT THE KERNEL.  theta|phi|g arrive already split: the conv3|theta|phi|g GEMM (gemm_nloop.h, `out2_split`)
//    writes qkv as per token 3 x [128 hi halves | 128 lo halves] (the same 1 536 bytes as 384 floats; theta pre-scaled by log2 e), so
//    the hi / lo split of a key row is done ONCE by its producer instead of once per query block (8 x per image) beside the matrix
//    stream, and the range guard of the 16-bit modes moves there with it;
//  * tiles come by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, no VALU) into a 4-slot ring, requested three
//    tiles ahead; rows are unpadded 512-byte [hi | lo] images, made conflict-free by XOR-swizzling the 16-byte chunk index through the
//    per-lane SOURCE address (phi: chunk ^ (key & 15) for ds_read_b128 of 16 different keys; g: chunk ^ ((key & 3) << 2) for
//    ds_read_b64_tr_b16 blocks of 4 keys x 64 bytes);
//  * a workgroup is 4 waves (128 queries, one key stream: no merge), each alone on its SIMD with the whole register file, and the
//    loop is software-pipelined by hand: while the matrix pipe runs S^T of tile t + 1 the vector pipe exponentiates / sums / splits
//    P of tile t; while it runs O^T of tile t the vector pipe takes the row maximum of tile t + 1.  The rare rescale (running maximum
//    raised by more than 2^8) is a branch BETWEEN iterations, so an iteration is one basic block the scheduler may interleave;
//  * FUSEW: the `w` conv + BN + block residual + LeakyReLU (model.py:56-59,105-113) is the tail, and the normalised O^T accumulators ARE its
//    A operand (lane = pixel; registers 8p .. 8p+7 of channel block dt = the 8 k-slots of K step (dt, p)): no LDS round trip.  The
//    nine 16-KB weight tiles (pack.py `w4`: rows [n][128 hi | 128 lo] with k in that register order, chunk-swizzled like phi) are
//    requested by the last four iterations' DMA slots and sit in LDS complete when the loop ends; the residual of tile n + 1 is in
//    flight while tile n multiplies.
//
// Operand maps of v_mfma_f32_32x32x16_f16: lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j], B[k = 8h + j][col r], j = 0..7;
// C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4h.
#pragma once
#include <hip/hip_runtime.h>
#include "attention.h"
#include "igemm_h16.h"

// Diagnostic builds (scratch/att4_diag.py; never set in the product): 1 = every tile request re-reads tile t & 3 (L2-hot source),
// 2 = no tile requests inside the loop, 21 = no residual loads in the tail, 22 = no output stores (wrong results: timing only)
#ifndef A4_DIAG
#define A4_DIAG 0
#endif



// -DA4_STAMPS (attention-only launches of scratch/att4_diag.py): every wave keeps the low word of s_memtime at six points of every iteration in
// lane t of six registers (v_writelane: no branch, no memory traffic inside the loop — a conditional store there splits the iteration into
// blocks and the register allocator fills them with copies); wave 0 of workgroup 0 stores them BEHIND `out` when the loop is done
#ifdef A4_STAMPS
#define A4_STAMP(k) do { const int a4_now = (int)__builtin_amdgcn_s_memtime(); const int a4_t = __builtin_amdgcn_readfirstlane(t); \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(a4_stv[k]) : "s"(a4_now), "s"(a4_t) : "m0"); } while (0)
#else
#define A4_STAMP(k) do { } while (0)
#endif

namespace bsr {

constexpr int kA4TokBytes = 3 * kAttD * 4;          // one token of the split qkv buffer: theta | phi | g, each [128 hi | 128 lo] halves
constexpr int kA4RowBytes = 512;                    // one operand row in LDS: 256 B hi plane + 256 B lo plane
constexpr int kA4KT = 32;                           // keys per tile
constexpr int kA4KBytes = kA4KT * kA4RowBytes;      // 16 KB: the phi rows of a tile (the g rows follow)
constexpr int kA4SlotBytes = 2 * kA4KBytes;         // 32 KB
constexpr int kA4Slots = 4;
constexpr int kA4WTiles = 9;                        // 32-channel tiles of the `w` GEMM (N = 288), 16 KB each
constexpr int kA4SmemBytes = kA4WTiles * kA4KBytes; // 144 KB >= the 128-KB ring
static_assert(kA4SmemBytes >= kA4Slots * kA4SlotBytes && kA4SmemBytes <= 160 * 1024, "LDS budget");

// one LDS-DMA piece: 64 lanes x 16 bytes -> 1 KiB at lds_addr (asm: behind the builtin hipcc 7.2 drains lgkmcnt before every later
// matrix instruction, igemm_h16.h); counted in vmcnt like a load, waited for with the explicit counted waits below
__device__ __forceinline__ void a4_dma(const char* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

typedef short a4_s16x4 __attribute__((__vector_size__(8)));
typedef short a4_s16x8 __attribute__((__vector_size__(16)));
__device__ __forceinline__ f16x8 a4_read_tr(const char* p) {      // keys {0..3} and {8..11} relative to the addressed row
  const a4_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) a4_s16x4*)(p));
  const a4_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) a4_s16x4*)(p + 8 * kA4RowBytes));
  return __builtin_bit_cast(f16x8, (a4_s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// PV1 = 1 (f16 mode only, opt-in): P.V with the hi planes only — one matrix instruction per product instead of three
template <bool FUSEW, int PV1 = 0>
__global__ __launch_bounds__(256, 1) void nonlocal_attention_h16_kernel(const char* __restrict__ qkvs, float* __restrict__ out, int tokens, AttWArgs wa) {
  extern __shared__ __attribute__((aligned(1024))) char a4_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int qblocks = tokens / 128;
  int img, qb;
  {   // the query blocks of one image share an XCD's L2 copy of its keys / values (attention.h)
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
  const char* base = qkvs + (size_t)img * tokens * kA4TokBytes;
  const int q = qb * 128 + wave * 32 + r;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)a4_smem;

  // ---- LDS-DMA plan: a tile is 16 phi pieces + 16 g pieces of 1 KiB (two 512-byte rows each); wave w moves pieces w, w + 4, w + 8, w + 12
  // of both.  Lane l of piece p lands at chunk position l & 31 of key 2p + (l >> 5) and fetches the chunk that belongs there.
  unsigned dk[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = 2 * (wave + 4 * i) + (lane >> 5), cpos = lane & 31;
    dk[i] = (unsigned)(key * kA4TokBytes + 512 + ((cpos ^ (key & 15)) << 4));
    dv[i] = (unsigned)(key * kA4TokBytes + 1024 + ((cpos ^ ((key & 3) << 2)) << 4));
  }
  auto dma_tile = [&](int kt, int slot) {
    const char* tb = base + (size_t)kt * (kA4KT * kA4TokBytes);
#pragma unroll
    for (int i = 0; i < 4; ++i) a4_dma(tb, dk[i], lds0 + slot * kA4SlotBytes + (wave + 4 * i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) a4_dma(tb, dv[i], lds0 + slot * kA4SlotBytes + kA4KBytes + (wave + 4 * i) * 1024);
  };
  // FUSEW: weight tiles 2j, 2j + 1 of `w` (32 KB, stored as the LDS image) take the place of tile NT + j: the same 8 pieces per wave
  const unsigned dw = (unsigned)(lane * 16);
  auto dma_wpair = [&](int j, int slot) {
    const char* tb = reinterpret_cast<const char*>(wa.w) + (size_t)j * kA4SlotBytes;
#pragma unroll
    for (int i = 0; i < 8; ++i) a4_dma(tb + (wave + 4 * i) * 1024, dw, lds0 + slot * kA4SlotBytes + (wave + 4 * i) * 1024);
  };

  const int NT = tokens / kA4KT;
  if constexpr (FUSEW) {      // the ninth weight tile lives above the ring: requested first, landed long before anyone looks
#pragma unroll
    for (int i = 0; i < 4; ++i)
      a4_dma(reinterpret_cast<const char*>(wa.w) + 8 * kA4KBytes + (wave + 4 * i) * 1024, dw, lds0 + 8 * kA4KBytes + (wave + 4 * i) * 1024);
  }
  dma_tile(0, 0);
  dma_tile(1, 1);

  // theta of this lane's query (pre-scaled by log2 e and split by the producer): K step s covers channels 16 s + 8 h .. +7
  f16x8 qh[kAttD / 16], ql[kAttD / 16];
  {
    const char* qrow = base + (size_t)q * kA4TokBytes + 16 * h;
#pragma unroll
    for (int s = 0; s < kAttD / 16; ++s) {
      qh[s] = *reinterpret_cast<const f16x8*>(qrow + 32 * s);
      ql[s] = *reinterpret_cast<const f16x8*>(qrow + 256 + 32 * s);
    }
  }

  // fragment addresses (bytes from the slot's phi / g base): the chunk swizzle makes them lane-dependent per K step / channel block
  unsigned kaddr[8], vaddr[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kaddr[ks] = (unsigned)(r * kA4RowBytes + (((2 * ks + h) ^ (r & 15)) << 4));
  {
    const int kk = (lane & 15) >> 2, gq1 = (lane >> 4) & 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      vaddr[dt] = (unsigned)((4 * h + kk) * kA4RowBytes + (((4 * dt + 2 * gq1 + ((lane & 3) >> 1)) ^ (kk << 2)) << 4) + 8 * (lane & 1));
  }

#ifdef A4_STAMPS
  int a4_stv[6] = {0, 0, 0, 0, 0, 0};
#endif
  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float l_run = 0.f;

  // Fragments live in registers one phase ahead of their matrix instructions: the phi fragments of tile t + 1 are read while O^T of tile
  // t - 1 multiplies, the transposed g fragments of tile t while S^T of tile t + 1 multiplies — no matrix instruction waits for LDS.
  f16x8 kfh[8], kfl[8], vfh[8], vfl[PV1 ? 1 : 8];
  auto read_k = [&](const char* kb, int ks) {
    kfh[ks] = *reinterpret_cast<const f16x8*>(kb + kaddr[ks]);
    kfl[ks] = *reinterpret_cast<const f16x8*>(kb + kaddr[ks] + 256);
  };
  auto read_v = [&](const char* vb, int g) {               // fragment g = (dt, t2): channels 32 dt .., keys 16 t2 ..
    vfh[g] = a4_read_tr(vb + vaddr[g >> 1] + (g & 1) * 16 * kA4RowBytes);
    if constexpr (PV1 == 0) vfl[g] = a4_read_tr(vb + vaddr[g >> 1] + (g & 1) * 16 * kA4RowBytes + 256);
  };
  auto s_step = [&](f32x16& s, int ks) {
    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[ks], qh[ks], s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[ks], ql[ks], s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[ks], qh[ks], s, 0, 0, 0);
  };
  auto xhalf_max = [&](float mx) -> float {                // the other 16 keys of this query sit on lane ^ 32
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return __builtin_fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  };

  // tiles 0 and 1 and theta have landed (everything this wave has requested so far), for every wave.  The empty statement makes the
  // compiler place ITS wait for the theta loads here — it cannot see the DMAs, so the wait it computes is vmcnt(0), and behind the
  // requests for tiles 2 and 3 that would wait for those too.
  __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::"v"(qh[0]), "v"(ql[7]));
  dma_tile(2, 2);
  dma_tile(3, 3);
  f32x16 sa, sb;
  float m_run;
  {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) read_k(a4_smem, ks);
#pragma unroll
    for (int i = 0; i < 16; ++i) sa[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) s_step(sa, ks);
    float mx = __builtin_fmaxf(sa[0], sa[1]);
#pragma unroll
    for (int i = 2; i < 16; i += 2) mx = __builtin_fmaxf(__builtin_fmaxf(sa[i], sa[i + 1]), mx);
    m_run = xhalf_max(mx);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) read_k(a4_smem + kA4SlotBytes, ks);
  }

  // FUSEW: the tail's residual (y3x, 147 KB per workgroup) and its output (the same again) are what the tail waits for — every workgroup of
  // the one-round grid is in its tail at once: 72 MB against HBM / the Infinity Cache, ~5 us for the loads and ~6 for the stores of a tail
  // whose matrix work is 4.5 us (A/B builds without them: profiles/HISTORY.md round 6).  The residual of channel tiles 0-3 is requested
  // during the LAST FOUR iterations of the key loop (one tile each: behind them only weight pieces wait, which nobody needs before the tail),
  // the rest when the loop is done — all of it BEFORE the first store: vmcnt retires in order and counts stores, so a load requested behind
  // a tile's stores cannot be waited for without waiting for those stores' acknowledgements.  (Requested over the last eight iterations
  // instead — six tiles in the loop, three in the last iteration: 63.6 against 63.9 us per launch; not worth eight peeled iterations.)
  constexpr int kResInLoop = FUSEW ? 4 : 0;
  const size_t tile_pix = (size_t)img * tokens + (size_t)qb * 128 + (size_t)wave * 32;
  const __amdgpu_buffer_rsrc_t rsrc_res = make_rsrc(FUSEW ? wa.res + tile_pix * wa.res_cs : reinterpret_cast<const float*>(qkvs));
  const unsigned rcs4 = (unsigned)wa.res_cs * 4u;
  // y3x rows (i & 3) + 8 (i >> 2) + 4h of the wave's 32 pixels, channel 32 n + r — one element of the residual tile of channel tile n
  auto load_res1 = [&](int n, int i) -> float {
    const unsigned l1 = 32 * n + r < wa.res_c ? (unsigned)(4 * h) * rcs4 + (unsigned)r * 4u : kLaneOff;
    if (A4_DIAG == 21) return 0.f;
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_res, l1 + (unsigned)(i & 3) * rcs4, (unsigned)(32 * n) * 4u + (unsigned)(i >> 2) * 8u * rcs4, 0));
  };
  [[maybe_unused]] f32x16 res[kA4WTiles];      // MFMA accumulator layout: the residual tile is the C operand of its tile's first matrix instruction

  // One iteration = 48 matrix instructions, and one wave per SIMD hides only what is issued BETWEEN its own matrix instructions: a gap takes
  // max(32, issue cycles of what stands in it) — 8 for the matrix instruction itself, 4 per vector / LDS instruction, 8 per exp — so the other
  // work of a phase is dealt over its 24 gaps by hand, at most ~24 cycles each, and a sched_barrier closes every gap (measured with the work
  // in 8 groups of 3 instead: 3 100 cycles per iteration, every piece serial to the matrix stream; profiles/HISTORY.md round 6).
  //   phase A  matrix: S^T(t + 1) from the phi fragments in registers (K step ks = 3 gaps)
  //            vector: P(t) = exp2(S(t) - m), row sum, hi / lo split: one PAIR of values per K step (sub sub exp | exp add add | cvt mix mix cvt)
  //            LDS: one transposed g(t) fragment per gap (gaps 0-15)
  //   -- tile t + 2 has landed for everyone; everyone holds g(t) in registers: slot t & 3 is free --
  //   phase B  matrix: O^T += g(t)^T P(t)^T as (key half, plane pair, channel block) with the channel block fastest
  //            vector: row maximum of S(t + 1), one max3 per two gaps        LDS: one phi fragment of tile t + 2 per gap (gaps 0-15)
  //            DMA: the 8 pieces of tile t + 4 into slot t & 3, gaps 16-23
  // MORE: a tile t + 4 exists (all but the last four iterations): the wait counts and the DMA source are compile-time, an iteration is ONE basic block
  auto iteration = [&](auto more_tag, int t, f32x16& s_cur, f32x16& s_nxt, const char* slot_cur, const char* slot_n2) {
    constexpr int POS = decltype(more_tag)::value;            // -1: steady state; 0..3: the last four iterations, t = NT - 4 + POS
    constexpr bool MORE = POS < 0;                            // a tile t + 4 exists
    constexpr int RES = POS < kResInLoop ? POS : -1;           // residual tile requested in this iteration's phase A
    f16x2 p2h[8], p2l[8];
    A4_STAMP(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) s_nxt[i] = 0.f;
    float d1 = 0.f, e0 = 0.f, e1 = 0.f;
#pragma unroll
    for (int a = 0; a < 24; ++a) {
      const int ks = a / 3, j = a % 3;
      if (j == 0) s_nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[ks], qh[ks], s_nxt, 0, 0, 0);
      else if (j == 1) s_nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[ks], ql[ks], s_nxt, 0, 0, 0);
      else s_nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[ks], qh[ks], s_nxt, 0, 0, 0);
      if (a < (PV1 ? 8 : 16)) {                              // fragment f = (dt, t2): channels 32 dt .., keys 16 t2 ..; hi plane, then lo plane
        const int f = PV1 ? a : a >> 1;
        const char* vp = slot_cur + kA4KBytes + vaddr[f >> 1] + (f & 1) * 16 * kA4RowBytes;
        if (PV1 || (a & 1) == 0) vfh[f] = a4_read_tr(vp); else vfl[PV1 ? 0 : f] = a4_read_tr(vp + 256);
      }
      if constexpr (RES >= 0) { if (a < 16) res[RES][a] = load_res1(RES, a); }
      if (j == 0) {
        const float d0 = s_cur[2 * ks] - m_run;
        d1 = s_cur[2 * ks + 1] - m_run;
        e0 = __builtin_amdgcn_exp2f(d0);                     // <= 2^8: inside the fp16 range
      } else if (j == 1) {
        e1 = __builtin_amdgcn_exp2f(d1);
        l_run += e0;         // one chain on purpose: as an (e0, e1) pair hipcc packs the sums into v_pk_add_f32 behind 2 moves each
        l_run += e1;
      } else {
        // hi = fp16(e), lo = fp16(e - hi) as mfma_common.h split2, with the subtraction as ONE mixed-precision fma per value (v_fma_mix_f32)
        const f16x2 hi = __builtin_convertvector(f32x2{e0, e1}, f16x2);
        const float l0 = __builtin_fmaf((float)hi[0], -1.f, e0), l1 = __builtin_fmaf((float)hi[1], -1.f, e1);
        p2h[ks] = hi;
        p2l[ks] = __builtin_convertvector(f32x2{l0, l1}, f16x2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    f16x8 ph[2], pl[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ph[t2][2 * j] = p2h[4 * t2 + j][0]; ph[t2][2 * j + 1] = p2h[4 * t2 + j][1];
        pl[t2][2 * j] = p2l[4 * t2 + j][0]; pl[t2][2 * j + 1] = p2l[4 * t2 + j][1];
      }
    A4_STAMP(1);
    constexpr int kYounger = (MORE || FUSEW) ? 8 + (RES >= 0 ? 16 : 0) + (RES >= 1 ? 16 : 0) : 0;      // requested after tile t + 2's pieces: the pieces of t + 3 and the residual tiles of this and the last iteration
    __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(A4_DIAG == 2 ? 0 : kYounger));      // tile t + 2 is in LDS (the 8 pieces of t + 3 may still fly); without the
    __builtin_amdgcn_s_barrier();                                                                   // tail nothing follows the last tile: drain (attention-only launches: tests)
    __builtin_amdgcn_sched_barrier(0);
    A4_STAMP(2);
    const char* tb = MORE ? base + (size_t)(A4_DIAG == 1 ? (t & 3) : t + 4) * (kA4KT * kA4TokBytes) : reinterpret_cast<const char*>(wa.w) + (size_t)(FUSEW ? t + 4 - NT : 0) * kA4SlotBytes;
    const unsigned slot_lds = lds0 + (unsigned)(t & 3) * kA4SlotBytes;
    float mx = 0.f;
#pragma unroll
    for (int b = 0; b < 24; ++b) {
      // consecutive matrix instructions go to different accumulator tiles; each tile still sums (lo.hi, hi.lo, hi.hi) of keys 0-15, then of keys 16-31
      constexpr int kPer = PV1 ? 1 : 3;
      if (b == 16) A4_STAMP(4);
      if (b == 8) A4_STAMP(5);
      if (b < 8 * kPer) {
        const int dt = b & 3, j = PV1 ? 2 : (b >> 2) % 3, t2 = (b >> 2) / kPer, f = 2 * dt + t2;
        if (j == 0) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl[PV1 ? 0 : f], ph[t2], o[dt], 0, 0, 0);
        else if (j == 1) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[f], pl[t2], o[dt], 0, 0, 0);
        else o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[f], ph[t2], o[dt], 0, 0, 0);
      }
      if (b < 16) {
        if ((b & 1) == 0) kfh[b >> 1] = *reinterpret_cast<const f16x8*>(slot_n2 + kaddr[b >> 1]);
        else kfl[b >> 1] = *reinterpret_cast<const f16x8*>(slot_n2 + kaddr[b >> 1] + 256);
        if ((b & 1) == 0) mx = b == 0 ? __builtin_fmaxf(s_nxt[0], s_nxt[1]) : __builtin_fmaxf(__builtin_fmaxf(s_nxt[b], s_nxt[b + 1]), mx);
      } else {
        // piece b - 16 of the next occupant of slot t & 3: phi pieces w + 4i, then g pieces; or (FUSEW, past the last tile) a weight-tile pair as it lies
        const int g = b - 16;
        if constexpr (MORE && A4_DIAG != 2) a4_dma(tb, g < 4 ? dk[g & 3] : dv[g & 3], slot_lds + (g < 4 ? 0 : kA4KBytes) + (wave + 4 * (g & 3)) * 1024);
        else if constexpr (FUSEW) a4_dma(tb + (wave + 4 * g) * 1024, dw, slot_lds + (wave + 4 * g) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    A4_STAMP(3);
    mx = xhalf_max(mx);
    // ---- the running maximum moves only when a tile exceeds it by 2^8 (attention.h), between iterations
    if ((MORE || t + 1 < NT) && __any(mx > m_run + kRescaleThreshold)) {
      const float m_new = __builtin_fmaxf(m_run, mx);
      const float scale = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= scale;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= scale;
    }
  };

  {
    std::integral_constant<int, -1> steady;
    int t = 0;
#pragma unroll 1
    for (; t < NT - 4; t += 4) {
      iteration(steady, t, sa, sb, a4_smem, a4_smem + 2 * kA4SlotBytes);
      iteration(steady, t + 1, sb, sa, a4_smem + kA4SlotBytes, a4_smem + 3 * kA4SlotBytes);
      iteration(steady, t + 2, sa, sb, a4_smem + 2 * kA4SlotBytes, a4_smem);
      iteration(steady, t + 3, sb, sa, a4_smem + 3 * kA4SlotBytes, a4_smem + kA4SlotBytes);
    }
    iteration(std::integral_constant<int, 0>{}, t, sa, sb, a4_smem, a4_smem + 2 * kA4SlotBytes);
    iteration(std::integral_constant<int, 1>{}, t + 1, sb, sa, a4_smem + kA4SlotBytes, a4_smem + 3 * kA4SlotBytes);
    iteration(std::integral_constant<int, 2>{}, t + 2, sa, sb, a4_smem + 2 * kA4SlotBytes, a4_smem);
    iteration(std::integral_constant<int, 3>{}, t + 3, sb, sa, a4_smem + 3 * kA4SlotBytes, a4_smem + kA4SlotBytes);
  }

#ifdef A4_STAMPS
  if (!FUSEW && blockIdx.x == 0 && wave == 0 && lane < 32)
    for (int k = 0; k < 6; ++k) reinterpret_cast<int*>(out + (size_t)gridDim.x * 128 * kAttD)[k * 32 + lane] = a4_stv[k];
#endif
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;

  if constexpr (!FUSEW) {
    // y[q][d], d = 32 dt + (i & 3) + 8 (i >> 2) + 4h: four consecutive channels per register quad
    float* orow = out + ((size_t)img * tokens + q) * kAttD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 v = {o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv, o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g4 + 4 * h) = v;
      }
  } else {
    // ---- the `w` GEMM: out[px][n] = LeakyReLU(sum_k att[px][k] W[k][n] + b[n] + y3x[px][n]); this wave's 32 pixels x all 9 channel tiles
    const __amdgpu_buffer_rsrc_t rsrc_out = make_rsrc(wa.out + tile_pix * wa.out_cs);
    const unsigned ocs4 = (unsigned)wa.out_cs * 4u;
    const float act_alpha = wa.act ? kLeakyAlpha : 1.f;
    // the rest of the residual (see kResInLoop above).  A residual tile is loaded in the accumulator layout and enters as the C operand of its
    // tile's first matrix instruction: no vector add, and the tiles wait in the accumulator half of the register file
#pragma unroll
    for (int n = kResInLoop; n < kA4WTiles; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) res[n][i] = load_res1(n, i);
    // A fragments: K step (dt, p) = registers 8p .. 8p + 7 of o[dt] (channels 32 dt + 16 p + 4h + (j & 3) + 8 (j >> 2)): the order pack.py's `w4` image uses
    f16x8 ahi[8], alo[8];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const f32x4 x0 = {o[dt][8 * p] * inv, o[dt][8 * p + 1] * inv, o[dt][8 * p + 2] * inv, o[dt][8 * p + 3] * inv};
        const f32x4 x1 = {o[dt][8 * p + 4] * inv, o[dt][8 * p + 5] * inv, o[dt][8 * p + 6] * inv, o[dt][8 * p + 7] * inv};
        split8(x0, x1, ahi[2 * dt + p], alo[2 * dt + p]);
      }
    float bias[kA4WTiles];
#pragma unroll
    for (int n = 0; n < kA4WTiles; ++n) bias[n] = wa.bias[32 * n + r];
    // every weight piece has landed (this wave's own: vmcnt 0 — the residual and bias loads above are younger, so this waits for them too;
    // the other waves': the barrier)
    __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
    __builtin_amdgcn_s_barrier();
    // The same gap discipline as the key loop: tile n multiplies (24 matrix instructions) while the weight fragments of tile n + 1 are read
    // (gaps 0-15), the residual of tile n + 1 is requested (gaps 0-15) and tile n - 1 gets its epilogue — residual add (gaps 0-7), LeakyReLU
    // (gaps 8-15), stores (gaps 16-23).  n = 9 is the drain: the epilogue of tile 8 alone.
    f16x8 wfh[2][8], wfl[2][8];
    auto read_w = [&](int n, int ks, int plane) {
      const char* wp = a4_smem + n * kA4KBytes + kaddr[ks] + plane * 256;
      if (plane == 0) wfh[n & 1][ks] = *reinterpret_cast<const f16x8*>(wp); else wfl[n & 1][ks] = *reinterpret_cast<const f16x8*>(wp);
    };
#pragma unroll
    for (int a = 0; a < 16; ++a) read_w(0, a >> 1, a & 1);
    f32x16 acc[2];
    float v[16];
#pragma unroll
    for (int n = 0; n <= kA4WTiles; ++n) {
      const int cur = n & 1, prv = cur ^ 1;
#pragma unroll
      for (int a = 0; a < 24; ++a) {
        const int ks = a / 3, j = a % 3;
        if (n < kA4WTiles) {
          if (a == 0) acc[cur] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[0], wfh[cur][0], res[n], 0, 0, 0);
          else if (j == 0) acc[cur] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ks], wfh[cur][ks], acc[cur], 0, 0, 0);
          else if (j == 1) acc[cur] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], wfl[cur][ks], acc[cur], 0, 0, 0);
          else acc[cur] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], wfh[cur][ks], acc[cur], 0, 0, 0);
        }
        if (n + 1 < kA4WTiles && a < 16) {
          read_w(n + 1, a >> 1, a & 1);
        }
        if (n >= 1) {
          const int m = n - 1;                                  // the tile whose epilogue runs in these gaps
          if (a < 8) {
            v[2 * a] = acc[prv][2 * a] + bias[m];
            v[2 * a + 1] = acc[prv][2 * a + 1] + bias[m];
          } else if (a < 16) {
            const int g = a - 8;
            const f32x2 t2 = f32x2{v[2 * g], v[2 * g + 1]} * act_alpha;
            v[2 * g] = __builtin_amdgcn_fmed3f(v[2 * g], t2[0], 3.4028234664e38f);
            v[2 * g + 1] = __builtin_amdgcn_fmed3f(v[2 * g + 1], t2[1], 3.4028234664e38f);
          } else {
            const unsigned vb = 32 * m + r < wa.n_store ? (unsigned)(4 * h) * ocs4 + (unsigned)r * 4u : kLaneOff;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int i = 2 * (a - 16) + e;
              if (A4_DIAG != 22 || v[i] == 12345.678f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i]), rsrc_out, vb + (unsigned)(i & 3) * ocs4, (unsigned)(32 * m) * 4u + (unsigned)(i >> 2) * 8u * ocs4, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// fp32 [tokens][384] -> the split layout (bsr_debug_attention only: in the forward the conv3|theta|phi|g GEMM writes it directly)
__global__ void a4_split_qkv_kernel(const float* __restrict__ qkv, char* __restrict__ dst, size_t n_pairs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one PAIR of channels
  if (i >= n_pairs) return;
  const size_t tok = i / 192;
  const int c2 = (int)(i % 192), grp = c2 / 64, c = (c2 % 64) * 2;
  f32x2 x = *reinterpret_cast<const f32x2*>(qkv + tok * 384 + grp * 128 + c);
  if (grp == 0) x = x * 1.4426950408889634f;
  f16x2 hi, lo;
  split2(x, hi, lo);
  char* row = dst + tok * kA4TokBytes + grp * 512 + c * 2;
  *reinterpret_cast<f16x2*>(row) = hi;
  *reinterpret_cast<f16x2*>(row + 256) = lo;
}

template <bool FUSEW, int PV1>
inline hipError_t a4_launch(const char* qkvs, float* out, int batch, int tokens, const AttWArgs& wa, hipStream_t stream) {
  if (tokens % 128 != 0) return hipErrorInvalidValue;
  auto kern = nonlocal_attention_h16_kernel<FUSEW, PV1>;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kA4SmemBytes);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(batch * (tokens / 128)), dim3(256), kA4SmemBytes, stream, qkvs, out, tokens, wa);
  return hipGetLastError();
}

// attention alone (probes, tests): split qkv in, fp32 [tokens][128] out
inline hipError_t launch_nonlocal_attention_h16(const float* qkv_split, float* out, int batch, int tokens, hipStream_t stream, bool pv1 = false) {
  if (pv1) return a4_launch<false, 1>(reinterpret_cast<const char*>(qkv_split), out, batch, tokens, AttWArgs{}, stream);
  return a4_launch<false, 0>(reinterpret_cast<const char*>(qkv_split), out, batch, tokens, AttWArgs{}, stream);
}
// attention + `w` tail in one launch; wa.w = the layer's `w4` image (pack.py), wa.bias its bias
inline hipError_t launch_nonlocal_attention_h16_w(const float* qkv_split, int batch, int tokens, const AttWArgs& wa, hipStream_t stream, bool pv1 = false) {
  if (wa.n_store > 288 || wa.res_c > 288 || wa.out2 != nullptr || wa.res == nullptr) return hipErrorInvalidValue;
  if (pv1) return a4_launch<true, 1>(reinterpret_cast<const char*>(qkv_split), nullptr, batch, tokens, wa, stream);
  return a4_launch<true, 0>(reinterpret_cast<const char*>(qkv_split), nullptr, batch, tokens, wa, stream);
}

}  // namespace bsr
