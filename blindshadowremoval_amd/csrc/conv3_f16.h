// 3x3 convolutions of the f16 mode (BASELINE configs[3]: single-fp16 operands, fp32 accumulate) — Conv2D 3x3 stride 1 and
// Conv2DTranspose 3x3 stride 2 (/root/reference/model.py:85,153,207-216) — round 6.
//
// The implicit-GEMM kernel these layers ran on (igemm_h16_kernel<.., NSPLIT = 1>) steps one TAP at a time: with single-fp16 operands a
// step is four matrix instructions (128 cycles of pipe) behind a weight-ring rotation in scalar registers, two LDS round trips that nothing
// covers, a counted wait and a workgroup barrier — in-kernel stamps (scratch/bench_h16.hip): 174 cycles per matrix instruction in the loop,
// 25 k of a workgroup's 39 k cycles on up3 (the split-precision form of the same loop, three times the matrix work per step: 90).  Here
//   * a step is a TRIO of taps (12 matrix instructions) with ONE barrier; a chunk of 32 input channels is three trios, and the weight ring
//     is exactly one chunk (three trio slots of 12 KB): the slot of a trio is a compile-time constant — no ring pointers, no rotation;
//   * the weights of a trio are ONE contiguous 12-KB image per 64-channel output block (pack.py `w3`: [block][chunk][tap][64 rows x 64 B], the
//     16-byte chunk index XOR-swizzled by (row >> 2) & 3 so that ds_read_b128 over 16 rows is conflict-free without padding) moved by
//     three LDS-DMA pieces per wave, requested two trios before their barrier;
//   * the fragments of tap t + 1 are read while tap t multiplies (two register sets), within a trio;
//   * the input tile of the NEXT chunk is requested at the start of a chunk through a raw buffer (out-of-image pixels are out-of-range
//     offsets: the hardware returns the zeros of TF's SAME padding), converted if it is fp32, and written to the other of two LDS tile
//     buffers during trio 1 — the loads are inline asm so that the only waits are the counted ones written here (the compiler cannot see
//     the DMAs and would drain them at every use of a loaded register).
// Same tile geometry and accumulator layout as igemm_h16.h (4 x 32 pixels per workgroup, wave = tile row, 64 output channels per block, the four
// output parities of the transposed conv as four accumulator sets), same epilogue; two workgroups per CU.
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_h16.h"

// Diagnostic builds (timing only, wrong results): 1 = no counted waits on the weight stream, 2 = no trio barriers, 3 = no weight DMA in the loop,
// 4 = no tile-chunk fetch / store in the loop
#ifndef C3_DIAG
#define C3_DIAG 0
#endif

namespace bsr {

template <bool TR>
struct C3Cfg {
  static constexpr int TH = 4, TW = 32, BN = 64, NI = 2, NPH = TR ? 4 : 1;
  static constexpr int IH = TR ? TH + 1 : TH + 2, IW = TR ? TW + 1 : TW + 2;
  static constexpr int PIX_B = 80;                                   // bytes per input pixel in LDS: 32 halves + 8 halves of pad (conflict-free b128 over 16 pixels)
  static constexpr int IN_BYTES = ((IH * IW * PIX_B + 1023) / 1024) * 1024;
  static constexpr int IN_PIECES = IH * IW * 4;                      // 16-byte pieces (8 channels) of one tile chunk
  static constexpr int IN_PER_THREAD = (IN_PIECES + 255) / 256;
  static constexpr int TAP_B = BN * 64, TRIO_B = 3 * TAP_B;
  static constexpr int W_OFF = 2 * IN_BYTES;
  static constexpr int SMEM_BYTES = W_OFF + 3 * TRIO_B;
  static_assert(SMEM_BYTES <= 80 * 1024, "two workgroups per CU");
  static_assert(3 * TRIO_B < 65536 && W_OFF < 65536, "every fragment address is a base register (input tile origin / weight ring origin) plus a 16-bit immediate");
};

typedef int c3_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ c3_v4i c3_rsrc(const void* base) {          // raw buffer, stride 0, 2-GiB window (mfma_common.h make_rsrc, as four plain words)
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  c3_v4i r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
  r[2] = 0x7fffffff;
  r[3] = 0x00020000;
  return r;
}

// IO: bit 0 = the input tensor is fp16 in HBM, bit 1 = the output is written as fp16 (igemm_h16.h)
template <bool TR, int IO>
__global__ __launch_bounds__(256, 2) void conv3_f16_kernel(ConvArgs p) {
  using C = C3Cfg<TR>;
  constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0;
  constexpr int IW = C::IW, NPH = C::NPH, NI = C::NI, NIN = (IN16 ? 1 : 2) * C::IN_PER_THREAD;      // load instructions of one tile-chunk fetch
  extern __shared__ __attribute__((aligned(1024))) char c3_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)c3_smem;

  // workgroup -> (tile, output block): the blocks of a tile get ids congruent mod 8 (one XCD, one L2 copy of their input tile: igemm_h16.h)
  int bid = blockIdx.x, nb = 0;
  if (p.n_blocks > 1) {
    const int tiles = (int)(gridDim.x / (unsigned)p.n_blocks);
    if ((tiles & 7) == 0) {
      const int run = 8 * p.n_blocks, within = bid % run;
      nb = within >> 3;
      bid = (bid / run) * 8 + (within & 7);
    } else {
      nb = bid % p.n_blocks;
      bid /= p.n_blocks;
    }
  }
  const int tile_x = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int tile_y = bid % p.tiles_y;
  const int img = bid / p.tiles_y;
  const int n0 = nb * C::BN;
  const int y0 = tile_y * C::TH, x0 = tile_x * C::TW;
  const int iy0 = y0 - 1, ix0 = x0 - 1;                               // TR: taps reach one pixel up / left; 3x3 stride 1 SAME: pad 1

  // ---- input staging plan: piece idx = (pixel, 8-channel group q); the raw buffer covers this image, out-of-image pixels are out of range
  constexpr unsigned IB = IN16 ? 2u : 4u;
  const c3_v4i in_rsrc = c3_rsrc(reinterpret_cast<const char*>(p.in) + ((size_t)img * p.H * p.W * p.in_cs + p.in_coff) * IB);
  unsigned in_voff[C::IN_PER_THREAD];
  int in_loff[C::IN_PER_THREAD];
#pragma unroll
  for (int i = 0; i < C::IN_PER_THREAD; ++i) {
    const int idx0 = tid + i * 256;
    const int idx = idx0 < C::IN_PIECES ? idx0 : C::IN_PIECES - 1;
    const int pix = idx >> 2, q = idx & 3;
    const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
    const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    in_voff[i] = ok ? (unsigned)(((iy * p.W + ix) * p.in_cs + q * 8) * (int)IB) : kLaneOff;
    in_loff[i] = idx0 < C::IN_PIECES ? pix * C::PIX_B + q * 16 : -1;
  }
  f32x4 st[NIN];                                                       // staging registers of the tile chunk in flight
  auto fetch_in = [&](int ch) {
    const unsigned soff = (unsigned)(ch * 32) * IB;
#pragma unroll
    for (int i = 0; i < C::IN_PER_THREAD; ++i) {
      if constexpr (IN16) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(st[i]) : "v"(in_voff[i]), "s"(in_rsrc), "s"(soff) : "memory");
      } else {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(st[2 * i]) : "v"(in_voff[i]), "s"(in_rsrc), "s"(soff) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(st[2 * i + 1]) : "v"(in_voff[i]), "s"(in_rsrc), "s"(soff) : "memory");
      }
    }
  };
  // C3_WAIT_IN(N): wait until at most N vector-memory operations issued after the fetch are outstanding, i.e. the fetch has landed; the
  // staging registers are tied to the wait so that nothing reads them before it
  static_assert(NIN == 3 || NIN == 4 || NIN == 6 || NIN == 8, "asm operand list of C3_WAIT_IN");
#define C3_WAIT_IN(N) do { \
    if constexpr (NIN == 3) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2])); \
    else if constexpr (NIN == 4) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[NIN > 3 ? 3 : 0])); \
    else if constexpr (NIN == 6) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[NIN > 3 ? 3 : 0]), "+v"(st[NIN > 4 ? 4 : 0]), "+v"(st[NIN > 5 ? 5 : 0])); \
    else asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[NIN > 3 ? 3 : 0]), "+v"(st[NIN > 4 ? 4 : 0]), "+v"(st[NIN > 5 ? 5 : 0]), "+v"(st[NIN > 6 ? 6 : 0]), "+v"(st[NIN > 7 ? 7 : 0])); \
  } while (0)
  auto store_in = [&](int buf) {
    char* dst = c3_smem + buf * C::IN_BYTES;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < C::IN_PER_THREAD; ++i) {
      if (in_loff[i] < 0) continue;
      if constexpr (IN16) {
        *reinterpret_cast<f32x4*>(dst + in_loff[i]) = st[i];
      } else {                                                         // fp32 activations: the values the fp32 form of this layer rounds at this point
        f16x8 hi, lo;
        split8(st[2 * i], st[2 * i + 1], hi, lo);
        amax = amax8(st[2 * i], st[2 * i + 1], amax);
        *reinterpret_cast<f16x8*>(dst + in_loff[i]) = hi;
      }
    }
    if constexpr (!IN16) range_report(amax, p.range_flag);
  };

  // ---- weights: trio tr of chunk ch is one 12-KB image; wave w moves its pieces w, w + 4, w + 8 into slot tr
  const char* w_blk = reinterpret_cast<const char*>(p.w) + (size_t)nb * p.nchunk * (3 * C::TRIO_B);
  const unsigned dma_v = (unsigned)(lane * 16);
  auto dma_trio = [&](int ch, int tr) {
    const char* src = w_blk + ((size_t)ch * 3 + tr) * C::TRIO_B;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const unsigned lds_addr = lds0 + C::W_OFF + tr * C::TRIO_B + (wave + 4 * i) * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_v), "s"(src + (wave + 4 * i) * 1024), "s"(lds_addr) : "memory", "m0");
    }
  };

  // ---- fragment addresses
  const unsigned a_base = (unsigned)((wave * IW + r) * C::PIX_B + h * 16);                 // + (dy * IW + dx) * 80 + g * 32 + buffer
  unsigned b_base[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) b_base[g] = (unsigned)(C::W_OFF + r * 64 + (((2 * g + h) ^ ((r >> 2) & 3)) << 4));      // + tap * 4096 + ni * 2048 + trio slot
  auto tap_px = [](int t) -> int {                                     // input-tile pixel offset of tap t
    if (TR) return ((t / 3 == 2 ? 0 : 1) * IW + (t % 3 == 2 ? 0 : 1));
    return (t / 3) * IW + (t % 3);
  };
  auto tap_ph = [](int t) -> int { return TR ? ((t / 3 == 1 ? 2 : 0) + (t % 3 == 1 ? 1 : 0)) : 0; };

  f32x16 acc[NPH][NI];
  {
    float bias_n[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + ni * 32 + r;
      bias_n[ni] = p.bias[n < p.n_pad ? n : 0];
    }
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ph][ni] = bias_tile(h, bias_n[ni]);
  }

  // ---- prologue: trios 0-2 of chunk 0 requested, tile chunk 0 staged
  dma_trio(0, 0);
  dma_trio(0, 1);
  dma_trio(0, 2);
  fetch_in(0);
  C3_WAIT_IN(0);
  store_in(0);
  __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
  __builtin_amdgcn_s_barrier();

  // fragments of one (tap, K group) unit: 1 A + NI B reads feed NI matrix instructions; two register sets — unit u + 1 is read while
  // unit u multiplies (whole-tap sets, 48 registers, pushed the transposed form — 128 accumulator registers — over the 256 of two waves
  // per SIMD: 200 spilled registers)
  f16x8 af[2], bf[2][NI];
  auto read_unit = [&](int set, unsigned a_cur, int u) {          // a_cur = a_base + (tile buffer of this chunk)
    const int t = u >> 1, g = u & 1, tr = t / 3, tt = t % 3;
    af[set] = *reinterpret_cast<const f16x8*>(c3_smem + a_cur + tap_px(t) * C::PIX_B + g * 32);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      bf[set][ni] = *reinterpret_cast<const f16x8*>(c3_smem + b_base[g] + tr * C::TRIO_B + tt * C::TAP_B + ni * 2048);
  };

  // One chunk = 3 trios = 9 taps = 36 matrix instructions.  VMEM operations of a wave, in issue order (D = 3 DMA pieces):
  //   [trio 0]  fetch(chunk + 1) NIN      barrier 0 waits for nothing new (trio 1 was published by the last barrier of the previous chunk)
  //             after barrier 0: DMA(next chunk, trio 0) D                         -> slot 0 is free once every wave has finished trio 0
  //   [trio 1]  the fetch has landed when at most D operations are younger: convert / write the tile into the other buffer
  //             barrier 1 publishes it;  after it: DMA(next, trio 1) D
  //   [trio 2]  barrier 2: DMA(next, trio 0) and (next, trio 1) must have landed: at most D younger (next, trio 2 is requested after it) ... see below
#pragma unroll 1
  for (int ch = 0; ch < p.nchunk; ++ch) {
    const bool more = ch + 1 < p.nchunk;
    const int buf = ch & 1;
    const unsigned a_cur = a_base + (unsigned)buf * C::IN_BYTES;
    if (more && C3_DIAG != 4) fetch_in(ch + 1);
    read_unit(0, a_cur, 0);
#pragma unroll
    for (int u = 0; u < 18; ++u) {
      const int t = u >> 1, set = u & 1;
      // fragments of the next unit: inside a trio only (the next trio's slot is published by the barrier in between)
      if (u % 6 != 5) read_unit(set ^ 1, a_cur, u + 1);
      if (u == 8 && more && C3_DIAG != 4) {                            // mid trio 1: the next tile chunk goes to the other buffer
        C3_WAIT_IN(3);                                                 // younger: DMA(next, trio 0)
        store_in(buf ^ 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int ph = tap_ph(t);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ph][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[set], bf[set][ni], acc[ph][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u % 6 == 5) {
        const int tr = u / 6;
        // end of trio tr: the next trio's weights must be in LDS for everyone (requested two barriers ago: at most the D pieces requested
        // one barrier ago — and the NIN loads of this chunk's fetch, at trio 0 — are younger), this wave's ds_writes of the next tile done
        if (C3_DIAG == 1 || C3_DIAG == 3 || C3_DIAG == 4) {
          __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));
        } else if (tr == 0) {
          if (more) __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(3 + NIN)); else __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(3));
        } else if (tr == 1 && !more) {
          __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));            // the last chunk has no fetch whose wait covers DMA(this chunk, trio 2)
        } else {
          __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(3));
        }
        if (C3_DIAG != 2) __builtin_amdgcn_s_barrier();
        if (more && C3_DIAG != 3) dma_trio(ch + 1, tr);               // slot tr is free: every wave is past trio tr
        if (tr < 2) read_unit(set ^ 1, a_cur, u + 1);
      }
    }
  }
  // NOTE on the counted waits: DMA(next, tr) is requested AFTER barrier tr; trio tr of the next chunk is read after barrier 2 of THIS chunk
  // (tr = 0), after barrier 0 / 1 of the next chunk (tr = 1 / 2).  Barrier 2 therefore needs DMA(next, 0) landed: younger = DMA(next, 1) = 3
  // (DMA(next, 2) follows the barrier) — vmcnt(3); barrier 0 of the next chunk needs DMA(next, 1): younger = DMA(next, 2) = 3 + that chunk's
  // fetch NIN; barrier 1 needs DMA(next, 2): younger = DMA(next + 1, 0) = 3.  The last chunk requests nothing: its waits are looser than needed.
  // ---- epilogue (igemm_h16.h): bias is in the accumulators; LeakyReLU; NHWC raw-buffer stores
  __builtin_amdgcn_s_setprio(3);
  constexpr int SX = TR ? 2 : 1;
  const size_t blk_pix = (size_t)img * p.Ho * p.Wo + (size_t)(SX * y0) * p.Wo + SX * x0;
  constexpr unsigned OB = OUT16 ? 2u : 4u;
  const unsigned lane_out = ((unsigned)(SX * 4 * h) * (unsigned)p.out_cs + (unsigned)r) * OB;
  const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.out) + (blk_pix * p.out_cs + p.out_coff) * OB));
  // fp16 output, 16-byte stores (round 6): in the accumulator layout a lane holds ONE channel of 16 pixels, so the fp16 pack wrote 2 bytes per
  // lane — 128 store instructions of 128 bytes per wave and tile, ~60 cycles each at the issue (in-kernel stamps: 8.7 k of a workgroup's
  // cycles, more than its matrix loop once that is trio-stepped).  The tile of one output parity goes through LDS instead ([32 pixels][64
  // channels] halves, rows of 144 bytes: the two half-waves' 2-byte writes and the 16-byte reads are conflict-free) and leaves as 16-byte
  // pieces, eight lanes per pixel = its whole 128-byte line: 16 store instructions per wave and tile.  Every LDS byte of the loop is dead
  // behind its last barrier; a wave uses its own 4.5-KB region and LDS executes a wave's accesses in order: no further barrier.
  bool wide = false;
  if constexpr (OUT16) wide = ((p.out_cs | p.out_coff | p.n_store) & 7) == 0;
  if (OUT16 && wide) {
    char* reg = c3_smem + wave * (32 * 144);
    const int piece = lane & 7, pc = lane >> 3;                       // this lane's 16-byte piece (8 channels) of pixel column pc + 8 q
    const unsigned v16 = n0 + piece * 8 < p.n_store ? (unsigned)((SX * pc) * p.out_cs + piece * 8) * 2u : kLaneOff;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      float amax = 0.f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        f32x16 v = acc[ph][ni];
        if (p.act) {
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const f32x2 y = leaky_relu2(f32x2{v[i], v[i + 1]});
            v[i] = y[0];
            v[i + 1] = y[1];
          }
        }
#pragma unroll
        for (int i = 0; i < 16; i += 4) amax = amax4(f32x4{v[i], v[i + 1], v[i + 2], v[i + 3]}, amax);
#pragma unroll
        for (int i = 0; i < 16; ++i)
          *reinterpret_cast<_Float16*>(reg + ((i & 3) + 8 * (i >> 2) + 4 * h) * 144 + (ni * 32 + r) * 2) = (_Float16)v[i];
      }
      range_report(amax, p.range_flag);
      const unsigned tile_off = (unsigned)((SX * wave + (TR ? (ph >> 1) : 0)) * p.Wo + (TR ? (ph & 1) : 0)) * (unsigned)p.out_cs + (unsigned)n0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        typedef unsigned c3_u4 __attribute__((ext_vector_type(4)));
        const c3_u4 d = *reinterpret_cast<const c3_u4*>(reg + (8 * q + pc) * 144 + piece * 16);
        __builtin_amdgcn_raw_buffer_store_b128(d, orsrc, v16, (tile_off + (unsigned)(SX * 8 * q) * (unsigned)p.out_cs) * 2u, 0);
      }
    }
    return;
  }
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nt = n0 + ni * 32;
      const unsigned voff = nt + r < p.n_store ? lane_out : kLaneOff;
      const unsigned tile_off = (unsigned)((SX * wave + (TR ? (ph >> 1) : 0)) * p.Wo + (TR ? (ph & 1) : 0)) * (unsigned)p.out_cs + (unsigned)nt;
      f32x16 v = acc[ph][ni];
      if (p.act) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const f32x2 y = leaky_relu2(f32x2{v[i], v[i + 1]});
          v[i] = y[0];
          v[i + 1] = y[1];
        }
      }
      if constexpr (OUT16) {
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 4) amax = amax4(f32x4{v[i], v[i + 1], v[i + 2], v[i + 3]}, amax);
        range_report(voff == kLaneOff ? 0.f : amax, p.range_flag);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = SX * ((i & 3) + 8 * (i >> 2));
        const unsigned soff = (tile_off + (unsigned)k * (unsigned)p.out_cs) * OB;
        if constexpr (OUT16)
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (_Float16)v[i]), orsrc, voff, soff, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i]), orsrc, voff, soff, 0);
      }
    }
}

template <bool TR, int IO>
inline hipError_t launch_conv3_f16(ConvArgs a, int batch, hipStream_t stream) {
  using C = C3Cfg<TR>;
  auto kern = conv3_f16_kernel<TR, IO>;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  const int mh = TR ? a.H : a.Ho, mw = TR ? a.W : a.Wo;
  if (mh % C::TH != 0 || mw % C::TW != 0 || a.nchunk < 1) return hipErrorInvalidValue;
  a.tiles_x = mw / C::TW;
  a.tiles_y = mh / C::TH;
  const int nblk = (a.n_store + C::BN - 1) / C::BN;
  a.n_blocks = nblk;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.tiles_x * a.tiles_y * batch * nblk)), dim3(256), C::SMEM_BYTES, stream, a);
  return hipGetLastError();
}

#undef C3_WAIT_IN

}  // namespace bsr
