// Implicit-GEMM convolution on the CDNA4 16-bit matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate), NHWC fp32 in / out.
//
// Same layers, tiling, LDS rings and epilogue as igemm_conv.h (Conv2D 3x3 / stride-2 3x3 / Conv2DTranspose 3x3 s2 + folded
// BatchNorm + LeakyReLU, /root/reference/model.py:115-177); what changes is the operand format inside the workgroup:
//
//  NSPLIT = 2 ("f32x3", split-precision contraction): every fp32 operand x is split at staging time into two fp16 planes,
//      hi = fp16(x) and lo = fp16(x - hi), and a K group of 16 channels is contracted with THREE matrix instructions into the
//      same fp32 accumulator: hi.hi + hi.lo + lo.hi (the lo.lo term, <= 2^-22 relative, is dropped).  The fp16 products are exact
//      in fp32.  Accuracy of the split itself: x - hi is at most 2^-11 |x|, and the lo plane is stored UNSCALED, so it is a
//      NORMAL fp16 number (x = hi + lo to 2^-22 |x|) only while |x| >= 2^-3; below that lo is an fp16 subnormal (the gfx950
//      matrix cores do not flush them) with spacing 2^-24, i.e. every operand carries an ABSOLUTE error of up to 2^-25 ~ 3e-8 —
//      2^-17 .. 2^-20 relative for the |w| ~ 0.004 .. 0.05 of a folded conv kernel.  That is "fp32-class" only in the sense the
//      tests state it: the same 1e-3 / 2e-5 end-to-end tolerances as the fp32 path hold with room (measured 4.5e-6 vs 4.3e-6), not
//      2^-22 per product.  Activations and outputs stay fp32 in HBM, so the kernels are drop-in for the fp32 ones; weights are
//      split offline (pack.py).  Range: |x| >= 65504 (fp16 max) makes hi = inf and lo = NaN where the fp32 path stays finite:
//      weights are range-checked at pack time; activations are checked on the device — a staged operand whose hi half is not
//      finite sets the handle's sticky range flag (BSR_ERR_RANGE from the next bsr_* call / bsr_check_range).
//  NSPLIT = 1 ("f16", BASELINE configs[3]): hi plane only, one matrix instruction per K group, half-width LDS tiles.
//
// LDS row of one pixel / one output channel: [CC halves hi | CC halves lo (NSPLIT = 2) | 8 halves pad]  = LDP 32-bit words;
// lane (r = l & 31, h = l >> 5) reads k = 8h .. 8h+7 of a 16-channel K group with one ds_read_b128 per plane — exactly the
// A / B operand map of v_mfma_f32_32x32x16_f16 (cdna_hip_programming.md §3) — at the same word addresses the fp32 kernel uses
// (row * LDP + 4h + 8g), so the bank behaviour is the fp32 kernel's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "igemm_conv.h"

#ifndef BSR_PAIR_S2_H16
#define BSR_PAIR_S2_H16 0
#endif
#ifndef BSR_DMA_ASM
#define BSR_DMA_ASM 1
#endif
#ifndef BSR_H16_AREUSE
#define BSR_H16_AREUSE 1    // transposed convs: taps that read the same input offset run back to back and share their A fragments
#endif
#ifndef BSR_H16_RING
#define BSR_H16_RING 4      // slots of the LDS-DMA weight ring: the image of step s + RING - 1 is requested at the top of step s
#endif
#ifndef BSR_H16_PACK_STORES
#define BSR_H16_PACK_STORES 0          // 1: the fp16 epilogues trade values between neighbouring lanes and store two channels (4 bytes) per lane — measured: transposed convs 370 vs 359 us, stem 85 vs 88 (the permutes and selects cost more than the halved store count saves): off
#endif
#ifndef BSR_H16_WIDE_STORES
#define BSR_H16_WIDE_STORES 1   // fp32 outputs leave through LDS as 16-byte stores (round 6; 0 = 16 dword stores per accumulator tile)
#endif
#ifndef BSR_H16_BDEEP
#define BSR_H16_BDEEP 0     // 1: f16 transposed convs read the B fragments of a whole step one step ahead (measured: 355.9 vs 358.1 us, nothing — off)
#endif
#ifndef BSR_H16_FETCH_TAP_F16
#define BSR_H16_FETCH_TAP_F16 5   // = T - 4 of the transposed 3x3 layers
#endif
#ifndef BSR_H16_RING_F16
#define BSR_H16_RING_F16 8  // the same for NSPLIT = 1 (f16 mode): its steps are a third as long, so the same latency spans more of them
#endif

namespace bsr {

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 encoding: vmcnt = simm16[15:14 | 3:0], expcnt [6:4], lgkmcnt [11:8])
constexpr int waitcnt_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }
// vmcnt <= n AND lgkmcnt == 0 (every LDS access of this wave done): what a workgroup barrier needs while LDS-DMAs stay in flight.
// __syncthreads() cannot be used there: hipcc drains vmcnt to 0 in front of it as soon as a global_load_lds is outstanding.
// fp16 epilogue stores, two channels per lane (round 5, opt-in BSR_H16_PACK_STORES: measured slower, see the switch).  In a 32x32 accumulator tile lane r holds ONE channel of 16 pixels, so a
// 2-byte store per value moved 128 bytes per instruction.  Neighbouring lanes (channels r, r ^ 1) trade values through a quad
// permute: for the pixel pair (k, k + step) the even lane ends up with both channels of pixel k, the odd lane with both of pixel
// k + step — one 4-byte store each: half the store instructions, 256 bytes per instruction, the same fp16 values (RNE) as before.
__device__ __forceinline__ unsigned pack_pair_f16(float own_k, float own_k1, bool odd) {
  const float send = odd ? own_k : own_k1;
  const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
  const _Float16 lo = (_Float16)(odd ? recv : own_k), hi = (_Float16)(odd ? own_k1 : recv);
  return (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
}

constexpr int waitcnt_vm_lgkm0(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4); }

template <int KH, int KW, int S, bool TR, int TH, int TW, int WM, int WN, int MI, int NI, int CC, int INB, int NSPLIT>
struct H16Cfg {
  static constexpr int T = KH * KW;
  static constexpr int IH = TR ? TH + 1 : (TH - 1) * S + KH;
  static constexpr int IW = TR ? TW + 1 : (TW - 1) * S + KW;
  static constexpr int LDP = (NSPLIT * CC + 8) / 2;             // 32-bit words per LDS row
  static constexpr int LO = CC / 2;                              // word offset of the lo plane inside a row
  static constexpr int G = CC / 16;                              // K groups (one 32x32x16 instruction per plane pair) per chunk
  static constexpr int BN = WN * NI * 32;
  static constexpr int BM = WM * MI * 32;
  static constexpr int NPH = TR ? 4 : 1;
  static constexpr int IN_WORDS = IH * IW * LDP;
  // Weight rows of the DMA-fed f32x3 layers with 32-channel chunks are stored UNPADDED (128 bytes: 4 hi + 4 lo 16-byte slots) with
  // the slot index XOR-swizzled by (row >> 1) & 7 (applied offline by pack.py): an LDS-DMA piece costs the issuing wave 100+
  // cycles beside the matrix stream, and 64 rows x 128 B are 8 pieces = 2 per wave where the padded 144-B rows needed 3.
  // Conflict-free for ds_read_b128: two 128-B rows fill one 256-B bank row and the 8 even (odd) rows of every 16-lane group
  // get 8 different slots.
  static constexpr bool SWZ = NSPLIT == 2 && CC == 32 && T > 1;
  static constexpr bool PAIR = S == 2 && CC == 16 && INB == 1 && BSR_PAIR_S2_H16;   // igemm_conv.h's paired half-line fetch: removes the same over-fetch here, but costs 4-11 % of time (longer prologue) -> off
  static constexpr int LDPW = SWZ ? 32 : LDP;                    // words per weight row
  static constexpr int W_WORDS = BN * LDPW;
  // Weights by LDS-DMA (multi-tap layers): with 16-bit operands a (chunk, tap) step is only NI * G * NSPLIT' matrix instructions
  // long (a few hundred cycles), far less than an L2 round trip, so the register-staged ring of igemm_conv.h (load at the start
  // of a step, ds_write at its end) exposes the load latency on every step.  global_load_lds_dwordx4 copies the packed image
  // straight into a 4-slot LDS ring THREE steps ahead, needs no registers, and stays in flight across two barriers.
  static constexpr bool DMAW = T > 1;
  static constexpr int W_CHUNKS = (W_WORDS * 4 + 1023) / 1024;   // 1 KiB = one wave-instruction of 64 lanes x 16 B
  static constexpr int W_DMA_PER_WAVE = (W_CHUNKS + 3) / 4;
  static constexpr int W_SLOT_WORDS = DMAW ? W_CHUNKS * 256 : W_WORDS;
  static constexpr int W_SLOTS = DMAW ? (NSPLIT == 1 ? BSR_H16_RING_F16 : BSR_H16_RING) : 3;
  static constexpr int SMEM_BYTES = (INB * IN_WORDS + W_SLOTS * W_SLOT_WORDS) * 4;
  static constexpr int IN_V8 = IH * IW * (CC / 8);               // 8-channel (32-byte) pieces of one input-tile chunk
  static constexpr int IN_PER_THREAD = (IN_V8 + 255) / 256;
  static constexpr int W_V4 = W_WORDS / 4;
  static constexpr int W_PER_THREAD = (W_V4 + 255) / 256;
  static_assert(NSPLIT == 1 || NSPLIT == 2, "one (f16) or two (f32x3) fp16 planes");
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(BM == TH * TW, "M tile must equal the spatial tile");
  static_assert(CC % 16 == 0, "channel chunk must be a multiple of the 16-wide K group");
  static_assert(!TR || ((T * (CC / 16)) % 2 == 0), "fragment slot parity");
  static_assert(!TR || (KH == 3 && KW == 3 && S == 1), "transposed path is ConvT(3, stride 2)");
  static_assert(INB == 1 || (T == 1 ? INB == 3 : INB == 2), "input buffers: 1, or 2 (taps > 1) / 3 (1x1)");
  static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
  static_assert(W_WORDS % 4 == 0, "weight image is copied in 16-byte pieces");
};

// split2 / split8 / amax8 / range_report (the operand split and the range guard of the 16-bit modes) live in mfma_common.h: gemm_tail.h uses them too.

// IO (NSPLIT = 1 only — the fp16 pack of BASELINE configs[3]): bit 0 = the INPUT tensor is fp16 in HBM (in_cs / in_coff count halves; a
// staged piece is one 16-byte load that goes to LDS unconverted — the fp32 form rounds the same values to fp16 at this point, so
// nothing is lost), bit 1 = the OUTPUT is written as fp16 (out_cs / out_coff count halves).  Both halve that tensor's HBM bytes.
template <int KH, int KW, int S, bool TR, int TH, int TW, int WM, int WN, int MI, int NI, int CC, int INB, int NSPLIT, int IO = 0>
__global__ __launch_bounds__(256, 2) void igemm_h16_kernel(ConvArgs p) {
  using C = H16Cfg<KH, KW, S, TR, TH, TW, WM, WN, MI, NI, CC, INB, NSPLIT>;
  constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0;
  static_assert(IO == 0 || NSPLIT == 1, "fp16 activation I/O belongs to the f16 mode");
  constexpr int T = C::T, IW = C::IW, LDP = C::LDP, LDPW = C::LDPW, BN = C::BN, NPH = C::NPH, G = C::G, LO = C::LO;
  constexpr bool SWZ = C::SWZ;
  constexpr bool PAIR = C::PAIR;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + INB * C::IN_WORDS;

  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

  int bid = blockIdx.x;
  int nb = blockIdx.y;
  if (p.n_blocks > 0) {
    // The N blocks of a pixel tile read the SAME input tile.  Workgroup ids go round-robin over the 8 XCDs, each with its own L2
    // (MI355X_MICROARCH.md), so N blocks with consecutive ids fetch that tile into several L2s: round 4 measured 1.85 x the
    // algorithmic read bytes on res*.conv1 at an L2 hit rate of 0.39.  Round 5: within a run of 8 * n_blocks ids the tile is
    // id % 8 and the N block id / 8 — a tile's N blocks are congruent mod 8 (one XCD, one L2 copy) and 8 ids apart (in flight together).
    const int tiles = (int)(gridDim.x / (unsigned)p.n_blocks);
    if ((tiles & 7) == 0) {
      const int run = 8 * p.n_blocks, within = bid % run;
      nb = within >> 3;
      bid = (bid / run) * 8 + (within & 7);
    } else {                       // N block fastest: the workgroups that re-read one input tile are at least dispatched together
      nb = bid % p.n_blocks;
      bid /= p.n_blocks;
    }
  }
  const int tile_x = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int tile_y = bid % p.tiles_y;
  const int img = bid / p.tiles_y;
  const int n0 = nb * BN;
  const int y0 = tile_y * TH, x0 = tile_x * TW;
  const int iy0 = TR ? y0 - 1 : y0 * S - p.pad_t;
  const int ix0 = TR ? x0 - 1 : x0 * S - p.pad_l;
  // IN16: p.in is an fp16 tensor; element offsets are the same, byte offsets half
  const char* in_img_b = reinterpret_cast<const char*>(p.in) + ((size_t)img * p.H * p.W * p.in_cs + p.in_coff) * (IN16 ? 2 : 4);

  int a_base[MI], b_base[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int px = (wm * MI + mi) * 32 + r;
    const int ty = px / TW, tx = px % TW;
    a_base[mi] = ((ty * S) * IW + tx * S) * LDP + 4 * h;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b_base[ni] = ((wn * NI + ni) * 32 + r) * LDPW + 4 * h;
  // swizzled rows: logical 16-byte slot (2g + h) of the hi plane, 4 + 2g + h of the lo plane -> physical slot ^ ((r >> 1) & 7)
  int b_sw[NI][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) b_sw[ni][ci] = ((wn * NI + ni) * 32 + r) * LDPW + 4 * ((2 * ci) ^ (h ^ ((r >> 1) & 7)));

  float bias_n[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + (wn * NI + ni) * 32 + r;
    bias_n[ni] = p.bias[n < p.n_pad ? n : 0];
  }
  f32x16 acc[NPH][MI][NI];
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ph][mi][ni] = bias_tile(h, bias_n[ni]);      // fp32 matrix instruction (igemm_conv.h): the bias keeps its 24 bits

  // staging addresses: wave-uniform base + per-thread constant, computed once (see igemm_conv.h)
  unsigned in_goff[C::IN_PER_THREAD], w_off[C::W_PER_THREAD];
  int in_loff[C::IN_PER_THREAD];
  unsigned in_okmask = 0u;
#pragma unroll
  for (int i = 0; i < C::IN_PER_THREAD; ++i) {
    const int idx0 = tid + i * 256;
    const int idx = idx0 < C::IN_V8 ? idx0 : C::IN_V8 - 1;
    const int pix = idx / (CC / 8), q = idx % (CC / 8);
    const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
    const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
    in_goff[i] = (unsigned)(((iyc * p.W + ixc) * p.in_cs + q * 8) * (IN16 ? 2 : 4));
    in_loff[i] = idx0 < C::IN_V8 ? pix * LDP + q * 4 : -1;            // words: 8 halves = 4 words per piece and plane
    in_okmask |= (ok ? 1u : 0u) << i;
  }
#pragma unroll
  for (int i = 0; i < C::W_PER_THREAD; ++i) {
    const int idx0 = tid + i * 256;
    w_off[i] = (unsigned)((idx0 < C::W_V4 ? idx0 : C::W_V4 - 1) * 16);
  }
  auto fetch_in = [&](int ch, f32x4 (&regs)[2 * C::IN_PER_THREAD]) {
    const char* base = in_img_b + (size_t)ch * CC * (IN16 ? 2 : 4);
#pragma unroll
    for (int i = 0; i < C::IN_PER_THREAD; ++i) {
      regs[2 * i] = *reinterpret_cast<const f32x4*>(base + in_goff[i]);              // IN16: these 16 bytes are the piece's 8 halves
      if constexpr (!IN16) regs[2 * i + 1] = *reinterpret_cast<const f32x4*>(base + in_goff[i] + 16);
    }
  };
  auto store_in = [&](int off, const f32x4 (&regs)[2 * C::IN_PER_THREAD]) {
    float amax = 0.f;                                                                // range guard: largest |x| this thread converts
#pragma unroll
    for (int i = 0; i < C::IN_PER_THREAD; ++i) {
      if (in_loff[i] >= 0) {
        f32x4 a = regs[2 * i], b = regs[2 * i + 1];
        if (!((in_okmask >> i) & 1u)) { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }      // TF SAME zero padding (all-zero bits are 0.0 in fp16 too)
        if constexpr (IN16) {
          *reinterpret_cast<f32x4*>(s_in + off + in_loff[i]) = a;
        } else {
          f16x8 hi, lo;
          split8(a, b, hi, lo);
          amax = amax8(a, b, amax);
          *reinterpret_cast<f16x8*>(s_in + off + in_loff[i]) = hi;
          if (NSPLIT == 2) *reinterpret_cast<f16x8*>(s_in + off + in_loff[i] + LO) = lo;
        }
      }
    }
    if constexpr (!IN16) range_report(amax, p.range_flag);
  };
  // weights: p.w is the packed fp16 LDS image [chunk][tap][n_pad][LDP words], copied verbatim
  auto fetch_w = [&](int step, f32x4 (&regs)[C::W_PER_THREAD]) {
    const char* base = reinterpret_cast<const char*>(p.w + ((size_t)step * p.n_pad + n0) * LDPW);
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i) regs[i] = *reinterpret_cast<const f32x4*>(base + w_off[i]);
  };
  auto store_w = [&](int off, const f32x4 (&regs)[C::W_PER_THREAD]) {
    char* dst = reinterpret_cast<char*>(s_w + off);
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i) {
      if (tid + i * 256 < C::W_V4) *reinterpret_cast<f32x4*>(dst + w_off[i]) = regs[i];
    }
  };

  // LDS-DMA of one step's weight image into ring slot `off`: wave w copies 1-KiB pieces w, w+4, ... (a last piece shorter than
  // 1 KiB spills into the slot's padding with clamped sources).  When the piece count is not a multiple of 4 the waves past the
  // remainder issue one instruction fewer per step (`dma_full` false) — each piece costs the CU's LDS-DMA path ~70 cycles, so none
  // is copied twice — and their counted waits allow correspondingly fewer instructions in flight.
  constexpr bool DMAW = C::DMAW;
  constexpr int WREM = C::W_CHUNKS % 4;
  const bool dma_full = WREM == 0 || wave < WREM;
  unsigned dma_src[C::W_DMA_PER_WAVE];
#pragma unroll
  for (int i = 0; i < C::W_DMA_PER_WAVE; ++i) {
    const int c = min(wave + 4 * i, C::W_CHUNKS - 1);
    dma_src[i] = (unsigned)min(c * 1024 + lane * 16, C::W_WORDS * 4 - 16);
  }
  auto dma_w = [&](int step, int off) {      // step = index of the packed (chunk, tap) weight image
#ifdef H16_DIAG_DMA_SAME
    step = 0;
#endif
    const char* base = reinterpret_cast<const char*>(p.w + ((size_t)step * p.n_pad + n0) * LDPW);
#pragma unroll
    for (int i = 0; i < C::W_DMA_PER_WAVE; ++i) {
#ifdef H16_DIAG_DMA_HALF
      if (i & 1) continue;
#endif
      if (i == C::W_DMA_PER_WAVE - 1 && !dma_full) continue;                         // wave-uniform
      const int c = min(wave + 4 * i, C::W_CHUNKS - 1);
#if BSR_DMA_ASM
      // Written as asm on purpose: behind __builtin_amdgcn_global_load_lds hipcc 7.2 puts a full s_waitcnt lgkmcnt(0) in front of
      // the first matrix instruction after every later ds_read (it no longer tracks which LDS reads are outstanding once an LDS-DMA
      // is), which exposes the fragment reads of every step.  The waits this kernel needs for the DMAs are the explicit counted ones.
      const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_w + off + c * 256);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :: "v"(dma_src[i]), "s"(base), "s"(lds_addr) : "memory", "m0");
#else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + dma_src[i]),
                                       (__attribute__((address_space(3))) void*)(s_w + off + c * 256), 16, 0, 0);
#endif
    }
  };

  // Transposed conv: the 9 taps read only 4 different input offsets ((1,1): taps 0 1 3 4, (1,0): 2 5, (0,1): 6 7, (0,0): 8).  Run in
  // that order, a tap group keeps its A fragments in registers and LDS serves 4 instead of 9 A reads per chunk; the 16-bit kernels
  // are bound by LDS read bandwidth (1 KiB of fragments per matrix instruction = the LDS peak at the full matrix rate).
  constexpr bool AREUSE = TR && INB == 1 && C::DMAW && BSR_H16_AREUSE;
  auto tap_at = [&](int q) -> int {          // execution position -> tap
    if (!AREUSE) return q;
    constexpr int ord[9] = {0, 1, 3, 4, 2, 5, 6, 7, 8};
    return ord[q % 9];
  };
  auto img_of = [&](int ch, int q) -> int {      // weight image of execution position q (may run into the next chunks)
    return (ch + q / T) * T + tap_at(q % T);
  };
#ifdef BSR_STAMPS
  unsigned long long st0 = __builtin_amdgcn_s_memtime(), st1 = 0, st2 = 0;
  unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  f32x4 in_regs[2 * C::IN_PER_THREAD];
  f32x4 in_regs2[PAIR ? 2 * C::IN_PER_THREAD : 1];      // PAIR: the odd chunk of a pair, fetched together with the even one
  f32x4 w_regs[C::W_PER_THREAD];
  const int nsteps = p.nchunk * T;

  constexpr int R = C::W_SLOTS;                      // DMA ring: slot i holds step s + i, the last one receives step s + R - 1
  int w_cur = 0, w_n1 = C::W_SLOT_WORDS, w_n2 = 2 * C::W_SLOT_WORDS;
  int w_far[R > 3 ? R - 3 : 1];                      // slots of steps s + 3 .. s + R - 1
#pragma unroll
  for (int i = 3; i < R; ++i) w_far[i - 3] = i * C::W_SLOT_WORDS;
  int in_cur = 0, in_n1 = (INB > 1) ? C::IN_WORDS : 0, in_n2 = (INB > 2) ? 2 * C::IN_WORDS : 0;

  // prologue: the first steps staged synchronously
  if constexpr (DMAW) {
    dma_w(img_of(0, 0), w_cur);
    if (nsteps > 1) dma_w(img_of(0, 1), w_n1);
    if (nsteps > 2) dma_w(img_of(0, 2), w_n2);
#pragma unroll
    for (int i = 3; i < R - 1; ++i)
      if (nsteps > i) dma_w(img_of(0, i), w_far[i - 3]);
    fetch_in(0, in_regs);
    if constexpr (PAIR) { if (p.nchunk > 1) fetch_in(1, in_regs2); }
    store_in(0, in_regs);
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    __syncthreads();
  } else {
    fetch_in(0, in_regs);
    fetch_w(0, w_regs);
    store_in(0, in_regs);
    store_w(0, w_regs);
    if (nsteps > 1) {
      fetch_w(1, w_regs);
      store_w(w_n1, w_regs);
      if (T == 1 && INB == 3) {
        fetch_in(1, in_regs);
        store_in(in_n1, in_regs);
      }
    }
    __syncthreads();
  }

  f16x8 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];
  auto read_frags = [&](int slot, int a_off, int w_off_, int g) {      // A at a_off (group offset included), B = K group g of ring slot w_off_
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ah[slot][mi] = *reinterpret_cast<const f16x8*>(s_in + a_base[mi] + a_off);
      if (NSPLIT == 2) al[slot][mi] = *reinterpret_cast<const f16x8*>(s_in + a_base[mi] + a_off + LO);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if constexpr (SWZ) {
        bh[slot][ni] = *reinterpret_cast<const f16x8*>(s_w + w_off_ + b_sw[ni][g]);
        bl[slot][ni] = *reinterpret_cast<const f16x8*>(s_w + w_off_ + b_sw[ni][2 + g]);
      } else {
        bh[slot][ni] = *reinterpret_cast<const f16x8*>(s_w + b_base[ni] + w_off_ + g * 8);
        if (NSPLIT == 2) bl[slot][ni] = *reinterpret_cast<const f16x8*>(s_w + b_base[ni] + w_off_ + g * 8 + LO);
      }
    }
  };
  auto tap_offset = [&](int t) -> int {
    if (TR) {
      const int a = t / 3, b = t % 3;
      return (((a == 2) ? 0 : 1) * IW + ((b == 2) ? 0 : 1)) * LDP;
    }
    return ((t / KW) * IW + (t % KW)) * LDP;
  };
  if constexpr (AREUSE) {
    f16x8 aH[G][MI], aL[G][MI];                       // A fragments of the current input offset, one set per K group
    auto read_a = [&](int g, int a_off) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        aH[g][mi] = *reinterpret_cast<const f16x8*>(s_in + a_base[mi] + a_off + g * 8);
        if (NSPLIT == 2) aL[g][mi] = *reinterpret_cast<const f16x8*>(s_in + a_base[mi] + a_off + g * 8 + LO);
      }
    };
    auto read_b = [&](int slot, int w_off_, int g) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if constexpr (SWZ) {
          bh[slot][ni] = *reinterpret_cast<const f16x8*>(s_w + w_off_ + b_sw[ni][g]);
          bl[slot][ni] = *reinterpret_cast<const f16x8*>(s_w + w_off_ + b_sw[ni][2 + g]);
        } else {
          bh[slot][ni] = *reinterpret_cast<const f16x8*>(s_w + b_base[ni] + w_off_ + g * 8);
          if (NSPLIT == 2) bl[slot][ni] = *reinterpret_cast<const f16x8*>(s_w + b_base[ni] + w_off_ + g * 8 + LO);
        }
      }
    };
    // BDEEP (round 5, f16 mode): the B fragments of a WHOLE step are read one step before their matrix instructions (two register sets
    // by step parity) instead of one K group ahead.  With fp16 operands a K group is two 32-cycle matrix instructions; an LDS round
    // trip under eight reading waves is 150-200 cycles, so every group waited for its fragments (98 cycles per matrix instruction
    // measured in round 3; the stamps of the attention experiment, profiles/HISTORY.md round 5, show the same 44-80).  The image of
    // step s + 1 is complete when step s starts (the previous barrier published it).
    constexpr bool BDEEP = BSR_H16_BDEEP && NSPLIT == 1;
    f16x8 Bh[BDEEP ? 2 : 1][G][NI];
    auto read_b_step = [&](int set, int w_off_) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Bh[set][g][ni] = *reinterpret_cast<const f16x8*>(s_w + b_base[ni] + w_off_ + g * 8);
    };
#pragma unroll
    for (int g = 0; g < G; ++g) read_a(g, tap_offset(tap_at(0)));
    if constexpr (BDEEP) read_b_step(0, w_cur); else read_b(0, w_cur, 0);
    __builtin_amdgcn_s_setprio(0);
#ifdef BSR_STAMPS
    st1 = __builtin_amdgcn_s_memtime();
#endif
    for (int ch = 0; ch < p.nchunk; ++ch) {
      const bool more = ch + 1 < p.nchunk;
#pragma unroll
      for (int q = 0; q < T; ++q) {
        const int t = tap_at(q);
        const int s = ch * T + q;
        // the step of a chunk at which the NEXT chunk's input tile is requested.  vmcnt retires in order, so the request may stay in
        // flight for R - 3 steps at most (after that it is older than the weight image the step's barrier waits for).  fp32 input: 24-48
        // staging registers, requested 4 steps ahead; fp16 input (IN16): 12 registers and steps a third as long — requested earlier
        // (BSR_H16_FETCH_TAP_F16), with a ring deep enough to keep it in flight
        constexpr int kInFetchTap = IN16 ? BSR_H16_FETCH_TAP_F16 : T - 4;
        const bool hasD = s + R - 1 < nsteps;
#ifndef H16_DIAG_NO_DMA
        if (hasD) dma_w(img_of(ch, q + R - 1), w_far[R - 4]);
#endif
        if (q == kInFetchTap && more) fetch_in(ch + 1, in_regs);
        __builtin_amdgcn_sched_barrier(0);
        const int ph = ((t / 3 == 1) ? 2 : 0) + ((t % 3 == 1) ? 1 : 0);
        const bool a_last = q + 1 < T && tap_offset(tap_at(q + 1 < T ? q + 1 : q)) != tap_offset(t);      // the next tap reads another input offset
        if constexpr (BDEEP) {
          if (q + 1 < T) read_b_step((q + 1) & 1, w_n1);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int cur = (q * G + g) & 1, nxt = cur ^ 1;
          if constexpr (!BDEEP) {
            if (g + 1 < G) {
              read_b(nxt, w_cur, g + 1);
            } else if (q + 1 < T) {
              read_b(nxt, w_n1, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              if constexpr (BDEEP) {
                acc[ph][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aH[g][mi], Bh[q & 1][g][ni], acc[ph][mi][ni], 0, 0, 0);
              } else {
                if (NSPLIT == 2) {
                  acc[ph][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aL[g][mi], bh[cur][ni], acc[ph][mi][ni], 0, 0, 0);
                  acc[ph][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aH[g][mi], bl[cur][ni], acc[ph][mi][ni], 0, 0, 0);
                }
                acc[ph][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aH[g][mi], bh[cur][ni], acc[ph][mi][ni], 0, 0, 0);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
          if (a_last) {                 // in place, behind the last instructions that read the old fragments
            read_a(g, tap_offset(tap_at(q + 1 < T ? q + 1 : q)));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // The image of step s+2 must have landed before this barrier publishes it (see the other path).  LDS reads still in
        // flight here (the next step's first B fragments, a tap group's new A fragments) are of slots / tiles nobody overwrites
        // before the next barrier, so lgkmcnt is left alone.
        constexpr int NW = (R - 3) * C::W_DMA_PER_WAVE, NIN = (IN16 ? 1 : 2) * C::IN_PER_THREAD;
        const bool near_fetch = q >= kInFetchTap && q <= kInFetchTap + R - 3;
#ifndef H16_DIAG_NO_VMWAIT
        constexpr int NW1 = (R - 3) * (C::W_DMA_PER_WAVE - 1);          // waves that skip the last piece
        if (!hasD) {
          __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
        } else if (near_fetch && more) {
          if (dma_full) __builtin_amdgcn_s_waitcnt(waitcnt_vm(NW + NIN)); else __builtin_amdgcn_s_waitcnt(waitcnt_vm(NW1 + NIN));
        } else {
          if (dma_full) __builtin_amdgcn_s_waitcnt(waitcnt_vm(NW)); else __builtin_amdgcn_s_waitcnt(waitcnt_vm(NW1));
        }
#endif
        __builtin_amdgcn_s_barrier();
        if (q == T - 1 && more) {
          store_in(0, in_regs);
          __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));
          __builtin_amdgcn_s_barrier();
#pragma unroll
          for (int g = 0; g < G; ++g) read_a(g, tap_offset(tap_at(0)));
          if constexpr (BDEEP) read_b_step(0, w_n1); else read_b((T * G) & 1, w_n1, 0);
        }
        const int tw = w_cur; w_cur = w_n1; w_n1 = w_n2; w_n2 = w_far[0];
#pragma unroll
        for (int i = 0; i + 1 < R - 3; ++i) w_far[i] = w_far[i + 1];
        w_far[R - 4] = tw;
      }
    }
  } else {
  read_frags(0, in_cur + tap_offset(0), w_cur, 0);
  __builtin_amdgcn_s_setprio(0);
#ifdef BSR_STAMPS
  st1 = __builtin_amdgcn_s_memtime();
#endif

  for (int ch = 0; ch < p.nchunk; ++ch) {
    const bool more = ch + 1 < p.nchunk;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int s = ch * T + t;
      const bool has1 = s + 1 < nsteps, has2 = s + 2 < nsteps;
      constexpr bool kRing1x1 = (T == 1 && INB == 3);
      constexpr int kInFetchTap = (T >= 4) ? T - 4 : 0;
      constexpr int kInStoreTap = (INB == 2) ? T - 2 : T - 1;
      const bool fetch_now = kRing1x1 ? has2 : (T > 1 && t == kInFetchTap && more);
      const bool stage_in = kRing1x1 ? has2 : (INB == 2 && t == kInStoreTap && more);
      const bool has3 = s + R - 1 < nsteps;              // a step is left to request
      if constexpr (DMAW) {
#ifndef H16_DIAG_NO_DMA
#ifdef H16_DMA_UNCOND
        dma_w(min(s + R - 1, nsteps - 1), w_far[R - 4]);
#else
        if (has3) dma_w(s + R - 1, w_far[R - 4]);
#endif
#endif
      } else {
        if (has2) fetch_w(s + 2, w_regs);
      }
      if constexpr (PAIR) {      // the two 64-byte halves of a 128-byte line are requested together (igemm_conv.h)
        if (fetch_now && (ch & 1)) {
          fetch_in(ch + 1, in_regs);
          if (ch + 2 < p.nchunk) fetch_in(ch + 2, in_regs2);
        }
      } else {
        if (fetch_now) fetch_in(kRing1x1 ? ch + 2 : ch + 1, in_regs);
      }
      __builtin_amdgcn_sched_barrier(0);

      const int ph = TR ? (((t / 3 == 1) ? 2 : 0) + ((t % 3 == 1) ? 1 : 0)) : 0;
      const int tap_off = tap_offset(t);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int cur = (t * G + g) & 1, nxt = cur ^ 1;
        if (g + 1 < G) {
          read_frags(nxt, in_cur + tap_off + (g + 1) * 8, w_cur, g + 1);
        } else if (t + 1 < T) {
          read_frags(nxt, in_cur + tap_offset(t + 1 < T ? t + 1 : 0), w_n1, 0);
        } else if (INB > 1) {
          if (has1) read_frags(nxt, in_n1 + tap_offset(0), w_n1, 0);
        }
        if (g == G - 1) {                                                           // write point: stage step s+2
          if constexpr (!DMAW) {
            if (has2) store_w(w_n2, w_regs);
          }
          if (stage_in) store_in(kRing1x1 ? in_n2 : in_n1, in_regs);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if (NSPLIT == 2) {
              // small terms first, the leading term last
              acc[ph][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][mi], bh[cur][ni], acc[ph][mi][ni], 0, 0, 0);
              acc[ph][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][mi], bl[cur][ni], acc[ph][mi][ni], 0, 0, 0);
            }
            acc[ph][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][mi], bh[cur][ni], acc[ph][mi][ni], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }

      if constexpr (DMAW) {
        // The image of step s+2 (DMA issued at the top of step s+3-R) must have landed before this barrier publishes it.  vmcnt
        // retires in issue order: younger than that DMA are the R-3 DMAs of steps s+3 .. s+R-1 and an input-tile fetch issued
        // since; they may stay in flight.
        constexpr int NW = (R - 3) * C::W_DMA_PER_WAVE, NIN = (IN16 ? 1 : 2) * C::IN_PER_THREAD;      // load instructions of one input-tile fetch
        const bool near_fetch = T > 1 && t >= kInFetchTap && t <= kInFetchTap + R - 3;      // compile-time once t is unrolled
#ifdef H16_DIAG_NO_VMWAIT
        __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));
#else
        // input-tile loads issued after the awaited DMA (in this step or the previous one): they may stay in flight
        int n_in = 0;
        if (near_fetch && more) {
          if constexpr (PAIR) n_in = (ch & 1) ? (ch + 2 < p.nchunk ? 2 : 1) : 0;
          else n_in = 1;
        }
        // INB == 1: the LDS reads still in flight here (the next step's first fragments) are of a tile / ring slot nobody
        // overwrites before the next barrier, and no ds_write is pending, so lgkmcnt is left alone
#define BSR_WAIT_STEP(n) do { if constexpr (INB == 1) __builtin_amdgcn_s_waitcnt(waitcnt_vm(n)); else __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(n)); } while (0)
        constexpr int NW1 = (R - 3) * (C::W_DMA_PER_WAVE - 1);          // waves that skip the last piece
        if (!has3) {
          BSR_WAIT_STEP(0);
        } else if (n_in == 2) {
          if (dma_full) BSR_WAIT_STEP(NW + 2 * NIN); else BSR_WAIT_STEP(NW1 + 2 * NIN);
        } else if (n_in == 1) {
          if (dma_full) BSR_WAIT_STEP(NW + NIN); else BSR_WAIT_STEP(NW1 + NIN);
        } else {
          if (dma_full) BSR_WAIT_STEP(NW); else BSR_WAIT_STEP(NW1);
        }
#undef BSR_WAIT_STEP
#endif
#ifndef H16_DIAG_NO_STEP_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
      } else {
        __syncthreads();
      }
      if (INB == 1 && t == T - 1 && more) {
        if constexpr (PAIR) {
          if (ch & 1) store_in(0, in_regs); else store_in(0, in_regs2);
        } else {
          store_in(0, in_regs);
        }
        if constexpr (DMAW) {
          __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));      // the tile's ds_writes are done (lgkmcnt); DMAs may stay in flight
          __builtin_amdgcn_s_barrier();
        } else {
          __syncthreads();
        }
        read_frags(((T * G) & 1), tap_offset(0), w_n1, 0);
      }
      if constexpr (DMAW) {
        const int tw = w_cur; w_cur = w_n1; w_n1 = w_n2; w_n2 = w_far[0];
#pragma unroll
        for (int i = 0; i + 1 < R - 3; ++i) w_far[i] = w_far[i + 1];
        w_far[R - 4] = tw;
        if (T > 1 && INB == 2 && t == T - 1) { const int ti = in_cur; in_cur = in_n1; in_n1 = ti; }
      } else {
        const int tw = w_cur; w_cur = w_n1; w_n1 = w_n2; w_n2 = tw;
        if (T == 1 && INB == 3) { const int ti = in_cur; in_cur = in_n1; in_n1 = in_n2; in_n2 = ti; }
        if (T > 1 && INB == 2 && t == T - 1) { const int ti = in_cur; in_cur = in_n1; in_n1 = ti; }
      }
    }
    if ((T * G) & 1) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) { ah[0][mi] = ah[1][mi]; al[0][mi] = al[1][mi]; }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) { bh[0][ni] = bh[1][ni]; bl[0][ni] = bl[1][ni]; }
    }
  }

  }

  __builtin_amdgcn_s_setprio(3);
#ifdef BSR_STAMPS
  st2 = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue: identical to igemm_conv_kernel's (bias is already in the accumulator; LeakyReLU; NHWC raw-buffer stores) ----
  static_assert(TW == 32, "epilogue assumes one tile row per 32-pixel MFMA tile");
  constexpr int SX = TR ? 2 : 1;
  const size_t blk_pix = (size_t)img * p.Ho * p.Wo + (size_t)(SX * y0) * p.Wo + SX * x0;
  constexpr unsigned OB = OUT16 ? 2u : 4u;                       // bytes per output element
  const unsigned lane_out = ((unsigned)(SX * 4 * h) * (unsigned)p.out_cs + (unsigned)r) * OB;
  // OUT16: packed two-channel stores (pack_pair_f16) when every pixel's channel run starts 4-byte aligned; the odd lane writes the NEXT pixel of a pair
  const bool odd = (r & 1) != 0;
  const bool pk = OUT16 && BSR_H16_PACK_STORES && ((p.out_cs | p.out_coff) & 1) == 0;
  const unsigned lane_out_pk = ((unsigned)(SX * (4 * h + (odd ? 1 : 0))) * (unsigned)p.out_cs + (unsigned)(r & ~1)) * 2u;
  const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.out) + (blk_pix * p.out_cs + p.out_coff) * OB));
  // fp32 output as 16-byte stores (round 6): in the accumulator layout a lane holds ONE channel of 16 pixels — 16 dword stores per 32x32
  // tile, 128 per wave for the four parities of a transposed conv, ~60 cycles each at the issue (in-kernel stamps: 8-9 k of a
  // workgroup's 52 k cycles).  The tile goes through LDS ([32 pixels][32 channels] floats, 128-byte rows: a half-wave's dword writes and
  // the 16-byte reads of two rows are both conflict-free) and leaves as 4 instructions of 16 bytes per lane, eight lanes per pixel.  The
  // loop's LDS is dead behind its last barrier; a wave uses its own 4-KB region and LDS executes a wave's accesses in order.  Same values.
  if constexpr (!OUT16 && BSR_H16_WIDE_STORES && MI == 1 && WN == 1 && WM == 4 && C::SMEM_BYTES >= 4 * 4096) {
    if (((p.out_cs | p.out_coff | p.n_store) & 3) == 0) {
      float* reg = smem + wave * 1024;
      const int piece = lane & 7, pc = lane >> 3;                     // this lane's 16-byte piece (4 channels) of pixel column pc + 8 q
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int nt = n0 + ni * 32;
          f32x16 v = acc[ph][0][ni];
          if (p.act) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              const f32x2 y = leaky_relu2(f32x2{v[i], v[i + 1]});
              v[i] = y[0];
              v[i + 1] = y[1];
            }
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) reg[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = v[i];
          const unsigned v16 = nt + piece * 4 < p.n_store ? (unsigned)((SX * pc) * p.out_cs + piece * 4) * 4u : kLaneOff;
          const unsigned tile_off = (unsigned)((SX * wm + (TR ? (ph >> 1) : 0)) * p.Wo + (TR ? (ph & 1) : 0)) * (unsigned)p.out_cs + (unsigned)nt;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            typedef unsigned h16_u4 __attribute__((ext_vector_type(4)));
            const h16_u4 d = *reinterpret_cast<const h16_u4*>(reg + (8 * q + pc) * 32 + piece * 4);
            __builtin_amdgcn_raw_buffer_store_b128(d, orsrc, v16, (tile_off + (unsigned)(SX * 8 * q) * (unsigned)p.out_cs) * 4u, 0);
          }
        }
      return;
    }
  }
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int nt = n0 + (wn * NI + ni) * 32;
        const unsigned voff = nt + r < p.n_store ? lane_out : kLaneOff;
        const int ty = wm * MI + mi;
        const unsigned tile_off = (unsigned)((SX * ty + (TR ? (ph >> 1) : 0)) * p.Wo + (TR ? (ph & 1) : 0)) * (unsigned)p.out_cs + (unsigned)nt;
        f32x16 v = acc[ph][mi][ni];
        if (p.act) {
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const f32x2 y = leaky_relu2(f32x2{v[i], v[i + 1]});
            v[i] = y[0];
            v[i + 1] = y[1];
          }
        }
        if constexpr (OUT16) {                 // the fp16 activation pack converts here: range guard on what is about to be rounded
          float amax = 0.f;
#pragma unroll
          for (int i = 0; i < 16; i += 4) amax = amax4(f32x4{v[i], v[i + 1], v[i + 2], v[i + 3]}, amax);
          range_report(voff == kLaneOff ? 0.f : amax, p.range_flag);
        }
        if (OUT16 && pk) {
          const bool pair_ok = nt + (r | 1) < p.n_store;
          const unsigned voff_pk = pair_ok ? lane_out_pk : kLaneOff;
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const int k = SX * ((i & 3) + 8 * (i >> 2));
            const unsigned soff = (tile_off + (unsigned)k * (unsigned)p.out_cs) * 2u;
            __builtin_amdgcn_raw_buffer_store_b32(pack_pair_f16(v[i], v[i + 1], odd), orsrc, voff_pk, soff, 0);
          }
          if (voff != kLaneOff && !pair_ok) {       // the last channel of an odd channel count has no partner: its own 2-byte stores
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int k = SX * ((i & 3) + 8 * (i >> 2));
              __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (_Float16)v[i]), orsrc, voff, (tile_off + (unsigned)k * (unsigned)p.out_cs) * 2u, 0);
            }
          }
        } else {
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
#ifdef BSR_STAMPS
  unsigned long long st2b = __builtin_amdgcn_s_memtime();
  if (p.stamps != nullptr && lane == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long st3 = __builtin_amdgcn_s_memtime();
    unsigned long long* d = p.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 4;
    unsigned long long rt3 = __builtin_amdgcn_s_memrealtime();
    d[0] = st1 - st0; d[1] = st2 - st1; d[2] = ((rt3 - rt0) << 32) | (st2b - st2); d[3] = st3 - st2;
  }
#endif
}

template <int KH, int KW, int S, bool TR, int TH, int TW, int WM, int WN, int MI, int NI, int CC, int INB, int NSPLIT, int IO = 0>
inline hipError_t launch_igemm_h16(ConvArgs a, int batch, hipStream_t stream) {
  using C = H16Cfg<KH, KW, S, TR, TH, TW, WM, WN, MI, NI, CC, INB, NSPLIT>;
  auto kern = igemm_h16_kernel<KH, KW, S, TR, TH, TW, WM, WN, MI, NI, CC, INB, NSPLIT, IO>;
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (C::SMEM_BYTES > 48 * 1024 && (dev < 0 || !once.done[dev])) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  const int mh = TR ? a.H : a.Ho, mw = TR ? a.W : a.Wo;
  a.tiles_x = mw / TW;
  a.tiles_y = mh / TH;
  const int nblk = (a.n_store + C::BN - 1) / C::BN;
  a.n_blocks = nblk > 1 ? nblk : 0;
  dim3 grid(a.tiles_x * a.tiles_y * batch * nblk, 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), C::SMEM_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace bsr
