// PNG files of the test loops' figure strips, built ON THE DEVICE (round 5).
//
// The reference's loops end every item with `cv2.imwrite` of an 8-bit RGB strip (/root/reference/utils.py:196-204, called from
// train_test_GSC.py:744-746, :889-890).  Rounds 2-4 encoded those strips on the host (pngio.py: Sub filter + zlib RLE, ~4 ms of CPU
// per 256x768 strip, ~9 ms per 256x1792 UCB strip): with the forward at 6 000+ images/s that encoder, not the GPU, set the loops'
// rate.  PNG is lossless, so any conforming file gives the reference's pixels back; this writer produces the simplest conforming
// file — filter type 0 on every scanline, the zlib stream as STORED deflate blocks — whose byte layout is a pure function of
// (H, W): every pixel byte has a fixed position, and the only data-dependent bytes are the zlib Adler-32 and the IDAT CRC-32, both
// of which reduce in parallel.  The host's share becomes one write() per file.
//
//   file = signature(8) | IHDR chunk(25) | IDAT length(4) "IDAT"(4) | zlib: 78 01 | blocks | adler32(4) | crc32(4) | IEND chunk(12)
//   block b = 5-byte header (BFINAL, LEN, ~LEN) + R whole scanlines (R = 65535 / (1 + 3W)), scanline = 00 | W x RGB
//
// png_rows_kernel: one wave per scanline.  The wave assembles the scanline's bytes of the FILE (with the block header / the chunk
// head in front of it where one starts there) in LDS, copies them out, and reduces its part of the two checksums:
//   * CRC-32 (reflected, poly 0xEDB88320): the register after a message is linear in (initial register, message), so
//     crc(A | B) = shift(crc(A), |B|) ^ crc_0(B) with shift(c, n) = c * x^(8n) mod P — zlib's crc32_combine, restated: multmodp / x2nmodp
//     below.  Every lane runs the table-driven loop over its piece (the stream's first piece starts from 0xFFFFFFFF, all others
//     from 0), shifts the result to the END of the checksummed stream and the pieces are XORed together: a wave reduction, then one
//     atomicXor per scanline — XOR is order-independent, so the result is deterministic.  Round 6 (rocprofv3: 106 -> 26-30 us per 16 strips, the
//     files byte for byte the same): the shift is TWO tabulated multiplications (lane: to the end of the scanline; row: to the end of the stream;
//     tables of the host in the kernel arguments) instead of ~20 bit-serial ones per lane — 70 of the 106 us by knock-out builds; the loop takes
//     four bytes per step (slice-by-4 tables, a lane's piece = whole dwords of LDS); the scanline's pixels sit on a 16-byte boundary of LDS, are
//     filled and copied out 16 bytes per lane (unaligned global stores); LDS is sized by the scanline (one round of workgroups).
//   * Adler-32 over the n raw bytes d_0 .. d_{n-1}: A = 1 + sum d_j, B = n + sum (n - j) d_j (mod 65521): two integer sums, 64-bit atomics.
// Round 6: ONE launch.  The workgroup whose arrival ticket says it is the file's LAST folds the accumulators, runs the 4 Adler bytes through the
// CRC, writes the fixed bytes — and clears accumulators and ticket again: the scratch is zero before and after every call (the caller
// zeroes it ONCE, when it allocates it).  Round 5's form was memset + rows + finish: its own knock-out builds put 0.054 of its 0.074-0.104
// ms into the dispatch of three dependent launches (profiles/HISTORY.md).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Diagnostic builds (scratch/png_knock.sh; wrong files, timing only): bit 0 = no CRC / Adler byte loop, 1 = no copy-out, 2 = no pixel fill,
// 3 = no shift to the end of the stream, 4 = no atomics / ticket / finish
#ifndef BSR_PNG_KNOCK
#define BSR_PNG_KNOCK 0
#endif

namespace bsr {

constexpr unsigned kCrcPoly = 0xEDB88320u;

// a(x) * b(x) mod P(x), bit-reflected representation (bit 31 = x^0): zlib crc32.c's multmodp
__host__ __device__ inline unsigned crc_multmodp(unsigned a, unsigned b) {
  unsigned m = 1u << 31, p = 0u;
  for (;;) {
    if (a & m) {
      p ^= b;
      if ((a & (m - 1u)) == 0u) break;
    }
    m >>= 1;
    b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
  }
  return p;
}

struct PngGeom {
  int H, W;                 // pixels
  int RB;                   // raw bytes per scanline: 1 + 3 W
  int R;                    // scanlines per stored block
  int nblocks;
  unsigned zlen;            // bytes of the zlib stream (IDAT data)
  unsigned file_bytes;
  unsigned ihdr_crc;
  unsigned x2n[32];         // x^(2^k) mod P
  // round 6: the shift of a lane's piece of a scanline to the end of the stream as TWO tabulated multiplications — x^(8 (bytes of the
  // scanline behind the piece)), a function of the lane, times x^(8 (bytes of the stream behind the scanline)), a function of the row —
  // instead of ~20 bit-serial ones per lane (rocprof + knock-out builds: 70 of the kernel's 110 us).  Tables of the host, in the kernel
  // arguments; rows beyond kPngRowTable use the bit-serial shift.
  int cb;                   // pixel bytes of a scanline per lane: 4 ceil(3 W / 256)
  unsigned lane_shift[64];
  unsigned row_shift[512];
};
constexpr int kPngRowTable = 512;

__host__ __device__ inline unsigned crc_update_byte(unsigned c, unsigned byte) {     // bitwise form (host, and the finish kernel's few bytes)
  c ^= byte;
  for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
  return c;
}

inline bool png_geometry(int H, int W, PngGeom* g) {
  if (H <= 0 || W <= 0 || (long long)3 * W + 1 > 16384 || H > 65535) return false;      // a scanline must fit the per-wave LDS segment
  // a loop encodes the same geometry thousands of times: the last one is kept per thread (the shift tables are ~330 polynomial multiplications)
  static thread_local PngGeom last;
  static thread_local bool have = false;
  if (have && last.H == H && last.W == W) { *g = last; return true; }
  struct Keep { PngGeom* g; ~Keep() { last = *g; have = true; } } keep{g};
  g->H = H; g->W = W;
  g->RB = 1 + 3 * W;
  g->R = 65535 / g->RB;
  g->nblocks = (H + g->R - 1) / g->R;
  g->zlen = 2u + 5u * (unsigned)g->nblocks + (unsigned)H * (unsigned)g->RB + 4u;
  g->file_bytes = 8u + 25u + 8u + g->zlen + 4u + 12u;
  const unsigned char ihdr[17] = {'I', 'H', 'D', 'R', (unsigned char)(W >> 24), (unsigned char)(W >> 16), (unsigned char)(W >> 8), (unsigned char)W,
                                  (unsigned char)(H >> 24), (unsigned char)(H >> 16), (unsigned char)(H >> 8), (unsigned char)H, 8, 2, 0, 0, 0};
  unsigned c = 0xFFFFFFFFu;
  for (int i = 0; i < 17; ++i) c = crc_update_byte(c, ihdr[i]);
  g->ihdr_crc = c ^ 0xFFFFFFFFu;
  unsigned p = 1u << 30;                                     // x^1
  g->x2n[0] = p;
  for (int k = 1; k < 32; ++k) g->x2n[k] = p = crc_multmodp(p, p);
  // x^(8 n) mod P on the host (crc_shift_bytes of the device, applied to x^0)
  auto x8n = [&](unsigned n) {
    unsigned c = 1u << 31, k = 3;
    while (n) {
      if (n & 1u) c = crc_multmodp(g->x2n[k & 31], c);
      n >>= 1;
      ++k;
    }
    return c;
  };
  const int nbp = 3 * W;                                     // pixel bytes of a scanline
  g->cb = 4 * ((nbp + 255) / 256);
  {
    // lane l covers PIXEL bytes [l cb, min((l + 1) cb, 3 W)) of its scanline (cb a multiple of 4: whole dwords of LDS, four bytes per CRC
    // step): behind it lie 3 W - min((l + 1) cb, 3 W) bytes of the scanline
    const unsigned step = x8n((unsigned)g->cb);
    int last = (nbp - 1) / g->cb;                            // the last lane with bytes; its piece may be short
    for (int l = 63; l >= 0; --l) {
      if (l >= last) g->lane_shift[l] = 1u << 31;            // nothing behind it: x^0
      else if (l == last - 1) g->lane_shift[l] = x8n((unsigned)(nbp - last * g->cb));
      else g->lane_shift[l] = crc_multmodp(g->lane_shift[l + 1], step);
    }
    // scanline y: the stream ("IDAT" + zlib stream without its Adler-32) continues for dist(y) bytes behind it; dist(H - 1) = 0 and one
    // scanline up adds RB bytes, plus the 5-byte header where a stored block starts in between
    const unsigned xa = x8n((unsigned)g->RB), xb = x8n((unsigned)g->RB + 5u);
    const int rows = H < kPngRowTable ? H : kPngRowTable;
    unsigned cur = 1u << 31;
    for (int y = H - 1; y >= 0; --y) {
      if (y < H - 1) cur = crc_multmodp(cur, ((y + 1) % g->R == 0) ? xb : xa);
      if (y < rows) g->row_shift[y] = cur;
    }
    for (int y = rows; y < kPngRowTable; ++y) g->row_shift[y] = 0u;
  }
  return true;
}

// c * x^(8 n) mod P: the CRC register c moved past n further (zero) bytes — zlib's x2nmodp(n, 3) folded into the product
__device__ inline unsigned crc_shift_bytes(const PngGeom& g, unsigned c, unsigned n) {
  unsigned k = 3;
  while (n) {
    if (n & 1u) c = crc_multmodp(g.x2n[k & 31], c);
    n >>= 1;
    ++k;
  }
  return c;
}

constexpr int kPngSegMax = 16384 + 16;                        // LDS bytes per wave at most: a scanline + the headers that may precede it
__host__ __device__ inline int png_seg_bytes(int RB) { return (RB + 11 + 15 + 15) & ~15; }     // LDS bytes per wave for this geometry: headers + scanline + the pad that aligns its pixels
constexpr int kPngWaves = 4;
constexpr int kPngTabBytes = 4096;
constexpr int kPngSub = 8;                                    // checksum sub-accumulators per file (workgroup b adds into b % 8: at most H / 32 atomics per address)
constexpr int kPngMaxFigs = 8;

// Source of a strip as FIGURES (round 5, bsr_png_encode_figs): the strip is n figures side by side, figure k a float32 [B][H][Wf] image
// of ch[k] = 1 or 3 channels whose pixels are ps[k] floats apart (a channel slice of a wider NHWC tensor is fine), optionally multiplied
// by a one-channel float32 image mul[k] (pixels mps[k] floats apart) and by scale[k]; a pixel's byte is round-half-even(clamp(v, 0, 1) * 255)
// — Logging.get_imgs / strips_on_device (fsrnet.py) for the whole strip, without the uint8 strip tensor or any of its elementwise passes.
struct PngFigs {
  const float* ptr[kPngMaxFigs];
  const float* mul[kPngMaxFigs];
  float scale[kPngMaxFigs];
  int ch[kPngMaxFigs], ps[kPngMaxFigs], mps[kPngMaxFigs];
  int n, Wf;                                                  // n == 0: the source is the uint8 strip
};

__device__ __forceinline__ unsigned char png_quantise(float v) {
#pragma clang fp contract(off)
  const float c = __builtin_fminf(__builtin_fmaxf(v, 0.f), 1.f) * 255.f;
  return (unsigned char)__builtin_rintf(c);
}

// the fixed bytes of a file and its two checksums, from the folded accumulators (one thread)
__device__ inline void png_finish_file(unsigned char* __restrict__ f, const PngGeom& g, unsigned long long sum_a, unsigned long long sum_b, unsigned crc_acc) {
  const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  for (int i = 0; i < 8; ++i) f[i] = sig[i];
  auto be32 = [&](unsigned off, unsigned v) { f[off] = (unsigned char)(v >> 24); f[off + 1] = (unsigned char)(v >> 16); f[off + 2] = (unsigned char)(v >> 8); f[off + 3] = (unsigned char)v; };
  be32(8, 13u);
  f[12] = 'I'; f[13] = 'H'; f[14] = 'D'; f[15] = 'R';
  be32(16, (unsigned)g.W);
  be32(20, (unsigned)g.H);
  f[24] = 8; f[25] = 2; f[26] = 0; f[27] = 0; f[28] = 0;     // 8 bits, truecolour, deflate, adaptive filtering, no interlace
  be32(29, g.ihdr_crc);
  be32(33, g.zlen);
  const unsigned long long n_raw = (unsigned long long)g.H * (unsigned long long)g.RB;
  const unsigned A = (unsigned)((1ull + sum_a % 65521ull) % 65521ull), Bv = (unsigned)((n_raw % 65521ull + sum_b % 65521ull) % 65521ull);
  const unsigned adler = (Bv << 16) | A;
  const unsigned tail = 41u + g.zlen - 4u;                   // file offset of the Adler-32
  be32(tail, adler);
  unsigned c = crc_acc;
  for (int i = 0; i < 4; ++i) c = crc_update_byte(c, f[tail + i]);
  be32(tail + 4, c ^ 0xFFFFFFFFu);
  be32(tail + 8, 0u);
  f[tail + 12] = 'I'; f[tail + 13] = 'E'; f[tail + 14] = 'N'; f[tail + 15] = 'D';
  be32(tail + 16, 0xAE426082u);
}

// pixels: [B][H][W][3] uint8 (device), or figs.n > 0.  out: B files, `stride` bytes apart.  acc: [B][kPngSub][4] 64-bit words, ZERO on entry and on exit:
// {sum d, sum (n - j) d, crc xor, arrival ticket (sub-accumulator 0 only)}; a workgroup reduces its four scanlines in LDS and adds ONCE, into sub-accumulator blockIdx.x % 8
// (one atomic per scanline into one address per file serialised 768 atomics per file: 0.11 ms per 16 strips, most of this kernel)
__global__ __launch_bounds__(256) void png_rows_kernel(const unsigned char* __restrict__ pixels, unsigned char* __restrict__ out, size_t stride,
                                                       unsigned long long* __restrict__ acc, PngGeom g, PngFigs figs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char png_smem[];
  __shared__ unsigned long long s_red[kPngWaves][3];
  __shared__ int s_last;
  unsigned* s_tab = reinterpret_cast<unsigned*>(png_smem);                       // 4 x 256-entry CRC tables (slice by 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* seg = png_smem + kPngTabBytes + (size_t)wave * (size_t)png_seg_bytes(g.RB);      // (sized by the scanline: a 768-pixel strip leaves room for ~14 workgroups per CU, the maximum for 2)
  {
    // slice-by-4 tables: T0 = the byte table; T_{k+1}[i] = T_k[i] moved past one more zero byte — four bytes per step instead of one, the
    // step's four look-ups independent of each other (the byte loop was a chain of 37-84 dependent LDS round trips per lane)
    unsigned c = (unsigned)tid;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
    s_tab[tid] = c;
    __syncthreads();
    for (int k = 1; k < 4; ++k) {
      c = s_tab[c & 0xFFu] ^ (c >> 8);
      s_tab[256 * k + tid] = c;
    }
  }
  const int item = blockIdx.y;
  const int y = blockIdx.x * kPngWaves + wave;
  const bool live = y < g.H;
  // the scanline's place in the file, and what precedes it inside this wave's segment
  int hdr = 0;                                               // bytes of the segment in front of the scanline
  unsigned file_off = 0;                                     // file offset of the segment's first byte
  if (live) {
    const int blk = y / g.R;
    const unsigned row_off = 8u + 25u + 8u + 2u + 5u * (unsigned)(blk + 1) + (unsigned)y * (unsigned)g.RB;
    if (y % g.R == 0) hdr = 5;
    if (y == 0) hdr = 5 + 2 + 4;                             // "IDAT" | 78 01 | block header
    file_off = row_off - (unsigned)hdr;
    seg += (16 - ((hdr + 1) & 15)) & 15;                     // the scanline's first PIXEL byte (segment byte hdr + 1) on a 16-byte boundary of LDS
    if (lane == 0) {
      int o = 0;
      if (y == 0) {
        seg[0] = 'I'; seg[1] = 'D'; seg[2] = 'A'; seg[3] = 'T'; seg[4] = 0x78; seg[5] = 0x01;
        o = 6;
      }
      if (hdr) {
        const int rows = min(g.R, g.H - blk * g.R);
        const unsigned len = (unsigned)rows * (unsigned)g.RB;
        seg[o] = (blk == g.nblocks - 1) ? 1 : 0;             // BFINAL, BTYPE = 00 (stored)
        seg[o + 1] = (unsigned char)len; seg[o + 2] = (unsigned char)(len >> 8);
        seg[o + 3] = (unsigned char)~len; seg[o + 4] = (unsigned char)(~len >> 8);
      }
      seg[hdr] = 0;                                          // filter type 0 (None)
    }
    const unsigned char* src = pixels + ((size_t)item * g.H + y) * (size_t)(3 * g.W);
    const int nb = 3 * g.W;
    if (BSR_PNG_KNOCK & 4) {
    } else if (figs.n > 0) {
      unsigned char* d = seg + hdr + 1;
      const size_t rowpix = ((size_t)item * g.H + y) * (size_t)figs.Wf;
      for (int k = 0; k < figs.n; ++k) {
        const float* fp = figs.ptr[k];
        const float* mp = figs.mul[k];
        const float sc = figs.scale[k];
        const int ch3 = figs.ch[k] == 3;
        const bool wide = ch3 && (reinterpret_cast<uintptr_t>(fp) & 15) == 0 && (figs.ps[k] & 3) == 0;      // a channel slice of an NHWC tensor: one 16-byte load per pixel
        for (int x = lane; x < figs.Wf; x += 64) {
#pragma clang fp contract(off)
          const float* px = fp + (rowpix + x) * (size_t)figs.ps[k];
          const float m = mp != nullptr ? mp[(rowpix + x) * (size_t)figs.mps[k]] : 1.f;
          float v0, v1, v2;
          if (wide) {
            typedef float png_f4 __attribute__((ext_vector_type(4)));
            const png_f4 q = *reinterpret_cast<const png_f4*>(px);
            v0 = q[0]; v1 = q[1]; v2 = q[2];
          } else {
            v0 = px[0]; v1 = ch3 ? px[1] : v0; v2 = ch3 ? px[2] : v0;
          }
          if (mp != nullptr) { v0 = v0 * m; v1 = v1 * m; v2 = v2 * m; }
          if (sc != 1.f) { v0 = v0 * sc; v1 = v1 * sc; v2 = v2 * sc; }
          unsigned char* o = d + 3 * (k * figs.Wf + x);
          o[0] = png_quantise(v0); o[1] = png_quantise(v1); o[2] = png_quantise(v2);
        }
      }
    } else if (((3 * g.W) & 15) == 0 && (reinterpret_cast<uintptr_t>(pixels) & 15) == 0) {
      typedef unsigned png_u4 __attribute__((ext_vector_type(4)));
      const png_u4* src16 = reinterpret_cast<const png_u4*>(src);
      png_u4* d16 = reinterpret_cast<png_u4*>(seg + hdr + 1);
      for (int i = lane; i < nb / 16; i += 64) d16[i] = src16[i];
    } else if (((3 * g.W) & 3) == 0 && (reinterpret_cast<uintptr_t>(pixels) & 3) == 0) {
      const unsigned* src4 = reinterpret_cast<const unsigned*>(src);
      unsigned* d4 = reinterpret_cast<unsigned*>(seg + hdr + 1);
      for (int i = lane; i < nb / 4; i += 64) d4[i] = src4[i];
    } else {
      for (int i = lane; i < nb; i += 64) seg[hdr + 1 + i] = src[i];
    }
  }
  __syncthreads();
  if (lane == 0) { s_red[wave][0] = 0ull; s_red[wave][1] = 0ull; s_red[wave][2] = 0ull; }
  if (live) {
  const int seg_len = hdr + g.RB;
  // copy the segment to the file image
  unsigned char* dst = out + (size_t)item * stride + file_off;
  if (!(BSR_PNG_KNOCK & 2)) {
    // 16-byte pieces on LDS's boundaries (the pixels' alignment), stored to whatever byte address of the file they belong to (global memory takes
    // unaligned stores on gfx950); the bytes in front of the first and behind the last whole piece go one per lane
    typedef unsigned png_u4 __attribute__((ext_vector_type(4)));
    const int head = min((hdr + 1) & 15, seg_len);          // segment bytes in front of the first 16-byte boundary
    const int npiece = (seg_len - head) / 16, tail0 = head + 16 * npiece;
    if (lane < head) dst[lane] = seg[lane];
    if (lane < seg_len - tail0) dst[tail0 + lane] = seg[tail0 + lane];
    for (int i = lane; i < npiece; i += 64) {
      const png_u4 v = *reinterpret_cast<const png_u4*>(seg + head + 16 * i);
      __builtin_memcpy(dst + head + 16 * i, &v, 16);
    }
  }
  // checksums: lane l owns PIXEL bytes [l cb, (l + 1) cb) of the scanline — whole dwords of LDS — and lane 0 also what stands in front of them
  // in the segment (headers, filter byte), so that the bytes of the scanline behind a lane's piece depend on the lane only
  const int nbp = 3 * g.W, px0 = hdr + 1;                      // segment byte of the first pixel byte
  const int p0 = min(lane * g.cb, nbp), p1 = min((lane + 1) * g.cb, nbp);
  const int b0 = lane == 0 ? 0 : px0 + p0, b1 = px0 + p1;
  unsigned c = (y == 0 && lane == 0) ? 0xFFFFFFFFu : 0u;     // the checksummed stream starts with "IDAT"
  unsigned long long sa = 0, sb = 0;
  const unsigned long long n_raw = (unsigned long long)g.H * (unsigned long long)g.RB;
  const long long raw0 = (long long)y * g.RB - hdr;         // raw index of segment byte 0 (header bytes are not raw data)
  auto byte_step = [&](int i) {
    const unsigned d = seg[i];
    c = s_tab[(c ^ d) & 0xFFu] ^ (c >> 8);
    if (i >= hdr) {
      sa += d;
      sb += (n_raw - (unsigned long long)(raw0 + i)) * d;
    }
  };
  if (!(BSR_PNG_KNOCK & 1)) {
    if (lane == 0)
      for (int i = 0; i < px0; ++i) byte_step(i);
    int i = px0 + p0;
    for (; i + 4 <= b1; i += 4) {
      const unsigned w = *reinterpret_cast<const unsigned*>(seg + i);      // (px0 is on a 16-byte boundary of LDS, p0 a multiple of 4)
      c ^= w;
      c = s_tab[768 + (c & 0xFFu)] ^ s_tab[512 + ((c >> 8) & 0xFFu)] ^ s_tab[256 + ((c >> 16) & 0xFFu)] ^ s_tab[c >> 24];
      const unsigned d0 = w & 0xFFu, d1 = (w >> 8) & 0xFFu, d2 = (w >> 16) & 0xFFu, d3 = w >> 24;
      const unsigned long long wt = n_raw - (unsigned long long)(raw0 + i);      // weight of the dword's first byte; the next ones weigh one less each
      sa += d0 + d1 + d2 + d3;
      sb += wt * (d0 + d1 + d2 + d3) - (unsigned long long)(d1 + 2u * d2 + 3u * d3);
    }
    for (; i < b1; ++i) byte_step(i);
  }
  // shift this piece to the end of the checksummed stream (just before the Adler bytes): stream = "IDAT" + zlib stream - adler
  const unsigned stream_pos = file_off - 37u + (unsigned)b1;                     // bytes of the stream up to and including this piece ("IDAT" sits at file offset 37)
  const unsigned stream_end = 4u + g.zlen - 4u;
  const bool tabled = g.H <= kPngRowTable;                    // (uniform)
  unsigned contrib = 0u;
  if (b1 > b0) {
    if (BSR_PNG_KNOCK & 8) contrib = c + stream_end - stream_pos;
    else if (tabled) contrib = crc_multmodp(g.lane_shift[lane], c);                 // to the end of the scanline ...
    else contrib = crc_shift_bytes(g, c, stream_end - stream_pos);
  }
  for (int o = 32; o >= 1; o >>= 1) {
    contrib ^= __shfl_xor(contrib, o);
    sa += __shfl_xor(sa, o);
    sb += __shfl_xor(sb, o);
  }
  if (tabled && lane == 0 && !(BSR_PNG_KNOCK & 8)) contrib = crc_multmodp(g.row_shift[y], contrib);      // ... and the scanline to the end of the stream
  if (lane == 0) { s_red[wave][0] = sa; s_red[wave][1] = sb; s_red[wave][2] = contrib; }
  }
  __syncthreads();
  if (tid == 0 && !(BSR_PNG_KNOCK & 16)) {
    unsigned long long ra = 0, rb = 0;
    unsigned rc = 0;
#pragma unroll
    for (int w = 0; w < kPngWaves; ++w) { ra += s_red[w][0]; rb += s_red[w][1]; rc ^= (unsigned)s_red[w][2]; }
    // reduced mod 65521 HERE: unreduced, the B term (n_raw - j) d summed over a whole file passes 2^64 from n_raw ~ 3.8e8 bytes on, and
    // 2^64 mod 65521 is not 0 (png_geometry admits files of 1.07e9 raw bytes); a workgroup's own sum stays below 2^53
    unsigned long long* base = acc + (size_t)item * kPngSub * 4;
    unsigned long long* a = base + (blockIdx.x % kPngSub) * 4;
    atomicAdd(a, ra % 65521ull);
    atomicAdd(a + 1, rb % 65521ull);
    atomicXor(reinterpret_cast<unsigned*>(a + 2), rc);
    // arrival ticket (word 3 of sub-accumulator 0).  Everything the last arriver needs from the others travels in ATOMICS, which are performed
    // at the device's coherence point: this thread's adds have been performed once vmcnt is 0 (it counts atomics), so the ticket add that
    // follows is ordered behind them, and the last arriver's exchanges see every workgroup's contribution.  No release / acquire FENCE: a
    // release writes back the XCD's dirty L2 lines — the 9 MB of file bytes these workgroups have just stored — once per workgroup (measured:
    // 0.105 instead of 0.074 ms per 16 strips); the file bytes themselves need no hand-off, nobody in this launch reads them.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ticket = __hip_atomic_fetch_add(base + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == (unsigned long long)gridDim.x - 1ull ? 1 : 0;
  }
  __syncthreads();
  if (!(BSR_PNG_KNOCK & 16) && s_last && wave == 0) {
    // fold the 8 x {sum, sum, xor} sub-accumulators and clear them (and the ticket) for the next call: 32 lanes, one atomic exchange each —
    // as a chain in one thread the 25 round trips were 0.03 ms of the call
    unsigned long long* base = acc + (size_t)item * kPngSub * 4;
    unsigned long long v = 0ull;
    if (lane < 4 * kPngSub) v = __hip_atomic_exchange(base + lane, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int word = lane & 3;
    unsigned long long f0 = word == 0 ? v : 0ull, f1 = word == 1 ? v : 0ull, f2 = word == 2 ? v : 0ull;
    for (int o = 16; o >= 1; o >>= 1) {
      f0 += __shfl_xor(f0, o);
      f1 += __shfl_xor(f1, o);
      f2 ^= __shfl_xor(f2, o);
    }
    if (lane == 0) png_finish_file(out + (size_t)item * stride, g, f0, f1, (unsigned)f2);
  }
}

inline hipError_t launch_png_encode(const unsigned char* pixels, const PngFigs& figs, int B, const PngGeom& g, unsigned char* out, size_t stride,
                                    unsigned long long* acc, hipStream_t stream) {
  const int smem_max = kPngTabBytes + kPngWaves * kPngSegMax, smem = kPngTabBytes + kPngWaves * png_seg_bytes(g.RB);
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(png_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem_max);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  hipLaunchKernelGGL(png_rows_kernel, dim3((unsigned)((g.H + kPngWaves - 1) / kPngWaves), (unsigned)B), dim3(256), smem, stream, pixels, out, stride, acc, g, figs);
  return hipGetLastError();
}

}  // namespace bsr
