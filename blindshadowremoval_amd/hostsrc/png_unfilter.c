/* Host-side helper of the loaders' worker processes (NOT part of libbsr_hip.so, no GPU code): PNG scanline reconstruction.
 *
 * The reference's loader reads every item with cv2.imread (/root/reference/dataset.py:150-152, 621-623); in this package the
 * workers decode with PIL, and for the 256x256 RGB files of the reference's own samples 2.0 of PIL's 2.2 ms per file is this
 * step (the zlib stream of a photograph is nearly stored: inflate takes 0.1 ms).  The filters are those of the PNG
 * specification (ISO/IEC 15948:2004, 9.2 "Filter types for filter method 0"): None, Sub, Up, Average, Paeth, all modulo 256,
 * bytes left of the first pixel and above the first scanline read as 0.
 *
 * bsr_png_unfilter(raw, h, rowbytes, bpp, out): raw = h scanlines of 1 filter-type byte + rowbytes filtered bytes (the
 * inflated IDAT stream of a non-interlaced image); out = h x rowbytes reconstructed bytes; bpp = bytes per complete pixel
 * (1 grey, 3 RGB, 4 RGBA at 8 bits).  Returns 0, or -(row+1) for a scanline with an undefined filter type.
 * Checked byte for byte against PIL on files of every filter type (tests/test_pngio.py). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int iabs(int v) { return v < 0 ? -v : v; }

static inline uint8_t paeth(int a, int b, int c)
{
    int pa = iabs(b - c), pb = iabs(a - c), pc = iabs(a + b - 2 * c);
    /* ties: a before b before c (the specification's order) */
    return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

#define PAETH_ROW(BPP)                                                                                   \
    static void paeth_row_##BPP(const uint8_t* in, const uint8_t* up, uint8_t* out, int n)             \
    {                                                                                                    \
        int a[BPP], c[BPP];                                                                              \
        for (int k = 0; k < BPP; ++k) { a[k] = 0; c[k] = 0; }                                            \
        int i = 0;                                                                                       \
        for (; i + BPP <= n; i += BPP)                                                                   \
            for (int k = 0; k < BPP; ++k) {                                                              \
                int b = up[i + k];                                                                       \
                int x = (in[i + k] + paeth(a[k], b, c[k])) & 255;                                        \
                out[i + k] = (uint8_t)x;                                                                 \
                a[k] = x; c[k] = b;                                                                      \
            }                                                                                            \
    }
PAETH_ROW(1)
PAETH_ROW(3)
PAETH_ROW(4)

#if defined(__SSE2__)
#include <emmintrin.h>
/* One PIXEL per step, its 3 or 4 channels in the 16-bit lanes of one register, the predictor chosen without a branch (the scalar
 * form mispredicts on photographs: 6 ns per byte; this one: ~1 ns).  pa = |b - c| does not depend on the pixel to the left, so the
 * chain from one pixel to the next is a - c -> |.| / + -> min -> compare -> select -> add. */
static inline __m128i load_px(const uint8_t* p, int bpp, int whole)
{
    uint32_t v = 0;
    if (whole) memcpy(&v, p, 4); else memcpy(&v, p, (size_t)bpp);
    return _mm_unpacklo_epi8(_mm_cvtsi32_si128((int)v), _mm_setzero_si128());
}
static inline __m128i abs16(__m128i v) { return _mm_max_epi16(v, _mm_sub_epi16(_mm_setzero_si128(), v)); }

static void paeth_row_simd(const uint8_t* in, const uint8_t* up, uint8_t* out, int n, int bpp)
{
    __m128i a = _mm_setzero_si128(), c = _mm_setzero_si128();
    const __m128i lo8 = _mm_set1_epi16(255);
    for (int i = 0; i + bpp <= n; i += bpp) {
        int whole = i + 4 <= n;                         /* 4 bytes may be read / written: the 4th is the next pixel's, rewritten by the next step */
        __m128i b = load_px(up + i, bpp, whole), x = load_px(in + i, bpp, whole);
        __m128i pa = _mm_sub_epi16(b, c), pb = _mm_sub_epi16(a, c);
        __m128i pc = abs16(_mm_add_epi16(pa, pb));
        pa = abs16(pa); pb = abs16(pb);
        __m128i m = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
        __m128i is_a = _mm_cmpeq_epi16(m, pa), is_b = _mm_cmpeq_epi16(m, pb);        /* ties: a, then b, then c */
        __m128i bc = _mm_or_si128(_mm_and_si128(is_b, b), _mm_andnot_si128(is_b, c));
        __m128i pred = _mm_or_si128(_mm_and_si128(is_a, a), _mm_andnot_si128(is_a, bc));
        c = b;
        a = _mm_and_si128(_mm_add_epi16(x, pred), lo8);
        uint32_t r = (uint32_t)_mm_cvtsi128_si32(_mm_packus_epi16(a, a));
        if (whole) memcpy(out + i, &r, 4); else memcpy(out + i, &r, (size_t)bpp);
    }
}
#define BSR_HAVE_PAETH_SIMD 1
#else
#define BSR_HAVE_PAETH_SIMD 0
#endif

/* Sub and Average with the pixel to the left carried in registers (reading it back from `out` costs a store-to-load forward per byte) */
#define SUB_AVG_ROWS(BPP)                                                                                \
    static void sub_row_##BPP(const uint8_t* in, uint8_t* out, int n)                                    \
    {                                                                                                    \
        uint8_t a[BPP];                                                                                  \
        for (int k = 0; k < BPP; ++k) a[k] = 0;                                                          \
        for (int i = 0; i + BPP <= n; i += BPP)                                                          \
            for (int k = 0; k < BPP; ++k) { a[k] = (uint8_t)(a[k] + in[i + k]); out[i + k] = a[k]; }     \
    }                                                                                                    \
    static void avg_row_##BPP(const uint8_t* in, const uint8_t* up, uint8_t* out, int n)                 \
    {                                                                                                    \
        unsigned a[BPP];                                                                                 \
        for (int k = 0; k < BPP; ++k) a[k] = 0;                                                          \
        for (int i = 0; i + BPP <= n; i += BPP)                                                          \
            for (int k = 0; k < BPP; ++k) { a[k] = (in[i + k] + ((a[k] + up[i + k]) >> 1)) & 255u; out[i + k] = (uint8_t)a[k]; } \
    }
SUB_AVG_ROWS(1)
SUB_AVG_ROWS(3)
SUB_AVG_ROWS(4)

static void paeth_row_any(const uint8_t* in, const uint8_t* up, uint8_t* out, int n, int bpp)
{
    for (int i = 0; i < n; ++i) {
        int a = i >= bpp ? out[i - bpp] : 0, c = i >= bpp ? up[i - bpp] : 0;
        out[i] = (uint8_t)(in[i] + paeth(a, up[i], c));
    }
}

#ifndef BSR_HOST_SHA
#define BSR_HOST_SHA "unbuilt"
#endif
static const char kHostShaTag[] = "BSR_HOST_SHA=" BSR_HOST_SHA;
const char* bsr_host_source_sha(void) { return kHostShaTag + 13; }

int bsr_png_unfilter(const uint8_t* raw, int h, int rowbytes, int bpp, uint8_t* out)
{
    if (h < 0 || rowbytes < 0 || bpp < 1) return -1;
    uint8_t* zero = (uint8_t*)calloc((size_t)rowbytes + 1, 1);
    if (!zero) return -1;
    const uint8_t* up = zero;
    for (int y = 0; y < h; ++y) {
        const uint8_t* in = raw + (size_t)y * (rowbytes + 1);
        uint8_t* o = out + (size_t)y * rowbytes;
        int ft = in[0];
        ++in;
        switch (ft) {
        case 0: memcpy(o, in, (size_t)rowbytes); break;
        case 1:
            if (bpp == 1) { sub_row_1(in, o, rowbytes); break; }
            if (bpp == 3 && rowbytes % 3 == 0) { sub_row_3(in, o, rowbytes); break; }
            if (bpp == 4 && rowbytes % 4 == 0) { sub_row_4(in, o, rowbytes); break; }
            for (int i = 0; i < rowbytes && i < bpp; ++i) o[i] = in[i];
            for (int i = bpp; i < rowbytes; ++i) o[i] = (uint8_t)(in[i] + o[i - bpp]);
            break;
        case 2:
            for (int i = 0; i < rowbytes; ++i) o[i] = (uint8_t)(in[i] + up[i]);
            break;
        case 3:
            if (bpp == 1) { avg_row_1(in, up, o, rowbytes); break; }
            if (bpp == 3 && rowbytes % 3 == 0) { avg_row_3(in, up, o, rowbytes); break; }
            if (bpp == 4 && rowbytes % 4 == 0) { avg_row_4(in, up, o, rowbytes); break; }
            for (int i = 0; i < rowbytes && i < bpp; ++i) o[i] = (uint8_t)(in[i] + (up[i] >> 1));
            for (int i = bpp; i < rowbytes; ++i) o[i] = (uint8_t)(in[i] + ((o[i - bpp] + up[i]) >> 1));
            break;
        case 4:
#if BSR_HAVE_PAETH_SIMD
            if ((bpp == 3 || bpp == 4) && rowbytes % bpp == 0) { paeth_row_simd(in, up, o, rowbytes, bpp); break; }
#endif
            if (bpp == 3 && rowbytes % 3 == 0) paeth_row_3(in, up, o, rowbytes);
            else if (bpp == 4 && rowbytes % 4 == 0) paeth_row_4(in, up, o, rowbytes);
            else if (bpp == 1) paeth_row_1(in, up, o, rowbytes);
            else paeth_row_any(in, up, o, rowbytes, bpp);
            break;
        default:
            free(zero);
            return -(y + 1);
        }
        up = o;
    }
    free(zero);
    return 0;
}
