/* Host-side helper of the loaders' worker processes (libbsr_host.so, no GPU code): inflate of a zlib stream (RFC 1950 / 1951).
 *
 * The reference reads its inputs with cv2.imread (/root/reference/dataset.py:150-152, 621-623); the UCB items are ordinary compressed
 * PNG files, and with the scanline reconstruction in C (png_unfilter.c) zlib 1.2.11's inflate — 1.3 ms per 256x256 RGB file, two files
 * per item — was half of a UCB item's host time.  This is the usual fast form of the same algorithm: a 64-bit bit buffer refilled eight
 * bytes at a time, two-level decode tables (11 / 8 root bits, canonical codes, LSB-first), literal / length / distance entries that
 * carry their base value and extra-bit count, 8-byte match copies.  The format is RFC 1951's; nothing here is taken from zlib's sources.
 *
 * bsr_inflate_zlib(src, src_len, dst, dst_len): src must be readable for src_len + 16 bytes (the caller pads with zeros), dst writable for
 * dst_len + 16 bytes; the stream must inflate to EXACTLY dst_len bytes and carry a matching Adler-32.  Returns 0, or a negative
 * code: -1 header, -2 truncated input, -3 bad block type / stored length, -4 bad code lengths, -5 bad symbol / distance,
 * -6 output size mismatch, -7 Adler-32 mismatch.  Checked against zlib on every kind of block (tests/test_pngio.py). */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define LROOT 11
#define DROOT 8
#define LTABLE_MAX (1 << 15)          /* generous: 2^11 root + sub-tables; an over-subscribed set is rejected before it can overflow */
#define DTABLE_MAX (1 << 15)

/* table entry: bits 0-7 code length (bits to drop), 8-15 kind / extra bits, 16-31 value */
#define K_LIT 0x00
#define K_EOB 0x40
#define K_SUB 0x80                    /* value = sub-table offset, low 6 bits of kind = sub-table bits */
#define K_BASE 0x20                   /* length or distance symbol: low 5 bits of kind = extra bits, value = base */
#define ENTRY(len, kind, val) ((uint32_t)(len) | ((uint32_t)(kind) << 8) | ((uint32_t)(val) << 16))

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

static inline uint32_t rev_bits(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

/* what a symbol's entry looks like, apart from its length: kind 0 = literal / length alphabet, 1 = distances, 2 = code lengths */
static inline uint32_t sym_entry(int alphabet, int sym, int len)
{
    if (alphabet == 2) return ENTRY(len, K_LIT, sym);
    if (alphabet == 1) return sym < 30 ? ENTRY(len, K_BASE | kDistExtra[sym], kDistBase[sym]) : 0u;     /* 30, 31: invalid (entry 0 = error) */
    if (sym < 256) return ENTRY(len, K_LIT, sym);
    if (sym == 256) return ENTRY(len, K_EOB, 0);
    return sym < 286 ? ENTRY(len, K_BASE | kLenExtra[sym - 257], kLenBase[sym - 257]) : 0u;
}

/* Canonical Huffman decode table for `n` symbols of the given code lengths (0 = unused).  Returns the number of entries used, or -1
 * for an over-subscribed set.  Incomplete sets are allowed (RFC 1951 permits a single distance code; unused slots stay 0 = error). */
static int build_table(const uint8_t* lens, int n, int alphabet, int root, uint32_t* table, int table_max)
{
    int count[16] = {0}, maxlen = 0;
    for (int i = 0; i < n; ++i) { count[lens[i]]++; if (lens[i] > maxlen) maxlen = lens[i]; }
    count[0] = 0;
    uint32_t next[16];
    uint32_t code = 0;
    long left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return -1;
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    const int rsize = 1 << root;
    memset(table, 0, (size_t)rsize * sizeof(uint32_t));
    if (maxlen == 0) return rsize;
    int used = rsize;
    /* pass 1 (only when codes longer than the root exist): the longest code behind every root prefix */
    uint8_t sub_bits[1 << LROOT];
    if (maxlen > root) {
        memset(sub_bits, 0, (size_t)rsize);
        uint32_t nx[16];
        memcpy(nx, next, sizeof(nx));
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t rc = rev_bits(nx[l]++, l);
            if (l > root) {
                const uint32_t low = rc & (uint32_t)(rsize - 1);
                if (l - root > sub_bits[low]) sub_bits[low] = (uint8_t)(l - root);
            }
        }
        for (int low = 0; low < rsize; ++low)
            if (sub_bits[low]) {
                const int sz = 1 << sub_bits[low];
                if (used + sz > table_max) return -1;
                table[low] = ENTRY(root, K_SUB | sub_bits[low], used);
                memset(table + used, 0, (size_t)sz * sizeof(uint32_t));
                used += sz;
            }
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rc = rev_bits(next[l]++, l);
        if (l <= root) {
            const uint32_t e = sym_entry(alphabet, s, l);
            for (uint32_t i = rc; i < (uint32_t)rsize; i += 1u << l) table[i] = e;
        } else {
            const uint32_t low = rc & (uint32_t)(rsize - 1);
            const uint32_t root_e = table[low];
            const int sb = (int)((root_e >> 8) & 0x3F);
            uint32_t* sub = table + (root_e >> 16);
            const uint32_t e = sym_entry(alphabet, s, l - root);
            for (uint32_t i = rc >> root; i < (1u << sb); i += 1u << (l - root)) sub[i] = e;
        }
    }
    return used;
}

typedef struct {
    const uint8_t* src;
    size_t n, pos;          /* pos: next byte to load into the bit buffer */
    uint64_t buf;
    int cnt;                /* valid bits in buf */
} Bits;

/* At least 56 valid bits afterwards — as long as the input lasts.  src is readable 16 bytes past n (zero padding): a well-formed stream
 * ends at least four bytes (its Adler-32) before n and the buffer runs at most eight bytes ahead of what was consumed, so its refills
 * never stop; a stream that runs past its end gets zeros until pos passes n + 8, then no more bits: the decoder's loops return -2 as soon
 * as a refill leaves them fewer than 56 (one symbol pair consumes at most 48, so the count never goes negative). */
static inline void refill(Bits* b)
{
    if (b->cnt <= 56 && b->pos <= b->n + 8) {
        uint64_t v;
        memcpy(&v, b->src + b->pos, 8);
        b->buf |= v << b->cnt;
        const int take = (63 - b->cnt) >> 3;
        b->pos += (size_t)take;
        b->cnt += take * 8;
    }
}
#define PEEK(b, k) ((uint32_t)((b)->buf & ((1ull << (k)) - 1ull)))
#define DROP(b, k) do { (b)->buf >>= (k); (b)->cnt -= (k); } while (0)
/* bytes of input really consumed so far (bits still in the buffer do not count) */
static inline size_t consumed(const Bits* b) { return b->pos - (size_t)(b->cnt >> 3); }

#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
/* the same two sums 32 bytes per step (round 6: the scalar form was 14 % of a photograph's inflate): a += sum b_i; s += 32 a_before +
 * sum (32 - i) b_i, kept unreduced for a run of at most 5 536 bytes.  Compiled for AVX2 whatever the build's -march; used when the CPU has it. */
__attribute__((target("avx2"))) static uint32_t adler32_avx2(const uint8_t* p, size_t n)
{
    uint32_t a = 1, s = 0;
    const __m256i weights = _mm256_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m256i ones = _mm256_set1_epi16(1), zero = _mm256_setzero_si256();
    while (n >= 32) {
        size_t k = (n < 5536 ? n : 5536) & ~(size_t)31;        /* 173 steps: every 32-bit lane stays far below 2^31 */
        n -= k;
        __m256i vs1 = zero, vs2 = zero, vs3 = zero;            /* byte sums (4 x 64 bit), weighted sums (8 x 32 bit), sum of the byte sums BEFORE each step */
        const uint32_t steps = (uint32_t)(k / 32);
        for (uint32_t i = 0; i < steps; ++i, p += 32) {
            const __m256i b = _mm256_loadu_si256((const __m256i*)p);
            vs3 = _mm256_add_epi64(vs3, vs1);
            vs1 = _mm256_add_epi64(vs1, _mm256_sad_epu8(b, zero));
            vs2 = _mm256_add_epi32(vs2, _mm256_madd_epi16(_mm256_maddubs_epi16(b, weights), ones));
        }
        uint64_t t1[4], t3[4];
        uint32_t t2[8];
        _mm256_storeu_si256((__m256i*)t1, vs1);
        _mm256_storeu_si256((__m256i*)t3, vs3);
        _mm256_storeu_si256((__m256i*)t2, vs2);
        const uint64_t s1 = t1[0] + t1[1] + t1[2] + t1[3], s3 = t3[0] + t3[1] + t3[2] + t3[3];
        uint64_t s2 = 0;
        for (int i = 0; i < 8; ++i) s2 += t2[i];
        s = (uint32_t)(((uint64_t)s + (uint64_t)k * a + 32u * s3 + s2) % 65521u);
        a = (uint32_t)(((uint64_t)a + s1) % 65521u);
    }
    while (n--) { a += *p++; s += a; }                         /* fewer than 32 bytes: no overflow */
    return ((s % 65521u) << 16) | (a % 65521u);
}
#define BSR_HAVE_ADLER_AVX2 1
#endif

static uint32_t adler32(const uint8_t* p, size_t n)
{
#ifdef BSR_HAVE_ADLER_AVX2
    if (__builtin_cpu_supports("avx2")) return adler32_avx2(p, n);
#endif
    uint32_t a = 1, s = 0;
    while (n) {
        size_t k = n < 5552 ? n : 5552;                /* the largest run before a 32-bit sum can overflow */
        n -= k;
        while (k >= 16) {                              /* s += 16 a + sum (16 - i) p[i];  a += sum p[i]: two independent sums per block */
            uint32_t sum = 0, wsum = 0;
            for (int i = 0; i < 16; ++i) { sum += p[i]; wsum += (uint32_t)(16 - i) * p[i]; }
            s += 16u * a + wsum;
            a += sum;
            p += 16; k -= 16;
        }
        while (k--) { a += *p++; s += a; }
        a %= 65521u; s %= 65521u;
    }
    return (s << 16) | a;
}

int bsr_inflate_zlib(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len)
{
    if (src_len < 6) return -1;
    const unsigned cmf = src[0], flg = src[1];
    if ((cmf & 0x0F) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return -1;      /* deflate, window <= 32 K, no preset dictionary */
    Bits b = {src, src_len, 2, 0, 0};
    size_t out = 0;
    static _Thread_local uint32_t ltab_tls[LTABLE_MAX], dtab_tls[DTABLE_MAX];
    uint32_t* const ltab = ltab_tls;                          /* one address computation per call: a thread-local access in a shared library is a function call */
    uint32_t* const dtab = dtab_tls;
    /* pair[i]: what the next LROOT bits i decode to when they START with one or two whole literals — byte 0 / 1 = the literals,
     * byte 2 = bits to drop, byte 3 = how many (0: not a literal, take the general path).  Serial Huffman decoding is a chain of
     * dependent table look-ups, one per symbol; photographs are ~95 % literals of 4-9 bits, so most look-ups here yield two bytes. */
    static _Thread_local uint32_t pair_tls[1 << LROOT];
    uint32_t* const pair = pair_tls;
    int last;
    do {
        refill(&b);
        if (b.cnt < 56) return -2;
        last = (int)PEEK(&b, 1); DROP(&b, 1);
        const int type = (int)PEEK(&b, 2); DROP(&b, 2);
        if (type == 0) {                                        /* stored */
            DROP(&b, b.cnt & 7);                                /* to the byte boundary */
            refill(&b);
            if (b.cnt < 56) return -2;
            const uint32_t len = PEEK(&b, 16); DROP(&b, 16);
            const uint32_t nlen = PEEK(&b, 16); DROP(&b, 16);
            if ((len ^ nlen) != 0xFFFFu) return -3;
            size_t at = consumed(&b);
            if (at + len > src_len) return -2;
            if (out + len > dst_len) return -6;
            memcpy(dst + out, src + at, len);
            out += len;
            b.pos = at + len; b.buf = 0; b.cnt = 0;
            continue;
        }
        if (type == 3) return -3;
        if (type == 1) {                                        /* fixed codes (RFC 1951, 3.2.6) */
            uint8_t l[288 + 32];
            int i = 0;
            for (; i < 144; ++i) l[i] = 8;
            for (; i < 256; ++i) l[i] = 9;
            for (; i < 280; ++i) l[i] = 7;
            for (; i < 288; ++i) l[i] = 8;
            for (i = 0; i < 32; ++i) l[288 + i] = 5;
            if (build_table(l, 288, 0, LROOT, ltab, LTABLE_MAX) < 0 || build_table(l + 288, 32, 1, DROOT, dtab, DTABLE_MAX) < 0) return -4;
        } else {                                                /* dynamic codes */
            const int hlit = (int)PEEK(&b, 5) + 257; DROP(&b, 5);
            const int hdist = (int)PEEK(&b, 5) + 1; DROP(&b, 5);
            const int hclen = (int)PEEK(&b, 4) + 4; DROP(&b, 4);
            if (hlit > 286 || hdist > 30) return -4;
            uint8_t cl[19] = {0};
            for (int i = 0; i < hclen; ++i) { refill(&b); if (b.cnt < 56) return -2; cl[kClOrder[i]] = (uint8_t)PEEK(&b, 3); DROP(&b, 3); }
            uint32_t ctab[1 << 7];
            if (build_table(cl, 19, 2, 7, ctab, 1 << 7) < 0) return -4;
            uint8_t l[286 + 30 + 138];
            int i = 0;
            while (i < hlit + hdist) {
                refill(&b);
                if (b.cnt < 56) return -2;
                const uint32_t e = ctab[PEEK(&b, 7)];
                if ((e & 0xFF) == 0) return -4;
                DROP(&b, e & 0xFF);
                const int sym = (int)(e >> 16);
                if (sym < 16) { l[i++] = (uint8_t)sym; continue; }
                int rep, val = 0;
                if (sym == 16) { if (i == 0) return -4; val = l[i - 1]; rep = 3 + (int)PEEK(&b, 2); DROP(&b, 2); }
                else if (sym == 17) { rep = 3 + (int)PEEK(&b, 3); DROP(&b, 3); }
                else { rep = 11 + (int)PEEK(&b, 7); DROP(&b, 7); }
                if (i + rep > hlit + hdist) return -4;
                while (rep--) l[i++] = (uint8_t)val;
            }
            if (l[256] == 0) return -4;                         /* no end-of-block code */
            if (build_table(l, hlit, 0, LROOT, ltab, LTABLE_MAX) < 0 || build_table(l + hlit, hdist, 1, DROOT, dtab, DTABLE_MAX) < 0) return -4;
        }
        if (consumed(&b) > src_len) return -2;
        for (uint32_t i = 0; i < (1u << LROOT); ++i) {
            const uint32_t e1 = ltab[i];
            const uint32_t l1 = e1 & 0xFF;
            if ((e1 & 0xFF00u) != 0 || l1 == 0) { pair[i] = 0; continue; }
            const uint32_t e2 = ltab[i >> l1];                  /* its upper l1 index bits are zeros, not stream bits: only a code that fits below counts */
            const uint32_t l2 = e2 & 0xFF;
            if ((e2 & 0xFF00u) == 0 && l2 != 0 && l1 + l2 <= LROOT) pair[i] = (e1 >> 16) | ((e2 >> 16) << 8) | ((l1 + l2) << 16) | (2u << 24);
            else pair[i] = (e1 >> 16) | (l1 << 16) | (1u << 24);
        }
        for (;;) {                                              /* the symbols of one block */
            /* the literal loop: one branch-free refill (>= 56 bits), then up to four look-ups of 11 bits, each worth one or two literals
             * written as one 16-bit store (the second byte is overwritten by the next symbol when it was not one yet) */
            while (out + 10 <= dst_len && b.pos <= b.n + 8) {
                uint64_t v;
                memcpy(&v, b.src + b.pos, 8);
                b.buf |= v << b.cnt;
                b.pos += (size_t)((63 - b.cnt) >> 3);
                b.cnt |= 56;
                uint32_t pr;
#define PAIR_STEP() pr = pair[b.buf & ((1u << LROOT) - 1u)]; if ((pr >> 24) == 0) break; \
                    { const uint16_t two = (uint16_t)pr; memcpy(dst + out, &two, 2); } out += pr >> 24; b.buf >>= (pr >> 16) & 0xFF; b.cnt -= (int)((pr >> 16) & 0xFF)
                PAIR_STEP(); PAIR_STEP(); PAIR_STEP(); PAIR_STEP();
#undef PAIR_STEP
            }
            refill(&b);                                         /* the look-ups above may have used 33 of the 56 bits */
            if (b.cnt < 56) return -2;
            uint32_t e = ltab[PEEK(&b, LROOT)];
            if ((e >> 8) & K_SUB) { DROP(&b, LROOT); e = ltab[(e >> 16) + PEEK(&b, (e >> 8) & 0x3F)]; }
            const uint32_t kind = (e >> 8) & 0xFF;
            if ((e & 0xFF) == 0) return -5;
            DROP(&b, e & 0xFF);
            if (kind == K_LIT) {
                if (out + 3 > dst_len) {                        /* the last bytes of the output: one at a time */
                    if (out >= dst_len) return -6;
                    dst[out++] = (uint8_t)(e >> 16);
                    continue;
                }
                dst[out++] = (uint8_t)(e >> 16);
                /* two more literals from the bits already in the buffer (photographs are mostly literals; a root-table literal takes <= 11
                 * bits, >= 41 are left after the first symbol): no refill, no bounds check */
                uint32_t e2 = ltab[PEEK(&b, LROOT)];
                if ((e2 & 0xFF00u) == 0 && (e2 & 0xFF) != 0) {
                    DROP(&b, e2 & 0xFF);
                    dst[out++] = (uint8_t)(e2 >> 16);
                    e2 = ltab[PEEK(&b, LROOT)];
                    if ((e2 & 0xFF00u) == 0 && (e2 & 0xFF) != 0) {
                        DROP(&b, e2 & 0xFF);
                        dst[out++] = (uint8_t)(e2 >> 16);
                    }
                }
                continue;
            }
            if (kind == K_EOB) break;
            /* length + distance */
            const int lx = (int)(kind & 0x1F);
            const uint32_t len = (e >> 16) + PEEK(&b, lx); DROP(&b, lx);
            refill(&b);
            if (b.cnt < 56) return -2;
            uint32_t d = dtab[PEEK(&b, DROOT)];
            if ((d >> 8) & K_SUB) { DROP(&b, DROOT); d = dtab[(d >> 16) + PEEK(&b, (d >> 8) & 0x3F)]; }
            if ((d & 0xFF) == 0) return -5;
            DROP(&b, d & 0xFF);
            const int dx = (int)((d >> 8) & 0x1F);
            const size_t dist = (size_t)(d >> 16) + PEEK(&b, dx); DROP(&b, dx);
            if (dist > out) return -5;
            if (out + len > dst_len) return -6;
            uint8_t* o = dst + out;
            const uint8_t* f = o - dist;
            if (dist >= 8) {                                    /* 8 bytes at a time; may write up to 7 bytes past the match (dst has 16 spare) */
                for (uint32_t k = 0; k < len; k += 8) { uint64_t v; memcpy(&v, f + k, 8); memcpy(o + k, &v, 8); }
            } else if (dist == 1) {
                memset(o, f[0], len);
            } else {
                for (uint32_t k = 0; k < len; ++k) o[k] = f[k];
            }
            out += len;
            if (consumed(&b) > src_len) return -2;
        }
        if (consumed(&b) > src_len) return -2;
    } while (!last);
    if (out != dst_len) return -6;
    DROP(&b, b.cnt & 7);
    size_t at = consumed(&b);
    if (at + 4 > src_len) return -2;
    const uint32_t want = ((uint32_t)src[at] << 24) | ((uint32_t)src[at + 1] << 16) | ((uint32_t)src[at + 2] << 8) | (uint32_t)src[at + 3];
    return adler32(dst, dst_len) == want ? 0 : -7;
}
