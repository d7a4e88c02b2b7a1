// Input preparation on the device: the per-sample numpy / matplotlib work of the reference's test loaders
// (/root/reference/dataset.py:619-638 `parse_fn_test_FFHQ`, :148-170 `parse_fn_test`; helpers /root/reference/utils.py:356-433
// `face_crop_and_resize`, :255-276 `generate_face_region`; /root/reference/warp.py:194-232 `generate_offset_map`, `generate_uv_map`)
// for a whole batch of rows at once.  The host keeps what is tiny and irregular — PNG decode, the crop box, the three Delaunay
// triangulations of <= 101 points (qhull via matplotlib.tri, exactly the reference's call) and their plane coefficients
// (`Triangulation.calculate_plane_coefficients`, what LinearTriInterpolator evaluates) — and ships it as ONE blob; the 65 536-point
// evaluations (seven interpolated channels, the hull mask and its 5x5 Gaussian, the bilinear crop-resize of image + ground truth) run
// here in float64 with the reference's operation order, fp contraction off, and land as the packed [B,S,S,16] float32 tensor the
// generator's callers split (channel layout: img3, gt3, uvm3, reg_in3, reg_out3, face1 — SURVEY.md Appendix D).
//
// Point location is brute force over the mesh's triangles (<= 200): per triangle three edge functions l_i = A_i x + B_i y + C_i
// (normalised barycentrics, built on the host), the triangle with the largest min(l_i) wins, inside = that value >= -1e-12 (grid
// points ON a hull edge — the offset meshes' anchors sit on the image border — count as inside, as in matplotlib's trifinder).
// The interpolant is continuous across edges, so which of two triangles sharing an edge wins changes the result by rounding only.
//
// Round 5: a workgroup is a 16x16-pixel BLOCK (it was 256 consecutive pixels = one image row at S = 256) and only walks the triangles
// that can touch it: thread k tests triangle k against the block's bounding box — an edge function whose maximum over the box is below
// -1e-9 is negative on every pixel of it, so that triangle can neither be `inside` (>= -1e-12) for one of them nor beat one that is —
// and the survivors are packed into LDS in their original order (ballot + prefix: the first of equal candidates still wins).  ~10
// triangles per block instead of ~160 per mesh; the same arithmetic per (pixel, triangle), the same winner, the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace bsr {

constexpr int kPrepTriDoubles = 18;     // 9 edge-function doubles + 3 channels x (a, b, c) plane coefficients
constexpr int kPrepMaxTri = 256;

struct PrepRow {            // one row of the batch; offsets are bytes from the blob start
  int64_t img_off, gt_off;  // RGB8 images [h][w][3] (gt_off == img_off when there is no separate ground truth)
  int32_t h, w;
  int32_t box[4];           // crop box x0, y0, x1, y1 in image pixels; may leave the image (zero extension, utils.py:414-425)
  int64_t tri_off[4];       // meshes: 0 uv map (3 channels), 1 reg_in (2), 2 reg_out (2), 3 face hull (1); kPrepTriDoubles doubles per triangle
  int32_t ntri[4];
};

// kernel 1: everything but the blur.  grid (S*S / 256, B); block 256.  hull: [B][S][S] raw hull mask (0 / 1) for kernel 2.
__global__ __launch_bounds__(256) void prep_rows_kernel(const unsigned char* __restrict__ blob, const PrepRow* __restrict__ rows,
                                                        const double* __restrict__ grid, int S, float* __restrict__ out, float* __restrict__ hull) {
  __shared__ double s_tri[kPrepMaxTri * kPrepTriDoubles];
  __shared__ int s_cnt[4];
  const PrepRow& row = rows[blockIdx.y];                        // read in place (wave-uniform scalar loads): a private copy indexed by the mesh number would live in scratch
  const int bpr = S / 16;                                       // S * S % 256 == 0 <=> S % 16 == 0 (checked by bsr_prep_rows)
  const int by = blockIdx.x / bpr, bx = blockIdx.x % bpr;
  const int oy = by * 16 + (threadIdx.x >> 4), ox = bx * 16 + (threadIdx.x & 15);
  const int pix = oy * S + ox;
  float* o = out + ((size_t)blockIdx.y * S * S + pix) * 16;

  // ---- crop + INTER_LINEAR resize of image and ground truth (dataset.resize_linear: float64, (1-w) a + w b per axis, x first) ----
  {
    const int n = row.box[2] - row.box[0];                        // the crop is square: side 2 * int(length)
    const double scale = (double)n / (double)S;
    auto axis = [&](int oidx, int& i0, int& i1, double& wgt) {
      double src = ((double)oidx + 0.5) * scale - 0.5;
      src = src > 0.0 ? src : 0.0;
      int f = (int)floor(src);
      i0 = f < n - 1 ? f : n - 1;
      i1 = i0 + 1 < n - 1 ? i0 + 1 : n - 1;
      wgt = src - (double)i0;
    };
    int y0, y1, x0, x1;
    double wy, wx;
    axis(oy, y0, y1, wy);
    axis(ox, x0, x1, wx);
    auto tap = [&](const unsigned char* im, int cy, int cx, int c) -> double {      // crop pixel (cy, cx): image pixel or 0 outside
      const int iy = cy + row.box[1], ix = cx + row.box[0];
      if (iy < 0 || iy >= row.h || ix < 0 || ix >= row.w) return 0.0;
      return (double)im[((size_t)iy * row.w + ix) * 3 + c] / 255.0;
    };
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const unsigned char* im = blob + (which ? row.gt_off : row.img_off);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double v = 0.0;
        if (n > 0) {
          const double top = tap(im, y0, x0, c) * (1.0 - wx) + tap(im, y0, x1, c) * wx;
          const double bot = tap(im, y1, x0, c) * (1.0 - wx) + tap(im, y1, x1, c) * wx;
          v = top * (1.0 - wy) + bot * wy;
        }
        o[which * 3 + c] = (float)v;
      }
    }
  }

  // ---- the four meshes ----
  const double px = grid[ox], py = grid[oy];                      // np.meshgrid(linspace(0,1,S), linspace(0,1,S)): x varies along columns
#pragma unroll 1
  for (int m = 0; m < 4; ++m) {
    const int nt = row.ntri[m] < kPrepMaxTri ? row.ntri[m] : kPrepMaxTri;
    const double* t = reinterpret_cast<const double*>(blob + row.tri_off[m]);
    __syncthreads();                                              // the previous mesh's triangles are no longer read
    // triangle threadIdx.x against this block's box [gx0, gx1] x [gy0, gy1] (grid is increasing): keep it unless an edge function is
    // below -1e-9 on the whole box (its maximum over a box is at a corner)
    const double gx0 = grid[bx * 16], gx1 = grid[bx * 16 + 15], gy0 = grid[by * 16], gy1 = grid[by * 16 + 15];
    bool keep = false;
    double e9[9];
    if ((int)threadIdx.x < nt) {
#pragma unroll
      for (int j = 0; j < 9; ++j) e9[j] = t[threadIdx.x * kPrepTriDoubles + j];
      keep = true;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double mx = fmax(e9[3 * i] * gx0, e9[3 * i] * gx1), my = fmax(e9[3 * i + 1] * gy0, e9[3 * i + 1] * gy1);
        keep = keep && ((mx + my) + e9[3 * i + 2] >= -1.0e-9);
      }
    }
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wv] = __popcll(bal);
    __syncthreads();
    int pos = __popcll(bal & ((1ull << lane) - 1ull));
    int nkeep = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = s_cnt[w];
      pos += w < wv ? c : 0;
      nkeep += c;
    }
    if (keep) {
      double* d = s_tri + pos * kPrepTriDoubles;
#pragma unroll
      for (int j = 0; j < 9; ++j) d[j] = e9[j];
#pragma unroll
      for (int j = 9; j < kPrepTriDoubles; ++j) d[j] = t[threadIdx.x * kPrepTriDoubles + j];
    }
    __syncthreads();
    int best = -1;
    double best_l = -1.0e300;
    for (int k = 0; k < nkeep; ++k) {
      const double* e = s_tri + k * kPrepTriDoubles;
      const double l0 = (e[0] * px + e[1] * py) + e[2];
      const double l1 = (e[3] * px + e[4] * py) + e[5];
      const double l2 = (e[6] * px + e[7] * py) + e[8];
      const double lm = fmin(l0, fmin(l1, l2));
      if (lm > best_l) { best_l = lm; best = k; }
    }
    const bool inside = best >= 0 && best_l >= -1.0e-12;
    const double* cf = s_tri + (best >= 0 ? best : 0) * kPrepTriDoubles + 9;
    // LinearTriInterpolator: z = a x + b y + c, evaluated left to right as numpy does
    const double z0 = (cf[0] * px + cf[1] * py) + cf[2];
    const double z1 = (cf[3] * px + cf[4] * py) + cf[5];
    const double z2 = (cf[6] * px + cf[7] * py) + cf[8];
    const double nan = __builtin_nan("");
    if (m == 0) {                       // uv map: np.nan_to_num -> 0 outside the landmark hull (warp.py:231)
      o[6] = inside ? (float)z0 : 0.f;
      o[7] = inside ? (float)z1 : 0.f;
      o[8] = inside ? (float)z2 : 0.f;
    } else if (m == 1 || m == 2) {      // offset maps: [my, mx, mx * 0], NOT nan_to_num'ed (warp.py:212-213; the anchors cover the square)
      const double my = inside ? z0 : nan, mx = inside ? z1 : nan;
      o[6 + 3 * m] = (float)my;
      o[7 + 3 * m] = (float)mx;
      o[8 + 3 * m] = (float)(mx * 0.0);
    } else {                            // face hull: interpolated x coordinate > 0 (utils.py:272-273; nan -> 0 -> false)
      hull[(size_t)blockIdx.y * S * S + pix] = (inside && z0 > 0.0) ? 1.f : 0.f;
    }
  }
}

// kernel 2: cv2.GaussianBlur(mask, (5,5), 0) = [1,4,6,4,1]/16 separable, BORDER_REFLECT_101, columns first then rows as
// dataset.gaussian_blur5 sums them (float64) -> channel 15
__global__ __launch_bounds__(256) void prep_blur_kernel(const float* __restrict__ hull, int S, float* __restrict__ out) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int oy = pix / S, ox = pix % S;
  const float* hm = hull + (size_t)blockIdx.y * S * S;
  const double k[5] = {1.0 / 16.0, 4.0 / 16.0, 6.0 / 16.0, 4.0 / 16.0, 1.0 / 16.0};
  auto refl = [&](int i) { return i < 0 ? -i : (i >= S ? 2 * S - 2 - i : i); };
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {               // outer sum over rows of the horizontally blurred image: sum_i k[i] * tmp[y + i - 2]
    const int yy = refl(oy + i - 2);
    double tmp = 0.0;
#pragma unroll
    for (int j = 0; j < 5; ++j) tmp = tmp + k[j] * (double)hm[(size_t)yy * S + refl(ox + j - 2)];
    acc = acc + k[i] * tmp;
  }
  out[((size_t)blockIdx.y * S * S + pix) * 16 + 15] = (float)acc;
}

// ---- PNG scanline reconstruction on the device (round 6) ----
// The loaders' workers inflate a PNG file and stop there: the FILTERED scanlines (RFC 2083 section 6: per row one filter-type byte, then
// w * c bytes, each the difference to a predictor built from the pixel to the left, the pixel above and the one above-left) travel in the
// item's ring slot as they are, and this kernel reconstructs the RGB8 image the preparation kernel reads — what hostsrc/png_unfilter.c
// did in the worker (17 % of a worker's time per item).  A row depends on the row above and, for the Average / Paeth filters, a pixel on
// its left neighbour, so the parallelism is the anti-diagonal: thread y owns row y and at step t reconstructs pixel x = t - y; the pixel
// above (thread y - 1, step t - 1) comes through a double-buffered LDS word per thread, left and above-left are the thread's own registers.
// One workgroup per image of at most 256 rows; c = 1 (grey: replicated), 3 or 4 (alpha dropped: channels never mix).
struct UnfilterItem {
  int64_t raw_off, out_off;   // bytes from the blob start: h x (1 + w c) filtered scanlines -> RGB8 [h][w][3] (grey8 [h][w] with grey_out)
  int32_t h, w, c, grey_out;  // grey_out != 0 (c = 1 only): the segmentation masks, one byte per pixel
};

__device__ __forceinline__ int png_paeth(int a, int b, int c) {      // a = left, b = above, c = above-left
  const int p = a + b - c;
  const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

constexpr int kUnfilterMaxRows = 256;    // one workgroup, one thread per row (the reference's images are 256 x 256; taller ones are reconstructed by the worker)
constexpr int kUnfilterSlack = 16;       // readable bytes the caller guarantees in front of and behind every filtered image (inside the blob)

// One image.  A thread walks its row in GROUPS of four pixels: the 4 C filtered bytes of a group are ONE unaligned load (global memory
// takes any byte address on gfx950) requested four groups = sixteen steps before their turn, the group's 12 output bytes ONE store —
// vmcnt retires in order, loads behind stores, so every memory instruction between a request and its use delays it: with a load and
// two byte stores per STEP the kernel waited for a memory round trip per step (0.45-0.58 ms per batch of 16 images).  No load stands
// inside a divergent branch (behind one the compiler drains vmcnt at the join), and the step's barrier waits for the wave's LDS word only.
template <int C, bool GREY = false>
__device__ __forceinline__ void png_unfilter_image(unsigned char* __restrict__ blob, const UnfilterItem& it, unsigned (*s_px)[256]) {
  constexpr int NCH = C == 1 ? 1 : 3;                           // channels reconstructed (alpha is never needed: channels do not mix)
  const int tid = threadIdx.x, h = min(it.h, kUnfilterMaxRows), w = it.w;
  const size_t rb = 1 + (size_t)w * C;
  const bool row = tid < h;
  const unsigned char* rp = blob + it.raw_off + (size_t)(row ? tid : 0) * rb;
  constexpr int OB = GREY ? 1 : 3;                              // output bytes per pixel
  unsigned char* op = blob + it.out_off + (size_t)(row ? tid : 0) * w * OB;
  const int ft = row ? rp[0] : 0;
  rp += 1;
  struct Group { unsigned d[C]; };
  auto fetch = [&](int x0) -> Group {                           // pixels x0 .. x0 + 3 (a group without a pixel of the row: any readable address)
    const unsigned char* q = rp + ((x0 <= -4 || x0 >= w) ? 0 : x0 * C);
    Group g;
    __builtin_memcpy(g.d, q, 4 * C);
    return g;
  };
  int left[3] = {0, 0, 0}, upl[3] = {0, 0, 0};
  const int m1 = -(int)(ft == 1), m2 = -(int)(ft == 2), m3 = -(int)(ft == 3), m4 = -(int)(ft == 4);      // all ones for the row's filter type
  Group win[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) win[i] = fetch(4 * i - tid);
  const int steps = (w + h - 1 + 15) & ~15;
  for (int t = 0; t < steps; t += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x0 = t + 4 * i - tid;
      const Group g = win[i];
      unsigned od[3] = {0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = x0 + k;
        const bool act = row && x >= 0 && x < w;
        const unsigned above = s_px[(k + 1) & 1][tid > 0 ? tid - 1 : 0];      // thread tid - 1 wrote its pixel x one step ago (step parity = k & 1: t + 4 i is even)
        int up[3];
        up[0] = tid > 0 ? (int)(above & 255u) : 0; up[1] = tid > 0 ? (int)((above >> 8) & 255u) : 0; up[2] = tid > 0 ? (int)((above >> 16) & 255u) : 0;
        // branch-free: rows of one wave carry different filter types, and every `if` on them costs the wave all its sides plus the exec-mask
        // bookkeeping (the first version: ~250 instructions per step); an idle thread computes on and keeps its state by the selects
        int o[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          if (ch < NCH) {
            const int bi = k * C + ch;                          // byte of the group
            const int a = left[ch], b = up[ch], c0 = upl[ch];
            const int pa = abs(b - c0), pb = abs(a - c0), pc = abs(a + b - 2 * c0);
            const int paeth = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c0);
            const int pred = (a & m1) | (b & m2) | (((a + b) >> 1) & m3) | (paeth & m4);      // (masks, not selects: the compiler turns a select chain on the filter type back into branches)
            o[ch] = (int)(((g.d[bi >> 2] >> (8 * (bi & 3))) & 255u) + (unsigned)pred) & 255;
            left[ch] = act ? o[ch] : left[ch];
            upl[ch] = act ? b : upl[ch];
          }
        }
        if (NCH == 1) { o[1] = o[0]; o[2] = o[0]; }
        const unsigned px = (unsigned)o[0] | ((unsigned)o[1] << 8) | ((unsigned)o[2] << 16);
        if (act) s_px[k & 1][tid] = px;
        if constexpr (GREY) {
          od[0] |= (px & 255u) << (8 * k);
        } else {
          const int sh = 8 * ((3 * k) & 3), dw = (3 * k) >> 2;          // the pixel's 24 bits into the group's 96 (static: k is unrolled)
          od[dw] |= px << sh;
          if (sh > 8) od[dw + 1] |= px >> (32 - sh);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      // the group's output: one 12-byte (grey: 4-byte) store; a group that straddles an end of the row goes bytewise
      if (row && x0 >= 0 && x0 + 3 < w) {
        __builtin_memcpy(op + OB * (size_t)x0, od, 4 * OB);
      } else if (row && x0 > -4 && x0 < w) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (x0 + k >= 0 && x0 + k < w)
#pragma unroll
            for (int ch = 0; ch < OB; ++ch) op[OB * (x0 + k) + ch] = (unsigned char)(od[(OB * k + ch) >> 2] >> (8 * ((OB * k + ch) & 3)));
      }
      win[i] = fetch(x0 + 16);
    }
  }
}

__global__ __launch_bounds__(256) void png_unfilter_kernel(unsigned char* __restrict__ blob, const UnfilterItem* __restrict__ items) {
  __shared__ unsigned s_px[2][256];
  const UnfilterItem it = items[blockIdx.x];
  if (it.c == 3) png_unfilter_image<3>(blob, it, s_px);
  else if (it.c == 4) png_unfilter_image<4>(blob, it, s_px);
  else if (it.grey_out) png_unfilter_image<1, true>(blob, it, s_px);
  else png_unfilter_image<1>(blob, it, s_px);
}

}  // namespace bsr
