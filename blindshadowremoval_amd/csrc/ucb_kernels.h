// UCB post-processing of FSRNet.test_step (/root/reference/train_test_GSC.py:411-748) ON THE DEVICE (round 5).
//
// What the reference does per item on the host with TensorFlow eager ops, numpy and cv2 — resize the generator outputs, the input, the
// ground truth and seven segmentation masks to the crop-box size and zero-pad them back (:437-477), gate the predicted shadow magnitude
// by per-region thresholds (:479-590), keep the large 4-connected components that are not hair (:594-615), apply the nose rule
// (:650-666), composite (:711-722), score SSIM / PSNR (:724-725) and lay the seven figures out as one strip (:744) — cost 26 ms of
// CPU per item in rounds 2-4 (blindshadowremoval_amd/ucb_post.py, the host statement of the same steps) and set the rate of
// FSRNet.test.  Here: three kernels per batch, every DECISION (thresholds, rounded masks, components, rules) bit-identical to
// ucb_post.ucb_postprocess:
//   * all arithmetic that feeds a comparison is done in the host statement's type and operation order with fp contraction off —
//     the bilinear resize is TensorFlow's CPU kernel's (compute_lerp: top = tl + (tr - tl) * xl; ... ; out = top + (bottom - top) * yl,
//     float32), means over the three channels are ((a + b) + c) / 3, counts are integers, float sums over the image use numpy's
//     pairwise order (128-element leaves with eight interleaved accumulators, then a balanced binary tree);
//   * region slices follow Python's slice rules (a negative start counts from the end).
// ucb_resize_kernel: one thread per output pixel.  The per-item work — round 5: ONE workgroup of 1024 threads per item walking its 65 536 pixels
// stage by stage (16 of 256 CUs busy for 1.2 ms per batch of 16) — is, since round 6, a CHAIN of small launches: every stage that is a map
// over pixels (mask boxes / counts, the gates, the per-pixel threshold, union, root sizes, the keep filter, the composite and the strip)
// covers all items with S*S/256 workgroups each, integer reductions by atomics into a per-item variable block in the scratch (a workgroup
// folds its own contribution in LDS first), the two float sums that feed comparisons keep numpy's pairwise order (a workgroup of 256
// pixels owns exactly two 128-element leaves; a one-workgroup-per-item launch folds the 512 leaf sums in the balanced tree and derives the
// scalars of the next stage).  Same arithmetic, same decisions, same bytes; what a stage reads of another workgroup's results crosses a
// kernel boundary (union-find parents: agent-scope atomics, as before).
// ucb_ssim_kernel: tf.image.ssim's 11x11 Gaussian window as two separable float32 passes through LDS + the squared error for PSNR,
// one partial sum per workgroup, folded in a fixed order (deterministic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsr {

constexpr int kUcbCh = 17;                // resized planes per pixel: gt 3 | pred 3 | tmp 3 | mp 1 | masks 7 (face_hair face mouth nose eyebrow eye glasses)
constexpr int kUcbFigs = 7;
constexpr int kUcbVarsBytes = 512;         // sizeof(UcbItemVars) rounded up

struct UcbScratch {                      // per-item arrays inside the caller's scratch block (all sized for N = S*S pixels)
  float* w;                              // [N][17]
  float* mp;                             // [N] gated magnitude
  float* out;                            // [N][3] composite
  float* fval;                           // [N] float32 values of a pairwise sum
  double* dval;                          // [N] float64 values of a pairwise sum
  unsigned* label;                       // [N] union-find parents
  unsigned* csize;                       // [N] component sizes (at the root)
  int* chair;                            // [N] signed hair sum per component (at the root)
  unsigned char* keep;                   // [N]
  double* ssim_part;                     // [2][nblk] partial sums of the SSIM map and of the squared error
  struct UcbItemVars* vars;              // per-item variables of the stage chain (round 6)
  float* leaf_f;                         // [N / 128] leaf sums of a float32 pairwise sum
  double* leaf_d;                        // [N / 128] leaf sums of a float64 pairwise sum
};

__host__ __device__ inline size_t ucb_item_scratch_bytes(int S) {
  const size_t N = (size_t)S * S;
  const size_t nblk = (size_t)((S + 15) / 16) * ((S + 15) / 16);
  size_t b = N * kUcbCh * 4 + N * 4 + N * 3 * 4 + N * 4 + N * 8 + N * 4 + N * 4 + N * 4 + N + 2 * nblk * 8;
  b = (b + 7) & ~size_t(7);
  b += kUcbVarsBytes + (N / 128) * (8 + 4);                  // the stage chain's variable block and leaf sums
  return (b + 255) & ~size_t(255);
}

__host__ __device__ inline UcbScratch ucb_scratch(void* base, int item, int S) {
  const size_t N = (size_t)S * S;
  const size_t nblk = (size_t)((S + 15) / 16) * ((S + 15) / 16);
  unsigned char* p = static_cast<unsigned char*>(base) + (size_t)item * ucb_item_scratch_bytes(S);
  UcbScratch s;
  s.dval = reinterpret_cast<double*>(p); p += N * 8;
  s.ssim_part = reinterpret_cast<double*>(p); p += 2 * nblk * 8;
  s.w = reinterpret_cast<float*>(p); p += N * kUcbCh * 4;
  s.mp = reinterpret_cast<float*>(p); p += N * 4;
  s.out = reinterpret_cast<float*>(p); p += N * 3 * 4;
  s.fval = reinterpret_cast<float*>(p); p += N * 4;
  s.label = reinterpret_cast<unsigned*>(p); p += N * 4;
  s.csize = reinterpret_cast<unsigned*>(p); p += N * 4;
  s.chair = reinterpret_cast<int*>(p); p += N * 4;
  s.keep = p; p += N;
  p = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(p) + 7) & ~uintptr_t(7));
  s.vars = reinterpret_cast<struct UcbItemVars*>(p); p += kUcbVarsBytes;
  s.leaf_d = reinterpret_cast<double*>(p); p += (N / 128) * 8;
  s.leaf_f = reinterpret_cast<float*>(p);
  return s;
}

// size of the crop box as the reference computes it: int(box[3] - box[1]) on float32 values (train_test_GSC.py:417-418)
__device__ inline int ucb_box_size(const float* box) {
#pragma clang fp contract(off)
  return (int)(box[3] - box[1]);
}

// rows10: [B][S][S][10] float32 = input 3 | ground truth 3 | con_rgb 3 | dif 1 of row 0 of each item; masks: [B][7][S][S] uint8 grey
// levels (cv2.imread(...) / 255.0, one of the three equal channels); boxes: [B][4] float32
__global__ __launch_bounds__(256) void ucb_resize_kernel(const float* __restrict__ rows10, const unsigned char* __restrict__ masks,
                                                         const float* __restrict__ boxes, int S, void* scratch) {
#pragma clang fp contract(off)
  const int item = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= S * S) return;
  const int oy = p / S, ox = p % S;
  const int size = ucb_box_size(boxes + 4 * item);
  float* w = ucb_scratch(scratch, item, S).w + (size_t)p * kUcbCh;
  if (size <= 0 || size > S || oy >= size || ox >= size) {          // the zero pad of :454-477
#pragma unroll
    for (int c = 0; c < kUcbCh; ++c) w[c] = 0.f;
    return;
  }
  // TensorFlow's half-pixel bilinear weights (resize_weights in ucb_post.py): in = (i + 0.5f) * scale - 0.5f
  const float scale = (float)S / (float)size;
  const float sy = ((float)oy + 0.5f) * scale - 0.5f, sx = ((float)ox + 0.5f) * scale - 0.5f;
  const float fy = floorf(sy), fx = floorf(sx);
  const int y0 = max((int)fy, 0), y1 = min((int)ceilf(sy), S - 1);
  const int x0 = max((int)fx, 0), x1 = min((int)ceilf(sx), S - 1);
  const float yl = sy - fy, xl = sx - fx;
  const float* r = rows10 + (size_t)item * S * S * 10;
  const float* tl = r + ((size_t)y0 * S + x0) * 10; const float* tr = r + ((size_t)y0 * S + x1) * 10;
  const float* bl = r + ((size_t)y1 * S + x0) * 10; const float* br = r + ((size_t)y1 * S + x1) * 10;
  auto lerp = [&](float a, float b, float c, float d) {
    const float top = a + (b - a) * xl;
    const float bottom = c + (d - c) * xl;
    return top + (bottom - top) * yl;
  };
  // rows10 channels: im 0-2, gt 3-5, con 6-8, dif 9  ->  w: gt 0-2, pred 3-5, tmp 6-8, mp 9
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    w[c] = lerp(tl[3 + c], tr[3 + c], bl[3 + c], br[3 + c]);
    w[3 + c] = lerp(tl[6 + c], tr[6 + c], bl[6 + c], br[6 + c]);
    w[6 + c] = lerp(tl[c], tr[c], bl[c], br[c]);
  }
  w[9] = lerp(tl[9], tr[9], bl[9], br[9]);
  const unsigned char* m = masks + (size_t)item * 7 * S * S;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const unsigned char* mk = m + (size_t)k * S * S;
    auto g = [&](int y, int x) { return (float)((double)mk[y * S + x] / 255.0); };      // np.asarray(.., float64) / 255.0, then float32
    w[10 + k] = rintf(lerp(g(y0, x0), g(y0, x1), g(y1, x0), g(y1, x1)));               // tf.round: half to even
  }
}

// Python's a[start:stop] on an axis of length n -> [lo, hi)
__device__ inline void py_slice(int start, int stop, int n, int& lo, int& hi) {
  if (start < 0) start += n;
  if (stop < 0) stop += n;
  lo = min(max(start, 0), n);
  hi = min(max(stop, 0), n);
  if (hi < lo) hi = lo;
}

// Parent pointers are updated by atomics (performed in L2): they are READ with agent-scope atomic loads too, so that no stale line of
// the CU's vector L1 is ever taken for a root.
__device__ inline unsigned uf_load(const unsigned* a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int uf_load(const int* a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned uf_find(const unsigned* L, unsigned x) {
  unsigned p = uf_load(L + x);
  while (p != x) { x = p; p = uf_load(L + x); }
  return x;
}
__device__ inline void uf_union(unsigned* L, unsigned a, unsigned b) {
  for (;;) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a > b) { const unsigned t = a; a = b; b = t; }        // the smaller index becomes the root
    const unsigned old = atomicMin(&L[b], a);
    if (old == b) return;
    b = old;
  }
}

enum { UCB_OK = 0, UCB_EMPTY_MASK = 1, UCB_BAD_BOX = 2 };

// integer variables of an item (atomic min / max / add targets of the pixel stages)
enum { NOSE_R0, NOSE_R1, NOSE_C0, NOSE_C1, MOUTH_R0, MOUTH_R1, MOUTH_C0, MOUTH_C1, BROW_CNT, BROW_R0, BROW_C0, FACE_C0, FACE_C1, FACE_CNT,
       FH_R0, FH_C0, FH_C1, FH_CNT, NOSE_CNT, MOUTH_CNT, CNT_SR, CNT_ROI, CNT_DEN, MAX_SIZE, KEEP_CNT, NOSE_SH, UCB_NVARS };
constexpr int kUcbMinVars[] = {NOSE_R0, NOSE_C0, MOUTH_R0, MOUTH_C0, BROW_R0, BROW_C0, FACE_C0, FH_R0, FH_C0};
constexpr int kUcbMaxVars[] = {NOSE_R1, NOSE_C1, MOUTH_R1, MOUTH_C1, FACE_C1, FH_C1, MAX_SIZE};

struct UcbItemVars {
  int v[UCB_NVARS];
  int fail;                                  // UCB_OK, or why the item gets a black strip (every later stage skips it)
  int size;
  int forehead_rule, roi_off, left_rule, nose_hit;
  int r1a, r1b, c1a, c1b, r2a, r2b, below_lo, below_hi, fr0, fr1, fc0, fc1, left_hi, ra, rb, ca, cb;
  double min_size;
};
static_assert(sizeof(UcbItemVars) <= kUcbVarsBytes, "variable block");

__device__ inline bool ucb_is_min(int k) { for (int i : kUcbMinVars) if (i == k) return true; return false; }
__device__ inline bool ucb_is_max(int k) { for (int i : kUcbMaxVars) if (i == k) return true; return false; }

// A workgroup's contribution to the item's variables: folded in LDS (s_v, initialised by ucb_wg_vars_begin), then one global atomic per
// variable the workgroup touched.  `mask` says which variables a stage updates.
__device__ inline void ucb_wg_vars_begin(int* s_v, int tid) {
  if (tid < UCB_NVARS) s_v[tid] = ucb_is_min(tid) ? 0x7fffffff : (ucb_is_max(tid) ? -1 : 0);
  __syncthreads();
}
__device__ inline void ucb_wg_vars_end(const int* s_v, UcbItemVars* g, int tid) {
  __syncthreads();
  if (tid < UCB_NVARS) {
    const int x = s_v[tid];
    if (ucb_is_min(tid)) { if (x != 0x7fffffff) atomicMin(&g->v[tid], x); }
    else if (ucb_is_max(tid)) { if (x != -1) atomicMax(&g->v[tid], x); }
    else if (x != 0) atomicAdd(&g->v[tid], x);
  }
}

// numpy's pairwise sum (np.add.reduce on a contiguous array of N = nleaf * 128 values): leaves of 128 with eight interleaved accumulators
// (ucb_leaf_sum: one thread per leaf, its 128 values in LDS), then a balanced binary tree over the leaf sums (ucb_tree_sum: one workgroup).
template <typename T>
__device__ inline T ucb_leaf_sum(const T* a) {
#pragma clang fp contract(off)
  T r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  for (int i = 8; i < 128; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  }
  return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}
template <typename T>
__device__ inline T ucb_tree_sum(const T* __restrict__ leaves, int nleaf, T* s_tree, int tid, int nthreads) {
#pragma clang fp contract(off)
  for (int i = tid; i < nleaf; i += nthreads) s_tree[i] = leaves[i];
  __syncthreads();
  for (int st = 1; st < nleaf; st *= 2) {
    for (int i = tid; i < nleaf; i += nthreads)
      if ((i % (2 * st)) == 0 && i + st < nleaf) s_tree[i] = s_tree[i] + s_tree[i + st];
    __syncthreads();
  }
  return s_tree[0];
}

// ---- the stage chain.  Pixel stages: grid (N / 256, B), 256 threads, pixel p = blockIdx.x * 256 + tid.  Item stages: grid (B), 512 threads.

__global__ void ucb_init_kernel(const float* __restrict__ boxes, int S, void* scratch) {       // grid (B), 64 threads
  const int item = blockIdx.x, tid = threadIdx.x;
  UcbItemVars* g = ucb_scratch(scratch, item, S).vars;
  if (tid < UCB_NVARS) g->v[tid] = ucb_is_min(tid) ? 0x7fffffff : (ucb_is_max(tid) && tid != MAX_SIZE ? -1 : 0);
  if (tid == 0) {
    const int size = ucb_box_size(boxes + 4 * item);
    g->size = size;
    g->fail = (size <= 0 || size > S) ? UCB_BAD_BOX : UCB_OK;
    g->forehead_rule = g->roi_off = g->left_rule = g->nose_hit = 0;
  }
}

// One mask's contribution from a wave: count, row and column bounds of the lanes where `in` holds, folded by shuffles; lane 0 posts them.
__device__ inline int ucb_wave_min(int v) { for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o)); return v; }
__device__ inline int ucb_wave_max(int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o)); return v; }
__device__ inline int ucb_wave_add(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ inline void ucb_wave_box(int* s_v, bool in, int y, int x, int r0, int r1, int c0, int c1, int cnt) {
  const unsigned long long m = __ballot(in);
  if (m == 0) return;                                           // wave-uniform
  const int ylo = ucb_wave_min(in ? y : 0x7fffffff), yhi = ucb_wave_max(in ? y : -1);
  const int xlo = ucb_wave_min(in ? x : 0x7fffffff), xhi = ucb_wave_max(in ? x : -1);
  if ((threadIdx.x & 63) == 0) {
    if (r0 >= 0) atomicMin(&s_v[r0], ylo);
    if (r1 >= 0) atomicMax(&s_v[r1], yhi);
    if (c0 >= 0) atomicMin(&s_v[c0], xlo);
    if (c1 >= 0) atomicMax(&s_v[c1], xhi);
    if (cnt >= 0) atomicAdd(&s_v[cnt], __popcll(m));
  }
}

// stage 1: bounding boxes and counts of the rounded masks (:479-489, :533-536, :565-567)
__global__ __launch_bounds__(256) void ucb_s1_kernel(int S, void* scratch) {
  __shared__ int s_v[UCB_NVARS];
  const int item = blockIdx.y, tid = threadIdx.x, p = blockIdx.x * 256 + tid;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  if (sc.vars->fail) return;
  ucb_wg_vars_begin(s_v, tid);
  const float4 w = *reinterpret_cast<const float4*>(sc.w + (size_t)p * kUcbCh + 11 - 3);      // channels 8 .. 11
  const float4 w2 = *reinterpret_cast<const float4*>(sc.w + (size_t)p * kUcbCh + 12);        // channels 12 .. 15
  const int y = p / S, x = p % S;
  ucb_wave_box(s_v, w2.y == 1.f, y, x, NOSE_R0, NOSE_R1, NOSE_C0, NOSE_C1, NOSE_CNT);
  ucb_wave_box(s_v, w2.x == 1.f, y, x, MOUTH_R0, MOUTH_R1, MOUTH_C0, MOUTH_C1, MOUTH_CNT);
  ucb_wave_box(s_v, w2.z == 1.f, y, x, BROW_R0, -1, BROW_C0, -1, -1);
  ucb_wave_box(s_v, w2.z != 0.f, y, x, -1, -1, -1, -1, BROW_CNT);    // np.sum(brow): the rounded mask is 0 / 1, three equal channels
  ucb_wave_box(s_v, w.w == 1.f, y, x, -1, -1, FACE_C0, FACE_C1, FACE_CNT);
  ucb_wg_vars_end(s_v, sc.vars, tid);
}

__global__ void ucb_a1_kernel(int S, void* scratch) {            // grid (B), 1 thread: what stage 1 decided
  UcbItemVars* g = ucb_scratch(scratch, blockIdx.x, S).vars;
  if (threadIdx.x != 0 || g->fail) return;
  if (g->v[NOSE_CNT] == 0 || g->v[MOUTH_CNT] == 0) { g->fail = UCB_EMPTY_MASK; return; }
  g->forehead_rule = 3 * g->v[BROW_CNT] > 30 ? 1 : 0;
}

// bbox of the face above the eyebrows (:535-538), only where the forehead rule applies
__global__ __launch_bounds__(256) void ucb_s1b_kernel(int S, void* scratch) {
  __shared__ int s_v[UCB_NVARS];
  const int item = blockIdx.y, tid = threadIdx.x, p = blockIdx.x * 256 + tid;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  if (sc.vars->fail || !sc.vars->forehead_rule) return;
  ucb_wg_vars_begin(s_v, tid);
  const int upper_brow = sc.vars->v[BROW_R0];
  const int y = p / S, x = p % S;
  ucb_wave_box(s_v, y < upper_brow && sc.w[(size_t)p * kUcbCh + 11] == 1.f, y, x, FH_R0, -1, FH_C0, FH_C1, FH_CNT);
  ucb_wg_vars_end(s_v, sc.vars, tid);
}

__global__ void ucb_a1b_kernel(int S, void* scratch) {           // grid (B), 1 thread: the slices of stages 2-3
  UcbItemVars* g = ucb_scratch(scratch, blockIdx.x, S).vars;
  if (threadIdx.x != 0 || g->fail) return;
  if (g->forehead_rule && g->v[FH_CNT] == 0) { g->fail = UCB_EMPTY_MASK; return; }
  if (3 * g->v[BROW_CNT] > 0 && g->v[FACE_CNT] == 0) { g->fail = UCB_EMPTY_MASK; return; }
  const double mid_nose_height = (g->v[NOSE_R1] + g->v[NOSE_R0]) / 2.0;
  const int upper_mouth = g->v[MOUTH_R0], lower_mouth = g->v[MOUTH_R1], left_mouth = g->v[MOUTH_C0], right_mouth = g->v[MOUTH_C1];
  py_slice((int)mid_nose_height, upper_mouth, S, g->r1a, g->r1b);
  py_slice(left_mouth, right_mouth, S, g->c1a, g->c1b);
  py_slice(upper_mouth, lower_mouth, S, g->r2a, g->r2b);
  py_slice(upper_mouth, S, S, g->below_lo, g->below_hi);
}

// stage 2: gate the magnitude around mustache and mouth (:473-499); stage 3: counts and the float32 sum for the "mouth and below" rules (:547-564)
__global__ __launch_bounds__(256) void ucb_s23_kernel(int S, void* scratch) {
#pragma clang fp contract(off)
  __shared__ int s_v[UCB_NVARS];
  __shared__ float s_f[256];
  const int item = blockIdx.y, tid = threadIdx.x, p = blockIdx.x * 256 + tid;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  const UcbItemVars* g = sc.vars;
  if (g->fail) return;
  ucb_wg_vars_begin(s_v, tid);
  const float* w = sc.w + (size_t)p * kUcbCh;
  const int y = p / S, x = p % S;
  float mp = w[9] * w[10];
  const bool incol = x >= g->c1a && x < g->c1b;
  if (incol && y >= g->r1a && y < g->r1b && mp < 0.018f) mp = mp * 0.f;
  if (incol && y >= g->r2a && y < g->r2b && mp < 0.02f) mp = mp * 0.f;
  sc.mp[p] = mp;
  const float roi = (y >= g->below_lo && y < g->below_hi) ? w[11] : 0.f;
  const float shadowed = mp > 0.01f ? 1.f : 0.f;
  const unsigned long long m_roi = __ballot(roi != 0.f), m_sr = __ballot(roi != 0.f && shadowed != 0.f);
  if ((tid & 63) == 0) {
    if (m_roi) atomicAdd(&s_v[CNT_ROI], __popcll(m_roi));
    if (m_sr) { atomicAdd(&s_v[CNT_SR], __popcll(m_sr)); atomicAdd(&s_v[CNT_DEN], __popcll(m_sr)); }
  }
  const float a = roi * w[6] * shadowed, b = roi * w[7] * shadowed, c = roi * w[8] * shadowed;
  s_f[tid] = ((a + b) + c) / 3.f;                               // np.mean(roi * tmp * shadowed, 2)
  ucb_wg_vars_end(s_v, sc.vars, tid);                            // (its barrier also publishes s_f)
  if (tid < 2) sc.leaf_f[blockIdx.x * 2 + tid] = ucb_leaf_sum<float>(s_f + 128 * tid);
}

__global__ __launch_bounds__(512) void ucb_a23_kernel(int S, void* scratch) {      // grid (B), 512 threads: the tree, then the scalars of stage 4
#pragma clang fp contract(off)
  __shared__ float s_tree[512];
  const int tid = threadIdx.x;
  const UcbScratch sc = ucb_scratch(scratch, blockIdx.x, S);
  UcbItemVars* g = sc.vars;
  if (g->fail) return;
  const float mean_num = ucb_tree_sum<float>(sc.leaf_f, S * S / 128, s_tree, tid, 512);
  if (tid != 0) return;
  const float frac = (float)(3 * g->v[CNT_SR]) / (float)(3 * g->v[CNT_ROI]);
  const float mean_below = mean_num / (float)g->v[CNT_DEN];
  g->roi_off = ((0.252f < frac && frac < 0.268f) || (0.3f < frac && frac < 0.31f && mean_below > 0.358f) || (0.295f < frac && frac < 0.3f && mean_below > 0.22f)) ? 1 : 0;
  g->fr0 = g->fr1 = g->fc0 = g->fc1 = 0;
  if (g->forehead_rule) {
    py_slice(g->v[FH_R0] + 20, g->v[BROW_R0] - 40, S, g->fr0, g->fr1);
    py_slice(g->v[FH_C0] + 40, g->v[FH_C1] - 40, S, g->fc0, g->fc1);
  }
  g->left_rule = 0;
  g->left_hi = 0;
  if (3 * g->v[BROW_CNT] > 0) {
    const int left_face = g->v[FACE_C0], right_face = g->v[FACE_C1];
    if (g->v[BROW_C0] - left_face == 0) {
      g->left_rule = 1;
      int lo;
      py_slice(0, (int)(left_face * 0.8 + right_face * 0.2), S, lo, g->left_hi);
    }
  }
}

// stage 4: per-pixel threshold and detection (:501-590), union-find initialisation
__global__ __launch_bounds__(256) void ucb_s4_kernel(int S, void* scratch) {
#pragma clang fp contract(off)
  const int item = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  const UcbItemVars* g = sc.vars;
  if (g->fail) return;
  const float* w = sc.w + (size_t)p * kUcbCh;
  const int y = p / S, x = p % S;
  const float hair = w[10] - w[11];
  const float intensity = ((w[6] + w[7]) + w[8]) / 3.f;
  float thr = 0.01f;
  if (hair > 0.f) thr = 0.02f;
  if (hair > 0.f && intensity < 0.13f) thr = 0.004f;
  if (g->forehead_rule && y >= g->fr0 && y < g->fr1 && x >= g->fc0 && x < g->fc1 && intensity < 0.4f) thr = -0.001f;
  const float roi = (y >= g->below_lo && y < g->below_hi) ? w[11] : 0.f;
  if (g->roi_off && roi > 0.f) thr = 1.0f;
  if (g->left_rule && x < g->left_hi && w[14] > 0.f && intensity > 0.1f) thr = 1.0f;
  const bool det = sc.mp[p] > thr;
  sc.keep[p] = det ? 1 : 0;
  // a detected pixel starts out pointing at the first pixel of its run inside this wave: stage 5 then only joins runs
  const int lane = threadIdx.x & 63;
  const unsigned long long km = __ballot(det);
  const bool left = lane > 0 && x > 0 && ((km >> (lane - 1)) & 1ull);
  const unsigned long long heads = __ballot(det && !left);
  unsigned start = (unsigned)p;
  if (det) start = (unsigned)(p - lane + 63 - __clzll(heads & ((2ull << lane) - 1ull)));
  __hip_atomic_store(sc.label + p, start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(sc.csize + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(sc.chair + p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// stage 5: 4-connected components (:594-615): join the runs stage 4 labelled with their left and upper neighbours ...
__global__ __launch_bounds__(256) void ucb_s5a_kernel(int S, void* scratch) {
  const int item = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  if (sc.vars->fail || !sc.keep[p]) return;
  const int y = p / S, x = p % S;
  const bool left = x > 0 && sc.keep[p - 1];
  if (left && (threadIdx.x & 63) == 0) uf_union(sc.label, (unsigned)p, (unsigned)(p - 1));        // a run that crosses waves
  // one join per stretch where this row's run touches the upper row's run: the pixel to the left has made it when both rows continue there
  if (y > 0 && sc.keep[p - S] && !(left && sc.keep[p - S - 1])) uf_union(sc.label, (unsigned)p, (unsigned)(p - S));
}
// ... then sizes / hair sums at the roots
__global__ __launch_bounds__(256) void ucb_s5b_kernel(int S, void* scratch) {
#pragma clang fp contract(off)
  const int item = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  if (sc.vars->fail) return;
  const bool k = sc.keep[p] != 0;
  unsigned root = (unsigned)p;
  int hair = 0;
  if (k) {
    root = uf_find(sc.label, (unsigned)p);
    __hip_atomic_store(sc.label + p, root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // a root keeps pointing at itself, so concurrent finds stay correct
    const float* w = sc.w + (size_t)p * kUcbCh;
    hair = (int)(w[10] - w[11]);
  }
  // one pair of atomics per (wave, component), not per pixel: a big component is one address
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(k);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned r = (unsigned)__shfl((int)root, leader);
    const bool mine = k && root == r;
    const unsigned long long m = __ballot(mine);
    const int hs = ucb_wave_add(mine ? hair : 0);
    if (lane == leader) {
      atomicAdd(&sc.csize[r], (unsigned)__popcll(m));
      if (hs != 0) atomicAdd(&sc.chair[r], hs);
    }
    todo &= ~m;
  }
}
__global__ __launch_bounds__(256) void ucb_s5c_kernel(int S, void* scratch) {       // the largest component
  __shared__ int s_v[UCB_NVARS];
  const int item = blockIdx.y, tid = threadIdx.x, p = blockIdx.x * 256 + tid;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  if (sc.vars->fail) return;
  ucb_wg_vars_begin(s_v, tid);
  if (sc.keep[p] && uf_load(sc.label + p) == (unsigned)p) atomicMax(&s_v[MAX_SIZE], (int)uf_load(sc.csize + p));
  ucb_wg_vars_end(s_v, sc.vars, tid);
}

// the keep filter (:603-615) and stage 6, part 1: the sums of the nose rule (:650-666)
__global__ __launch_bounds__(256) void ucb_s56_kernel(int S, void* scratch) {
#pragma clang fp contract(off)
  __shared__ int s_v[UCB_NVARS];
  __shared__ double s_d[256];
  const int item = blockIdx.y, tid = threadIdx.x, p = blockIdx.x * 256 + tid;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  if (sc.vars->fail) return;
  ucb_wg_vars_begin(s_v, tid);
  const double min_size = 0.45 * (double)sc.vars->v[MAX_SIZE];
  unsigned char k = 0;
  if (sc.keep[p]) {
    const unsigned root = uf_load(sc.label + p);
    const unsigned sz = uf_load(sc.csize + root);
    if ((double)sz >= min_size && (double)uf_load(sc.chair + root) / (double)sz < 0.8) k = 1;
  }
  sc.keep[p] = k;                                               // only this thread reads keep[p] in this launch
  const float* w = sc.w + (size_t)p * kUcbCh;
  const float mean3 = ((w[6] + w[7]) + w[8]) / 3.f;            // np.mean(tmp, 2): float32
  const double sh = (double)k * (double)mean3;                 // keep is a float64 array in the host statement
  s_d[tid] = sh;
  const unsigned long long m_k = __ballot(k != 0), m_n = __ballot((double)w[13] * sh > 0.0);
  if ((tid & 63) == 0) {
    if (m_k) atomicAdd(&s_v[KEEP_CNT], __popcll(m_k));
    if (m_n) atomicAdd(&s_v[NOSE_SH], __popcll(m_n));
  }
  ucb_wg_vars_end(s_v, sc.vars, tid);
  if (tid < 2) sc.leaf_d[blockIdx.x * 2 + tid] = ucb_leaf_sum<double>(s_d + 128 * tid);
}

__global__ __launch_bounds__(512) void ucb_a6_kernel(int S, void* scratch) {        // grid (B), 512 threads: the nose rule's verdict
#pragma clang fp contract(off)
  __shared__ double s_tree[512];
  const int tid = threadIdx.x;
  const UcbScratch sc = ucb_scratch(scratch, blockIdx.x, S);
  UcbItemVars* g = sc.vars;
  if (g->fail) return;
  const double sum_sh = ucb_tree_sum<double>(sc.leaf_d, S * S / 128, s_tree, tid, 512);
  if (tid != 0) return;
  const double mean_intensity = sum_sh / (double)g->v[KEEP_CNT];
  const double frac_nose = (double)g->v[NOSE_SH] / (double)(float)g->v[NOSE_CNT];
  g->nose_hit = 0;
  if ((0.15 < frac_nose && frac_nose < 0.25) || (0.30 < frac_nose && frac_nose < 0.31) || (0.34 < frac_nose && frac_nose < 0.35)) {
    const double mid_nose_height = (g->v[NOSE_R1] + g->v[NOSE_R0]) / 2.0, mid_nose_width = (g->v[NOSE_C1] + g->v[NOSE_C0]) / 2.0;
    const int reach = mean_intensity < 0.15 ? 5 : 65;
    g->nose_hit = 1;
    py_slice((int)mid_nose_height, (int)(double)(g->v[NOSE_R1] + reach), S, g->ra, g->rb);
    py_slice((int)(mid_nose_width - 35), (int)(mid_nose_width + 35), S, g->ca, g->cb);
  }
}

// stage 6, part 2 + stage 7: the nose rule applied, the composite (:711-722) and the seven figures (:744) as one uint8 strip (utils.py:217-233:
// clip, * 255, round half to even).  losses: [B][2] = ssim, psnr (ucb_ssim_finish_kernel); strips: [B][S][7 S][3] uint8; figs: optional
// [B][7][S][S][3] float32; status: [B] (UCB_EMPTY_MASK where the host statement raises on an empty nose / mouth / forehead / face mask)
__global__ __launch_bounds__(256) void ucb_s7_kernel(int S, void* scratch, unsigned char* __restrict__ strips, float* __restrict__ figs, int* __restrict__ status) {
#pragma clang fp contract(off)
  const int item = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const int N = S * S;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  const UcbItemVars* g = sc.vars;
  unsigned char* strip = strips + (size_t)item * N * kUcbFigs * 3;
  const int y = p / S, x = p % S;
  if (p == 0) status[item] = g->fail;
  if (g->fail) {                                                // a black strip; NaN losses are left to the finish kernel
#pragma unroll
    for (int k = 0; k < kUcbFigs; ++k) {
      unsigned char* dst = strip + ((size_t)y * (kUcbFigs * S) + (size_t)k * S + x) * 3;
      dst[0] = dst[1] = dst[2] = 0;
      if (figs != nullptr) { float* fd = figs + (((size_t)item * kUcbFigs + k) * N + p) * 3; fd[0] = fd[1] = fd[2] = 0.f; }
    }
    sc.out[(size_t)p * 3] = sc.out[(size_t)p * 3 + 1] = sc.out[(size_t)p * 3 + 2] = 0.f;
    return;
  }
  unsigned char keep = sc.keep[p];
  if (g->nose_hit && y >= g->ra && y < g->rb && x >= g->ca && x < g->cb) keep = 0;
  const float* w = sc.w + (size_t)p * kUcbCh;
  const float d = keep ? 1.f : 0.f;
  const float mp2 = sc.mp[p] * 2.f;
  float f[kUcbFigs][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float tmp = w[6 + c], pred = w[3 + c];
    const float o = fminf(fmaxf(pred * d + tmp * (1.f - d), 0.f), 1.f);
    sc.out[(size_t)p * 3 + c] = o;
    f[0][c] = tmp; f[1][c] = o; f[2][c] = mp2; f[3][c] = w[c]; f[4][c] = d; f[5][c] = pred; f[6][c] = w[13] * tmp;
  }
#pragma unroll
  for (int k = 0; k < kUcbFigs; ++k) {
    unsigned char* dst = strip + ((size_t)y * (kUcbFigs * S) + (size_t)k * S + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c] = (unsigned char)rintf(fminf(fmaxf(f[k][c], 0.f), 1.f) * 255.f);
    if (figs != nullptr) {
      float* fd = figs + (((size_t)item * kUcbFigs + k) * N + p) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) fd[c] = f[k][c];
    }
  }
}

// tf.image.ssim(gt, out, 1.0) / tf.image.psnr (:724-725) as blindshadowremoval_amd/metrics.py states them: 11-tap Gaussian (sigma 1.5),
// 'VALID', float32, vertical then horizontal pass over x, y, x^2, y^2, xy; one 16x16 tile of the (S-10)^2 map per workgroup.
constexpr int kSsimTile = 16, kSsimWin = 11, kSsimIn = kSsimTile + kSsimWin - 1;

__global__ __launch_bounds__(256) void ucb_ssim_kernel(int S, void* scratch) {
  __shared__ float s_x[kSsimIn][kSsimIn + 1], s_y[kSsimIn][kSsimIn + 1];
  __shared__ float s_v[5][kSsimTile][kSsimIn + 1];
  __shared__ double s_red[2][256];
  const int item = blockIdx.y, tid = threadIdx.x;
  const int tiles = (S + kSsimTile - 1) / kSsimTile;
  const int ty0 = (blockIdx.x / tiles) * kSsimTile, tx0 = (blockIdx.x % tiles) * kSsimTile;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  const int M = S - kSsimWin + 1;                              // size of the SSIM map
  float g[kSsimWin];
  {
    double e[kSsimWin], sum = 0.0;
    for (int i = 0; i < kSsimWin; ++i) { const double x = i - (kSsimWin - 1) / 2.0; e[i] = exp(-(x * x) / (2.0 * 1.5 * 1.5)); sum += e[i]; }
    for (int i = 0; i < kSsimWin; ++i) g[i] = (float)(e[i] / sum);
  }
  double acc_ssim = 0.0, acc_se = 0.0;
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
    for (int i = tid; i < kSsimIn * kSsimIn; i += 256) {
      const int yy = i / kSsimIn, xx = i % kSsimIn;
      const int y = ty0 + yy, x = tx0 + xx;
      float a = 0.f, b = 0.f;
      if (y < S && x < S) { a = sc.w[((size_t)y * S + x) * kUcbCh + c]; b = sc.out[((size_t)y * S + x) * 3 + c]; }
      s_x[yy][xx] = a; s_y[yy][xx] = b;
    }
    __syncthreads();
    // squared error of this tile's own 16x16 pixels (every pixel of the image belongs to exactly one tile)
    {
      const int yy = tid / kSsimTile, xx = tid % kSsimTile;
      if (ty0 + yy < S && tx0 + xx < S) { const double d = (double)s_x[yy][xx] - (double)s_y[yy][xx]; acc_se += d * d; }
    }
    for (int i = tid; i < kSsimTile * kSsimIn; i += 256) {      // vertical pass
      const int yy = i / kSsimIn, xx = i % kSsimIn;
      float vx = 0.f, vy = 0.f, vxx = 0.f, vyy = 0.f, vxy = 0.f;
      for (int k = 0; k < kSsimWin; ++k) {
        const float a = s_x[yy + k][xx], b = s_y[yy + k][xx];
        vx += g[k] * a; vy += g[k] * b; vxx += g[k] * (a * a); vyy += g[k] * (b * b); vxy += g[k] * (a * b);
      }
      s_v[0][yy][xx] = vx; s_v[1][yy][xx] = vy; s_v[2][yy][xx] = vxx; s_v[3][yy][xx] = vyy; s_v[4][yy][xx] = vxy;
    }
    __syncthreads();
    {
      const int yy = tid / kSsimTile, xx = tid % kSsimTile;
      if (ty0 + yy < M && tx0 + xx < M) {
        float mx = 0.f, my = 0.f, xx2 = 0.f, yy2 = 0.f, xy = 0.f;
        for (int k = 0; k < kSsimWin; ++k) {
          mx += g[k] * s_v[0][yy][xx + k]; my += g[k] * s_v[1][yy][xx + k]; xx2 += g[k] * s_v[2][yy][xx + k];
          yy2 += g[k] * s_v[3][yy][xx + k]; xy += g[k] * s_v[4][yy][xx + k];
        }
        const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
        const float sxx = xx2 - mx * mx, syy = yy2 - my * my, sxy = xy - mx * my;
        const float lum = (2.f * mx * my + c1) / (mx * mx + my * my + c1);
        const float cs = (2.f * sxy + c2) / (sxx + syy + c2);
        acc_ssim += (double)(lum * cs);
      }
    }
  }
  s_red[0][tid] = acc_ssim; s_red[1][tid] = acc_se;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (tid < s) { s_red[0][tid] += s_red[0][tid + s]; s_red[1][tid] += s_red[1][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) {
    const int nblk = tiles * tiles;
    sc.ssim_part[blockIdx.x] = s_red[0][0];
    sc.ssim_part[nblk + blockIdx.x] = s_red[1][0];
  }
}

__global__ __launch_bounds__(64) void ucb_ssim_finish_kernel(int S, void* scratch, const int* __restrict__ status, float* __restrict__ losses, int B) {   // grid (B), one wave
  const int item = blockIdx.x, lane = threadIdx.x;
  const UcbScratch sc = ucb_scratch(scratch, item, S);
  const int tiles = (S + kSsimTile - 1) / kSsimTile, nblk = tiles * tiles;
  double a = 0.0, e = 0.0;
  for (int i = lane; i < nblk; i += 64) { a += sc.ssim_part[i]; e += sc.ssim_part[nblk + i]; }      // a fixed order: lane partials, then a butterfly
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); e += __shfl_xor(e, o); }
  if (lane != 0) return;
  const int M = S - kSsimWin + 1;
  if (status[item] != UCB_OK) { losses[2 * item] = __builtin_nanf(""); losses[2 * item + 1] = __builtin_nanf(""); return; }
  losses[2 * item] = (float)(a / ((double)M * M * 3.0));
  losses[2 * item + 1] = (float)(20.0 * log10(1.0) - 10.0 * log10(e / ((double)S * S * 3.0)));
}

inline hipError_t launch_ucb_post(const float* rows10, const unsigned char* masks, const float* boxes, int B, int S, float* losses,
                                  unsigned char* strips, float* figs, int* status, void* scratch, hipStream_t stream) {
  const int N = S * S;
  hipLaunchKernelGGL(ucb_resize_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256), 0, stream, rows10, masks, boxes, S, scratch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const dim3 px((unsigned)(N / 256), (unsigned)B), it((unsigned)B);
  hipLaunchKernelGGL(ucb_init_kernel, it, dim3(64), 0, stream, boxes, S, scratch);
  hipLaunchKernelGGL(ucb_s1_kernel, px, dim3(256), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_a1_kernel, it, dim3(64), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_s1b_kernel, px, dim3(256), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_a1b_kernel, it, dim3(64), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_s23_kernel, px, dim3(256), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_a23_kernel, it, dim3(512), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_s4_kernel, px, dim3(256), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_s5a_kernel, px, dim3(256), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_s5b_kernel, px, dim3(256), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_s5c_kernel, px, dim3(256), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_s56_kernel, px, dim3(256), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_a6_kernel, it, dim3(512), 0, stream, S, scratch);
  hipLaunchKernelGGL(ucb_s7_kernel, px, dim3(256), 0, stream, S, scratch, strips, figs, status);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int tiles = (S + kSsimTile - 1) / kSsimTile;
  hipLaunchKernelGGL(ucb_ssim_kernel, dim3((unsigned)(tiles * tiles), (unsigned)B), dim3(256), 0, stream, S, scratch);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ucb_ssim_finish_kernel, dim3((unsigned)B), dim3(64), 0, stream, S, scratch, status, losses, B);
  return hipGetLastError();
}

}  // namespace bsr
