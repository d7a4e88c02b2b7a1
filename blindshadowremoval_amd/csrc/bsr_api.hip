// libbsr_hip.so — host side of the MI355X-native GSC generator forward (C ABI in include/bsr_hip.h).
// Orchestrates Generator.call (/root/reference/model.py:228-290) as a fixed sequence of hand-written
// gfx950 kernels over one pre-planned NHWC workspace; concatenations are channel slices of shared buffers.
#include "../../include/bsr_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "attention.h"
#include "attention_h16.h"
#include "conv3_f16.h"
#include "conv_n16.h"
#include "gemm_nloop.h"
#include "glue_kernels.h"
#include "igemm_conv.h"
#include "igemm_h16.h"
#include "png_kernels.h"
#include "prep_kernels.h"
#include "stem7.h"
#include "ucb_kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return fail(BSR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));                \
  } while (0)

// Every entry point runs on the handle's device and puts the caller's current device back on return.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) err = hipSetDevice(device);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// ---- packed-weight blob (written by blindshadowremoval_amd/pack.py) ----
constexpr uint32_t kBlobMagic = 0x57525342u;  // "BSRW"
constexpr uint32_t kBlobVersion = 1;
struct BlobHeader {
  uint32_t magic, version, n_entries, reserved;
};
struct BlobEntry {
  char name[40];
  uint64_t offset;   // bytes from blob start, 16-byte aligned
  uint64_t nfloats;
  int32_t dims[4];   // conv weights: {nchunk, taps, n_pad, CC+4}; bias: {n_pad,0,0,0}
};

struct LayerW {
  const float* w = nullptr;
  const float* b = nullptr;
  int nchunk = 0, taps = 0, n_pad = 0, ldp = 0;
};

enum KClass { K_CONV3 = 0, K_CONVT = 1, K_CONV1 = 2, K_ATT = 3, K_CONV7 = 4, K_GLUE = 5, K_CONVT_NI2 = 6 };

constexpr int CS_Y3X = 288;  // y3x = conv3 output + block input, all 9 channel tiles kept (TSM inputs are wider than 257: model_with_TSM.py:105-113)

// Channel plan of the bottleneck trunk.  GSC (/root/reference/model.py:238,259): xa = cat[x 96 | uv 3], blocks 0-2 are 257
// wide, xh = cat[x_hole 257 | bmask | uv 3].  TSM (/root/reference/model_with_TSM.py:272,293) inserts the ShareLayer output:
// xa = cat[x 96 | x_share 192 | uv 3] = 291, blocks 0-2 keep 291, xh = cat[x_hole 291 | bmask | x_share 582 | uv 3] = 877.
struct Variant {
  bool tsm;
  int c_a, cs_a, uv_a;        // res0 input: real channels, stride, uv slot
  int c_r, cs_r;              // blocks 0-2 output
  int c_h, cs_h, uv_h;        // blocks 3-5 input/output, uv slot
};
constexpr Variant kGSC{false, 99, 120, 96, 257, 264, 261, 264, 258};
// 16-bit matrix-core modes (BSR_DTYPE_F16 / BSR_DTYPE_F32X3): K chunks are 32 channels, so the 257 / 261-wide tensors get stride 288
constexpr Variant kGSC16{false, 99, 128, 96, 257, 288, 261, 288, 258};
constexpr Variant kTSM{true, 291, 312, 288, 291, 312, 877, 888, 874};
constexpr Variant kTSM16{true, 291, 320, 288, 291, 320, 877, 896, 874};          // TSM widths at the 16-bit kernels' 32-channel granularity
constexpr int CS_CF = 64;    // f = clr_up3 output; the gs channel of cat[gs, f] (model.py:267) is read from the gs output

struct Plan {  // float offsets into the workspace for a (B,H,W) problem
  size_t x1, c3, c2, xa, t1, t2, y3[6], qkv, att[6], r[6], xh, ybuf, qh, f1, f2, cf, probe, reg32, share, total;
};

Plan make_plan(size_t B, size_t H, size_t W, const Variant& v = kGSC) {
  Plan p;
  size_t off = 0;
  auto take = [&](size_t floats) {
    size_t o = off;
    off += (floats + 63) & ~size_t(63);
    return o;
  };
  const size_t px = B * H * W, cells = px / 64;
  p.x1 = take(px * 32);
  p.c3 = take(px / 4 * 128);
  p.c2 = take(px / 16 * 160);
  p.xa = take(cells * v.cs_a);
  p.t1 = take(cells * 128);
  p.t2 = take(cells * 128);
  for (int i = 0; i < 6; ++i) p.y3[i] = take(cells * CS_Y3X);
  p.qkv = take(cells * 384);
  for (int i = 0; i < 6; ++i) p.att[i] = take(cells * 128);
  for (int i = 0; i < 6; ++i) p.r[i] = take(cells * (i < 3 ? v.cs_r : v.cs_h));
  p.xh = take(cells * v.cs_h);
  p.ybuf = take(px * 64);
  p.qh = take(px * 16);
  p.f1 = take(px / 16 * 128);
  p.f2 = take(px / 4 * 96);
  p.cf = take(px * CS_CF);
  p.probe = take(cells * 2);
  p.reg32 = take(v.tsm ? cells * 4 : 0);
  p.share = take(v.tsm ? cells * 2 * v.c_r : 0);
  p.total = off;
  return p;
}

}  // namespace

struct bsr_handle {
  int device = 0;
  float* d_blob = nullptr;
  std::unordered_map<std::string, LayerW> layers;
  Variant var = kGSC;
  int dtype = BSR_DTYPE_F32;     // BSR_DTYPE_F16 / BSR_DTYPE_F32X3: 16-bit matrix cores on the 3x3 / stride-2 / transposed 3x3 layers (igemm_h16.h), fp32 kernels elsewhere
  float head_bias[2] = {0.f, 0.f};
  const float* tail_w = nullptr;
  const float* clr_gs_w = nullptr;
  float* ws = nullptr;
  size_t ws_floats = 0;
  Plan plan{};
  int B = 0, H = 0, W = 0;       // shape of the last forward
  bool ran = false;
  bool att_in_lds = false;       // the last forward ran attention + `w` as ONE launch: the attention output never reached the att<i> workspace slots
  // Range guard of the 16-bit modes (igemm_h16.h): one word of pinned, device-mapped host memory; a kernel that stages an activation
  // outside the fp16 range stores 1 to it (over PCIe, only when it happens).  Sticky until bsr_check_range().
  unsigned* range_flag = nullptr;
  bool fuse_heads = true;        // env BSR_FUSE_HEADS=0: always the two-launch heads (A/B measurements, bit-identity tests)
  bool conv1_gemm = true;        // env BSR_CONV1_GEMM=0: res*.conv1 of the 16-bit modes on the implicit-GEMM kernel at every batch (A/B measurements, bit-identity tests)
  bool att_pv1 = false;          // f16 mode: P.V of the attention with the hi planes only (one matrix instruction per product instead of three).  Measured on
                                 // the f16 parity tests: margins 60 / 75 / 65 % of F16_TOL used against 64 / 69 / 75 % with the split product — the mode's
                                 // error is set by its fp16 activations, not by this — and 1.4 % of the forward (profiles/HISTORY.md round 6); f32x3 keeps the split
  bool conv3_f16 = true;         // env BSR_CONV3_F16=0: the f16 mode's 3x3 / transposed 3x3 layers on igemm_h16_kernel<.., NSPLIT = 1> (the form small test shapes
                                 // and A/B measurements compare against; same operands, another summation order)
  bool fuse_attw = true;         // env BSR_FUSE_ATTW=0: attention and the `w` GEMM as two launches (A/B measurements, bit-identity tests)
  bool timing = false;
  std::vector<hipEvent_t> ev;    // event pool, pairs
  std::vector<int> ev_class;
  std::vector<std::string> ev_name;   // layer name of each event pair ("up3", "res2.conv2", "attention4", ...)
  size_t ev_used = 0;
};

namespace {

const char* const kRangeMsg =
    "an activation exceeded the fp16 range (|x| >= 65520) in a forward of this 16-bit-mode handle: its outputs are not trustworthy "
    "(inf / NaN where the fp32 path stays finite).  Re-run those inputs on a BSR_DTYPE_F32 handle; bsr_check_range() clears the condition";

int find_layer(bsr_handle* h, const char* name, int nchunk, int taps, int ldp, int n_min, LayerW* out) {
  auto it = h->layers.find(name);
  if (it == h->layers.end()) return fail(BSR_ERR_BLOB, std::string("blob has no layer '") + name + "'");
  const LayerW& l = it->second;
  if (l.w == nullptr || l.b == nullptr || l.nchunk != nchunk || l.taps != taps || l.ldp != ldp || l.n_pad < n_min) {
    char buf[256];
    snprintf(buf, sizeof buf, "layer '%s': blob has {chunks %d, taps %d, n_pad %d, ldp %d}, kernel needs {%d, %d, >=%d, %d}", name,
             l.nchunk, l.taps, l.n_pad, l.ldp, nchunk, taps, n_min, ldp);
    return fail(BSR_ERR_BLOB, buf);
  }
  *out = l;
  return BSR_OK;
}

struct Launcher {
  bsr_handle* h;
  hipStream_t s;
  int rc = BSR_OK;

  void begin(int cls, const char* name) {
    if (!h->timing) return;
    if (h->ev_used + 2 > h->ev.size()) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        h->ev.push_back(e);
      }
      h->ev_class.push_back(cls);
      h->ev_name.emplace_back();
    }
    h->ev_class[h->ev_used / 2] = cls;
    h->ev_name[h->ev_used / 2] = name;
    hipEventRecord(h->ev[h->ev_used], s);
  }
  void end() {
    if (!h->timing || h->ev_used + 2 > h->ev.size()) return;
    hipEventRecord(h->ev[h->ev_used + 1], s);
    h->ev_used += 2;
  }
  void check(hipError_t e, const char* what) {
    if (e != hipSuccess && rc == BSR_OK) rc = fail(BSR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  }

  // KH,KW,S,TR,TH,TW,WM,WN,MI,NI,CC,PF_IN
  // io (f16 mode only): bit 0 = `in` is an fp16 tensor, bit 1 = `out` is written as fp16 (the fp16 activation pack of configs[3])
  template <int KH, int KW, int S, bool TR, int NI, int CC, int INB>
  void conv(int cls, const char* name, const float* in, int in_cs, int in_coff, int k_pad, int H, int W, float* out, int out_cs,
            int out_coff, int n_store, int act, int io = 0) {
    if (rc != BSR_OK) return;
    using C = bsr::ConvCfg<KH, KW, S, TR, 4, 32, 4, 1, 1, NI, CC, INB>;
    constexpr bool k33 = (KH == 3 && KW == 3);
    constexpr bool k11 = (KH == 1 && KW == 1);
    constexpr int CCH = CC == 24 ? 32 : CC;                       // K chunk of the 16-bit kernels (multiple of 16)
    const bool h16 = h->dtype != BSR_DTYPE_F32;
    // 16-bit modes: the 3x3-conv layers take fp16 operands (f16) or hi/lo planes (f32x3); the 1x1 layers are split-precision in both
    const int nsplit = (k33 && h->dtype == BSR_DTYPE_F16) ? 1 : 2;
    LayerW l;
    const int nb = (n_store + C::BN - 1) / C::BN;
    if (h16) {
      if (k_pad % CCH != 0) { rc = fail(BSR_ERR_ARG, std::string("layer '") + name + "': K is not a multiple of the 16-bit kernels' chunk"); return; }
      // words per weight row: padded rows, or the unpadded swizzled 128-byte rows of the DMA-fed f32x3 layers (H16Cfg::LDPW)
      const int ldpw = nsplit == 2 ? bsr::H16Cfg<KH, KW, S, TR, 4, 32, 4, 1, 1, NI, CCH, INB, 2>::LDPW : (CCH + 8) / 2;
      rc = find_layer(h, name, k_pad / CCH, KH * KW, ldpw, nb * C::BN, &l);
    } else {
      rc = find_layer(h, name, k_pad / CC, KH * KW, CC + 4, nb * C::BN, &l);
    }
    if (rc != BSR_OK) return;
    bsr::ConvArgs a{};
    a.in = in; a.in_cs = in_cs; a.in_coff = in_coff; a.H = H; a.W = W;
    a.out = out; a.out_cs = out_cs; a.out_coff = out_coff;
    a.Ho = TR ? 2 * H : (H + S - 1) / S;
    a.Wo = TR ? 2 * W : (W + S - 1) / S;
    a.w = l.w; a.bias = l.b; a.nchunk = l.nchunk; a.n_pad = l.n_pad; a.n_store = n_store;
    // TF SAME pad-before: total = max((out-1)*s + k - in, 0), before = total / 2
    auto pad_before = [](int in, int k, int s) {
      int o = (in + s - 1) / s, tot = (o - 1) * s + k - in;
      return tot > 0 ? tot / 2 : 0;
    };
    a.pad_t = pad_before(H, KH, S);
    a.pad_l = pad_before(W, KW, S);
    a.act = act;
    a.range_flag = h->range_flag;
    const int mh = TR ? H : a.Ho, mw = TR ? W : a.Wo;
    if (mh % 4 != 0 || mw % 32 != 0) {
      rc = fail(BSR_ERR_ARG, std::string("layer '") + name + "': feature map is not a multiple of the 4x32 tile");
      return;
    }
    // f16 mode, stride-1 3x3 and transposed 3x3: the trio-stepped kernel of conv3_f16.h from the layer's `w3` image (pack.py)
    if constexpr (k33 && S == 1) {
      if (h->dtype == BSR_DTYPE_F16 && h->conv3_f16) {
        char nm3[48];
        snprintf(nm3, sizeof nm3, "%s.w3", name);
        LayerW l3;
        const int nblk = (n_store + 63) / 64;
        rc = find_layer(h, nm3, nblk * (k_pad / 32), 9, 16, 64, &l3);
        if (rc != BSR_OK) return;
        a.w = l3.w; a.bias = l3.b; a.nchunk = k_pad / 32; a.n_pad = nblk * 64;
        begin(cls, name);
        if (io == 3) check(bsr::launch_conv3_f16<TR, 3>(a, h->B, s), name);
        else if (io == 2) check(bsr::launch_conv3_f16<TR, 2>(a, h->B, s), name);
        else if (io == 1) check(bsr::launch_conv3_f16<TR, 1>(a, h->B, s), name);
        else check(bsr::launch_conv3_f16<TR, 0>(a, h->B, s), name);
        end();
        return;
      }
    }
    begin(cls, name);
    static_assert(k33 || k11, "igemm layers are 3x3 or 1x1");
    // Small batches (round 4): the 1/8-resolution trunk layers have B x 8 pixel tiles of 4x32 — fewer workgroups than CUs below
    // B = 16.  Their 2x32-pixel variant (two waves on the pixel rows x two on the channel halves, the same per-element accumulation
    // order: bit-identical) doubles the grid.
    constexpr bool kTrunk = !TR && S == 1 && NI == 2 && (CC == 32 || (k11 && CC == 24));
    bool half_tile = false;
    if constexpr (kTrunk) half_tile = !h16 && mh % 2 == 0 && (long long)(mh / 4) * (mw / 32) * h->B * nb < bsr::device_cu_count();
    if (half_tile) {
      if constexpr (kTrunk) check(bsr::launch_igemm_conv<KH, KW, S, TR, 2, 32, 2, 2, 1, 1, CC, INB>(a, h->B, s), name);
    } else if (!h16)
      check(bsr::launch_igemm_conv<KH, KW, S, TR, 4, 32, 4, 1, 1, NI, CC, INB>(a, h->B, s), name);
    else if (nsplit == 2)
      check(bsr::launch_igemm_h16<KH, KW, S, TR, 4, 32, 4, 1, 1, NI, CCH, INB, 2>(a, h->B, s), name);
    else if constexpr (k33) {
      if (io == 3) check(bsr::launch_igemm_h16<KH, KW, S, TR, 4, 32, 4, 1, 1, NI, CCH, INB, 1, 3>(a, h->B, s), name);
      else if (io == 2) check(bsr::launch_igemm_h16<KH, KW, S, TR, 4, 32, 4, 1, 1, NI, CCH, INB, 1, 2>(a, h->B, s), name);
      else if (io == 1) check(bsr::launch_igemm_h16<KH, KW, S, TR, 4, 32, 4, 1, 1, NI, CCH, INB, 1, 1>(a, h->B, s), name);
      else check(bsr::launch_igemm_h16<KH, KW, S, TR, 4, 32, 4, 1, 1, NI, CCH, INB, 1, 0>(a, h->B, s), name);
    }
    end();
  }
  // 1x1 conv as a resident-activation GEMM (K = NCH*32) over all N
  template <int NI, int NCH>
  void gemm(int cls, const char* name, const float* in, int in_cs, size_t pixels, float* out, int out_cs, int n_store, int act,
            const float* res1 = nullptr, int res1_cs = 0, int res1_c = 0,
            float* out2 = nullptr, int out2_cs = 0, int n_split = 0, int n_store1 = 0) {
    if (rc != BSR_OK) return;
    using C = bsr::GemmNLoopCfg<NI, NCH>;
    LayerW l;
    const int tiles = (n_store + 31) / 32;
    // two workgroups per CU share the N range; small batches (fewer than 2 x CUs workgroups that way) split it further, down to one
    // channel group per range — every 32-channel tile is computed the same way whatever range it falls in (bit-identical)
    int kNSplit = 2;
    {
      const long long mblocks = (long long)(pixels / C::BM);
      const int want = (int)((2LL * bsr::device_cu_count() + mblocks - 1) / (mblocks > 0 ? mblocks : 1));
      const int most = (tiles + NI - 1) / NI;
      kNSplit = want < 2 ? 2 : (want > most ? most : want);
#ifdef BSR_NL_NSPLIT
      kNSplit = BSR_NL_NSPLIT > most ? most : BSR_NL_NSPLIT;
#endif
    }
    rc = find_layer(h, name, NCH, 1, 36, (tiles + NI - 1) * 32, &l);   // the last group of a range may read (zero) rows past its tiles
    if (rc != BSR_OK) return;
    if (pixels % C::BM != 0) { rc = fail(BSR_ERR_ARG, std::string("layer '") + name + "': pixel count is not a multiple of 128"); return; }
    bsr::ConvArgs a{};
    a.in = in; a.in_cs = in_cs; a.in_coff = 0; a.out = out; a.out_cs = out_cs; a.out_coff = 0;
    a.w = l.w; a.bias = l.b; a.nchunk = l.nchunk; a.n_pad = l.n_pad; a.n_store = n_store; a.act = act;
    a.res1 = res1; a.res1_cs = res1_cs; a.res1_c = res1_c;
    a.out2 = out2; a.out2_cs = out2_cs; a.n_split = n_split; a.n_store1 = n_store1;
    a.range_flag = h->range_flag;
    a.out2_split = (h->dtype != BSR_DTYPE_F32 && out2 != nullptr) ? 1 : 0;      // conv3 | theta|phi|g of the 16-bit modes: qkv in the split layout of attention_h16.h
    begin(cls, name);
    if (h->dtype == BSR_DTYPE_F32)
      check(bsr::launch_gemm_nloop<NI, NCH, 0>(a, pixels, kNSplit, s), name);
    else
      check(bsr::launch_gemm_nloop<NI, NCH, 2>(a, pixels, kNSplit, s), name);      // split-precision in both 16-bit modes
    end();
  }

  void heads(const float* y, int H, int W, float* qh, const float* inputs, float* gs, float* mask22, size_t npix) {
    if (rc != BSR_OK) return;
    LayerW l;
    rc = find_layer(h, "heads", 2, 7, 36, 16, &l);
    if (rc != BSR_OK) return;
    if (l.n_pad != 16) { rc = fail(BSR_ERR_BLOB, "layer 'heads' must be packed with n_pad 16"); return; }
    if (H % 8 != 0 || W % 32 != 0) { rc = fail(BSR_ERR_ARG, "layer 'heads': image is not a multiple of its tile"); return; }
    bsr::ConvN16Args a{};
    a.in = y; a.in_cs = 64; a.H = H; a.W = W; a.w = l.w; a.bias = l.b; a.out = qh; a.out_cs = 16; a.act = 0;
    a.pad_t = 3; a.pad_l = 0;
    a.inputs = inputs; a.gs_out = gs; a.mask22 = mask22; a.b_mask = h->head_bias[0]; a.b_con = h->head_bias[1];
    a.range_flag = h->range_flag;
    const int dt = h->dtype;
    int resident = 0;
    check(bsr::launch_conv_n16<7, 1, false, false, 2, 0, false, true>(a, h->B, s, &resident), "heads");      // query: resident workgroup slots
    const bool fuse = rc == BSR_OK && h->fuse_heads && bsr::conv_n16_fuse_pays(h->B, H, 8, resident);
    begin(K_CONV7, "heads");
    if (fuse) {
      if (dt == BSR_DTYPE_F32) check(bsr::launch_conv_n16<7, 1, false, false, 2, 0, false, true>(a, h->B, s), "heads");
      else if (dt == BSR_DTYPE_F16) check(bsr::launch_conv_n16<7, 1, false, false, 2, 2, true, true>(a, h->B, s), "heads");
      else check(bsr::launch_conv_n16<7, 1, false, false, 2, 2, false, true>(a, h->B, s), "heads");
      end();
      return;
    }
    if (dt == BSR_DTYPE_F32) check(bsr::launch_conv_n16<7, 1, false, false, 2, 0>(a, h->B, s), "heads");
    else if (dt == BSR_DTYPE_F16) check(bsr::launch_conv_n16<7, 1, false, false, 2, 2, true>(a, h->B, s), "heads");
    else check(bsr::launch_conv_n16<7, 1, false, false, 2, 2>(a, h->B, s), "heads");
    end();
    begin(K_GLUE, "heads_post");
    hipLaunchKernelGGL(bsr::heads_post_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, qh, inputs, h->head_bias[0], h->head_bias[1], gs,
                       mask22, W, npix);
    check(hipGetLastError(), "heads_post");
    end();
  }

  template <int KH, int KW, bool GS, bool TAIL, int RW>
  void conv16(int cls, const char* name, const float* in, int in_cs, int H, int W, float* out, int out_cs, int act, const float* gs,
              const float* inputs, float* con_rgb, float* dif, float* packed = nullptr) {
    if (rc != BSR_OK) return;
    LayerW l;
    rc = find_layer(h, name, 2, KH * KW, 36, 16, &l);
    if (rc != BSR_OK) return;
    if (l.n_pad != 16) { rc = fail(BSR_ERR_BLOB, std::string("layer '") + name + "' must be packed with n_pad 16"); return; }
    bsr::ConvN16Args a{};
    a.in = in; a.in_cs = in_cs; a.H = H; a.W = W; a.w = l.w; a.bias = l.b; a.out = out; a.out_cs = out_cs; a.act = act;
    a.pad_t = (KH - 1) / 2; a.pad_l = (KW - 1) / 2;
    a.gs = gs; a.w_gs = h->clr_gs_w; a.tail_w = h->tail_w; a.inputs = inputs; a.con_rgb = con_rgb; a.dif = dif; a.packed = packed;
    a.range_flag = h->range_flag;
    if (H % (4 * RW) != 0 || W % 32 != 0) { rc = fail(BSR_ERR_ARG, std::string("layer '") + name + "': image is not a multiple of its tile"); return; }
    begin(cls, name);
    if (h->dtype == BSR_DTYPE_F32)
      check(bsr::launch_conv_n16<KH, KW, GS, TAIL, RW, 0>(a, h->B, s), name);
    else if (h->dtype == BSR_DTYPE_F16)
      check(bsr::launch_conv_n16<KH, KW, GS, TAIL, RW, 2, true>(a, h->B, s), name);      // f16 mode: its input tensor is fp16
    else
      check(bsr::launch_conv_n16<KH, KW, GS, TAIL, RW, 2>(a, h->B, s), name);            // split precision
    end();
  }
};

int ensure_workspace(bsr_handle* h, int B, int H, int W, hipStream_t s) {
  Plan p = make_plan(B, H, W, h->var);
  bool fresh = false;
  if (p.total > h->ws_floats) {
    HIP_TRY(hipStreamSynchronize(s));
    if (h->ws) HIP_TRY(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_floats = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->ws), p.total * sizeof(float)));
    h->ws_floats = p.total;
    h->B = 0;
    fresh = true;
  }
  if (fresh) {
    // a new allocation holds arbitrary bits (NaN x zero weight = NaN): clear all of it once
    HIP_TRY(hipMemsetAsync(h->ws, 0, p.total * sizeof(float), s));
    h->ran = false;
  } else if (B != h->B || H != h->H || W != h->W) {
    // A new shape moves every buffer.  All of them are fully rewritten by their producers each forward except the channel-pad
    // lanes of the 1/8-resolution concat buffers (xa, xh, r0..r5: real channels < stride), which must read as exact zeros:
    // clear just those (a few MB per image instead of the whole workspace).
    const size_t cells = (size_t)B * H * W / 64;
    const Variant& v = h->var;
    HIP_TRY(hipMemsetAsync(h->ws + p.xa, 0, cells * v.cs_a * sizeof(float), s));
    HIP_TRY(hipMemsetAsync(h->ws + p.xh, 0, cells * v.cs_h * sizeof(float), s));
    for (int i = 0; i < 6; ++i) HIP_TRY(hipMemsetAsync(h->ws + p.r[i], 0, cells * (i < 3 ? v.cs_r : v.cs_h) * sizeof(float), s));
    h->ran = false;
  }
  h->plan = p;
  h->B = B; h->H = H; h->W = W;
  return BSR_OK;
}

}  // namespace

extern "C" {

int bsr_abi_version(void) { return 8; }

#ifndef BSR_SRC_SHA
#define BSR_SRC_SHA "unhashed"
#endif
// the tag makes the hash findable in the FILE (build.library_sha16 reads it without loading the library into the process)
static const char kSrcShaTag[] = "BSR_SRC_SHA=" BSR_SRC_SHA;
const char* bsr_source_sha(void) { return kSrcShaTag + 12; }

const char* bsr_last_error(void) { return g_last_error.c_str(); }

size_t bsr_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return make_plan(B, H, W).total * sizeof(float);
}

int bsr_create(bsr_handle** out, int device, const void* packed_weights, size_t nbytes, int dtype) {
  if (out == nullptr || packed_weights == nullptr) return fail(BSR_ERR_ARG, "bsr_create: null argument");
  *out = nullptr;
  if (dtype != BSR_DTYPE_F32 && dtype != BSR_DTYPE_F16 && dtype != BSR_DTYPE_F32X3)
    return fail(BSR_ERR_ARG, "bsr_create: dtype must be BSR_DTYPE_F32, BSR_DTYPE_F16 or BSR_DTYPE_F32X3");
  if (nbytes < sizeof(BlobHeader)) return fail(BSR_ERR_BLOB, "bsr_create: blob shorter than its header");
  const uint8_t* blob = static_cast<const uint8_t*>(packed_weights);
  BlobHeader hd;
  memcpy(&hd, blob, sizeof hd);
  if (hd.magic != kBlobMagic || hd.version != kBlobVersion) return fail(BSR_ERR_BLOB, "bsr_create: bad blob magic/version");
  if ((int)hd.reserved != dtype) return fail(BSR_ERR_BLOB, "bsr_create: the blob was packed for another dtype (pack_generator(weights, dtype) must match bsr_create's dtype)");
  const size_t table_end = sizeof(BlobHeader) + (size_t)hd.n_entries * sizeof(BlobEntry);
  if (table_end > nbytes) return fail(BSR_ERR_BLOB, "bsr_create: blob entry table exceeds blob size");
  DeviceGuard guard(device);
  HIP_TRY(guard.err);
  bsr_handle* h = new bsr_handle();
  h->device = device;
  h->dtype = dtype;
  if (const char* e_ = getenv("BSR_FUSE_HEADS")) h->fuse_heads = atoi(e_) != 0;
  if (const char* e_ = getenv("BSR_FUSE_ATTW")) h->fuse_attw = atoi(e_) != 0;
  if (const char* e_ = getenv("BSR_CONV3_F16")) h->conv3_f16 = atoi(e_) != 0;
  h->att_pv1 = dtype == BSR_DTYPE_F16;
  if (const char* e_ = getenv("BSR_CONV1_GEMM")) h->conv1_gemm = atoi(e_) != 0;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->d_blob), nbytes);
  if (e == hipSuccess) e = hipMemcpy(h->d_blob, blob, nbytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (h->d_blob) hipFree(h->d_blob);
    delete h;
    return fail(BSR_ERR_HIP, std::string("bsr_create: weight upload: ") + hipGetErrorString(e));
  }
  for (uint32_t i = 0; i < hd.n_entries; ++i) {
    BlobEntry en;
    memcpy(&en, blob + sizeof(BlobHeader) + (size_t)i * sizeof(BlobEntry), sizeof en);
    en.name[sizeof(en.name) - 1] = 0;
    if (en.offset % 16 != 0 || en.offset + en.nfloats * 4 > nbytes) {
      bsr_destroy(h);
      return fail(BSR_ERR_BLOB, std::string("bsr_create: entry '") + en.name + "' is out of bounds or misaligned");
    }
    std::string nm(en.name);
    const float* dptr = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(h->d_blob) + en.offset);
    if (nm == "heads.bias") {
      if (en.nfloats != 2) { bsr_destroy(h); return fail(BSR_ERR_BLOB, "bsr_create: heads.bias must hold 2 floats"); }
      memcpy(h->head_bias, blob + en.offset, 8);
    } else if (nm == "clr_conv1.gs") {
      if (en.nfloats != 256) { bsr_destroy(h); return fail(BSR_ERR_BLOB, "bsr_create: clr_conv1.gs must hold 16x16 floats"); }
      h->clr_gs_w = dptr;
    } else if (nm == "tail.w") {
      if (en.nfloats != 16 * 16 + 16 + 48 + 3) { bsr_destroy(h); return fail(BSR_ERR_BLOB, "bsr_create: tail.w has the wrong size"); }
      h->tail_w = dptr;
    } else if (nm.size() > 2 && nm.compare(nm.size() - 2, 2, ".w") == 0) {
      LayerW& l = h->layers[nm.substr(0, nm.size() - 2)];
      l.w = dptr;
      l.nchunk = en.dims[0]; l.taps = en.dims[1]; l.n_pad = en.dims[2]; l.ldp = en.dims[3];
      if ((uint64_t)l.nchunk * l.taps * l.n_pad * l.ldp != en.nfloats) {
        bsr_destroy(h);
        return fail(BSR_ERR_BLOB, std::string("bsr_create: entry '") + en.name + "' dims do not match its size");
      }
    } else if (nm.size() > 2 && nm.compare(nm.size() - 2, 2, ".b") == 0) {
      h->layers[nm.substr(0, nm.size() - 2)].b = dptr;
    }
  }
  if (h->tail_w == nullptr || h->clr_gs_w == nullptr) { bsr_destroy(h); return fail(BSR_ERR_BLOB, "bsr_create: blob lacks 'tail.w' / 'clr_conv1.gs'"); }
  if (dtype != BSR_DTYPE_F32) {
    e = hipHostMalloc(reinterpret_cast<void**>(&h->range_flag), 64, hipHostMallocMapped);
    if (e != hipSuccess) { h->range_flag = nullptr; bsr_destroy(h); return fail(BSR_ERR_HIP, std::string("bsr_create: range flag: ") + hipGetErrorString(e)); }
    *h->range_flag = 0u;
  }
  {   // GSC or TSM weights?  (res0.conv1 has K = 120 -> 5 chunks of 24, or K = 312 -> 13)
    auto it = h->layers.find("res0.conv1");
    if (it == h->layers.end()) { bsr_destroy(h); return fail(BSR_ERR_BLOB, "bsr_create: blob has no 'res0.conv1'"); }
    // res0.conv1: GSC K = 120 (5 x 24) | 128 (4 x 32, 16-bit modes); TSM K = 312 (13 x 24) | 320 (10 x 32)
    const int nch = it->second.nchunk;
    if (dtype == BSR_DTYPE_F32) h->var = nch == 13 ? kTSM : kGSC;
    else h->var = nch == 10 ? kTSM16 : kGSC16;
  }
  *out = h;
  return BSR_OK;
}

void bsr_destroy(bsr_handle* h) {
  if (h == nullptr) return;
  DeviceGuard guard(h->device);
  for (hipEvent_t e : h->ev) hipEventDestroy(e);
  if (h->ws) hipFree(h->ws);
  if (h->d_blob) hipFree(h->d_blob);
  if (h->range_flag) hipHostFree(h->range_flag);
  delete h;
}

int bsr_reserve(bsr_handle* h, int B, int H, int W) {
  if (h == nullptr || B <= 0 || H <= 0 || W <= 0) return fail(BSR_ERR_ARG, "bsr_reserve: bad argument");
  DeviceGuard guard(h->device);
  HIP_TRY(guard.err);
  Plan p = make_plan(B, H, W, h->var);
  if (p.total > h->ws_floats) {
    HIP_TRY(hipDeviceSynchronize());
    if (h->ws) HIP_TRY(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_floats = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->ws), p.total * sizeof(float)));
    HIP_TRY(hipMemset(h->ws, 0, p.total * sizeof(float)));
    h->ws_floats = p.total;
    h->B = 0;
    h->ran = false;
  }
  return BSR_OK;
}

int bsr_check_range(bsr_handle* h, void* stream) {
  if (h == nullptr) return fail(BSR_ERR_ARG, "bsr_check_range: null handle");
  if (h->range_flag == nullptr) return BSR_OK;                 // BSR_DTYPE_F32: nothing is ever converted to fp16
  DeviceGuard guard(h->device);
  HIP_TRY(guard.err);
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  // read-and-clear in ONE atomic exchange: a report raised by another stream's forward between a separate read and clear would be
  // lost (one stream per handle remains the rule — bsr_hip.h — but the flag itself no longer depends on it)
  if (__atomic_exchange_n(h->range_flag, 0u, __ATOMIC_SEQ_CST) == 0u) return BSR_OK;
  return fail(BSR_ERR_RANGE, kRangeMsg);
}

int bsr_peek_range(bsr_handle* h) {
  if (h == nullptr) return fail(BSR_ERR_ARG, "bsr_peek_range: null handle");
  if (h->range_flag == nullptr) return BSR_OK;
  // no synchronisation, no clear: the word lives in host memory the kernels write over PCIe; what it says covers every forward that
  // has COMPLETED so far (the caller waited for an event of the one it asks about)
  if (__atomic_load_n(h->range_flag, __ATOMIC_SEQ_CST) == 0u) return BSR_OK;
  return fail(BSR_ERR_RANGE, kRangeMsg);
}

int bsr_set_timing(bsr_handle* h, int enable) {
  if (h == nullptr) return fail(BSR_ERR_ARG, "bsr_set_timing: null handle");
  h->timing = enable != 0;
  h->ev_used = 0;
  return BSR_OK;
}

int bsr_get_timing(bsr_handle* h, float ms[BSR_NUM_CLASSES], int launches[BSR_NUM_CLASSES]) {
  if (h == nullptr || ms == nullptr || launches == nullptr) return fail(BSR_ERR_ARG, "bsr_get_timing: null argument");
  for (int i = 0; i < BSR_NUM_CLASSES; ++i) { ms[i] = 0.f; launches[i] = 0; }
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    HIP_TRY(hipEventSynchronize(h->ev[i + 1]));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]));
    const int c = h->ev_class[i / 2];
    ms[c] += t;
    launches[c] += 1;
  }
  return BSR_OK;
}

int bsr_timing_launches(bsr_handle* h) { return h == nullptr ? 0 : (int)(h->ev_used / 2); }

int bsr_timing_entry(bsr_handle* h, int i, char* name, size_t name_cap, float* ms, int* cls) {
  if (h == nullptr || name == nullptr || ms == nullptr || cls == nullptr || name_cap == 0) return fail(BSR_ERR_ARG, "bsr_timing_entry: null argument");
  if (i < 0 || (size_t)(2 * i + 1) >= h->ev_used) return fail(BSR_ERR_ARG, "bsr_timing_entry: index out of range");
  HIP_TRY(hipEventSynchronize(h->ev[2 * i + 1]));
  HIP_TRY(hipEventElapsedTime(ms, h->ev[2 * i], h->ev[2 * i + 1]));
  *cls = h->ev_class[i];
  snprintf(name, name_cap, "%s", h->ev_name[i].c_str());
  return BSR_OK;
}

size_t bsr_handle_workspace_bytes(const bsr_handle* h, int B, int H, int W) {
  if (h == nullptr || B <= 0 || H <= 0 || W <= 0) return 0;
  return make_plan(B, H, W, h->var).total * sizeof(float);
}

static int forward_impl(bsr_handle* h, const float* inputs, const float* uv, const float* reg, int frame, int share, int B, int H, int W,
                        float* gs, float* con_rgb, float* mask22, float* dif, void* stream, float* packed = nullptr) {
  if (h == nullptr || inputs == nullptr || uv == nullptr || gs == nullptr || mask22 == nullptr || (packed == nullptr && (con_rgb == nullptr || dif == nullptr)))
    return fail(BSR_ERR_ARG, "bsr_forward: null argument");
  if (B <= 0) return fail(BSR_ERR_ARG, "bsr_forward: B must be positive");
  const Variant& V = h->var;
  if (V.tsm != (reg != nullptr))
    return fail(BSR_ERR_ARG, V.tsm ? "bsr_forward: this handle holds TSM weights: call bsr_forward_tsm" : "bsr_forward_tsm: this handle holds GSC weights: call bsr_forward");
  if (V.tsm && (frame <= 0 || B % frame != 0 || H != W))
    return fail(BSR_ERR_ARG, "bsr_forward_tsm: B must be a multiple of frame and the image square (warp.py assumes a square map)");
  if (H <= 0 || W <= 0 || H % 32 != 0 || W % 256 != 0)
    return fail(BSR_ERR_ARG, "bsr_forward: H must be a multiple of 32 and W a multiple of 256 (reference: 256x256)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (h->range_flag != nullptr && *reinterpret_cast<volatile unsigned*>(h->range_flag) != 0u)
    return fail(BSR_ERR_RANGE, kRangeMsg);      // an earlier forward overflowed fp16: sticky until bsr_check_range() acknowledges it
  DeviceGuard guard(h->device);
  HIP_TRY(guard.err);
  int rc = ensure_workspace(h, B, H, W, s);
  if (rc != BSR_OK) return rc;
  h->ev_used = 0;
  const Plan& p = h->plan;
  float* ws = h->ws;
  const size_t npix = (size_t)B * H * W, ncell = npix / 64;
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
  Launcher L{h, s};
  auto glue_begin = [&](const char* what) { L.begin(K_GLUE, what); };
  auto glue_end = [&](const char* what) { L.check(hipGetLastError(), what); L.end(); };

  // conv1 = Conv(32, 7x7) + BN + LeakyReLU (model.py:203,230): dedicated stem kernel (7 row taps x 21 contiguous floats)
  {
    LayerW l;
    const bool x3 = h->dtype != BSR_DTYPE_F32;                     // split precision in both 16-bit modes
    L.rc = find_layer(h, "conv1", 1, 7, x3 ? 36 : 28, 32, &l);
    if (L.rc == BSR_OK) {
      bsr::StemArgs a{inputs, ws + p.x1, l.w, l.b, H, W, 0, 0, 0, h->range_flag};
      L.begin(K_CONV7, "conv1");
      if (h->dtype == BSR_DTYPE_F16)
        L.check((bsr::launch_stem7<4, 2, true>(a, B, s)), "conv1");                 // x1 stored as fp16
      else if (x3)
        L.check(bsr::launch_stem7<4, 2>(a, B, s), "conv1");
      else
        L.check(bsr::launch_stem7<4, 0>(a, B, s), "conv1");
      L.end();
    }
  }
  // down1..3 = Conv(stride 2) (model.py:207-209,231-233); x2 / x3 land in their skip-concat slots (model.py:244-245)
  // f16 mode (the fp16 pack of BASELINE configs[3]): the full- / half- / quarter-resolution tensors between the 3x3-conv layers —
  // x1, c3 = [up2 | x2], c2 = [up1 | x3], y, f1, f2, f — live in HBM as fp16 (same workspace slots, first half used); the
  // 1/8-resolution trunk (xa, t1, t2, r*, y3x, qkv, att) and the four outputs stay fp32.  IO16: 1 = fp16 input, 2 = fp16 output.
  const bool p16 = h->dtype == BSR_DTYPE_F16;
  const int io_both = p16 ? 3 : 0, io_in = p16 ? 1 : 0, io_out = p16 ? 2 : 0;
  L.conv<3, 3, 2, false, 2, 16, 1>(K_CONV3, "down1", ws + p.x1, 32, 0, 32, H, W, ws + p.c3, 128, 64, 64, 1, io_both);
  L.conv<3, 3, 2, false, 2, 16, 1>(K_CONV3, "down2", ws + p.c3, 128, 64, 64, H2, W2, ws + p.c2, 160, 96, 64, 1, io_both);
  L.conv<3, 3, 2, false, 3, 16, 1>(K_CONV3, "down3", ws + p.c2, 160, 96, 64, H4, W4, ws + p.xa, V.cs_a, 0, 96, 1, io_in);
  // uv = resize(uv, [h,w]); x = cat[x, uv] (model.py:237-238) and the uv slot of cat[x_hole, bmask, uv] (model.py:259)
  glue_begin("uv_resize8");
  hipLaunchKernelGGL(bsr::uv_resize8_kernel, dim3((unsigned)((ncell * 3 + 255) / 256)), dim3(256), 0, s, uv, H, W, ws + p.xa, V.cs_a, V.uv_a,
                     ws + p.xh, V.cs_h, V.uv_h, ncell);
  glue_end("uv_resize8");

  // TSM: x_share = ShareLayer(x, reg, frame, share) into channels [96, 288) of xa (model_with_TSM.py:271-272)
  auto share_layer = [&](const float* x, int x_cs, int C, float* out, int out_cs, int out_coff) {
    glue_begin("share_layer");
    if (share) hipLaunchKernelGGL(bsr::share_reduce_kernel, dim3((unsigned)(ncell / frame)), dim3(128), 0, s, x, x_cs, C, ws + p.reg32, H8, frame, ws + p.share);
    hipLaunchKernelGGL(bsr::share_unwarp_kernel, dim3((unsigned)ncell), dim3(128), 0, s, ws + p.share, x, x_cs, C, ws + p.reg32, H8, frame, share, out, out_cs, out_coff);
    glue_end("share_layer");
  };
  if (V.tsm) {
    glue_begin("reg_resize8");
    hipLaunchKernelGGL(bsr::reg_resize8_kernel, dim3((unsigned)((ncell * 4 + 255) / 256)), dim3(256), 0, s, reg, H, W, ws + p.reg32, ncell);
    glue_end("reg_resize8");
    share_layer(ws + p.xa, V.cs_a, 96, ws + p.xa, V.cs_a, 96);
  }

  // res*.conv1 (1x1, 99|257|261 -> 128, + BN + LeakyReLU) at full batches: the resident-activation GEMM with ONE workgroup per CU and
  // all of N per workgroup (gemm_nloop.h, MINW = 1) — same operands, same matrix-instruction order per output element as the
  // implicit-GEMM kernels (igemm_h16_kernel<1,1,1,..,NSPLIT = 2> / igemm_conv_kernel<1,1,1,..,CC = 24>): the same bits
  auto conv1_gemm = [&](const char* nm, const float* x, int x_cs) -> bool {
    if (V.tsm || !h->conv1_gemm || L.rc != BSR_OK) return false;
    if (h->dtype == BSR_DTYPE_F32) return false;      // fp32: the implicit-GEMM kernel (the resident form over 24-channel chunks was built and is no faster: profiles/HISTORY.md round 5)
    if (ncell % 128 != 0 || (long long)(ncell / 128) * 2 < bsr::device_cu_count()) return false;
    if (x_cs != 128 && x_cs != 288) return false;
    LayerW l;
    L.rc = find_layer(h, nm, x_cs / 32, 1, 36, 128, &l);
    if (L.rc != BSR_OK) return true;
    bsr::ConvArgs a{};
    a.in = x; a.in_cs = x_cs; a.in_coff = 0; a.out = ws + p.t1; a.out_cs = 128; a.out_coff = 0;
    a.w = l.w; a.bias = l.b; a.nchunk = l.nchunk; a.n_pad = l.n_pad; a.n_store = 128; a.act = 1;
    a.range_flag = h->range_flag;
    L.begin(K_CONV1, nm);
    if (x_cs == 128) L.check(bsr::launch_gemm_nloop<4, 4, 2, 1>(a, ncell, 1, s), nm);
    else L.check(bsr::launch_gemm_nloop<4, 9, 2, 1>(a, ncell, 1, s), nm);
    L.end();
    return true;
  };

  // ResBottleneck + NonLocalBlock (model.py:98-113, 23-61)
  auto res_block = [&](int i, const float* x, int x_cs, int x_c) {
    float* r_out = ws + p.r[i];
    const int o_cs = i < 3 ? V.cs_r : V.cs_h;
    char nm[32];
    float* y3 = ws + p.y3[i];
    snprintf(nm, sizeof nm, "res%d.conv1", i);
    if (!conv1_gemm(nm, x, x_cs)) L.conv<1, 1, 1, false, 2, 24, 3>(K_CONV1, nm, x, x_cs, 0, x_cs, H8, W8, ws + p.t1, 128, 0, 128, 1);
    // conv3+BN (128 -> 257 = y3) and theta|phi|g (257 -> 3x128, no activation in between: model.py:101,33-46) are ONE
    // K = 128 GEMM: the qkv weights are composed offline with conv3's (pack.py), N = [y3 288 | qkv 384].  The y3 output also absorbs
    // the block's skip: y3x = y3 + pad(x), so that the `w` GEMM below reads ONE residual.
    // (The two as ONE launch — the conv2 tile through LDS into a GEMM tail — was built in round 4, bit-identical and no faster: profiles/HISTORY.md;
    // sources at git e99927a.)
    snprintf(nm, sizeof nm, "res%d.conv2", i);
    L.conv<3, 3, 1, false, 2, 32, 1>(K_CONV3, nm, ws + p.t1, 128, 0, 128, H8, W8, ws + p.t2, 128, 0, 128, 1);
    snprintf(nm, sizeof nm, "res%d.c3q", i);
    L.gemm<3, 4>(K_CONV1, nm, ws + p.t2, 128, ncell, y3, CS_Y3X, 288 + 384, 0, x, x_cs, x_cs < 288 ? x_cs : 288, ws + p.qkv, 384, 288, CS_Y3X);
    // z = y3 + BN(w(att)); out = LeakyReLU(pad(x) + pad(z))  (model.py:56-59, 105-113) = LeakyReLU(y3x + BN(w(att))).
    // ONE launch — the `w` GEMM runs as the tail of the attention kernel on the workgroup's own 128 pixels (attention.h /
    // attention_h16.h, FUSEW; the attention output never goes to HBM).  fp32 small batches (the 4- / 2-wave attention shapes) keep the
    // two launches; both forms give the same bits (tests/test_gpu_parity.py).
    // 16-bit modes: the one-wave-per-SIMD kernel of attention_h16.h (round 6), whose normalised O^T accumulators are the tail's A operand, at
    // every batch (that kernel has one workgroup shape).
    const bool h16 = h->dtype != BSR_DTYPE_F32;
    const bool fuse_w = h->fuse_attw && (h16 || bsr::attention_auto_qw(B, H8 * W8) == 4);
    h->att_in_lds = fuse_w;
    if (fuse_w && L.rc == BSR_OK) {
      LayerW l;
      // fp32: the [4][1][n_pad][36] image gemm_tail.h streams through its ring; 16-bit modes (round 6): the `w4` image of
      // attention_h16.h — nine 16-KB tiles in the k order of the O^T accumulators, resident in LDS when the key loop ends
      snprintf(nm, sizeof nm, h16 ? "res%d.w4" : "res%d.w", i);
      L.rc = h16 ? find_layer(h, nm, 9, 1, 128, 32, &l) : find_layer(h, nm, 4, 1, 36, 12 * 32, &l);
      if (L.rc == BSR_OK) {
        bsr::AttWArgs wa{};
        wa.w = l.w; wa.bias = l.b; wa.n_pad = h16 ? 288 : l.n_pad;
        wa.res = y3; wa.res_cs = CS_Y3X; wa.res_c = CS_Y3X;
        wa.out = r_out; wa.out_cs = o_cs; wa.n_store = o_cs < 288 ? o_cs : 288; wa.n_store1 = wa.n_store; wa.act = 1; wa.stagger = 1;
        snprintf(nm, sizeof nm, "res%d.attw", i);
        L.begin(K_ATT, nm);
        if (!h16)
          L.check(bsr::launch_nonlocal_attention_w(ws + p.qkv, B, H8 * W8, wa, s), "attention+w");
        else
          L.check(bsr::launch_nonlocal_attention_h16_w(ws + p.qkv, B, H8 * W8, wa, s, h->att_pv1), "attention_h16+w");
        L.end();
      }
    } else {
      if (L.rc == BSR_OK) {
        snprintf(nm, sizeof nm, "res%d.attention", i);
        L.begin(K_ATT, nm);
        if (!h16)
          L.check(bsr::launch_nonlocal_attention(ws + p.qkv, ws + p.att[i], B, H8 * W8, s), "attention");
        else
          L.check(bsr::launch_nonlocal_attention_h16(ws + p.qkv, ws + p.att[i], B, H8 * W8, s, h->att_pv1), "attention_h16");
        L.end();
      }
      snprintf(nm, sizeof nm, "res%d.w", i);
      L.gemm<3, 4>(K_CONV1, nm, ws + p.att[i], 128, ncell, r_out, o_cs, o_cs < 288 ? o_cs : 288, 1, y3, CS_Y3X, CS_Y3X);
    }
    if (x_c > 288 && L.rc == BSR_OK) {      // the block output keeps the wider of x / y (model.py:105-113): channels the GEMM does not cover
      glue_begin("lrelu_copy");
      hipLaunchKernelGGL(bsr::lrelu_copy_kernel, dim3((unsigned)((ncell * (x_c - 288) + 255) / 256)), dim3(256), 0, s, x, x_cs, r_out, o_cs, 288, x_c, ncell);
      glue_end("lrelu_copy");
    }
  };
  if ((H8 * W8) % 128 != 0) return fail(BSR_ERR_ARG, "bsr_forward: (H/8)*(W/8) must be a multiple of 128");
  res_block(0, ws + p.xa, V.cs_a, V.c_a);
  res_block(1, ws + p.r[0], V.cs_r, V.c_r);
  res_block(2, ws + p.r[1], V.cs_r, V.c_r);

  // greyscale decoder: up1..3 = ConvT (model.py:243-245)
  L.conv<3, 3, 1, true, 1, 24, 1>(K_CONVT, "up1", ws + p.r[2], V.cs_r, 0, V.cs_r, H8, W8, ws + p.c2, 160, 0, 96, 1, io_out);
  L.conv<3, 3, 1, true, 2, 32, 1>(K_CONVT_NI2, "up2", ws + p.c2, 160, 0, 160, H4, W4, ws + p.c3, 128, 0, 64, 1, io_both);
  L.conv<3, 3, 1, true, 2, 32, 1>(K_CONVT_NI2, "up3", ws + p.c3, 128, 0, 128, H2, W2, ws + p.ybuf, 64, 0, 64, 1, io_both);
  // heads conv2 (mask) / conv3 (con): 7x7, 64 -> 1 each (model.py:246-247) as one 7x1 MFMA conv with N = (kx, head); the 7 horizontal
  // taps + tanh / gs / mask22 (model.py:246-252) either inside the same kernel (row-strip workgroups, when the batch has enough
  // strips to fill the chip) or by heads_post_kernel from the qh scratch tensor — bit-identical results either way
  L.heads(ws + p.ybuf, H, W, ws + p.qh, inputs, gs, mask22, npix);
  // bmask / x_hole (model.py:256-259)
  glue_begin("bmask_xhole");
  hipLaunchKernelGGL(bsr::bmask_xhole_kernel, dim3((unsigned)ncell), dim3(64), 0, s, gs, inputs, H, W, ws + p.r[2], V.cs_r, V.c_r, ws + p.xh,
                     V.cs_h, ws + p.probe);
  glue_end("bmask_xhole");
  if (V.tsm) share_layer(ws + p.xh, V.cs_h, V.c_r, ws + p.xh, V.cs_h, V.c_r + 1);    // model_with_TSM.py:292-293

  res_block(3, ws + p.xh, V.cs_h, V.c_h);
  res_block(4, ws + p.r[3], V.cs_h, V.c_h);
  res_block(5, ws + p.r[4], V.cs_h, V.c_h);

  // colour decoder (model.py:264-269)
  L.conv<3, 3, 1, true, 2, 24, 1>(K_CONVT, "clr_up1", ws + p.r[5], V.cs_h, 0, V.cs_h, H8, W8, ws + p.f1, 128, 0, 128, 1, io_out);
  L.conv<3, 3, 1, true, 1, 32, 1>(K_CONVT, "clr_up2", ws + p.f1, 128, 0, 128, H4, W4, ws + p.f2, 96, 0, 96, 1, io_both);
  L.conv<3, 3, 1, true, 2, 32, 1>(K_CONVT_NI2, "clr_up3", ws + p.f2, 96, 0, 96, H2, W2, ws + p.cf, CS_CF, 0, 64, 1, io_both);
  // clr_conv1 (3x3 over cat[gs, f]) + clr_conv2 + clr_conv3 + dif, one kernel (model.py:267-269,288)
  L.conv16<3, 3, true, true, 2>(K_CONV3, "clr_conv1", ws + p.cf, CS_CF, H, W, nullptr, 0, 1, gs, inputs, con_rgb, dif, packed);
  if (L.rc == BSR_OK) h->ran = true;
  return L.rc;
}

int bsr_forward(bsr_handle* h, const float* inputs, const float* uv, int B, int H, int W, float* gs, float* con_rgb, float* mask22,
                float* dif, void* stream) {
  return forward_impl(h, inputs, uv, nullptr, 1, 0, B, H, W, gs, con_rgb, mask22, dif, stream);
}

int bsr_forward_packed(bsr_handle* h, const float* inputs, const float* uv, int B, int H, int W, float* gs, float* con_rgb_dif, float* mask22,
                       void* stream) {
  if (con_rgb_dif == nullptr) return fail(BSR_ERR_ARG, "bsr_forward_packed: null con_rgb_dif");
  return forward_impl(h, inputs, uv, nullptr, 1, 0, B, H, W, gs, nullptr, mask22, nullptr, stream, con_rgb_dif);
}

int bsr_forward_tsm(bsr_handle* h, const float* inputs, const float* uv, const float* reg, int B, int H, int W, int frame, int share,
                    float* gs, float* con_rgb, float* mask22, float* dif, void* stream) {
  if (reg == nullptr) return fail(BSR_ERR_ARG, "bsr_forward_tsm: null reg");
  return forward_impl(h, inputs, uv, reg, frame, share != 0, B, H, W, gs, con_rgb, mask22, dif, stream);
}

int bsr_prep_rows(int device, const void* d_blob, size_t blob_bytes, size_t rows_off, size_t grid_off, int B, int S, float* out, float* hull_tmp,
                  void* stream) {
  if (d_blob == nullptr || out == nullptr || hull_tmp == nullptr) return fail(BSR_ERR_ARG, "bsr_prep_rows: null argument");
  if (B <= 0 || S <= 0 || (S * S) % 256 != 0) return fail(BSR_ERR_ARG, "bsr_prep_rows: B must be positive and S*S a multiple of 256");
  if (rows_off % 8 != 0 || grid_off % 8 != 0) return fail(BSR_ERR_ARG, "bsr_prep_rows: table offsets must be 8-byte aligned");
  // the two tables the kernels index directly must lie inside the blob (what the row records point to — images, triangle tables — is
  // device memory this library cannot read back cheaply: prep.py validates every record against blob_bytes before the upload)
  if (rows_off > blob_bytes || (size_t)B * sizeof(bsr::PrepRow) > blob_bytes - rows_off || grid_off > blob_bytes ||
      (size_t)S * sizeof(double) > blob_bytes - grid_off)
    return fail(BSR_ERR_ARG, "bsr_prep_rows: the row / grid tables do not fit in blob_bytes");
  DeviceGuard guard(device);
  HIP_TRY(guard.err);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned char* blob = static_cast<const unsigned char*>(d_blob);
  const dim3 grid((unsigned)(S * S / 256), (unsigned)B);
  hipLaunchKernelGGL(bsr::prep_rows_kernel, grid, dim3(256), 0, s, blob, reinterpret_cast<const bsr::PrepRow*>(blob + rows_off),
                     reinterpret_cast<const double*>(blob + grid_off), S, out, hull_tmp);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(bsr::prep_blur_kernel, grid, dim3(256), 0, s, hull_tmp, S, out);
  HIP_TRY(hipGetLastError());
  return BSR_OK;
}

int bsr_png_unfilter(int device, void* d_blob, size_t blob_bytes, size_t items_off, int n, void* stream) {
  if (d_blob == nullptr) return fail(BSR_ERR_ARG, "bsr_png_unfilter: null argument");
  if (n <= 0 || items_off % 8 != 0) return fail(BSR_ERR_ARG, "bsr_png_unfilter: n must be positive and items_off 8-byte aligned");
  // the item table must lie inside the blob; what its records point to is validated by the caller before the upload (prep.py), as for bsr_prep_rows
  if (items_off > blob_bytes || (size_t)n * sizeof(bsr::UnfilterItem) > blob_bytes - items_off)
    return fail(BSR_ERR_ARG, "bsr_png_unfilter: the item table does not fit in blob_bytes");
  DeviceGuard guard(device);
  HIP_TRY(guard.err);
  unsigned char* blob = static_cast<unsigned char*>(d_blob);
  hipLaunchKernelGGL(bsr::png_unfilter_kernel, dim3((unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream), blob,
                     reinterpret_cast<const bsr::UnfilterItem*>(blob + items_off));
  HIP_TRY(hipGetLastError());
  return BSR_OK;
}

size_t bsr_png_file_bytes(int H, int W) {
  bsr::PngGeom g;
  return bsr::png_geometry(H, W, &g) ? (size_t)g.file_bytes : 0;
}

size_t bsr_png_scratch_bytes(int B) { return B > 0 ? (size_t)B * bsr::kPngSub * 4 * sizeof(unsigned long long) : 0; }

int bsr_png_encode(int device, const unsigned char* pixels, int B, int H, int W, unsigned char* out, size_t out_stride, void* scratch, void* stream) {
  if (pixels == nullptr || out == nullptr || scratch == nullptr) return fail(BSR_ERR_ARG, "bsr_png_encode: null argument");
  bsr::PngGeom g;
  if (B <= 0 || !bsr::png_geometry(H, W, &g)) return fail(BSR_ERR_ARG, "bsr_png_encode: B, H, W must be positive, W <= 5461 and H <= 65535");
  if (out_stride < g.file_bytes) return fail(BSR_ERR_ARG, "bsr_png_encode: out_stride is smaller than bsr_png_file_bytes(H, W)");
  if (reinterpret_cast<uintptr_t>(scratch) % 8 != 0) return fail(BSR_ERR_ARG, "bsr_png_encode: scratch must be 8-byte aligned");
  DeviceGuard guard(device);
  HIP_TRY(guard.err);
  bsr::PngFigs none{};
  HIP_TRY(bsr::launch_png_encode(pixels, none, B, g, out, out_stride, static_cast<unsigned long long*>(scratch), static_cast<hipStream_t>(stream)));
  return BSR_OK;
}

int bsr_png_encode_figs(int device, int n_figs, const float* const* figs, const float* const* muls, const float* scales, const int* channels,
                        const int* pixel_strides, const int* mul_strides, int B, int H, int Wf, unsigned char* out, size_t out_stride, void* scratch,
                        void* stream) {
  if (figs == nullptr || channels == nullptr || pixel_strides == nullptr || out == nullptr || scratch == nullptr)
    return fail(BSR_ERR_ARG, "bsr_png_encode_figs: null argument");
  if (n_figs < 1 || n_figs > bsr::kPngMaxFigs) return fail(BSR_ERR_ARG, "bsr_png_encode_figs: 1 to 8 figures per strip");
  bsr::PngGeom g;
  if (B <= 0 || Wf <= 0 || !bsr::png_geometry(H, n_figs * Wf, &g))
    return fail(BSR_ERR_ARG, "bsr_png_encode_figs: B, H, Wf must be positive, the strip's width n_figs * Wf <= 5461 and H <= 65535");
  if (out_stride < g.file_bytes) return fail(BSR_ERR_ARG, "bsr_png_encode_figs: out_stride is smaller than bsr_png_file_bytes(H, n_figs * Wf)");
  if (reinterpret_cast<uintptr_t>(scratch) % 8 != 0) return fail(BSR_ERR_ARG, "bsr_png_encode_figs: scratch must be 8-byte aligned");
  bsr::PngFigs f{};
  f.n = n_figs;
  f.Wf = Wf;
  for (int k = 0; k < n_figs; ++k) {
    if (figs[k] == nullptr || (channels[k] != 1 && channels[k] != 3) || pixel_strides[k] < channels[k])
      return fail(BSR_ERR_ARG, "bsr_png_encode_figs: every figure needs a pointer, 1 or 3 channels and a pixel stride of at least its channels");
    f.ptr[k] = figs[k];
    f.mul[k] = muls != nullptr ? muls[k] : nullptr;
    f.scale[k] = scales != nullptr ? scales[k] : 1.f;
    f.ch[k] = channels[k];
    f.ps[k] = pixel_strides[k];
    f.mps[k] = (mul_strides != nullptr && f.mul[k] != nullptr) ? mul_strides[k] : 1;
    if (f.mul[k] != nullptr && f.mps[k] < 1) return fail(BSR_ERR_ARG, "bsr_png_encode_figs: a multiplier needs a positive pixel stride");
  }
  DeviceGuard guard(device);
  HIP_TRY(guard.err);
  HIP_TRY(bsr::launch_png_encode(nullptr, f, B, g, out, out_stride, static_cast<unsigned long long*>(scratch), static_cast<hipStream_t>(stream)));
  return BSR_OK;
}

size_t bsr_ucb_post_scratch_bytes(int B, int S) {
  if (B <= 0 || (S != 32 && S != 64 && S != 128 && S != 256)) return 0;
  return (size_t)B * bsr::ucb_item_scratch_bytes(S);
}

int bsr_ucb_post(int device, const float* rows10, const unsigned char* masks, const float* boxes, int B, int S, float* losses,
                 unsigned char* strips, float* figs, int* status, void* scratch, void* stream) {
  if (rows10 == nullptr || masks == nullptr || boxes == nullptr || losses == nullptr || strips == nullptr || status == nullptr || scratch == nullptr)
    return fail(BSR_ERR_ARG, "bsr_ucb_post: null argument");
  if (B <= 0 || (S != 32 && S != 64 && S != 128 && S != 256))
    return fail(BSR_ERR_ARG, "bsr_ucb_post: B must be positive and S one of 32, 64, 128, 256 (reference: 256)");
  if (reinterpret_cast<uintptr_t>(scratch) % 256 != 0) return fail(BSR_ERR_ARG, "bsr_ucb_post: scratch must be 256-byte aligned");
  DeviceGuard guard(device);
  HIP_TRY(guard.err);
  HIP_TRY(bsr::launch_ucb_post(rows10, masks, boxes, B, S, losses, strips, figs, status, scratch, static_cast<hipStream_t>(stream)));
  return BSR_OK;
}

int bsr_clock_trace(int device, unsigned long long* out, int samples, int spin, const int* stop, int* taken, void* stream) {
  if (out == nullptr || stop == nullptr || taken == nullptr || samples <= 0 || spin < 0) return fail(BSR_ERR_ARG, "bsr_clock_trace: bad argument");
  DeviceGuard guard(device);
  HIP_TRY(guard.err);
  hipLaunchKernelGGL(bsr::clock_trace_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out, samples, spin, stop, taken);
  HIP_TRY(hipGetLastError());
  return BSR_OK;
}

int bsr_debug_split_qkv(const float* qkv, void* qkv_split, int B, int tokens, void* stream) {
  if (qkv == nullptr || qkv_split == nullptr || B <= 0 || tokens <= 0) return fail(BSR_ERR_ARG, "bsr_debug_split_qkv: bad argument");
  const size_t pairs = (size_t)B * tokens * 192;
  hipLaunchKernelGGL(bsr::a4_split_qkv_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), qkv,
                     static_cast<char*>(qkv_split), pairs);
  HIP_TRY(hipGetLastError());
  return BSR_OK;
}

int bsr_debug_attention_split(const void* qkv_split, float* y, int B, int tokens, int pv1, void* stream) {
  if (qkv_split == nullptr || y == nullptr) return fail(BSR_ERR_ARG, "bsr_debug_attention_split: null argument");
  if (B <= 0 || tokens <= 0 || tokens % 128 != 0) return fail(BSR_ERR_ARG, "bsr_debug_attention_split: tokens must be a positive multiple of 128");
  HIP_TRY(bsr::launch_nonlocal_attention_h16(static_cast<const float*>(qkv_split), y, B, tokens, static_cast<hipStream_t>(stream), pv1 != 0));
  return BSR_OK;
}

int bsr_debug_attention_dtype(const float* qkv, float* y, int B, int tokens, int dtype, void* stream) {
  if (qkv == nullptr || y == nullptr) return fail(BSR_ERR_ARG, "bsr_debug_attention: null argument");
  if (B <= 0 || tokens <= 0 || tokens % 128 != 0) return fail(BSR_ERR_ARG, "bsr_debug_attention: tokens must be a positive multiple of 128");
  if (dtype == BSR_DTYPE_F32) {
    HIP_TRY(bsr::launch_nonlocal_attention(qkv, y, B, tokens, static_cast<hipStream_t>(stream)));
  } else if (dtype == BSR_DTYPE_F32X3 || dtype == BSR_DTYPE_F16) {
    // the kernel of the 16-bit modes reads theta|phi|g as their producer leaves them (split into fp16 planes: attention_h16.h); this
    // hook splits a scratch copy first (allocated and freed here: a test hook, not a hot path)
    void* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)B * tokens * bsr::kA4TokBytes));
    int rc = bsr_debug_split_qkv(qkv, tmp, B, tokens, stream);
    if (rc == BSR_OK) rc = bsr_debug_attention_split(tmp, y, B, tokens, 0, stream);
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    hipFree(tmp);
    if (rc != BSR_OK) return rc;
    HIP_TRY(e);
  } else {
    return fail(BSR_ERR_ARG, "bsr_debug_attention: unknown dtype");
  }
  return BSR_OK;
}

int bsr_debug_attention_qw(const float* qkv, float* y, int B, int tokens, int qw, void* stream) {
  if (qkv == nullptr || y == nullptr) return fail(BSR_ERR_ARG, "bsr_debug_attention_qw: null argument");
  if (B <= 0 || tokens <= 0 || tokens % 128 != 0) return fail(BSR_ERR_ARG, "bsr_debug_attention_qw: tokens must be a positive multiple of 128");
  if (qw != 0 && qw != 1 && qw != 2 && qw != 4) return fail(BSR_ERR_ARG, "bsr_debug_attention_qw: qw must be 0 (automatic), 1, 2 or 4 query waves per workgroup");
  HIP_TRY(bsr::launch_nonlocal_attention(qkv, y, B, tokens, static_cast<hipStream_t>(stream), qw));
  return BSR_OK;
}

int bsr_debug_attention(const float* qkv, float* y, int B, int tokens, void* stream) {
  return bsr_debug_attention_dtype(qkv, y, B, tokens, BSR_DTYPE_F32, stream);
}

int bsr_probe(bsr_handle* h, const char* name, float* dst, size_t cap_floats, int shape4[4], void* stream) {
  if (h == nullptr || name == nullptr || dst == nullptr || shape4 == nullptr) return fail(BSR_ERR_ARG, "bsr_probe: null argument");
  if (!h->ran) return fail(BSR_ERR_STATE, "bsr_probe: no forward has run on this handle");
  DeviceGuard guard(h->device);
  HIP_TRY(guard.err);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const Plan& p = h->plan;
  const int B = h->B, H = h->H, W = h->W;
  struct Src { size_t off; int hh, ww, cs, coff, c; bool half; };
  const bool p16 = h->dtype == BSR_DTYPE_F16;          // the tensors the f16 mode keeps as fp16 (forward_impl)
  Src src{};
  std::string nm(name);
  auto res_idx = [&](const char* prefix) -> int {
    size_t n = strlen(prefix);
    if (nm.size() == n + 1 && nm.compare(0, n, prefix) == 0 && nm[n] >= '0' && nm[n] <= '5') return nm[n] - '0';
    return -1;
  };
  int i;
  if (nm == "x1") src = {p.x1, H, W, 32, 0, 32, p16};
  else if (nm == "x2") src = {p.c3, H / 2, W / 2, 128, 64, 64, p16};
  else if (nm == "x3") src = {p.c2, H / 4, W / 4, 160, 96, 64, p16};
  else if (nm == "x0") src = {p.xa, H / 8, W / 8, h->var.cs_a, 0, h->var.c_a};
  else if ((i = res_idx("res")) >= 0) src = {p.r[i], H / 8, W / 8, i < 3 ? h->var.cs_r : h->var.cs_h, 0, i < 3 ? h->var.c_r : h->var.c_h};
  else if ((i = res_idx("att")) >= 0) {
    // fused attention + w (fp32, full batches): the attention output stays in LDS, the att<i> slots hold whatever an earlier
    // forward left there — refuse rather than hand out stale data
    if (h->att_in_lds)
      return fail(BSR_ERR_STATE, "bsr_probe: att<i> does not exist for the last forward — attention and the `w` GEMM ran as one launch and the "
                                 "attention output never left LDS (create the handle with BSR_FUSE_ATTW=0 in the environment to probe it)");
    src = {p.att[i], H / 8, W / 8, 128, 0, 128};
  }
  else if ((i = res_idx("y3x")) >= 0) src = {p.y3[i], H / 8, W / 8, CS_Y3X, 0, CS_Y3X};
  else if (nm == "up1") src = {p.c2, H / 4, W / 4, 160, 0, 96, p16};
  else if (nm == "up2") src = {p.c3, H / 2, W / 2, 128, 0, 64, p16};
  else if (nm == "y") src = {p.ybuf, H, W, 64, 0, 64, p16};
  else if (nm == "d32") src = {p.probe, H / 8, W / 8, 2, 0, 1};
  else if (nm == "bmask") src = {p.probe, H / 8, W / 8, 2, 1, 1};
  else if (nm == "xh") src = {p.xh, H / 8, W / 8, h->var.cs_h, 0, h->var.c_h};
  else if (nm == "f1") src = {p.f1, H / 4, W / 4, 128, 0, 128, p16};
  else if (nm == "f2") src = {p.f2, H / 2, W / 2, 96, 0, 96, p16};
  else if (nm == "f") src = {p.cf, H, W, CS_CF, 0, 64, p16};
  else return fail(BSR_ERR_STATE, std::string("bsr_probe: unknown probe '") + name + "'");
  const size_t npix = (size_t)B * src.hh * src.ww;
  shape4[0] = B; shape4[1] = src.hh; shape4[2] = src.ww; shape4[3] = src.c;
  if (npix * src.c > cap_floats) return fail(BSR_ERR_ARG, "bsr_probe: destination too small");
  hipLaunchKernelGGL(bsr::slice_copy_kernel, dim3((unsigned)((npix * src.c + 255) / 256)), dim3(256), 0, s, h->ws + src.off, src.cs, src.coff,
                     src.c, dst, npix, src.half ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return BSR_OK;
}

}  // extern "C"
