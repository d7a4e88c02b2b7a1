// Diagnostic harness (not product code): times one igemm_conv_kernel configuration on random data and, with
// -DBSR_STAMPS, reports where a wave's cycles go (prologue / main loop / epilogue).
#define BSR_STAMPS 1
#include "../blindshadowremoval_amd/csrc/igemm_conv.h"
#include "../blindshadowremoval_amd/csrc/gemm_nloop.h"
#include "../blindshadowremoval_amd/csrc/conv_n16.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace bsr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static int g_mode = 0;
template <int KH, int KW, int S, bool TR, int NI, int CC, int INB, int WN = 1>
int run(const char* name, int B, int H, int W, int Cin, int Cout) {
  using C = ConvCfg<KH, KW, S, TR, 4, 32, 4, WN, 1, NI, CC, INB>;
  constexpr int NW = 4 * WN;
  const int T = KH * KW, nchunk = Cin / CC, n_pad = ((Cout + C::BN - 1) / C::BN) * C::BN;
  const int Ho = TR ? 2 * H : H / S, Wo = TR ? 2 * W : W / S;
  size_t n_in = (size_t)B * H * W * Cin, n_out = (size_t)B * Ho * Wo * Cout, n_w = (size_t)nchunk * T * n_pad * (CC + 4);
  std::vector<float> h_in(n_in), h_w(n_w);
  for (auto& v : h_in) v = g_mode == 1 ? 0.f : (g_mode == 2 ? 0.25f : (float)rand() / RAND_MAX - 0.5f);
  for (auto& v : h_w) v = g_mode == 1 ? 0.f : (g_mode == 2 ? 0.03125f : ((float)rand() / RAND_MAX - 0.5f) * 0.1f);
  float *d_in, *d_out, *d_w, *d_b;
  CK(hipMalloc(&d_in, n_in * 4)); CK(hipMalloc(&d_out, n_out * 4)); CK(hipMalloc(&d_w, n_w * 4)); CK(hipMalloc(&d_b, n_pad * 4));
  CK(hipMemcpy(d_in, h_in.data(), n_in * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, h_w.data(), n_w * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_b, 0, n_pad * 4));
  ConvArgs a{};
  a.in = d_in; a.in_cs = Cin; a.in_coff = 0; a.H = H; a.W = W; a.out = d_out; a.out_cs = Cout; a.out_coff = 0; a.Ho = Ho; a.Wo = Wo;
  a.w = d_w; a.bias = d_b; a.nchunk = nchunk; a.n_pad = n_pad; a.n_store = Cout; a.pad_t = (S == 2) ? 0 : (KH - 1) / 2; a.pad_l = (S == 2) ? 0 : (KW - 1) / 2; a.act = 1;
  const int mh = TR ? H : Ho, mw = TR ? W : Wo;
  size_t nblk = (size_t)(mw / 32) * (mh / 4) * B * (n_pad / C::BN);
  unsigned long long* d_st;
  CK(hipMalloc(&d_st, nblk * NW * 4 * 8));
  a.stamps = d_st;
  const int nsteps_h = nchunk * T;
  unsigned long long* d_st2; CK(hipMalloc(&d_st2, (size_t)64 * NW * nsteps_h * 3 * 8)); CK(hipMemset(d_st2, 0, (size_t)64 * NW * nsteps_h * 3 * 8));
  a.stamps2 = getenv("BSR_TIMELINE") ? d_st2 : nullptr;
  unsigned long long* d_st3; CK(hipMalloc(&d_st3, nblk * NW * 6 * 8)); a.stamps3 = d_st3;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  const int iters = getenv("BSR_ITERS") ? atoi(getenv("BSR_ITERS")) : 6;      // many back-to-back launches = the sustained clock
  for (int it = 0; it < iters; ++it) {
    CK(hipEventRecord(e0));
    if (getenv("BSR_SOLO")) {          // one workgroup per CU (LDS request > half the CU's): a wave's matrix rate WITHOUT a SIMD partner
      auto kern = igemm_conv_kernel<KH, KW, S, TR, 4, 32, 4, WN, 1, NI, CC, INB>;
      const int smem = 100 * 1024;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      ConvArgs b = a;
      b.tiles_x = mw / 32; b.tiles_y = mh / 4;
      hipLaunchKernelGGL(kern, dim3(b.tiles_x * b.tiles_y * B, (Cout + C::BN - 1) / C::BN), dim3(C::NT), smem, 0, b, GemmTailArgs{});
      CK(hipGetLastError());
    } else
    CK((launch_igemm_conv<KH, KW, S, TR, 4, 32, 4, WN, 1, NI, CC, INB>(a, B, 0)));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it > 0) best = std::min(best, ms);
  }
  std::vector<unsigned long long> st(nblk * NW * 4);
  CK(hipMemcpy(st.data(), d_st, nblk * NW * 4 * 8, hipMemcpyDeviceToHost));
  double pro = 0, loop = 0, epi = 0, epi_issue = 0;
  double rt = 0;
  for (size_t i = 0; i < nblk * NW; ++i) { pro += st[i * 4]; loop += st[i * 4 + 1]; epi_issue += (st[i * 4 + 2] & 0xffffffffull); rt += (double)(st[i * 4 + 2] >> 32); epi += st[i * 4 + 3]; }
  printf("   wave lifetime %.0f cycles = %.0f ticks of the 100 MHz realtime counter -> shader clock %.2f GHz\n", (pro + loop + epi) / (nblk * (double)NW), rt / (nblk * (double)NW), (pro + loop + epi) / rt * 0.1);
  double nw = nblk * (double)NW;
  double flops = 2.0 * B * (TR ? H * W : Ho * Wo) * (double)T * Cin * Cout;
  double mfma_per_wave = (double)nchunk * T * (CC / 2) * NI;       // MFMAs a wave issues
  printf("%-10s %7.1f us  %6.1f TFLOP/s | blocks %zu, per wave (cycles): prologue %.0f  loop %.0f  epilogue %.0f (issue %.0f) | loop ticks per MFMA %.2f\n",
         name, best * 1e3, flops / best / 1e9, nblk, pro / nw, loop / nw, epi / nw, epi_issue / nw, loop / nw / mfma_per_wave);
  {   // where the prologue goes (cycles per wave)
    std::vector<unsigned long long> s3(nblk * NW * 6);
    CK(hipMemcpy(s3.data(), d_st3, s3.size() * 8, hipMemcpyDeviceToHost));
    double dd[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < nblk * NW; ++i) for (int k = 0; k < 5; ++k) dd[k] += (double)(s3[i * 6 + k + 1] - s3[i * 6 + k]);
    printf("   prologue split: address set-up %.0f | loads issued %.0f | loads landed + LDS written %.0f | barrier %.0f | accumulators + first fragments %.0f\n",
           dd[0] / nw, dd[1] / nw, dd[2] / nw, dd[3] / nw, dd[4] / nw);
    hipFree(d_st3);
  }
  if (getenv("BSR_TIMELINE") && nblk >= 1088) {      // per-step timeline of 64 mid-kernel workgroups: matrix phase and barrier wait per step
    std::vector<unsigned long long> t2((size_t)64 * NW * nsteps_h * 3);
    CK(hipMemcpy(t2.data(), d_st2, t2.size() * 8, hipMemcpyDeviceToHost));
    double work = 0, wait = 0, gap = 0; size_t n = 0, ng = 0;
    std::vector<double> works;
    for (size_t bw = 0; bw < (size_t)64 * NW; ++bw)
      for (int st = 0; st < nsteps_h; ++st) {
        const unsigned long long* e = &t2[(bw * nsteps_h + st) * 3];
        if (e[2] == 0) continue;
        work += (double)(e[1] - e[0]); wait += (double)(e[2] - e[1]); works.push_back((double)(e[1] - e[0])); ++n;
        if (st + 1 < nsteps_h) { const unsigned long long* f = e + 3; if (f[2] != 0) { gap += (double)(f[0] - e[2]); ++ng; } }
      }
    std::sort(works.begin(), works.end());
    printf("   timeline (%zu steps): matrix phase %.0f cycles avg (p10 %.0f, p50 %.0f, p90 %.0f; %d MFMAs -> %.1f per MFMA), barrier wait %.0f, between steps %.0f\n", n, work / n,
           works[n / 10], works[n / 2], works[n * 9 / 10], (CC / 2) * NI, work / n / ((CC / 2) * NI), wait / n, gap / (ng ? ng : 1));
  }
  hipFree(d_in); hipFree(d_out); hipFree(d_w); hipFree(d_b); hipFree(d_st); hipFree(d_st2);
  return 0;
}

template <int KH, int KW, bool GS, bool TAIL, int RW>
int run_n16(const char* name, int B, int H, int W) {
  using C = ConvN16Cfg<KH, KW, GS, TAIL, RW>;
  size_t npx = (size_t)B * H * W;
  float *d_in, *d_out, *d_w, *d_b, *d_gs, *d_wgs, *d_tail, *d_inp, *d_rgb, *d_dif;
  CK(hipMalloc(&d_in, npx * 64 * 4)); CK(hipMalloc(&d_out, npx * 16 * 4)); CK(hipMalloc(&d_w, 2 * C::W_FLOATS * 4)); CK(hipMalloc(&d_b, 64));
  CK(hipMalloc(&d_gs, npx * 4)); CK(hipMalloc(&d_wgs, 1024)); CK(hipMalloc(&d_tail, 2048)); CK(hipMalloc(&d_inp, npx * 12)); CK(hipMalloc(&d_rgb, npx * 12)); CK(hipMalloc(&d_dif, npx * 4));
  std::vector<float> h_in(npx * 64), h_w(2 * C::W_FLOATS);
  for (auto& v : h_in) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : h_w) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  CK(hipMemcpy(d_in, h_in.data(), npx * 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, h_w.data(), 2 * C::W_FLOATS * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_b, 0, 64)); CK(hipMemset(d_gs, 0, npx * 4)); CK(hipMemset(d_wgs, 0, 1024)); CK(hipMemset(d_tail, 0, 2048)); CK(hipMemset(d_inp, 0, npx * 12));
  ConvN16Args a{};
  a.in = d_in; a.in_cs = 64; a.H = H; a.W = W; a.w = d_w; a.bias = d_b; a.out = d_out; a.out_cs = 16; a.act = 1; a.pad_t = (KH - 1) / 2; a.pad_l = (KW - 1) / 2;
  a.gs = d_gs; a.w_gs = d_wgs; a.tail_w = d_tail; a.inputs = d_inp; a.con_rgb = d_rgb; a.dif = d_dif;
  size_t nblk = (size_t)(W / 32) * (H / C::TH) * B;
  unsigned long long* d_st; CK(hipMalloc(&d_st, nblk * 16 * 8)); a.stamps = d_st;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0)); CK((launch_conv_n16<KH, KW, GS, TAIL, RW>(a, B, 0))); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it > 0) best = std::min(best, ms);
  }
  std::vector<unsigned long long> st(nblk * 16);
  CK(hipMemcpy(st.data(), d_st, nblk * 16 * 8, hipMemcpyDeviceToHost));
  double pro = 0, loop = 0, epi = 0, rt = 0;
  for (size_t i = 0; i < nblk * 4; ++i) { pro += st[i * 4]; loop += st[i * 4 + 1]; rt += st[i * 4 + 2]; epi += st[i * 4 + 3]; }
  // persistent kernel: 512 workgroups wrote stamps, each covering nblk / 512 tiles
  const size_t ntile = nblk;
  nblk = std::min<size_t>(nblk, 512);
  pro = loop = epi = rt = 0;
  for (size_t i = 0; i < nblk * 4; ++i) { pro += st[i * 4]; loop += st[i * 4 + 1]; rt += st[i * 4 + 2]; epi += st[i * 4 + 3]; }
  double nw = nblk * 4.0;
  double mf = 2.0 * KH * KW * 2 * 4 * (2 * RW) * ((double)ntile / nblk);
  printf("%-12s %7.1f us  %6.1f TFLOP/s | workgroups %zu | per wave: prologue %.0f  loop %.0f (%.1f cyc/MFMA)  epilogue %.0f | clock %.2f GHz, lifetime %.1f us\n", name, best * 1e3,
         2.0 * npx * KH * KW * 64 * 16 / best / 1e9, nblk, pro / nw, loop / nw, loop / nw / mf, epi / nw, (pro + loop + epi) / rt * 0.1, rt / nw * 0.01);
  return 0;
}

int run_gemm(const char* name, int pixels, int N, int nsplit, bool res) {
  const int K = 128, n_pad = ((N + 31) / 32 + 3) * 32;
  size_t n_in = (size_t)pixels * K, n_out = (size_t)pixels * N, n_w = (size_t)4 * n_pad * 36;
  float *d_in, *d_out, *d_w, *d_b, *d_r;
  CK(hipMalloc(&d_in, n_in * 4)); CK(hipMalloc(&d_out, n_out * 4)); CK(hipMalloc(&d_w, n_w * 4)); CK(hipMalloc(&d_b, n_pad * 4)); CK(hipMalloc(&d_r, n_out * 4));
  std::vector<float> h_in(n_in), h_w(n_w);
  for (auto& v : h_in) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : h_w) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  CK(hipMemcpy(d_in, h_in.data(), n_in * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, h_w.data(), n_w * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_b, 0, n_pad * 4)); CK(hipMemset(d_r, 0, n_out * 4));
  ConvArgs a{};
  a.in = d_in; a.in_cs = K; a.out = d_out; a.out_cs = N; a.w = d_w; a.bias = d_b; a.nchunk = 4; a.n_pad = n_pad; a.n_store = N; a.act = 1;
  if (res) { a.res1 = d_r; a.res1_cs = N; a.res1_c = N < 288 ? N : 288; }
  size_t nblk = (size_t)(pixels / 128) * nsplit;
  unsigned long long* d_st; CK(hipMalloc(&d_st, nblk * 16 * 8)); a.stamps = d_st;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 6; ++it) {
    CK(hipEventRecord(e0)); CK((launch_gemm_nloop<3, 4>(a, pixels, nsplit, 0))); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it > 0) best = std::min(best, ms);
  }
  std::vector<unsigned long long> st(nblk * 16);
  CK(hipMemcpy(st.data(), d_st, nblk * 16 * 8, hipMemcpyDeviceToHost));
  double pro = 0, loop = 0, epi = 0, rt = 0;
  for (size_t i = 0; i < nblk * 4; ++i) { pro += st[i * 4]; loop += st[i * 4 + 1]; epi += st[i * 4 + 3]; rt += (double)(st[i * 4 + 2] >> 32); }
  double nw = nblk * 4.0;
  printf("%-12s %7.1f us  %6.1f TFLOP/s | blocks %zu | per wave: prologue %.0f  loop(excl epi) %.0f  epilogues %.0f | clock %.2f GHz, lifetime %.1f us\n", name, best * 1e3,
         2.0 * pixels * K * N / best / 1e9, nblk, pro / nw, loop / nw, epi / nw, (pro + loop + epi) / rt * 0.1, rt / nw * 0.01);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1) g_mode = atoi(argv[1]);
  if (argc > 2 && argv[2][0] == 'u') {            // clock cross-check (round 3): the dominant instantiation at two dispatch lengths
    if (run<3, 3, 1, true, 2, 32, 1>("up3 B=32", 32, 128, 128, 128, 64)) return 1;
    if (run<3, 3, 1, true, 1, 32, 1, 2>("up3 8w", 32, 128, 128, 128, 64)) return 1;          // 8 waves: WN = 2, NI = 1 per wave
    if (run<3, 3, 1, false, 2, 32, 1>("conv2", 32, 32, 32, 128, 128)) return 1;
    if (run<3, 3, 1, false, 1, 32, 1, 2>("conv2 8w", 32, 32, 32, 128, 128)) return 1;
    if (run<3, 3, 1, false, 2, 32, 1, 2>("conv2 8w ni2", 32, 32, 32, 128, 128)) return 1;
    if (run<3, 3, 2, false, 2, 16, 1>("down1", 32, 256, 256, 32, 64)) return 1;
    if (run<3, 3, 2, false, 1, 16, 1, 2>("down1 8w", 32, 256, 256, 32, 64)) return 1;
    return 0;
  }
  if (argc > 2 && argv[2][0] == 's') {            // the stride-2 encoder convs
    if (run<3, 3, 2, false, 2, 16, 1>("down1", 32, 256, 256, 32, 64)) return 1;
    if (run<3, 3, 2, false, 2, 32, 1>("down1/cc32", 32, 256, 256, 32, 64)) return 1;
    if (run<3, 3, 2, false, 2, 16, 1>("down2", 32, 128, 128, 64, 64)) return 1;
    if (run<3, 3, 2, false, 2, 32, 1>("down2/cc32", 32, 128, 128, 64, 64)) return 1;
    if (run<3, 3, 2, false, 3, 16, 1>("down3", 32, 64, 64, 64, 96)) return 1;
    if (run<3, 3, 2, false, 1, 16, 1>("down3/ni1", 32, 64, 64, 64, 96)) return 1;
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'n') {            // only the 16-channel kernels
    if (run_n16<3, 3, true, true, 2>("clr_conv1", 32, 256, 256)) return 1;
    if (run_n16<7, 1, false, false, 2>("heads", 32, 256, 256)) return 1;
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'g') {            // only the resident-activation GEMMs
    if (run_gemm("c3q", 32768, 672, 2, false)) return 1;
    if (run_gemm("c3q+res", 32768, 672, 2, true)) return 1;
    if (run_gemm("c3q+res/3", 32768, 672, 3, true)) return 1;
    if (run_gemm("w/3", 32768, 288, 3, true)) return 1;
    if (run_gemm("c3q/split1", 32768, 672, 1, false)) return 1;
    if (run_gemm("w", 32768, 288, 2, true)) return 1;
    if (run_gemm("w/nores", 32768, 288, 2, false)) return 1;
    return 0;
  }
  if (run_n16<3, 3, true, true, 2>("clr_conv1", 32, 256, 256)) return 1;
  if (run_n16<3, 3, true, true, 1>("clr_conv1/rw1", 32, 256, 256)) return 1;
  if (run_n16<7, 1, false, false, 2>("heads", 32, 256, 256)) return 1;
  if (run_n16<7, 1, false, false, 1>("heads/rw1", 32, 256, 256)) return 1;
  if (run_gemm("c3q", 32768, 672, 2, false)) return 1;
  if (run_gemm("c3q/split1", 32768, 672, 1, false)) return 1;
  if (run_gemm("w", 32768, 288, 2, true)) return 1;
  if (run_gemm("w/nores", 32768, 288, 2, false)) return 1;
  if (run<3, 3, 1, true, 2, 32, 1>("up3", 32, 128, 128, 128, 64)) return 1;
  if (run<3, 3, 1, true, 2, 32, 2>("up3/inb2", 32, 128, 128, 128, 64)) return 1;
  if (run<3, 3, 1, true, 1, 32, 1>("up3/ni1", 32, 128, 128, 128, 64)) return 1;
  if (run<3, 3, 1, false, 2, 32, 1>("res.conv2", 32, 32, 32, 128, 128)) return 1;
  if (run<3, 3, 1, false, 4, 32, 1>("conv2/ni4", 32, 32, 32, 128, 128)) return 1;
  if (run<1, 1, 1, false, 3, 32, 3>("res.conv3", 32, 32, 32, 128, 264)) return 1;
  if (run<1, 1, 1, false, 4, 24, 3>("qkv", 32, 32, 32, 264, 384)) return 1;
  return 0;
}
