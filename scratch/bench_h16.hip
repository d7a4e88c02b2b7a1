// Diagnostic harness (not product code): times one igemm_h16_kernel configuration on random data and reports where a wave's
// cycles go (prologue / main loop / epilogue).   hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/bench_h16.hip -o scratch/bench_h16
#define BSR_STAMPS 1
#include "../blindshadowremoval_amd/csrc/igemm_h16.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace bsr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KH, int KW, int S, bool TR, int NI, int CC, int INB, int NSPLIT, int TH = 4, int MI = 1, int WM = 4, int WN = 1>
int run(const char* name, int B, int H, int W, int Cin, int Cout) {
  using C = H16Cfg<KH, KW, S, TR, TH, 32, WM, WN, MI, NI, CC, INB, NSPLIT>;
  const int T = KH * KW, nchunk = Cin / CC, n_pad = ((Cout + C::BN - 1) / C::BN) * C::BN;
  const int Ho = TR ? 2 * H : H / S, Wo = TR ? 2 * W : W / S;
  size_t n_in = (size_t)B * H * W * Cin, n_out = (size_t)B * Ho * Wo * Cout, n_w = (size_t)nchunk * T * n_pad * C::LDP;
  std::vector<float> h_in(n_in);
  std::vector<_Float16> h_w(n_w * 2);
  for (auto& v : h_in) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : h_w) v = (_Float16)(((float)rand() / RAND_MAX - 0.5f) * 0.1f);
  float *d_in, *d_out, *d_w, *d_b;
  CK(hipMalloc(&d_in, n_in * 4)); CK(hipMalloc(&d_out, n_out * 4)); CK(hipMalloc(&d_w, n_w * 4)); CK(hipMalloc(&d_b, n_pad * 4));
  CK(hipMemcpy(d_in, h_in.data(), n_in * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, h_w.data(), n_w * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_b, 0, n_pad * 4));
  ConvArgs a{};
  a.in = d_in; a.in_cs = Cin; a.in_coff = 0; a.H = H; a.W = W; a.out = d_out; a.out_cs = Cout; a.out_coff = 0; a.Ho = Ho; a.Wo = Wo;
  a.w = d_w; a.bias = d_b; a.nchunk = nchunk; a.n_pad = n_pad; a.n_store = Cout; a.pad_t = (S == 2) ? 0 : (KH - 1) / 2; a.pad_l = (S == 2) ? 0 : (KW - 1) / 2; a.act = 1;
  const int mh = TR ? H : Ho, mw = TR ? W : Wo;
  size_t nblk = (size_t)(mw / 32) * (mh / TH) * B * (n_pad / C::BN);
  unsigned long long* d_st;
  CK(hipMalloc(&d_st, nblk * 16 * 8));
  a.stamps = d_st;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 6; ++it) {
    CK(hipEventRecord(e0));
    CK((launch_igemm_h16<KH, KW, S, TR, TH, 32, WM, WN, MI, NI, CC, INB, NSPLIT>(a, B, 0)));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it > 0) best = std::min(best, ms);
  }
  std::vector<unsigned long long> st(nblk * 16);
  CK(hipMemcpy(st.data(), d_st, nblk * 16 * 8, hipMemcpyDeviceToHost));
  double pro = 0, loop = 0, epi = 0, epi_issue = 0, rt = 0;
  for (size_t i = 0; i < nblk * 4; ++i) { pro += st[i * 4]; loop += st[i * 4 + 1]; epi_issue += (st[i * 4 + 2] & 0xffffffffull); rt += (double)(st[i * 4 + 2] >> 32); epi += st[i * 4 + 3]; }
  double nw = nblk * 4.0;
  double flops = 2.0 * B * (TR ? H * W : Ho * Wo) * (double)T * Cin * Cout;
  double mfma_per_wave = (double)nchunk * T * (CC / 16) * NI * MI * (NSPLIT == 2 ? 3 : 1);
  double bytes = (double)n_in * 4 + (double)n_out * 4;
  printf("%-12s %7.1f us  %6.1f TFLOP/s  %5.2f TB/s(alg) | blocks %zu, per wave (cycles): prologue %.0f  loop %.0f  epilogue %.0f (issue %.0f) | loop cycles per MFMA %.1f | clock %.2f GHz\n",
         name, best * 1e3, flops / best / 1e9, bytes / best / 1e9, nblk, pro / nw, loop / nw, epi / nw, epi_issue / nw, loop / nw / mfma_per_wave, (pro + loop + epi) / rt * 0.1);
  hipFree(d_in); hipFree(d_out); hipFree(d_w); hipFree(d_b); hipFree(d_st);
  return 0;
}

int main() {
  if (run<3, 3, 1, true, 2, 32, 1, 2>("up3 x3", 32, 128, 128, 128, 64)) return 1;
  if (run<3, 3, 1, true, 1, 32, 1, 2, 4, 2, 2, 2>("up3 x3 w22", 32, 128, 128, 128, 64)) return 1;
  if (run<3, 3, 1, true, 2, 32, 1, 1>("up3 f16", 32, 128, 128, 128, 64)) return 1;
  if (run<3, 3, 1, true, 1, 32, 1, 1, 4, 2, 2, 2>("up3 f16 w22", 32, 128, 128, 128, 64)) return 1;
  if (run<3, 3, 1, true, 1, 32, 1, 2, 4, 2, 2, 2>("up2 x3 w22", 32, 64, 64, 160, 64)) return 1;
  if (run<3, 3, 1, true, 2, 32, 1, 2>("clr_up3 x3", 32, 128, 128, 96, 64)) return 1;
  if (run<3, 3, 1, true, 2, 32, 1, 2>("up2 x3", 32, 64, 64, 160, 64)) return 1;
  if (run<3, 3, 1, true, 1, 32, 1, 2>("clr_up2 x3", 32, 64, 64, 128, 96)) return 1;
  if (run<3, 3, 1, false, 2, 32, 1, 2>("res.conv2 x3", 32, 32, 32, 128, 128)) return 1;
  if (run<3, 3, 1, false, 2, 32, 1, 1>("res.conv2 f16", 32, 32, 32, 128, 128)) return 1;
  if (run<3, 3, 2, false, 2, 16, 1, 1>("down1 f16", 32, 256, 256, 32, 64)) return 1;
  if (run<3, 3, 2, false, 2, 16, 1, 2>("down1 x3", 32, 256, 256, 32, 64)) return 1;
  if (run<3, 3, 2, false, 2, 16, 1, 2>("down2 x3", 32, 128, 128, 64, 64)) return 1;
  return 0;
}
