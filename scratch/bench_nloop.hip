// Diagnostic harness (not product code): the c3q GEMM of the 16-bit modes (gemm_nloop_kernel<3, 4, 2>, N = 288 fp32 + 384 split) on random
// data, with -DBSR_STAMPS: where a wave's cycles go, per blockIdx.y range.   hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/bench_nloop.hip -o /tmp/bn && /tmp/bn
#define BSR_STAMPS 1
#include "../blindshadowremoval_amd/csrc/gemm_nloop.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace bsr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int run(const char* name, int pixels, int nsplit, bool res, bool split) {
  const int K = 128, N = 672, n_pad = ((N + 31) / 32 + 3) * 32;
  size_t n_in = (size_t)pixels * K, n_w = (size_t)4 * n_pad * 36;
  float *d_in, *d_out, *d_w, *d_b, *d_r, *d_q;
  CK(hipMalloc(&d_in, n_in * 4)); CK(hipMalloc(&d_out, (size_t)pixels * 288 * 4)); CK(hipMalloc(&d_w, n_w * 4)); CK(hipMalloc(&d_b, n_pad * 4));
  CK(hipMalloc(&d_r, (size_t)pixels * 288 * 4)); CK(hipMalloc(&d_q, (size_t)pixels * 384 * 4));
  std::vector<float> h_in(n_in);
  std::vector<_Float16> h_w(n_w * 2);
  for (auto& v : h_in) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : h_w) v = (_Float16)(((float)rand() / RAND_MAX - 0.5f) * 0.1f);
  CK(hipMemcpy(d_in, h_in.data(), n_in * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, h_w.data(), n_w * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_b, 0, n_pad * 4)); CK(hipMemset(d_r, 0, (size_t)pixels * 288 * 4));
  ConvArgs a{};
  a.in = d_in; a.in_cs = K; a.out = d_out; a.out_cs = 288; a.w = d_w; a.bias = d_b; a.nchunk = 4; a.n_pad = n_pad; a.n_store = N; a.act = 0;
  a.out2 = d_q; a.out2_cs = 384; a.n_split = 288; a.n_store1 = 288; a.out2_split = split ? 1 : 0;
  if (res) { a.res1 = d_r; a.res1_cs = 288; a.res1_c = 288; }
  size_t nblk = (size_t)(pixels / 128) * nsplit;
  unsigned long long* d_st; CK(hipMalloc(&d_st, nblk * 16 * 8)); a.stamps = d_st;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 8; ++it) {
    CK(hipEventRecord(e0)); CK((launch_gemm_nloop<3, 4, 2>(a, pixels, nsplit, 0))); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it > 1) best = std::min(best, ms);
  }
  std::vector<unsigned long long> st(nblk * 16);
  CK(hipMemcpy(st.data(), d_st, nblk * 16 * 8, hipMemcpyDeviceToHost));
  printf("%-14s %7.1f us\n", name, best * 1e3);
  const size_t gx = pixels / 128;
  for (int y = 0; y < nsplit; ++y) {
    double pro = 0, loop = 0, epi = 0, rt = 0;
    for (size_t b = 0; b < gx; ++b)
      for (int w = 0; w < 4; ++w) {
        const unsigned long long* d = &st[((y * gx + b) * 4 + w) * 4];
        pro += d[0]; loop += d[1]; epi += d[3]; rt += (double)(d[2] >> 32);
      }
    const double nw = gx * 4.0;
    printf("    range %d per wave: prologue %.0f  loop(excl epi) %.0f  epilogues %.0f cycles | clock %.2f GHz, lifetime %.1f us\n", y, pro / nw, loop / nw, epi / nw,
           (pro + loop + epi) / rt * 0.1, rt / nw * 0.01);
  }
  hipFree(d_in); hipFree(d_out); hipFree(d_w); hipFree(d_b); hipFree(d_r); hipFree(d_q); hipFree(d_st);
  return 0;
}

int run_conv1(const char* name, int pixels) {
  const int K = 288, N = 128, n_pad = ((N + 31) / 32 + 3) * 32;
  size_t n_in = (size_t)pixels * K, n_w = (size_t)9 * n_pad * 36;
  float *d_in, *d_out, *d_w, *d_b;
  CK(hipMalloc(&d_in, n_in * 4)); CK(hipMalloc(&d_out, (size_t)pixels * N * 4)); CK(hipMalloc(&d_w, n_w * 4)); CK(hipMalloc(&d_b, n_pad * 4));
  std::vector<float> h_in(n_in);
  std::vector<_Float16> h_w(n_w * 2);
  for (auto& v : h_in) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : h_w) v = (_Float16)(((float)rand() / RAND_MAX - 0.5f) * 0.1f);
  CK(hipMemcpy(d_in, h_in.data(), n_in * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, h_w.data(), n_w * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_b, 0, n_pad * 4));
  ConvArgs a{};
  a.in = d_in; a.in_cs = K; a.out = d_out; a.out_cs = N; a.w = d_w; a.bias = d_b; a.nchunk = 9; a.n_pad = n_pad; a.n_store = N; a.act = 1;
  size_t nblk = (size_t)(pixels / 128);
  unsigned long long* d_st; CK(hipMalloc(&d_st, nblk * 16 * 8)); a.stamps = d_st;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 8; ++it) {
    CK(hipEventRecord(e0)); CK((launch_gemm_nloop<4, 9, 2, 1>(a, pixels, 1, 0))); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it > 1) best = std::min(best, ms);
  }
  std::vector<unsigned long long> st(nblk * 16);
  CK(hipMemcpy(st.data(), d_st, nblk * 16 * 8, hipMemcpyDeviceToHost));
  double pro = 0, loop = 0, epi = 0, rt = 0, rtmax = 0;
  for (size_t b = 0; b < nblk; ++b)
    for (int w = 0; w < 4; ++w) {
      const unsigned long long* d = &st[(b * 4 + w) * 4];
      pro += d[0]; loop += d[1]; epi += d[3]; rt += (double)(d[2] >> 32); rtmax = std::max(rtmax, (double)(d[2] >> 32));
    }
  const double nw = nblk * 4.0;
  printf("%-14s %7.1f us | per wave: prologue %.0f  loop(excl epi) %.0f  epilogue %.0f cycles | clock %.2f GHz, lifetime mean %.1f us max %.1f us\n", name, best * 1e3, pro / nw, loop / nw, epi / nw,
         (pro + loop + epi) / rt * 0.1, rt / nw * 0.01, rtmax * 0.01);
  hipFree(d_in); hipFree(d_out); hipFree(d_w); hipFree(d_b); hipFree(d_st);
  return 0;
}

int main() {
  if (run_conv1("res.conv1", 32768)) return 1;
  if (run("c3q", 32768, 2, true, true)) return 1;
  if (run("c3q nores", 32768, 2, false, true)) return 1;
  if (run("c3q fp32 qkv", 32768, 2, true, false)) return 1;
  return 0;
}
