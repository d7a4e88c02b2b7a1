// Fused single-head attention for NonLocalBlock (/root/reference/model.py:51-53) in SPLIT PRECISION on the fp16 matrix cores:
// the same flash-style algorithm as attention.h (S^T = phi . theta^T per 32-key tile, online softmax in registers with the
// query on the lane, O^T += g^T . P^T with exp2(S^T) used directly as the B operand), but every fp32 operand — theta, phi, g
// and the probabilities P — is split into hi + lo fp16 halves and each contraction issues hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation (igemm_h16.h explains the numerics: ~2^-22 per product).
//
// Operand maps of v_mfma_f32_32x32x16_f16 (cdna_hip_programming.md §3): lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j]
// and B[k = 8h + j][col r], j = 0..7; C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4h.
//  * S^T: A = phi tile in LDS, [key][128 hi | 128 lo halves], one ds_read_b128 per plane and 16-channel K step; B = theta of the
//    lane's query, split once into registers.
//  * O^T: B = P^T straight from the S^T accumulator: registers 8t..8t+7 of lane half h are keys 16t + 8(j >> 2) + 4h + (j & 3).
//    A = g^T must present the same key order along k, so the g tile is stored TRANSPOSED and key-PERMUTED in LDS:
//    row = channel d, position p = 16t + 8h + 4a + b holds key 16t + 8a + 4h + b — again one ds_read_b128 per plane and K step.
//    The transpose is done by the staging threads (thread = one channel x 16 keys: coalesced dword loads, 16-byte LDS stores).
#pragma once
#include <hip/hip_runtime.h>
#include "attention.h"
#include "igemm_h16.h"

namespace bsr {

constexpr int kAx3LdK = 132;                                   // words per phi row: 64 (hi) + 64 (lo) + 4 pad
constexpr int kAx3LdV = 36;                                    // words per g^T row: 16 (hi) + 16 (lo) + 4 pad
constexpr int kAx3StageWords = kAttKT * kAx3LdK + kAttD * kAx3LdV;
constexpr int kAx3SmemBytes = 2 * kAx3StageWords * 4;

__global__ __launch_bounds__(256, 1) void nonlocal_attention_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out, int tokens) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, r = lane & 31;
  const int qblocks = tokens / 128;
  int img, qb;
  {   // query blocks of one image share an XCD's L2 copy of K/V (attention.h)
    const int nblk = gridDim.x, b = blockIdx.x;
    const int per_round = 8 * qblocks;
    if (nblk % per_round == 0) {
      const int round = b / per_round, within = b % per_round;
      img = round * 8 + (within % 8);
      qb = within / 8;
    } else {
      img = b / qblocks;
      qb = b % qblocks;
    }
  }
  const float* base = qkv + (size_t)img * tokens * (3 * kAttD);
  const int q = qb * 128 + wave * 32 + r;

  // theta of this lane's query, pre-scaled by log2(e) (softmax in base 2), split: K step s covers channels 16s + 8h .. +7
  f16x8 qh[kAttD / 16], ql[kAttD / 16];
#pragma unroll
  for (int s = 0; s < kAttD / 16; ++s) {
    const float* src = base + (size_t)q * (3 * kAttD) + 16 * s + 8 * h;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src) * 1.4426950408889634f;
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + 4) * 1.4426950408889634f;
    split8(a, b, qh[s], ql[s]);
  }

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging: phi as 8-channel pieces (512 per tile, 2 per thread); g as (channel, 16-key half) columns (256 per tile, 1 per thread)
  f32x4 kreg[4];
  float vreg[16];
  const int vd = tid & 127, vt = tid >> 7;
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int key = idx >> 4, c8 = idx & 15;
      const float* row = base + (size_t)(kt * kAttKT + key) * (3 * kAttD) + kAttD + c8 * 8;
      kreg[2 * i] = *reinterpret_cast<const f32x4*>(row);
      kreg[2 * i + 1] = *reinterpret_cast<const f32x4*>(row + 4);
    }
    const float* col = base + (size_t)(kt * kAttKT + vt * 16) * (3 * kAttD) + 2 * kAttD + vd;
#pragma unroll
    for (int j = 0; j < 16; ++j) vreg[j] = col[(size_t)j * (3 * kAttD)];
  };
  auto publish = [&](int buf) {
    float* sk = smem + buf * kAx3StageWords;
    float* sv = sk + kAttKT * kAx3LdK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int key = idx >> 4, c8 = idx & 15;
      f16x8 hi, lo;
      split8(kreg[2 * i], kreg[2 * i + 1], hi, lo);
      *reinterpret_cast<f16x8*>(sk + key * kAx3LdK + c8 * 4) = hi;
      *reinterpret_cast<f16x8*>(sk + key * kAx3LdK + 64 + c8 * 4) = lo;
    }
    // g^T row vd, K step vt: position 8h' + 4a + b  <-  key offset 8a + 4h' + b
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      f16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = vreg[8 * (j >> 2) + 4 * hh + (j & 3)];
        const _Float16 xh = (_Float16)x;
        hi[j] = xh;
        lo[j] = (_Float16)(x - (float)xh);
      }
      *reinterpret_cast<f16x8*>(sv + vd * kAx3LdV + 8 * vt + 4 * hh) = hi;
      *reinterpret_cast<f16x8*>(sv + vd * kAx3LdV + 16 + 8 * vt + 4 * hh) = lo;
    }
  };

  const int nkt = tokens / kAttKT;
  fetch(0);
  publish(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) fetch(kt + 1);
    const float* sk = smem + buf * kAx3StageWords;
    const float* sv = sk + kAttKT * kAx3LdK;

    f32x16 s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < kAttD / 16; ++ks) {
      const f16x8 kh = *reinterpret_cast<const f16x8*>(sk + r * kAx3LdK + ks * 8 + 4 * h);
      const f16x8 kl = *reinterpret_cast<const f16x8*>(sk + r * kAx3LdK + 64 + ks * 8 + 4 * h);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
    }

    // online softmax, exactly as attention.h: the decision covers this tile's P before any of it is exponentiated
    float mx = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__any(mx > m_run + kRescaleThreshold)) {
      const float m_new = fmaxf(m_run, mx);
      const float scale = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= scale;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= scale;
    }
    float psum = 0.f;
    f16x8 ph[2], pl[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float pv = __builtin_amdgcn_exp2f(s[i] - m_run);       // <= 2^8: inside the fp16 range
      psum += pv;
      const _Float16 x = (_Float16)pv;
      ph[i >> 3][i & 7] = x;
      pl[i >> 3][i & 7] = (_Float16)(pv - (float)x);
    }
    l_run += psum;

#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f16x8 vh = *reinterpret_cast<const f16x8*>(sv + (32 * dt + r) * kAx3LdV + 8 * t + 4 * h);
        const f16x8 vl = *reinterpret_cast<const f16x8*>(sv + (32 * dt + r) * kAx3LdV + 16 + 8 * t + 4 * h);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[t], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[t], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[t], o[dt], 0, 0, 0);
      }

    if (kt + 1 < nkt) {
      publish(buf ^ 1);
      __syncthreads();
    }
  }

  // y[q][d], d = 32 dt + (i & 3) + 8 (i >> 2) + 4h: four consecutive channels per register quad
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  float* orow = out + ((size_t)img * tokens + q) * kAttD;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 v = {o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv, o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv};
      *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g4 + 4 * h) = v;
    }
}

inline hipError_t launch_nonlocal_attention_x3(const float* qkv, float* out, int batch, int tokens, hipStream_t stream) {
  static PerDeviceOnce once;
  const int dev = PerDeviceOnce::current();
  if (dev < 0 || !once.done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nonlocal_attention_x3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kAx3SmemBytes);
    if (e != hipSuccess) return e;
    if (dev >= 0) once.done[dev] = true;
  }
  hipLaunchKernelGGL(nonlocal_attention_x3_kernel, dim3(batch * (tokens / 128)), dim3(256), kAx3SmemBytes, stream, qkv, out, tokens);
  return hipGetLastError();
}

}  // namespace bsr
