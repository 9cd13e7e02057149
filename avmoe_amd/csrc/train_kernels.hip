// Training-loop pieces either side of the adapter path: a fused Adam step over one flat parameter
// bucket (the reference's optimizer is torch.optim.Adam over the adapter parameters, AVE/main_trans_v3.py:322) and a
// device-side expert-activation histogram (the reference loops over idx.tolist() on the host, main_trans_v3.py:183-207).
#include "../../include/avmoe.h"
#include "common.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>

namespace avmoe {

// torch.optim.Adam semantics (no amsgrad, L2 weight decay added to the gradient), one thread per 4 elements
__global__ void __launch_bounds__(256) kk_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                               float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  if (i0 + 3 < n) {
    float4 pp = *(float4*)(p + i0), mm = *(float4*)(m + i0), vv = *(float4*)(v + i0);
    const float4 gg = *(const float4*)(g + i0);
    float* P = (float*)&pp; float* M = (float*)&mm; float* V = (float*)&vv; const float* G = (const float*)&gg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = G[k] * gscale + wd * P[k];
      M[k] = b1 * M[k] + (1.f - b1) * gr;
      V[k] = b2 * V[k] + (1.f - b2) * gr * gr;
      P[k] -= (lr / bc1) * M[k] / (sqrtf(V[k]) / bc2_sqrt + eps);
    }
    *(float4*)(p + i0) = pp; *(float4*)(m + i0) = mm; *(float4*)(v + i0) = vv;
  } else {
    for (long i = i0; i < n; ++i) {
      const float gr = g[i] * gscale + wd * p[i];
      m[i] = b1 * m[i] + (1.f - b1) * gr;
      v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
      p[i] -= (lr / bc1) * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
    }
  }
}

// counts[idx[s]] += 1 (integer atomics: exact and order-independent)
__global__ void kk_expert_hist(const int64_t* __restrict__ idx, long S, int E, long long* __restrict__ counts) {
  for (long s = (long)blockIdx.x * 256 + threadIdx.x; s < S; s += (long)gridDim.x * 256) {
    const int64_t e = idx[s];
    if (e >= 0 && e < E) atomicAdd((unsigned long long*)(counts + e), 1ull);
  }
}

// dst1 += src1 ; dst2 += src2 in one launch (fp32 sums, rounded once): the two token tensors of a site pair each collect their
// second gradient this way when the two sites ran on two streams.  16 bytes per lane; blocks [0, nb1) serve the first pair.
template <typename T>
__global__ void __launch_bounds__(256) kk_add2(T* __restrict__ d1, const T* __restrict__ s1, long n1, T* __restrict__ d2, const T* __restrict__ s2, long n2, int nb1) {
  constexpr int V = 16 / sizeof(T);
  const bool first = (int)blockIdx.x < nb1;
  T* d = first ? d1 : d2; const T* s = first ? s1 : s2;
  const long n = first ? n1 : n2, nblk = first ? nb1 : (long)gridDim.x - nb1, b = first ? blockIdx.x : blockIdx.x - nb1;
  for (long i = (b * 256 + threadIdx.x) * V; i < n; i += nblk * 256 * V) {
    if (i + V <= n) {
      uint4 a = *(const uint4*)(d + i); const uint4 c = *(const uint4*)(s + i);
      if constexpr (sizeof(T) == 4) {
        float* A = (float*)&a; const float* C = (const float*)&c;
#pragma unroll
        for (int k = 0; k < 4; ++k) A[k] += C[k];
      } else {
        unsigned* A = (unsigned*)&a; const unsigned* C = (const unsigned*)&c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float lo = __builtin_bit_cast(float, A[k] << 16) + __builtin_bit_cast(float, C[k] << 16);
          const float hi = __builtin_bit_cast(float, A[k] & 0xffff0000u) + __builtin_bit_cast(float, C[k] & 0xffff0000u);
          A[k] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
        }
      }
      *(uint4*)(d + i) = a;
    } else {
      for (long j = i; j < n; ++j) d[j] = (T)((float)d[j] + (float)s[j]);
    }
  }
}

}  // namespace avmoe

using namespace avmoe;

// idx[s][j] = expert with the j-th largest probability of frame s; equal probabilities keep the lower expert index first, so
// column 0 is exactly the forward's first-max argmax (net_trans_v3.py:479).  One thread per frame, E <= 16: selection by rank.
__global__ void __launch_bounds__(256) kk_router_topk(const float* __restrict__ probs, long S, int E, int k, long long* __restrict__ idx) {
  for (long s = (long)blockIdx.x * 256 + threadIdx.x; s < S; s += (long)gridDim.x * 256) {
    float p[AVMOE_MAX_EXPERTS];
    for (int e = 0; e < E; ++e) p[e] = probs[s * E + e];
    for (int e = 0; e < E; ++e) {
      int rank = 0;                                    // experts that come before e
      for (int f = 0; f < E; ++f) rank += (p[f] > p[e]) || (p[f] == p[e] && f < e);
      if (rank < k) idx[s * k + rank] = e;
    }
  }
}

extern "C" {

int avmoe_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) { set_last_error("avmoe_adam_step: bad argument"); return ERR_BAD_ARG; }
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u) != 0) {
    set_last_error("avmoe_adam_step: buffers must be 16-byte aligned"); return ERR_ALIGNMENT;
  }
  if (n == 0) return OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const long nblk = (n + 1023) / 1024;
  hipLaunchKernelGGL(kk_adam, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2,
                     eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
  AVMOE_CHECK_LAUNCH("adam_step");
  return OK;
}

int avmoe_add2(void* dst1, const void* src1, int64_t n1, void* dst2, const void* src2, int64_t n2, int32_t dtype, void* stream) {
  if ((n1 > 0 && (!dst1 || !src1)) || (n2 > 0 && (!dst2 || !src2)) || n1 < 0 || n2 < 0 || (dtype != AVMOE_F32 && dtype != AVMOE_BF16)) {
    set_last_error("avmoe_add2: bad argument"); return ERR_BAD_ARG;
  }
  if ((((uintptr_t)dst1 | (uintptr_t)src1 | (uintptr_t)dst2 | (uintptr_t)src2) & 15u) != 0) { set_last_error("avmoe_add2: buffers must be 16-byte aligned"); return ERR_ALIGNMENT; }
  if (n1 + n2 == 0) return OK;
  const long per = 256L * 16 * 4;                       // bytes one block moves per sweep
  const int esz = dtype == AVMOE_BF16 ? 2 : 4;
  auto blocks = [&](long n) { return n == 0 ? 0L : std::max<long>(1, std::min<long>((n * esz + per - 1) / per, 2048)); };
  const long nb1 = blocks(n1), nb2 = blocks(n2);
  if (dtype == AVMOE_BF16) hipLaunchKernelGGL(kk_add2<__bf16>, dim3((unsigned)(nb1 + nb2)), dim3(256), 0, (hipStream_t)stream, (__bf16*)dst1, (const __bf16*)src1, (long)n1, (__bf16*)dst2, (const __bf16*)src2, (long)n2, (int)nb1);
  else hipLaunchKernelGGL(kk_add2<float>, dim3((unsigned)(nb1 + nb2)), dim3(256), 0, (hipStream_t)stream, (float*)dst1, (const float*)src1, (long)n1, (float*)dst2, (const float*)src2, (long)n2, (int)nb1);
  AVMOE_CHECK_LAUNCH("add2");
  return OK;
}

int avmoe_router_topk(const float* probs, int64_t S, int32_t E, int32_t k, int64_t* idx, void* stream) {
  if (!probs || !idx || S < 0 || E < 1 || E > AVMOE_MAX_EXPERTS || k < 1 || k > E) { set_last_error("avmoe_router_topk: bad argument"); return ERR_BAD_ARG; }
  if (S == 0) return OK;
  hipLaunchKernelGGL(kk_router_topk, dim3((unsigned)std::min<long>((S + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, probs, (long)S,
                     (int)E, (int)k, (long long*)idx);
  AVMOE_CHECK_LAUNCH("router_topk");
  return OK;
}

int avmoe_expert_histogram(const int64_t* idx, int64_t S, int32_t E, int64_t* counts, void* stream) {
  if (!idx || !counts || S < 0 || E < 1) { set_last_error("avmoe_expert_histogram: bad argument"); return ERR_BAD_ARG; }
  if (S == 0) return OK;
  const unsigned nblk = (unsigned)std::min<long>((S + 255) / 256, 1024);
  hipLaunchKernelGGL(kk_expert_hist, dim3(nblk), dim3(256), 0, (hipStream_t)stream, idx, (long)S, (int)E, (long long*)counts);
  AVMOE_CHECK_LAUNCH("expert_histogram");
  return OK;
}

}  // extern "C"
