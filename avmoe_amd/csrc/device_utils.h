// Small device helpers shared by the bottleneck-space kernels (wave64 reductions, T <-> f32).
#pragma once
#include <hip/hip_runtime.h>

namespace avmoe {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

__device__ __forceinline__ float bf2f(unsigned short h) {
  return __builtin_bit_cast(float, ((unsigned int)h) << 16);
}
__device__ __forceinline__ unsigned short f2bf(float x) {
  return __builtin_bit_cast(unsigned short, (__bf16)x);
}

// element access for the activation / operand type T (float or __bf16)
template <typename T> __device__ __forceinline__ float ldT(const T* p, long i);
template <> __device__ __forceinline__ float ldT<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ldT<__bf16>(const __bf16* p, long i) {
  return bf2f(((const unsigned short*)p)[i]);
}
template <typename T> __device__ __forceinline__ void stT(T* p, long i, float v);
template <> __device__ __forceinline__ void stT<float>(float* p, long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stT<__bf16>(__bf16* p, long i, float v) {
  ((unsigned short*)p)[i] = f2bf(v);
}

// four consecutive elements (index i a multiple of 4, base 16-byte aligned): one 16-byte / 8-byte access
template <typename T> __device__ __forceinline__ float4 ld4T(const T* p, long i);
template <> __device__ __forceinline__ float4 ld4T<float>(const float* p, long i) { return *(const float4*)(p + i); }
template <> __device__ __forceinline__ float4 ld4T<__bf16>(const __bf16* p, long i) {
  const u32x2_t u = *(const u32x2_t*)((const unsigned short*)p + i);
  return make_float4(__builtin_bit_cast(float, u[0] << 16), __builtin_bit_cast(float, u[0] & 0xffff0000u),
                     __builtin_bit_cast(float, u[1] << 16), __builtin_bit_cast(float, u[1] & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4T(T* p, long i, const float4& v);
template <> __device__ __forceinline__ void st4T<float>(float* p, long i, const float4& v) { *(float4*)(p + i) = v; }
template <> __device__ __forceinline__ void st4T<__bf16>(__bf16* p, long i, const float4& v) {
  u32x2_t u;
  u[0] = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16); u[1] = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
  *(u32x2_t*)((unsigned short*)p + i) = u;
}

// wave64 all-reduce (every lane gets the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// the same for a double per lane (row statistics whose consumer takes a DIFFERENCE of sums: variance, softmax denominators)
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block (256 threads) sum through LDS scratch of >= 4 floats; result valid in every thread
__device__ __forceinline__ float block_sum256(float v, float* scratch4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch4[threadIdx.x >> 6] = v;
  __syncthreads();
  return scratch4[0] + scratch4[1] + scratch4[2] + scratch4[3];
}

}  // namespace avmoe
