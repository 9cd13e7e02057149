// dev probe (round 4): do the fp32-MFMA mat-vecs of the bottleneck-space kernels (mmT: A operand from LDS, four dependent
// v_mfma_f32_16x16x4_f32 per 16-column tile, results consumed by VALU right away) return the same bits when ANOTHER kernel hammers the
// matrix pipe of the same SIMDs from a second stream?   build:  hipcc --offload-arch=gfx950 -O3 scripts/mfma_probe.hip -o avmoe_amd/lib/variants/mfma_probe
//   ./mfma_probe [reps] [aggressor mode: 0 none, 1 bf16 32x32x16 MFMA loop, 2 + ds_read_b32, 3 + ds_read_b128, 4 / 5 LDS reads only, 6 / 7 16x16 MFMAs + b128, 8 / 9 sixteen chains of bf16 16x16x32 (+ b128)] [victim dynamic LDS bytes] [aggressor blocks, default 1024] [victim launches per rep, default 8; 0 = aggressor alone]
//   -DMIT=n: mitigations under test (see mmT); 8 = the A operand as two ds_read_b64; scripts/run_mfma_probe.sh runs the matrix kept in profiles/r05_mfma_probe.txt
// The victim computes, per tile step, W = P x M (16 tokens x 32 -> 32) with mmT and the same numbers with plain FMAs through shuffles,
// and counts the lanes whose results differ by more than 1e-4 relative.  Expected: 0, with or without the aggressor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ int opaque0() { int v = 0; asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float& at(float4& v, int x) { return ((float*)&v)[x]; }
__device__ __forceinline__ float at(const float4& v, int x) { return ((const float*)&v)[x]; }

// MIT 20 / 21 (round 6): the mat-vec in the form the TUNED kernels run (csrc/tile_fast.hip::mmT_split, csrc/tile_stream.hip::mm_presplit): both
// operands as two bf16 planes, hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16 -- does THAT form move beside a dense-MFMA aggressor?
// 21: + the per-block column sums of those kernels (DPP row sums, rsum16) against a shuffle butterfly over the same values.
__device__ __forceinline__ void split8(const float4& v0, const float4& v1, bf16x8& hi, bf16x8& lo) {
  const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) { const __bf16 h = (__bf16)f[i]; hi[i] = h; lo[i] = (__bf16)(f[i] - (float)h); }
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float rsum16(float v) { v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v); return v; }
template <int NJ>
__device__ __forceinline__ f32x4 mmT(const float* Mt, int ld, int col0, const float4* p, int r, int q) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* mp = Mt + (col0 + r) * ld + 4 * q;
#if defined(MIT) && MIT == 22      // the split form with NO dependent matrix instructions: every product into a zero accumulator, summed by the VALU
#pragma unroll
  for (int j = 0; j < NJ; j += 2) {
    bf16x8 ah, al, ph, pl;
    split8(*(const float4*)(mp + 16 * j), *(const float4*)(mp + 16 * (j + 1)), ah, al);
    split8(p[j], p[j + 1], ph, pl);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ph, z, 0, 0, 0);
    const f32x4 a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pl, z, 0, 0, 0);
    const f32x4 a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ph, z, 0, 0, 0);
    acc += (a0 + a1) + a2;
  }
  return acc;
#endif
#if defined(MIT) && MIT == 23      // the exact-fp32 form with no dependent matrix instructions
  {
    f32x4 part[NJ * 4];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float4 av = *(const float4*)(mp + 16 * j);
#pragma unroll
      for (int x = 0; x < 4; ++x) part[4 * j + x] = __builtin_amdgcn_mfma_f32_16x16x4f32(at(av, x), at(p[j], x), z, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NJ * 4; ++i) acc += part[i];
    return acc;
  }
#endif
#if defined(MIT) && (MIT == 20 || MIT == 21)
#pragma unroll
  for (int j = 0; j < NJ; j += 2) {
    bf16x8 ah, al, ph, pl;
    split8(*(const float4*)(mp + 16 * j), *(const float4*)(mp + 16 * (j + 1)), ah, al);
    split8(p[j], p[j + 1], ph, pl);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ph, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ph, acc, 0, 0, 0);
  }
  return acc;
#endif
#ifndef MIT
#define MIT 0      // mitigation under test: 1 = drain the LDS counter + s_nop before the MFMAs, 2 = the LDS data through a VALU move, 3 = both
#endif
  float4 a[NJ];
#if MIT == 8       // the same 16 bytes per lane as two 8-byte LDS reads (ds_read_b64 x 2 instead of ds_read_b128)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float2 lo, hi;
    const unsigned ad = (unsigned)(size_t)(mp + 16 * j);
    asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(ad) : "memory");
    a[j] = make_float4(lo.x, lo.y, hi.x, hi.y);
  }
#else
#pragma unroll
  for (int j = 0; j < NJ; ++j) a[j] = *(const float4*)(mp + 16 * j);
#endif
#if MIT == 1 || MIT == 3
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
  asm volatile("s_nop 7" ::: "memory");
#endif
#if MIT == 2 || MIT == 3
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int x = 0; x < 4; ++x) asm volatile("v_mov_b32 %0, %0" : "+v"(at(a[j], x)));
#endif
#if MIT == 4       // two independent accumulator chains
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at(a[j], 0), at(p[j], 0), acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(at(a[j], 1), at(p[j], 1), acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at(a[j], 2), at(p[j], 2), acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(at(a[j], 3), at(p[j], 3), acc2, 0, 0, 0);
  }
  return acc + acc2;
#else
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at(a[j], x), at(p[j], x), acc, 0, 0, 0);
#if MIT == 5
      asm volatile("s_nop 7\n s_nop 7" ::: "memory");
#endif
#if MIT == 9
      asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
#endif
#if MIT == 12
      asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
#endif
#if MIT == 13
      asm volatile("s_sleep 2" ::: "memory");
#endif
    }
  }
  return acc;
#endif
}
template <int NJ>
__device__ __forceinline__ f32x4 mmV(const float* Mt, int ld, int col0, const float4* p, int r, int q) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
      for (int xx = 0; xx < 4; ++xx) {
        const float pv = __shfl(at(p[j], xx), r + 16 * qq);
        const int k = 16 * j + 4 * qq + xx;
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[x] += pv * Mt[(col0 + 4 * q + x) * ld + k];
      }
  return acc;
}

constexpr int LD = 36;
__global__ void __launch_bounds__(256) victim(const float* __restrict__ M, const float* __restrict__ P, int tiles, unsigned* __restrict__ bad, float* __restrict__ sink,
                                               float* __restrict__ outU, float* __restrict__ outV) {
#if MIT == 6
  __builtin_amdgcn_s_setprio(3);
#endif
  __shared__ float s_M[2][32 * LD];
  extern __shared__ float hog[];      // dynamic LDS request (argv[3] / MIT 7): sets how many victim blocks fit a compute unit and what is left for the aggressor
  if (threadIdx.x == 0 && tiles < 0) hog[0] = 1.f;
  for (int i = threadIdx.x; i < 2 * 32 * 32; i += 256) { const int m = i >> 10, n = (i >> 5) & 31, k = i & 31; s_M[m][n * LD + k] = M[i]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
  unsigned nbad = 0;
  float keep = 0.f;
  for (int t = blockIdx.x * 4 + wave; t < tiles; t += gridDim.x * 4) {
    const int oz = opaque0();
    float4 p[2];
    p[0] = *(const float4*)(P + ((long)t * 16 + r) * 32 + 4 * q);
    p[1] = *(const float4*)(P + ((long)t * 16 + r) * 32 + 16 + 4 * q);
    float u3 = 0.f, v3 = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 w = mmT<2>(s_M[m] + oz, LD, 16 * ct, p, r, q);
#pragma unroll
        for (int x = 0; x < 4; ++x) u3 += w[x] * at(p[ct], x);
      }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 w = mmV<2>(s_M[m] + oz, LD, 16 * ct, p, r, q);
#pragma unroll
        for (int x = 0; x < 4; ++x) v3 += w[x] * at(p[ct], x);
      }
    if (fabsf(u3 - v3) > 1e-4f * fmaxf(fabsf(v3), 1.f)) ++nbad;
#if defined(MIT) && MIT == 21
    {
      const float cs = rsum16(u3);                     // column sum over the tile's 16 tokens, as flush_cols_we forms it
      float bs = u3;
      for (int o = 1; o < 16; o <<= 1) bs += __shfl_xor(bs, o, 64);
      if (fabsf(cs - bs) > 1e-5f * fmaxf(fabsf(bs), 1.f)) ++nbad;
    }
#endif
    if (outU) { outU[(long)t * 64 + lane] = u3; outV[(long)t * 64 + lane] = v3; }
    keep += u3;
  }
  if (nbad) atomicAdd(bad, nbad);
  sink[blockIdx.x * 256 + threadIdx.x] = keep;
}

// aggressor: long dependent-free chains of the widest bf16 MFMA on every wave (optionally with LDS traffic in between)
__global__ void __launch_bounds__(256) aggressor_lds(int iters, int lds, float* __restrict__ sink) {      // LDS traffic only
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = (float)i * 1e-6f;
  __syncthreads();
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (lds & 1) acc += s[(threadIdx.x * 17 + i) & 4095];
    if (lds & 2) { const float4 t4 = *(const float4*)(s + ((threadIdx.x * 4 + 16 * i) & 4092)); acc += t4.x + t4.w; }
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) aggressor16(int iters, int lds, int f32, float* __restrict__ sink) {      // 16x16 MFMAs (bf16 x32 or f32 x4) + LDS
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = (float)i * 1e-6f;
  __syncthreads();
  const bf16x8 a = {(__bf16)1.f, (__bf16)0.5f, (__bf16)0.25f, (__bf16)1.f, (__bf16)0.5f, (__bf16)0.25f, (__bf16)1.f, (__bf16)0.5f};
  f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (f32) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(0.5f, 0.25f, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(0.5f, 0.25f, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(0.5f, 0.25f, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(0.5f, 0.25f, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c3, 0, 0, 0);
    }
    if (lds & 2) { const float4 t4 = *(const float4*)(s + ((threadIdx.x * 4 + 16 * i) & 4092)); acc += t4.x + t4.w; }
  }
  sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + acc;
}
__global__ void __launch_bounds__(256) aggressor16w(int iters, int lds, float* __restrict__ sink) {      // 16 chains of 16x16x32 bf16 (64 accumulator registers, like a GEMM tile)
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = (float)i * 1e-6f;
  __syncthreads();
  const bf16x8 a = {(__bf16)1.f, (__bf16)0.5f, (__bf16)0.25f, (__bf16)1.f, (__bf16)0.5f, (__bf16)0.25f, (__bf16)1.f, (__bf16)0.5f};
  f32x4 c[16] = {};
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c[k], 0, 0, 0);
    if (lds & 2) { const float4 t4 = *(const float4*)(s + ((threadIdx.x * 4 + 16 * i) & 4092)); acc += t4.x + t4.w; }
  }
  float r = acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) r += c[k][k & 3];
  sink[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) aggressor(int iters, int lds, float* __restrict__ sink) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = (float)i * 1e-6f;
  __syncthreads();
  const bf16x8 a = {(__bf16)1.f, (__bf16)0.5f, (__bf16)0.25f, (__bf16)1.f, (__bf16)0.5f, (__bf16)0.25f, (__bf16)1.f, (__bf16)0.5f};
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c3, 0, 0, 0);
    if (lds & 1) acc += s[(threadIdx.x * 17 + i) & 4095];
    if (lds & 2) { const float4 t4 = *(const float4*)(s + ((threadIdx.x * 4 + 16 * i) & 4092)); acc += t4.x + t4.w; }
  }
  sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + acc;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20, mode = argc > 2 ? atoi(argv[2]) : 1;
  const int agrid = argc > 4 ? atoi(argv[4]) : 1024, nvict = argc > 5 ? atoi(argv[5]) : 8;      // aggressor blocks ; victim launches per rep (0: aggressor alone, for its duration)
  const int tiles = 1 << 17;
  std::vector<float> hM(2 * 32 * 32), hP((size_t)tiles * 16 * 32);
  unsigned seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 65536.f - 0.5f; };
  for (auto& v : hM) v = rnd();
  for (auto& v : hP) v = rnd();
  float *dM, *dP, *sink1, *sink2; unsigned* dbad;
  hipMalloc(&dM, hM.size() * 4); hipMalloc(&dP, hP.size() * 4); hipMalloc(&sink1, 4096 * 256 * 4); hipMalloc(&sink2, 4096 * 256 * 4); hipMalloc(&dbad, 4);
  hipMemcpy(dM, hM.data(), hM.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dP, hP.data(), hP.size() * 4, hipMemcpyHostToDevice);
  float *dU, *dV; hipMalloc(&dU, (size_t)tiles * 64 * 4); hipMalloc(&dV, (size_t)tiles * 64 * 4);
  std::vector<float> hU((size_t)tiles * 64), hV((size_t)tiles * 64);
  unsigned long badU = 0, badV = 0;
#if MIT == 7
  const int vict_lds = argc > 3 ? atoi(argv[3]) : 140 * 1024;      // with the 9 KB of static LDS: 149 KB -- no 16 KB block of the aggressor fits beside it on the CU
#else
  const int vict_lds = argc > 3 ? atoi(argv[3]) : 0;               // argv[3]: dynamic LDS bytes per victim block (own blocks per CU = 160 KB / (9 KB + this))
#endif
  if (vict_lds > 0) hipFuncSetAttribute((const void*)victim, hipFuncAttributeMaxDynamicSharedMemorySize, vict_lds);
  {
    int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)victim, 256, vict_lds);
    printf("victim: %d B dynamic + 9216 B static LDS per block -> %d resident blocks per CU (aggressor blocks take 16384 B each)\n", vict_lds, nb);
  }
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  unsigned total = 0;
  for (int rep = 0; rep < reps; ++rep) {
    hipMemsetAsync(dbad, 0, 4, s1);
    hipStreamSynchronize(s1);
    const auto t0 = std::chrono::steady_clock::now();
    // mode: 1 MFMA only ; 2 MFMA + ds_read_b32 ; 3 MFMA + ds_read_b128 ; 4 ds_read_b32 only ; 5 ds_read_b128 only
    if (mode >= 1 && mode <= 3) hipLaunchKernelGGL(aggressor, dim3(agrid), dim3(256), 0, s2, 20000, mode == 2 ? 1 : (mode == 3 ? 2 : 0), sink2);
    if (mode == 6 || mode == 7) hipLaunchKernelGGL(aggressor16, dim3(agrid), dim3(256), 0, s2, 60000, 2, mode == 7, sink2);      // 6: bf16 16x16x32 + b128 ; 7: f32 16x16x4 + b128
    if (mode == 8 || mode == 9) hipLaunchKernelGGL(aggressor16w, dim3(agrid), dim3(256), 0, s2, 10000, mode == 9 ? 2 : 0, sink2);      // 8: 16 chains of bf16 16x16x32 ; 9: + b128
    if (mode == 4 || mode == 5) hipLaunchKernelGGL(aggressor_lds, dim3(agrid), dim3(256), 0, s2, 80000, mode == 4 ? 1 : 2, sink2);
    for (int k = 0; k < nvict; ++k) hipLaunchKernelGGL(victim, dim3(1024), dim3(256), vict_lds, s1, dM, dP, tiles, dbad, sink1, k == 3 ? dU : nullptr, k == 3 ? dV : nullptr);
    hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rep == reps - 1) printf("last rep: %.2f ms for the aggressor (%d blocks) + %d victim launches on two streams\n", ms, mode ? agrid : 0, nvict);
    if (rep < 3 && nvict > 3) {      // which of the two methods is off?  exact reference (double) on the host for the lanes the GPU flagged (u3 != v3)
      hipMemcpy(hU.data(), dU, hU.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hV.data(), dV, hV.size() * 4, hipMemcpyDeviceToHost);
      int shown = 0;
      for (long i = 0; i < (long)tiles * 64; ++i) {
        const float u = hU[i], v = hV[i];
        if (fabsf(u - v) <= 1e-4f * fmaxf(fabsf(v), 1.f)) continue;
        const long t = i / 64; const int lane = (int)(i % 64), r = lane & 15, q = lane >> 4;
        double ref = 0.0;
        for (int m = 0; m < 2; ++m)
          for (int ct = 0; ct < 2; ++ct)
            for (int x = 0; x < 4; ++x) {
              double w = 0.0;
              for (int k = 0; k < 32; ++k) w += (double)hP[((size_t)t * 16 + r) * 32 + k] * hM[(m * 32 + 16 * ct + 4 * q + x) * 32 + k];
              ref += w * hP[((size_t)t * 16 + r) * 32 + 16 * ct + 4 * q + x];
            }
        const bool ub = fabs(u - ref) > 1e-4 * fmax(fabs(ref), 1.0), vb = fabs(v - ref) > 1e-4 * fmax(fabs(ref), 1.0);
        badU += ub; badV += vb;
        if (shown++ < 4) printf("   tile %ld lane %d: mfma %.6f  valu %.6f  exact %.6f\n", t, lane, u, v, ref);
      }
    }
    unsigned b = 0; hipMemcpy(&b, dbad, 4, hipMemcpyDeviceToHost);
    if (b) printf("rep %d: %u mismatching lanes\n", rep, b);
    total += b;
  }
  printf("MFMA_PROBE MIT %d mode %d lds %d: %u mismatches over %d reps; of the lanes checked exactly: MFMA path wrong %lu, VALU path wrong %lu (%s)\n", MIT, mode, vict_lds, total, reps,
         badU, badV, hipGetErrorString(hipGetLastError()));
  return 0;
}
