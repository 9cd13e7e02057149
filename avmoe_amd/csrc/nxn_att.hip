// AVVP unimodal N x N block (mgn.py:132-139), the softmax of the token-token scores as ONE kernel per chunk of frames:
//
//     att[s] = softmax_rows(X[s] X[s]^T)          X: (frames, N, C) bf16,  att: (frames, N, Np) bf16,  lse: (frames, N) f32
//
// A block owns 128 query rows of one frame and keeps their MFMA fragments in registers; the key tiles (128 tokens x C, 24 - 48 KB)
// stream through the LDS -- twice: sweep 1 accumulates the running (max, sum exp) of every row, sweep 2 recomputes the scores and
// stores exp(score - lse) -- so the scores never exist outside the accumulators and X (0.8 - 0.9 MB per frame) is served by the L2.
// Products are computed transposed (A operand = key fragment, B operand = query fragment; the key rows of a 16-column tile are a
// permutation of the tile's columns) so that lane (r, q) ends up with 32 CONSECUTIVE columns of row r: 64 contiguous bytes per lane
// and slab, stored straight from registers.  With the row log-sum-exp given (the backward of a chunked site) only sweep 2 runs.
// Built for C = 96 and C = 192 (bf16, N a multiple of 128): the stage-0 / stage-1 sites, where N is large; everything else takes
// the engine's softmax epilogues (gemm.hip).
//
// Round 5: every fragment read of the product loops is issued by hand two to three steps ahead of the matrix instructions that use it
// (inline assembly + counted lgkmcnt, compile-time offsets through static_for).  The C = 192 instances keep 414 - 461 registers, so a block
// is alone on its CU with one wave per SIMD, and the compiler's `ds_read -> s_waitcnt lgkmcnt(0) -> 2 - 4 products` on ONE register quad left
// the matrix pipe idle for an LDS latency per read: dx (query / key owner) 5.67 / 5.46 -> 4.59 / 4.33 ms, y 4.80 -> 3.72, xr 4.14 -> 3.05 ms
// per launch at the cfg-3 stage-0 visual site (64 x 10 frames of 2304 tokens), cfg-3 step 550 -> 528 - 536 ms.  The C = 96 instances (two
// blocks per CU, two waves per SIMD) measure the same as before: there the second wave had been covering the latency.
#include "kernels.h"
#include "common.h"
#include "prof.h"
#include <algorithm>
#include <type_traits>

namespace avmoe {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ unsigned pack2(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}


// LDS reads issued and waited for by hand (round 5, from dx_stream3.hip): where a block keeps 450 registers it is alone on its CU, one wave
// per SIMD, and the compiler's `read -> s_waitcnt lgkmcnt(0) -> products` leaves the matrix pipe idle for one LDS latency per read.  The
// offsets have to be immediates: the loops run over compile-time indices (static_for).
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
template <int OFF> __device__ __forceinline__ void lds_rd128(u32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void lds_rdtr(u32x2& d, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void lds_wait(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void lds_wait2(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N> __device__ __forceinline__ void lds_waittr(u32x2& a, u32x2& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int I, int END> struct static_for_t {
  template <typename F> static __device__ __forceinline__ void run(F&& f) { f(std::integral_constant<int, I>{}); static_for_t<I + 1, END>::run(f); }
};
template <int END> struct static_for_t<END, END> { template <typename F> static __device__ __forceinline__ void run(F&&) {} };
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_t<B, E>::run(f); }

#ifndef NXN_ATT_MINB6
#define NXN_ATT_MINB6 2      // blocks per CU the C = 192 instance of kk_nxn_att is compiled for (2: 256 registers, 36 spilled; 1: 512)
#endif
template <int KS>          // C = 32 KS
__global__ void __launch_bounds__(256, (KS <= 3 ? 2 : NXN_ATT_MINB6)) kk_nxn_att(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Kt, float* __restrict__ lse_g,
                                                      unsigned short* __restrict__ att, const unsigned short* __restrict__ att_in, int N, int Np, int have_lse) {
  // att_in != nullptr (have_lse set): the softmax BACKWARD instead -- keys = dxr, lse_g = the row dots, output = att_in * (X dxr^T - rowdot)
  constexpr int C = 32 * KS, RB = C * 2 + 16, CPR = C / 8, NLD = 128 * CPR / 256;          // LDS row pitch, 16-byte chunks per key row, loads per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const long f = blockIdx.y;
  const unsigned short* Xf = X + f * (long)N * C;
  const unsigned short* Kf = Kt + f * (long)N * C;          // the key rows (X itself for the softmax)
  const int i0 = blockIdx.x * 128 + 32 * wave;                     // this wave's 32 query rows
  bf16x8 qf[2][KS];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[tm][ks] = *(const bf16x8*)(Xf + (long)(i0 + 16 * tm + r) * C + ks * 32 + 8 * q);
  float mrow[2] = {-INFINITY, -INFINITY}, lrow[2] = {0.f, 0.f}, lse[2];
  if (have_lse) { lse[0] = lse_g[f * N + i0 + r]; lse[1] = lse_g[f * N + i0 + 16 + r]; }
  const int ntile = N / 128;
  u32x4 st[NLD];
  auto gload = [&](int jt) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i;
      st[i] = *(const u32x4*)(Kf + (long)(jt * 128 + c / CPR) * C + (c % CPR) * 8);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i;
      *(u32x4*)(smem + (c / CPR) * RB + (c % CPR) * 16) = st[i];
    }
  };
  for (int sweep = have_lse ? 1 : 0; sweep < (att ? 2 : 1); ++sweep) {       // att == nullptr: the row statistics only (the forward's xr comes from kk_nxn_bwd<.., 2>)
    gload(0);
    for (int jt = 0; jt < ntile; ++jt) {
      __syncthreads();                                             // the previous tile's fragments have been read
      lstore();
      __syncthreads();
      if (jt + 1 < ntile) gload(jt + 1);                           // in flight during this tile's products
      f32x4 acc[2][8];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[tm][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      {   // key row of fragment row r of tile t: column (r >> 2) * 32 + 4 t + (r & 3) of this key tile  ->  lane (r, q) owns columns 32 q + 4 t + e
          // 8 KS steps (ks, t), the fragment reads issued by hand three steps ahead of the two products that use each
        constexpr int NA = 8 * KS;
        const unsigned aX = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem + ((r >> 2) * 32 + (r & 3)) * RB + q * 16);
        u32x4 kfr[4];
        auto issue = [&](auto ic) {
          constexpr int st_ = decltype(ic)::value, ks = st_ / 8, t = st_ % 8;
          lds_rd128<4 * t * RB + ks * 64>(kfr[st_ % 4], aX);
        };
        issue(std::integral_constant<int, 0>{});
        issue(std::integral_constant<int, 1>{});
        issue(std::integral_constant<int, 2>{});
        static_for<0, NA>([&](auto ic) {
          constexpr int st_ = decltype(ic)::value, ks = st_ / 8, t = st_ % 8;
          if constexpr (st_ + 3 < NA) issue(std::integral_constant<int, st_ + 3>{});
          lds_wait<(NA - 1 - st_ < 3 ? NA - 1 - st_ : 3)>(kfr[st_ % 4]);
          const bf16x8 kf = __builtin_bit_cast(bf16x8, kfr[st_ % 4]);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) acc[tm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[tm][ks], acc[tm][t], 0, 0, 0);
        });
      }
      if (sweep == 0) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          float mx = mrow[tm];
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, acc[tm][t][e]);
          float s = lrow[tm] * __expf(mrow[tm] - mx);
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += __expf(acc[tm][t][e] - mx);
          mrow[tm] = mx; lrow[tm] = s;
        }
      } else {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const long off = (f * N + i0 + 16 * tm + r) * (long)Np + jt * 128 + 32 * q;
          unsigned short* dst = att + off;
#pragma unroll
          for (int h = 0; h < 4; ++h) {                              // 8 columns = tiles 2 h and 2 h + 1
            const f32x4 a = acc[tm][2 * h], b = acc[tm][2 * h + 1];
            const float l = lse[tm];
            if (att_in) {
              const u32x4 w = *(const u32x4*)(att_in + off + 8 * h);
              auto lo = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
              auto hi = [](unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
              *(u32x4*)(dst + 8 * h) = u32x4{pack2(lo(w[0]) * (a[0] - l), hi(w[0]) * (a[1] - l)), pack2(lo(w[1]) * (a[2] - l), hi(w[1]) * (a[3] - l)),
                                             pack2(lo(w[2]) * (b[0] - l), hi(w[2]) * (b[1] - l)), pack2(lo(w[3]) * (b[2] - l), hi(w[3]) * (b[3] - l))};
            } else {
              *(u32x4*)(dst + 8 * h) = u32x4{pack2(__expf(a[0] - l), __expf(a[1] - l)), pack2(__expf(a[2] - l), __expf(a[3] - l)),
                                             pack2(__expf(b[0] - l), __expf(b[1] - l)), pack2(__expf(b[2] - l), __expf(b[3] - l))};
            }
          }
        }
      }
    }
    if (sweep == 0) {                                                // rows are spread over the four q lanes of each r
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        float m = mrow[tm], l = lrow[tm];
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
          const float mn = fmaxf(m, m2);
          l = l * __expf(m - mn) + l2 * __expf(m2 - mn); m = mn;
        }
        lse[tm] = m + __logf(l);
        if (q == 0) lse_g[f * N + i0 + 16 * tm + r] = lse[tm];
      }
    }
  }
}


// ---- the backward WITHOUT the softmax in memory (round 4) ------------------------------------------------------------------------
// With the row log-sum-exp kept, att[i][j] = exp(X_i . X_j - lse_i) costs one product per tile to re-form -- in the accumulators, where
// the transposed product leaves lane (r, q) with 32 CONSECUTIVE key columns of query row r.  Eight of them, columns 32 q + 8 s .. + 7,
// are exactly what lane (r, kb = q) has to supply as the B operand of a 16 x 16 x 32 step whose contraction runs over the keys
// { 32 q' + 8 s + p } (q' = 0 .. 3 the four k-blocks, p = 0 .. 7): the probabilities go from the accumulators into the next
// product WITHOUT moving between lanes, and the A operand -- the dxr (or X) tile transposed, channels as rows -- is read from the LDS
// tile with the transposing read at rows 32 q + 8 s (+ 4).  So:
//   kk_nxn_y   y = att dxr  (128 query rows x C in the accumulators over all key tiles), then rowdot_i = X_i . y_i and dX_i += y_i
//              -- replaces [att re-formed and stored] + [GEMM att dxr -> fp32 y] + [row-dot kernel]: att is neither written nor read
//   kk_nxn_ds  dS = att * (X dxr^T - rowdot) with att re-formed the same way (two products per tile) -- replaces the read of att
// (the two products against dS stay engine GEMMs on the stored dS).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_t;

template <int KS, int MODE>      // MODE 0: y = att dxr (+ rowdot, dX += y) ; 2: the FORWARD's xr = att^T X (the block owns 128 rows of xr; lse per streamed row)
__global__ void __launch_bounds__(256, (KS <= 3 ? 2 : 1)) kk_nxn_bwd(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Dx,
                                                                      const float* __restrict__ lse_g, float* __restrict__ rowdot, unsigned short* __restrict__ out,
                                                                      int N, int Np) {
  // MODE 0: out = dX (S, N, C), += y ; rowdot written.   MODE 2: out = xr (S, N, C), Dx unused.
  constexpr bool XR = MODE == 2;
  constexpr int C = 32 * KS, RB = C * 2 + 16, CPR = C / 8, NLD = 128 * CPR / 256, CT = C / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sD = XR ? smem : smem + 128 * RB;                   // (xr: the second product runs against the streamed X tile itself)
  float* s_lse = (float*)(smem + (XR ? 1 : 2) * 128 * RB);  // xr: the streamed rows' log-sum-exp
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const long f = blockIdx.y;
  const unsigned short* Xf = X + f * (long)N * C;
  const unsigned short* Df = XR ? Xf : Dx + f * (long)N * C;
  const int i0 = blockIdx.x * 128 + 32 * wave;
  bf16x8 qf[2][KS];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[tm][ks] = *(const bf16x8*)(Xf + (long)(i0 + 16 * tm + r) * C + ks * 32 + 8 * q);
  float lse[2] = {0.f, 0.f};
  if constexpr (!XR) { lse[0] = lse_g[f * N + i0 + r]; lse[1] = lse_g[f * N + i0 + 16 + r]; }
  f32x4 accY[2][CT];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) accY[tm][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntile = N / 128;
  u32x4 st[NLD], sd[XR ? 1 : NLD];
  float lnext = 0.f;
  auto gload = [&](int jt) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i;
      const long off = (long)(jt * 128 + c / CPR) * C + (c % CPR) * 8;
      st[i] = *(const u32x4*)(Xf + off);
      if constexpr (!XR) sd[i] = *(const u32x4*)(Df + off);
    }
    if constexpr (XR) { if (tid < 128) lnext = lse_g[f * N + jt * 128 + tid]; }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i;
      *(u32x4*)(sX + (c / CPR) * RB + (c % CPR) * 16) = st[i];
      if constexpr (!XR) *(u32x4*)(sD + (c / CPR) * RB + (c % CPR) * 16) = sd[i];
    }
    if constexpr (XR) { if (tid < 128) s_lse[tid] = lnext; }
  };
  gload(0);
  for (int jt = 0; jt < ntile; ++jt) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (jt + 1 < ntile) gload(jt + 1);
    f32x4 acc[2][8];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[tm][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {   // 8 KS steps (ks, t): the streamed rows' fragments, three steps ahead of the two products that use each (hand-issued LDS reads)
      constexpr int NA = 8 * KS;
      const unsigned aX = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(sX + ((r >> 2) * 32 + (r & 3)) * RB + q * 16);
      u32x4 kfr[4];
      auto issue = [&](auto ic) {
        constexpr int st_ = decltype(ic)::value, ks = st_ / 8, t = st_ % 8;
        lds_rd128<4 * t * RB + ks * 64>(kfr[st_ % 4], aX);
      };
      issue(std::integral_constant<int, 0>{});
      issue(std::integral_constant<int, 1>{});
      issue(std::integral_constant<int, 2>{});
      static_for<0, NA>([&](auto ic) {
        constexpr int st_ = decltype(ic)::value, ks = st_ / 8, t = st_ % 8;
        if constexpr (st_ + 3 < NA) issue(std::integral_constant<int, st_ + 3>{});
        lds_wait<(NA - 1 - st_ < 3 ? NA - 1 - st_ : 3)>(kfr[st_ % 4]);
        const bf16x8 kf = __builtin_bit_cast(bf16x8, kfr[st_ % 4]);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[tm][ks], acc[tm][t], 0, 0, 0);
      });
    }
    if constexpr (XR) {                    // att[i][j] for the block's own row j = r: the log-sum-exp belongs to the streamed row i = 32 q + 4 t + e
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float4 l4 = *(const float4*)(s_lse + 32 * q + 4 * t);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[tm][t][e] = __expf(acc[tm][t][e] - lv[e]);
      }
    } else {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[tm][t][e] = __expf(acc[tm][t][e] - lse[tm]);          // att, this lane's 32 key columns of row r
    }
    {
      // rounded to bf16 as the stored softmax was (same numbers as the path it replaces), as B operands: step s = columns 32 q + 8 s .. + 7
      bf16x8 pb[2][4];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
          const f32x4 a = acc[tm][2 * sidx], b = acc[tm][2 * sidx + 1];
          pb[tm][sidx] = bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
        }
      {   // 4 CT steps (ct, s): dxr^T fragments (channel 16 ct + r, keys 32 q + 8 s + 0 .. 7) by transposing reads, three steps ahead
        constexpr int NC = 4 * CT;
        const unsigned aT = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(sD + (32 * q + (r >> 2)) * RB + (4 * (r & 3)) * 2);
        u32x2 t1[4], t2[4];
        auto issue = [&](auto ic) {
          constexpr int st_ = decltype(ic)::value, ct = st_ / 4, sidx = st_ % 4, off = 8 * sidx * RB + 32 * ct;
          lds_rdtr<off>(t1[st_ % 4], aT);
          lds_rdtr<off + 4 * RB>(t2[st_ % 4], aT);
        };
        issue(std::integral_constant<int, 0>{});
        issue(std::integral_constant<int, 1>{});
        issue(std::integral_constant<int, 2>{});
        static_for<0, NC>([&](auto ic) {
          constexpr int st_ = decltype(ic)::value, ct = st_ / 4, sidx = st_ % 4;
          if constexpr (st_ + 3 < NC) issue(std::integral_constant<int, st_ + 3>{});
          lds_waittr<2 * (NC - 1 - st_ < 3 ? NC - 1 - st_ : 3)>(t1[st_ % 4], t2[st_ % 4]);
          const bf16x8 af = __builtin_bit_cast(bf16x8, u32x4{t1[st_ % 4][0], t1[st_ % 4][1], t2[st_ % 4][0], t2[st_ % 4][1]});
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) accY[tm][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pb[tm][sidx], accY[tm][ct], 0, 0, 0);
        });
      }
    }
  }
  if constexpr (XR) {                      // xr[own row][16 ct + 4 q + e]
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const long row = f * N + i0 + 16 * tm + r;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const f32x4 y = accY[tm][ct];
        unsigned* dst = (unsigned*)(out + row * C + 16 * ct + 4 * q);
        dst[0] = pack2(y[0], y[1]); dst[1] = pack2(y[2], y[3]);
      }
    }
  } else {
    // lane (r, q): y[query row][16 ct + 4 q + e]  ->  rowdot = X . y (over the row: the lane's entries, then the four q lanes), dX += y
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const long row = f * N + i0 + 16 * tm + r;
      float dot = 0.f;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const long o = row * C + 16 * ct + 4 * q;
        const unsigned xl = *(const unsigned*)(X + o), xh = *(const unsigned*)(X + o + 2);
        const unsigned ol = *(const unsigned*)(out + o), oh = *(const unsigned*)(out + o + 2);
        auto lo = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
        auto hi = [](unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
        const f32x4 y = accY[tm][ct];
        dot += (lo(xl) * y[0] + hi(xl) * y[1]) + (lo(xh) * y[2] + hi(xh) * y[3]);
        *(unsigned*)(out + o) = pack2(lo(ol) + y[0], hi(ol) + y[1]);
        *(unsigned*)(out + o + 2) = pack2(lo(oh) + y[2], hi(oh) + y[3]);
      }
      dot += __shfl_xor(dot, 16, 64);
      dot += __shfl_xor(dot, 32, 64);
      if (q == 0) rowdot[row] = dot;
    }
  }
}

// The two products of the scores' gradient against X, dS formed in the accumulators and never stored (dS_ij = att_ij (X_i . dxr_j - rowdot_i)):
//   KEY = false: the block owns 128 QUERY rows i (X_i as fragments), streams (X_j, dxr_j) tiles:            dX_i += sum_j dS_ij X_j
//   KEY = true : the block owns 128 KEY rows j (X_j AND dxr_j as fragments), streams X_i tiles with the rows' (lse_i, rowdot_i):   dX_j += sum_i dS_ij X_i
// Same accumulator trick as kk_nxn_bwd: lane (r, q) holds 32 consecutive streamed columns of own row r, which is the B operand of the
// second product as it stands; its A operand is the streamed X tile read transposed.  A tile goes through in two halves of 64 streamed rows
// (scores, their gradient and dS of a half are 80 registers instead of 160).
template <int KS, bool KEY>
__global__ void __launch_bounds__(256, (KS <= 3 ? 2 : 1)) kk_nxn_dx(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Dx,
                                                                     const float* __restrict__ lse_g, const float* __restrict__ rowdot,
                                                                     unsigned short* __restrict__ out, int N) {
  constexpr int C = 32 * KS, RB = C * 2 + 16, CPR = C / 8, NLD = 128 * CPR / 256, CT = C / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sD = smem + 128 * RB;                               // (query owner: the streamed dxr tile)
  float* s_lse = (float*)(smem + 128 * RB);                 // (key owner: the streamed rows' log-sum-exp and row dots)
  float* s_rd = s_lse + 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const long f = blockIdx.y;
  const unsigned short* Xf = X + f * (long)N * C;
  const unsigned short* Df = Dx + f * (long)N * C;
  const int i0 = blockIdx.x * 128 + 32 * wave;
  bf16x8 qf[2][KS], df[KEY ? 2 : 1][KEY ? KS : 1];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[tm][ks] = *(const bf16x8*)(Xf + (long)(i0 + 16 * tm + r) * C + ks * 32 + 8 * q);
      if constexpr (KEY) df[tm][ks] = *(const bf16x8*)(Df + (long)(i0 + 16 * tm + r) * C + ks * 32 + 8 * q);
    }
  float lse[2] = {0.f, 0.f}, rd[2] = {0.f, 0.f};
  if constexpr (!KEY) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) { lse[tm] = lse_g[f * N + i0 + 16 * tm + r]; rd[tm] = rowdot[f * N + i0 + 16 * tm + r]; }
  }
  f32x4 accY[2][CT];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) accY[tm][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntile = N / 128;
  u32x4 st[NLD], sd[KEY ? 1 : NLD];
  float lnext = 0.f;
  auto gload = [&](int jt) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i;
      const long off = (long)(jt * 128 + c / CPR) * C + (c % CPR) * 8;
      st[i] = *(const u32x4*)(Xf + off);
      if constexpr (!KEY) sd[i] = *(const u32x4*)(Df + off);
    }
    if constexpr (KEY) lnext = tid < 128 ? lse_g[f * N + jt * 128 + tid] : rowdot[f * N + jt * 128 + tid - 128];
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i;
      *(u32x4*)(sX + (c / CPR) * RB + (c % CPR) * 16) = st[i];
      if constexpr (!KEY) *(u32x4*)(sD + (c / CPR) * RB + (c % CPR) * 16) = sd[i];
    }
    if constexpr (KEY) s_lse[tid] = lnext;                  // (s_rd follows s_lse)
  };
  auto rb = [](float v) { return (float)(__bf16)v; };       // (the softmax the engine's products read was bf16: the same factor)
  gload(0);
  for (int jt = 0; jt < ntile; ++jt) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (jt + 1 < ntile) gload(jt + 1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                           // streamed rows 32 q + 16 h + 4 t + e of the tile
      f32x4 acc[2][4], g[2][4];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[tm][t] = f32x4{0.f, 0.f, 0.f, 0.f}; g[tm][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      {   // 4 KS steps (ks, t): the streamed rows' fragments, two steps ahead of the products that use them
        constexpr int NA = 4 * KS, NRD = KEY ? 1 : 2;
        const unsigned aX = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(sX + ((r >> 2) * 32 + 16 * h + (r & 3)) * RB + q * 16);
        const unsigned aD = aX + (unsigned)(sD - sX);
        u32x4 kfr[3], kdr[3];
        auto issue = [&](auto ic) {
          constexpr int st_ = decltype(ic)::value, ks = st_ / 4, t = st_ % 4, off = 4 * t * RB + ks * 64;
          lds_rd128<off>(kfr[st_ % 3], aX);
          if constexpr (!KEY) lds_rd128<off>(kdr[st_ % 3], aD);
        };
        issue(std::integral_constant<int, 0>{});
        issue(std::integral_constant<int, 1>{});
        static_for<0, NA>([&](auto ic) {
          constexpr int st_ = decltype(ic)::value, ks = st_ / 4, t = st_ % 4;
          if constexpr (st_ + 2 < NA) issue(std::integral_constant<int, st_ + 2>{});
          constexpr int pend = NRD * (NA - 1 - st_ < 2 ? NA - 1 - st_ : 2);
          if constexpr (KEY) lds_wait<pend>(kfr[st_ % 3]); else lds_wait2<pend>(kfr[st_ % 3], kdr[st_ % 3]);
          const bf16x8 kf = __builtin_bit_cast(bf16x8, kfr[st_ % 3]);
          if constexpr (KEY) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
              acc[tm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[tm][ks], acc[tm][t], 0, 0, 0);      // X_i . X_j
              g[tm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, df[tm][ks], g[tm][t], 0, 0, 0);          // X_i . dxr_j
            }
          } else {
            const bf16x8 kd = __builtin_bit_cast(bf16x8, kdr[st_ % 3]);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
              acc[tm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[tm][ks], acc[tm][t], 0, 0, 0);      // X_j . X_i
              g[tm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kd, qf[tm][ks], g[tm][t], 0, 0, 0);          // dxr_j . X_i
            }
          }
        });
      }
      bf16x8 pb[2][2];
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        float ds[2][2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * sp + u;
          float lv[4], rv[4];
          if constexpr (KEY) {
            const float4 l4 = *(const float4*)(s_lse + 32 * q + 16 * h + 4 * t), r4 = *(const float4*)(s_rd + 32 * q + 16 * h + 4 * t);
            lv[0] = l4.x; lv[1] = l4.y; lv[2] = l4.z; lv[3] = l4.w; rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
          }
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float l = KEY ? lv[e] : lse[tm], d0 = KEY ? rv[e] : rd[tm];
              ds[tm][u][e] = rb(__expf(acc[tm][t][e] - l)) * (g[tm][t][e] - d0);
            }
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          pb[tm][sp] = bf16x8{(__bf16)ds[tm][0][0], (__bf16)ds[tm][0][1], (__bf16)ds[tm][0][2], (__bf16)ds[tm][0][3],
                              (__bf16)ds[tm][1][0], (__bf16)ds[tm][1][1], (__bf16)ds[tm][1][2], (__bf16)ds[tm][1][3]};
      }
      {   // 2 CT steps (ct, sp): X^T fragments (channel 16 ct + r, streamed rows 32 q + 16 h + 8 sp + 0 .. 7) by transposing reads, three steps ahead
        constexpr int NC = 2 * CT;
        const unsigned aT = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(sX + (32 * q + 16 * h + (r >> 2)) * RB + (4 * (r & 3)) * 2);
        u32x2 t1[4], t2[4];
        auto issue = [&](auto ic) {
          constexpr int st_ = decltype(ic)::value, ct = st_ / 2, sp = st_ % 2, off = 8 * sp * RB + 32 * ct;
          lds_rdtr<off>(t1[st_ % 4], aT);
          lds_rdtr<off + 4 * RB>(t2[st_ % 4], aT);
        };
        issue(std::integral_constant<int, 0>{});
        issue(std::integral_constant<int, 1>{});
        issue(std::integral_constant<int, 2>{});
        static_for<0, NC>([&](auto ic) {
          constexpr int st_ = decltype(ic)::value, ct = st_ / 2, sp = st_ % 2;
          if constexpr (st_ + 3 < NC) issue(std::integral_constant<int, st_ + 3>{});
          constexpr int pend = 2 * (NC - 1 - st_ < 3 ? NC - 1 - st_ : 3);
          lds_waittr<pend>(t1[st_ % 4], t2[st_ % 4]);
          const bf16x8 af = __builtin_bit_cast(bf16x8, u32x4{t1[st_ % 4][0], t1[st_ % 4][1], t2[st_ % 4][0], t2[st_ % 4][1]});
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) accY[tm][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pb[tm][sp], accY[tm][ct], 0, 0, 0);
        });
      }
    }
  }
  // lane (r, q): the sum for own row 16 tm + r, channels 16 ct + 4 q + e  ->  dX += it
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const long row = f * N + i0 + 16 * tm + r;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const long o = row * C + 16 * ct + 4 * q;
      const unsigned ol = *(const unsigned*)(out + o), oh = *(const unsigned*)(out + o + 2);
      auto lo = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
      auto hi = [](unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
      const f32x4 y = accY[tm][ct];
      *(unsigned*)(out + o) = pack2(lo(ol) + y[0], hi(ol) + y[1]);
      *(unsigned*)(out + o + 2) = pack2(lo(oh) + y[2], hi(oh) + y[3]);
    }
  }
}

}  // namespace

bool nxn_att_ok(int bf16, int N, int C, int Np) { return bf16 && (C == 96 || C == 192) && N % 128 == 0 && N >= 128 && Np % 8 == 0; }

static int nxn_launch(const char* name, const void* X, const void* K, int frames, int N, int C, int Np, float* lse, void* out, const void* att_in, int have_lse,
                      hipStream_t st) {
  if (!nxn_att_ok(1, N, C, Np)) { set_last_error("nxn_att: shape not served (C = 96 / 192, N a multiple of 128)"); return ERR_UNSUPPORTED; }
  if (frames <= 0) return OK;
  const double bytes = (double)frames * N * ((double)C * 2 * (att_in ? 2 : 1) + (double)Np * 2 * (att_in ? 2 : 1) + 4);
  const double flops = (have_lse ? 1.0 : 2.0) * 2.0 * frames * (double)N * N * C;
  ProfScope ps_(name, (long)frames * N, bytes, flops, st);
  const dim3 grid((unsigned)(N / 128), (unsigned)frames);
  if (C == 96) hipLaunchKernelGGL(kk_nxn_att<3>, grid, dim3(256), 128 * (96 * 2 + 16), st, (const unsigned short*)X, (const unsigned short*)K, lse,
                                  (unsigned short*)out, (const unsigned short*)att_in, N, Np, have_lse);
  else hipLaunchKernelGGL(kk_nxn_att<6>, grid, dim3(256), 128 * (192 * 2 + 16), st, (const unsigned short*)X, (const unsigned short*)K, lse,
                          (unsigned short*)out, (const unsigned short*)att_in, N, Np, have_lse);
  AVMOE_CHECK_LAUNCH("nxn_att");
  return OK;
}
int k_nxn_att(const void* X, int frames, int N, int C, int Np, float* lse, void* att, int have_lse, hipStream_t st) {
  return nxn_launch("k_nxn_att", X, X, frames, N, C, Np, lse, att, nullptr, have_lse, st);
}
// dS = att * (X dxr^T - rowdot): the softmax backward of the same block, the gradient of the scores never stored
int k_nxn_att_bwd(const void* X, const void* dxr, int frames, int N, int C, int Np, const float* rowdot, const void* att, void* dS, hipStream_t st) {
  return nxn_launch("k_nxn_att_bwd", X, dxr, frames, N, C, Np, (float*)rowdot, dS, att, 1, st);
}

// y = att dxr without att in memory (re-formed from the kept row log-sum-exp): rowdot = X . y written, dX += y
static int nxn_bwd_launch(int mode, const void* X, const void* dxr, int frames, int N, int C, int Np, const float* lse, float* rowdot, void* out, hipStream_t st) {
  if (!nxn_att_ok(1, N, C, Np)) { set_last_error("nxn_bwd: shape not served (C = 96 / 192, N a multiple of 128)"); return ERR_UNSUPPORTED; }
  if (frames <= 0) return OK;
  const double bytes = (double)frames * N * ((double)C * 2 * (mode == 2 ? 2 : 4) + 8);
  const double flops = 2.0 * 2.0 * frames * (double)N * N * C;
  ProfScope ps_(mode == 2 ? "k_nxn_xr" : "k_nxn_y", (long)frames * N, bytes, flops, st);
  const dim3 grid((unsigned)(N / 128), (unsigned)frames);
  const int lds = mode == 2 ? 128 * (C * 2 + 16) + 512 : 2 * 128 * (C * 2 + 16);
#define NXB(KS_, DS_)                                                                                                     \
  do {                                                                                                                    \
    static LdsAttrOnce attr;                                                                                              \
    AVMOE_TRY(attr.ensure((const void*)kk_nxn_bwd<KS_, DS_>, lds, "nxn_bwd"));                                            \
    hipLaunchKernelGGL((kk_nxn_bwd<KS_, DS_>), grid, dim3(256), lds, st, (const unsigned short*)X, (const unsigned short*)dxr, lse, rowdot, \
                       (unsigned short*)out, N, Np);                                                                      \
  } while (0)
  if (C == 96) { if (mode == 2) NXB(3, 2); else NXB(3, 0); }
  else { if (mode == 2) NXB(6, 2); else NXB(6, 0); }
#undef NXB
  AVMOE_CHECK_LAUNCH("nxn_bwd");
  return OK;
}
int k_nxn_y(const void* X, const void* dxr, int frames, int N, int C, int Np, const float* lse, float* rowdot, void* dX, hipStream_t st) {
  return nxn_bwd_launch(0, X, dxr, frames, N, C, Np, lse, rowdot, dX, st);
}
// the forward's xr = att^T X from the row log-sum-exp (k_nxn_att with att == nullptr leaves it): the softmax is never stored
int k_nxn_xr(const void* X, int frames, int N, int C, int Np, const float* lse, void* xr, hipStream_t st) {
  return nxn_bwd_launch(2, X, X, frames, N, C, Np, lse, nullptr, xr, st);
}

// dX += dS X (query owner) or dS^T X (key owner) with dS re-formed in the accumulators: neither att nor dS exists in memory
int k_nxn_dx(int key, const void* X, const void* dxr, int frames, int N, int C, int Np, const float* lse, const float* rowdot, void* dX, hipStream_t st) {
  if (!nxn_att_ok(1, N, C, Np)) { set_last_error("nxn_dx: shape not served (C = 96 / 192, N a multiple of 128)"); return ERR_UNSUPPORTED; }
  if (frames <= 0) return OK;
  const double bytes = (double)frames * N * ((double)C * 2 * (key ? 4 : 4) + 8);
  const double flops = 3.0 * 2.0 * frames * (double)N * N * C;
  ProfScope ps_(key ? "k_nxn_dxk" : "k_nxn_dxq", (long)frames * N, bytes, flops, st);
  const dim3 grid((unsigned)(N / 128), (unsigned)frames);
  const int lds = key ? 128 * (C * 2 + 16) + 1024 : 2 * 128 * (C * 2 + 16);
#define NXD(KS_, KEY_)                                                                                                   \
  {                                                                                                                      \
    static LdsAttrOnce attr;                                                                                             \
    AVMOE_TRY(attr.ensure((const void*)kk_nxn_dx<KS_, KEY_>, lds, "nxn_dx"));                                            \
    hipLaunchKernelGGL((kk_nxn_dx<KS_, KEY_>), grid, dim3(256), lds, st, (const unsigned short*)X, (const unsigned short*)dxr, lse, rowdot, \
                       (unsigned short*)dX, N);                                                                          \
  }
  if (C == 96) { if (key) NXD(3, true) else NXD(3, false) }
  else { if (key) NXD(6, true) else NXD(6, false) }
#undef NXD
  AVMOE_CHECK_LAUNCH("nxn_dx");
  return OK;
}

}  // namespace avmoe
