// Bottleneck-space kernels, register-resident form for the shape every ViT-B/16 / HTS-AT site of the reference has:
// bottleneck 64 in 2 conv groups (reduction 12 of 768 channels), 32 latent tokens (net_trans_v3.py:296-487).
// Same arithmetic as the run-time-shaped kernels of tile_kernels.hip (names: oracle/algebra_ref.py), different data path:
//
//   * a wavefront owns 16 tokens of ONE expert; lane (r = lane & 15, q = lane >> 4) holds, for token r, the bottleneck entries
//     dd = 16 c + 4 q + x  (c = 0..3 chunk, x = 0..3)  of that expert -- exactly one 16-byte global access per chunk,
//     so Z-space tensors go HBM <-> registers directly, no LDS staging;
//   * the waves of a block take DIFFERENT experts of the SAME tiles (WE<E> below), so that the experts' 64-byte segments of a
//     row are requested together; every expert's per-frame constants sit in LDS at once;
//   * every per-token mat-vec  W[tok][n] = sum_k P[tok][k] M[k][n]  is computed TRANSPOSED on the fp32 matrix pipe:
//     A operand = M^T (lane supplies M[k(step,q)][16 ct + r], from LDS), B operand = P^T (lane supplies its own register
//     P[r][k(step,q)]), and D^T leaves W[tok r][16 ct + 4 q + x] in the same lane layout -- so chains of mat-vecs and
//     elementwise work never leave the register file;
//   * sums over the bottleneck index = in-lane + 2 cross-row shuffles; sums over tokens (BatchNorm statistics) are carried
//     in per-lane accumulators over the whole kernel and folded once at its end (no float atomics: reproducible); sums over the
//     experts of one token go through LDS in expert order.
#include "kernels.h"
#include "device_utils.h"
#include "prof.h"
#include <algorithm>

// waves per SIMD each kernel is compiled for (register budget = 512 / value); measured best on MI355X
#ifndef LB_MIDB
#define LB_MIDB 3
#endif
#ifndef LB_MID
#define LB_MID 4
#endif
#ifndef LB_POST
#define LB_POST 4
#endif
#ifndef LB_POSTB
#define LB_POSTB 2
#endif
// Before a tile loop: everything the per-expert prologue left pending (spill reloads included) is waited for once.  Left to the
// compiler, that wait lands INSIDE the loop (the prologue's loads sit under conditions) -- behind the tile's stores, where the
// in-order memory counter makes it wait for the stores to be acknowledged, every tile.
// the per-site switches of the kernels; KF_ASSUME_FULL (development builds): every one of them a compile-time constant "on"
#if KF_ASSUME_FULL
#define F_LN_BEFORE(a) true
#define F_USE_BN(a) true
#define F_BN_TRAIN(a) true
#define F_LN_POST(a) true
#define F_USE_GATE(a) true
#else
#define F_LN_BEFORE(a) (a.ln_before)
#define F_USE_BN(a) (a.use_bn)
#define F_BN_TRAIN(a) (a.bn_train)
#define F_LN_POST(a) (a.ln_post)
#define F_USE_GATE(a) (a.use_gate)
#endif
#ifndef KF_NO_TILES
#define KF_NO_TILES 0          // development builds: 1 = the per-expert prologues / epilogues only (what the tile loops amortise)
#endif
#define DRAIN_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
#ifndef LB_MIDB_PREFETCH
#define LB_MIDB_PREFETCH 0       // mid_bwd: the same prefetch -- measured neutral (15 spills at three waves per SIMD), off
#endif
#ifndef LB_POSTB_PREFETCH
#define LB_POSTB_PREFETCH 1       // post_small_bwd (bf16, split dApost): next tile's rows requested before the current tile is computed
#endif
#ifndef LB_PRE
#define LB_PRE 3
#endif
#ifndef LB_PREB
#define LB_PREB 2
#endif

namespace avmoe {

#include "tile_fast_dev.h"

namespace {

// =====================================================================================================
// MID backward      (BN2-moment terms + BN1/ReLU mask; algebra_ref.py MID backward)
// =====================================================================================================
struct FMidArgs { int relu_of_e[MAX_E]; FastDims t; int moments; };

template <typename T, int E>
__global__ void __launch_bounds__(WE<E>::NTHR, LB_MIDB) kf_mid_bwd(FMidArgs a, const void* __restrict__ Z_, const float* __restrict__ bn1, const float* __restrict__ dsm,
                                                  const float* __restrict__ sdSzz, void* __restrict__ dzp_, float* __restrict__ colpart) {
  constexpr int DZ = E * FDD, NS = WE<E>::NS, NTHR = WE<E>::NTHR;
  const T* Z = (const T*)Z_; T* dzp = (T*)dzp_;
  __shared__ float s_Se[E][2 * FDG * LD32];
  __shared__ float s_bne[E][5 * FDD];
  __shared__ float s_col[WE<E>::NW * FDD];
  const FastDims& t = a.t;
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int e = wave % E, ts = wave / E;
  const int n_beg = blockIdx.x * t.per, n_end = KF_NO_TILES ? n_beg : min(t.N, n_beg + t.per);
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  if (a.moments)
    for (int i = threadIdx.x; i < E * 2 * FDG * FDG; i += NTHR) {
      const int ee = i >> 11, gi = (i >> 10) & 1, k = (i >> 5) & 31, c = i & 31;
      s_Se[ee][(gi * FDG + c) * LD32 + k] = sdSzz[(long)(gi * E + ee) * FDG * FDG + k * FDG + c];     // transposed (mmT)
    }
  for (int i = threadIdx.x; i < E * FDD; i += NTHR) {
    const int ee = i >> 6, dd = i & 63, col = (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31);
    float* b = s_bne[ee];
    b[dd] = bn1[col]; b[FDD + dd] = bn1[DZ + col]; b[2 * FDD + dd] = bn1[2 * DZ + col];
    b[3 * FDD + dd] = bn1[3 * DZ + col]; b[4 * FDD + dd] = a.moments ? dsm[2 * DZ + col] : 0.f;
  }
  __syncthreads();
  const float* s_S = s_Se[e];
  const float* s_bn = s_bne[e];
  const bool relu = a.relu_of_e[e];
  float4 cs0[4], cs1[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { cs0[c] = zero4(); cs1[c] = zero4(); }
  DRAIN_VMEM();
  // bf16: the rows of the NEXT tile are requested (raw) before the current tile is computed and waited for before its stores
  constexpr bool PF = sizeof(T) == 2 && LB_MIDB_PREFETCH;
  RawRow<T> nz, ndz;
  zero_raw(nz); zero_raw(ndz);
  if constexpr (PF) {
    const int n0 = n_beg + 16 * ts;
    if (n0 < n_end && n0 + r < t.N) { const long row = ((long)s * t.N + n0 + r) * DZ; ldraw_row<E>(Z + row, e, q, nz); ldraw_row<E>(dzp + row, e, q, ndz); }
  }
  for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NS) {
    const int oz = opaque0();
    const bool ok = n0 + r < t.N;
    const long row = ((long)s * t.N + n0 + r) * DZ;
    float4 z[4], dz[4], zp[4], dyo[4];
    if constexpr (PF) {
      unpack_row(nz, z); unpack_row(ndz, dz);
      zero_raw(nz); zero_raw(ndz);
      if (n0 + 16 * NS < n_end && n0 + 16 * NS + r < t.N) { ldraw_row<E>(Z + row + 16L * NS * DZ, e, q, nz); ldraw_row<E>(dzp + row + 16L * NS * DZ, e, q, ndz); }
    } else {
      zero_row(z); zero_row(dz);
      if (ok) { ld_row<T, E>(Z + row, e, q, z); ld_row<T, E>(dzp + row, e, q, dz); }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 sc = ld4(s_bn + oz + 2 * FDD + 16 * c + 4 * q), sh = ld4(s_bn + oz + 3 * FDD + 16 * c + 4 * q);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float y = at(z[c], x) * at(sc, x) + at(sh, x);
        at(zp[c], x) = relu ? fmaxf(y, 0.f) : y;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gi = c >> 1, ct = c & 1;
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      if (a.moments) w = mmT<2, sizeof(T) == 2>(s_S + oz + gi * FDG * LD32, LD32, 16 * ct, zp + 2 * gi, r, q);
      const float4 mean = ld4(s_bn + oz + 16 * c + 4 * q), rstd = ld4(s_bn + oz + FDD + 16 * c + 4 * q), dm = ld4(s_bn + oz + 4 * FDD + 16 * c + 4 * q);
      float4 dy;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float zv = at(z[c], x);
        const float zh = (zv - at(mean, x)) * at(rstd, x);
        const float d = at(dz[c], x) + at(dm, x) + w[x];
        const float v = (!ok || (relu && at(zp[c], x) <= 0.f)) ? 0.f : rndT<T>(d);     // as stored: the BN1 sums see the same numbers
        at(dy, x) = v;
        at(cs0[c], x) += v; at(cs1[c], x) += v * zh;
      }
      dyo[c] = dy;
    }
    if constexpr (PF) __builtin_amdgcn_s_waitcnt(0x0F70);          // the prefetched rows, before this tile's store is issued
    if (ok) st_row<T, E>(dzp + row, e, q, dyo);
  }
  flush_cols_we<E>(cs0, s_col, colpart, blk, 2, e, ts);
  flush_cols_we<E>(cs1, s_col, colpart, blk, 3, e, ts);
}

// =====================================================================================================
// MID forward: z' = act(BN1(z)) -> Zp (operand of the second-moment GEMM) + column sums of z'   (net_trans_v3.py:397-400)
// =====================================================================================================
struct FMidFArgs { int relu_of_e[MAX_E]; FastDims t; };

template <typename T, int E>
__global__ void __launch_bounds__(WE<E>::NTHR, LB_MID) kf_mid(FMidFArgs a, const void* __restrict__ Z_, const float* __restrict__ bn1, void* __restrict__ Zp_,
                                                 float* __restrict__ colpart) {
  constexpr int DZ = E * FDD, NS = WE<E>::NS;
  T* Zp = (T*)Zp_; const T* Z = (const T*)Z_;
  __shared__ float s_col[WE<E>::NW * FDD];
  const FastDims& t = a.t;
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int e = wave % E, ts = wave / E;
  const int n_beg = blockIdx.x * t.per, n_end = KF_NO_TILES ? n_beg : min(t.N, n_beg + t.per);
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  const bool relu = a.relu_of_e[e];
  float4 sc[4], sh[4], cs0[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    sc[c] = ld4(bn1 + 2 * DZ + zcol<E>(c, e, q)); sh[c] = ld4(bn1 + 3 * DZ + zcol<E>(c, e, q)); cs0[c] = zero4();
  }
  constexpr int UT = 4;                                            // tiles per step: all their row loads in flight together
  DRAIN_VMEM();
  for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NS * UT) {
    RawRow<T> z[UT];
    bool ok[UT];
    long tok[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int n = n0 + 16 * NS * u;
      ok[u] = n < n_end && n + r < t.N;
      tok[u] = (long)s * t.N + n + r;
      zero_raw(z[u]);
      if (ok[u]) ldraw_row<E>(Z + tok[u] * DZ, e, q, z[u]);
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      if (!ok[u]) continue;
      float4 zr[4], zp[4];
      unpack_row(z[u], zr);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          float y = at(zr[c], x) * at(sc[c], x) + at(sh[c], x);
          if (relu) y = fmaxf(y, 0.f);
          y = rndT<T>(y);
          at(zp[c], x) = y;
          at(cs0[c], x) += y;
        }
      }
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) st_seg<T>(Zp + tok[u] * DZ + gi * (E * FDG) + e * FDG, zp[2 * gi], zp[2 * gi + 1], q);
    }
  }
  flush_cols_we<E>(cs0, s_col, colpart, blk, 0, e, ts);
}

// =====================================================================================================
// POST_SMALL forward  (net_trans_v3.py:430-434,485-486)
// =====================================================================================================
struct FPostArgs { P16 gate; int relu_of_e[MAX_E]; FastDims t; int ln_post, use_gate; float ln_eps; };

template <typename T, int E>
__global__ void __launch_bounds__(WE<E>::NTHR, LB_POST) kf_post_small(FPostArgs a, const void* __restrict__ Z_, const float* __restrict__ bn1, const float* __restrict__ Gq,
                                                     const float* __restrict__ uvh, const float* __restrict__ probs, void* __restrict__ Apost_,
                                                     float* __restrict__ rpmup) {
  constexpr int DZ = E * FDD, NS = WE<E>::NS, NTHR = WE<E>::NTHR;
  T* Apost = (T*)Apost_; const T* Z = (const T*)Z_;
  __shared__ float s_Ge[E][2 * FDG * LD32];
  __shared__ float s_ce[E][4 * FDD];      // us, vh, sc, sh
  const FastDims& t = a.t;
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int e = wave % E, ts = wave / E;
  const int n_beg = blockIdx.x * t.per, n_end = KF_NO_TILES ? n_beg : min(t.N, n_beg + t.per);
  {
    for (int i = threadIdx.x; i < E * 2 * FDG * FDG; i += NTHR) {
      const int ee = i >> 11, gi = (i >> 10) & 1, k = (i >> 5) & 31, c = i & 31;
      s_Ge[ee][(gi * FDG + c) * LD32 + k] = Gq[(long)(gi * E + ee) * FDG * FDG + k * FDG + c];     // transposed (mmT)
    }
    for (int i = threadIdx.x; i < E * FDD; i += NTHR) {
      const int ee = i >> 6, dd = i & 63, col = (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31);
      float* c_ = s_ce[ee];
      c_[dd] = uvh[col]; c_[FDD + dd] = uvh[DZ + col]; c_[2 * FDD + dd] = bn1[2 * DZ + col]; c_[3 * FDD + dd] = bn1[3 * DZ + col];
    }
    const float* s_G = s_Ge[e];
    const float* s_c = s_ce[e];
    float H1 = 0.f, H2 = 0.f;
    for (int gi = 0; gi < 2; ++gi) { H1 += uvh[2 * DZ + gi * E + e]; H2 += uvh[2 * DZ + 2 * E + gi * E + e]; }
    __syncthreads();
    const bool relu = a.relu_of_e[e];
    const float gate = F_USE_GATE(a) ? a.gate.p[e][0] : 1.f;
    const float qv = probs[(long)s * E + e] * gate;
    DRAIN_VMEM();
    RawRow<T> nz;                                             // the next tile's row of Z, raw
    zero_raw(nz);
    { const int n0 = n_beg + 16 * ts; if (n0 < n_end && n0 + r < t.N) ldraw_row<E>(Z + ((long)s * t.N + n0 + r) * DZ, e, q, nz); }
    for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NS) {
      const int oz = opaque0();
      const bool ok = n0 + r < t.N;
      const long tok = (long)s * t.N + n0 + r;
      float4 zp[4], zraw[4];
      unpack_row(nz, zraw);                                  // requested one tile ago
      zero_raw(nz);
      if (n0 + 16 * NS < n_end && n0 + 16 * NS + r < t.N) ldraw_row<E>(Z + (tok + 16 * NS) * DZ, e, q, nz);      // the next tile's row: in flight during this tile's arithmetic
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 sc = ld4(s_c + oz + 2 * FDD + 16 * c + 4 * q), sh = ld4(s_c + oz + 3 * FDD + 16 * c + 4 * q);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float y = at(zraw[c], x) * at(sc, x) + at(sh, x);
          at(zp[c], x) = relu ? fmaxf(y, 0.f) : y;
        }
      }
      float rp = 1.f, mup = 0.f;
      if (F_LN_POST(a)) {
        float so = 0.f, soo = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int gi = c >> 1, ct = c & 1;
          const f32x4 w = mmT<2, sizeof(T) == 2>(s_G + oz + gi * FDG * LD32, LD32, 16 * ct, zp + 2 * gi, r, q);
          const float4 us = ld4(s_c + oz + 16 * c + 4 * q), vh = ld4(s_c + oz + FDD + 16 * c + 4 * q);
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float zv = at(zp[c], x);
            so += zv * at(us, x);
            soo += zv * (w[x] + 2.f * at(vh, x));
          }
        }
        const float So = qsum4(so) + H1, Soo = qsum4(soo) + H2;
        mup = So / (float)t.C;
        rp = rsqrtf(fmaxf(Soo / (float)t.C - mup * mup, 0.f) + a.ln_eps);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);          // the prefetched row, before this tile's stores are issued (see kf_post_small_bwd)
      if (ok) {
        const float sc = qv * rp;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          const float4 v0 = make_float4(sc * zp[2 * gi].x, sc * zp[2 * gi].y, sc * zp[2 * gi].z, sc * zp[2 * gi].w);
          const float4 v1 = make_float4(sc * zp[2 * gi + 1].x, sc * zp[2 * gi + 1].y, sc * zp[2 * gi + 1].z, sc * zp[2 * gi + 1].w);
          st_seg<T>(Apost + (tok * 2 + gi) * t.KPp + e * FDG, v0, v1, q);
        }
        if (q == 2) { rpmup[(long)e * t.NT + tok] = rp; rpmup[(long)t.NT * E + (long)e * t.NT + tok] = mup; }
      }
    }
  }
  // The 3 E scalar columns [q rp, -q rp mup, q] of every (token, group) row of Apost, for all experts at once: one thread per row
  // writes its 6 E bytes as a run (from rp / mup as stored above, token-contiguous) -- inside the expert passes they were three 2-byte
  // stores per lane, token and expert: 28 % of the kernel's time.
  __syncthreads();
  float qe[E];
#pragma unroll
  for (int e = 0; e < E; ++e) qe[e] = probs[(long)s * E + e] * (F_USE_GATE(a) ? a.gate.p[e][0] : 1.f);
  for (int idx = threadIdx.x; idx < 2 * (n_end - n_beg); idx += NTHR) {
    const long tok = (long)s * t.N + n_beg + (idx >> 1);
    float v[3 * E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float rp = rpmup[(long)e * t.NT + tok], mup = rpmup[(long)t.NT * E + (long)e * t.NT + tok];
      v[3 * e] = qe[e] * rp; v[3 * e + 1] = -qe[e] * rp * mup; v[3 * e + 2] = qe[e];
    }
    T* dst = Apost + (tok * 2 + (idx & 1)) * t.KPp + E * FDG;
    if constexpr (sizeof(T) == 2) {
      unsigned* d32 = (unsigned*)dst;                        // (E * FDG is even and KPp a multiple of 4: 4-byte aligned)
#pragma unroll
      for (int j = 0; j + 1 < 3 * E; j += 2) d32[j >> 1] = (unsigned)f2bf(v[j]) | ((unsigned)f2bf(v[j + 1]) << 16);
      if constexpr ((3 * E) & 1) stT<T>(dst, 3 * E - 1, v[3 * E - 1]);
    } else {
#pragma unroll
      for (int j = 0; j < 3 * E; ++j) stT<T>(dst, j, v[j]);
    }
  }
}

// =====================================================================================================
// POST_SMALL backward
// =====================================================================================================
struct FPostBArgs { P16 gate; int relu_of_e[MAX_E]; FastDims t; int ln_post, use_gate; const void* ZpS; float* dSooT; const float* dApx;
                    int dapw; };      // dapw: row width of the split dApost's T columns (E * 32: whole lines)   // ZpS / dSooT: gram64 mode

// D16: dApost arrives as T columns (the E x 32 bottleneck entries per group, row stride KPp) + an fp32 side array dApx
// [token][group][16] with the 3 E scalar columns (the streaming GEMM's split output); otherwise one fp32 array.
template <typename T, int E, bool D16>
__global__ void __launch_bounds__(WE<E>::NTHR, LB_POSTB) kf_post_small_bwd(FPostBArgs a, const void* __restrict__ Z_, const float* __restrict__ bn1, const float* __restrict__ Gq,
                                                         const float* __restrict__ uvh, const float* __restrict__ probs, const float* __restrict__ rpmup,
                                                         const void* __restrict__ dAp_, void* __restrict__ dzp_, void* __restrict__ Zp_, void* __restrict__ Zw_,
                                                         float* __restrict__ colpart, float* __restrict__ blkscal) {
  constexpr int DZ = E * FDD, NS = WE<E>::NS, NTHR = WE<E>::NTHR;
  T* Zp = (T*)Zp_; T* Zw = (T*)Zw_; const T* Z = (const T*)Z_; T* dzp = (T*)dzp_;
  const float* dAp = (const float*)dAp_; const T* dAp16 = (const T*)dAp_;
  __shared__ float s_Ge[E][2 * FDG * LD32];
  __shared__ float s_ce[E][4 * FDD];      // us, vh, sc, sh
  __shared__ float s_col[WE<E>::NW * FDD];
  __shared__ float s_sc[WE<E>::NW];
  const FastDims& t = a.t;
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int n_beg = blockIdx.x * t.per, n_end = KF_NO_TILES ? n_beg : min(t.N, n_beg + t.per);
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  const int e = wave % E, ts = wave / E;
  {
    for (int i = threadIdx.x; i < E * 2 * FDG * FDG; i += NTHR) {
      const int ee = i >> 11, gi = (i >> 10) & 1, k = (i >> 5) & 31, c = i & 31;
      s_Ge[ee][(gi * FDG + c) * LD32 + k] = Gq[(long)(gi * E + ee) * FDG * FDG + k * FDG + c];     // transposed (mmT)
    }
    for (int i = threadIdx.x; i < E * FDD; i += NTHR) {
      const int ee = i >> 6, dd = i & 63, col = (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31);
      float* c_ = s_ce[ee];
      c_[dd] = uvh[col]; c_[FDD + dd] = uvh[DZ + col]; c_[2 * FDD + dd] = bn1[2 * DZ + col]; c_[3 * FDD + dd] = bn1[3 * DZ + col];
    }
    const float* s_G = s_Ge[e];
    const float* s_c = s_ce[e];
    __syncthreads();
    const bool relu = a.relu_of_e[e];
    const float gate = F_USE_GATE(a) ? a.gate.p[e][0] : 1.f;
    const float qv = probs[(long)s * E + e] * gate;
    float sdq = 0.f, sdSo = 0.f, sdSoo = 0.f;
    float4 cs0[4], cs1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { cs0[c] = zero4(); cs1[c] = zero4(); }
    // PFB (bf16 + split dApost): the raw rows of the NEXT tile are requested before the current one is computed -- a wave keeps two
    // tiles of loads in flight (+ 24 registers), which is what bounds these latency-bound kernels
    constexpr bool PFB = D16 && sizeof(T) == 2 && LB_POSTB_PREFETCH;
    RawRow<T> nz, nd;
    float nx[8];
    auto prefetch = [&](int n0p) {
      zero_raw(nz); zero_raw(nd);
#pragma unroll
      for (int i = 0; i < 8; ++i) nx[i] = 0.f;
      if (n0p < n_end && n0p + r < t.N) {
        const long tk = (long)s * t.N + n0p + r;
        ldraw_row<E>(Z + tk * DZ, e, q, nz);
        if constexpr (sizeof(T) == 2) {
          nd.v[0] = *(const uint4*)(dAp16 + (tk * 2) * a.dapw + e * FDG + seg_off8(q));
          nd.v[1] = *(const uint4*)(dAp16 + (tk * 2 + 1) * a.dapw + e * FDG + seg_off8(q));
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          const float* p = a.dApx + (tk * 2 + gi) * 16 + 3 * e;
          nx[3 * gi] = p[0]; nx[3 * gi + 1] = p[1]; nx[3 * gi + 2] = p[2];
        }
        nx[6] = rpmup[(long)e * t.NT + tk]; nx[7] = rpmup[(long)t.NT * E + (long)e * t.NT + tk];
      }
    };
    if constexpr (PFB) prefetch(n_beg + 16 * ts);
    DRAIN_VMEM();
    int qsel = 0;      // the lane (of the four that hold a token) that takes this tile's scalar terms: four times as many partial sums
    for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NS, qsel = (qsel + 1) & 3) {
      const int oz = opaque0();
      const bool ok = n0 + r < t.N;
      const long tok = (long)s * t.N + n0 + r;
      float4 zp[4], d[4], zraw[4], dzo[4];
      float da1 = 0.f, da2 = 0.f, da3 = 0.f, rp = 1.f, mup = 0.f;
      const bool saved_zp = a.ZpS != nullptr;                 // (a stored copy of z': no longer produced by the forward)
      if constexpr (PFB) {
        unpack_row(nz, zraw); unpack_row(nd, d);
        da1 = nx[0] + nx[3]; da2 = nx[1] + nx[4]; da3 = nx[2] + nx[5];
        if (ok) { rp = nx[6]; mup = nx[7]; }
        prefetch(n0 + 16 * NS);
      } else {
        zero_row(zraw);
        if (ok) ld_row<T, E>((saved_zp ? (const T*)a.ZpS : Z) + tok * DZ, e, q, zraw);
        if constexpr (D16) {
          zero_row(d);
          if (ok) { ld_seg<T>(dAp16 + (tok * 2) * a.dapw + e * FDG, d[0], d[1], q); ld_seg<T>(dAp16 + (tok * 2 + 1) * a.dapw + e * FDG, d[2], d[3], q); }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4& z = zraw[c];
        if constexpr (!D16) d[c] = ok ? ld4(dAp + (tok * 2 + (c >> 1)) * t.KPp + e * FDG + (c & 1) * 16 + 4 * q) : zero4();
        const float4 sc = ld4(s_c + oz + 2 * FDD + 16 * c + 4 * q), sh = ld4(s_c + oz + 3 * FDD + 16 * c + 4 * q);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float y = at(z, x) * at(sc, x) + at(sh, x);
          at(zp[c], x) = saved_zp ? at(z, x) : (relu ? fmaxf(y, 0.f) : y);
        }
      }
      if (!PFB && ok) {
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          const float* p = D16 ? a.dApx + (tok * 2 + gi) * 16 + 3 * e : dAp + (tok * 2 + gi) * t.KPp + E * FDG + 3 * e;
          da1 += p[0]; da2 += p[1]; da3 += p[2];
        }
        rp = rpmup[(long)e * t.NT + tok]; mup = rpmup[(long)t.NT * E + (long)e * t.NT + tok];
      }
      float zz = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int x = 0; x < 4; ++x) zz += at(d[c], x) * at(zp[c], x);
      zz = qsum4(zz);
      float dSo = 0.f, dSoo = 0.f;
      if (ok) {
        const float dq = rp * zz + rp * da1 - rp * mup * da2 + da3;
        if (F_LN_POST(a)) {
          const float drp = qv * zz + qv * da1 - qv * mup * da2;
          float dmup = -qv * rp * da2;
          const float dvarp = drp * (-0.5f) * rp * rp * rp;
          dSoo = dvarp / (float)t.C;
          dmup -= 2.f * mup * dvarp;
          dSo = dmup / (float)t.C;
        }
        if (q == qsel) { sdq += dq; sdSo += dSo; sdSoo += dSoo; }
      }
      const float k1 = qv * rp;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int gi = c >> 1, ct = c & 1;
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (F_LN_POST(a)) w = mmT<2, sizeof(T) == 2>(s_G + oz + gi * FDG * LD32, LD32, 16 * ct, zp + 2 * gi, r, q);
        const float4 us = ld4(s_c + oz + 16 * c + 4 * q), vh = ld4(s_c + oz + FDD + 16 * c + 4 * q);
        float4 o;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float zv = at(zp[c], x);
          float dzv = k1 * at(d[c], x);
          if (F_LN_POST(a)) {
            dzv += dSo * at(us, x) + dSoo * (2.f * w[x] + 2.f * at(vh, x));
            at(cs0[c], x) += dSo * zv; at(cs1[c], x) += dSoo * zv;
          }
          at(o, x) = dzv;
        }
        dzo[c] = o;
      }
      // the prefetched rows are waited for HERE, before this tile's stores are issued: a wait placed after them would (the memory
      // counter being in-order) also wait for the stores to be acknowledged -- once per tile
      if constexpr (PFB) __builtin_amdgcn_s_waitcnt(0x0F70);
      if (ok) {
        st_row<T, E>(dzp + tok * DZ, e, q, dzo);
        if (a.dSooT && q == 0) a.dSooT[(long)e * t.NT + tok] = dSoo;
#pragma unroll
        for (int gi = 0; gi < 2 && a.dSooT == nullptr; ++gi) {
          const long seg = tok * DZ + gi * (E * FDG) + e * FDG;
          const float4& z0 = zp[2 * gi]; const float4& z1 = zp[2 * gi + 1];
          st_seg<T>(Zp + seg, z0, z1, q);
          st_seg<T>(Zw + seg, make_float4(dSoo * z0.x, dSoo * z0.y, dSoo * z0.z, dSoo * z0.w),
                    make_float4(dSoo * z1.x, dSoo * z1.y, dSoo * z1.z, dSoo * z1.w), q);
        }
      }
    }
    flush_cols_we<E>(cs0, s_col, colpart, blk, 0, e, ts);
    flush_cols_we<E>(cs1, s_col, colpart, blk, 1, e, ts);
    const float v0 = expert_scalar<E>(wave_sum(sdq), s_sc, e), v1 = expert_scalar<E>(wave_sum(sdSo), s_sc, e), v2 = expert_scalar<E>(wave_sum(sdSoo), s_sc, e);
    if (ts == 0 && lane == 0) { float* o = blkscal + ((long)blk * E + e) * 4; o[0] = v0; o[1] = v1; o[2] = v2; }
  }
}

// =====================================================================================================
// PRE_SMALL forward   (net_trans_v3.py:385-395)
// =====================================================================================================
struct FPreArgs { P16 glat; int lat_of_e[MAX_E]; int nxn_of_e[MAX_E]; long sxr_off[MAX_E]; FastDims t; int ln_before; float ln_eps;
                  const float* ZR; const float* sxr;         // x + g xr experts (AVVP N x N block, frame attention): xr through Wt, row sums
                  const float* L2g; float* L2w; };           // fused logits (Dims::fuse_l2): two per-group partial planes (NT, KL) to add; the sums go to L2w

template <typename T, int E, bool XR>      // XR: the site has x + g xr experts (AVVP N x N block, frame attention)
__global__ void __launch_bounds__(WE<E>::NTHR, LB_PRE) kf_pre_small(FPreArgs a, void* __restrict__ Z_, const float* __restrict__ L2, const float* __restrict__ sxs,
                                                    const float* __restrict__ TT, const float* __restrict__ TW, const float* __restrict__ Tsum,
                                                    const float* __restrict__ wsum, const float* __restrict__ dconst, void* __restrict__ aout_,
                                                    float* __restrict__ rmu, float* __restrict__ colpart) {
  constexpr int DZ = E * FDD, NS = WE<E>::NS, NTHR = WE<E>::NTHR;
  T* aout = (T*)aout_; T* Z = (T*)Z_;
  // dynamic LDS: per cross-modal expert [TT^T (FK x LD32) | TW^T slice (FDD x LD32) | tb (FK)], then per expert [wsum | dconst], then
  // the column-flush scratch
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  constexpr int LATF = FK * LD32 + FDD * LD32 + FK;
  float* s_ce = s_dyn + a.t.El * LATF;
  float* s_col = s_ce + E * 2 * FDD;
  const FastDims& t = a.t;
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int n_beg = blockIdx.x * t.per, n_end = KF_NO_TILES ? n_beg : min(t.N, n_beg + t.per);
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  const int e = wave % E, ts = wave / E;
  for (int ee = 0; ee < E; ++ee) {          // every expert's per-frame constants, by the whole block
    const int ll = a.lat_of_e[ee];
    if (ll < 0) continue;
    float* lt = s_dyn + ll * LATF;
    const float* tt = TT + ((long)s * t.El + ll) * FK * FK;
    kf_fill<4>(FK * FK, NTHR, [&](int i) { return tt[i]; }, [&](int i, float v) { lt[(i & 31) * LD32 + (i >> 5)] = v; });     // transposed (mmT)
    kf_fill<8>(FK * FDD, NTHR,
               [&](int i) { const int k = i >> 6, dd = i & 63; return TW[((long)s * t.KLT + (long)ll * FK + k) * DZ + (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31)]; },
               [&](int i, float v) { const int k = i >> 6, dd = i & 63; lt[FK * LD32 + dd * LD32 + k] = v; });
    if (threadIdx.x < FK) lt[FK * LD32 + FDD * LD32 + threadIdx.x] = Tsum[(long)s * t.KLT + (long)ll * FK + threadIdx.x] / (float)t.C;
  }
  for (int i = threadIdx.x; i < E * FDD; i += NTHR) {
    const int ee = i >> 6, dd = i & 63, col = (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31);
    s_ce[ee * 2 * FDD + dd] = wsum[col]; s_ce[ee * 2 * FDD + FDD + dd] = dconst[col];
  }
  {
    const int l = a.lat_of_e[e];
    const bool nxn = XR && a.nxn_of_e[e] != 0;
    const float gv = (nxn || l >= 0) ? a.glat.p[e][0] : 0.f;
    const float* s_TT = s_dyn + (l >= 0 ? l : 0) * LATF;
    const float* s_TWt = s_TT + FK * LD32;
    const float* s_tb = s_TWt + FDD * LD32;
    const float* s_c = s_ce + e * 2 * FDD;
    __syncthreads();
    float4 cs0[4], cs1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { cs0[c] = zero4(); cs1[c] = zero4(); }
    DRAIN_VMEM();
    for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NS) {
      const int oz = opaque0();
      const bool ok = n0 + r < t.N;
      const long tok = (long)s * t.N + n0 + r;
      float Sx = 0.f, Sxx = 1.f;
      float4 z[4], zo[4], lg[2];
      RawRow<T> zraw_;
      zero_raw(zraw_); lg[0] = zero4(); lg[1] = zero4();
      if (ok) {                    // every load of the tile, requested together
        Sx = sxs[tok]; Sxx = sxs[t.NT + tok];
        ldraw_row<E>(Z + tok * DZ, e, q, zraw_);
        if (l >= 0) {
          const long lo = tok * t.KL + (long)l * FK + 4 * q;
          if (a.L2g) {             // per-group partial sums out of the down projection's pass over X: group 0 + group 1
            const float4 a0 = ld4(a.L2g + lo), a1 = ld4(a.L2g + lo + 16);
            const float4 b0 = ld4(a.L2g + (long)t.NT * t.KL + lo), b1 = ld4(a.L2g + (long)t.NT * t.KL + lo + 16);
            lg[0] = make_float4(a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w);
            lg[1] = make_float4(a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w);
          } else { lg[0] = ld4(L2 + lo); lg[1] = ld4(L2 + lo + 16); }
        }
      }
      unpack_row(zraw_, z);
      float4 av[2] = {zero4(), zero4()};
      if (l >= 0) {
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int x = 0; x < 4; ++x) mx = fmaxf(mx, at(lg[j], x));
        mx = qmax4(mx);
        float sum = 0.f;
        float4 ex[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int x = 0; x < 4; ++x) { at(ex[j], x) = __expf(at(lg[j], x) - mx); sum += at(ex[j], x); }
        sum = qsum4(sum);
        const float inv = ok ? 1.f / sum : 0.f;
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float4 tb = ld4(s_tb + oz + 16 * j + 4 * q);
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float v = rndT<T>(at(ex[j], x) * inv);
            at(av[j], x) = v;
            u1 += v * at(tb, x); u2 += v * at(lg[j], x);
          }
        }
        u1 = qsum4(u1); u2 = qsum4(u2);
        float u3 = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 w = mmT<2, sizeof(T) == 2>(s_TT + oz, LD32, 16 * ct, av, r, q);
#pragma unroll
          for (int x = 0; x < 4; ++x) u3 += w[x] * at(av[ct], x);
        }
        u3 = qsum4(u3);
        Sx += gv * (float)t.C * u1;
        Sxx += 2.f * gv * u2 + gv * gv * u3;
      }
      if (nxn && ok) {             // x' = x + g xr : sums of x' from the sums of x, xr and x . xr   (mgn.py:132-139)
        const float* sxr = a.sxr + a.sxr_off[e];
        Sx += gv * sxr[tok];
        Sxx += 2.f * gv * sxr[2L * t.NT + tok] + gv * gv * sxr[(long)t.NT + tok];
      }
      float mu = 0.f, rr = 1.f;
      if (F_LN_BEFORE(a)) {
        mu = Sx / (float)t.C;
        rr = rsqrtf(fmaxf(Sxx / (float)t.C - mu * mu, 0.f) + a.ln_eps);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
        if (l >= 0) p = mmT<2, sizeof(T) == 2>(s_TWt + oz, LD32, 16 * c, av, r, q);
        if (nxn && ok) {           // xr through Wt (fp32 rows in the Z layout)
          const float4 zr4 = ld4(a.ZR + tok * DZ + (c >> 1) * (E * FDG) + e * FDG + (c & 1) * 16 + 4 * q);
          p = f32x4{zr4.x, zr4.y, zr4.z, zr4.w};
        }
        const float4 ws = ld4(s_c + oz + 16 * c + 4 * q), dc = ld4(s_c + oz + FDD + 16 * c + 4 * q);
        float4 o;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float zr = at(z[c], x) + gv * p[x];
          const float zv = rndT<T>(F_LN_BEFORE(a) ? rr * (zr - mu * at(ws, x)) + at(dc, x) : zr);   // as stored: BN1 statistics of the stored z
          at(o, x) = zv;
          if (ok) { at(cs0[c], x) += zv; at(cs1[c], x) += zv * zv; }
        }
        zo[c] = o;
      }
      // every store of the tile is issued here, behind its last load: a wait that follows a store also waits (in-order counter)
      // for the store to be acknowledged
      if (ok) {
        if (l >= 0) st_seg<T>(aout + (long)l * t.aL + tok * FK, av[0], av[1], q);
        if (l >= 0 && a.L2w) {       // the logits themselves (the backward's hop-2 block reads them)
          float* lw = a.L2w + tok * t.KL + (long)l * FK + 4 * q;
          *(float4*)lw = lg[0]; *(float4*)(lw + 16) = lg[1];
        }
        st_row<T, E>(Z + tok * DZ, e, q, zo);
        if (q == 0) { rmu[(long)e * t.NT + tok] = rr; rmu[(long)t.NT * E + (long)e * t.NT + tok] = mu; }
      }
    }
    flush_cols_we<E>(cs0, s_col, colpart, blk, 0, e, ts);
    flush_cols_we<E>(cs1, s_col, colpart, blk, 1, e, ts);
  }
}

// =====================================================================================================
// PRE_SMALL backward, two kernels:
//   kf_pre_small_bwd : BN1 input gradient + folded-LayerNorm sums + dzraw for every expert (136 VGPRs, three waves per SIMD);
//   kf_pre_lat_bwd   : the hop-2 block of the cross-modal experts (softmax backward over the latent tokens, the mat-vecs against
//                      TW / TT on the matrix pipe), from dzraw as stored and the expert's LayerNorm sums (dslat).
// As ONE kernel the hop-2 block's registers (256 + spills) set the occupancy of the whole sweep -- two waves per SIMD for the
// unimodal experts' passes too: 363 us at the cfg-2 audio site, 188 us of it the first kernel's work once that runs alone.
// =====================================================================================================
#ifndef LB_PRELB
#define LB_PRELB 2
#endif
struct FPreBArgs { int lat_of_e[MAX_E]; int nxn_of_e[MAX_E]; int first_of_slot[MAX_E]; long sxr_off[MAX_E]; P16 glat; FastDims t;
                   int ln_before, use_bn, bn_train; const float* ZR; const float* sxr; void* dZR; float* dsr; };

template <typename T, int E, bool XR>
__global__ void __launch_bounds__(WE<E>::NTHR, LB_PREB) kf_pre_small_bwd(FPreBArgs a, const void* __restrict__ Z_, const float* __restrict__ wsum,
                                                        const float* __restrict__ dconst, const float* __restrict__ rmu,
                                                        const float* __restrict__ bn1, const float* __restrict__ dsm, const void* __restrict__ dy_in_,
                                                        void* __restrict__ dZx_, void* __restrict__ dL2x_, float* __restrict__ dslat,
                                                        float* __restrict__ rs2x, float* __restrict__ colpart, float* __restrict__ blkscal) {
  constexpr int DZ = E * FDD, NS = WE<E>::NS, NW = WE<E>::NW, NTHR = WE<E>::NTHR;
  const T* Z = (const T*)Z_; const T* dy_in = (const T*)dy_in_;
  T* dZx = (T*)dZx_; T* dL2x = (T*)dL2x_;
  __shared__ float s_bne[E][7 * FDD];       // mean, rstd, sc, mdy, mdyz, wsum, dconst
  __shared__ float s_col[NW * FDD];
  __shared__ float s_sc[NW];
  __shared__ float s_ln[2][NW][16][2];      // (dSx, dSxx) of each wave's expert for the tile's 16 tokens, two tiles deep
  const FastDims& t = a.t;
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int e = wave % E, ts = wave / E;
  const int n_beg = blockIdx.x * t.per, n_end = KF_NO_TILES ? n_beg : min(t.N, n_beg + t.per);
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  for (int i = threadIdx.x; i < E * FDD; i += NTHR) {
    const int ee = i >> 6, dd = i & 63, col = (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31);
    float* b = s_bne[ee];
    b[dd] = bn1[col]; b[FDD + dd] = bn1[DZ + col]; b[2 * FDD + dd] = bn1[2 * DZ + col];
    b[3 * FDD + dd] = F_BN_TRAIN(a) ? dsm[3 * DZ + col] : 0.f; b[4 * FDD + dd] = F_BN_TRAIN(a) ? dsm[4 * DZ + col] : 0.f;
    b[5 * FDD + dd] = wsum[col]; b[6 * FDD + dd] = dconst[col];
  }
  const float* s_bn = s_bne[e];
  const int l = a.lat_of_e[e];
  const bool nxn = XR && a.nxn_of_e[e] != 0;
  const float gv = nxn ? a.glat.p[e][0] : 0.f;
  __syncthreads();
  float sdg = 0.f;
  float4 cs0[4], cs1[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { cs0[c] = zero4(); cs1[c] = zero4(); }
  DRAIN_VMEM();
  // the rows of the NEXT tile are requested (raw) before the current tile is computed and waited for before its stores (a load
  // behind a store would -- the memory counter being in-order -- wait for the store to be acknowledged)
  RawRow<T> nz_, ndy_;
  float nrr = 1.f, nmu = 0.f;
  auto request = [&](int n0) {
    zero_raw(nz_); zero_raw(ndy_); nrr = 1.f; nmu = 0.f;
    if (n0 < n_end && n0 + r < t.N) {
      const long tok = (long)s * t.N + n0 + r;
      ldraw_row<E>(Z + tok * DZ, e, q, nz_); ldraw_row<E>(dy_in + tok * DZ, e, q, ndy_);
      if (F_LN_BEFORE(a)) { nrr = rmu[(long)e * t.NT + tok]; nmu = rmu[(long)t.NT * E + (long)e * t.NT + tok]; }
    }
  };
  request(n_beg + 16 * ts);
  int par = 0;
  int qsel = 0;      // the lane (of the four that hold a token) that takes this tile's scalar terms: four times as many partial sums
  for (int nb = n_beg; nb < n_end; nb += 16 * NS, par ^= 1, qsel = (qsel + 1) & 3) {      // (uniform trip count: one barrier per step)
    const int n0 = nb + 16 * ts;
    const int oz = opaque0();
    const bool ok = n0 < n_end && n0 + r < t.N;
    const long tok = (long)s * t.N + n0 + r;
    const RawRow<T> zraw_ = nz_, dyraw_ = ndy_;
    const float rr = nrr, mu = nmu;
    request(n0 + 16 * NS);
    const float irr = 1.f / rr;
    // ---- BN1 input gradient, folded-LayerNorm sums, dzraw ----
    float4 dzr[4], zrow[4], dyrow[4];
    float s_dr = 0.f, s_dmu = 0.f;
    unpack_row(zraw_, zrow); unpack_row(dyraw_, dyrow);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4& z = zrow[c];
      const float4& dyv = dyrow[c];
      const float4 mean = ld4(s_bn + oz + 16 * c + 4 * q), rstd = ld4(s_bn + oz + FDD + 16 * c + 4 * q), sc = ld4(s_bn + oz + 2 * FDD + 16 * c + 4 * q);
      const float4 mdy = ld4(s_bn + oz + 3 * FDD + 16 * c + 4 * q), mdyz = ld4(s_bn + oz + 4 * FDD + 16 * c + 4 * q);
      const float4 ws = ld4(s_bn + oz + 5 * FDD + 16 * c + 4 * q), dc = ld4(s_bn + oz + 6 * FDD + 16 * c + 4 * q);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        float v = 0.f;
        if (ok) {
          const float zv = at(z, x);
          float dz = at(dyv, x);
          if (F_USE_BN(a)) {
            if (F_BN_TRAIN(a)) dz = at(sc, x) * (dz - at(mdy, x) - (zv - at(mean, x)) * at(rstd, x) * at(mdyz, x));
            else dz = at(sc, x) * dz;
          }
          if (F_LN_BEFORE(a)) {
            const float zc = (zv - at(dc, x)) * irr;
            at(cs0[c], x) += dz; at(cs1[c], x) += -rr * mu * dz;
            s_dr += dz * zc; s_dmu += dz * at(ws, x);
            v = rr * dz;
          } else v = dz;
        }
        at(dzr[c], x) = v;
      }
    }
    float szr = 0.f;
    if (nxn) {                   // x' = x + g xr : d(xr Wt^T) = g dzraw, and dzraw . (xr Wt^T) for the gate
      if (ok) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 zr4 = ld4(a.ZR + tok * DZ + (c >> 1) * (E * FDG) + e * FDG + (c & 1) * 16 + 4 * q);
#pragma unroll
          for (int x = 0; x < 4; ++x) szr += at(dzr[c], x) * at(zr4, x);
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          float4 g0, g1;
#pragma unroll
          for (int x = 0; x < 4; ++x) { at(g0, x) = gv * at(dzr[2 * gi], x); at(g1, x) = gv * at(dzr[2 * gi + 1], x); }
          st_seg<T>((T*)a.dZR + tok * DZ + gi * (E * FDG) + e * FDG, g0, g1, q);
        }
      }
      szr = qsum4(szr);
    }
    float dSx = 0.f, dSxx = 0.f;
    if (F_LN_BEFORE(a)) {
      const float sdr = qsum4(s_dr), sdm = qsum4(s_dmu);
      float dmu = -rr * sdm;
      const float dvar = sdr * (-0.5f) * rr * rr * rr;
      dSxx = dvar / (float)t.C;
      dmu -= 2.f * mu * dvar;
      dSx = dmu / (float)t.C;
    }
    if (!ok) { dSx = 0.f; dSxx = 0.f; }
    if (q == 0) { s_ln[par][wave][r][0] = dSx; s_ln[par][wave][r][1] = dSxx; }      // for the sum over the experts below
    if (nxn && ok && q == qsel) {   // the gate's share of the statistics gradients of this expert's xr slot
      const float* sxr = a.sxr + a.sxr_off[e];
      sdg += dSx * sxr[tok] + dSxx * (2.f * sxr[2L * t.NT + tok] + 2.f * gv * sxr[(long)t.NT + tok]) + szr;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // the prefetched rows, before this tile's stores are issued
    if (ok) {
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) st_seg<T>(dZx + tok * DZ + gi * (E * FDG) + e * FDG, dzr[2 * gi], dzr[2 * gi + 1], q);
      if (q == 0 && l >= 0) { dslat[(2L * l) * t.NT + tok] = dSx; dslat[(2L * l + 1) * t.NT + tok] = dSxx; }      // this expert's own sums: pre_lat_bwd
    }
    __syncthreads();             // every expert's sums of this step are in s_ln[par]
    if (XR && nxn && a.first_of_slot[e] && ok && q == 0) {
      // statistics gradients to (sum xr, sum xr^2, x . xr) of the xr slot: the experts that share it, in expert order
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int ee = 0; ee < E; ++ee)
        if (a.nxn_of_e[ee] && a.sxr_off[ee] == a.sxr_off[e]) {
          const float g = a.glat.p[ee][0], dx = s_ln[par][ts * E + ee][r][0], dxx = s_ln[par][ts * E + ee][r][1];
          v0 += g * dx; v1 += 2.f * g * g * dxx; v2 += 2.f * g * dxx;
        }
      float* dsr = a.dsr + a.sxr_off[e];
      dsr[tok] = v0; dsr[(long)t.NT + tok] = v1; dsr[2L * t.NT + tok] = v2;
    }
    if (e == 0 && ok && q == 0) {
      float accx = 0.f, accxx = 0.f;
#pragma unroll
      for (int ee = 0; ee < E; ++ee) { accx += s_ln[par][ts * E + ee][r][0]; accxx += s_ln[par][ts * E + ee][r][1]; }
      stT<T>(dL2x, tok * t.KLp + t.KL, accx);
      stT<T>(dL2x, tok * t.KLp + t.KL + 1, 1.f);
      rs2x[tok] = 2.f * accxx;
    }
  }
  flush_cols_we<E>(cs0, s_col, colpart, blk, 0, e, ts);
  flush_cols_we<E>(cs1, s_col, colpart, blk, 1, e, ts);
  const float vg = expert_scalar<E>(wave_sum(sdg), s_sc, e);
  if (ts == 0 && lane == 0) blkscal[((long)blk * E + e) * 4 + 3] = vg;      // (a cross-modal expert's slot is rewritten by pre_lat_bwd)
}

// The hop-2 block: the block's waves take DIFFERENT cross-modal experts of the same tiles (wave = tile slot * El + latent index);
// El is a run-time value (1 .. 4 cross-modal experts), the row stride E * 64 a compile-time one.
struct FPreLArgs { int e_of_lat[MAX_E]; P16 glat; FastDims t; };

template <typename T, int E>
__global__ void __launch_bounds__(256, LB_PRELB) kf_pre_lat_bwd(FPreLArgs a, const float* __restrict__ L2, const float* __restrict__ TT,
                                                       const float* __restrict__ TW, const float* __restrict__ Tsum, const void* __restrict__ ain_,
                                                       const void* __restrict__ dZx_, const float* __restrict__ dslat, void* __restrict__ dL2x_,
                                                       void* __restrict__ aw_, void* __restrict__ ag_, float* __restrict__ blkscal,
                                                       float* __restrict__ dtbp) {
  constexpr int DZ = E * FDD;
  const T* ain = (const T*)ain_; const T* dZx = (const T*)dZx_;
  T* dL2x = (T*)dL2x_; T* aw_o = (T*)aw_; T* ag_o = (T*)ag_;
  // dynamic LDS: per cross-modal expert [TT^T (FK x LD32) | TW (FK x LD64) | TW^T (FDD x LD32) | tb (FK)], then scratch [4][64] + [4]
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  constexpr int LATF = FK * LD32 + FK * LD64 + FDD * LD32 + FK;
  const FastDims& t = a.t;
  const int El = t.El, NSL = El >= 3 ? 1 : 4 / El;          // tile slots (El = 3: the fourth wave idles)
  float* s_col = s_dyn + El * LATF;
  float* s_sc = s_col + 4 * FDD;
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const bool act = wave < El * NSL;
  const int l = act ? wave % El : 0, ts = act ? wave / El : 0;
  const int e = a.e_of_lat[l];
  const int n_beg = blockIdx.x * t.per, n_end = (KF_NO_TILES || !act) ? n_beg : min(t.N, n_beg + t.per);
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  for (int ll = 0; ll < El; ++ll) {
    const int ee = a.e_of_lat[ll];
    float* lt = s_dyn + ll * LATF;
    const float* tt = TT + ((long)s * t.El + ll) * FK * FK;
    kf_fill<4>(FK * FK, 256, [&](int i) { return tt[i]; }, [&](int i, float v) { lt[(i & 31) * LD32 + (i >> 5)] = v; });     // transposed (mmT)
    kf_fill<8>(FK * FDD, 256,
               [&](int i) { const int k = i >> 6, dd = i & 63; return TW[((long)s * t.KLT + (long)ll * FK + k) * DZ + (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31)]; },
               [&](int i, float v) { const int k = i >> 6, dd = i & 63; lt[FK * LD32 + k * LD64 + dd] = v; lt[FK * LD32 + FK * LD64 + dd * LD32 + k] = v; });
    if (threadIdx.x < FK) lt[FK * LD32 + FK * LD64 + FDD * LD32 + threadIdx.x] = Tsum[(long)s * t.KLT + (long)ll * FK + threadIdx.x] / (float)t.C;
  }
  const float* s_TT = s_dyn + l * LATF;
  const float* s_TW = s_TT + FK * LD32;       // [k][dd]
  const float* s_TWt = s_TW + FK * LD64;      // [dd][k]
  const float* s_tb = s_TWt + FDD * LD32;
  const float gv = a.glat.p[e][0];
  __syncthreads();
  float sdg = 0.f;
  float4 ck[2];
  ck[0] = zero4(); ck[1] = zero4();
  DRAIN_VMEM();
  // the rows of the NEXT tile are requested (raw) before the current tile is computed and waited for before its stores: load
  // latency and the acknowledgement of the previous tile's stores pass under the mat-vecs instead of in front of them
  RawRow<T> ndraw_;
  RawSeg<T> naraw_;
  float4 nlg[2];
  float ndSx = 0.f, ndSxx = 0.f;
  auto request = [&](int n0) {
    zero_raw(ndraw_); zero_raw(naraw_); nlg[0] = zero4(); nlg[1] = zero4(); ndSx = 0.f; ndSxx = 0.f;
    if (n0 < n_end && n0 + r < t.N) {
      const long tok = (long)s * t.N + n0 + r;
      ldraw_row<E>(dZx + tok * DZ, e, q, ndraw_);
      ldraw_seg(ain + (long)l * t.aL + tok * FK, q, naraw_);
      const long lo = tok * t.KL + (long)l * FK + 4 * q;
      nlg[0] = ld4(L2 + lo); nlg[1] = ld4(L2 + lo + 16);
      ndSx = dslat[(2L * l) * t.NT + tok]; ndSxx = dslat[(2L * l + 1) * t.NT + tok];
    }
  };
  request(n_beg + 16 * ts);
  int qsel = 0;      // the lane (of the four that hold a token) that takes this tile's scalar terms: four times as many partial sums
  for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NSL, qsel = (qsel + 1) & 3) {
    const int oz = opaque0();
    const bool ok = n0 + r < t.N;
    const long tok = (long)s * t.N + n0 + r;
    const RawRow<T> draw_ = ndraw_;
    const RawSeg<T> araw_ = naraw_;
    float4 dzr[4], av[2], lg[2];
    lg[0] = nlg[0]; lg[1] = nlg[1];
    const float dSx = ndSx, dSxx = ndSxx;
    request(n0 + 16 * NSL);
    unpack_row(draw_, dzr);
    unpack_rawseg(araw_, av[0], av[1]);
    float u1 = 0.f, u2 = 0.f;
    float4 tb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      tb[j] = ld4(s_tb + oz + 16 * j + 4 * q);
#pragma unroll
      for (int x = 0; x < 4; ++x) { u1 += at(av[j], x) * at(tb[j], x); u2 += at(av[j], x) * at(lg[j], x); }
    }
    u1 = qsum4(u1); u2 = qsum4(u2);
    float dgr = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {                                 // dzraw . (a TW)
      const f32x4 p = mmT<2, sizeof(T) == 2>(s_TWt + oz, LD32, 16 * c, av, r, q);
#pragma unroll
      for (int x = 0; x < 4; ++x) dgr += p[x] * at(dzr[c], x);
    }
    dgr = qsum4(dgr);
    const float du1 = dSx * gv * (float)t.C, du2 = 2.f * gv * dSxx, du3 = gv * gv * dSxx;
    float u3 = 0.f, sada = 0.f;
    float4 da[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const f32x4 ta = mmT<2, sizeof(T) == 2>(s_TT + oz, LD32, 16 * ct, av, r, q);
      const f32x4 twd = mmT<4, sizeof(T) == 2>(s_TW + oz, LD64, 16 * ct, dzr, r, q);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float ac = at(av[ct], x);
        u3 += ta[x] * ac;
        float d = 0.f;
        if (ok) {
          d = gv * twd[x] + du1 * at(tb[ct], x) + du2 * at(lg[ct], x) + 2.f * du3 * ta[x];
          sada += ac * d;
          at(ck[ct], x) += du1 * ac;
        }
        at(da[ct], x) = d;
      }
    }
    u3 = qsum4(u3); sada = qsum4(sada);
    if (ok && q == qsel) sdg += dSx * (float)t.C * u1 + dSxx * (2.f * u2 + 2.f * gv * u3) + dgr;
    __builtin_amdgcn_s_waitcnt(0x0F70);          // the prefetched rows, before this tile's stores are issued
    if (ok) {
      float4 v0[2], v1[2], v2[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float ac = at(av[j], x);
          at(v0[j], x) = du2 * ac + ac * (at(da[j], x) - sada); at(v1[j], x) = du3 * ac; at(v2[j], x) = gv * ac;
        }
      const long so = tok * t.KLp + (long)l * FK, sa = (long)l * t.aL + tok * FK;      // rows of dL2x ; planes of aw / ag
      st_seg<T>(dL2x + so, v0[0], v0[1], q); st_seg<T>(aw_o + sa, v1[0], v1[1], q); st_seg<T>(ag_o + sa, v2[0], v2[1], q);
    }
  }
  // per cross-modal expert: the gate partial and the column sums of du1 * a, folded over the expert's tile slots
  const float ws = wave_sum(sdg);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float v = rsum16(at(ck[j], x));
      if (r == 0) s_col[wave * FDD + 16 * j + 4 * q + x] = v;
    }
  if (lane == 0) s_sc[wave] = ws;
  __syncthreads();
  if (act && ts == 0) {
    float vg = 0.f;
    for (int u = 0; u < NSL; ++u) vg += s_sc[u * El + l];
    if (lane == 0) blkscal[((long)blk * E + e) * 4 + 3] = vg;
    if (lane < FK) {
      float v = 0.f;
      for (int u = 0; u < NSL; ++u) v += s_col[(u * El + l) * FDD + lane];
      dtbp[(long)blk * t.KL + (long)l * FK + lane] = v;
    }
  }
}

FastDims make_fd(const Dims& d, int per) {
  FastDims t;
  t.S = d.S; t.N = d.N; t.C = d.C; t.El = d.El; t.KL = d.KL; t.KLT = d.KLT; t.KLp = d.KLp; t.KPp = d.KPp; t.NT = d.NT; t.per = per; t.aL = d.aL;
  return t;
}
void fast_grid(const Dims& d, dim3* grid, int* per) {
  const int bps = d.nblk_tok / d.S;
  *per = (int)round_up(cdiv(d.N, bps), 16);
  *grid = dim3((unsigned)bps, (unsigned)d.S);
}

}  // namespace

// The register-resident kernels cover exactly: bottleneck 64 in 2 groups, 32 latent tokens, 2 - 4 experts (any variant).
bool tile_fast_shape(const Dims& d) {
  return d.DD == FDD && d.dgp == FDG && d.g == 2 && d.K == FK && d.Kp == FK && d.E >= 2 && d.E <= 4;
}
bool tile_fast_ok(const Dims& d) { return tile_fast_shape(d) && !d.gen; }      // (d.gen at this shape: the development A/B against tile_gen.inc)

// (kernels still in the expert-outer form run 256 threads whatever E is: LAUNCH_TEX*)
#define NTHR_OF(NE) (WE<NE>::NTHR)
// dynamic LDS above the default limit needs the attribute once per kernel function
// per kernel function: its static LDS (asked once) and the per-device flag of the dynamic-LDS limit (raised once to what the CU has left)
struct FastFn { const void* fn; size_t stat; LdsAttrOnce once; };
static FastFn* fast_fn(const void* fn) {
  static FastFn cache[128] = {};      // (a few dozen instantiations; a race between two host threads writes the same values twice)
  for (int i = 0; i < 128; ++i) {
    if (cache[i].fn == fn) return &cache[i];
    if (!cache[i].fn) {
      hipFuncAttributes fa;
      if (hipFuncGetAttributes(&fa, fn) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      cache[i].stat = fa.sharedSizeBytes; cache[i].fn = fn;
      return &cache[i];
    }
  }
  return nullptr;
}
// dynamic LDS above the default limit needs the attribute once per kernel function (and device)
static int fast_lds(const void* fn, size_t bytes, const char* what) {
  if (bytes <= 64 * 1024) return OK;
  FastFn* f = fast_fn(fn);
  if (!f) { set_last_error("%s: kernel attributes", what); return ERR_LAUNCH; }
  if (f->stat + bytes > 160 * 1024) { set_last_error("%s: %zu + %zu bytes of LDS", what, f->stat, bytes); return ERR_UNSUPPORTED; }
  return f->once.ensure(fn, (int)(160 * 1024 - f->stat), what);
}
// Dims::excl (avmoe_moe_desc::shared_gpu: kernels of another stream may be on the GPU).  scripts/mfma_probe.hip, round 6 (profiles/r06_mfma_probe.txt):
// the split-bf16 mat-vecs of these kernels go wrong beside another kernel's long matrix instructions exactly like the fp32 ones, so this family
// takes the CU's LDS too: a grid of at most one block per CU (every small site) asks for 150 KB per block -- nothing fits beside it, and nothing is
// lost; a larger grid asks for 80 KB, two blocks fill a CU (eight waves instead of twelve; a foreign block can only slip in beside the first or
// the last block of a CU).  Large bf16 sites run the passes that use the matrix pipe in csrc/tile_stream.hip, whose blocks own their CU anyway.
static size_t fast_excl_lds(const void* fn, size_t sh, unsigned nblocks, int excl) {
  if (!excl) return sh;
  const FastFn* f = fast_fn(fn);
  if (!f) return sh;
  const size_t target = (int)nblocks <= cu_count() ? (size_t)150 * 1024 : (size_t)80 * 1024;
  return std::max(sh, target > f->stat ? target - f->stat : (size_t)0);
}
#define LAUNCH_K(FN, NTH, SH, ...)                                                                     \
  do {                                                                                                 \
    const size_t sh__ = fast_excl_lds((const void*)(FN), (SH), grid.x * grid.y, d.excl);               \
    AVMOE_TRY(fast_lds((const void*)(FN), sh__, #FN));                                                 \
    hipLaunchKernelGGL((FN), grid, dim3(NTH), sh__, st, __VA_ARGS__);                                  \
  } while (0)
#define LAUNCH_TE1(bf16, KERN, NE, NTH, SH, ...)                                                       \
  do {                                                                                                 \
    if (bf16) LAUNCH_K((KERN<__bf16, NE>), NTH, SH, __VA_ARGS__);                                      \
    else LAUNCH_K((KERN<float, NE>), NTH, SH, __VA_ARGS__);                                            \
  } while (0)
// 4 experts: the cfg-2 configuration; 2 (1 + 1): what the reference's AVE / AVVP launchers ship (AVE/train.sh:7-8); 3: 1 + 2 / 2 + 1
// (wave-per-expert blocks: 256 / 256 / 192 threads)
#define LAUNCH_TE(bf16, KERN, SH, ...)                                                                 \
  do {                                                                                                 \
    if (d.E == 4) LAUNCH_TE1(bf16, KERN, 4, NTHR_OF(4), SH, __VA_ARGS__);                              \
    else if (d.E == 2) LAUNCH_TE1(bf16, KERN, 2, NTHR_OF(2), SH, __VA_ARGS__);                         \
    else LAUNCH_TE1(bf16, KERN, 3, NTHR_OF(3), SH, __VA_ARGS__);                                       \
  } while (0)
// the same kernels launched with 256 threads whatever E is (pre_lat_bwd: its waves map to the cross-modal experts)
#define LAUNCH_TE256(bf16, KERN, SH, ...)                                                              \
  do {                                                                                                 \
    if (d.E == 4) LAUNCH_TE1(bf16, KERN, 4, 256, SH, __VA_ARGS__);                                     \
    else if (d.E == 2) LAUNCH_TE1(bf16, KERN, 2, 256, SH, __VA_ARGS__);                                \
    else LAUNCH_TE1(bf16, KERN, 3, 256, SH, __VA_ARGS__);                                              \
  } while (0)

#define LAUNCH_TEX1(bf16, KERN, NE, XR_, SH, ...)                                                      \
  do {                                                                                                 \
    if (bf16) LAUNCH_K((KERN<__bf16, NE, XR_>), NTHR_OF(NE), SH, __VA_ARGS__);                         \
    else LAUNCH_K((KERN<float, NE, XR_>), NTHR_OF(NE), SH, __VA_ARGS__);                               \
  } while (0)
#define LAUNCH_TEX2(bf16, KERN, XR_, SH, ...)                                                          \
  do {                                                                                                 \
    if (d.E == 4) LAUNCH_TEX1(bf16, KERN, 4, XR_, SH, __VA_ARGS__);                                    \
    else if (d.E == 2) LAUNCH_TEX1(bf16, KERN, 2, XR_, SH, __VA_ARGS__);                               \
    else LAUNCH_TEX1(bf16, KERN, 3, XR_, SH, __VA_ARGS__);                                             \
  } while (0)
#define LAUNCH_TEX(bf16, KERN, SH, ...)                                                                \
  do {                                                                                                 \
    if (d.nxn) LAUNCH_TEX2(bf16, KERN, true, SH, __VA_ARGS__);                                         \
    else LAUNCH_TEX2(bf16, KERN, false, SH, __VA_ARGS__);                                              \
  } while (0)

int kf_pre_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  const Dims& d = pl.d;
  dim3 grid; int per; fast_grid(d, &grid, &per);
  FPreArgs a;
  for (int e = 0; e < MAX_E; ++e) {
    a.glat.p[e] = prm.e[e].gate_lat; a.lat_of_e[e] = d.lat_of_e[e]; a.nxn_of_e[e] = e < d.E ? d.nxn_of_e[e] : 0;
    a.sxr_off[e] = (e < d.E && d.xr_of_e[e] > 0) ? (long)d.xr_of_e[e] * 3 * d.NT : 0;
  }
  a.ZR = (const float*)(saved + pl.o_ZR); a.sxr = (const float*)(saved + pl.o_sxr);
  a.L2g = d.fuse_l2 ? (const float*)(scratch + pl.o_L2g) : nullptr; a.L2w = d.fuse_l2 ? (float*)(saved + pl.o_L2) : nullptr;
  if (d.fuse_l2 && d.g != 2) { set_last_error("pre_small: fused logits are built for two groups"); return ERR_UNSUPPORTED; }
  a.t = make_fd(d, per); a.ln_before = d.ln_before; a.ln_eps = d.ln_eps;
  const size_t sh = ((size_t)d.El * (FK * LD32 + FDD * LD32 + FK) + (size_t)d.E * 2 * FDD + 4 * FDD) * sizeof(float);
  LAUNCH_TEX(d.bf16, kf_pre_small, sh, a, (void*)(saved + pl.o_Z), (const float*)(saved + pl.o_L2), (const float*)(saved + pl.o_sx),
            (const float*)(saved + pl.o_TT), (const float*)(saved + pl.o_TW), (const float*)(saved + pl.o_Tsum),
            (const float*)(saved + pl.o_wsum), (const float*)(saved + pl.o_dconst), (void*)(saved + pl.o_a), (float*)(saved + pl.o_rmu),
            (float*)(scratch + pl.o_colpart));
  AVMOE_CHECK_LAUNCH("pre_small (64/32)");
  return OK;
}

int kf_mid(const Plan& pl, char* saved, char* scratch, hipStream_t st) {
  const Dims& d = pl.d;
  dim3 grid; int per; fast_grid(d, &grid, &per);
  FMidFArgs a;
  for (int e = 0; e < MAX_E; ++e) a.relu_of_e[e] = d.relu_of_e[e];
  a.t = make_fd(d, per);
  LAUNCH_TE(d.bf16, kf_mid, 0, a, (const void*)(saved + pl.o_Z), (const float*)(saved + pl.o_bn1), (void*)(scratch + pl.o_Zp),
            (float*)(scratch + pl.o_colpart));
  AVMOE_CHECK_LAUNCH("mid (64/32)");
  return OK;
}

int kf_post_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  const Dims& d = pl.d;
  dim3 grid; int per; fast_grid(d, &grid, &per);
  FPostArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.gate.p[e] = prm.e[e].gate; a.relu_of_e[e] = d.relu_of_e[e]; }
  a.t = make_fd(d, per); a.ln_post = d.ln_post; a.use_gate = d.use_gate && !d.gate_w; a.ln_eps = d.ln_eps;
  LAUNCH_TE(d.bf16, kf_post_small, 0, a, (const void*)(saved + pl.o_Z), (const float*)(saved + pl.o_bn1), (const float*)(saved + pl.o_Gq),
            (const float*)(saved + pl.o_uvh), (const float*)(saved + pl.o_probs), (void*)(saved + pl.o_Apost), (float*)(saved + pl.o_rpmup));
  AVMOE_CHECK_LAUNCH("post_small (64/32)");
  return OK;
}

int kf_post_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st, int dap16) {
  const Dims& d = pl.d;
  dim3 grid; int per; fast_grid(d, &grid, &per);
  FPostBArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.gate.p[e] = prm.e[e].gate; a.relu_of_e[e] = d.relu_of_e[e]; }
  a.t = make_fd(d, per); a.ln_post = d.ln_post; a.use_gate = d.use_gate && !d.gate_w;
  a.ZpS = nullptr;          // z' is recomputed from z (no stored copy any more: the Gram kernel forms it on the fly too)
  a.dSooT = d.gram64 ? (float*)(scratch + pl.o_dSooT) : nullptr;
  a.dApx = (const float*)(scratch + pl.o_dApx); a.dapw = d.E * d.dgp;
  if (dap16 && !d.bf16) { set_last_error("post_small_bwd: split dApost is a bf16 form"); return ERR_BAD_ARG; }
#define POSTB_ARGS a, (const void*)(saved + pl.o_Z), (const float*)(saved + pl.o_bn1), (const float*)(saved + pl.o_Gq),                       \
            (const float*)(saved + pl.o_uvh), (const float*)(saved + pl.o_probs), (const float*)(saved + pl.o_rpmup),                     \
            (const void*)(scratch + pl.o_dAp), (void*)(scratch + pl.o_dzp), (void*)(scratch + pl.o_Zp), (void*)(scratch + pl.o_Zw),       \
            (float*)(scratch + pl.o_colpart), (float*)(scratch + pl.o_blkscal)
  if (dap16) LAUNCH_TEX2(true, kf_post_small_bwd, true, 0, POSTB_ARGS);
  else LAUNCH_TEX2(d.bf16, kf_post_small_bwd, false, 0, POSTB_ARGS);
#undef POSTB_ARGS
  AVMOE_CHECK_LAUNCH("post_small_bwd (64/32)");
  return OK;
}

int kf_mid_bwd(const Plan& pl, char* saved, char* scratch, hipStream_t st) {
  const Dims& d = pl.d;
  dim3 grid; int per; fast_grid(d, &grid, &per);
  FMidArgs a;
  for (int e = 0; e < MAX_E; ++e) a.relu_of_e[e] = d.relu_of_e[e];
  a.t = make_fd(d, per); a.moments = d.use_bn && d.training;
  LAUNCH_TE(d.bf16, kf_mid_bwd, 0, a, (const void*)(saved + pl.o_Z), (const float*)(saved + pl.o_bn1),
            (const float*)(scratch + pl.o_dsm), (const float*)(scratch + pl.o_sdSzz), (void*)(scratch + pl.o_dzp),
            (float*)(scratch + pl.o_colpart));
  AVMOE_CHECK_LAUNCH("mid_bwd (64/32)");
  return OK;
}

int kf_pre_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  const Dims& d = pl.d;
  dim3 grid; int per; fast_grid(d, &grid, &per);
  FPreBArgs a;
  {
    bool seen[MAX_E] = {};
    for (int e = 0; e < MAX_E; ++e) {
      a.glat.p[e] = prm.e[e].gate_lat; a.lat_of_e[e] = d.lat_of_e[e];
      a.nxn_of_e[e] = 0; a.first_of_slot[e] = 0; a.sxr_off[e] = 0;
      if (e < d.E && d.nxn_of_e[e]) {
        const int slot = d.xr_of_e[e];
        a.nxn_of_e[e] = 1; a.sxr_off[e] = (long)slot * 3 * d.NT;
        a.first_of_slot[e] = !seen[slot]; seen[slot] = true;
      }
    }
  }
  a.ZR = (const float*)(saved + pl.o_ZR); a.sxr = (const float*)(saved + pl.o_sxr);
  a.dZR = (void*)(scratch + pl.o_dZR); a.dsr = (float*)(scratch + pl.o_dsr);
  a.t = make_fd(d, per); a.ln_before = d.ln_before; a.use_bn = d.use_bn; a.bn_train = d.use_bn && d.training;
  LAUNCH_TEX(d.bf16, kf_pre_small_bwd, 0, a, (const void*)(saved + pl.o_Z), (const float*)(saved + pl.o_wsum),
            (const float*)(saved + pl.o_dconst), (const float*)(saved + pl.o_rmu),
            (const float*)(saved + pl.o_bn1), (const float*)(scratch + pl.o_dsm), (const void*)(scratch + pl.o_dzp),
            (void*)(scratch + pl.o_Zw), (void*)(scratch + pl.o_dL2x), (float*)(scratch + pl.o_dslat),
            (float*)(scratch + pl.o_rs2x), (float*)(scratch + pl.o_colpart), (float*)(scratch + pl.o_blkscal));
  AVMOE_CHECK_LAUNCH("pre_small_bwd (64/32)");
  return OK;
}

// the hop-2 block of the cross-modal experts; after kf_pre_small_bwd on the same stream (reads dZx and dslat, rewrites the experts'
// gate partials in blkscal)
int kf_pre_lat_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  const Dims& d = pl.d;
  if (d.KL == 0) return OK;
  dim3 grid; int per; fast_grid(d, &grid, &per);
  FPreLArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.glat.p[e] = prm.e[e].gate_lat; a.e_of_lat[e] = 0; }
  for (int e = 0; e < d.E; ++e) if (d.lat_of_e[e] >= 0) a.e_of_lat[d.lat_of_e[e]] = e;
  a.t = make_fd(d, per);
  const size_t sh = ((size_t)d.El * (FK * LD32 + FK * LD64 + FDD * LD32 + FK) + 4 * FDD + 4) * sizeof(float);
  LAUNCH_TE256(d.bf16, kf_pre_lat_bwd, sh, a, (const float*)(saved + pl.o_L2), (const float*)(saved + pl.o_TT), (const float*)(saved + pl.o_TW),
            (const float*)(saved + pl.o_Tsum), (const void*)(saved + pl.o_a), (const void*)(scratch + pl.o_Zw),
            (const float*)(scratch + pl.o_dslat), (void*)(scratch + pl.o_dL2x), (void*)(scratch + pl.o_aw), (void*)(scratch + pl.o_ag),
            (float*)(scratch + pl.o_blkscal), (float*)(scratch + pl.o_dtbp));
  AVMOE_CHECK_LAUNCH("pre_lat_bwd (64/32)");
  return OK;
}

}  // namespace avmoe
