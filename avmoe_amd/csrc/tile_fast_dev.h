// Device helpers shared by the register-resident bottleneck-space kernels (tile_fast.hip) and their streaming form (tile_stream.hip):
// the lane layout (16 tokens x 4 quarter-rows per wave), packed bf16 segment accesses, the transposed mat-vec on the matrix pipe,
// the per-block folds of column / scalar accumulators.  Included inside `namespace avmoe`; everything lives in an anonymous namespace
// (one copy per translation unit).
#pragma once
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int FDD = 64;     // bottleneck width of one expert (2 groups x 32)
constexpr int FDG = 32;     // per group
constexpr int FK = 32;      // latent tokens
constexpr int LD32 = 36;    // leading dim of LDS matrices with 32 columns   (4*ld = 16 mod 32: conflict-free A-operand reads)
constexpr int LD64 = 68;    // ... with 64 columns

struct FastDims { int S, N, C, El, KL, KLT, KLp, KPp, NT, per; long aL; };      // aL: plane stride of a / aw / ag ([slot][token][32])

// reductions over the 4 lanes that hold one token (same r, q = 0..3): the gfx950 row swaps v_permlane16_swap (rows 0<->1,
// 2<->3) and v_permlane32_swap (rows 0,1 <-> 2,3) -- plain VALU, no LDS crossbar round trip as with ds_bpermute
__device__ __forceinline__ float qsum4(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float w = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned x = __float_as_uint(w);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float qmax4(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float w = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const unsigned x = __float_as_uint(w);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// sum over the 16 lanes of a row (all tokens of the tile, fixed q): DPP adds -- quad swaps, then half-row and row mirrors
// (after the quad steps every quad is uniform, so a mirror pairs the right partners)
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float rsum16(float v) {
  v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);     // row_half_mirror
  v += dpp_f<0x140>(v);     // row_mirror
  return v;
}
// an integer the optimiser cannot see through (always 0): added to LDS offsets inside the tile loops so that the per-expert
// constants are re-read from LDS each tile instead of being hoisted into (and spilled from) registers
__device__ __forceinline__ int opaque0() { int v = 0; asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <typename T> __device__ __forceinline__ float rndT(float v);
template <> __device__ __forceinline__ float rndT<float>(float v) { return v; }
template <> __device__ __forceinline__ float rndT<__bf16>(float v) { return bf2f(f2bf(v)); }
// ---- 32-wide bf16 segments (one expert's 32 bottleneck entries of a group / one latent slot's 32 tokens) as ONE 16-byte access
// per lane: lanes q and q^1 trade quads with v_permlane16_swap, so that lane q even holds entries 4q .. 4q+7 of the first
// 16-chunk and lane q odd entries 4(q-1) .. 4(q-1)+7 of the second -- the four q lanes cover the 64-byte segment contiguously.
// (fp32 tensors keep their two 16-byte accesses per lane.)  The swap is its own inverse: loads use it the other way round.
__device__ __forceinline__ int seg_off8(int q) { return (q & 1) * 16 + (q >> 1) * 8; }
template <typename T> __device__ __forceinline__ void st_seg(T* seg, const float4& c0, const float4& c1, int q);
template <> __device__ __forceinline__ void st_seg<float>(float* seg, const float4& c0, const float4& c1, int q) {
  *(float4*)(seg + 4 * q) = c0; *(float4*)(seg + 16 + 4 * q) = c1;
}
template <> __device__ __forceinline__ void st_seg<__bf16>(__bf16* seg, const float4& c0, const float4& c1, int q) {
  const unsigned a0 = (unsigned)f2bf(c0.x) | ((unsigned)f2bf(c0.y) << 16), a1 = (unsigned)f2bf(c0.z) | ((unsigned)f2bf(c0.w) << 16);
  const unsigned b0 = (unsigned)f2bf(c1.x) | ((unsigned)f2bf(c1.y) << 16), b1 = (unsigned)f2bf(c1.z) | ((unsigned)f2bf(c1.w) << 16);
  const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
  *(uint4*)(seg + seg_off8(q)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
}
template <typename T> __device__ __forceinline__ void ld_seg(const T* seg, float4& c0, float4& c1, int q);
template <> __device__ __forceinline__ void ld_seg<float>(const float* seg, float4& c0, float4& c1, int q) {
  c0 = *(const float4*)(seg + 4 * q); c1 = *(const float4*)(seg + 16 + 4 * q);
}
template <> __device__ __forceinline__ void ld_seg<__bf16>(const __bf16* seg, float4& c0, float4& c1, int q) {
  const uint4 u = *(const uint4*)(seg + seg_off8(q));
  const auto s0 = __builtin_amdgcn_permlane16_swap(u.x, u.z, false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(u.y, u.w, false, false);
  c0 = make_float4(__uint_as_float(s0[0] << 16), __uint_as_float(s0[0] & 0xffff0000u), __uint_as_float(s1[0] << 16), __uint_as_float(s1[0] & 0xffff0000u));
  c1 = make_float4(__uint_as_float(s0[1] << 16), __uint_as_float(s0[1] & 0xffff0000u), __uint_as_float(s1[1] << 16), __uint_as_float(s1[1] & 0xffff0000u));
}
// one expert's 64 bottleneck entries of a Z-space row ([group][expert][32], element type T): chunks v[0..3]
template <typename T, int E> __device__ __forceinline__ void ld_row(const T* row, int e, int q, float4 (&v)[4]) {
  ld_seg<T>(row + e * FDG, v[0], v[1], q); ld_seg<T>(row + E * FDG + e * FDG, v[2], v[3], q);
}
template <typename T, int E> __device__ __forceinline__ void st_row(T* row, int e, int q, const float4 (&v)[4]) {
  st_seg<T>(row + e * FDG, v[0], v[1], q); st_seg<T>(row + E * FDG + e * FDG, v[2], v[3], q);
}
// the same row as RAW registers (bf16: 2 x 16 bytes): several tiles' loads are kept in flight in this form and only
// unpacked (quad exchange + widen) when a tile is computed
template <typename T> struct RawRow;
template <> struct RawRow<float> { float4 v[4]; };
template <> struct RawRow<__bf16> { uint4 v[2]; };
template <int E> __device__ __forceinline__ void ldraw_row(const float* row, int e, int q, RawRow<float>& o) {
  o.v[0] = *(const float4*)(row + e * FDG + 4 * q); o.v[1] = *(const float4*)(row + e * FDG + 16 + 4 * q);
  o.v[2] = *(const float4*)(row + E * FDG + e * FDG + 4 * q); o.v[3] = *(const float4*)(row + E * FDG + e * FDG + 16 + 4 * q);
}
template <int E> __device__ __forceinline__ void ldraw_row(const __bf16* row, int e, int q, RawRow<__bf16>& o) {
  o.v[0] = *(const uint4*)(row + e * FDG + seg_off8(q)); o.v[1] = *(const uint4*)(row + E * FDG + e * FDG + seg_off8(q));
}
__device__ __forceinline__ void zero_raw(RawRow<float>& o) { o.v[0] = o.v[1] = o.v[2] = o.v[3] = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void zero_raw(RawRow<__bf16>& o) { o.v[0] = o.v[1] = make_uint4(0u, 0u, 0u, 0u); }
__device__ __forceinline__ void unpack_row(const RawRow<float>& i, float4 (&v)[4]) { v[0] = i.v[0]; v[1] = i.v[1]; v[2] = i.v[2]; v[3] = i.v[3]; }
__device__ __forceinline__ void unpack_seg(const uint4& u, float4& c0, float4& c1) {
  const auto s0 = __builtin_amdgcn_permlane16_swap(u.x, u.z, false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(u.y, u.w, false, false);
  c0 = make_float4(__uint_as_float(s0[0] << 16), __uint_as_float(s0[0] & 0xffff0000u), __uint_as_float(s1[0] << 16), __uint_as_float(s1[0] & 0xffff0000u));
  c1 = make_float4(__uint_as_float(s0[1] << 16), __uint_as_float(s0[1] & 0xffff0000u), __uint_as_float(s1[1] << 16), __uint_as_float(s1[1] & 0xffff0000u));
}
__device__ __forceinline__ void unpack_row(const RawRow<__bf16>& i, float4 (&v)[4]) { unpack_seg(i.v[0], v[0], v[1]); unpack_seg(i.v[1], v[2], v[3]); }
// one 32-entry segment in raw form
template <typename T> struct RawSeg;
template <> struct RawSeg<float> { float4 v[2]; };
template <> struct RawSeg<__bf16> { uint4 v; };
__device__ __forceinline__ void zero_raw(RawSeg<float>& o) { o.v[0] = o.v[1] = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void zero_raw(RawSeg<__bf16>& o) { o.v = make_uint4(0u, 0u, 0u, 0u); }
__device__ __forceinline__ void ldraw_seg(const float* seg, int q, RawSeg<float>& o) { o.v[0] = *(const float4*)(seg + 4 * q); o.v[1] = *(const float4*)(seg + 16 + 4 * q); }
__device__ __forceinline__ void ldraw_seg(const __bf16* seg, int q, RawSeg<__bf16>& o) { o.v = *(const uint4*)(seg + seg_off8(q)); }
__device__ __forceinline__ void unpack_rawseg(const RawSeg<float>& i, float4& c0, float4& c1) { c0 = i.v[0]; c1 = i.v[1]; }
__device__ __forceinline__ void unpack_rawseg(const RawSeg<__bf16>& i, float4& c0, float4& c1) { unpack_seg(i.v, c0, c1); }
__device__ __forceinline__ void zero_row(float4 (&v)[4]) { v[0] = v[1] = v[2] = v[3] = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float& at(float4& v, int x) { return ((float*)&v)[x]; }
__device__ __forceinline__ float at(const float4& v, int x) { return ((const float*)&v)[x]; }

// column offset of chunk c (dd = 16 c + 4 q ..) of expert e inside a Z-space row [group][expert][32]
template <int E> __device__ __forceinline__ int zcol(int c, int e, int q) { return (c >> 1) * (E * FDG) + e * FDG + (c & 1) * 16 + 4 * q; }

// W[tok r][col0 + 4 q + x] = sum over NJ chunks of  P[r][16 j + 4 q' + x'] * M[16 j + 4 q' + x'][col0 + ...]
// Mt: the matrix TRANSPOSED in LDS, Mt[n][k] (leading dim ld, a multiple of 4): the A operands of the four MFMA steps x' = 0..3
// of a chunk are then one 16-byte read  Mt[col0 + r][16 j + 4 q .. + 3].   p[j]: this lane's chunk registers.
#ifndef KF_NO_MFMA
#define KF_NO_MFMA 0           // development builds (timing only): 1 = the mat-vecs skip the matrix pipe and the LDS reads
#endif
// The same product on the bf16 matrix pipe in split form (hi.hi + hi.lo + lo.hi of two bf16 planes per operand, fp32 accumulation:
// tile_gen.inc::mmT_split has the derivation and the reason): what the bf16 instantiations of this file run.  Three
// v_mfma_f32_16x16x32_bf16 per pair of 16-entry chunks instead of eight v_mfma_f32_16x16x4_f32: the cfg-2 step 5.12 -> 5.05 ms
// (same-box A/B, round 4).  KF_MMT_BF16=0 (development builds): the exact-fp32 instruction everywhere.
#ifndef KF_MMT_BF16
#define KF_MMT_BF16 1
#endif
typedef __attribute__((ext_vector_type(8))) __bf16 kf_bf16x8;
__device__ __forceinline__ void kf_split8(const float4& v0, const float4& v1, kf_bf16x8& hi, kf_bf16x8& lo) {
  const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)f[i];
    hi[i] = h;
    lo[i] = (__bf16)(f[i] - (float)h);
  }
}
template <int NJ>
__device__ __forceinline__ f32x4 mmT_split(const float* Mt, int ld, int col0, const float4* p, int r, int q) {
  static_assert(NJ % 2 == 0, "chunk pairs");
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* mp = Mt + (col0 + r) * ld + 4 * q;
#pragma unroll
  for (int j = 0; j < NJ; j += 2) {
    kf_bf16x8 ah, al, ph, pl;
    kf_split8(*(const float4*)(mp + 16 * j), *(const float4*)(mp + 16 * (j + 1)), ah, al);
    kf_split8(p[j], p[j + 1], ph, pl);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ph, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ph, acc, 0, 0, 0);
  }
  return acc;
}
template <int NJ, bool SPLIT>
__device__ __forceinline__ f32x4 mmT(const float* Mt, int ld, int col0, const float4* p, int r, int q) {
#if KF_MMT_BF16
  if constexpr (SPLIT) return mmT_split<NJ>(Mt, ld, col0, p, r, q);
#endif
#if KF_NO_MFMA
  return f32x4{p[0].x, p[0].y, p[0].z, p[0].w};
#endif
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* mp = Mt + (col0 + r) * ld + 4 * q;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float4 a = *(const float4*)(mp + 16 * j);
#pragma unroll
    for (int x = 0; x < 4; ++x) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at(a, x), at(p[j], x), acc, 0, 0, 0);
  }
  return acc;
}

// A block's per-frame constants go global -> LDS in BATCHES: U independent loads per thread in flight, then their stores.  As plain loops
// (`for (i = tid; i < n; i += threads) lds[f(i)] = g[h(i)]`) the compiler kept them rolled -- load, s_waitcnt vmcnt(0), store, branch: one
// memory round trip per element and thread, 24 in a row in kf_pre_lat_bwd's prologue (19 us per block, a quarter of the kernel; a
// timing-only build without the fills: -103 us per cfg-2 step over the six kernels).  The loads are unconditional (clamped index: a load under a
// condition is waited for on the spot), the stores conditional.  Used for the latent-token matrices (kf_pre_small, kf_pre_lat_bwd: 126 -> 101 us
// and 47 -> 36 us at the two cfg-2 sites); the d x d matrices of mid_bwd / post_small / post_small_bwd measured the same either way and keep
// the plain loops.
template <int U, typename LD, typename ST>
__device__ __forceinline__ void kf_fill(int n, int nthr, LD&& ld, ST&& st) {
  for (int i0 = threadIdx.x; i0 < n; i0 += U * nthr) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld(min(i0 + u * nthr, n - 1));
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * nthr < n) st(i0 + u * nthr, v[u]);
  }
}

// ---- wave-per-expert form --------------------------------------------------------------------------------------------------
// The waves of a block take DIFFERENT experts of the SAME 16-token tiles (wave = tile slot * E + expert), so the E 64-byte segments
// of a Z-space row [group][expert][32] are requested within the same few hundred cycles -- one DRAM page, neighbouring sectors of
// the same lines -- instead of E sweeps over the block's tokens apart.  Measured on kf_mid_bwd at the cfg-2 audio site (same
// bytes, same occupancy): 183 -> 125 us; with an expert-outer loop the HBM traffic is the same but every sweep touches one
// 64-byte sector in four of each row.
template <int E> struct WE {
  static constexpr int NS = (E == 2) ? 2 : 1;      // tile slots: tiles a block works on at a time
  static constexpr int NW = E * NS, NTHR = 64 * NW;
};
// fold per-lane token-slot accumulators (acc[c][x] for dd = 16 c + 4 q + x of expert e) over the 16 token slots and the tile slots,
// then write colpart[blk][slot][colmap(e, dd)]          s_x: [NW][64]
template <int E>
__device__ __forceinline__ void flush_cols_we(float4 (&acc)[4], float* s_x, float* colpart, int blk, int slot, int e, int ts) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float v = rsum16(at(acc[c], x));
      if (r == 0) s_x[wave * FDD + 16 * c + 4 * q + x] = v;
    }
  __syncthreads();
  if (ts == 0) {
    const int dd = lane;
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < WE<E>::NS; ++u) v += s_x[(u * E + e) * FDD + dd];
    colpart[((long)blk * 4 + slot) * (E * FDD) + (dd >> 5) * (E * FDG) + e * FDG + (dd & 31)] = v;
  }
  __syncthreads();
}
// sum of per-wave values (already wave-uniform) over the tile slots of expert e          s_w: [NW]
template <int E>
__device__ __forceinline__ float expert_scalar(float v, float* s_w, int e) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) s_w[wave] = v;
  __syncthreads();
  float o = 0.f;
#pragma unroll
  for (int u = 0; u < WE<E>::NS; ++u) o += s_w[u * E + e];
  return o;
}

}  // namespace
