// Bottleneck-space kernels, tiled for CDNA4: one wavefront owns a tile of 16 tokens of one frame and every
// per-token mat-vec (hop-2 softmax x K x K / K x d matrices, d x d quadratic forms, BN2-moment terms) is a
// 16 x n x k product on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32), with the per-frame matrices staged
// once per block in LDS.  Arithmetic and names: oracle/algebra_ref.py (PRE_SMALL, MID, POST_SMALL and their backward).
//
// Register layouts inside a wave (lane l, r = l & 15, q = l >> 4):
//   "C layout"  the MFMA result: lane holds rows (tokens) 4q+e, e = 0..3, of column 16*ct + r
//   A operand   read from a per-wave LDS tile  At[16][lda]:  At[r][4*kk + q]        (lda  = 2 mod 32: conflict-free)
//   B operand   read from a per-block LDS matrix Bs[k][ldb]: Bs[4*kk + q][col0 + r] (ldb  = 16 mod 32: conflict-free)
// Row reductions over columns = in-lane over column tiles, then xor-shuffles 1,2,4,8 inside the 16-lane group.
// Column sums over tokens (BatchNorm etc.) = xor-shuffles 16,32, then a per-wave LDS accumulator, flushed per block
// into colpart[blk][slot][col] (summed over blocks by kk_reduce_colpart: no float atomics, reproducible).
#include "kernels.h"
#include "device_utils.h"
#include "prof.h"
#include <algorithm>
#include <cstdlib>

namespace avmoe {

#define DISPATCH_T(bf16, KERN, grid, block, shmem, st, ...)                                   \
  do {                                                                                        \
    if (bf16) hipLaunchKernelGGL((KERN<__bf16>), grid, block, shmem, st, __VA_ARGS__);        \
    else hipLaunchKernelGGL((KERN<float>), grid, block, shmem, st, __VA_ARGS__);              \
  } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <typename T> __device__ __forceinline__ float rndT(float v);
template <> __device__ __forceinline__ float rndT<float>(float v) { return v; }
template <> __device__ __forceinline__ float rndT<__bf16>(float v) { return bf2f(f2bf(v)); }

// sum over the 16 lanes that share q (all columns of a C-layout row)
__device__ __forceinline__ float rsum16(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
// sum over the 4 lanes that share r (A-layout token spread over q, or the 4 row-quads of one column)
__device__ __forceinline__ float qsum4(float v) {
  v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float qmax4(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// 16 x 16 tile of  At (16 x 4*k4) . Bs (4*k4 x ..)  at columns col0..col0+15
__device__ __forceinline__ f32x4 tile_mm(const float* At, int lda, const float* Bs, int ldb, int k4, int col0, int r, int q) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* ap = At + r * lda + q;
  const float* bp = Bs + q * ldb + col0 + r;
  for (int kk = 0; kk < k4; ++kk) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * kk], bp[4 * kk * ldb], acc, 0, 0, 0);
  }
  return acc;
}

static inline int pad_lda(int k) { return (int)round_up(k, 32) + 2; }    // = 2 (mod 32)
static inline int pad_ldb(int n) { return (int)round_up(n, 32) + 16; }   // = 16 (mod 32)
static void tile_grid(const Dims& d, dim3* grid, int* per) {
  const int bps = d.nblk_tok / d.S;
  *per = (int)round_up(cdiv(d.N, bps), 16);
  *grid = dim3((unsigned)bps, (unsigned)d.S);
}
// excl (Dims::excl, avmoe_moe_desc::shared_gpu): other kernels may share the GPU with this call -- the launch then asks for 150 KB of
// dynamic LDS, one block per compute unit, so that no block of another kernel lands beside it (tile_gen.inc::gen_lds_request has the
// reason; these any-shape kernels do their mat-vecs with LDS operands on the fp32 matrix pipe: the pattern that was seen to go wrong)
static int set_lds(const void* fn, size_t& bytes, const char* what, int excl) {
  if (excl) bytes = std::max(bytes, (size_t)150 * 1024);
  if (bytes <= 65536) return OK;
  if (bytes > 160 * 1024) { set_last_error("%s needs %zu B of LDS (num_tk / bottleneck too large)", what, bytes); return ERR_UNSUPPORTED; }
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { set_last_error("%s: LDS attribute: %s", what, hipGetErrorString(e)); return ERR_LAUNCH; }
  return OK;
}

struct TileDims {
  int S, N, C, E, K, Kp, El, KL, KLT, KLp, DD, DZ, dgp, g, KPp, NT, per;
  long aL;       // plane stride of a / aw / ag ([slot][token][Kp])
  int k4;        // ceil(K / 4): contraction steps over the latent index
  int lda_k;     // leading dim of per-wave K-wide tiles
  int ldb_k;     // leading dim of LDS matrices with K columns
  int lda_d;     // leading dim of per-wave DD-wide tiles
  int ldb_d;     // leading dim of LDS matrices with DD columns
  int ldb_g;     // leading dim of LDS matrices with dgp columns
  int lg_pr;     // log2(DD / 4) if a power of two, else -1   (float4 per slab row)
  int lg_sw;     // log2(dgp / 4) if a power of two, else -1  (float4 per group segment)
};
static TileDims make_td(const Dims& d, int per) {
  TileDims t;
  t.S = d.S; t.N = d.N; t.C = d.C; t.E = d.E; t.K = d.K; t.Kp = d.Kp; t.El = d.El; t.KL = d.KL; t.KLT = d.KLT; t.KLp = d.KLp; t.aL = d.aL;
  t.DD = d.DD; t.DZ = d.DZ; t.dgp = d.dgp; t.g = d.g; t.KPp = d.KPp; t.NT = d.NT; t.per = per;
  t.k4 = cdiv(d.K, 4);
  t.lda_k = pad_lda(4 * t.k4); t.ldb_k = pad_ldb(4 * t.k4);
  t.lda_d = pad_lda(d.DD); t.ldb_d = pad_ldb(d.DD); t.ldb_g = pad_ldb(d.dgp);
  auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  t.lg_pr = lg(d.DD / 4); t.lg_sw = lg(d.dgp / 4);
  return t;
}
__device__ __forceinline__ int colmap(const TileDims& a, int e, int dd) {
  if (a.lg_sw >= 0) { const int sh = a.lg_sw + 2; return ((dd >> sh) * a.E + e) * a.dgp + (dd & (a.dgp - 1)); }
  return (dd / a.dgp) * a.E * a.dgp + e * a.dgp + (dd % a.dgp);
}
// flush per-wave column accumulators (LDS, [waves][nslot][width]) of expert e into colpart
__device__ __forceinline__ void flush_colacc(const TileDims& a, float* s_col, int nslot, int e, float* colpart, int blk, int slot0) {
  const int nw = blockDim.x >> 6;
  __syncthreads();
  for (int i = threadIdx.x; i < nslot * a.DD; i += blockDim.x) {
    const int which = i / a.DD, dd = i % a.DD;
    float v = 0.f;
    for (int w = 0; w < nw; ++w) { float* p = s_col + (w * nslot + which) * a.DD + dd; v += *p; *p = 0.f; }
    colpart[((long)blk * 4 + slot0 + which) * a.DZ + colmap(a, e, dd)] = v;
  }
  __syncthreads();
}
// block sum of per-wave scalars: every wave passes v[0..3] (valid in all lanes after wave_sum); out4[i] written for mask bits
__device__ __forceinline__ void flush_scal4(float* s_sc, float v0, float v1, float v2, float v3, float* out4, unsigned mask) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) { s_sc[wave * 4 + 0] = v0; s_sc[wave * 4 + 1] = v1; s_sc[wave * 4 + 2] = v2; s_sc[wave * 4 + 3] = v3; }
  __syncthreads();
  if (threadIdx.x < 4 && ((mask >> threadIdx.x) & 1u)) {
    float acc = 0.f;
    for (int w = 0; w < nw; ++w) acc += s_sc[w * 4 + threadIdx.x];
    out4[threadIdx.x] = acc;
  }
}

// -----------------------------------------------------------------------------------------------------
// Slab I/O: the 16 x DD block of one expert inside a row-major array whose rows hold [group][expert][dgp]
// (element (row, gi, jp) at base[row*row_stride + gi*grp_stride + col0 + jp]).  The whole wave moves it with
// 16-byte accesses, MAXV of them in flight per lane, through the per-wave LDS tile [16][ld] -- so the compute
// phases never wait on global memory inside their loops.
// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void slab_index(const TileDims& t, int idx, int& row, int& gi, int& c4) {
  const int per_row = t.DD >> 2, segw = t.dgp >> 2;
  int rem;
  if (t.lg_pr >= 0) { row = idx >> t.lg_pr; rem = idx & (per_row - 1); } else { row = idx / per_row; rem = idx - row * per_row; }
  if (t.lg_sw >= 0) { gi = rem >> t.lg_sw; c4 = rem & (segw - 1); } else { gi = rem / segw; c4 = rem - gi * segw; }
}
template <int MAXV, typename F>
__device__ __forceinline__ void slab_load(const float* __restrict__ base, long row_stride, long grp_stride, int col0, const TileDims& t,
                                          int nvalid, float* dst, int ld, int lane, F&& xform) {
  const int total = 16 * (t.DD >> 2);
  for (int i0 = 0; i0 < total; i0 += 64 * MAXV) {
    float4 v[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      const int idx = i0 + u * 64 + lane;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < total) {
        int row, gi, c4; slab_index(t, idx, row, gi, c4);
        if (row < nvalid) v[u] = *(const float4*)(base + row * row_stride + gi * grp_stride + col0 + 4 * c4);
      }
    }
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      const int idx = i0 + u * 64 + lane;
      if (idx < total) {
        int row, gi, c4; slab_index(t, idx, row, gi, c4);
        const int dd = gi * t.dgp + 4 * c4;
        float2* d2 = (float2*)(dst + row * ld + dd);
        d2[0] = make_float2(xform(dd, v[u].x), xform(dd + 1, v[u].y));
        d2[1] = make_float2(xform(dd + 2, v[u].z), xform(dd + 3, v[u].w));
      }
    }
  }
}
// Register-staged variant for software pipelining (slabs with at most 64*MAXV float4, i.e. DD <= 16*MAXV):
// slab_fetch issues the loads of a (future) tile, slab_commit writes them to the LDS tile one iteration later.
template <int MAXV>
__device__ __forceinline__ void slab_fetch(const float* __restrict__ base, long row_stride, long grp_stride, int col0, const TileDims& t,
                                           int nvalid, int lane, float4 (&v)[MAXV]) {
  const int total = 16 * (t.DD >> 2);
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int idx = u * 64 + lane;
    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < total) {
      int row, gi, c4; slab_index(t, idx, row, gi, c4);
      if (row < nvalid) v[u] = *(const float4*)(base + row * row_stride + gi * grp_stride + col0 + 4 * c4);
    }
  }
}
template <int MAXV, typename F>
__device__ __forceinline__ void slab_commit(const float4 (&v)[MAXV], const TileDims& t, float* dst, int ld, int lane, F&& xform) {
  const int total = 16 * (t.DD >> 2);
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int idx = u * 64 + lane;
    if (idx < total) {
      int row, gi, c4; slab_index(t, idx, row, gi, c4);
      const int dd = gi * t.dgp + 4 * c4;
      float2* d2 = (float2*)(dst + row * ld + dd);
      d2[0] = make_float2(xform(dd, v[u].x), xform(dd + 1, v[u].y));
      d2[1] = make_float2(xform(dd + 2, v[u].z), xform(dd + 3, v[u].w));
    }
  }
}
// fp32 slab store: value(row, dd) = src[row*ld + dd] * rowscale(row)
template <typename F>
__device__ __forceinline__ void slab_store_f32(float* __restrict__ base, long row_stride, long grp_stride, int col0, const TileDims& t,
                                               int nvalid, const float* src, int ld, int lane, F&& rowscale) {
  const int total = 16 * (t.DD >> 2);
  for (int idx = lane; idx < total; idx += 64) {
    int row, gi, c4; slab_index(t, idx, row, gi, c4);
    if (row < nvalid) {
      const int dd = gi * t.dgp + 4 * c4;
      const float sc = rowscale(row);
      const float2* s2 = (const float2*)(src + row * ld + dd);
      const float2 a = s2[0], b = s2[1];
      *(float4*)(base + row * row_stride + gi * grp_stride + col0 + 4 * c4) = make_float4(a.x * sc, a.y * sc, b.x * sc, b.y * sc);
    }
  }
}
// T-typed slab store (bf16: 8 bytes per lane)
template <typename T, typename F>
__device__ __forceinline__ void slab_store_T(void* __restrict__ base_, long elem0, long row_stride, long grp_stride, int col0,
                                             const TileDims& t, int nvalid, const float* src, int ld, int lane, F&& rowscale) {
  const int total = 16 * (t.DD >> 2);
  for (int idx = lane; idx < total; idx += 64) {
    int row, gi, c4; slab_index(t, idx, row, gi, c4);
    if (row < nvalid) {
      const int dd = gi * t.dgp + 4 * c4;
      const float sc = rowscale(row);
      const float2* s2 = (const float2*)(src + row * ld + dd);
      const float2 a = s2[0], b = s2[1];
      const long o = elem0 + row * row_stride + gi * grp_stride + col0 + 4 * c4;
      if constexpr (sizeof(T) == 4) {
        *(float4*)((float*)base_ + o) = make_float4(a.x * sc, a.y * sc, b.x * sc, b.y * sc);
      } else {
        uint2 pk;
        pk.x = (unsigned)f2bf(a.x * sc) | ((unsigned)f2bf(a.y * sc) << 16);
        pk.y = (unsigned)f2bf(b.x * sc) | ((unsigned)f2bf(b.y * sc) << 16);
        *(uint2*)((unsigned short*)base_ + o) = pk;
      }
    }
  }
}

// =====================================================================================================
// PRE_SMALL forward   (net_trans_v3.py:385-395)
// =====================================================================================================
struct PreTArgs { P16 glat; int lat_of_e[MAX_E]; int nxn_of_e[MAX_E]; long sxr_off[MAX_E]; TileDims t; int ln_before; float ln_eps; const float* ZR; const float* sxr; };

template <typename T>
__global__ void __launch_bounds__(256) kt_pre_small(PreTArgs a, float* __restrict__ Z, const float* __restrict__ L2, const float* __restrict__ sxs, const float* __restrict__ TT,
                                                    const float* __restrict__ TW, const float* __restrict__ Tsum, const float* __restrict__ wsum, const float* __restrict__ dconst,
                                                    void* __restrict__ aout_, float* __restrict__ rmu, float* __restrict__ colpart) {
  T* aout = (T*)aout_;
  const TileDims& t = a.t;
  extern __shared__ float sm[];
  const int K = t.K, DD = t.DD, K4 = 4 * t.k4;
  float* s_TT = sm;                               // [K4][ldb_k]
  float* s_TW = s_TT + K4 * t.ldb_k;              // [K4][ldb_d]
  float* s_tb = s_TW + K4 * t.ldb_d;              // [K4]
  float* s_a = s_tb + K4;                         // 4 waves x [16][lda_k]
  float* s_rv = s_a + 4 * 16 * t.lda_k;           // 4 waves x [3][16]   A-layout -> C-layout scalars
  float* s_col = s_rv + 4 * 48;                   // 4 waves x [2][DD]
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int n_beg = blockIdx.x * t.per, n_end = min(t.N, n_beg + t.per);
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  float* At = s_a + wave * 16 * t.lda_k;
  float* rv = s_rv + wave * 48;
  float* mycol = s_col + wave * 2 * DD;
  for (int i = threadIdx.x; i < 4 * 2 * DD; i += 256) s_col[i] = 0.f;
  for (int i = threadIdx.x; i < 4 * 16 * t.lda_k; i += 256) s_a[i] = 0.f;      // columns >= 4*k4 stay zero

  for (int e = 0; e < t.E; ++e) {
    const int l = a.lat_of_e[e];
    const bool nxn = a.nxn_of_e[e] != 0;
    float gv = 0.f;
    __syncthreads();
    if (nxn) gv = a.glat.p[e][0];
    if (l >= 0) {
      gv = a.glat.p[e][0];
      const float* tt = TT + ((long)s * t.El + l) * K * K;
      for (int i = threadIdx.x; i < K4 * t.ldb_k; i += 256) {
        const int k = i / t.ldb_k, c = i % t.ldb_k;
        s_TT[i] = (k < K && c < K) ? tt[k * K + c] : 0.f;
      }
      for (int i = threadIdx.x; i < K4 * t.ldb_d; i += 256) {
        const int k = i / t.ldb_d, dd = i % t.ldb_d;
        s_TW[i] = (k < K && dd < DD) ? TW[((long)s * t.KLT + (long)l * t.Kp + k) * t.DZ + colmap(t, e, dd)] : 0.f;
      }
      for (int i = threadIdx.x; i < K4; i += 256) s_tb[i] = i < K ? Tsum[(long)s * t.KLT + (long)l * t.Kp + i] / (float)t.C : 0.f;
    }
    __syncthreads();
    for (int n0 = n_beg + 16 * wave; n0 < n_end; n0 += 64) {
      const long t0 = (long)s * t.N + n0;
      float Sx[4], Sxx[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int n = n0 + 4 * q + x;
        Sx[x] = n < t.N ? sxs[t0 + 4 * q + x] : 0.f;
        Sxx[x] = n < t.N ? sxs[t.NT + t0 + 4 * q + x] : 1.f;
        if (nxn && n < t.N) {        // x' = x + g xr : sums of x' from the sums of x, xr and x . xr   (mgn.py:132-139)
          const long ti = t0 + 4 * q + x;
          const float* sxr = a.sxr + a.sxr_off[e];              // this expert's xr slot
          Sx[x] += gv * sxr[ti];
          Sxx[x] += 2.f * gv * sxr[2L * t.NT + ti] + gv * gv * sxr[(long)t.NT + ti];
        }
      }
      if (l >= 0) {
        // ---- softmax of the hop-2 logits, token r spread over the 4 lanes q (k = 4*kk + q) ----
        const bool rowok = (n0 + r) < t.N;
        const float* l2 = L2 + (t0 + r) * t.KL + (long)l * t.Kp;
        float mx = -INFINITY;
        for (int kk = 0; kk < t.k4; ++kk) { const int k = 4 * kk + q; if (rowok && k < K) mx = fmaxf(mx, l2[k]); }
        mx = qmax4(mx);
        float sum = 0.f;
        for (int kk = 0; kk < t.k4; ++kk) { const int k = 4 * kk + q; if (rowok && k < K) sum += __expf(l2[k] - mx); }
        sum = qsum4(sum);
        const float inv = rowok ? 1.f / sum : 0.f;
        float u1 = 0.f, u2 = 0.f;
        wsync();
        for (int kk = 0; kk < t.k4; ++kk) {
          const int k = 4 * kk + q;
          float av = 0.f;
          if (rowok && k < K) {
            const float lv = l2[k];
            av = rndT<T>(__expf(lv - mx) * inv);
            u1 += av * s_tb[k]; u2 += av * lv;
          }
          At[r * t.lda_k + k] = av;
          if (rowok && k < t.Kp) stT<T>(aout, (long)l * t.aL + (t0 + r) * t.Kp + k, av);
        }
        if (rowok) for (int k = K4 + q; k < t.Kp; k += 4) stT<T>(aout, (long)l * t.aL + (t0 + r) * t.Kp + k, 0.f);
        u1 = qsum4(u1); u2 = qsum4(u2);
        if (q == 0) { rv[r] = u1; rv[16 + r] = u2; }
        wsync();
        // ---- u3 = a^T (T T^T) a  through the matrix pipe ----
        float u3[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ct = 0; ct * 16 < K; ++ct) {
          const f32x4 w = tile_mm(At, t.lda_k, s_TT, t.ldb_k, t.k4, ct * 16, r, q);
#pragma unroll
          for (int x = 0; x < 4; ++x) u3[x] += w[x] * At[(4 * q + x) * t.lda_k + ct * 16 + r];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          u3[x] = rsum16(u3[x]);
          Sx[x] += gv * (float)t.C * rv[4 * q + x];
          Sxx[x] += 2.f * gv * rv[16 + 4 * q + x] + gv * gv * u3[x];
        }
      }
      float mu[4], rr[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        mu[x] = 0.f; rr[x] = 1.f;
        if (a.ln_before) {
          mu[x] = Sx[x] / (float)t.C;
          rr[x] = rsqrtf(fmaxf(Sxx[x] / (float)t.C - mu[x] * mu[x], 0.f) + a.ln_eps);
        }
      }
      // ---- LN-folded down projection, column tile by column tile ----
      for (int ct = 0; ct * 16 < DD; ++ct) {
        const int dd = ct * 16 + r;
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
        if (l >= 0) p = tile_mm(At, t.lda_k, s_TW, t.ldb_d, t.k4, ct * 16, r, q);
        float c0 = 0.f, c1 = 0.f;
        if (dd < DD) {
          const int col = colmap(t, e, dd);
          const float ws = wsum[col], dc = dconst[col];
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            if (n0 + 4 * q + x < t.N) {
              const long zi = (t0 + 4 * q + x) * t.DZ + col;
              const float zr = Z[zi] + gv * (nxn ? a.ZR[zi] : p[x]);
              const float z = a.ln_before ? rr[x] * (zr - mu[x] * ws) + dc : zr;
              Z[zi] = z;
              c0 += z; c1 += z * z;
            }
          }
        }
        c0 = qsum4(c0); c1 = qsum4(c1);
        if (q == 0 && dd < DD) { mycol[dd] += c0; mycol[DD + dd] += c1; }
      }
      if (r == 0) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          if (n0 + 4 * q + x < t.N) { rmu[(long)e * t.NT + (t0 + 4 * q + x)] = rr[x]; rmu[(long)t.NT * t.E + (long)e * t.NT + (t0 + 4 * q + x)] = mu[x]; }
      }
    }
    flush_colacc(t, s_col, 2, e, colpart, blk, 0);
  }
}


// Algorithmic bytes of the bottleneck-space passes (every element each pass has to read or write, once; zsz / esz are the
// element sizes of the Z-space and T-typed tensors) -- the numerators of their HBM rooflines.
static double bytes_pre_small(const Dims& d) { return (double)d.NT * (2.0 * d.DZ * d.zsz + (d.KL ? (double)d.KL * (4 + d.esz) : 0.0) + 8.0 + 8.0 * d.E); }
static double bytes_post_small(const Dims& d) { return (double)d.NT * ((double)d.DZ * d.zsz + (double)d.g * d.KPp * d.esz + 8.0 * d.E); }
static double bytes_post_small_bwd(const Dims& d) {
  const double zspace = d.gram64 ? (double)d.DZ * (d.esz + d.zsz) + 4.0 * d.E       // read the saved z', write dz' and dSoo
                                 : (double)d.DZ * (2.0 * d.zsz + 2.0 * d.esz);      // read Z, write dz', z', dSoo z'
  return (double)d.NT * (zspace + (double)d.g * d.KPp * 4.0 + 8.0 * d.E);
}
static double bytes_mid_bwd(const Dims& d) { return (double)d.NT * (3.0 * d.DZ * d.zsz); }
static double bytes_pre_small_bwd(const Dims& d) {
  return (double)d.NT * ((double)d.DZ * (2.0 * d.zsz + d.esz) + (d.KL ? (double)d.KL * (4 + 4 * d.esz) : 0.0) + 8.0 * d.E + 12.0);
}
int k_pre_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  const Dims& d = pl.d;
  const bool stream = tile_fast_ok(d) && kfs_serves_pre_small(d);
  ProfScope ps_(stream ? "k_pre_small (stream)" : "k_pre_small", (long)d.NT, bytes_pre_small(d), 0.0, st);
  for (int e = 0; e < d.E; ++e)
    if (d.nxn_of_e[e] && !prm.e[e].gate_lat) { set_last_error("moe: expert %d lacks gate_av", e); return ERR_BAD_ARG; }
  if (tile_fast_ok(d)) {
    const int rc = kfs_pre_small(pl, saved, scratch, prm, st);       // streaming form (large bf16 sites); 1 = not served
    return rc == 1 ? kf_pre_small(pl, saved, scratch, prm, st) : rc;
  }
  if (d.gen) return kg_pre_small(pl, saved, scratch, prm, st);
  dim3 grid; int per; tile_grid(d, &grid, &per);
  PreTArgs a;
  for (int e = 0; e < MAX_E; ++e) {
    a.glat.p[e] = prm.e[e].gate_lat; a.lat_of_e[e] = d.lat_of_e[e]; a.nxn_of_e[e] = d.nxn_of_e[e];
    a.sxr_off[e] = (e < d.E && d.xr_of_e[e] > 0) ? (long)d.xr_of_e[e] * 3 * d.NT : 0;
  }
  a.t = make_td(d, per); a.ln_before = d.ln_before; a.ln_eps = d.ln_eps;
  a.ZR = (const float*)(saved + pl.o_ZR); a.sxr = (const float*)(saved + pl.o_sxr);
  const TileDims& t = a.t;
  const int K4 = 4 * t.k4;
  size_t sh = (size_t)(K4 * t.ldb_k + K4 * t.ldb_d + K4 + 4 * 16 * t.lda_k + 4 * 48 + 4 * 2 * t.DD) * sizeof(float);
  AVMOE_TRY(set_lds(d.bf16 ? (const void*)kt_pre_small<__bf16> : (const void*)kt_pre_small<float>, sh, "pre_small", d.excl));
  DISPATCH_T(d.bf16, kt_pre_small, grid, dim3(256), sh, st, a, (float*)(saved + pl.o_Z), (const float*)(saved + pl.o_L2),
             (const float*)(saved + pl.o_sx), (const float*)(saved + pl.o_TT), (const float*)(saved + pl.o_TW),
             (const float*)(saved + pl.o_Tsum), (const float*)(saved + pl.o_wsum), (const float*)(saved + pl.o_dconst),
             (void*)(saved + pl.o_a), (float*)(saved + pl.o_rmu), (float*)(scratch + pl.o_colpart));
  AVMOE_CHECK_LAUNCH("pre_small");
  return OK;
}

// =====================================================================================================
// POST_SMALL forward  (net_trans_v3.py:430-434,485-486)
// =====================================================================================================
struct PostTArgs { P16 gate; int relu_of_e[MAX_E]; TileDims t; int ln_post, use_gate; float ln_eps; };

// z' tile of expert e: C-layout values written to the per-wave LDS tile Zt[16][lda_d] (rows = tokens)
__device__ __forceinline__ void load_zp_tile(const TileDims& t, int e, bool relu, const float* Z, const float* s_sc, const float* s_sh,
                                             long t0, int n0, float* Zt, int r, int q) {
  for (int ct = 0; ct * 16 < t.DD; ++ct) {
    const int dd = ct * 16 + r;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      float y = 0.f;
      if (dd < t.DD && n0 + 4 * q + x < t.N) {
        y = Z[(t0 + 4 * q + x) * t.DZ + colmap(t, e, dd)] * s_sc[dd] + s_sh[dd];
        if (relu) y = fmaxf(y, 0.f);
      }
      Zt[(4 * q + x) * t.lda_d + dd] = y;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) kt_post_small(PostTArgs a, const float* __restrict__ Z, const float* __restrict__ bn1, const float* __restrict__ Gq, const float* __restrict__ uvh,
                                                     const float* __restrict__ probs, void* __restrict__ Apost_, float* __restrict__ rpmup) {
  T* Apost = (T*)Apost_;
  const TileDims& t = a.t;
  extern __shared__ float sm[];
  const int DD = t.DD, dgp = t.dgp, g4 = (dgp + 3) / 4;
  float* s_G = sm;                                // [g][4*g4][ldb_g]
  float* s_us = s_G + t.g * 4 * g4 * t.ldb_g;     // DD
  float* s_vh = s_us + DD;
  float* s_sc = s_vh + DD;
  float* s_sh = s_sc + DD;
  float* s_z = s_sh + DD;                         // 4 waves x [16][lda_d]
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int n_beg = blockIdx.x * t.per, n_end = min(t.N, n_beg + t.per);
  float* Zt = s_z + wave * 16 * t.lda_d;
  for (int e = 0; e < t.E; ++e) {
    __syncthreads();
    for (int i = threadIdx.x; i < t.g * 4 * g4 * t.ldb_g; i += 256) {
      const int gi = i / (4 * g4 * t.ldb_g), rem = i % (4 * g4 * t.ldb_g), k = rem / t.ldb_g, c = rem % t.ldb_g;
      s_G[i] = (k < dgp && c < dgp) ? Gq[((long)(gi * t.E + e)) * dgp * dgp + k * dgp + c] : 0.f;
    }
    float H1 = 0.f, H2 = 0.f;
    for (int gi = 0; gi < t.g; ++gi) { H1 += uvh[2 * t.DZ + gi * t.E + e]; H2 += uvh[2 * t.DZ + t.g * t.E + gi * t.E + e]; }
    for (int dd = threadIdx.x; dd < DD; dd += 256) {
      const int col = colmap(t, e, dd);
      s_us[dd] = uvh[col]; s_vh[dd] = uvh[t.DZ + col]; s_sc[dd] = bn1[2 * t.DZ + col]; s_sh[dd] = bn1[3 * t.DZ + col];
    }
    __syncthreads();
    const bool relu = a.relu_of_e[e];
    const float gate = a.use_gate ? a.gate.p[e][0] : 1.f;
    const float qv = probs[(long)s * t.E + e] * gate;
    for (int n0 = n_beg + 16 * wave; n0 < n_end; n0 += 64) {
      const long t0 = (long)s * t.N + n0;
      wsync();
      load_zp_tile(t, e, relu, Z, s_sc, s_sh, t0, n0, Zt, r, q);
      wsync();
      float rp[4] = {1.f, 1.f, 1.f, 1.f}, mup[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.ln_post) {
        float so[4] = {0.f, 0.f, 0.f, 0.f}, soo[4] = {0.f, 0.f, 0.f, 0.f};
        for (int gi = 0; gi < t.g; ++gi)
          for (int ct = 0; ct * 16 < dgp; ++ct) {
            const f32x4 w = tile_mm(Zt + gi * dgp, t.lda_d, s_G + gi * 4 * g4 * t.ldb_g, t.ldb_g, g4, ct * 16, r, q);
            const int jp = ct * 16 + r, dd = gi * dgp + jp;
            if (jp < dgp) {
#pragma unroll
              for (int x = 0; x < 4; ++x) {
                const float zv = Zt[(4 * q + x) * t.lda_d + dd];
                so[x] += zv * s_us[dd];
                soo[x] += zv * (w[x] + 2.f * s_vh[dd]);
              }
            }
          }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float So = rsum16(so[x]) + H1, Soo = rsum16(soo[x]) + H2;
          mup[x] = So / (float)t.C;
          rp[x] = rsqrtf(fmaxf(Soo / (float)t.C - mup[x] * mup[x], 0.f) + a.ln_eps);
        }
      }
      for (int ct = 0; ct * 16 < DD; ++ct) {
        const int dd = ct * 16 + r;
        if (dd < DD) {
#pragma unroll
          for (int x = 0; x < 4; ++x)
            if (n0 + 4 * q + x < t.N)
              stT<T>(Apost, ((t0 + 4 * q + x) * t.g + dd / dgp) * t.KPp + e * dgp + (dd % dgp), qv * rp[x] * Zt[(4 * q + x) * t.lda_d + dd]);
        }
      }
      if (r < t.g) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          if (n0 + 4 * q + x < t.N) {
            const long base = ((t0 + 4 * q + x) * t.g + r) * t.KPp + t.E * dgp + 3 * e;
            stT<T>(Apost, base + 0, qv * rp[x]); stT<T>(Apost, base + 1, -qv * rp[x] * mup[x]); stT<T>(Apost, base + 2, qv);
          }
      }
      if (r == 0) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          if (n0 + 4 * q + x < t.N) { rpmup[(long)e * t.NT + (t0 + 4 * q + x)] = rp[x]; rpmup[(long)t.NT * t.E + (long)e * t.NT + (t0 + 4 * q + x)] = mup[x]; }
      }
    }
  }
}

int k_post_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  const Dims& d = pl.d;
  const bool stream = tile_fast_ok(d) && kfs_serves_post_small(d);
  ProfScope ps_(stream ? "k_post_small (stream)" : "k_post_small", (long)d.NT, bytes_post_small(d), 0.0, st);
  if (tile_fast_ok(d)) {
    const int rc = kfs_post_small(pl, saved, scratch, prm, st);       // streaming form (large bf16 sites); 1 = not served
    return rc == 1 ? kf_post_small(pl, saved, scratch, prm, st) : rc;
  }
  if (d.gen) return kg_post_small(pl, saved, scratch, prm, st);
  dim3 grid; int per; tile_grid(d, &grid, &per);
  PostTArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.gate.p[e] = prm.e[e].gate; a.relu_of_e[e] = d.relu_of_e[e]; }
  a.t = make_td(d, per); a.ln_post = d.ln_post; a.use_gate = d.use_gate && !d.gate_w; a.ln_eps = d.ln_eps;
  const TileDims& t = a.t;
  const int g4 = cdiv(t.dgp, 4);
  size_t sh = (size_t)(t.g * 4 * g4 * t.ldb_g + 4 * t.DD + 4 * 16 * t.lda_d) * sizeof(float);
  AVMOE_TRY(set_lds(d.bf16 ? (const void*)kt_post_small<__bf16> : (const void*)kt_post_small<float>, sh, "post_small", d.excl));
  DISPATCH_T(d.bf16, kt_post_small, grid, dim3(256), sh, st, a, (const float*)(saved + pl.o_Z), (const float*)(saved + pl.o_bn1),
             (const float*)(saved + pl.o_Gq), (const float*)(saved + pl.o_uvh), (const float*)(saved + pl.o_probs),
             (void*)(saved + pl.o_Apost), (float*)(saved + pl.o_rpmup));
  AVMOE_CHECK_LAUNCH("post_small");
  return OK;
}

}  // namespace avmoe

namespace avmoe {

// =====================================================================================================
// POST_SMALL backward
// =====================================================================================================
struct PostBTArgs { P16 gate; int relu_of_e[MAX_E]; TileDims t; int ln_post, use_gate; };

template <typename T, bool PF>
__global__ void __launch_bounds__(256) kt_post_small_bwd(PostBTArgs a, const float* __restrict__ Z, const float* __restrict__ bn1,
                                                         const float* __restrict__ Gq, const float* __restrict__ uvh,
                                                         const float* __restrict__ probs, const float* __restrict__ rpmup,
                                                         const float* __restrict__ dAp, float* __restrict__ dzp, void* __restrict__ Zp_,
                                                         void* __restrict__ Zw_, float* __restrict__ colpart, float* __restrict__ blkscal) {
  const TileDims& t = a.t;
  extern __shared__ float sm[];
  const int DD = t.DD, dgp = t.dgp, g4 = (dgp + 3) / 4, nw = blockDim.x >> 6, ld = t.lda_d;
  float* s_G = sm;
  float* s_us = s_G + t.g * 4 * g4 * t.ldb_g;
  float* s_vh = s_us + DD;
  float* s_sc = s_vh + DD;
  float* s_sh = s_sc + DD;
  float* s_til = s_sh + DD;                       // nw x 3 x [16][ld] : z', dAz, dz'
  float* s_row = s_til + nw * 3 * 16 * ld;        // nw x [6][16]      : da1, da2, da3, rp, mup, dSoo
  float* s_col = s_row + nw * 96;                 // nw x [2][DD]
  float* s_scal = s_col + nw * 2 * DD;            // nw x 4
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int n_beg = blockIdx.x * t.per, n_end = min(t.N, n_beg + t.per);
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  float* Zt = s_til + wave * 3 * 16 * ld;
  float* Dt = Zt + 16 * ld;
  float* Ot = Dt + 16 * ld;
  float* rw = s_row + wave * 96;
  float* mycol = s_col + wave * 2 * DD;
  for (int i = threadIdx.x; i < nw * 2 * DD; i += blockDim.x) s_col[i] = 0.f;
  for (int i = threadIdx.x; i < nw * 3 * 16 * ld; i += blockDim.x) s_til[i] = 0.f;
  for (int e = 0; e < t.E; ++e) {
    __syncthreads();
    for (int i = threadIdx.x; i < t.g * 4 * g4 * t.ldb_g; i += blockDim.x) {
      const int gi = i / (4 * g4 * t.ldb_g), rem = i % (4 * g4 * t.ldb_g), k = rem / t.ldb_g, c = rem % t.ldb_g;
      s_G[i] = (k < dgp && c < dgp) ? Gq[((long)(gi * t.E + e)) * dgp * dgp + k * dgp + c] : 0.f;
    }
    for (int dd = threadIdx.x; dd < DD; dd += blockDim.x) {
      const int col = colmap(t, e, dd);
      s_us[dd] = uvh[col]; s_vh[dd] = uvh[t.DZ + col]; s_sc[dd] = bn1[2 * t.DZ + col]; s_sh[dd] = bn1[3 * t.DZ + col];
    }
    __syncthreads();
    const bool relu = a.relu_of_e[e];
    const float gate = a.use_gate ? a.gate.p[e][0] : 1.f;
    const float qv = probs[(long)s * t.E + e] * gate;
    float sdq = 0.f, sdSo = 0.f, sdSoo = 0.f;
    // software pipeline (PF): the loads of tile i+1 are in flight while tile i is computed and stored
    float4 pz[4], pd[4];
    float pr[5] = {0.f, 0.f, 0.f, 1.f, 0.f};
    auto fetch = [&](int n0) {
      const long t0 = (long)s * t.N + n0;
      const int nvalid = min(16, t.N - n0);
      slab_fetch<4>(Z + t0 * t.DZ, t.DZ, (long)t.E * dgp, e * dgp, t, nvalid, lane, pz);
      slab_fetch<4>(dAp + t0 * t.g * t.KPp, (long)t.g * t.KPp, t.KPp, e * dgp, t, nvalid, lane, pd);
      pr[0] = pr[1] = pr[2] = 0.f; pr[3] = 1.f; pr[4] = 0.f;
      if (lane < nvalid) {
        for (int gi = 0; gi < t.g; ++gi) {
          const float* p = dAp + ((t0 + lane) * t.g + gi) * t.KPp + t.E * dgp + 3 * e;
          pr[0] += p[0]; pr[1] += p[1]; pr[2] += p[2];
        }
        pr[3] = rpmup[(long)e * t.NT + (t0 + lane)]; pr[4] = rpmup[(long)t.NT * t.E + (long)e * t.NT + (t0 + lane)];
      }
    };
    if (PF && n_beg + 16 * wave < n_end) fetch(n_beg + 16 * wave);
    for (int n0 = n_beg + 16 * wave; n0 < n_end; n0 += 16 * nw) {
      const long t0 = (long)s * t.N + n0;
      const int nvalid = min(16, t.N - n0);
      wsync();
      // ---- phase A: everything this tile needs, in coalesced 16-byte loads ----
      auto zx = [&](int dd, float z) { const float y = z * s_sc[dd] + s_sh[dd]; return relu ? fmaxf(y, 0.f) : y; };
      if constexpr (PF) {
        slab_commit<4>(pz, t, Zt, ld, lane, zx);
        slab_commit<4>(pd, t, Dt, ld, lane, [](int, float v) { return v; });
        if (lane < 16) { rw[lane] = pr[0]; rw[16 + lane] = pr[1]; rw[32 + lane] = pr[2]; rw[48 + lane] = pr[3]; rw[64 + lane] = pr[4]; }
        if (n0 + 16 * nw < n_end) fetch(n0 + 16 * nw);
      } else {
        slab_load<4>(Z + t0 * t.DZ, t.DZ, (long)t.E * dgp, e * dgp, t, nvalid, Zt, ld, lane, zx);
        slab_load<4>(dAp + t0 * t.g * t.KPp, (long)t.g * t.KPp, t.KPp, e * dgp, t, nvalid, Dt, ld, lane, [](int, float v) { return v; });
        if (lane < 16) {
          float d1 = 0.f, d2 = 0.f, d3 = 0.f, rpv = 1.f, mupv = 0.f;
          if (lane < nvalid) {
            for (int gi = 0; gi < t.g; ++gi) {
              const float* p = dAp + ((t0 + lane) * t.g + gi) * t.KPp + t.E * dgp + 3 * e;
              d1 += p[0]; d2 += p[1]; d3 += p[2];
            }
            rpv = rpmup[(long)e * t.NT + (t0 + lane)]; mupv = rpmup[(long)t.NT * t.E + (long)e * t.NT + (t0 + lane)];
          }
          rw[lane] = d1; rw[16 + lane] = d2; rw[32 + lane] = d3; rw[48 + lane] = rpv; rw[64 + lane] = mupv;
        }
      }
      wsync();
      // ---- phase B: bottleneck-space arithmetic from LDS ----
      float zz[4] = {0.f, 0.f, 0.f, 0.f};
      for (int ct = 0; ct * 16 < DD; ++ct) {
        const int dd = ct * 16 + r;
        if (dd < DD) {
#pragma unroll
          for (int x = 0; x < 4; ++x) zz[x] += Dt[(4 * q + x) * ld + dd] * Zt[(4 * q + x) * ld + dd];
        }
      }
      float rp[4], dSo[4], dSoo[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int row = 4 * q + x;
        zz[x] = rsum16(zz[x]);
        rp[x] = rw[48 + row];
        const float mup = rw[64 + row], da1 = rw[row], da2 = rw[16 + row], da3 = rw[32 + row];
        dSo[x] = 0.f; dSoo[x] = 0.f;
        if (row < nvalid) {
          const float dq = rp[x] * zz[x] + rp[x] * da1 - rp[x] * mup * da2 + da3;
          if (a.ln_post) {
            const float drp = qv * zz[x] + qv * da1 - qv * mup * da2;
            float dmup = -qv * rp[x] * da2;
            const float dvarp = drp * (-0.5f) * rp[x] * rp[x] * rp[x];
            dSoo[x] = dvarp / (float)t.C;
            dmup -= 2.f * mup * dvarp;
            dSo[x] = dmup / (float)t.C;
          }
          if (r == 0) { sdq += dq; sdSo += dSo[x]; sdSoo += dSoo[x]; rw[80 + row] = dSoo[x]; }
        } else if (r == 0) rw[80 + row] = 0.f;
      }
      for (int gi = 0; gi < t.g; ++gi)
        for (int ct = 0; ct * 16 < dgp; ++ct) {
          f32x4 w = {0.f, 0.f, 0.f, 0.f};
          if (a.ln_post) w = tile_mm(Zt + gi * dgp, ld, s_G + gi * 4 * g4 * t.ldb_g, t.ldb_g, g4, ct * 16, r, q);
          const int jp = ct * 16 + r, dd = gi * dgp + jp;
          float c0 = 0.f, c1 = 0.f;
          if (jp < dgp) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const int row = 4 * q + x;
              const float zv = Zt[row * ld + dd];
              float dz = qv * rp[x] * Dt[row * ld + dd];
              if (a.ln_post) {
                dz += dSo[x] * s_us[dd] + dSoo[x] * (2.f * w[x] + 2.f * s_vh[dd]);
                c0 += dSo[x] * zv; c1 += dSoo[x] * zv;
              }
              Ot[row * ld + dd] = dz;
            }
          }
          c0 = qsum4(c0); c1 = qsum4(c1);
          if (q == 0 && jp < dgp) { mycol[dd] += c0; mycol[DD + dd] += c1; }
        }
      wsync();
      // ---- phase C: coalesced stores ----
      slab_store_f32(dzp + t0 * t.DZ, t.DZ, (long)t.E * dgp, e * dgp, t, nvalid, Ot, ld, lane, [](int) { return 1.f; });
      slab_store_T<T>(Zp_, t0 * t.DZ, t.DZ, (long)t.E * dgp, e * dgp, t, nvalid, Zt, ld, lane, [](int) { return 1.f; });
      slab_store_T<T>(Zw_, t0 * t.DZ, t.DZ, (long)t.E * dgp, e * dgp, t, nvalid, Zt, ld, lane, [&](int row) { return rw[80 + row]; });
    }
    flush_colacc(t, s_col, 2, e, colpart, blk, 0);
    flush_scal4(s_scal, wave_sum(sdq), wave_sum(sdSo), wave_sum(sdSoo), 0.f, blkscal + ((long)blk * t.E + e) * 4, 0x7u);
  }
}

// =====================================================================================================
// MID backward
// =====================================================================================================
struct MidBTArgs { int relu_of_e[MAX_E]; TileDims t; int moments; };

template <typename T>
__global__ void __launch_bounds__(256) kt_mid_bwd(MidBTArgs a, const float* __restrict__ Z, const float* __restrict__ bn1,
                                                  const float* __restrict__ dsm, const float* __restrict__ sdSzz, float* __restrict__ dzp,
                                                  float* __restrict__ colpart) {
  const TileDims& t = a.t;
  extern __shared__ float sm[];
  const int DD = t.DD, dgp = t.dgp, g4 = (dgp + 3) / 4, nw = blockDim.x >> 6, ld = t.lda_d;
  float* s_S = sm;                                // [g][4*g4][ldb_g]
  float* s_bn = s_S + t.g * 4 * g4 * t.ldb_g;     // [5][DD]: mean, rstd, sc, sh, dmz/NT
  float* s_til = s_bn + 5 * DD;                   // nw x 3 x [16][ld] : raw z, z', dz' -> dy
  float* s_col = s_til + nw * 3 * 16 * ld;        // nw x [2][DD]
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int n_beg = blockIdx.x * t.per, n_end = min(t.N, n_beg + t.per);
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  float* Zr = s_til + wave * 3 * 16 * ld;
  float* Zt = Zr + 16 * ld;
  float* Dt = Zt + 16 * ld;
  float* mycol = s_col + wave * 2 * DD;
  for (int i = threadIdx.x; i < nw * 2 * DD; i += blockDim.x) s_col[i] = 0.f;
  for (int i = threadIdx.x; i < nw * 3 * 16 * ld; i += blockDim.x) s_til[i] = 0.f;
  for (int e = 0; e < t.E; ++e) {
    __syncthreads();
    if (a.moments)
      for (int i = threadIdx.x; i < t.g * 4 * g4 * t.ldb_g; i += blockDim.x) {
        const int gi = i / (4 * g4 * t.ldb_g), rem = i % (4 * g4 * t.ldb_g), k = rem / t.ldb_g, c = rem % t.ldb_g;
        s_S[i] = (k < dgp && c < dgp) ? sdSzz[((long)(gi * t.E + e)) * dgp * dgp + k * dgp + c] : 0.f;
      }
    for (int dd = threadIdx.x; dd < DD; dd += blockDim.x) {
      const int col = colmap(t, e, dd);
      s_bn[dd] = bn1[col]; s_bn[DD + dd] = bn1[t.DZ + col]; s_bn[2 * DD + dd] = bn1[2 * t.DZ + col];
      s_bn[3 * DD + dd] = bn1[3 * t.DZ + col]; s_bn[4 * DD + dd] = a.moments ? dsm[2 * t.DZ + col] : 0.f;
    }
    __syncthreads();
    const bool relu = a.relu_of_e[e];
    for (int n0 = n_beg + 16 * wave; n0 < n_end; n0 += 16 * nw) {
      const long t0 = (long)s * t.N + n0;
      const int nvalid = min(16, t.N - n0);
      wsync();
      slab_load<4>(Z + t0 * t.DZ, t.DZ, (long)t.E * dgp, e * dgp, t, nvalid, Zr, ld, lane, [](int, float v) { return v; });
      slab_load<4>(dzp + t0 * t.DZ, t.DZ, (long)t.E * dgp, e * dgp, t, nvalid, Dt, ld, lane, [](int, float v) { return v; });
      wsync();
      if (a.moments) {                       // z' tile (A operand of the BN2-moment term)
        for (int ct = 0; ct * 16 < DD; ++ct) {
          const int dd = ct * 16 + r;
          if (dd < DD) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const float y = Zr[(4 * q + x) * ld + dd] * s_bn[2 * DD + dd] + s_bn[3 * DD + dd];
              Zt[(4 * q + x) * ld + dd] = relu ? fmaxf(y, 0.f) : y;
            }
          }
        }
        wsync();
      }
      for (int gi = 0; gi < t.g; ++gi)
        for (int ct = 0; ct * 16 < dgp; ++ct) {
          f32x4 w = {0.f, 0.f, 0.f, 0.f};
          if (a.moments) w = tile_mm(Zt + gi * dgp, ld, s_S + gi * 4 * g4 * t.ldb_g, t.ldb_g, g4, ct * 16, r, q);
          const int jp = ct * 16 + r, dd = gi * dgp + jp;
          float c0 = 0.f, c1 = 0.f;
          if (jp < dgp) {
            const float mean = s_bn[dd], rstd = s_bn[DD + dd], sc = s_bn[2 * DD + dd], sh = s_bn[3 * DD + dd], dm = s_bn[4 * DD + dd];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const int row = 4 * q + x;
              const float z = Zr[row * ld + dd];
              const float zh = (z - mean) * rstd;
              const float y = z * sc + sh;
              const float dz = Dt[row * ld + dd] + dm + w[x];
              const float dy = (row >= nvalid || (relu && y <= 0.f)) ? 0.f : dz;
              Dt[row * ld + dd] = dy;
              c0 += dy; c1 += dy * zh;
            }
          }
          c0 = qsum4(c0); c1 = qsum4(c1);
          if (q == 0 && jp < dgp) { mycol[dd] += c0; mycol[DD + dd] += c1; }
        }
      wsync();
      slab_store_f32(dzp + t0 * t.DZ, t.DZ, (long)t.E * dgp, e * dgp, t, nvalid, Dt, ld, lane, [](int) { return 1.f; });
    }
    flush_colacc(t, s_col, 2, e, colpart, blk, 2);
  }
}

// =====================================================================================================
// PRE_SMALL backward
// =====================================================================================================
struct PreBTArgs { P16 glat; int lat_of_e[MAX_E]; int nxn_of_e[MAX_E]; int first_of_slot[MAX_E]; long sxr_off[MAX_E]; TileDims t; int ln_before, use_bn, bn_train, dd4;
                   const float* ZR; const float* sxr; void* dZR; float* dsr; };

template <typename T>
__global__ void __launch_bounds__(256) kt_pre_small_bwd(PreBTArgs a, const float* __restrict__ Z, const float* __restrict__ L2, const float* __restrict__ TT, const float* __restrict__ TW,
                                                        const float* __restrict__ Tsum, const float* __restrict__ wsum, const float* __restrict__ dconst, const void* __restrict__ ain_,
                                                        const float* __restrict__ rmu, const float* __restrict__ bn1, const float* __restrict__ dsm, const float* __restrict__ dy_in,
                                                        void* __restrict__ dZx_, void* __restrict__ dL2x_, void* __restrict__ aw_, void* __restrict__ ag_, float* __restrict__ dsxs, float* __restrict__ rs2x,
                                                        float* __restrict__ colpart, float* __restrict__ blkscal, float* __restrict__ dtbp) {
  const T* ain = (const T*)ain_;
  T* dZx = (T*)dZx_; T* dL2x = (T*)dL2x_; T* aw_o = (T*)aw_; T* ag_o = (T*)ag_;
  const TileDims& t = a.t;
  extern __shared__ float sm[];
  const int K = t.K, DD = t.DD, K4 = 4 * t.k4, D4 = 4 * a.dd4, nw = blockDim.x >> 6;
  float* s_TT = sm;                               // [K4][ldb_k]
  float* s_TW = s_TT + K4 * t.ldb_k;              // [K4][ldb_d]   B of a . TW       (contraction k)
  float* s_TWt = s_TW + K4 * t.ldb_d;             // [D4][ldb_k]   B of dzraw . TW^T (contraction dd)
  float* s_tb = s_TWt + D4 * t.ldb_k;             // [K4]
  float* s_bn = s_tb + K4;                        // [7][DD]: mean, rstd, sc, mdy, mdyz, wsum, dconst
  float* s_a = s_bn + 7 * DD;                     // nw x [16][lda_k]   a tile
  float* s_da = s_a + nw * 16 * t.lda_k;          // nw x [16][lda_k]   da tile
  float* s_dz = s_da + nw * 16 * t.lda_k;         // nw x [16][lda_d]   dzraw tile
  float* s_rv = s_dz + nw * 16 * t.lda_d;         // nw x 32
  float* s_col = s_rv + nw * 32;                  // nw x [2][DD]
  float* s_kcol = s_col + nw * 2 * DD;            // nw x [K4]
  float* s_scal = s_kcol + nw * K4;               // nw x 4
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int n_beg = blockIdx.x * t.per, n_end = min(t.N, n_beg + t.per);
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  float* At = s_a + wave * 16 * t.lda_k;
  float* Da = s_da + wave * 16 * t.lda_k;
  float* Dt = s_dz + wave * 16 * t.lda_d;
  float* rv = s_rv + wave * 32;
  float* mycol = s_col + wave * 2 * DD;
  float* mykcol = s_kcol + wave * K4;
  for (int i = threadIdx.x; i < nw * 2 * DD; i += blockDim.x) s_col[i] = 0.f;
  for (int i = threadIdx.x; i < nw * K4; i += blockDim.x) s_kcol[i] = 0.f;
  for (int i = threadIdx.x; i < 2 * nw * 16 * t.lda_k; i += blockDim.x) s_a[i] = 0.f;       // a and da tiles
  for (int i = threadIdx.x; i < nw * 16 * t.lda_d; i += blockDim.x) s_dz[i] = 0.f;

  for (int e = 0; e < t.E; ++e) {
    const int l = a.lat_of_e[e];
    const bool nxn = a.nxn_of_e[e] != 0;
    float gv = 0.f;
    __syncthreads();
    if (nxn) gv = a.glat.p[e][0];
    if (l >= 0) {
      gv = a.glat.p[e][0];
      const float* tt = TT + ((long)s * t.El + l) * K * K;
      for (int i = threadIdx.x; i < K4 * t.ldb_k; i += blockDim.x) {
        const int k = i / t.ldb_k, c = i % t.ldb_k;
        s_TT[i] = (k < K && c < K) ? tt[k * K + c] : 0.f;
      }
      for (int i = threadIdx.x; i < K4 * t.ldb_d; i += blockDim.x) {
        const int k = i / t.ldb_d, dd = i % t.ldb_d;
        s_TW[i] = (k < K && dd < DD) ? TW[((long)s * t.KLT + (long)l * t.Kp + k) * t.DZ + colmap(t, e, dd)] : 0.f;
      }
      for (int i = threadIdx.x; i < D4 * t.ldb_k; i += blockDim.x) {
        const int dd = i / t.ldb_k, k = i % t.ldb_k;
        s_TWt[i] = (k < K && dd < DD) ? TW[((long)s * t.KLT + (long)l * t.Kp + k) * t.DZ + colmap(t, e, dd)] : 0.f;
      }
      for (int i = threadIdx.x; i < K4; i += blockDim.x) s_tb[i] = i < K ? Tsum[(long)s * t.KLT + (long)l * t.Kp + i] / (float)t.C : 0.f;
    }
    for (int dd = threadIdx.x; dd < DD; dd += blockDim.x) {
      const int col = colmap(t, e, dd);
      s_bn[dd] = bn1[col]; s_bn[DD + dd] = bn1[t.DZ + col]; s_bn[2 * DD + dd] = bn1[2 * t.DZ + col];
      s_bn[3 * DD + dd] = a.bn_train ? dsm[3 * t.DZ + col] : 0.f; s_bn[4 * DD + dd] = a.bn_train ? dsm[4 * t.DZ + col] : 0.f;
      s_bn[5 * DD + dd] = wsum[col]; s_bn[6 * DD + dd] = dconst[col];
    }
    __syncthreads();
    float sdg = 0.f;
    for (int n0 = n_beg + 16 * wave; n0 < n_end; n0 += 16 * nw) {
      const long t0 = (long)s * t.N + n0;
      bool ok[4];
      float rr[4], mu[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        ok[x] = n0 + 4 * q + x < t.N;
        rr[x] = (a.ln_before && ok[x]) ? rmu[(long)e * t.NT + (t0 + 4 * q + x)] : 1.f;
        mu[x] = (a.ln_before && ok[x]) ? rmu[(long)t.NT * t.E + (long)e * t.NT + (t0 + 4 * q + x)] : 0.f;
      }
      // ---- BN1 input gradient, folded-LayerNorm sums, dzraw tile ----
      float s_dr[4] = {0.f, 0.f, 0.f, 0.f}, s_dmu[4] = {0.f, 0.f, 0.f, 0.f}, s_zr[4] = {0.f, 0.f, 0.f, 0.f};
      wsync();
      for (int ct = 0; ct * 16 < DD; ++ct) {
        const int dd = ct * 16 + r;
        float c0 = 0.f, c1 = 0.f;
        if (dd < DD) {
          const int col = colmap(t, e, dd);
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            float dzr = 0.f;
            if (ok[x]) {
              const long zi = (t0 + 4 * q + x) * t.DZ + col;
              const float z = Z[zi];
              float dz = dy_in[zi];
              if (a.use_bn) {
                if (a.bn_train) dz = s_bn[2 * DD + dd] * (dz - s_bn[3 * DD + dd] - (z - s_bn[dd]) * s_bn[DD + dd] * s_bn[4 * DD + dd]);
                else dz = s_bn[2 * DD + dd] * dz;
              }
              if (a.ln_before) {
                const float zc = (z - s_bn[6 * DD + dd]) / rr[x];
                c0 += dz; c1 += -rr[x] * mu[x] * dz;
                s_dr[x] += dz * zc; s_dmu[x] += dz * s_bn[5 * DD + dd];
                dzr = rr[x] * dz;
              } else dzr = dz;
              stT<T>(dZx, zi, dzr);
              if (nxn) { stT<T>((T*)a.dZR, zi, gv * dzr); s_zr[x] += dzr * a.ZR[zi]; }
            }
            Dt[(4 * q + x) * t.lda_d + dd] = dzr;
          }
        }
        c0 = qsum4(c0); c1 = qsum4(c1);
        if (q == 0 && dd < DD) { mycol[dd] += c0; mycol[DD + dd] += c1; }
      }
      float dSx[4], dSxx[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        dSx[x] = 0.f; dSxx[x] = 0.f;
        if (a.ln_before) {
          const float sdr = rsum16(s_dr[x]), sdm = rsum16(s_dmu[x]);
          float dmu = -rr[x] * sdm;
          const float dvar = sdr * (-0.5f) * rr[x] * rr[x] * rr[x];
          dSxx[x] = dvar / (float)t.C;
          dmu -= 2.f * mu[x] * dvar;
          dSx[x] = dmu / (float)t.C;
        }
        if (r == 0 && ok[x]) {
          const long ti = t0 + 4 * q + x;
          if (e == 0) { dsxs[ti] = dSx[x]; dsxs[t.NT + ti] = dSxx[x]; }
          else { dsxs[ti] += dSx[x]; dsxs[t.NT + ti] += dSxx[x]; }
        }
        if (nxn) {                   // x' = x + g xr : statistics gradients to (sum xr, sum xr^2, x . xr) and to the gate
          const float zr = rsum16(s_zr[x]);
          if (r == 0 && ok[x]) {
            const long ti = t0 + 4 * q + x;
            const float v0 = gv * dSx[x], v1 = 2.f * gv * gv * dSxx[x], v2 = 2.f * gv * dSxx[x];
            float* dsr = a.dsr + a.sxr_off[e];                  // this expert's xr slot (shared by the AVVP experts)
            const float* sxr = a.sxr + a.sxr_off[e];
            if (a.first_of_slot[e]) { dsr[ti] = v0; dsr[(long)t.NT + ti] = v1; dsr[2L * t.NT + ti] = v2; }
            else { dsr[ti] += v0; dsr[(long)t.NT + ti] += v1; dsr[2L * t.NT + ti] += v2; }
            sdg += dSx[x] * sxr[ti] + dSxx[x] * (2.f * sxr[2L * t.NT + ti] + 2.f * gv * sxr[(long)t.NT + ti]) + zr;
          }
        }
      }
      if (l >= 0) {
        // ---- a tile (A operand + C layout source), u1, u2 ----
        const bool rowok = (n0 + r) < t.N;
        const long arow = (t0 + r) * t.KL + (long)l * t.Kp, aplane = (long)l * t.aL + (t0 + r) * t.Kp;      // row of L2 ; plane row of a
        float u1 = 0.f, u2 = 0.f;
        for (int kk = 0; kk < t.k4; ++kk) {
          const int k = 4 * kk + q;
          float av = 0.f;
          if (rowok && k < K) { av = ldT<T>(ain, aplane + k); u1 += av * s_tb[k]; u2 += av * L2[arow + k]; }
          At[r * t.lda_k + k] = av;
        }
        u1 = qsum4(u1); u2 = qsum4(u2);
        if (q == 0) { rv[r] = u1; rv[16 + r] = u2; }
        wsync();
        float u3[4] = {0.f, 0.f, 0.f, 0.f}, dgr[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ct = 0; ct * 16 < DD; ++ct) {                       // dzraw . (a TW)
          const f32x4 p = tile_mm(At, t.lda_k, s_TW, t.ldb_d, t.k4, ct * 16, r, q);
#pragma unroll
          for (int x = 0; x < 4; ++x) dgr[x] += p[x] * Dt[(4 * q + x) * t.lda_d + ct * 16 + r];
        }
        float du1[4], du2[4], du3[4], sada[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int x = 0; x < 4; ++x) { du1[x] = dSx[x] * gv * (float)t.C; du2[x] = 2.f * gv * dSxx[x]; du3[x] = gv * gv * dSxx[x]; }
        for (int ct = 0; ct * 16 < K; ++ct) {                        // da tile
          const f32x4 ta = tile_mm(At, t.lda_k, s_TT, t.ldb_k, t.k4, ct * 16, r, q);
          const f32x4 twd = tile_mm(Dt, t.lda_d, s_TWt, t.ldb_k, a.dd4, ct * 16, r, q);
          const int k = ct * 16 + r;
          float ck = 0.f;
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float ac = At[(4 * q + x) * t.lda_k + k];
            u3[x] += ta[x] * ac;
            float da = 0.f;
            if (k < K && ok[x]) {
              da = gv * twd[x] + du1[x] * s_tb[k] + du2[x] * L2[(t0 + 4 * q + x) * t.KL + (long)l * t.Kp + k] + 2.f * du3[x] * ta[x];
              sada[x] += ac * da;
              ck += du1[x] * ac;
            }
            Da[(4 * q + x) * t.lda_k + k] = da;
          }
          ck = qsum4(ck);
          if (q == 0 && k < K) mykcol[k] += ck;
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          sada[x] = rsum16(sada[x]);
          u3[x] = rsum16(u3[x]); dgr[x] = rsum16(dgr[x]);
          if (r == 0 && ok[x]) sdg += dSx[x] * (float)t.C * rv[4 * q + x] + dSxx[x] * (2.f * rv[16 + 4 * q + x] + 2.f * gv * u3[x]) + dgr[x];
        }
        wsync();
        for (int ct = 0; ct * 16 < t.Kp; ++ct) {
          const int k = ct * 16 + r;
          if (k < t.Kp) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
              if (ok[x]) {
                const long o = (t0 + 4 * q + x) * t.KLp + (long)l * t.Kp + k, oa = (long)l * t.aL + (t0 + 4 * q + x) * t.Kp + k;
                float v0 = 0.f, v1 = 0.f, v2 = 0.f;
                if (k < K) {
                  const float ac = At[(4 * q + x) * t.lda_k + k], da = Da[(4 * q + x) * t.lda_k + k];
                  v0 = du2[x] * ac + ac * (da - sada[x]); v1 = du3[x] * ac; v2 = gv * ac;
                }
                stT<T>(dL2x, o, v0); stT<T>(aw_o, oa, v1); stT<T>(ag_o, oa, v2);
              }
          }
        }
      }
      if (e == t.E - 1 && r == 0) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          if (ok[x]) {
            const long ti = t0 + 4 * q + x;
            stT<T>(dL2x, ti * t.KLp + t.KL, dsxs[ti]);
            stT<T>(dL2x, ti * t.KLp + t.KL + 1, 1.f);
            rs2x[ti] = 2.f * dsxs[t.NT + ti];
          }
      }
    }
    flush_colacc(t, s_col, 2, e, colpart, blk, 0);
    flush_scal4(s_scal, 0.f, 0.f, 0.f, wave_sum(sdg), blkscal + ((long)blk * t.E + e) * 4, 0x8u);
    if (l >= 0) {
      __syncthreads();
      for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float v = 0.f;
        for (int w = 0; w < nw; ++w) { v += s_kcol[w * K4 + k]; s_kcol[w * K4 + k] = 0.f; }
        dtbp[(long)blk * t.KL + (long)l * t.Kp + k] = v;
      }
      __syncthreads();
    }
  }
}

}  // namespace avmoe

namespace avmoe {

// waves per block (1 .. 4): the count that puts the most waves on a CU within its 160 KB of LDS -- e.g. 3 waves x 2 blocks
// rather than 4 waves x 1 block when a 4-wave block needs more than half of it (bottlenecks above 64)
static int pick_waves(size_t fixed_floats, size_t per_wave_floats, size_t* bytes) {
  static const bool old_rule = dev_env("AVMOE_PICK_WAVES_POW2") != nullptr;      // dev switch
  int best = 0, best_occ = 0;
  for (int nw = 4; nw >= 1; --nw) {
    if (old_rule && nw == 3) continue;
    const size_t b = (fixed_floats + nw * per_wave_floats) * sizeof(float);
    if (b > 160 * 1024) continue;
    if (old_rule) { best = nw; break; }
    const int occ = nw * (int)std::min<size_t>(8, (160 * 1024) / b);
    if (occ > best_occ) { best_occ = occ; best = nw; }
  }
  if (best) *bytes = (fixed_floats + best * per_wave_floats) * sizeof(float);
  return best;
}

int k_post_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads, hipStream_t st, int dap16) {
  const Dims& d = pl.d;
  const bool stream = tile_fast_ok(d) && kfs_serves_post_small_bwd(d, dap16);
  ProfScope ps_(stream ? "k_post_small_bwd (stream)" : "k_post_small_bwd", (long)d.NT, bytes_post_small_bwd(d) - (dap16 ? (double)d.NT * d.g * (d.E * d.dgp * 2.0 + (d.KPp - 16) * 4.0 - d.E * d.dgp * 2.0) : 0.0), 0.0, st);
  if (dap16 && !tile_fast_ok(d) && !d.gen) { set_last_error("post_small_bwd: split dApost needs a register-resident path"); return ERR_BAD_ARG; }
  if (tile_fast_ok(d)) {
    const int rc = kfs_post_small_bwd(pl, saved, scratch, prm, st, dap16);       // streaming form (large bf16 sites); 1 = not served
    if (rc < 0) return rc;
    if (rc == 1) AVMOE_TRY(kf_post_small_bwd(pl, saved, scratch, prm, st, dap16));
    return k_post_small_bwd_finalize(pl, saved, scratch, prm, grads, st);
  }
  if (d.gen) {
    AVMOE_TRY(kg_post_small_bwd(pl, saved, scratch, prm, st, dap16));
    return k_post_small_bwd_finalize(pl, saved, scratch, prm, grads, st);
  }
  dim3 grid; int per; tile_grid(d, &grid, &per);
  PostBTArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.gate.p[e] = prm.e[e].gate; a.relu_of_e[e] = d.relu_of_e[e]; }
  a.t = make_td(d, per); a.ln_post = d.ln_post; a.use_gate = d.use_gate && !d.gate_w;
  const TileDims& t = a.t;
  const int g4 = cdiv(t.dgp, 4);
  size_t sh;
  const int nw = pick_waves((size_t)t.g * 4 * g4 * t.ldb_g + 4 * t.DD, (size_t)3 * 16 * t.lda_d + 96 + 2 * t.DD + 4, &sh);
  if (!nw) { set_last_error("post_small_bwd: LDS budget"); return ERR_UNSUPPORTED; }
  const bool pf = d.DD <= 64;
#define KPSB(TT_, PF_) kt_post_small_bwd<TT_, PF_>
  const void* fn = d.bf16 ? (pf ? (const void*)KPSB(__bf16, true) : (const void*)KPSB(__bf16, false))
                          : (pf ? (const void*)KPSB(float, true) : (const void*)KPSB(float, false));
  AVMOE_TRY(set_lds(fn, sh, "post_small_bwd", d.excl));
#define LAUNCH_PSB(TT_, PF_) hipLaunchKernelGGL((KPSB(TT_, PF_)), grid, dim3(64 * nw), sh, st, a, (const float*)(saved + pl.o_Z), (const float*)(saved + pl.o_bn1), \
                     (const float*)(saved + pl.o_Gq), (const float*)(saved + pl.o_uvh), (const float*)(saved + pl.o_probs), \
                     (const float*)(saved + pl.o_rpmup), (const float*)(scratch + pl.o_dAp), (float*)(scratch + pl.o_dzp), \
                     (void*)(scratch + pl.o_Zp), (void*)(scratch + pl.o_Zw), (float*)(scratch + pl.o_colpart), (float*)(scratch + pl.o_blkscal))
  if (d.bf16) { if (pf) LAUNCH_PSB(__bf16, true); else LAUNCH_PSB(__bf16, false); }
  else { if (pf) LAUNCH_PSB(float, true); else LAUNCH_PSB(float, false); }
#undef LAUNCH_PSB
#undef KPSB
  AVMOE_CHECK_LAUNCH("post_small_bwd");
  return k_post_small_bwd_finalize(pl, saved, scratch, prm, grads, st);
}

int k_mid_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads, hipStream_t st) {
  const Dims& d = pl.d;
  const bool stream = tile_fast_ok(d) && kfs_serves_mid_bwd(d);
  ProfScope ps_(stream ? "k_mid_bwd (stream)" : "k_mid_bwd", (long)d.NT, bytes_mid_bwd(d), 0.0, st);
  if (tile_fast_ok(d)) {
    const int rc = kfs_mid_bwd(pl, saved, scratch, st);       // streaming form (large bf16 sites); 1 = not served
    if (rc < 0) return rc;
    if (rc == 1) AVMOE_TRY(kf_mid_bwd(pl, saved, scratch, st));
    return k_mid_bwd_finalize(pl, saved, scratch, prm, grads, st);
  }
  if (d.gen) {
    AVMOE_TRY(kg_mid_bwd(pl, saved, scratch, st));
    return k_mid_bwd_finalize(pl, saved, scratch, prm, grads, st);
  }
  dim3 grid; int per; tile_grid(d, &grid, &per);
  MidBTArgs a;
  for (int e = 0; e < MAX_E; ++e) a.relu_of_e[e] = d.relu_of_e[e];
  a.t = make_td(d, per); a.moments = d.use_bn && d.training;
  const TileDims& t = a.t;
  const int g4 = cdiv(t.dgp, 4);
  size_t sh;
  const int nw = pick_waves((size_t)t.g * 4 * g4 * t.ldb_g + 5 * t.DD, (size_t)3 * 16 * t.lda_d + 2 * t.DD, &sh);
  if (!nw) { set_last_error("mid_bwd: LDS budget"); return ERR_UNSUPPORTED; }
  AVMOE_TRY(set_lds(d.bf16 ? (const void*)kt_mid_bwd<__bf16> : (const void*)kt_mid_bwd<float>, sh, "mid_bwd", d.excl));
  DISPATCH_T(d.bf16, kt_mid_bwd, grid, dim3(64 * nw), sh, st, a, (const float*)(saved + pl.o_Z), (const float*)(saved + pl.o_bn1),
             (const float*)(scratch + pl.o_dsm), (const float*)(scratch + pl.o_sdSzz), (float*)(scratch + pl.o_dzp),
             (float*)(scratch + pl.o_colpart));
  AVMOE_CHECK_LAUNCH("mid_bwd");
  return k_mid_bwd_finalize(pl, saved, scratch, prm, grads, st);
}

int k_pre_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads, hipStream_t st) {
  const Dims& d = pl.d;
  if (tile_fast_ok(d) && kfs_serves_pre_bwd(d)) {      // large bf16 sites: both parts in one streaming pass (tile_stream.hip)
    {
      ProfScope ps_("k_pre_bwd (stream)", (long)d.NT, (double)d.NT * ((double)d.DZ * (2.0 * d.zsz + d.esz) + 8.0 * d.E + 4.0 + (d.KL ? (double)d.KL * (4 + 4 * d.esz) + d.KLp * d.esz - d.KL * d.esz : 0.0)), 0.0, st);
      AVMOE_TRY(kfs_pre_bwd(pl, saved, scratch, prm, st));
    }
    ProfScope ps_("k_pre_bwd_finalize", 0.0, 0.0, st);
    return k_pre_small_bwd_finalize(pl, saved, scratch, prm, grads, st);
  }
  if (tile_fast_ok(d)) {      // two kernels: every expert's BN1 / LayerNorm part, then the cross-modal experts' hop-2 block (tile_fast.hip)
    const double lat = (double)d.NT * ((double)d.El * d.DD * d.esz + (d.KL ? (double)d.KL * (4 + 4 * d.esz) : 0.0) + 8.0 * d.El);
    {
      ProfScope ps_("k_pre_small_bwd", (long)d.NT, (double)d.NT * ((double)d.DZ * (2.0 * d.zsz + d.esz) + 8.0 * d.E + 8.0 * d.El + 12.0), 0.0, st);
      AVMOE_TRY(kf_pre_small_bwd(pl, saved, scratch, prm, st));
    }
    if (d.KL) {
      ProfScope ps_("k_pre_lat_bwd", (long)d.NT, lat, 0.0, st);
      AVMOE_TRY(kf_pre_lat_bwd(pl, saved, scratch, prm, st));
    }
    ProfScope ps_("k_pre_bwd_finalize", 0.0, 0.0, st);
    return k_pre_small_bwd_finalize(pl, saved, scratch, prm, grads, st);
  }
  ProfScope ps_("k_pre_small_bwd", (long)d.NT, bytes_pre_small_bwd(d), 0.0, st);
  if (d.gen) {
    AVMOE_TRY(kg_pre_small_bwd(pl, saved, scratch, prm, st));
    return k_pre_small_bwd_finalize(pl, saved, scratch, prm, grads, st);
  }
  dim3 grid; int per; tile_grid(d, &grid, &per);
  PreBTArgs a;
  {
    bool seen[MAX_E] = {};
    for (int e = 0; e < MAX_E; ++e) {
      a.glat.p[e] = prm.e[e].gate_lat; a.lat_of_e[e] = d.lat_of_e[e]; a.nxn_of_e[e] = d.nxn_of_e[e];
      a.first_of_slot[e] = 0; a.sxr_off[e] = 0;
      if (e < d.E && d.nxn_of_e[e]) {
        const int slot = d.xr_of_e[e];
        a.sxr_off[e] = (long)slot * 3 * d.NT;
        a.first_of_slot[e] = !seen[slot]; seen[slot] = true;
      }
    }
  }
  a.ZR = (const float*)(saved + pl.o_ZR); a.sxr = (const float*)(saved + pl.o_sxr);
  a.dZR = (void*)(scratch + pl.o_dZR); a.dsr = (float*)(scratch + pl.o_dsr);
  a.t = make_td(d, per); a.ln_before = d.ln_before; a.use_bn = d.use_bn; a.bn_train = d.use_bn && d.training; a.dd4 = cdiv(d.DD, 4);
  const TileDims& t = a.t;
  const int K4 = 4 * t.k4, D4 = 4 * a.dd4;
  size_t sh;
  const int nw = pick_waves((size_t)K4 * t.ldb_k + (size_t)K4 * t.ldb_d + (size_t)D4 * t.ldb_k + K4 + 7 * t.DD,
                            (size_t)2 * 16 * t.lda_k + 16 * t.lda_d + 32 + 2 * t.DD + K4 + 4, &sh);
  if (!nw) { set_last_error("pre_small_bwd: K=%d, bottleneck %d exceed the LDS budget", d.K, d.DD); return ERR_UNSUPPORTED; }
  AVMOE_TRY(set_lds(d.bf16 ? (const void*)kt_pre_small_bwd<__bf16> : (const void*)kt_pre_small_bwd<float>, sh, "pre_small_bwd", d.excl));
  DISPATCH_T(d.bf16, kt_pre_small_bwd, grid, dim3(64 * nw), sh, st, a, (const float*)(saved + pl.o_Z), (const float*)(saved + pl.o_L2),
             (const float*)(saved + pl.o_TT), (const float*)(saved + pl.o_TW), (const float*)(saved + pl.o_Tsum),
             (const float*)(saved + pl.o_wsum), (const float*)(saved + pl.o_dconst), (const void*)(saved + pl.o_a),
             (const float*)(saved + pl.o_rmu), (const float*)(saved + pl.o_bn1), (const float*)(scratch + pl.o_dsm),
             (const float*)(scratch + pl.o_dzp), (void*)(scratch + pl.o_Zw), (void*)(scratch + pl.o_dL2x), (void*)(scratch + pl.o_aw),
             (void*)(scratch + pl.o_ag), (float*)(scratch + pl.o_dsxs), (float*)(scratch + pl.o_rs2x),
             (float*)(scratch + pl.o_colpart), (float*)(scratch + pl.o_blkscal), (float*)(scratch + pl.o_dtbp));
  AVMOE_CHECK_LAUNCH("pre_small_bwd");
  return k_pre_small_bwd_finalize(pl, saved, scratch, prm, grads, st);
}

}  // namespace avmoe
