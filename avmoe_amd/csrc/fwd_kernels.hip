// Forward bottleneck-space kernels (everything between the GEMMs).  Arithmetic and names follow
// oracle/algebra_ref.py::AlgebraRef.forward; reference lines are cited there and in DESIGN.md.
//
// Buffers of the operand type T (float | __bf16) are passed as void* and cast inside the kernel.
#include "kernels.h"
#include "colsum_fin.h"
#include "moe_run.h"
#include "device_utils.h"
#include "prof.h"
#include "gemm.h"
#include <algorithm>

namespace avmoe {

#define DISPATCH_T(bf16, KERN, grid, block, shmem, st, ...)                                   \
  do {                                                                                        \
    if (bf16) hipLaunchKernelGGL((KERN<__bf16>), grid, block, shmem, st, __VA_ARGS__);        \
    else hipLaunchKernelGGL((KERN<float>), grid, block, shmem, st, __VA_ARGS__);              \
  } while (0)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T> __device__ __forceinline__ float roundT(float v);
template <> __device__ __forceinline__ float roundT<float>(float v) { return v; }
template <> __device__ __forceinline__ float roundT<__bf16>(float v) { return bf2f(f2bf(v)); }

static inline unsigned grid1d(long n, int cap = 4096) { return (unsigned)std::max<long>(1, std::min<long>((n + 255) / 256, cap)); }

// ---------------------------------------------------------------------------------------------
// generic helpers
// ---------------------------------------------------------------------------------------------
__global__ void kk_fill_f32(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}
int k_fill_f32(float* p, long n, float v, hipStream_t st) {
  ProfScope ps_("k_fill_f32", 0.0, 0.0, st);
  if (n <= 0) return OK;
  hipLaunchKernelGGL(kk_fill_f32, dim3(grid1d(n)), dim3(256), 0, st, p, n, v);
  AVMOE_CHECK_LAUNCH("fill_f32");
  return OK;
}

// Deterministic column sums  out[slot][c] = scale * sum_r in[slot*slot_in + r*row_stride + c]   (no float atomics anywhere).
// A block owns CW columns; its threads form NS = blockDim / CW row streams (stream k adds rows k, k + NS, ..., four
// independent loads in flight), combined in double through LDS in a fixed order.  Many rows: CW = 16, 64 streams per block
// (short serial chains, ncol/16 blocks); few rows: CW = 64, 4 streams.
template <int CW, int NTHR>
__global__ void __launch_bounds__(NTHR) kk_colsum_f32(const float* in, long R, int ncol, long row_stride, long slot_in, float* out,
                                                      long slot_out, float scale) {
  constexpr int NS = NTHR / CW;
  __shared__ double red[NS][CW];
  const int c = threadIdx.x % CW, k = threadIdx.x / CW;
  const int col = blockIdx.x * CW + c;
  const float* p = in + (long)blockIdx.y * slot_in + col;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < ncol) {
    long r = k;
    for (; r + 3L * NS < R; r += 4L * NS) {
      a0 += p[r * row_stride]; a1 += p[(r + NS) * row_stride]; a2 += p[(r + 2L * NS) * row_stride]; a3 += p[(r + 3L * NS) * row_stride];
    }
    for (; r < R; r += NS) a0 += p[r * row_stride];
  }
  red[k][c] = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
  __syncthreads();
  if (k == 0 && col < ncol) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NS; ++w) s += red[w][c];
    out[(long)blockIdx.y * slot_out + col] = (float)(s * scale);
  }
}
int k_colsum_f32(const float* in, long R, int ncol, long row_stride, int nslot, long slot_in, float* out, long slot_out, float scale,
                 hipStream_t st) {
  if (ncol <= 0 || nslot <= 0) return OK;
  if (R >= 128) hipLaunchKernelGGL((kk_colsum_f32<16, 1024>), dim3(cdiv(ncol, 16), nslot), dim3(1024), 0, st, in, R, ncol, row_stride, slot_in, out, slot_out, scale);
  else hipLaunchKernelGGL((kk_colsum_f32<64, 256>), dim3(cdiv(ncol, 64), nslot), dim3(256), 0, st, in, R, ncol, row_stride, slot_in, out, slot_out, scale);
  AVMOE_CHECK_LAUNCH("colsum_f32");
  return OK;
}
// Two independent column sums of the same row-count class in ONE launch (blockIdx.y picks the job; each job runs exactly the code and
// the summation order of kk_colsum_f32 with one slot, so the results are bit-identical to two launches).
struct ColsumJob { const float* in; long R; int ncol; long row_stride; float* out; float scale; };
template <int CW, int NTHR>
__global__ void __launch_bounds__(NTHR) kk_colsum2_f32(ColsumJob j0, ColsumJob j1) {
  constexpr int NS = NTHR / CW;
  __shared__ double red[NS][CW];
  const ColsumJob j = blockIdx.y == 0 ? j0 : j1;
  const int c = threadIdx.x % CW, k = threadIdx.x / CW;
  const int col = blockIdx.x * CW + c;
  if ((int)blockIdx.x * CW >= j.ncol) return;                 // (whole block past this job's columns)
  const float* p = j.in + col;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < j.ncol) {
    long r = k;
    for (; r + 3L * NS < j.R; r += 4L * NS) {
      a0 += p[r * j.row_stride]; a1 += p[(r + NS) * j.row_stride]; a2 += p[(r + 2L * NS) * j.row_stride]; a3 += p[(r + 3L * NS) * j.row_stride];
    }
    for (; r < j.R; r += NS) a0 += p[r * j.row_stride];
  }
  red[k][c] = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
  __syncthreads();
  if (k == 0 && col < j.ncol) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NS; ++w) s += red[w][c];
    j.out[col] = (float)(s * j.scale);
  }
}
int k_colsum2_f32(const float* in0, long R0, int ncol0, long rs0, float* out0, float scale0,
                  const float* in1, long R1, int ncol1, long rs1, float* out1, float scale1, hipStream_t st) {
  if (ncol0 <= 0 || ncol1 <= 0 || (R0 >= 128) != (R1 >= 128)) {       // different classes (or an empty job): two launches
    AVMOE_TRY(k_colsum_f32(in0, R0, ncol0, rs0, 1, 0, out0, 0, scale0, st));
    return k_colsum_f32(in1, R1, ncol1, rs1, 1, 0, out1, 0, scale1, st);
  }
  const ColsumJob j0{in0, R0, ncol0, rs0, out0, scale0}, j1{in1, R1, ncol1, rs1, out1, scale1};
  const int nc = std::max(ncol0, ncol1);
  if (R0 >= 128) hipLaunchKernelGGL((kk_colsum2_f32<16, 1024>), dim3(cdiv(nc, 16), 2), dim3(1024), 0, st, j0, j1);
  else hipLaunchKernelGGL((kk_colsum2_f32<64, 256>), dim3(cdiv(nc, 64), 2), dim3(256), 0, st, j0, j1);
  AVMOE_CHECK_LAUNCH("colsum2_f32");
  return OK;
}
int k_reduce_colpart(const Plan& pl, char* scratch, int slot0, int nslots, hipStream_t st) {
  const Dims& d = pl.d;
  return k_colsum_f32((const float*)(scratch + pl.o_colpart) + (long)slot0 * d.DZ, d.nblk_tok, d.DZ, 4L * d.DZ, nslots, d.DZ,
                      (float*)(scratch + pl.o_colsum) + (long)slot0 * d.DZ, d.DZ, 1.f, st);
}

template <typename T>
__global__ void kk_cast(const float* src, long rows, int cols, long ld_src, void* dst_, long ld_dst) {
  T* dst = (T*)dst_;
  const long total = rows * ld_dst;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / ld_dst;
    const int c = (int)(i % ld_dst);
    stT<T>(dst, i, c < cols ? src[r * ld_src + c] : 0.f);
  }
}
int k_cast(int bf16_out, const float* src, long rows, int cols, long ld_src, void* dst, long ld_dst, hipStream_t st) {
  ProfScope ps_("k_cast", 0.0, 0.0, st);
  const long total = rows * ld_dst;
  if (total <= 0) return OK;
  DISPATCH_T(bf16_out, kk_cast, dim3(grid1d(total)), dim3(256), 0, st, src, rows, cols, ld_src, dst, ld_dst);
  AVMOE_CHECK_LAUNCH("cast");
  return OK;
}

// ---------------------------------------------------------------------------------------------
// weight preparation -- ONE launch, every block picks its job from its index (the jobs are independent):
//   remap operands  WcK / WcT / WfT in T, rw = Wf 1, wbar = mean_n Wc, mean(bc)     (conv_adapter / fc: net_trans_v3.py:445-446,469-470)
//   experts         Wt = Wd * gamma_before (LayerNorm folded into the down projection: net_trans_v3.py:392-395), wsum, dconst,
//                   the stacked latent tokens in T
//   constants       the ones row and the (forward: zero) dm1 / N row of every frame's Text, with their row sums
// ---------------------------------------------------------------------------------------------
struct PrepExpArgs {
  P16 down, lnbw, lnbb, tok;
  int e_of_lat[MAX_E];
  int E, g, dg, dgp, Cg, C, K, Kp, KL, ln_before;
};
struct PrepAllArgs {
  const float *Wc, *bc, *Wf;
  int N, M, Mk, Mb, Np, C, Cy, S, KL, KLT;
  int b_rw, b_wbar, b_scal, b_exp, b_const, b_end;      // first block of every job after the casts (which own [0, b_rw))
  PrepExpArgs x;
};

template <typename T>
__device__ __forceinline__ void prep_remap_cast(const PrepAllArgs& a, T* WcK, T* WcT, T* WfT, int bx, int nbx) {
  const long n1 = (long)a.N * a.Mk, n2 = (long)(a.M + 1) * a.Np, n3 = (long)a.C * a.Cy;
  for (long i = (long)bx * 256 + threadIdx.x; i < n1 + n2 + n3; i += (long)nbx * 256) {
    if (i < n1) {                                   // WcK[n] = [Wc[n,:] | bc[n] | 1 | 0..]
      const int n = (int)(i / a.Mk), m = (int)(i % a.Mk);
      stT<T>(WcK, i, m < a.M ? a.Wc[(long)n * a.M + m] : (m == a.M ? a.bc[n] : (m == a.M + 1 ? 1.f : 0.f)));
    } else if (i < n1 + n2) {                       // WcT[m] = Wc[:,m]^T ; row M = bc
      const long j = i - n1;
      const int m = (int)(j / a.Np), n = (int)(j % a.Np);
      stT<T>(WcT, j, n < a.N ? (m < a.M ? a.Wc[(long)n * a.M + m] : a.bc[n]) : 0.f);
    } else {
      const long j = i - n1 - n2;
      stT<T>(WfT, j, a.Wf[j]);
    }
  }
}
// rw[c] = sum_y T(Wf[c][y]) (T-rounded weights, so the folded bias matches the GEMM operands); one wave per row
template <typename T>
__device__ __forceinline__ void prep_rw(const PrepAllArgs& a, float* rw, int bx) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = bx * 4 + wave;
  if (c >= a.C) return;
  float s = 0.f;
  const float* row = a.Wf + (long)c * a.Cy;
  for (int y0 = lane; y0 < a.Cy; y0 += 64 * 8) {            // eight loads requested before the first add (same order of additions)
    float v[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) v[x] = row[min(y0 + 64 * x, a.Cy - 1)];
#pragma unroll
    for (int x = 0; x < 8; ++x)
      if (y0 + 64 * x < a.Cy) s += roundT<T>(v[x]);
  }
  s = wave_sum(s);
  if (lane == 0) rw[c] = s;
}
// wbar[m] = mean_n Wc[n][m], zero in the padding m >= M: a block owns 16 columns x 16 row streams (stream k adds rows k, k + 16, ..,
// eight independent loads in flight: at N = 4096 a stream is 256 rows long and the walk is latency-bound), combined in double in a fixed
// order (no float atomics: bit-reproducible)
__device__ __forceinline__ void prep_wbar(const PrepAllArgs& a, float* wbar, int bx) {
  __shared__ double red[16][16];
  const int c = threadIdx.x & 15, k = threadIdx.x >> 4;
  const int m = bx * 16 + c;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  if (m < a.M) {
    const float* p = a.Wc + m;
    int n = k;
    for (; n + 31 * 16 < a.N; n += 32 * 16) {                 // long columns (N in the thousands): 32 loads requested before the first add
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = p[(long)(n + 16 * u) * a.M];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc[u & 7] += v[u];        // (accumulator u & 7 takes rows n + 16 u: the same partition and order as below)
    }
    for (; n + 7 * 16 < a.N; n += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += p[(long)(n + 16 * u) * a.M];
    }
    for (; n < a.N; n += 16) acc[0] += p[(long)n * a.M];
  }
  red[k][c] = (((double)acc[0] + (double)acc[1]) + ((double)acc[2] + (double)acc[3])) + (((double)acc[4] + (double)acc[5]) + ((double)acc[6] + (double)acc[7]));
  __syncthreads();
  if (k == 0 && m < a.Mb) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][c];
    wbar[m] = m < a.M ? (float)(t / a.N) : 0.f;
  }
}
// scal[0] = mean(bc) ; scal[1] = 1 (the unit gate of the "v1" experts, moe_forward.cpp::with_unit_gates)
__device__ __forceinline__ void prep_scal(const PrepAllArgs& a, float* scal) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int n = threadIdx.x; n < a.N; n += 256) acc += a.bc[n];
  acc = block_sum256(acc, red);
  if (threadIdx.x == 0) { scal[0] = acc / (float)a.N; scal[1] = 1.f; }
}
// one block per row (i, e, jp) of Wt, then one per stacked latent-token row
template <typename T>
__device__ __forceinline__ void prep_experts(const PrepExpArgs& a, T* Wt, float* wsum, float* dconst, T* T0T, int bx) {
  __shared__ float red[4];
  const int nrow = a.g * a.E * a.dgp;
  if (bx < nrow) {
    const int row = bx;
    const int i = row / (a.E * a.dgp), e = (row / a.dgp) % a.E, jp = row % a.dgp;
    const float* Wd = a.down.p[e];
    float s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x; c < a.Cg; c += 256) {
      const float w = jp < a.dg ? Wd[(long)(i * a.dg + jp) * a.Cg + c] : 0.f;
      const float gm = a.ln_before ? a.lnbw.p[e][i * a.Cg + c] : 1.f;
      const float bt = a.ln_before ? a.lnbb.p[e][i * a.Cg + c] : 0.f;
      const float wt = roundT<T>(w * gm);
      stT<T>(Wt, (long)row * a.Cg + c, wt);
      s1 += wt;
      s2 += w * bt;
    }
    s1 = block_sum256(s1, red);
    s2 = block_sum256(s2, red);
    if (threadIdx.x == 0) { wsum[row] = s1; dconst[row] = s2; }
  } else {
    const int r = bx - nrow;
    if (r < a.KL) {
      const int l = r / a.Kp, k = r % a.Kp;
      const float* tk = a.tok.p[a.e_of_lat[l]];
      for (int c = threadIdx.x; c < a.C; c += 256) stT<T>(T0T, (long)r * a.C + c, k < a.K ? tk[(long)k * a.C + c] : 0.f);
    }
  }
}
// Text[s][KL][:] = 1 (ones row), Text[s][KL+1][:] = 0 (dm1 / N row, written by the backward) and their row sums in Tsum
template <typename T>
__device__ __forceinline__ void prep_const_rows(const PrepAllArgs& a, T* Text, float* Tsum, int bx, int nbx) {
  const long n3 = (long)a.S * 2 * a.C, rows = (long)a.S * a.KLT;
  for (long j = (long)bx * 256 + threadIdx.x; j < n3; j += (long)nbx * 256) {
    const int s = (int)(j / (2 * a.C)), rr = (int)((j / a.C) % 2), c = (int)(j % a.C);
    stT<T>(Text, ((long)s * a.KLT + a.KL + rr) * a.C + c, rr == 0 ? 1.f : 0.f);
    if (c == 0) {
      const long row = (long)s * a.KLT + a.KL + rr;
      Tsum[row] = rr == 0 ? (float)a.C : 0.f; Tsum[rows + row] = rr == 0 ? (float)a.C : 0.f;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) kk_prep_all(PrepAllArgs a, void* WcK, void* WcT, void* WfT, float* rw, float* wbar, float* scal,
                                                   void* Wt, float* wsum, float* dconst, void* T0T, void* Text, float* Tsum) {
  const int b = blockIdx.x;
  if (b < a.b_rw) prep_remap_cast<T>(a, (T*)WcK, (T*)WcT, (T*)WfT, b, a.b_rw);
  else if (b < a.b_wbar) prep_rw<T>(a, rw, b - a.b_rw);
  else if (b < a.b_scal) prep_wbar(a, wbar, b - a.b_wbar);
  else if (b < a.b_exp) prep_scal(a, scal);
  else if (b < a.b_const) prep_experts<T>(a.x, (T*)Wt, wsum, dconst, (T*)T0T, b - a.b_exp);
  else prep_const_rows<T>(a, (T*)Text, Tsum, b - a.b_const, a.b_end - a.b_const);
}

int k_prep_all(const Plan& pl, char* saved, const avmoe_moe_ptrs& prm, hipStream_t st) {
  ProfScope ps_("k_prep_all", 0.0, 0.0, st);
  const Dims& d = pl.d;
  if (!prm.conv_w || !prm.conv_b || !prm.fc_w) { set_last_error("moe: conv_adapter / fc parameters missing"); return ERR_BAD_ARG; }
  PrepAllArgs a;
  a.Wc = prm.conv_w; a.bc = prm.conv_b; a.Wf = prm.fc_w;
  a.N = d.N; a.M = d.M; a.Mk = d.Mk; a.Mb = d.Mb; a.Np = d.Np; a.C = d.C; a.Cy = d.Cy; a.S = d.S; a.KL = d.KL; a.KLT = d.KLT;
  for (int e = 0; e < MAX_E; ++e) {
    a.x.down.p[e] = prm.e[e].down_w; a.x.lnbw.p[e] = prm.e[e].lnb_w; a.x.lnbb.p[e] = prm.e[e].lnb_b;
    a.x.tok.p[e] = prm.e[e].my_tokens; a.x.e_of_lat[e] = d.e_of_lat[e];
  }
  for (int e = 0; e < d.E; ++e) {
    const avmoe_expert_ptrs& x = prm.e[e];
    if (!x.down_w || !x.up_w) { set_last_error("moe: expert %d has no down/up weights", e); return ERR_BAD_ARG; }
    if (d.ln_before && (!x.lnb_w || !x.lnb_b)) { set_last_error("moe: expert %d lacks ln_before", e); return ERR_BAD_ARG; }
    if (d.ln_post && (!x.lnp_w || !x.lnp_b)) { set_last_error("moe: expert %d lacks ln_post", e); return ERR_BAD_ARG; }
    if (d.use_gate && !x.gate) { set_last_error("moe: expert %d lacks gate", e); return ERR_BAD_ARG; }
    if (d.use_bn && (!x.bn1_w || !x.bn1_b || !x.bn2_w || !x.bn2_b || !x.bn1_rm || !x.bn1_rv || !x.bn2_rm || !x.bn2_rv)) {
      set_last_error("moe: expert %d lacks BatchNorm parameters / running statistics", e); return ERR_BAD_ARG;
    }
    if (d.lat_of_e[e] >= 0 && (!x.my_tokens || !x.gate_lat)) { set_last_error("moe: expert %d lacks my_tokens / gate_av", e); return ERR_BAD_ARG; }
  }
  a.x.E = d.E; a.x.g = d.g; a.x.dg = d.dg; a.x.dgp = d.dgp; a.x.Cg = d.Cg; a.x.C = d.C; a.x.K = d.K; a.x.Kp = d.Kp; a.x.KL = d.KL;
  a.x.ln_before = d.ln_before;
  const long tot = (long)d.N * d.Mk + (long)(d.M + 1) * d.Np + (long)d.C * d.Cy;
  a.b_rw = (int)grid1d(tot, 2048);
  a.b_wbar = a.b_rw + cdiv(d.C, 4);
  a.b_scal = a.b_wbar + cdiv(d.Mb, 16);
  a.b_exp = a.b_scal + 1;
  a.b_const = a.b_exp + d.g * d.E * d.dgp + d.KL;
  a.b_end = a.b_const + (int)grid1d((long)d.S * 2 * d.C, 256);
  DISPATCH_T(d.bf16, kk_prep_all, dim3((unsigned)a.b_end), dim3(256), 0, st, a, (void*)(saved + pl.o_WcK), (void*)(saved + pl.o_WcT),
             (void*)(saved + pl.o_WfT), (float*)(saved + pl.o_rw), (float*)(saved + pl.o_wbar), (float*)(saved + pl.o_scal),
             (void*)(saved + pl.o_Wt), (float*)(saved + pl.o_wsum), (float*)(saved + pl.o_dconst), (void*)(saved + pl.o_T0T),
             (void*)(saved + pl.o_Text), (float*)(saved + pl.o_Tsum));
  AVMOE_CHECK_LAUNCH("prep_all");
  return OK;
}

// ---------------------------------------------------------------------------------------------
// token statistics: row sum / sum of squares (one wave per row, 16-byte loads), column means
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) kk_rowstats(const void* X_, long rows, int C, float* out) {
  const T* X = (const T*)X_;
  constexpr int EPV = 16 / sizeof(T);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const u32x4_t* p = (const u32x4_t*)(X + row * C);
    float s = 0.f, ss = 0.f;
    for (int v = lane; v < C / EPV; v += 64) {
      const u32x4_t w = p[v];
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        // NB: __builtin_bit_cast(float, w[e]) on a vector-element lvalue miscompiles (ROCm 7.2): go through a scalar
        for (int e = 0; e < 4; ++e) { const unsigned int bits = w[e]; const float x = __uint_as_float(bits); s += x; ss += x * x; }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = bf2f((unsigned short)(w[e] & 0xFFFFu)), x1 = bf2f((unsigned short)(w[e] >> 16));
          s += x0 + x1; ss += x0 * x0 + x1 * x1;
        }
      }
    }
    s = wave_sum(s); ss = wave_sum(ss);
    if (lane == 0) { out[row] = s; out[rows + row] = ss; }
  }
}
int k_rowstats(int bf16, const void* X, long rows, int C, float* out, hipStream_t st) {
  ProfScope ps_("k_rowstats", 0.0, 0.0, st);
  if (rows <= 0) return OK;
  DISPATCH_T(bf16, kk_rowstats, dim3((unsigned)std::min<long>((rows + 3) / 4, 8192)), dim3(256), 0, st, X, rows, C, out);
  AVMOE_CHECK_LAUNCH("rowstats");
  return OK;
}

// One pass over X: per-token sum / sum of squares (LayerNorm) AND per-block column partial sums (router mean).
// grid (nchunk, S); 16 lanes per row (a wave reads 4 rows at once, every lane 16-byte vectors lane, lane+16, ..), two row
// quartets in flight per wave; row reduction = 4 shuffles inside the 16-lane group.  xpart[s][chunk][C].
template <typename T>
__global__ void __launch_bounds__(256) kk_xstats(const void* X_, int N, int C, int rows_per_blk, float* sx, long NT, float* xpart) {
  const T* X = (const T*)X_;
  constexpr int EPV = 16 / sizeof(T);
  constexpr int NV = 96 / EPV;                          // C <= 16 * NV * EPV = 1536
  __shared__ float s_col[4][1536];
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l = lane & 15, grp = lane >> 4;
  const int n0 = blockIdx.x * rows_per_blk, n1 = min(N, n0 + rows_per_blk);
  const int nvec = C / EPV;
  float cacc[NV][EPV];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int e = 0; e < EPV; ++e) cacc[v][e] = 0.f;
  for (int nb = n0 + 16 * wave; nb < n1; nb += 32 * 2) {
    // two row quartets per iteration (rows nb + grp and nb + 4 + grp .. interleaved over the 4 waves in steps of 16)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int n = nb + 4 * h + grp;
      const bool ok = n < n1;
      const long row = (long)s * N + n;
      const uint4* p = (const uint4*)(X + row * C);
      float rs = 0.f, rss = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int iv = l + 16 * v;
        if (ok && iv < nvec) {
          const uint4 w = p[iv];
          const unsigned int ww[4] = {w.x, w.y, w.z, w.w};
          if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float x = __uint_as_float(ww[e]); rs += x; rss += x * x; cacc[v][e] += x; }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x0 = bf2f((unsigned short)(ww[e] & 0xFFFFu)), x1 = bf2f((unsigned short)(ww[e] >> 16));
              rs += x0 + x1; rss += x0 * x0 + x1 * x1; cacc[v][2 * e] += x0; cacc[v][2 * e + 1] += x1;
            }
          }
        }
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { rs += __shfl_xor(rs, o, 64); rss += __shfl_xor(rss, o, 64); }
      if (l == 0 && ok) { sx[row] = rs; sx[NT + row] = rss; }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int iv = l + 16 * v;
    if (iv < nvec)
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        float c = cacc[v][e];
        c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);        // the wave's 4 row groups
        if (grp == 0) s_col[wave][iv * EPV + e] = c;
      }
  }
  __syncthreads();
  float* out = xpart + ((long)s * gridDim.x + blockIdx.x) * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    out[c] = s_col[0][c] + s_col[1][c] + s_col[2][c] + s_col[3][c];
  }
}
int k_xstats(const Plan& pl, const void* X, char* saved, char* scratch, hipStream_t st) {
  const Dims& d = pl.d;
  ProfScope ps_("k_xstats", (long)d.NT, (double)d.NT * ((double)d.C * d.esz + 8.0), 0.0, st);
  if (d.C > 1536) { set_last_error("xstats: C=%d too wide", d.C); return ERR_UNSUPPORTED; }
  const int rpb = (int)round_up(cdiv(d.N, d.xchunks), 64);      // a block sweeps 64 rows per step (4 waves x 4 quartets x 4 rows)
  const int nchunk = cdiv(d.N, rpb);                            // <= d.xchunks (xpart is sized by that)
  DISPATCH_T(d.bf16, kk_xstats, dim3(nchunk, d.S), dim3(256), 0, st, X, d.N, d.C, rpb, (float*)(saved + pl.o_sx), (long)d.NT,
             (float*)(scratch + pl.o_xpart));
  AVMOE_CHECK_LAUNCH("xstats");
  return k_colsum_f32((const float*)(scratch + pl.o_xpart), nchunk, d.C, d.C, d.S, (long)nchunk * d.C, (float*)(saved + pl.o_rin),
                      2L * d.C, 1.f / (float)d.N, st);
}

// out[i] = sum_p parts[p * n + i]  (the per-group partial row sums of the fused statistics, in group order)
__global__ void __launch_bounds__(256) kk_sum_parts(const float* __restrict__ parts, int nparts, long n, float* __restrict__ out) {
  if (n % 4 == 0) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
      float4 a = *(const float4*)(parts + i);
      for (int p = 1; p < nparts; ++p) { const float4 b = *(const float4*)(parts + (long)p * n + i); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
      *(float4*)(out + i) = a;
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      float a = parts[i];
      for (int p = 1; p < nparts; ++p) a += parts[(long)p * n + i];
      out[i] = a;
    }
  }
}
// The two finishing sums of the fused X statistics in ONE launch (they are independent): blocks [0, nb_sum) add the per-group row sums
// (kk_sum_parts), the others the per-tile column sums of every frame into the router's token mean  rin[s][c] = scale * sum_tile xpart
__global__ void __launch_bounds__(256) kk_xstats_fin(const float* __restrict__ parts, int nparts, long n, float* __restrict__ out, int nb_sum,
                                                     const float* __restrict__ xpart, int tiles, int C, int S, float* __restrict__ rin, long rin_ld,
                                                     float scale) {
  if ((int)blockIdx.x < nb_sum) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)nb_sum * 1024) {
      if (i + 3 < n) {
        float4 a = *(const float4*)(parts + i);
        for (int p = 1; p < nparts; ++p) { const float4 b = *(const float4*)(parts + (long)p * n + i); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        *(float4*)(out + i) = a;
      } else {
        for (long k = i; k < n; ++k) { float a = parts[k]; for (int p = 1; p < nparts; ++p) a += parts[(long)p * n + k]; out[k] = a; }
      }
    }
    return;
  }
  // a block owns 64 columns of one frame x 4 tile streams (stream u adds tiles u, u + 4, .., two loads in flight), combined through LDS
  // in a fixed order
  __shared__ float red[4][64];
  const int ncb = (C + 63) / 64;
  const int b = (int)blockIdx.x - nb_sum, s = b / ncb, cb = b - s * ncb;
  const int l = threadIdx.x & 63, u = threadIdx.x >> 6;
  const int c = cb * 64 + l;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    const float* p = xpart + (long)s * tiles * C + c;
    int tl = u;
    for (; tl + 4 < tiles; tl += 8) { a0 += p[(long)tl * C]; a1 += p[(long)(tl + 4) * C]; }
    if (tl < tiles) a0 += p[(long)tl * C];
  }
  red[u][l] = a0 + a1;
  __syncthreads();
  if (u == 0 && c < C) rin[(long)s * rin_ld + c] = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) * scale;
}
int k_xstats_fin(const float* parts, int nparts, long n, float* out, const float* xpart, int tiles, int C, int S, float* rin, long rin_ld,
                 float scale, hipStream_t st) {
  ProfScope ps_("k_xstats_fin", 0.0, 0.0, st);
  if (n % 4) { set_last_error("xstats_fin: row count not a multiple of 4"); return ERR_UNSUPPORTED; }
  const int nb_sum = (int)std::min<long>(cdiv(n, 1024), 2048), nb_col = S * cdiv(C, 64);
  hipLaunchKernelGGL(kk_xstats_fin, dim3((unsigned)(nb_sum + nb_col)), dim3(256), 0, st, parts, nparts, n, out, nb_sum, xpart, tiles, C, S, rin,
                     rin_ld, scale);
  AVMOE_CHECK_LAUNCH("xstats_fin");
  return OK;
}
int k_sum_parts(const float* parts, int nparts, long n, float* out, hipStream_t st) {
  ProfScope ps_("k_sum_parts", 0.0, 0.0, st);
  hipLaunchKernelGGL(kk_sum_parts, dim3((unsigned)std::min<long>(cdiv(n, 1024), 2048)), dim3(256), 0, st, parts, nparts, n, out);
  AVMOE_CHECK_LAUNCH("sum_parts");
  return OK;
}

template <typename T>
__global__ void __launch_bounds__(256) kk_colmean(const void* X_, int N, int C, float* out, long out_ld) {
  const T* X = (const T*)X_;
  const int s = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const T* p = X + (long)s * N * C + c;
  float acc = 0.f;
  for (int n = 0; n < N; ++n) acc += ldT<T>(p, (long)n * C);
  out[(long)s * out_ld + c] = acc / (float)N;
}
int k_colmean(int bf16, const void* X, int S, int N, int C, float* out, long out_ld, hipStream_t st) {
  ProfScope ps_("k_colmean", 0.0, 0.0, st);
  DISPATCH_T(bf16, kk_colmean, dim3(cdiv(C, 256), S), dim3(256), 0, st, X, N, C, out, out_ld);
  AVMOE_CHECK_LAUNCH("colmean");
  return OK;
}

// ---------------------------------------------------------------------------------------------
// hop 1 helpers
// ---------------------------------------------------------------------------------------------
// One launch after Q = T0 Wf:
//   blocks [0, nb_q): qr[kc] = T0[kc] . rw ; qb[kc] = T0[kc] . bf  (one wave per latent row), written into qrqb AND into the
//                     extension columns of every frame's Rext row:  Rext[s][kc][M] = qr[kc], [M+1] = qb[kc], rest of the padding 0
//   the others:       Rext[s][Kcy][:] = 0 (the ybar row has no logits) ; BmX[s][Kcy][:] = wbar
template <typename T>
__global__ void __launch_bounds__(256) kk_qrqb_fill(const void* T0T_, const float* rw, const float* bf, float* qrqb, void* Rext_, void* BmX_,
                                                    const float* wbar, int nb_q, int S, int Kcy, int Kcyb, int C, int M, int Mk, int Mb) {
  const T* T0T = (const T*)T0T_;
  T* Rext = (T*)Rext_; T* BmX = (T*)BmX_;
  if ((int)blockIdx.x < nb_q) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kc = blockIdx.x * 4 + wave;
    if (kc >= Kcy) return;
    float a = 0.f, b = 0.f;
    for (int c0 = lane; c0 < C; c0 += 64 * 8) {              // eight channels requested before the first add (same order of additions)
      float t[8], r[8], f[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) { const int c = min(c0 + 64 * x, C - 1); t[x] = ldT<T>(T0T, (long)kc * C + c); r[x] = rw[c]; f[x] = bf[c]; }
#pragma unroll
      for (int x = 0; x < 8; ++x)
        if (c0 + 64 * x < C) { a += t[x] * r[x]; b += t[x] * f[x]; }
    }
    a = wave_sum(a); b = wave_sum(b);
    if (lane == 0) { qrqb[kc] = a; qrqb[Kcy + kc] = b; }
    const int padw = Mk - M;
    for (int i = lane; i < S * padw; i += 64) {
      const int s = i / padw, w = i - s * padw;
      stT<T>(Rext, ((long)s * Kcyb + kc) * Mk + M + w, w == 0 ? a : (w == 1 ? b : 0.f));
    }
    return;
  }
  const long n1 = (long)S * Mk, n2 = (long)S * Mb;
  for (long i = (long)(blockIdx.x - nb_q) * 256 + threadIdx.x; i < n1 + n2; i += (long)(gridDim.x - nb_q) * 256) {
    if (i < n1) {
      const int s = (int)(i / Mk), m = (int)(i % Mk);
      stT<T>(Rext, ((long)s * Kcyb + Kcy) * Mk + m, 0.f);
    } else {
      const long j = i - n1;
      const int s = (int)(j / Mb), m = (int)(j % Mb);
      stT<T>(BmX, ((long)s * Kcyb + Kcy) * Mb + m, wbar[m]);
    }
  }
}
int k_qrqb_fill(const Plan& pl, char* saved, const float* bf, hipStream_t st) {
  ProfScope ps_("k_qrqb_fill", 0.0, 0.0, st);
  const Dims& d = pl.d;
  const int nb_q = d.Kcy > 0 ? cdiv(d.Kcy, 4) : 0;
  if (nb_q > 0 && !bf) { set_last_error("moe: fc.bias missing"); return ERR_BAD_ARG; }
  const int nb_f = (int)grid1d((long)d.S * (d.Mk + d.Mb), 1024);
  DISPATCH_T(d.bf16, kk_qrqb_fill, dim3((unsigned)(nb_q + nb_f)), dim3(256), 0, st, (const void*)(saved + pl.o_T0T),
             (const float*)(saved + pl.o_rw), bf, (float*)(saved + pl.o_qrqb), (void*)(saved + pl.o_Rext), (void*)(saved + pl.o_BmX),
             (const float*)(saved + pl.o_wbar), nb_q, d.S, d.Kcy, d.Kcyb, d.C, d.M, d.Mk, d.Mb);
  AVMOE_CHECK_LAUNCH("qrqb_fill");
  return OK;
}

// row softmax (unscaled logits, net_trans_v3.py:381): f32 in -> T out, padding columns zeroed.  Rows come
// in groups of `grp` of which the first `valid` are real; the others are written as zeros.
template <typename T>
__global__ void __launch_bounds__(256) kk_softmax_rows(const float* in, long rows, int n, int ld_in, void* out_, int ld_out,
                                                       int grp, int valid, int slot, int kvalid) {
  T* out = (T*)out_;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    if ((int)(row % grp) >= valid || (int)((row % grp) % slot) >= kvalid) {
      for (int j = lane; j < ld_out; j += 64) stT<T>(out, row * ld_out + j, 0.f);
      continue;
    }
    const float* p = in + row * ld_in;
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, p[j]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int j = lane; j < n; j += 64) sm += __expf(p[j] - mx);
    sm = wave_sum(sm);
    const float inv = 1.f / sm;
    for (int j = lane; j < ld_out; j += 64) stT<T>(out, row * ld_out + j, j < n ? __expf(p[j] - mx) * inv : 0.f);
  }
}
// the same with the row in registers (NV 16-byte vectors per lane: rows of up to 256 NV entries, leading dimensions multiples of 4):
// ONE pass over the logits, one exp per entry, 16-byte loads and 8 / 16-byte stores
template <typename T, int NV>
__global__ void __launch_bounds__(256) kk_softmax_rows_reg(const float* in, long rows, int n, int ld_in, void* out_, int ld_out,
                                                           int grp, int valid, int slot, int kvalid) {
  T* out = (T*)out_;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    T* o = out + row * ld_out;
    if ((int)(row % grp) >= valid || (int)((row % grp) % slot) >= kvalid) {
      for (int j = 4 * lane; j < ld_out; j += 256) st4T<T>(o, j, make_float4(0.f, 0.f, 0.f, 0.f));
      continue;
    }
    const float* p = in + row * ld_in;
    float4 v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int j = 4 * lane + 256 * k;
      // (a whole vector lies inside the padded row: ld_in % 4 == 0.  The loads are UNCONDITIONAL -- clamped offset, entries beyond n replaced
      //  afterwards -- so that all NV of a lane are in flight at once: under `if (j < ld_in)` each was its own branch and full wait.)
      v[k] = *(const float4*)(p + (j < ld_in ? j : 0));
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int j = 4 * lane + 256 * k;
      if (j + 3 >= n) { float* e = (float*)&v[k]; for (int x = 0; x < 4; ++x) if (j + x >= n) e[x] = -INFINITY; }
      mx = fmaxf(mx, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
    }
    mx = wave_max(mx);
    double smd = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k].x = __expf(v[k].x - mx); v[k].y = __expf(v[k].y - mx); v[k].z = __expf(v[k].z - mx); v[k].w = __expf(v[k].w - mx);   // exp(-inf) = 0: the padding
      smd += ((double)v[k].x + v[k].y) + ((double)v[k].z + v[k].w);
    }
    const float inv = (float)(1.0 / wave_sum_d(smd));
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int j = 4 * lane + 256 * k;
      if (j < ld_out) st4T<T>(o, j, make_float4(v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv));
    }
  }
}
int k_softmax_rows(int bf16_out, const float* in, long rows, int n, int ld_in, void* out, int ld_out, int grp, int valid,
                   int slot, int kvalid, hipStream_t st) {
  ProfScope ps_("k_softmax_rows", 0.0, 0.0, st);
  if (rows <= 0) return OK;
  const dim3 grid((unsigned)std::min<long>((rows + 3) / 4, 8192));
  const bool reg = ld_in % 4 == 0 && ld_out % 4 == 0 && ld_out <= ld_in && n <= 4096;
  if (reg && n <= 1024) {
    if (bf16_out) hipLaunchKernelGGL((kk_softmax_rows_reg<__bf16, 4>), grid, dim3(256), 0, st, in, rows, n, ld_in, out, ld_out, grp, valid, slot, kvalid);
    else hipLaunchKernelGGL((kk_softmax_rows_reg<float, 4>), grid, dim3(256), 0, st, in, rows, n, ld_in, out, ld_out, grp, valid, slot, kvalid);
  } else if (reg) {
    if (bf16_out) hipLaunchKernelGGL((kk_softmax_rows_reg<__bf16, 16>), grid, dim3(256), 0, st, in, rows, n, ld_in, out, ld_out, grp, valid, slot, kvalid);
    else hipLaunchKernelGGL((kk_softmax_rows_reg<float, 16>), grid, dim3(256), 0, st, in, rows, n, ld_in, out, ld_out, grp, valid, slot, kvalid);
  } else {
    DISPATCH_T(bf16_out, kk_softmax_rows, grid, dim3(256), 0, st, in, rows, n, ld_in, out, ld_out, grp, valid, slot, kvalid);
  }
  AVMOE_CHECK_LAUNCH("softmax_rows");
  return OK;
}

// Text[s][row] = T0 + TV + ab (x) rw + bf   (cross-modal slots)  |  T0 + TV  (latent-on-X slots)
// and the router's second mean  rin[s][C + c] = TV[s][Kcy] + mean(bc) rw + bf
struct FinishTArgs {
  P16 tok; int e_of_lat[MAX_E];
  int S, C, K, Kp, KL, KLT, Kcy, Kcyb, Kcx, Mb, M, src, lat0;
};
// one WAVE per (frame, latent row): 16-byte accesses where C allows, the two row sums folded inside the wave (no block barriers --
// with one short block per row the kernel was bound by block turnover, not by its bytes)
template <typename T>
__global__ void __launch_bounds__(256) kk_finish_T(FinishTArgs a, const float* TV, const void* BmX_, const float* rw, const float* bf,
                                                   const float* scal, void* Text_, float* rin, float* Tsum, int nrow) {
  const T* BmX = (const T*)BmX_;
  T* Text = (T*)Text_;
  const int lane = threadIdx.x & 63;
  const int gr = blockIdx.x * 4 + (threadIdx.x >> 6);        // (frame, latent row)
  if (gr >= nrow) return;
  const int rows = a.src == 0 ? a.Kcyb : a.Kcx;
  const int s = gr / rows, kr = gr - s * rows;
  const float* tvr = TV + (long)gr * a.C;
  const bool vec = (a.C & 3) == 0;
  if (a.src == 0 && kr == a.Kcy) {
    float* dst = rin + (long)s * 2 * a.C + a.C;
    const float s0 = scal[0];
    if (vec) {
      for (int c = 4 * lane; c < a.C; c += 256) {
        const float4 t = *(const float4*)(tvr + c), r = *(const float4*)(rw + c), b = *(const float4*)(bf + c);
        *(float4*)(dst + c) = make_float4(t.x + s0 * r.x + b.x, t.y + s0 * r.y + b.y, t.z + s0 * r.z + b.z, t.w + s0 * r.w + b.w);
      }
    } else {
      for (int c = lane; c < a.C; c += 64) dst[c] = tvr[c] + s0 * rw[c] + bf[c];
    }
    return;
  }
  const int slot = a.lat0 + kr / a.Kp, k = kr % a.Kp;
  const long trow = (long)s * a.KLT + (long)slot * a.Kp + k, trows = (long)a.S * a.KLT;
  T* out = Text + trow * a.C;
  if (k >= a.K) {                                    // padding rows of a slot stay exactly zero
    if (vec) { for (int c = 4 * lane; c < a.C; c += 256) st4T<T>(out, c, make_float4(0.f, 0.f, 0.f, 0.f)); }
    else { for (int c = lane; c < a.C; c += 64) stT<T>(out, c, 0.f); }
    if (lane == 0) { Tsum[trow] = 0.f; Tsum[trows + trow] = 0.f; }
    return;
  }
  const float* tok = a.tok.p[a.e_of_lat[slot]] + (long)k * a.C;
  const float ab = a.src == 0 ? ldT<T>(BmX, ((long)s * a.Kcyb + kr) * a.Mb + a.M) : 0.f;
  double sm = 0.0, ss = 0.0;                         // row sum / sum of squares of the row AS STORED (the LayerNorm folds use the same numbers); in double:
                                                     // the variance the folds form from them is a difference of the two
  if (vec) {
    for (int c = 4 * lane; c < a.C; c += 256) {
      const float4 tk = *(const float4*)(tok + c), tv = *(const float4*)(tvr + c);
      float4 v = make_float4(tk.x + tv.x, tk.y + tv.y, tk.z + tv.z, tk.w + tv.w);
      if (a.src == 0) {
        const float4 r = *(const float4*)(rw + c), b = *(const float4*)(bf + c);
        v.x += ab * r.x + b.x; v.y += ab * r.y + b.y; v.z += ab * r.z + b.z; v.w += ab * r.w + b.w;
      }
      v.x = roundT<T>(v.x); v.y = roundT<T>(v.y); v.z = roundT<T>(v.z); v.w = roundT<T>(v.w);
      st4T<T>(out, c, v);
      sm += ((double)v.x + v.y) + ((double)v.z + v.w); ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
  } else {
    for (int c = lane; c < a.C; c += 64) {
      float v = tok[c] + tvr[c];
      if (a.src == 0) v += ab * rw[c] + bf[c];
      v = roundT<T>(v);
      stT<T>(out, c, v);
      sm += v; ss += (double)v * v;
    }
  }
  sm = wave_sum_d(sm); ss = wave_sum_d(ss);
  if (lane == 0) { Tsum[trow] = (float)sm; Tsum[trows + trow] = (float)ss; }
}
int k_finish_T(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, int src, hipStream_t st) {
  const Dims& d = pl.d;
  FinishTArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.tok.p[e] = prm.e[e].my_tokens; a.e_of_lat[e] = d.e_of_lat[e]; }
  a.S = d.S; a.C = d.C; a.K = d.K; a.Kp = d.Kp; a.KL = d.KL; a.KLT = d.KLT; a.Kcy = d.Kcy; a.Kcyb = d.Kcyb; a.Kcx = d.Kcx;
  a.Mb = d.Mb; a.M = d.M; a.src = src; a.lat0 = src == 0 ? 0 : d.Ey;
  const int rows = src == 0 ? d.Kcyb : d.Kcx;
  if (rows <= 0) return OK;
  DISPATCH_T(d.bf16, kk_finish_T, dim3((unsigned)cdiv((long)d.S * rows, 4)), dim3(256), 0, st, a, (const float*)(scratch + pl.o_TV),
             (const void*)(saved + pl.o_BmX), (const float*)(saved + pl.o_rw), (const float*)prm.fc_b,
             (const float*)(saved + pl.o_scal), (void*)(saved + pl.o_Text), (float*)(saved + pl.o_rin), (float*)(saved + pl.o_Tsum), d.S * rows);
  AVMOE_CHECK_LAUNCH("finish_T");
  return OK;
}

// ---------------------------------------------------------------------------------------------
// router  (net_trans_v3.py:460-466,477-479): one block per frame, fp32, fixed reduction order.
// ---------------------------------------------------------------------------------------------
struct RouterArgs {
  const float *W1, *b1, *W2, *b2, *W3, *b3, *noise;
  int C2, E, S;
};
// one block per frame: layer-1 pre-activations from the engine GEMM -- summed here over its split-K slabs, in slab order, when it
// ran split (no separate reduce pass) --, then bias / ReLU, the two small layers, softmax and first-max argmax.  fp32, fixed order.
__global__ void __launch_bounds__(128) kk_router_tail(RouterArgs a, float* rh1, const float* slabs, int ks, float* rh2, float* probs,
                                                      float* probs_out, int64_t* idx_out, float* lb_zero) {
  __shared__ float s_h1[128], s_h2[32], s_lg[MAX_E];
  const int s = blockIdx.x, t = threadIdx.x;
  if (lb_zero && s == 0 && t == 0) *lb_zero = 0.f;          // (sites without the load-balancing loss report 0)
  {
    float pre;
    if (ks > 1) {
      pre = 0.f;
      const long per = (long)a.S * 128;
      for (int k = 0; k < ks; ++k) pre += slabs[(long)k * per + (long)s * 128 + t];
    } else pre = rh1[(long)s * 128 + t];
    const float h = fmaxf(pre + a.b1[t], 0.f);
    s_h1[t] = h; rh1[(long)s * 128 + t] = h;                 // kept for the backward
  }
  __syncthreads();
  if (t < 32) {
    float acc = 0.f;
    for (int i = 0; i < 128; ++i) acc += a.W2[t * 128 + i] * s_h1[i];
    const float h = fmaxf(acc + a.b2[t], 0.f);
    s_h2[t] = h; rh2[(long)s * 32 + t] = h;
  }
  __syncthreads();
  if (t < a.E) {
    float acc = 0.f;
    for (int i = 0; i < 32; ++i) acc += a.W3[t * 32 + i] * s_h2[i];
    s_lg[t] = acc + a.b3[t] + (a.noise ? a.noise[(long)s * a.E + t] : 0.f);
  }
  __syncthreads();
  if (t == 0) {
    float mx = s_lg[0];
    for (int e = 1; e < a.E; ++e) mx = fmaxf(mx, s_lg[e]);
    float sum = 0.f;
    for (int e = 0; e < a.E; ++e) sum += expf(s_lg[e] - mx);
    int best = 0; float bp = -1.f;
    for (int e = 0; e < a.E; ++e) {
      const float p = expf(s_lg[e] - mx) / sum;
      probs[(long)s * a.E + e] = p;
      if (probs_out) probs_out[(long)s * a.E + e] = p;
      if (p > bp) { bp = p; best = e; }            // strict '>' : first maximum wins (torch.argmax)
    }
    if (idx_out) idx_out[s] = best;
  }
}
// load-balancing loss  -sum_e log(mean_s p[s,e])  (reference quirk: PVT_AVSModel_v2.py:314-318)
__global__ void __launch_bounds__(256) kk_lb_loss(const float* probs, int S, int E, float* lb) {
  __shared__ float red[4];
  float total = 0.f;
  for (int e = 0; e < E; ++e) {
    float acc = 0.f;
    for (int s = threadIdx.x; s < S; s += 256) acc += probs[(long)s * E + e];
    acc = block_sum256(acc, red);
    total += -logf(acc / (float)S);
  }
  if (threadIdx.x == 0) *lb = total;
}
int k_router(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const float* noise, float* probs_out,
             int64_t* idx_out, float* lb_out, hipStream_t st) {
  ProfScope ps_("k_router", 0.0, 0.0, st);
  const Dims& d = pl.d;
  if (!prm.r0_w || !prm.r0_b || !prm.r2_w || !prm.r2_b || !prm.r4_w || !prm.r4_b) {
    set_last_error("moe: router parameters missing"); return ERR_BAD_ARG;
  }
  RouterArgs a{prm.r0_w, prm.r0_b, prm.r2_w, prm.r2_b, prm.r4_w, prm.r4_b, noise, 2 * d.C, d.E, d.S};
  int ks = 1;
  {   // layer 1 on the matrix pipe in exact fp32 whatever the activation dtype (bit-stable argmax): rh1 = rin W1^T
    GemmArgs g;
    g.dtype = GEMM_F32; g.out_dtype = GEMM_F32;
    g.A = saved + pl.o_rin; g.B = prm.r0_w; g.C = saved + pl.o_rh1;
    g.M = d.S; g.N = 128; g.K = 2 * d.C; g.lda = 2L * d.C; g.ldb = 2L * d.C; g.sCi = 128;
    g.tile = 64; g.slabs = (float*)(scratch + pl.o_slabs); g.ksplit = choose_ksplit(g, slab_floats(d));   // few tiles, long K
    g.keep_slabs = 1;                                      // the tail kernel adds the slabs: no reduce launch
    ks = g.ksplit;
    AVMOE_TRY(launch_gemm(g, st));
  }
  hipLaunchKernelGGL(kk_router_tail, dim3(d.S), dim3(128), 0, st, a, (float*)(saved + pl.o_rh1), (const float*)(scratch + pl.o_slabs), ks,
                     (float*)(saved + pl.o_rh2), (float*)(saved + pl.o_probs), probs_out, idx_out, (lb_out && !d.lb_loss) ? lb_out : nullptr);
  AVMOE_CHECK_LAUNCH("router");
  if (lb_out && d.lb_loss) {
    hipLaunchKernelGGL(kk_lb_loss, dim3(1), dim3(256), 0, st, (const float*)(saved + pl.o_probs), d.S, d.E, lb_out);
    AVMOE_CHECK_LAUNCH("lb_loss");
  }
  return OK;
}

// ---------------------------------------------------------------------------------------------
// sub-ops of the C ABI (tests / partial adoption; not on the product path)
// ---------------------------------------------------------------------------------------------
// noise[s][e] = (e == hot) * value : logit "noise" that turns the router's softmax into an exact one-hot (avmoe_expert_forward_*)
__global__ void kk_onehot_noise(float* noise, long n, int E, int hot, float value) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) noise[i] = ((int)(i % E) == hot) ? value : 0.f;
}
int k_onehot_noise(float* noise, int S, int E, int hot, float value, hipStream_t st) {
  hipLaunchKernelGGL(kk_onehot_noise, dim3(grid1d((long)S * E, 256)), dim3(256), 0, st, noise, (long)S * E, E, hot, value);
  AVMOE_CHECK_LAUNCH("onehot_noise");
  return OK;
}
// Z[row][col] += rowb[row % period] + colb[col]   (either bias may be NULL): the biases of conv_adapter / fc on a materialised remap
template <typename T>
__global__ void kk_add_bias(void* Z_, long rows, int cols, int period, const float* rowb, const float* colb) {
  T* Z = (T*)Z_;
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / cols; const int col = (int)(i - row * cols);
    stT<T>(Z, i, ldT<T>(Z, i) + (rowb ? rowb[row % period] : 0.f) + (colb ? colb[col] : 0.f));
  }
}
int k_add_bias(bool bf16, void* Z, long rows, int cols, int period, const float* rowb, const float* colb, hipStream_t st) {
  DISPATCH_T(bf16, kk_add_bias, dim3(grid1d(rows * cols, 4096)), dim3(256), 0, st, Z, rows, cols, period, rowb, colb);
  AVMOE_CHECK_LAUNCH("add_bias");
  return OK;
}

// (PRE_SMALL / POST_SMALL live in tile_kernels.hip: 16-token MFMA tiles)

// ---------------------------------------------------------------------------------------------
// BN1 finalize: column partials -> batch mean / biased var -> (mean, rstd, scale, shift); running
// statistics updated in place (momentum, unbiased var).  Eval mode uses the running statistics.
// net_trans_v3.py:397-398 ; torch BatchNorm2d semantics.
// ---------------------------------------------------------------------------------------------
struct Bn1Args {
  P16 w, b; W16 rm, rv; N16 nbt;
  int E, g, dg, dgp, DZ, nblk, NT, use_bn, training;
  float eps, momentum;
};
// (cs0, cs1: the two column sums of this column over all blocks -- only read in training mode)
__device__ __forceinline__ void bn1_finalize_col(const Bn1Args& a, int col, float cs0, float cs1, float* bn1) {
  const int i = col / (a.E * a.dgp), e = (col / a.dgp) % a.E, jp = col % a.dgp;
  float mean = 0.f, rstd = 0.f, sc = 0.f, sh = 0.f;
  if (jp < a.dg) {
    const int j = i * a.dg + jp;
    if (!a.use_bn) { mean = 0.f; rstd = 1.f; sc = 1.f; sh = 0.f; }
    else {
      float var;
      if (a.training) {
        const double s0 = cs0, s1 = cs1;
        const double m = s0 / a.NT;
        const double v = fmax(s1 / a.NT - m * m, 0.0);
        mean = (float)m; var = (float)v;
        const double unb = a.NT > 1 ? v * ((double)a.NT / (a.NT - 1)) : v;
        a.rm.p[e][j] = (1.f - a.momentum) * a.rm.p[e][j] + a.momentum * mean;
        a.rv.p[e][j] = (1.f - a.momentum) * a.rv.p[e][j] + a.momentum * (float)unb;
        if (j == 0 && a.nbt.p[e]) a.nbt.p[e][0] += 1;          // bn1.num_batches_tracked (one column per expert gets here)
      } else { mean = a.rm.p[e][j]; var = a.rv.p[e][j]; }
      rstd = rsqrtf(var + a.eps);
      sc = a.w.p[e][j] * rstd;
      sh = a.b.p[e][j] - mean * sc;
    }
  }
  bn1[col] = mean; bn1[a.DZ + col] = rstd; bn1[2 * a.DZ + col] = sc; bn1[3 * a.DZ + col] = sh;
}
__global__ void kk_bn1_finalize(Bn1Args a, const float* colpart, float* bn1) {      // eval mode / no BatchNorm: no sums needed
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= a.DZ) return;
  bn1_finalize_col(a, col, 0.f, 0.f, bn1);
}
struct Bn1Fin : NoExtra {
  Bn1Args a; float* bn1;
  __device__ void operator()(int col, float s0, float s1) const { bn1_finalize_col(a, col, s0, s1, bn1); }
};
int k_bn1_finalize(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  ProfScope ps_("k_bn1_finalize", 0.0, 0.0, st);
  const Dims& d = pl.d;
  Bn1Args a;
  for (int e = 0; e < MAX_E; ++e) {
    a.w.p[e] = prm.e[e].bn1_w; a.b.p[e] = prm.e[e].bn1_b; a.rm.p[e] = prm.e[e].bn1_rm; a.rv.p[e] = prm.e[e].bn1_rv; a.nbt.p[e] = prm.e[e].bn1_nbt;
  }
  a.E = d.E; a.g = d.g; a.dg = d.dg; a.dgp = d.dgp; a.DZ = d.DZ; a.nblk = d.nblk_tok; a.NT = d.NT; a.use_bn = d.use_bn;
  a.training = d.training; a.eps = d.bn_eps; a.momentum = d.bn_momentum;
  if (d.use_bn && d.training)      // the column sums over the blocks and the per-column finalize in one launch
    return launch_colsum_fin((const float*)(scratch + pl.o_colpart), d.nblk_tok, d.DZ, 4L * d.DZ, d.DZ, Bn1Fin{{}, a, (float*)(saved + pl.o_bn1)}, st);
  hipLaunchKernelGGL(kk_bn1_finalize, dim3(cdiv(d.DZ, 256)), dim3(256), 0, st, a, (const float*)(scratch + pl.o_colsum),
                     (float*)(saved + pl.o_bn1));
  AVMOE_CHECK_LAUNCH("bn1_finalize");
  return OK;
}

// ---------------------------------------------------------------------------------------------
// MID: z' = act(z * scale + shift) -> Zp (T, operand of the second-moment GEMM) + column sums of z'
// (thread <-> column, rows looped: no reduction inside the block).   net_trans_v3.py:397-400
// ---------------------------------------------------------------------------------------------
struct MidArgs { int relu_of_e[MAX_E]; int E, dgp, DZ, NT; };
template <typename T>
__global__ void __launch_bounds__(256) kk_mid(MidArgs a, const float* Z, const float* bn1, void* Zp_, float* colpart, int rows_per_blk) {
  T* Zp = (T*)Zp_;
  const long r0 = (long)blockIdx.x * rows_per_blk, r1 = min((long)a.NT, r0 + rows_per_blk);
  for (int col = threadIdx.x; col < a.DZ; col += 256) {
    const float sc = bn1[2 * a.DZ + col], sh = bn1[3 * a.DZ + col];
    const bool relu = a.relu_of_e[(col / a.dgp) % a.E];
    float acc = 0.f;
    for (long t = r0; t < r1; ++t) {
      float y = Z[t * a.DZ + col] * sc + sh;
      if (relu) y = fmaxf(y, 0.f);
      y = roundT<T>(y);
      stT<T>(Zp, t * a.DZ + col, y);
      acc += y;
    }
    colpart[((long)blockIdx.x * 4 + 0) * a.DZ + col] = acc;
  }
}
__global__ void kk_colsum_finalize(const float* colsum, int DZ, int slot, float scale, float* out) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= DZ) return;
  out[col] = colsum[(long)slot * DZ + col] * scale;
}
// ---- merged groups -----------------------------------------------------------------------------------------------------------
__global__ void kk_merge_expand(P16 down, P16 up, float* __restrict__ mWd, float* __restrict__ mWu, int E, int d, int C, int mg) {
  const int dgt = d / mg, Cgt = C / mg;
  const long per = (long)d * C, total = (long)E * per;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int e = (int)(i / per);
    const long r = i - (long)e * per;
    {                                                      // down: (d, C) from (d, C / mg)
      const int j = (int)(r / C), c = (int)(r % C);
      mWd[i] = (c / Cgt == j / dgt) ? down.p[e][(long)j * Cgt + c % Cgt] : 0.f;
    }
    {                                                      // up: (C, d) from (C, d / mg)
      const int c = (int)(r / d), j = (int)(r % d);
      mWu[i] = (j / dgt == c / Cgt) ? up.p[e][(long)c * dgt + j % dgt] : 0.f;
    }
  }
}
int k_merge_expand(const Plan& pl, char* saved, const avmoe_moe_ptrs& prm, hipStream_t st) {
  const Dims& d = pl.d;
  P16 dn, up;
  for (int e = 0; e < MAX_E; ++e) { dn.p[e] = prm.e[e].down_w; up.p[e] = prm.e[e].up_w; }
  for (int e = 0; e < d.E; ++e)
    if (!dn.p[e] || !up.p[e]) { set_last_error("moe: expert %d has no down/up weights", e); return ERR_BAD_ARG; }
  const long total = (long)d.E * d.d * d.C;
  hipLaunchKernelGGL(kk_merge_expand, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, dn, up,
                     (float*)(saved + pl.o_mWd), (float*)(saved + pl.o_mWu), d.E, d.d, d.C, d.mg);
  AVMOE_CHECK_LAUNCH("merge_expand");
  return OK;
}
avmoe_moe_ptrs merged_params(const Plan& pl, const avmoe_moe_ptrs& prm, char* saved) {
  avmoe_moe_ptrs p = prm;
  const Dims& d = pl.d;
  for (int e = 0; e < d.E; ++e) {
    p.e[e].down_w = (float*)(saved + pl.o_mWd) + (size_t)e * d.d * d.C;
    p.e[e].up_w = (float*)(saved + pl.o_mWu) + (size_t)e * d.d * d.C;
  }
  return p;
}

int k_mid(const Plan& pl, char* saved, char* scratch, hipStream_t st) {
  const Dims& d = pl.d;
  ProfScope ps_("k_mid", (long)d.NT, (double)d.NT * d.DZ * (double)(d.zsz + d.esz), 0.0, st);
  MidArgs a;
  for (int e = 0; e < MAX_E; ++e) a.relu_of_e[e] = d.relu_of_e[e];
  a.E = d.E; a.dgp = d.dgp; a.DZ = d.DZ; a.NT = d.NT;
  const int nblk = d.nblk_tok;
  const int rpb = cdiv(d.NT, nblk);
  if (tile_fast_ok(d)) AVMOE_TRY(kf_mid(pl, saved, scratch, st));
  else if (d.gen) AVMOE_TRY(kg_mid(pl, saved, scratch, st));
  else DISPATCH_T(d.bf16, kk_mid, dim3(nblk), dim3(256), 0, st, a, (const float*)(saved + pl.o_Z), (const float*)(saved + pl.o_bn1),
                  (void*)(scratch + pl.o_Zp), (float*)(scratch + pl.o_colpart), rpb);
  AVMOE_TRY(k_reduce_colpart(pl, scratch, 0, 1, st));
  hipLaunchKernelGGL(kk_colsum_finalize, dim3(cdiv(d.DZ, 256)), dim3(256), 0, st, (const float*)(scratch + pl.o_colsum),
                     d.DZ, 0, 1.f / (float)d.NT, (float*)(saved + pl.o_mz));
  AVMOE_CHECK_LAUNCH("mid");
  return OK;
}

// (POST_PREP lives in weight_kernels.hip)

}  // namespace avmoe
