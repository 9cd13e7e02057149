// AVS unimodal expert, self_attention_version "v1" (PVT_AVSModel_v2.py:141-142, 210-214): the expert's input is REPLACED by
//
//     nn.MultiheadAttention(C, num_heads = 4, dropout = 0.2)(x, x, x)        with x of shape (S, N, C), batch_first = False
//
// i.e. the SEQUENCE axis is the frames S and the batch axis the tokens N: every token attends over the frames (of all clips
// of the batch) at its own position.  torch.nn.functional.multi_head_attention_forward, need_weights = True path:
//
//     [q | k | v] = x Win^T + bin ;  per (token n, head h):  P = softmax_s'( q k^T / sqrt(dh) ) ;  Pd = dropout(P) ;  o = Pd v
//     MHA(x) = concat_h(o) Wout^T + bout
//
// Here the stage produces  xr = MHA(x) - x  for one expert ("xr slot"); the bottleneck path then treats the expert like the
// AVVP ones, input x + 1 * xr (moe_forward.cpp), and the backward hands back d xr.  Every product is a call of the GEMM engine
// on strided views of the token-major tensors (token n = batch index 1, head h = batch index 2, frame s = row); the glue is
// elementwise.  Dropout is a caller-supplied multiplier (include/avmoe.h, sa_keep).
#include "kernels.h"
#include "moe_run.h"
#include "device_utils.h"
#include "prof.h"
#include "gemm.h"
#include <algorithm>
#include <cmath>

namespace avmoe {

#define DISPATCH_T(bf16, KERN, grid, block, shmem, st, ...)                                   \
  do {                                                                                        \
    if (bf16) hipLaunchKernelGGL((KERN<__bf16>), grid, block, shmem, st, __VA_ARGS__);        \
    else hipLaunchKernelGGL((KERN<float>), grid, block, shmem, st, __VA_ARGS__);              \
  } while (0)

namespace {

inline unsigned grid1(long n, int cap = 8192) { return (unsigned)std::max<long>(1, std::min<long>((n + 255) / 256, cap)); }

// rows[t][c] += bias[c]
template <typename T>
__global__ void km_add_bias(void* rows_, const float* bias, long rows, int cols) {
  T* p = (T*)rows_;
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) stT<T>(p, i, ldT<T>(p, i) + bias[i % cols]);
}
// Pd[b][s][s'] = P[b][s][s'] * keep[b][s][s']   (row stride Sp, padding stays 0)
template <typename T>
__global__ void km_keep(const void* P_, const float* keep, void* Pd_, long mats, int S, int Sp) {
  const T* P = (const T*)P_; T* Pd = (T*)Pd_;
  const long total = mats * S * Sp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int j = (int)(i % Sp);
    const long row = i / Sp;
    stT<T>(Pd, i, j < S ? ldT<T>(P, i) * keep[row * S + j] : 0.f);
  }
}
// d P = d Pd * keep   (fp32, in place)
__global__ void km_keep_bwd(float* dP, const float* keep, long mats, int S, int Sp) {
  const long total = mats * S * Sp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int j = (int)(i % Sp);
    if (j < S) dP[i] *= keep[(i / Sp) * S + j];
  }
}
// xr = (O Wout^T) + bout - X
template <typename T>
__global__ void km_finish(void* xr_, const float* bout, const void* X_, long rows, int C) {
  T* xr = (T*)xr_; const T* X = (const T*)X_;
  const long total = rows * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    stT<T>(xr, i, ldT<T>(xr, i) + bout[i % C] - ldT<T>(X, i));
}
// part[chunk][c] = sum over the rows of the chunk of rows[t][c]   (grid: column blocks x row chunks; summed over the chunks by
// k_colsum_f32 -- deterministic, no atomics)
template <typename T>
__global__ void __launch_bounds__(256) km_colsum_part(const void* rows_, long rows, int cols, int rows_per_chunk, float* part) {
  const T* p = (const T*)rows_;
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  const long t0 = (long)blockIdx.y * rows_per_chunk, t1 = t0 + rows_per_chunk < rows ? t0 + rows_per_chunk : rows;
  float acc = 0.f;
  if (c < cols)
    for (long t = t0 + wave; t < t1; t += 4) acc += ldT<T>(p, t * cols + c);
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < cols) part[(long)blockIdx.y * cols + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

int colsum_rows(int bf16, const void* rows, long nrows, int cols, float* part /* >= 256 * cols floats */, float* out, hipStream_t st) {
  const int chunks = (int)std::max<long>(1, std::min<long>(256, nrows / 64));
  const int rpc = (int)((nrows + chunks - 1) / chunks);
  DISPATCH_T(bf16, km_colsum_part, dim3(cdiv(cols, 64), chunks), dim3(256), 0, st, rows, nrows, cols, rpc, part);
  AVMOE_CHECK_LAUNCH("mha colsum");
  return k_colsum_f32(part, chunks, cols, cols, 1, 0, out, 0, 1.f, st);
}

struct Slot {
  const Dims& d;
  char* sv; char* sc; const Plan& pl;
  int slot;
  size_t esz;
  char* Win() const { return sv + pl.o_mWin + (size_t)slot * 3 * d.C * d.C * esz; }
  char* Wout() const { return sv + pl.o_mWout + (size_t)slot * d.C * d.C * esz; }
  char* QKV() const { return sv + pl.o_mQKV + (size_t)slot * d.NT * 3 * d.C * esz; }
  char* P() const { return sv + pl.o_mP + (size_t)slot * d.N * d.H * d.S * d.Sp * esz; }
  char* Pd() const { return sv + pl.o_mPd + (size_t)slot * d.N * d.H * d.S * d.Sp * esz; }
  char* O() const { return sv + pl.o_mO + (size_t)slot * d.NT * d.C * esz; }
  char* xr() const { return sv + pl.o_xr + (size_t)slot * d.NT * d.C * esz; }
};

// the per-(token, head) views of [q | k | v] / of a (S, N, C) tensor: frame s is the row, stride N * width
void head_view_a(GemmArgs& g, const Dims& d, int width) { g.lda = (long)d.N * width; g.sA1 = width; g.sA2 = d.dh; }
void head_view_b(GemmArgs& g, const Dims& d, int width) { g.ldb = (long)d.N * width; g.sB1 = width; g.sB2 = d.dh; }
void head_view_c(GemmArgs& g, const Dims& d, int width) { g.sCi = (long)d.N * width; g.sC1 = width; g.sC2 = d.dh; }
void mat_view_a(GemmArgs& g, const Dims& d) { g.lda = d.Sp; g.sA1 = (long)d.H * d.S * d.Sp; g.sA2 = (long)d.S * d.Sp; }
void mat_view_c(GemmArgs& g, const Dims& d) { g.sCi = d.Sp; g.sC1 = (long)d.H * d.S * d.Sp; g.sC2 = (long)d.S * d.Sp; }

}  // namespace

int mha_frames_forward(const Plan& pl, const void* X, const avmoe_expert_ptrs& ep, int slot, char* sv, char* sc, hipStream_t st) {
  const Dims& d = pl.d;
  ProfScope ps_("mha_frames_forward", 0.0, 0.0, st);
  if (!ep.sa_in_w || !ep.sa_in_b || !ep.sa_out_w || !ep.sa_out_b) { set_last_error("moe: self_attention.* pointers missing"); return ERR_BAD_ARG; }
  const int dt = d.bf16 ? GEMM_BF16 : GEMM_F32;
  const Slot b{d, sv, sc, pl, slot, (size_t)d.esz};
  const long mats = (long)d.N * d.H;
  auto base = [&]() { GemmArgs g; g.dtype = dt; g.out_dtype = dt; return g; };
  AVMOE_TRY(k_cast(d.bf16, ep.sa_in_w, 3L * d.C, d.C, d.C, b.Win(), d.C, st));
  AVMOE_TRY(k_cast(d.bf16, ep.sa_out_w, d.C, d.C, d.C, b.Wout(), d.C, st));
  {                                                        // [q | k | v] = X Win^T (+ bin)
    GemmArgs g = base();
    g.A = X; g.B = b.Win(); g.C = b.QKV(); g.M = d.NT; g.N = 3 * d.C; g.K = d.C; g.lda = d.C; g.ldb = d.C; g.sCi = 3L * d.C;
    AVMOE_TRY(launch_gemm(g, st));
  }
  DISPATCH_T(d.bf16, km_add_bias, dim3(grid1((long)d.NT * 3 * d.C)), dim3(256), 0, st, (void*)b.QKV(), (const float*)ep.sa_in_b, (long)d.NT, 3 * d.C);
  AVMOE_CHECK_LAUNCH("mha add_bias");
  {                                                        // scores[n][h] = q k^T / sqrt(dh)
    GemmArgs g = base();
    g.A = b.QKV(); g.B = b.QKV() + (size_t)d.C * d.esz; g.C = sc + pl.o_mSc; g.out_dtype = GEMM_F32;
    g.M = d.S; g.N = d.S; g.K = d.dh; g.nb1 = d.N; g.nb2 = d.H; g.alpha = 1.f / std::sqrt((float)d.dh);
    head_view_a(g, d, 3 * d.C); head_view_b(g, d, 3 * d.C); mat_view_c(g, d);
    AVMOE_TRY(launch_gemm(g, st));
  }
  AVMOE_TRY(k_softmax_rows(d.bf16, (const float*)(sc + pl.o_mSc), mats * d.S, d.S, d.Sp, b.P(), d.Sp, 1, 1, 1, 1, st));
  const char* Pd = b.P();
  if (ep.sa_keep) {
    DISPATCH_T(d.bf16, km_keep, dim3(grid1(mats * d.S * d.Sp)), dim3(256), 0, st, (const void*)b.P(), (const float*)ep.sa_keep, (void*)b.Pd(), mats, d.S, d.Sp);
    AVMOE_CHECK_LAUNCH("mha keep");
    Pd = b.Pd();
  }
  {                                                        // o[n][h] = Pd v  -> O (S, N, C)
    GemmArgs g = base();
    g.A = Pd; g.B = b.QKV() + (size_t)2 * d.C * d.esz; g.C = b.O();
    g.M = d.S; g.N = d.dh; g.K = d.S; g.nb1 = d.N; g.nb2 = d.H; g.b_layout = MN_MAJOR;
    mat_view_a(g, d); head_view_b(g, d, 3 * d.C); head_view_c(g, d, d.C);
    AVMOE_TRY(launch_gemm(g, st));
  }
  {                                                        // xr = O Wout^T + bout - X
    GemmArgs g = base();
    g.A = b.O(); g.B = b.Wout(); g.C = b.xr(); g.M = d.NT; g.N = d.C; g.K = d.C; g.lda = d.C; g.ldb = d.C; g.sCi = d.C;
    AVMOE_TRY(launch_gemm(g, st));
  }
  DISPATCH_T(d.bf16, km_finish, dim3(grid1((long)d.NT * d.C)), dim3(256), 0, st, (void*)b.xr(), (const float*)ep.sa_out_b, X, (long)d.NT, d.C);
  AVMOE_CHECK_LAUNCH("mha finish");
  return OK;
}

// dxr (NT, C) in T -> parameter gradients of the expert's self_attention.*, dX += d MHA / d x  (the "- dxr" of the replaced
// input is applied by k_nxn_axpy)
int mha_frames_backward(const Plan& pl, const void* X, const avmoe_expert_ptrs& ep, const avmoe_expert_ptrs& eg, int slot, const void* dxr,
                        char* sv, char* sc, float* slabs, size_t slab_cap, void* dX, hipStream_t st) {
  const Dims& d = pl.d;
  ProfScope ps_("mha_frames_backward", 0.0, 0.0, st);
  const int dt = d.bf16 ? GEMM_BF16 : GEMM_F32;
  const Slot b{d, sv, sc, pl, slot, (size_t)d.esz};
  const long mats = (long)d.N * d.H;
  float* sink = (float*)(sc + pl.o_mdW);                   // [3C * C | 3C | C]  for gradients nobody asked for
  float* dWin = eg.sa_in_w ? eg.sa_in_w : sink;
  float* dWout = eg.sa_out_w ? eg.sa_out_w : sink;
  float* dbin = eg.sa_in_b ? eg.sa_in_b : sink + (size_t)3 * d.C * d.C;
  float* dbout = eg.sa_out_b ? eg.sa_out_b : sink + (size_t)3 * d.C * d.C + 3 * d.C;
  const char* Pd = ep.sa_keep ? b.Pd() : b.P();
  char* dO = sc + pl.o_mdO; char* dS = sc + pl.o_mdS; char* dQKV = sc + pl.o_mdQKV; float* dP = (float*)(sc + pl.o_mSc);
  const float alpha = 1.f / std::sqrt((float)d.dh);
  auto base = [&]() { GemmArgs g; g.dtype = dt; g.out_dtype = dt; g.slabs = slabs; return g; };
  auto run = [&](GemmArgs& g, bool split) {
    if (split) g.ksplit = choose_ksplit(g, slab_cap);
    return launch_gemm(g, st);
  };
  float* part = (float*)(sc + pl.o_mpart);
  AVMOE_TRY(colsum_rows(d.bf16, dxr, (long)d.NT, d.C, part, dbout, st));
  {                                                        // d Wout = dxr^T O
    GemmArgs g = base();
    g.A = dxr; g.B = b.O(); g.C = dWout; g.out_dtype = GEMM_F32;
    g.M = d.C; g.N = d.C; g.K = d.NT; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.C; g.ldb = d.C; g.sCi = d.C;
    AVMOE_TRY(run(g, true));
  }
  {                                                        // d O = dxr Wout
    GemmArgs g = base();
    g.A = dxr; g.B = b.Wout(); g.C = dO; g.M = d.NT; g.N = d.C; g.K = d.C; g.lda = d.C; g.b_layout = MN_MAJOR; g.ldb = d.C; g.sCi = d.C;
    AVMOE_TRY(run(g, false));
  }
  {                                                        // d Pd[n][h] = dO v^T
    GemmArgs g = base();
    g.A = dO; g.B = b.QKV() + (size_t)2 * d.C * d.esz; g.C = dP; g.out_dtype = GEMM_F32;
    g.M = d.S; g.N = d.S; g.K = d.dh; g.nb1 = d.N; g.nb2 = d.H;
    head_view_a(g, d, d.C); head_view_b(g, d, 3 * d.C); mat_view_c(g, d);
    AVMOE_TRY(run(g, false));
  }
  {                                                        // d v[n][h] = Pd^T dO
    GemmArgs g = base();
    g.A = Pd; g.B = dO; g.C = dQKV + (size_t)2 * d.C * d.esz;
    g.M = d.S; g.N = d.dh; g.K = d.S; g.nb1 = d.N; g.nb2 = d.H; g.a_layout = g.b_layout = MN_MAJOR;
    mat_view_a(g, d); head_view_b(g, d, d.C); head_view_c(g, d, 3 * d.C);
    AVMOE_TRY(run(g, false));
  }
  if (ep.sa_keep) {
    hipLaunchKernelGGL(km_keep_bwd, dim3(grid1(mats * d.S * d.Sp)), dim3(256), 0, st, dP, (const float*)ep.sa_keep, mats, d.S, d.Sp);
    AVMOE_CHECK_LAUNCH("mha keep_bwd");
  }
  AVMOE_TRY(k_softmax_rows_bwd(d.bf16, b.P(), dP, mats * d.S, d.S, d.Sp, dS, nullptr, 1, 1, st));
  {                                                        // d q[n][h] = dS k / sqrt(dh)
    GemmArgs g = base();
    g.A = dS; g.B = b.QKV() + (size_t)d.C * d.esz; g.C = dQKV; g.alpha = alpha;
    g.M = d.S; g.N = d.dh; g.K = d.S; g.nb1 = d.N; g.nb2 = d.H; g.b_layout = MN_MAJOR;
    mat_view_a(g, d); head_view_b(g, d, 3 * d.C); head_view_c(g, d, 3 * d.C);
    AVMOE_TRY(run(g, false));
  }
  {                                                        // d k[n][h] = dS^T q / sqrt(dh)
    GemmArgs g = base();
    g.A = dS; g.B = b.QKV(); g.C = dQKV + (size_t)d.C * d.esz; g.alpha = alpha;
    g.M = d.S; g.N = d.dh; g.K = d.S; g.nb1 = d.N; g.nb2 = d.H; g.a_layout = g.b_layout = MN_MAJOR;
    mat_view_a(g, d); head_view_b(g, d, 3 * d.C); head_view_c(g, d, 3 * d.C);
    AVMOE_TRY(run(g, false));
  }
  AVMOE_TRY(colsum_rows(d.bf16, dQKV, (long)d.NT, 3 * d.C, part, dbin, st));
  {                                                        // d Win = dQKV^T X
    GemmArgs g = base();
    g.A = dQKV; g.B = X; g.C = dWin; g.out_dtype = GEMM_F32;
    g.M = 3 * d.C; g.N = d.C; g.K = d.NT; g.a_layout = g.b_layout = MN_MAJOR; g.lda = 3L * d.C; g.ldb = d.C; g.sCi = d.C;
    AVMOE_TRY(run(g, true));
  }
  {                                                        // dX += dQKV Win
    GemmArgs g = base();
    g.A = dQKV; g.B = b.Win(); g.C = dX; g.M = d.NT; g.N = d.C; g.K = 3 * d.C; g.lda = 3L * d.C; g.b_layout = MN_MAJOR; g.ldb = d.C;
    g.sCi = d.C; g.accumulate = 1;
    AVMOE_TRY(run(g, false));
  }
  return OK;
}

}  // namespace avmoe
