// Backward bottleneck-space kernels.  Every formula is the one in oracle/algebra_ref.py::AlgebraRef.backward
// (validated there against autograd); stage names match.  Full-width tensors are only touched by GEMMs.
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

__device__ __forceinline__ void wave_lds_sync_b() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <typename T> __device__ __forceinline__ float roundTb(float v);
template <> __device__ __forceinline__ float roundTb<float>(float v) { return v; }
template <> __device__ __forceinline__ float roundTb<__bf16>(float v) { return bf2f(f2bf(v)); }

static inline unsigned grid1db(long n, int cap = 4096) { return (unsigned)std::max<long>(1, std::min<long>((n + 255) / 256, cap)); }


// (the per-token backward kernels live in tile_kernels.hip; this file keeps their finalize / weight-space parts)

// finalize of POST_SMALL backward in ONE launch: the column sums of the per-block partials with their epilogue (dusum, dvh) and, in
// E extra blocks, the per-expert sums over the frames: dp[s][e], dH1[e], dH2[e], grads.gate
struct PostFinArgs { P16 gate; W16 ggate; int S, E, DZ, nblk, bps, use_gate; };
struct PostFin {
  PostFinArgs a; const float* blkscal; const float* probs; float* dsm; float* dp;
  // dsm layout: [0]=dusum [1]=dvh [2]=dmz/NT [3]=mdy [4]=mdyz [5]=ddconst [6]=dwsum [7]=spare (each DZ) ; then dH1[E], dH2[E]
  __device__ void operator()(int col, float s0, float s1) const { dsm[col] = s0; dsm[a.DZ + col] = 2.f * s1; }
  template <int NTHR> __device__ void extra(int e, int) const {
    __shared__ float red[NTHR / 64];
    float h1 = 0.f, h2 = 0.f, gsum = 0.f;
    const float gate = a.use_gate ? a.gate.p[e][0] : 1.f;
    for (int s = threadIdx.x; s < a.S; s += NTHR) {
      float dq = 0.f;
      for (int b = 0; b < a.bps; ++b) {
        const float* p = blkscal + (((long)s * a.bps + b) * a.E + e) * 4;
        dq += p[0]; h1 += p[1]; h2 += p[2];
      }
      dp[(long)s * a.E + e] = gate * dq;
      gsum += probs[(long)s * a.E + e] * dq;
    }
    h1 = block_sum_fixed<NTHR>(h1, red); h2 = block_sum_fixed<NTHR>(h2, red); gsum = block_sum_fixed<NTHR>(gsum, red);
    if (threadIdx.x == 0) {
      dsm[8 * a.DZ + e] = h1; dsm[8 * a.DZ + a.E + e] = h2;
      if (a.use_gate && a.ggate.p[e]) a.ggate.p[e][0] = gsum;
    }
  }
};

int k_post_small_bwd_finalize(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads,
                              hipStream_t st) {
  const Dims& d = pl.d;
  PostFinArgs f;
  for (int e = 0; e < MAX_E; ++e) { f.gate.p[e] = prm.e[e].gate; f.ggate.p[e] = grads.e[e].gate; }
  f.S = d.S; f.E = d.E; f.DZ = d.DZ; f.nblk = d.nblk_tok; f.bps = d.nblk_tok / d.S; f.use_gate = d.use_gate && !d.gate_w;      // (gate_w: dAp already carries the gate -- dp = dq, and dgate comes from weight space)
  return launch_colsum_fin((const float*)(scratch + pl.o_colpart), d.nblk_tok, d.DZ, 4L * d.DZ, d.DZ,
                           PostFin{f, (const float*)(scratch + pl.o_blkscal), (const float*)(saved + pl.o_probs), (float*)(scratch + pl.o_dsm),
                                   (float*)(scratch + pl.o_dp)}, st, d.E);
}

// (POST_PREP backward lives in weight_kernels.hip)

// ---------------------------------------------------------------------------------------------
// MID backward: total dz' (direct + BN2-moment terms), ReLU mask, BN1 reduction sums.
// ---------------------------------------------------------------------------------------------
struct MidBwdArgs {
  int relu_of_e[MAX_E]; W16 gw1, gb1;
  int S, N, E, DD, DZ, dgp, dg, g, NT, use_bn, training, nblk;
};
struct MidBwdFin : NoExtra {      // epilogue of the column sums (slots 2, 3 of the per-block partials) -- colsum_fin.h
  MidBwdArgs a; float* dsm;
  __device__ void operator()(int col, float cs0, float cs1) const {
  const double s0 = cs0, s1 = cs1;
  dsm[3 * a.DZ + col] = (float)(s0 / a.NT);
  dsm[4 * a.DZ + col] = (float)(s1 / a.NT);
  const int i = col / (a.E * a.dgp), e = (col / a.dgp) % a.E, jp = col % a.dgp;
  if (a.use_bn && jp < a.dg) {
    const int j = i * a.dg + jp;
    if (a.gw1.p[e]) a.gw1.p[e][j] = (float)s1;
    if (a.gb1.p[e]) a.gb1.p[e][j] = (float)s0;
  }
  }
};
int k_mid_bwd_finalize(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads, hipStream_t st) {
  const Dims& d = pl.d;
  MidBwdArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.relu_of_e[e] = d.relu_of_e[e]; a.gw1.p[e] = grads.e[e].bn1_w; a.gb1.p[e] = grads.e[e].bn1_b; }
  a.S = d.S; a.N = d.N; a.E = d.E; a.DD = d.DD; a.DZ = d.DZ; a.dgp = d.dgp; a.dg = d.dg; a.g = d.g; a.NT = d.NT;
  a.use_bn = d.use_bn; a.training = d.training; a.nblk = d.nblk_tok;
  return launch_colsum_fin((const float*)(scratch + pl.o_colpart) + 2L * d.DZ, d.nblk_tok, d.DZ, 4L * d.DZ, d.DZ,
                           MidBwdFin{{}, a, (float*)(scratch + pl.o_dsm)}, st);
}

// ---------------------------------------------------------------------------------------------
// PRE_SMALL backward: BN1 input gradient, folded-LayerNorm statistics, hop-2 softmax.
// ---------------------------------------------------------------------------------------------
// finalize in ONE launch: the column sums of the per-block partials (ddconst, dwsum) and, in extra blocks, one per expert with a
// latent / attention gate (its gradient: a sum over the blocks' scalars) and a few for dtbar[s][kc] = sum over the frame's blocks
struct PreFinArgs { W16 gglat; int lat_of_e[MAX_E]; int nxn_of_e[MAX_E]; int S, E, DZ, KL, nblk, bps; };
struct PreFin {
  PreFinArgs a; const float* blkscal; const float* dtbp; float* dsm; float* dtbar;
  __device__ void operator()(int col, float s0, float s1) const { dsm[5 * a.DZ + col] = s0; dsm[6 * a.DZ + col] = s1; }
  template <int NTHR> __device__ void extra(int bx, int nextra) const {
    if (bx < a.E) {
      __shared__ float red[NTHR / 64];
      const int e = bx;
      if (a.lat_of_e[e] < 0 && !a.nxn_of_e[e]) return;
      float acc = 0.f;
      for (int b = threadIdx.x; b < a.nblk; b += NTHR) acc += blkscal[((long)b * a.E + e) * 4 + 3];
      acc = block_sum_fixed<NTHR>(acc, red);
      if (threadIdx.x == 0 && a.gglat.p[e]) a.gglat.p[e][0] = acc;
      return;
    }
    for (long i = (long)(bx - a.E) * NTHR + threadIdx.x; i < (long)a.S * a.KL; i += (long)(nextra - a.E) * NTHR) {
      const int s = (int)(i / a.KL), kc = (int)(i % a.KL);
      float acc = 0.f;
      for (int b = 0; b < a.bps; ++b) acc += dtbp[((long)s * a.bps + b) * a.KL + kc];
      dtbar[i] = acc;
    }
  }
};
int k_pre_small_bwd_finalize(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads,
                             hipStream_t st) {
  const Dims& d = pl.d;
  PreFinArgs f;
  for (int e = 0; e < MAX_E; ++e) { f.gglat.p[e] = grads.e[e].gate_lat; f.lat_of_e[e] = d.lat_of_e[e]; f.nxn_of_e[e] = d.nxn_of_e[e]; }
  f.S = d.S; f.E = d.E; f.DZ = d.DZ; f.KL = d.KL; f.nblk = d.nblk_tok; f.bps = d.nblk_tok / d.S;
  const int nb_dtbar = d.KL > 0 ? std::max(1, std::min(64, cdiv((long)d.S * d.KL, 1024))) : 0;
  return launch_colsum_fin((const float*)(scratch + pl.o_colpart), d.nblk_tok, d.DZ, 4L * d.DZ, d.DZ,
                           PreFin{f, (const float*)(scratch + pl.o_blkscal), (const float*)(scratch + pl.o_dtbp), (float*)(scratch + pl.o_dsm),
                                  (float*)(scratch + pl.o_dtbar)}, st, d.E + nb_dtbar);
}

// ---------------------------------------------------------------------------------------------
// router backward (mixture weights, LB loss, 3-layer MLP)   net_trans_v3.py:460-466,477-478
// rbw layout: dlog [S][E] | dh2r [S][32] | dh1 [S][128] | drin [S][2C]
// ---------------------------------------------------------------------------------------------
struct RouterBwdArgs { const float *W1, *W2, *W3; int C2, E, S, lb_loss; const float* lb_grad; };
// Launch 1: per frame dlog / dh2 / dh1 (softmax, LB loss, the two small layers), one block per frame
__global__ void __launch_bounds__(256) kk_router_bwd_a(RouterBwdArgs a, const float* probs, const float* dp, const float* rh1,
                                                       const float* rh2, float* rbw) {
  __shared__ float s_dl[MAX_E], s_d2[32], s_pm[MAX_E];
  const int s = blockIdx.x;
  float* dh1 = rbw + (long)s * 128;
  float* dh2 = rbw + (long)a.S * (128 + a.C2) + (long)s * 32;
  float* dlog = rbw + (long)a.S * (128 + a.C2 + 32) + (long)s * a.E;
  if (a.lb_loss && a.lb_grad) {                  // column means of p for the LB loss: 256 / 16 frame streams per expert
    __shared__ float s_part[16][MAX_E];
    const int e = threadIdx.x % MAX_E, u = threadIdx.x / MAX_E;
    float acc = 0.f;
    if (e < a.E) for (int ss = u; ss < a.S; ss += 16) acc += probs[(long)ss * a.E + e];
    s_part[u][e] = acc;
    __syncthreads();
    if (threadIdx.x < a.E) {
      float t = 0.f;
      for (int k = 0; k < 16; ++k) t += s_part[k][threadIdx.x];
      s_pm[threadIdx.x] = t / (float)a.S;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float dot = 0.f, dpv[MAX_E];
    for (int e = 0; e < a.E; ++e) {
      dpv[e] = dp[(long)s * a.E + e];
      if (a.lb_loss && a.lb_grad) dpv[e] += a.lb_grad[0] * (-1.f / ((float)a.S * s_pm[e]));
      dot += probs[(long)s * a.E + e] * dpv[e];
    }
    for (int e = 0; e < a.E; ++e) { const float v = probs[(long)s * a.E + e] * (dpv[e] - dot); s_dl[e] = v; dlog[e] = v; }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    float acc = 0.f;
    for (int e = 0; e < a.E; ++e) acc += s_dl[e] * a.W3[e * 32 + threadIdx.x];
    const float v = rh2[(long)s * 32 + threadIdx.x] > 0.f ? acc : 0.f;
    s_d2[threadIdx.x] = v; dh2[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    float acc = 0.f;
    for (int j = 0; j < 32; ++j) acc += s_d2[j] * a.W2[j * 128 + threadIdx.x];
    dh1[threadIdx.x] = rh1[(long)s * 128 + threadIdx.x] > 0.f ? acc : 0.f;
  }
}
// Last launch, three independent jobs picked by the block index:
//   [0, nb_small)  the weight gradients of the two small layers and all biases: one output per lane, 4 frame streams per output,
//                  combined through LDS in a fixed order
//   then nb_c      dm1 / N (from drin, the engine GEMM before) into the extra row of Text[s] -- operand of the dX GEMM
//   then the rest  dW1 = sum over the split-K slabs of dh1^T rin, in slab order (the engine GEMM kept its slabs: no reduce launch)
__global__ void __launch_bounds__(256) kk_router_bwd_fin(int S, int E, int C2, int nb_small, int nb_c, const float* rbw, const float* rh1,
                                                         const float* rh2, float* gW1, float* gb1, float* gW2, float* gb2, float* gW3, float* gb3,
                                                         const float* slabs, int ks, int bf16, void* Text_, int KLT, int KL, int C, int N) {
  const float* dh1 = rbw;
  const float* drin = rbw + (long)S * 128;
  const float* dh2 = rbw + (long)S * (128 + C2);
  const float* dlog = rbw + (long)S * (128 + C2 + 32);
  if ((int)blockIdx.x < nb_small) {
    __shared__ float red[4][64];
    const int n2 = 32 * 128, n3 = E * 32, nb = 128 + 32 + E;
    const int l = threadIdx.x & 63, u = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + l;
    float acc = 0.f;
    float* dst = nullptr;
    const float *pa = nullptr, *pb = nullptr;                // acc = sum over this stream's frames of pa[s * sa] * (pb ? pb[s * sb] : 1)
    long sa = 0, sb = 0;
    if (i < n2) {
      const int j = i / 128, c = i % 128;
      pa = dh2 + j; sa = 32; pb = rh1 + c; sb = 128;
      if (gW2) dst = gW2 + i;
    } else if (i < n2 + n3) {
      const int k = i - n2, j = k / 32, c = k % 32;
      pa = dlog + j; sa = E; pb = rh2 + c; sb = 32;
      if (gW3) dst = gW3 + k;
    } else if (i < n2 + n3 + nb) {
      const int k = i - n2 - n3;
      if (k < 128) { pa = dh1 + k; sa = 128; if (gb1) dst = gb1 + k; }
      else if (k < 160) { pa = dh2 + (k - 128); sa = 32; if (gb2) dst = gb2 + (k - 128); }
      else { pa = dlog + (k - 160); sa = E; if (gb3) dst = gb3 + (k - 160); }
    }
    if (pa)
      for (int s0 = u; s0 < S; s0 += 4 * 8) {                // eight frames requested before the first add (same order of additions as a plain loop)
        float va[8], vb[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const long sc = min(s0 + 4 * x, S - 1);
          va[x] = pa[sc * sa]; vb[x] = pb ? pb[sc * sb] : 1.f;
        }
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (s0 + 4 * x < S) acc += va[x] * vb[x];
      }
    red[u][l] = acc;
    __syncthreads();
    if (u == 0 && dst) *dst = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    return;
  }
  if ((int)blockIdx.x < nb_small + nb_c) {
    const long total = (long)S * C;
    const float invN = 1.f / (float)N;
    for (long idx = (long)(blockIdx.x - nb_small) * 256 + threadIdx.x; idx < total; idx += (long)nb_c * 256) {
      const int s = (int)(idx / C), i = (int)(idx % C);
      const long o = ((long)s * KLT + KL + 1) * C + i;
      const float v = drin[(long)s * C2 + i] * invN;
      if (bf16) ((unsigned short*)Text_)[o] = f2bf(v); else ((float*)Text_)[o] = v;
    }
    return;
  }
  const long per = 128L * C2;
  const int nb_w = (int)gridDim.x - nb_small - nb_c;
  for (long idx = ((long)(blockIdx.x - nb_small - nb_c) * 256 + threadIdx.x) * 4; idx < per; idx += (long)nb_w * 1024) {
    f32x4_t acc = *(const f32x4_t*)(slabs + idx);
    for (int k = 1; k < ks; ++k) { const f32x4_t b = *(const f32x4_t*)(slabs + (long)k * per + idx); acc += b; }
    *(f32x4_t*)(gW1 + idx) = acc;
  }
}
int k_router_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads,
                 const float* lb_grad, hipStream_t st) {
  ProfScope ps_("k_router_bwd", 0.0, 0.0, st);
  const Dims& d = pl.d;
  RouterBwdArgs a{prm.r0_w, prm.r2_w, prm.r4_w, 2 * d.C, d.E, d.S, d.lb_loss, lb_grad};
  hipLaunchKernelGGL(kk_router_bwd_a, dim3(d.S), dim3(256), 0, st, a, (const float*)(saved + pl.o_probs),
                     (const float*)(scratch + pl.o_dp), (const float*)(saved + pl.o_rh1), (const float*)(saved + pl.o_rh2),
                     (float*)(scratch + pl.o_rbw));
  float* rbw = (float*)(scratch + pl.o_rbw);
  float* dh1 = rbw;
  float* drin = rbw + (long)d.S * 128;
  {   // drin = dh1 W1   (fp32 engine GEMM)
    GemmArgs g;
    g.dtype = GEMM_F32; g.out_dtype = GEMM_F32;
    g.A = dh1; g.B = prm.r0_w; g.C = drin;
    g.M = d.S; g.N = 2 * d.C; g.K = 128; g.lda = 128; g.b_layout = MN_MAJOR; g.ldb = 2L * d.C; g.sCi = 2L * d.C; g.tile = 64;
    AVMOE_TRY(launch_gemm(g, st));
  }
  int ks = 1;
  if (grads.r0_w) {   // dW1 = dh1^T rin ; split over the frames, the slabs are added by the last launch (no reduce pass)
    GemmArgs g;
    g.dtype = GEMM_F32; g.out_dtype = GEMM_F32;
    g.A = dh1; g.B = saved + pl.o_rin; g.C = grads.r0_w;
    g.M = 128; g.N = 2 * d.C; g.K = d.S; g.a_layout = g.b_layout = MN_MAJOR; g.lda = 128; g.ldb = 2L * d.C; g.sCi = 2L * d.C;
    g.tile = 64; g.slabs = (float*)(scratch + pl.o_slabs); g.ksplit = choose_ksplit(g, slab_floats(d));
    g.keep_slabs = 1;
    ks = g.ksplit;
    AVMOE_TRY(launch_gemm(g, st));
  }
  const int nb_small = cdiv(32 * 128 + d.E * 32 + 160 + d.E, 64);
  const int nb_c = (int)grid1db((long)d.S * d.C, 512);
  const int nb_w = (grads.r0_w && ks > 1) ? (int)grid1db(128L * 2 * d.C / 4, 512) : 0;
  hipLaunchKernelGGL(kk_router_bwd_fin, dim3((unsigned)(nb_small + nb_c + nb_w)), dim3(256), 0, st, d.S, d.E, 2 * d.C, nb_small, nb_c,
                     (const float*)(scratch + pl.o_rbw), (const float*)(saved + pl.o_rh1), (const float*)(saved + pl.o_rh2),
                     grads.r0_w, grads.r0_b, grads.r2_w, grads.r2_b, grads.r4_w, grads.r4_b, (const float*)(scratch + pl.o_slabs), ks,
                     d.bf16, (void*)(saved + pl.o_Text), d.KLT, d.KL, d.C, d.N);
  AVMOE_CHECK_LAUNCH("router_bwd");
  return OK;
}

// ---------------------------------------------------------------------------------------------
// row softmax backward: dl = a * (da - sum(a * da)); optional transposed copy out_t[n][row_in_group]
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) kk_softmax_rows_bwd(const void* a_, const float* da, long rows, int n, int ld, void* out_,
                                                           void* outT_, int grp, int ld_t) {
  const T* a = (const T*)a_;
  T* out = (T*)out_; T* outT = (T*)outT_;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float dot = 0.f;
    for (int j = lane; j < n; j += 64) dot += ldT<T>(a, row * ld + j) * da[row * ld + j];
    dot = wave_sum(dot);
    for (int j = lane; j < ld; j += 64) {
      const float v = j < n ? ldT<T>(a, row * ld + j) * (da[row * ld + j] - dot) : 0.f;
      stT<T>(out, row * ld + j, v);
      if (outT && j < n) stT<T>(outT, ((row / grp) * n + j) * ld_t + (row % grp), v);
    }
  }
}
// the same with the row in registers (round 5): NV 4-entry vectors per lane (rows of up to 256 NV entries, ld % 4 == 0), every load
// unconditional and in flight at once (clamped offset, entries beyond n zeroed afterwards), ONE pass over a and da.  The loops above stayed
// rolled -- a two- and a four-byte load, a wait, an FMA per entry: 2 x 16 memory round trips per 1024-entry row.
template <typename T, int NV>
__global__ void __launch_bounds__(256) kk_softmax_rows_bwd_reg(const void* a_, const float* da, long rows, int n, int ld, void* out_,
                                                               void* outT_, int grp, int ld_t) {
  const T* a = (const T*)a_;
  T* out = (T*)out_; T* outT = (T*)outT_;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float4 av[NV], dv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int j = 4 * lane + 256 * k, jc = j < ld ? j : 0;
      av[k] = ld4T<T>(a, row * ld + jc);
      dv[k] = *(const float4*)(da + row * ld + jc);
    }
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int j = 4 * lane + 256 * k;
      float* ae = (float*)&av[k];
      const float* de = (const float*)&dv[k];
#pragma unroll
      for (int x = 0; x < 4; ++x) { if (j + x >= n) ae[x] = 0.f; dot += ae[x] * (j + x < n ? de[x] : 0.f); }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int j = 4 * lane + 256 * k;
      if (j >= ld) continue;
      const float* ae = (const float*)&av[k];
      const float* de = (const float*)&dv[k];
      float4 v;
      float* ve = (float*)&v;
#pragma unroll
      for (int x = 0; x < 4; ++x) ve[x] = j + x < n ? ae[x] * (de[x] - dot) : 0.f;
      st4T<T>(out, row * ld + j, v);
      if (outT) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          if (j + x < n) stT<T>(outT, ((row / grp) * n + j + x) * ld_t + (row % grp), ve[x]);
      }
    }
  }
}
int k_softmax_rows_bwd(int bf16, const void* a, const float* da, long rows, int n, int ld, void* out_dl, void* out_t, int grp,
                       int ldT, hipStream_t st) {
  ProfScope ps_("k_softmax_rows_bwd", 0.0, 0.0, st);
  if (rows <= 0) return OK;
  if (ld % 4 == 0 && ld <= 4096 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)da % 16) == 0 && ((uintptr_t)out_dl % 16) == 0) {
    const dim3 grid((unsigned)std::min<long>((rows + 3) / 4, 8192));
    if (ld <= 1024) {
      if (bf16) hipLaunchKernelGGL((kk_softmax_rows_bwd_reg<__bf16, 4>), grid, dim3(256), 0, st, a, da, rows, n, ld, out_dl, out_t, grp, ldT);
      else hipLaunchKernelGGL((kk_softmax_rows_bwd_reg<float, 4>), grid, dim3(256), 0, st, a, da, rows, n, ld, out_dl, out_t, grp, ldT);
    } else {
      if (bf16) hipLaunchKernelGGL((kk_softmax_rows_bwd_reg<__bf16, 16>), grid, dim3(256), 0, st, a, da, rows, n, ld, out_dl, out_t, grp, ldT);
      else hipLaunchKernelGGL((kk_softmax_rows_bwd_reg<float, 16>), grid, dim3(256), 0, st, a, da, rows, n, ld, out_dl, out_t, grp, ldT);
    }
    AVMOE_CHECK_LAUNCH("softmax_rows_bwd");
    return OK;
  }
  DISPATCH_T(bf16, kk_softmax_rows_bwd, dim3((unsigned)std::min<long>((rows + 3) / 4, 8192)), dim3(256), 0, st, a, da, rows, n, ld,
             out_dl, out_t, grp, ldT);
  AVMOE_CHECK_LAUNCH("softmax_rows_bwd");
  return OK;
}

// ---------------------------------------------------------------------------------------------
// hop-1 backward glue
// ---------------------------------------------------------------------------------------------
// dT += dtbar / C ; split into the cross-modal block dTy (T, Kcyb rows: last = dm2) and the latent-on-X block dTx (T);
// dabx[s][kc'] = dTy[s][kc'] . rw
template <typename T>
__global__ void __launch_bounds__(256) kk_finish_dT(const float* dT, const float* dtbar, const float* rbw_drin, const float* rw,
                                                    void* dTy_, void* dTx_, float* dabx, int S, int KL, int Kcy, int Kcyb, int Kcx,
                                                    int C) {
  T* dTy = (T*)dTy_; T* dTx = (T*)dTx_;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long rows = (long)S * (Kcyb + Kcx);
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const int s = (int)(r / (Kcyb + Kcx)), q = (int)(r % (Kcyb + Kcx));
    float dot = 0.f;
    const bool vec = (C & 3) == 0;                           // 16-byte loads, 8 / 16-byte stores
    if (q < Kcyb) {
      T* dst = dTy + ((long)s * Kcyb + q) * C;
      const float* src = q < Kcy ? dT + ((long)s * KL + q) * C : rbw_drin + (long)s * 2 * C + C;      // (row Kcy: dm2)
      const float add = q < Kcy ? dtbar[(long)s * KL + q] / (float)C : 0.f;
      if (vec) {
        for (int c = 4 * lane; c < C; c += 256) {
          const float4 x = *(const float4*)(src + c), w = *(const float4*)(rw + c);
          const float4 v = make_float4(roundTb<T>(x.x + add), roundTb<T>(x.y + add), roundTb<T>(x.z + add), roundTb<T>(x.w + add));
          st4T<T>(dst, c, v);
          dot += (v.x * w.x + v.y * w.y) + (v.z * w.z + v.w * w.w);
        }
      } else {
        for (int c = lane; c < C; c += 64) {
          const float v = roundTb<T>(src[c] + add);
          stT<T>(dst, c, v);
          dot += v * rw[c];
        }
      }
      dot = wave_sum(dot);
      if (lane == 0) dabx[(long)s * Kcyb + q] = dot;
    } else {
      const int kx = q - Kcyb;
      T* dst = dTx + ((long)s * Kcx + kx) * C;
      const float* src = dT + ((long)s * KL + Kcy + kx) * C;
      const float add = dtbar[(long)s * KL + Kcy + kx] / (float)C;
      if (vec) {
        for (int c = 4 * lane; c < C; c += 256) {
          const float4 x = *(const float4*)(src + c);
          st4T<T>(dst, c, make_float4(x.x + add, x.y + add, x.z + add, x.w + add));
        }
      } else {
        for (int c = lane; c < C; c += 64) stT<T>(dst, c, src[c] + add);
      }
    }
  }
}
// Two independent sums over the frames in ONE launch:
//   blocks [0, nb0): dT0[kc][c] = sum_s (dT[s][kc][c] + dtbar[s][kc] / C)   block = 32 columns x 8 frame streams, eight frames of a
//                    stream requested before the first add (a plain loop is one global round trip per frame), combined through LDS in
//                    a fixed order
//   the others:      drw[c] = sum_r dTy[r][c] abx[r] ; dbf[c] = sum_r dTy[r][c]   over the S*Kcyb rows r, first stage: block (channel
//                    tile cx, chunk) -> rowpart[chunk][2][C] (a sum over the chunks follows); eight rows in flight
template <typename T>
__global__ void __launch_bounds__(256) kk_dT0_dTy(const float* dT, const float* dtbar, float* dT0, int S, int KL, int C, int nb0,
                                                  const void* dTy_, const void* BmX_, const float* scal, float* rowpart, long rows,
                                                  int rows_per_chunk, int Kcy, int Kcyb, int Mb, int M) {
  if ((int)blockIdx.x < nb0) {
    __shared__ float red[8][32];
    const long n1 = (long)KL * C;
    const int l = threadIdx.x & 31, u = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + l;
    double acc = 0.0;
    if (i < n1) {
      const int kc = (int)(i / C);
      const float ic = 1.f / (float)C;
      for (int s0 = u; s0 < S; s0 += 8 * 8) {
        float v[8], w[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const int s = min(s0 + 8 * x, S - 1);              // (clamped: the loads stay unconditional, the adds are masked)
          v[x] = dT[(long)s * n1 + i]; w[x] = dtbar[(long)s * KL + kc];
        }
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (s0 + 8 * x < S) acc += v[x] + w[x] * ic;
      }
    }
    red[u][l] = (float)acc;
    __syncthreads();
    if (u == 0 && i < n1) dT0[i] = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) + ((red[4][l] + red[5][l]) + (red[6][l] + red[7][l]));
    return;
  }
  const T* dTy = (const T*)dTy_; const T* BmX = (const T*)BmX_;
  const int ncx = (C + 255) / 256;
  const int bx = ((int)blockIdx.x - nb0) % ncx, by = ((int)blockIdx.x - nb0) / ncx;
  const int c = bx * 256 + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)by * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float drw = 0.f, dbf = 0.f;
  const float s0 = scal[0];
  for (long rb = r0; rb < r1; rb += 8) {
    float v[8], ab[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const long r = min(rb + x, rows - 1);
      v[x] = ldT<T>(dTy, r * C + c);
      ab[x] = ldT<T>(BmX, r * Mb + M);
    }
#pragma unroll
    for (int x = 0; x < 8; ++x)
      if (rb + x < r1) { const float a = (int)((rb + x) % Kcyb) < Kcy ? ab[x] : s0; drw += v[x] * a; dbf += v[x]; }
  }
  rowpart[((long)by * 2 + 0) * C + c] = drw;
  rowpart[((long)by * 2 + 1) * C + c] = dbf;
}
// ONE launch: blocks [0, ncast): dBmT = T([dBm | dabx | 0]), the (frame, latent row) rows flattened -- four consecutive entries per thread
// (16-byte loads where the row length allows), grid-stride;  the others: the two sums over the frames  dwbar[m] = sum_s dBm[s][Kcy][m]
// and  dbcbar = sum_s dabx[s][Kcy]  -- 16 outputs x 16 frame streams per block, every stream's loads requested before the first add
// (the walk over the frames is a chain of global round trips otherwise), combined through LDS in a fixed order
template <typename T>
__global__ void __launch_bounds__(256) kk_prep_dBm(const float* dBm, const float* dabx, void* dBmT_, float* dvec_wbar, int rows, int S, int Kcy,
                                                   int Kcyb, int M, int Mb, int ncast) {
  T* dBmT = (T*)dBmT_;
  if ((int)blockIdx.x < ncast) {
    const long nq = (long)rows * (Mb / 4);                   // Mb % 4 == 0 (moe_plan.cpp: padded to 8)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long)ncast * 256) {
      const long row = i / (Mb / 4);
      const int m = (int)(i - row * (Mb / 4)) * 4;
      const int q = (int)(row % Kcyb);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < Kcy) {
        if (m + 3 < M) v = *(const float4*)(dBm + row * Mb + m);
        else {
          const float ab = dabx[row];
          float* e = (float*)&v;
#pragma unroll
          for (int x = 0; x < 4; ++x) e[x] = m + x < M ? dBm[row * Mb + m + x] : (m + x == M ? ab : 0.f);
        }
      }
      st4T<T>(dBmT, row * Mb + m, v);
    }
    return;
  }
  __shared__ float red[16][16];
  const int l = threadIdx.x & 15, u = threadIdx.x >> 4;
  const int m = ((int)blockIdx.x - ncast) * 16 + l;            // m < M: dwbar[m] ; m == Mb: dbcbar (dvec layout: [dwbar (Mb) | dbcbar])
  double acc = 0.0;
  if (m < M || m == Mb) {
    const float* p = m < M ? dBm + (long)Kcy * Mb + m : dabx + Kcy;
    const long st = m < M ? (long)Kcyb * Mb : (long)Kcyb;
    for (int s0 = u; s0 < S; s0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) v[x] = s0 + 16 * x < S ? p[(long)(s0 + 16 * x) * st] : 0.f;
#pragma unroll
      for (int x = 0; x < 8; ++x) acc += v[x];
    }
  }
  red[u][l] = (float)acc;
  __syncthreads();
  if (u == 0 && (m < M || m == Mb)) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][l];
    dvec_wbar[m] = t;
  }
}
// dqr[kc] = sum_{s,n} dL1 bc[n] ; dqb[kc] = sum_{s,n} dL1     (partials per (s,kc) row, then over s)
template <typename T>
__global__ void __launch_bounds__(256) kk_dqrqb_part(const void* dL1_, const float* bc, float* part, long rows, int n, int ld) {
  const T* dL1 = (const T*)dL1_;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float a = 0.f, b = 0.f;
    for (int j = lane; j < n; j += 64) { const float v = ldT<T>(dL1, row * ld + j); a += v * bc[j]; b += v; }
    a = wave_sum(a); b = wave_sum(b);
    if (lane == 0) { part[row] = a; part[rows + row] = b; }
  }
}
// final assembly of the remap / token parameter gradients
struct Hop1FinArgs { W16 gtok; int e_of_lat[MAX_E]; int S, N, M, Mk, Mb, C, Cy, K, Kp, KL, Kcy, Kcyb; };
// drw[c] += sum_kc T0[kc][c] dqr[kc] ; dbf[c] += sum_kc T0[kc][c] dqb[kc]      (in place in dvec; thread per channel)
// (a block owns 64 channels; its four waves take every fourth latent row and are combined through LDS in a fixed order)
// dqr / dqb themselves are sums over the frames of the per-(frame, row) partials `part` ([2][S * Kcyb]): every block adds them up
// for itself (thread = latent row x 4 frame streams, fixed order -- they are S x Kcyb numbers), block 0 also leaves them in dqp_fin
template <typename T>
__global__ void __launch_bounds__(256) kk_hop1_vec(const void* T0T_, const float* part, float* dqp_fin, float* dvec, int C, int Kcy, int Kcyb, int S) {
  const T* T0T = (const T*)T0T_;
  __shared__ float red[2][4][64];
  extern __shared__ float s_q[];                     // dqr [Kcyb] | dqb [Kcyb]
  const int cl = threadIdx.x & 63, pt = threadIdx.x >> 6;
  const long rows = (long)S * Kcyb;
  for (int k0 = 0; k0 < Kcyb; k0 += 128) {           // 128 latent rows at a time: rows k0 + cl and k0 + 64 + cl, their loads in flight together
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};            // (sums of both signs over all frames: accumulated in double)
    for (int s0 = pt; s0 < S; s0 += 4 * 8) {           // eight frames of this stream per trip: 32 loads requested before the first add
      float v[2][2][8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kc = min(k0 + 64 * h + cl, Kcyb - 1);   // (clamped: unconditional loads, masked adds)
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const long o = (long)min(s0 + 4 * x, S - 1) * Kcyb + kc;
          v[h][0][x] = part[o]; v[h][1][x] = part[rows + o];
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (s0 + 4 * x < S) { acc[h][0] += v[h][0][x]; acc[h][1] += v[h][1][x]; }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kc = k0 + 64 * h + cl;
      red[0][pt][cl] = (float)acc[h][0]; red[1][pt][cl] = (float)acc[h][1];
      __syncthreads();
      if (pt == 0 && kc < Kcyb) {
        const float q0 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        const float q1 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
        s_q[kc] = q0; s_q[Kcyb + kc] = q1;
        if (blockIdx.x == 0) { dqp_fin[kc] = q0; dqp_fin[Kcyb + kc] = q1; }
      }
      __syncthreads();
    }
  }
  const int c = blockIdx.x * 64 + cl;
  float a0 = 0.f, a1 = 0.f;
  if (c < C)
    for (int kb = pt; kb < Kcy; kb += 4 * 8) {             // eight rows of T0 in flight
      float t[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) t[x] = ldT<T>(T0T, (long)min(kb + 4 * x, Kcy - 1) * C + c);
#pragma unroll
      for (int x = 0; x < 8; ++x)
        if (kb + 4 * x < Kcy) { a0 += t[x] * s_q[kb + 4 * x]; a1 += t[x] * s_q[Kcyb + kb + 4 * x]; }
    }
  red[0][pt][cl] = a0; red[1][pt][cl] = a1;
  __syncthreads();
  if (pt == 0 && c < C) {
    dvec[c] += (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    dvec[C + c] += (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  }
}
// final assembly of the remap / token parameter gradients (pure elementwise; grid.y selects the tensor)
template <typename T>
__global__ void __launch_bounds__(256) kk_hop1_finalize(Hop1FinArgs a, const float* dWcK, const float* dWf, const float* dvec, const float* dT0,
                                                        const float* dqp_fin, const float* rw, const float* bf, float* gWc, float* gbc,
                                                        float* gWf, float* gbf) {
  const float* drw = dvec; const float* dbf = dvec + a.C; const float* dwbar = dvec + 2 * a.C;
  const float* dqr = dqp_fin; const float* dqb = dqp_fin + a.Kcyb;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.y == 0) {                             // conv_adapter weight (N x M) and bias
    if (i < (unsigned)a.N * a.M) {
      const unsigned n = i / a.M, m = i - n * a.M;
      if (gWc) gWc[i] = dWcK[(long)n * a.Mk + m] + dwbar[m] / (float)a.N;
    } else if (i < (unsigned)a.N * a.M + a.N) {
      const unsigned n = i - (unsigned)a.N * a.M;
      if (gbc) gbc[n] = dWcK[(long)n * a.Mk + a.M] + dvec[2 * a.C + a.Mb] / (float)a.N;
    }
  } else if (blockIdx.y == 1) {                      // fc weight (C x Cy) and bias:  rw = Wf 1  =>  every column of row c gets drw[c]
    if (i < (unsigned)a.C * a.Cy) {
      const unsigned c = i / a.Cy;
      if (gWf) gWf[i] = dWf[i] + drw[c];
    } else if (i < (unsigned)a.C * a.Cy + a.C) {
      const unsigned c = i - (unsigned)a.C * a.Cy;
      if (gbf) gbf[c] = dbf[c];
    }
  } else {                                           // latent tokens
    if (i < (unsigned)a.KL * a.C) {
      const unsigned kc = i / a.C, c = i - kc * a.C;
      float v = dT0[i];
      if ((int)kc < a.Kcy) v += dqr[kc] * rw[c] + dqb[kc] * bf[c];
      float* gt = a.gtok.p[a.e_of_lat[kc / a.Kp]];
      if (gt && (int)(kc % a.Kp) < a.K) gt[(long)(kc % a.Kp) * a.C + c] = v;
    }
  }
}

// down projection / ln_before gradients from dWt (+ dwsum, ddconst)
struct DownBwdArgs { P16 down, lnbw, lnbb; W16 gdown, glnbw, glnbb; int E, g, dg, dgp, Cg, DZ, ln_before; };
__global__ void __launch_bounds__(256) kk_down_bwd(DownBwdArgs a, const float* dWt, const float* dsm) {
  // block = 64 channels x 4 bottleneck-row streams of one (expert, group); the two LayerNorm sums are combined through LDS
  __shared__ float red[2][4][64];
  const int l = threadIdx.x & 63, u = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l, gi = blockIdx.y % a.g, e = blockIdx.y / a.g;
  const bool on = c < a.Cg;
  const float gb = (a.ln_before && on) ? a.lnbw.p[e][gi * a.Cg + c] : 1.f;
  const float bb = (a.ln_before && on) ? a.lnbb.p[e][gi * a.Cg + c] : 0.f;
  float dgb = 0.f, dbb = 0.f;
  if (on) {
    // batches of four rows: every load of a batch in flight before its stores (as one loop -- load, store, next row -- the store to the
    // gradient, which the compiler cannot tell apart from the inputs, kept each row's loads behind the previous row's store: a memory
    // round trip per row)
    const float* dsm6 = a.ln_before ? dsm + 6 * a.DZ : nullptr;
    const float* dsm5 = a.ln_before ? dsm + 5 * a.DZ : nullptr;
    for (int jp0 = u; jp0 < a.dg; jp0 += 16) {
      float dwt[4], ddc[4], wd[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int jp = min(jp0 + 4 * b, a.dg - 1), row = (gi * a.E + e) * a.dgp + jp;
        dwt[b] = dWt[(long)row * a.Cg + c];
        ddc[b] = 0.f;
        if (a.ln_before) { dwt[b] += dsm6[row]; ddc[b] = dsm5[row]; }          // (a.ln_before: block-uniform)
        wd[b] = a.down.p[e][(long)(gi * a.dg + jp) * a.Cg + c];
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int jp = jp0 + 4 * b;
        if (jp < a.dg) {
          if (a.gdown.p[e]) a.gdown.p[e][(long)(gi * a.dg + jp) * a.Cg + c] = dwt[b] * gb + ddc[b] * bb;
          dgb += dwt[b] * wd[b]; dbb += ddc[b] * wd[b];
        }
      }
    }
  }
  red[0][u][l] = dgb; red[1][u][l] = dbb;
  __syncthreads();
  if (u == 0 && on && a.ln_before) {
    if (a.glnbw.p[e]) a.glnbw.p[e][gi * a.Cg + c] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
    if (a.glnbb.p[e]) a.glnbb.p[e][gi * a.Cg + c] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
  }
}

// ---------------------------------------------------------------------------------------------
// AVVP N x N block helpers
// ---------------------------------------------------------------------------------------------
// per token: sum xr, sum xr^2, x . xr    (one wave per row)
template <typename T>
__global__ void __launch_bounds__(256) kk_xrstats(const void* X_, const void* R_, long rows, int C, float* out) {
  const T* X = (const T*)X_; const T* R = (const T*)R_;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float x = ldT<T>(X, row * C + c), v = ldT<T>(R, row * C + c);
      s0 += v; s1 += v * v; s2 += x * v;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { out[row] = s0; out[rows + row] = s1; out[2 * rows + row] = s2; }
  }
}
// the same for C % 4 == 0: 16 lanes per row, four consecutive entries per lane and access (rows of 96 - 192 channels: a wave per row
// leaves most of its lanes idle)
template <typename T>
__global__ void __launch_bounds__(256) kk_xrstats_v4(const void* X_, const void* R_, long rows, int C, float* out) {
  const T* X = (const T*)X_; const T* R = (const T*)R_;
  const int l = threadIdx.x & 15;
  for (long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4); row < rows; row += (long)gridDim.x * 16) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = 4 * l; c < C; c += 64) {
      const float4 x = ld4T<T>(X, row * C + c), v = ld4T<T>(R, row * C + c);
      s0 += (v.x + v.y) + (v.z + v.w); s1 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); s2 += (x.x * v.x + x.y * v.y) + (x.z * v.z + x.w * v.w);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (l == 0) { out[row] = s0; out[rows + row] = s1; out[2 * rows + row] = s2; }
  }
}
int k_xrstats(const Plan& pl, const void* X, char* saved, int slot, hipStream_t st) {
  ProfScope ps_("k_xrstats", 0.0, 0.0, st);
  const Dims& d = pl.d;
  if (d.C % 4 == 0)
    DISPATCH_T(d.bf16, kk_xrstats_v4, dim3((unsigned)std::min<long>((d.NT + 15) / 16, 16384)), dim3(256), 0, st, X,
               (const void*)(saved + pl.o_xr + (size_t)slot * d.NT * d.C * d.esz), (long)d.NT, d.C, (float*)(saved + pl.o_sxr) + (size_t)slot * 3 * d.NT);
  else
    DISPATCH_T(d.bf16, kk_xrstats, dim3((unsigned)std::min<long>((d.NT + 3) / 4, 8192)), dim3(256), 0, st, X,
               (const void*)(saved + pl.o_xr + (size_t)slot * d.NT * d.C * d.esz), (long)d.NT, d.C, (float*)(saved + pl.o_sxr) + (size_t)slot * 3 * d.NT);
  AVMOE_CHECK_LAUNCH("xrstats");
  return OK;
}
// dxr[t][c] += dsr2[t] * X[t][c] + dsr0[t] ;  dX[t][c] += dsr2[t] * xr[t][c]  (- dxr[t][c] when xr = f(X) - X REPLACES the
// input: the direct route of that expert's input gradient is then not a route at all -- AVS "v1")
template <typename T>
__global__ void kk_nxn_axpy(const void* X_, const void* R_, const float* dsr, long NT, int C, void* dxr_, void* dX_, int replaces) {
  const T* X = (const T*)X_; const T* R = (const T*)R_;
  T* dxr = (T*)dxr_; T* dX = (T*)dX_;
  const long total = NT * C;
  if ((C & 3) == 0) {                                        // four consecutive entries of a row per thread and access
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 1024) {
      const long t = i / C;
      const float d0 = dsr[t], d2 = dsr[2 * NT + t];
      const float4 a = ld4T<T>(dxr, i), x = ld4T<T>(X, i), o = ld4T<T>(dX, i), r = ld4T<T>(R, i);
      const float4 g = make_float4(roundTb<T>(a.x + d2 * x.x + d0), roundTb<T>(a.y + d2 * x.y + d0), roundTb<T>(a.z + d2 * x.z + d0), roundTb<T>(a.w + d2 * x.w + d0));
      st4T<T>(dxr, i, g);
      const float4 sub = replaces ? g : make_float4(0.f, 0.f, 0.f, 0.f);
      st4T<T>(dX, i, make_float4(o.x + d2 * r.x - sub.x, o.y + d2 * r.y - sub.y, o.z + d2 * r.z - sub.z, o.w + d2 * r.w - sub.w));
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long t = i / C;
    const float d0 = dsr[t], d2 = dsr[2 * NT + t];
    const float g = roundTb<T>(ldT<T>(dxr, i) + d2 * ldT<T>(X, i) + d0);
    stT<T>(dxr, i, g);
    stT<T>(dX, i, ldT<T>(dX, i) + d2 * ldT<T>(R, i) - (replaces ? g : 0.f));
  }
}
__global__ void kk_merge_gather(W16 gdown, W16 gup, const float* __restrict__ gWd, const float* __restrict__ gWu, int E, int d, int C, int mg) {
  const int dgt = d / mg, Cgt = C / mg;
  const long per = (long)d * Cgt, total = (long)E * per;          // elements of one grouped weight: d * C / mg == C * d / mg
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int e = (int)(i / per);
    const long r = i - (long)e * per;
    if (gdown.p[e]) {                                      // (d, C / mg): row j reads its group's channel block
      const int j = (int)(r / Cgt), cc = (int)(r % Cgt);
      gdown.p[e][r] = gWd[((long)e * d + j) * C + (j / dgt) * Cgt + cc];
    }
    if (gup.p[e]) {                                        // (C, d / mg): row c reads its group's bottleneck block
      const int c = (int)(r / dgt), jj = (int)(r % dgt);
      gup.p[e][r] = gWu[((long)e * C + c) * d + (c / Cgt) * dgt + jj];
    }
  }
}
int k_merge_gather(const Plan& pl, char* scratch, const avmoe_moe_ptrs& grads, hipStream_t st) {
  const Dims& d = pl.d;
  W16 gd, gu;
  for (int e = 0; e < MAX_E; ++e) { gd.p[e] = grads.e[e].down_w; gu.p[e] = grads.e[e].up_w; }
  const long total = (long)d.E * d.d * (d.C / d.mg);
  hipLaunchKernelGGL(kk_merge_gather, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, gd, gu,
                     (const float*)(scratch + pl.o_gWd), (const float*)(scratch + pl.o_gWu), d.E, d.d, d.C, d.mg);
  AVMOE_CHECK_LAUNCH("merge_gather");
  return OK;
}
avmoe_moe_ptrs merged_grads(const Plan& pl, const avmoe_moe_ptrs& grads, char* scratch) {
  avmoe_moe_ptrs p = grads;
  const Dims& d = pl.d;
  for (int e = 0; e < d.E; ++e) {        // the kernels write dense gradients here; k_merge_gather hands the diagonal blocks to the caller
    if (grads.e[e].down_w) p.e[e].down_w = (float*)(scratch + pl.o_gWd) + (size_t)e * d.d * d.C;
    if (grads.e[e].up_w) p.e[e].up_w = (float*)(scratch + pl.o_gWu) + (size_t)e * d.d * d.C;
  }
  return p;
}

// N x N block, backward: y = att dxr (fp32, rows x C) is both the direct gradient term (dX += y) and, dotted with X, the row term of the
// softmax backward (sum_j att_ij d att_ij = X_i . y_i).  16 lanes per row.
template <typename T>
__global__ void __launch_bounds__(256) kk_nxn_rowdot(const void* X_, const float* __restrict__ y, long rows, int C, void* dX_, float* __restrict__ rowdot) {
  const T* X = (const T*)X_; T* dX = (T*)dX_;
  const int l = threadIdx.x & 15;
  for (long r = (long)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (long)gridDim.x * 16) {
    float acc = 0.f;
    if ((C & 3) == 0) {                                      // four consecutive entries per lane and access
      for (int c = 4 * l; c < C; c += 64) {
        const float4 yv = *(const float4*)(y + r * C + c), x = ld4T<T>(X, r * C + c), o = ld4T<T>(dX, r * C + c);
        acc += (x.x * yv.x + x.y * yv.y) + (x.z * yv.z + x.w * yv.w);
        st4T<T>(dX, r * C + c, make_float4(o.x + yv.x, o.y + yv.y, o.z + yv.z, o.w + yv.w));
      }
    } else {
      for (int c = l; c < C; c += 16) {
        const float yv = y[r * C + c];
        acc += ldT<T>(X, r * C + c) * yv;
        stT<T>(dX, r * C + c, ldT<T>(dX, r * C + c) + yv);
      }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (l == 0) rowdot[r] = acc;
  }
}
int k_nxn_rowdot(int bf16, const void* X, const float* y, long rows, int C, void* dX, float* rowdot, hipStream_t st) {
  ProfScope ps_("k_nxn_rowdot", 0.0, 0.0, st);
  if (rows <= 0) return OK;
  DISPATCH_T(bf16, kk_nxn_rowdot, dim3((unsigned)std::min<long>((rows + 15) / 16, 16384)), dim3(256), 0, st, X, y, rows, C, dX, rowdot);
  AVMOE_CHECK_LAUNCH("nxn_rowdot");
  return OK;
}

int k_nxn_axpy(const Plan& pl, const void* X, char* saved, char* scratch, void* dX, int slot, int replaces, hipStream_t st) {
  ProfScope ps_("k_nxn_axpy", 0.0, 0.0, st);
  const Dims& d = pl.d;
  DISPATCH_T(d.bf16, kk_nxn_axpy, dim3(grid1db((long)d.NT * d.C, 16384)), dim3(256), 0, st, X,
             (const void*)(saved + pl.o_xr + (size_t)slot * d.NT * d.C * d.esz), (const float*)(scratch + pl.o_dsr) + (size_t)slot * 3 * d.NT,
             (long)d.NT, d.C, (void*)(scratch + pl.o_dxr), dX, replaces);
  AVMOE_CHECK_LAUNCH("nxn_axpy");
  return OK;
}

// ---- host wrappers for the glue ---------------------------------------------------------------
int k_finish_dT(const Plan& pl, char* saved, char* scratch, hipStream_t st) {
  const Dims& d = pl.d;
  const float* drin = (const float*)(scratch + pl.o_rbw) + (long)d.S * 128;
  const long rows = (long)d.S * (d.Kcyb + d.Kcx);
  DISPATCH_T(d.bf16, kk_finish_dT, dim3((unsigned)std::min<long>((rows + 3) / 4, 8192)), dim3(256), 0, st,
             (const float*)(scratch + pl.o_dT), (const float*)(scratch + pl.o_dtbar), drin, (const float*)(saved + pl.o_rw),
             (void*)(scratch + pl.o_dTy), (void*)(scratch + pl.o_dTx), (float*)(scratch + pl.o_dabx), d.S, d.KL, d.Kcy, d.Kcyb, d.Kcx, d.C);
  {
    const long nrows = (long)d.S * d.Kcyb;
    const int nchunk = (int)std::min<long>(512, std::max<long>(1, nrows / 8));      // (short serial chains per thread: 8 rows, or nrows / 512 when there are many)
    const int rpc = cdiv(nrows, nchunk);
    const int nb0 = d.KL > 0 ? cdiv((long)d.KL * d.C, 32) : 0;
    DISPATCH_T(d.bf16, kk_dT0_dTy, dim3((unsigned)(nb0 + cdiv(d.C, 256) * nchunk)), dim3(256), 0, st, (const float*)(scratch + pl.o_dT),
               (const float*)(scratch + pl.o_dtbar), (float*)(scratch + pl.o_dT0), d.S, d.KL, d.C, nb0, (const void*)(scratch + pl.o_dTy),
               (const void*)(saved + pl.o_BmX), (const float*)(saved + pl.o_scal), (float*)(scratch + pl.o_rowpart), nrows, rpc,
               d.Kcy, d.Kcyb, d.Mb, d.M);
    AVMOE_TRY(k_colsum_f32((const float*)(scratch + pl.o_rowpart), nchunk, 2 * d.C, 2L * d.C, 1, 0, (float*)(scratch + pl.o_dvec), 0, 1.f, st));
  }
  AVMOE_CHECK_LAUNCH("finish_dT");
  return OK;
}
int k_prep_dBm(const Plan& pl, char* scratch, hipStream_t st) {
  const Dims& d = pl.d;
  const float* dBm = (const float*)(scratch + pl.o_dBm);
  const float* dabx = (const float*)(scratch + pl.o_dabx);
  float* dvec = (float*)(scratch + pl.o_dvec);
  // dwbar (row Kcy of every frame's dBm), zero in the padding m >= M (dvec lies inside the backward's one memset: moe_plan.h) ; dbcbar
  const int rows = d.S * d.Kcyb;
  if (d.Mb % 4) { set_last_error("prep_dBm: padded row length %d not a multiple of 4", d.Mb); return ERR_UNSUPPORTED; }
  const int ncast = (int)std::min<long>(cdiv((long)rows * (d.Mb / 4), 256L), 4096);
  DISPATCH_T(d.bf16, kk_prep_dBm, dim3((unsigned)(ncast + cdiv(d.Mb + 1, 16))), dim3(256), 0, st, dBm, dabx, (void*)(scratch + pl.o_dBmT),
             dvec + 2 * d.C, rows, d.S, d.Kcy, d.Kcyb, d.M, d.Mb, ncast);
  AVMOE_CHECK_LAUNCH("prep_dBm");
  return OK;
}
int k_dqrqb(const Plan& pl, char* scratch, const float* bc, hipStream_t st) {
  ProfScope ps_("k_dqrqb", 0.0, 0.0, st);
  const Dims& d = pl.d;
  const long rows = (long)d.S * d.Kcyb;
  float* part = (float*)(scratch + pl.o_dqp);              // per (frame, row) partials; k_hop1_finalize adds them over the frames
  DISPATCH_T(d.bf16, kk_dqrqb_part, dim3((unsigned)std::min<long>((rows + 3) / 4, 8192)), dim3(256), 0, st,
             (const void*)(scratch + pl.o_dL1), bc, part, rows, d.N, d.Np);
  AVMOE_CHECK_LAUNCH("dqrqb");
  return OK;
}
int k_hop1_finalize(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads, hipStream_t st) {
  ProfScope ps_("k_hop1_finalize", 0.0, 0.0, st);
  const Dims& d = pl.d;
  Hop1FinArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.gtok.p[e] = grads.e[e].my_tokens; a.e_of_lat[e] = d.e_of_lat[e]; }
  a.S = d.S; a.N = d.N; a.M = d.M; a.Mk = d.Mk; a.Mb = d.Mb; a.C = d.C; a.Cy = d.Cy; a.K = d.K; a.Kp = d.Kp; a.KL = d.KL; a.Kcy = d.Kcy; a.Kcyb = d.Kcyb;
  const long big = std::max(std::max((long)d.N * d.M + d.N, (long)d.C * d.Cy + d.C), (long)d.KL * d.C);
  if (big >= (1L << 31)) { set_last_error("hop1_finalize: parameter tensor too large"); return ERR_UNSUPPORTED; }
  float* dqp_fin = (float*)(scratch + pl.o_dqp) + 2L * d.S * d.Kcyb;
  if (d.Kcy > 0)
    DISPATCH_T(d.bf16, kk_hop1_vec, dim3(cdiv(d.C, 64)), dim3(256), (size_t)2 * d.Kcyb * sizeof(float), st, (const void*)(saved + pl.o_T0T),
               (const float*)(scratch + pl.o_dqp), dqp_fin, (float*)(scratch + pl.o_dvec), d.C, d.Kcy, d.Kcyb, d.S);
  DISPATCH_T(d.bf16, kk_hop1_finalize, dim3((unsigned)cdiv(big, 256), 3), dim3(256), 0, st, a, (const float*)(scratch + pl.o_dWcK),
             (const float*)(scratch + pl.o_dWf), (const float*)(scratch + pl.o_dvec), (const float*)(scratch + pl.o_dT0), dqp_fin,
             (const float*)(saved + pl.o_rw), (const float*)prm.fc_b, grads.conv_w, grads.conv_b, grads.fc_w, grads.fc_b);
  AVMOE_CHECK_LAUNCH("hop1_finalize");
  return OK;
}
int k_down_bwd(const Plan& pl, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads, hipStream_t st) {
  ProfScope ps_("k_down_bwd", 0.0, 0.0, st);
  const Dims& d = pl.d;
  DownBwdArgs a;
  for (int e = 0; e < MAX_E; ++e) {
    a.down.p[e] = prm.e[e].down_w; a.lnbw.p[e] = prm.e[e].lnb_w; a.lnbb.p[e] = prm.e[e].lnb_b;
    a.gdown.p[e] = grads.e[e].down_w; a.glnbw.p[e] = grads.e[e].lnb_w; a.glnbb.p[e] = grads.e[e].lnb_b;
  }
  a.E = d.E; a.g = d.g; a.dg = d.dg; a.dgp = d.dgp; a.Cg = d.Cg; a.DZ = d.DZ; a.ln_before = d.ln_before;
  hipLaunchKernelGGL(kk_down_bwd, dim3((unsigned)cdiv(d.Cg, 64), (unsigned)(d.E * d.g)), dim3(256), 0, st, a, (const float*)(scratch + pl.o_dWt),
                     (const float*)(scratch + pl.o_dsm));
  AVMOE_CHECK_LAUNCH("down_bwd");
  return OK;
}

}  // namespace avmoe
