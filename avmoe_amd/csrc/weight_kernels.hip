// Weight-space kernels (C x d sized work): BatchNorm-2 statistics from the d-space moments, the folded output
// weights Bpost, the d x d Gram / sums the LayerNorm-post statistics need, and the chain rule back through those
// folds.  net_trans_v3.py:401-403,430-434 ; names follow oracle/algebra_ref.py (POST weights / BN2 stats).
//
// All of them are tiny in FLOPs; what matters is latency, so every kernel stages the per-(group, expert) d x d
// matrices in LDS, keeps each channel's up-projection row in LDS, and spreads channels over many blocks.
#include "kernels.h"
#include "device_utils.h"
#include "prof.h"
#include <algorithm>

namespace avmoe {

#define DISPATCH_T(bf16, KERN, grid, block, shmem, st, ...)                                   \
  do {                                                                                        \
    if (bf16) hipLaunchKernelGGL((KERN<__bf16>), grid, block, shmem, st, __VA_ARGS__);        \
    else hipLaunchKernelGGL((KERN<float>), grid, block, shmem, st, __VA_ARGS__);              \
  } while (0)

static inline unsigned grid1dw(long n, int cap = 4096) { return (unsigned)std::max<long>(1, std::min<long>((n + 255) / 256, cap)); }

constexpr int GCS = 8;       // channel chunks of the Gram kernels

struct WArgs {
  P16 up, w2, b2, lpw, lpb, gate; W16 rm, rv, gup, gw2, gb2, glpw, glpb, ggate; N16 nbt2;
  int E, g, dg, dgp, Cg, C, KPp, NT, DZ, use_bn, training, ln_post;
  int gate_w;            // Dims::gate_w: Bpost_e carries gate_e ; dgate_e is formed here
  float eps, momentum;
};
static void fill_w(const Dims& d, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs* grads, WArgs* a) {
  for (int e = 0; e < MAX_E; ++e) {
    a->up.p[e] = prm.e[e].up_w; a->w2.p[e] = prm.e[e].bn2_w; a->b2.p[e] = prm.e[e].bn2_b;
    a->lpw.p[e] = prm.e[e].lnp_w; a->lpb.p[e] = prm.e[e].lnp_b; a->rm.p[e] = prm.e[e].bn2_rm; a->rv.p[e] = prm.e[e].bn2_rv;
    a->nbt2.p[e] = prm.e[e].bn2_nbt;
    a->gup.p[e] = grads ? grads->e[e].up_w : nullptr; a->gw2.p[e] = grads ? grads->e[e].bn2_w : nullptr;
    a->gb2.p[e] = grads ? grads->e[e].bn2_b : nullptr; a->glpw.p[e] = grads ? grads->e[e].lnp_w : nullptr;
    a->glpb.p[e] = grads ? grads->e[e].lnp_b : nullptr;
    a->gate.p[e] = prm.e[e].gate; a->ggate.p[e] = grads ? grads->e[e].gate : nullptr;
  }
  a->gate_w = d.use_gate && d.gate_w;
  a->E = d.E; a->g = d.g; a->dg = d.dg; a->dgp = d.dgp; a->Cg = d.Cg; a->C = d.C; a->KPp = d.KPp; a->NT = d.NT; a->DZ = d.DZ;
  a->use_bn = d.use_bn; a->training = d.training; a->ln_post = d.ln_post; a->eps = d.bn_eps; a->momentum = d.bn_momentum;
}

// ---------------------------------------------------------------------------------------------
// BN2 statistics: mo[c] = Wu[c,:] . mz ; E[o^2][c] = Wu[c,:] Szz Wu[c,:]^T    grid (g*E, ceil(Cg/8))
// ---------------------------------------------------------------------------------------------
// block = BS_CH channels x 32 bottleneck lanes: lane jl owns rows j = jl, jl + 32, .. of Szz Wu[c,:]^T; the two sums are
// folded over the lanes in double
constexpr int BS_CH = 8;
__global__ void __launch_bounds__(256) kw_bn2_stats(WArgs a, const float* mz, const float* Szz, float* bn2) {
  extern __shared__ float sm[];
  const int dg = a.dg, dgp = a.dgp, ldm = dgp + 1;
  float* s_S = sm;                     // dgp x ldm
  float* s_m = s_S + dgp * ldm;        // dgp
  float* s_w = s_m + dgp;              // BS_CH x dgp
  const int cb = blockIdx.x, i = cb / a.E, e = cb % a.E;
  const int jl = threadIdx.x & 31, cc = threadIdx.x >> 5;
  const int cl = blockIdx.y * BS_CH + cc;                   // channel inside the group
  const bool on = cl < a.Cg;
  const int c = i * a.Cg + (on ? cl : 0);
  const bool stats = a.use_bn && a.training;
  if (stats && i == 0 && blockIdx.y == 0 && threadIdx.x == 0 && a.nbt2.p[e]) a.nbt2.p[e][0] += 1;      // bn2.num_batches_tracked
  if (stats) {
    for (int k = threadIdx.x; k < dgp * dgp; k += 256) { const int r = k / dgp; s_S[r * ldm + (k - r * dgp)] = Szz[(long)cb * dgp * dgp + k]; }
    for (int k = threadIdx.x; k < dgp; k += 256) s_m[k] = mz[(long)cb * dgp + k];
    for (int j = jl; j < dgp; j += 32) s_w[cc * dgp + j] = (on && j < dg) ? a.up.p[e][(long)c * dg + j] : 0.f;
  }
  __syncthreads();
  float mo = 0.f, rs2 = 1.f, k2 = 1.f, h2 = 0.f;
  if (a.use_bn) {
    float v2;
    if (a.training) {
      const float* wu = s_w + cc * dgp;
      double dmo = 0.0, eo2 = 0.0;
      for (int j = jl; j < dg; j += 32) {
        dmo += (double)wu[j] * s_m[j];
        float row = 0.f;
        for (int l = 0; l < dg; ++l) row += s_S[j * ldm + l] * wu[l];
        eo2 += (double)wu[j] * row;
      }
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) { dmo += __shfl_xor(dmo, o, 64); eo2 += __shfl_xor(eo2, o, 64); }
      mo = (float)dmo;
      const double v = fmax(eo2 - dmo * dmo, 0.0);
      v2 = (float)v;
      const double unb = a.NT > 1 ? v * ((double)a.NT / (a.NT - 1)) : v;
      if (on && jl == 0) {
        a.rm.p[e][c] = (1.f - a.momentum) * a.rm.p[e][c] + a.momentum * mo;
        a.rv.p[e][c] = (1.f - a.momentum) * a.rv.p[e][c] + a.momentum * (float)unb;
      }
    } else { mo = a.rm.p[e][c]; v2 = a.rv.p[e][c]; }
    rs2 = rsqrtf(v2 + a.eps);
    k2 = a.w2.p[e][c] * rs2;
    h2 = a.b2.p[e][c] - mo * k2;
  }
  if (on && jl == 0) {
    const long EC = (long)a.E * a.C, idx = (long)e * a.C + c;
    bn2[idx] = mo; bn2[EC + idx] = rs2; bn2[2 * EC + idx] = k2; bn2[3 * EC + idx] = h2;
  }
}

// thread per (c, k'): Bpost[c][k']
template <typename T>
__device__ __forceinline__ void build_bpost_body(const WArgs& a, const float* bn2, void* Bpost_, int bx, int nbx) {
  T* Bpost = (T*)Bpost_;
  const long total = (long)a.C * a.KPp;
  const long EC = (long)a.E * a.C;
  for (long idx = (long)bx * 256 + threadIdx.x; idx < total; idx += (long)nbx * 256) {
    const int c = (int)(idx / a.KPp), kp = (int)(idx % a.KPp);
    float v = 0.f;
    if (kp < a.E * a.dgp) {
      const int e = kp / a.dgp, jp = kp % a.dgp;
      if (jp < a.dg) {
        const float gp = a.ln_post ? a.lpw.p[e][c] : 1.f;
        v = gp * a.up.p[e][(long)c * a.dg + jp] * bn2[2 * EC + (long)e * a.C + c];
      }
    } else if (kp < a.E * a.dgp + 3 * a.E) {
      const int r = kp - a.E * a.dgp, e = r / 3, w = r % 3;
      const float gp = a.ln_post ? a.lpw.p[e][c] : 1.f;
      const float bp = a.ln_post ? a.lpb.p[e][c] : 0.f;
      const float h2 = bn2[3 * EC + (long)e * a.C + c];
      v = w == 0 ? gp * h2 : (w == 1 ? gp : bp);
    }
    if (a.gate_w && kp < a.E * a.dgp + 3 * a.E) v *= a.gate.p[kp < a.E * a.dgp ? kp / a.dgp : (kp - a.E * a.dgp) / 3][0];      // the expert's gate (net_trans_v3.py:434)
    stT<T>(Bpost, idx, v);
  }
}

// ---------------------------------------------------------------------------------------------
// Channel contractions per (group, expert):  Gram[j][l] = sum_c w[c] U[c][j] U[c][l] ,  v1[j] = sum_c x1[c] U1[c][j] ,
// v2[j] = sum_c x2[c] U[c][j] , s1 = sum x3 , s2 = sum x3^2.       grid (g*E, GCS), partials summed by kw_gram_finish.
//   mode 0 (forward):  U = Wu*k2 (= Wh), w = 1, v1 = usum (x1 = 1, U1 = U), v2 = vh (x2 = h2), s1/s2 = H1/H2 (x3 = h2)
//   mode 1 (backward): U = Wu, w = dv2, v1 = dmz (x1 = dmo, U1 = Wu)
// partial layout per (chunk, cb): [dgp*dgp | dgp | dgp | 2]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void gram_body(const WArgs& a, int mode, const float* bn2, const float* dmodv, float* gpart, int cb, int chunk, int ncb) {
  extern __shared__ float sm[];
  const int dg = a.dg, dgp = a.dgp;
  const int i = cb / a.E, e = cb % a.E;
  const int cc = (a.Cg + GCS - 1) / GCS;
  const int c0 = chunk * cc, c1 = min(a.Cg, c0 + cc), nc = max(0, c1 - c0);
  float* s_U = sm;                     // cc x dgp
  float* s_w = s_U + cc * dgp;         // cc : Gram weight
  float* s_x1 = s_w + cc;              // cc
  float* s_x2 = s_x1 + cc;             // cc
  const long EC = (long)a.E * a.C;
  const long cbase = (long)e * a.C + (long)i * a.Cg;
  for (int k = threadIdx.x; k < nc * dgp; k += 256) {
    const int cl = k / dgp, j = k % dgp;
    float u = 0.f;
    if (j < dg) {
      u = a.up.p[e][((long)i * a.Cg + c0 + cl) * dg + j];
      if (mode == 0) u *= bn2[2 * EC + cbase + c0 + cl];
    }
    s_U[k] = u;
  }
  for (int k = threadIdx.x; k < nc; k += 256) {
    if (mode == 0) { s_w[k] = 1.f; s_x1[k] = 1.f; s_x2[k] = bn2[3 * EC + cbase + c0 + k]; }
    else { s_w[k] = dmodv[EC + cbase + c0 + k]; s_x1[k] = dmodv[cbase + c0 + k]; s_x2[k] = a.gate_w ? dmodv[2 * EC + cbase + c0 + k] : 0.f; }      // (x2: the per-channel dgate terms, summed as s1)
  }
  __syncthreads();
  const int stride = dgp * dgp + 2 * dgp + 2;
  float* out = gpart + ((long)chunk * ncb + cb) * stride;
  for (int pr = threadIdx.x; pr < dgp * dgp; pr += 256) {
    const int j = pr / dgp, l = pr % dgp;
    float acc = 0.f;
    for (int c = 0; c < nc; ++c) acc += s_w[c] * s_U[c * dgp + j] * s_U[c * dgp + l];
    out[pr] = acc;
  }
  for (int j = threadIdx.x; j < dgp; j += 256) {
    float v1 = 0.f, v2 = 0.f;
    for (int c = 0; c < nc; ++c) { const float u = s_U[c * dgp + j]; v1 += s_x1[c] * u; v2 += s_x2[c] * u; }
    out[dgp * dgp + j] = v1; out[dgp * dgp + dgp + j] = v2;
  }
  if (threadIdx.x == 0) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < nc; ++c) { s1 += s_x2[c]; s2 += s_x2[c] * s_x2[c]; }
    out[dgp * dgp + 2 * dgp] = s1; out[dgp * dgp + 2 * dgp + 1] = s2;
  }
}
__global__ void __launch_bounds__(256) kw_gram(WArgs a, int mode, const float* bn2, const float* dmodv, float* gpart) {
  gram_body(a, mode, bn2, dmodv, gpart, blockIdx.x, blockIdx.y, gridDim.x);
}
// forward: the Gram partials (mode 0) and the folded output weights Bpost -- both hang on bn2 only -- in ONE launch
template <typename T>
__global__ void __launch_bounds__(256) kw_gram_bpost(WArgs a, const float* bn2, float* gpart, void* Bpost, int ncb) {
  const int ngram = ncb * GCS;
  if ((int)blockIdx.x < ngram) gram_body(a, 0, bn2, nullptr, gpart, blockIdx.x % ncb, blockIdx.x / ncb, ncb);
  else build_bpost_body<T>(a, bn2, Bpost, (int)blockIdx.x - ngram, (int)gridDim.x - ngram);
}
// sum the GCS partials and scatter into the consumers' layouts
//   mode 0: Gq[cb][dgp*dgp], uvh = [usum (DZ) | vh (DZ) | H1[g*E] | H2[g*E]]
//   mode 1: sdSzz[cb][..] = 2/NT * Gram ,  dsm[2*DZ + cb*dgp + j] = v1 / NT
__global__ void kw_gram_finish(WArgs a, int mode, const float* gpart, float* outG, float* outV) {
  const int dgp = a.dgp, nb = a.g * a.E;
  const int stride = dgp * dgp + 2 * dgp + 2;
  const long total = (long)nb * stride;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cb = (int)(idx / stride), k = (int)(idx % stride);
    float acc = 0.f;
    for (int ch = 0; ch < GCS; ++ch) acc += gpart[((long)ch * nb + cb) * stride + k];
    const float inv = 1.f / (float)a.NT;
    if (k < dgp * dgp) outG[(long)cb * dgp * dgp + k] = mode == 0 ? acc : 2.f * acc * inv;
    else if (k < dgp * dgp + dgp) {
      const int j = k - dgp * dgp;
      if (mode == 0) outV[(long)cb * dgp + j] = acc; else outV[2 * a.DZ + (long)cb * dgp + j] = acc * inv;
    } else if (k < dgp * dgp + 2 * dgp) {
      if (mode == 0) outV[a.DZ + (long)cb * dgp + (k - dgp * dgp - dgp)] = acc;
    } else if (mode == 0) {
      const int w = k - dgp * dgp - 2 * dgp;
      outV[2 * a.DZ + (long)w * nb + cb] = acc;
    } else if (a.gate_w && k == dgp * dgp + 2 * dgp && cb < a.E) {        // mode 1: dgate_e = the s1 sums of the expert's groups, in group order
      float tot = 0.f;
      for (int i = 0; i < a.g; ++i)
        for (int ch = 0; ch < GCS; ++ch) tot += gpart[((long)ch * nb + (long)i * a.E + cb) * stride + k];
      if (a.ggate.p[cb]) a.ggate.p[cb][0] = tot;
    }
  }
}
static int run_gram(const Plan& pl, const WArgs& a, int mode, const float* bn2, const float* dmodv, float* gpart, float* outG,
                    float* outV, hipStream_t st) {
  const Dims& d = pl.d;
  const int cc = cdiv(d.Cg, GCS);
  const size_t sh = (size_t)(cc * d.dgp + 3 * cc) * sizeof(float);
  hipLaunchKernelGGL(kw_gram, dim3(d.g * d.E, GCS), dim3(256), sh, st, a, mode, bn2, dmodv, gpart);
  const long total = (long)d.g * d.E * (d.dgp * d.dgp + 2 * d.dgp + 2);
  hipLaunchKernelGGL(kw_gram_finish, dim3(grid1dw(total)), dim3(256), 0, st, a, mode, (const float*)gpart, outG, outV);
  AVMOE_CHECK_LAUNCH("gram");
  return OK;
}

int k_post_prep(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  ProfScope ps_("k_post_prep", 0.0, 0.0, st);
  const Dims& d = pl.d;
  WArgs a; fill_w(d, prm, nullptr, &a);
  const size_t sh = (size_t)(d.dgp * (d.dgp + 1) + d.dgp + BS_CH * d.dgp) * sizeof(float);
  if (sh > 160 * 1024) { set_last_error("post_prep: bottleneck per group %d needs %zu B of LDS", d.dg, sh); return ERR_UNSUPPORTED; }
  if (sh > 65536 && hipFuncSetAttribute((const void*)kw_bn2_stats, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
    set_last_error("post_prep: LDS attribute"); return ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kw_bn2_stats, dim3(d.g * d.E, cdiv(d.Cg, BS_CH)), dim3(256), sh, st, a, (const float*)(saved + pl.o_mz),
                     (const float*)(saved + pl.o_Szz), (float*)(saved + pl.o_bn2));
  {
    const int ncb = d.g * d.E, cc = cdiv(d.Cg, GCS);
    const size_t shg = (size_t)(cc * d.dgp + 3 * cc) * sizeof(float);
    const int nb_bp = (int)grid1dw((long)d.C * d.KPp, 1024);
    DISPATCH_T(d.bf16, kw_gram_bpost, dim3((unsigned)(ncb * GCS + nb_bp)), dim3(256), shg, st, a, (const float*)(saved + pl.o_bn2),
               (float*)(scratch + pl.o_gpart), (void*)(saved + pl.o_Bpost), ncb);
    const long total = (long)ncb * (d.dgp * d.dgp + 2 * d.dgp + 2);
    hipLaunchKernelGGL(kw_gram_finish, dim3(grid1dw(total)), dim3(256), 0, st, a, 0, (const float*)(scratch + pl.o_gpart), (float*)(saved + pl.o_Gq),
                       (float*)(saved + pl.o_uvh));
  }
  AVMOE_CHECK_LAUNCH("post_prep");
  return OK;
}

// ---------------------------------------------------------------------------------------------
// POST_PREP backward, per channel: dBpost, dG, dusum, dvh, dH  ->  up_sampler / bn2 / ln_post gradients and
// (dmo, dv2) per channel; then the channel contraction (kw_gram mode 1) gives dmz / NT and 2 dSzz / NT.
// grid (g*E, ceil(Cg/8))
// ---------------------------------------------------------------------------------------------
// block = PB_CH channels x 32 bottleneck lanes of one (group, expert): lane jl owns rows j = jl, jl + 32, .. of the d x d
// contractions (LDS matrices padded to an odd leading dim), the per-channel sums are folded with 5 shuffles
constexpr int PB_CH = 8, PB_JMAX = 4;      // bottleneck per group <= 32 * PB_JMAX
__global__ void __launch_bounds__(256) kw_post_prep_bwd(WArgs a, const float* bn2, const float* dBp, const float* dGq, const float* dsm,
                                                        const float* mz, const float* Szz, float* dmodv) {
  extern __shared__ float sm[];
  const int dg = a.dg, dgp = a.dgp, ldm = dgp + 1;
  float* s_dG = sm;                      // dgp x ldm
  float* s_S = s_dG + dgp * ldm;         // dgp x ldm
  float* s_v = s_S + dgp * ldm;          // 3*dgp : dusum, dvh, mz
  float* s_w = s_v + 3 * dgp;            // PB_CH x dgp
  const int cb = blockIdx.x, i = cb / a.E, e = cb % a.E;
  const int jl = threadIdx.x & 31, cc = threadIdx.x >> 5;
  const int cl = blockIdx.y * PB_CH + cc;
  const bool on = cl < a.Cg;
  const int c = i * a.Cg + cl;
  const bool stats = a.use_bn && a.training;
  for (int k = threadIdx.x; k < dgp * dgp; k += 256) {
    const int r = k / dgp, col = k - r * dgp;
    s_dG[r * ldm + col] = a.ln_post ? dGq[(long)cb * dgp * dgp + k] : 0.f;
    s_S[r * ldm + col] = stats ? Szz[(long)cb * dgp * dgp + k] : 0.f;
  }
  for (int k = threadIdx.x; k < dgp; k += 256) {
    s_v[k] = a.ln_post ? dsm[(long)cb * dgp + k] : 0.f;
    s_v[dgp + k] = a.ln_post ? dsm[a.DZ + (long)cb * dgp + k] : 0.f;
    s_v[2 * dgp + k] = stats ? mz[(long)cb * dgp + k] : 0.f;
  }
  for (int j = jl; j < dgp; j += 32) s_w[cc * dgp + j] = (on && j < dg) ? a.up.p[e][(long)c * dg + j] : 0.f;
  __syncthreads();
  const long EC = (long)a.E * a.C, idx = (long)e * a.C + (on ? c : 0);
  const float mo = bn2[idx], rs2 = bn2[EC + idx], k2 = bn2[2 * EC + idx], h2 = bn2[3 * EC + idx];
  const float* wu = s_w + cc * dgp;
  const float gp = (a.ln_post && on) ? a.lpw.p[e][c] : 1.f;
  const float* dBrow = dBp + (long)(on ? c : 0) * a.KPp;
  const float* dBmain = dBrow + e * dgp;
  // gate_w: dBp is the gradient of the GATED weights gate_e * Bpost_e.  dgate_e = sum over the expert's columns of dBp * Bpost (ungated), per
  // channel here (summed over the channels by kw_gram mode 1 / kw_gram_finish); everything below sees gt * dBp = the gradient of Bpost itself.
  const float gt = a.gate_w ? a.gate.p[e][0] : 1.f;
  const float dBh_r = dBrow[a.E * dgp + 3 * e + 0], dBg_r = dBrow[a.E * dgp + 3 * e + 1], dBb_r = dBrow[a.E * dgp + 3 * e + 2];
  const float dBh = gt * dBh_r, dBg = gt * dBg_r, dBb = gt * dBb_r;
  float dH1 = 0.f, dH2 = 0.f;
  if (a.ln_post) { dH1 = dsm[8 * a.DZ + e]; dH2 = dsm[8 * a.DZ + a.E + e]; }
  float dWh[PB_JMAX];
  float dk2 = 0.f, dgp_acc = 0.f, dh2p = 0.f;
#pragma unroll
  for (int jj = 0; jj < PB_JMAX; ++jj) {
    const int j = jl + 32 * jj;
    dWh[jj] = 0.f;
    if (j < dg) {
      float v = gp * gt * dBmain[j];
      if (a.ln_post) {
        float acc = 0.f;
        for (int l = 0; l < dg; ++l) acc += s_dG[j * ldm + l] * wu[l];
        v += 2.f * k2 * acc + s_v[j] + s_v[dgp + j] * h2;
      }
      dWh[jj] = v;
      dk2 += v * wu[j];
      dgp_acc += dBmain[j] * (wu[j] * k2);          // (ungated: scaled below)
      if (a.ln_post) dh2p += s_v[dgp + j] * (wu[j] * k2);
    }
  }
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) { dk2 += __shfl_xor(dk2, o, 64); dgp_acc += __shfl_xor(dgp_acc, o, 64); dh2p += __shfl_xor(dh2p, o, 64); }
  if (a.gate_w) {
    const float bp = (a.ln_post && on) ? a.lpb.p[e][c] : 0.f;
    if (on && jl == 0) dmodv[2 * EC + idx] = gp * dgp_acc + gp * h2 * dBh_r + gp * dBg_r + bp * dBb_r;
  }
  dgp_acc *= gt;
  float dh2 = gp * dBh + dh2p;
  if (a.ln_post) {
    dh2 += dH1 + 2.f * h2 * dH2;
    if (on && jl == 0) {
      if (a.glpw.p[e]) a.glpw.p[e][c] = dgp_acc + dBh * h2 + dBg;
      if (a.glpb.p[e]) a.glpb.p[e][c] = dBb;
    }
  }
  float dmo = 0.f, dv2 = 0.f;
  if (a.use_bn) {
    if (on && jl == 0 && a.gb2.p[e]) a.gb2.p[e][c] = dh2;
    dmo = -k2 * dh2;
    dk2 -= mo * dh2;
    if (on && jl == 0 && a.gw2.p[e]) a.gw2.p[e][c] = dk2 * rs2;
    dv2 = dk2 * (on ? a.w2.p[e][c] : 0.f) * (-0.5f) * rs2 * rs2 * rs2;
    if (a.training) dmo -= 2.f * mo * dv2; else { dmo = 0.f; dv2 = 0.f; }
  }
  if (on && jl == 0) { dmodv[idx] = dmo; dmodv[EC + idx] = dv2; }
  float* gu = (on && a.gup.p[e]) ? a.gup.p[e] + (long)c * dg : nullptr;
#pragma unroll
  for (int jj = 0; jj < PB_JMAX; ++jj) {
    const int j = jl + 32 * jj;
    if (j < dg) {
      float v = dWh[jj] * k2;
      if (stats) {
        float acc = 0.f;
        for (int l = 0; l < dg; ++l) acc += s_S[j * ldm + l] * wu[l];
        v += dmo * s_v[2 * dgp + j] + 2.f * dv2 * acc;
      }
      if (gu) gu[j] = v;
    }
  }
}

int k_post_prep_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads, hipStream_t st) {
  ProfScope ps_("k_post_prep_bwd", 0.0, 0.0, st);
  const Dims& d = pl.d;
  WArgs a; fill_w(d, prm, &grads, &a);
  if (d.dg > 32 * PB_JMAX) { set_last_error("post_prep_bwd: bottleneck per group %d > %d", d.dg, 32 * PB_JMAX); return ERR_UNSUPPORTED; }
  const size_t sh = (size_t)(2 * d.dgp * (d.dgp + 1) + 3 * d.dgp + PB_CH * d.dgp) * sizeof(float);
  if (sh > 160 * 1024) { set_last_error("post_prep_bwd: bottleneck per group %d needs %zu B of LDS", d.dg, sh); return ERR_UNSUPPORTED; }
  if (sh > 65536 && hipFuncSetAttribute((const void*)kw_post_prep_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
    set_last_error("post_prep_bwd: LDS attribute"); return ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kw_post_prep_bwd, dim3(d.g * d.E, cdiv(d.Cg, PB_CH)), dim3(256), sh, st, a, (const float*)(saved + pl.o_bn2),
                     (const float*)(scratch + pl.o_dBp), (const float*)(scratch + pl.o_dGq), (const float*)(scratch + pl.o_dsm),
                     (const float*)(saved + pl.o_mz), (const float*)(saved + pl.o_Szz), (float*)(scratch + pl.o_dmodv));
  AVMOE_CHECK_LAUNCH("post_prep_bwd");
  return run_gram(pl, a, 1, nullptr, (const float*)(scratch + pl.o_dmodv), (float*)(scratch + pl.o_gpart), (float*)(scratch + pl.o_sdSzz),
                  (float*)(scratch + pl.o_dsm), st);
}

}  // namespace avmoe
