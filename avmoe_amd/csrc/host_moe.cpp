// CPU (host memory, fp32) implementation of the adapter-site ABI of include/avmoe.h -- SURVEY 8(b): "a host (CPU, C++) implementation of the
// same ABI is the restatement used for CPU timing and no-GPU CI".  Built with g++ into avmoe_amd/lib/libavmoe_host.so (avmoe_amd/build.py:
// the host_ prefix keeps it out of the HIP library); declared in include/avmoe_host.h; pinned on the reference's vectors in the CPU suite
// (tests/test_host_golden.py).  It is NOT part of the product path (which has no CPU fallback) and shares no code with it: the reference's
// own formulation is evaluated directly, op by op, with a hand-written reverse pass -- no bottleneck-space re-factorisation -- so it is an
// independent second statement of the arithmetic beside oracle/avmoe_oracle.py (eager PyTorch + autograd).
//
// Follows, token-major (X (S, N, C), Y (S, M, Cy)):
//   remap + router + mixture   MoEAdapter.forward          AVMOE/AVE/nets/net_trans_v3.py:468-487
//   cross-modal expert         ExpertAdapter.forward       net_trans_v3.py:377-403, 430-435
//   unimodal expert                                        net_trans_v3.py:405-422, 430-435
//   AVS logit noise, probs, load-balancing loss            AVS/avs_scripts/avs_s4/model/PVT_AVSModel_v2.py:294-296, 312-318
//   AVVP unimodal N x N block                              AVVP/nets/mgn.py:132-139
//   AVS "v2" latent self attention                         PVT_AVSModel_v2.py:215-227
// Served: every variant, AVS self_attention_version "v1" (MultiheadAttention across the frames) included since round 6 (it was AVMOE_ERR_UNSUPPORTED here; the HIP
// library and the Python oracle have it), train and eval BatchNorm, every flag of the descriptor.
#include "../../include/avmoe_host.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
  return code;
}
typedef std::vector<float> V;

// C (M x N) = alpha * op(A) op(B) + beta * C ; op(A) is M x K: A stored (M, lda) or, transposed, (K, lda) ; op(B) is K x N: B stored (K, ldb) or (N, ldb)
void gemm(bool tA, bool tB, int M, int N, int K, float alpha, const float* A, long lda, const float* B, long ldb, float beta, float* C, long ldc) {
#pragma omp parallel for schedule(static) if ((long)M * N * K > 65536)
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += (double)(tA ? A[(long)k * lda + i] : A[(long)i * lda + k]) * (double)(tB ? B[(long)j * ldb + k] : B[(long)k * ldb + j]);
      C[(long)i * ldc + j] = alpha * (float)s + (beta == 0.f ? 0.f : beta * C[(long)i * ldc + j]);
    }
}
void softmax_rows(float* a, int rows, int n, long ld) {
  for (int i = 0; i < rows; ++i) {
    float* p = a + (long)i * ld;
    float mx = p[0];
    for (int j = 1; j < n; ++j) mx = std::fmax(mx, p[j]);
    double s = 0.0;
    for (int j = 0; j < n; ++j) { p[j] = std::exp(p[j] - mx); s += p[j]; }
    for (int j = 0; j < n; ++j) p[j] = (float)(p[j] / s);
  }
}
// dL = P * (dP - rowsum(P dP)), in place on dP
void softmax_rows_bwd(const float* P, float* dP, int rows, int n, long ld) {
  for (int i = 0; i < rows; ++i) {
    const float* p = P + (long)i * ld; float* d = dP + (long)i * ld;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += (double)p[j] * d[j];
    for (int j = 0; j < n; ++j) d[j] = p[j] * (d[j] - (float)s);
  }
}
struct LN { V mu, rs; };     // per row
void ln_fwd(const float* x, float* y, long rows, int C, const float* w, const float* b, float eps, LN& st) {
  st.mu.resize(rows); st.rs.resize(rows);
#pragma omp parallel for schedule(static) if (rows * C > 65536)
  for (long i = 0; i < rows; ++i) {
    const float* p = x + i * C;
    double m = 0.0; for (int c = 0; c < C; ++c) m += p[c]; m /= C;
    double v = 0.0; for (int c = 0; c < C; ++c) v += (p[c] - m) * (p[c] - m); v /= C;
    const float r = (float)(1.0 / std::sqrt(v + eps));
    st.mu[i] = (float)m; st.rs[i] = r;
    for (int c = 0; c < C; ++c) y[i * C + c] = (p[c] - (float)m) * r * w[c] + b[c];
  }
}
// dx from dy ; dw, db accumulated (may be null)
void ln_bwd(const float* x, const float* dy, float* dx, long rows, int C, const float* w, const LN& st, float* dw, float* db) {
  std::vector<double> aw(C, 0.0), ab(C, 0.0);
  for (long i = 0; i < rows; ++i) {
    const float* p = x + i * C; const float* g = dy + i * C;
    const float m = st.mu[i], r = st.rs[i];
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < C; ++c) { const float xh = (p[c] - m) * r, gh = g[c] * w[c]; s1 += gh; s2 += (double)gh * xh; aw[c] += (double)g[c] * xh; ab[c] += g[c]; }
    for (int c = 0; c < C; ++c) { const float xh = (p[c] - m) * r, gh = g[c] * w[c]; dx[i * C + c] = r * (gh - (float)(s1 / C) - xh * (float)(s2 / C)); }
  }
  if (dw) for (int c = 0; c < C; ++c) dw[c] = (float)aw[c];
  if (db) for (int c = 0; c < C; ++c) db[c] = (float)ab[c];
}
struct BN { V mu, rs; };     // per channel
// train: batch statistics over the rows (biased variance), running statistics updated when `update` ; eval: running statistics
void bn_fwd(const float* x, float* y, long rows, int C, const float* w, const float* b, float eps, bool train, float mom, float* rm, float* rv, int64_t* nbt,
            bool update, BN& st) {
  st.mu.assign(C, 0.f); st.rs.assign(C, 0.f);
  for (int c = 0; c < C; ++c) {
    double m, v;
    if (train) {
      m = 0.0; for (long i = 0; i < rows; ++i) m += x[i * C + c]; m /= rows;
      v = 0.0; for (long i = 0; i < rows; ++i) v += (x[i * C + c] - m) * (x[i * C + c] - m); v /= rows;
      if (update && rm && rv) {
        rm[c] = (1.f - mom) * rm[c] + mom * (float)m;
        rv[c] = (1.f - mom) * rv[c] + mom * (float)(rows > 1 ? v * rows / (rows - 1) : v);
      }
    } else { m = rm[c]; v = rv[c]; }
    st.mu[c] = (float)m; st.rs[c] = (float)(1.0 / std::sqrt(v + eps));
  }
  if (train && update && nbt) *nbt += 1;
  for (long i = 0; i < rows; ++i)
    for (int c = 0; c < C; ++c) y[i * C + c] = (x[i * C + c] - st.mu[c]) * st.rs[c] * w[c] + b[c];
}
void bn_bwd(const float* x, const float* dy, float* dx, long rows, int C, const float* w, bool train, const BN& st, float* dw, float* db) {
  for (int c = 0; c < C; ++c) {
    double s1 = 0.0, s2 = 0.0;
    for (long i = 0; i < rows; ++i) { const float xh = (x[i * C + c] - st.mu[c]) * st.rs[c]; s1 += dy[i * C + c]; s2 += (double)dy[i * C + c] * xh; }
    if (dw) dw[c] = (float)s2;
    if (db) db[c] = (float)s1;
    for (long i = 0; i < rows; ++i) {
      const float xh = (x[i * C + c] - st.mu[c]) * st.rs[c];
      dx[i * C + c] = train ? w[c] * st.rs[c] * (dy[i * C + c] - (float)(s1 / rows) - xh * (float)(s2 / rows)) : w[c] * st.rs[c] * dy[i * C + c];
    }
  }
}
// grouped 1x1 convolution on token rows: y (rows, Co) = x (rows, Ci) W^T per group ; W (Co, Ci / g): output chunk i reads input chunk i
void gconv_fwd(const float* x, float* y, long rows, int Ci, int Co, int g, const float* W) {
  const int ci = Ci / g, co = Co / g;
  for (int i = 0; i < g; ++i) gemm(false, true, (int)rows, co, ci, 1.f, x + i * ci, Ci, W + (long)i * co * ci, ci, 0.f, y + i * co, Co);
}
void gconv_bwd(const float* x, const float* dy, float* dx, float* dW, long rows, int Ci, int Co, int g, const float* W) {
  const int ci = Ci / g, co = Co / g;
  for (int i = 0; i < g; ++i) {
    gemm(false, false, (int)rows, ci, co, 1.f, dy + i * co, Co, W + (long)i * co * ci, ci, 0.f, dx + i * ci, Ci);      // dx = dy W
    if (dW) gemm(true, false, co, ci, (int)rows, 1.f, dy + i * co, Co, x + i * ci, Ci, 0.f, dW + (long)i * co * ci, ci);      // dW = dy^T x
  }
}

struct Expert {      // what one expert's forward leaves for its backward (all (S N, .) token-major)
  bool lat = false;                 // two-hop latent attention in front (cross-modal experts: over the remapped tokens; AVS v2 unimodal: over x itself)
  bool self = false;                // ... over x itself
  bool nxn = false;                 // AVVP unimodal: x' = x + gate_av softmax_rows(x x^T)^T x   (A1 holds the (S, N, N) softmax)
  bool mha = false;                 // AVS unimodal, self_attention_version "v1": x' = MultiheadAttention_4(x, x, x) ACROSS THE FRAMES (sequence = S, batch = N;
                                    // PVT_AVSModel_v2.py:210-214) -- Xp = x', QKV (S N, 3C), A1 = the (N 4, S, S) softmax before the dropout multiplier, Xr = heads' outputs (S N, C)
  V QKV;
  bool relu = false;                // cross-modal experts only
  V A1, T, A2, Xr, Xp;              // cross-modal: (S, K, N) ; (S, K, C) ; (S, N, K) ; (S N, C) ; x' = x + gate_av xr
  V U, Z, Zb, O, Ob, Op, pre;       // LN_before(x') ; down ; BN1 (+ ReLU applied into Za) ; up ; BN2 ; LN_post ; = what the gate multiplies
  V Za;
  LN lnb, lnp; BN bn1, bn2;
};

struct Ctx {
  int S, N, C, M, Cy, E, Em, d, g, K;
  bool avvp, v2, v1;
  bool bn, gate, lnb, lnp, train, lb;
  float bn_eps, ln_eps, mom;
  V Yt, Yf, rin, h1, h2, logit, p;
  std::vector<Expert> ex;
};

int setup(const avmoe_moe_desc* q, Ctx& c) {
  if (!q) return fail(AVMOE_ERR_BAD_ARG, "host: null descriptor");
  if (q->dtype != AVMOE_F32) return fail(AVMOE_ERR_UNSUPPORTED, "host: fp32 activations only");
  if (q->self_attn == AVMOE_SELF_ATTN_MHA_V1 && q->C % 4) return fail(AVMOE_ERR_BAD_ARG, "host: self_attention_version v1 needs C divisible by its 4 heads");
  c.S = q->S; c.N = q->N; c.C = q->C; c.M = q->M; c.Cy = q->Cy; c.Em = q->E_m; c.E = q->E_m + q->E_s; c.d = q->d; c.g = q->groups; c.K = q->K;
  if (c.S <= 0 || c.N <= 0 || c.C <= 0 || c.M <= 0 || c.Cy <= 0 || c.E <= 0 || c.E > AVMOE_MAX_EXPERTS || c.d <= 0 || c.g <= 0 || c.d % c.g || c.C % c.g || (c.Em > 0 && c.K <= 0))
    return fail(AVMOE_ERR_BAD_ARG, "host: bad extents");
  c.avvp = q->variant == AVMOE_VARIANT_AVVP || q->self_attn == AVMOE_SELF_ATTN_NXN; c.v2 = q->self_attn == AVMOE_SELF_ATTN_LATENT_V2;
  c.v1 = q->self_attn == AVMOE_SELF_ATTN_MHA_V1;
  if (c.v2 && c.K <= 0) return fail(AVMOE_ERR_BAD_ARG, "host: bad extents");
  c.bn = q->use_bn; c.gate = q->use_gate; c.lnb = q->ln_before; c.lnp = q->ln_post; c.train = q->training; c.lb = q->lb_loss;
  c.bn_eps = q->bn_eps; c.ln_eps = q->ln_eps; c.mom = q->bn_momentum;
  return AVMOE_OK;
}

// the forward ; update: advance the BatchNorm running statistics / counters (the backward recomputes the forward with update = false)
int forward(Ctx& c, const float* X, const float* Y, const avmoe_moe_ptrs& P, const float* noise, float* out, float* probs, int64_t* idx, float* lb, bool update) {
  const int S = c.S, N = c.N, C = c.C, M = c.M, Cy = c.Cy, E = c.E, K = c.K;
  const long NT = (long)S * N;
  if (!P.conv_w || !P.conv_b || !P.fc_w || !P.fc_b || !P.r0_w || !P.r0_b || !P.r2_w || !P.r2_b || !P.r4_w || !P.r4_b) return fail(AVMOE_ERR_BAD_ARG, "host: remap / router parameter missing");
  // ---- remap (net_trans_v3.py:469-471): Yt[s] = Wc Y[s] + bc ; Yf = Yt Wf^T + bf
  c.Yt.assign(NT * Cy, 0.f); c.Yf.assign(NT * C, 0.f);
  for (int s = 0; s < S; ++s) {
    gemm(false, false, N, Cy, M, 1.f, P.conv_w, M, Y + (long)s * M * Cy, Cy, 0.f, c.Yt.data() + (long)s * N * Cy, Cy);
    for (int n = 0; n < N; ++n) for (int k = 0; k < Cy; ++k) c.Yt[((long)s * N + n) * Cy + k] += P.conv_b[n];
  }
  gemm(false, true, (int)NT, C, Cy, 1.f, c.Yt.data(), Cy, P.fc_w, Cy, 0.f, c.Yf.data(), C);
  for (long i = 0; i < NT; ++i) for (int k = 0; k < C; ++k) c.Yf[i * C + k] += P.fc_b[k];
  // ---- router (:472-479): means over the tokens, 2C -> 128 -> 32 -> E, softmax, first-max argmax
  c.rin.assign((long)S * 2 * C, 0.f);
  for (int s = 0; s < S; ++s)
    for (int k = 0; k < C; ++k) {
      double a = 0.0, b = 0.0;
      for (int n = 0; n < N; ++n) { a += X[((long)s * N + n) * C + k]; b += c.Yf[((long)s * N + n) * C + k]; }
      c.rin[(long)s * 2 * C + k] = (float)(a / N); c.rin[(long)s * 2 * C + C + k] = (float)(b / N);
    }
  c.h1.assign((long)S * 128, 0.f); c.h2.assign((long)S * 32, 0.f); c.logit.assign((long)S * E, 0.f); c.p.assign((long)S * E, 0.f);
  gemm(false, true, S, 128, 2 * C, 1.f, c.rin.data(), 2 * C, P.r0_w, 2 * C, 0.f, c.h1.data(), 128);
  for (int s = 0; s < S; ++s) for (int j = 0; j < 128; ++j) c.h1[s * 128 + j] = std::fmax(c.h1[s * 128 + j] + P.r0_b[j], 0.f);
  gemm(false, true, S, 32, 128, 1.f, c.h1.data(), 128, P.r2_w, 128, 0.f, c.h2.data(), 32);
  for (int s = 0; s < S; ++s) for (int j = 0; j < 32; ++j) c.h2[s * 32 + j] = std::fmax(c.h2[s * 32 + j] + P.r2_b[j], 0.f);
  gemm(false, true, S, E, 32, 1.f, c.h2.data(), 32, P.r4_w, 32, 0.f, c.logit.data(), E);
  for (int s = 0; s < S; ++s) for (int e = 0; e < E; ++e) c.logit[s * E + e] += P.r4_b[e] + (noise ? noise[s * E + e] : 0.f);
  c.p = c.logit;
  softmax_rows(c.p.data(), S, E, E);
  for (int s = 0; s < S; ++s) {
    int best = 0;
    for (int e = 1; e < E; ++e) if (c.p[s * E + e] > c.p[s * E + best]) best = e;
    if (idx) idx[s] = best;
    if (probs) for (int e = 0; e < E; ++e) probs[s * E + e] = c.p[s * E + e];
  }
  if (lb) {      // -sum_e log(mean_s p): the reference's kl_div against the constant 1 (PVT_AVSModel_v2.py:314-318)
    double v = 0.0;
    if (c.lb) for (int e = 0; e < E; ++e) { double m = 0.0; for (int s = 0; s < S; ++s) m += c.p[s * E + e]; v -= std::log(m / S); }
    *lb = (float)v;
  }
  // ---- experts (multimodal first, then singlemodal: :482) and the mixture (:485-486)
  c.ex.assign(E, Expert());
  for (long i = 0; i < NT * C; ++i) out[i] = 0.f;
  for (int e = 0; e < E; ++e) {
    Expert& x = c.ex[e];
    const avmoe_expert_ptrs& q = P.e[e];
    x.relu = e < c.Em;
    x.lat = e < c.Em || c.v2; x.self = !(e < c.Em); x.nxn = !(e < c.Em) && c.avvp && !c.v1; x.mha = !(e < c.Em) && c.v1;
    if (!q.down_w || !q.up_w) return fail(AVMOE_ERR_BAD_ARG, "host: expert %d lacks its projections", e);
    const float* xin = X;
    if (x.lat) {
      if (!q.my_tokens || !q.gate_lat) return fail(AVMOE_ERR_BAD_ARG, "host: expert %d lacks my_tokens / gate_av (gate_self)", e);
      x.A1.assign((long)S * K * N, 0.f); x.T.assign((long)S * K * C, 0.f); x.A2.assign(NT * K, 0.f); x.Xr.assign(NT * C, 0.f); x.Xp.assign(NT * C, 0.f);
      for (int s = 0; s < S; ++s) {
        const float* Xs = X + (long)s * N * C;
        const float* Yf = x.self ? Xs : c.Yf.data() + (long)s * N * C;                         // the token set the latent tokens summarise
        float* A1 = x.A1.data() + (long)s * K * N; float* T = x.T.data() + (long)s * K * C; float* A2 = x.A2.data() + (long)s * N * K;
        gemm(false, true, K, N, C, 1.f, q.my_tokens, C, Yf, C, 0.f, A1, N);                    // hop 1 (:380-383): latent tokens read the remapped tokens
        softmax_rows(A1, K, N, N);
        std::memcpy(T, q.my_tokens, sizeof(float) * K * C);
        gemm(false, false, K, C, N, 1.f, A1, N, Yf, C, 1.f, T, C);
        gemm(false, true, N, K, C, 1.f, Xs, C, T, C, 0.f, A2, K);                              // hop 2 (:385-388): x reads the latent tokens
        softmax_rows(A2, N, K, K);
        gemm(false, false, N, C, K, 1.f, A2, K, T, C, 0.f, x.Xr.data() + (long)s * N * C, C);
      }
      for (long i = 0; i < NT * C; ++i) x.Xp[i] = X[i] + q.gate_lat[0] * x.Xr[i];              // (:390)
      xin = x.Xp.data();
    } else if (x.nxn) {
      if (!q.gate_lat) return fail(AVMOE_ERR_BAD_ARG, "host: expert %d lacks gate_av", e);
      x.A1.assign((long)S * N * N, 0.f); x.Xr.assign(NT * C, 0.f); x.Xp.assign(NT * C, 0.f);
      for (int s = 0; s < S; ++s) {
        const float* Xs = X + (long)s * N * C; float* att = x.A1.data() + (long)s * N * N;
        gemm(false, true, N, N, C, 1.f, Xs, C, Xs, C, 0.f, att, N);                            // mgn.py:134-136
        softmax_rows(att, N, N, N);
        gemm(true, false, N, C, N, 1.f, att, N, Xs, C, 0.f, x.Xr.data() + (long)s * N * C, C);  // xr = att^T x (mgn.py:137-139: x_cn @ att)
      }
      for (long i = 0; i < NT * C; ++i) x.Xp[i] = X[i] + q.gate_lat[0] * x.Xr[i];
      xin = x.Xp.data();
    }
    else if (x.mha) {                // torch.nn.functional.multi_head_attention_forward, batch_first = False: q scaled by 1 / sqrt(dh), softmax over the key
                                     // frames, the caller's dropout multiplier on the weights (sa_keep: (N 4, S, S), or NULL), out_proj
      if (!q.sa_in_w || !q.sa_in_b || !q.sa_out_w || !q.sa_out_b) return fail(AVMOE_ERR_BAD_ARG, "host: expert %d lacks self_attention.*", e);
      constexpr int H = 4;
      const int dh = C / H;
      const float scale = 1.f / std::sqrt((float)dh);
      x.QKV.assign(NT * 3 * C, 0.f); x.A1.assign((long)N * H * S * S, 0.f); x.Xr.assign(NT * C, 0.f); x.Xp.assign(NT * C, 0.f);
      gemm(false, true, (int)NT, 3 * C, C, 1.f, X, C, q.sa_in_w, C, 0.f, x.QKV.data(), 3 * C);
      for (long i = 0; i < NT; ++i) for (int k = 0; k < 3 * C; ++k) x.QKV[i * 3 * C + k] += q.sa_in_b[k];
#pragma omp parallel for schedule(static)
      for (int nh = 0; nh < N * H; ++nh) {
        const int n = nh / H, hh = nh % H;
        float* att = x.A1.data() + (long)nh * S * S;
        for (int s = 0; s < S; ++s)
          for (int u2 = 0; u2 < S; ++u2) {
            const float* qv = x.QKV.data() + ((long)s * N + n) * 3 * C + hh * dh;
            const float* kv = x.QKV.data() + ((long)u2 * N + n) * 3 * C + C + hh * dh;
            double a = 0.0;
            for (int j = 0; j < dh; ++j) a += (double)(qv[j] * scale) * kv[j];
            att[(long)s * S + u2] = (float)a;
          }
        softmax_rows(att, S, S, S);
        const float* keep = q.sa_keep ? q.sa_keep + (long)nh * S * S : nullptr;
        for (int s = 0; s < S; ++s)
          for (int j = 0; j < dh; ++j) {
            double a = 0.0;
            for (int u2 = 0; u2 < S; ++u2) a += (double)(att[(long)s * S + u2] * (keep ? keep[(long)s * S + u2] : 1.f)) * x.QKV[((long)u2 * N + n) * 3 * C + 2 * C + hh * dh + j];
            x.Xr[((long)s * N + n) * C + hh * dh + j] = (float)a;
          }
      }
      gemm(false, true, (int)NT, C, C, 1.f, x.Xr.data(), C, q.sa_out_w, C, 0.f, x.Xp.data(), C);
      for (long i = 0; i < NT; ++i) for (int k = 0; k < C; ++k) x.Xp[i * C + k] += q.sa_out_b[k];
      xin = x.Xp.data();             // REPLACES x (no residual, no gate)
    }
    const float* u = xin;
    if (c.lnb) {
      if (!q.lnb_w || !q.lnb_b) return fail(AVMOE_ERR_BAD_ARG, "host: expert %d lacks ln_before", e);
      x.U.assign(NT * C, 0.f); ln_fwd(xin, x.U.data(), NT, C, q.lnb_w, q.lnb_b, c.ln_eps, x.lnb); u = x.U.data();
    }
    x.Z.assign(NT * c.d, 0.f); gconv_fwd(u, x.Z.data(), NT, C, c.d, c.g, q.down_w);
    const float* z = x.Z.data();
    if (c.bn) {
      if (!q.bn1_w || !q.bn1_b || !q.bn2_w || !q.bn2_b || (!c.train && (!q.bn1_rm || !q.bn1_rv || !q.bn2_rm || !q.bn2_rv))) return fail(AVMOE_ERR_BAD_ARG, "host: expert %d lacks BatchNorm tensors", e);
      x.Zb.assign(NT * c.d, 0.f); bn_fwd(z, x.Zb.data(), NT, c.d, q.bn1_w, q.bn1_b, c.bn_eps, c.train, c.mom, q.bn1_rm, q.bn1_rv, q.bn1_nbt, update, x.bn1); z = x.Zb.data();
    }
    x.Za.assign(z, z + NT * c.d);
    if (x.relu) for (auto& v : x.Za) v = std::fmax(v, 0.f);                                     // ReLU: cross-modal expert only (:400)
    x.O.assign(NT * C, 0.f); gconv_fwd(x.Za.data(), x.O.data(), NT, c.d, C, c.g, q.up_w);
    const float* o = x.O.data();
    if (c.bn) { x.Ob.assign(NT * C, 0.f); bn_fwd(o, x.Ob.data(), NT, C, q.bn2_w, q.bn2_b, c.bn_eps, c.train, c.mom, q.bn2_rm, q.bn2_rv, q.bn2_nbt, update, x.bn2); o = x.Ob.data(); }
    if (c.lnp) {
      if (!q.lnp_w || !q.lnp_b) return fail(AVMOE_ERR_BAD_ARG, "host: expert %d lacks ln_post", e);
      x.Op.assign(NT * C, 0.f); ln_fwd(o, x.Op.data(), NT, C, q.lnp_w, q.lnp_b, c.ln_eps, x.lnp); o = x.Op.data();
    }
    x.pre.assign(o, o + NT * C);
    const float gt = (c.gate && q.gate) ? q.gate[0] : 1.f;
    for (int s = 0; s < S; ++s) {
      const float w = c.p[s * E + e] * gt;
      for (long i = (long)s * N * C; i < (long)(s + 1) * N * C; ++i) out[i] += w * x.pre[i];
    }
  }
  return AVMOE_OK;
}

}  // namespace

extern "C" {

const char* avmoe_host_last_error(void) { return g_err; }
size_t avmoe_host_moe_saved_bytes(const avmoe_moe_desc* desc) { (void)desc; return 16; }      // the backward recomputes the forward: nothing is kept

int avmoe_host_moe_forward(const avmoe_moe_desc* desc, const float* X, const float* Y, const avmoe_moe_ptrs* params, const float* noise, float* out,
                           float* probs, int64_t* idx, float* lb, void* saved) {
  (void)saved;
  Ctx c;
  if (int rc = setup(desc, c)) return rc;
  if (!X || !Y || !params || !out) return fail(AVMOE_ERR_BAD_ARG, "host forward: null pointer");
  return forward(c, X, Y, *params, noise, out, probs, idx, lb, true);
}

int avmoe_host_moe_backward(const avmoe_moe_desc* desc, const float* X, const float* Y, const avmoe_moe_ptrs* params, const float* noise, const float* dOut,
                            const float* lb_grad, void* saved, float* dX, float* dY, const avmoe_moe_ptrs* grads) {
  (void)saved;
  Ctx c;
  if (int rc = setup(desc, c)) return rc;
  if (!X || !Y || !params || !dOut || !dX || !dY || !grads) return fail(AVMOE_ERR_BAD_ARG, "host backward: null pointer");
  const avmoe_moe_ptrs& P = *params; const avmoe_moe_ptrs& G = *grads;
  const int S = c.S, N = c.N, C = c.C, M = c.M, Cy = c.Cy, E = c.E, K = c.K, d = c.d;
  const long NT = (long)S * N;
  V out(NT * C);
  float lbv = 0.f;
  if (int rc = forward(c, X, Y, P, noise, out.data(), nullptr, nullptr, &lbv, false)) return rc;
  for (long i = 0; i < NT * C; ++i) dX[i] = 0.f;
  V dYf(NT * C, 0.f), dp((long)S * E, 0.f);
  V t0(NT * C), t1(NT * C), tz(NT * d), tz2(NT * d);
  for (int e = 0; e < E; ++e) {
    Expert& x = c.ex[e];
    const avmoe_expert_ptrs& q = P.e[e]; const avmoe_expert_ptrs& gq = G.e[e];
    const float gt = (c.gate && q.gate) ? q.gate[0] : 1.f;
    // mixture + gate: out += p[s, e] gate pre
    double dgate = 0.0;
    for (int s = 0; s < S; ++s) {
      double a = 0.0;
      const float pe = c.p[s * E + e];
      for (long i = (long)s * N * C; i < (long)(s + 1) * N * C; ++i) { a += (double)dOut[i] * x.pre[i]; t0[i] = dOut[i] * pe * gt; }
      dp[s * E + e] = (float)(a * gt); dgate += a * pe;
    }
    if (c.gate && gq.gate) gq.gate[0] = (float)dgate;
    float* g = t0.data(); float* h = t1.data();            // g: gradient of the current stage's output, h: scratch for its input's
    if (c.lnp) { ln_bwd(c.bn ? x.Ob.data() : x.O.data(), g, h, NT, C, q.lnp_w, x.lnp, gq.lnp_w, gq.lnp_b); std::swap(g, h); }
    if (c.bn) { bn_bwd(x.O.data(), g, h, NT, C, q.bn2_w, c.train, x.bn2, gq.bn2_w, gq.bn2_b); std::swap(g, h); }
    gconv_bwd(x.Za.data(), g, tz.data(), gq.up_w, NT, d, C, c.g, q.up_w);
    float* gz = tz.data(); float* hz = tz2.data();
    if (x.relu) for (long i = 0; i < NT * d; ++i) if (!(x.Za[i] > 0.f)) gz[i] = 0.f;
    if (c.bn) { bn_bwd(x.Z.data(), gz, hz, NT, d, q.bn1_w, c.train, x.bn1, gq.bn1_w, gq.bn1_b); std::swap(gz, hz); }
    const float* xin = (x.lat || x.nxn || x.mha) ? x.Xp.data() : X;
    gconv_bwd(c.lnb ? x.U.data() : xin, gz, g, gq.down_w, NT, C, d, c.g, q.down_w);      // g: gradient of LN_before's output (or of x')
    if (c.lnb) { ln_bwd(xin, g, h, NT, C, q.lnb_w, x.lnb, gq.lnb_w, gq.lnb_b); std::swap(g, h); }
    if (x.mha) {                      // x' = out_proj(heads(softmax(q k^T / sqrt(dh)) keep . v)): back through out_proj, the heads, in_proj -- into dX
      constexpr int H = 4;
      const int dh = C / H;
      const float scale = 1.f / std::sqrt((float)dh);
      if (gq.sa_out_w) gemm(true, false, C, C, (int)NT, 1.f, g, C, x.Xr.data(), C, 0.f, gq.sa_out_w, C);
      if (gq.sa_out_b) for (int k = 0; k < C; ++k) { double a = 0.0; for (long i = 0; i < NT; ++i) a += g[i * C + k]; gq.sa_out_b[k] = (float)a; }
      gemm(false, false, (int)NT, C, C, 1.f, g, C, q.sa_out_w, C, 0.f, h, C);                // h = d heads' outputs
      V dQKV(NT * 3 * C, 0.f);
#pragma omp parallel for schedule(static)
      for (int nh = 0; nh < N * H; ++nh) {
        const int n = nh / H, hh = nh % H;
        const float* att = x.A1.data() + (long)nh * S * S;
        const float* keep = q.sa_keep ? q.sa_keep + (long)nh * S * S : nullptr;
        V dP((long)S * S);
        for (int s = 0; s < S; ++s)
          for (int u2 = 0; u2 < S; ++u2) {
            const float kp = keep ? keep[(long)s * S + u2] : 1.f;
            double a = 0.0;
            for (int j = 0; j < dh; ++j) {
              const float dho = h[((long)s * N + n) * C + hh * dh + j];
              a += (double)dho * x.QKV[((long)u2 * N + n) * 3 * C + 2 * C + hh * dh + j];
              dQKV[((long)u2 * N + n) * 3 * C + 2 * C + hh * dh + j] += att[(long)s * S + u2] * kp * dho;      // dv (this (n, head) alone writes these entries)
            }
            dP[(long)s * S + u2] = (float)a * kp;
          }
        softmax_rows_bwd(att, dP.data(), S, S, S);                                             // -> d scores
        for (int s = 0; s < S; ++s)
          for (int u2 = 0; u2 < S; ++u2) {
            const float ds = dP[(long)s * S + u2];
            for (int j = 0; j < dh; ++j) {
              dQKV[((long)s * N + n) * 3 * C + hh * dh + j] += ds * scale * x.QKV[((long)u2 * N + n) * 3 * C + C + hh * dh + j];
              dQKV[((long)u2 * N + n) * 3 * C + C + hh * dh + j] += ds * scale * x.QKV[((long)s * N + n) * 3 * C + hh * dh + j];
            }
          }
      }
      if (gq.sa_in_w) gemm(true, false, 3 * C, C, (int)NT, 1.f, dQKV.data(), 3 * C, X, C, 0.f, gq.sa_in_w, C);
      if (gq.sa_in_b) for (int k = 0; k < 3 * C; ++k) { double a = 0.0; for (long i = 0; i < NT; ++i) a += dQKV[i * 3 * C + k]; gq.sa_in_b[k] = (float)a; }
      gemm(false, false, (int)NT, C, 3 * C, 1.f, dQKV.data(), 3 * C, q.sa_in_w, C, 1.f, dX, C);
    } else {
      for (long i = 0; i < NT * C; ++i) dX[i] += g[i];                                     // x' = x + ...
    }
    if (x.lat) {
      double dga = 0.0;
      for (long i = 0; i < NT * C; ++i) { dga += (double)g[i] * x.Xr[i]; g[i] *= q.gate_lat[0]; }      // g = d xr
      if (gq.gate_lat) gq.gate_lat[0] = (float)dga;
      V dT0((long)K * C, 0.f), dT((long)K * C), dA2((long)N * K), dA1((long)K * N);
      for (int s = 0; s < S; ++s) {
        const float* Xs = X + (long)s * N * C;
        const float* Yf = x.self ? Xs : c.Yf.data() + (long)s * N * C;
        float* dsrc = (x.self ? dX : dYf.data()) + (long)s * N * C;                            // gradient of the summarised token set
        const float* A1 = x.A1.data() + (long)s * K * N; const float* T = x.T.data() + (long)s * K * C; const float* A2 = x.A2.data() + (long)s * N * K;
        const float* dxr = g + (long)s * N * C;
        gemm(false, true, N, K, C, 1.f, dxr, C, T, C, 0.f, dA2.data(), K);                     // xr = A2 T
        gemm(true, false, K, C, N, 1.f, A2, K, dxr, C, 0.f, dT.data(), C);
        softmax_rows_bwd(A2, dA2.data(), N, K, K);                                             // -> dL2 ; L2 = x T^T
        gemm(false, false, N, C, K, 1.f, dA2.data(), K, T, C, 1.f, dX + (long)s * N * C, C);
        gemm(true, false, K, C, N, 1.f, dA2.data(), K, Xs, C, 1.f, dT.data(), C);
        for (long i = 0; i < (long)K * C; ++i) dT0[i] += dT[i];                                // T = T0 + A1 Yf
        gemm(false, true, K, N, C, 1.f, dT.data(), C, Yf, C, 0.f, dA1.data(), N);
        gemm(true, false, N, C, K, 1.f, A1, N, dT.data(), C, 1.f, dsrc, C);
        softmax_rows_bwd(A1, dA1.data(), K, N, N);                                             // -> dL1 ; L1 = T0 Yf^T
        gemm(false, false, K, C, N, 1.f, dA1.data(), N, Yf, C, 1.f, dT0.data(), C);
        gemm(true, false, N, C, K, 1.f, dA1.data(), N, q.my_tokens, C, 1.f, dsrc, C);
      }
      if (gq.my_tokens) std::memcpy(gq.my_tokens, dT0.data(), sizeof(float) * K * C);
    } else if (x.nxn) {
      double dga = 0.0;
      for (long i = 0; i < NT * C; ++i) { dga += (double)g[i] * x.Xr[i]; g[i] *= q.gate_lat[0]; }      // g = d xr
      if (gq.gate_lat) gq.gate_lat[0] = (float)dga;
      V datt((long)N * N);
      for (int s = 0; s < S; ++s) {
        const float* Xs = X + (long)s * N * C; const float* att = x.A1.data() + (long)s * N * N; const float* dxr = g + (long)s * N * C;
        float* dXs = dX + (long)s * N * C;
        gemm(false, true, N, N, C, 1.f, Xs, C, dxr, C, 0.f, datt.data(), N);                   // xr = att^T x : d att[i][j] = x_i . dxr_j
        gemm(false, false, N, C, N, 1.f, att, N, dxr, C, 1.f, dXs, C);                         //               dx_i += sum_j att[i][j] dxr_j
        softmax_rows_bwd(att, datt.data(), N, N, N);                                           // -> d scores ; scores = x x^T
        gemm(false, false, N, C, N, 1.f, datt.data(), N, Xs, C, 1.f, dXs, C);
        gemm(true, false, N, C, N, 1.f, datt.data(), N, Xs, C, 1.f, dXs, C);
      }
    }
  }
  // ---- router: p = softmax(logits) ; + the load-balancing loss -sum_e log(mean_s p)
  if (c.lb && lb_grad) for (int e = 0; e < E; ++e) { double m = 0.0; for (int s = 0; s < S; ++s) m += c.p[s * E + e]; m /= S; for (int s = 0; s < S; ++s) dp[s * E + e] -= lb_grad[0] / (float)(S * m); }
  softmax_rows_bwd(c.p.data(), dp.data(), S, E, E);                                            // dp -> dlogits
  V dh2((long)S * 32), dh1((long)S * 128), drin((long)S * 2 * C);
  if (G.r4_w) gemm(true, false, E, 32, S, 1.f, dp.data(), E, c.h2.data(), 32, 0.f, G.r4_w, 32);
  if (G.r4_b) for (int e = 0; e < E; ++e) { double a = 0.0; for (int s = 0; s < S; ++s) a += dp[s * E + e]; G.r4_b[e] = (float)a; }
  gemm(false, false, S, 32, E, 1.f, dp.data(), E, P.r4_w, 32, 0.f, dh2.data(), 32);
  for (long i = 0; i < (long)S * 32; ++i) if (!(c.h2[i] > 0.f)) dh2[i] = 0.f;
  if (G.r2_w) gemm(true, false, 32, 128, S, 1.f, dh2.data(), 32, c.h1.data(), 128, 0.f, G.r2_w, 128);
  if (G.r2_b) for (int j = 0; j < 32; ++j) { double a = 0.0; for (int s = 0; s < S; ++s) a += dh2[s * 32 + j]; G.r2_b[j] = (float)a; }
  gemm(false, false, S, 128, 32, 1.f, dh2.data(), 32, P.r2_w, 128, 0.f, dh1.data(), 128);
  for (long i = 0; i < (long)S * 128; ++i) if (!(c.h1[i] > 0.f)) dh1[i] = 0.f;
  if (G.r0_w) gemm(true, false, 128, 2 * C, S, 1.f, dh1.data(), 128, c.rin.data(), 2 * C, 0.f, G.r0_w, 2 * C);
  if (G.r0_b) for (int j = 0; j < 128; ++j) { double a = 0.0; for (int s = 0; s < S; ++s) a += dh1[s * 128 + j]; G.r0_b[j] = (float)a; }
  gemm(false, false, S, 2 * C, 128, 1.f, dh1.data(), 128, P.r0_w, 2 * C, 0.f, drin.data(), 2 * C);
  for (int s = 0; s < S; ++s)
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < C; ++k) {
        dX[((long)s * N + n) * C + k] += drin[(long)s * 2 * C + k] / N;
        dYf[((long)s * N + n) * C + k] += drin[(long)s * 2 * C + C + k] / N;
      }
  // ---- remap: Yf = Yt Wf^T + bf ; Yt[s] = Wc Y[s] + bc
  V dYt(NT * Cy);
  gemm(false, false, (int)NT, Cy, C, 1.f, dYf.data(), C, P.fc_w, Cy, 0.f, dYt.data(), Cy);
  if (G.fc_w) gemm(true, false, C, Cy, (int)NT, 1.f, dYf.data(), C, c.Yt.data(), Cy, 0.f, G.fc_w, Cy);
  if (G.fc_b) for (int k = 0; k < C; ++k) { double a = 0.0; for (long i = 0; i < NT; ++i) a += dYf[i * C + k]; G.fc_b[k] = (float)a; }
  V dWc((long)N * M, 0.f);
  for (int s = 0; s < S; ++s) {
    gemm(true, false, M, Cy, N, 1.f, P.conv_w, M, dYt.data() + (long)s * N * Cy, Cy, 0.f, dY + (long)s * M * Cy, Cy);
    gemm(false, true, N, M, Cy, 1.f, dYt.data() + (long)s * N * Cy, Cy, Y + (long)s * M * Cy, Cy, 1.f, dWc.data(), M);
  }
  if (G.conv_w) std::memcpy(G.conv_w, dWc.data(), sizeof(float) * N * M);
  if (G.conv_b) for (int n = 0; n < N; ++n) { double a = 0.0; for (int s = 0; s < S; ++s) for (int k = 0; k < Cy; ++k) a += dYt[((long)s * N + n) * Cy + k]; G.conv_b[n] = (float)a; }
  return AVMOE_OK;
}

}  // extern "C"
