/*
 * chaorec_oracle.c -- CPU restatement of ChaoRec's GCN-propagate + BPR + full-rank hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under chaorec_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Every function restates a reference call site (file:line relative to the reference root)
 * in plain scalar C with the rounding order spelled out, so that the HIP kernels can be
 * compared bit-for-bit where the arithmetic is order-defined (SpMM, scoring dot, GEMM,
 * sampler) and to a stated tolerance where libm differs (BPR's exp/log).
 *
 * Pinning: the reference has no tests.  This oracle is pinned against
 * outputs of the reference's own classes imported in the build container
 * (tests/golden/gen_golden.py -> tests/golden/ *.npz; tests/test_oracle_golden.py).
 * For LightGCN/MMGCN the third-party propagate (torch-geometric 2.1.0 / torch-scatter 2.0.9,
 * requirements.txt:49-50, not installed) is restated by oracle/pyg_standin.py, so those
 * goldens pin "reference model code + restated propagate"; FREEDOM/metrics/sampler goldens
 * are pure reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: products and sums below must round separately unless fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- P1/P7/P12 propagate:  out[c] += norm_e * x[r]  ------------------------------------
 * Model/LightGCN.py:40-43 (propagate -> message: norm.view(-1,1) * x_j, add-aggregate at
 * edge_index[1]); BasicGCN.py:48-53; torch.sparse.mm at Model/FREEDOM.py:168,174.
 * CSR row = destination, entries in the reference's edge order.  message() materialises the
 * product (one rounding), scatter_add_ then adds sequentially (second rounding). */
void oracle_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                         const float *x, float *y, int64_t n_rows, int32_t D, float alpha,
                         const float *z, float beta, float *acc, const float *acc_init,
                         float acc_w) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t r = 0; r < n_rows; ++r) {
    for (int32_t k = 0; k < D; ++k) {
      float s = 0.0f;
      for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
        const float t = val[e] * x[(size_t)col[e] * D + k];
        s = s + t;
      }
      s = alpha * s;
      if (z) {
        const float t = beta * z[(size_t)r * D + k];
        s = s + t;
      }
      if (y) y[(size_t)r * D + k] = s;
      if (acc) {
        /* Model/LightGCN.py:90-93: final_embeddings += weights[i] * embs[i] */
        float a0;
        if (acc_init) a0 = acc_w * acc_init[(size_t)r * D + k];
        else a0 = acc[(size_t)r * D + k];
        const float t = acc_w * s;
        acc[(size_t)r * D + k] = a0 + t;
      }
    }
  }
}

/* The un-fused reference formulation (edge list, scatter-add in edge order), used to check
 * that the CSR form above really is the same arithmetic: out[dst[e]] += w[e] * x[src[e]]. */
void oracle_scatter_edges_f32(const int64_t *src, const int64_t *dst, const float *w,
                              const float *x, float *out, int64_t n_edges, int32_t D) {
  for (int64_t e = 0; e < n_edges; ++e) {
    for (int32_t k = 0; k < D; ++k) {
      const float t = w[e] * x[(size_t)src[e] * D + k];
      out[(size_t)dst[e] * D + k] = out[(size_t)dst[e] * D + k] + t;
    }
  }
}

/* ---- P4/P5/P9/P13 BPR ------------------------------------------------------------------
 * variant 0: Model/LightGCN.py:97-121   -mean(log(sigmoid(d) + 1e-5)) + reg*(mean u^2 + ...)
 * variant 1: Model/FREEDOM.py:185-192   -mean(logsigmoid(d))
 * variant 2: Model/MMGCN.py:188-196     -mean(log(sigmoid(d)))
 * Sums in double here (the oracle is the accurate side; the kernel's fp32 butterflies are
 * compared with a tolerance).  out[0]=total out[1]=bpr out[2]=reg; coef[b] = d bpr / d d_b. */
void oracle_bpr_fwd_f32(const float *tab_u, const float *tab_i, const int64_t *users,
                        const int64_t *pos, const int64_t *neg, int32_t B, int32_t D,
                        int32_t variant, float reg_weight, double *out, double *coef) {
  double tsum = 0.0, ru = 0.0, rp = 0.0, rn = 0.0;
  for (int32_t b = 0; b < B; ++b) {
    const float *u = tab_u + (size_t)users[b] * D;
    const float *p = tab_i + (size_t)pos[b] * D;
    const float *n = tab_i + (size_t)neg[b] * D;
    double sp = 0.0, sn = 0.0;
    for (int32_t k = 0; k < D; ++k) {
      sp += (double)u[k] * p[k];
      sn += (double)u[k] * n[k];
      ru += (double)u[k] * u[k];
      rp += (double)p[k] * p[k];
      rn += (double)n[k] * n[k];
    }
    const double d = sp - sn;
    const double s = 1.0 / (1.0 + exp(-d));
    double term, c;
    if (variant == 0) {
      term = log(s + 1e-5);
      c = -(s * (1.0 - s)) / (s + 1e-5);
    } else if (variant == 1) {
      term = fmin(d, 0.0) - log1p(exp(-fabs(d)));
      c = -(1.0 - s);
    } else {
      term = log(s);
      c = -(1.0 - s);
    }
    tsum += term;
    if (coef) coef[b] = c / B;
  }
  const double bpr = -tsum / B;
  const double denom = (double)B * D;
  const double reg = reg_weight != 0.0f ? (double)reg_weight * (ru / denom + rp / denom + rn / denom) : 0.0;
  out[0] = bpr + reg;
  out[1] = bpr;
  out[2] = reg;
}

/* autograd of the above (index_select backward = index_add in batch order). */
void oracle_bpr_bwd_f32(const float *tab_u, const float *tab_i, const int64_t *users,
                        const int64_t *pos, const int64_t *neg, int32_t B, int32_t D,
                        const double *coef, float reg_weight, double grad_out, double *g_u,
                        double *g_i) {
  const double r2 = grad_out * 2.0 * reg_weight / ((double)B * D);
  for (int32_t b = 0; b < B; ++b) {
    const size_t ou = (size_t)users[b] * D, op = (size_t)pos[b] * D, on = (size_t)neg[b] * D;
    const double c = coef[b] * grad_out;
    for (int32_t k = 0; k < D; ++k) {
      const double u = tab_u[ou + k], p = tab_i[op + k], n = tab_i[on + k];
      g_u[ou + k] += c * (p - n) + r2 * u;
      g_i[op + k] += c * u + r2 * p;
      g_i[on + k] += -c * u + r2 * n;
    }
  }
}

/* The same index_add in FLOAT and in a stated order -- role by role (the rows `emb[users]`, then `emb[pos]`, then `emb[neg]`
 * of Model/LightGCN.py:113-121 / Model/MMGCN.py:193-197), batch order inside a role, every addend rounded to float and added to
 * the float row's ONE running sum.  (torch's CPU autograd accumulates the three index_select gradients separately and adds
 * the three tensors: another association of the same addends -- this is a defined order, not torch's.)  It is what the product's ORDERED backward
 * launch (chaorec_bpr_bwd_ordered_f32) reproduces bit for bit (the expressions are the kernel's: c = coef * grad_out,
 * r2 = (2 reg / (B D)) * grad_out, no fused multiply-add: -ffp-contract=off on both sides).  g_u / g_i may alias. */
void oracle_bpr_bwd_ordered_f32(const float *tab_u, const float *tab_i, const int64_t *users,
                                const int64_t *pos, const int64_t *neg, int32_t B, int32_t D,
                                const float *coef, float reg_weight, float grad_out, float *g_u, float *g_i) {
  const float r2_unit = 2.0f * reg_weight / ((float)B * (float)D);
  const float r2 = r2_unit * grad_out;
  for (int role = 0; role < 3; ++role) {
    for (int32_t b = 0; b < B; ++b) {
      const size_t ou = (size_t)users[b] * D, op = (size_t)pos[b] * D, on = (size_t)neg[b] * D;
      const float c = coef[b] * (grad_out * 1.0f);
      for (int32_t k = 0; k < D; ++k) {
        const float u = tab_u[ou + k], p = tab_i[op + k], n = tab_i[on + k];
        if (role == 0) {
          const float v = c * (p - n) + r2 * u;
          g_u[ou + k] += v;
        } else if (role == 1) {
          const float cu = c * u;
          const float v = cu + r2 * p;
          g_i[op + k] += v;
        } else {
          const float cu = -c * u;
          const float v = cu + r2 * n;
          g_i[on + k] += v;
        }
      }
    }
  }
}

/* ---- S sampler -------------------------------------------------------------------------
 * dataload.py:74-79: draw uniformly from all items until the draw is not in the user's
 * history.  The generator is this build's counter-based one (the reference's is Python's
 * Mersenne Twister: parity is distributional).  Same function as the HIP kernel. */
static uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint32_t sampler_draw(uint64_t seed, uint64_t step, uint32_t b, uint32_t attempt,
                             uint32_t num_item) {
  const uint64_t h = mix64(seed ^ mix64(step ^ mix64(((uint64_t)b << 32) | attempt)));
  return (uint32_t)(((h >> 32) * (uint64_t)num_item) >> 32);
}
void oracle_sample_negatives(const int64_t *hist_rowptr, const int32_t *hist_col,
                             const int64_t *users, int32_t B, int32_t num_item, uint64_t seed,
                             uint64_t step, int64_t id_offset, int64_t *out_neg) {
  for (int32_t b = 0; b < B; ++b) {
    const int64_t u = users[b];
    uint32_t cand = 0;
    for (uint32_t attempt = 0;; ++attempt) {
      cand = sampler_draw(seed, step, (uint32_t)b, attempt, (uint32_t)num_item);
      int seen = 0; /* linear scan, as `neg_item not in self.user_item_dict[user]` (a list) */
      for (int64_t e = hist_rowptr[u]; e < hist_rowptr[u + 1]; ++e)
        if ((uint32_t)hist_col[e] == cand) { seen = 1; break; }
      if (!seen || attempt >= 4096u) break;
    }
    out_neg[b] = (int64_t)cand + id_offset;
  }
}

/* ---- R scoring + mask + top-K ----------------------------------------------------------
 * Model/LightGCN.py:147-155: score = user @ item.T; score[row][hist] = 1e-6; topk(50).
 * The dot product's summation order is implementation-defined in torch (BLAS); this build
 * fixes it to the f32-MFMA chain order (two interleaved half-ranges of k, fused multiply-add)
 * so kernel and oracle agree bit-for-bit; vs torch the difference is rounding-level. */
float oracle_score_dot(const float *u, const float *i, int32_t D) {
  float acc = 0.0f;
  const int32_t chunk = D <= 128 ? D : 64; /* K-dim chunking of the kernel, see chaorec_hip.h */
  const int32_t half = chunk / 2;
  for (int32_t b = 0; b < D; b += chunk) {
    for (int32_t s = 0; s < half; ++s) {
      acc = fmaf(i[b + s], u[b + s], acc);
      acc = fmaf(i[b + half + s], u[b + half + s], acc);
    }
  }
  return acc;
}

typedef struct { float v; int64_t i; } oracle_pair;
static int pair_cmp(const void *a, const void *b) {
  const oracle_pair *x = (const oracle_pair *)a, *y = (const oracle_pair *)b;
  if (x->v > y->v) return -1;
  if (x->v < y->v) return 1;
  return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0); /* ties: lowest index first */
}

void oracle_score_topk_f32(const float *user_emb, const float *item_emb, int64_t n_users,
                           int64_t n_items, int32_t D, const int64_t *hist_rowptr,
                           const int32_t *hist_col, float mask_value, int32_t K,
                           int64_t id_offset, int64_t *out_idx, float *out_val) {
#pragma omp parallel
  {
    oracle_pair *row = (oracle_pair *)malloc((size_t)n_items * sizeof(oracle_pair));
#pragma omp for schedule(dynamic, 16)
    for (int64_t u = 0; u < n_users; ++u) {
      for (int64_t i = 0; i < n_items; ++i) {
        row[i].v = oracle_score_dot(user_emb + (size_t)u * D, item_emb + (size_t)i * D, D);
        row[i].i = i;
      }
      if (hist_rowptr)
        for (int64_t e = hist_rowptr[u]; e < hist_rowptr[u + 1]; ++e) row[hist_col[e]].v = mask_value;
      qsort(row, (size_t)n_items, sizeof(oracle_pair), pair_cmp);
      for (int32_t k = 0; k < K; ++k) {
        out_idx[(size_t)u * K + k] = row[k].i + id_offset;
        out_val[(size_t)u * K + k] = row[k].v;
      }
    }
    free(row);
  }
}

/* ---- dense GEMM (nn.Linear fwd/bwd): k-ascending fmaf chain ---------------------------- */
void oracle_gemm_f32(const float *A, const float *B, float *C, const float *bias, int64_t M,
                     int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t transA,
                     int32_t transB, int32_t accumulate, int32_t act) {
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m) {
    for (int64_t n = 0; n < N; ++n) {
      float acc = 0.0f;
      for (int64_t k = 0; k < K; ++k) {
        const float a = transA ? A[k * lda + m] : A[m * lda + k];
        const float b = transB ? B[n * ldb + k] : B[k * ldb + n];
        acc = fmaf(a, b, acc);
      }
      if (bias) acc = acc + bias[n];
      if (accumulate) acc = C[m * ldc + n] + acc;
      if (act == 1) acc = acc > 0.0f ? acc : acc * 0.01f;
      if (act == 2) acc = acc > 0.0f ? acc : acc * 0.2f;   /* nn.LeakyReLU(0.2), Model/NGCF.py:32 */
      C[m * ldc + n] = acc;
    }
  }
}

/* ---- Adam (torch.optim.Adam defaults, main.py:397) -------------------------------------- */
void oracle_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr,
                          float b1, float b2, float eps, float wd, int32_t step) {
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  const float fbc1 = (float)bc1, fbc2s = (float)sqrt(bc2);
  for (int64_t i = 0; i < n; ++i) {
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.0f) gi = gi + wd * pi;
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / fbc2s + eps;
    p[i] = pi - (lr / fbc1) * (mi / denom);
  }
}
