// Fused BPR step + negative sampler.
//
// Replaces: Model/LightGCN.py:97-121 (variant 0: log(sigmoid(d)+1e-5) + L2 means),
//           Model/FREEDOM.py:185-192 (variant 1: logsigmoid), Model/MMGCN.py:188-202
//           (variant 2: log(sigmoid)), their autograd backward, and the python rejection
//           sampler dataload.py:74-79.
//
// One wave64 per (user,pos,neg) triple: at D=64 a lane owns one feature, a row gather is one
// 256-B coalesced read, the three dot products are wave butterflies (no LDS).  The batch is
// launch/latency-bound (0.79 MB gathered at B=1024), so the forward is two launches with a
// fixed reduction order (bit-reproducible loss) and the backward is one launch of 256-B
// atomic row adds (the full-rate atomic shape on gfx950).
#include "common.h"
#include <cstdlib>

namespace chaorec {

__device__ __forceinline__ float sigmoidf_acc(float d) { return 1.0f / (1.0f + expf(-d)); }

// the wave of triple b: three row gathers, five wave butterflies, lane 0 writes the term / coefficient / L2 parts
__device__ __forceinline__ float bpr_terms_wave(const float *__restrict__ tab_u, const float *__restrict__ tab_i,
                                                int64_t iu, int64_t ip, int64_t in, int b, int B, int D, int variant,
                                                float *__restrict__ coef, float *__restrict__ ws) {
  const int lane = threadIdx.x & 63;
  const float *pu = tab_u + (size_t)iu * D;
  const float *pp = tab_i + (size_t)ip * D;
  const float *pn = tab_i + (size_t)in * D;
  float sp = 0.f, sn = 0.f, ru = 0.f, rp = 0.f, rn = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float u = pu[k], p = pp[k], n = pn[k];
    sp += u * p;
    sn += u * n;
    ru += u * u;
    rp += p * p;
    rn += n * n;
  }
  sp = wave_sum(sp);
  sn = wave_sum(sn);
  ru = wave_sum(ru);
  rp = wave_sum(rp);
  rn = wave_sum(rn);
  float c = 0.f;
  if (lane == 0) {
    const float d = sp - sn;
    const float invB = 1.0f / (float)B;
    float term;
    if (variant == CHAOREC_BPR_LOG_SIGMOID_EPS) {
      const float s = sigmoidf_acc(d);
      term = logf(s + 1e-5f);
      c = -invB * (s * (1.0f - s)) / (s + 1e-5f);
    } else if (variant == CHAOREC_BPR_LOGSIGMOID) {
      // logsigmoid(d) = min(d,0) - log1p(exp(-|d|))
      term = fminf(d, 0.f) - log1pf(expf(-fabsf(d)));
      c = -invB * sigmoidf_acc(-d);
    } else {
      const float s = sigmoidf_acc(d);
      term = logf(s);
      c = -invB * (1.0f - s);
    }
    coef[b] = c;
    ws[b] = term;
    ws[B + b] = ru;
    ws[2 * B + b] = rp;
    ws[3 * B + b] = rn;
  }
  return __shfl(c, 0, 64);
}

__global__ __launch_bounds__(256) void bpr_fwd_terms_kernel(
    const float *__restrict__ tab_u, const float *__restrict__ tab_i,
    const int64_t *__restrict__ users, const int64_t *__restrict__ pos,
    const int64_t *__restrict__ neg, int B, int D, int variant, float *__restrict__ coef,
    float *__restrict__ ws) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  bpr_terms_wave(tab_u, tab_i, users[b], pos[b], neg[b], b, B, D, variant, coef, ws);
}

// draw_triple is defined below (sampler section)
__device__ __forceinline__ void draw_triple(const int64_t *__restrict__, int64_t, const int64_t *__restrict__,
                                            const int32_t *__restrict__, int64_t, uint32_t, uint64_t, uint64_t,
                                            uint32_t, int64_t &, int64_t &, int64_t &, const int64_t *__restrict__,
                                            int64_t);

// The same forward with the batch drawn IN the launch: lane 0 of triple b's wave picks the training edge and rejects
// negatives against the user's history, the wave shares the three ids by shuffle, gathers and reduces.  The ids are
// written out for the backward launch.  One launch less per step than draw_batch + bpr_fwd_terms.
__global__ __launch_bounds__(256) void bpr_fwd_terms_drawn_kernel(
    const float *__restrict__ tab_u, const float *__restrict__ tab_i, const int64_t *__restrict__ edges,
    int64_t n_edges, const int64_t *__restrict__ hist_rowptr, const int32_t *__restrict__ hist_col, int64_t num_user,
    uint32_t num_item, uint64_t seed, uint64_t step, const int64_t *__restrict__ step_dev,
    int64_t *__restrict__ out_users, int64_t *__restrict__ out_pos, int64_t *__restrict__ out_neg, int B, int D,
    int variant, float *__restrict__ coef, float *__restrict__ ws, const int64_t *__restrict__ perm,
    const int64_t *__restrict__ perm_pos) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  if (step_dev) step += (uint64_t)step_dev[0];
  int64_t u = 0, p = 0, n = 0;
  if (lane == 0) {
    draw_triple(edges, n_edges, hist_rowptr, hist_col, num_user, num_item, seed, step, (uint32_t)b, u, p, n, perm,
                perm ? perm_pos[0] : 0);
    out_users[b] = u;
    out_pos[b] = p;
    out_neg[b] = n;
  }
  u = __shfl(u, 0, 64);
  p = __shfl(p, 0, 64);
  n = __shfl(n, 0, 64);
  bpr_terms_wave(tab_u, tab_i, u, p, n, b, B, D, variant, coef, ws);
}

// Forward terms AND the backward's row updates in one launch, for a loss that is differentiated with d(loss) = 1
// (a plain loss.backward()): the coefficient of sample b depends on its own score difference only, never on the
// batch total, so the wave that just reduced the triple adds its three gradient rows right away -- the same
// 256-B atomic row adds, the same values as bpr_bwd_kernel with grad_out = 1.  g_u / g_i must be zero where no
// sample lands (the caller keeps such a buffer: see chaorec_spmm_csr_adam_f32's clear_z).
__global__ __launch_bounds__(256) void bpr_fwd_bwd_drawn_kernel(
    const float *__restrict__ tab_u, const float *__restrict__ tab_i, const int64_t *__restrict__ edges,
    int64_t n_edges, const int64_t *__restrict__ hist_rowptr, const int32_t *__restrict__ hist_col, int64_t num_user,
    uint32_t num_item, uint64_t seed, uint64_t step, const int64_t *__restrict__ step_dev,
    const int64_t *__restrict__ in_users, const int64_t *__restrict__ in_pos, const int64_t *__restrict__ in_neg,
    int64_t *__restrict__ out_users, int64_t *__restrict__ out_pos, int64_t *__restrict__ out_neg, int B, int D,
    int variant, float reg_weight, float *__restrict__ coef, float *__restrict__ ws, const int64_t *__restrict__ perm,
    const int64_t *__restrict__ perm_pos, float *g_u, float *g_i, int32_t *__restrict__ adam_step, float beta1,
    float beta2, float *__restrict__ adam_bc, int64_t pos_offset, uint32_t *row_bits, int64_t bits_item_offset) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // the optimizer's step count and this step's bias corrections (two double pow()s on one otherwise idle thread,
  // under the other lanes' draw + gathers): nothing else in this launch reads them, the Adam epilogue comes later
  if (adam_step && blockIdx.x == 0 && threadIdx.x == blockDim.x - 1) {
    const int st = adam_step[0] + 1;
    adam_step[0] = st;
    adam_bias_corrections(st, beta1, beta2, adam_bc[0], adam_bc[1]);
  }
  if (b >= B) return;
  int64_t u = 0, p = 0, n = 0;
  if (edges) {
    if (step_dev) step += (uint64_t)step_dev[0];
    if (lane == 0) {
      draw_triple(edges, n_edges, hist_rowptr, hist_col, num_user, num_item, seed, step, (uint32_t)b, u, p, n, perm,
                  perm ? perm_pos[0] + pos_offset : 0);
      out_users[b] = u;
      out_pos[b] = p;
      out_neg[b] = n;
    }
    u = __shfl(u, 0, 64);
    p = __shfl(p, 0, 64);
    n = __shfl(n, 0, 64);
  } else {
    u = in_users[b];
    p = in_pos[b];
    n = in_neg[b];
  }
  const float c = bpr_terms_wave(tab_u, tab_i, u, p, n, b, B, D, variant, coef, ws) * 1.0f;
  const float r2 = 1.0f * 2.0f * reg_weight / ((float)B * (float)D);
  const size_t ou = (size_t)u * D, op = (size_t)p * D, on = (size_t)n * D;
  if (g_u) {                               // (NULL: the ordered launch that follows adds the rows -- CHAOREC_BPR_ORDERED=2)
    for (int k = lane; k < D; k += 64) {
      const float uu = tab_u[ou + k], pp = tab_i[op + k], nn = tab_i[on + k];
      atomicAdd(g_u + ou + k, c * (pp - nn) + r2 * uu);
      atomicAdd(g_i + op + k, c * uu + r2 * pp);
      atomicAdd(g_i + on + k, -c * uu + r2 * nn);
    }
  }
  // the rows of the gradient buffer this sample touched, for the row-sparse backward propagates
  // (chaorec_spmm_csr_rowsparse_f32): bit r of a bitmap over the joined table's rows, items from bits_item_offset on
  if (row_bits && lane < 3) {
    const int64_t r = lane == 0 ? u : bits_item_offset + (lane == 1 ? p : n);
    atomicOr(row_bits + (r >> 5), 1u << (r & 31));
  }
}

// One block, fixed order: thread t sums elements t, t+256, ... then a fixed LDS tree.
__global__ __launch_bounds__(256) void bpr_fwd_finalize_kernel(const float *__restrict__ ws, int B,
                                                               int D, float reg_weight,
                                                               float *__restrict__ out_loss,
                                                               float *__restrict__ out_total,
                                                               int64_t *__restrict__ advance,
                                                               int64_t *__restrict__ advance_pos,
                                                               float *__restrict__ loss_accum = nullptr,
                                                               int32_t *__restrict__ adam_step = nullptr,
                                                               float beta1 = 0.f, float beta2 = 0.f,
                                                               float *__restrict__ adam_bc = nullptr) {
  __shared__ float red[4][256];
  const int t = threadIdx.x;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = t; i < B; i += 256) {
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] += ws[q * B + i];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[q][t] = a[q];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
#pragma unroll
      for (int q = 0; q < 4; ++q) red[q][t] += red[q][t + s];
    }
    __syncthreads();
  }
  if (t == 0) {
    const float bpr = -red[0][0] / (float)B;
    const float denom = (float)B * (float)D;
    float reg = 0.f;
    if (reg_weight != 0.f) reg = reg_weight * (red[1][0] / denom + red[2][0] / denom + red[3][0] / denom);
    out_loss[0] = bpr + reg;
    out_loss[1] = bpr;
    out_loss[2] = reg;
    if (out_total) out_total[0] = bpr + reg;
    // every wave of the terms launch has read the batch counter by now: move it on for the next step
    if (advance) advance[0] += 1;
    if (advance_pos) advance_pos[0] += B;       // the epoch permutation's read position
    if (loss_accum) loss_accum[0] += bpr + reg;  // per-epoch loss bookkeeping (train_and_evaluate.py:48 without the sync)
    if (adam_step) {                             // the optimizer's step count and this step's bias corrections
      const int st = adam_step[0] + 1;
      adam_step[0] = st;
      adam_bias_corrections(st, beta1, beta2, adam_bc[0], adam_bc[1]);
    }
  }
}

// The same reduction for n_steps consecutive steps' workspaces in ONE launch (a replay of k captured steps runs its
// loss bookkeeping once: nothing later in a step reads what the finalize writes, only the NEXT step's batch draw does,
// and that reads `advance` / `advance_pos` plus a per-launch offset).  One WORKGROUP per step reduces that step's workspace
// -- the single-step kernel's sums, thread for thread and tree level for tree level -- and parks (bpr, reg) in `scratch`;
// the last workgroup to arrive (a ticket in scratch) then does the bookkeeping step by step, in order: the same sequence of
// additions into loss_accum as n_steps single launches.  (One workgroup walking the k workspaces one after the other took
// 25 us per 10-step replay at sports size, 2 % of the replay, all of it on the critical path.)
__global__ __launch_bounds__(256) void bpr_fwd_finalize_steps_kernel(const float *__restrict__ ws, int64_t ws_stride,
                                                                     int n_steps, int B, int D, float reg_weight,
                                                                     float *__restrict__ out_loss,
                                                                     float *__restrict__ out_total,
                                                                     int64_t *__restrict__ advance,
                                                                     int64_t *__restrict__ advance_pos,
                                                                     float *__restrict__ loss_accum, float *scratch) {
  __shared__ float red[4][256];
  __shared__ int last_s;
  const int t = threadIdx.x, st = blockIdx.x;
  int *ticket = reinterpret_cast<int *>(scratch + 2 * n_steps);
  {
    const float *w = ws + (size_t)st * ws_stride;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < B; i += 256) {
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] += w[q * B + i];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[q][t] = a[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[q][t] += red[q][t + s];
      }
      __syncthreads();
    }
    if (t == 0) {
      const float bpr = -red[0][0] / (float)B;
      const float denom = (float)B * (float)D;
      float reg = 0.f;
      if (reg_weight != 0.f) reg = reg_weight * (red[1][0] / denom + red[2][0] / denom + red[3][0] / denom);
      scratch[2 * st] = bpr;
      scratch[2 * st + 1] = reg;
      __threadfence();
      last_s = atomicAdd(ticket, 1) == n_steps - 1;
    }
    __syncthreads();
  }
  if (!last_s || t != 0) return;
  __threadfence();
  for (int s2 = 0; s2 < n_steps; ++s2) {
    const float bpr = __hip_atomic_load(scratch + 2 * s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float reg = __hip_atomic_load(scratch + 2 * s2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s2 == n_steps - 1) {
      out_loss[0] = bpr + reg;
      out_loss[1] = bpr;
      out_loss[2] = reg;
      if (out_total) out_total[0] = bpr + reg;
    }
    if (loss_accum) loss_accum[0] += bpr + reg;
  }
  if (advance) advance[0] += n_steps;
  if (advance_pos) advance_pos[0] += (int64_t)n_steps * B;
  *ticket = 0;
}

__global__ __launch_bounds__(256) void bpr_bwd_kernel(
    const float *__restrict__ tab_u, const float *__restrict__ tab_i,
    const int64_t *__restrict__ users, const int64_t *__restrict__ pos,
    const int64_t *__restrict__ neg, int B, int D, const float *__restrict__ coef,
    float reg_weight, const float *__restrict__ grad_out, float *g_u, float *g_i) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const float go = grad_out ? grad_out[0] : 1.0f;
  const float c = coef[b] * go;
  const float r2 = go * 2.0f * reg_weight / ((float)B * (float)D);
  const size_t ou = (size_t)users[b] * D, op = (size_t)pos[b] * D, on = (size_t)neg[b] * D;
  for (int k = lane; k < D; k += 64) {
    const float u = tab_u[ou + k], p = tab_i[op + k], n = tab_i[on + k];
    atomicAdd(g_u + ou + k, c * (p - n) + r2 * u);
    atomicAdd(g_i + op + k, c * u + r2 * p);
    atomicAdd(g_i + on + k, -c * u + r2 * n);
  }
}

// ---- several BPR terms over one user table (Model/FREEDOM.py:203-215: mf_loss + reg_weight * (mf_t_loss + mf_v_loss)) ----
// T item tables (the propagated id embeddings, the projected text rows, the projected image rows), each with its own
// positive / negative row ids, share the user table and the batch's users.  One launch for all T * B triples, one
// finalize that also forms  sum_k w_k * loss_k, one backward launch: 4 launches per step where T calls of
// chaorec_bpr_fwd_f32 / chaorec_bpr_bwd_f32 plus the weighted sum in torch took 13.
constexpr int kBprMaxTerms = 4;
struct BprMultiArgs {
  const float *tab_i[kBprMaxTerms];
  const int64_t *pos[kBprMaxTerms];
  const int64_t *neg[kBprMaxTerms];
  float *g_i[kBprMaxTerms];
  // backward only: term k's item table is a block of rows GATHERED from a longer table (FREEDOM's projected batch rows,
  // ops.linear_rows): row r of it is row scatter_rows[k][r] there, and its gradient is ALSO added into
  // scatter_out[k][scatter_rows[k][r]] -- the zero-filled [n_table_rows, D] buffer + index_add_ of the gathering node's
  // backward, inside this launch.  NULL: no scatter.
  const int64_t *scatter_rows[kBprMaxTerms];
  float *scatter_out[kBprMaxTerms];
  int T;
};

__global__ __launch_bounds__(256) void bpr_multi_fwd_terms_kernel(const float *__restrict__ tab_u,
                                                                  const int64_t *__restrict__ users, const BprMultiArgs A,
                                                                  int B, int D, int variant, float *__restrict__ coef,
                                                                  float *__restrict__ ws) {
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int k = w / B, b = w - k * B;
  if (k >= A.T) return;
  bpr_terms_wave(tab_u, A.tab_i[k], users[b], A.pos[k][b], A.neg[k][b], b, B, D, variant, coef + (size_t)k * B,
                 ws + (size_t)k * 4 * B);
}

// one block; term by term the reduction of bpr_fwd_finalize_kernel (thread t sums elements t, t + 256, ... then a fixed LDS
// tree), then total = ((w_0 l_0 + w_1 l_1) + w_2 l_2) + ...
__global__ __launch_bounds__(256) void bpr_multi_finalize_kernel(const float *__restrict__ ws, int T, int B,
                                                                 const float *__restrict__ wvec,
                                                                 float *__restrict__ losses, float *__restrict__ out_total) {
  __shared__ float red[256];
  const int t = threadIdx.x;
  float total = 0.f;
  for (int k = 0; k < T; ++k) {
    const float *w = ws + (size_t)k * 4 * B;
    float a = 0.f;
    for (int i = t; i < B; i += 256) a += w[i];
    red[t] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) red[t] += red[t + s];
      __syncthreads();
    }
    if (t == 0) {
      const float l = -red[0] / (float)B;
      losses[k] = l;
      total = k == 0 ? l * wvec[0] : total + l * wvec[k];
    }
    __syncthreads();
  }
  if (t == 0) out_total[0] = total;
}

__global__ __launch_bounds__(256) void bpr_multi_bwd_kernel(const float *__restrict__ tab_u, const int64_t *__restrict__ users,
                                                            const BprMultiArgs A, int B, int D, const float *__restrict__ coef,
                                                            const float *__restrict__ wvec, const float *__restrict__ grad_out,
                                                            float *g_u) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int k = w / B, b = w - k * B;
  if (k >= A.T) return;
  const float go = (grad_out ? grad_out[0] : 1.0f) * wvec[k];
  const float c = coef[(size_t)k * B + b] * go;
  const float *tab_i = A.tab_i[k];
  float *g_i = A.g_i[k];
  const int64_t pr = A.pos[k][b], nr = A.neg[k][b];
  const size_t ou = (size_t)users[b] * D, op = (size_t)pr * D, on = (size_t)nr * D;
  const int64_t *srows = A.scatter_rows[k];
  float *sout = A.scatter_out[k];
  size_t sp = 0, sn = 0;
  if (srows) sp = (size_t)srows[pr] * D, sn = (size_t)srows[nr] * D;     // wave-uniform
  for (int q = lane; q < D; q += 64) {
    const float u = tab_u[ou + q], p = tab_i[op + q], n = tab_i[on + q];
    atomicAdd(g_u + ou + q, c * (p - n));
    atomicAdd(g_i + op + q, c * u);
    atomicAdd(g_i + on + q, -c * u);
    if (srows) {
      atomicAdd(sout + sp + q, c * u);
      atomicAdd(sout + sn + q, -c * u);
    }
  }
}

// ---- ORDERED backward: the same row sums without atomics ---------------------------------------------------------
// The atomic row adds above are order-dependent once THREE addends meet in one element ((a + b) + c != a + (c + b)); the
// order the memory system applies them in moves with the load on the chip, so two runs of one training step can differ
// in the last bit of a few gradient elements -- and Adam, in the first steps of a run, turns a last-bit difference of an
// element that is (nearly) all cancellation into a difference of a fraction of lr in the parameter (round 6: the rare
// "two-stream divergence" of the MMGCN steps and VBPR's run-to-run spread were this; it shows in one-stream runs as
// often, tools/stream_stress.py, profiles/r06_stream_bisect.txt).  The ordered launch gives every destination row ONE
// owner wave that adds the row's contributions in ascending slot order:
//   * a slot = one row contribution of one sample; the slots of a GROUP (one destination buffer) are numbered
//     role-major, sample-minor;
//   * every workgroup computes the destination-row KEY (the row's address) of every slot of its group into LDS, then
//     each of its 16 waves takes one slot: a wave whose key occurs at a lower slot leaves; the first occurrence scans the
//     keys upwards from its own slot and adds the contribution of every slot with its key to the row (read once, written
//     once, no atomic) -- the expressions are the atomic kernels', only the order is fixed.
// O(n^2 / 64) key compares per group of n slots: 3 B = 3072 slots cost ~3 us of the chip.  Groups beyond kOrdMaxSlots (the
// keys must fit LDS) stay with the atomic launch.
constexpr int kOrdMaxSlots = 16384;         // 128 KB of 8-byte keys
constexpr int kOrdWaves = 16;

struct OrdArgs {
  const float *tab_u;
  const int64_t *users;
  BprMultiArgs A;                           // tab_i / pos / neg / g_i / scatter per term
  const float *coef;                        // [T, B]
  const float *wvec;                        // [T] or NULL (single term: 1)
  const float *grad_out;                    // device scalar or NULL
  float *g_u;
  float r2_unit;                            // single term: 2 reg / (B D), times grad_out; multi: 0
  int B, D, T;
  int joined;                               // 1: ONE group of 3 B slots (users, pos, neg of term 0: g_u and g_i may alias)
  int n_groups;
  int wg_begin[2 * kBprMaxTerms + 2];       // first workgroup of group g (a 1-D grid: no workgroup without slots); [n_groups] = all
};
// groups (blockIdx.y) when !joined: 0 = the users of all terms (T B slots); 1 + 2 k = term k's item rows (2 B: pos, neg);
// 2 + 2 k = term k's scattered rows (2 B), empty without scatter_rows[k]
struct OrdSlot {
  int term, b, role;                        // role 0 user, 1 pos, 2 neg, 3 scattered pos, 4 scattered neg
};
__device__ __forceinline__ int ord_group_slots(const OrdArgs &P, int g) {
  if (P.joined) return 3 * P.B;
  if (g == 0) return P.T * P.B;
  const int k = (g - 1) >> 1;
  if (((g - 1) & 1) && !P.A.scatter_rows[k]) return 0;
  return 2 * P.B;
}
__device__ __forceinline__ OrdSlot ord_slot(const OrdArgs &P, int g, int j) {
  OrdSlot s;
  const int q = j / P.B;
  s.b = j - q * P.B;
  if (P.joined) {
    s.term = 0, s.role = q;
  } else if (g == 0) {
    s.term = q, s.role = 0;
  } else {
    s.term = (g - 1) >> 1;
    s.role = (((g - 1) & 1) ? 3 : 1) + q;
  }
  return s;
}
__device__ __forceinline__ float *ord_dst(const OrdArgs &P, const OrdSlot s) {
  const int D = P.D;
  if (s.role == 0) return P.g_u + (size_t)P.users[s.b] * D;
  const int64_t r = (s.role & 1) ? P.A.pos[s.term][s.b] : P.A.neg[s.term][s.b];
  if (s.role <= 2) return P.A.g_i[s.term] + (size_t)r * D;
  return P.A.scatter_out[s.term] + (size_t)P.A.scatter_rows[s.term][r] * D;
}

template <int NQ>
__global__ __launch_bounds__(64 * kOrdWaves) void bpr_bwd_ordered_kernel(const OrdArgs P) {
  extern __shared__ uint64_t ord_keys[];
  int g = 0;
  while (g + 1 < P.n_groups && (int)blockIdx.x >= P.wg_begin[g + 1]) ++g;       // (block-uniform, <= 9 steps)
  const int n = ord_group_slots(P, g);
  const int wg = (int)blockIdx.x - P.wg_begin[g];                                 // this workgroup's number inside its group
  for (int j = threadIdx.x; j < n; j += blockDim.x) ord_keys[j] = (uint64_t)reinterpret_cast<uintptr_t>(ord_dst(P, ord_slot(P, g, j)));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int me = wg * kOrdWaves + wave;
  if (me >= n) return;
  const uint64_t key = ord_keys[me];
  // the first occurrence of a key owns its row
  for (int base = 0; base < me; base += 64) {
    const int j = base + lane;
    if (__any(j < me && ord_keys[j] == key)) return;
  }
  float *dst = reinterpret_cast<float *>((uintptr_t)key);
  const int D = P.D;
  float acc[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) acc[i] = lane + 64 * i < D ? dst[lane + 64 * i] : 0.f;
  const float go0 = P.grad_out ? P.grad_out[0] : 1.0f;
  const float r2 = P.r2_unit * go0;
  auto contribution = [&](int j, float (&v)[NQ]) __attribute__((always_inline)) {
    const OrdSlot s = ord_slot(P, g, __builtin_amdgcn_readfirstlane(j));
    const float c = P.coef[(size_t)s.term * P.B + s.b] * (go0 * (P.wvec ? P.wvec[s.term] : 1.0f));
    const float *tab_i = P.A.tab_i[s.term];
    const size_t ou = (size_t)P.users[s.b] * D, op = (size_t)P.A.pos[s.term][s.b] * D, on = (size_t)P.A.neg[s.term][s.b] * D;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = lane + 64 * i;
      v[i] = 0.f;
      if (q < D) {
        if (s.role == 0) {
          const float p = tab_i[op + q], nn = tab_i[on + q];
          v[i] = P.joined ? c * (p - nn) + r2 * P.tab_u[ou + q] : c * (p - nn);
        } else {
          const float u = P.tab_u[ou + q];
          const float cu = (s.role & 1) ? c * u : -c * u;
          v[i] = (P.joined && s.role <= 2) ? cu + r2 * tab_i[((s.role & 1) ? op : on) + q] : cu;
        }
      }
    }
  };
  for (int base = me & ~63; base < n; base += 64) {
    const int j = base + lane;
    unsigned long long m = __ballot(j >= me && j < n && ord_keys[j] == key);
    while (m) {                              // wave-uniform; up to four contributions' loads in flight, added in slot order
      int t[4], cnt = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t[i] = 0;
        if (m) {
          t[i] = base + (int)__builtin_ctzll(m);
          m &= m - 1;
          cnt = i + 1;
        }
      }
      float v[4][NQ];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < cnt) contribution(t[i], v[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < cnt) {
#pragma unroll
          for (int x = 0; x < NQ; ++x) acc[x] += v[i][x];
        }
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i)
    if (lane + 64 * i < D) dst[lane + 64 * i] = acc[i];
}

// -> false: this shape stays with the atomic launch
static bool launch_bpr_ordered(OrdArgs &P, int groups, hipStream_t st) {
  int max_slots = 0, total_wg = 0;
  P.n_groups = groups;
  for (int g = 0; g < groups; ++g) {
    int n;
    if (P.joined) n = 3 * P.B;
    else if (g == 0) n = P.T * P.B;
    else n = (((g - 1) & 1) && !P.A.scatter_rows[(g - 1) >> 1]) ? 0 : 2 * P.B;
    P.wg_begin[g] = total_wg;
    total_wg += (n + kOrdWaves - 1) / kOrdWaves;
    max_slots = n > max_slots ? n : max_slots;
  }
  P.wg_begin[groups] = total_wg;
  if (max_slots > kOrdMaxSlots || P.D > 256 || total_wg == 0) return false;
  const dim3 grid((unsigned)total_wg);
  const size_t lds = (size_t)max_slots * sizeof(uint64_t);
  const int nq = (P.D + 63) / 64;
  if (nq <= 1) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bpr_bwd_ordered_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bpr_bwd_ordered_kernel<1>, grid, dim3(64 * kOrdWaves), lds, st, P);
  } else if (nq == 2) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bpr_bwd_ordered_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bpr_bwd_ordered_kernel<2>, grid, dim3(64 * kOrdWaves), lds, st, P);
  } else {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bpr_bwd_ordered_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bpr_bwd_ordered_kernel<4>, grid, dim3(64 * kOrdWaves), lds, st, P);
  }
  return true;
}

// ---- sampler -------------------------------------------------------------------------------
// Counter-based generator: splitmix64 finaliser over (seed, step, b, attempt).  Stateless, so a
// draw does not depend on launch geometry or on how many draws other samples needed.
// (mix64: common.h)
__device__ __forceinline__ uint32_t sampler_draw(uint64_t seed, uint64_t step, uint32_t b,
                                                 uint32_t attempt, uint32_t num_item) {
  uint64_t h = mix64(seed ^ mix64(step ^ mix64(((uint64_t)b << 32) | attempt)));
  // Lemire multiply-shift onto [0, num_item): bias <= num_item / 2^32, far below test resolution
  return (uint32_t)(((h >> 32) * (uint64_t)num_item) >> 32);
}

__global__ __launch_bounds__(256) void sample_negatives_kernel(
    const int64_t *__restrict__ hist_rowptr, const int32_t *__restrict__ hist_col,
    const int64_t *__restrict__ users, int B, uint32_t num_item, uint64_t seed, uint64_t step,
    const int64_t *__restrict__ step_dev, int64_t id_offset, int64_t *__restrict__ out_neg) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (step_dev) step += (uint64_t)step_dev[0];  // device-resident batch counter: the launch is graph-capturable
  const int64_t u = users[b];
  const int64_t h0 = hist_rowptr[u], h1 = hist_rowptr[u + 1];
  uint32_t cand = 0;
  for (uint32_t attempt = 0;; ++attempt) {
    cand = sampler_draw(seed, step, (uint32_t)b, attempt, num_item);
    // binary search in the user's ascending history row
    int64_t lo = h0, hi = h1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((uint32_t)hist_col[mid] < cand) lo = mid + 1; else hi = mid;
    }
    const bool seen = (lo < h1) && ((uint32_t)hist_col[lo] == cand);
    if (!seen) break;
    if (attempt >= 4096u) break;  // a user who interacted with ~every item: give up, bounded
  }
  out_neg[b] = (int64_t)cand + id_offset;
}

// sample b of batch `step`: a training edge picked uniformly (user, positive as LOCAL item id) and one negative the
// user has not interacted with (rejection against the ascending history row, bounded)
__device__ __forceinline__ void draw_triple(const int64_t *__restrict__ edges, int64_t n_edges,
                                            const int64_t *__restrict__ hist_rowptr,
                                            const int32_t *__restrict__ hist_col, int64_t num_user, uint32_t num_item,
                                            uint64_t seed, uint64_t step, uint32_t b, int64_t &u, int64_t &p,
                                            int64_t &n, const int64_t *__restrict__ perm,
                                            int64_t perm_pos) {
  uint64_t idx;
  if (perm) {
    // epoch permutation (DataLoader(shuffle=True)): every edge once.  A read position past the end (a caller that
    // draws more batches than the epoch holds) is clamped, never dereferenced.
    const int64_t q = perm_pos + (int64_t)b;
    idx = (uint64_t)perm[q < n_edges ? (q < 0 ? 0 : q) : n_edges - 1];
  } else {
    const uint64_t hsel = mix64(seed ^ mix64(step ^ mix64(0xED6E5ull ^ ((uint64_t)b << 32))));
    // 64x64 -> high 64 multiply-shift onto [0, n_edges)
    idx = (uint64_t)(((unsigned __int128)hsel * (unsigned __int128)(uint64_t)n_edges) >> 64);
  }
  u = edges[2 * idx];
  p = edges[2 * idx + 1] - num_user;
  const int64_t h0 = hist_rowptr[u], h1 = hist_rowptr[u + 1];
  uint32_t cand = 0;
  for (uint32_t attempt = 0;; ++attempt) {
    cand = sampler_draw(seed, step, b, attempt, num_item);
    int64_t lo = h0, hi = h1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((uint32_t)hist_col[mid] < cand) lo = mid + 1; else hi = mid;
    }
    const bool seen = (lo < h1) && ((uint32_t)hist_col[lo] == cand);
    if (!seen || attempt >= 4096u) break;
  }
  n = (int64_t)cand;
}

// One launch per batch: pick B training edges uniformly (counter-based, with replacement across batches),
// gather (user, positive) and draw one negative each -- the DataLoader(shuffle) + TrainingDataset.__getitem__
// pair of main.py:194-195 / dataload.py:74-79 for a streaming trainer.  Outputs LOCAL ids (item - num_user),
// what Model.loss() computes first anyway.  The negative stream is sample_negatives_kernel's.
__global__ __launch_bounds__(256) void draw_batch_kernel(
    const int64_t *__restrict__ edges, int64_t n_edges, const int64_t *__restrict__ hist_rowptr,
    const int32_t *__restrict__ hist_col, int B, int64_t num_user, uint32_t num_item, uint64_t seed, uint64_t step,
    const int64_t *__restrict__ step_dev, int64_t *__restrict__ out_users, int64_t *__restrict__ out_pos,
    int64_t *__restrict__ out_neg, int64_t item_offset) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (step_dev) step += (uint64_t)step_dev[0];
  int64_t u, p, n;
  draw_triple(edges, n_edges, hist_rowptr, hist_col, num_user, num_item, seed, step, (uint32_t)b, u, p, n, nullptr, 0);
  out_users[b] = u;
  out_pos[b] = p + item_offset;      // (item_offset = num_user: the GLOBAL ids dataload.py:74-79 hands to Model.loss())
  out_neg[b] = n + item_offset;
}

// The batch BEFORE the forward (a training step whose forward is restricted to the rows its loss reads): draw_batch_kernel's
// triples -- or a given batch -- plus the three table rows of every sample flagged in a row bitmap (bit u, bit
// bits_item_offset + pos, bit bits_item_offset + neg) and every row flagged FIRST by this launch appended to a list (no
// duplicates, arbitrary order; *list_n zero on entry).  The triples are bpr_fwd_bwd_drawn_kernel's for the same
// (seed, step, permutation position): a step that draws here and hands the ids to that kernel trains on the same batch.
__global__ __launch_bounds__(256) void batch_rows_kernel(
    const int64_t *__restrict__ edges, int64_t n_edges, const int64_t *__restrict__ hist_rowptr,
    const int32_t *__restrict__ hist_col, int B, int64_t num_user, uint32_t num_item, uint64_t seed, uint64_t step,
    const int64_t *__restrict__ step_dev, const int64_t *__restrict__ perm, const int64_t *__restrict__ perm_pos,
    int64_t pos_offset, int64_t *users, int64_t *pos, int64_t *neg, uint32_t *row_bits, int64_t bits_item_offset,
    int32_t *list, int32_t *list_n, int64_t list_cap) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t u, p, n;
  if (edges) {
    if (step_dev) step += (uint64_t)step_dev[0];
    draw_triple(edges, n_edges, hist_rowptr, hist_col, num_user, num_item, seed, step, (uint32_t)b, u, p, n, perm,
                perm ? perm_pos[0] + pos_offset : 0);
    users[b] = u;
    pos[b] = p;
    neg[b] = n;
  } else {
    u = users[b];
    p = pos[b];
    n = neg[b];
  }
  const int64_t rows[3] = {u, bits_item_offset + p, bits_item_offset + n};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t m = 1u << (rows[k] & 31);
    const uint32_t old = atomicOr(row_bits + (rows[k] >> 5), m);
    if (list && !(old & m)) {
      const int at = atomicAdd(list_n, 1);
      if (at < list_cap) list[at] = (int32_t)rows[k];
    }
  }
}

// rows = [pos - offset ; neg - offset]: Model.loss()'s first lines (Model/FREEDOM.py:195-196: pos_items - self.num_user,
// neg_items - self.num_user) and the row list of the batch's 2 B items in one launch instead of three
__global__ __launch_bounds__(256) void shift_cat_kernel(const int64_t *__restrict__ pos, const int64_t *__restrict__ neg,
                                                        int64_t offset, int B, int64_t *__restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= 2 * B) return;
  out[b] = (b < B ? pos[b] : neg[b - B]) - offset;
}

}  // namespace chaorec

using namespace chaorec;

extern "C" int chaorec_draw_batch(const int64_t *edges, int64_t n_edges, const int64_t *hist_rowptr,
                                  const int32_t *hist_col, int32_t B, int64_t num_user, int32_t num_item,
                                  uint64_t seed, uint64_t step, const int64_t *step_dev, int64_t *out_users,
                                  int64_t *out_pos, int64_t *out_neg, int64_t item_offset, void *stream) {
  if (!edges || !hist_rowptr || !out_users || !out_pos || !out_neg)
    return fail(CHAOREC_E_INVALID, "draw_batch: NULL argument");
  if (B <= 0 || n_edges <= 0 || num_item <= 0) return fail(CHAOREC_E_INVALID, "draw_batch: bad sizes");
  hipLaunchKernelGGL(draw_batch_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, edges, n_edges,
                     hist_rowptr, hist_col, B, num_user, (uint32_t)num_item, seed, step, step_dev, out_users,
                     out_pos, out_neg, item_offset);
  return check_launch("draw_batch_kernel");
}

extern "C" int chaorec_batch_rows(const int64_t *edges, int64_t n_edges, const int64_t *hist_rowptr, const int32_t *hist_col,
                                  int32_t B, int64_t num_user, int32_t num_item, uint64_t seed, uint64_t step,
                                  const int64_t *step_dev, const int64_t *perm, const int64_t *perm_pos, int64_t pos_offset,
                                  int64_t *users, int64_t *pos, int64_t *neg, uint32_t *row_bits, int64_t bits_item_offset,
                                  int32_t *list, int32_t *list_n, int64_t list_cap, void *stream) {
  if (!users || !pos || !neg || !row_bits) return fail(CHAOREC_E_INVALID, "batch_rows: NULL argument");
  if (edges && !hist_rowptr) return fail(CHAOREC_E_INVALID, "batch_rows: a drawn batch needs the history rows");
  if (B <= 0 || (edges && (n_edges <= 0 || num_item <= 0))) return fail(CHAOREC_E_INVALID, "batch_rows: bad sizes");
  if ((list == nullptr) != (list_n == nullptr) || (list && list_cap <= 0))
    return fail(CHAOREC_E_INVALID, "batch_rows: list, list_n and list_cap come together");
  if ((perm == nullptr) != (perm_pos == nullptr)) return fail(CHAOREC_E_INVALID, "batch_rows: perm and perm_pos come together");
  hipLaunchKernelGGL(batch_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, edges, n_edges, hist_rowptr,
                     hist_col, B, num_user, (uint32_t)num_item, seed, step, step_dev, perm, perm_pos, pos_offset, users, pos,
                     neg, row_bits, bits_item_offset, list, list_n, list_cap);
  return check_launch("batch_rows_kernel");
}

extern "C" int chaorec_shift_cat_i64(const int64_t *pos, const int64_t *neg, int64_t offset, int32_t B, int64_t *out,
                                     void *stream) {
  if (!pos || !neg || !out) return fail(CHAOREC_E_INVALID, "shift_cat: NULL argument");
  if (B <= 0) return fail(CHAOREC_E_INVALID, "shift_cat: B=%d", B);
  hipLaunchKernelGGL(shift_cat_kernel, dim3((2 * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos, neg, offset, B,
                     out);
  return check_launch("shift_cat_kernel");
}

extern "C" int chaorec_bpr_fwd_f32(const float *tab_u, const float *tab_i, const int64_t *users,
                                   const int64_t *pos, const int64_t *neg, int32_t B, int32_t D,
                                   int32_t variant, float reg_weight, float *out_loss, float *out_total,
                                   float *coef, float *workspace, void *stream) {
  if (!tab_u || !tab_i || !users || !pos || !neg || !out_loss || !coef || !workspace)
    return fail(CHAOREC_E_INVALID, "bpr_fwd: NULL argument");
  if (B <= 0 || D <= 0) return fail(CHAOREC_E_INVALID, "bpr_fwd: B=%d D=%d", B, D);
  if (variant < 0 || variant > 2) return fail(CHAOREC_E_INVALID, "bpr_fwd: variant %d", variant);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bpr_fwd_terms_kernel, dim3((B + 3) / 4), dim3(256), 0, st, tab_u, tab_i, users,
                     pos, neg, B, D, variant, coef, workspace);
  int rc = check_launch("bpr_fwd_terms_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(bpr_fwd_finalize_kernel, dim3(1), dim3(256), 0, st, workspace, B, D, reg_weight,
                     out_loss, out_total, (int64_t *)nullptr, (int64_t *)nullptr);
  return check_launch("bpr_fwd_finalize_kernel");
}

extern "C" int chaorec_bpr_fwd_drawn_f32(const float *tab_u, const float *tab_i, const int64_t *edges, int64_t n_edges,
                                         const int64_t *hist_rowptr, const int32_t *hist_col, int64_t num_user,
                                         int32_t num_item, uint64_t seed, uint64_t step, const int64_t *step_dev,
                                         int32_t B, int32_t D, int32_t variant, float reg_weight, int64_t *out_users,
                                         int64_t *out_pos, int64_t *out_neg, float *out_loss, float *out_total,
                                         float *coef, float *workspace, int64_t *advance, const int64_t *perm,
                                         int64_t *perm_pos, void *stream) {
  if (!tab_u || !tab_i || !edges || !hist_rowptr || !out_users || !out_pos || !out_neg || !out_loss || !coef || !workspace)
    return fail(CHAOREC_E_INVALID, "bpr_fwd_drawn: NULL argument");
  if (perm && !perm_pos) return fail(CHAOREC_E_INVALID, "bpr_fwd_drawn: perm without perm_pos");
  if (B <= 0 || D <= 0 || n_edges <= 0 || num_item <= 0) return fail(CHAOREC_E_INVALID, "bpr_fwd_drawn: bad sizes");
  if (variant < 0 || variant > 2) return fail(CHAOREC_E_INVALID, "bpr_fwd_drawn: variant %d", variant);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bpr_fwd_terms_drawn_kernel, dim3((B + 3) / 4), dim3(256), 0, st, tab_u, tab_i, edges, n_edges,
                     hist_rowptr, hist_col, num_user, (uint32_t)num_item, seed, step, step_dev, out_users, out_pos,
                     out_neg, B, D, variant, coef, workspace, perm, (const int64_t *)perm_pos);
  int rc = check_launch("bpr_fwd_terms_drawn_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(bpr_fwd_finalize_kernel, dim3(1), dim3(256), 0, st, workspace, B, D, reg_weight, out_loss,
                     out_total, advance, (perm && advance) ? perm_pos : (int64_t *)nullptr);
  return check_launch("bpr_fwd_finalize_kernel");
}

extern "C" int chaorec_bpr_fwd_bwd_f32(const float *tab_u, const float *tab_i, const int64_t *edges, int64_t n_edges,
                                       const int64_t *hist_rowptr, const int32_t *hist_col, int64_t num_user,
                                       int32_t num_item, uint64_t seed, uint64_t step, const int64_t *step_dev,
                                       const int64_t *in_users, const int64_t *in_pos, const int64_t *in_neg,
                                       int32_t B, int32_t D, int32_t variant, float reg_weight, int64_t *out_users,
                                       int64_t *out_pos, int64_t *out_neg, float *coef, float *workspace,
                                       const int64_t *perm, const int64_t *perm_pos, float *g_u, float *g_i,
                                       int32_t *adam_step, float beta1, float beta2, float *adam_bc, void *stream) {
  return chaorec_bpr_fwd_bwd_at_f32(tab_u, tab_i, edges, n_edges, hist_rowptr, hist_col, num_user, num_item, seed, step,
                                    step_dev, in_users, in_pos, in_neg, B, D, variant, reg_weight, out_users, out_pos,
                                    out_neg, coef, workspace, perm, perm_pos, 0, g_u, g_i, adam_step, beta1, beta2, adam_bc,
                                    nullptr, 0, stream);
}

extern "C" int chaorec_bpr_bwd_ordered_f32(const float *tab_u, const float *tab_i, const int64_t *users, const int64_t *pos,
                                           const int64_t *neg, int32_t B, int32_t D, const float *coef, float reg_weight,
                                           const float *grad_out, float *g_u, float *g_i, void *stream);

extern "C" int chaorec_bpr_fwd_bwd_at_f32(const float *tab_u, const float *tab_i, const int64_t *edges, int64_t n_edges,
                                          const int64_t *hist_rowptr, const int32_t *hist_col, int64_t num_user,
                                          int32_t num_item, uint64_t seed, uint64_t step, const int64_t *step_dev,
                                          const int64_t *in_users, const int64_t *in_pos, const int64_t *in_neg,
                                          int32_t B, int32_t D, int32_t variant, float reg_weight, int64_t *out_users,
                                          int64_t *out_pos, int64_t *out_neg, float *coef, float *workspace,
                                          const int64_t *perm, const int64_t *perm_pos, int64_t pos_offset, float *g_u,
                                          float *g_i, int32_t *adam_step, float beta1, float beta2, float *adam_bc,
                                          uint32_t *row_bits, int64_t bits_item_offset, void *stream) {
  if (!tab_u || !tab_i || !coef || !workspace || !g_u || !g_i) return fail(CHAOREC_E_INVALID, "bpr_fwd_bwd: NULL argument");
  if (pos_offset < 0) return fail(CHAOREC_E_INVALID, "bpr_fwd_bwd: pos_offset %lld", (long long)pos_offset);
  if (adam_step && !adam_bc) return fail(CHAOREC_E_INVALID, "bpr_fwd_bwd: adam_step without adam_bc");
  if (edges) {
    if (!hist_rowptr || !out_users || !out_pos || !out_neg) return fail(CHAOREC_E_INVALID, "bpr_fwd_bwd: NULL draw argument");
    if (n_edges <= 0 || num_item <= 0) return fail(CHAOREC_E_INVALID, "bpr_fwd_bwd: bad sizes");
    if (perm && !perm_pos) return fail(CHAOREC_E_INVALID, "bpr_fwd_bwd: perm without perm_pos");
  } else if (!in_users || !in_pos || !in_neg) {
    return fail(CHAOREC_E_INVALID, "bpr_fwd_bwd: neither an edge list to draw from nor a batch");
  }
  if (B <= 0 || D <= 0) return fail(CHAOREC_E_INVALID, "bpr_fwd_bwd: B=%d D=%d", B, D);
  if (variant < 0 || variant > 2) return fail(CHAOREC_E_INVALID, "bpr_fwd_bwd: variant %d", variant);
  // CHAOREC_BPR_ORDERED=2: the fused steps' gradient rows too are added by the ordered, atomic-free launch (one more launch per
  // step -- ~7 us of a 121 us sports step, nothing of a configs[4] step --: a LightGCN run is then the same bits every time,
  // and the fused step equals the autograd step bit for bit also on batches with repeated rows).  Read per call.
  const char *env = std::getenv("CHAOREC_BPR_ORDERED");
  bool ordered = env && env[0] == '2' && 3 * (int64_t)B <= kOrdMaxSlots && D <= 256;
  const int64_t *ids_u = edges ? out_users : in_users, *ids_p = edges ? out_pos : in_pos, *ids_n = edges ? out_neg : in_neg;
  hipLaunchKernelGGL(bpr_fwd_bwd_drawn_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, tab_u, tab_i, edges,
                     n_edges, hist_rowptr, hist_col, num_user, (uint32_t)num_item, seed, step, step_dev, in_users, in_pos,
                     in_neg, out_users, out_pos, out_neg, B, D, variant, reg_weight, coef, workspace, perm, perm_pos,
                     ordered ? (float *)nullptr : g_u, g_i, adam_step, beta1, beta2, adam_bc, pos_offset, row_bits,
                     bits_item_offset);
  int rc = check_launch("bpr_fwd_bwd_drawn_kernel");
  if (rc || !ordered) return rc;
  return chaorec_bpr_bwd_ordered_f32(tab_u, tab_i, ids_u, ids_p, ids_n, B, D, coef, reg_weight, nullptr, g_u, g_i, stream);
}

extern "C" int chaorec_bpr_finalize_steps_f32(const float *workspace, int64_t ws_stride, int32_t n_steps, int32_t B,
                                              int32_t D, float reg_weight, float *out_loss, float *out_total,
                                              float *loss_accum, int64_t *advance, int64_t *perm_pos, float *scratch,
                                              void *stream) {
  if (!workspace || !out_loss || !scratch) return fail(CHAOREC_E_INVALID, "bpr_finalize_steps: NULL argument");
  if (B <= 0 || D <= 0 || n_steps < 1 || ws_stride < 4 * (int64_t)B)
    return fail(CHAOREC_E_INVALID, "bpr_finalize_steps: B=%d D=%d n_steps=%d ws_stride=%lld", B, D, n_steps, (long long)ws_stride);
  hipLaunchKernelGGL(bpr_fwd_finalize_steps_kernel, dim3(n_steps), dim3(256), 0, (hipStream_t)stream, workspace, ws_stride,
                     n_steps, B, D, reg_weight, out_loss, out_total, advance, perm_pos, loss_accum, scratch);
  return check_launch("bpr_fwd_finalize_steps_kernel");
}

extern "C" int chaorec_bpr_finalize_f32(const float *workspace, int32_t B, int32_t D, float reg_weight, float *out_loss,
                                        float *out_total, float *loss_accum, int64_t *advance, int64_t *perm_pos,
                                        int32_t *adam_step, float beta1, float beta2, float *adam_bc, void *stream) {
  if (!workspace || !out_loss) return fail(CHAOREC_E_INVALID, "bpr_finalize: NULL argument");
  if (B <= 0 || D <= 0) return fail(CHAOREC_E_INVALID, "bpr_finalize: B=%d D=%d", B, D);
  if (adam_step && !adam_bc) return fail(CHAOREC_E_INVALID, "bpr_finalize: adam_step without adam_bc");
  hipLaunchKernelGGL(bpr_fwd_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, B, D, reg_weight,
                     out_loss, out_total, advance, perm_pos, loss_accum, adam_step, beta1, beta2, adam_bc);
  return check_launch("bpr_fwd_finalize_kernel");
}

extern "C" int chaorec_bpr_bwd_f32(const float *tab_u, const float *tab_i, const int64_t *users,
                                   const int64_t *pos, const int64_t *neg, int32_t B, int32_t D,
                                   const float *coef, float reg_weight, const float *grad_out,
                                   float *g_u, float *g_i, void *stream) {
  if (!tab_u || !tab_i || !users || !pos || !neg || !coef || !g_u || !g_i)
    return fail(CHAOREC_E_INVALID, "bpr_bwd: NULL argument");
  if (B <= 0 || D <= 0) return fail(CHAOREC_E_INVALID, "bpr_bwd: B=%d D=%d", B, D);
  hipLaunchKernelGGL(bpr_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, tab_u,
                     tab_i, users, pos, neg, B, D, coef, reg_weight, grad_out, g_u, g_i);
  return check_launch("bpr_bwd_kernel");
}

extern "C" int chaorec_bpr_bwd_ordered_f32(const float *tab_u, const float *tab_i, const int64_t *users,
                                           const int64_t *pos, const int64_t *neg, int32_t B, int32_t D,
                                           const float *coef, float reg_weight, const float *grad_out,
                                           float *g_u, float *g_i, void *stream) {
  if (!tab_u || !tab_i || !users || !pos || !neg || !coef || !g_u || !g_i)
    return fail(CHAOREC_E_INVALID, "bpr_bwd_ordered: NULL argument");
  if (B <= 0 || D <= 0) return fail(CHAOREC_E_INVALID, "bpr_bwd_ordered: B=%d D=%d", B, D);
  OrdArgs P;
  P.tab_u = tab_u, P.users = users, P.coef = coef, P.wvec = nullptr, P.grad_out = grad_out, P.g_u = g_u;
  P.r2_unit = 2.0f * reg_weight / ((float)B * (float)D);
  P.B = B, P.D = D, P.T = 1, P.joined = 1;
  for (int k = 0; k < kBprMaxTerms; ++k) {
    P.A.tab_i[k] = tab_i, P.A.pos[k] = pos, P.A.neg[k] = neg, P.A.g_i[k] = g_i;
    P.A.scatter_rows[k] = nullptr, P.A.scatter_out[k] = nullptr;
  }
  P.A.T = 1;
  if (!launch_bpr_ordered(P, 1, (hipStream_t)stream))     // (too many slots for the key table / D > 256: atomics)
    return chaorec_bpr_bwd_f32(tab_u, tab_i, users, pos, neg, B, D, coef, reg_weight, grad_out, g_u, g_i, stream);
  return check_launch("bpr_bwd_ordered_kernel");
}

static int fill_multi(BprMultiArgs &A, int32_t T, const float *const *tabs, const int64_t *const *pos,
                      const int64_t *const *neg, float *const *g_i, const char *who) {
  if (T < 1 || T > kBprMaxTerms) return fail(CHAOREC_E_INVALID, "%s: T=%d must be in [1, %d]", who, T, kBprMaxTerms);
  if (!tabs || !pos || !neg) return fail(CHAOREC_E_INVALID, "%s: NULL argument", who);
  A.T = T;
  for (int k = 0; k < kBprMaxTerms; ++k) {
    const int j = k < T ? k : 0;
    if (!tabs[j] || !pos[j] || !neg[j] || (g_i && !g_i[j])) return fail(CHAOREC_E_INVALID, "%s: NULL term %d", who, j);
    A.tab_i[k] = tabs[j], A.pos[k] = pos[j], A.neg[k] = neg[j], A.g_i[k] = g_i ? g_i[j] : nullptr;
    A.scatter_rows[k] = nullptr, A.scatter_out[k] = nullptr;
  }
  return CHAOREC_OK;
}

extern "C" int chaorec_bpr_multi_fwd_f32(const float *tab_u, const int64_t *users, int32_t T, const float *const *tabs,
                                         const int64_t *const *pos, const int64_t *const *neg, int32_t B, int32_t D,
                                         int32_t variant, const float *wvec, float *losses, float *out_total, float *coef,
                                         float *workspace, void *stream) {
  if (!tab_u || !users || !wvec || !losses || !out_total || !coef || !workspace)
    return fail(CHAOREC_E_INVALID, "bpr_multi_fwd: NULL argument");
  if (B <= 0 || D <= 0) return fail(CHAOREC_E_INVALID, "bpr_multi_fwd: B=%d D=%d", B, D);
  if (variant < 0 || variant > 2) return fail(CHAOREC_E_INVALID, "bpr_multi_fwd: variant %d", variant);
  BprMultiArgs A;
  int rc = fill_multi(A, T, tabs, pos, neg, nullptr, "bpr_multi_fwd");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bpr_multi_fwd_terms_kernel, dim3((T * B + 3) / 4), dim3(256), 0, st, tab_u, users, A, B, D, variant,
                     coef, workspace);
  rc = check_launch("bpr_multi_fwd_terms_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(bpr_multi_finalize_kernel, dim3(1), dim3(256), 0, st, workspace, T, B, wvec, losses, out_total);
  return check_launch("bpr_multi_finalize_kernel");
}

extern "C" int chaorec_bpr_multi_bwd_f32(const float *tab_u, const int64_t *users, int32_t T, const float *const *tabs,
                                         const int64_t *const *pos, const int64_t *const *neg, int32_t B, int32_t D,
                                         const float *coef, const float *wvec, const float *grad_out, float *g_u,
                                         float *const *g_i, const int64_t *const *scatter_rows, float *const *scatter_out,
                                         void *stream) {
  if (!tab_u || !users || !wvec || !coef || !g_u || !g_i) return fail(CHAOREC_E_INVALID, "bpr_multi_bwd: NULL argument");
  if (B <= 0 || D <= 0) return fail(CHAOREC_E_INVALID, "bpr_multi_bwd: B=%d D=%d", B, D);
  if ((scatter_rows == nullptr) != (scatter_out == nullptr))
    return fail(CHAOREC_E_INVALID, "bpr_multi_bwd: scatter_rows and scatter_out come together");
  BprMultiArgs A;
  int rc = fill_multi(A, T, tabs, pos, neg, g_i, "bpr_multi_bwd");
  if (rc) return rc;
  if (scatter_rows) {
    for (int k = 0; k < T; ++k) {
      if ((scatter_rows[k] == nullptr) != (scatter_out[k] == nullptr))
        return fail(CHAOREC_E_INVALID, "bpr_multi_bwd: term %d has one of scatter_rows / scatter_out", k);
      A.scatter_rows[k] = scatter_rows[k], A.scatter_out[k] = scatter_out[k];
    }
  }
  hipLaunchKernelGGL(bpr_multi_bwd_kernel, dim3((T * B + 3) / 4), dim3(256), 0, (hipStream_t)stream, tab_u, users, A, B, D,
                     coef, wvec, grad_out, g_u);
  return check_launch("bpr_multi_bwd_kernel");
}

extern "C" int chaorec_bpr_multi_bwd_ordered_f32(const float *tab_u, const int64_t *users, int32_t T, const float *const *tabs,
                                                 const int64_t *const *pos, const int64_t *const *neg, int32_t B, int32_t D,
                                                 const float *coef, const float *wvec, const float *grad_out, float *g_u,
                                                 float *const *g_i, const int64_t *const *scatter_rows,
                                                 float *const *scatter_out, void *stream) {
  if (!tab_u || !users || !wvec || !coef || !g_u || !g_i) return fail(CHAOREC_E_INVALID, "bpr_multi_bwd_ordered: NULL argument");
  if (B <= 0 || D <= 0) return fail(CHAOREC_E_INVALID, "bpr_multi_bwd_ordered: B=%d D=%d", B, D);
  if ((scatter_rows == nullptr) != (scatter_out == nullptr))
    return fail(CHAOREC_E_INVALID, "bpr_multi_bwd_ordered: scatter_rows and scatter_out come together");
  OrdArgs P;
  int rc = fill_multi(P.A, T, tabs, pos, neg, g_i, "bpr_multi_bwd_ordered");
  if (rc) return rc;
  if (scatter_rows) {
    for (int k = 0; k < T; ++k) {
      if ((scatter_rows[k] == nullptr) != (scatter_out[k] == nullptr))
        return fail(CHAOREC_E_INVALID, "bpr_multi_bwd_ordered: term %d has one of scatter_rows / scatter_out", k);
      P.A.scatter_rows[k] = scatter_rows[k], P.A.scatter_out[k] = scatter_out[k];
    }
  }
  P.tab_u = tab_u, P.users = users, P.coef = coef, P.wvec = wvec, P.grad_out = grad_out, P.g_u = g_u;
  P.r2_unit = 0.f;
  P.B = B, P.D = D, P.T = T, P.joined = 0;
  if (!launch_bpr_ordered(P, 1 + 2 * T, (hipStream_t)stream))
    return chaorec_bpr_multi_bwd_f32(tab_u, users, T, tabs, pos, neg, B, D, coef, wvec, grad_out, g_u, g_i, scatter_rows,
                                     scatter_out, stream);
  return check_launch("bpr_bwd_ordered_kernel");
}

extern "C" int chaorec_sample_negatives(const int64_t *hist_rowptr, const int32_t *hist_col,
                                        const int64_t *users, int32_t B, int32_t num_item,
                                        uint64_t seed, uint64_t step, const int64_t *step_dev,
                                        int64_t id_offset, int64_t *out_neg, void *stream) {
  if (!hist_rowptr || !users || !out_neg) return fail(CHAOREC_E_INVALID, "sample: NULL argument");
  if (B <= 0 || num_item <= 0) return fail(CHAOREC_E_INVALID, "sample: B=%d num_item=%d", B, num_item);
  hipLaunchKernelGGL(sample_negatives_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     hist_rowptr, hist_col, users, B, (uint32_t)num_item, seed, step, step_dev, id_offset,
                     out_neg);
  return check_launch("sample_negatives_kernel");
}
