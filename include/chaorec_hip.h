/*
 * chaorec_hip.h -- C-ABI of libchaorec_hip.so: the MI355X (gfx950) kernels under
 * ChaoRec's GCN-propagation + BPR-training + full-rank-evaluation hot path.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are
 *     only ENQUEUED, nothing here synchronises or allocates;
 *   - the caller owns every buffer, including workspaces (sizes from the *_workspace_bytes
 *     queries, which are host-only and launch nothing);
 *   - return value: 0 = ok, <0 = error (CHAOREC_E_*); chaorec_last_error() returns a
 *     host string describing the last failure on the calling thread;
 *   - row-major dense tables, fp32, leading dimension == D unless an ld* argument exists.
 *
 * The reference (Ricardo-Ping/ChaoRec) is pure Python with no FFI layer: each entry point
 * below names the reference library call it replaces (file:line relative to the reference
 * root).  The Python binding a maintainer adds is shown in INTEGRATION.md.
 */
#ifndef CHAOREC_HIP_H
#define CHAOREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHAOREC_ABI_VERSION 16  /* 2: gemm workspace, score stats, rank metrics; 3: edge dropout, weighted sampling,
                                  row cosine, two-pass reductions, BPR forward with the batch drawn in the launch,
                                  SpMM dynamic-values mode, gemm act 2, 256-B aligned score workspace;
                                  4: SpMM with the Adam update in its epilogue, BPR forward + backward in one launch,
                                  stand-alone BPR finalize with loss / optimizer bookkeeping, layer mean in the last
                                  forward SpMM, scoring with carried thresholds, split-bf16 NT GEMM;
                                  5: weighted_sample_keys (sharded edge pruning);
                                  6: BPR batch at an offset + one finalize for k captured steps, NGCF elementwise backward;
                                  7: Adam over a feature table with a low-rank, row-sparse gradient (dense / lazy / flush), split-bf16 TN GEMM
                                     (weight gradients), multi-tensor Adam;
                                  8: rows_mean (layer mean of exchanged rows: fused sharded LightGCN step);
                                  9: split-bf16 NN GEMM (input gradients, accumulate epilogue), MMGCN's layer tail
                                     (leaky_cat_add / leaky_split_bwd), normalize_rows, multi-term BPR, draw_batch item_offset, shift_cat,
                                     peer-to-peer exchange kernels;
                                  10: multi-term BPR backward scatters the gradient of gathered row blocks itself;
                                  11: row-sparse backward propagates (row bitmaps written by the BPR launch, read by
                                      chaorec_spmm_csr_rowsparse_f32, cleared by chaorec_spmm_csr_adam_f32);
                                  12: the frontier-restricted step: chaorec_batch_rows (the batch before the forward), list
                                      launches with a layer-mean epilogue and workgroup-per-row launches for long rows,
                                      chaorec_expand_row_bits over rectangular blocks, rows_list_from_bits, rows_mean_by_bits,
                                      zero_rows_by_bits, rows_copy_by_bits, or_words, peer-to-peer exchange of flagged rows, frontier pack / unpack;
                                  13: scoring: CHAOREC_SCORE_FRONT / _BACK (one call as two phases), a raised-threshold pass for
                                      users whose candidate lists overflow (long item ranges), chaorec_score_topk_stats out10;
                                  14: chaorec_edge_dot_f32 (edge scores over a CSR's entries);
                                  15: ordered (atomic-free, run-to-run reproducible) BPR backward launches;
                                  16: chaorec_spmm_csr_rowlist_f32's long_cnt is int32[4] (column-striped very long rows) */

#define CHAOREC_OK 0
#define CHAOREC_E_INVALID (-1)     /* bad argument (NULL, negative size, unsupported D/K) */
#define CHAOREC_E_LAUNCH (-2)      /* hipLaunch / hipMemsetAsync reported an error */
#define CHAOREC_E_WORKSPACE (-3)   /* workspace too small */

int chaorec_abi_version(void);
const char *chaorec_last_error(void);

/* ---------------------------------------------------------------------------------------
 * P1/P7/P12: normalised-adjacency propagate  y = alpha * (A x) [+ beta * z],  A in CSR.
 *
 * Replaces: PyG MessagePassing.propagate = index_select + message (norm * x_j) + scatter_add
 *           (Model/LightGCN.py:40-43, BasicGCN.py:48-53,78-82) and torch.sparse.mm(adj, x)
 *           (Model/FREEDOM.py:168,174).
 *
 * Row r of the CSR is the DESTINATION node; its entries are the incoming edges in the
 * reference's edge order (stable sort by destination), so mode 0 reproduces the reference
 * CPU accumulation order:  s = 0; for e in row: s = s + (val[e] * x[col[e]])  -- product and
 * sum rounded separately (no FMA), exactly what message() then scatter_add_ do.
 * Epilogue, in this order, every step rounded to fp32:
 *     s = alpha * s                       (alpha == 1 is exact)
 *     if z:   s = s + (beta * z[r])
 *     y[r] = s                            (y may be NULL when only acc is wanted)
 *     if acc: a0 = acc_init ? (acc_w * acc_init[r]) : acc[r];   acc[r] = a0 + (acc_w * s)
 * The acc epilogue is LightGCN.forward's layer mean (Model/LightGCN.py:86-93):
 * final += (1/(L+1)) * x_l, with acc_init = x_0 on the first layer.
 * Backward (A symmetric for LightGCN/FREEDOM-ui/BasicGCN) is the same call with the
 * transposed CSR:  g_l = A^T g_{l+1} + beta * G.
 *
 * D must be a multiple of 4, 4 <= D <= 1024.  rowptr is int64 (nnz of config 5 > 2^31).
 * schedule (optional, may be NULL): the array built by chaorec_spmm_build_schedule() for this graph and D,
 *   copied to the device.  It fixes which workgroup walks which rows (longest rows first, one heavy group per
 *   workgroup) and carries a 64-B descriptor per row {row, degree, first entry, first 6 (col,val) pairs} so the
 *   kernel's dependent-load chain is descriptor -> x instead of order -> rowptr -> (col,val) -> x.  It changes
 *   scheduling and latency only, never results.
 * mode: 0 = ordered (bit-reproducible, reference order).  CHAOREC_SPMM_DYNAMIC_VALUES (1): same, but the schedule
 *   was built for this structure under OTHER values (the value array is rewritten every step, e.g. edge dropout):
 *   the kernel then ignores the values stored in the schedule and reads all of them from val[].
 * ------------------------------------------------------------------------------------- */
#define CHAOREC_SPMM_DYNAMIC_VALUES 1
int chaorec_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                         const float *x, float *y, int64_t n_rows, int64_t n_cols, int32_t D,
                         float alpha, const float *z, float beta,
                         float *acc, const float *acc_init, float acc_w,
                         const int32_t *schedule, int32_t mode, void *stream);

/* The LAST forward propagate of LightGCN.forward with the WHOLE layer mean in its epilogue (Model/LightGCN.py:86-93):
 *     s = (A x)[r];  a = w * terms[0][r];  a = a + w * terms[k][r] (k = 1 .. n_terms-1, in order);  mean_out[r] = a + w * s
 * -- the reference's accumulation order (final = 0 + w x_0 + w x_1 + ... + w x_L), every step rounded, so the result is
 * bit-identical to chaorec_spmm_csr_f32's per-layer acc epilogue; what it saves is that epilogue's read-modify-write of
 * the [N, D] mean in every earlier layer (terms = the earlier layers' outputs x_0 .. x_{L-1}, host array of DEVICE
 * pointers, n_terms <= 3, and <= 2 for D > 64: register budget).  y may be NULL (x_L itself is not needed). */
int chaorec_spmm_csr_mean_f32(const int64_t *rowptr, const int32_t *col, const float *val, const float *x,
                              float *y, int64_t n_rows, int64_t n_cols, int32_t D, float *mean_out,
                              const float *const *terms, int32_t n_terms, float w, const int32_t *schedule,
                              int32_t mode, void *stream);

/* The LAST backward propagate of a training step with the optimizer update in its epilogue
 * (train_and_evaluate.py:46-47: loss.backward(); optimizer.step(), for a model whose parameter table IS the
 * propagate's input: LightGCN, Model/LightGCN.py:76-95):
 *     g[r]   = alpha * (A x)[r] + beta * z[r]         -- chaorec_spmm_csr_f32's arithmetic, e.g. g_0 = A^T g_1 + w G
 *     Adam(param[r], g[r], exp_avg[r], exp_avg_sq[r]) -- chaorec_adam_step_f32's arithmetic, bit for bit
 * in one launch: no separate optimizer launch and no second pass over the gradient (grad_out may be NULL).
 * bias_corr: device float[2] = {1 - beta1^t, sqrt(1 - beta2^t)} of this step (chaorec_bpr_finalize_f32 writes it).
 * clear_z != 0: every row of z that was non-zero is zeroed after it was read -- z is then the batch gradient buffer
 * that chaorec_bpr_fwd_bwd_f32 adds into, kept all-zero between steps instead of being zero-filled every step
 * (refused when x == z: another wave could still gather the row).  D <= 256. */
/* clear_bits_a / _b (optional, n_words_* uint32 words each): row bitmaps of the row-sparse propagates below that this
 * launch zeroes as a side job -- it is the step's last launch, every reader of the step's bitmaps is done. */
int chaorec_spmm_csr_adam_f32(const int64_t *rowptr, const int32_t *col, const float *val, const float *x,
                              float *grad_out, int64_t n_rows, int64_t n_cols, int32_t D, float alpha,
                              float *z, float beta, const int32_t *schedule, int32_t mode, float *param,
                              float *exp_avg, float *exp_avg_sq, const float *bias_corr, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int32_t clear_z,
                              uint32_t *clear_bits_a, int64_t n_words_a, uint32_t *clear_bits_b, int64_t n_words_b,
                              void *stream);

/* A backward propagate whose operands are ROW-SPARSE (train_and_evaluate.py:46 loss.backward() through
 * Model/LightGCN.py:81-83 for a BPR batch): the batch gradient G has 3 B non-zero rows out of N, the first backward
 * propagate's result is non-zero in their neighbours only.  Same arithmetic, entry order and result as
 * chaorec_spmm_csr_f32's  y = alpha (A x) + beta z  (bit for bit, up to the sign of an exact zero); the bitmaps only gate
 * LOADS: a source row whose bit in src_bits is clear is not gathered (its term is val * (+0)), a row of z whose bit in
 * z_bits is clear is not read.  A set bit means "may be non-zero": supersets are fine.  out_bits (optional, all-zero on
 * entry) receives such a superset for y.  Bitmaps: uint32 words, bit r & 31 of word r >> 5; any of the three may be NULL.
 * 64 <= D <= 256. */
/* row_bits (optional): a superset of the rows of y that can be non-zero -- normally chaorec_expand_row_bits of src_bits.
 * A row whose bit is clear (and whose z row is not flagged) walks no entries at all; it stores zeros when write_zeros != 0
 * (the next launch reads y densely) and nothing otherwise (the next launch gathers flagged rows only).  Per-entry bitmap
 * tests alone leave such a launch at ~80 % of the dense one (a 4-byte L2 request per entry instead of a row gather); with
 * the row mask its cost follows the frontier, not the graph. */
int chaorec_spmm_csr_rowsparse_f32(const int64_t *rowptr, const int32_t *col, const float *val, const float *x,
                                   float *y, int64_t n_rows, int64_t n_cols, int32_t D, float alpha, const float *z,
                                   float beta, const int32_t *schedule, int32_t mode, const uint32_t *src_bits,
                                   const uint32_t *z_bits, uint32_t *out_bits, const uint32_t *row_bits,
                                   int32_t write_zeros, void *stream);

/* bits_out |= bits_self | {columns of the rows flagged in bits_in}: the rows a propagate can make non-zero when its source
 * is non-zero in the flagged rows only.  bits_in: a bitmap over the CSR's n_rows rows; bits_self (optional) and bits_out:
 * bitmaps over its n_out_rows COLUMNS (symmetric graph: bits_self = bits_in, n_out_rows = n_rows; a user shard's
 * rectangular blocks B_g / B_g^T: the batch rows of the other side).  Work ~ the flagged rows' entries.  list / list_n
 * (optional; *list_n zero on entry, list_cap entries): every row whose bit this launch sets first is also appended. */
int chaorec_expand_row_bits(const int64_t *rowptr, const int32_t *col, int64_t n_rows, const uint32_t *bits_in,
                            const uint32_t *bits_self, int64_t n_out_rows, uint32_t *bits_out, int32_t *list,
                            int32_t *list_n, int64_t list_cap, void *stream);

/* y[r, :] = 0 for every row flagged in `bits` (a buffer that is non-zero in a frontier's rows only goes back to all-zero
 * without a pass over the whole of it).  D a multiple of 4. */
int chaorec_zero_rows_by_bits_f32(float *y, int64_t n_rows, int32_t D, const uint32_t *bits, void *stream);

/* list[0 .. *list_n) = the rows flagged in `bits` (arbitrary order; *list_n zero on entry, list_cap entries): the work list
 * of a chaorec_spmm_csr_rowlist_f32 launch over a frontier that exists as a bitmap only. */
int chaorec_rows_list_from_bits(const uint32_t *bits, int64_t n_rows, int32_t *list, int32_t *list_n, int64_t list_cap,
                                void *stream);

/* out[r] = ((w t0[r] + w t1[r]) + ..) for the rows flagged in `bits` (chaorec_rows_mean_f32's association; n_terms <= 8): the
 * layer mean of Model/LightGCN.py:85-95 for a light step's item rows of a user shard. */
int chaorec_rows_mean_by_bits_f32(const float *const *terms, int32_t n_terms, float w, float *out, int64_t n_rows, int32_t D,
                                  const uint32_t *bits, void *stream);

/* dst[w] = src[0][w] | .. | src[n_src - 1][w] (src: n_src bitmaps of n_words words, back to back): the union of the ranks'
 * row bitmaps after an all-gather -- RCCL has no bitwise-or reduction.  dst may be one of the sources. */
int chaorec_or_words_u32(uint32_t *dst, const uint32_t *src, int32_t n_src, int64_t n_words, void *stream);

/* chaorec_spmm_csr_rowsparse_f32's arithmetic for the rows of a device-side LIST only (y[r] for r in list[0 .. *list_n); other
 * rows of y are not touched): the first backward propagate of a BPR step, whose output is non-zero in the 1-hop image of the
 * 3 B batch rows -- 1-2 % of the graph at BASELINE configs[4], where even a launch that only LOOKS at every row's descriptor
 * costs a third of the dense one.  One lane group per listed row, entries in CSR order, unflagged sources skipped (their
 * term is +0): bit-identical rows.  64 <= D <= 256. */
/* mean_out (optional; y may then be NULL): the layer mean of Model/LightGCN.py:85-95 for the listed rows,
 * mean_out[r] = ((w t0[r] + w t1[r]) + ..) + w y[r] over n_mean_terms <= 4 earlier-layer tables (chaorec_spmm_csr_mean_f32's
 * association): the LAST forward propagate of a training step, which needs the propagated table in its batch's rows only. */
int chaorec_spmm_csr_rowlist_f32(const int64_t *rowptr, const int32_t *col, const float *val, const float *x, float *y,
                                 int64_t n_rows, int32_t D, float alpha, const float *z, float beta, const uint32_t *src_bits,
                                 const uint32_t *z_bits, const int32_t *list, const int32_t *list_n, int64_t list_cap,
                                 float *mean_out, const float *const *mean_terms, int32_t n_mean_terms, float mean_w,
                                 int32_t *long_list, int32_t *long_cnt, int64_t long_cap, int32_t long_threshold,
                                 void *stream);
/* long_list / long_cnt (optional): listed rows with more than long_threshold entries are deferred to long_list (long_cap
 * entries; long_cnt: int32[4] since ABI 16, zero on entry and zero again afterwards) and computed by a second launch with one
 * WORKGROUP per row instead of being the tail of a 16..64-lane group (a popular item's row has 1e4-1e5 entries): without
 * src_bits (a forward propagate of a light step: every entry gathered) the gathers are shared by 256 threads and the products
 * summed in entry order through an LDS tile -- and a row above 8192 entries (CHAOREC_ROWLIST_STRIPE_T) by D / 32 workgroups,
 * one per 128-byte column stripe (a list of more than 65 536 rows: above four times that): the same chain of adds per output
 * element, D / 32 times the bytes in flight; with src_bits (the backward's first propagate) the workgroup scans 1024 entries per
 * round and queues the few flagged ones in entry order for one lane group to gather and add -- the sequential CSR-order sum
 * either way. */

/* destination rows handled by one wave64 for feature width D (host helper, launches nothing) */
int chaorec_spmm_rows_per_wave(int32_t D);
/* rows with more entries than this are walked cooperatively by their workgroup; a schedule built outside the library
 * (chaorec_amd/graph.py: schedule_tensors, the same words from tensor operations on the device) groups rows by it */
int chaorec_spmm_long_threshold(void);

/* Host-side schedule builder for chaorec_spmm_csr_f32 (HOST pointers, no GPU work): int32 elements needed,
 * and the build itself.  Graph-static: build once per (graph, D), keep the copy in HBM next to the CSR. */
int64_t chaorec_spmm_schedule_len(int64_t n_rows, int32_t D);
int chaorec_spmm_build_schedule(const int64_t *rowptr, const int32_t *col, const float *val,
                                int64_t n_rows, int32_t D, int32_t *out, int64_t out_len);

/* ---------------------------------------------------------------------------------------
 * P4/P5/P9/P13: fused BPR step on a batch of (user, pos, neg) triples.
 *
 * Replaces: three row gathers + mul + sum + log(sigmoid) + mean (+ L2 means)
 *           Model/LightGCN.py:97-121 (variant 0), Model/FREEDOM.py:185-192 (variant 1),
 *           Model/MMGCN.py:188-202 (variant 2) and their autograd backward (index_add).
 *
 * users[b] indexes rows of tab_u, pos[b]/neg[b] index rows of tab_i (LOCAL row ids).
 *   d_b  = sum_k u*p - sum_k u*n
 *   variant 0: t_b = log(sigmoid(d_b) + 1e-5)     variant 1: t_b = logsigmoid(d_b)
 *   variant 2: t_b = log(sigmoid(d_b))
 *   bpr  = -(1/B) sum_b t_b
 *   reg  = reg_weight * (mean(u^2) + mean(p^2) + mean(n^2)),  means over B*D  (0 if reg_weight==0)
 *   out_loss[0] = bpr + reg, out_loss[1] = bpr, out_loss[2] = reg;  out_total (optional, may be NULL) also
 *   receives bpr + reg as a stand-alone scalar (a separate allocation keeps the host-side autograd simple)
 *   coef[b] = d(bpr)/d(d_b)   (kept for the backward)
 * workspace: 4*B floats.  Reductions run in a fixed order: results are run-to-run identical.
 *
 * bwd: g_u[users[b]] += go*(coef_b*(p-n) + (2*reg_weight/(B*D))*u), g_i[pos[b]] += go*(coef_b*u + ..*p),
 *      g_i[neg[b]] += go*(-coef_b*u + ..*n);  go = *grad_out (device scalar) or 1 if NULL.
 *      Accumulated with fp32 atomics (duplicates inside a batch); g_u/g_i must be zeroed or
 *      hold the gradient being accumulated.
 * ------------------------------------------------------------------------------------- */
#define CHAOREC_BPR_LOG_SIGMOID_EPS 0
#define CHAOREC_BPR_LOGSIGMOID 1
#define CHAOREC_BPR_LOG_SIGMOID 2

int chaorec_bpr_fwd_f32(const float *tab_u, const float *tab_i,
                        const int64_t *users, const int64_t *pos, const int64_t *neg,
                        int32_t B, int32_t D, int32_t variant, float reg_weight,
                        float *out_loss, float *out_total, float *coef, float *workspace, void *stream);

int chaorec_bpr_bwd_f32(const float *tab_u, const float *tab_i,
                        const int64_t *users, const int64_t *pos, const int64_t *neg,
                        int32_t B, int32_t D, const float *coef, float reg_weight,
                        const float *grad_out, float *g_u, float *g_i, void *stream);
/* The same row sums WITHOUT atomics, reproducible run to run: every destination row has one owner wave that adds the row's
 * contributions in ascending (role, sample) order into ONE running float sum -- first the batch's contributions to
 * `emb[users]`, then to `emb[pos]`, then to `emb[neg]` (Model/LightGCN.py:113-121, Model/MMGCN.py:193-197), each in batch order: a
 * DEFINED order (oracle_bpr_bwd_ordered_f32 restates it), not torch's -- autograd sums the three index_select gradients as three
 * separately accumulated tensors, another association of the same addends.  fp32 atomic adds are applied in an order that
 * moves with the load on the chip; three addends in one element then differ in the last bit between runs, and Adam's first
 * steps turn that into a visible parameter difference.  Same arguments; g_u / g_i may alias (one joined table).  Falls back
 * to the atomic launch beyond 16384 slots (3 B) or D > 256.  chaorec_bpr_fwd_bwd_f32 / _at_f32 (the fused LightGCN steps) add
 * their gradient rows through this launch as well when the environment says CHAOREC_BPR_ORDERED=2 (read per call). */
int chaorec_bpr_bwd_ordered_f32(const float *tab_u, const float *tab_i,
                                const int64_t *users, const int64_t *pos, const int64_t *neg,
                                int32_t B, int32_t D, const float *coef, float reg_weight,
                                const float *grad_out, float *g_u, float *g_i, void *stream);

/* ---------------------------------------------------------------------------------------
 * S: uniform negative sampler with history rejection.
 *
 * Replaces: TrainingDataset.__getitem__'s `random.sample(all_set,1)` rejection loop
 *           (dataload.py:74-79).  Parity is distributional (uniform over items the user has
 *           not interacted with), not bit-wise: the reference uses Python's Mersenne Twister.
 * hist_* is the user -> interacted LOCAL item ids CSR, ids ascending inside a row.
 * Draws are a pure function of (seed, step, b, attempt): reproducible and order-free.
 * step_dev (optional device int64 scalar) is added to `step`, so a captured hipGraph can advance the stream
 * of draws from a device-resident batch counter.
 * out_neg[b] = local item id + id_offset (the reference hands out GLOBAL ids = item + num_user).
 * The reference draws a SECOND item per sample by the same rule (dataload.py:81-84: `int_items`, handed out only to MCLN,
 * :103-104): that is this function again under  seed ^ CHAOREC_SECOND_DRAW_SALT  -- an independent stream of the same
 * counter generator, never in the user's history either.
 * ------------------------------------------------------------------------------------- */
#define CHAOREC_SECOND_DRAW_SALT 0x9E3779B97F4A7C15ull
int chaorec_sample_negatives(const int64_t *hist_rowptr, const int32_t *hist_col,
                             const int64_t *users, int32_t B, int32_t num_item,
                             uint64_t seed, uint64_t step, const int64_t *step_dev, int64_t id_offset,
                             int64_t *out_neg, void *stream);

/* One launch per training batch for a streaming trainer: B edges picked uniformly from `edges` ([n_edges, 2]
 * int64, GLOBAL item ids, device), their (user, positive) gathered and one negative drawn for each by the rule
 * above (same draw stream as chaorec_sample_negatives).  Replaces DataLoader(shuffle=True) +
 * TrainingDataset.__getitem__ (main.py:194-195, dataload.py:74-106).  Outputs LOCAL item ids + item_offset (ABI 9:
 * item_offset = num_user gives the GLOBAL ids the reference's dataset hands to Model.loss()). */
/* The batch BEFORE the forward: chaorec_draw_batch's triples (edges != NULL; the same triples chaorec_bpr_fwd_bwd_at_f32
 * draws for the same seed / step / permutation position, LOCAL item ids) or a given batch (edges == NULL: users / pos / neg
 * are inputs), and the three table rows of every sample flagged in row_bits (bit u, bit bits_item_offset + pos,
 * bit bits_item_offset + neg); list / list_n (optional, *list_n zero on entry): every row this launch flags first is
 * appended.  For a training step whose forward propagates are restricted to the rows its loss reads
 * (train_and_evaluate.py:43-48 never looks at the other rows of Model/LightGCN.py:95's mean). */
int chaorec_batch_rows(const int64_t *edges, int64_t n_edges, const int64_t *hist_rowptr, const int32_t *hist_col,
                       int32_t B, int64_t num_user, int32_t num_item, uint64_t seed, uint64_t step,
                       const int64_t *step_dev, const int64_t *perm, const int64_t *perm_pos, int64_t pos_offset,
                       int64_t *users, int64_t *pos, int64_t *neg, uint32_t *row_bits, int64_t bits_item_offset,
                       int32_t *list, int32_t *list_n, int64_t list_cap, void *stream);

int chaorec_draw_batch(const int64_t *edges, int64_t n_edges, const int64_t *hist_rowptr,
                       const int32_t *hist_col, int32_t B, int64_t num_user, int32_t num_item,
                       uint64_t seed, uint64_t step, const int64_t *step_dev, int64_t *out_users,
                       int64_t *out_pos, int64_t *out_neg, int64_t item_offset, void *stream);

/* out[0:B] = pos - offset, out[B:2B] = neg - offset: Model.loss()'s id shift (Model/FREEDOM.py:195-196) and the row list of
 * the batch's 2 B items in one launch (ABI 9). */
int chaorec_shift_cat_i64(const int64_t *pos, const int64_t *neg, int64_t offset, int32_t B, int64_t *out, void *stream);

/* The fused form the training loop uses (north_star: "fused BPR negative-sample + gather + pairwise-logsigmoid +
 * L2-reg kernel using wavefront shuffles"): chaorec_draw_batch + chaorec_bpr_fwd_f32 in ONE forward launch -- lane 0 of
 * sample b's wave draws (user, positive, negative) exactly as chaorec_draw_batch does for (seed, step + *step_dev, b),
 * the wave shares the ids by shuffle, gathers the three rows and reduces.  out_users / out_pos / out_neg receive the
 * ids (LOCAL item ids; the backward launch chaorec_bpr_bwd_f32 takes them).  Everything else as chaorec_bpr_fwd_f32;
 * results are bit-identical to the two-call form.  advance (optional, may be step_dev itself): a device counter that
 * the single-block finalize launch increments by one after the draw, so a captured step needs no separate
 * counter kernel.  perm / perm_pos (optional): draw the training edges from an epoch permutation instead -- sample b
 * takes edge perm[*perm_pos + b] (DataLoader(shuffle=True): every edge once per epoch; negatives as above); with
 * `advance` set, *perm_pos moves on by B in the finalize launch. */
int chaorec_bpr_fwd_drawn_f32(const float *tab_u, const float *tab_i, const int64_t *edges, int64_t n_edges,
                              const int64_t *hist_rowptr, const int32_t *hist_col, int64_t num_user,
                              int32_t num_item, uint64_t seed, uint64_t step, const int64_t *step_dev,
                              int32_t B, int32_t D, int32_t variant, float reg_weight, int64_t *out_users,
                              int64_t *out_pos, int64_t *out_neg, float *out_loss, float *out_total,
                              float *coef, float *workspace, int64_t *advance, const int64_t *perm,
                              int64_t *perm_pos, void *stream);

/* Forward terms AND backward row updates of the BPR(+L2) loss in ONE launch, for a loss differentiated with
 * d(loss) = 1 (a plain loss.backward(), train_and_evaluate.py:46): sample b's coefficient depends on its own score
 * difference only, so the wave that reduced the triple adds its three gradient rows at once (the arithmetic of
 * chaorec_bpr_bwd_f32 with grad_out = 1).  The batch is drawn in the launch when `edges` != NULL (exactly as
 * chaorec_bpr_fwd_drawn_f32; out_* receive the ids) or given by in_users / in_pos / in_neg (LOCAL item ids).
 * g_u / g_i must be zero wherever no sample lands.  The loss itself comes from chaorec_bpr_finalize_f32 on the
 * same workspace.  adam_step / adam_bc (optional): torch.optim.Adam's step count is incremented and the new step's
 * bias corrections {1 - beta1^t, sqrt(1 - beta2^t)} are written here, by one otherwise idle thread (consumed later in
 * the step by chaorec_spmm_csr_adam_f32). */
int chaorec_bpr_fwd_bwd_f32(const float *tab_u, const float *tab_i, const int64_t *edges, int64_t n_edges,
                            const int64_t *hist_rowptr, const int32_t *hist_col, int64_t num_user,
                            int32_t num_item, uint64_t seed, uint64_t step, const int64_t *step_dev,
                            const int64_t *in_users, const int64_t *in_pos, const int64_t *in_neg,
                            int32_t B, int32_t D, int32_t variant, float reg_weight, int64_t *out_users,
                            int64_t *out_pos, int64_t *out_neg, float *coef, float *workspace,
                            const int64_t *perm, const int64_t *perm_pos, float *g_u, float *g_i,
                            int32_t *adam_step, float beta1, float beta2, float *adam_bc, void *stream);

/* The single-block, fixed-order reduction of a BPR forward's workspace ([4, B]: terms, sum u^2, sum p^2, sum n^2)
 * into out_loss[3] = {total, bpr, reg} (+ out_total[0] = total), and the step's scalar bookkeeping, all optional:
 * loss_accum[0] += total (the per-epoch loss sum of train_and_evaluate.py:48 without its per-batch host sync);
 * advance[0] += 1 and perm_pos[0] += B (batch counter / epoch-permutation position of the in-launch draw);
 * adam_step[0] += 1 and adam_bc[2] = {1 - beta1^t, sqrt(1 - beta2^t)} for the new t (torch.optim.Adam's step count
 * and bias corrections, consumed by chaorec_spmm_csr_adam_f32). */
int chaorec_bpr_finalize_f32(const float *workspace, int32_t B, int32_t D, float reg_weight, float *out_loss,
                             float *out_total, float *loss_accum, int64_t *advance, int64_t *perm_pos,
                             int32_t *adam_step, float beta1, float beta2, float *adam_bc, void *stream);

/* The two above for a replay of k captured steps that runs its loss bookkeeping ONCE (chaorec_amd/optim.py:
 * FusedLightGCNStep, steps_per_replay > 1): step j of the replay draws its batch with step = j and
 * pos_offset = j * B (read position *perm_pos + pos_offset of the epoch permutation) into its own workspace
 * workspace + j * ws_stride, and ONE chaorec_bpr_finalize_steps_f32 after the last step reduces the k workspaces in
 * order -- the same sums and the same sequence of additions into loss_accum as k single launches -- writes the LAST
 * step's out_loss / out_total and moves *advance on by k and *perm_pos by k * B.  scratch: 2 k + 1 floats, the last one an
 * int ticket that is zero on entry and zero again afterwards (one workgroup per step; the last to arrive does the
 * bookkeeping in step order). */
int chaorec_bpr_fwd_bwd_at_f32(const float *tab_u, const float *tab_i, const int64_t *edges, int64_t n_edges,
                               const int64_t *hist_rowptr, const int32_t *hist_col, int64_t num_user,
                               int32_t num_item, uint64_t seed, uint64_t step, const int64_t *step_dev,
                               const int64_t *in_users, const int64_t *in_pos, const int64_t *in_neg,
                               int32_t B, int32_t D, int32_t variant, float reg_weight, int64_t *out_users,
                               int64_t *out_pos, int64_t *out_neg, float *coef, float *workspace,
                               const int64_t *perm, const int64_t *perm_pos, int64_t pos_offset, float *g_u,
                               float *g_i, int32_t *adam_step, float beta1, float beta2, float *adam_bc,
                               uint32_t *row_bits, int64_t bits_item_offset, void *stream);
/* (row_bits, optional: bit u, bit bits_item_offset + pos, bit bits_item_offset + neg of every sample are set -- the
 *  rows of the gradient buffer the launch touched, for chaorec_spmm_csr_rowsparse_f32) */
int chaorec_bpr_finalize_steps_f32(const float *workspace, int64_t ws_stride, int32_t n_steps, int32_t B,
                                   int32_t D, float reg_weight, float *out_loss, float *out_total,
                                   float *loss_accum, int64_t *advance, int64_t *perm_pos, float *scratch, void *stream);

/* ---------------------------------------------------------------------------------------
 * R: all-items scoring + history mask + top-K, never materialising the [U, I] matrix.
 *
 * Replaces: torch.matmul(user, item.T) + per-user python mask loop + torch.topk
 *           (Model/LightGCN.py:147-155, Model/FREEDOM.py:230-238, Model/MMGCN.py:220-230)
 *           and the kNN build torch.mm + topk (Model/FREEDOM.py:114-118) with hist == NULL.
 *
 *   score[u][i] = sum_k user_emb[u][k] * item_emb[i][k]          (precision below)
 *   score[u][i] = mask_value   for every i in hist row u          (1e-6 / 1e-5 in the reference)
 *   out = the K largest per user, descending; ties -> LOWEST item index first.
 *   out_idx = item index + id_offset (int64), out_val = the (masked) score.
 *
 * The RESULT is the same for every `precision` value: the exact top-K of the fp32 scores defined by the
 * k-ordered fmaf chain below (what v_mfma_f32_32x32x2_f32 computes), bit-identical to
 * oracle/chaorec_oracle.c:oracle_score_dot().  With the K-dim cut into chunks of C floats (C = D for
 * D <= 128, C = 64 above), per chunk base b:
 *     for s in [0, C/2): acc = fmaf(u[b+s], i[b+s], acc); acc = fmaf(u[b+C/2+s], i[b+C/2+s], acc)
 * `precision` selects the route to it:
 *   0  fastest exact route.  D in {64, 128} and >= 4096 items: the [U, I] sweep runs on the bf16 MFMA pipe
 *      (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate) as a PREFILTER with a proven per-item error bound
 *      |s~ - s| <= e_uj = 1.05 * 2^-8 * ||u|| * ||i_j||: every item whose upper bound s~ + e exceeds a per-user
 *      threshold T_u becomes a candidate, every candidate is re-scored with the exact fp32 chain and the top-K is
 *      ranked on those values; the answer is certified when the K-th best exact score is > T_u (anything outside
 *      the candidates is then strictly beaten by K items).  T_u is estimated from a sample of the items; a user that
 *      cannot be certified (list overflow, too few / too many candidates) gets all its scores computed exactly.
 *      From 524 288 items and 2 U I D >= 3e13 flops per call on (environment CHAOREC_PF_CLS_MIN_ITEMS: an item count as the
 *      only condition, 0 = never) the prefilter works on a
 *      NORM-SORTED copy of the item table: items in descending order of their norm class (exponent + 3 mantissa bits), the
 *      bound taken per run of classes instead of per item -- e_u = c ||u|| N_run folded into the users' operand scale, so
 *      the bound costs no MFMA (one in nine at D = 128) --, thresholds sampled from every 32nd / 16th / 8th item of that
 *      order.  Same candidates-superset guarantee, same exact re-score, same result bit for bit; the workspace holds the
 *      permutation (chaorec_score_topk_workspace_bytes accounts for it under the same environment).
 *      Otherwise: route 2.
 *   1  one unthresholded fp32 MFMA pass (A/B runs and tests).
 *   2  fp32 MFMA sweep with a sampled per-user threshold: tau0 = 32nd best score over every s-th 32-item tile,
 *      the full pass keeps only scores above it, a certification step counts them and any user with fewer
 *      than K is re-ranked without a threshold.
 * D in {8, 16, 32, 64, 128} (users' fragment register-resident) or a multiple of 64 above 128
 * (streamed; the kNN build over modality features);  1 <= K <= 64;  n_items >= K.
 * hist_rowptr may be NULL (no mask).  hist_col ascending inside a row.
 * workspace: chaorec_score_topk_workspace_bytes() bytes, 256-byte aligned; nothing in it survives the call.  The
 * call issues kernels only (no memset / memcpy nodes), so it may be captured in a hipGraph.
 * ------------------------------------------------------------------------------------- */
size_t chaorec_score_topk_workspace_bytes(int64_t n_users, int64_t n_items, int32_t K, int32_t D);

int chaorec_score_topk_f32(const float *user_emb, const float *item_emb,
                           int64_t n_users, int64_t n_items, int32_t D,
                           const int64_t *hist_rowptr, const int32_t *hist_col,
                           float mask_value, int32_t K, int64_t id_offset,
                           int64_t *out_idx, float *out_val,
                           void *workspace, size_t workspace_bytes,
                           int32_t precision, void *stream);

/* chaorec_score_topk_f32 (route 0) with the per-user thresholds CARRIED from one call to the next: an evaluation
 * loop ranks the same users once per epoch (train_and_evaluate.py:655-659) and one epoch moves the scores little.
 *   hint_out (optional, [n_users] float, device): receives, per user, one float below the exact score of rank
 *            hint_rank (K < hint_rank <= 128; 80 is a good value for K = 50) -- the next call's threshold.
 *   hint_in  (optional): such an array from the previous call (may alias hint_out).  Pass A sweeps with T_u = hint_in
 *            (no sampling pass, about half the candidates of a sampled threshold); users it cannot certify -- scores
 *            moved too much since -- are queued on the device and retried with a sampled threshold (pass B), then, if
 *            need be, ranked exactly.  hint_in == NULL: pass B for everybody (= chaorec_score_topk_f32).
 *            A queue of at most 16 users skips pass B and is ranked exactly, per user, at once.
 *   flags    CHAOREC_SCORE_LIGHT: launch no pass B at all (three launches and their one-wave critical path less) -- for
 *            a caller that saw a short queue last time; whatever pass A leaves is ranked exactly per user, which is slow
 *            only if that expectation was wrong.
 *   counters_out (optional, device int32[4]): {users pass A queued, users ranked by the exact route, users of the wide
 *            selection, queued users taken by the exact route} of THIS call -- what a caller bases the next call's flags on.
 * A threshold never changes the result, only the work: the output is bit-identical to chaorec_score_topk_f32's
 * whatever the hints hold (NaN / inf / stale values included).  Needs the prefilter route (D in {64,128}, >= 4096
 * items, K <= 64); otherwise the hints are ignored and hint_out / counters_out are left untouched.
 *   flags    CHAOREC_SCORE_FRONT / CHAOREC_SCORE_BACK: run only the first / only the second PHASE of the call.  The front
 *            is everything up to and including the call's first sweep over all users (pack, sampling when there are no
 *            hints, sweep); the back is the rest (selection, retry passes, exact routes).  Two calls with the same
 *            arguments and the same workspace, FRONT then BACK (in stream order, or on two streams with an event between
 *            them), are one whole call.  For a caller that ranks several user ranges: range k's back phase -- gather- and
 *            VALU-bound -- beside range k + 1's front phase -- MFMA-bound -- on another stream, one workspace per range in
 *            flight.  Neither flag (or both): the whole call.  On the routes without a prefilter the BACK call does
 *            everything and the FRONT call nothing.
 * Long item ranges (>= 131072 items), where the exact per-user route streams the whole item table per user: a user whose
 * candidate lists overflowed (threshold too low) first gets a RAISED threshold -- the exact score of a rank >= K among
 * the candidates its lists kept -- and one more compact sweep + selection (pass C); only what that cannot certify
 * either is ranked by the exact routes. */
#define CHAOREC_SCORE_LIGHT 1
#define CHAOREC_SCORE_FRONT 2
#define CHAOREC_SCORE_BACK 4
int chaorec_score_topk_hinted_f32(const float *user_emb, const float *item_emb,
                                  int64_t n_users, int64_t n_items, int32_t D,
                                  const int64_t *hist_rowptr, const int32_t *hist_col,
                                  float mask_value, int32_t K, int64_t id_offset,
                                  int64_t *out_idx, float *out_val,
                                  void *workspace, size_t workspace_bytes,
                                  const float *hint_in, float *hint_out, int32_t hint_rank, int32_t flags,
                                  int32_t *counters_out, void *stream);

/* Monitoring: what the prefilter route of the LAST scoring call on this workspace did (same sizes).
 * out10 (device, 10 x uint64): [0] users handed to the exact routes, [1] candidates re-scored in total, [2] longest
 * per-lane sweep list (entries), [3] users, [4..8] users the last selection pass could not certify, by reason (list
 * overflow, fewer than K candidates, more candidates than the selection holds, K-th best not above the sweep
 * threshold, unused), [9] users that were given a raised threshold and one more pass (pass C).  All zeros if that call
 * did not take the prefilter route. */
int chaorec_score_topk_stats(const void *workspace, int64_t n_users, int64_t n_items, int32_t K, int32_t D,
                             uint64_t *out10, void *stream);

/* ---------------------------------------------------------------------------------------
 * M: ranking metrics of every evaluation row in one launch.
 *
 * Replaces: utils.gene_metrics' python loop over users x K x {precision, recall, ndcg, hit_rate, map}
 *           (utils.py:112-139) with the per-user formulas of metrics.py:13-57 and their corner cases: set()
 *           semantics for precision / recall / hit_rate, `item in test_list` per position for ndcg / map,
 *           len(test_list) (duplicates included) as denominator, 0 for an empty list, natural-log discounts.
 *
 *   rank_idx   [n_users, rank_stride] int64 global item ids, best first (chaorec_score_topk_f32's out_idx)
 *   row_user   [n_rows] user of each evaluation row; pos_rowptr/pos_items: the rows' positives as a CSR
 *              (val.npy / test.npy rows [user, pos...]), global item ids, duplicates allowed
 *   k_list     HOST array of n_k <= 8 cut-offs, each <= min(64, rank_stride)
 *   discount   HOST array, discount[p] = 1 / log(p + 2) for p < max(k) as the caller's libm gives it (the
 *              per-user terms are then bit-identical to a host evaluation with the same table)
 *   out        DEVICE [n_k][5] fp64: precision, recall, ndcg, hit_rate, map, each averaged over n_rows
 * fp64 throughout; the sums over rows are reduced in a fixed order (run-to-run identical).
 * ------------------------------------------------------------------------------------- */
size_t chaorec_rank_metrics_workspace_bytes(int64_t n_rows, int32_t n_k);
int chaorec_rank_metrics_f64(const int64_t *rank_idx, int64_t n_users, int64_t rank_stride,
                             const int64_t *row_user, const int64_t *pos_rowptr, const int64_t *pos_items,
                             int64_t n_rows, const int32_t *k_list, int32_t n_k, const double *discount,
                             double *out, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Dense fp32 GEMM on the f32 MFMA pipe (exact fp32 products, fp32 accumulate):
 *   C[M,N] = op(A) * op(B) (+ bias[N]) (+ C if accumulate)
 * transA/transB: 0 = as stored [M,K]/[K,N], 1 = stored transposed [K,M]/[N,K].
 *
 * Replaces: nn.Linear forward/backward on modality features and MMGCN's per-layer Linears
 *           (Model/FREEDOM.py:59-60,209,212; Model/MMGCN.py:40,97,102-131; BasicGCN.py:40).
 *   forward  y = x W^T + b      : transA=0, transB=1 (W stored [N,K])
 *   grad x   = gy W             : transA=0, transB=0
 *   grad W   = gy^T x           : transA=1, transB=0
 * act: 0 none, 1 leaky_relu(0.01) applied to the result (F.leaky_relu default slope), 2 leaky_relu(0.2)
 *      (nn.LeakyReLU(0.2), Model/NGCF.py:32).
 * Few output tiles with a long reduction (weight gradients: K = number of graph nodes) are split along K into
 * slabs in `workspace` (size from chaorec_gemm_workspace_bytes) and summed in a fixed order by a second launch:
 * deterministic, equal to the unsplit chain to rounding. */
size_t chaorec_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int chaorec_gemm_f32(const float *A, const float *B, float *C, const float *bias,
                     int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                     int32_t transA, int32_t transB, int32_t accumulate, int32_t act,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * y = x W^T (+ bias, + leaky-relu) on the bf16 MFMA pipe at fp32-grade accuracy: the FORWARD of the modality
 * projections image_trs / text_trs over the [I, 4096] / [I, 384] feature tables (Model/FREEDOM.py:59-60,209,212) and
 * of MMGCN's Linears (Model/MMGCN.py:40,97,102-131; BasicGCN.py:40).
 *   C[M,N] = A[M,K] . B[N,K]^T, both operands k-contiguous (torch's nn.Linear layout), fp32 in / fp32 out.
 * Every operand is split exactly into three bf16 planes (8+8+8 significand bits) and six of the nine plane products
 * are accumulated in fp32 by v_mfma_f32_32x32x16_bf16: |C - exact| <= ~4e-7 * sum_k |a||b| (an fp32 GEMM's accuracy
 * class; NOT bit-identical to chaorec_gemm_f32 / oracle_gemm_f32, whose k-ascending fmaf chain stays available), at
 * 2.7x the f32 MFMA rate -- the skinny N = 64 projections become HBM-bound (A is read once).
 * act: 0 none, 1 leaky_relu(0.01), 2 leaky_relu(0.2).  workspace: chaorec_gemm_nt_bf16x3_workspace_bytes(M,N,K)
 * (K-slabs of outputs with few tiles, summed in a fixed order).
 * ------------------------------------------------------------------------------------- */
size_t chaorec_gemm_nt_bf16x3_workspace_bytes(int64_t M, int64_t N, int64_t K);
int chaorec_gemm_nt_bf16x3(const float *A, const float *B, float *C, const float *bias, int64_t M, int64_t N,
                           int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t act, void *workspace,
                           size_t workspace_bytes, void *stream);

/* The weight gradient of those Linears on the same pipe:  C[M, N] = A[K, M]^T B[K, N]  (A = the gradient of the layer's
 * output, B = its input, K = the rows both run over -- every graph node for Model/MMGCN.py:97-131's layers), same
 * three-plane split, same accuracy class, slabs along K summed in a fixed order (deterministic).  Replaces
 * autograd's  grad_output.t() @ input  (an fp32 GEMM).  lda >= M, ldb >= N, ldc >= N. */
size_t chaorec_gemm_tn_bf16x3_workspace_bytes(int64_t M, int64_t N, int64_t K);
int chaorec_gemm_tn_bf16x3(const float *A, const float *B, float *C, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldb, int64_t ldc, void *workspace, size_t workspace_bytes, void *stream);

/* The INPUT gradient of those Linears,  C[M, N] = A[M, K] . B[K, N]  with B = the layer's weight [out = K, in = N] as it
 * lies in memory (autograd's  grad_output @ weight, Model/MMGCN.py:97-131's layers backwards): same pipe, same split,
 * same accuracy class.  lda >= K, ldb >= N, ldc >= N.  accumulate != 0: C += A B (a node's second gradient flow added in
 * the epilogue instead of by a separate launch; a + b is commutative, so the bits are those of the separate add). */
size_t chaorec_gemm_nn_bf16x3_workspace_bytes(int64_t M, int64_t N, int64_t K);
int chaorec_gemm_nn_bf16x3(const float *A, const float *B, float *C, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldb, int64_t ldc, int32_t accumulate, void *workspace, size_t workspace_bytes,
                           void *stream);

/* The two Linears MMGCN applies to the same x -- conv.lin (BasicGCN.py:40) and linear_layer (Model/MMGCN.py:104-131) -- as ONE
 * product each way, their operands concatenated virtually (second pointers, no copies), same pipe and accuracy class (ABI 9):
 *   nt_dual  [C1 | C2] = act1/act2( A [B1; B2]^T + [bias1 | bias2] )      B1 [N1, K], B2 [N2, K]; C1 [M, N1], C2 [M, N2]
 *   nn_dual  C = [A1 | A2] [B1; B2]                                      A1 [M, K1], A2 [M, K2]; B1 [K1, N], B2 [K2, N]
 *   tn_dual  [C1; C2] = [A1 | A2]^T B                                    A1 [K, M1], A2 [K, M2]; B [K, N]; C1 [M1, N], C2 [M2, N]
 * The split sizes (N1 / K1 / M1) and the second operands' leading dimensions are multiples of 4, the second pointers 16-byte
 * aligned.  nt_dual serves shapes without split-K only (error otherwise: call the products separately). */
int chaorec_gemm_nt_bf16x3_dual(const float *A, const float *B1, const float *B2, float *C1, float *C2, const float *bias1,
                                const float *bias2, int64_t M, int64_t N1, int64_t N2, int64_t K, int64_t lda, int64_t ldb1,
                                int64_t ldb2, int64_t ldc1, int64_t ldc2, int32_t act1, int32_t act2, void *stream);
size_t chaorec_gemm_nn_bf16x3_dual_workspace_bytes(int64_t M, int64_t N, int64_t K);
int chaorec_gemm_nn_bf16x3_dual(const float *A1, const float *A2, const float *B1, const float *B2, float *C, int64_t M,
                                int64_t N, int64_t K1, int64_t K2, int64_t lda1, int64_t lda2, int64_t ldb1, int64_t ldb2,
                                int64_t ldc, void *workspace, size_t workspace_bytes, void *stream);
size_t chaorec_gemm_tn_bf16x3_dual_workspace_bytes(int64_t M, int64_t N, int64_t K);
int chaorec_gemm_tn_bf16x3_dual(const float *A1, const float *A2, const float *B, float *C1, float *C2, int64_t M1, int64_t M2,
                                int64_t N, int64_t K, int64_t lda1, int64_t lda2, int64_t ldb, int64_t ldc1, int64_t ldc2,
                                void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Fused Adam step over one flat fp32 parameter (torch.optim.Adam defaults, main.py:397):
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * t = *step_dev when step_dev != NULL (a device counter, so the launch can sit in a captured hipGraph),
 * else `step`.
 * ------------------------------------------------------------------------------------- */
int chaorec_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                          int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int32_t step, const int32_t *step_dev, void *stream);

/* The same update for up to chaorec_adam_multi_max() parameter tensors in ONE launch (host arrays of `count` device
 * pointers and element counts; the pointers travel in the kernel argument): a model with ~50 small tensors
 * (Model/MMGCN.py) otherwise pays one launch latency per tensor and step.  Same arithmetic, bit for bit. */
int32_t chaorec_adam_multi_max(void);
int chaorec_adam_multi_f32(int32_t count, float *const *param, const float *const *grad, float *const *exp_avg,
                           float *const *exp_avg_sq, const int64_t *numel, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int32_t step, const int32_t *step_dev, void *stream);

/* ---------------------------------------------------------------------------------------
 * Adam over a trainable feature table whose gradient is  G = gy W  (rank R <= 64, non-zero in the few rows of the batch).
 *
 * Replaces: the dense [n_rows, K] gradient of nn.Embedding.from_pretrained(feat, freeze=False).weight behind
 *           trs(feat.weight)[pos | neg] (Model/FREEDOM.py:59-60, 209-213: autograd materialises gy W, 186 MB at
 *           clothing size) + torch.optim.Adam streaming it (main.py:397).
 *
 * param, exp_avg, exp_avg_sq [n_rows, K] (K % 4 == 0, 16-byte aligned); gy [n_rows, R] = gradient with respect to the
 * projection's output (zero rows for items outside the batch); W [R, K] = the projection's weight (nn.Linear layout).
 *   g[n, c] = fmaf chain over r ascending of gy[n, r] * W[r, c], starting from +0; a row whose gy row is all zero takes
 *   g = +0; then chaorec_adam_step_f32's arithmetic with t = *step_dev (or `step`).
 * mode 0 (dense): every row is updated: the tables always equal what chaorec_adam_step_f32 on the dense gradient gives.
 * mode 1 (lazy):  only rows with a non-zero gy row are read or written; such a row first replays the zero-gradient
 *                 steps it sat out (last[strip, row] + 1 .. t - 1: same operations, order and bias corrections), then
 *                 takes step t.  `last`: int32 [chaorec_adam_lowrank_strips(K), n_rows], initialised to the step count at
 *                 which lazy updating starts.
 * mode 2 (flush): every row catches up to step t inclusive with zero gradients and no new step (gy, W unused): after
 *                 it the three tables are bit-identical to mode 0's.  Readers of `param` must flush first, or
 * mode 3 (catch up): the same for the rows with a non-zero row in `gy` only (gy is then a flag array [n_rows, R], W
 *                 unused): the rows of a batch, before its forward gathers them.
 * bc_table (optional, float[2 * bc_len] from chaorec_adam_bias_table): (1 - beta1^s, sqrt(1 - beta2^s)) per step s,
 * spares the replay two double-precision pow() per missed step; steps >= bc_len are computed in the launch.
 * rowlist / rowcount / rowcap (modes 1 and 3): these modes walk a list of distinct rows (int32 ids in device memory,
 * *rowcount <= rowcap of them).  rows_given != 0: the caller filled it (chaorec_unique_rows over the batch's item ids)
 * and the launch visits exactly these rows -- a listed row whose gy row is zero takes the step with g = +0 (mode 1),
 * which is what mode 0 does to it; mode 3 then needs no flag array (gy may be NULL).  rows_given == 0: the launch
 * fills the list itself with the rows whose gy (or flag) row is non-zero (rowcap >= n_rows).
 *
 * mode 0 with step_dev: the step used is *step_dev + step (ABI 9) -- a launch issued before the optimizer advanced its device
 * counter (on a side stream, beside the rest of the backward pass) passes step = 1, every other caller 0.
 *
 * chaorec_unique_rows: list[0 .. *count) = the distinct values of rows[0 .. n) (item ids of a batch, duplicates
 * allowed; ids outside [0, n_rows) are dropped), order unspecified.  claim: int32 [n_rows] scratch and stamp_dev: int32 [1], both zero-initialised once and
 * then left to the launches: each one takes the stamp *stamp_dev + 1, marks the rows it lists with it and stores it
 * back, so nothing is cleared between launches (also not between replays of a captured hipGraph).  One workgroup:
 * meant for batches of a few thousand ids.
 * ------------------------------------------------------------------------------------- */
int32_t chaorec_adam_lowrank_strips(int32_t K);
int chaorec_adam_bias_table(float *table, int32_t n_steps, float beta1, float beta2, void *stream);
int chaorec_unique_rows(const int64_t *rows, int64_t n, int64_t n_rows, int32_t *claim, int32_t *stamp_dev,
                        int32_t *list, int32_t *count, void *stream);
int chaorec_adam_lowrank_f32(float *param, const float *gy, const float *W, float *exp_avg, float *exp_avg_sq,
                             int64_t n_rows, int32_t K, int32_t R, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int32_t step, const int32_t *step_dev, int32_t mode, int32_t *last,
                             const float *bc_table, int32_t bc_len, int32_t *rowlist, int32_t *rowcount,
                             int32_t rowcap, int32_t rows_given, void *stream);

/* ---------------------------------------------------------------------------------------
 * Per-step edge dropout + symmetric renormalisation of a structure-static CSR (NGCF).
 *
 * Replaces: dropout_adj + add_self_loops + degree + deg^-1/2[row]*deg^-1/2[col] inside every NGCFConv.forward
 *           (Model/NGCF.py:38-58), which rebuilds the edge list of the graph on every call.
 *
 * The graph is the destination-major CSR of the bidirectional train edges WITH one self loop per node (the
 * structure of chaorec_spmm_csr_f32's operand, built once).  entry_row[k] = destination of entry k, col[k] = its
 * source, transpose_entry[k] = the entry holding the reversed edge (a bijection; a self loop maps to itself).
 *   keep_k = 1 for self loops (they are appended after the dropout), else u_k >= p with
 *            u_k = 2^-24 * (mix64(seed ^ mix64(step' ^ mix64((salt << 48) ^ k))) >> 40), step' = step + *step_dev
 *            (or keep_in[k] != 0 when keep_in is given: an externally drawn mask, used by the parity tests)
 *   deg[n]  = number of kept entries whose SOURCE is n          (degree(row) on the kept list, int32 in deg_ws)
 *   val[k]   = keep_k                  ? (1/sqrt(deg[col[k]])) * (1/sqrt(deg[entry_row[k]])) : 0
 *   val_t[k] = keep_transpose_entry[k] ? the same product : 0      (the values of A^T in the same structure)
 * p == 0 keeps everything.  deg_ws: n_nodes int32.  salt < 65536 separates the conv layers of one step.
 * Not bit-comparable with the reference's mask (torch's generator); the keep law (independent, rate 1-p) is.
 * ------------------------------------------------------------------------------------- */
int chaorec_edge_dropout_norm(const int32_t *entry_row, const int32_t *col, const int32_t *transpose_entry,
                              int64_t nnz, int64_t n_nodes, float p, uint64_t seed, uint64_t step,
                              const int64_t *step_dev, uint32_t salt, const uint8_t *keep_in,
                              int32_t *deg_ws, float *val, float *val_t, void *stream);

/* ---------------------------------------------------------------------------------------
 * Weighted sampling WITHOUT replacement as a keep mask (FREEDOM's degree-sensitive pruning).
 *
 * Replaces: torch.multinomial(edge_values, k) (Model/FREEDOM.py:151), which is limited to 2^24 categories; the
 *           reference only uses the drawn SET (the pruned graph is coalesced right after, :152-162).
 *
 * Exponential race: key_e = |log(u_e)| / w_e with u_e uniform in (0,1) from the counter generator
 * (mix64(seed ^ mix64(step' ^ mix64((0x5A3B << 48) ^ e)))); the k smallest keys are kept -- the same law as k
 * sequential draws without replacement with probabilities proportional to w.  The k-th smallest 64-bit key
 * ((fp32 key bits << 32) | 32 fresh hash bits) is found by an exact radix select (6 passes, integer histograms),
 * so the mask is a deterministic function of (weights, seed, step').  keep[e] = key_e <= that key: exactly k
 * ones unless two entries share all 64 key bits.  w_e <= 0 is never kept (while k <= #positive weights).
 * keys_out (optional, may be NULL): the n 64-bit keys, for the tests.
 * ------------------------------------------------------------------------------------- */
size_t chaorec_weighted_sample_workspace_bytes(void);
int chaorec_weighted_sample_keep(const float *weights, int64_t n, int64_t k, uint64_t seed, uint64_t step,
                                 const int64_t *step_dev, void *workspace, size_t workspace_bytes,
                                 uint8_t *keep, uint64_t *keys_out, void *stream);

/* The keys of that race alone, entry j numbered ids[j] (NULL: j): what a RANK of a user-sharded job computes for its
 * own share of the edge list (chaorec_amd/dist.py:ShardedFREEDOM.pre_epoch_processing) -- with the edges' numbers in
 * the whole list the keys, hence the kept set { key <= the k-th smallest key of ALL ranks }, do not depend on the
 * sharding; the k-th smallest key over the ranks is found by a radix select over all-reduced histograms. */
int chaorec_weighted_sample_keys(const float *weights, const int64_t *ids, int64_t n, uint64_t seed, uint64_t step,
                                 const int64_t *step_dev, uint64_t *keys_out, void *stream);

/* ---------------------------------------------------------------------------------------
 * Row-wise cosine re-weighting of a propagated layer (LayerGCN).
 *
 * Replaces: _weights = F.cosine_similarity(all_embeddings, ego_embeddings, dim=-1)
 *           all_embeddings = torch.einsum('a,ab->ab', _weights, all_embeddings)       (Model/LayerGCN.py:125-127)
 *           and their autograd backward -- ~12 + ~25 elementwise / reduction launches per layer in the reference.
 *
 * fwd:  w_r = <y_r, e_r> / (max(|y_r|, 1e-8) * max(|e_r|, 1e-8));  out_r = w_r * y_r;  w_out[r] = w_r (optional).
 * bwd:  with a = max(|y|, eps), b = max(|e|, eps), s = <grad_out, y>:
 *         grad_y = w grad_out + s (e/(a b) - [|y| > eps] w y / a^2),  grad_e = s (y/(a b) - [|e| > eps] w e / b^2)
 * y, e, out, grads: fp32 [n_rows, D] row-major, D a multiple of 4 in [4, 1024].  One launch each, one pass over HBM.
 * ------------------------------------------------------------------------------------- */
int chaorec_row_cosine_scale_fwd_f32(const float *y, const float *e, float *out, float *w_out,
                                     int64_t n_rows, int32_t D, void *stream);
int chaorec_row_cosine_scale_bwd_f32(const float *grad_out, const float *y, const float *e, float *grad_y,
                                     float *grad_e, int64_t n_rows, int32_t D, void *stream);

/* NGCF's elementwise backward, one launch each (n a multiple of 4; 16-B aligned pointers).
 * Replaces: torch.where(y > 0, g, 0.2 g) (three launches: the backward of nn.LeakyReLU(0.2), Model/NGCF.py:32,80-84) and
 *           the backward of x_j * x_i (Model/NGCF.py:78: two multiplies and the add into the aggregate's other gradient).
 *   chaorec_leaky_bwd_f32:     grad_in = y > 0 ? grad_out : slope * grad_out
 *   chaorec_mul_pair_bwd_f32:  grad_s += grad_t * x (separately rounded product and sum);  grad_x = grad_t * s */
int chaorec_leaky_bwd_f32(const float *y, const float *grad_out, float slope, float *grad_in, int64_t n, void *stream);
int chaorec_mul_pair_bwd_f32(const float *grad_t, const float *s, const float *x, float *grad_s, float *grad_x,
                             int64_t n, void *stream);

/* out = w * terms[0] + w * terms[1] + ... + w * terms[n_terms-1] over n floats, accumulated in that order (products and
 * sums rounded separately): LightGCN's layer mean (Model/LightGCN.py:86-93) for rows whose propagated values arrive
 * AFTER the propagate launches -- the replicated item rows of a user-row shard, summed over the ranks by the per-layer
 * exchange (chaorec_amd/dist.py).  Same association as chaorec_spmm_csr_mean_f32's epilogue.  terms: HOST array of
 * n_terms (<= 8) device pointers; n a multiple of 4. */
int chaorec_rows_mean_f32(const float *const *terms, int32_t n_terms, float w, float *out, int64_t n, void *stream);

/* ---------------------------------------------------------------------------------------
 * Deterministic two-pass reductions (fixed order, no atomics, no semaphores, no memset nodes).
 *
 * Replaces: the bias gradient of every nn.Linear (grad_out.sum(0), autograd of Model/MMGCN.py:97-131,
 *           Model/FREEDOM.py:209,212, Model/MGCN.py gates) and the scalar means of the regularisers
 *           (Model/MMGCN.py:198-199, Model/LayerGCN.py:147-155, Model/MGCN.py:301) INSIDE captured training steps:
 *           torch's multi-block reductions zero a semaphore buffer with a memset, and a memset node of a captured
 *           hipGraph does not replay on this stack (stale / garbage sums from the second replay on).
 *
 * colsum: out[c] = sum_r x[r*ldx + c]  (x fp32 [M, N] row-major).   sum: out[0] = scale * sum_i x[i].
 * workspace: chaorec_reduce_workspace_bytes(M, N) bytes (use M = n, N = 1 for the scalar sum).
 * ------------------------------------------------------------------------------------- */
size_t chaorec_reduce_workspace_bytes(int64_t M, int64_t N);
int chaorec_colsum_f32(const float *x, int64_t M, int64_t N, int64_t ldx, float *out, void *workspace,
                       size_t workspace_bytes, void *stream);
int chaorec_sum_f32(const float *x, int64_t n, float scale, float *out, void *workspace, size_t workspace_bytes,
                    void *stream);

/* ---------------------------------------------------------------------------------------
 * Peer-to-peer exchange of a sharded layer's item partial (the "direct" reduce-scatter + all-gather in which
 * every row block crosses one xGMI link once), as two plain kernels over IPC-mapped peer buffers ("mailboxes"):
 *   pull_sum:    out[0:n] = sum over r = 0..world-1, in that order, of peers[r][offset : offset + n]
 *   pull_gather: out[r * block : (r + 1) * block] = peers[r][0 : block]   for every r
 * peers: HOST array of `world` (<= 16) device pointers -- this rank's own mailbox at its own index, the others mapped from
 * the peers' processes (hipIpcOpenMemHandle): the partial mailboxes (whole buffer) for pull_sum, the result mailboxes (one
 * block) for pull_gather.  The caller orders the phases across the ranks (a barrier between mailbox write, pull_sum and
 * pull_gather; chaorec_amd/dist.py: a one-element all-reduce on the stream).  offset, n, block in floats, multiples of 4.
 * Plain launches: capturable in a hipGraph (ABI 9).
 * ------------------------------------------------------------------------------------- */
int chaorec_exchange_pull_sum_f32(const void *const *peers, int32_t world, int64_t offset, int64_t n, float *out,
                                  void *stream);
int chaorec_exchange_pull_gather_f32(const void *const *peers, int32_t world, int64_t block, float *out, void *stream);

/* The same exchange for a buffer that is non-zero in a FRONTIER's rows only (ABI 12; the gradient seed and the frontier
 * partials of dist.FusedShardedLightGCNStep's row-sparse backward).  bits: a bitmap over the buffer's n_rows rows, identical
 * on every rank, a superset of the rows that are non-zero on any rank.  Only flagged rows move:
 *   chaorec_rows_copy_by_bits_f32:        dst[r] = src[r]                      (the partial into the mailbox)
 *   chaorec_exchange_pull_sum_rows_f32:   out[r - row0] = sum over ranks, rank order, of peers[q][r]   for the flagged rows r of
 *                                         this rank's block [row0, row0 + n_block)
 *   chaorec_exchange_pull_gather_rows_f32: out[r] = peers[r / n_block][r - (r / n_block) n_block]      for every flagged row
 * The other rows of the caller's buffer are not touched (zeros on every rank, and zeros is their sum).  Row pointers in
 * units of rows of D floats (D a multiple of 4). */
int chaorec_rows_copy_by_bits_f32(float *dst, const float *src, int64_t n_rows, int32_t D, const uint32_t *bits, void *stream);
int chaorec_exchange_pull_sum_rows_f32(const void *const *peers, int32_t world, int64_t row0, int64_t n_block, int64_t n_rows,
                                       int32_t D, const uint32_t *bits, float *out, void *stream);
int chaorec_exchange_pull_gather_rows_f32(const void *const *peers, int32_t world, int64_t n_block, int64_t n_rows, int32_t D,
                                          const uint32_t *bits, float *out, void *stream);

/* The COMPACT form of a frontier exchange, for collectives that cannot skip rows (RCCL): the bitmap being the same on every
 * rank, "flagged row number k in bitmap order" names the same row everywhere.  pack: prefix[w] = flagged rows before word w
 * (prefix: int32 [n_words + 1] scratch, prefix[n_words] = their total), compact[k] = src[row k] for k < cap, the rest of
 * compact zeroed; the caller all-reduces compact ([cap, D]); unpack: dst[row k] = compact[k].  Rows beyond cap are dropped:
 * only for frontiers with a static bound (the batch items of all ranks: <= 2 B world). */
int chaorec_frontier_pack_f32(const float *src, int64_t n_rows, int32_t D, const uint32_t *bits, int32_t *prefix, float *compact,
                              int64_t cap, void *stream);
int chaorec_frontier_unpack_f32(float *dst, int64_t n_rows, int32_t D, const uint32_t *bits, const int32_t *prefix,
                                const float *compact, int64_t cap, void *stream);

/* ---------------------------------------------------------------------------------------
 * Several BPR terms over ONE user table and one batch of users (Model/FREEDOM.py:203-215:
 *   mf_loss + reg_weight * (mf_t_loss + mf_v_loss), each -mean(logsigmoid(s+ - s-)) over its own item table):
 *   losses[k] = BPR(tab_u[users], tabs[k][pos[k]], tabs[k][neg[k]]),  out_total = sum_k wvec[k] * losses[k]
 * T <= 4 terms; tabs / pos / neg / g_i: HOST arrays of T device pointers (copied into the kernel argument).
 * coef [T, B] and workspace [T, 4 B] floats are kept for the backward, which adds
 *   grad_out * wvec[k] * d losses[k]  into g_u (shared) and g_i[k] (atomic row adds; the buffers must be zero where no
 * sample lands).  wvec, grad_out: device pointers (grad_out may be NULL = 1).  Two launches forward, one backward.
 * ------------------------------------------------------------------------------------- */
int chaorec_bpr_multi_fwd_f32(const float *tab_u, const int64_t *users, int32_t T, const float *const *tabs,
                              const int64_t *const *pos, const int64_t *const *neg, int32_t B, int32_t D,
                              int32_t variant, const float *wvec, float *losses, float *out_total, float *coef,
                              float *workspace, void *stream);
/* scatter_rows / scatter_out (both NULL, or HOST arrays of T device pointers, entries NULL per term): term k's item table
 * is a block of rows gathered from a longer table (Model/FREEDOM.py:208-213 reads the batch rows of the projected feature
 * table: here only those rows are projected) -- row r of it is row scatter_rows[k][r] of that table, and the backward adds
 * its gradient ALSO into scatter_out[k][scatter_rows[k][r], :] (zero where no sample lands): the [I, D] row gradient the
 * feature table's optimizer takes, without a separate zero fill + index_add launch per table. */
int chaorec_bpr_multi_bwd_f32(const float *tab_u, const int64_t *users, int32_t T, const float *const *tabs,
                              const int64_t *const *pos, const int64_t *const *neg, int32_t B, int32_t D,
                              const float *coef, const float *wvec, const float *grad_out, float *g_u,
                              float *const *g_i, const int64_t *const *scatter_rows, float *const *scatter_out,
                              void *stream);
/* ... and its ordered, atomic-free form (see chaorec_bpr_bwd_ordered_f32): g_u, every g_i[k] and every scatter_out[k] must be
 * DISTINCT buffers (each is one group of rows with one owner wave per row). */
int chaorec_bpr_multi_bwd_ordered_f32(const float *tab_u, const int64_t *users, int32_t T, const float *const *tabs,
                                      const int64_t *const *pos, const int64_t *const *neg, int32_t B, int32_t D,
                                      const float *coef, const float *wvec, const float *grad_out, float *g_u,
                                      float *const *g_i, const int64_t *const *scatter_rows, float *const *scatter_out,
                                      void *stream);

/* ---------------------------------------------------------------------------------------
 * Edge scores over the stored entries of a CSR: out[k] = <a[entry_row[k]], b[col[k]]>, k in [0, nnz).
 * Replaces the family's per-edge gather + product + row sum -- `torch.sum(x[row] * x[col], dim=1)` over an edge list
 * (Model/DCCF.py:109-111 cosine edge weights, Model/DDRec.py:197-204 the threshold filter, Model/GRCN.py:31 / Model/MGAT.py:43
 * attention logits) and the gradient torch.sparse.mm gives a sparse operand's values (d val[k] = <gy[row_k], x[col_k]>) --
 * which writes two [nnz, D] gathers and their product to HBM, by one pass that reads the two rows of every entry.
 * a [n_rows, D], b [n_cols, D] contiguous fp32, D a multiple of 4; entry_row / col int32 [nnz] (the CSR's entry -> row map and
 * column array).  Backward (through the caller's autograd): the two SpMMs  ga = G b,  gb = G^T a  over the same structure with
 * the incoming gradient as values (chaorec_spmm_csr_f32, CHAOREC_SPMM_DYNAMIC_VALUES).
 * ------------------------------------------------------------------------------------- */
int chaorec_edge_dot_f32(const int32_t *entry_row, const int32_t *col, const float *a, const float *b, float *out,
                         int64_t nnz, int32_t D, void *stream);

/* ---------------------------------------------------------------------------------------
 * MMGCN's layer tail in one pass each way (Model/MMGCN.py:102-131, per layer:
 *   h = F.leaky_relu(conv(x));  x_hat = F.leaky_relu(linear(x)) + id_embedding;  cat((h, x_hat), dim=1) ):
 *   out[r] = [ leaky_relu(s[r]) | u[r] + id[r] ]       s [n, d1], u / id [n, d2] (id may be NULL), out [n, d1 + d2]
 * and backwards from the concatenation's gradient:
 *   grad_s = g[:, :d1] * leaky'(cat[:, :d1]),  grad_u = g[:, d1:] * leaky'(uy),  grad_id (optional) = g[:, d1:]
 * (uy = the activated Linear output u; cat = the forward's output).  d1, d2 multiples of 4, everything contiguous.
 * ------------------------------------------------------------------------------------- */
int chaorec_leaky_cat_add_f32(const float *s, const float *u, const float *id, float *out, int64_t n_rows, int32_t d1,
                              int32_t d2, float slope, void *stream);
int chaorec_leaky_split_bwd_f32(const float *grad_cat, const float *cat, const float *uy, float *grad_s, float *grad_u,
                                float *grad_id, int64_t n_rows, int32_t d1, int32_t d2, float slope, void *stream);

/* F.normalize(torch.cat((a, b), dim=0), p=2, dim=1, eps) (Model/MMGCN.py:99-100) without the concatenation: rows
 * [0, rows_a) are read from a, rows [rows_a, n_rows) from b (b may be NULL when rows_a == n_rows); y [n_rows, D],
 * norm [n_rows] = |x_r| for the backward:  grad_x = (grad_y - y <grad_y, y>) / |x|  where |x| >= eps, grad_y / eps below;
 * rows < skip_rows of grad_x are left unwritten (the caller does not need them).  D a multiple of 4. */
int chaorec_normalize_rows_fwd_f32(const float *a, const float *b, int64_t rows_a, int64_t n_rows, int32_t D, float eps,
                                   float *y, float *norm, void *stream);
int chaorec_normalize_rows_bwd_f32(const float *grad_y, const float *y, const float *norm, int64_t skip_rows,
                                   int64_t n_rows, int32_t D, float eps, float *grad_x, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CHAOREC_HIP_H */
