// M: ranking metrics of all users in one launch.
//
// Replaces utils.gene_metrics' python loop over users x K x 5 metrics (utils.py:112-139) with the per-user formulas of
// metrics.py:13-57, including their corner cases: set() semantics for precision / recall / hit_rate (a duplicate in
// ranked[:k] counts once), `item in test_list` per position for ndcg / map (duplicates count every time), len(test_list)
// with duplicates as the denominator, 0 for an empty test list, natural-log discounts.
//
// One thread per evaluation row ([user, pos...] of val.npy / test.npy as a CSR), fp64 like the reference's python floats.
// The discounts 1/log(p+2) are computed by the HOST (numpy's log, the reference's own) and passed in, so every per-user
// term is bit-identical to the host restatement; the sums over users are reduced in a fixed order (per-block partials,
// then one block) -> run-to-run identical, equal to the reference to summation-order rounding (1e-13).
#include "common.h"

namespace chaorec {

constexpr int kMetMaxK = 64;   // ranks looked at per user
constexpr int kMetMaxNK = 8;   // cut-offs per call

struct MetricArgs {
  const int64_t *rank_idx;     // [n_users, rank_stride] global item ids, best first
  int64_t rank_stride;
  const int64_t *row_user;     // [n_rows]
  const int64_t *pos_rowptr;   // [n_rows + 1]
  const int64_t *pos_items;    // global item ids, duplicates allowed
  int64_t n_rows;
  int n_k, kmax;
  int k_list[kMetMaxNK];
  double disc[kMetMaxK];       // 1 / log(p + 2)
  double idcg[kMetMaxK + 1];   // prefix sums of disc
  double *partial;             // [blocks][n_k * 5]
};

__global__ __launch_bounds__(256) void rank_metrics_kernel(const MetricArgs A) {
  __shared__ double red[4][kMetMaxNK * 5];
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (no per-thread array of the n_k * 5 values: indexed by a run-time cut-off number it lived in scratch memory, 336 B
  //  per lane; every cut-off's five values go from registers straight into the wave reduction instead)
  const bool valid = row < A.n_rows;
  unsigned long long hit = 0ull, first = 0ull;   // bit p: ranked[p] in test_list / first occurrence of its id
  int64_t len = 0;
  if (valid) {
    const int64_t u = A.row_user[row];
    const int64_t pb = A.pos_rowptr[row], pe = A.pos_rowptr[row + 1];
    len = pe - pb;
    const int64_t *top = A.rank_idx + u * A.rank_stride;
    for (int p = 0; p < A.kmax; ++p) {
      const int64_t it = top[p];
      bool h = false;
      for (int64_t j = pb; j < pe; ++j) h = h || (A.pos_items[j] == it);
      bool dup = false;
      for (int q = 0; q < p; ++q) dup = dup || (top[q] == it);
      if (h) hit |= 1ull << p;
      if (!dup) first |= 1ull << p;
    }
  }
  const double dlen = (double)len;
  // fixed-order block reduction: lanes by butterfly, waves in index order
  const int n = A.n_k * 5;
  for (int ki = 0; ki < A.n_k; ++ki) {
    const int k = A.k_list[ki];
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0;
    if (valid) {
      const unsigned long long km = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
      const double inter = (double)__popcll(hit & first & km);
      double dcg = 0.0, ap = 0.0;
      int cum = 0;
      for (int p = 0; p < k; ++p) {
        if ((hit >> p) & 1ull) {
          ++cum;
          dcg += A.disc[p];
          ap += (double)cum / (double)(p + 1);
        }
      }
      const int64_t m = len < k ? len : k;
      v0 = inter / (double)k;
      v1 = len > 0 ? inter / dlen : 0.0;
      v2 = len > 0 ? dcg / A.idcg[m] : 0.0;
      v3 = inter > 0.0 ? 1.0 : 0.0;
      v4 = len > 0 ? ap / dlen : 0.0;
    }
    for (int o = 32; o > 0; o >>= 1) {
      v0 += __shfl_xor(v0, o, 64);
      v1 += __shfl_xor(v1, o, 64);
      v2 += __shfl_xor(v2, o, 64);
      v3 += __shfl_xor(v3, o, 64);
      v4 += __shfl_xor(v4, o, 64);
    }
    if (lane == 0) {
      red[wave][ki * 5 + 0] = v0;
      red[wave][ki * 5 + 1] = v1;
      red[wave][ki * 5 + 2] = v2;
      red[wave][ki * 5 + 3] = v3;
      red[wave][ki * 5 + 4] = v4;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n)
    A.partial[(size_t)blockIdx.x * n + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void rank_metrics_finalize_kernel(const double *__restrict__ partial, int blocks, int n,
                                                                   int64_t n_rows, double *__restrict__ out) {
  const int i = threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int b = 0; b < blocks; ++b) s += partial[(size_t)b * n + i];
  out[i] = s / (double)n_rows;
}

}  // namespace chaorec

using namespace chaorec;

extern "C" size_t chaorec_rank_metrics_workspace_bytes(int64_t n_rows, int32_t n_k) {
  if (n_rows <= 0 || n_k <= 0) return 0;
  return (size_t)((n_rows + 255) / 256) * (size_t)n_k * 5 * sizeof(double);
}

extern "C" int chaorec_rank_metrics_f64(const int64_t *rank_idx, int64_t n_users, int64_t rank_stride,
                                        const int64_t *row_user, const int64_t *pos_rowptr, const int64_t *pos_items,
                                        int64_t n_rows, const int32_t *k_list, int32_t n_k, const double *discount,
                                        double *out, void *workspace, size_t workspace_bytes, void *stream) {
  if (!rank_idx || !row_user || !pos_rowptr || !k_list || !discount || !out)
    return fail(CHAOREC_E_INVALID, "rank_metrics: NULL argument");
  if (n_rows <= 0 || n_users <= 0) return fail(CHAOREC_E_INVALID, "rank_metrics: bad sizes");
  if (n_k < 1 || n_k > kMetMaxNK) return fail(CHAOREC_E_INVALID, "rank_metrics: n_k=%d must be in [1,%d]", n_k, kMetMaxNK);
  MetricArgs a;
  a.kmax = 0;
  for (int i = 0; i < n_k; ++i) {
    if (k_list[i] < 1 || k_list[i] > kMetMaxK || k_list[i] > rank_stride)
      return fail(CHAOREC_E_INVALID, "rank_metrics: k=%d must be in [1, min(%d, rank_stride)]", k_list[i], kMetMaxK);
    a.k_list[i] = k_list[i];
    if (k_list[i] > a.kmax) a.kmax = k_list[i];
  }
  const size_t need = chaorec_rank_metrics_workspace_bytes(n_rows, n_k);
  if (need > workspace_bytes || !workspace) return fail(CHAOREC_E_WORKSPACE, "rank_metrics: workspace %zu < %zu", workspace_bytes, need);
  a.idcg[0] = 0.0;
  for (int p = 0; p < kMetMaxK; ++p) {
    a.disc[p] = p < a.kmax ? discount[p] : 0.0;
    a.idcg[p + 1] = a.idcg[p] + a.disc[p];   // same left-to-right sum as the reference's sum(...) over range(min(len, k))
  }
  a.rank_idx = rank_idx;
  a.rank_stride = rank_stride;
  a.row_user = row_user;
  a.pos_rowptr = pos_rowptr;
  a.pos_items = pos_items;
  a.n_rows = n_rows;
  a.n_k = n_k;
  a.partial = (double *)workspace;
  const int blocks = (int)((n_rows + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rank_metrics_kernel, dim3(blocks), dim3(256), 0, st, a);
  int rc = check_launch("rank_metrics_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(rank_metrics_finalize_kernel, dim3(1), dim3(64), 0, st, (const double *)workspace, blocks, n_k * 5,
                     n_rows, out);
  return check_launch("rank_metrics_finalize_kernel");
}
