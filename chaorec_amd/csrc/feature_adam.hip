// Adam over a trainable FEATURE table whose gradient has rank <= 64 and is non-zero in few rows (SURVEY 8(a) P12/P13,
// FREEDOM: Model/FREEDOM.py:59-60 makes the [I, 4096] image and [I, 384] text features trainable, :209-213 read them
// only as  trs(feature.weight)[pos | neg],  so   d loss / d feature = gy W   with gy [I, R] non-zero in the <= 2 B rows of
// the batch and W = trs.weight [R, K]).
//
// The reference materialises that dense [I, K] gradient (186 MB at clothing size), and torch.optim.Adam then streams
// param, grad, exp_avg, exp_avg_sq.  Here the gradient is never written: one launch forms  g[n, :] = gy[n, :] W  per row
// on the fly (a k-ascending fmaf chain, W's column strip resident in LDS) and applies chaorec_adam_step_f32's
// arithmetic to it, bit for bit.
//
//   MODE 0 (dense)  every row is updated every step: 6 arrays x 4 B per element of HBM traffic, nothing else.
//   MODE 1 (lazy)   only the rows with a non-zero gy row are touched.  A row that sat out steps s+1 .. t-1 first
//                   replays exactly those zero-gradient updates (same operations, same order, same bias corrections),
//                   then takes step t: after a flush the table, exp_avg and exp_avg_sq are bit-identical to MODE 0's,
//                   at the traffic of the touched rows only.  `last[strip, row]` = the step the strip of the row is
//                   current for.
//   MODE 2 (flush)  every row catches up to the current step (no gradient): what a reader of the table needs first.
//   MODE 3 (catch up) the same for the rows with a non-zero row in `gy` (here a flag array): the rows of the next batch,
//                   before the forward gathers them.
// Modes 1 and 3 walk a LIST of distinct rows (adam_lowrank_rows_kernel): the caller's (chaorec_unique_rows over the batch's
// item ids), or one the launch builds from the non-zero rows of gy (rows_from_gy_kernel) -- scanning gy inside the update
// kernel was 16 strips x I short dependent loads, more than the update itself.
//
// Layout: a workgroup owns a strip of 256 columns (one float4 per lane) and a chunk of rows (MODE 0: 16 waves take groups
// of four rows round-robin; modes 1-3: a few rows per wave), all loads of a group in flight together.  MODE 0 is HBM-bound
// (24 B per element); the lazy modes are bound by Adam's arithmetic (two IEEE divisions and a square root per element and
// replayed step) and by their short dependent chains.  Also here: chaorec_adam_multi_f32 (many small tensors, one launch).
#include "common.h"

namespace chaorec {

constexpr int kStripCols = 256;
constexpr int kUnr = 4;
constexpr int kMaxRank = 64;

struct LowrankArgs {
  float *p, *m, *v;
  const float *gy, *W;
  int64_t n_rows;
  int K, R;
  int64_t rows_per_wg;
  AdamConsts ac;
  int step_host;
  const int32_t *step_dev;
  int32_t *last;            // [n_strips, n_rows]  (MODE 1, 2)
  const float2 *bc_table;   // [bc_len]: (1 - b1^s, sqrt(1 - b2^s)) for s < bc_len, entry 0 unused
  int bc_len;
  const int32_t *rowlist;   // optional (modes 1, 3): the distinct rows to visit, *rowcount of them
  const int32_t *rowcount;
};

// The distinct values of rows[0..n) in arbitrary order, and how many: one workgroup.  claim[row] remembers the stamp of
// the launch that listed the row last and every launch takes a new stamp, so nothing is cleared between launches.
__global__ __launch_bounds__(1024) void unique_rows_kernel(const int64_t *__restrict__ rows, int64_t n, int64_t n_rows,
                                                           int32_t *__restrict__ claim, int32_t *stamp_dev,
                                                           int32_t *__restrict__ list, int32_t *__restrict__ count) {
  __shared__ int n_s;
  const int stamp = stamp_dev[0] + 1;      // a fresh stamp per launch, kept in device memory (hipGraph replays)
  if (threadIdx.x == 0) n_s = 0;
  __syncthreads();
  for (int64_t j = threadIdx.x; j < n; j += blockDim.x) {
    const int64_t row = rows[j];
    if (row < 0 || row >= n_rows) continue;          // (an id outside the table is not listed: nothing is touched for it)
    if (atomicExch(&claim[row], stamp) != stamp) list[atomicAdd(&n_s, 1)] = (int)row;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    count[0] = n_s;
    stamp_dev[0] = stamp;
  }
}

__global__ __launch_bounds__(256) void adam_bias_table_kernel(float2 *table, int n, float b1, float b2) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  float a = 1.f, b = 1.f;
  if (s > 0) adam_bias_corrections(s, b1, b2, a, b);
  table[s] = make_float2(a, b);
}

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float *p) {
  const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_store4(const float4 &x, float *p) {
  v4f t = {x.x, x.y, x.z, x.w};
  __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p));
}

__device__ __forceinline__ void adam4(float4 &p, const float4 &g, float4 &m, float4 &v, const AdamConsts &ac, float bc1,
                                      float bc2s) {
  adam_update(p.x, g.x, m.x, v.x, ac, bc1, bc2s);
  adam_update(p.y, g.y, m.y, v.y, ac, bc1, bc2s);
  adam_update(p.z, g.z, m.z, v.z, ac, bc1, bc2s);
  adam_update(p.w, g.w, m.w, v.w, ac, bc1, bc2s);
}

// MODE 0: every row, every step.  Streams p, m, v once: HBM-bound.
template <int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void adam_lowrank_dense_kernel(LowrankArgs a) {
  extern __shared__ __attribute__((aligned(16))) float w_s[];   // [R][256] floats = 64 KB at R = 64: two workgroups per CU
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x;
  const int c0 = strip * kStripCols + lane * 4;
  const bool col_ok = c0 < a.K;
  // step count and bias corrections of this step: one thread, handed over through the (not yet filled) strip buffer
  if (threadIdx.x == 0) {
    // (device counter + host offset: a launch issued BEFORE the optimizer advanced the counter passes offset 1)
    const int step = a.step_dev ? a.step_dev[0] + a.step_host : a.step_host;
    adam_bias_corrections(step, a.ac.b1, a.ac.b2, w_s[0], w_s[1]);
  }
  __syncthreads();
  const float bc1 = w_s[0], bc2s = w_s[1];
  __syncthreads();
  for (int i = threadIdx.x; i < a.R * 64; i += WAVES * 64) {
    const int r = i >> 6, l = i & 63, c = strip * kStripCols + l * 4;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < a.K) w = *reinterpret_cast<const float4 *>(a.W + (int64_t)r * a.K + c);
    *reinterpret_cast<float4 *>(w_s + r * kStripCols + l * 4) = w;
  }
  __syncthreads();
  const int64_t rb = (int64_t)blockIdx.y * a.rows_per_wg;
  const int64_t re = rb + a.rows_per_wg < a.n_rows ? rb + a.rows_per_wg : a.n_rows;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t base = rb + wave * kUnr; base < re; base += WAVES * kUnr) {
    float gyl[kUnr];
    float4 pp[kUnr], mm[kUnr], vv[kUnr];
#pragma unroll
    for (int u = 0; u < kUnr; ++u) {
      const int64_t row = base + u;
      gyl[u] = 0.f;
      if (row < re && lane < a.R) gyl[u] = a.gy[row * a.R + lane];
      if (row < re && col_ok) {
        const int64_t o = row * a.K + c0;
        if (NT) {
          pp[u] = nt_load4(a.p + o);
          mm[u] = nt_load4(a.m + o);
          vv[u] = nt_load4(a.v + o);
        } else {
          pp[u] = *reinterpret_cast<const float4 *>(a.p + o);
          mm[u] = *reinterpret_cast<const float4 *>(a.m + o);
          vv[u] = *reinterpret_cast<const float4 *>(a.v + o);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kUnr; ++u) {
      const int64_t row = base + u;
      if (row >= re) continue;
      float4 g = zero4;
      if (__any((__float_as_uint(gyl[u]) << 1) != 0u)) {
        for (int r = 0; r < a.R; ++r) {
          const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gyl[u]), r));
          const float4 w = *reinterpret_cast<const float4 *>(w_s + r * kStripCols + lane * 4);
          g.x = fmaf(s, w.x, g.x);
          g.y = fmaf(s, w.y, g.y);
          g.z = fmaf(s, w.z, g.z);
          g.w = fmaf(s, w.w, g.w);
        }
      }
      if (col_ok) {
        adam4(pp[u], g, mm[u], vv[u], a.ac, bc1, bc2s);
        const int64_t o = row * a.K + c0;
        if (NT) {
          nt_store4(mm[u], a.m + o);
          nt_store4(vv[u], a.v + o);
          nt_store4(pp[u], a.p + o);
        } else {
          *reinterpret_cast<float4 *>(a.m + o) = mm[u];
          *reinterpret_cast<float4 *>(a.v + o) = vv[u];
          *reinterpret_cast<float4 *>(a.p + o) = pp[u];
        }
      }
    }
  }
}

// the bias corrections of a step the per-wave window below does not hold (rare: a row that sat out > 60 steps, or a run
// longer than the table)
__device__ __forceinline__ float2 bias_of_step(const float2 *table, int len, int s, float b1, float b2) {
  if (s < len) return table[s];
  float x, y;
  adam_bias_corrections(s, b1, b2, x, y);
  return make_float2(x, y);
}

__global__ void zero_count_kernel(int32_t *count) {
  if (threadIdx.x == 0) count[0] = 0;
}

// the rows with a non-zero gy row, as a list (when the caller has none): one wave per row
__global__ __launch_bounds__(256) void rows_from_gy_kernel(const float *__restrict__ gy, int64_t n_rows, int R,
                                                           int32_t *__restrict__ list, int32_t *__restrict__ count) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float x = lane < R ? gy[row * R + lane] : 0.f;
  if (__any((__float_as_uint(x) << 1) != 0u) && lane == 0) list[atomicAdd(count, 1)] = (int)row;
}

// MODES 1 (lazy step), 2 (flush), 3 (catch up): the rows of a list (mode 2: every row), RPW per wave with all their loads
// in flight together; the replay loop runs from registers (the bias corrections of the last 64 steps sit one per lane).
// MODE 1 runs 16 waves per workgroup: its workgroups each stage a 64 KB strip of W, so few large ones beat many small.
template <int MODE, int RPW, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void adam_lowrank_rows_kernel(LowrankArgs a) {
  extern __shared__ __attribute__((aligned(16))) float w_s[];   // MODE 1: [R][256] floats
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int strip = blockIdx.x;
  const int c0 = strip * kStripCols + lane * 4;
  const bool col_ok = c0 < a.K;
  const int c0c = col_ok ? c0 : a.K - 4;                 // (lanes past the table load a valid address and store nothing)
  const int step = a.step_dev ? a.step_dev[0] : a.step_host;
  const int64_t n_visit = a.rowlist ? (int64_t)a.rowcount[0] : a.n_rows;
  if ((int64_t)blockIdx.y * (WAVES * RPW) >= n_visit) return;
  if (MODE == 1) {
    for (int i = threadIdx.x; i < a.R * 64; i += WAVES * 64) {
      const int r = i >> 6, l = i & 63, c = strip * kStripCols + l * 4;
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < a.K) w = *reinterpret_cast<const float4 *>(a.W + (int64_t)r * a.K + c);
      *reinterpret_cast<float4 *>(w_s + r * kStripCols + l * 4) = w;
    }
    __syncthreads();
  }
  const int64_t j0 = ((int64_t)blockIdx.y * WAVES + wave) * RPW;
  if (j0 >= n_visit) return;
  int32_t *last = a.last + (int64_t)strip * a.n_rows;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int upto = MODE == 1 ? step - 1 : step;          // the last zero-gradient step a stale row replays
  // lane i: the corrections of step  win0 + i  (the window ends at `step`; lr / bc1 once per step, not per element)
  const int win0 = step - 63;
  const int ws = win0 + lane;
  float2 tab = make_float2(1.f, 1.f);
  if (ws >= 1) tab = bias_of_step(a.bc_table, a.bc_len, ws, a.ac.b1, a.ac.b2);
  int64_t rows[RPW];
  bool valid[RPW];
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    valid[u] = j0 + u < n_visit;
    const int64_t j = valid[u] ? j0 + u : n_visit - 1;
    rows[u] = a.rowlist ? (int64_t)a.rowlist[j] : j;
    if (rows[u] < 0 || rows[u] >= a.n_rows) {            // (a list entry outside the table: skipped, never dereferenced)
      valid[u] = false;
      rows[u] = 0;
    }
  }
  float gyl[RPW];
  int lst[RPW];
  float4 pp[RPW], mm[RPW], vv[RPW];
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int64_t row = rows[u];
    gyl[u] = 0.f;
    if (MODE == 1 && lane < a.R) gyl[u] = a.gy[row * a.R + lane];
    lst[u] = last[row];
    const int64_t o = row * a.K + c0c;
    pp[u] = *reinterpret_cast<const float4 *>(a.p + o);
    mm[u] = *reinterpret_cast<const float4 *>(a.m + o);
    vv[u] = *reinterpret_cast<const float4 *>(a.v + o);
  }
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    if (!valid[u]) continue;
    const int64_t row = rows[u];
    const int l0 = __builtin_amdgcn_readfirstlane(lst[u]);
    if (MODE != 1 && l0 >= step) continue;               // already current: nothing to replay, nothing to write
    for (int s = l0 + 1; s <= upto; ++s) {
      float b1c, b2c;
      if (s >= win0) {
        b1c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tab.x), s - win0));
        b2c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tab.y), s - win0));
      } else {
        const float2 t = bias_of_step(a.bc_table, a.bc_len, s, a.ac.b1, a.ac.b2);
        b1c = t.x;
        b2c = t.y;
      }
      adam4(pp[u], zero4, mm[u], vv[u], a.ac, b1c, b2c);
    }
    if (MODE == 1) {
      float4 g = zero4;
      if (__any((__float_as_uint(gyl[u]) << 1) != 0u)) {
        for (int r = 0; r < a.R; ++r) {
          const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gyl[u]), r));
          const float4 w = *reinterpret_cast<const float4 *>(w_s + r * kStripCols + lane * 4);
          g.x = fmaf(s, w.x, g.x);
          g.y = fmaf(s, w.y, g.y);
          g.z = fmaf(s, w.z, g.z);
          g.w = fmaf(s, w.w, g.w);
        }
      }
      const float b1c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tab.x), 63));
      const float b2c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tab.y), 63));
      adam4(pp[u], g, mm[u], vv[u], a.ac, b1c, b2c);
    }
    if (col_ok) {
      const int64_t o = row * a.K + c0;
      *reinterpret_cast<float4 *>(a.m + o) = mm[u];
      *reinterpret_cast<float4 *>(a.v + o) = vv[u];
      *reinterpret_cast<float4 *>(a.p + o) = pp[u];
    }
    if (lane == 0) last[row] = step;
  }
}

// ---- many small parameters, one launch -------------------------------------------------------------------------
// A model like MMGCN has ~50 parameter tensors of a few thousand elements: one Adam launch each is 50 x ~4.5 us of launch
// latency per step.  The tensors' pointers travel BY VALUE in the kernel argument (no device-side table to keep in sync,
// and a captured hipGraph bakes them in like any other argument); block b finds its tensor by the prefix of block counts.
constexpr int kMultiMax = 48;
constexpr int kMultiBlockElems = 4096;     // elements per block: 256 threads x 4 x float4

struct AdamMultiArgs {
  float *p[kMultiMax];
  const float *g[kMultiMax];
  float *m[kMultiMax];
  float *v[kMultiMax];
  int32_t n[kMultiMax];
  int32_t first_block[kMultiMax + 1];
  int count;
  AdamConsts ac;
  int step_host;
  const int32_t *step_dev;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamMultiArgs a) {
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const int step = a.step_dev ? a.step_dev[0] : a.step_host;
    adam_bias_corrections(step, a.ac.b1, a.ac.b2, bc[0], bc[1]);
  }
  int t = 0;                                          // largest t with first_block[t] <= blockIdx.x
#pragma unroll 1
  for (int hi = a.count; hi - t > 1;) {
    const int mid = (t + hi) >> 1;
    if ((int)blockIdx.x >= a.first_block[mid]) t = mid; else hi = mid;
  }
  __syncthreads();
  const float bc1 = bc[0], bc2s = bc[1];
  float *__restrict__ p = a.p[t];
  const float *__restrict__ g = a.g[t];
  float *__restrict__ m = a.m[t];
  float *__restrict__ v = a.v[t];
  const int n = a.n[t];
  const int e0 = ((int)blockIdx.x - a.first_block[t]) * kMultiBlockElems;
  const int e1 = e0 + kMultiBlockElems < n ? e0 + kMultiBlockElems : n;
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  if (vec) {
    const int q1 = e0 + ((e1 - e0) & ~3);
    for (int i = e0 + 4 * (int)threadIdx.x; i < q1; i += 1024) {
      float4 pp = *reinterpret_cast<float4 *>(p + i), mm = *reinterpret_cast<float4 *>(m + i);
      float4 vv = *reinterpret_cast<float4 *>(v + i);
      const float4 gg = *reinterpret_cast<const float4 *>(g + i);
      adam4(pp, gg, mm, vv, a.ac, bc1, bc2s);
      *reinterpret_cast<float4 *>(m + i) = mm;
      *reinterpret_cast<float4 *>(v + i) = vv;
      *reinterpret_cast<float4 *>(p + i) = pp;
    }
    for (int i = q1 + (int)threadIdx.x; i < e1; i += 256) adam_update(p[i], g[i], m[i], v[i], a.ac, bc1, bc2s);
  } else {
    for (int i = e0 + (int)threadIdx.x; i < e1; i += 256) adam_update(p[i], g[i], m[i], v[i], a.ac, bc1, bc2s);
  }
}

}  // namespace chaorec

using namespace chaorec;

extern "C" int32_t chaorec_adam_multi_max(void) { return kMultiMax; }

extern "C" int chaorec_adam_multi_f32(int32_t count, float *const *param, const float *const *grad, float *const *exp_avg,
                                      float *const *exp_avg_sq, const int64_t *numel, float lr, float beta1, float beta2,
                                      float eps, float weight_decay, int32_t step, const int32_t *step_dev, void *stream) {
  if (count < 0 || count > kMultiMax) return fail(CHAOREC_E_INVALID, "adam_multi: count=%d (0..%d)", count, kMultiMax);
  if (count == 0) return CHAOREC_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !numel) return fail(CHAOREC_E_INVALID, "adam_multi: NULL argument");
  if (!step_dev && step < 1) return fail(CHAOREC_E_INVALID, "adam_multi: step=%d", step);
  AdamMultiArgs a;
  int blocks = 0, k = 0;
  for (int i = 0; i < count; ++i) {
    if (numel[i] == 0) continue;
    if (!param[i] || !grad[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0 || numel[i] > (int64_t)1 << 30)
      return fail(CHAOREC_E_INVALID, "adam_multi: tensor %d: NULL pointer or numel=%lld", i, (long long)numel[i]);
    a.p[k] = param[i]; a.g[k] = grad[i]; a.m[k] = exp_avg[i]; a.v[k] = exp_avg_sq[i];
    a.n[k] = (int32_t)numel[i];
    a.first_block[k] = blocks;
    blocks += (int)((numel[i] + kMultiBlockElems - 1) / kMultiBlockElems);
    ++k;
  }
  if (k == 0) return CHAOREC_OK;
  for (int i = k; i <= kMultiMax; ++i) a.first_block[i] = blocks;
  for (int i = k; i < kMultiMax; ++i) { a.p[i] = nullptr; a.g[i] = nullptr; a.m[i] = nullptr; a.v[i] = nullptr; a.n[i] = 0; }
  a.count = k;
  a.ac = make_adam_consts(lr, beta1, beta2, eps, weight_decay);
  a.step_host = step;
  a.step_dev = step_dev;
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("adam_multi_kernel");
}

extern "C" int chaorec_adam_bias_table(float *table, int32_t n_steps, float beta1, float beta2, void *stream) {
  if (!table || n_steps < 1) return fail(CHAOREC_E_INVALID, "adam_bias_table: table=%p n_steps=%d", (void *)table, n_steps);
  adam_bias_table_kernel<<<(unsigned)((n_steps + 255) / 256), 256, 0, (hipStream_t)stream>>>((float2 *)table, n_steps, beta1,
                                                                                           beta2);
  return check_launch("adam_bias_table");
}

extern "C" int chaorec_unique_rows(const int64_t *rows, int64_t n, int64_t n_rows, int32_t *claim, int32_t *stamp_dev,
                                   int32_t *list, int32_t *count, void *stream) {
  if ((!rows && n > 0) || !claim || !stamp_dev || !list || !count || n < 0 || n_rows < 0 || n_rows > INT32_MAX)
    return fail(CHAOREC_E_INVALID, "unique_rows: null pointer / n=%lld n_rows=%lld", (long long)n, (long long)n_rows);
  unique_rows_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(rows, n, n_rows, claim, stamp_dev, list, count);
  return check_launch("unique_rows_kernel");
}

extern "C" int32_t chaorec_adam_lowrank_strips(int32_t K) { return K > 0 ? (K + kStripCols - 1) / kStripCols : 0; }

extern "C" int chaorec_adam_lowrank_f32(float *param, const float *gy, const float *W, float *exp_avg, float *exp_avg_sq,
                                        int64_t n_rows, int32_t K, int32_t R, float lr, float beta1, float beta2,
                                        float eps, float weight_decay, int32_t step, const int32_t *step_dev,
                                        int32_t mode, int32_t *last, const float *bc_table, int32_t bc_len,
                                        int32_t *rowlist, int32_t *rowcount, int32_t rowcap, int32_t rows_given,
                                        void *stream) {
  if (n_rows == 0) return CHAOREC_OK;                      // (an empty table has no storage to point at)
  if (!param || !exp_avg || !exp_avg_sq) return fail(CHAOREC_E_INVALID, "adam_lowrank: null table pointer");
  if (mode < 0 || mode > 3) return fail(CHAOREC_E_INVALID, "adam_lowrank: mode=%d", mode);
  if (mode != 0 && !last) return fail(CHAOREC_E_INVALID, "adam_lowrank: mode %d needs `last`", mode);
  if ((mode == 1 || mode == 3) && (!rowlist || !rowcount || rowcap < 1 || (!rows_given && rowcap < n_rows)))
    return fail(CHAOREC_E_INVALID, "adam_lowrank: mode %d needs a row list buffer (%d entries; %lld when the launch "
                "fills it)", mode, rowcap, (long long)n_rows);
  if ((mode <= 1 && (!gy || !W)) || (mode == 3 && !gy && !rows_given))
    return fail(CHAOREC_E_INVALID, "adam_lowrank: gy / W missing");
  if (n_rows < 0 || K < 4 || K % 4 || R < 1 || R > kMaxRank)
    return fail(CHAOREC_E_INVALID, "adam_lowrank: n_rows=%lld K=%d (multiple of 4) R=%d (<= %d)", (long long)n_rows, K, R,
                kMaxRank);
  if (!step_dev && step < 1) return fail(CHAOREC_E_INVALID, "adam_lowrank: step=%d", step);
  if ((((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (mode <= 1 ? (uintptr_t)W : 0)) & 15) != 0)
    return fail(CHAOREC_E_INVALID, "adam_lowrank: tables must be 16-byte aligned");
  if (n_rows == 0) return CHAOREC_OK;
  hipStream_t st = (hipStream_t)stream;
  LowrankArgs a;
  a.p = param; a.m = exp_avg; a.v = exp_avg_sq; a.gy = gy; a.W = W;
  a.n_rows = n_rows; a.K = K; a.R = R;
  a.ac = make_adam_consts(lr, beta1, beta2, eps, weight_decay);
  a.step_host = step; a.step_dev = step_dev;
  a.last = last; a.bc_table = (const float2 *)bc_table; a.bc_len = bc_table ? bc_len : 0;
  a.rowlist = (mode == 1 || mode == 3) ? rowlist : nullptr; a.rowcount = rowcount;
  a.rows_per_wg = 0;
  const int strips = (K + kStripCols - 1) / kStripCols;
  if (mode == 0) {
    // streaming: 16 waves per workgroup (two workgroups per CU hold their 64 KB strips of W: 32 waves per CU keep enough
    // 16-byte loads in flight), up to ~8 workgroups per CU over the whole grid, non-temporal accesses (every byte is touched
    // once): 246 -> 224 us at [11384, 4096] against 4-wave workgroups with plain loads
    constexpr int kWavesDense = 16;
    const int rows_per_pass = kWavesDense * kUnr;
    // (~0.5 MB of table traffic per workgroup, so that the 64 KB strip of W it stages first stays a small part of it;
    // 4-wave workgroups were slower for the narrow [I, 384] text table too: 45 vs 35 us)
    int64_t wgs = (6 * n_rows * (int64_t)K * 4) >> 19;
    wgs = wgs < 256 ? 256 : (wgs > 2048 ? 2048 : wgs);
    const int64_t chunks0 = (wgs + strips - 1) / strips;
    int64_t rows_per_wg = (n_rows + chunks0 - 1) / chunks0;
    rows_per_wg = (rows_per_wg + rows_per_pass - 1) / rows_per_pass * rows_per_pass;
    const int64_t chunks = (n_rows + rows_per_wg - 1) / rows_per_wg;
    if (chunks > 65535) return fail(CHAOREC_E_INVALID, "adam_lowrank: too many row chunks");
    a.rows_per_wg = rows_per_wg;
    hipLaunchKernelGGL((adam_lowrank_dense_kernel<1, kWavesDense>), dim3((unsigned)strips, (unsigned)chunks),
                       dim3(kWavesDense * 64), (size_t)R * kStripCols * sizeof(float), st, a);
    return check_launch("adam_lowrank_dense_kernel");
  }
  int64_t n_visit = n_rows;
  if (mode == 1 || mode == 3) {
    if (!rows_given) {    // no list from the caller: the rows with a non-zero gy (or flag) row, listed here
      zero_count_kernel<<<1, 64, 0, st>>>(rowcount);      // (a kernel, not a memset node: hipGraph replays)
      rows_from_gy_kernel<<<(unsigned)((n_rows + 3) / 4), 256, 0, st>>>(gy, n_rows, R, rowlist, rowcount);
      int rc = check_launch("rows_from_gy_kernel");
      if (rc) return rc;
    }
    n_visit = rows_given ? (int64_t)rowcap : n_rows;   // (workgroups past the device-side count leave at once)
  }
  constexpr int kRpwStep = 4, kRpwReplay = 2, kWavesStep = 16;
  const int per_wg = mode == 1 ? kWavesStep * kRpwStep : 4 * kRpwReplay;
  const int64_t chunks = (n_visit + per_wg - 1) / per_wg;
  if (chunks > 65535) return fail(CHAOREC_E_INVALID, "adam_lowrank: too many row chunks (%lld)", (long long)chunks);
  const dim3 grid((unsigned)strips, (unsigned)chunks);
  if (mode == 1) {
    hipLaunchKernelGGL((adam_lowrank_rows_kernel<1, kRpwStep, kWavesStep>), grid, dim3(kWavesStep * 64),
                       (size_t)R * kStripCols * sizeof(float), st, a);
  } else if (mode == 2) {
    hipLaunchKernelGGL((adam_lowrank_rows_kernel<2, kRpwReplay, 4>), grid, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((adam_lowrank_rows_kernel<3, kRpwReplay, 4>), grid, dim3(256), 0, st, a);
  }
  return check_launch("adam_lowrank_rows_kernel");
}
