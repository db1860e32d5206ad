// CSR SpMM over the normalised user-item Laplacian:  y = alpha * (A x) [+ beta z], fp32.
//
// Replaces the reference's propagate (index_select + norm*x_j + scatter_add,
// Model/LightGCN.py:40-43, BasicGCN.py:48-53) and torch.sparse.mm (Model/FREEDOM.py:168,174).
//
// HBM-bound gather: 2 flop per 4 gathered bytes, so no MFMA here.  Layout choices:
//   * one LPR-lane group per destination row, each lane owning one float4 of the row
//     (D=64 -> 16 lanes x 16 B = one 256-B row per group, 4 rows per wave64), so every
//     gathered source row is read as whole 128-B lines with dwordx4 loads;
//   * the group loads LPR (col,val) pairs with ONE coalesced load and broadcasts them with
//     lane shuffles instead of every lane re-reading the same index;
//   * 4 source rows are in flight per group before the first add (4 x 4 KiB per wave),
//     the adds themselves stay in CSR order with separately rounded product and sum, so the
//     result is bit-identical to a sequential scatter_add_ in the reference's edge order.
#include "common.h"
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace chaorec {

__device__ __forceinline__ float4 shfl4(float4 v, int src) {
  return make_float4(__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64), __shfl(v.w, src, 64));
}

constexpr int kInline = 6;      // (col,val) pairs carried inside a row descriptor
constexpr int kDescDwords = 16;  // 64-B descriptor: row, deg|flag, e0 lo/hi, 6 cols, 6 vals

#ifndef CHAOREC_SPMM_UNR
#define CHAOREC_SPMM_UNR 8
#endif
#ifndef CHAOREC_SPMM_LONG_T
#define CHAOREC_SPMM_LONG_T (4 * CHAOREC_SPMM_UNR)
#endif
#ifndef CHAOREC_SPMM_SP_COMPACT2
#define CHAOREC_SPMM_SP_COMPACT2 1      // long rows of a gated launch walk a compacted list of their flagged entries
#endif
#ifndef CHAOREC_SPMM_MINW
#define CHAOREC_SPMM_MINW 1
#endif
// The GATED instantiations at D >= 128 (SP, the backward launches of a light step over a large graph) trade gathers in flight
// per wave for WAVES in flight: 4 instead of 8 source rows per group and step, a quarter of the long-row tile, and a register
// budget of SEVEN waves per SIMD instead of the four the compiler settles at when left alone (126 VGPRs; 72 with two spilled).
// Their time is a latency chain per row -- descriptor -> (col, val) -> bitmap -> gather -> store, 11 ms of it at BASELINE
// configs[4] whole with not one source flagged -- that only more rows in flight shorten (tools/gated_bench.py, that launch):
//     4 waves (rounds 3-6)  24.7 ms  0.46 of the HBM peak        6 waves (tile / 2)   20.4 ms  0.556
//     5 waves               21.8-22.0                            7 waves (tile / 4)   19.0 ms  0.598
//     8 waves (22 spills)   26.3                                 6 waves, 8 gathers in flight (37 spills)  24.3
// The dense launches gain 1-2 % from such settings at D = 128 and lose 20-45 % at D = 64 (sports, cache resident): they keep
// theirs.  CHAOREC_SPMM_SP_HIOCC=0: the rounds 3-6 form.
#ifndef CHAOREC_SPMM_SP_HIOCC
#define CHAOREC_SPMM_SP_HIOCC 1
#endif
#ifndef CHAOREC_SPMM_SP_UNR
#define CHAOREC_SPMM_SP_UNR 4
#endif
#ifndef CHAOREC_SPMM_SP_UH
#define CHAOREC_SPMM_SP_UH 2
#endif
#ifndef CHAOREC_SPMM_SP_MINW
#define CHAOREC_SPMM_SP_MINW 7
#endif
template <int LPR, bool SP>
constexpr bool spmm_hi_occ() { return CHAOREC_SPMM_SP_HIOCC && SP && LPR >= 32; }
// (D >= 128 otherwise: at least four waves per SIMD -- the Adam instantiation took 140 VGPRs = three waves when left alone;
//  held to 128 the configs[4] launch with the Adam epilogue goes 41.1 -> 39.0 ms)
template <int LPR, bool SP>
constexpr int spmm_min_waves() { return spmm_hi_occ<LPR, SP>() ? CHAOREC_SPMM_SP_MINW : (LPR >= 32 && CHAOREC_SPMM_MINW < 4 ? 4 : CHAOREC_SPMM_MINW); }

// Adam epilogue (ADAM instantiations): the row this group just produced is the GRADIENT of row r of a parameter
// table -- the last backward propagate of a LightGCN step, g_0 = A^T g_1 + w G -- and the optimizer update of that
// row (torch.optim.Adam, main.py:397) is applied right here: one launch and one pass over the gradient less per
// step.  `bc` holds the step's two bias corrections (device memory: the step counter lives there too).  clear_z:
// rows of z that were non-zero are zeroed after use (z = the batch gradient G, kept all-zero between steps
// instead of being zero-filled every step; only legal when this launch does not GATHER from z).
struct AdamEpi {
  float *p, *m, *v;
  const float *bc;
  AdamConsts c;
  int clear_z;
};

// Row-sparse operands (SP instantiations; the backward propagates of a BPR-trained model): the gradient a BPR batch leaves
// in G has 3 B non-zero rows out of N, the first backward propagate's output is non-zero in their neighbours only -- most
// gathers of those launches would fetch rows of exact zeros.  A bitmap per operand (bit r set = row r may be non-zero; a
// superset is fine) gates the LOADS: a source row whose bit is clear is not gathered, a z row whose bit is clear is not
// read.  The arithmetic is untouched: the skipped terms are val * (+0) = +0 added to the running sum (weights are
// positive, cleared rows hold +0), so the result is bit-identical to the dense launch (up to the sign of an exact zero).
// out_bits receives a superset of the output's non-zero rows (a row with any gathered source or a flagged z row; rows
// walked by the cooperative long-row section are always flagged).  `clear`: bitmap words this launch zeroes as a side job
// (the launch after the last reader of a bitmap cleans it for the next step; any instantiation).
struct RowSparse {
  const uint32_t *src_bits;
  const uint32_t *z_bits;
  uint32_t *out_bits;
  // row_bits: which OUTPUT rows can be non-zero at all (a superset: the 1-hop image of the source's rows, computed by
  // chaorec_expand_row_bits from the few flagged rows instead of by scanning every entry).  A row whose bit is clear -- and
  // whose z row is not flagged -- does nothing: no entry scan, no gathers; it stores zeros when write_zeros is set (the
  // next launch reads y densely) and nothing at all otherwise (the next launch gathers from flagged rows only).
  const uint32_t *row_bits;
  int write_zeros;
  uint32_t *clear[2];
  int64_t n_clear[2];
};
__device__ __forceinline__ bool row_bit(const uint32_t *bits, int64_t r) { return (bits[r >> 5] >> (r & 31)) & 1u; }

template <int LPR, int CPL, bool ADAM, bool SP = false>
__global__ __launch_bounds__(256, (spmm_min_waves<LPR, SP>())) void spmm_csr_ordered_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ x, float *__restrict__ y,
    int64_t n_rows, int D4, float alpha, const float *z, float beta,
    float *acc, const float *__restrict__ acc_init, float acc_w,
    const int32_t *__restrict__ sched, int64_t n_groups, int dyn_val, const AdamEpi ae,
    const float *__restrict__ mean_t1, const float *__restrict__ mean_t2, const RowSparse sa) {
  if (sa.clear[0] || sa.clear[1]) {        // (kernel-uniform; two short bitmaps: a few words per thread at most)
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gs = (int64_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int w = 0; w < 2; ++w)
      if (sa.clear[w])
        for (int64_t i = gt; i < sa.n_clear[w]; i += gs) sa.clear[w][i] = 0u;
  }
  constexpr int NG = kWave / LPR;   // lane groups = destination rows per wave
  constexpr int UNR = spmm_hi_occ<LPR, SP>() ? CHAOREC_SPMM_SP_UNR : CHAOREC_SPMM_UNR;  // gathered rows in flight per group (short-row phase)
  constexpr int UNR2 = (kWave / NG) < 16 ? (kWave / NG) : 16;  // per group in the long-row phase
  constexpr int LONG_T = CHAOREC_SPMM_LONG_T;  // rows above this are walked by the whole wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR;
  const int li = lane % LPR;
  const int64_t wslot = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // The index chain of a plain CSR walk, schedule -> rowptr -> (col,val) -> x, is four dependent loads and
  // the waves of this kernel spend ~3/4 of their life waiting on exactly that (SQ_WAIT_ANY), not on bandwidth.
  // So the host packs a 64-B ROW DESCRIPTOR per row, stored in schedule order: {row, deg|flag, e0, the first
  // kInline (col,val) pairs}.  One coalesced load per wave fetches its 4 descriptors; rows of <= kInline
  // entries (the median row) then need only descriptor -> x -> store.  Entry order is unchanged.
  int64_t r = -1, e0 = 0;
  int deg = 0;
  bool blk_long = true;   // without a schedule every block runs the cooperative section
  int icol[kInline];
  float ival[kInline];
#pragma unroll
  for (int j = 0; j < kInline; ++j) {
    icol[j] = 0;
    ival[j] = 0.f;
  }
  int n_inl = 0;          // entries of this row already in registers
  if (sched) {
    constexpr int LD = LPR >= 16 ? 16 : LPR;      // lanes of a group that hold descriptor dwords
    constexpr int DW = 16 / LD;                    // dwords per such lane
    const int32_t *dp = sched + ((size_t)wslot * NG + sub) * 16;
    int dws[DW];
#pragma unroll
    for (int k = 0; k < DW; ++k) dws[k] = (li < LD) ? dp[li + k * LD] : 0;
    // (every lane reading its group's whole descriptor with four 16-byte loads instead -- no shuffles -- was measured in round 6:
    //  sports step 119.5 -> 122.4 us, configs[4] gated launch 17.9 -> 18.1 ms, with the Adam epilogue 38.5 -> 39.5, the dense
    //  launch unchanged over two repeats (33.7 / 33.8 against 33.9 / 33.8): not kept)
    auto dword = [&](int j) { return __shfl(dws[j / LD], sub * LPR + (j % LD), 64); };
    r = dword(0);
    const int d1 = dword(1);
    deg = d1 & 0x7fffffff;
    blk_long = d1 < 0;
    e0 = ((int64_t)dword(3) << 32) | (int64_t)(unsigned int)dword(2);
#pragma unroll
    for (int j = 0; j < kInline; ++j) {
      icol[j] = dword(4 + j);
      ival[j] = __int_as_float(dword(4 + kInline + j));
    }
    // CHAOREC_SPMM_DYNAMIC_VALUES: the schedule was built for this STRUCTURE but the values changed since (per-step
    // edge dropout): take the first entries' values from val[] -- the address comes from the descriptor, so the
    // loads travel together with the x gathers instead of adding a hop
    if (dyn_val) {
#pragma unroll
      for (int j = 0; j < kInline; ++j)
        if (j < deg) ival[j] = val[e0 + j];
    }
    n_inl = kInline;
  } else {
    // Surplus waves of the last block stay (they take part in the block barriers below) with no rows.
    r = wslot < n_groups ? wslot * NG + sub : -1;
    if (r >= n_rows) r = -1;
    if (r >= 0) {
      e0 = rowptr[r];
      deg = (int)(rowptr[r + 1] - e0);
    }
  }
  const bool row_ok = r >= 0;
  bool active = true;           // SP: can this output row be non-zero?
  if constexpr (SP) {
    if (sa.row_bits && row_ok) {
      active = row_bit(sa.row_bits, r) || (z && sa.z_bits && row_bit(sa.z_bits, r));
      if (!active) deg = 0;     // (no entries to walk: the row's sum is the empty sum)
    }
  }
  const bool is_long = (NG > 1) && deg > LONG_T;
  const int deg1 = is_long ? 0 : deg;
  const int rest = max(deg1 - n_inl, 0);            // entries still to be fetched from the CSR arrays
  const int dmax = group_uniform_max_i32<LPR>(rest);  // wave-uniform trip counts keep every shuffle fully active

  float4 sum[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) sum[q] = make_float4(0.f, 0.f, 0.f, 0.f);

  const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(x);

  // epilogue operands (beta*z, the running layer mean) only depend on the row: fetch them now, under the
  // gathers, instead of as one more dependent load at the very end of the wave
  const float4 *z4 = reinterpret_cast<const float4 *>(z);
  const float4 *init4 = reinterpret_cast<const float4 *>(acc_init);
  float4 *acc4 = reinterpret_cast<float4 *>(acc);
  bool zflag = true;            // SP: is this row of z worth reading?
  bool any_src = !SP;           // SP: did this row gather anything (group-uniform)?
  if constexpr (SP) {
    if (z && sa.z_bits && row_ok) zflag = row_bit(sa.z_bits, r);
  }
  float4 zpre[CPL], apre[CPL];
  float4 tpre[(LPR <= 16 && !ADAM) ? CPL : 1];
  float4 ppre[ADAM ? CPL : 1], mpre[ADAM ? CPL : 1], vpre[ADAM ? CPL : 1];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int chunk = li + q * LPR;
    zpre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    apre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_ok && chunk < D4) {
      const size_t o = (size_t)r * (size_t)D4 + chunk;
      if (z && zflag) zpre[q] = z4[o];
      if (acc) apre[q] = acc_init ? init4[o] : acc4[o];
      // whole layer mean in ONE epilogue (the last forward propagate): the earlier layers' rows ride in the slots of
      // the unused z operand and (narrow rows only: register budget) one extra slot
      if (mean_t1) zpre[q] = reinterpret_cast<const float4 *>(mean_t1)[o];
      if constexpr (LPR <= 16 && !ADAM) {
        if (mean_t2) tpre[q] = reinterpret_cast<const float4 *>(mean_t2)[o];
      }
      if constexpr (ADAM) {
        ppre[q] = reinterpret_cast<const float4 *>(ae.p)[o];
        mpre[q] = reinterpret_cast<const float4 *>(ae.m)[o];
        vpre[q] = reinterpret_cast<const float4 *>(ae.v)[o];
      }
    }
  }

  // ---- phase 1: every group walks its own (short) row -----------------------------------------
  // the first (col,val) block of the remainder is requested BEFORE the inline gathers so that its round trip
  // overlaps theirs
  const int64_t e0r = e0 + n_inl;
  int c_first = 0;
  float v_first = 0.f;
  int b_first = 1;
  if (li < rest) {
    c_first = col[e0r + li];
    v_first = val[e0r + li];
  }
  // first the entries that came with the descriptor (all in flight at once) ...
  if (sched) {
    float4 xin[kInline][CPL];
    bool sflag[kInline];
#pragma unroll
    for (int j = 0; j < kInline; ++j) {
      sflag[j] = j < deg1;
      if constexpr (SP) {       // (the LPR lanes of a group ask for the same word: one request)
        if (sa.src_bits && sflag[j]) sflag[j] = row_bit(sa.src_bits, icol[j]);
        any_src = any_src || sflag[j];
      }
    }
    if constexpr (SP) {         // the remainder's first block: its bits travel with the inline gathers
      if (sa.src_bits && li < rest) b_first = row_bit(sa.src_bits, c_first) ? 1 : 0;
    }
#pragma unroll
    for (int j = 0; j < kInline; ++j) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const int chunk = li + q * LPR;
        xin[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sflag[j] && chunk < D4) xin[j][q] = x4[(size_t)icol[j] * (size_t)D4 + chunk];
      }
    }
#pragma unroll
    for (int j = 0; j < kInline; ++j) {
      if (sflag[j]) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) sum[q] = add_rn4(sum[q], mul_rn4(ival[j], xin[j][q]));
      }
    }
  } else if constexpr (SP) {
    if (sa.src_bits && li < rest) b_first = row_bit(sa.src_bits, c_first) ? 1 : 0;
  }
  // ... then the rest of the row from the CSR arrays, UNR source rows in flight
  for (int base = 0; base < dmax; base += LPR) {
    int c = c_first;
    float v = v_first;
    int bt = b_first;
    if (base > 0) {
      c = 0;
      v = 0.f;
      bt = 1;
      if (base + li < rest) {  // one coalesced (col,val) load per group, broadcast below
        c = col[e0r + base + li];
        v = val[e0r + base + li];
        if constexpr (SP) {
          if (sa.src_bits) bt = row_bit(sa.src_bits, c) ? 1 : 0;
        }
      }
    }
    const int n = min(LPR, rest - base);
    const int nmax = min(LPR, dmax - base);
    // SP with a source bitmap: the group's FLAGGED entries of this block, compacted in entry order (the skipped ones would add
    // val * (+0) = +0) -- UNR real gathers in flight per step instead of UNR slots of which a third is live (the gated
    // backward launch of a light step ran at 0.39 of the HBM peak against the dense launch's 0.8: half as many rows in
    // flight per wave, and two steps per block where one does).  gm: the group's flags, consumed lowest bit first.
    // (a 32-bit word for groups of up to 32 lanes: lowest set bit, clear it and test it are 3 instructions per slot instead of
    //  ~10 on a 64-bit pair -- the gated launch is bound by VALU issue, profiles/r06_gated_occupancy.txt)
    using gmask_t = typename std::conditional<(LPR <= 32), uint32_t, unsigned long long>::type;
    gmask_t gm = 0;
    int jmax = nmax;
    if constexpr (SP) {
      if (sa.src_bits) {
        const unsigned long long bal = __ballot(bt != 0 && li < n);
        gm = (gmask_t)((bal >> (sub * LPR)) & (LPR == 64 ? ~0ull : ((1ull << (LPR & 63)) - 1ull)));
        // (the longest group's count from the ballot itself: scalar bit counts instead of a DPP reduction over the lanes)
        constexpr unsigned long long gmask = LPR == 64 ? ~0ull : ((1ull << (LPR & 63)) - 1ull);
        jmax = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) jmax = max(jmax, (int)__popcll((bal >> (g * LPR)) & gmask));
      }
    }
    for (int j = 0; j < jmax; j += UNR) {
      int cj[UNR];
      float vj[UNR];
      bool p[UNR];
      float4 xv[UNR][CPL];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        int src = sub * LPR + ((j + u) & (LPR - 1));
        p[u] = (j + u) < n;
        if constexpr (SP) {
          if (sa.src_bits) {              // (kernel-uniform)
            p[u] = gm != 0;
            if constexpr (sizeof(gmask_t) == 4) src = sub * LPR + (p[u] ? (int)__builtin_ctz((uint32_t)gm) : 0);
            else src = sub * LPR + (p[u] ? (int)__builtin_ctzll((unsigned long long)gm) : 0);
            gm &= gm - 1;                 // (0 stays 0)
          }
          any_src = any_src || p[u];
        }
        cj[u] = __shfl(c, src, 64);
        vj[u] = __shfl(v, src, 64);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          const int chunk = li + q * LPR;
          xv[u][q] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p[u] && chunk < D4) xv[u][q] = x4[(size_t)cj[u] * (size_t)D4 + chunk];
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (p[u]) {
#pragma unroll
          for (int q = 0; q < CPL; ++q) sum[q] = add_rn4(sum[q], mul_rn4(vj[u], xv[u][q]));
        }
      }
    }
  }

  // ---- phase 2: long rows, one at a time, gathered by ALL groups ---------------------------
  // The sum stays the sequential CSR-order sum: the groups only share the LOADS.  A long row's
  // time is set by how many source rows one wave keeps in flight, so for D = 64 / 128 the walk is
  // software-pipelined: two register buffers of HALF = 8*NG source rows ping-pong (one in flight
  // while the other is reduced), (col,val) blocks of 64 entries are fetched a block ahead, and
  // the ordered reduction goes through an 8 KiB LDS tile (every lane reads entry e of its own
  // float4 column, a broadcast read) instead of NG x 4 lane shuffles per entry.
  if constexpr (NG == 2 || NG == 4) {
    // Block-cooperative, still sequential: the 4 waves of the workgroup take the row's 32-entry chunks
    // round-robin.  Each wave gathers its chunk and parks the products in its own 8 KiB LDS tile while
    // the previous chunks are being summed; the running sum is then handed from chunk to chunk through
    // LDS (`carry`, ordered by the `seq` ticket), so only the adds -- ~8 cycles per entry -- are serial,
    // not the memory latency.  The additions happen in entry order: bit-identical to one lane walking
    // the row alone.
#ifndef CHAOREC_SPMM_UH
#define CHAOREC_SPMM_UH 8
#endif
    constexpr int UH = spmm_hi_occ<LPR, SP>() ? CHAOREC_SPMM_SP_UH : CHAOREC_SPMM_UH;
    constexpr int HALF = NG * UH;       // entries per chunk (32 for D=64, 16 for D=128)
    constexpr int FPL = LPR * 4 / 64;   // features per lane in the ordered sum (1 or 2)
    constexpr int MAXL = 4 * NG;        // rows per block
    __shared__ float4 tile_all[4][HALF * LPR];
    __shared__ float carry[64 * FPL];
    __shared__ float4 long_sum[MAXL][LPR];
    __shared__ long long long_e0[MAXL];
    __shared__ int long_n[MAXL];
    __shared__ int n_long_s, seq_s;
    const int wv = threadIdx.x >> 6;
    float4 *tile = tile_all[wv];
    const float *tilef = reinterpret_cast<const float *>(tile);
    int my_slot = -1;
    int nl = 0;
    if (blk_long) {   // block-uniform (the host sets the flag on all descriptors of a block)
    if (threadIdx.x == 0) n_long_s = 0;
    __syncthreads();
    if (is_long && li == 0) {
      my_slot = atomicAdd(&n_long_s, 1);
      long_e0[my_slot] = e0;
      long_n[my_slot] = deg;
    }
    my_slot = __shfl(my_slot, sub * LPR, 64);
    __syncthreads();
    nl = n_long_s;  // block-uniform
    }
    // SP with a source bitmap: a long row is walked over its FLAGGED entries only.  One coalesced pass over (col, val) by the
    // whole workgroup, one bitmap probe per entry -- all in flight together --, the flagged ones compacted in entry order into
    // an LDS list (SB entries of the row per turn); the chunks of the ordered walk then take HALF entries of the LIST each:
    // every gather a wave has in flight is a real one and the carry chain is as long as the row's flagged part, not the row.
    // The sum is the dense walk's: the skipped terms are val * (+0) = +0.  (The gated backward launch of a light step ran at
    // 0.39 of the HBM peak against the dense launch's 0.8 with a third of every chunk's gathers live.)
    constexpr int SB = 256;
    __shared__ int lc[(SP && CHAOREC_SPMM_SP_COMPACT2) ? SB : 1];
    __shared__ float lv[(SP && CHAOREC_SPMM_SP_COMPACT2) ? SB : 1];
    __shared__ int wcnt[4];
    bool compact = false;
    if constexpr (SP && CHAOREC_SPMM_SP_COMPACT2) compact = sa.src_bits != nullptr;      // (kernel-uniform)
    for (int t = 0; t < nl; ++t) {
      const int n = long_n[t];
      const int64_t le0 = long_e0[t];
      if (threadIdx.x == 0) seq_s = 0;
      __syncthreads();
      float4 xv[UH];
      float vv[UH];
      int chunk_base = 0;                   // chunks summed so far (the ticket counts on over the turns of a compacted walk)
      for (int sb0 = 0; sb0 < n; sb0 += compact ? SB : n) {
        int n_eff = n;                      // entries this turn walks: the row, or this turn's list
        if constexpr (SP && CHAOREC_SPMM_SP_COMPACT2) {
          if (compact) {
            const int e = sb0 + (int)threadIdx.x;
            int ce = 0;
            float ve = 0.f;
            if (e < n) {
              ce = col[le0 + e];
              ve = val[le0 + e];
            }
            const bool fe = e < n && row_bit(sa.src_bits, ce);
            const unsigned long long bm = __ballot(fe);
            if (lane == 0) wcnt[wv] = (int)__popcll(bm);
            __syncthreads();
            int total = 0, mybase = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              if (w == wv) mybase = total;
              total += wcnt[w];
            }
            if (fe) {
              const int pos = mybase + (int)__popcll(bm & ((1ull << lane) - 1ull));
              lc[pos] = ce;
              lv[pos] = ve;
            }
            __syncthreads();
            n_eff = total;
          }
        }
        const int nchunks = (n_eff + HALF - 1) / HALF;
        auto gather = [&](int k) {
          int c = 0;
          float v = 0.f;
          int bt = 1;
          if (lane < HALF && k * HALF + lane < n_eff) {
            if (compact) {
              c = lc[k * HALF + lane];
              v = lv[k * HALF + lane];
            } else {
              c = col[le0 + k * HALF + lane];
              v = val[le0 + k * HALF + lane];
              if constexpr (SP) {
                if (sa.src_bits) bt = row_bit(sa.src_bits, c) ? 1 : 0;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < UH; ++u) {
            const int idx = u * NG + sub;
            const int cj = __shfl(c, idx, 64);
            vv[u] = __shfl(v, idx, 64);
            xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            bool live = k * HALF + idx < n_eff && li < D4;
            if constexpr (SP) {
              // (the shuffle FIRST, by every lane: `live && __shfl(..)` would leave the lanes that are not live out of it,
              //  and they are other lanes' sources.  A skipped row parks v * (+0) = +0 in the tile.)
              const bool flagged = __shfl(bt, idx, 64) != 0;
              live = live && flagged;
            }
            if (live) xv[u] = x4[(size_t)cj * (size_t)D4 + li];
          }
        };
        if (wv < nchunks) gather(wv);
        for (int k = wv; k < nchunks; k += 4) {
#pragma unroll
          for (int u = 0; u < UH; ++u) tile[(u * NG + sub) * LPR + li] = mul_rn4(vv[u], xv[u]);
          if (k + 4 < nchunks) gather(k + 4);  // next chunk's loads fly while we wait for the carry
          while (__hip_atomic_load(&seq_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != chunk_base + k)
            __builtin_amdgcn_s_sleep(1);
          float a[FPL];
#pragma unroll
          for (int f = 0; f < FPL; ++f) a[f] = (chunk_base + k) == 0 ? 0.f : carry[lane * FPL + f];
          const int cntv = min(HALF, n_eff - k * HALF);
#pragma unroll 8
          for (int e = 0; e < cntv; ++e) {
#pragma unroll
            for (int f = 0; f < FPL; ++f) a[f] = add_rn(a[f], tilef[e * (LPR * 4) + lane * FPL + f]);
          }
          // (a compacted walk does not know its last chunk before its last turn: it always hands on through `carry`)
          float *dst = (!compact && k == nchunks - 1) ? reinterpret_cast<float *>(long_sum[t]) : carry;
#pragma unroll
          for (int f = 0; f < FPL; ++f) dst[lane * FPL + f] = a[f];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_store(&seq_s, chunk_base + k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        chunk_base += nchunks;
        __syncthreads();                    // (compacted: every chunk of this turn is in the carry, the list may be rewritten)
      }
      if constexpr (SP && CHAOREC_SPMM_SP_COMPACT2) {
        if (compact) {
          if (wv == 0) {
            float *dst = reinterpret_cast<float *>(long_sum[t]);
#pragma unroll
            for (int f = 0; f < FPL; ++f) dst[lane * FPL + f] = chunk_base ? carry[lane * FPL + f] : 0.f;
          }
          __syncthreads();
        }
      }
    }
    if (my_slot >= 0) {
      sum[0] = long_sum[my_slot][li];
      any_src = true;             // (a long row's gathers are not tracked: always flagged)
    }
  } else if constexpr (NG > 1) {
    static_assert(!SP, "the narrow-row long walk does not look at the row bitmaps: SP is built for LPR >= 16 only");
    unsigned long long lm = __ballot(is_long && li == 0);
    while (lm) {
      const int gl = (int)(__builtin_ctzll(lm) / LPR);
      lm &= lm - 1;
      const int owner = gl * LPR;
      const int n = __shfl(deg, owner, 64);
      const int64_t le0 = ((int64_t)__shfl((int)(e0 >> 32), owner, 64) << 32) |
                          (int64_t)(unsigned int)__shfl((int)(e0 & 0xffffffffll), owner, 64);
      float4 a[CPL];
#pragma unroll
      for (int q = 0; q < CPL; ++q) a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int base = 0; base < n; base += NG * UNR2) {
        int c = 0;
        float v = 0.f;
        if (lane < NG * UNR2 && base + lane < n) {
          c = col[le0 + base + lane];
          v = val[le0 + base + lane];
        }
        float vv[UNR2];
        float4 xv[UNR2][CPL];
#pragma unroll
        for (int u = 0; u < UNR2; ++u) {
          const int idx = u * NG + sub;
          const int cj = __shfl(c, idx, 64);
          vv[u] = __shfl(v, idx, 64);
          const bool p = base + idx < n;
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            const int chunk = li + q * LPR;
            xv[u][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p && chunk < D4) xv[u][q] = x4[(size_t)cj * (size_t)D4 + chunk];
          }
        }
#pragma unroll
        for (int u = 0; u < UNR2; ++u) {
          float4 t[CPL];
#pragma unroll
          for (int q = 0; q < CPL; ++q) t[q] = mul_rn4(vv[u], xv[u][q]);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const bool p = base + u * NG + g < n;  // wave-uniform
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
              const float4 tg = shfl4(t[q], g * LPR + li);
              if (p) a[q] = add_rn4(a[q], tg);
            }
          }
        }
      }
      if (sub == gl) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) sum[q] = a[q];
      }
    }
  }

  if (!row_ok) return;
  if constexpr (SP) {
    if (sa.out_bits && li == 0 && (any_src || (z && zflag))) atomicOr(sa.out_bits + (r >> 5), 1u << (r & 31));
  }
  float4 *y4 = reinterpret_cast<float4 *>(y);
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int chunk = li + q * LPR;
    if (chunk >= D4) continue;
    const size_t o = (size_t)r * (size_t)D4 + chunk;
    float4 s = mul_rn4(alpha, sum[q]);
    if (z) s = add_rn4(s, mul_rn4(beta, zpre[q]));
    if constexpr (SP) {
      if (y && (active || sa.write_zeros)) y4[o] = s;       // (an inactive row: s = +0)
    } else {
      if (y) y4[o] = s;
    }
    if (acc) {
      float4 a0 = acc_init ? mul_rn4(acc_w, apre[q]) : apre[q];
      if (mean_t1) a0 = add_rn4(a0, mul_rn4(acc_w, zpre[q]));
      if constexpr (LPR <= 16 && !ADAM) {
        if (mean_t2) a0 = add_rn4(a0, mul_rn4(acc_w, tpre[q]));
      }
      acc4[o] = add_rn4(a0, mul_rn4(acc_w, s));
    }
    if constexpr (ADAM) {
      const float bc1 = ae.bc[0], bc2_sqrt = ae.bc[1];
      float4 pp = ppre[q], mm = mpre[q], vv = vpre[q];
      adam_update(pp.x, s.x, mm.x, vv.x, ae.c, bc1, bc2_sqrt);
      adam_update(pp.y, s.y, mm.y, vv.y, ae.c, bc1, bc2_sqrt);
      adam_update(pp.z, s.z, mm.z, vv.z, ae.c, bc1, bc2_sqrt);
      adam_update(pp.w, s.w, mm.w, vv.w, ae.c, bc1, bc2_sqrt);
      reinterpret_cast<float4 *>(ae.m)[o] = mm;
      reinterpret_cast<float4 *>(ae.v)[o] = vv;
      reinterpret_cast<float4 *>(ae.p)[o] = pp;
      if (ae.clear_z && z && (zpre[q].x != 0.f || zpre[q].y != 0.f || zpre[q].z != 0.f || zpre[q].w != 0.f))
        reinterpret_cast<float4 *>(const_cast<float *>(z))[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int LPR, int CPL>
static int launch_spmm(const int64_t *rowptr, const int32_t *col, const float *val, const float *x,
                       float *y, int64_t n_rows, int D4, float alpha, const float *z, float beta,
                       float *acc, const float *acc_init, float acc_w, const int32_t *sched, int dyn_val,
                       hipStream_t st, const AdamEpi *adam = nullptr, const float *mean_t1 = nullptr,
                       const float *mean_t2 = nullptr, const RowSparse *rs = nullptr) {
  RowSparse sa;
  std::memset(&sa, 0, sizeof(sa));
  if (rs) sa = *rs;
  const bool sparse = sa.src_bits || sa.z_bits || sa.out_bits || sa.row_bits;
  constexpr int RPW = kWave / LPR;
  const int64_t waves = (n_rows + RPW - 1) / RPW;
  const int64_t blocks = (waves + 3) / 4;     // the schedule has exactly 4 * blocks wave slots
  if (blocks > 0x7fffffffLL) return fail(CHAOREC_E_INVALID, "spmm: grid too large");
  if (mean_t2 && LPR > 16) return fail(CHAOREC_E_INVALID, "spmm: three mean terms need D <= 64");
  if (adam) {
    if constexpr (CPL == 1) {
      if (sparse) return fail(CHAOREC_E_INVALID, "spmm+adam: no row-sparse form (bitmaps can only be cleared here)");
      hipLaunchKernelGGL((spmm_csr_ordered_kernel<LPR, CPL, true>), dim3((unsigned)blocks), dim3(256), 0, st,
                         rowptr, col, val, x, y, n_rows, D4, alpha, z, beta, acc, acc_init, acc_w, sched,
                         waves, dyn_val, *adam, (const float *)nullptr, (const float *)nullptr, sa);
    } else {
      return fail(CHAOREC_E_INVALID, "spmm+adam: D > 256 not built");
    }
  } else if (sparse) {
    if constexpr (CPL == 1 && LPR >= 16) {     // (D = 64 .. 256: the widths whose long-row walks look at the bitmaps)
      hipLaunchKernelGGL((spmm_csr_ordered_kernel<LPR, CPL, false, true>), dim3((unsigned)blocks), dim3(256), 0, st,
                         rowptr, col, val, x, y, n_rows, D4, alpha, z, beta, acc, acc_init, acc_w, sched,
                         waves, dyn_val, AdamEpi{}, mean_t1, mean_t2, sa);
    } else {
      return fail(CHAOREC_E_INVALID, "spmm (row-sparse): built for 64 <= D <= 256");
    }
  } else {
    hipLaunchKernelGGL((spmm_csr_ordered_kernel<LPR, CPL, false>), dim3((unsigned)blocks), dim3(256), 0, st,
                       rowptr, col, val, x, y, n_rows, D4, alpha, z, beta, acc, acc_init, acc_w, sched,
                       waves, dyn_val, AdamEpi{}, mean_t1, mean_t2, sa);
  }
  return check_launch("spmm_csr_ordered_kernel");
}

static int rows_per_wave(int D) {
  if (D < 4 || (D & 3)) return 0;
  int lpr = 1;
  while (lpr < D / 4 && lpr < 64) lpr <<= 1;
  return 64 / lpr;
}

}  // namespace chaorec

using namespace chaorec;

static int spmm_dispatch(const int64_t *rowptr, const int32_t *col, const float *val, const float *x, float *y,
                         int64_t n_rows, int64_t n_cols, int32_t D, float alpha, const float *z, float beta,
                         float *acc, const float *acc_init, float acc_w, const int32_t *schedule, int32_t mode,
                         void *stream, const AdamEpi *adam, const float *mean_t1 = nullptr,
                         const float *mean_t2 = nullptr, const RowSparse *rs = nullptr) {
  if (!rowptr || !x || (!y && !acc && !adam)) return fail(CHAOREC_E_INVALID, "spmm: NULL rowptr/x or no output");
  if (n_rows < 0 || n_cols < 0) return fail(CHAOREC_E_INVALID, "spmm: negative size");
  if (D < 4 || D > 1024 || (D & 3)) return fail(CHAOREC_E_INVALID, "spmm: D=%d must be a multiple of 4 in [4,1024]", D);
  if (mode != 0 && mode != CHAOREC_SPMM_DYNAMIC_VALUES) return fail(CHAOREC_E_INVALID, "spmm: unknown mode %d", mode);
  const int dyn_val = (mode & CHAOREC_SPMM_DYNAMIC_VALUES) && schedule;
  if (acc_init && !acc) return fail(CHAOREC_E_INVALID, "spmm: acc_init without acc");
  if (n_rows == 0) return CHAOREC_OK;
  hipStream_t st = (hipStream_t)stream;
  const int D4 = D / 4;
#define CHAOREC_SPMM_ARGS rowptr, col, val, x, y, n_rows, D4, alpha, z, beta, acc, acc_init, acc_w, schedule, dyn_val, st, adam, mean_t1, mean_t2, rs
  if (D4 <= 1) return launch_spmm<1, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 2) return launch_spmm<2, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 4) return launch_spmm<4, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 8) return launch_spmm<8, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 16) return launch_spmm<16, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 32) return launch_spmm<32, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 64) return launch_spmm<64, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 128) return launch_spmm<64, 2>(CHAOREC_SPMM_ARGS);
  if (D4 <= 192) return launch_spmm<64, 3>(CHAOREC_SPMM_ARGS);
  return launch_spmm<64, 4>(CHAOREC_SPMM_ARGS);
#undef CHAOREC_SPMM_ARGS
}

extern "C" int chaorec_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                                    const float *x, float *y, int64_t n_rows, int64_t n_cols,
                                    int32_t D, float alpha, const float *z, float beta, float *acc,
                                    const float *acc_init, float acc_w, const int32_t *schedule,
                                    int32_t mode, void *stream) {
  return spmm_dispatch(rowptr, col, val, x, y, n_rows, n_cols, D, alpha, z, beta, acc, acc_init, acc_w, schedule, mode,
                       stream, nullptr);
}

extern "C" int chaorec_spmm_csr_rowsparse_f32(const int64_t *rowptr, const int32_t *col, const float *val, const float *x,
                                              float *y, int64_t n_rows, int64_t n_cols, int32_t D, float alpha, const float *z,
                                              float beta, const int32_t *schedule, int32_t mode, const uint32_t *src_bits,
                                              const uint32_t *z_bits, uint32_t *out_bits, const uint32_t *row_bits,
                                              int32_t write_zeros, void *stream) {
  if (!y) return fail(CHAOREC_E_INVALID, "spmm (row-sparse): NULL y");
  if (z_bits && !z) return fail(CHAOREC_E_INVALID, "spmm (row-sparse): z_bits without z");
  if (row_bits && z && !z_bits) return fail(CHAOREC_E_INVALID, "spmm (row-sparse): row_bits with an unflagged z");
  RowSparse rs;
  std::memset(&rs, 0, sizeof(rs));
  rs.src_bits = src_bits;
  rs.z_bits = z_bits;
  rs.out_bits = out_bits;
  rs.row_bits = row_bits;
  rs.write_zeros = write_zeros ? 1 : 0;
  return spmm_dispatch(rowptr, col, val, x, y, n_rows, n_cols, D, alpha, z, beta, nullptr, nullptr, 0.f, schedule, mode, stream,
                       nullptr, nullptr, nullptr, &rs);
}

extern "C" int chaorec_spmm_csr_mean_f32(const int64_t *rowptr, const int32_t *col, const float *val, const float *x,
                                         float *y, int64_t n_rows, int64_t n_cols, int32_t D, float *mean_out,
                                         const float *const *terms, int32_t n_terms, float w, const int32_t *schedule,
                                         int32_t mode, void *stream) {
  if (!mean_out || !terms || n_terms < 1 || n_terms > 3 || !terms[0]) return fail(CHAOREC_E_INVALID, "spmm_mean: bad terms");
  for (int k = 0; k < n_terms; ++k)
    if (!terms[k]) return fail(CHAOREC_E_INVALID, "spmm_mean: NULL term %d", k);
  if (n_terms == 3 && D > 64) return fail(CHAOREC_E_INVALID, "spmm_mean: three terms need D <= 64 (D=%d)", D);
  return spmm_dispatch(rowptr, col, val, x, y, n_rows, n_cols, D, 1.0f, nullptr, 0.f, mean_out, terms[0], w, schedule, mode,
                       stream, nullptr, n_terms > 1 ? terms[1] : nullptr, n_terms > 2 ? terms[2] : nullptr);
}

extern "C" int chaorec_spmm_csr_adam_f32(const int64_t *rowptr, const int32_t *col, const float *val, const float *x,
                                         float *grad_out, int64_t n_rows, int64_t n_cols, int32_t D, float alpha,
                                         float *z, float beta, const int32_t *schedule, int32_t mode, float *param,
                                         float *exp_avg, float *exp_avg_sq, const float *bias_corr, float lr,
                                         float beta1, float beta2, float eps, float weight_decay, int32_t clear_z,
                                         uint32_t *clear_bits_a, int64_t n_words_a, uint32_t *clear_bits_b, int64_t n_words_b,
                                         void *stream) {
  if (!param || !exp_avg || !exp_avg_sq || !bias_corr) return fail(CHAOREC_E_INVALID, "spmm+adam: NULL argument");
  if (n_words_a < 0 || n_words_b < 0) return fail(CHAOREC_E_INVALID, "spmm+adam: negative bitmap length");
  if (D > 256) return fail(CHAOREC_E_INVALID, "spmm+adam: D=%d > 256 not built", D);
  if (clear_z && (const float *)z == x) return fail(CHAOREC_E_INVALID, "spmm+adam: clear_z while gathering from z");
  AdamEpi ae;
  ae.p = param;
  ae.m = exp_avg;
  ae.v = exp_avg_sq;
  ae.bc = bias_corr;
  ae.c = make_adam_consts(lr, beta1, beta2, eps, weight_decay);
  ae.clear_z = clear_z ? 1 : 0;
  RowSparse rs;
  std::memset(&rs, 0, sizeof(rs));
  rs.clear[0] = clear_bits_a, rs.n_clear[0] = clear_bits_a ? n_words_a : 0;
  rs.clear[1] = clear_bits_b, rs.n_clear[1] = clear_bits_b ? n_words_b : 0;
  return spmm_dispatch(rowptr, col, val, x, grad_out, n_rows, n_cols, D, alpha, z, beta, nullptr, nullptr, 0.f, schedule,
                       mode, stream, &ae, nullptr, nullptr, &rs);
}

namespace chaorec {
// bits_out |= bits_self | { c : A[r, c] != 0 for some r flagged in bits_in }: the rows a propagate can make non-zero when
// its source is non-zero in the flagged rows only (a frontier expansion: work ~ the flagged rows' entries, not the
// graph's).  bits_in lives over the CSR's rows, bits_self / bits_out over its COLUMNS (a symmetric graph: bits_self =
// bits_in; a user-shard's rectangular blocks: the batch rows of the other side).  list / list_n (optional): every row whose
// bit this launch sets FIRST is appended (no duplicates, arbitrary order; list_n is the device-side length, zero on entry)
// -- the rows chaorec_spmm_csr_rowlist_f32 computes.
__global__ __launch_bounds__(256) void expand_row_bits_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                              int64_t n_rows, const uint32_t *__restrict__ bits_in,
                                                              int64_t n_words, const uint32_t *__restrict__ bits_self,
                                                              int64_t n_out_rows, uint32_t *bits_out, int32_t *list,
                                                              int32_t *list_n, int64_t list_cap) {
  // One WORKGROUP per pair of bitmap words (most pairs are empty: a load and an exit); a flagged row's entries are walked by
  // all 256 threads -- a popular item's row has 1e4-1e5 entries, and the `old` value of every atomicOr is needed (the list),
  // so each step of the walk is a round trip.
  __shared__ int wtot[4];
  __shared__ int wbase;
  const int lane = threadIdx.x;
  const int64_t wave = blockIdx.x;
  const int64_t n_self_words = bits_self ? (n_out_rows + 31) >> 5 : 0;
  auto flag = [&](int64_t row) {
    const uint32_t m = 1u << (row & 31);
    const uint32_t old = atomicOr(bits_out + (row >> 5), m);
    if (list && !(old & m)) {
      const int at = atomicAdd(list_n, 1);
      if (at < list_cap) list[at] = (int32_t)row;
    }
  };
  for (int k = 0; k < 2; ++k) {
    const int64_t wi = 2 * wave + k;
    if (wi < n_self_words && lane < 32) {
      const uint32_t own = bits_self[wi];
      if (((own >> lane) & 1u) && wi * 32 + lane < n_out_rows) flag(wi * 32 + lane);
    }
    if (wi >= n_words) continue;
    uint32_t word = bits_in[wi];          // wave-uniform
    while (word) {
      const int b = __builtin_ctz(word);
      word &= word - 1;
      const int64_t r = wi * 32 + b;
      if (r >= n_rows) break;
      const int64_t e0 = rowptr[r], e1 = rowptr[r + 1];
      // Eight entries per thread in flight (a step of the walk is a round trip of returning atomics), and ONE list-length
      // atomic per workgroup and step: N1 has 3 M rows at BASELINE configs[4], and 3 M appends -- 45 k wave-aggregated adds --
      // to one address queue up behind each other at its L2 channel (the expansion was 1.2 ms of every light step).
#ifndef CHAOREC_EXPAND_EP
#define CHAOREC_EXPAND_EP 8
#endif
      constexpr int EP = CHAOREC_EXPAND_EP;
      for (int64_t eb = e0; eb < e1; eb += 256 * EP) {          // (block-uniform trip count: barriers inside)
        const int64_t e = eb + lane;
        int c[EP];
        uint32_t old[EP];
#pragma unroll
        for (int j = 0; j < EP; ++j) c[j] = e + 256 * j < e1 ? col[e + 256 * j] : -1;
#pragma unroll
        for (int j = 0; j < EP; ++j) {
          old[j] = 0xFFFFFFFFu;
          if (c[j] >= 0) old[j] = atomicOr(bits_out + (c[j] >> 5), 1u << (c[j] & 31));
        }
        if (list) {                                             // (kernel-uniform)
          int mine = 0;
#pragma unroll
          for (int j = 0; j < EP; ++j) mine += (c[j] >= 0 && !(old[j] & (1u << (c[j] & 31)))) ? 1 : 0;
          // exclusive prefix of `mine` over the workgroup: wave scan + the four wave totals through LDS
          int incl = mine;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if ((lane & 63) >= d) incl += up;
          }
          if ((lane & 63) == 63) wtot[lane >> 6] = incl;
          __syncthreads();
          if (lane == 0) {
            const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            wbase = total ? atomicAdd(list_n, total) : 0;
          }
          __syncthreads();
          int at = wbase + incl - mine;
          for (int w = 0; w < (lane >> 6); ++w) at += wtot[w];
#pragma unroll
          for (int j = 0; j < EP; ++j) {
            if (c[j] >= 0 && !(old[j] & (1u << (c[j] & 31)))) {
              if (at < list_cap) list[at] = (int32_t)c[j];
              ++at;
            }
          }
          __syncthreads();                                      // (wtot / wbase are rewritten by the next step)
        }
      }
    }
  }
}

// y[r] = 0 for every row r flagged in a bitmap (one wave per word; an empty word is a load and an exit): how a buffer that
// is non-zero in a frontier's rows only goes back to all-zero without a pass over the whole of it.
__global__ __launch_bounds__(256) void zero_rows_by_bits_kernel(float *__restrict__ y, int64_t n_rows, int D4,
                                                                const uint32_t *__restrict__ bits, int64_t n_words) {
  const int lane = threadIdx.x & 63;
  const int64_t wi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wi >= n_words) return;
  uint32_t word = bits[wi];
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  while (word) {
    const int b = __builtin_ctz(word);
    word &= word - 1;
    const int64_t r = wi * 32 + b;
    if (r >= n_rows) break;
    for (int c = lane; c < D4; c += 64) reinterpret_cast<float4 *>(y)[(size_t)r * D4 + c] = zero;
  }
}

// list[0 .. *list_n) = the rows flagged in a bitmap (arbitrary order; *list_n zero on entry): the work list of a list launch
// over a frontier that exists as a bitmap only (a user shard's item frontier after the union over the ranks).
__global__ __launch_bounds__(256) void rows_list_from_bits_kernel(const uint32_t *__restrict__ bits, int64_t n_rows,
                                                                  int64_t n_words, int32_t *list, int32_t *list_n,
                                                                  int64_t list_cap) {
  // one list-length atomic per WORKGROUP (its 256 words' counts scanned through LDS): a 3 M-row frontier is 1e5 non-empty
  // words, and as many adds to one address queue up at its L2 channel (see expand_row_bits_kernel)
  __shared__ int wtot[4];
  __shared__ int wbase;
  const int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t word = wi < n_words ? bits[wi] : 0u;
  if (wi < n_words && (wi + 1) * 32 > n_rows)
    word &= (n_rows - wi * 32 >= 32) ? ~0u : ((1u << (n_rows - wi * 32)) - 1u);   // (bits past the end: ignored)
  const int cnt = __popc(word);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wtot[wv] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    wbase = total ? atomicAdd(list_n, total) : 0;
  }
  __syncthreads();
  int at = wbase + incl - cnt;
  for (int w = 0; w < wv; ++w) at += wtot[w];
  while (word) {
    const int b = __builtin_ctz(word);
    word &= word - 1;
    const int64_t r = wi * 32 + b;
    if (r < n_rows && at < list_cap) list[at] = (int32_t)r;
    ++at;
  }
}

// out[r] = ((w t0[r] + w t1[r]) + ..) for the rows flagged in a bitmap (chaorec_rows_mean_f32's association): the layer mean of
// a light step's item rows, whose propagated values arrive with the exchanges.
struct BitsMeanTerms {
  const float4 *t[8];
  int n;
};
__global__ __launch_bounds__(256) void rows_mean_by_bits_kernel(const BitsMeanTerms T, float w, float4 *__restrict__ out,
                                                                int64_t n_rows, int D4, const uint32_t *__restrict__ bits,
                                                                int64_t n_words) {
  const int lane = threadIdx.x & 63;
  const int64_t wi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wi >= n_words) return;
  uint32_t word = bits[wi];
  while (word) {
    const int b = __builtin_ctz(word);
    word &= word - 1;
    const int64_t r = wi * 32 + b;
    if (r >= n_rows) break;
    for (int c = lane; c < D4; c += 64) {
      const size_t o = (size_t)r * D4 + c;
      float4 a = mul_rn4(w, T.t[0][o]);
      for (int k = 1; k < T.n; ++k) a = add_rn4(a, mul_rn4(w, T.t[k][o]));
      out[o] = a;
    }
  }
}

// dst[w] = src[0][w] | src[1][w] | .. | src[n_src - 1][w]: the union of the ranks' row bitmaps after an all-gather (RCCL
// has no bitwise-or reduction).
__global__ __launch_bounds__(256) void or_words_kernel(uint32_t *dst, const uint32_t *src, int n_src,
                                                       int64_t n_words) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  uint32_t acc = 0u;
  for (int k = 0; k < n_src; ++k) acc |= src[(size_t)k * n_words + w];
  dst[w] = acc;
}

// y[r] = alpha * (A x)[r] + beta * z[r] for the rows r of a LIST only (the frontier of a row-sparse backward propagate: 1-2 %
// of the graph at BASELINE configs[4]): one LPR-lane group per listed row, a fixed grid striding over the list.  The row's
// entries are walked in CSR order; an entry whose source row is not flagged in src_bits is skipped (its term is +0), the
// others are gathered and added with separately rounded product and sum -- the arithmetic, and the bits, of
// spmm_csr_ordered_kernel.  Rows outside the list are not touched.
// mean (optional): the layer mean of the LAST forward propagate for the listed rows -- mean.out[r] = ((w t0[r] + w t1[r]) + ..)
// + w s with s the row's propagated value, chaorec_spmm_csr_mean_f32's association -- for a training step that needs the
// propagated table in its batch's rows only; y may then be NULL.
struct ListMean {
  const float *t[4];
  int n;
  float w;
  float *out;
};

// Rows above `long_t` entries (long_list given) are not walked here: one lane group taking a popular item's 1e4-1e5 entries one
// after the other would be the launch's tail.  They are appended to long_list and spmm_rowlist_long_kernel -- a workgroup
// per row -- takes them.
struct LongRows {
  int32_t *list;      // rows deferred by the short-row kernel: long rows from the front, STRIPED rows from the back
  int32_t *cnt;       // [0] = long rows, [1] = workgroups of the long-row kernel that are done, [2] = striped rows, [3] unused
                      // (all zero between launches)
  int64_t cap;
  int t;              // a row with more entries than this is a long row
  int vt;             // ... and with more than this a STRIPED one (INT_MAX: none): see spmm_rowlist_long_ws_kernel
};

__device__ __forceinline__ void rowlist_epilogue(int64_t r, int D4, int li, float4 sum, float alpha, const float *z, float beta,
                                                 float4 zrow, float *y, const ListMean &mean) {
  const size_t o = (size_t)r * D4 + li;
  float4 s = mul_rn4(alpha, sum);
  if (z) s = add_rn4(s, mul_rn4(beta, zrow));
  if (y) reinterpret_cast<float4 *>(y)[o] = s;
  if (mean.out) {
    float4 a = mul_rn4(mean.w, reinterpret_cast<const float4 *>(mean.t[0])[o]);
    for (int k = 1; k < mean.n; ++k) a = add_rn4(a, mul_rn4(mean.w, reinterpret_cast<const float4 *>(mean.t[k])[o]));
    reinterpret_cast<float4 *>(mean.out)[o] = add_rn4(a, mul_rn4(mean.w, s));
  }
}

// Eight waves per SIMD asked for (60 VGPRs with four gathers in flight per group instead of 88 with eight): the launches over
// N1's list are latency chains per row like the gated launch above.  configs[4] whole, tools/rowlist_n1_bench.py: the backward's
// first propagate 2.99 -> 2.24 ms, the forward over N1 13.05 -> 12.85 ms.  (Asking for seven gives the same register count and
// NO gain -- 3.04-3.08 ms -- with 2, 4 or 6 gathers in flight alike: what the bound changes is the compiler's schedule, not
// only the occupancy.)
#ifndef CHAOREC_ROWLIST_MINW
#define CHAOREC_ROWLIST_MINW 8
#endif
#ifndef CHAOREC_ROWLIST_UNR
#define CHAOREC_ROWLIST_UNR 4
#endif
template <int LPR>
__global__ __launch_bounds__(256, CHAOREC_ROWLIST_MINW) void spmm_rowlist_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                           const float *__restrict__ val, const float *__restrict__ x,
                                                           float *__restrict__ y, int D4, float alpha, const float *z, float beta,
                                                           const uint32_t *__restrict__ src_bits,
                                                           const uint32_t *__restrict__ z_bits, const int32_t *__restrict__ list,
                                                           const int32_t *__restrict__ list_n, int64_t list_cap,
                                                           const ListMean mean, const LongRows lr) {
  constexpr int NG = kWave / LPR;
  constexpr int UNR = CHAOREC_ROWLIST_UNR;  // gathered rows in flight per group (ungated walk)
  const int lane = threadIdx.x & 63, sub = lane / LPR, li = lane % LPR;
  const int64_t n = min((int64_t)list_n[0], list_cap);
  const int64_t slots = (int64_t)gridDim.x * (blockDim.x >> 6) * NG;
  const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(x);
  for (int64_t i = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * NG + sub;; i += slots) {
    // (wave-uniform exit: the groups of a wave walk i, i+1, ..; shuffles below stay inside a group's own lanes)
    if (i - sub >= n) break;
    bool ok = i < n;
    const int64_t r = ok ? list[i] : 0;
    const int64_t e0 = ok ? rowptr[r] : 0;
    int deg = ok ? (int)(rowptr[r + 1] - e0) : 0;
    if (lr.list && deg > lr.t) {          // (group-uniform)
      if (li == 0) {
        if (deg > lr.vt) {
          const int at = atomicAdd(lr.cnt + 2, 1);
          if (at < lr.cap) lr.list[lr.cap - 1 - at] = (int32_t)r;
        } else {
          const int at = atomicAdd(lr.cnt, 1);
          if (at < lr.cap) lr.list[at] = (int32_t)r;
        }
      }
      ok = false;
      deg = 0;
    }
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 zrow = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok && z && li < D4 && (!z_bits || row_bit(z_bits, r))) zrow = reinterpret_cast<const float4 *>(z)[(size_t)r * D4 + li];
    const int dmax = group_uniform_max_i32<LPR>(deg);
    if (!src_bits) {
      // every entry is gathered: (col, val) blocks of LPR entries, UNR source rows in flight, adds in entry order
      for (int base = 0; base < dmax; base += LPR) {
        int c = 0;
        float v = 0.f;
        if (base + li < deg) {
          c = col[e0 + base + li];
          v = val[e0 + base + li];
        }
        const int nn = min(LPR, deg - base);
        const int nmax = min(LPR, dmax - base);
        for (int j = 0; j < nmax; j += UNR) {
          int cj[UNR];
          float vj[UNR];
          float4 xv[UNR];
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int src = sub * LPR + ((j + u) & (LPR - 1));
            cj[u] = __shfl(c, src, 64);
            vj[u] = __shfl(v, src, 64);
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j + u < nn && li < D4) xv[u] = x4[(size_t)cj[u] * (size_t)D4 + li];
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u)
            if (j + u < nn) sum = add_rn4(sum, mul_rn4(vj[u], xv[u]));
        }
      }
    } else {
      for (int base = 0; base < dmax; base += LPR) {
        int c = 0, keep = 0;
        float v = 0.f;
        if (base + li < deg) {
          c = col[e0 + base + li];
          v = val[e0 + base + li];
          keep = row_bit(src_bits, c) ? 1 : 0;
        }
        // the flagged entries of this block, in entry order (a group-local mask: LPR <= 64 bits)
        unsigned long long m = __ballot(keep != 0);
        m = (m >> (sub * LPR)) & (LPR == 64 ? ~0ull : ((1ull << LPR) - 1ull));
        while (__any(m != 0ull)) {            // (trip counts differ between the wave's groups: every lane stays in the loop)
          const bool has = m != 0ull;
          const int j = has ? __builtin_ctzll(m) : 0;
          if (has) m &= m - 1;
          const int src = sub * LPR + j;
          const int cj = __shfl(c, src, 64);
          const float vj = __shfl(v, src, 64);
          if (has && li < D4) sum = add_rn4(sum, mul_rn4(vj, x4[(size_t)cj * (size_t)D4 + li]));
        }
      }
    }
    if (ok && li < D4) rowlist_epilogue(r, D4, li, sum, alpha, z, beta, zrow, y, mean);
  }
}

// The long rows of a list launch: ONE WORKGROUP per row.  The row's entries are taken CH at a time: every lane group gathers
// K source rows and parks the products v * x in an LDS tile, then the row's D columns are summed by D threads walking the
// tile in ENTRY ORDER -- the sequential CSR-order sum of the other kernels, bit for bit; only the loads are shared.  The
// walk is a three-stage pipeline: while round i is summed, round i+1's source rows and round i+2's (col, val) are in
// flight, so a round costs one memory latency, not two dependent ones plus the sum.  The last workgroup to finish zeroes
// the list's counters for the next launch.
template <int LPR>
__global__ __launch_bounds__(256) void spmm_rowlist_long_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                const float *__restrict__ val, const float *__restrict__ x,
                                                                float *__restrict__ y, int D4, float alpha, const float *z,
                                                                float beta, const uint32_t *__restrict__ z_bits,
                                                                const ListMean mean, const LongRows lr) {
  constexpr int NGB = 256 / LPR;        // lane groups per workgroup
  constexpr int K = 16;                 // source rows per group and round
  constexpr int CH = NGB * K;           // entries per round (256 / 128 / 64 for D4 = 16 / 32 / 64)
  __shared__ float4 tile[CH * LPR];     // 64 KiB
  const int tid = threadIdx.x, g = tid / LPR, li = tid % LPR;
  const int64_t n = min((int64_t)lr.cnt[0], lr.cap);
  const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(x);
  const float *tile_f = reinterpret_cast<const float *>(tile);
  const int D = 4 * D4;
  constexpr int DT = 4 * LPR;           // (D4 <= LPR: the tile's rows are LPR float4 wide, the first D4 in use)
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t r = lr.list[i];
    const int64_t e0 = rowptr[r];
    const int deg = (int)(rowptr[r + 1] - e0);
    float s = 0.f;                      // column tid of the running sum (threads tid < D)
    // BRANCH-FREE loads (an index past the row's end is clamped to its last entry: loaded, parked, never summed) -- a
    // conditional load makes the compiler wait for each gather before it issues the next
    int cn[K];                          // round i+2's (col, val)
    float vn[K];
    float4 xv[K];                       // round i+1's source rows, as loaded
    float vx[K];                        //             ... and their values
    const int lic = min(li, D4 - 1);
    auto load_cv = [&](int base) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int64_t e = e0 + min(base + g * K + k, deg - 1);
        cn[k] = col[e];
        vn[k] = val[e];
      }
    };
    auto gather = [&]() {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        xv[k] = x4[(size_t)cn[k] * (size_t)D4 + lic];
        vx[k] = vn[k];
      }
    };
    load_cv(0);
    gather();
    load_cv(CH);
    for (int base = 0; base < deg; base += CH) {
      __syncthreads();                  // (the previous round's sums have read the tile)
#pragma unroll
      for (int k = 0; k < K; ++k) tile[(g * K + k) * LPR + li] = mul_rn4(vx[k], xv[k]);
      __syncthreads();
      if (base + CH < deg) {            // (block-uniform)
        gather();                       // round i+1's source rows ...
        load_cv(base + 2 * CH);         // ... and round i+2's (col, val)
      }
      if (tid < D) {
        // entry e's product of column tid sits at tile_f[e * DT + tid]
        const float *t = tile_f + tid;
        const int nn = min(CH, deg - base);
        if (nn == CH) {
#pragma unroll 16
          for (int e = 0; e < CH; ++e) s = add_rn(s, t[e * DT]);
        } else {
          for (int e = 0; e < nn; ++e) s = add_rn(s, t[e * DT]);
        }
      }
    }
    __syncthreads();
    // the epilogue works on float4s (lanes li < D4 of group 0): hand the column sums over through the tile
    if (tid < D) reinterpret_cast<float *>(tile)[tid] = s;
    __syncthreads();
    if (g == 0 && li < D4) {
      float4 zrow = make_float4(0.f, 0.f, 0.f, 0.f);
      if (z && (!z_bits || row_bit(z_bits, r))) zrow = reinterpret_cast<const float4 *>(z)[(size_t)r * D4 + li];
      rowlist_epilogue(r, D4, li, tile[li], alpha, z, beta, zrow, y, mean);
    }
    __syncthreads();
  }
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(lr.cnt + 1, 1) == (int)gridDim.x - 1) {
      lr.cnt[0] = 0;
      lr.cnt[1] = 0;
    }
  }
}
// The same launch with SPECIALISED WAVES (round 4's second form; `CHAOREC_ROWLIST_LONG_TEAMS` = 0 selects the kernel above).
// The one-slot walk above costs one gather latency per round (~2.5 us for random 512-B rows of a multi-GB table, against
// 0.4-0.8 us of ordered adds), and keeping two rounds in flight in ONE wave's registers did not work: the compiler's wait at
// the top of the loop body became vmcnt(0) whatever the arrangement.  vmcnt is a PER-WAVE counter, so here the
// rounds in flight belong to different waves: T loader teams of 4 waves take the rounds round-robin -- a team gathers its
// round's source rows right after it parked the previous one's products, and waits for them (its own vmcnt(0)) T phases
// later --, SW summing waves never touch global memory and add tile p-1 in entry order while tile p is being written (two
// LDS tiles).  One workgroup barrier per phase.  The sum is the same sequential CSR-order sum, bit for bit.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int LPR>
constexpr int ws_summer_waves() { return (4 * LPR + 63) / 64; }

// One row (or one 128-byte column stripe of it) through the loader teams and the summing waves.  LPRX lanes per entry; x4w, c0:
// the window's first float4 column (x4w = x4 + c0), D4w its width; D4 the tables' row stride in float4.
template <int LPRX, int T, int SW>
__device__ __forceinline__ void long_ws_row(const int32_t *__restrict__ col, const float *__restrict__ val,
                                            const float4 *__restrict__ x4w, float *__restrict__ y, int D4, int D4w, int c0,
                                            float alpha, const float *z, float beta, const uint32_t *__restrict__ z_bits,
                                            const ListMean &mean, int64_t r, int64_t e0, int deg, float4 *tile0, float4 *tile1) {
  constexpr int LW = 4, K = 16;
  constexpr int NGB = LW * 64 / LPRX;     // lane groups per loader team
  constexpr int CH = NGB * K;             // entries per round: a 64 KiB tile
  constexpr int DT = 4 * LPRX;
  const int tid = threadIdx.x, wave = tid >> 6;
  const bool summer = wave < SW;
  const int team = summer ? -1 : (wave - SW) / LW;
  const int lt = summer ? 0 : tid - (SW + team * LW) * 64;      // thread index inside its team
  const int g = lt / LPRX, li = lt % LPRX;
  const int lic = min(li, D4w - 1);
  const int D = 4 * D4w;
  const int rounds = (deg + CH - 1) / CH;
  float s = 0.f;                        // (summing waves: column tid of the running sum)
  int cn[K];
  float4 xv[K];
  float vv[K];
  if (!summer) {                        // the team's first round (= its number), and the columns of its second
#pragma unroll
    for (int k = 0; k < K; ++k) cn[k] = col[e0 + min(team * CH + g * K + k, deg - 1)];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      xv[k] = x4w[(size_t)cn[k] * (size_t)D4 + lic];
      vv[k] = val[e0 + min(team * CH + g * K + k, deg - 1)];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) cn[k] = col[e0 + min((team + T) * CH + g * K + k, deg - 1)];
  }
  for (int pb = 0; pb < rounds + 1; pb += T) {
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
      const int p = pb + tt;            // phase p: round p is parked, round p - 1 is summed
      if (!summer && team == tt && p < rounds) {
        float4 *tw = (p & 1) ? tile1 : tile0;
#pragma unroll
        for (int k = 0; k < K; ++k) tw[(g * K + k) * LPRX + li] = mul_rn4(vv[k], xv[k]);
#pragma unroll
        for (int k = 0; k < K; ++k) {   // the team's next round (p + T): clamped past the row's end, never summed
          xv[k] = x4w[(size_t)cn[k] * (size_t)D4 + lic];
          vv[k] = val[e0 + min((p + T) * CH + g * K + k, deg - 1)];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) cn[k] = col[e0 + min((p + 2 * T) * CH + g * K + k, deg - 1)];
      }
      if (summer && p >= 1 && p - 1 < rounds && tid < D) {
        const float *t = reinterpret_cast<const float *>(((p - 1) & 1) ? tile1 : tile0) + tid;
        const int nn = min(CH, deg - (p - 1) * CH);
        if (nn == CH) {
          // The chain of dependent adds is what a striped row's time is made of.  The next 16 products are read from LDS while
          // the current 16 are added; the scheduling barriers keep the two groups apart -- left alone, the compiler issued a
          // batch's reads LAST-needed first and waited for all of them before the first add, the LDS latency once per 16 adds
          // (configs[4]'s batch launch, tools/rowlist_r0_bench.py: 1.75 ms that way, 1.5 ms this way; a branch-free reload, a
          // raised wave priority and conflict-free parking of the tile rows each measured the same or worse).
          constexpr int SB = 16;
          static_assert(CH % (2 * SB) == 0, "two batches per turn");
          float a[SB], b[SB];
#pragma unroll
          for (int q = 0; q < SB; ++q) a[q] = t[q * DT];
#pragma unroll 1
          for (int e = 0; e < CH; e += 2 * SB) {
#pragma unroll
            for (int q = 0; q < SB; ++q) b[q] = t[(e + SB + q) * DT];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < SB; ++q) s = add_rn(s, a[q]);
            __builtin_amdgcn_sched_barrier(0);
            if (e + 2 * SB < CH) {
#pragma unroll
              for (int q = 0; q < SB; ++q) a[q] = t[(e + 2 * SB + q) * DT];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < SB; ++q) s = add_rn(s, b[q]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          for (int e = 0; e < nn; ++e) s = add_rn(s, t[e * DT]);
        }
      }
      lds_barrier();
    }
  }
  // the epilogue works on float4s: hand the column sums over through tile 0
  if (summer && tid < D) reinterpret_cast<float *>(tile0)[tid] = s;
  lds_barrier();
  if (tid < D4w) {
    float4 zrow = make_float4(0.f, 0.f, 0.f, 0.f);
    if (z && (!z_bits || row_bit(z_bits, r))) zrow = reinterpret_cast<const float4 *>(z)[(size_t)r * D4 + c0 + tid];
    rowlist_epilogue(r, D4, c0 + tid, tile0[tid], alpha, z, beta, zrow, y, mean);
  }
  lds_barrier();
}

// STRIPED rows (round 6).  One workgroup moves ~50 GB/s of gathered rows (two teams x 64 KiB in flight, ~2.5 us a round trip),
// so a popular item's 2e5 entries x 512 B took 2 ms however many other workgroups sat idle -- the whole of a light step's
// launch over its batch's rows (0.11 of the HBM rate).  The ORDER of a row's sum cannot be cut, its COLUMNS can: a row above
// `vt` entries is taken by D4 / 8 workgroups, each gathering one 128-byte stripe (8 float4 = a whole cache line) of every
// source row and summing its 32 columns in entry order -- the same chain per output element, bit for bit, with D4 / 8 times
// the bytes in flight.  (The chain itself, one dependent add per entry, is what is left: ~0.4 ms for 2e5 entries.)
constexpr int kStripeW = 8;             // float4 per stripe
constexpr int64_t kStripeListCap = 65536;   // lists above this many rows stripe four times later (see chaorec_spmm_csr_rowlist_f32)
template <int LPR, int T>
__global__ __launch_bounds__((ws_summer_waves<LPR>() + 4 * T) * 64) void spmm_rowlist_long_ws_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ x, float *__restrict__ y, int D4, float alpha, const float *z, float beta,
    const uint32_t *__restrict__ z_bits, const ListMean mean, const LongRows lr) {
  extern __shared__ float4 ws_tiles[];
  constexpr int SW = ws_summer_waves<LPR>();
  constexpr int CHF = (4 * 64 / LPR) * 16;     // entries per round of a full-width row (a 64 KiB tile either way)
  float4 *tile0 = ws_tiles, *tile1 = ws_tiles + CHF * LPR;
  const int tid = threadIdx.x;
  const int64_t n = min((int64_t)lr.cnt[0], lr.cap);
  const int64_t nv = min((int64_t)lr.cnt[2], lr.cap - n);
  const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(x);
  // the striped rows first (the launch's longest pieces of work): item = (row, stripe), stripes of a row on neighbouring ids
  const int S = D4 / kStripeW;
  for (int64_t it = blockIdx.x; it < nv * S; it += gridDim.x) {
    const int64_t r = lr.list[lr.cap - 1 - it / S];
    const int c0 = (int)(it % S) * kStripeW;
    const int64_t e0 = rowptr[r];
    const int deg = (int)(rowptr[r + 1] - e0);
    long_ws_row<kStripeW, T, SW>(col, val, x4 + c0, y, D4, kStripeW, c0, alpha, z, beta, z_bits, mean, r, e0, deg, tile0, tile1);
  }
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t r = lr.list[i];
    const int64_t e0 = rowptr[r];
    const int deg = (int)(rowptr[r + 1] - e0);
    long_ws_row<LPR, T, SW>(col, val, x4, y, D4, D4, 0, alpha, z, beta, z_bits, mean, r, e0, deg, tile0, tile1);
  }
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(lr.cnt + 1, 1) == (int)gridDim.x - 1) {
      lr.cnt[0] = 0;
      lr.cnt[1] = 0;
      lr.cnt[2] = 0;
    }
  }
}

// The long rows of a GATED list launch (the backward's first propagate: a listed row's entries count only where the source is
// flagged, a handful among a popular item's 1e4-1e5): one WORKGROUP per row scans 1024 entries per round -- coalesced (col, val)
// loads, a bit test each -- and queues the flagged ones in ENTRY ORDER (wave ballots + a prefix over the 4 x 4 counts); the
// queue is then gathered and added in that order by one lane group: the sequential CSR-order sum over the flagged entries,
// which is the dense sum bit for bit (the others contribute +0).  One lane group scanning such a row alone was the launch.
__global__ __launch_bounds__(256) void spmm_rowlist_long_gated_kernel(const int64_t *__restrict__ rowptr,
                                                                      const int32_t *__restrict__ col,
                                                                      const float *__restrict__ val, const float *__restrict__ x,
                                                                      float *__restrict__ y, int D4, float alpha, const float *z,
                                                                      float beta, const uint32_t *__restrict__ src_bits,
                                                                      const uint32_t *__restrict__ z_bits, const ListMean mean,
                                                                      const LongRows lr) {
  constexpr int Q = 2048, U4 = 4;
  __shared__ int q_c[Q];
  __shared__ float q_v[Q];
  __shared__ int cnt[U4][4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t n = min((int64_t)lr.cnt[0], lr.cap);
  const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(x);
  const int lic = min(lane, D4 - 1);
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t r = lr.list[i];
    const int64_t e0 = rowptr[r];
    const int deg = (int)(rowptr[r + 1] - e0);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);     // (wave 0, lanes < D4)
    int qn = 0;                                        // queued entries (block-uniform)
    auto flush = [&]() {                               // (called by every thread; the queue is complete and visible)
      if (wave == 0) {
        for (int j = 0; j < qn; j += 8) {
          float4 xv[8];
          float vv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int jj = min(j + u, qn - 1);
            xv[u] = x4[(size_t)q_c[jj] * (size_t)D4 + lic];
            vv[u] = q_v[jj];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (j + u < qn) sum = add_rn4(sum, mul_rn4(vv[u], xv[u]));
        }
      }
      __syncthreads();
      qn = 0;
    };
    for (int base = 0; base < deg; base += U4 * 256) {
      int c[U4];
      float v[U4];
      bool ok[U4];
#pragma unroll
      for (int u = 0; u < U4; ++u) {
        const int e = base + u * 256 + tid;
        ok[u] = e < deg;
        const int64_t ec = e0 + min(e, deg - 1);
        c[u] = col[ec];
        v[u] = val[ec];
      }
      int rank_in_wave[U4], n_wave[U4];
#pragma unroll
      for (int u = 0; u < U4; ++u) {
        ok[u] = ok[u] && row_bit(src_bits, c[u]);
        const unsigned long long b = __ballot(ok[u]);
        rank_in_wave[u] = __popcll(b & ((1ull << lane) - 1ull));
        n_wave[u] = __popcll(b);
        if (lane == 0) cnt[u][wave] = n_wave[u];
      }
      __syncthreads();
      int before[U4], total = 0;                       // entry order = u-major, wave-minor, lane
#pragma unroll
      for (int u = 0; u < U4; ++u) {
        int mine = 0, all_u = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          if (w < wave) mine += cnt[u][w];
          all_u += cnt[u][w];
        }
        before[u] = total + mine;
        total += all_u;
      }
      if (qn + total > Q) {                            // (block-uniform) -- cannot happen twice: total <= 1024 < Q
        __syncthreads();
        flush();
      }
#pragma unroll
      for (int u = 0; u < U4; ++u)
        if (ok[u]) {
          q_c[qn + before[u] + rank_in_wave[u]] = c[u];
          q_v[qn + before[u] + rank_in_wave[u]] = v[u];
        }
      qn += total;
      __syncthreads();                                 // (the counts are re-written next round; the queue is visible)
    }
    flush();
    if (wave == 0 && lane < D4) {
      float4 zrow = make_float4(0.f, 0.f, 0.f, 0.f);
      if (z && (!z_bits || row_bit(z_bits, r))) zrow = reinterpret_cast<const float4 *>(z)[(size_t)r * D4 + lane];
      rowlist_epilogue(r, D4, lane, sum, alpha, z, beta, zrow, y, mean);
    }
  }
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(lr.cnt + 1, 1) == (int)gridDim.x - 1) {
      lr.cnt[0] = 0;
      lr.cnt[1] = 0;
    }
  }
}
}  // namespace chaorec

extern "C" int chaorec_expand_row_bits(const int64_t *rowptr, const int32_t *col, int64_t n_rows, const uint32_t *bits_in,
                                       const uint32_t *bits_self, int64_t n_out_rows, uint32_t *bits_out, int32_t *list,
                                       int32_t *list_n, int64_t list_cap, void *stream) {
  if (!rowptr || !col || !bits_in || !bits_out) return fail(CHAOREC_E_INVALID, "expand_row_bits: NULL argument");
  if (n_rows <= 0 || n_rows > 0x7fffffffLL || n_out_rows <= 0 || n_out_rows > 0x7fffffffLL)
    return fail(CHAOREC_E_INVALID, "expand_row_bits: n_rows=%lld n_out_rows=%lld", (long long)n_rows, (long long)n_out_rows);
  if ((list == nullptr) != (list_n == nullptr) || (list && list_cap <= 0))
    return fail(CHAOREC_E_INVALID, "expand_row_bits: list, list_n and list_cap come together");
  const int64_t n_words = (n_rows + 31) / 32;
  const int64_t n_self = bits_self ? (n_out_rows + 31) / 32 : 0;
  const int64_t groups = (std::max(n_words, n_self) + 1) / 2;
  hipLaunchKernelGGL(expand_row_bits_kernel, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, rowptr, col,
                     n_rows, bits_in, n_words, bits_self, n_out_rows, bits_out, list, list_n, list_cap);
  return check_launch("expand_row_bits_kernel");
}

extern "C" int chaorec_zero_rows_by_bits_f32(float *y, int64_t n_rows, int32_t D, const uint32_t *bits, void *stream) {
  if (!y || !bits) return fail(CHAOREC_E_INVALID, "zero_rows_by_bits: NULL argument");
  if (n_rows <= 0 || D <= 0 || (D & 3)) return fail(CHAOREC_E_INVALID, "zero_rows_by_bits: n_rows=%lld D=%d", (long long)n_rows, D);
  const int64_t n_words = (n_rows + 31) / 32;
  hipLaunchKernelGGL(zero_rows_by_bits_kernel, dim3((unsigned)((n_words + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, n_rows,
                     D / 4, bits, n_words);
  return check_launch("zero_rows_by_bits_kernel");
}

extern "C" int chaorec_rows_list_from_bits(const uint32_t *bits, int64_t n_rows, int32_t *list, int32_t *list_n, int64_t list_cap,
                                           void *stream) {
  if (!bits || !list || !list_n) return fail(CHAOREC_E_INVALID, "rows_list_from_bits: NULL argument");
  if (n_rows <= 0 || list_cap <= 0) return fail(CHAOREC_E_INVALID, "rows_list_from_bits: bad sizes");
  const int64_t n_words = (n_rows + 31) / 32;
  hipLaunchKernelGGL(rows_list_from_bits_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bits,
                     n_rows, n_words, list, list_n, list_cap);
  return check_launch("rows_list_from_bits_kernel");
}

extern "C" int chaorec_rows_mean_by_bits_f32(const float *const *terms, int32_t n_terms, float w, float *out, int64_t n_rows,
                                             int32_t D, const uint32_t *bits, void *stream) {
  if (!terms || !out || !bits) return fail(CHAOREC_E_INVALID, "rows_mean_by_bits: NULL argument");
  if (n_terms < 1 || n_terms > 8) return fail(CHAOREC_E_INVALID, "rows_mean_by_bits: n_terms=%d must be in [1, 8]", n_terms);
  if (n_rows <= 0 || D <= 0 || (D & 3)) return fail(CHAOREC_E_INVALID, "rows_mean_by_bits: n_rows=%lld D=%d", (long long)n_rows, D);
  BitsMeanTerms T;
  T.n = n_terms;
  for (int k = 0; k < 8; ++k) {
    if (k < n_terms && !terms[k]) return fail(CHAOREC_E_INVALID, "rows_mean_by_bits: NULL term %d", k);
    T.t[k] = (const float4 *)(k < n_terms ? terms[k] : terms[0]);
  }
  const int64_t n_words = (n_rows + 31) / 32;
  hipLaunchKernelGGL(rows_mean_by_bits_kernel, dim3((unsigned)((n_words + 3) / 4)), dim3(256), 0, (hipStream_t)stream, T, w,
                     (float4 *)out, n_rows, D / 4, bits, n_words);
  return check_launch("rows_mean_by_bits_kernel");
}

extern "C" int chaorec_or_words_u32(uint32_t *dst, const uint32_t *src, int32_t n_src, int64_t n_words, void *stream) {
  if (!dst || !src) return fail(CHAOREC_E_INVALID, "or_words: NULL argument");
  if (n_src <= 0 || n_words <= 0) return fail(CHAOREC_E_INVALID, "or_words: n_src=%d n_words=%lld", n_src, (long long)n_words);
  hipLaunchKernelGGL(or_words_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n_src,
                     n_words);
  return check_launch("or_words_kernel");
}

extern "C" int chaorec_spmm_csr_rowlist_f32(const int64_t *rowptr, const int32_t *col, const float *val, const float *x, float *y,
                                            int64_t n_rows, int32_t D, float alpha, const float *z, float beta,
                                            const uint32_t *src_bits, const uint32_t *z_bits, const int32_t *list,
                                            const int32_t *list_n, int64_t list_cap, float *mean_out,
                                            const float *const *mean_terms, int32_t n_mean_terms, float mean_w,
                                            int32_t *long_list, int32_t *long_cnt, int64_t long_cap, int32_t long_threshold,
                                            void *stream) {
  if (!rowptr || !col || !val || !x || !list || !list_n) return fail(CHAOREC_E_INVALID, "spmm (row list): NULL argument");
  if (!y && !mean_out) return fail(CHAOREC_E_INVALID, "spmm (row list): neither y nor mean_out");
  if (n_rows <= 0 || list_cap <= 0) return fail(CHAOREC_E_INVALID, "spmm (row list): bad sizes");
  const int D4 = D / 4;
  if (D < 64 || D > 256 || (D & 3)) return fail(CHAOREC_E_INVALID, "spmm (row list): D=%d must be a multiple of 4 in [64, 256]", D);
  if ((long_list == nullptr) != (long_cnt == nullptr) || (long_list && (long_cap <= 0 || long_threshold < 1)))
    return fail(CHAOREC_E_INVALID, "spmm (row list): long_list, long_cnt, long_cap and long_threshold come together");
  ListMean mean;
  std::memset(&mean, 0, sizeof(mean));
  if (mean_out) {
    if (!mean_terms || n_mean_terms < 1 || n_mean_terms > 4) return fail(CHAOREC_E_INVALID, "spmm (row list): 1..4 mean terms");
    for (int k = 0; k < n_mean_terms; ++k) {
      if (!mean_terms[k]) return fail(CHAOREC_E_INVALID, "spmm (row list): NULL mean term %d", k);
      mean.t[k] = mean_terms[k];
    }
    mean.n = n_mean_terms, mean.w = mean_w, mean.out = mean_out;
  }
  LongRows lr;
  lr.list = long_list, lr.cnt = long_cnt, lr.cap = long_cap, lr.t = long_threshold;
  static const int teams = [] {
    const char *e = std::getenv("CHAOREC_ROWLIST_LONG_TEAMS");
    return (e && std::atoi(e) == 0) ? 0 : 2;      // (three teams: 128 VGPRs at 14 waves per workgroup, spills, slower)
  }();
  // striped rows: only the ungated long-row launch with specialised waves takes them, and only whole stripes
  const char *se = std::getenv("CHAOREC_ROWLIST_STRIPE_T");      // (read per call: the tests move it; 0: none)
  const int stripe_t = (se && *se) ? std::atoi(se) : 8192;
  // A launch over a SHORT list (a batch's rows: a few hundred long rows, the longest of them the launch's tail) stripes from
  // stripe_t entries on.  A long list (N1: 2e4 long rows) keeps every workgroup busy near the HBM rate anyway; there only the
  // very longest rows are worth it, the ones whose single workgroup would still be walking when the rest is done (configs[4]
  // whole, forward over N1's list, tools/rowlist_n1_bench.py: 13.8 ms without stripes, 13.3 / 12.9 / 13.0 / 13.3 ms with
  // stripes above 16 / 32 / 64 / 128 K entries).
  const int st_eff = list_cap <= kStripeListCap ? stripe_t : 4 * stripe_t;
  lr.vt = (!src_bits && teams && st_eff > 0 && D4 % kStripeW == 0 && D4 > kStripeW)
              ? std::max(st_eff, long_threshold) : INT_MAX;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(2048), block(256);        // a fixed grid striding over the device-side list
#define CHAOREC_ROWLIST_ARGS rowptr, col, val, x, y, D4, alpha, z, beta, src_bits, z_bits, list, list_n, list_cap, mean, lr
  if (D4 <= 16) hipLaunchKernelGGL((spmm_rowlist_kernel<16>), grid, block, 0, st, CHAOREC_ROWLIST_ARGS);
  else if (D4 <= 32) hipLaunchKernelGGL((spmm_rowlist_kernel<32>), grid, block, 0, st, CHAOREC_ROWLIST_ARGS);
  else hipLaunchKernelGGL((spmm_rowlist_kernel<64>), grid, block, 0, st, CHAOREC_ROWLIST_ARGS);
#undef CHAOREC_ROWLIST_ARGS
  int rc = check_launch("spmm_rowlist_kernel");
  if (rc != CHAOREC_OK || !long_list) return rc;
  const dim3 lgrid(1024);                   // one workgroup per long row, striding
  if (src_bits) {
#define CHAOREC_ROWLIST_GATED_ARGS rowptr, col, val, x, y, D4, alpha, z, beta, src_bits, z_bits, mean, lr
    hipLaunchKernelGGL(spmm_rowlist_long_gated_kernel, lgrid, block, 0, st, CHAOREC_ROWLIST_GATED_ARGS);
#undef CHAOREC_ROWLIST_GATED_ARGS
    return check_launch("spmm_rowlist_long_gated_kernel");
  }
#define CHAOREC_ROWLIST_LONG_ARGS rowptr, col, val, x, y, D4, alpha, z, beta, z_bits, mean, lr
  if (teams) {
    // specialised waves: T loader teams of 4 waves + the summing waves, two 64 KiB LDS tiles (dynamic: above the 64 KiB of
    // static LDS a kernel may declare)
    constexpr unsigned kTiles = 2 * 65536;
#define CHAOREC_WS_LAUNCH(LPR_, T_)                                                                                   \
  do {                                                                                                                \
    static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&spmm_rowlist_long_ws_kernel<LPR_, T_>), \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, kTiles);          \
    if (attr_ != hipSuccess) return fail(CHAOREC_E_LAUNCH, "spmm (row list): cannot get %u bytes of LDS", kTiles);  \
    hipLaunchKernelGGL((spmm_rowlist_long_ws_kernel<LPR_, T_>), lgrid, dim3((ws_summer_waves<LPR_>() + 4 * T_) * 64), \
                       kTiles, st, CHAOREC_ROWLIST_LONG_ARGS);                                                        \
  } while (0)
    if (D4 <= 16) CHAOREC_WS_LAUNCH(16, 2);
    else if (D4 <= 32) CHAOREC_WS_LAUNCH(32, 2);
    else CHAOREC_WS_LAUNCH(64, 2);
#undef CHAOREC_WS_LAUNCH
    return check_launch("spmm_rowlist_long_ws_kernel");
  }
  if (D4 <= 16) hipLaunchKernelGGL((spmm_rowlist_long_kernel<16>), lgrid, block, 0, st, CHAOREC_ROWLIST_LONG_ARGS);
  else if (D4 <= 32) hipLaunchKernelGGL((spmm_rowlist_long_kernel<32>), lgrid, block, 0, st, CHAOREC_ROWLIST_LONG_ARGS);
  else hipLaunchKernelGGL((spmm_rowlist_long_kernel<64>), lgrid, block, 0, st, CHAOREC_ROWLIST_LONG_ARGS);
#undef CHAOREC_ROWLIST_LONG_ARGS
  return check_launch("spmm_rowlist_long_kernel");
}

extern "C" int chaorec_spmm_rows_per_wave(int32_t D) { return rows_per_wave(D); }
// rows with more entries than this are walked cooperatively by their workgroup (the schedule builders group by it)
extern "C" int chaorec_spmm_long_threshold(void) { return CHAOREC_SPMM_LONG_T; }

// ---- host-side schedule builder (no GPU work) --------------------------------------------------
// Wave slot s (4 per workgroup) handles the G = rows_per_wave(D) consecutive rows of one group.  Groups are
// sorted by their heaviest row and dealt out so that workgroup b receives the b-th heaviest group plus one
// group from each lighter quantile: long rows start first AND land in different workgroups (a workgroup walks
// its long rows one at a time).  Every row gets a 64-B descriptor in slot order.
extern "C" int64_t chaorec_spmm_schedule_len(int64_t n_rows, int32_t D) {
  const int g = rows_per_wave(D);
  if (g == 0 || n_rows <= 0) return 0;
  const int64_t groups = (n_rows + g - 1) / g;
  const int64_t blocks = (groups + 3) / 4;
  return blocks * 4 * g * kDescDwords;
}

extern "C" int chaorec_spmm_build_schedule(const int64_t *rowptr, const int32_t *col, const float *val,
                                           int64_t n_rows, int32_t D, int32_t *out, int64_t out_len) {
  const int g = rows_per_wave(D);
  if (!rowptr || !out || g == 0 || n_rows <= 0) return fail(CHAOREC_E_INVALID, "build_schedule: bad argument");
  const int64_t need = chaorec_spmm_schedule_len(n_rows, D);
  if (out_len < need) return fail(CHAOREC_E_WORKSPACE, "build_schedule: out_len %lld < %lld", (long long)out_len, (long long)need);
  const int64_t groups = (n_rows + g - 1) / g;
  const int64_t nb = (groups + 3) / 4;
  // Which rows share a wave.  A wave's g lane groups finish together with its LONGEST row, so rows are grouped by
  // degree, not by index (a descriptor carries its row id: any row can sit in any slot):
  //   * rows above LONG_T (walked cooperatively by the whole block, one after the other) are dealt one per group,
  //     longest first, each with the g-1 SHORTEST rows as company -- never two long rows behind each other in a block
  //     unless there are more long rows than blocks;
  //   * all other rows, sorted by degree, fill the remaining groups g at a time: equal trip counts inside a wave.
  std::vector<int64_t> rows_sorted(n_rows);
  for (int64_t r = 0; r < n_rows; ++r) rows_sorted[r] = r;
  std::stable_sort(rows_sorted.begin(), rows_sorted.end(), [&](int64_t a, int64_t b) {
    return rowptr[a + 1] - rowptr[a] > rowptr[b + 1] - rowptr[b];
  });
  int64_t n_long = 0;
  if (g > 1)
    while (n_long < n_rows && rowptr[rows_sorted[n_long] + 1] - rowptr[rows_sorted[n_long]] > CHAOREC_SPMM_LONG_T) ++n_long;
  if (n_long > groups) n_long = groups;       // (more long rows than waves: the surplus is grouped like the rest)
  std::vector<int64_t> slot_row((size_t)groups * g, -1);
  {
    int64_t lo = n_long, hi = n_rows - 1;     // unassigned rows: rows_sorted[lo .. hi], longest at lo
    for (int64_t gi = 0; gi < n_long; ++gi) {
      slot_row[(size_t)gi * g] = rows_sorted[gi];
      for (int s2 = 1; s2 < g && hi >= lo; ++s2) slot_row[(size_t)gi * g + s2] = rows_sorted[hi--];
    }
    for (int64_t gi = n_long; gi < groups; ++gi)
      for (int s2 = 0; s2 < g && lo <= hi; ++s2) slot_row[(size_t)gi * g + s2] = rows_sorted[lo++];
  }
  std::vector<int64_t> heavy(groups, 0), order(groups);
  for (int64_t gi = 0; gi < groups; ++gi) {
    order[gi] = gi;
    for (int s2 = 0; s2 < g; ++s2) {
      const int64_t r = slot_row[(size_t)gi * g + s2];
      if (r >= 0) heavy[gi] = std::max(heavy[gi], rowptr[r + 1] - rowptr[r]);
    }
  }
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return heavy[a] > heavy[b]; });
  std::memset(out, 0, (size_t)need * sizeof(int32_t));
  // Which groups share a workgroup (its LDS and wave slots are released when the LAST of its 4 waves ends):
  //   * a group with a long row gets its own workgroup, together with the three LIGHTEST groups left -- their waves
  //     are done early and then help walking the long row (the cooperative phase uses all 4 waves);
  //   * the other workgroups take 4 consecutive groups of the degree-sorted list: waves of one block end together.
  // Workgroups are numbered heavy first (the dispatcher hands them out in order).
  int64_t n_long_groups = 0;
  if (g > 1)
    while (n_long_groups < groups && heavy[order[n_long_groups]] > CHAOREC_SPMM_LONG_T) ++n_long_groups;
  if (n_long_groups > nb) n_long_groups = nb;
  std::vector<int64_t> block_group((size_t)nb * 4, -1);
  {
    int64_t lo = n_long_groups, hi = groups - 1;
    for (int64_t b = 0; b < n_long_groups; ++b) {
      block_group[(size_t)b * 4] = order[b];
      for (int j = 1; j < 4 && hi >= lo; ++j) block_group[(size_t)b * 4 + j] = order[hi--];
    }
    for (int64_t b = n_long_groups; b < nb; ++b)
      for (int j = 0; j < 4 && lo <= hi; ++j) block_group[(size_t)b * 4 + j] = order[lo++];
  }
  for (int64_t b = 0; b < nb; ++b) {
    bool any_long = false;
    int64_t grp[4];
    for (int j = 0; j < 4; ++j) {
      grp[j] = block_group[(size_t)b * 4 + j];
      if (grp[j] >= 0 && g > 1 && heavy[grp[j]] > CHAOREC_SPMM_LONG_T) any_long = true;
    }
    for (int j = 0; j < 4; ++j) {
      for (int s = 0; s < g; ++s) {
        int32_t *d = out + (((size_t)(4 * b + j)) * g + s) * kDescDwords;
        const int64_t r = grp[j] >= 0 ? slot_row[(size_t)grp[j] * g + s] : -1;
        if (r < 0 || r >= n_rows) {
          d[0] = -1;
          d[1] = any_long ? (int32_t)0x80000000u : 0;
          continue;
        }
        const int64_t e0 = rowptr[r], deg = rowptr[r + 1] - e0;
        if (deg > 0x7fffffffLL) return fail(CHAOREC_E_INVALID, "build_schedule: row %lld too long", (long long)r);
        d[0] = (int32_t)r;
        d[1] = (int32_t)deg | (any_long ? (int32_t)0x80000000u : 0);
        d[2] = (int32_t)(e0 & 0xffffffffll);
        d[3] = (int32_t)(e0 >> 32);
        for (int q = 0; q < kInline && q < deg; ++q) {
          d[4 + q] = col[e0 + q];
          std::memcpy(&d[4 + kInline + q], &val[e0 + q], 4);
        }
      }
    }
  }
  return CHAOREC_OK;
}
