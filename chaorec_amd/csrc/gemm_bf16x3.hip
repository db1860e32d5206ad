// C[M,N] = A[M,K] . B[N,K]^T (+ bias, + leaky-relu), and C[M,N] = A[K,M]^T . B[K,N], on the bf16 MFMA pipe at fp32-grade
// accuracy.
//
// Replaces the FORWARD of the modality projections and of every other nn.Linear on the path (y = x W^T + b:
// Model/FREEDOM.py:59-60,209,212 image_trs / text_trs over the 4096- / 384-wide feature tables;
// Model/MMGCN.py:40,97,102-131; BasicGCN.py:40) -- north_star's "modality-feature projection (v_feat/t_feat x W) on
// MFMA bf16".
//
// A plain bf16 GEMM has 8 significand bits per operand; the reference computes these products in fp32.  So every
// fp32 operand is split EXACTLY into three bf16 planes, x = h + m + l (h = rne_bf16(x), m = rne_bf16(x - h),
// l = x - h - m: 8 + 8 + 8 bits cover fp32's 24), and the product a.b is accumulated from six of the nine plane
// products on v_mfma_f32_32x32x16_bf16 (fp32 accumulate; a bf16 x bf16 product is exact in fp32):
//     ah.bh + ah.bm + am.bh + ah.bl + al.bh + am.bm          (dropped: am.bl, al.bm, al.bl <= 3 * 2^-24 |a||b|)
// -> |error| <= ~4e-7 * sum_k |a_k||b_k| including the fp32 accumulation order: the accuracy class of an fp32 GEMM (it
// is NOT bit-identical to oracle_gemm_f32's k-ascending chain; chaorec_gemm_f32 stays for that).  Six bf16 MFMAs do
// a 32x32x16 block in 192 cycles; the f32 MFMA (v_mfma_f32_32x32x2_f32) needs 512: 2.7x the matrix rate, which turns
// the skinny projections (N = 64: 32 flop per byte of A) from MFMA-bound into HBM-bound.
//
// Tile: 128 x 64 x 32 per workgroup, 4 waves, wave w owns rows [32w, 32w + 32) x 64 columns (2 accumulators).
// Global -> registers (float4 along k, both operands are k-contiguous in memory: no transposition anywhere) -> split
// -> LDS as three bf16 planes per operand ([plane][row][32 k + 8 pad] : a lane's MFMA fragment, 8 consecutive k of
// one row, is one 16-B LDS read) -> MFMA.  The next k-tile is fetched while the current one is multiplied.
// K is cut into slabs over blockIdx.z when the output has few tiles (the same fixed-order slab sum as
// chaorec_gemm_f32: gemm_reduce_slabs_kernel).
#include "common.h"
#include <cstdlib>

namespace chaorec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef CHAOREC_X3_WAVE2X2
#define CHAOREC_X3_WAVE2X2 1
#endif
// 128-wide tiles: waves as 2 x 2 of 64 x 64 instead of 4 x 1 of 32 x 128 (12 LDS fragment reads per 24 MFMAs instead of 15:
// 768 x 768 x 60 499 TN 683 -> 658 us, NT 552 -> 546 us; same bits)
constexpr bool kWave2x2 = CHAOREC_X3_WAVE2X2 != 0;
#ifndef CHAOREC_X3_PF2
#define CHAOREC_X3_PF2 0
#endif
constexpr bool kPrefetch2 = CHAOREC_X3_PF2 != 0;    // 128-wide tiles: two k-tiles of operands in flight (see the main loop; measured SLOWER)
constexpr int XBM = 128, XBK = 32, XPAD = 8;   // (row stride 40 bf16 = 80 B: 16-B aligned, banks skewed)
// N tile: 64 (the skinny projections, N = 64: one tile covers the output's width) or 128 (wide outputs: every staged and
// split A element then feeds twice the MFMAs -- the three-plane split is VALU work of the same order as the MFMA time)

union Frag8 {
  uint4 u;
  bf16x8 v;
};

// x = h + m + l exactly (finite x), three bf16 bit patterns per element, for TWO elements at a time, packed (first element in
// the low half) as the LDS planes want them.
//   CHAOREC_X3_SPLIT 0 (rounds 2-5): h = rne_bf16(x), m = rne_bf16(x - h), l = x - h - m.  19 VALU per element with the packing: the split
//     cost more than the six MFMAs it feeds (9.3 VALU per MFMA on MMGCN's weight-gradient product, MFMA pipe busy 0.35).
//   CHAOREC_X3_SPLIT 1 (round 6, default): the three BYTES of the significand -- h = x & 0xFFFF0000, m = (x - h) & 0xFFFF0000,
//     l = x - h - m; both subtractions exact, every plane of x's sign -- and one v_perm_b32 per packed pair: 4 + 1.5 VALU per
//     element.  |m| < 2^-7 |x|, |l| < 2^-15 |x|: the three dropped products (m l, l m, l l) stay below 2^-21 |a||b| each, inside the
//     1e-6 of sum |a||b| the products promise (a rounded h halves |m|: 2^-23).  Round 5 built this form and shelved it because
//     MMGCN's runs "stopped being reproducible" with it -- that was the BPR backward's atomic adds all along (round 6,
//     profiles/r06_stream_bisect.txt), which a coarser split only made more visible.
//   (v_cvt_pk_bf16_f32, gfx950's one-instruction pair rounding, was tried for form 0 in round 6: correct, and the products ran
//    THREE TIMES slower -- 384 -> 1 228 us on the weight gradient; profiles/r06_f_cvt_pk_split.txt.  Not used.)
#ifndef CHAOREC_X3_SPLIT
#define CHAOREC_X3_SPLIT 1
#endif
__device__ __forceinline__ uint32_t rne_bf16_bits(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b + 0x7FFFu + ((b >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ uint32_t pack_hi16(float lo, float hi) {      // (lo >> 16) | (hi & 0xFFFF0000): one v_perm_b32
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ void split3x2(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l) {
#if CHAOREC_X3_SPLIT
  const float h0 = __uint_as_float(__float_as_uint(x0) & 0xFFFF0000u), h1 = __uint_as_float(__float_as_uint(x1) & 0xFFFF0000u);
  const float r0 = x0 - h0, r1 = x1 - h1;
  const float m0 = __uint_as_float(__float_as_uint(r0) & 0xFFFF0000u), m1 = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
  const float s0 = r0 - m0, s1 = r1 - m1;
  h = pack_hi16(x0, x1);
  m = pack_hi16(r0, r1);
  l = pack_hi16(s0, s1);
#else
  const uint32_t h0 = rne_bf16_bits(x0), h1 = rne_bf16_bits(x1);
  const float r0 = x0 - __uint_as_float(h0 << 16), r1 = x1 - __uint_as_float(h1 << 16);
  const uint32_t m0 = rne_bf16_bits(r0), m1 = rne_bf16_bits(r1);
  const float s0 = r0 - __uint_as_float(m0 << 16), s1 = r1 - __uint_as_float(m1 << 16);
  h = h0 | (h1 << 16);
  m = m0 | (m1 << 16);
  l = (__float_as_uint(s0) >> 16) | (__float_as_uint(s1) & 0xFFFF0000u);      // (at most 8 significant bits left: exact)
#endif
}

// four consecutive k of one row -> 8 B per plane
__device__ __forceinline__ void split3x4(const float4 x, uint2 &h, uint2 &m, uint2 &l) {
  split3x2(x.x, x.y, h.x, m.x, l.x);
  split3x2(x.z, x.w, h.y, m.y, l.y);
}

extern __global__ void gemm_reduce_slabs_kernel(const float *__restrict__ slabs, int splits, float *__restrict__ C,
                                                const float *__restrict__ bias, int64_t M, int64_t N, int64_t ldc,
                                                int accumulate, int act, float *__restrict__ C2, int64_t ldc2,
                                                int64_t row_split);

// Optional SECOND segments of the operands: a virtual concatenation without the copy, so that the two Linears MMGCN applies
// to the same x (Model/MMGCN.py:102-131: conv.lin and linear_layer) run as ONE product each way --
//   forward   x [Wc; Wl]^T       B's memory rows n >= b_split come from B2; columns n >= c_split of the result go to C2
//                                (with bias2 / act2)
//   gx        [gc | gu] [Wc; Wl] A's memory columns k >= a_split come from A2, B's memory rows k >= b_split from B2
//   dW        [gc | gu]^T x      A's memory columns m >= a_split come from A2; rows m >= c_split of the result go to C2
// A is always split along its memory COLUMN index, B along its memory ROW index, C along columns (c_rows = 0) or rows.
// Splits are multiples of 4 and the second pointers 16-byte aligned: a float4 never straddles a boundary.
struct X3Seg {
  const float *A2;
  const float *B2;
  float *C2;
  const float *bias2;
  int64_t lda2, ldb2, ldc2;
  int64_t a_split, b_split, c_split;     // INT64_MAX: no second segment
  int act2, c_rows;
  int xcd;                               // workgroup ids are regrouped per XCD (see the kernel); every launch carries it
};
constexpr int64_t kNoSplit = INT64_MAX;
// CHAOREC_X3_XCD=0: tiles in dispatch order (rounds 2-5 and the first half of round 6), for A/B runs
static int x3_xcd() {                    // (read per call: the test compares the two orders in one process)
  const char *e = std::getenv("CHAOREC_X3_XCD");
  return (e && e[0] == '0') ? 0 : 1;
}
static X3Seg no_seg() {
  X3Seg g;
  g.xcd = x3_xcd();
  g.A2 = g.B2 = g.bias2 = nullptr;
  g.C2 = nullptr;
  g.lda2 = g.ldb2 = g.ldc2 = 0;
  g.a_split = g.b_split = g.c_split = kNoSplit;
  g.act2 = g.c_rows = 0;
  return g;
}

// NT (TA = TB = false): A [M, K], B [N, K] (k contiguous in both: the forward and, through W^T, the input gradient of a Linear).
// TN (TA = TB = true): A [K, M], B [K, N] (the reduction runs over the ROWS of both operands: the weight gradient
//             dW = gy^T x, Model/MMGCN.py's Linears over all graph nodes).  A thread then fetches a 4 (m) x 4 (k)
//             block of A -- four float4 along m, one per k -- and a 4 (n) x 2 (k) block of B, transposes them in
//             registers (a choice of components, no instructions) and writes the same k-major bf16 planes to LDS.
// TA / TB: the operand is k-MAJOR in memory (A [K, M] / B [K, N]).  (false, false) = NT, (true, true) = TN, (false, true)
// = NN: C = A[M,K] . B[K,N] -- a Linear's input gradient gy . W with W as it lies in memory (W^T as a small copy in
// front of every such product was ~20 launches of 4.6 us per MMGCN step).
template <bool TA, bool TB, int XBN, bool SEG = false>   // SEG: the operands have second segments (X3Seg)
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                             float *__restrict__ C, const float *__restrict__ bias,
                                                             int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                                                             int64_t ldc, int act, int64_t k_per_split,
                                                             float *__restrict__ slabs, int accumulate, const X3Seg seg_in) {
  // (without SEG every split is the constant "none": the segment selects fold away)
  X3Seg seg = seg_in;
  if constexpr (!SEG) {
    seg.A2 = seg.B2 = seg.bias2 = nullptr;
    seg.C2 = nullptr;
    seg.a_split = seg.b_split = seg.c_split = kNoSplit;
    seg.c_rows = 0;
  }
  __shared__ uint16_t As[3][XBM][XBK + XPAD];
  __shared__ uint16_t Bs[3][XBN][XBK + XPAD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  // Tile of this workgroup.  Workgroups are dispatched in the order x + gridDim.x * y; the side with FEWER tiles varies fastest,
  // so that the tiles which share a panel of the long operand run next to each other: with the row tile fastest, the six
  // column tiles of one 128-row block of MMGCN's [60 499, 772] operand were 473 workgroups -- a whole round -- apart and the
  // block was fetched from HBM six times.
  //
  // XCD grouping (round 6).  The dispatcher deals workgroups round-robin over the eight XCDs (observed, not promised: a wrong
  // guess costs speed only), each with its own 4 MiB L2: neighbours in dispatch order never share an L2, and MMGCN's
  // [60 499, 768] operand came over the fabric once per column tile, a weight gradient's k-slab once per XCD.  Dispatch id d
  // therefore works on item (d % 8) * (n / 8) + d / 8 (the bijective form for n % 8 != 0) of the list ordered slab-major,
  // then by the side with more tiles: an XCD gets a CONTIGUOUS run of that list -- whole row panels with all their column
  // tiles, whole k-slabs.
  const int64_t gxy = (int64_t)gridDim.x * (int64_t)gridDim.y;
  int64_t wid = (int64_t)blockIdx.x + (int64_t)gridDim.x * ((int64_t)blockIdx.y + (int64_t)gridDim.y * (int64_t)blockIdx.z);
  if (seg.xcd) {
    const int64_t nwg = gxy * (int64_t)gridDim.z, q = nwg >> 3, rr = nwg & 7, x = wid & 7;
    wid = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (wid >> 3);
  }
  const int64_t zi = wid / gxy, lin = wid - zi * gxy;
  const bool n_fast = gridDim.y <= gridDim.x;
  const int64_t m0 = (n_fast ? lin / gridDim.y : lin % gridDim.x) * XBM;
  const int64_t n0 = (n_fast ? lin % gridDim.y : lin / gridDim.x) * XBN;
  const int64_t kb = zi * k_per_split, ke = min(K, kb + k_per_split);

  constexpr int NJ = XBN / 32;      // accumulators per wave (32 x 32 each)
  constexpr int NB = XBN / 32;      // B float4 per thread per k-tile (NT) / B blocks per thread (TN)
  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  // this thread's float4s of a k-tile: A 128 x 32 = 1024 float4 (4 per thread), B 64 x 32 = 512 (2 per thread)
  float4 ra[4], rb[NB];
  const bool a_vec = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                     (!seg.A2 || (((seg.lda2 | seg.a_split) & 3) == 0 && (reinterpret_cast<uintptr_t>(seg.A2) & 15) == 0));
  const bool b_vec = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) &&
                     (!seg.B2 || ((seg.ldb2 & 3) == 0 && (reinterpret_cast<uintptr_t>(seg.B2) & 15) == 0));
  // (row, k): indices of the VIRTUAL operand; col_split: its memory columns (k) past the split live in base2;
  // row_split: its memory rows past the split live in base2
  auto load4 = [&](const float *base, int64_t ld, int64_t row, int64_t n_rows, int64_t k, bool vec, const float *base2,
                   int64_t ld2, int64_t col_split, int64_t row_split) -> float4 {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < n_rows && k < ke) {
      const float *src = base + row * ld + k;
      if (k >= col_split) src = base2 + row * ld2 + (k - col_split);
      if (row >= row_split) src = base2 + (row - row_split) * ld2 + k;
      if (vec && k + 3 < ke) {
        x = *reinterpret_cast<const float4 *>(src);
      } else {
        x.x = src[0];
        if (k + 1 < ke) x.y = src[1];
        if (k + 2 < ke) x.z = src[2];
        if (k + 3 < ke) x.w = src[3];
      }
    }
    return x;
  };
  // TN: float4 along the m / n dimension of row k (zero past the matrix or past this slab's k range)
  auto load4t = [&](const float *base, int64_t ld, int64_t k, int64_t col, int64_t n_cols, bool vec, const float *base2,
                    int64_t ld2, int64_t col_split, int64_t row_split) -> float4 {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < ke && col < n_cols) {
      const float *src = base + k * ld + col;
      if (col >= col_split) src = base2 + k * ld2 + (col - col_split);
      if (k >= row_split) src = base2 + (k - row_split) * ld2 + col;
      if (vec && col + 3 < n_cols) {
        x = *reinterpret_cast<const float4 *>(src);
      } else {
        x.x = src[0];
        if (col + 1 < n_cols) x.y = src[1];
        if (col + 2 < n_cols) x.z = src[2];
        if (col + 3 < n_cols) x.w = src[3];
      }
    }
    return x;
  };
  const int a_mq = t & 31, a_kq = t >> 5;         // TN: A block = rows 4 a_mq .. +3, k 4 a_kq .. +3
  const int b_nq = t & 15, b_kp = t >> 4;         // TN: B block = rows 4 b_nq .. +3, k 2 b_kp, 2 b_kp + 1
  auto fetch_into = [&](int64_t k0, float4 (&ra)[4], float4 (&rb)[NB]) __attribute__((always_inline)) {
    if constexpr (TA) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        ra[j] = load4t(A, lda, k0 + 4 * a_kq + j, m0 + 4 * a_mq, M, a_vec, seg.A2, seg.lda2, seg.a_split, kNoSplit);
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int v = t + p * 256;                  // float4 index: row = v / 8, k4 = (v % 8) * 4
        ra[p] = load4(A, lda, m0 + (v >> 3), M, k0 + ((v & 7) << 2), a_vec, seg.A2, seg.lda2, seg.a_split, kNoSplit);
      }
    }
    if constexpr (TB) {
#pragma unroll
      for (int bb = 0; bb < NB / 2; ++bb)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          rb[2 * bb + j] = load4t(B, ldb, k0 + 2 * b_kp + j, n0 + 4 * (b_nq + 16 * bb), N, b_vec, seg.B2, seg.ldb2, kNoSplit,
                                  seg.b_split);
    } else {
#pragma unroll
      for (int p = 0; p < NB; ++p) {
        const int v = t + p * 256;
        rb[p] = load4(B, ldb, n0 + (v >> 3), N, k0 + ((v & 7) << 2), b_vec, seg.B2, seg.ldb2, kNoSplit, seg.b_split);
      }
    }
  };
  auto fetch = [&](int64_t k0) __attribute__((always_inline)) { fetch_into(k0, ra, rb); };
  auto stash_from = [&](const float4 (&ra)[4], const float4 (&rb)[NB]) __attribute__((always_inline)) {
    if constexpr (TA) {
      const float ax[4][4] = {{ra[0].x, ra[1].x, ra[2].x, ra[3].x}, {ra[0].y, ra[1].y, ra[2].y, ra[3].y},
                              {ra[0].z, ra[1].z, ra[2].z, ra[3].z}, {ra[0].w, ra[1].w, ra[2].w, ra[3].w}};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2 hh, mm, ll;
        split3x4(make_float4(ax[i][0], ax[i][1], ax[i][2], ax[i][3]), hh, mm, ll);
        // (tile row 4 mq + i is kept at LDS row 32 i + mq: a wave's 32 m-quads then write 32 consecutive LDS rows --
        // with the natural order they are 4 rows = 320 B apart and land in two banks)
        *reinterpret_cast<uint2 *>(&As[0][32 * i + a_mq][4 * a_kq]) = hh;
        *reinterpret_cast<uint2 *>(&As[1][32 * i + a_mq][4 * a_kq]) = mm;
        *reinterpret_cast<uint2 *>(&As[2][32 * i + a_mq][4 * a_kq]) = ll;
      }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int v = t + p * 256;
        uint2 hh, mm, ll;
        split3x4(ra[p], hh, mm, ll);
        const int row = v >> 3, k4 = (v & 7) << 2;
        *reinterpret_cast<uint2 *>(&As[0][row][k4]) = hh;
        *reinterpret_cast<uint2 *>(&As[1][row][k4]) = mm;
        *reinterpret_cast<uint2 *>(&As[2][row][k4]) = ll;
      }
    }
    if constexpr (TB) {
#pragma unroll
      for (int bb = 0; bb < NB / 2; ++bb) {
        const float4 r0 = rb[2 * bb], r1 = rb[2 * bb + 1];
        const float bx[4][2] = {{r0.x, r1.x}, {r0.y, r1.y}, {r0.z, r1.z}, {r0.w, r1.w}};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint32_t hh, mm, ll;
          split3x2(bx[i][0], bx[i][1], hh, mm, ll);
          // tile row 4 (nq + 16 bb) + i of a 64-row half is kept at LDS row 64 bb + 16 i + nq (see the A tile)
          *reinterpret_cast<uint32_t *>(&Bs[0][64 * bb + 16 * i + b_nq][2 * b_kp]) = hh;
          *reinterpret_cast<uint32_t *>(&Bs[1][64 * bb + 16 * i + b_nq][2 * b_kp]) = mm;
          *reinterpret_cast<uint32_t *>(&Bs[2][64 * bb + 16 * i + b_nq][2 * b_kp]) = ll;
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < NB; ++p) {
        const int v = t + p * 256;
        uint2 hh, mm, ll;
        split3x4(rb[p], hh, mm, ll);
        const int row = v >> 3, k4 = (v & 7) << 2;
        *reinterpret_cast<uint2 *>(&Bs[0][row][k4]) = hh;
        *reinterpret_cast<uint2 *>(&Bs[1][row][k4]) = mm;
        *reinterpret_cast<uint2 *>(&Bs[2][row][k4]) = ll;
      }
    }
  };

  // the MFMAs of the k-tile that lies in LDS
  auto multiply = [&]() __attribute__((always_inline)) {
    if constexpr (XBN == 128 && kWave2x2) {
      // 2 x 2 waves of 64 x 64 (accumulator j = 2 i + jj: rows 64 (wave / 2) + 32 i, columns 64 (wave % 2) + 32 jj): two A
      // and two B fragments per plane feed the four accumulators -- 12 LDS reads per 24 MFMAs instead of 15
#pragma unroll
      for (int ks = 0; ks < XBK; ks += 16) {
        Frag8 a2[2][3], b2[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int R = 64 * (wave >> 1) + 32 * i + r, Rn = 64 * (wave & 1) + 32 * i + r;
            const int arow = TA ? 32 * (R & 3) + (R >> 2) : R;
            const int brow = TB ? 64 * (Rn >> 6) + 16 * (Rn & 3) + ((Rn & 63) >> 2) : Rn;
            a2[i][pl].u = *reinterpret_cast<const uint4 *>(&As[pl][arow][ks + 8 * h]);
            b2[i][pl].u = *reinterpret_cast<const uint4 *>(&Bs[pl][brow][ks + 8 * h]);
          }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * i + jj;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i][1].v, b2[jj][1].v, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i][2].v, b2[jj][0].v, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i][0].v, b2[jj][2].v, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i][1].v, b2[jj][0].v, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i][0].v, b2[jj][1].v, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i][0].v, b2[jj][0].v, acc[j], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < XBK; ks += 16) {
        Frag8 a[3], b[NJ][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          // LDS row of tile row (32 wave + r) resp. (32 j + r): the identity, or the TN stash's permutation
          const int arow = TA ? (r & 3) * 32 + wave * 8 + (r >> 2) : wave * 32 + r;
          a[pl].u = *reinterpret_cast<const uint4 *>(&As[pl][arow][ks + 8 * h]);
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            // (TN: tile row 32 j + r lies in the 64-row half j / 2, at 64 (j / 2) + 16 (r & 3) + 8 (j & 1) + (r >> 2))
            const int brow = TB ? 64 * (j >> 1) + (r & 3) * 16 + (j & 1) * 8 + (r >> 2) : j * 32 + r;
            b[j][pl].u = *reinterpret_cast<const uint4 *>(&Bs[pl][brow][ks + 8 * h]);
          }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          // smallest products first (0 = h, 1 = m, 2 = l)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[j][1].v, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[j][0].v, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[j][2].v, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[j][0].v, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[j][1].v, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[j][0].v, acc[j], 0, 0, 0);
        }
      }
    }
  };
  if constexpr (kPrefetch2 && XBN == 128) {
    // TWO k-tiles in flight in registers (round 6 experiment, off: -DCHAOREC_X3_PF2=1 builds it).  The idea: a tile's loads
    // are issued under the previous tile's MFMAs, ~0.65 us of them, and an HBM round trip under load is longer.  Measured
    // (tools/gemm_wide_bench.py, 252 VGPRs instead of 220, no spills, still two waves per SIMD): 60 499 x 772 x 768 NT 604 ->
    // 696 us, TN 564 -> 694 us, MMGCN's step 3.57 -> 3.90 ms.  The loads' latency is not what the k-tile waits for.
    float4 ra1[4], rb1[NB];
    if (kb < ke) fetch_into(kb, ra, rb);
    if (kb + XBK < ke) fetch_into(kb + XBK, ra1, rb1);
    for (int64_t k0 = kb; k0 < ke; k0 += 2 * XBK) {
      __syncthreads();                            // the previous tile's fragment reads are done
      stash_from(ra, rb);
      __syncthreads();
      if (k0 + 2 * XBK < ke) fetch_into(k0 + 2 * XBK, ra, rb);
      multiply();
      if (k0 + XBK < ke) {                        // (block-uniform)
        __syncthreads();
        stash_from(ra1, rb1);
        __syncthreads();
        if (k0 + 3 * XBK < ke) fetch_into(k0 + 3 * XBK, ra1, rb1);
        multiply();
      }
    }
  } else {
    if (kb < ke) fetch(kb);
    for (int64_t k0 = kb; k0 < ke; k0 += XBK) {
      __syncthreads();                              // the previous tile's fragment reads are done
      stash_from(ra, rb);
      __syncthreads();
      if (k0 + XBK < ke) fetch(k0 + XBK);           // next tile's loads fly under this tile's MFMAs
      multiply();
    }
  }

  // C[m][n]: accumulator register q of lane (r, h) is row 32*wave + (q & 3) + 8 * (q >> 2) + 4h, column 32 j + r
  const bool to_slab = slabs != nullptr;
  float *dst = to_slab ? slabs + (size_t)zi * (size_t)M * (size_t)N : C;
  const int64_t ldd = to_slab ? N : ldc;
  constexpr bool W22 = XBN == 128 && kWave2x2;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int64_t n = n0 + (W22 ? 64 * (wave & 1) + 32 * (j & 1) : j * 32) + r;
    if (n >= N) continue;
    const bool col2 = !seg.c_rows && n >= seg.c_split;            // this column belongs to the second output
    float bv = 0.f;
    if (!to_slab) bv = col2 ? (seg.bias2 ? seg.bias2[n - seg.c_split] : 0.f) : (bias ? bias[n] : 0.f);
    const int a_here = col2 ? seg.act2 : act;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int64_t m = m0 + (W22 ? 64 * (wave >> 1) + 32 * (j >> 1) : wave * 32) + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (m >= M) continue;
      float v = acc[j][q];
      float *out = dst + m * ldd + n;
      if (!to_slab) {
        if (col2) out = seg.C2 + m * seg.ldc2 + (n - seg.c_split);
        if (seg.c_rows && m >= seg.c_split) out = seg.C2 + (m - seg.c_split) * seg.ldc2 + n;
        v = v + bv;
        if (accumulate) v = *out + v;
        if (a_here == 1) v = v > 0.f ? v : v * 0.01f;
        if (a_here == 2) v = v > 0.f ? v : v * 0.2f;
      }
      *out = v;
    }
  }
}

struct XPlan {
  int splits;
  int64_t k_per_split;
};

// the N tile: 128 once the output is at least two 64-wide tiles wide
static int pick_bn(int64_t N) { return N >= 128 ? 128 : 64; }

static XPlan plan_x(int64_t M, int64_t N, int64_t K) {
  XPlan p;
  const int XBN = pick_bn(N);
  const int64_t tiles = ((M + XBM - 1) / XBM) * ((N + XBN - 1) / XBN);
  int64_t s = 1;
  // few output tiles, long reduction: fill the chip by cutting K.  (Not at 400+ tiles: [60499, 768] x [768, 64] has 473 --
  // all resident at once --, and three slabs of 15.5 MB each cost a 55 us second pass on top of their own writes.)
  if (tiles < 400 && K >= 512) {
    s = (1024 + tiles - 1) / tiles;
    if (s > K / 256) s = K / 256;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
  }
  int64_t per = (K + s - 1) / s;
  per = (per + XBK - 1) / XBK * XBK;
  p.k_per_split = per;
  p.splits = (int)((K + per - 1) / per);
  if (p.splits < 1) p.splits = 1;
  return p;
}

// weight gradients: outputs of a few tiles, reductions over all graph nodes -- up to 256 slabs of at least 128 k.
// The slab count is chosen by ROUNDS: a CU holds two 128-wide (three 64-wide) workgroups, the chip 512 (768) at a time, and a
// launch takes ceil(workgroups / that) rounds of K / slabs each.  Round 5's rule (ceil(1024 / tiles) slabs) gave MMGCN's
// largest product -- 42 tiles x 25 slabs = 1 050 workgroups -- a THIRD round for its last 26 workgroups: 732 us where two
// rounds take ~500.  Among the slab counts with the fewest k-steps on the critical path the smallest wins (fewer slab bytes).
static XPlan plan_x_tn(int64_t M, int64_t N, int64_t K) {
  XPlan p;
  const int XBN = pick_bn(N);
  const int64_t tiles = ((M + XBM - 1) / XBM) * ((N + XBN - 1) / XBN);
  int64_t s = 1;
  if (tiles < 512 && K >= 512) {
    const int64_t slots = XBN == 128 ? 512 : 768;
    int64_t smax = K / 128;
    if (smax > 256) smax = 256;
    if (smax < 1) smax = 1;
    int64_t best = INT64_MAX;
    for (int64_t c = 1; c <= smax; ++c) {
      const int64_t rounds = (tiles * c + slots - 1) / slots;
      const int64_t ksteps = (((K + c - 1) / c) + XBK - 1) / XBK;      // k-tiles per workgroup
      const int64_t cost = rounds * (ksteps + 6);                       // (+ a workgroup's prologue / epilogue, in k-tiles)
      if (cost < best) {
        best = cost;
        s = c;
      }
    }
  }
  int64_t per = (K + s - 1) / s;
  per = (per + XBK - 1) / XBK * XBK;
  p.k_per_split = per;
  p.splits = (int)((K + per - 1) / per);
  if (p.splits < 1) p.splits = 1;
  return p;
}

}  // namespace chaorec

using namespace chaorec;

extern "C" size_t chaorec_gemm_tn_bf16x3_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const XPlan p = plan_x_tn(M, N, K);
  return p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int chaorec_gemm_tn_bf16x3(const float *A, const float *B, float *C, int64_t M, int64_t N, int64_t K,
                                      int64_t lda, int64_t ldb, int64_t ldc, void *workspace, size_t workspace_bytes,
                                      void *stream) {
  if (!A || !B || !C) return fail(CHAOREC_E_INVALID, "gemm_tn_bf16x3: NULL argument");
  if (M < 0 || N < 0 || K <= 0 || lda < M || ldb < N || ldc < N) return fail(CHAOREC_E_INVALID, "gemm_tn_bf16x3: bad size");
  if (M == 0 || N == 0) return CHAOREC_OK;
  const XPlan p = plan_x_tn(M, N, K);
  const size_t need = p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
  if (need > workspace_bytes || (need && !workspace))
    return fail(CHAOREC_E_WORKSPACE, "gemm_tn_bf16x3: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  float *slabs = p.splits > 1 ? (float *)workspace : nullptr;
  const int XBN = pick_bn(N);
  const dim3 grid((unsigned)((M + XBM - 1) / XBM), (unsigned)((N + XBN - 1) / XBN), (unsigned)p.splits);
  if (XBN == 128)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<true, true, 128>), grid, dim3(256), 0, st, A, B, C, (const float *)nullptr, M, N, K,
                       lda, ldb, ldc, 0, p.k_per_split, slabs, 0, no_seg());
  else
    hipLaunchKernelGGL((gemm_bf16x3_kernel<true, true, 64>), grid, dim3(256), 0, st, A, B, C, (const float *)nullptr, M, N, K, lda,
                       ldb, ldc, 0, p.k_per_split, slabs, 0, no_seg());
  int rc = check_launch("gemm_bf16x3_kernel<TN>");
  if (rc || p.splits == 1) return rc;
  hipLaunchKernelGGL(gemm_reduce_slabs_kernel, dim3((unsigned)((M * N + 31) / 32)), dim3(32 * reduce_lanes(p.splits, M * N)), 0, st, slabs, p.splits, C,
                     (const float *)nullptr, M, N, ldc, 0, 0, (float *)nullptr, (int64_t)0, kNoSplit);
  return check_launch("gemm_reduce_slabs_kernel");
}

extern "C" size_t chaorec_gemm_nt_bf16x3_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const XPlan p = plan_x(M, N, K);
  return p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int chaorec_gemm_nt_bf16x3(const float *A, const float *B, float *C, const float *bias, int64_t M, int64_t N,
                                      int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t act, void *workspace,
                                      size_t workspace_bytes, void *stream) {
  if (!A || !B || !C) return fail(CHAOREC_E_INVALID, "gemm_nt_bf16x3: NULL argument");
  if (M < 0 || N < 0 || K <= 0) return fail(CHAOREC_E_INVALID, "gemm_nt_bf16x3: bad size");
  if (act < 0 || act > 2) return fail(CHAOREC_E_INVALID, "gemm_nt_bf16x3: act %d", act);
  if (M == 0 || N == 0) return CHAOREC_OK;
  const XPlan p = plan_x(M, N, K);
  const size_t need = p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
  if (need > workspace_bytes || (need && !workspace))
    return fail(CHAOREC_E_WORKSPACE, "gemm_nt_bf16x3: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  float *slabs = p.splits > 1 ? (float *)workspace : nullptr;
  const int XBN = pick_bn(N);
  const dim3 grid((unsigned)((M + XBM - 1) / XBM), (unsigned)((N + XBN - 1) / XBN), (unsigned)p.splits);
  if (XBN == 128)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false, 128>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, act,
                       p.k_per_split, slabs, 0, no_seg());
  else
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false, 64>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, act,
                       p.k_per_split, slabs, 0, no_seg());
  int rc = check_launch("gemm_bf16x3_kernel<NT>");
  if (rc || p.splits == 1) return rc;
  hipLaunchKernelGGL(gemm_reduce_slabs_kernel, dim3((unsigned)((M * N + 31) / 32)), dim3(32 * reduce_lanes(p.splits, M * N)), 0, st, slabs, p.splits, C,
                     bias, M, N, ldc, 0, act, (float *)nullptr, (int64_t)0, kNoSplit);
  return check_launch("gemm_reduce_slabs_kernel");
}

extern "C" size_t chaorec_gemm_nn_bf16x3_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  return chaorec_gemm_nt_bf16x3_workspace_bytes(M, N, K);
}

// C[M,N] = A[M,K] . B[K,N]  (B k-major: a Linear's weight [out = K, in = N] as it lies in memory)
extern "C" int chaorec_gemm_nn_bf16x3(const float *A, const float *B, float *C, int64_t M, int64_t N, int64_t K,
                                      int64_t lda, int64_t ldb, int64_t ldc, int32_t accumulate, void *workspace,
                                      size_t workspace_bytes, void *stream) {
  if (!A || !B || !C) return fail(CHAOREC_E_INVALID, "gemm_nn_bf16x3: NULL argument");
  if (M < 0 || N < 0 || K <= 0 || lda < K || ldb < N || ldc < N) return fail(CHAOREC_E_INVALID, "gemm_nn_bf16x3: bad size");
  if (M == 0 || N == 0) return CHAOREC_OK;
  const XPlan p = plan_x(M, N, K);
  const size_t need = p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
  if (need > workspace_bytes || (need && !workspace))
    return fail(CHAOREC_E_WORKSPACE, "gemm_nn_bf16x3: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  float *slabs = p.splits > 1 ? (float *)workspace : nullptr;
  const int XBN = pick_bn(N);
  const dim3 grid((unsigned)((M + XBM - 1) / XBM), (unsigned)((N + XBN - 1) / XBN), (unsigned)p.splits);
  if (XBN == 128)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, true, 128>), grid, dim3(256), 0, st, A, B, C, (const float *)nullptr, M, N,
                       K, lda, ldb, ldc, 0, p.k_per_split, slabs, accumulate ? 1 : 0, no_seg());
  else
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, true, 64>), grid, dim3(256), 0, st, A, B, C, (const float *)nullptr, M, N, K,
                       lda, ldb, ldc, 0, p.k_per_split, slabs, accumulate ? 1 : 0, no_seg());
  int rc = check_launch("gemm_bf16x3_kernel<NN>");
  if (rc || p.splits == 1) return rc;
  hipLaunchKernelGGL(gemm_reduce_slabs_kernel, dim3((unsigned)((M * N + 31) / 32)), dim3(32 * reduce_lanes(p.splits, M * N)), 0, st, slabs, p.splits, C,
                     (const float *)nullptr, M, N, ldc, accumulate ? 1 : 0, 0, (float *)nullptr, (int64_t)0, kNoSplit);
  return check_launch("gemm_reduce_slabs_kernel");
}

// ---- the two Linears MMGCN applies to the same x as ONE product each way (X3Seg) ---------------------------------------
static bool seg_ok(const void *p, int64_t ld, int64_t split) {
  return p && (ld & 3) == 0 && (split & 3) == 0 && split > 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

// [C1 | C2] = act1/act2(A [B1; B2]^T + [bias1 | bias2]):  B1 [N1, K], B2 [N2, K], C1 [M, N1], C2 [M, N2]
extern "C" int chaorec_gemm_nt_bf16x3_dual(const float *A, const float *B1, const float *B2, float *C1, float *C2,
                                           const float *bias1, const float *bias2, int64_t M, int64_t N1, int64_t N2,
                                           int64_t K, int64_t lda, int64_t ldb1, int64_t ldb2, int64_t ldc1, int64_t ldc2,
                                           int32_t act1, int32_t act2, void *stream) {
  if (!A || !B1 || !C1 || !C2) return fail(CHAOREC_E_INVALID, "gemm_nt_bf16x3_dual: NULL argument");
  if (M < 0 || N1 <= 0 || N2 <= 0 || K <= 0 || act1 < 0 || act1 > 2 || act2 < 0 || act2 > 2)
    return fail(CHAOREC_E_INVALID, "gemm_nt_bf16x3_dual: bad size / act");
  if (!seg_ok(B2, ldb2, N1)) return fail(CHAOREC_E_INVALID, "gemm_nt_bf16x3_dual: N1, ldb2 multiples of 4, B2 16-byte aligned");
  if (M == 0) return CHAOREC_OK;
  const int64_t N = N1 + N2;
  const XPlan p = plan_x(M, N, K);
  if (p.splits != 1) return fail(CHAOREC_E_INVALID, "gemm_nt_bf16x3_dual: this shape wants split-K (call the two products separately)");
  X3Seg g = no_seg();
  g.B2 = B2, g.ldb2 = ldb2, g.b_split = N1;
  g.C2 = C2, g.ldc2 = ldc2, g.c_split = N1, g.bias2 = bias2, g.act2 = act2;
  const int XBN = pick_bn(N);
  const dim3 grid((unsigned)((M + XBM - 1) / XBM), (unsigned)((N + XBN - 1) / XBN), 1);
  hipStream_t st = (hipStream_t)stream;
  if (XBN == 128)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false, 128, true>), grid, dim3(256), 0, st, A, B1, C1, bias1, M, N, K, lda, ldb1,
                       ldc1, act1, p.k_per_split, (float *)nullptr, 0, g);
  else
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false, 64, true>), grid, dim3(256), 0, st, A, B1, C1, bias1, M, N, K, lda, ldb1,
                       ldc1, act1, p.k_per_split, (float *)nullptr, 0, g);
  return check_launch("gemm_bf16x3_kernel<NT, dual>");
}

// C = [A1 | A2] [B1; B2]:  A1 [M, K1], A2 [M, K2], B1 [K1, N], B2 [K2, N]  (gx = gc Wc + gu Wl in one accumulation)
extern "C" size_t chaorec_gemm_nn_bf16x3_dual_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  return chaorec_gemm_nt_bf16x3_workspace_bytes(M, N, K);
}
extern "C" int chaorec_gemm_nn_bf16x3_dual(const float *A1, const float *A2, const float *B1, const float *B2, float *C,
                                           int64_t M, int64_t N, int64_t K1, int64_t K2, int64_t lda1, int64_t lda2,
                                           int64_t ldb1, int64_t ldb2, int64_t ldc, void *workspace, size_t workspace_bytes,
                                           void *stream) {
  if (!A1 || !B1 || !C) return fail(CHAOREC_E_INVALID, "gemm_nn_bf16x3_dual: NULL argument");
  if (M < 0 || N <= 0 || K1 <= 0 || K2 <= 0) return fail(CHAOREC_E_INVALID, "gemm_nn_bf16x3_dual: bad size");
  if (!seg_ok(A2, lda2, K1) || !seg_ok(B2, ldb2, K1))
    return fail(CHAOREC_E_INVALID, "gemm_nn_bf16x3_dual: K1, lda2, ldb2 multiples of 4, A2 / B2 16-byte aligned");
  if (M == 0) return CHAOREC_OK;
  const int64_t K = K1 + K2;
  const XPlan p = plan_x(M, N, K);
  const size_t need = p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
  if (need > workspace_bytes || (need && !workspace))
    return fail(CHAOREC_E_WORKSPACE, "gemm_nn_bf16x3_dual: workspace %zu < %zu", workspace_bytes, need);
  X3Seg g = no_seg();
  g.A2 = A2, g.lda2 = lda2, g.a_split = K1;
  g.B2 = B2, g.ldb2 = ldb2, g.b_split = K1;
  float *slabs = p.splits > 1 ? (float *)workspace : nullptr;
  const int XBN = pick_bn(N);
  const dim3 grid((unsigned)((M + XBM - 1) / XBM), (unsigned)((N + XBN - 1) / XBN), (unsigned)p.splits);
  hipStream_t st = (hipStream_t)stream;
  if (XBN == 128)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, true, 128, true>), grid, dim3(256), 0, st, A1, B1, C, (const float *)nullptr, M, N,
                       K, lda1, ldb1, ldc, 0, p.k_per_split, slabs, 0, g);
  else
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, true, 64, true>), grid, dim3(256), 0, st, A1, B1, C, (const float *)nullptr, M, N,
                       K, lda1, ldb1, ldc, 0, p.k_per_split, slabs, 0, g);
  int rc = check_launch("gemm_bf16x3_kernel<NN, dual>");
  if (rc || p.splits == 1) return rc;
  hipLaunchKernelGGL(gemm_reduce_slabs_kernel, dim3((unsigned)((M * N + 31) / 32)), dim3(32 * reduce_lanes(p.splits, M * N)), 0, st,
                     slabs, p.splits, C, (const float *)nullptr, M, N, ldc, 0, 0, (float *)nullptr, (int64_t)0, kNoSplit);
  return check_launch("gemm_reduce_slabs_kernel");
}

// [C1; C2] = [A1 | A2]^T B:  A1 [K, M1], A2 [K, M2], B [K, N], C1 [M1, N], C2 [M2, N]  (both weight gradients of the pair)
extern "C" size_t chaorec_gemm_tn_bf16x3_dual_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const XPlan p = plan_x_tn(M, N, K);            // (the dual product always runs the plain form: its own slab count)
  return p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}
extern "C" int chaorec_gemm_tn_bf16x3_dual(const float *A1, const float *A2, const float *B, float *C1, float *C2, int64_t M1,
                                           int64_t M2, int64_t N, int64_t K, int64_t lda1, int64_t lda2, int64_t ldb,
                                           int64_t ldc1, int64_t ldc2, void *workspace, size_t workspace_bytes, void *stream) {
  if (!A1 || !B || !C1 || !C2) return fail(CHAOREC_E_INVALID, "gemm_tn_bf16x3_dual: NULL argument");
  if (M1 <= 0 || M2 <= 0 || N <= 0 || K <= 0) return fail(CHAOREC_E_INVALID, "gemm_tn_bf16x3_dual: bad size");
  if (!seg_ok(A2, lda2, M1)) return fail(CHAOREC_E_INVALID, "gemm_tn_bf16x3_dual: M1, lda2 multiples of 4, A2 16-byte aligned");
  const int64_t M = M1 + M2;
  const XPlan p = plan_x_tn(M, N, K);
  const size_t need = p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
  if (need > workspace_bytes || (need && !workspace))
    return fail(CHAOREC_E_WORKSPACE, "gemm_tn_bf16x3_dual: workspace %zu < %zu", workspace_bytes, need);
  X3Seg g = no_seg();
  g.A2 = A2, g.lda2 = lda2, g.a_split = M1;
  g.C2 = C2, g.ldc2 = ldc2, g.c_split = M1, g.c_rows = 1;
  float *slabs = p.splits > 1 ? (float *)workspace : nullptr;
  const int XBN = pick_bn(N);
  const dim3 grid((unsigned)((M + XBM - 1) / XBM), (unsigned)((N + XBN - 1) / XBN), (unsigned)p.splits);
  hipStream_t st = (hipStream_t)stream;
  if (XBN == 128)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<true, true, 128, true>), grid, dim3(256), 0, st, A1, B, C1, (const float *)nullptr, M, N, K,
                       lda1, ldb, ldc1, 0, p.k_per_split, slabs, 0, g);
  else
    hipLaunchKernelGGL((gemm_bf16x3_kernel<true, true, 64, true>), grid, dim3(256), 0, st, A1, B, C1, (const float *)nullptr, M, N, K,
                       lda1, ldb, ldc1, 0, p.k_per_split, slabs, 0, g);
  int rc = check_launch("gemm_bf16x3_kernel<TN, dual>");
  if (rc || p.splits == 1) return rc;
  hipLaunchKernelGGL(gemm_reduce_slabs_kernel, dim3((unsigned)((M * N + 31) / 32)), dim3(32 * reduce_lanes(p.splits, M * N)), 0, st,
                     slabs, p.splits, C1, (const float *)nullptr, M, N, ldc1, 0, 0, C2, ldc2, M1);
  return check_launch("gemm_reduce_slabs_kernel");
}
