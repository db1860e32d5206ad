// Sustained issue rate of v_mfma_f32_32x32x16_bf16 on gfx950 against waves per SIMD, independent accumulator chains
// per wave and the chain length before the accumulators are read (a sign-bit collect like the scoring sweep's).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int CH, int LEN, bool COLLECT>
__global__ __launch_bounds__(64) void mfma_loop(const uint4 *__restrict__ src, int iters, unsigned *__restrict__ out) {
  union U { uint4 u; bf16x8 v; };
  U a[LEN], b[CH];
#pragma unroll
  for (int q = 0; q < LEN; ++q) a[q].u = src[(q * 64 + threadIdx.x) & 1023];
#pragma unroll
  for (int c = 0; c < CH; ++c) b[c].u = src[(512 + c * 64 + threadIdx.x) & 1023];
  unsigned keep = 0;
  for (int it = 0; it < iters; ++it) {
    f32x16 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
#pragma unroll
    for (int q = 0; q < LEN; ++q)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q].v, b[c].v, acc[c], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (COLLECT) {
        unsigned qb = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) qb = __builtin_amdgcn_alignbit(qb, __float_as_uint(acc[c][r]), 31);
        keep += qb;
      } else {
        keep += __float_as_uint(acc[c][0]) >> 31;
      }
    }
    a[0].u.x += keep & 1u;       // (keeps the loop from being hoisted)
  }
  if (keep == 0x12345678u) out[0] = keep;
}

template <int CH, int LEN, bool COLLECT>
static void run(int waves_per_simd, const uint4 *src, unsigned *out) {
  const int iters = 4000, cus = 256;
  const dim3 grid(cus * 4 * waves_per_simd), block(64);
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  hipLaunchKernelGGL((mfma_loop<CH, LEN, COLLECT>), grid, block, 0, 0, src, 10, out);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((mfma_loop<CH, LEN, COLLECT>), grid, block, 0, 0, src, iters, out);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms = 0;
  hipEventElapsedTime(&ms, s, e);
  const double mfma_per_simd = (double)waves_per_simd * iters * CH * LEN;
  const double cyc = ms * 1e-3 * 2.45e9;          // at the 2.45 GHz the scoring kernels were measured at
  printf("chains %d  length %d  collect %d  waves/SIMD %d: %8.3f ms  %5.1f cycles per MFMA per SIMD (32 = pipe rate) -> %4.1f %% of the pipe\n",
         CH, LEN, (int)COLLECT, waves_per_simd, ms, cyc / mfma_per_simd, 100.0 * 32.0 * mfma_per_simd / cyc);
}

int main() {
  uint4 *src;
  unsigned *out;
  hipMalloc(&src, 1024 * sizeof(uint4));
  hipMalloc(&out, 64);
  std::vector<unsigned> h(4096);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c003c00u + (unsigned)(i * 2654435761u >> 20);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int w : {1, 2, 3, 4}) {
    run<1, 5, false>(w, src, out);
    run<2, 5, false>(w, src, out);
    run<3, 5, false>(w, src, out);
    run<3, 5, true>(w, src, out);
    run<4, 5, false>(w, src, out);
    run<3, 16, false>(w, src, out);
  }
  return 0;
}
