// Cycles of v_mfma_f32_32x32x8_bf16_1k next to v_mfma_f32_32x32x16_bf16 on gfx950: chains of NX16 x16 steps + NX8 x8 steps.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int NX16, int NX8>
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters) {
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)in[(threadIdx.x + i * 8 + j) & 1023]; b[i][j] = (__bf16)in[(threadIdx.x * 3 + i + j) & 1023]; }
  s16x4 a4, b4;
  for (int j = 0; j < 4; ++j) { a4[j] = (short)threadIdx.x; b4[j] = (short)(threadIdx.x + j); }
  f32x16 acc0, acc1, s0 = {}, s1 = {};
  for (int it = 0; it < iters; ++it) {
    for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
#pragma unroll
    for (int q = 0; q < NX16; ++q) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q & 7], b[q & 7], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q & 7], b[(q + 1) & 7], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < NX8; ++q) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc1, 0, 0, 0);
    }
    s0 += acc0; s1 += acc1;
  }
  float r = 0; for (int i = 0; i < 16; ++i) r += s0[i] + s1[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int A, int B>
void run(const char *name, const float *in, float *out) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  const int iters = 4000;
  k<A, B><<<256 * 3, 256>>>(in, out, 10);
  hipEventRecord(s);
  k<A, B><<<256 * 3, 256>>>(in, out, iters);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  printf("%-28s %8.3f ms  -> %.1f ns per (x16 pair-step equivalent) chain step\n", name, ms, ms * 1e6 / iters / 3 / (A + B));
}
int main() {
  float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 256 * 3 * 256 * 4); hipMemset(in, 0, 4096);
  run<8, 0>("8 x16", in, out);
  run<9, 0>("9 x16", in, out);
  run<8, 1>("8 x16 + 1 x8", in, out);
  run<8, 2>("8 x16 + 2 x8", in, out);
  run<0, 9>("9 x8", in, out);
  run<4, 0>("4 x16", in, out);
  run<5, 0>("5 x16", in, out);
  run<4, 1>("4 x16 + 1 x8", in, out);
  return 0;
}
