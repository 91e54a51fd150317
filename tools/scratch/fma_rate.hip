// Diagnostic: issue rate of v_fma_f32 / v_pk_fma_f32 on gfx950 at 1, 2, 4 waves per SIMD (answers: is a wave64 fp32 FMA 2 or 4
// SIMD cycles, and does the packed form double the rate?).   hipcc --offload-arch=gfx950 -O3 fma_rate.hip -o fma_rate.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void fma_kernel(float* out, int iters, float a, float b) {
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
    if (PK) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          f32x2 c = {acc[i], acc[i + 1]}, x = {a, a}, y = {b, b};
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y));
          acc[i] = c[0]; acc[i + 1] = c[1];
        }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int pk = 0; pk < 2; ++pk)
    for (int wps : {1, 2, 4}) {                       // waves per SIMD: blocks of 256 threads = 1 wave per SIMD; wps blocks per CU
      const int blocks = 256 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(fma_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(fma_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fmas = (double)blocks * 256 * iters * 128;      // lane FMAs
        if (rep) printf("%s waves/SIMD %d: %.3f ms, %.1f TFLOP/s, %.2f SIMD-cycles per wave-instruction at 2.0 GHz\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms,
                        2 * fmas / ms / 1e9, ms * 1e-3 * 2.0e9 / ((double)iters * (pk ? 64 : 128) * wps));
      }
    }
  return 0;
}
