// micro-benchmarks: v_fma_f32 vs v_pk_fma_f32 issue rate, ds_read_b128 / b64 rate (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_fma(float* out, int iters, float w) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], w, 0.5f);
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k_pk(float* out, int iters, float w) {
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const f2 wv = {w, w}, c = {0.5f, 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], wv, c);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
__global__ __launch_bounds__(512) void k_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float* p = lds + (threadIdx.x >> 6) * 512 + lane * 4;
  f4 s = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE == 0) { f4 v = *(const f4*)(p + r * 256 % 4096); s += v; }
      if (MODE == 1) { f2 v = *(const f2*)(p + r * 256 % 4096); s.x += v.x; s.y += v.y; }
      if (MODE == 2) { float v = *(p + r * 256 % 4096); s.x += v; }
      if (MODE == 3) { f4 v = *(const f4*)(lds + r * 16); s += v; }     // broadcast (all lanes one address)
    }
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}
template <class F> float run(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 512 * 4);
  const int blocks = 2048, iters = 2000;
  float ms = run([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0001f); });
  printf("v_fma_f32   : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * blocks * 512 * iters * 64.0 / ms / 1e9);
  ms = run([&] { hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0001f); });
  printf("v_pk_fma_f32: %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * blocks * 512 * iters * 64.0 / ms / 1e9);
  const char* names[4] = {"ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_read_b128 bcast"};
  const int bytes[4] = {16, 8, 4, 16};
  for (int m = 0; m < 4; ++m) {
    if (m == 0) ms = run([&] { hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(512), 0, 0, out, iters); });
    if (m == 1) ms = run([&] { hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(512), 0, 0, out, iters); });
    if (m == 2) ms = run([&] { hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(512), 0, 0, out, iters); });
    if (m == 3) ms = run([&] { hipLaunchKernelGGL(k_lds<3>, dim3(blocks), dim3(512), 0, 0, out, iters); });
    const double tot = (double)blocks * 512 * iters * 8 * bytes[m];
    printf("%-20s: %.3f ms  %.1f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", names[m], ms, tot / ms / 1e9, tot / (ms * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
