// micro-benchmark: v_fma_f32 issue rate with 1 / 2 / 3 distinct VGPR source operands (register-file bank conflicts?)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters, float ws) {
  float acc[8], nb[24], wk[9];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 24; ++i) nb[i] = in[threadIdx.x + i * 512];
  for (int i = 0; i < 9; ++i) wk[i] = in[threadIdx.x + (24 + i) * 512];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            if (MODE == 0) acc[i * 4 + j] = fmaf(wk[kh * 3 + kw], nb[(i + kh) * 6 + j + kw], acc[i * 4 + j]);   // 3 VGPRs
            if (MODE == 1) acc[i * 4 + j] = fmaf(ws, nb[(i + kh) * 6 + j + kw], acc[i * 4 + j]);                // SGPR weight
            if (MODE == 2) acc[i * 4 + j] = fmaf(ws, acc[i * 4 + j], 0.25f);                                    // 1 VGPR
            if (MODE == 3) {   // 9 scalar weights that are rewritten by the scalar ALU every iteration (as after an s_load / s_mov)
              const int wi = (it * 9 + kh * 3 + kw) | 0x3f000000;
              acc[i * 4 + j] = fmaf(__builtin_bit_cast(float, wi), nb[(i + kh) * 6 + j + kw], acc[i * 4 + j]);
            }
          }
    asm volatile("" ::: "memory");
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> float run(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float *out, *in; hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&in, 64 * 512 * 4); hipMemset(in, 0, 64 * 512 * 4);
  const int blocks = 2048, iters = 1000;
  const char* names[4] = {"fma(v, v, v)  conv pattern", "fma(s, v, v)  scalar weight", "fma(s, v, c)  one VGPR", "fma(s9, v, v) 9 rewritten scalars"};
  for (int m = 0; m < 4; ++m) {
    float ms = 0;
    if (m == 0) ms = run([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, in, iters, 1.0001f); });
    if (m == 1) ms = run([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, in, iters, 1.0001f); });
    if (m == 2) ms = run([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, out, in, iters, 1.0001f); });
    if (m == 3) ms = run([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, out, in, iters, 1.0001f); });
    printf("%-30s %.3f ms  %.1f TFLOP/s\n", names[m], ms, 2.0 * blocks * 512 * (double)iters * 72 / ms / 1e9);
  }
  return 0;
}
