// Diagnostic: is an fp32 product rebuilt from three bf16 pieces per operand on the bf16 matrix pipe as accurate as an fp32
// FMA chain?   a = a_hi + a_mid + a_lo exactly (truncation splits: 8 + 8 + 8 significant bits); the six leading cross
// products hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//   hipcc --offload-arch=gfx950 -O3 -o bf3_numerics bf3_numerics.hip && ./bf3_numerics
// Prints, for D = A[32xK] B[Kx32] with K = 16..262144: error vs fp64 of (a) the bf16x3 path, (b) v_mfma_f32_32x32x2_f32
// (an exact fp32 fma chain), (c) bf16x3 with only 3 products; normalised by sum|a b|; plus the mean signed error (bias).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split3(float v, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const unsigned h = u & 0xffff0000u;
  const float r1 = v - __builtin_bit_cast(float, h);          // exact
  const unsigned m = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
  const float r2 = r1 - __builtin_bit_cast(float, m);         // exact, <= 8 significant bits
  hi = h >> 16; mid = m >> 16; lo = __builtin_bit_cast(unsigned, r2) >> 16;
}

// A row-major [32][K], B stored as Bt [32][K] (column n of B contiguous in k): both fragments are 8 consecutive k per lane
template <int NPROD>
__global__ __launch_bounds__(64) void bf3_kernel(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ D, int K) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 a[3], b[3];
    for (int j = 0; j < 8; ++j) {
      unsigned x, y, z;
      split3(A[(long long)r * K + k0 + 8 * h + j], x, y, z);
      a[0][j] = (short)x; a[1][j] = (short)y; a[2][j] = (short)z;
      split3(Bt[(long long)r * K + k0 + 8 * h + j], x, y, z);
      b[0][j] = (short)x; b[1][j] = (short)y; b[2][j] = (short)z;
    }
    if (NPROD >= 6) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

__global__ __launch_bounds__(64) void f32_kernel(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ D, int K) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(long long)r * K + k0 + h], Bt[(long long)r * K + k0 + h], acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

int main() {
  const int KMAX = 262144;
  std::vector<float> A(32LL * KMAX), B(32LL * KMAX);
  srand(1);
  auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0) * (1.f + (rand() % 7)); };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 32 * 32 * 4);
  for (int K : {16, 256, 4096, 65536, 262144}) {
    // repack to row stride K
    std::vector<float> a(32LL * K), b(32LL * K);
    for (int r = 0; r < 32; ++r)
      for (int k = 0; k < K; ++k) { a[(long long)r * K + k] = A[(long long)r * KMAX + k]; b[(long long)r * K + k] = B[(long long)r * KMAX + k]; }
    hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(32 * 32), mag(32 * 32);
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double s = 0, t = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)a[(long long)m * K + k] * (double)b[(long long)n * K + k]; s += p; t += fabs(p); }
        ref[m * 32 + n] = s; mag[m * 32 + n] = t;
      }
    std::vector<float> out(32 * 32);
    for (int variant = 0; variant < 3; ++variant) {
      if (variant == 0) hipLaunchKernelGGL(bf3_kernel<6>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
      if (variant == 1) hipLaunchKernelGGL(f32_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
      if (variant == 2) hipLaunchKernelGGL(bf3_kernel<3>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
      hipMemcpy(out.data(), dD, out.size() * 4, hipMemcpyDeviceToHost);
      double mx = 0, rms = 0, bias = 0;
      for (int i = 0; i < 32 * 32; ++i) {
        const double e = ((double)out[i] - ref[i]) / mag[i];
        mx = fmax(mx, fabs(e)); rms += e * e; bias += e;
      }
      printf("K %7d %-22s err/sum|ab|: max %.3e rms %.3e mean(signed) %+.3e\n", K,
             variant == 0 ? "bf16x3, 6 products" : (variant == 1 ? "fp32 mfma (fma chain)" : "bf16x3, 3 products"), mx, sqrt(rms / 1024), bias / 1024);
    }
  }
  return 0;
}
