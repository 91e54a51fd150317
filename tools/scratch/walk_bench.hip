// micro-benchmark: the live-kernel walk of conv133_kernel in isolation (diagnostic): do LDS reads and FMAs overlap?
//   hipcc --offload-arch=gfx950 -O3 walk_bench.hip -o walk_bench && ./walk_bench
// 512-thread workgroups, 2 per CU; a wave owns 4 output planes, a lane a 2x4 micro-tile; per "visit" it reads a 4x6
// neighbourhood (4 x (b128 + b64)) from a staged 18 x 40 plane and then runs K kernels (9 broadcast weights + 72 FMAs).
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int PITCH = 48, CHS = 18 * PITCH + 16;

__device__ unsigned long long g_clk[2];
template <int MODE, int K>   // MODE bit0: neighbourhood reads, bit1: weight reads, bit2: FMAs, bit3: halo columns as a second b128
__global__ __launch_bounds__(512, 2) void walk(float* out, const float* in, int visits) {
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  __shared__ __attribute__((aligned(16))) float lds[8 * CHS];
  __shared__ __attribute__((aligned(16))) float wl[32 * 8 * 12];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lx = lane & 7, ly = lane >> 3;
  for (int i = tid; i < 8 * CHS; i += 512) lds[i] = in[i];
  for (int i = tid; i < 32 * 8 * 12; i += 512) wl[i] = in[i];
  __syncthreads();
  float acc[4][2][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[a][i][j] = 0.f;
  const float* lane_tp = lds + (ly * 2) * PITCH + lx * 4;
  float nb[4][8];
  float wk[9];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) nb[r][c] = in[tid + r * 8 + c];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = in[k];
  for (int v = 0; v < visits; ++v) {
    const int cl = (v * 5 + wave) & 7;
    if (MODE & 1) {
      const float* tp = lane_tp + cl * CHS;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 a = *reinterpret_cast<const float4*>(tp + r * PITCH);
        nb[r][0] = a.x; nb[r][1] = a.y; nb[r][2] = a.z; nb[r][3] = a.w;
        if (MODE & 8) {
          const float4 b = *reinterpret_cast<const float4*>(tp + r * PITCH + 4);
          nb[r][4] = b.x; nb[r][5] = b.y; nb[r][6] = b.z; nb[r][7] = b.w;
          asm volatile("" :: "v"(nb[r][6]), "v"(nb[r][7]));
        } else {
          const float2 b = *reinterpret_cast<const float2*>(tp + r * PITCH + 4);
          nb[r][4] = b.x; nb[r][5] = b.y;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < K; ++a) {
      if (MODE & 2) {
        const float* wp = wl + ((wave * 4 + a) * 8 + cl) * 12;
        const float4 w0 = *reinterpret_cast<const float4*>(wp);
        const float4 w1 = *reinterpret_cast<const float4*>(wp + 4);
        wk[0] = w0.x; wk[1] = w0.y; wk[2] = w0.z; wk[3] = w0.w; wk[4] = w1.x; wk[5] = w1.y; wk[6] = w1.z; wk[7] = w1.w; wk[8] = wp[8];
      }
      if (MODE & 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) acc[a][i][j] = fmaf(wk[kh * 3 + kw], nb[i + kh][j + kw], acc[a][i][j]);
      } else {
        // keep the loaded values alive
        float s = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += wk[k];
        asm volatile("" :: "v"(s));
      }
    }
    if (!(MODE & 4)) {
      float s = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) s += nb[r][c];
      asm volatile("" :: "v"(s));
    }
  }
  float s = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[a][i][j];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) { atomicAdd(&g_clk[0], __builtin_readcyclecounter() - c0); atomicAdd(&g_clk[1], wall_clock64() - r0); }
}

template <class F> float run(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

template <int MODE, int K> void one(float* out, float* in, const char* name, int blocks) {
  const int visits = 2000;
  float ms = run([&] { hipLaunchKernelGGL((walk<MODE, K>), dim3(blocks), dim3(512), 0, 0, out, in, visits); });
  const double kern = (double)blocks * 8 * visits * K;
  // cycles of CU time per kernel-wave: blocks/256 rounds... report ns per (CU, kernel-wave) and TFLOP/s
  unsigned long long h[2]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), 16); unsigned long long z[2] = {0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, 16);
  printf("%-44s K=%d blocks=%4d  %.3f ms  %6.1f TFLOP/s  %.1f CU-clk(2.0GHz)/kernel  sclk %.0f MHz\n", name, K, blocks, ms,
         (MODE & 4) ? kern * 64 * 144 / ms / 1e9 : 0.0, ms * 1e-3 * 2.0e9 * 256 / kern, 100.0 * (double)h[0] / (double)h[1]);
}


// software-pipelined walk (1 kernel per visit): the reads of visit v+1 are issued before the FMAs of visit v
template <int DEPTH>
__global__ __launch_bounds__(512, 2) void walk_pipe(float* out, const float* in, int visits) {
  __shared__ __attribute__((aligned(16))) float lds[8 * CHS];
  __shared__ __attribute__((aligned(16))) float wl[32 * 8 * 12];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lx = lane & 7, ly = lane >> 3;
  for (int i = tid; i < 8 * CHS; i += 512) lds[i] = in[i];
  for (int i = tid; i < 32 * 8 * 12; i += 512) wl[i] = in[i];
  __syncthreads();
  float acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const float* lane_tp = lds + (ly * 2) * PITCH + lx * 4;
  float nb[2][4][6];
  float wk[2][9];
  auto rd = [&](int v, int s) {
    const int cl = (v * 5 + wave) & 7;
    const float* tp = lane_tp + cl * CHS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float4 a = *reinterpret_cast<const float4*>(tp + r * PITCH);
      const float2 b = *reinterpret_cast<const float2*>(tp + r * PITCH + 4);
      nb[s][r][0] = a.x; nb[s][r][1] = a.y; nb[s][r][2] = a.z; nb[s][r][3] = a.w; nb[s][r][4] = b.x; nb[s][r][5] = b.y;
    }
    const float* wp = wl + ((wave * 4 + (v & 3)) * 8 + cl) * 12;
    const float4 w0 = *reinterpret_cast<const float4*>(wp);
    const float4 w1 = *reinterpret_cast<const float4*>(wp + 4);
    wk[s][0] = w0.x; wk[s][1] = w0.y; wk[s][2] = w0.z; wk[s][3] = w0.w; wk[s][4] = w1.x; wk[s][5] = w1.y; wk[s][6] = w1.z; wk[s][7] = w1.w; wk[s][8] = wp[8];
  };
  auto fm = [&](int s) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) acc[i][j] = fmaf(wk[s][kh * 3 + kw], nb[s][i + kh][j + kw], acc[i][j]);
  };
  if (DEPTH == 0) {
    for (int v = 0; v < visits; ++v) { rd(v, 0); fm(0); }
  } else {
    rd(0, 0);
    for (int v = 0; v < visits; v += 2) {
      rd(v + 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      fm(0);
      __builtin_amdgcn_sched_barrier(0);
      rd(v + 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      fm(1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * 512 + tid] = s;
}
template <int DEPTH> void onep(float* out, float* in, const char* name, int blocks) {
  const int visits = 2000;
  float ms = run([&] { hipLaunchKernelGGL((walk_pipe<DEPTH>), dim3(blocks), dim3(512), 0, 0, out, in, visits); });
  const double kern = (double)blocks * 8 * visits;
  printf("%-44s K=1 blocks=%4d  %.3f ms  %6.1f TFLOP/s  %.1f CU-clk(2.0GHz)/kernel\n", name, blocks, ms, kern * 64 * 144 / ms / 1e9, ms * 1e-3 * 2.0e9 * 256 / kern);
}

int main() {
  float *out, *in; hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20);
  { float *o2, *i2; hipMalloc(&o2, 4096 * 512 * 4); hipMalloc(&i2, 1 << 20); hipMemset(i2, 0, 1 << 20);
    onep<0>(o2, i2, "single accumulator set, not pipelined", 512); onep<1>(o2, i2, "single accumulator set, pipelined", 512); }
  for (int blocks : {512}) {
    one<4, 1>(out, in, "FMAs only", blocks);
    one<1, 1>(out, in, "neighbourhood reads only", blocks);
    one<2, 1>(out, in, "weight reads only", blocks);
    one<3, 1>(out, in, "both reads, no FMAs", blocks);
    one<5, 1>(out, in, "neighbourhood + FMAs", blocks);
    one<6, 1>(out, in, "weights + FMAs", blocks);
    one<7, 1>(out, in, "all (1 kernel per visit)", blocks);
    one<7, 2>(out, in, "all (2 kernels per visit)", blocks);
    one<7, 4>(out, in, "all (4 kernels per visit)", blocks);
    one<9, 1>(out, in, "neighbourhood reads only, halo as b128", blocks);
    one<11, 1>(out, in, "both reads, halo as b128", blocks);
    one<15, 1>(out, in, "all, halo as b128 (1 kernel per visit)", blocks);
    one<15, 2>(out, in, "all, halo as b128 (2 kernels per visit)", blocks);
    one<15, 4>(out, in, "all, halo as b128 (4 kernels per visit)", blocks);
  }
  return 0;
}
