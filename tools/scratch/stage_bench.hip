// micro-benchmark: the staging access pattern of conv133_kernel in isolation (diagnostic).
//   hipcc --offload-arch=gfx950 -O3 stage_bench.hip -o stage_bench && ./stage_bench
// A workgroup (512 threads) owns a 16x32 output tile of one (n, d) slice and walks C input planes in chunks of 8:
// wave w loads plane chunk*8+w of the chunk (18 rows x 10 float4 groups, halo included) into registers, writes it to LDS,
// barrier.  Variants: channel-major (NCDHW: planes of a slice are D*H*W*4 bytes apart) against slice-major (NDCHW:
// planes of a slice adjacent) addressing; with/without LDS write and barriers; epilogue stores of Q planes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct P {
  const float* x; float* y;
  int C, Q, D, H, W;
  long long xs_c, xs_d, xs_n;     // element strides of the source
  long long ys_c, ys_d, ys_n;
  int tiles_x, tiles_y, total;
  int mode;                       // bit0: LDS write + barriers, bit1: epilogue store, bit2: skip loads
};

__device__ __forceinline__ int xcd_remap(int b, int n) { const int per = n / 8; return (b % 8) * per + b / 8; }

__global__ __launch_bounds__(512, 2) void stage_kernel(P p) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 18 * 40];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int item = (gridDim.x % 8 == 0) ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  if (item >= p.total) return;
  int t = item;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; t /= p.tiles_y;
  const int d = t % p.D; const int n = t / p.D;
  const int h0 = ty * 16, w0 = tx * 32;
  int goff[3], loff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int u = lane + 64 * i; if (u >= 180) u = 179;
    const int r = u / 10, q = u - r * 10;
    int hi = h0 - 1 + r; hi = hi < 0 ? 0 : (hi >= p.H ? p.H - 1 : hi);
    int gc = w0 - 4 + 4 * q; gc = gc < 0 ? 0 : (gc + 3 >= p.W ? p.W - 4 : gc);
    goff[i] = hi * p.W + gc;
    loff[i] = r * 40 + 4 * q;
  }
  float acc = 0.f;
  const int nch = p.C / 8;
  f32x4_t v[3];
  auto pref = [&](int c) {
    const float* base = p.x + n * p.xs_n + d * p.xs_d + (long long)(c * 8 + wave) * p.xs_c;
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const f32x4_t*>(base + goff[i]);
  };
  if (!(p.mode & 4)) pref(0);
  for (int c = 0; c < nch; ++c) {
    if (p.mode & 1) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float* dst = lds + wave * 720 + loff[i];
        f32x4_t w = v[i];
        w.x = fmaxf(w.x * 1.01f + 0.5f, 0.01f * w.x); w.y = fmaxf(w.y * 1.01f + 0.5f, 0.01f * w.y);
        w.z = fmaxf(w.z * 1.01f + 0.5f, 0.01f * w.z); w.w = fmaxf(w.w * 1.01f + 0.5f, 0.01f * w.w);
        *reinterpret_cast<f32x4_t*>(dst) = w;
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (c + 1 < nch && !(p.mode & 4)) pref(c + 1);
    if (p.mode & 1) {
      acc += lds[(tid * 7 + c) % (8 * 720)];
      __syncthreads();
    }
  }
  // epilogue: each wave stores Q/8 planes of the 16x32 tile, a lane a 2x4 micro-tile (two float4 rows)
  const int lx = lane & 7, ly = lane >> 3;
  if (p.mode & 2) {
    const int per = p.Q / 8;
    for (int a = 0; a < per; ++a) {
      float* yp = p.y + n * p.ys_n + d * p.ys_d + (long long)(wave * per + a) * p.ys_c;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4_t o; o.x = acc; o.y = acc + 1; o.z = acc + 2; o.w = acc + 3;
        *reinterpret_cast<f32x4_t*>(yp + (long long)(h0 + ly * 2 + i) * p.W + w0 + lx * 4) = o;
      }
    }
  } else if (acc == 123.456f) p.y[tid] = acc;
}

__global__ void copy_kernel(const f32x4_t* __restrict__ a, f32x4_t* __restrict__ b, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read_kernel(const f32x4_t* __restrict__ a, float* out, long long n4) {
  float s = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) { f32x4_t v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 1.2345f) out[0] = s;
}

template <class F> float run(F f, int it = 10) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < it; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / it;
}

int main() {
  const int B = 2, C = 64, Q = 32, D = 128, H = 128, W = 128;
  const long long HW = (long long)H * W;
  float *x, *y;
  const size_t xb = (size_t)B * C * D * HW * 4, yb = (size_t)B * Q * D * HW * 4;
  hipMalloc(&x, xb); hipMalloc(&y, yb); hipMemset(x, 0, xb); hipMemset(y, 0, yb);
  P p; p.x = x; p.y = y; p.C = C; p.Q = Q; p.D = D; p.H = H; p.W = W;
  p.tiles_x = W / 32; p.tiles_y = H / 16; p.total = B * D * p.tiles_x * p.tiles_y;
  printf("workgroups %d, read %.2f GB, write %.2f GB\n", p.total, xb / 1e9, yb / 1e9);
  {
    float ms = run([&] { hipLaunchKernelGGL(read_kernel, dim3(256 * 16), dim3(256), 0, 0, (const f32x4_t*)x, y, (long long)(xb / 16)); });
    printf("%-52s %.3f ms  %.0f GB/s\n", "plain coalesced read of x", ms, xb / ms / 1e6);
    ms = run([&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, (const f32x4_t*)x, (f32x4_t*)y, (long long)(yb / 16)); });
    printf("%-52s %.3f ms  %.0f GB/s (r+w)\n", "plain copy of |y| bytes", ms, 2.0 * yb / ms / 1e6);
  }
  for (int layout = 0; layout < 2; ++layout) {
    if (layout == 0) { p.xs_c = D * HW; p.xs_d = HW; p.xs_n = C * D * HW; p.ys_c = D * HW; p.ys_d = HW; p.ys_n = Q * D * HW; }
    else { p.xs_c = HW; p.xs_d = C * HW; p.xs_n = C * D * HW; p.ys_c = HW; p.ys_d = Q * HW; p.ys_n = Q * D * HW; }
    const char* ln = layout == 0 ? "NCDHW" : "NDCHW";
    const struct { int mode; const char* name; } V[] = {
      {0, "loads only (no LDS, no barrier, no store)"}, {1, "loads + in_act + LDS write + 2 barriers"},
      {2, "loads + epilogue stores"}, {3, "loads + LDS + barriers + epilogue stores"}, {4 | 2, "epilogue stores only"},
      {4 | 1, "LDS write + barriers only"}};
    for (auto& v : V) {
      p.mode = v.mode;
      float ms = run([&] { hipLaunchKernelGGL(stage_kernel, dim3(p.total), dim3(512), 0, 0, p); });
      double bytes = ((v.mode & 4) ? 0.0 : (double)xb) + ((v.mode & 2) ? (double)yb : 0.0);
      printf("%s %-46s %.3f ms  %.0f GB/s\n", ln, v.name, ms, bytes / ms / 1e6);
    }
  }
  return 0;
}
