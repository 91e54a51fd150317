// Round 5 diagnostic: what does the memory system give a persistent one-workgroup-per-CU kernel that reads activations the way
// conv133_mm_kernel does -- per chunk 16 channel planes x ROWS rows x 512 B (a 4 x 128 tile + halo of a 128-wide plane, planes 8 MB
// apart), NW waves issuing 16-byte-per-lane loads with DEPTH requests in flight per wave -- and optionally stores 32 planes x 2 KB
// per item?  No LDS, no arithmetic beyond a checksum: the ceiling of the access pattern itself.
//   hipcc --offload-arch=gfx950 -O3 req_path.hip -o req_path.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ sink, int planes, int rows_per_tile,
                                          int tiles_per_plane, int items, int nchunks, int store, int G) {
  // item = (batch-depth slice s, tile t): reads nchunks x 16 planes (channel c of slice s) rows [4 t - 1, 4 t + rows_per_tile - 1)
  const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long plane_elems = 128ll * 128;       // one depth slice of one channel: 64 KB
  const long long chan_stride = 128ll * plane_elems;  // 128 depth slices per channel: 8 MB
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = blockIdx.x; it < items; it += G) {
    const int s = it / tiles_per_plane, t = it % tiles_per_plane;
    for (int c0 = 0; c0 < nchunks * 16; c0 += 16) {
      // requests of this chunk: 16 channels x rows_per_tile rows x 128 floats = rows_per_tile * 32 float4 per channel
      const int per_chan = rows_per_tile * 32;
      const int total = 16 * per_chan;               // float4 requests per chunk
      for (int base = wave * 64 * DEPTH; base < total; base += nw * 64 * DEPTH) {
        f4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int r = base + d * 64 + lane;
          const int ch = r / per_chan, q = r % per_chan;
          int row = 4 * t - 1 + q / 32;
          row = row < 0 ? 0 : (row > 127 ? 127 : row);
          const float* p = x + (long long)((c0 + ch) % planes) * chan_stride + (long long)s * plane_elems + row * 128 + (q % 32) * 4;
          v[d] = r < total ? *reinterpret_cast<const f4*>(p) : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
      }
    }
    if (store == 1) {                                // 32 out planes x 4 rows x 512 B, 16 bytes per lane
      for (int r = threadIdx.x; r < 32 * 4 * 32; r += blockDim.x) {
        const int ch = r / 128, q = r % 128;
        float* p = y + (long long)ch * chan_stride + (long long)s * plane_elems + (4 * t + q / 32) * 128 + (q % 32) * 4;
        *reinterpret_cast<f4*>(p) = acc;
      }
    } else if (store >= 2 && wave < 4) {             // as K1m's epilogue: the first four waves only, 4 bytes per lane, 32 consecutive pixels of two
      const int fq = lane & 31, fh = lane >> 5;      // channels per instruction; wave w owns tile row w (store == 3: the same through atomics)
      for (int i = 0; i < 16; ++i) {
        const int ch = (i & 3) + 8 * (i >> 2) + 4 * fh;
        for (int a = 0; a < 4; ++a) {
          float* p = y + (long long)ch * chan_stride + (long long)s * plane_elems + (4 * t + wave) * 128 + a * 32 + fq;
          if (store == 3) asm volatile("global_atomic_add_f32 %0, %1, off" :: "v"(p), "v"(acc[0]) : "memory");
          else *p = acc[a];
        }
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = 1.f;
}

int main() {
  const int planes = 64;
  const size_t bytes = (size_t)planes * 128 * 128 * 128 * 4;   // 64 channels x 8 MB = 512 MB
  float *x, *y, *sink;
  hipMalloc(&x, bytes); hipMalloc(&y, bytes / 2); hipMalloc(&sink, 64);
  hipMemset(x, 0, bytes); hipMemset(y, 0, bytes / 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int slices = 128, tiles = 32, items = slices * tiles;   // one batch item of a 128^3 plane set
  printf("%-8s %-6s %-6s %-5s %-6s %10s %12s %12s\n", "waves", "depth", "rows", "store", "chunks", "ms", "read TB/s", "payload TB/s");
  for (int store : {0, 1, 2, 3})
    for (int rows : {6})
      for (int nw : {8})
        for (int depth : {8, 16}) {
          const int nchunks = 4;
          auto launch = [&]() {
            if (depth == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(64 * nw), 0, 0, x, y, sink, planes, rows, tiles, items, nchunks, store, 256);
            else if (depth == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(64 * nw), 0, 0, x, y, sink, planes, rows, tiles, items, nchunks, store, 256);
            else hipLaunchKernelGGL(k<16>, dim3(256), dim3(64 * nw), 0, 0, x, y, sink, planes, rows, tiles, items, nchunks, store, 256);
          };
          launch(); hipDeviceSynchronize();
          hipEventRecord(e0);
          for (int i = 0; i < 5; ++i) launch();
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
          const double rd = (double)items * nchunks * 16 * rows * 512.0, payload = (double)items * nchunks * 16 * 4 * 512.0 + (store ? (double)items * 32 * 4 * 512.0 : 0.0);   // (store 2 / 3: four loader + four storing waves of the eight)
          printf("%-8d %-6d %-6d %-5d %-6d %10.3f %12.2f %12.2f\n", nw, depth, rows, store, nchunks, ms, rd / ms / 1e9, payload / ms / 1e9);
        }
  return 0;
}
