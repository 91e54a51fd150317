// Diagnostic: do the matrix instructions of one wave and the vector instructions of ANOTHER wave on the same SIMD overlap on gfx950?
// One 512-thread workgroup per CU = 2 waves per SIMD: waves 0-3 issue 108 independent-accumulator v_mfma_f32_32x32x16_bf16 per
// "tile" (9 accumulators of 16 registers), waves 4-7 issue NV vector instructions per tile (v_perm / v_and / v_sub mix, the
// conversion's instructions).  Prints the time of each alone and of both together; optional LDS fragment reads in the MFMA wave.
//   hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LDSR>
__global__ __launch_bounds__(512, 2) void k(float* out, int tiles, int mode, int nv) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  float res = 0.f;
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    bf16x8 a[3], b[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) { a[s] = bf16x8{1, 2, 3, 4, 5, 6, 7, (short)lane}; b[s] = bf16x8{(short)s, 2, 3, 4, 5, 6, 7, (short)lane}; }
    for (int tile = 0; tile < tiles; ++tile) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (LDSR) {
          const unsigned char* base = lds + ((tile * 2 + half) & 7) * 4096 + (lane & 31) * 400 + (lane >> 5) * 16;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            a[s] = *reinterpret_cast<const bf16x8*>(base + s * 13000 % 16384 / 16 * 16);
            b[s] = *reinterpret_cast<const bf16x8*>(base + 16384 + s * 12800);
          }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          f32x16 c = acc[t];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
          acc[t] = c;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) res += acc[t][i];
  } else {
    if (!(mode & 2)) return;
    unsigned x = lane * 77u + 1u, y = lane * 13u + 7u, z = 0x12345u;
    float f = lane * 0.5f, g = 1.5f;
    for (int tile = 0; tile < tiles; ++tile)
      for (int i = 0; i < nv; i += 10) {
        asm volatile(
            "v_perm_b32 %0, %1, %2, %3\n v_and_b32 %1, 0xffff0000, %0\n v_sub_f32 %4, %4, %5\n v_perm_b32 %2, %0, %1, %3\n"
            "v_and_b32 %0, 0xffff0000, %2\n v_sub_f32 %5, %5, %4\n v_perm_b32 %1, %2, %0, %3\n v_fma_f32 %4, %5, %4, %5\n"
            "v_max_f32 %5, %4, %5\n v_cndmask_b32 %0, %1, %2, vcc\n"
            : "+v"(x), "+v"(y), "+v"(z) : "v"(0x07060302u), "v"(f), "v"(g) : "vcc");
      }
    res = __builtin_bit_cast(float, x ^ y ^ z) + f + g;
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles = 2000;
  auto run = [&](int ldsr, int mode, int nv) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (ldsr) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, tiles, mode, nv);
      else hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, tiles, mode, nv);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
  };
  for (int ldsr = 0; ldsr < 2; ++ldsr) {
    const float tm = run(ldsr, 1, 0);
    printf("lds reads %d: MFMA wave alone: %.3f ms = %.0f ns per tile of 108 MFMAs (%.1f ns per MFMA; 32 cycles at 2.4 GHz = 13.3 ns)\n", ldsr, tm, tm * 1e6 / tiles, tm * 1e6 / tiles / 108);
    for (int nv : {250, 500, 750, 1000}) {
      const float tv = run(ldsr, 2, nv), tb = run(ldsr, 3, nv);
      printf("  %4d VALU per tile: VALU wave alone %.3f ms (%.2f ns per instr), both %.3f ms  (sum %.3f, max %.3f)\n", nv, tv, tv * 1e6 / tiles / nv, tb, tm + tv, tm > tv ? tm : tv);
    }
  }
  return 0;
}
