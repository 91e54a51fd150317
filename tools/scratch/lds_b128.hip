// LDS ds_read_b128 / ds_write_b128 conflict probe (gfx950): cycles per wave-instruction for lane->address maps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* __restrict__ offs, unsigned long long* out, int iters, int write) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const int off = offs[threadIdx.x & 63];
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned a = (unsigned)(size_t)(lds + off) ;
  u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
  for (int i = 0; i < iters; ++i) {
    if (write) {
      asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n"
                   "ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(acc) : "memory");
    } else {
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8\n ds_read_b128 %2, %8\n ds_read_b128 %3, %8\n"
                   "ds_read_b128 %4, %8\n ds_read_b128 %5, %8\n ds_read_b128 %6, %8\n ds_read_b128 %7, %8\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
      acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc.x; }
}
int main() {
  int* d; unsigned long long* o;
  hipMalloc(&d, 64 * 4); hipMalloc(&o, 16);
  struct P { const char* name; std::function<int(int)> f; };
  std::vector<P> pats = {
    {"linear lane*16", [](int l) { return l * 16; }},
    {"lane*32 (all)", [](int l) { return l * 32; }},
    {"(l&31)*32 + (l>>5)*16   [W rows, no swizzle]", [](int l) { return (l & 31) * 32 + (l >> 5) * 16; }},
    {"(l&31)*32 + ((l>>5)^((l&31)>>4))*16 [W rows, swizzled]", [](int l) { int fq = l & 31, h = l >> 5; return fq * 32 + ((h ^ (fq >> 4)) << 4); }},
    {"(l&31)*48 + (l>>5)*16   [A frag, PXB 48]", [](int l) { return (l & 31) * 48 + (l >> 5) * 16; }},
    {"(l&31)*64 + (l>>5)*16   [stride 64]", [](int l) { return (l & 31) * 64 + (l >> 5) * 16; }},
    {"(l&31)*80 + (l>>5)*16   [stride 80]", [](int l) { return (l & 31) * 80 + (l >> 5) * 16; }},
    {"(l&15)*416 + (l>>4)*16  [bf3 v1 CSTR 416]", [](int l) { return (l & 15) * 416 + (l >> 4) * 16; }},
    {"(l&31)*400 + (l>>5)*16  [bf3 v2 CSTR2 400]", [](int l) { return (l & 31) * 400 + (l >> 5) * 16; }},
    {"all same address (broadcast)", [](int l) { return 0; }},
    {"lane*256 (same bank)", [](int l) { return l * 256; }},
    {"(l&31)*144+(l>>5)*16 [SPITCH 36 floats write]", [](int l) { return (l & 31) * 144 + (l >> 5) * 16; }},
    {"(l&31)*32 + (l>>5)*16 + ((l&31)>>3)*16 ... rot", [](int l) { int fq = l & 31, h = l >> 5; return fq * 32 + (((h + (fq >> 3)) & 1) << 4); }},
  };
  for (auto& p : pats) {
    int h[64];
    for (int l = 0; l < 64; ++l) h[l] = p.f(l);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int write = 0; write < 2; ++write) {
      const int iters = 2000;
      probe<<<1, 64>>>(d, o, iters, write);
      probe<<<1, 64>>>(d, o, iters, write);
      hipDeviceSynchronize();
      unsigned long long r[2];
      hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
      // s_memtime ticks at 100 MHz; report ns per instruction
      printf("%-58s %s  %.2f cycles/instr\n", p.name, write ? "write" : "read ", (double)r[0] / (iters * 8.0));
    }
  }
  return 0;
}
