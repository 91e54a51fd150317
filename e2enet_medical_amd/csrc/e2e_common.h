// Shared helpers for the gfx950 kernels of libe2e_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "e2e_hip.h"

typedef const float __attribute__((address_space(1)))* gfloat_p;

namespace e2e {

void set_error(const char* fmt, ...);
// dispatch diagnostics: which kernel variant the last entry point on this thread selected (e2e_last_kernel)
void note_kernel(const char* fmt, ...);

// shader clock the two hot kernel families actually run at (e2e_diag_kernel_clock): workgroup 0 of every launch adds its
// s_memtime span (shader-clock cycles) and s_memrealtime span (constant 100 MHz) to a device-side pair -- one atomic per launch
void mm_clock_read(unsigned long long out[2], bool reset);        // conv133_mm.hip
void wgrad_clock_read(unsigned long long out[2], bool reset);     // conv133_wgrad_bf3.hip

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return E2E_ERR_LAUNCH;
  }
  return E2E_OK;
}

#define E2E_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      e2e::set_error(__VA_ARGS__);      \
      return E2E_ERR_ARG;               \
    }                                   \
  } while (0)

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

// normalise-on-load: lrelu(a*v + b)
__device__ __forceinline__ float in_act(float v, float a, float b, float slope) {
  float u = fmaf(v, a, b);
  return u > 0.f ? u : u * slope;
}

// full-wave (64 lane) sum, result valid in every lane; the first four steps are DPP moves inside the vector ALU
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), only the last two cross the 16-lane rows
// through ds_bpermute.  Six dependent ds_bpermute round trips cost several hundred cycles each under LDS load.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// the same for a double (two DPP moves per step); used where a sum must not carry an fp32 rounding that is coherent over
// a whole plane (InstanceNorm statistics of the deep levels)
__device__ __forceinline__ double dpp_move_d(double v, const int ctrl_sel) {
  const long long b = __builtin_bit_cast(long long, v);
  int lo = (int)b, hi = (int)(b >> 32);
  switch (ctrl_sel) {
    case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break;
  }
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_dpp_d(double v) {
  v += dpp_move_d(v, 0);
  v += dpp_move_d(v, 1);
  v += dpp_move_d(v, 2);
  v += dpp_move_d(v, 3);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// full-wave (64 lane) sum, result valid in every lane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// XCD-aware remap of a linear workgroup id (MI355X deals consecutive ids round-robin over the 8 XCDs):
// ids that share id % 8 run on one XCD (one L2), so hand each XCD a contiguous run of logical work items.
// Returns the logical index for `id` in a grid padded to a multiple of 8; callers bounds-check it.
__device__ __forceinline__ int xcd_remap(int id, int padded_total) {
  int per = padded_total >> 3;
  return (id & 7) * per + (id >> 3);
}

// Zero `bytes` (a multiple of 4) on the stream with a kernel.  Not hipMemsetAsync: captured into a HIP graph (ROCm 7.0) the
// memset nodes of the small accumulator buffers that several ops reuse one after the other were not reliably ordered
// against the neighbouring kernel nodes once eager work ran between two replays (NaN sums; tests/test_gpu_net.py graph
// test); a kernel node is ordered like every other kernel.
__global__ inline void zero_words_kernel(unsigned* p, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0u;
}
inline void zero_async(void* p, size_t bytes, hipStream_t st) {
  const long long n = (long long)(bytes / 4);
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned*)p, n);
}

// conv133_wgrad_bf3.hip: dense weight gradient on the bf16 matrix pipe with fp32-exact (three-piece) operands; planned and
// dispatched by e2e_conv133_wgrad (conv133_wgrad.hip)
struct WgBf3Params {
  const e2e_in_chan_t* chans;
  const float* dy;
  float* slab;                 // [chunks][Cout][Cin][9]
  int B, Cin, Cout, Di, Hi, Wi, Do, sd;
  int tiles_x, tiles_y, tiles_per_n, tiles_per_chunk;
  int segs;                    // chunks per batch item
  int cblocks;                 // ceil(Cin / 32)
  int geom;                    // 0: 4 x 32-pixel tiles, 1: 8 x 16-pixel tiles (planes 16..31 wide; v5 kernel only)
  int h2;                      // 1: fp16 two-piece operands, three products (v5 kernel only); 0: bf16 three-piece, six products
  const unsigned* dy_absmax;   // h2: bit pattern of max |dy| over the tensor (device; nullptr: dy is taken unscaled)
  const unsigned* x_absmax;    // h2: bit pattern of (a bound of) max |x| over the input planes after normalise-on-load (nullptr: fixed 2^3)
};
int launch_wgrad_bf3(const WgBf3Params& p, int nchunks, int pairs, hipStream_t st);

}  // namespace e2e
