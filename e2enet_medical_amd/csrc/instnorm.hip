// K2 / K7: InstanceNorm statistics finalize and InstanceNorm + LeakyReLU backward (gfx950).
// Reference semantics: nn.InstanceNorm3d(eps=1e-5, affine=True, instance statistics in train and eval) followed
// by nn.LeakyReLU(0.01) (unetpp_d.py:99-100,111); backward = autograd of the same.
#include "e2e_common.h"

namespace {

// Chan et al. pairwise combination of (count, mean, M2)
struct Stat {
  double n, mean, m2;
};
__device__ __forceinline__ Stat combine(const Stat a, const Stat b) {
  if (b.n == 0.0) return a;
  if (a.n == 0.0) return b;
  Stat r;
  r.n = a.n + b.n;
  const double delta = b.mean - a.mean;
  r.mean = a.mean + delta * (b.n / r.n);
  r.m2 = a.m2 + b.m2 + delta * delta * (a.n * b.n / r.n);
  return r;
}

__global__ __launch_bounds__(256) void in_finalize_kernel(const double* __restrict__ part, int np,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          int C) {
  const int nc = blockIdx.x;
  const int c = nc % C;
  const double* pp = part + (long long)nc * np * 3;
  Stat s{0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < np; i += 256) {
    Stat t{pp[i * 3], pp[i * 3 + 1], pp[i * 3 + 2]};
    s = combine(s, t);
  }
  __shared__ double sh[3][256];
  sh[0][threadIdx.x] = s.n;
  sh[1][threadIdx.x] = s.mean;
  sh[2][threadIdx.x] = s.m2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      Stat a{sh[0][threadIdx.x], sh[1][threadIdx.x], sh[2][threadIdx.x]};
      Stat b{sh[0][threadIdx.x + off], sh[1][threadIdx.x + off], sh[2][threadIdx.x + off]};
      a = combine(a, b);
      sh[0][threadIdx.x] = a.n;
      sh[1][threadIdx.x] = a.mean;
      sh[2][threadIdx.x] = a.m2;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double var = sh[2][0] / sh[0][0];   // biased variance
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma[c], b = beta[c];
    scale[nc] = (float)(g * rstd);
    shift[nc] = (float)(b - sh[1][0] * g * rstd);
    mean_out[nc] = (float)sh[1][0];
    rstd_out[nc] = (float)rstd;
  }
}

// ---- backward ---------------------------------------------------------------------------------------------------
#ifndef IN_BWD_VARIANT
#define IN_BWD_VARIANT 1      // A/B builds (tools/scratch/r06_k7.sh, profiles/r06_k7_ab.txt): 1 two chunks per iteration in the apply pass (+1.5 %), 2 nontemporal loads of y there (its last use: no effect)
#endif
typedef float fvec4 __attribute__((ext_vector_type(4)));
#define IN_LD(p) (*reinterpret_cast<const fvec4*>(p))
#if IN_BWD_VARIANT & 2
#define IN_LDY(p) __builtin_nontemporal_load(reinterpret_cast<const fvec4*>(p))
#else
#define IN_LDY(p) IN_LD(p)
#endif
// pass 1: s1 = sum du, s2 = sum du * xhat   per (n, c), fp64 accumulation; every block leaves ONE record (s1, s2) -- plain stores,
// no atomics, no zeroing launch in front (round 6) -- which the apply pass adds up in a fixed order
__global__ __launch_bounds__(256) void in_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float slope,
                                                            double* __restrict__ sums, int C, long long spatial,
                                                            unsigned* __restrict__ absmax, double* __restrict__ s3_zero) {
  const int nc = blockIdx.y;
  if (absmax != nullptr && blockIdx.x == 0 && nc == 0 && threadIdx.x == 0) *absmax = 0u;   // the apply pass (next launch) records max |dy|
  if (blockIdx.x == 0 && threadIdx.x == 0) s3_zero[(long long)nc * 3 + 2] = 0.0;            // ... and accumulates sum dy here
  const float mu = mean[nc], rs = rstd[nc], sca = scale[nc], shf = shift[nc];
  const float* dzp = dz + (long long)nc * spatial;
  const float* yp = y + (long long)nc * spatial;
  float s1 = 0.f, s2 = 0.f;
  double d1 = 0.0, d2 = 0.0;
  const long long stride = (long long)gridDim.x * 256 * 4;
  const bool vec = (spatial % 4) == 0;
  int it = 0;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < spatial; i += stride) {
    float dv[4], yv[4];
    if (vec) {
      const float4 a = *reinterpret_cast<const float4*>(dzp + i);
      const float4 q = *reinterpret_cast<const float4*>(yp + i);
      dv[0] = a.x; dv[1] = a.y; dv[2] = a.z; dv[3] = a.w;
      yv[0] = q.x; yv[1] = q.y; yv[2] = q.z; yv[3] = q.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool in = i + k < spatial;
        dv[k] = in ? dzp[i + k] : 0.f;
        yv[k] = in ? yp[i + k] : mu;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (yv[k] - mu) * rs;
      const float u = fmaf(yv[k], sca, shf);        // the forward's own u (e2e::in_act): the SAME branch in both directions
      const float du = u > 0.f ? dv[k] : dv[k] * slope;
      s1 += du;
      s2 = fmaf(du, xh, s2);
    }
    if ((++it & 15) == 0) {   // flush the fp32 running sums into fp64 every 64 elements
      d1 += s1; d2 += s2; s1 = 0.f; s2 = 0.f;
    }
  }
  d1 += s1; d2 += s2;
  d1 = e2e::wave_sum_d(d1);
  d2 = e2e::wave_sum_d(d2);
  __shared__ double sh[2][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sh[0][wave] = d1; sh[1][wave] = d2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* r = sums + ((long long)nc * gridDim.x + blockIdx.x) * 2;         // (here `sums` is the record area of the workspace)
    r[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    r[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

// pass 2: dy = gamma * rstd * (du - s1/N - xhat * s2/N), in place; s3 = sum dy (bias gradient); absmax (optional): the bit
// pattern of max |dy| over the whole tensor (atomicMax on the unsigned pattern of a non-negative float: order-independent, so
// deterministic), from which the fp16 two-piece weight gradient takes its power-of-two scale (conv133_wgrad_bf3.hip).
// Round 6: (s1, s2) of the block's (n, c) are the sum of `nrec` records (of pass 1, or of the last writers of dz: rec != nullptr),
// added up by the block's first wave in a fixed order; with rec == nullptr they are read from sums[nc] (in_bwd_tile_sums_kernel).
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(float* __restrict__ dz, const float* __restrict__ y,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ gamma, float slope,
                                                           double* __restrict__ sums, const double* __restrict__ rec, int nrec,
                                                           int B, int C, long long spatial, unsigned* __restrict__ absmax,
                                                           double* __restrict__ rec3) {
  const int nc = blockIdx.y;
  const int c = nc % C;
  const float mu = mean[nc], rs = rstd[nc], g = gamma[c], sca = scale[nc], shf = shift[nc];
  __shared__ double s12[2];
  if (rec != nullptr) {
    if (threadIdx.x < 64) {
      const double* r = rec + (long long)nc * nrec * 2;
      double a = 0.0, b = 0.0;
      for (int i = threadIdx.x; i < nrec; i += 64) { a += r[2 * i]; b += r[2 * i + 1]; }
      a = e2e::wave_sum_d(a);
      b = e2e::wave_sum_d(b);
      if (threadIdx.x == 0) {
        s12[0] = a; s12[1] = b;
        if (blockIdx.x == 0) { sums[(long long)nc * 3] = a; sums[(long long)nc * 3 + 1] = b; }     // for the parameter sums
      }
    }
  } else if (threadIdx.x == 0) {
    s12[0] = sums[(long long)nc * 3]; s12[1] = sums[(long long)nc * 3 + 1];
  }
  __syncthreads();
  const double inv_n = 1.0 / (double)spatial;
  const float m1 = (float)(s12[0] * inv_n);
  const float m2 = (float)(s12[1] * inv_n);
  const float grs = g * rs;
  float* dzp = dz + (long long)nc * spatial;
  const float* yp = y + (long long)nc * spatial;
  double acc = 0.0;
  float amax = 0.f;
  const long long stride = (long long)gridDim.x * 256 * 4;
  const bool vec = (spatial % 4) == 0;
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
#if IN_BWD_VARIANT & 1
  // two chunks per iteration: four loads in flight per thread
  if (vec) {
    for (; i + stride < spatial; i += 2 * stride) {
      const fvec4 a0 = IN_LD(dzp + i), q0 = IN_LDY(yp + i), a1 = IN_LD(dzp + i + stride), q1 = IN_LDY(yp + i + stride);
      fvec4 o0, o1;
      float part = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh0 = (q0[k] - mu) * rs, xh1 = (q1[k] - mu) * rs;
        const float u0 = fmaf(q0[k], sca, shf), u1 = fmaf(q1[k], sca, shf);
        const float du0 = u0 > 0.f ? a0[k] : a0[k] * slope, du1 = u1 > 0.f ? a1[k] : a1[k] * slope;
        o0[k] = grs * (du0 - m1 - xh0 * m2);
        o1[k] = grs * (du1 - m1 - xh1 * m2);
        part += o0[k];
        amax = fmaxf(amax, fabsf(o0[k]));
      }
      acc += part;
      part = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { part += o1[k]; amax = fmaxf(amax, fabsf(o1[k])); }
      acc += part;
      *reinterpret_cast<fvec4*>(dzp + i) = o0;
      *reinterpret_cast<fvec4*>(dzp + i + stride) = o1;
    }
  }
#endif
  for (; i < spatial; i += stride) {
    float dv[4], yv[4], o[4];
    if (vec) {
      const fvec4 a = IN_LD(dzp + i);
      const fvec4 q = IN_LDY(yp + i);
      dv[0] = a[0]; dv[1] = a[1]; dv[2] = a[2]; dv[3] = a[3];
      yv[0] = q[0]; yv[1] = q[1]; yv[2] = q[2]; yv[3] = q[3];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool in = i + k < spatial;
        dv[k] = in ? dzp[i + k] : 0.f;
        yv[k] = in ? yp[i + k] : mu;
      }
    }
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (yv[k] - mu) * rs;
      const float u = fmaf(yv[k], sca, shf);        // the forward's own u (e2e::in_act)
      const float du = u > 0.f ? dv[k] : dv[k] * slope;
      o[k] = grs * (du - m1 - xh * m2);
      if (vec || i + k < spatial) { part += o[k]; amax = fmaxf(amax, fabsf(o[k])); }
    }
    acc += part;
    if (vec) {
      *reinterpret_cast<float4*>(dzp + i) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i + k < spatial) dzp[i + k] = o[k];
    }
  }
  acc = e2e::wave_sum_d(acc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  __shared__ double sh[4];
  __shared__ float shm[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sh[wave] = acc; shm[wave] = amax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    rec3[(long long)nc * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];    // one record per block (no atomic: the bias gradient is bit-reproducible)
    if (absmax != nullptr) atomicMax(absmax, __builtin_bit_cast(unsigned, fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]))));
  }
}

// parameter gradients: sums over the batch (s3 was zeroed by the first pass of this call, accumulated by the apply pass).
// (A "last block of the apply pass does this" variant was measured in round 6: the device-scope fence it needs in EVERY block
//  writes the XCD's L2 back -- 16 k fences per launch took the apply pass from 3.4 to ~12 ms per step.  A separate 5 us launch it is.)
__global__ __launch_bounds__(64) void in_bwd_params_kernel(const double* __restrict__ sums, const double* __restrict__ rec3, int nblocks,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dbias, int B, int C) {
  const int c = blockIdx.x;                                   // one wave per channel
  double s3 = 0.0;
  if (dbias != nullptr) {
    // sum dy: the apply pass's per-block records of every batch item, lane t takes records t, t + 64, ... and the wave adds its
    // lanes in a fixed tree -- the same order in every run.  (Until round 6 an fp64 atomic per block: the conv bias in front of an
    // InstanceNorm has an analytically zero gradient, what is computed is the rounding residue of sum dy, and THAT depended on the
    // order of the atomics in its last fp32 bit -- tests/test_gpu_net.py::test_full_size_128_training_properties caught it once.)
    for (int n = 0; n < B; ++n) {
      const double* q = rec3 + ((long long)n * C + c) * nblocks;
      for (int b = threadIdx.x; b < nblocks; b += 64) s3 += q[b];
    }
    s3 = e2e::wave_sum_d(s3);
  }
  if (threadIdx.x != 0) return;
  double s1 = 0.0, s2 = 0.0;
  for (int n = 0; n < B; ++n) {
    const double* r = sums + ((long long)n * C + c) * 3;
    s1 += r[0];
    s2 += r[1];
  }
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  if (dbias) dbias[c] = (float)s3;
}

// first pass done by the last writers of dz (conv133_sparse.hip): add their tile records up, one wave per (n, c), fixed order
__global__ __launch_bounds__(64) void in_bwd_tile_sums_kernel(const double* __restrict__ part, double* __restrict__ sums, int np,
                                                              unsigned* __restrict__ absmax) {
  const int nc = blockIdx.x;
  if (absmax != nullptr && nc == 0 && threadIdx.x == 0) *absmax = 0u;
  const double* p = part + (long long)nc * np * 2;
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < np; i += 64) { a += p[2 * i]; b += p[2 * i + 1]; }
  a = e2e::wave_sum_d(a);
  b = e2e::wave_sum_d(b);
  if (threadIdx.x == 0) { sums[(long long)nc * 3] = a; sums[(long long)nc * 3 + 1] = b; sums[(long long)nc * 3 + 2] = 0.0; }
}

}  // namespace

extern "C" int e2e_in_stats_finalize(const double* part, int np, const float* gamma, const float* beta, float eps,
                                     float* scale, float* shift, float* mean, float* rstd, int B, int C,
                                     void* stream) {
  E2E_REQUIRE(part && gamma && beta && scale && shift && mean && rstd, "in_stats_finalize: null pointer");
  E2E_REQUIRE(np > 0 && B > 0 && C > 0, "in_stats_finalize: bad dims");
  hipLaunchKernelGGL(in_finalize_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, part, np, gamma, beta, eps,
                     scale, shift, mean, rstd, C);
  return e2e::check_launch("in_finalize_kernel");
}

// doubles of workspace e2e_in_lrelu_bwd needs for (B, C): sums [B*C*3] + pass-1 records [B*C*256*2] + the apply pass's sum-dy records
// [B*C*256]; no state survives a call
extern "C" long long e2e_in_lrelu_bwd_ws_doubles(int B, int C) { return (long long)B * C * (3 + 3 * 256) + 2; }

extern "C" int e2e_in_lrelu_bwd(float* dz_dy, const float* y, const float* mean, const float* rstd, const float* scale,
                                const float* shift, const float* gamma, float slope, float* dgamma, float* dbeta,
                                float* dbias, float* sums, int B, int C, long long spatial, const double* tile_sums, int np,
                                unsigned* dy_absmax, void* stream) {
  E2E_REQUIRE(dz_dy && y && mean && rstd && scale && shift && gamma && dgamma && dbeta && sums, "in_lrelu_bwd: null pointer");
  E2E_REQUIRE(B > 0 && C > 0 && spatial > 0, "in_lrelu_bwd: bad dims");
  hipStream_t st = (hipStream_t)stream;
  double* ds = reinterpret_cast<double*>(sums);
  double* recs = ds + (long long)B * C * 3;
  double* rec3 = recs + (long long)B * C * 256 * 2;
  long long blocks = e2e::cdivll(spatial, 256 * 4 * 4);
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  dim3 grid((unsigned)blocks, B * C);
  // three launches per conv block (round 5: zero + reduce + apply + params); the first pass leaves per-block records (plain stores:
  // deterministic, nothing to zero) that the apply pass adds up in a fixed order
  if (tile_sums == nullptr) {
    hipLaunchKernelGGL(in_bwd_reduce_kernel, grid, dim3(256), 0, st, dz_dy, y, mean, rstd, scale, shift, slope, recs, C,
                       spatial, dy_absmax, ds);
    hipLaunchKernelGGL(in_bwd_apply_kernel, grid, dim3(256), 0, st, dz_dy, y, mean, rstd, scale, shift, gamma, slope, ds,
                       (const double*)recs, (int)blocks, B, C, spatial, dy_absmax, rec3);
  } else {
    E2E_REQUIRE(np > 0, "in_lrelu_bwd: tile_sums without a record count");
    hipLaunchKernelGGL(in_bwd_tile_sums_kernel, dim3(B * C), dim3(64), 0, st, tile_sums, ds, np, dy_absmax);
    hipLaunchKernelGGL(in_bwd_apply_kernel, grid, dim3(256), 0, st, dz_dy, y, mean, rstd, scale, shift, gamma, slope, ds,
                       (const double*)nullptr, 0, B, C, spatial, dy_absmax, rec3);
  }
  hipLaunchKernelGGL(in_bwd_params_kernel, dim3(C), dim3(64), 0, st, (const double*)ds, (const double*)rec3, (int)blocks,
                     dgamma, dbeta, dbias, B, C);
  return e2e::check_launch("in_lrelu_bwd");
}
