// K4 / K5: max pooling (kernel == stride) and the 1x1x1 segmentation heads, gfx950.
// Reference: nn.MaxPool3d(k) built at unetpp_d.py:523-524 ("down" fusion branch) and
// nn.Conv3d(C, K, 1, bias=False) at unetpp_d.py:394-401 (used :480-483).  Both read a conv block's pre-norm
// output and apply its InstanceNorm affine + LeakyReLU on load (z = lrelu(scale * x + shift)).
// Both are HBM-streaming ops: one coalesced pass over the input, nothing staged.
#include "e2e_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ max pool
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope,
                                                          float* __restrict__ y, int D, int H, int W, int kd, int kh, int kw,
                                                          int Do, int Ho, int Wo) {
  const int nc = blockIdx.y;
  const long long ospatial = (long long)Do * Ho * Wo;
  const long long oi = (long long)blockIdx.x * 256 + threadIdx.x;
  if (oi >= ospatial) return;
  const int wo = (int)(oi % Wo);
  const long long r = oi / Wo;
  const int ho = (int)(r % Ho), dq = (int)(r / Ho);
  float a = 1.f, b = 0.f, sl = 1.f;
  if (scale) { a = scale[nc]; b = shift[nc]; sl = slope; }
  const float* xp = x + (long long)nc * D * H * W;
  float m = -INFINITY;
  for (int i = 0; i < kd; ++i)
    for (int j = 0; j < kh; ++j)
      for (int k = 0; k < kw; ++k) {
        const float v = e2e::in_act(xp[((long long)(dq * kd + i) * H + (ho * kh + j)) * W + (wo * kw + k)], a, b, sl);
        m = (v > m || v != v) ? v : m;    // ATen: NaN propagates, first maximum wins
      }
  y[(long long)nc * ospatial + oi] = m;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope,
                                                          const float* __restrict__ dy, float* __restrict__ dx,
                                                          int accumulate, int D, int H, int W, int kd, int kh, int kw, int Do,
                                                          int Ho, int Wo) {
  const int nc = blockIdx.y;
  const long long ospatial = (long long)Do * Ho * Wo;
  const long long oi = (long long)blockIdx.x * 256 + threadIdx.x;
  if (oi >= ospatial) return;
  const int wo = (int)(oi % Wo);
  const long long r = oi / Wo;
  const int ho = (int)(r % Ho), dq = (int)(r / Ho);
  float a = 1.f, b = 0.f, sl = 1.f;
  if (scale) { a = scale[nc]; b = shift[nc]; sl = slope; }
  const float* xp = x + (long long)nc * D * H * W;
  float* dxp = dx + (long long)nc * D * H * W;
  float m = -INFINITY;
  int best = 0;
  for (int i = 0; i < kd; ++i)
    for (int j = 0; j < kh; ++j)
      for (int k = 0; k < kw; ++k) {
        const float v = e2e::in_act(xp[((long long)(dq * kd + i) * H + (ho * kh + j)) * W + (wo * kw + k)], a, b, sl);
        if (v > m || v != v) { m = v; best = (i * kh + j) * kw + k; }
      }
  const float g = dy[(long long)nc * ospatial + oi];
  for (int i = 0; i < kd; ++i)
    for (int j = 0; j < kh; ++j)
      for (int k = 0; k < kw; ++k) {
        const float val = ((i * kh + j) * kw + k) == best ? g : 0.f;
        float* dst = dxp + ((long long)(dq * kd + i) * H + (ho * kh + j)) * W + (wo * kw + k);
        if (accumulate) *dst += val;
        else *dst = val;
      }
}

// kw == 2 fast path: one thread = two adjacent pooled outputs -> each window row is one aligned float4 of the input
__global__ __launch_bounds__(256) void maxpool_fwd_w2_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float slope,
                                                             float* __restrict__ y, int D, int H, int W, int kd, int kh, int Do,
                                                             int Ho, int Wo) {
  const int nc = blockIdx.y;
  const long long pairs = (long long)Do * Ho * (Wo / 2);
  const long long pi = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pi >= pairs) return;
  const int wp = (int)(pi % (Wo / 2));
  const long long r = pi / (Wo / 2);
  const int ho = (int)(r % Ho), dq = (int)(r / Ho);
  float a = 1.f, b = 0.f, sl = 1.f;
  if (scale) { a = scale[nc]; b = shift[nc]; sl = slope; }
  const float* xp = x + (long long)nc * D * H * W;
  float m0 = -INFINITY, m1 = -INFINITY;
  for (int i = 0; i < kd; ++i)
    for (int j = 0; j < kh; ++j) {
      const float4 q = *reinterpret_cast<const float4*>(xp + ((long long)(dq * kd + i) * H + (ho * kh + j)) * W + wp * 4);
      const float v0 = e2e::in_act(q.x, a, b, sl), v1 = e2e::in_act(q.y, a, b, sl);
      const float v2 = e2e::in_act(q.z, a, b, sl), v3 = e2e::in_act(q.w, a, b, sl);
      m0 = (v0 > m0 || v0 != v0) ? v0 : m0;
      m0 = (v1 > m0 || v1 != v1) ? v1 : m0;
      m1 = (v2 > m1 || v2 != v2) ? v2 : m1;
      m1 = (v3 > m1 || v3 != v3) ? v3 : m1;
    }
  *reinterpret_cast<float2*>(y + (long long)nc * Do * Ho * Wo + ((long long)dq * Ho + ho) * Wo + wp * 2) = make_float2(m0, m1);
}

// Fused first pass of the source's InstanceNorm + LeakyReLU backward (round 4): when this launch is the LAST writer of dx -- the
// engine issues the pooling backward behind the other consumers' data gradients for that purpose -- every thread holds the final
// dz of the cells of its windows next to their pre-norm values y (it re-reads them for the argmax anyway) and the block writes
//   sum dz lrelu'(u),  sum dz lrelu'(u) xhat      (u = a y + b, xhat = (y - mean) rstd; instnorm.hip: in_bwd_reduce_kernel)
// of its cells to part[(nc * gridDim.x + blockIdx.x) * 2 ..]: one record per block, plain stores, added up in a fixed order by
// e2e_in_lrelu_bwd.  HBM-bound kernel: the sums cost no extra traffic.
__global__ __launch_bounds__(256) void maxpool_bwd_w2_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float slope,
                                                             const float* __restrict__ dy, float* __restrict__ dx,
                                                             int accumulate, int D, int H, int W, int kd, int kh, int Do, int Ho,
                                                             int Wo, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             double* __restrict__ part) {
  const int nc = blockIdx.y;
  const long long pairs = (long long)Do * Ho * (Wo / 2);
  const long long pi = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool active = pi < pairs;
  float s1 = 0.f, s2 = 0.f;
  if (active) {
    const int wp = (int)(pi % (Wo / 2));
    const long long r = pi / (Wo / 2);
    const int ho = (int)(r % Ho), dq = (int)(r / Ho);
    float a = 1.f, b = 0.f, sl = 1.f;
    if (scale) { a = scale[nc]; b = shift[nc]; sl = slope; }
    float mu = 0.f, rs = 0.f;
    if (part != nullptr) { mu = mean[nc]; rs = rstd[nc]; }
    const float* xp = x + (long long)nc * D * H * W;
    float* dxp = dx + (long long)nc * D * H * W;
    float m0 = -INFINITY, m1 = -INFINITY;
    int b0 = 0, b1 = 0;
    float4 qs[2][2];                                        // the window rows (kd, kh <= 2), kept for the fused sums
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (i >= kd || j >= kh) continue;
        const float4 q = *reinterpret_cast<const float4*>(xp + ((long long)(dq * kd + i) * H + (ho * kh + j)) * W + wp * 4);
        qs[i][j] = q;
        const float v0 = e2e::in_act(q.x, a, b, sl), v1 = e2e::in_act(q.y, a, b, sl);
        const float v2 = e2e::in_act(q.z, a, b, sl), v3 = e2e::in_act(q.w, a, b, sl);
        const int base = (i * kh + j) * 2;
        if (v0 > m0 || v0 != v0) { m0 = v0; b0 = base; }
        if (v1 > m0 || v1 != v1) { m0 = v1; b0 = base + 1; }
        if (v2 > m1 || v2 != v2) { m1 = v2; b1 = base; }
        if (v3 > m1 || v3 != v3) { m1 = v3; b1 = base + 1; }
      }
    const float2 g = *reinterpret_cast<const float2*>(dy + (long long)nc * Do * Ho * Wo + ((long long)dq * Ho + ho) * Wo + wp * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (i >= kd || j >= kh) continue;
        const int base = (i * kh + j) * 2;
        const long long off = ((long long)(dq * kd + i) * H + (ho * kh + j)) * W + wp * 4;
        float4* dst = reinterpret_cast<float4*>(dxp + off);
        float4 v = make_float4(b0 == base ? g.x : 0.f, b0 == base + 1 ? g.x : 0.f, b1 == base ? g.y : 0.f, b1 == base + 1 ? g.y : 0.f);
        if (accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *dst = v;
        if (part != nullptr) {
          const float4 q = qs[i][j];
          const float ys[4] = {q.x, q.y, q.z, q.w}, dz[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = fmaf(a, ys[e], b);
            const float du = u > 0.f ? dz[e] : dz[e] * sl;
            s1 += du;
            s2 = fmaf(du, (ys[e] - mu) * rs, s2);
          }
        }
      }
  }
  if (part != nullptr) {
    double d1 = e2e::wave_sum_d((double)s1), d2 = e2e::wave_sum_d((double)s2);
    __shared__ double sh[2][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh[0][wave] = d1; sh[1][wave] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double* rec = part + ((long long)nc * gridDim.x + blockIdx.x) * 2;
      rec[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
      rec[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
  }
}

// ------------------------------------------------------------------------------------------------ 1x1x1 head
// logits[n,k,v] = sum_c W[k,c] z[n,c,v]; KB classes per pass kept in registers, weights are wave-uniform scalars.
template <int KB>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float slope,
                                                       const float* __restrict__ w, float* __restrict__ logits, int C, int K,
                                                       long long spatial) {
  const int n = blockIdx.y;
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= spatial) return;
  for (int k0 = 0; k0 < K; k0 += KB) {
    float acc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) acc[k] = 0.f;
    for (int cb = 0; cb < C; cb += 16) {          // two-level summation: blocks of 16 channels (see conv133_kernel)
      float part[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) part[k] = 0.f;
      const int ce = cb + 16 < C ? cb + 16 : C;
      for (int c = cb; c < ce; ++c) {
        float a = 1.f, b = 0.f, sl = 1.f;
        if (scale) { a = scale[(long long)n * C + c]; b = shift[(long long)n * C + c]; sl = slope; }
        const float z = e2e::in_act(x[((long long)n * C + c) * spatial + v], a, b, sl);
#pragma unroll
        for (int k = 0; k < KB; ++k)
          if (k0 + k < K) part[k] = fmaf(w[(long long)(k0 + k) * C + c], z, part[k]);
      }
#pragma unroll
      for (int k = 0; k < KB; ++k) acc[k] += part[k];
    }
#pragma unroll
    for (int k = 0; k < KB; ++k)
      if (k0 + k < K) logits[((long long)n * K + k0 + k) * spatial + v] = acc[k];
  }
}

// dx[n,c,v] (+)= sum_k W[k,c] dlogits[n,k,v]
template <int KB>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                         float* __restrict__ dx, int accumulate, int C, int K,
                                                         long long spatial) {
  const int n = blockIdx.y;
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= spatial) return;
  for (int k0 = 0; k0 < K; k0 += KB) {
    float g[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) g[k] = (k0 + k < K) ? dl[((long long)n * K + k0 + k) * spatial + v] : 0.f;
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < KB; ++k)
        if (k0 + k < K) s = fmaf(w[(long long)(k0 + k) * C + c], g[k], s);
      float* dst = dx + ((long long)n * C + c) * spatial + v;
      if (accumulate || k0 > 0) *dst += s;
      else *dst = s;
    }
  }
}

// dW[k,c] = sum_{n,v} dlogits[n,k,v] z[n,c,v]: block = (voxel chunk, channel c); fp64 atomics into ws[K*C]
template <int KB>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float slope,
                                                         const float* __restrict__ dl, double* __restrict__ acc_out, int C,
                                                         int K, long long spatial, int B) {
  const int c = blockIdx.y;
  __shared__ double sh[4][KB];
  for (int k0 = 0; k0 < K; k0 += KB) {
    float acc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) acc[k] = 0.f;
    double dacc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) dacc[k] = 0.0;
    for (int n = 0; n < B; ++n) {
      float a = 1.f, b = 0.f, sl = 1.f;
      if (scale) { a = scale[(long long)n * C + c]; b = shift[(long long)n * C + c]; sl = slope; }
      const float* xp = x + ((long long)n * C + c) * spatial;
      int it = 0;
      for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < spatial; v += (long long)gridDim.x * 256) {
        const float z = e2e::in_act(xp[v], a, b, sl);
#pragma unroll
        for (int k = 0; k < KB; ++k)
          if (k0 + k < K) acc[k] = fmaf(dl[((long long)n * K + k0 + k) * spatial + v], z, acc[k]);
        if ((++it & 63) == 0) {
#pragma unroll
          for (int k = 0; k < KB; ++k) { dacc[k] += acc[k]; acc[k] = 0.f; }
        }
      }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const double t = e2e::wave_sum_d(dacc[k] + (double)acc[k]);
      if (lane == 0) sh[wave][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < KB && k0 + threadIdx.x < K)
      atomicAdd(&acc_out[(long long)(k0 + threadIdx.x) * C + c],
                sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    __syncthreads();
  }
}

// ---- float4 variants (spatial % 4 == 0): one thread = 4 consecutive voxels --------------------------------------------
template <int KB>
__global__ __launch_bounds__(256) void head_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope,
                                                          const float* __restrict__ w, float* __restrict__ logits, int C,
                                                          int K, long long spatial) {
  const int n = blockIdx.y;
  const long long v = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (v >= spatial) return;
  for (int k0 = 0; k0 < K; k0 += KB) {
    float acc[KB][4];
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
    for (int cb = 0; cb < C; cb += 16) {          // two-level summation: blocks of 16 channels (see conv133_kernel)
      float part[KB][4];
#pragma unroll
      for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[k][j] = 0.f;
      const int ce = cb + 16 < C ? cb + 16 : C;
#pragma unroll 8
      for (int c = cb; c < ce; ++c) {
        float a = 1.f, b = 0.f, sl = 1.f;
        if (scale) { a = scale[(long long)n * C + c]; b = shift[(long long)n * C + c]; sl = slope; }
        const float4 q = *reinterpret_cast<const float4*>(x + ((long long)n * C + c) * spatial + v);
        const float z[4] = {e2e::in_act(q.x, a, b, sl), e2e::in_act(q.y, a, b, sl), e2e::in_act(q.z, a, b, sl),
                            e2e::in_act(q.w, a, b, sl)};
#pragma unroll
        for (int k = 0; k < KB; ++k)
          if (k0 + k < K) {
            const float wk = w[(long long)(k0 + k) * C + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) part[k][j] = fmaf(wk, z[j], part[k][j]);
          }
      }
#pragma unroll
      for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] += part[k][j];
    }
#pragma unroll
    for (int k = 0; k < KB; ++k)
      if (k0 + k < K)
        *reinterpret_cast<float4*>(logits + ((long long)n * K + k0 + k) * spatial + v) =
            make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
  }
}

template <int KB>
__global__ __launch_bounds__(256) void head_dgrad_v4_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                            float* __restrict__ dx, int accumulate, int C, int K,
                                                            long long spatial) {
  const int n = blockIdx.y;
  const long long v = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (v >= spatial) return;
  for (int k0 = 0; k0 < K; k0 += KB) {
    float g[KB][4];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + k < K) q = *reinterpret_cast<const float4*>(dl + ((long long)n * K + k0 + k) * spatial + v);
      g[k][0] = q.x; g[k][1] = q.y; g[k][2] = q.z; g[k][3] = q.w;
    }
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KB; ++k)
        if (k0 + k < K) {
          const float wk = w[(long long)(k0 + k) * C + c];
#pragma unroll
          for (int j = 0; j < 4; ++j) s[j] = fmaf(wk, g[k][j], s[j]);
        }
      float4* dst = reinterpret_cast<float4*>(dx + ((long long)n * C + c) * spatial + v);
      float4 o = make_float4(s[0], s[1], s[2], s[3]);
      if (accumulate || k0 > 0) { const float4 t = *dst; o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
      *dst = o;
    }
  }
}

// dW[k,c]: block = (voxel chunk, group of CG channels): dlogits are re-read C / CG times instead of C times
template <int KB, int CG>
__global__ __launch_bounds__(256) void head_wgrad_v4_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float slope,
                                                            const float* __restrict__ dl, double* __restrict__ acc_out,
                                                            int C, int K, long long spatial, int B) {
  const int c0 = blockIdx.y * CG;
  __shared__ double sh[4][CG * KB];
  for (int k0 = 0; k0 < K; k0 += KB) {
    float acc[CG][KB];
#pragma unroll
    for (int i = 0; i < CG; ++i)
#pragma unroll
      for (int k = 0; k < KB; ++k) acc[i][k] = 0.f;
    for (int n = 0; n < B; ++n) {
      float a[CG], b[CG], sl[CG];
#pragma unroll
      for (int i = 0; i < CG; ++i) {
        a[i] = 1.f; b[i] = 0.f; sl[i] = 1.f;
        if (scale && c0 + i < C) { a[i] = scale[(long long)n * C + c0 + i]; b[i] = shift[(long long)n * C + c0 + i]; sl[i] = slope; }
      }
      for (long long v = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; v < spatial; v += (long long)gridDim.x * 1024) {
        float g[KB][4];
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k0 + k < K) q = *reinterpret_cast<const float4*>(dl + ((long long)n * K + k0 + k) * spatial + v);
          g[k][0] = q.x; g[k][1] = q.y; g[k][2] = q.z; g[k][3] = q.w;
        }
#pragma unroll
        for (int i = 0; i < CG; ++i) {
          if (c0 + i >= C) continue;
          const float4 q = *reinterpret_cast<const float4*>(x + ((long long)n * C + c0 + i) * spatial + v);
          const float z[4] = {e2e::in_act(q.x, a[i], b[i], sl[i]), e2e::in_act(q.y, a[i], b[i], sl[i]),
                              e2e::in_act(q.z, a[i], b[i], sl[i]), e2e::in_act(q.w, a[i], b[i], sl[i])};
#pragma unroll
          for (int k = 0; k < KB; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][k] = fmaf(g[k][j], z[j], acc[i][k]);
        }
      }
    }
    // (a thread accumulates a few hundred products in fp32; everything across threads is summed in fp64)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < CG; ++i)
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const double t = e2e::wave_sum_d((double)acc[i][k]);
        if (lane == 0) sh[wave][i * KB + k] = t;
      }
    __syncthreads();
    if (threadIdx.x < CG * KB) {
      const int i = threadIdx.x / KB, k = threadIdx.x - i * KB;
      if (c0 + i < C && k0 + k < K)
        atomicAdd(&acc_out[(long long)(k0 + k) * C + c0 + i],
                  sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    }
    __syncthreads();
  }
}

__global__ void d2f_kernel(const double* __restrict__ src, float* __restrict__ dst, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}

}  // namespace

extern "C" int e2e_maxpool_fwd(const float* x, const float* scale, const float* shift, float slope, float* y, int B,
                               int C, int D, int H, int W, int kd, int kh, int kw, void* stream) {
  E2E_REQUIRE(x && y, "maxpool_fwd: null pointer");
  E2E_REQUIRE(kd >= 1 && kh >= 1 && kw >= 1 && D >= kd && H >= kh && W >= kw, "maxpool_fwd: bad dims");
  const int Do = D / kd, Ho = H / kh, Wo = W / kw;
  if (kw == 2 && (W % 4) == 0) {      // (W % 4 == 0 => Wo even and every window row pair is a 16-byte aligned float4)
    dim3 grid2((unsigned)e2e::cdivll((long long)Do * Ho * (Wo / 2), 256), B * C);
    hipLaunchKernelGGL(maxpool_fwd_w2_kernel, grid2, dim3(256), 0, (hipStream_t)stream, x, scale, shift, slope, y, D, H, W, kd,
                       kh, Do, Ho, Wo);
    return e2e::check_launch("maxpool_fwd_w2_kernel");
  }
  dim3 grid((unsigned)e2e::cdivll((long long)Do * Ho * Wo, 256), B * C);
  hipLaunchKernelGGL(maxpool_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, scale, shift, slope, y, D, H, W, kd, kh,
                     kw, Do, Ho, Wo);
  return e2e::check_launch("maxpool_fwd_kernel");
}

// records per (n, c) that a fused launch writes; 0: this shape cannot carry the fused sums (cells outside every window, or the
// generic kernel)
extern "C" int e2e_maxpool_bwd_num_records(int D, int H, int W, int kd, int kh, int kw) {
  if (kd < 1 || kh < 1 || kd > 2 || kh > 2 || kw != 2 || (W % 4) != 0 || D % kd || H % kh) return 0;
  return (int)e2e::cdivll((long long)(D / kd) * (H / kh) * (W / 4), 256);
}

extern "C" int e2e_maxpool_bwd(const float* x, const float* scale, const float* shift, float slope, const float* dy,
                               float* dx, int accumulate, int B, int C, int D, int H, int W, int kd, int kh, int kw,
                               const float* mean, const float* rstd, double* tile_sums, void* stream) {
  E2E_REQUIRE(x && dy && dx, "maxpool_bwd: null pointer");
  E2E_REQUIRE(kd >= 1 && kh >= 1 && kw >= 1 && D >= kd && H >= kh && W >= kw, "maxpool_bwd: bad dims");
  E2E_REQUIRE(tile_sums == nullptr || (mean && rstd && scale && e2e_maxpool_bwd_num_records(D, H, W, kd, kh, kw) > 0),
              "maxpool_bwd: fused InstanceNorm-backward sums need a normalised source and a shape of e2e_maxpool_bwd_num_records");
  hipStream_t st = (hipStream_t)stream;
  const int Do = D / kd, Ho = H / kh, Wo = W / kw;
  if (!accumulate && (D % kd || H % kh || W % kw)) {   // cells outside every window receive no gradient
    e2e::zero_async(dx, (size_t)B * C * D * H * W * sizeof(float), st);
  }
  if (kw == 2 && (W % 4) == 0 && kd <= 2 && kh <= 2) {
    dim3 grid2((unsigned)e2e::cdivll((long long)Do * Ho * (Wo / 2), 256), B * C);
    hipLaunchKernelGGL(maxpool_bwd_w2_kernel, grid2, dim3(256), 0, st, x, scale, shift, slope, dy, dx, accumulate, D, H, W, kd, kh,
                       Do, Ho, Wo, mean, rstd, tile_sums);
    return e2e::check_launch("maxpool_bwd_w2_kernel");
  }
  dim3 grid((unsigned)e2e::cdivll((long long)Do * Ho * Wo, 256), B * C);
  hipLaunchKernelGGL(maxpool_bwd_kernel, grid, dim3(256), 0, st, x, scale, shift, slope, dy, dx, accumulate, D, H, W, kd, kh,
                     kw, Do, Ho, Wo);
  return e2e::check_launch("maxpool_bwd_kernel");
}

#define DISPATCH_KB(K, ...)                                   \
  if ((K) <= 4) { constexpr int KB = 4; __VA_ARGS__; }        \
  else if ((K) <= 8) { constexpr int KB = 8; __VA_ARGS__; }   \
  else { constexpr int KB = 16; __VA_ARGS__; }

extern "C" int e2e_head1x1_fwd(const float* x, const float* scale, const float* shift, float slope, const float* w,
                               float* logits, int B, int C, int K, long long spatial, void* stream) {
  E2E_REQUIRE(x && w && logits, "head1x1_fwd: null pointer");
  E2E_REQUIRE(B > 0 && C > 0 && K > 0 && spatial > 0, "head1x1_fwd: bad dims");
  if (spatial % 4 == 0) {
    dim3 grid4((unsigned)e2e::cdivll(spatial, 1024), B);
    DISPATCH_KB(K, hipLaunchKernelGGL((head_fwd_v4_kernel<KB>), grid4, dim3(256), 0, (hipStream_t)stream, x, scale, shift, slope, w,
                                      logits, C, K, spatial));
    return e2e::check_launch("head_fwd_v4_kernel");
  }
  dim3 grid((unsigned)e2e::cdivll(spatial, 256), B);
  DISPATCH_KB(K, hipLaunchKernelGGL((head_fwd_kernel<KB>), grid, dim3(256), 0, (hipStream_t)stream, x, scale, shift, slope, w,
                                    logits, C, K, spatial));
  return e2e::check_launch("head_fwd_kernel");
}

extern "C" int e2e_head1x1_dgrad(const float* dlogits, const float* w, float* dx, int accumulate, int B, int C, int K,
                                 long long spatial, void* stream) {
  E2E_REQUIRE(dlogits && w && dx, "head1x1_dgrad: null pointer");
  if (spatial % 4 == 0) {
    dim3 grid4((unsigned)e2e::cdivll(spatial, 1024), B);
    DISPATCH_KB(K, hipLaunchKernelGGL((head_dgrad_v4_kernel<KB>), grid4, dim3(256), 0, (hipStream_t)stream, dlogits, w, dx,
                                      accumulate, C, K, spatial));
    return e2e::check_launch("head_dgrad_v4_kernel");
  }
  dim3 grid((unsigned)e2e::cdivll(spatial, 256), B);
  DISPATCH_KB(K, hipLaunchKernelGGL((head_dgrad_kernel<KB>), grid, dim3(256), 0, (hipStream_t)stream, dlogits, w, dx,
                                    accumulate, C, K, spatial));
  return e2e::check_launch("head_dgrad_kernel");
}

extern "C" long long e2e_head1x1_wgrad_ws_bytes(int B, int C, int K, long long spatial) {
  (void)B; (void)spatial;
  return (long long)C * K * (long long)sizeof(double);
}

extern "C" int e2e_head1x1_wgrad(const float* x, const float* scale, const float* shift, float slope,
                                 const float* dlogits, float* dw, void* ws, int B, int C, int K, long long spatial,
                                 void* stream) {
  E2E_REQUIRE(x && dlogits && dw && ws, "head1x1_wgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  double* acc = reinterpret_cast<double*>(ws);
  e2e::zero_async(acc, (size_t)C * K * sizeof(double), st);
  long long blocks = e2e::cdivll(spatial, 256 * 16);
  if (blocks > 128) blocks = 128;
  if (blocks < 1) blocks = 1;
  if (spatial % 4 == 0 && K <= 16) {
    long long b4 = e2e::cdivll(spatial, 1024 * 8);
    if (b4 > 256) b4 = 256;
    if (b4 < 1) b4 = 1;
    dim3 grid4((unsigned)b4, e2e::cdiv(C, 8));
    if (K <= 4) hipLaunchKernelGGL((head_wgrad_v4_kernel<4, 8>), grid4, dim3(256), 0, st, x, scale, shift, slope, dlogits, acc, C, K, spatial, B);
    else hipLaunchKernelGGL((head_wgrad_v4_kernel<8, 8>), grid4, dim3(256), 0, st, x, scale, shift, slope, dlogits, acc, C, K, spatial, B);
  } else {
  dim3 grid((unsigned)blocks, C);
  DISPATCH_KB(K, hipLaunchKernelGGL((head_wgrad_kernel<KB>), grid, dim3(256), 0, st, x, scale, shift, slope, dlogits, acc, C, K,
                                    spatial, B));
  }
  hipLaunchKernelGGL(d2f_kernel, dim3((unsigned)e2e::cdivll((long long)C * K, 256)), dim3(256), 0, st, acc, dw, (long long)C * K);
  return e2e::check_launch("head1x1_wgrad");
}
