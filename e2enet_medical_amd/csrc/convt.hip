// K3: transposed convolution with kernel == stride (non-overlapping up-sampling), forward / data gradient /
// weight gradient, gfx950.  Reference: nn.ConvTranspose3d(Cin, Cout, k, k, bias=False) built at
// unetpp_d.py:521-522, DSFF-masked on (Cin, Cout) pairs (core_channel.py:324: names containing 'up').
//
//   y[n,o,kd*d+i,kh*h+j,kw*w+k] = sum_c z[n,c,d,h,w] * W[c,o,i,j,k],   z = lrelu(scale*x + shift)
//
// The op is a per-voxel [Cin] x [Cin, Cout*KT] product; its input tensor is small next to its output (1/8 of
// the voxels), so forward and data gradient are written in gather form straight from global memory (the
// re-reads of z / dy are L2 hits) with the weights as wave-uniform scalars and the DSFF liveness bits walked
// with scalar bit ops.  The dense weight gradient (huge reduction over voxels) runs on the fp32 MFMA.
#include "e2e_common.h"
#include <cstdlib>

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const f32x4_t __attribute__((address_space(1)))* gf4_p;

// flat voxel index inside one sample -> (d, h, w).  A sample has fewer than 2^31 voxels (checked by the entry points),
// so this is 32-bit unsigned arithmetic: the 64-bit `%` and `/` the index types would imply cost ~100 vector
// instructions each and used to dominate the staging code of these kernels.
__device__ __forceinline__ void decode_dhw(long long v, int W, int H, int& dv, int& hv, int& wv) {
  const unsigned u = (unsigned)v;
  const unsigned r = u / (unsigned)W;
  wv = (int)(u - r * (unsigned)W);
  const unsigned d = r / (unsigned)H;
  hv = (int)(r - d * (unsigned)H);
  dv = (int)d;
}

__device__ __forceinline__ void tap_ijk(int t, int kh, int kw, int& i, int& j, int& k) {
  k = t % kw;
  const int r = t / kw;
  j = r % kh;
  i = r / kh;
}

// ---- forward: one thread = VPL input voxels of one output channel --------------------------------------------
template <int KT, int VPL>
__global__ __launch_bounds__(256) void convT_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float slope,
                                                        const float* __restrict__ w, const unsigned* __restrict__ live,
                                                        float* __restrict__ y, int Cin, int Cout, int D, int H, int W,
                                                        int kd, int kh, int kw) {
  const int o = blockIdx.y, n = blockIdx.z;
  const long long spatial = (long long)D * H * W;
  const long long v0 = ((long long)blockIdx.x * 256 + threadIdx.x) * VPL;   // VPL consecutive voxels (same row if W % VPL == 0)
  float acc[VPL][KT];
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int t = 0; t < KT; ++t) acc[v][t] = 0.f;
  const int words = e2e::cdiv(Cin, 32);
  const bool vec_ok = (VPL == 1) || (spatial % VPL == 0);
  for (int wd = 0; wd < words; ++wd) {
    unsigned bits = live ? live[(long long)o * words + wd] : 0xffffffffu;
    const int remain = Cin - wd * 32;
    if (remain < 32) bits &= (1u << remain) - 1u;
    bits = __builtin_amdgcn_readfirstlane(bits);
    // two-level summation (as in conv133_kernel): the live planes of one 32-plane word form a partial sum that is flushed
    // into the outer accumulator -- one fp32 chain over up to 320 planes is noisier than the CPU path's blocked sum
    float part[VPL][KT];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
      for (int t = 0; t < KT; ++t) part[v][t] = 0.f;
    while (bits) {
      const int c = wd * 32 + __builtin_ctz(bits);
      bits &= bits - 1;
      const float* wp = w + ((long long)c * Cout + o) * KT;
      float wt[KT];
#pragma unroll
      for (int t = 0; t < KT; ++t) wt[t] = wp[t];
      float a = 1.f, b = 0.f, sl = 1.f;
      if (scale) { a = scale[(long long)n * Cin + c]; b = shift[(long long)n * Cin + c]; sl = slope; }
      const float* xp = x + ((long long)n * Cin + c) * spatial;
      float xv[VPL];
      if (VPL == 4 && vec_ok && v0 + 3 < spatial) {
        const float4 q = *reinterpret_cast<const float4*>(xp + v0);
        xv[0] = q.x; xv[1 % VPL] = q.y; xv[2 % VPL] = q.z; xv[3 % VPL] = q.w;
      } else {
#pragma unroll
        for (int v = 0; v < VPL; ++v) xv[v] = (v0 + v < spatial) ? xp[v0 + v] : 0.f;
      }
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const float z = e2e::in_act(xv[v], a, b, sl);
#pragma unroll
        for (int t = 0; t < KT; ++t) part[v][t] = fmaf(wt[t], z, part[v][t]);
      }
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
      for (int t = 0; t < KT; ++t) acc[v][t] += part[v][t];
  }
  const int Ho = H * kh, Wo = W * kw;
  float* yp = y + ((long long)n * Cout + o) * spatial * KT;
  if (VPL == 2 && kw == 2 && (W % 2) == 0 && v0 + 1 < spatial) {
    int dv, hv, wv;
    decode_dhw(v0, W, H, dv, hv, wv);
#pragma unroll
    for (int t = 0; t < KT; t += 2) {
      int i, j, k;
      tap_ijk(t, kh, kw, i, j, k);
      *reinterpret_cast<float4*>(yp + ((long long)(dv * kd + i) * Ho + (hv * kh + j)) * Wo + wv * 2) =
          make_float4(acc[0][t], acc[0][(t + 1) % KT], acc[1 % VPL][t], acc[1 % VPL][(t + 1) % KT]);
    }
    return;
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const long long vi = v0 + v;
    if (vi >= spatial) continue;
    int dv, hv, wv;
    decode_dhw(vi, W, H, dv, hv, wv);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      int i, j, k;
      tap_ijk(t, kh, kw, i, j, k);
      yp[((long long)(dv * kd + i) * Ho + (hv * kh + j)) * Wo + (wv * kw + k)] = acc[v][t];
    }
  }
}

// ---- data gradient: one thread = one input voxel of one input channel ---------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void convT_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          const unsigned* __restrict__ live_t, float* __restrict__ dx,
                                                          int accumulate, int Cin, int Cout, int D, int H, int W, int kd,
                                                          int kh, int kw) {
  const int c = blockIdx.y, n = blockIdx.z;
  const long long spatial = (long long)D * H * W;
  const long long vi = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool in = vi < spatial;
  const long long vv = in ? vi : 0;
  int dv, hv, wv;
  decode_dhw(vv, W, H, dv, hv, wv);
  const int Ho = H * kh, Wo = W * kw;
  long long offs[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    int i, j, k;
    tap_ijk(t, kh, kw, i, j, k);
    offs[t] = ((long long)(dv * kd + i) * Ho + (hv * kh + j)) * Wo + (wv * kw + k);
  }
  float acc = 0.f;
  const int words = e2e::cdiv(Cout, 32);
  for (int wd = 0; wd < words; ++wd) {
    unsigned bits = live_t ? live_t[(long long)c * words + wd] : 0xffffffffu;
    const int remain = Cout - wd * 32;
    if (remain < 32) bits &= (1u << remain) - 1u;
    bits = __builtin_amdgcn_readfirstlane(bits);
    while (bits) {
      const int o = wd * 32 + __builtin_ctz(bits);
      bits &= bits - 1;
      const float* wp = w + ((long long)c * Cout + o) * KT;
      const float* dyp = dy + ((long long)n * Cout + o) * spatial * KT;
#pragma unroll
      for (int t = 0; t < KT; ++t) acc = fmaf(wp[t], dyp[offs[t]], acc);
    }
  }
  if (in) {
    float* dst = dx + ((long long)n * Cin + c) * spatial + vi;
    if (accumulate) *dst += acc;
    else *dst = acc;
  }
}

// ---- weight gradient (dense) on the fp32 MFMA ------------------------------------------------------------------
// dW[c, o, t] = sum_{n, v} z[n, c, v] * dy[n, o, out(v, t)].
// One workgroup = (32 input channels) x (32 output channels) x all KT taps over a chunk of voxel tiles.
// Wave (ch, oh) owns the 16 x 16 sub-block (c half, o half): KT accumulator tiles of v_mfma_f32_16x16x4_f32
// (exact fp32 fma chain, so the sum is a plain fp32 accumulation like the reference's).  The reduction index is
// the voxel: it lives inside the MFMA K dimension and in the loop, so no cross-lane reduction is needed.
// Partial sums per chunk go to a slab; a second kernel adds the slabs in fixed order (deterministic).
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WG_TPX = 64;                 // voxels staged per step
constexpr int WG_ZS = WG_TPX + 2;          // channel stride in LDS, == 2 (mod 32): conflict-free A/B fragment reads

template <int KT>
__global__ __launch_bounds__(256) void convT_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope,
                                                          const float* __restrict__ dy, float* __restrict__ slab, int B,
                                                          int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw,
                                                          int tiles_per_chunk, int nchunks) {
  __shared__ float zs[32 * WG_ZS];
  __shared__ float ds[32 * KT * WG_ZS];
  const int chunk = blockIdx.x;
  const int cblocks = e2e::cdiv(Cin, 32);
  const int cb = blockIdx.y % cblocks, ob = blockIdx.y / cblocks;
  const long long spatial = (long long)D * H * W;
  const long long tiles_per_n = e2e::cdivll(spatial, WG_TPX);
  const long long total_tiles = tiles_per_n * B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ch = wave & 1, oh = wave >> 1;
  const int Ho = H * kh, Wo = W * kw;

  f32x4 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int ti = 0; ti < tiles_per_chunk; ++ti) {
    const long long tile = (long long)chunk * tiles_per_chunk + ti;
    if (tile >= total_tiles) break;
    const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
    const long long vbase = (tile - (long long)n * tiles_per_n) * WG_TPX;
    // stage z[32 c][64 v]
    for (int idx = tid; idx < 32 * WG_TPX; idx += 256) {
      const int cl = idx / WG_TPX, pv = idx - cl * WG_TPX;
      const int c = cb * 32 + cl;
      const long long vi = vbase + pv;
      float val = 0.f;
      if (c < Cin && vi < spatial) {
        val = x[((long long)n * Cin + c) * spatial + vi];
        if (scale) val = e2e::in_act(val, scale[(long long)n * Cin + c], shift[(long long)n * Cin + c], slope);
      }
      zs[cl * WG_ZS + pv] = val;
    }
    // stage dy[32 o][KT][64 v]
    for (int idx = tid; idx < 32 * KT * WG_TPX; idx += 256) {
      const int pv = idx % WG_TPX;
      const int rest = idx / WG_TPX;
      const int t = rest % KT, ol = rest / KT;
      const int o = ob * 32 + ol;
      const long long vi = vbase + pv;
      float val = 0.f;
      if (o < Cout && vi < spatial) {
        int dv, hv, wv;
        decode_dhw(vi, W, H, dv, hv, wv);
        int i, j, k;
        tap_ijk(t, kh, kw, i, j, k);
        val = dy[((long long)n * Cout + o) * spatial * KT + ((long long)(dv * kd + i) * Ho + (hv * kh + j)) * Wo + (wv * kw + k)];
      }
      ds[(ol * KT + t) * WG_ZS + pv] = val;
    }
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    const float* ap = zs + (ch * 16 + li) * WG_ZS + lk;
    const float* bp = ds + ((oh * 16 + li) * KT) * WG_ZS + lk;
#pragma unroll 4
    for (int k0 = 0; k0 < WG_TPX; k0 += 4) {
      const float a = ap[k0];
#pragma unroll
      for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[t * WG_ZS + k0], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
  // D[i = c][j = o]: col = lane & 15 -> o, row = (lane >> 4) * 4 + reg -> c
  float* sp = slab + (long long)chunk * Cin * Cout * KT;
  const int o = ob * 32 + oh * 16 + (lane & 15);
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = cb * 32 + ch * 16 + (lane >> 4) * 4 + r;
      if (c < Cin && o < Cout) sp[((long long)c * Cout + o) * KT + t] = acc[t][r];
    }
}

// ---- v2 (kw == 2, even W): software-pipelined staging ---------------------------------------------------------------
// The op is HBM-bound when every byte is read once (25.6 FLOP/B at 64 -> 32 channels), so the workgroup covers NCB
// blocks of 32 input channels against one block of 32 output channels and stages the dy tile once for all of them.
// A tile = 64 consecutive input voxels.  dy arrives as aligned float4 = (k=0,1) x (voxel pair) of one output row and
// is scattered to an LDS image [tap][o][voxel] whose channel stride == 2 (mod 32): conflict-free B fragments.  Loads
// of tile t+1 are issued into registers before the MFMA phase of tile t and committed after it.

template <int KDH, int NCB>     // KDH = kd * kh (output rows per input voxel row), KT = 2 * KDH
__global__ __launch_bounds__(256 * NCB) void convT_wgrad_v2_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, float slope,
                                                                   const float* __restrict__ dy, float* __restrict__ slab,
                                                                   int B, int Cin, int Cout, int D, int H, int W, int kd, int kh,
                                                                   int tiles_per_chunk, int cgroups) {
  constexpr int KT = 2 * KDH;
  constexpr int TS = WG_TPX + 2;                       // 66 == 2 (mod 32)
  constexpr int YCW = 32 / (4 * NCB);                  // dy channels staged per wave
  constexpr int YIT = KDH * 32 / 64;                   // wave iterations per dy channel (KDH rows x 32 voxel pairs)
  static_assert(KDH == 2 || KDH == 4, "kd*kh in {2,4}");
  __shared__ __attribute__((aligned(16))) float zs[NCB * 32 * TS];
  __shared__ __attribute__((aligned(16))) float ds[KT * 32 * TS];

  const int chunk = blockIdx.x;
  const int cg = blockIdx.y % cgroups, ob = blockIdx.y / cgroups;
  const long long spatial = (long long)D * H * W;
  const long long tiles_per_n = e2e::cdivll(spatial, WG_TPX);
  const long long total_tiles = tiles_per_n * B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cbl = wave >> 2, ch = wave & 1, oh = (wave >> 1) & 1;
  const int Ho = H * kh, Wo = W * 2;
  const long long ospatial = spatial * KT;
  const int cbase = cg * NCB * 32;

  f32x4 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long tile_lo = (long long)chunk * tiles_per_chunk;
  long long tile_hi = tile_lo + tiles_per_chunk;
  if (tile_hi > total_tiles) tile_hi = total_tiles;

  // z: each wave stages 8 channels x 64 voxels = 128 float4 -> 2 iterations; lane -> (channel k = g / 16, group g % 16)
  f32x4_t vz[2], vy[YCW][YIT];
  float za[2], zb[2];
  const bool row_tiles = (W % WG_TPX) == 0;            // => spatial % 64 == 0 as well: no ragged last tile
  long long yoff[YIT];
#pragma unroll
  for (int it = 0; it < YIT; ++it) {
    const int g = lane + 64 * it;
    const int rr = g >> 5, vp = g & 31;
    const int i = rr / kh, j = rr - i * kh;
    yoff[it] = ((long long)i * Ho + j) * Wo + 4 * vp;
  }
  auto prefetch = [&](long long tile) {
    const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
    const long long vbase = (tile - (long long)n * tiles_per_n) * WG_TPX;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int g = lane + 64 * it;
      const int c = cbase + wave * 8 + (g >> 4);
      const long long vi = vbase + (g & 15) * 4;
      const bool ok = c < Cin && vi + 3 < spatial;
      const long long off = ok ? ((long long)n * Cin + c) * spatial + vi : 0;
      vz[it] = *reinterpret_cast<gf4_p>((gfloat_p)x + off);
      za[it] = 1.f; zb[it] = 0.f;
      if (scale != nullptr && c < Cin) { za[it] = scale[(long long)n * Cin + c]; zb[it] = shift[(long long)n * Cin + c]; }
    }
    if (row_tiles) {      // the 64-voxel tile lies inside one input row: wave-uniform base + per-thread constant offsets
      int dv, hv, w0;
      decode_dhw(vbase, W, H, dv, hv, w0);
      const long long tbase = ((long long)n * Cout + ob * 32 + wave * YCW) * ospatial +
                              ((long long)dv * kd * Ho + (long long)hv * kh) * Wo + 2 * w0;
#pragma unroll
      for (int k = 0; k < YCW; ++k)
#pragma unroll
        for (int it = 0; it < YIT; ++it) {
          const bool ok = ob * 32 + wave * YCW + k < Cout;
          vy[k][it] = *reinterpret_cast<gf4_p>((gfloat_p)dy + (ok ? tbase + k * ospatial + yoff[it] : 0));
        }
      return;
    }
#pragma unroll
    for (int k = 0; k < YCW; ++k) {
      const int o = ob * 32 + wave * YCW + k;
#pragma unroll
      for (int it = 0; it < YIT; ++it) {
        const int g = lane + 64 * it;                 // (row rr = g / 32, voxel pair vp = g % 32)
        const int rr = g >> 5, vp = g & 31;
        const long long vi = vbase + 2 * vp;
        const bool ok = o < Cout && vi + 1 < spatial;
        long long off = 0;
        if (ok) {
          int dv, hv, wv;
          decode_dhw(vi, W, H, dv, hv, wv);
          const int i = rr / kh, j = rr - i * kh;
          off = ((long long)n * Cout + o) * ospatial + ((long long)(dv * kd + i) * Ho + (hv * kh + j)) * Wo + 2 * wv;
        }
        vy[k][it] = *reinterpret_cast<gf4_p>((gfloat_p)dy + off);
      }
    }
  };
  auto commit = [&](long long tile) {
    const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
    const long long vbase = (tile - (long long)n * tiles_per_n) * WG_TPX;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int g = lane + 64 * it;
      const int cl = wave * 8 + (g >> 4);
      const long long vi = vbase + (g & 15) * 4;
      const bool ok = cbase + cl < Cin && vi + 3 < spatial;
      float2* dst = reinterpret_cast<float2*>(zs + cl * TS + (g & 15) * 4);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = vz[it][j];
        if (scale != nullptr) t = e2e::in_act(t, za[it], zb[it], slope);
        v[j] = ok ? t : 0.f;
      }
      dst[0] = make_float2(v[0], v[1]);
      dst[1] = make_float2(v[2], v[3]);
    }
#pragma unroll
    for (int k = 0; k < YCW; ++k) {
      const int ol = wave * YCW + k;
#pragma unroll
      for (int it = 0; it < YIT; ++it) {
        const int g = lane + 64 * it;
        const int rr = g >> 5, vp = g & 31;
        const bool ok = ob * 32 + ol < Cout && vbase + 2 * vp + 1 < spatial;
        const f32x4_t q = vy[k][it];                  // (k=0,v) (k=1,v) (k=0,v+1) (k=1,v+1)
        float2* d0 = reinterpret_cast<float2*>(ds + ((rr * 2 + 0) * 32 + ol) * TS + 2 * vp);
        float2* d1 = reinterpret_cast<float2*>(ds + ((rr * 2 + 1) * 32 + ol) * TS + 2 * vp);
        *d0 = ok ? make_float2(q[0], q[2]) : make_float2(0.f, 0.f);
        *d1 = ok ? make_float2(q[1], q[3]) : make_float2(0.f, 0.f);
      }
    }
  };

  if (tile_lo < tile_hi) {
    prefetch(tile_lo);
    for (long long tile = tile_lo; tile < tile_hi; ++tile) {
      commit(tile);
      __syncthreads();
      if (tile + 1 < tile_hi) prefetch(tile + 1);
      const int li = lane & 15, lk = lane >> 4;
      const float* ap = zs + (cbl * 32 + ch * 16 + li) * TS + lk;
      const float* bp = ds + (oh * 16 + li) * TS + lk;
#pragma unroll 4
      for (int k0 = 0; k0 < WG_TPX; k0 += 4) {
        const float a = ap[k0];
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[t * 32 * TS + k0], acc[t], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  // D[i = c][j = o]: col = lane & 15 -> o, row = (lane >> 4) * 4 + reg -> c
  float* sp = slab + (long long)chunk * Cin * Cout * KT;
  const int o = ob * 32 + oh * 16 + (lane & 15);
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = cbase + cbl * 32 + ch * 16 + (lane >> 4) * 4 + r;
      if (c < Cin && o < Cout) sp[((long long)c * Cout + o) * KT + t] = acc[t][r];
    }
}

// ---- v3 of the weight gradient (kw == 2, even W): the v2 pipeline on the bf16 matrix pipe with fp32-exact operands -------
// v2 is bound by its fp32 MFMAs (128 x 32 cycles per 64-voxel tile and wave, two waves per SIMD: 0.125 ms of a 0.25 ms
// launch at 64 -> 32 @64^3).  Both operands are split into three bf16 pieces when they are staged ([piece][row][64 voxels]
// bf16, row stride 144 B); the reduction dimension (voxels) is contiguous in both images, so an A / B fragment is one
// ds_read_b128; 96 v_mfma_f32_16x16x32_bf16 x 16 cycles per tile and wave.  The split3 / pack_hi16 helpers and the data
// gradient of the same scheme follow below (convT_dgrad_bf3_kernel); they are declared here.
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float v, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, v);
  const float r1 = v - __builtin_bit_cast(float, h & 0xffff0000u);            // exact
  m = __builtin_bit_cast(unsigned, r1);
  const float r2 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);           // exact, <= 8 significant bits
  l = __builtin_bit_cast(unsigned, r2);
}
// (S1 >> 16) | (S0 & 0xffff0000): the bf16 (truncated) pieces of two values in one word, `lo` in the low half
__device__ __forceinline__ unsigned pack_hi16(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// ---- round 6: the same three GEMMs with fp16 TWO-piece operands (NP = 2: three products per fp32 product instead of six; NP = 3
// keeps the bf16 three-piece form).  x = hi + lo, hi = rn16(x), lo = rn16(x - hi): 11 + 11 significant bits, products lo*hi +
// hi*lo + hi*hi through v_mfma_f32_16x16x32_f16, fp32 accumulation (conv133_mm.hip, conv133_wgrad_bf3.hip; numerics gate:
// tests/test_gpu_ops.py::test_split_operand_products_vs_fp64).  fp16 has 5 exponent bits, so every operand is moved into range by
// an exact power of two taken from a device word (bit pattern of a bound of max |operand|): the activations from the bound
// e2e_conv133_input_ranges derives from the producer's InstanceNorm parameters, the weights from their measured maximum, dy from
// max |dy of the consuming conv| x the L1 norm of that conv's weights over this tensor's channels.  Without the words a launch
// stays on the bf16 form (8 exponent bits: no range to manage).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_ct __attribute__((ext_vector_type(2)));
typedef float f32x2_ct __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_pair(float a, float b, unsigned& hw, unsigned& lw) {
  const f16x2_ct h2 = __builtin_convertvector((f32x2_ct{a, b}), f16x2_ct);
  hw = __builtin_bit_cast(unsigned, h2);
  // lo = rn16(v - hi) as ONE mixed-precision FMA per value: fma(hi as f16, -1, v) is exact in fp32, rounded to fp16 into the low / high half
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lw) : "v"(hw), "v"(a), "v"(b));
}
// the NP packed words (two values each: `a` in the low half) of a value pair, piece p at dst + p * pstride
template <int NP>
__device__ __forceinline__ void store_pair(unsigned char* dst, int pstride, float a, float b) {
  if constexpr (NP == 2) {
    unsigned hw, lw;
    split2_pair(a, b, hw, lw);
    *reinterpret_cast<unsigned*>(dst) = hw;
    *reinterpret_cast<unsigned*>(dst + pstride) = lw;
  } else {
    unsigned ha, ma, la, hb, mb, lb;
    split3(a, ha, ma, la);
    split3(b, hb, mb, lb);
    *reinterpret_cast<unsigned*>(dst) = pack_hi16(ha, hb);
    *reinterpret_cast<unsigned*>(dst + pstride) = pack_hi16(ma, mb);
    *reinterpret_cast<unsigned*>(dst + 2 * pstride) = pack_hi16(la, lb);
  }
}
// a fragment of eight values held in registers (weights): piece p -> out[p]
template <int NP>
__device__ __forceinline__ void split_frag(const float (&v)[8], bf16x8_t (&out)[NP]) {
  if constexpr (NP == 2) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2_pair(v[2 * j], v[2 * j + 1], hw[j], lw[j]);
    out[0] = __builtin_bit_cast(bf16x8_t, u32x4_t{hw[0], hw[1], hw[2], hw[3]});
    out[1] = __builtin_bit_cast(bf16x8_t, u32x4_t{lw[0], lw[1], lw[2], lw[3]});
  } else {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(v[j], h[j], m[j], l[j]);
    out[0] = __builtin_bit_cast(bf16x8_t, u32x4_t{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7])});
    out[1] = __builtin_bit_cast(bf16x8_t, u32x4_t{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7])});
    out[2] = __builtin_bit_cast(bf16x8_t, u32x4_t{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7])});
  }
}
// acc += A B rebuilt from the pieces: small terms first
template <int NP>
__device__ __forceinline__ f32x4 mma_pieces(const bf16x8_t (&a)[NP], const bf16x8_t (&b)[NP], f32x4 c) {
  if constexpr (NP == 2) {
    const f16x8_t a0 = __builtin_bit_cast(f16x8_t, a[0]), a1 = __builtin_bit_cast(f16x8_t, a[1]);
    const f16x8_t b0 = __builtin_bit_cast(f16x8_t, b[0]), b1 = __builtin_bit_cast(f16x8_t, b[1]);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c, 0, 0, 0);      // lo * hi
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c, 0, 0, 0);      // hi * lo
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c, 0, 0, 0);      // hi * hi
  } else {
    // lo*hi, mid*mid, hi*lo, then mid*hi, hi*mid, then hi*hi
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
  }
  return c;
}
// exponent k of the power of two that moves a tensor bounded by the product of the floats whose bit patterns are *wa and *wb (either
// may be null = 1) into [2^14, 2^15): k = 141 - E, clamped to +-100 (2^k and 2^-k normal); a zero bound takes E = 1, Inf / NaN propagate
__device__ __forceinline__ int ct_scale_exp(const unsigned* wa, const unsigned* wb) {
  float b = 1.f;
  if (wa != nullptr) b *= __builtin_bit_cast(float, __builtin_nontemporal_load(wa));
  if (wb != nullptr) b *= __builtin_bit_cast(float, __builtin_nontemporal_load(wb));
  int E = (int)((__builtin_bit_cast(unsigned, b) >> 23) & 0xffu);
  E = E < 1 ? 1 : E;
  const int k = 141 - E;
  return k > 100 ? 100 : (k < -100 ? -100 : k);
}
__device__ __forceinline__ float ct_pow2(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }

// TPX = input voxels per tile: 64 (one 138 KB workgroup per CU) or, for KDH = 4, 32: two 77 KB workgroups per CU, one converting
// and committing its tile while the other issues its matrix instructions (the phases of ONE workgroup are serial: commit, barrier,
// matrix phase, barrier, with a single LDS image).
template <int KDH, int NCB, int TPX, int NP>     // KDH = kd * kh (output rows per input voxel row), KT = 2 * KDH; NP pieces per operand
__global__ __launch_bounds__(256 * NCB, TPX == 32 ? 2 : 1) void convT_wgrad_bf3_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, float slope,
                                                                   const float* __restrict__ dy, float* __restrict__ slab,
                                                                   int B, int Cin, int Cout, int D, int H, int W, int kd, int kh,
                                                                   int tiles_per_chunk, int cgroups, const unsigned* __restrict__ x_word,
                                                                   const unsigned* __restrict__ dy_word_a, const unsigned* __restrict__ dy_word_b) {
  constexpr int KT = 2 * KDH;
  constexpr int RS = TPX * 2 + 16;                     // bytes per staged row (TPX voxels bf16 + 16: odd multiple of 16, conflict-free b128)
  constexpr int ZG = TPX / 4;                          // float4 groups per input channel row
  constexpr int ZIT = 8 * ZG / 64;                     // wave iterations over its 8 input channels
  constexpr int VP = TPX / 2;                          // voxel pairs (= float4 of dy) per output row
  constexpr int ZP = NCB * 32 * RS, YP = KT * 32 * RS;  // bytes per piece
  constexpr int YCW = 32 / (4 * NCB);                  // dy channels staged per wave
  constexpr int YIT = KDH * VP / 64;                   // wave iterations per dy channel (KDH rows x VP voxel pairs)
  static_assert((KDH == 2 || KDH == 4) && (TPX == 64 || (TPX == 32 && KDH == 4)), "kd*kh in {2,4}; 32-voxel tiles for kd*kh = 4");
  static_assert((TPX == 32 ? 2 : 1) * NP * (NCB * 32 + 2 * KDH * 32) * RS <= 163840, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char zs[NP * ZP];     // [piece][channel][voxel] bf16 / fp16
  __shared__ __attribute__((aligned(16))) unsigned char ds[NP * YP];     // [piece][tap][out channel][voxel]
  // fp16 two-piece form: operand scales 2^kx (activations) and 2^ky (dy), the slab un-scaled by the two exact factors
  const int kx = NP == 2 ? ct_scale_exp(x_word, nullptr) : 0, ky = NP == 2 ? ct_scale_exp(dy_word_a, dy_word_b) : 0;
  const float xsc = ct_pow2(kx), ysc = ct_pow2(ky), unx = ct_pow2(-kx), uny = ct_pow2(-ky);

  const int chunk = blockIdx.x;
  const int cg = blockIdx.y % cgroups, ob = blockIdx.y / cgroups;
  const long long spatial = (long long)D * H * W;
  const long long tiles_per_n = e2e::cdivll(spatial, TPX);
  const long long total_tiles = tiles_per_n * B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cbl = wave >> 2, ch = wave & 1, oh = (wave >> 1) & 1;
  const int Ho = H * kh, Wo = W * 2;
  const long long ospatial = spatial * KT;
  const int cbase = cg * NCB * 32;

  f32x4 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long tile_lo = (long long)chunk * tiles_per_chunk;
  long long tile_hi = tile_lo + tiles_per_chunk;
  if (tile_hi > total_tiles) tile_hi = total_tiles;

  // z: each wave stages 8 channels x 64 voxels = 128 float4 -> 2 iterations; lane -> (channel k = g / 16, group g % 16)
  f32x4_t vz[ZIT], vy[YCW][YIT];
  float za[ZIT], zb[ZIT];
  // (=> spatial % 64 == 0 as well: no ragged last tile; the fast path keeps per-lane byte offsets in 32 bits: 96 spatial x 4 B)
  const bool row_tiles = (W % TPX) == 0 && spatial < (1ll << 24);
  long long yoff[YIT];
#pragma unroll
  for (int it = 0; it < YIT; ++it) {
    const int g = lane + 64 * it;
    const int rr = g / VP, vp = g % VP;
    const int i = rr / kh, j = rr - i * kh;
    yoff[it] = ((long long)i * Ho + j) * Wo + 4 * vp;
  }
  // row tiles (W % 64 == 0): a tile's addresses are a wave-uniform base (scalar arithmetic) plus per-lane byte offsets that
  // never change; the normalise-on-load coefficients change only with the batch item.  (s_memtime stamps: issuing the ten
  // float4 loads of a tile with per-lane 64-bit index arithmetic, two integer divisions and four coefficient loads took
  // ~2500 cycles, as long as the tile's matrix phase.)
  unsigned zoffb[ZIT], yoffb[YCW][YIT];
  bool zcok[ZIT];
#pragma unroll
  for (int it = 0; it < ZIT; ++it) {
    const int g = lane + 64 * it;
    zcok[it] = cbase + wave * 8 + g / ZG < Cin;
    zoffb[it] = zcok[it] ? (unsigned)(((long long)(wave * 8 + g / ZG) * spatial + (g % ZG) * 4) * 4) : 0u;
  }
#pragma unroll
  for (int k = 0; k < YCW; ++k)
#pragma unroll
    for (int it = 0; it < YIT; ++it)
      yoffb[k][it] = (ob * 32 + wave * YCW + k < Cout) ? (unsigned)(((long long)k * ospatial + yoff[it]) * 4) : 0u;
  int cur_n = -1;
  auto prefetch = [&](long long tile) {
    const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
    const long long vbase = (tile - (long long)n * tiles_per_n) * TPX;
    if (row_tiles) {
      if (n != cur_n) {
        cur_n = n;
#pragma unroll
        for (int it = 0; it < ZIT; ++it) {
          const int c = cbase + wave * 8 + (lane + 64 * it) / ZG;
          za[it] = 1.f; zb[it] = 0.f;
          if (scale != nullptr && c < Cin) { za[it] = scale[(long long)n * Cin + c]; zb[it] = shift[(long long)n * Cin + c]; }
        }
      }
      int dv, hv, w0;
      decode_dhw(vbase, W, H, dv, hv, w0);
      const char* zb8 = reinterpret_cast<const char*>(x + ((long long)n * Cin + cbase) * spatial + vbase);
      const int ych = ob * 32 + wave * YCW < Cout ? ob * 32 + wave * YCW : 0;      // (waves past the last channel read channel 0 and drop it)
      const char* yb8 = reinterpret_cast<const char*>(dy + ((long long)n * Cout + ych) * ospatial +
                                                      ((long long)dv * kd * Ho + (long long)hv * kh) * Wo + 2 * w0);
#pragma unroll
      for (int it = 0; it < ZIT; ++it) vz[it] = *reinterpret_cast<gf4_p>((gfloat_p)(zb8 + zoffb[it]));
#pragma unroll
      for (int k = 0; k < YCW; ++k)
#pragma unroll
        for (int it = 0; it < YIT; ++it) vy[k][it] = *reinterpret_cast<gf4_p>((gfloat_p)(yb8 + yoffb[k][it]));
      return;
    }
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int g = lane + 64 * it;
      const int c = cbase + wave * 8 + g / ZG;
      const long long vi = vbase + (g % ZG) * 4;
      const bool ok = c < Cin && vi + 3 < spatial;
      const long long off = ok ? ((long long)n * Cin + c) * spatial + vi : 0;
      vz[it] = *reinterpret_cast<gf4_p>((gfloat_p)x + off);
      za[it] = 1.f; zb[it] = 0.f;
      if (scale != nullptr && c < Cin) { za[it] = scale[(long long)n * Cin + c]; zb[it] = shift[(long long)n * Cin + c]; }
    }
#pragma unroll
    for (int k = 0; k < YCW; ++k) {
      const int o = ob * 32 + wave * YCW + k;
#pragma unroll
      for (int it = 0; it < YIT; ++it) {
        const int g = lane + 64 * it;                 // (row rr = g / VP, voxel pair vp = g % VP)
        const int rr = g / VP, vp = g % VP;
        const long long vi = vbase + 2 * vp;
        const bool ok = o < Cout && vi + 1 < spatial;
        long long off = 0;
        if (ok) {
          int dv, hv, wv;
          decode_dhw(vi, W, H, dv, hv, wv);
          const int i = rr / kh, j = rr - i * kh;
          off = ((long long)n * Cout + o) * ospatial + ((long long)(dv * kd + i) * Ho + (hv * kh + j)) * Wo + 2 * wv;
        }
        vy[k][it] = *reinterpret_cast<gf4_p>((gfloat_p)dy + off);
      }
    }
  };
  auto commit = [&](long long tile) {
    const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
    const long long vbase = (tile - (long long)n * tiles_per_n) * TPX;
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int g = lane + 64 * it;
      const int cl = wave * 8 + g / ZG;
      const long long vi = vbase + (g % ZG) * 4;
      const bool ok = cbase + cl < Cin && vi + 3 < spatial;
      unsigned char* dst = zs + cl * RS + (g % ZG) * 8;
      float tz[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = vz[it][j];
        if (scale != nullptr) t = e2e::in_act(t, za[it], zb[it], slope);
        tz[j] = ok ? (NP == 2 ? t * xsc : t) : 0.f;
      }
      store_pair<NP>(dst, ZP, tz[0], tz[1]);
      store_pair<NP>(dst + 4, ZP, tz[2], tz[3]);
    }
#pragma unroll
    for (int k = 0; k < YCW; ++k) {
      const int ol = wave * YCW + k;
#pragma unroll
      for (int it = 0; it < YIT; ++it) {
        const int g = lane + 64 * it;
        const int rr = g / VP, vp = g % VP;
        const bool ok = ob * 32 + ol < Cout && vbase + 2 * vp + 1 < spatial;
        f32x4_t q = vy[k][it];                        // (k=0,v) (k=1,v) (k=0,v+1) (k=1,v+1)
        if (!ok) q = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (NP == 2) q *= ysc;
        unsigned char* d0 = ds + ((rr * 2 + 0) * 32 + ol) * RS + 4 * vp;      // tap (rr, 0): voxels 2 vp, 2 vp + 1
        unsigned char* d1 = ds + ((rr * 2 + 1) * 32 + ol) * RS + 4 * vp;
        store_pair<NP>(d0, YP, q[0], q[2]);
        store_pair<NP>(d1, YP, q[1], q[3]);
      }
    }
  };

#ifdef CT_DIAG
  unsigned long long stamp[16]; int ns = 0;
#define CSTAMP() do { if (ns < 16) stamp[ns++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CSTAMP() do {} while (0)
#endif
  if (tile_lo < tile_hi) {
    CSTAMP();
    prefetch(tile_lo);
    for (long long tile = tile_lo; tile < tile_hi; ++tile) {
      CSTAMP();
      commit(tile);
      CSTAMP();
      __syncthreads();
      CSTAMP();
      if (tile + 1 < tile_hi) prefetch(tile + 1);
      CSTAMP();
      const int li = lane & 15, lk = lane >> 4;
      // A: channel row, 8 consecutive voxels 32 kb + 8 lk ..; B: out channel row of tap t, the same voxels
      const unsigned char* ap = zs + (cbl * 32 + ch * 16 + li) * RS + lk * 16;
      const unsigned char* bp = ds + (oh * 16 + li) * RS + lk * 16;
#pragma unroll
      for (int kb = 0; kb < TPX / 32; ++kb) {
        bf16x8_t af[NP];
#pragma unroll
        for (int sp = 0; sp < NP; ++sp) af[sp] = *reinterpret_cast<const bf16x8_t*>(ap + sp * ZP + kb * 64);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          bf16x8_t bf[NP];
#pragma unroll
          for (int sp = 0; sp < NP; ++sp) bf[sp] = *reinterpret_cast<const bf16x8_t*>(bp + sp * YP + t * 32 * RS + kb * 64);
          acc[t] = mma_pieces<NP>(af, bf, acc[t]);
        }
      }
      CSTAMP();
      __syncthreads();
    }
  }
#ifdef CT_DIAG
  if (tid == 64 && blockIdx.y == 0 && (blockIdx.x == 100 || blockIdx.x == 101))
    printf("WG %d tiles %d | pre %u | t0: wait+commit %u bar %u pf %u mma %u | t1: bar %u commit %u bar %u pf %u mma %u | t2: bar %u commit %u bar %u\n", blockIdx.x, (int)(tile_hi - tile_lo),
           (unsigned)(stamp[1]-stamp[0]), (unsigned)(stamp[2]-stamp[1]), (unsigned)(stamp[3]-stamp[2]), (unsigned)(stamp[4]-stamp[3]), (unsigned)(stamp[5]-stamp[4]),
           (unsigned)(stamp[6]-stamp[5]), (unsigned)(stamp[7]-stamp[6]), (unsigned)(stamp[8]-stamp[7]), (unsigned)(stamp[9]-stamp[8]), (unsigned)(stamp[10]-stamp[9]),
           (unsigned)(stamp[11]-stamp[10]), (unsigned)(stamp[12]-stamp[11]), (unsigned)(stamp[13]-stamp[12]));
#endif
  // D[i = c][j = o]: col = lane & 15 -> o, row = (lane >> 4) * 4 + reg -> c
  float* sp = slab + (long long)chunk * Cin * Cout * KT;
  const int o = ob * 32 + oh * 16 + (lane & 15);
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = cbase + cbl * 32 + ch * 16 + (lane >> 4) * 4 + r;
      if (c < Cin && o < Cout) sp[((long long)c * Cout + o) * KT + t] = NP == 2 ? acc[t][r] * unx * uny : acc[t][r];
    }
}


// ---- data gradient v2 (kw == 2, even W): dy tile staged once in LDS ---------------------------------------------------
// dx[c, v] = sum_o sum_t W[c, o, t] * dy[o, out(v, t)].  A tile = 64 consecutive input voxels (lane = voxel); the dy
// values of 32 output channels x KT taps are loaded as aligned float4 and scattered to an LDS image [tap][o][voxel];
// each wave then walks its input channels and, per channel, the live output channels (scalar bit ops), reading dy
// from LDS conflict free.  dy is read from HBM once instead of once per live (c, o) pair.
template <int KDH>
__global__ __launch_bounds__(256) void convT_dgrad_v2_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             const unsigned* __restrict__ live_t, float* __restrict__ dx,
                                                             int accumulate, int B, int Cin, int Cout, int D, int H, int W,
                                                             int kd, int kh) {
  constexpr int KT = 2 * KDH;
  constexpr int OC = 32;                               // output channels per staged chunk
  constexpr int TS = 64;                               // lane = voxel: consecutive lanes, consecutive banks
  constexpr int NUY = OC * KDH * 32 / 256;             // float4 loads per thread per chunk
  __shared__ __attribute__((aligned(16))) float ds[KT * OC * TS];

  const long long spatial = (long long)D * H * W;
  const long long tiles_per_n = e2e::cdivll(spatial, 64);
  const long long tile = blockIdx.x;
  const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
  const long long vbase = (tile - (long long)n * tiles_per_n) * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Ho = H * kh, Wo = W * 2;
  const long long ospatial = spatial * KT;
  const int words = e2e::cdiv(Cout, 32);
  const long long vi_lane = vbase + lane;

  for (int o0 = 0; o0 < Cout; o0 += OC) {
    // ---- stage dy[o0 .. o0+32) for this tile: unit = (o, output row rr, voxel pair vp) ----
    f32x4_t v[NUY];
#pragma unroll
    for (int i = 0; i < NUY; ++i) {
      const int u = tid + i * 256;
      const int ol = u / (KDH * 32);
      const int rem = u - ol * (KDH * 32);
      const int rr = rem >> 5, vp = rem & 31;
      const long long vi = vbase + 2 * vp;
      const int o = o0 + ol;
      const bool ok = o < Cout && vi + 1 < spatial;
      long long off = 0;
      if (ok) {
        int dv, hv, wv;
        decode_dhw(vi, W, H, dv, hv, wv);
        const int ii = rr / kh, jj = rr - ii * kh;
        off = ((long long)n * Cout + o) * ospatial + ((long long)(dv * kd + ii) * Ho + (hv * kh + jj)) * Wo + 2 * wv;
      }
      v[i] = *reinterpret_cast<gf4_p>((gfloat_p)dy + off);
    }
    if (o0 > 0) __syncthreads();                       // previous chunk fully consumed
#pragma unroll
    for (int i = 0; i < NUY; ++i) {
      const int u = tid + i * 256;
      const int ol = u / (KDH * 32);
      const int rem = u - ol * (KDH * 32);
      const int rr = rem >> 5, vp = rem & 31;
      const bool ok = o0 + ol < Cout && vbase + 2 * vp + 1 < spatial;
      float2* d0 = reinterpret_cast<float2*>(ds + ((rr * 2 + 0) * OC + ol) * TS + 2 * vp);
      float2* d1 = reinterpret_cast<float2*>(ds + ((rr * 2 + 1) * OC + ol) * TS + 2 * vp);
      *d0 = ok ? make_float2(v[i][0], v[i][2]) : make_float2(0.f, 0.f);
      *d1 = ok ? make_float2(v[i][1], v[i][3]) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    // ---- compute: wave w owns input channels w, w+4, ... ----
    for (int c = wave; c < Cin; c += 4) {
      unsigned bits = live_t ? live_t[(long long)c * words + (o0 >> 5)] : 0xffffffffu;
      const int remain = Cout - o0;
      if (remain < 32) bits &= (1u << remain) - 1u;
      bits = __builtin_amdgcn_readfirstlane(bits);
      float acc = 0.f;
      while (bits) {
        const int ol = __builtin_ctz(bits);
        bits &= bits - 1;
        const float* wp = w + ((long long)c * Cout + o0 + ol) * KT;
        const float* dp = ds + ol * TS + lane;
#pragma unroll
        for (int t = 0; t < KT; ++t) acc = fmaf(wp[t], dp[t * OC * TS], acc);
      }
      if (vi_lane < spatial) {
        float* dst = dx + ((long long)n * Cin + c) * spatial + vi_lane;
        if (accumulate || o0 > 0) *dst += acc;
        else *dst = acc;
      }
    }
  }
}

// DSFF liveness of the (in channel c, out channel o) kernel from the bit tables of e2e_dsff_expand (null = all alive).  The
// GEMM kernels below multiply whole tiles, so a pruned kernel is dropped where the weight enters a fragment -- the result
// does not depend on pruned weights being exact zeros in memory.
__device__ __forceinline__ bool ct_alive_rows(const unsigned* live_t, int Cout, int c, int o) {   // live_t: [Cin][ceil(Cout/32)]
  return live_t == nullptr || ((live_t[(long long)c * ((Cout + 31) >> 5) + (o >> 5)] >> (o & 31)) & 1u);
}
__device__ __forceinline__ bool ct_alive_cols(const unsigned* live, int Cin, int c, int o) {      // live: [Cout][ceil(Cin/32)]
  return live == nullptr || ((live[(long long)o * ((Cin + 31) >> 5) + (c >> 5)] >> (c & 31)) & 1u);
}

// ---- data gradient v3 (kw == 2): dense GEMM on the fp32 matrix cores ----------------------------------------------------
// dx[c, v] = sum_{(t, o)} W[c, o, t] * dy[o, out(v, t)]:  M = 64 input channels per workgroup (16 per wave), N = 32
// consecutive input voxels per tile, K = 32 output channels x KT taps per chunk.  At DSFF density 0.2 the dense GEMM does
// 5x the useful FLOPs, but the sparse walk of v2 issues one scalar weight load and KT dependent LDS reads per live (c, o)
// pair with two waves per SIMD -- latency bound at ~4x the HBM time of this op -- while the matrix pipe runs the dense
// product in less time than that (dead kernels enter the fragments as zeros, so the result is the same sum).
//   * a wave keeps its 16 x K slice of W in registers (K/4 A fragments) for the whole run of tiles;
//   * the dy tile [K][32 voxels] is staged through LDS (row stride 48: the two k-rows of a 32-lane read group land on
//     disjoint banks), software pipelined: the float4 loads of the next tile are issued before the MFMA phase.
template <int KDH>
__global__ __launch_bounds__(256) void convT_dgrad_v3_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             const unsigned* __restrict__ live_t,
                                                             float* __restrict__ dx, int accumulate, int B, int Cin, int Cout,
                                                             int D, int H, int W, int kd, int kh, int tiles_per_wg) {
  constexpr int KT = 2 * KDH;
  constexpr int OC = 32, TV = 32, TS = 48;
  constexpr int K = OC * KT, KS = K / 4;               // k-steps per chunk
  constexpr int NUY = OC * KDH * (TV / 2) / 256;       // float4 loads per thread per tile
  __shared__ __attribute__((aligned(16))) float ds[K * TS];

  const long long spatial = (long long)D * H * W;
  const long long tiles_per_n = e2e::cdivll(spatial, TV);
  const long long total_tiles = tiles_per_n * B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int Ho = H * kh, Wo = W * 2;
  const long long ospatial = spatial * KT;
  const int cbase = blockIdx.y * 64 + wave * 16;
  const long long tile_lo = (long long)blockIdx.x * tiles_per_wg;
  long long tile_hi = tile_lo + tiles_per_wg;
  if (tile_hi > total_tiles) tile_hi = total_tiles;
  if (tile_lo >= tile_hi) return;

  f32x4_t v[NUY];
  // per-thread part of the dy offsets, constant over the tiles when a tile (32 consecutive voxels) never crosses a row
  // (W % 32 == 0): unit = (o, output row rr, voxel pair vp) -> o * ospatial + (ii * Ho + jj) * Wo + 4 vp; the tile adds
  // a wave-uniform base.  (The general path below re-derives (d, h, w) per unit with 64-bit divisions: ~1000 vector
  // instructions per tile and thread, more than the MFMA phase.)
  const bool row_tiles = (W % TV) == 0;
  long long uoff[NUY];
#pragma unroll
  for (int i = 0; i < NUY; ++i) {
    const int u = tid + i * 256;
    const int ol = u / (KDH * (TV / 2));
    const int rem = u - ol * (KDH * (TV / 2));
    const int rr = rem / (TV / 2), vp = rem - rr * (TV / 2);
    const int ii = rr / kh, jj = rr - ii * kh;
    uoff[i] = (long long)ol * ospatial + ((long long)ii * Ho + jj) * Wo + 4 * vp;
  }
  auto prefetch = [&](long long tile, int o0) {
    const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
    const long long vbase = (tile - (long long)n * tiles_per_n) * TV;
    if (row_tiles) {
      int dv, hv, w0;
      decode_dhw(vbase, W, H, dv, hv, w0);
      const long long tbase = ((long long)n * Cout + o0) * ospatial + ((long long)dv * kd * Ho + (long long)hv * kh) * Wo + 2 * w0;
#pragma unroll
      for (int i = 0; i < NUY; ++i) {
        const int u = tid + i * 256;
        const int ol = u / (KDH * (TV / 2));
        const bool ok = o0 + ol < Cout;                  // (spatial % 32 == 0 here: no ragged last tile)
        v[i] = *reinterpret_cast<gf4_p>((gfloat_p)dy + (ok ? tbase + uoff[i] : 0));
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NUY; ++i) {
      const int u = tid + i * 256;                       // unit = (o, output row rr, voxel pair vp)
      const int ol = u / (KDH * (TV / 2));
      const int rem = u - ol * (KDH * (TV / 2));
      const int rr = rem / (TV / 2), vp = rem - rr * (TV / 2);
      const long long vi = vbase + 2 * vp;
      const int o = o0 + ol;
      const bool ok = o < Cout && vi + 1 < spatial;
      long long off = 0;
      if (ok) {
        int dv, hv, wv;
        decode_dhw(vi, W, H, dv, hv, wv);
        const int ii = rr / kh, jj = rr - ii * kh;
        off = ((long long)n * Cout + o) * ospatial + ((long long)(dv * kd + ii) * Ho + (hv * kh + jj)) * Wo + 2 * wv;
      }
      v[i] = *reinterpret_cast<gf4_p>((gfloat_p)dy + off);
    }
  };
  auto commit = [&](long long tile, int o0) {
    const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
    const long long vbase = (tile - (long long)n * tiles_per_n) * TV;
#pragma unroll
    for (int i = 0; i < NUY; ++i) {
      const int u = tid + i * 256;
      const int ol = u / (KDH * (TV / 2));
      const int rem = u - ol * (KDH * (TV / 2));
      const int rr = rem / (TV / 2), vp = rem - rr * (TV / 2);
      const bool ok = o0 + ol < Cout && vbase + 2 * vp + 1 < spatial;
      float2* d0 = reinterpret_cast<float2*>(ds + ((rr * 2 + 0) * OC + ol) * TS + 2 * vp);
      float2* d1 = reinterpret_cast<float2*>(ds + ((rr * 2 + 1) * OC + ol) * TS + 2 * vp);
      *d0 = ok ? make_float2(v[i][0], v[i][2]) : make_float2(0.f, 0.f);
      *d1 = ok ? make_float2(v[i][1], v[i][3]) : make_float2(0.f, 0.f);
    }
  };

  for (int o0 = 0; o0 < Cout; o0 += OC) {
    // A fragments of this wave: row c = cbase + li, k = 4 s + lk = t * 32 + ol
    float afrag[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + lk;
      const int t = k / OC, ol = k - t * OC;
      const int c = cbase + li, o = o0 + ol;
      afrag[s] = (c < Cin && o < Cout && ct_alive_rows(live_t, Cout, c, o)) ? w[((long long)c * Cout + o) * KT + t] : 0.f;
    }
    prefetch(tile_lo, o0);
    for (long long tile = tile_lo; tile < tile_hi; ++tile) {
      commit(tile, o0);
      __syncthreads();
      if (tile + 1 < tile_hi) prefetch(tile + 1, o0);    // in flight during the MFMA phase
      f32x4 acc[TV / 16];
#pragma unroll
      for (int b = 0; b < TV / 16; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* bp = ds + lk * TS + li;
      // B fragments in groups of G k-steps, two register sets: the LDS reads of group g + 1 are in flight while the
      // MFMAs of group g issue (left to the compiler this loop was read -> wait -> 2 MFMAs, one LDS round trip per k-step)
      constexpr int G = 8, NB = TV / 16;
      float bfr[2][G][NB];
#pragma unroll
      for (int i = 0; i < G; ++i)
#pragma unroll
        for (int b = 0; b < NB; ++b) bfr[0][i][b] = bp[4 * i * TS + b * 16];
#pragma unroll
      for (int g = 0; g < KS / G; ++g) {
        if (g + 1 < KS / G) {
#pragma unroll
          for (int i = 0; i < G; ++i)
#pragma unroll
            for (int b = 0; b < NB; ++b) bfr[(g + 1) & 1][i][b] = bp[4 * ((g + 1) * G + i) * TS + b * 16];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[g * G + i], bfr[g & 1][i][b], acc[b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // D[i = c][j = v]: col = lane & 15 -> v, row = (lane >> 4) * 4 + reg -> c
      const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
      const long long vbase = (tile - (long long)n * tiles_per_n) * TV;
#pragma unroll
      for (int b = 0; b < TV / 16; ++b) {
        const long long vi = vbase + b * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = cbase + lk * 4 + r;
          if (c < Cin && vi < spatial) {
            float* dst = dx + ((long long)n * Cin + c) * spatial + vi;
            if (accumulate || o0 > 0) *dst += acc[b][r];
            else *dst = acc[b][r];
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---- forward v2 (kw == 2, W % 32 == 0, Cin <= 256): dense GEMM on the bf16 matrix pipe with fp32-exact operands ------------
// y[v, (o, t)] = sum_c z[v, c] W[c, (o, t)]:  M = 32 consecutive input voxels of a row per tile, K = Cin (NKB blocks of 32),
// N = Cout x KT output columns.  The gather kernel above re-reads the input once per output channel (through L2) and is
// bound by scalar weight loads and the per-voxel FMA chains (2.9 TB/s at 64 -> 32 @64^3); here
//   * a wave keeps the three-piece B fragments of its 8 / NKB column tiles (16 columns = 16 / KT output channels x KT taps)
//     in registers for its whole run of tiles (96 VGPRs);
//   * the z tile (normalise-on-load applied) is split and staged voxel-major, [piece][32 voxels][K bf16 + 16 B], once per
//     tile for all columns of the workgroup (4 waves x 8 / NKB x 16 columns); the next tile's loads fly during the matrix phase;
//   * D[voxel][column]: a lane holds 4 consecutive voxels of one (o, i, j, k) column; the k = 0 / 1 columns are neighbouring
//     lanes, so one DPP pair swap turns them into two aligned float4 stores of the interleaved output row.
template <int KDH, int NKB, int NP>
__global__ __launch_bounds__(256, 2) void convT_fwd_bf3_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float slope,
                                                               const float* __restrict__ w, const unsigned* __restrict__ live,
                                                               float* __restrict__ y, int B, int Cin,
                                                               int Cout, int D, int H, int W, int kd, int kh, int tiles_per_wg,
                                                               const unsigned* __restrict__ x_word, const unsigned* __restrict__ w_word) {
  constexpr int KT = 2 * KDH;
  constexpr int TV = 32;
  constexpr int K = NKB * 32;
  constexpr int NTW = 8 / NKB;                          // 16-column tiles per wave
  constexpr int RS = K * 2 + 16;                        // bytes per staged voxel and piece (odd multiple of 16)
  constexpr int PSZ = TV * RS;
  constexpr int NRD = K / 64;                           // staging rounds: 256 threads x (2 channels x 4 voxels)
  __shared__ __attribute__((aligned(16))) unsigned char zs[NP * PSZ];
  const int kx = NP == 2 ? ct_scale_exp(x_word, nullptr) : 0, kwt = NP == 2 ? ct_scale_exp(w_word, nullptr) : 0;
  const float xsc = ct_pow2(kx), wsc = ct_pow2(kwt), unx = ct_pow2(-kx), unw = ct_pow2(-kwt);

  const long long spatial = (long long)D * H * W;
  const long long tiles_per_n = spatial / TV;
  const long long total_tiles = tiles_per_n * B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int Ho = H * kh, Wo = W * 2;
  const long long ospatial = spatial * KT;
  const int ncols = Cout * KT;
  const int col0 = (blockIdx.y * 4 + wave) * NTW * 16;   // first column of this wave
  const long long tile_lo = (long long)blockIdx.x * tiles_per_wg;
  long long tile_hi = tile_lo + tiles_per_wg;
  if (tile_hi > total_tiles) tile_hi = total_tiles;
  if (tile_lo >= tile_hi) return;

  // B fragments: column n = col0 + 16 nt + li = (o, t), k = 32 kb + 8 lk + j = input channel
  bf16x8_t bfr[NTW][NKB][NP];
  long long coff[NTW];                                   // offset of the column's output row origin inside a sample
  bool cok[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int col = col0 + nt * 16 + li;
    cok[nt] = col < ncols;
    const int o = cok[nt] ? col / KT : 0, t = cok[nt] ? col - o * KT : 0;
    const int rr = t >> 1, kk = t & 1;
    const int ii = rr / kh, jj = rr - ii * kh;
    coff[nt] = (long long)o * ospatial + ((long long)ii * Ho + jj) * Wo + 4 * kk;   // + 4 kk: the second float4 of the pair
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = kb * 32 + lk * 8 + j;
        wv[j] = (cok[nt] && c < Cin && ct_alive_cols(live, Cin, c, o)) ? w[((long long)c * Cout + o) * KT + t] : 0.f;
        if (NP == 2) wv[j] *= wsc;
      }
      split_frag<NP>(wv, bfr[nt][kb]);
    }
  }

  // staging: thread -> channel pair cp = tid / 8 (+ 32 per round), voxel group vq = tid % 8 (4 voxels)
  const int cp = tid >> 3, vq = tid & 7;
  f32x4_t vx[NRD][2];
  float za[NRD][2], zb[NRD][2];
  auto prefetch = [&](long long tile) {
    const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
    const long long vbase = (tile - (long long)n * tiles_per_n) * TV;
#pragma unroll
    for (int r = 0; r < NRD; ++r)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = r * 64 + cp * 2 + e;
        const bool ok = c < Cin;
        vx[r][e] = *reinterpret_cast<gf4_p>((gfloat_p)x + (((long long)n * Cin + (ok ? c : 0)) * spatial + vbase + vq * 4));
        za[r][e] = 1.f; zb[r][e] = 0.f;
        if (scale != nullptr && ok) { za[r][e] = scale[(long long)n * Cin + c]; zb[r][e] = shift[(long long)n * Cin + c]; }
      }
  };
  auto commit = [&]() {
#pragma unroll
    for (int r = 0; r < NRD; ++r) {
      float tz[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool ok = r * 64 + cp * 2 + e < Cin;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float t = vx[r][e][q];
          if (scale != nullptr) t = e2e::in_act(t, za[r][e], zb[r][e], slope);
          tz[e][q] = ok ? (NP == 2 ? t * xsc : t) : 0.f;
        }
      }
      unsigned char* dst = zs + (vq * 4) * RS + (r * 64 + cp * 2) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) store_pair<NP>(dst + q * RS, PSZ, tz[0][q], tz[1][q]);
    }
  };

  prefetch(tile_lo);
  for (long long tile = tile_lo; tile < tile_hi; ++tile) {
    commit();
    __syncthreads();
    if (tile + 1 < tile_hi) prefetch(tile + 1);          // in flight during the matrix phase
    f32x4 acc[2][NTW];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A fragment: voxel 16 mt + li, channels 32 kb + 8 lk .. + 7
    const unsigned char* ap = zs + li * RS + lk * 16;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        bf16x8_t af[NP];
#pragma unroll
        for (int sp = 0; sp < NP; ++sp) af[sp] = *reinterpret_cast<const bf16x8_t*>(ap + sp * PSZ + mt * 16 * RS + kb * 64);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = mma_pieces<NP>(af, bfr[nt][kb], acc[mt][nt]);
      }
    // D: row (voxel) = 16 mt + 4 lk + i, column = lane & 15.  Lanes 2p, 2p + 1 hold taps k = 0, 1 of one (o, i, j): after the
    // pair swap lane k = 0 owns output floats [2 v .. 2 v + 3] and lane k = 1 the next four of the interleaved row.
    const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
    const long long vbase = (tile - (long long)n * tiles_per_n) * TV;
    int dv, hv, w0;
    decode_dhw(vbase, W, H, dv, hv, w0);
    float* yb = y + (long long)n * Cout * ospatial + ((long long)dv * kd * Ho + (long long)hv * kh) * Wo + 2 * w0;
    const bool odd = (lane & 1) != 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        f32x4 a = acc[mt][nt];
        if (NP == 2) a = a * unx * unw;
        const float s0 = odd ? a[0] : a[2], s1 = odd ? a[1] : a[3];
        // quad_perm [1, 0, 3, 2]: swap with the neighbouring lane
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xb1, 0xf, 0xf, true));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xb1, 0xf, 0xf, true));
        const f32x4_t o4 = odd ? f32x4_t{r0, a[2], r1, a[3]} : f32x4_t{a[0], r0, a[1], r1};
        if (cok[nt]) *reinterpret_cast<f32x4_t*>(yb + coff[nt] + 2 * (mt * 16 + lk * 4)) = o4;
      }
    __syncthreads();
  }
}

// ---- data gradient v4 (kw == 2, W % 32 == 0): the v3 GEMM on the bf16 matrix pipe with fp32-exact operands ----------------
// v3 is bound by its fp32 MFMAs (128 x 32 cycles per 32-voxel tile and wave: ~0.15 ms of a 0.31 ms launch at 64 -> 32 @64^3,
// the staging and the HBM stream not overlapped with them).  Every fp32 value is split without error into three bf16
// pieces (hi, mid, lo) and a product is rebuilt from the six leading cross terms, as in conv133_wgrad_bf3.hip / conv133_dense.hip
// (error class of an fp32 FMA chain): 96 v_mfma_f32_16x16x32_bf16 x 16 cycles per tile and wave.
//   * K order k = o * KT + t (the memory order of W[c][o][t]): a lane's A fragment (8 consecutive k of row c) is 32
//     contiguous bytes of W; a wave keeps the split fragments of its 16 x K slice in registers (K/32 x 3 x 4 VGPRs);
//   * the dy tile is staged voxel-major, [piece][32 voxels][K bf16 + 16 B] (row stride an odd multiple of 16 B:
//     conflict-free ds_read_b128 B fragments); a float4 of dy (two voxels x taps (2 rr, 2 rr + 1) of one output channel)
//     becomes one packed word per voxel and piece; the loads of the next tile are in flight during the matrix phase.
template <int KDH, int NP>
__global__ __launch_bounds__(256, 2) void convT_dgrad_bf3_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                 const unsigned* __restrict__ live_t, float* __restrict__ dx, int accumulate, int B, int Cin, int Cout,
                                                                 int D, int H, int W, int kd, int kh, int tiles_per_wg,
                                                                 const unsigned* __restrict__ w_word, const unsigned* __restrict__ dy_word_a,
                                                                 const unsigned* __restrict__ dy_word_b) {
  constexpr int KT = 2 * KDH;
  constexpr int OC = 32, TV = 32;
  constexpr int K = OC * KT, NKB = K / 32;             // 16x16x32 k-blocks per chunk
  constexpr int S = K * 2 + 16;                        // bytes per staged voxel and piece (528 / 272: odd multiples of 16)
  constexpr int PSZ = TV * S;
  constexpr int NUY = OC * KDH * (TV / 2) / 256;       // float4 loads per thread per tile
  __shared__ __attribute__((aligned(16))) unsigned char ds[NP * PSZ];
  const int kwt = NP == 2 ? ct_scale_exp(w_word, nullptr) : 0, ky = NP == 2 ? ct_scale_exp(dy_word_a, dy_word_b) : 0;
  const float wsc = ct_pow2(kwt), ysc = ct_pow2(ky), unw = ct_pow2(-kwt), uny = ct_pow2(-ky);

  const long long spatial = (long long)D * H * W;
  const long long tiles_per_n = spatial / TV;          // (W % 32 == 0: a tile is 32 consecutive voxels of one row)
  const long long total_tiles = tiles_per_n * B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int Ho = H * kh, Wo = W * 2;
  const long long ospatial = spatial * KT;
  const int cbase = blockIdx.y * 64 + wave * 16;
  const long long tile_lo = (long long)blockIdx.x * tiles_per_wg;
  long long tile_hi = tile_lo + tiles_per_wg;
  if (tile_hi > total_tiles) tile_hi = total_tiles;
  if (tile_lo >= tile_hi) return;

  f32x4_t v[NUY];
  long long uoff[NUY];
  int ulds[NUY];
#pragma unroll
  for (int i = 0; i < NUY; ++i) {
    const int u = tid + i * 256;                        // unit = (o, output row rr, voxel pair vp), vp fastest (coalesced float4s)
    const int ol = u / (KDH * (TV / 2));
    const int rem = u - ol * (KDH * (TV / 2));
    const int rr = rem / (TV / 2), vp = rem - rr * (TV / 2);
    const int ii = rr / kh, jj = rr - ii * kh;
    uoff[i] = (long long)ol * ospatial + ((long long)ii * Ho + jj) * Wo + 4 * vp;
    ulds[i] = (2 * vp) * S + (ol * KT + 2 * rr) * 2;
  }
  auto prefetch = [&](long long tile, int o0) {
    const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
    const long long vbase = (tile - (long long)n * tiles_per_n) * TV;
    int dv, hv, w0;
    decode_dhw(vbase, W, H, dv, hv, w0);
    const long long tbase = ((long long)n * Cout + o0) * ospatial + ((long long)dv * kd * Ho + (long long)hv * kh) * Wo + 2 * w0;
#pragma unroll
    for (int i = 0; i < NUY; ++i) {
      const int ol = (tid + i * 256) / (KDH * (TV / 2));
      v[i] = *reinterpret_cast<gf4_p>((gfloat_p)dy + (o0 + ol < Cout ? tbase + uoff[i] : 0));
    }
  };
  auto commit = [&](int o0) {
#pragma unroll
    for (int i = 0; i < NUY; ++i) {
      const int ol = (tid + i * 256) / (KDH * (TV / 2));
      const bool ok = o0 + ol < Cout;
      f32x4_t q = ok ? v[i] : f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (NP == 2) q *= ysc;
      unsigned char* d0 = ds + ulds[i];
      store_pair<NP>(d0, PSZ, q[0], q[1]);
      store_pair<NP>(d0 + S, PSZ, q[2], q[3]);
    }
  };

  for (int o0 = 0; o0 < Cout; o0 += OC) {
    // A fragments of this wave: row c = cbase + li, k-block kb, k = 32 kb + 8 lk + j  <->  W[c][o0 + k / KT][k % KT]
    bf16x8_t af[NKB][NP];
    {
      const int c = cbase + li;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const int k0 = kb * 32 + lk * 8;
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int o = o0 + (k0 + j) / KT;
          wv[j] = (c < Cin && o < Cout && ct_alive_rows(live_t, Cout, c, o)) ? w[((long long)c * Cout + o0) * KT + k0 + j] : 0.f;
          if (NP == 2) wv[j] *= wsc;
        }
        split_frag<NP>(wv, af[kb]);
      }
    }
    prefetch(tile_lo, o0);
    for (long long tile = tile_lo; tile < tile_hi; ++tile) {
      commit(o0);
      __syncthreads();
      if (tile + 1 < tile_hi) prefetch(tile + 1, o0);    // in flight during the matrix phase
      // accumulate mode / later channel chunks: the old dx values are requested now, not behind the matrix phase (a dependent
      // load -> add -> store per tile there cost 0.08 ms of a 0.26 ms launch at 64 -> 32 @64^3)
      const int n = (int)((unsigned)tile / (unsigned)tiles_per_n);
      const long long vbase = (tile - (long long)n * tiles_per_n) * TV;
      const bool rmw = accumulate || o0 > 0;
      float* dstp[TV / 16][4];
      float old[TV / 16][4];
#pragma unroll
      for (int b = 0; b < TV / 16; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = cbase + lk * 4 + r;
          dstp[b][r] = dx + ((long long)n * Cin + (c < Cin ? c : 0)) * spatial + vbase + b * 16 + li;
          old[b][r] = rmw ? *dstp[b][r] : 0.f;
        }
      f32x4 acc[TV / 16];
#pragma unroll
      for (int b = 0; b < TV / 16; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
      // B fragment: voxel 16 b + li, k = 32 kb + 8 lk .. + 7
      const unsigned char* bp = ds + li * S + lk * 16;
      bf16x8_t bf[2][TV / 16][NP];
#pragma unroll
      for (int b = 0; b < TV / 16; ++b)
#pragma unroll
        for (int sp = 0; sp < NP; ++sp) bf[0][b][sp] = *reinterpret_cast<const bf16x8_t*>(bp + sp * PSZ + b * 16 * S);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) {
#pragma unroll
          for (int b = 0; b < TV / 16; ++b)
#pragma unroll
            for (int sp = 0; sp < NP; ++sp)
              bf[(kb + 1) & 1][b][sp] = *reinterpret_cast<const bf16x8_t*>(bp + sp * PSZ + b * 16 * S + (kb + 1) * 64);
        }
#pragma unroll
        for (int b = 0; b < TV / 16; ++b) acc[b] = mma_pieces<NP>(af[kb], bf[kb & 1][b], acc[b]);
      }
      // D[i = c][j = v]: col = lane & 15 -> v, row = (lane >> 4) * 4 + reg -> c
#pragma unroll
      for (int b = 0; b < TV / 16; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (cbase + lk * 4 + r < Cin) *dstp[b][r] = (NP == 2 ? acc[b][r] * unw * uny : acc[b][r]) + old[b][r];
      __syncthreads();
    }
  }
}


__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                          long long numel, int nchunks) {
  // out[e] = sum_k slab[k][e]: 4 waves x 4 independent running sums per element, combined in a fixed order
  // (deterministic); the serial one-thread-per-element loop over up to 512 slabs was latency bound
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < numel) {
    for (int k = w; k < nchunks; k += 16) {
      s0 += slab[(long long)k * numel + e];
      if (k + 4 < nchunks) s1 += slab[(long long)(k + 4) * numel + e];
      if (k + 8 < nchunks) s2 += slab[(long long)(k + 8) * numel + e];
      if (k + 12 < nchunks) s3 += slab[(long long)(k + 12) * numel + e];
    }
  }
  part[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && e < numel) out[e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

inline int wgrad_chunks(long long total_tiles, int pairs, int* tiles_per_chunk, int per_cu = 1) {
  // aim at ~256 workgroups (one 84 KB workgroup per CU, equal work each; per_cu = 2: 512) and keep the slabs few: every chunk is
  // a slab the reduction has to read (1024 chunks cost 14 % more on the 64 -> 32 @64^3 layer); at least 8 tiles per chunk
  const int target = 256;
  long long want = (long long)target * per_cu / (pairs > 0 ? pairs : 1);
  if (want < 1) want = 1;
  long long tpc = e2e::cdivll(total_tiles, want);
  if (tpc < 8) tpc = 8;
  if (tpc > total_tiles) tpc = total_tiles;
  *tiles_per_chunk = (int)tpc;
  return (int)e2e::cdivll(total_tiles, tpc);
}

}  // namespace

#define DISPATCH_KT(KTV, ...)                  \
  switch (KTV) {                               \
    case 1: { constexpr int KT = 1; __VA_ARGS__; break; } \
    case 2: { constexpr int KT = 2; __VA_ARGS__; break; } \
    case 4: { constexpr int KT = 4; __VA_ARGS__; break; } \
    case 8: { constexpr int KT = 8; __VA_ARGS__; break; } \
    default: e2e::set_error("convT: kernel volume %d unsupported", KTV); return E2E_ERR_UNSUPPORTED; \
  }

static int check_k(int kd, int kh, int kw) {
  return (kd == 1 || kd == 2) && (kh == 1 || kh == 2) && (kw == 1 || kw == 2);
}

// fp16 two-piece operands (round 6) where the caller hands over the range words; E2E_CT_H2=0 keeps the bf16 three-piece form
static int ct_h2_env() { static const int v = getenv("E2E_CT_H2") ? atoi(getenv("E2E_CT_H2")) : 1; return v; }

extern "C" int e2e_convT_fwd(const float* x, const float* scale, const float* shift, float slope, const float* w,
                             const unsigned* live, float* y, int B, int Cin, int Cout, int D, int H, int W, int kd,
                             int kh, int kw, const unsigned* x_absmax, const unsigned* w_absmax, void* stream) {
  E2E_REQUIRE(x && w && y, "convT_fwd: null pointer");
  E2E_REQUIRE(check_k(kd, kh, kw), "convT_fwd: kernel must be in {1,2}^3");
  E2E_REQUIRE((long long)D * H * W * kd * kh * kw < (1ll << 31), "convT_fwd: a sample must have fewer than 2^31 output voxels");
  E2E_REQUIRE((scale == nullptr) == (shift == nullptr), "convT_fwd: scale/shift must both be given");
  hipStream_t st = (hipStream_t)stream;
  const long long spatial = (long long)D * H * W;
  const int kt = kd * kh * kw;
  static const int use_bf3 = getenv("E2E_CT_BF3") ? atoi(getenv("E2E_CT_BF3")) : 1;
  const int kdh = kd * kh;
  if (use_bf3 && kw == 2 && (kdh == 2 || kdh == 4) && W % 32 == 0 && Cin <= 256 && (spatial / 32) * B >= 256) {
    const int nkb = Cin <= 64 ? 2 : (Cin <= 128 ? 4 : 8);
    const int cols_per_wg = 4 * (8 / nkb) * 16;
    const int cgroups = e2e::cdiv(Cout * kt, cols_per_wg);
    const long long total_tiles = (spatial / 32) * B;
    const bool h2 = ct_h2_env() && x_absmax != nullptr && w_absmax != nullptr;
    // two 4-wave workgroups per CU, one round; the fp16 two-piece form with <= 64 input channels needs 164 registers: THREE
    // workgroups per CU, i.e. half again as many requests in flight on a kernel bound by the CU's request path (64 -> 32 @64^3 x 2:
    // 0.189 -> 0.165 ms; 1024 workgroups 0.193; the 128-channel form does not move: profiles/r06_kbench_convt.txt)
    long long wgs = ((h2 && nkb == 2) ? 768 : 512) / cgroups;
    if (wgs < 1) wgs = 1;
    int tpw = (int)e2e::cdivll(total_tiles, wgs);
    if (tpw < 4) tpw = 4;
    dim3 grid((unsigned)e2e::cdivll(total_tiles, tpw), cgroups);
    e2e::note_kernel("convT_fwd_%s<%d,%d> wgs=%u cgroups=%d tiles_per_wg=%d", h2 ? "h2" : "bf3", kdh, nkb, grid.x, cgroups, tpw);
#define LAUNCH_F3(KDH, NKB, NP) hipLaunchKernelGGL((convT_fwd_bf3_kernel<KDH, NKB, NP>), grid, dim3(256), 0, st, x, scale, shift, slope, w, live, y, \
                                                   B, Cin, Cout, D, H, W, kd, kh, tpw, x_absmax, w_absmax)
    if (h2) {
      if (kdh == 4) { if (nkb == 2) LAUNCH_F3(4, 2, 2); else if (nkb == 4) LAUNCH_F3(4, 4, 2); else LAUNCH_F3(4, 8, 2); }
      else { if (nkb == 2) LAUNCH_F3(2, 2, 2); else if (nkb == 4) LAUNCH_F3(2, 4, 2); else LAUNCH_F3(2, 8, 2); }
    } else {
      if (kdh == 4) { if (nkb == 2) LAUNCH_F3(4, 2, 3); else if (nkb == 4) LAUNCH_F3(4, 4, 3); else LAUNCH_F3(4, 8, 3); }
      else { if (nkb == 2) LAUNCH_F3(2, 2, 3); else if (nkb == 4) LAUNCH_F3(2, 4, 3); else LAUNCH_F3(2, 8, 3); }
    }
#undef LAUNCH_F3
    return e2e::check_launch("convT_fwd_bf3_kernel");
  }
  e2e::note_kernel("convT_fwd_gather<%d,%d>", kt, spatial >= 4096 ? (kt <= 4 ? 4 : 2) : 1);
  if (spatial >= 4096 && kt <= 4) {
    dim3 grid((unsigned)e2e::cdivll(spatial, 256 * 4), Cout, B);
    DISPATCH_KT(kt, hipLaunchKernelGGL((convT_fwd_kernel<KT, 4>), grid, dim3(256), 0, st, x, scale, shift, slope, w, live, y,
                                       Cin, Cout, D, H, W, kd, kh, kw));
  } else if (spatial >= 4096) {
    dim3 grid((unsigned)e2e::cdivll(spatial, 256 * 2), Cout, B);
    DISPATCH_KT(kt, hipLaunchKernelGGL((convT_fwd_kernel<KT, 2>), grid, dim3(256), 0, st, x, scale, shift, slope, w, live, y,
                                       Cin, Cout, D, H, W, kd, kh, kw));
  } else {
    dim3 grid((unsigned)e2e::cdivll(spatial, 256), Cout, B);
    DISPATCH_KT(kt, hipLaunchKernelGGL((convT_fwd_kernel<KT, 1>), grid, dim3(256), 0, st, x, scale, shift, slope, w, live, y,
                                       Cin, Cout, D, H, W, kd, kh, kw));
  }
  return e2e::check_launch("convT_fwd_kernel");
}

static int dg_min_tiles() {
  return 256;                                               // 256 tiles (16^3 x 2) still win over the gather kernel, 64 do not
}

extern "C" int e2e_convT_dgrad(const float* dy, const float* w, const unsigned* live_t, float* dx, int accumulate,
                               int B, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw, const unsigned* w_absmax,
                               const unsigned* dy_bound_a, const unsigned* dy_bound_b, void* stream) {
  E2E_REQUIRE(dy && w && dx, "convT_dgrad: null pointer");
  E2E_REQUIRE(check_k(kd, kh, kw), "convT_dgrad: kernel must be in {1,2}^3");
  E2E_REQUIRE((long long)D * H * W * kd * kh * kw < (1ll << 31), "convT_dgrad: a sample must have fewer than 2^31 output voxels");
  hipStream_t st = (hipStream_t)stream;
  const long long spatial = (long long)D * H * W;
  // v3 / v4 (dense GEMM on the matrix cores) for the large planes; E2E_CT_BF3=0 keeps the fp32 matrix instructions (v3)
  const int no_v3 = 0;
  static const int use_bf3 = getenv("E2E_CT_BF3") ? atoi(getenv("E2E_CT_BF3")) : 1;
  if (!no_v3 && kw == 2 && (kd * kh == 2 || kd * kh == 4) && (W % 2) == 0 && spatial % 4 == 0 && e2e::cdivll(spatial, 32) * B >= dg_min_tiles()) {
    const long long total_tiles = e2e::cdivll(spatial, 32) * B;
    const int cgroups = e2e::cdiv(Cin, 64);
    const int target = 512;                                 // two 4-wave workgroups fit a CU (218 VGPRs): exactly one round (768: 0.174 -> 0.203 ms)
    long long wgs = target / cgroups;
    if (wgs < 1) wgs = 1;
    int tpw = (int)e2e::cdivll(total_tiles, wgs);
    if (tpw < 4) tpw = 4;
    dim3 grid((unsigned)e2e::cdivll(total_tiles, tpw), cgroups);
    if (use_bf3 && W % 32 == 0) {
      const bool h2 = ct_h2_env() && w_absmax != nullptr && dy_bound_a != nullptr;
      e2e::note_kernel("convT_dgrad_%s<%d> wgs=%u cgroups=%d tiles_per_wg=%d", h2 ? "h2" : "bf3", kd * kh, grid.x, cgroups, tpw);
#define LAUNCH_D3(KDH, NP) hipLaunchKernelGGL((convT_dgrad_bf3_kernel<KDH, NP>), grid, dim3(256), 0, st, dy, w, live_t, dx, accumulate, B, Cin, Cout, \
                                              D, H, W, kd, kh, tpw, w_absmax, dy_bound_a, dy_bound_b)
      if (h2) { if (kd * kh == 4) LAUNCH_D3(4, 2); else LAUNCH_D3(2, 2); }
      else { if (kd * kh == 4) LAUNCH_D3(4, 3); else LAUNCH_D3(2, 3); }
#undef LAUNCH_D3
      return e2e::check_launch("convT_dgrad_bf3_kernel");
    }
    e2e::note_kernel("convT_dgrad_v3<%d> wgs=%u cgroups=%d tiles_per_wg=%d", kd * kh, grid.x, cgroups, tpw);
    if (kd * kh == 4)
      hipLaunchKernelGGL((convT_dgrad_v3_kernel<4>), grid, dim3(256), 0, st, dy, w, live_t, dx, accumulate, B, Cin, Cout, D, H, W, kd, kh, tpw);
    else
      hipLaunchKernelGGL((convT_dgrad_v3_kernel<2>), grid, dim3(256), 0, st, dy, w, live_t, dx, accumulate, B, Cin, Cout, D, H, W, kd, kh, tpw);
    return e2e::check_launch("convT_dgrad_v3_kernel");
  }
  // v2 needs enough 64-voxel tiles to fill the chip (one workgroup per tile); small planes keep the gather kernel
  if (kw == 2 && (kd * kh == 2 || kd * kh == 4) && (W % 2) == 0 && spatial % 4 == 0 && e2e::cdivll(spatial, 64) * B >= 1024) {
    const unsigned tiles = (unsigned)(e2e::cdivll(spatial, 64) * B);
    e2e::note_kernel("convT_dgrad_v2<%d> tiles=%u", kd * kh, tiles);
    if (kd * kh == 4)
      hipLaunchKernelGGL((convT_dgrad_v2_kernel<4>), dim3(tiles), dim3(256), 0, st, dy, w, live_t, dx, accumulate, B, Cin, Cout, D, H, W, kd, kh);
    else
      hipLaunchKernelGGL((convT_dgrad_v2_kernel<2>), dim3(tiles), dim3(256), 0, st, dy, w, live_t, dx, accumulate, B, Cin, Cout, D, H, W, kd, kh);
    return e2e::check_launch("convT_dgrad_v2_kernel");
  }
  dim3 grid((unsigned)e2e::cdivll(spatial, 256), Cin, B);
  e2e::note_kernel("convT_dgrad_gather<%d>", kd * kh * kw);
  DISPATCH_KT(kd * kh * kw, hipLaunchKernelGGL((convT_dgrad_kernel<KT>), grid, dim3(256), 0, st, dy, w, live_t, dx, accumulate,
                                               Cin, Cout, D, H, W, kd, kh, kw));
  return e2e::check_launch("convT_dgrad_kernel");
}

// (32-voxel tiles of the bf16x3 weight gradient, two workgroups per CU, were measured SLOWER than one workgroup per CU with 64-voxel
// tiles on every level but the 8^3 one -- 0.215 -> 0.27 ms at 64 -> 32 @64^3, profiles/r04_convt_wgrad_tiles.txt -- and removed in
// round 5)
static bool convT_use_v2(int D, int H, int W, int kd, int kh, int kw) {
  return kw == 2 && (kd * kh == 2 || kd * kh == 4) && (W % 2) == 0 && ((long long)D * H * W) % 4 == 0;
}
static int convT_v2_ncb(int Cin) { return Cin > 32 ? 2 : 1; }

extern "C" long long e2e_convT_wgrad_ws_bytes(int B, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw) {
  const long long total_tiles = e2e::cdivll((long long)D * H * W, WG_TPX) * B;
  int tpc;
  int pairs = e2e::cdiv(Cin, 32) * e2e::cdiv(Cout, 32);
  if (convT_use_v2(D, H, W, kd, kh, kw)) pairs = e2e::cdiv(Cin, 32 * convT_v2_ncb(Cin)) * e2e::cdiv(Cout, 32);
  int nchunks = wgrad_chunks(total_tiles, pairs, &tpc);
  return (long long)nchunks * Cin * Cout * kd * kh * kw * (long long)sizeof(float);
}

extern "C" int e2e_convT_wgrad(const float* x, const float* scale, const float* shift, float slope, const float* dy,
                               float* dw, void* ws, int B, int Cin, int Cout, int D, int H, int W, int kd, int kh,
                               int kw, const unsigned* x_absmax, const unsigned* dy_bound_a, const unsigned* dy_bound_b, void* stream) {
  E2E_REQUIRE(x && dy && dw && ws, "convT_wgrad: null pointer");
  E2E_REQUIRE(check_k(kd, kh, kw), "convT_wgrad: kernel must be in {1,2}^3");
  E2E_REQUIRE((long long)D * H * W * kd * kh * kw < (1ll << 31), "convT_wgrad: a sample must have fewer than 2^31 output voxels");
  hipStream_t st = (hipStream_t)stream;
  const long long total_tiles = e2e::cdivll((long long)D * H * W, WG_TPX) * B;
  int tpc;
  const int kt = kd * kh * kw;
  float* slab = reinterpret_cast<float*>(ws);
  const long long numel_all = (long long)Cin * Cout * kt;
  if (convT_use_v2(D, H, W, kd, kh, kw)) {
    const int ncb = convT_v2_ncb(Cin);
    const int cgroups = e2e::cdiv(Cin, 32 * ncb);
    const int pairs2 = cgroups * e2e::cdiv(Cout, 32);
    const int nch = wgrad_chunks(total_tiles, pairs2, &tpc);
    dim3 grid2(nch, pairs2);
    const int kdh = kd * kh;
    static const int use_bf3 = getenv("E2E_CT_BF3") ? atoi(getenv("E2E_CT_BF3")) : 1;
    if (use_bf3) {
      const bool h2 = ct_h2_env() && x_absmax != nullptr && dy_bound_a != nullptr;
      e2e::note_kernel("convT_wgrad_%s<%d,%d> chunks=%d pairs=%d", h2 ? "h2" : "bf3", kdh, ncb, nch, pairs2);
#define LAUNCH_B3(KDH, NCB, TPX, NP) hipLaunchKernelGGL((convT_wgrad_bf3_kernel<KDH, NCB, TPX, NP>), grid2, dim3(256 * NCB), 0, st, x, scale, shift, \
                                                        slope, dy, slab, B, Cin, Cout, D, H, W, kd, kh, tpc, cgroups, x_absmax, dy_bound_a, dy_bound_b)
      if (h2) {
        if (kdh == 4) { if (ncb == 2) LAUNCH_B3(4, 2, 64, 2); else LAUNCH_B3(4, 1, 64, 2); }
        else { if (ncb == 2) LAUNCH_B3(2, 2, 64, 2); else LAUNCH_B3(2, 1, 64, 2); }
      } else {
        if (kdh == 4) { if (ncb == 2) LAUNCH_B3(4, 2, 64, 3); else LAUNCH_B3(4, 1, 64, 3); }
        else { if (ncb == 2) LAUNCH_B3(2, 2, 64, 3); else LAUNCH_B3(2, 1, 64, 3); }
      }
#undef LAUNCH_B3
      hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)e2e::cdivll(numel_all, 64)), dim3(256), 0, st, slab, dw, numel_all, nch);
      return e2e::check_launch("convT_wgrad_bf3");
    }
    e2e::note_kernel("convT_wgrad_v2<%d,%d> chunks=%d pairs=%d", kdh, ncb, nch, pairs2);
#define LAUNCH_V2(KDH, NCB) hipLaunchKernelGGL((convT_wgrad_v2_kernel<KDH, NCB>), grid2, dim3(256 * NCB), 0, st, x, scale, shift, \
                                               slope, dy, slab, B, Cin, Cout, D, H, W, kd, kh, tpc, cgroups)
    if (kdh == 4) { if (ncb == 2) LAUNCH_V2(4, 2); else LAUNCH_V2(4, 1); }
    else { if (ncb == 2) LAUNCH_V2(2, 2); else LAUNCH_V2(2, 1); }
#undef LAUNCH_V2
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)e2e::cdivll(numel_all, 64)), dim3(256), 0, st, slab, dw, numel_all, nch);
    return e2e::check_launch("convT_wgrad_v2");
  }
  const int pairs = e2e::cdiv(Cin, 32) * e2e::cdiv(Cout, 32);
  const int nchunks = wgrad_chunks(total_tiles, pairs, &tpc);
  dim3 grid(nchunks, pairs);
  e2e::note_kernel("convT_wgrad_v1<%d> chunks=%d pairs=%d", kt, nchunks, pairs);
  DISPATCH_KT(kt, hipLaunchKernelGGL((convT_wgrad_kernel<KT>), grid, dim3(256), 0, st, x, scale, shift, slope, dy, slab, B,
                                     Cin, Cout, D, H, W, kd, kh, kw, tpc, nchunks));
  const long long numel = (long long)Cin * Cout * kt;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)e2e::cdivll(numel, 64)), dim3(256), 0, st, slab, dw, numel, nchunks);
  return e2e::check_launch("convT_wgrad");
}
