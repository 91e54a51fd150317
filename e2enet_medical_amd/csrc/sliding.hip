// K11: sliding-window inference aggregation on device (gfx950).
// Reference: _internal_maybe_mirror_and_pred_3D (neural_network.py:500-565: result += flip(softmax(net(flip(x)))) / n,
// result *= gaussian) and _internal_predict_3D_3Dconv_tiled (neural_network.py:383-407: overlap-add, count map,
// divide, argmax).  The reference moves every tile to the host and adds in numpy; here everything stays in HBM.
#include "e2e_common.h"

namespace {

__device__ __forceinline__ long long flipped_index(int x, int y, int z, int X, int Y, int Z, int axes) {
  const int fx = (axes & 1) ? X - 1 - x : x;
  const int fy = (axes & 2) ? Y - 1 - y : y;
  const int fz = (axes & 4) ? Z - 1 - z : z;
  return ((long long)fx * Y + fy) * Z + fz;
}

__global__ __launch_bounds__(256) void flip3d_kernel(const float* __restrict__ src, float* __restrict__ dst, int X, int Y, int Z,
                                                     int axes) {
  const long long spatial = (long long)X * Y * Z;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= spatial) return;
  const int z = (int)(i % Z);
  const long long r = i / Z;
  const int y = (int)(r % Y), x = (int)(r / Y);
  dst[(long long)blockIdx.y * spatial + i] = src[(long long)blockIdx.y * spatial + flipped_index(x, y, z, X, Y, Z, axes)];
}

template <int KB>
__global__ __launch_bounds__(256) void softmax_flip_acc_kernel(const float* __restrict__ logits, float* __restrict__ result,
                                                               float w, int first, int K, int X, int Y, int Z, int axes) {
  const long long spatial = (long long)X * Y * Z;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // index in the flipped (network) frame
  if (i >= spatial) return;
  const int z = (int)(i % Z);
  const long long r = i / Z;
  const int y = (int)(r % Y), x = (int)(r / Y);
  const long long o = flipped_index(x, y, z, X, Y, Z, axes);
  for (int k0 = 0; k0 < K; k0 += KB) {
    // softmax needs all classes: first pass max / sum over all K, then emit this block
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, logits[(long long)k * spatial + i]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(logits[(long long)k * spatial + i] - m);
    const float inv = 1.f / s;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      if (k0 + k >= K) break;
      const float p = expf(logits[(long long)(k0 + k) * spatial + i] - m) * inv;
      float* dst = result + (long long)(k0 + k) * spatial + o;
      // reference: result += (1 / n) * pred -- multiply, round, then add (no fma contraction)
      const float wp = __fmul_rn(w, p);
      if (first) *dst = wp;
      else *dst = __fadd_rn(*dst, wp);
    }
  }
}

// the same accumulation for an inference_apply_nonlin other than softmax (reference neural_network.py:531-560 applies whatever
// the attribute holds): NL 0 = identity (the constructor's default `lambda x: x`, :80), NL 2 = sigmoid (region-based heads)
template <int NL>
__global__ __launch_bounds__(256) void nonlin_flip_acc_kernel(const float* __restrict__ logits, float* __restrict__ result,
                                                              float w, int first, int K, int X, int Y, int Z, int axes) {
  const long long spatial = (long long)X * Y * Z;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= spatial) return;
  const int z = (int)(i % Z);
  const long long r = i / Z;
  const int y = (int)(r % Y), x = (int)(r / Y);
  const long long o = flipped_index(x, y, z, X, Y, Z, axes);
  for (int k = 0; k < K; ++k) {
    const float v = logits[(long long)k * spatial + i];
    const float p = NL == 2 ? 1.f / (1.f + expf(-v)) : v;
    float* dst = result + (long long)k * spatial + o;
    const float wp = __fmul_rn(w, p);
    if (first) *dst = wp;
    else *dst = __fadd_rn(*dst, wp);
  }
}

__global__ __launch_bounds__(256) void sw_accumulate_kernel(const float* __restrict__ patch, const float* __restrict__ gauss,
                                                            float* __restrict__ agg, float* __restrict__ cnt, int K, int X, int Y,
                                                            int Z, int px, int py, int pz, int x0, int y0, int z0) {
  const long long ps = (long long)px * py * pz;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ps) return;
  const int z = (int)(i % pz);
  const long long r = i / pz;
  const int y = (int)(r % py), x = (int)(r / py);
  const long long o = ((long long)(x0 + x) * Y + (y0 + y)) * Z + (z0 + z);
  const float g = gauss ? gauss[i] : 1.f;
  const long long vs = (long long)X * Y * Z;
  for (int k = 0; k < K; ++k) {
    // reference: patch *= gaussian (rounded), then aggregated += patch (neural_network.py:562-563, :392-393)
    if (patch != nullptr) {                                   // (null: a tile another rank evaluates -- weight map only)
      const float pg = __fmul_rn(patch[(long long)k * ps + i], g);
      agg[(long long)k * vs + o] = __fadd_rn(agg[(long long)k * vs + o], pg);
    }
    cnt[(long long)k * vs + o] = __fadd_rn(cnt[(long long)k * vs + o], g);
  }
}

__global__ __launch_bounds__(256) void sw_finalize_kernel(const float* __restrict__ agg, const float* __restrict__ cnt,
                                                          float* __restrict__ probs, long long* __restrict__ seg, int K, int X, int Y,
                                                          int Z, int cx0, int cy0, int cz0, int CX, int CY, int CZ) {
  const long long cs = (long long)CX * CY * CZ;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= cs) return;
  const int z = (int)(i % CZ);
  const long long r = i / CZ;
  const int y = (int)(r % CY), x = (int)(r / CY);
  const long long o = ((long long)(cx0 + x) * Y + (cy0 + y)) * Z + (cz0 + z);
  const long long vs = (long long)X * Y * Z;
  float best = -INFINITY;
  int arg = 0;
  for (int k = 0; k < K; ++k) {
    const float p = __fdiv_rn(agg[(long long)k * vs + o], cnt[(long long)k * vs + o]);
    probs[(long long)k * cs + i] = p;
    if (p > best || (p != p && best == best)) { best = p; arg = k; }   // first maximum; NaN counts as maximal (numpy)
  }
  seg[i] = arg;
}
// ---- N1: fold ensembling and segmentation export (reference e2enet/inference/predict.py:282-301,
// e2enet/inference/segmentation_export.py:118-136) ----------------------------------------------------------------
// dst (+)= src (first != 0: dst = src);  n_folds > 0: dst = dst / n_folds afterwards (softmax /= len(params), float32)
__global__ __launch_bounds__(256) void ensemble_acc_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n,
                                                           int first, int n_folds) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = first ? src[i] : __fadd_rn(dst[i], src[i]);
  if (n_folds > 0) v = __fdiv_rn(v, (float)n_folds);
  dst[i] = v;
}

// seg[bbox + t] = argmax_k probs[k, permuted t] (first maximum) or, with region classes, the last region i whose
// probability exceeds 0.5 (0 when none).  probs is [K, X, Y, Z] in network axis order; the output voxel t = (a, b, c)
// lives in the transposed frame: source index = a * sa + b * sb + c * sc (strides of softmax.transpose(...)).
__global__ __launch_bounds__(256) void export_argmax_kernel(const float* __restrict__ probs, unsigned char* __restrict__ seg, int K,
                                                            long long kstride, int A, int B, int C, long long sa, long long sb,
                                                            long long sc, int OA, int OB, int OC, int a0, int b0, int c0,
                                                            const int* __restrict__ regions, int n_regions) {
  const long long n = (long long)A * B * C;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const long long r = i / C;
  const int b = (int)(r % B), a = (int)(r / B);
  if (a0 + a >= OA || b0 + b >= OB || c0 + c >= OC) return;      // bbox clipped to the original size (reference :127)
  const long long o = a * sa + b * sb + c * sc;
  int lab = 0;
  if (regions == nullptr) {
    float best = -INFINITY;
    for (int k = 0; k < K; ++k) {
      const float p = probs[(long long)k * kstride + o];
      if (p > best || (p != p && best == best)) { best = p; lab = k; }
    }
  } else {
    for (int k = 0; k < n_regions; ++k)
      if (probs[(long long)k * kstride + o] > 0.5f) lab = regions[k];
  }
  seg[((long long)(a0 + a) * OB + (b0 + b)) * OC + (c0 + c)] = (unsigned char)lab;
}
// ---- N1: softmax volume resampled to the original (pre-resampling) grid on the device ---------------------------------
// Reference: save_segmentation_nifti_from_softmax (segmentation_export.py:84-104) -> resample_data_or_seg(is_seg=False,
// order=1, order_z=0) (preprocessing.py:113-202) -> skimage.transform.resize(order=1, mode='edge', anti_aliasing=False),
// which (scikit-image 0.19.3) is scipy.ndimage.zoom(order=1, mode='nearest', grid_mode=True) on the float64 image; with a
// separate low-resolution axis every slice is resized in the plane (cast back to float32) and the axis itself is then
// picked by nearest neighbour (map_coordinates order 0, mode 'nearest', coordinates scale * (i + 0.5) - 0.5).
// Arithmetic as scipy's NI_ZoomShift: coordinate = (o + 0.5) * (in / out) - 0.5 in double, clamped to [0, in - 1], floor and
// fraction, weights (1 - t, t), the 2^n taps visited last axis fastest, coefficient multiplied by the axis weights in axis
// order and added to a double sum (this file is built with -ffp-contract=off), result cast to float32.
// lowres < 0: trilinear over all three axes; lowres = 0..2: nearest along that axis, bilinear in the plane.
__device__ __forceinline__ void lin_coord(int o, int n_in, int n_out, int& i0, int& i1, double& t) {
  double c = ((double)o + 0.5) * ((double)n_in / (double)n_out) - 0.5;
  if (c < 0.0) c = 0.0;
  if (c > (double)(n_in - 1)) c = (double)(n_in - 1);
  const double f = floor(c);
  i0 = (int)f;
  t = c - f;
  i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
}
__device__ __forceinline__ int near_coord(int o, int n_in, int n_out) {
  double c = ((double)n_in / (double)n_out) * ((double)o + 0.5) - 0.5;
  if (c < 0.0) c = 0.0;
  if (c > (double)(n_in - 1)) c = (double)(n_in - 1);
  return (int)floor(c + 0.5);
}
__global__ __launch_bounds__(256) void resample_linear_kernel(const float* __restrict__ src, float* __restrict__ dst, int K,
                                                              long long kstride, int A, int B, int C, long long sa,
                                                              long long sb, long long sc, int OA, int OB, int OC, int lowres) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long ovol = (long long)OA * OB * OC;
  if (idx >= ovol) return;
  const int oc = (int)(idx % OC);
  const int ob = (int)((idx / OC) % OB);
  const int oa = (int)(idx / ((long long)OC * OB));
  const int n_in[3] = {A, B, C}, n_out[3] = {OA, OB, OC}, o[3] = {oa, ob, oc};
  const long long st[3] = {sa, sb, sc};
  int i0[3], i1[3];
  double w0[3], w1[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (d == lowres || n_in[d] == n_out[d]) {
      // nearest along the separate axis; an axis whose size does not change has coordinate o exactly (weight 1 on tap 0)
      i0[d] = i1[d] = (d == lowres) ? near_coord(o[d], n_in[d], n_out[d]) : o[d];
      w0[d] = 1.0; w1[d] = 0.0;
    } else {
      double t;
      lin_coord(o[d], n_in[d], n_out[d], i0[d], i1[d], t);
      w0[d] = 1.0 - t; w1[d] = t;
    }
  }
  for (int k = 0; k < K; ++k) {
    const float* sp = src + (long long)k * kstride;
    double acc = 0.0;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
          const double wa = ta ? w1[0] : w0[0], wb = tb ? w1[1] : w0[1], wc = tc ? w1[2] : w0[2];
          if (wa == 0.0 || wb == 0.0 || wc == 0.0) continue;       // (taps scipy does not visit: nearest axis, unchanged axis)
          double coeff = (double)sp[(ta ? i1[0] : i0[0]) * st[0] + (tb ? i1[1] : i0[1]) * st[1] + (tc ? i1[2] : i0[2]) * st[2]];
          coeff *= wa;
          coeff *= wb;
          coeff *= wc;
          acc += coeff;
        }
    dst[(long long)k * ovol + idx] = (float)acc;
  }
}

}  // namespace

extern "C" int e2e_ensemble_accumulate(float* dst, const float* src, long long n, int first, int n_folds, void* stream) {
  E2E_REQUIRE(dst && src && n > 0 && n_folds >= 0, "ensemble_accumulate: bad arguments");
  hipLaunchKernelGGL(ensemble_acc_kernel, dim3((unsigned)e2e::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n, first, n_folds);
  return e2e::check_launch("ensemble_acc_kernel");
}

extern "C" int e2e_export_argmax_u8(const float* probs, unsigned char* seg, int K, long long kstride, int A, int B, int C,
                                    long long sa, long long sb, long long sc, int OA, int OB, int OC, int a0, int b0, int c0,
                                    const int* regions, int n_regions, void* stream) {
  E2E_REQUIRE(probs && seg && K > 0 && A > 0 && B > 0 && C > 0, "export_argmax_u8: bad arguments");
  E2E_REQUIRE(a0 >= 0 && b0 >= 0 && c0 >= 0 && OA > 0 && OB > 0 && OC > 0, "export_argmax_u8: bad placement");
  E2E_REQUIRE(regions == nullptr || (n_regions > 0 && n_regions <= K), "export_argmax_u8: bad region list");
  hipLaunchKernelGGL(export_argmax_kernel, dim3((unsigned)e2e::cdivll((long long)A * B * C, 256)), dim3(256), 0, (hipStream_t)stream, probs,
                     seg, K, kstride, A, B, C, sa, sb, sc, OA, OB, OC, a0, b0, c0, regions, n_regions);
  return e2e::check_launch("export_argmax_kernel");
}

extern "C" int e2e_flip3d(const float* src, float* dst, int NC, int X, int Y, int Z, int axes, void* stream) {
  E2E_REQUIRE(src && dst && src != dst && NC > 0 && X > 0 && Y > 0 && Z > 0, "flip3d: bad arguments");
  dim3 grid((unsigned)e2e::cdivll((long long)X * Y * Z, 256), NC);
  hipLaunchKernelGGL(flip3d_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, X, Y, Z, axes);
  return e2e::check_launch("flip3d_kernel");
}

extern "C" int e2e_softmax_flip_acc(const float* logits, float* result, float w, int first, int K, int X, int Y, int Z,
                                    int axes, void* stream) {
  E2E_REQUIRE(logits && result && K > 0, "softmax_flip_acc: bad arguments");
  dim3 grid((unsigned)e2e::cdivll((long long)X * Y * Z, 256));
  hipLaunchKernelGGL((softmax_flip_acc_kernel<32>), grid, dim3(256), 0, (hipStream_t)stream, logits, result, w, first, K, X, Y, Z, axes);
  return e2e::check_launch("softmax_flip_acc_kernel");
}

extern "C" int e2e_nonlin_flip_acc(const float* logits, float* result, float w, int first, int K, int X, int Y, int Z,
                                   int axes, int nonlin, void* stream) {
  E2E_REQUIRE(logits && result && K > 0, "nonlin_flip_acc: bad arguments");
  E2E_REQUIRE(nonlin >= 0 && nonlin <= 2, "nonlin_flip_acc: nonlin must be 0 (identity), 1 (softmax over the classes) or 2 (sigmoid)");
  if (nonlin == 1) return e2e_softmax_flip_acc(logits, result, w, first, K, X, Y, Z, axes, stream);
  dim3 grid((unsigned)e2e::cdivll((long long)X * Y * Z, 256));
  if (nonlin == 0) hipLaunchKernelGGL((nonlin_flip_acc_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, logits, result, w, first, K, X, Y, Z, axes);
  else hipLaunchKernelGGL((nonlin_flip_acc_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, logits, result, w, first, K, X, Y, Z, axes);
  return e2e::check_launch("nonlin_flip_acc_kernel");
}

extern "C" int e2e_sw_accumulate(const float* patch, const float* gauss, float* agg, float* cnt, int K, int X, int Y,
                                 int Z, int px, int py, int pz, int x0, int y0, int z0, void* stream) {
  E2E_REQUIRE(agg && cnt && K > 0, "sw_accumulate: bad arguments");
  E2E_REQUIRE(x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + px <= X && y0 + py <= Y && z0 + pz <= Z, "sw_accumulate: tile outside volume");
  dim3 grid((unsigned)e2e::cdivll((long long)px * py * pz, 256));
  hipLaunchKernelGGL(sw_accumulate_kernel, grid, dim3(256), 0, (hipStream_t)stream, patch, gauss, agg, cnt, K, X, Y, Z, px, py, pz, x0, y0, z0);
  return e2e::check_launch("sw_accumulate_kernel");
}

extern "C" int e2e_sw_finalize_argmax(const float* agg, const float* cnt, float* probs, long long* seg, int K, int X,
                                      int Y, int Z, int cx0, int cy0, int cz0, int CX, int CY, int CZ, void* stream) {
  E2E_REQUIRE(agg && cnt && probs && seg && K > 0, "sw_finalize_argmax: bad arguments");
  E2E_REQUIRE(cx0 >= 0 && cy0 >= 0 && cz0 >= 0 && cx0 + CX <= X && cy0 + CY <= Y && cz0 + CZ <= Z, "sw_finalize_argmax: crop outside volume");
  dim3 grid((unsigned)e2e::cdivll((long long)CX * CY * CZ, 256));
  hipLaunchKernelGGL(sw_finalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, agg, cnt, probs, seg, K, X, Y, Z, cx0, cy0, cz0, CX, CY, CZ);
  return e2e::check_launch("sw_finalize_kernel");
}

extern "C" int e2e_resample_linear(const float* src, float* dst, int K, long long kstride, int A, int B, int C, long long sa,
                                   long long sb, long long sc, int OA, int OB, int OC, int lowres_axis, void* stream) {
  E2E_REQUIRE(src && dst && src != dst && K > 0 && A > 0 && B > 0 && C > 0 && OA > 0 && OB > 0 && OC > 0, "resample_linear: bad arguments");
  E2E_REQUIRE(lowres_axis >= -1 && lowres_axis <= 2, "resample_linear: lowres_axis must be -1 (none) or 0..2");
  hipLaunchKernelGGL(resample_linear_kernel, dim3((unsigned)e2e::cdivll((long long)OA * OB * OC, 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, K, kstride, A, B, C, sa, sb, sc, OA, OB, OC, lowres_axis);
  return e2e::check_launch("resample_linear_kernel");
}
