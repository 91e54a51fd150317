// N3: the training input feed on the device -- the spatial and intensity transforms of
// get_moreDA_augmentation (e2enet/training/data_augmentation/data_augmentation_moreDA.py:66-111), which the reference runs in
// 24 CPU worker processes (batchgenerators 0.24, third party, not under /root/reference).  One 50 ms training step consumes
// 40 batches/s of 2 x 4 x 128^3 voxels; these kernels turn a raw loaded patch into a network batch in a few ms.
//
//   e2e_aug_spatial     SpatialTransform (rotation + scaling as ONE affine gather, centre crop): data order 3 (the reference's
//                       order_data: cubic B-spline = scipy map_coordinates(order=3, mode='constant', cval 0) over the coefficient
//                       image of e2e_aug_bspline_prefilter_axis) or order 1; seg order 0 (cval border_val_seg) or order 1
//                       (batchgenerators' interpolate_img(is_seg=True): per label a linear interpolation of the binary mask,
//                       assigned where >= 0.5, labels in ascending order)
//   e2e_aug_bspline_prefilter_axis   scipy.ndimage.spline_filter1d(order=3, mode='mirror') along one axis (the recursive
//                       prefilter scipy applies in front of every order-3 interpolation; 'constant' and the pre-padded 'nearest'
//                       mode use the mirror boundary too)
//   (MirrorTransform: e2e_flip3d of sliding.hip, per sample, after the intensity transforms like in the reference)
//   e2e_aug_stats       per-(sample, channel) min / max / mean / std (np.std, ddof 0): ContrastAugmentation, Gamma
//   e2e_aug_pointwise   GaussianNoise (counter-based generator, Box-Muller), BrightnessMultiplicative, ContrastAugmentation,
//                       Gamma (power step and retain_stats step), all parameterised per (sample, channel)
//   e2e_aug_blur_axis   GaussianBlurTransform: one axis of scipy.ndimage.gaussian_filter (truncate 4, mode 'reflect')
//   e2e_aug_lowres      SimulateLowResolutionTransform with order_upsample 1: nearest down-sampling to round(shape * zoom)
//                       followed by the linear up-sampling back, evaluated as one composite gather
//   e2e_aug_lowres_down / e2e_aug_lowres_up3   the same transform with the reference's order_upsample = 3: the low-resolution
//                       volume is materialised edge-padded by 12 (scipy's pre-padding for mode 'nearest'), prefiltered, and
//                       up-sampled with the cubic B-spline, clipped to the low-resolution volume's range (skimage resize clip=True)
//   e2e_aug_finish      MaskTransform (data = 0 where seg < 0, for the channels normalised inside the mask) and
//                       RemoveLabelTransform(-1, 0)
// All of them are HBM streaming kernels.  Parity: unpinned by construction (batchgenerators is absent from the image); every
// kernel is tested, given the drawn parameters, against scipy.ndimage / numpy on the CPU (tests/test_gpu_augment.py).
#include "e2e_common.h"

namespace {

// ---- cubic B-spline (scipy.ndimage, order 3) --------------------------------------------------------------------------------
// weights of the four taps floor(x) - 1 .. floor(x) + 2 at offset t = x - floor(x)
__device__ __forceinline__ void bspline3_weights(double t, double (&w)[4]) {
  const double u = 1.0 - t;
  w[0] = u * u * u / 6.0;
  w[1] = (3.0 * t * t * t - 6.0 * t * t + 4.0) / 6.0;
  w[2] = (-3.0 * t * t * t + 3.0 * t * t + 3.0 * t + 1.0) / 6.0;
  w[3] = t * t * t / 6.0;
}
// tap index outside [0, n): the coefficient image continues by mirror (d c b | a b c d | c b a), period 2 n - 2
__device__ __forceinline__ int mirror_index(int i, int n) {
  if (n <= 1) return 0;
  const int s2 = 2 * n - 2;
  if (i < 0) {
    i = s2 * (-i / s2) + i;
    return i <= 1 - n ? i + s2 : -i;
  }
  if (i >= n) {
    i -= s2 * (i / s2);
    if (i >= n) i = s2 - i;
  }
  return i;
}
// value of the spline with coefficient image c [D, H, W] at (cd, ch, cw); the caller has checked the coordinate range
__device__ __forceinline__ double bspline3_at(const float* __restrict__ c, int D, int H, int W, double cd, double ch, double cw) {
  const double fd = floor(cd), fh = floor(ch), fw = floor(cw);
  double wd[4], wh[4], ww[4];
  bspline3_weights(cd - fd, wd);
  bspline3_weights(ch - fh, wh);
  bspline3_weights(cw - fw, ww);
  int id[4], ih[4], iw[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    id[k] = mirror_index((int)fd - 1 + k, D);
    ih[k] = mirror_index((int)fh - 1 + k, H);
    iw[k] = mirror_index((int)fw - 1 + k, W);
  }
  double acc = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    double sa = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float* row = c + ((long long)id[a] * H + ih[b]) * W;
      double sb = 0.0;
#pragma unroll
      for (int e = 0; e < 4; ++e) sb += ww[e] * (double)row[iw[e]];
      sa += wh[b] * sb;
    }
    acc += wd[a] * sa;
  }
  return acc;
}

// The recursive prefilter of a line (scipy ni_splines.c: gain 6, pole z = sqrt(3) - 2, causal initialisation
// _init_causal_mirror, anticausal _init_anticausal_mirror): fp64 arithmetic, fp32 storage (the causal results included).
#define BSPLINE_POLE (-0.26794919243112270647)

// strided axis (D or H): one thread per line, neighbouring threads on neighbouring inner positions (coalesced)
// (src and dst may be the same buffer: the low-resolution path filters in place -- no __restrict__)
__global__ __launch_bounds__(256) void bspline_prefilter_strided_kernel(const float* src, float* dst,
                                                                        long long n_outer, int len, long long inner) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_outer * inner) return;
  const long long o = t / inner, i = t - o * inner;
  const float* s = src + o * len * inner + i;
  float* d = dst + o * len * inner + i;
  // pass 1 reads src (initial sum + causal recursion), pass 2 reads the causal results in dst
  const double z = BSPLINE_POLE;
  if (len < 2) { if (len == 1) d[0] = s[0]; return; }
  const double zn1 = pow(z, (double)(len - 1));
  double c0 = 6.0 * ((double)s[0] + zn1 * (double)s[(long long)(len - 1) * inner]);
  double zi = z;
  const int hz = len - 1 < 40 ? len - 1 : 40;
  for (int k = 1; k < hz; ++k) {
    c0 += zi * 6.0 * ((double)s[(long long)k * inner] + zn1 * (double)s[(long long)(len - 1 - k) * inner]);
    zi *= z;
  }
  c0 /= 1.0 - zn1 * zn1;
  double prev = c0;
  float last2 = 0.f, last1 = (float)prev;
  d[0] = (float)prev;
  for (int k = 1; k < len; ++k) {
    prev = 6.0 * (double)s[(long long)k * inner] + z * (double)last1;      // (scipy recurses on the stored value: float here)
    last2 = last1; last1 = (float)prev;
    d[(long long)k * inner] = last1;
  }
  double nxt = (z * (double)last2 + (double)last1) * z / (z * z - 1.0);
  float nf = (float)nxt;
  d[(long long)(len - 1) * inner] = nf;
  for (int k = len - 2; k >= 0; --k) {
    nxt = z * ((double)nf - (double)d[(long long)k * inner]);
    nf = (float)nxt;
    d[(long long)k * inner] = nf;
  }
}

// contiguous axis (W): a block stages LPB lines in LDS (coalesced loads), one thread filters one line there, coalesced stores
template <int LPB>
__global__ __launch_bounds__(256) void bspline_prefilter_rows_kernel(const float* src, float* dst,
                                                                     long long n_lines, int len) {
  extern __shared__ float rows[];                       // [LPB][pitch], pitch odd: the LPB serial walkers hit distinct banks
  const int pitch = len | 1;
  const long long line0 = (long long)blockIdx.x * LPB;
  const int nl = n_lines - line0 < LPB ? (int)(n_lines - line0) : LPB;
  for (int e = threadIdx.x; e < nl * len; e += 256) {
    const int l = e / len, k = e - l * len;
    rows[l * pitch + k] = src[(line0 + l) * len + k];
  }
  __syncthreads();
  if ((int)threadIdx.x < nl) {
    float* r = rows + threadIdx.x * pitch;
    const double z = BSPLINE_POLE;
    if (len >= 2) {
      const double zn1 = pow(z, (double)(len - 1));
      double c0 = 6.0 * ((double)r[0] + zn1 * (double)r[len - 1]);
      double zi = z;
      const int hz = len - 1 < 40 ? len - 1 : 40;
      for (int k = 1; k < hz; ++k) { c0 += zi * 6.0 * ((double)r[k] + zn1 * (double)r[len - 1 - k]); zi *= z; }
      c0 /= 1.0 - zn1 * zn1;
      float last = (float)c0;
      r[0] = last;
      for (int k = 1; k < len; ++k) { last = (float)(6.0 * (double)r[k] + z * (double)last); r[k] = last; }
      float nf = (float)((z * (double)r[len - 2] + (double)r[len - 1]) * z / (z * z - 1.0));
      r[len - 1] = nf;
      for (int k = len - 2; k >= 0; --k) { nf = (float)(z * ((double)nf - (double)r[k])); r[k] = nf; }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nl * len; e += 256) {
    const int l = e / len, k = e - l * len;
    dst[(line0 + l) * len + k] = rows[l * pitch + k];
  }
}

// ---- spatial: out[b, c, o] = interp(in[b, c], A_b (o - c_out) + t_b) ---------------------------------------------------------
// mat: per sample 12 doubles (row-major 3 x 4: A | t) mapping zero-centred output coordinates o - (size - 1) / 2 to input
// voxel coordinates; computed on the host.
__global__ __launch_bounds__(256) void aug_spatial_kernel(const float* __restrict__ data, const float* __restrict__ coef,
                                                          const int* __restrict__ cubic, const float* __restrict__ seg,
                                                          float* __restrict__ odata, float* __restrict__ oseg,
                                                          const double* __restrict__ mat, int C, int CS, int Di, int Hi, int Wi,
                                                          int Do, int Ho, int Wo, int order_seg, float cval_seg) {
  const long long ovol = (long long)Do * Ho * Wo;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (idx >= ovol) return;
  const int ow = (int)(idx % Wo), oh = (int)((idx / Wo) % Ho), od = (int)(idx / ((long long)Wo * Ho));
  const double* m = mat + b * 12;
  const double z = od - 0.5 * (Do - 1), y = oh - 0.5 * (Ho - 1), x = ow - 0.5 * (Wo - 1);
  const double cd = m[0] * z + m[1] * y + m[2] * x + m[3];
  const double ch = m[4] * z + m[5] * y + m[6] * x + m[7];
  const double cw = m[8] * z + m[9] * y + m[10] * x + m[11];
  const long long ivol = (long long)Di * Hi * Wi;
  // ---- data: order 3 on the coefficient image (samples flagged in `cubic`), mode 'constant', cval 0: a coordinate outside
  //      [0, n - 1] returns cval, taps of an inside coordinate that fall outside read the mirrored coefficients ----
  if (coef != nullptr && (cubic == nullptr || cubic[b] != 0)) {
    const bool inside = cd >= 0.0 && cd <= (double)(Di - 1) && ch >= 0.0 && ch <= (double)(Hi - 1) && cw >= 0.0 && cw <= (double)(Wi - 1);
    for (int c = 0; c < C; ++c)
      odata[((long long)b * C + c) * ovol + idx] =
          inside ? (float)bspline3_at(coef + ((long long)b * C + c) * ivol, Di, Hi, Wi, cd, ch, cw) : 0.f;
  } else
  // ---- data: order 1, mode 'constant', cval 0 (scipy map_coordinates: a tap outside the volume contributes cval) ----
  {
    const double fd = floor(cd), fh = floor(ch), fw = floor(cw);
    const int d0 = (int)fd, h0 = (int)fh, w0 = (int)fw;
    const double td = cd - fd, th = ch - fh, tw = cw - fw;
    // scipy: a coordinate outside [0, n - 1] (beyond half a voxel of slack is irrelevant for order 1: it returns cval as a whole)
    const bool inside = cd >= 0.0 && cd <= (double)(Di - 1) && ch >= 0.0 && ch <= (double)(Hi - 1) && cw >= 0.0 && cw <= (double)(Wi - 1);
    for (int c = 0; c < C; ++c) {
      float v = 0.f;
      if (inside) {
        const float* p = data + ((long long)b * C + c) * ivol;
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int di = d0 + a < Di ? d0 + a : Di - 1, hi = h0 + bb < Hi ? h0 + bb : Hi - 1, wi = w0 + e < Wi ? w0 + e : Wi - 1;
              const double wgt = (a ? td : 1.0 - td) * (bb ? th : 1.0 - th) * (e ? tw : 1.0 - tw);
              acc += wgt * (double)p[((long long)di * Hi + hi) * Wi + wi];
            }
        v = (float)acc;
      }
      odata[((long long)b * C + c) * ovol + idx] = v;
    }
  }
  // ---- seg: mode 'constant' ----
  if (seg != nullptr) {
    const bool inside = cd >= 0.0 && cd <= (double)(Di - 1) && ch >= 0.0 && ch <= (double)(Hi - 1) && cw >= 0.0 && cw <= (double)(Wi - 1);
    if (order_seg == 0) {       // nearest, round half up like scipy: floor(c + 0.5); outside: cval
      const int di = (int)floor(cd + 0.5), hi = (int)floor(ch + 0.5), wi = (int)floor(cw + 0.5);
      for (int c = 0; c < CS; ++c) {
        float v = cval_seg;
        if (inside) {
          const int dd = di < Di ? di : Di - 1, hh = hi < Hi ? hi : Hi - 1, ww = wi < Wi ? wi : Wi - 1;
          v = seg[((long long)b * CS + c) * ivol + ((long long)dd * Hi + hh) * Wi + ww];
        }
        oseg[((long long)b * CS + c) * ovol + idx] = v;
      }
    } else {
      // batchgenerators interpolate_img(is_seg=True, order=1): result = 0; for every label c of the image in ascending order:
      // result[map_coordinates(img == c, coords, order=1, cval) >= 0.5] = c.  Outside the volume every mask interpolates to
      // cval (-1 < 0.5): the voxel keeps 0.
      const double fd = floor(cd), fh = floor(ch), fw = floor(cw);
      const int d0 = (int)fd, h0 = (int)fh, w0 = (int)fw;
      const double td = cd - fd, th = ch - fh, tw = cw - fw;
      for (int c = 0; c < CS; ++c) {
        float res = 0.f;
        if (inside) {
          const float* p = seg + ((long long)b * CS + c) * ivol;
          float lab[8];
          double wgt[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int a = k >> 2, bb = (k >> 1) & 1, e = k & 1;
            const int di = d0 + a < Di ? d0 + a : Di - 1, hi = h0 + bb < Hi ? h0 + bb : Hi - 1, wi = w0 + e < Wi ? w0 + e : Wi - 1;
            lab[k] = p[((long long)di * Hi + hi) * Wi + wi];
            wgt[k] = (a ? td : 1.0 - td) * (bb ? th : 1.0 - th) * (e ? tw : 1.0 - tw);
          }
          // ascending label order: the largest label whose mask reaches 0.5 wins
          float best = -INFINITY;
          bool any = false;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (lab[j] == lab[k]) sum += wgt[j];
            if (sum >= 0.5 && (!any || lab[k] > best)) { best = lab[k]; any = true; }
          }
          if (any) res = best;
        }
        oseg[((long long)b * CS + c) * ovol + idx] = res;
      }
    }
  }
}

// ---- statistics: stats[bc] = (min, max, sum, sum of squares) in double; zeroed / seeded by the launcher ------------------
__global__ __launch_bounds__(256) void aug_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, long long vol,
                                                        int chunks) {
  const int bc = blockIdx.y;
  const float* p = x + (long long)bc * vol;
  const long long per = e2e::cdivll(vol, chunks);
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < vol ? lo + per : vol;
  double s = 0.0, s2 = 0.0;
  float mn = INFINITY, mx = -INFINITY;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = p[i];
    s += (double)v; s2 += (double)v * (double)v;
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  s = e2e::wave_sum_d(s); s2 = e2e::wave_sum_d(s2);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off, 64)); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
  __shared__ double sh[4][4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[w][0] = (double)mn; sh[w][1] = (double)mx; sh[w][2] = s; sh[w][3] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = sh[0][0], b = sh[0][1], c = sh[0][2], d = sh[0][3];
    for (int i = 1; i < 4; ++i) { a = fmin(a, sh[i][0]); b = fmax(b, sh[i][1]); c += sh[i][2]; d += sh[i][3]; }
    double* o = stats + ((long long)bc * chunks + blockIdx.x) * 4;
    o[0] = a; o[1] = b; o[2] = c; o[3] = d;
  }
}
// (min, max, mean, std) per (sample, channel) from the chunk records, fixed order
__global__ void aug_stats_final_kernel(const double* __restrict__ part, double* __restrict__ out, long long vol, int chunks, int nbc) {
  const int bc = blockIdx.x * blockDim.x + threadIdx.x;
  if (bc >= nbc) return;
  const double* p = part + (long long)bc * chunks * 4;
  double a = p[0], b = p[1], c = p[2], d = p[3];
  for (int i = 1; i < chunks; ++i) { a = fmin(a, p[i * 4]); b = fmax(b, p[i * 4 + 1]); c += p[i * 4 + 2]; d += p[i * 4 + 3]; }
  const double mean = c / (double)vol;
  double var = d / (double)vol - mean * mean;
  if (var < 0.0) var = 0.0;
  out[bc * 4] = a; out[bc * 4 + 1] = b; out[bc * 4 + 2] = mean; out[bc * 4 + 3] = sqrt(var);
}

// ---- counter-based generator: two rounds of a 64-bit mix (splitmix64 finaliser) per draw; Box-Muller ------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float gauss_at(unsigned long long seed, unsigned long long counter) {
  const unsigned long long r = mix64(mix64(seed) ^ (counter * 0xd1342543de82ef95ull));
  const float u1 = ((float)(unsigned)(r >> 40) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
  const float u2 = (float)(unsigned)((r >> 8) & 0xffffffu) * (1.0f / 16777216.0f);   // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// ---- pointwise: one op per launch, parameters per (sample, channel): prm[bc * 8 .. + 7]; prm[0] == 0 -> channel untouched ----
enum { OP_NOISE = 1, OP_MUL = 2, OP_CONTRAST = 3, OP_GAMMA_POW = 4, OP_RENORM = 5 };
__global__ __launch_bounds__(256) void aug_pointwise_kernel(float* __restrict__ x, const double* __restrict__ prm, int op,
                                                            long long vol, unsigned long long seed) {
  const int bc = blockIdx.y;
  const double* q = prm + (long long)bc * 8;
  if (q[0] == 0.0) return;
  float* p = x + (long long)bc * vol;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < vol; i += (long long)gridDim.x * 256) {
    float v = p[i];
    switch (op) {
      case OP_NOISE:        // data + N(0, q[1])          (GaussianNoiseTransform: the drawn "variance" is used as the scale)
        v += (float)q[1] * gauss_at(seed + (unsigned long long)bc, (unsigned long long)i);
        break;
      case OP_MUL:          // data * multiplier          (BrightnessMultiplicativeTransform)
        v = (float)((double)v * q[1]);
        break;
      case OP_CONTRAST: {   // (data - mean) * factor + mean, clipped to [min, max]   (ContrastAugmentationTransform, preserve_range)
        double t = ((double)v - q[2]) * q[1] + q[2];
        t = t < q[3] ? q[3] : (t > q[4] ? q[4] : t);
        v = (float)t;
        break;
      }
      case OP_GAMMA_POW: {  // ((+-data - min) / (range + 1e-7)) ** gamma * range + min, sign restored   (augment_gamma)
        const double s = q[5] != 0.0 ? -(double)v : (double)v;           // q[5]: invert_image
        double t = pow((s - q[2]) / (q[3] + 1e-7), q[1]) * q[3] + q[2];  // q[2] = min, q[3] = range of the (inverted) data
        v = (float)(q[5] != 0.0 ? -t : t);
        break;
      }
      case OP_RENORM: {     // retain_stats: (data - mean_new) / (std_new + 1e-8) * sd + mn   (on the inverted data when inverted)
        const double s = q[5] != 0.0 ? -(double)v : (double)v;
        const double t = (s - q[1]) / (q[2] + 1e-8) * q[4] + q[3];
        v = (float)(q[5] != 0.0 ? -t : t);
        break;
      }
    }
    p[i] = v;
  }
}

// ---- Gaussian blur along one axis: scipy.ndimage.gaussian_filter1d(sigma, truncate = 4, mode = 'reflect') -------------------
// wts[bc * 16 ..]: radius (as float), then the normalised weights w[0..radius] (w[0] = centre), radius <= 12; radius 0 = copy
__global__ __launch_bounds__(256) void aug_blur_axis_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            const float* __restrict__ wts, int D, int H, int W, int axis) {
  const int bc = blockIdx.y;
  const long long vol = (long long)D * H * W;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= vol) return;
  const float* wq = wts + (long long)bc * 16;
  const int r = (int)wq[0];
  const float* p = src + (long long)bc * vol;
  if (r == 0) { dst[(long long)bc * vol + idx] = p[idx]; return; }
  const int w = (int)(idx % W), h = (int)((idx / W) % H), d = (int)(idx / ((long long)W * H));
  const int n = axis == 0 ? D : (axis == 1 ? H : W);
  const int pos = axis == 0 ? d : (axis == 1 ? h : w);
  const long long stride = axis == 0 ? (long long)H * W : (axis == 1 ? W : 1);
  const long long base = idx - (long long)pos * stride;
  double acc = 0.0;
  for (int k = -r; k <= r; ++k) {
    int j = pos + k;
    // 'reflect' (d c b a | a b c d | d c b a): period 2 n
    while (j < 0 || j >= n) j = j < 0 ? -j - 1 : 2 * n - 1 - j;
    acc += (double)wq[1 + (k < 0 ? -k : k)] * (double)p[base + (long long)j * stride];
  }
  dst[(long long)bc * vol + idx] = (float)acc;
}

// ---- low-resolution simulation: nearest down-sampling to lo = round(shape * zoom), linear up-sampling back ------------------
// both with skimage resize = scipy zoom (grid_mode = True, mode 'nearest') coordinates; lo[bc * 3 ..] = low-res shape (0 = untouched)
__device__ __forceinline__ int lr_near(int o, int n_in, int n_out) {       // zoom order 0: source index of low-res voxel o
  double c = ((double)o + 0.5) * ((double)n_in / (double)n_out) - 0.5;
  if (c < 0.0) c = 0.0;
  if (c > (double)(n_in - 1)) c = (double)(n_in - 1);
  return (int)floor(c + 0.5);
}
__global__ __launch_bounds__(256) void aug_lowres_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         const int* __restrict__ lo, int D, int H, int W) {
  const int bc = blockIdx.y;
  const long long vol = (long long)D * H * W;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= vol) return;
  const int* l = lo + bc * 3;
  const float* p = src + (long long)bc * vol;
  if (l[0] == 0) { dst[(long long)bc * vol + idx] = p[idx]; return; }
  const int o[3] = {(int)(idx / ((long long)W * H)), (int)((idx / W) % H), (int)(idx % W)};
  const int n[3] = {D, H, W};
  int i0[3], i1[3];
  double t[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // up-sampling (order 1) coordinate in the low-res grid, then each low-res tap's own source voxel (order 0 down-sampling)
    double c = ((double)o[a] + 0.5) * ((double)l[a] / (double)n[a]) - 0.5;
    if (c < 0.0) c = 0.0;
    if (c > (double)(l[a] - 1)) c = (double)(l[a] - 1);
    const double f = floor(c);
    const int j0 = (int)f, j1 = j0 + 1 < l[a] ? j0 + 1 : l[a] - 1;
    t[a] = c - f;
    i0[a] = lr_near(j0, n[a], l[a]);
    i1[a] = lr_near(j1, n[a], l[a]);
  }
  double acc = 0.0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double wgt = (a ? t[0] : 1.0 - t[0]) * (b ? t[1] : 1.0 - t[1]) * (e ? t[2] : 1.0 - t[2]);
        acc += wgt * (double)p[((long long)(a ? i1[0] : i0[0]) * H + (b ? i1[1] : i0[1])) * W + (e ? i1[2] : i0[2])];
      }
  dst[(long long)bc * vol + idx] = (float)acc;
}

// ---- low-resolution simulation with cubic up-sampling (order_upsample = 3) ---------------------------------------------------
// down: dst [(ld + 2 pad), (lh + 2 pad), (lw + 2 pad)] = the nearest-down-sampled volume, edge padded (scipy zoom pre-pads by 12 for
// mode 'nearest' before the spline prefilter)
__global__ __launch_bounds__(256) void aug_lowres_down_kernel(const float* __restrict__ src, float* __restrict__ dst, int D, int H,
                                                              int W, int ld, int lh, int lw, int pad) {
  const int pd = ld + 2 * pad, ph = lh + 2 * pad, pw = lw + 2 * pad;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)pd * ph * pw) return;
  int k = (int)(idx % pw) - pad, j = (int)((idx / pw) % ph) - pad, i = (int)(idx / ((long long)pw * ph)) - pad;
  i = i < 0 ? 0 : (i >= ld ? ld - 1 : i);
  j = j < 0 ? 0 : (j >= lh ? lh - 1 : j);
  k = k < 0 ? 0 : (k >= lw ? lw - 1 : k);
  dst[idx] = src[((long long)lr_near(i, D, ld) * H + lr_near(j, H, lh)) * W + lr_near(k, W, lw)];
}
// up: dst [D, H, W] = clip(spline(coef)(grid-mode coordinate + pad), min, max of the low-resolution volume)
__global__ __launch_bounds__(256) void aug_lowres_up3_kernel(const float* __restrict__ coef, float* __restrict__ dst,
                                                             const double* __restrict__ minmax, int D, int H, int W, int ld, int lh,
                                                             int lw, int pad) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)D * H * W) return;
  const int ow = (int)(idx % W), oh = (int)((idx / W) % H), od = (int)(idx / ((long long)W * H));
  const double cd = ((double)od + 0.5) * ((double)ld / (double)D) - 0.5 + pad;
  const double ch = ((double)oh + 0.5) * ((double)lh / (double)H) - 0.5 + pad;
  const double cw = ((double)ow + 0.5) * ((double)lw / (double)W) - 0.5 + pad;
  double v = bspline3_at(coef, ld + 2 * pad, lh + 2 * pad, lw + 2 * pad, cd, ch, cw);
  float f = (float)v;
  const float mn = (float)minmax[0], mx = (float)minmax[1];
  dst[idx] = f < mn ? mn : (f > mx ? mx : f);
}

// ---- finish: MaskTransform (set_outside_to 0 where seg channel 0 < 0) and RemoveLabelTransform(-1, 0) -----------------------
__global__ __launch_bounds__(256) void aug_finish_kernel(float* __restrict__ data, float* __restrict__ seg,
                                                         const int* __restrict__ use_mask, int C, int CS, long long vol) {
  const int b = blockIdx.y;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= vol) return;
  const bool outside = seg[(long long)b * CS * vol + idx] < 0.f;
  if (outside && use_mask != nullptr)
    for (int c = 0; c < C; ++c)
      if (use_mask[c]) data[((long long)b * C + c) * vol + idx] = 0.f;
  for (int c = 0; c < CS; ++c) {
    float* s = seg + ((long long)b * CS + c) * vol + idx;
    if (*s == -1.f) *s = 0.f;
  }
}

}  // namespace

extern "C" int e2e_aug_spatial(const float* data, const float* coef, const int* cubic, const float* seg, float* out_data,
                               float* out_seg, const double* mat, int B, int C, int CS, int Di, int Hi, int Wi, int Do, int Ho,
                               int Wo, int order_seg, float cval_seg, void* stream) {
  E2E_REQUIRE(data && out_data && mat && B > 0 && C > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0, "aug_spatial: bad arguments");
  E2E_REQUIRE((seg == nullptr) == (out_seg == nullptr) && (seg == nullptr || CS > 0), "aug_spatial: seg / out_seg must come together");
  E2E_REQUIRE(order_seg == 0 || order_seg == 1, "aug_spatial: order_seg must be 0 or 1");
  dim3 grid((unsigned)e2e::cdivll((long long)Do * Ho * Wo, 256), B);
  hipLaunchKernelGGL(aug_spatial_kernel, grid, dim3(256), 0, (hipStream_t)stream, data, coef, cubic, seg, out_data, out_seg, mat, C,
                     CS, Di, Hi, Wi, Do, Ho, Wo, order_seg, cval_seg);
  return e2e::check_launch("aug_spatial_kernel");
}

extern "C" int e2e_aug_bspline_prefilter_axis(const float* src, float* dst, int nvol, int D, int H, int W, int axis, void* stream) {
  E2E_REQUIRE(src && dst && nvol > 0 && D > 0 && H > 0 && W > 0 && axis >= 0 && axis <= 2, "aug_bspline_prefilter_axis: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (axis == 2) {
    const long long lines = (long long)nvol * D * H;
    const int pitch = W | 1;
    E2E_REQUIRE((long long)16 * pitch * 4 <= 160 * 1024, "aug_bspline_prefilter_axis: rows longer than 2559 voxels are not served");
    if ((long long)64 * pitch * 4 <= 64 * 1024)
      hipLaunchKernelGGL((bspline_prefilter_rows_kernel<64>), dim3((unsigned)e2e::cdivll(lines, 64)), dim3(256), (size_t)64 * pitch * 4, st, src, dst, lines, W);
    else {
      // (set before every such launch: a process-wide flag would leave the attribute unset on the other devices of the process)
      E2E_REQUIRE(hipFuncSetAttribute((const void*)bspline_prefilter_rows_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                  "aug_bspline_prefilter_axis: cannot raise the dynamic LDS limit");
      hipLaunchKernelGGL((bspline_prefilter_rows_kernel<16>), dim3((unsigned)e2e::cdivll(lines, 16)), dim3(256), (size_t)16 * pitch * 4, st, src, dst, lines, W);
    }
    return e2e::check_launch("bspline_prefilter_rows_kernel");
  }
  const long long n_outer = axis == 0 ? nvol : (long long)nvol * D;
  const int len = axis == 0 ? D : H;
  const long long inner = axis == 0 ? (long long)H * W : W;
  hipLaunchKernelGGL(bspline_prefilter_strided_kernel, dim3((unsigned)e2e::cdivll(n_outer * inner, 256)), dim3(256), 0, st, src, dst,
                     n_outer, len, inner);
  return e2e::check_launch("bspline_prefilter_strided_kernel");
}

extern "C" long long e2e_aug_stats_ws_bytes(int nbc) { return (long long)nbc * 64 * 4 * (long long)sizeof(double); }

extern "C" int e2e_aug_stats(const float* x, double* stats, double* ws, int nbc, long long vol, void* stream) {
  E2E_REQUIRE(x && stats && ws && nbc > 0 && vol > 0, "aug_stats: bad arguments");
  const int chunks = 64;
  hipLaunchKernelGGL(aug_stats_kernel, dim3(chunks, nbc), dim3(256), 0, (hipStream_t)stream, x, ws, vol, chunks);
  hipLaunchKernelGGL(aug_stats_final_kernel, dim3(e2e::cdiv(nbc, 64)), dim3(64), 0, (hipStream_t)stream, ws, stats, vol, chunks, nbc);
  return e2e::check_launch("aug_stats_kernel");
}

extern "C" int e2e_aug_pointwise(float* x, const double* prm, int op, int nbc, long long vol, unsigned long long seed, void* stream) {
  E2E_REQUIRE(x && prm && nbc > 0 && vol > 0 && op >= OP_NOISE && op <= OP_RENORM, "aug_pointwise: bad arguments");
  long long blocks = e2e::cdivll(vol, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aug_pointwise_kernel, dim3((unsigned)blocks, nbc), dim3(256), 0, (hipStream_t)stream, x, prm, op, vol, seed);
  return e2e::check_launch("aug_pointwise_kernel");
}

extern "C" int e2e_aug_blur_axis(const float* src, float* dst, const float* wts, int nbc, int D, int H, int W, int axis, void* stream) {
  E2E_REQUIRE(src && dst && src != dst && wts && nbc > 0 && axis >= 0 && axis <= 2, "aug_blur_axis: bad arguments");
  dim3 grid((unsigned)e2e::cdivll((long long)D * H * W, 256), nbc);
  hipLaunchKernelGGL(aug_blur_axis_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, wts, D, H, W, axis);
  return e2e::check_launch("aug_blur_axis_kernel");
}

extern "C" int e2e_aug_lowres(const float* src, float* dst, const int* lo_shape, int nbc, int D, int H, int W, void* stream) {
  E2E_REQUIRE(src && dst && src != dst && lo_shape && nbc > 0, "aug_lowres: bad arguments");
  dim3 grid((unsigned)e2e::cdivll((long long)D * H * W, 256), nbc);
  hipLaunchKernelGGL(aug_lowres_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, lo_shape, D, H, W);
  return e2e::check_launch("aug_lowres_kernel");
}

extern "C" int e2e_aug_lowres_down(const float* src, float* dst, int D, int H, int W, int ld, int lh, int lw, int pad, void* stream) {
  E2E_REQUIRE(src && dst && src != dst && D > 0 && H > 0 && W > 0 && ld > 0 && lh > 0 && lw > 0 && pad >= 0, "aug_lowres_down: bad arguments");
  const long long n = (long long)(ld + 2 * pad) * (lh + 2 * pad) * (lw + 2 * pad);
  hipLaunchKernelGGL(aug_lowres_down_kernel, dim3((unsigned)e2e::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, D, H, W,
                     ld, lh, lw, pad);
  return e2e::check_launch("aug_lowres_down_kernel");
}

extern "C" int e2e_aug_lowres_up3(const float* coef, float* dst, const double* minmax, int D, int H, int W, int ld, int lh, int lw,
                                  int pad, void* stream) {
  E2E_REQUIRE(coef && dst && minmax && D > 0 && H > 0 && W > 0 && ld > 0 && lh > 0 && lw > 0 && pad >= 2, "aug_lowres_up3: bad arguments");
  hipLaunchKernelGGL(aug_lowres_up3_kernel, dim3((unsigned)e2e::cdivll((long long)D * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                     coef, dst, minmax, D, H, W, ld, lh, lw, pad);
  return e2e::check_launch("aug_lowres_up3_kernel");
}

extern "C" int e2e_aug_finish(float* data, float* seg, const int* use_mask, int B, int C, int CS, long long vol, void* stream) {
  E2E_REQUIRE(data && seg && B > 0 && C > 0 && CS > 0 && vol > 0, "aug_finish: bad arguments");
  dim3 grid((unsigned)e2e::cdivll(vol, 256), B);
  hipLaunchKernelGGL(aug_finish_kernel, grid, dim3(256), 0, (hipStream_t)stream, data, seg, use_mask, C, CS, vol);
  return e2e::check_launch("aug_finish_kernel");
}
