// K9: DSFF kernel statistics (gfx950): kernel-L1 magnitudes, exact k-th order statistic, death mask, mask /
// liveness-bit expansion.  Reference: Masking.kernel_death (core_channel.py:647-666).
#include "e2e_common.h"

namespace {

// l1[r, c] = sum_kd( sum_kh( sum_kw |w| ) ) with each sum taken left to right: exactly the association order of the
// reference's three chained torch.sum(dim=-1) (core_channel.py:652-655) for kernel extents <= 3.
__global__ __launch_bounds__(256) void kernel_l1_kernel(const float* __restrict__ w, float* __restrict__ l1, long long n,
                                                        int kd, int kh, int kw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = w + i * (kd * kh * kw);
  float sd = 0.f;
  for (int a = 0; a < kd; ++a) {
    float sh = 0.f;
    for (int b = 0; b < kh; ++b) {
      float sw = 0.f;
      for (int c = 0; c < kw; ++c) {
        const float v = fabsf(p[(a * kh + b) * kw + c]);
        sw = (c == 0) ? v : __fadd_rn(sw, v);
      }
      sh = (b == 0) ? sw : __fadd_rn(sh, sw);
    }
    sd = (a == 0) ? sh : __fadd_rn(sd, sh);
  }
  l1[i] = sd;
}

// Exact k-th smallest of n non-negative floats: 4-pass 8-bit radix select on the IEEE bit patterns (which order
// like the values for non-negative floats), one workgroup, wavefront-level histogram via LDS atomics.
__global__ __launch_bounds__(1024) void kth_value_kernel(const float* __restrict__ v, int n, int k, float* __restrict__ out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_k;
  if (threadIdx.x == 0) { s_prefix = 0u; s_k = (unsigned)k; }
  unsigned mask = 0u;
  for (int pass = 0; pass < 4; ++pass) {
    const int sft = 24 - 8 * pass;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned prefix = s_prefix;
    for (int i = threadIdx.x; i < n; i += 1024) {
      const unsigned key = __float_as_uint(v[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> sft) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned kk = s_k, cum = 0u;
      int b = 0;
      for (; b < 256; ++b) {
        if (cum + hist[b] > kk) break;
        cum += hist[b];
      }
      if (b > 255) b = 255;
      s_k = kk - cum;
      s_prefix = prefix | ((unsigned)b << sft);
    }
    mask |= 255u << sft;
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = __uint_as_float(s_prefix);
}

__global__ __launch_bounds__(256) void death_kernel(const float* __restrict__ l1, const float* __restrict__ thr,
                                                    unsigned char* __restrict__ kmask, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (l1[i] <= *thr) kmask[i] = 0;
}

// growth_mode = 'gradient' (core_channel.py:771-790): score[r, c, a] = sum_kh( sum_kw |grad| ) of the kernels that are dead, 0 for the
// live ones -- the reference's TWO chained torch.sum(dim=-1) (the depth extent of the kernel is NOT summed: a transposed-conv weight
// [Cin, Cout, 2, 2, 2] has two scores per kernel), each taken left to right.  The gradient the reference reads is weight.grad AFTER
// clip_grad_norm_ scaled it in place (nnUNetTrainer_simple.py:573): with `sq` the fused optimizer's squared global norm the same
// coefficient is applied here (RN(g * coef), as torch's mul_), with sq == nullptr the tensor is taken as it is.
__global__ __launch_bounds__(256) void grad_score_kernel(const float* __restrict__ g, const double* __restrict__ sq, float max_norm,
                                                         const unsigned char* __restrict__ kmask, float* __restrict__ score,
                                                         long long n, int kd, int kh, int kw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // (r, c, a)
  if (i >= n * kd) return;
  float coef = 1.f;
  if (sq != nullptr) {
    coef = max_norm / ((float)sqrt(*sq) + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
  }
  const float* p = g + i * (kh * kw);
  float sh = 0.f;
  for (int b = 0; b < kh; ++b) {
    float sw = 0.f;
    for (int c = 0; c < kw; ++c) {
      const float v = fabsf(sq != nullptr ? __fmul_rn(p[b * kw + c], coef) : p[b * kw + c]);
      sw = (c == 0) ? v : __fadd_rn(sw, v);
    }
    sh = (b == 0) ? sw : __fadd_rn(sh, sw);
  }
  score[i] = __fmul_rn(sh, kmask[i / kd] ? 0.f : 1.f);               // data_sum * (mask_sum < 1).float()
}

// new_mask[idx] = 1 for every (r, c, a) whose score exceeds the threshold (strictly, :786-787)
__global__ __launch_bounds__(256) void grow_above_kernel(const float* __restrict__ score, const float* __restrict__ thr,
                                                         unsigned char* __restrict__ kmask, long long n, int kd) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * kd) return;
  if (score[i] > *thr) kmask[i / kd] = 1;
}

__global__ __launch_bounds__(256) void expand_mask_kernel(const unsigned char* __restrict__ kmask, float* __restrict__ mask,
                                                          long long n, int ks) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * ks) return;
  mask[i] = kmask[i / ks] ? 1.f : 0.f;
}

// bits[r][wd] bit b = kmask[r][wd*32 + b]
__global__ __launch_bounds__(256) void rows_bits_kernel(const unsigned char* __restrict__ kmask, unsigned* __restrict__ bits,
                                                        int R, int Cc) {
  const int words = e2e::cdiv(Cc, 32);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R * words) return;
  const int r = i / words, wd = i - r * words;
  unsigned b = 0u;
  for (int j = 0; j < 32; ++j) {
    const int c = wd * 32 + j;
    if (c < Cc && kmask[(long long)r * Cc + c]) b |= 1u << j;
  }
  bits[i] = b;
}
// bits[c][wd] bit b = kmask[wd*32 + b][c]
__global__ __launch_bounds__(256) void cols_bits_kernel(const unsigned char* __restrict__ kmask, unsigned* __restrict__ bits,
                                                        int R, int Cc) {
  const int words = e2e::cdiv(R, 32);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Cc * words) return;
  const int c = i / words, wd = i - c * words;
  unsigned b = 0u;
  for (int j = 0; j < 32; ++j) {
    const int r = wd * 32 + j;
    if (r < R && kmask[(long long)r * Cc + c]) b |= 1u << j;
  }
  bits[i] = b;
}

// quad words for the 1x3x3 conv kernels: word [r / 4][c / 8], bit (c % 8) * 4 + r % 4 = kmask[r][c] (transpose = 0),
// or word [c / 4][r / 8], bit (r % 8) * 4 + c % 4 (transpose = 1)
__global__ __launch_bounds__(256) void quad_bits_kernel(const unsigned char* __restrict__ kmask, unsigned* __restrict__ bits,
                                                        int R, int Cc, int transpose) {
  const int Qn = transpose ? Cc : R, Pn = transpose ? R : Cc;      // quad rows run over q, words over p
  const int words = e2e::cdiv(Pn, 8);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= e2e::cdiv(Qn, 4) * words) return;
  const int qg = i / words, wd = i - qg * words;
  unsigned b = 0u;
  for (int j = 0; j < 32; ++j) {
    const int q = qg * 4 + (j & 3), pp = wd * 8 + (j >> 2);
    if (q < Qn && pp < Pn) {
      const int r = transpose ? pp : q, c = transpose ? q : pp;
      if (kmask[(long long)r * Cc + c]) b |= 1u << j;
    }
  }
  bits[i] = b;
}

__global__ __launch_bounds__(256) void kmask_from_weights_kernel(const float* __restrict__ w, unsigned char* __restrict__ kmask,
                                                                 long long n, int ks) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  bool any = false;
  for (int t = 0; t < ks; ++t) any |= (w[i * ks + t] != 0.f);
  kmask[i] = any ? 1 : 0;
}
}  // namespace

extern "C" int e2e_dsff_kernel_l1(const float* w, float* l1, int R, int Cc, int kd, int kh, int kw, void* stream) {
  E2E_REQUIRE(w && l1 && R > 0 && Cc > 0 && kd > 0 && kh > 0 && kw > 0, "dsff_kernel_l1: bad arguments");
  const long long n = (long long)R * Cc;
  hipLaunchKernelGGL(kernel_l1_kernel, dim3((unsigned)e2e::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, w, l1, n, kd, kh, kw);
  return e2e::check_launch("kernel_l1_kernel");
}

extern "C" int e2e_dsff_kth_value(const float* v, int n, int k, float* out, void* ws, void* stream) {
  (void)ws;
  E2E_REQUIRE(v && out && n > 0 && k >= 0 && k < n, "dsff_kth_value: need 0 <= k < n");
  hipLaunchKernelGGL(kth_value_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, n, k, out);
  return e2e::check_launch("kth_value_kernel");
}

extern "C" int e2e_dsff_death(const float* l1, const float* thr, unsigned char* kmask, int n, void* stream) {
  E2E_REQUIRE(l1 && thr && kmask && n > 0, "dsff_death: bad arguments");
  hipLaunchKernelGGL(death_kernel, dim3(e2e::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, l1, thr, kmask, n);
  return e2e::check_launch("death_kernel");
}

extern "C" int e2e_dsff_grad_score(const float* grad, const double* sq_norm, float max_norm, const unsigned char* kmask, float* score,
                                   int R, int Cc, int kd, int kh, int kw, void* stream) {
  E2E_REQUIRE(grad && kmask && score && R > 0 && Cc > 0 && kd > 0 && kh > 0 && kw > 0, "dsff_grad_score: bad arguments");
  const long long n = (long long)R * Cc;
  hipLaunchKernelGGL(grad_score_kernel, dim3((unsigned)e2e::cdivll(n * kd, 256)), dim3(256), 0, (hipStream_t)stream, grad, sq_norm, max_norm,
                     kmask, score, n, kd, kh, kw);
  return e2e::check_launch("grad_score_kernel");
}

extern "C" int e2e_dsff_grow_above(const float* score, const float* thr, unsigned char* kmask, int R, int Cc, int kd, void* stream) {
  E2E_REQUIRE(score && thr && kmask && R > 0 && Cc > 0 && kd > 0, "dsff_grow_above: bad arguments");
  const long long n = (long long)R * Cc;
  hipLaunchKernelGGL(grow_above_kernel, dim3((unsigned)e2e::cdivll(n * kd, 256)), dim3(256), 0, (hipStream_t)stream, score, thr, kmask, n, kd);
  return e2e::check_launch("grow_above_kernel");
}

extern "C" int e2e_dsff_expand(const unsigned char* kmask, float* mask, unsigned* bits_rows, unsigned* bits_cols, int R,
                               int Cc, int ks, void* stream) {
  E2E_REQUIRE(kmask && R > 0 && Cc > 0 && ks > 0, "dsff_expand: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)R * Cc;
  if (mask) hipLaunchKernelGGL(expand_mask_kernel, dim3((unsigned)e2e::cdivll(n * ks, 256)), dim3(256), 0, st, kmask, mask, n, ks);
  if (bits_rows) hipLaunchKernelGGL(rows_bits_kernel, dim3(e2e::cdiv(R * e2e::cdiv(Cc, 32), 256)), dim3(256), 0, st, kmask, bits_rows, R, Cc);
  if (bits_cols) hipLaunchKernelGGL(cols_bits_kernel, dim3(e2e::cdiv(Cc * e2e::cdiv(R, 32), 256)), dim3(256), 0, st, kmask, bits_cols, R, Cc);
  return e2e::check_launch("dsff_expand");
}

extern "C" int e2e_dsff_expand_quads(const unsigned char* kmask, unsigned* quads_rows, unsigned* quads_cols, int R, int Cc,
                                     void* stream) {
  E2E_REQUIRE(kmask && R > 0 && Cc > 0, "dsff_expand_quads: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (quads_rows)
    hipLaunchKernelGGL(quad_bits_kernel, dim3(e2e::cdiv(e2e::cdiv(R, 4) * e2e::cdiv(Cc, 8), 256)), dim3(256), 0, st, kmask, quads_rows, R, Cc, 0);
  if (quads_cols)
    hipLaunchKernelGGL(quad_bits_kernel, dim3(e2e::cdiv(e2e::cdiv(Cc, 4) * e2e::cdiv(R, 8), 256)), dim3(256), 0, st, kmask, quads_cols, R, Cc, 1);
  return e2e::check_launch("dsff_expand_quads");
}

extern "C" int e2e_dsff_kmask_from_weights(const float* w, unsigned char* kmask, int R, int Cc, int ks, void* stream) {
  E2E_REQUIRE(w && kmask && R > 0 && Cc > 0 && ks > 0, "dsff_kmask_from_weights: bad arguments");
  const long long n = (long long)R * Cc;
  hipLaunchKernelGGL(kmask_from_weights_kernel, dim3((unsigned)e2e::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, w, kmask, n, ks);
  return e2e::check_launch("kmask_from_weights_kernel");
}
