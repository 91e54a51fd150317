// K8: softmax + soft-Dice + cross-entropy for one deep-supervision scale, forward and gradient (gfx950).
// Reference: DC_and_CE_loss (dice_loss.py:302-359) = RobustCrossEntropyLoss (crossentropy.py:4-12, mean over
// voxels) + SoftDiceLoss(softmax, batch_dice, do_bg=False, smooth) (dice_loss.py:156-192) with
// tp/fp/fn of get_tp_fp_fn_tn (dice_loss.py:100-153); the per-scale weight of MultipleOutputLoss2
// (deep_supervision.py:31-43) is folded into the gradient.
//   acc layout (fp64): [B][K][3] (tp, fp, fn) followed by one CE sum.
#include "e2e_common.h"

namespace {
constexpr int KMAX = 32;

template <int KB>
__global__ __launch_bounds__(256) void dc_ce_reduce_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                           double* __restrict__ acc, int K, long long spatial) {
  const int n = blockIdx.y;
  const float* lp = logits + (long long)n * K * spatial;
  const float* tp_ = target + (long long)n * spatial;
  float tp[KB], fp[KB], fn[KB];
#pragma unroll
  for (int k = 0; k < KB; ++k) { tp[k] = 0.f; fp[k] = 0.f; fn[k] = 0.f; }
  double dtp[KB], dfp[KB], dfn[KB];
#pragma unroll
  for (int k = 0; k < KB; ++k) { dtp[k] = 0.0; dfp[k] = 0.0; dfn[k] = 0.0; }
  double ce = 0.0;
  int it = 0;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < spatial; v += (long long)gridDim.x * 256) {
    float l[KB];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      l[k] = (k < K) ? lp[(long long)k * spatial + v] : -INFINITY;
      m = fmaxf(m, l[k]);
    }
    const int t = (int)tp_[v];
    // cross entropy as torch's log_softmax forms it: (l_t - max) - log(sum exp(l - max)); -log(p_t) would turn into +Inf where
    // p_t underflows (logits apart by more than ~100: large InstanceNorm weights) although the loss is finite
    float lt = (unsigned)t < (unsigned)K ? 0.f : NAN;     // label outside [0, K): torch's CrossEntropyLoss raises; here the loss turns NaN
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      if (k == t) lt = l[k] - m;
      l[k] = (k < K) ? expf(l[k] - m) : 0.f;
      s += l[k];
    }
    const float inv = 1.f / s;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const float pk = l[k] * inv;
      if (k == t) { tp[k] += pk; fn[k] += 1.f - pk; }
      else fp[k] += pk;
    }
    ce -= (double)(lt - logf(s));
    if ((++it & 31) == 0) {
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        dtp[k] += tp[k]; dfp[k] += fp[k]; dfn[k] += fn[k];
        tp[k] = 0.f; fp[k] = 0.f; fn[k] = 0.f;
      }
    }
  }
  __shared__ double sh[4][3 * KB + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const double a = e2e::wave_sum_d(dtp[k] + (double)tp[k]);
    const double b = e2e::wave_sum_d(dfp[k] + (double)fp[k]);
    const double c = e2e::wave_sum_d(dfn[k] + (double)fn[k]);
    if (lane == 0) { sh[wave][3 * k] = a; sh[wave][3 * k + 1] = b; sh[wave][3 * k + 2] = c; }
  }
  ce = e2e::wave_sum_d(ce);
  if (lane == 0) sh[wave][3 * KB] = ce;
  __syncthreads();
  if (threadIdx.x < 3 * K) {
    const int i = threadIdx.x;
    atomicAdd(&acc[(long long)n * K * 3 + i], sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i]);
  }
  if (threadIdx.x == 255) atomicAdd(&acc[(long long)gridDim.y * K * 3], sh[0][3 * KB] + sh[1][3 * KB] + sh[2][3 * KB] + sh[3][3 * KB]);
}

// gradient: dlogit_j = weight * [ (p_j - [t == j]) / (B * spatial) + p_j * (g_j - sum_k g_k p_k) ],
// g_k = dDiceLoss/dp_k = -(1/M) * ((t == k) ? (2*Dn - N)/Dn^2 : -N/Dn^2),  N = 2tp + s, Dn = 2tp + fp + fn + s + 1e-8,
// M = number of (sample, foreground class) dice terms (batch_dice: K-1, else B*(K-1)); class 0 has no dice term.
template <int KB>
__global__ __launch_bounds__(256) void dc_ce_grad_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                         const double* __restrict__ acc, float weight, int batch_dice,
                                                         float smooth, float* __restrict__ dlogits, float* __restrict__ loss_out,
                                                         int B, int K, long long spatial) {
  const int n = blockIdx.y;
  __shared__ float g_hit[KB], g_miss[KB];
  __shared__ float dice_sum_sh;
  if (threadIdx.x < KB) {
    const int k = threadIdx.x;
    float gh = 0.f, gm = 0.f;
    if (k >= 1 && k < K) {
      double tp = 0, fp = 0, fn = 0;
      if (batch_dice) {
        for (int b = 0; b < B; ++b) {
          tp += acc[((long long)b * K + k) * 3]; fp += acc[((long long)b * K + k) * 3 + 1]; fn += acc[((long long)b * K + k) * 3 + 2];
        }
      } else {
        tp = acc[((long long)n * K + k) * 3]; fp = acc[((long long)n * K + k) * 3 + 1]; fn = acc[((long long)n * K + k) * 3 + 2];
      }
      const double N = 2 * tp + smooth, Dn = 2 * tp + fp + fn + smooth + 1e-8;
      const double M = batch_dice ? (double)(K - 1) : (double)B * (K - 1);
      gh = (float)(-(2 * Dn - N) / (Dn * Dn) / M);
      gm = (float)(N / (Dn * Dn) / M);
    }
    g_hit[k] = gh;
    g_miss[k] = gm;
  }
  if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) {
    // loss value: weight * (mean CE - mean dice)
    double dsum = 0.0;
    if (batch_dice) {
      for (int k = 1; k < K; ++k) {
        double tp = 0, fp = 0, fn = 0;
        for (int b = 0; b < B; ++b) {
          tp += acc[((long long)b * K + k) * 3]; fp += acc[((long long)b * K + k) * 3 + 1]; fn += acc[((long long)b * K + k) * 3 + 2];
        }
        dsum += (2 * tp + smooth) / (2 * tp + fp + fn + smooth + 1e-8);
      }
      dsum /= (double)(K - 1);
    } else {
      for (int b = 0; b < B; ++b)
        for (int k = 1; k < K; ++k) {
          const double tp = acc[((long long)b * K + k) * 3], fp = acc[((long long)b * K + k) * 3 + 1], fn = acc[((long long)b * K + k) * 3 + 2];
          dsum += (2 * tp + smooth) / (2 * tp + fp + fn + smooth + 1e-8);
        }
      dsum /= (double)B * (K - 1);
    }
    const double ce = acc[(long long)B * K * 3] / ((double)B * (double)spatial);
    *loss_out += (float)(weight * (ce - dsum));
  }
  __syncthreads();
  if (dlogits == nullptr) return;                      // value only (validation batches)
  const float inv_cnt = 1.f / ((float)B * (float)spatial);
  const float* lp = logits + (long long)n * K * spatial;
  float* dp = dlogits + (long long)n * K * spatial;
  const float* tg = target + (long long)n * spatial;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < spatial; v += (long long)gridDim.x * 256) {
    float l[KB];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      l[k] = (k < K) ? lp[(long long)k * spatial + v] : -INFINITY;
      m = fmaxf(m, l[k]);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      l[k] = (k < K) ? expf(l[k] - m) : 0.f;
      s += l[k];
    }
    const float inv = 1.f / s;
    const int t = (int)tg[v];
    float dot = 0.f;
    float g[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      l[k] *= inv;
      g[k] = (k == t) ? g_hit[k] : g_miss[k];
      dot = fmaf(g[k], l[k], dot);
    }
#pragma unroll
    for (int k = 0; k < KB; ++k)
      if (k < K) dp[(long long)k * spatial + v] = weight * ((l[k] - (k == t ? 1.f : 0.f)) * inv_cnt + l[k] * (g[k] - dot));
  }
}
// batch dice across data-parallel ranks (reference nnUNetTrainerV2_DDP.py:263-268 gathers the per-sample numerators and
// denominators and sums them): fold the [B][K][3] rows into row 0 (rows 1.. zeroed) so that ONE small all-reduce of
// row 0 gives every rank the global tp/fp/fn; dc_ce_grad(batch_dice = 1) then sums the rows as before.
__global__ void dc_ce_fold_batch_kernel(double* __restrict__ acc, int B, int K) {
  const int i = threadIdx.x;
  if (i >= 3 * K) return;
  double s = 0.0;
  for (int b = 0; b < B; ++b) {
    s += acc[(long long)b * K * 3 + i];
    if (b > 0) acc[(long long)b * K * 3 + i] = 0.0;
  }
  acc[i] = s;
}

// online evaluation of a validation batch (reference nnUNetTrainer_simple.py:373-405): hard tp / fp / fn voxel counts per
// class of argmax(softmax(logits)) against the target, summed over the batch.  counts [K][3] (class 0 is filled too,
// the reference reports classes 1..K-1).
template <int KB>
__global__ __launch_bounds__(256) void online_eval_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                          unsigned long long* __restrict__ counts, int K, long long spatial) {
  __shared__ unsigned int h[3][KB];
  for (int i = threadIdx.x; i < 3 * KB; i += 256) (&h[0][0])[i] = 0u;
  __syncthreads();
  const int n = blockIdx.y;
  const float* lp = logits + (long long)n * K * spatial;
  const float* tg = target + (long long)n * spatial;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < spatial; v += (long long)gridDim.x * 256) {
    float l[KB];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      l[k] = (k < K) ? lp[(long long)k * spatial + v] : -INFINITY;
      m = fmaxf(m, l[k]);
    }
    // argmax of the softmax = first maximum of exp(l - m) (the common 1/sum factor keeps order and ties)
    int seg = 0;
    float best = -1.f;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const float ek = (k < K) ? expf(l[k] - m) : -1.f;
      if (ek > best) { best = ek; seg = k; }
    }
    const int t = (int)tg[v];
    if (seg == t) atomicAdd(&h[0][seg], 1u);
    else {
      atomicAdd(&h[1][seg], 1u);
      if ((unsigned)t < (unsigned)K) atomicAdd(&h[2][t], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * K; i += 256) {
    const int k = i / 3, j = i - 3 * k;
    if (h[j][k]) atomicAdd(&counts[i], (unsigned long long)h[j][k]);
  }
}
}  // namespace

#define DISPATCH_LK(K, ...)                                    \
  if ((K) <= 4) { constexpr int KB = 4; __VA_ARGS__; }         \
  else if ((K) <= 8) { constexpr int KB = 8; __VA_ARGS__; }    \
  else if ((K) <= 16) { constexpr int KB = 16; __VA_ARGS__; }  \
  else { constexpr int KB = 32; __VA_ARGS__; }

extern "C" long long e2e_loss_ws_bytes(int B, int K) { return ((long long)B * K * 3 + 1) * (long long)sizeof(double); }

static unsigned loss_blocks(long long spatial) {
  long long b = e2e::cdivll(spatial, 256 * 8);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (unsigned)b;
}

extern "C" int e2e_dc_ce_reduce(const float* logits, const float* target, void* acc, int B, int K, long long spatial,
                                void* stream) {
  E2E_REQUIRE(logits && target && acc, "dc_ce_reduce: null pointer");
  E2E_REQUIRE(B > 0 && K > 1 && K <= KMAX && spatial > 0, "dc_ce_reduce: need 2 <= K <= 32");
  hipStream_t st = (hipStream_t)stream;
  e2e::zero_async(acc, (size_t)e2e_loss_ws_bytes(B, K), st);
  dim3 grid(loss_blocks(spatial), B);
  DISPATCH_LK(K, hipLaunchKernelGGL((dc_ce_reduce_kernel<KB>), grid, dim3(256), 0, st, logits, target, (double*)acc, K, spatial));
  return e2e::check_launch("dc_ce_reduce_kernel");
}

extern "C" int e2e_dc_ce_grad(const float* logits, const float* target, const void* acc, float weight, int batch_dice,
                              float smooth, float* dlogits, float* loss_out, int B, int K, long long spatial,
                              void* stream) {
  E2E_REQUIRE(logits && target && acc && loss_out, "dc_ce_grad: null pointer");
  E2E_REQUIRE(B > 0 && K > 1 && K <= KMAX && spatial > 0, "dc_ce_grad: need 2 <= K <= 32");
  dim3 grid(dlogits ? loss_blocks(spatial) : 1u, dlogits ? B : 1);
  DISPATCH_LK(K, hipLaunchKernelGGL((dc_ce_grad_kernel<KB>), grid, dim3(256), 0, (hipStream_t)stream, logits, target,
                                    (const double*)acc, weight, batch_dice, smooth, dlogits, loss_out, B, K, spatial));
  return e2e::check_launch("dc_ce_grad_kernel");
}

extern "C" int e2e_dc_ce_fold_batch(void* acc, int B, int K, void* stream) {
  E2E_REQUIRE(acc, "dc_ce_fold_batch: null pointer");
  E2E_REQUIRE(B > 0 && K > 1 && K <= KMAX, "dc_ce_fold_batch: need 2 <= K <= 32");
  hipLaunchKernelGGL(dc_ce_fold_batch_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, (double*)acc, B, K);
  return e2e::check_launch("dc_ce_fold_batch_kernel");
}

extern "C" int e2e_online_eval_counts(const float* logits, const float* target, long long* counts, int B, int K,
                                      long long spatial, void* stream) {
  E2E_REQUIRE(logits && target && counts, "online_eval_counts: null pointer");
  E2E_REQUIRE(B > 0 && K > 1 && K <= KMAX && spatial > 0, "online_eval_counts: need 2 <= K <= 32");
  hipStream_t st = (hipStream_t)stream;
  e2e::zero_async(counts, (size_t)K * 3 * sizeof(long long), st);
  dim3 grid(loss_blocks(spatial), B);
  DISPATCH_LK(K, hipLaunchKernelGGL((online_eval_kernel<KB>), grid, dim3(256), 0, st, logits, target,
                                    (unsigned long long*)counts, K, spatial));
  return e2e::check_launch("online_eval_kernel");
}


// ---- deep-supervision targets: nearest-neighbour gather through per-axis index vectors (downsampling.py:87-107) ----
namespace {
__global__ __launch_bounds__(256) void ds_target_gather_kernel(const float* __restrict__ seg, float* __restrict__ out,
                                                               const int* __restrict__ idx_d, const int* __restrict__ idx_h,
                                                               const int* __restrict__ idx_w, int D, int H, int W, int d, int h,
                                                               int w, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % w);
  long long t = i / w;
  const int y = (int)(t % h);
  t /= h;
  const int z = (int)(t % d);
  const long long bc = t / d;
  out[i] = seg[((bc * D + idx_d[z]) * H + idx_h[y]) * W + idx_w[x]];
}
}  // namespace

extern "C" int e2e_ds_target_gather(const float* seg, float* out, const int* idx_d, const int* idx_h, const int* idx_w,
                                    int BC, int D, int H, int W, int d, int h, int w, void* stream) {
  E2E_REQUIRE(seg && out && idx_d && idx_h && idx_w, "ds_target_gather: null pointer");
  E2E_REQUIRE(BC > 0 && D > 0 && H > 0 && W > 0 && d > 0 && h > 0 && w > 0, "ds_target_gather: bad dims");
  const long long total = (long long)BC * d * h * w;
  hipLaunchKernelGGL(ds_target_gather_kernel, dim3((unsigned)e2e::cdivll(total, 256)), dim3(256), 0, (hipStream_t)stream, seg, out,
                     idx_d, idx_h, idx_w, D, H, W, d, h, w, total);
  return e2e::check_launch("ds_target_gather_kernel");
}
