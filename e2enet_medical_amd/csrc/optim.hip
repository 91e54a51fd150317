// K10: clip_grad_norm_ + SGD (momentum, Nesterov, weight decay) + DSFF mask as multi-tensor kernels (gfx950).
// Reference: torch.nn.utils.clip_grad_norm_(params, 12) and torch.optim.SGD(lr, weight_decay=3e-5, momentum=0.99,
// nesterov=True).step() (nnUNetTrainer_simple.py:573-574, :369-370) followed by Masking.apply_mask
// (core_channel.py:427-434: weight *= mask, momentum_buffer *= mask).
#include "e2e_common.h"

namespace {

// grid = (XB, n): XB blocks per tensor, float4 main part + scalar tail; blocks beyond a small tensor's size exit at once,
// the 2.6 M-element tensors get all XB blocks (a fixed 16-32 blocks per tensor left the big ones latency bound)
__global__ __launch_bounds__(256) void sqnorm_kernel(const e2e_param_t* __restrict__ table, double* __restrict__ out) {
  const e2e_param_t t = table[blockIdx.y];
  const bool vec = (reinterpret_cast<unsigned long long>(t.grad) & 15ull) == 0;      // gradients may be slices of a flat buffer
  const long long n4 = vec ? t.numel >> 2 : 0;
  float s = 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(t.grad);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 g = g4[i];
    s = fmaf(g.x, g.x, s); s = fmaf(g.y, g.y, s); s = fmaf(g.z, g.z, s); s = fmaf(g.w, g.w, s);
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < t.numel; i += (long long)gridDim.x * 256)
    s = fmaf(t.grad[i], t.grad[i], s);
  // (a thread sums at most a few hundred squares in fp32; everything across threads is fp64)
  const double d = e2e::wave_sum_d((double)s);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = sh[0] + sh[1] + sh[2] + sh[3];
    if (tot != 0.0) atomicAdd(out, tot);
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(const e2e_param_t* __restrict__ table, const double* __restrict__ sq,
                                                  float max_norm, float lr, float wd, float mom, int nesterov, int first) {
  const e2e_param_t t = table[blockIdx.y];
  // clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1, always multiplied in (fp32 like torch)
  const float total = (float)sqrt(*sq);
  float coef = max_norm / (total + 1e-6f);
  coef = coef > 1.f ? 1.f : coef;
  const bool vec = ((reinterpret_cast<unsigned long long>(t.param) | reinterpret_cast<unsigned long long>(t.grad) |
                     reinterpret_cast<unsigned long long>(t.momentum) | reinterpret_cast<unsigned long long>(t.mask)) & 15ull) == 0;
  const long long n4 = vec ? t.numel >> 2 : 0;
  auto upd = [&](float& p, float gr, float& buf, float m, bool has_mask) {
    float g = gr * coef;
    g = g + wd * p;
    buf = first ? g : buf * mom + g;
    const float step = nesterov ? g + mom * buf : buf;
    p = p - lr * step;
    if (has_mask) { p *= m; buf *= m; }
  };
  float4* p4 = reinterpret_cast<float4*>(t.param);
  float4* b4 = reinterpret_cast<float4*>(t.momentum);
  const float4* g4 = reinterpret_cast<const float4*>(t.grad);
  const float4* m4 = reinterpret_cast<const float4*>(t.mask);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 p = p4[i];
    const float4 g = g4[i];
    float4 b = first ? make_float4(0.f, 0.f, 0.f, 0.f) : b4[i];
    const float4 m = t.mask ? m4[i] : make_float4(1.f, 1.f, 1.f, 1.f);
    upd(p.x, g.x, b.x, m.x, t.mask != nullptr);
    upd(p.y, g.y, b.y, m.y, t.mask != nullptr);
    upd(p.z, g.z, b.z, m.z, t.mask != nullptr);
    upd(p.w, g.w, b.w, m.w, t.mask != nullptr);
    p4[i] = p;
    b4[i] = b;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < t.numel; i += (long long)gridDim.x * 256) {
      float p = t.param[i], b = first ? 0.f : t.momentum[i];
      upd(p, t.grad[i], b, t.mask ? t.mask[i] : 1.f, t.mask != nullptr);
      t.param[i] = p;
      t.momentum[i] = b;
    }
}

__global__ __launch_bounds__(256) void apply_mask_kernel(const e2e_param_t* __restrict__ table) {
  const e2e_param_t t = table[blockIdx.y];
  if (!t.mask) return;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < t.numel; i += (long long)gridDim.x * 256) {
    const float m = t.mask[i];
    t.param[i] *= m;
    if (t.momentum) t.momentum[i] *= m;
  }
}
}  // namespace

extern "C" int e2e_grad_sqnorm(const e2e_param_t* table, int n, double* sq_out, void* stream) {
  E2E_REQUIRE(table && sq_out && n > 0, "grad_sqnorm: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  e2e::zero_async(sq_out, sizeof(double), st);
  hipLaunchKernelGGL(sqnorm_kernel, dim3(128, n), dim3(256), 0, st, table, sq_out);
  return e2e::check_launch("sqnorm_kernel");
}

extern "C" int e2e_sgd_clip_mask_step(const e2e_param_t* table, int n, const double* sq_norm, float max_norm, float lr,
                                      float weight_decay, float momentum, int nesterov, int first_step, void* stream) {
  E2E_REQUIRE(table && sq_norm && n > 0, "sgd_clip_mask_step: bad arguments");
  hipLaunchKernelGGL(sgd_kernel, dim3(128, n), dim3(256), 0, (hipStream_t)stream, table, sq_norm, max_norm, lr, weight_decay,
                     momentum, nesterov, first_step);
  return e2e::check_launch("sgd_kernel");
}

extern "C" int e2e_apply_mask(const e2e_param_t* table, int n, void* stream) {
  E2E_REQUIRE(table && n > 0, "apply_mask: bad arguments");
  hipLaunchKernelGGL(apply_mask_kernel, dim3(32, n), dim3(256), 0, (hipStream_t)stream, table);
  return e2e::check_launch("apply_mask_kernel");
}
