/*
 * e2e_hip.h -- C ABI of libe2e_hip.so: the MI355X (gfx950) kernels of the E2ENet
 * shiftConvPP + DSFF hot path.
 *
 * The reference (boqian333/E2ENet-Medical) has no FFI: its hot path bottoms out in
 * PyTorch operators.  Every entry point below names the reference call site(s) whose
 * arithmetic it replaces (paths relative to the reference repo root).  Conventions:
 *   - all pointers are DEVICE pointers unless marked host; fp32 NCDHW activations;
 *   - no ownership transfer: the caller (PyTorch-ROCm tensors) owns every buffer,
 *     workspaces are passed in; nothing is allocated or synchronised inside;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it and
 *     graph-capturable;
 *   - return 0 on success, negative E2E_ERR_* otherwise; e2e_last_error() returns a
 *     thread-local message.
 */
#ifndef E2E_HIP_H
#define E2E_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define E2E_OK 0
#define E2E_ERR_ARG (-1)
#define E2E_ERR_LAUNCH (-2)
#define E2E_ERR_UNSUPPORTED (-3)

const char* e2e_last_error(void);
int e2e_abi_version(void);
/* Diagnostics: name and launch shape of the kernel variant the last conv133 / convT entry point called on this
 * thread selected (size-dependent dispatch; the parity tests assert that the benchmarked shapes reach the intended
 * variants).  Thread-local, valid until the next call. */
const char* e2e_last_kernel(void);

/* One input plane (channel) of a convolution whose input is the *virtual* concatenation
 * [skip, up, down] (unetpp_d.py:453-478) followed by the restricted depth shift
 * (unetpp_d.py:45-59).  Neither the concat nor the shift is ever materialised: the load
 * stage reads plane `ptr` at depth d - dshift and applies the producer's InstanceNorm
 * affine + LeakyReLU on the fly (unetpp_d.py:111).                                      */
typedef struct {
  const float* ptr;     /* element [n=0, this channel, 0,0,0] of the source tensor        */
  const float* scale;   /* IN scale for (n=0, channel); NULL => raw source (no transform)  */
  const float* shift;   /* IN shift, same indexing                                        */
  long long nstride;    /* elements between batch items of the source tensor              */
  int ab_nstride;       /* elements between batch items of scale/shift (= source C)       */
  int dshift;           /* s(c): shifted[d] = x[d - s], zero outside                      */
  float slope;          /* LeakyReLU negative slope applied after the affine (1 = none)   */
  int reserved;
} e2e_in_chan_t;

/* One output plane of a scatter epilogue (conv dgrad): the gradient of virtual-concat
 * channel c computed at depth d is stored at depth d - dshift of `ptr` (un-shift on
 * store), `accumulate` != 0 adds to what is there.                                      */
typedef struct {
  float* ptr;
  long long nstride;
  int dshift;
  int accumulate;
} e2e_out_chan_t;

/* One channel of a fused InstanceNorm-backward reduction (round 4): the LAST writer of a gradient buffer holds the final dz of
 * every element it stores and writes, per 16 x 32 tile of every depth slice, this channel's  sum dz * lrelu'(u)  and
 * sum dz * lrelu'(u) * xhat  (u = scale * y + shift, xhat = (y - mean) * rstd; autograd of unetpp_d.py:99-100, :111) -- the
 * first pass of e2e_in_lrelu_bwd, which then only adds the tile records up (fixed order: deterministic) and runs its apply pass.
 * Records: part[((n * C + c) * np + tile) * 2 + {0, 1}], np = e2e_conv133_num_partials(D, H, W, 1, 1), every record of a
 * channel is written exactly once per launch (plain stores).  y == NULL: nothing to do for this channel.                    */
typedef struct {
  const float* y;        /* pre-norm plane of this channel (batch item 0) */
  const float* scale;    /* per-(n, c) coefficients of the channel at n = 0 ... */
  const float* shift;
  const float* mean;
  const float* rstd;
  double* part;          /* tile records of (n = 0, c) */
  long long nstride;     /* floats between batch items of y */
  long long part_nstride;   /* doubles between batch items of part (= 2 * C * np) */
  int ab_nstride;        /* elements between batch items of the coefficients (= C) */
  float slope;
} e2e_in_sum_chan_t;

/* ---- K1: 1x3x3 convolution, forward ------------------------------------------------
 * Replaces: torch_shift.forward (unetpp_d.py:45-59) + torch.cat (unetpp_d.py:453-478) +
 * nn.Conv3d k(1,3,3) pad(0,1,1) stride (sd,sh,sw) bias (unetpp_d.py:93,108) and the
 * statistics pass of nn.InstanceNorm3d (unetpp_d.py:99,111).
 *   chans   [Cin]            device table (see e2e_in_chan_t)
 *   w       [Cout,Cin,1,3,3] (DSFF-masked weights; dead kernels are exact zeros)
 *   live    quad words [ceil(Cout/4), ceil(Cin/8)] (quads_rows of e2e_dsff_expand_quads): bit (c%8)*4 + o%4 of
 *           word [o/4][c/8] set <=> kernel (o,c) is alive; NULL => dense
 *   y       [B,Cout,Do,Ho,Wo] pre-norm output, Do=(Di-1)/sd+1, Ho=(Hi-1)/sh+1, ...
 *   part    [B,Cout,np,3] per-tile (count, mean, M2) partials, fp64 (ABI 9; fp32 before), np =
 *           e2e_conv133_num_partials(Do,Ho,Wo,sh,sw); NULL => no statistics
 */
int e2e_conv133_num_partials(int Do, int Ho, int Wo, int sh, int sw);
int e2e_conv133_fwd(const e2e_in_chan_t* chans, int Cin, const float* w, const float* bias,
                    const unsigned* live, float* y, double* part, int B, int Cout, int Di, int Hi,
                    int Wi, int sd, int sh, int sw, void* stream);

/* Forward with a workspace: on the deep levels (planes no larger than a 16 x 16 tile, hundreds of input planes) the
 * input-plane chunks of a tile are split over several workgroups that store raw partial sums in `ws`
 * ([parts][B][Cout][Do][Ho][Wo]); a second kernel adds them up in a fixed order (deterministic), adds the bias and
 * writes the InstanceNorm partials.  e2e_conv133_fwd_ws_bytes returns the bytes this shape wants (0: it does not
 * split and e2e_conv133_fwd_splitk behaves exactly like e2e_conv133_fwd).                                              */
long long e2e_conv133_fwd_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw);
int e2e_conv133_fwd_splitk(const e2e_in_chan_t* chans, int Cin, const float* w, const float* bias,
                           const unsigned* live, float* y, double* part, int B, int Cout, int Di, int Hi, int Wi,
                           int sd, int sh, int sw, float* ws, long long ws_bytes, void* stream);


/* ---- K6a: 1x3x3 convolution, data gradient ------------------------------------------
 * Replaces: autograd of the Conv3d + cat + torch_shift chain w.r.t. its inputs
 * (nnUNetTrainer_simple.py:572 l.backward()).
 *   dy    [B,Cout,Do,Ho,Wo] gradient w.r.t. the pre-norm conv output
 *   outs  [Cin] destinations, one per virtual-concat input channel (un-shift on store)
 *   live_t quad words [ceil(Cin/4), ceil(Cout/8)] (quads_cols of e2e_dsff_expand_quads); NULL => dense
 */
int e2e_conv133_dgrad(const float* dy, const float* w, const unsigned* live_t, const e2e_out_chan_t* outs,
                      int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw, void* stream);

/* Data gradient with a workspace: the split-K form of the deep levels (see e2e_conv133_fwd_splitk); the parts hold raw
 * sums per virtual-concat channel, a second kernel adds them in a fixed order and applies the scatter epilogue (un-shift
 * on store, zero-fill, accumulate flag).  ws_bytes from e2e_conv133_dgrad_ws_bytes (0: this shape does not split).      */
long long e2e_conv133_dgrad_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw);
int e2e_conv133_dgrad_splitk(const float* dy, const float* w, const unsigned* live_t, const e2e_out_chan_t* outs,
                             int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw, float* ws,
                             long long ws_bytes, void* stream);


/* ---- K1s: DSFF-masked 1x3x3 convolution on a load-balanced plan (conv133_sparse.hip) ---------------------------------
 * The same operators as e2e_conv133_fwd / e2e_conv133_dgrad (unetpp_d.py:45-59, :453-478, :93/:108 and their autograd; the
 * mask is Masking's, core_channel.py:427-434) for stride-1 layers with planes wider than 16 voxels, Wi % 4 == 0 and more than
 * 8 channels on both sides (e2e_conv133_sparse_eligible).  The order in which input planes are chunked and the assignment of
 * output planes to the eight waves of a workgroup are chosen on the HOST from the kernel map so that every wave carries the same
 * work in every chunk; weights are read from a packed copy in that order.
 *   e2e_conv133_sparse_plan    host function.  kmask [R][Cc] uint8 in HOST memory (weight dims 0 and 1); transpose 0: forward
 *       (Q = R output planes, P = Cc input planes), 1: data gradient (Q = Cc virtual-concat channels, P = R dy channels).  Writes
 *       (host arrays) qslot [groups*32] (output plane of slot wave*4 + a, -1 empty), pslot [groups*nchunks*8] (input plane of each
 *       chunk slot, -1 empty; the same chunking for every group: they share the staged planes through L2), quads
 *       [groups*8*nchunks] (bit cl*4 + a of word [group][wave][chunk] <=> kernel alive), woff [groups*nchunks*8] (first slot of
 *       each wave's kernel list inside the chunk's packed block), kmax (slots of a block = the most live kernels of any chunk) and
 *       flush_every (chunks per flush of the two-level summation); groups = ceil(Q/32), nchunks = ceil(P/8).  A pure function of
 *       the kernel map (every data-parallel rank derives the same plan).
 *   e2e_conv133_sparse_pack    device: packed weights [groups][nchunks][kmax][12] -- only the LIVE kernels, wave by wave in walk
 *       order -- for a TABLE of jobs in one launch (run it after every optimizer step / parameter load).  reverse = 1 flips the
 *       taps (data gradient).
 *   e2e_conv133_fwd_sparse     chans_plan [groups][nchunks*8]: the e2e_in_chan_t of pslot's planes (ptr NULL for empty slots)
 *   e2e_conv133_dgrad_sparse   pslot_t as returned by the plan (dy channels); outs_plan [groups][32]: the e2e_out_chan_t of
 *       qslot's channels (ptr NULL for empty slots); insum_plan (optional, same order): channels whose gradient buffer this
 *       launch writes LAST also get their InstanceNorm-backward sums here (e2e_in_sum_chan_t)                                                                          */
typedef struct {
  const float* w;            /* the layer's weight tensor */
  float* wpk;                /* e2e_conv133_sparse_wpk_floats(P, Q) floats */
  const int* qslot;          /* device copies of the plan arrays */
  const int* pslot;
  const unsigned* quads;
  const int* woff;
  int groups, nchunks;
  int wq_stride, wp_stride;  /* element strides of w for (output plane, input plane): forward (Cin*9, 9), data gradient (9, Cin*9) */
  int reverse;
  int kmax;
} e2e_sparse_pack_job_t;
int e2e_conv133_sparse_eligible(int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw);
long long e2e_conv133_sparse_wpk_floats(int P, int Q, int kmax);
int e2e_conv133_sparse_plan(const unsigned char* kmask_host, int R, int Cc, int transpose, int* qslot, int* pslot, unsigned* quads,
                            int* woff, int* kmax, int* flush_every);
int e2e_conv133_sparse_pack(const e2e_sparse_pack_job_t* jobs, int njobs, long long max_threads, void* stream);
int e2e_conv133_fwd_sparse(const e2e_in_chan_t* chans_plan, int Cin, const float* wpk, const float* bias, const unsigned* quads,
                           const int* woff, int kmax, const int* qslot, int flush_every, float* y, double* part, int B, int Cout,
                           int Di, int Hi, int Wi, void* stream);
int e2e_conv133_dgrad_sparse(const float* dy, const float* wpk_t, const unsigned* quads_t, const int* woff_t, int kmax_t,
                             const int* pslot_t, const e2e_out_chan_t* outs_plan, const e2e_in_sum_chan_t* insum_plan /* [groups][32] or NULL */,
                             int flush_every, int B, int Cin, int Cout, int Di, int Hi, int Wi, void* stream);

/* ---- K1m (round 5): the same operators as e2e_conv133_fwd / e2e_conv133_dgrad (unetpp_d.py:45-59, :453-478, :93/:108 and their
 * autograd; DSFF-pruned kernels contribute nothing: they are packed as zeros from `live`, core_channel.py:427-434) as a GEMM on the
 * fp16 matrix pipe with fp32-exact two-piece operands (three matrix products per fp32 product; conv133_mm.hip) -- every stride-1
 * layer with Wi % 32 == 0, Hi % 16 == 0, Hi > 16 and 17..320 channels on both sides, masked or not.
 *   wpk         the layer's packed weights of this direction: a buffer of e2e_conv133_mm_ws_bytes(...) bytes (0: shape not served)
 *               that e2e_conv133_mm_pack filled -- ONE launch pair for all layers and both directions, after an optimizer step or a
 *               parameter load (round 6; until round 5 every conv launch re-packed its weights)
 *   w_absmax    the device word e2e_conv133_mm_pack recorded max |w| in (its power-of-two scale); NULL: packed with the fixed 2^8
 *   x_absmax    NULL, or a device word holding (a bound of) max |x| over all input planes after normalise-on-load
 *               (e2e_conv133_input_ranges derives one from the parameters; e2e_absmax_word measures one): the activations are
 *               moved into the fp16 range by the power of two it implies.  NULL keeps the fixed 2^3 of round 5: |x| > 8188 -> Inf
 *   dy_absmax   NULL, or the word e2e_in_lrelu_bwd(dy_absmax) left behind for this dy: its power-of-two scale (without it dy is
 *               taken as it is: only for O(1) test data -- full-resolution gradients of 1e-7 are below the fp16 range)
 *   outs        as e2e_conv133_dgrad; destinations with `accumulate` set are updated by no-return global_atomic_add_f32 -- every element
 *               by exactly one lane per launch, launches stream-ordered: the same value as a load / add / store and deterministic;
 *               the buffers must be device (coarse-grained) memory */
typedef struct {
  const float* w;            /* the layer's weight tensor [Cout, Cin, 1, 3, 3] */
  const unsigned* quads;     /* DSFF liveness quad words of this direction (e2e_dsff_expand_quads) or NULL (dense) */
  void* wpk;                 /* e2e_conv133_mm_ws_bytes(...) bytes */
  unsigned* w_absmax;        /* one word per layer (shared by its two directions) */
  int P, Q;                  /* reduction-side / output-side channels: forward (Cin, Cout), data gradient (Cout, Cin) */
  int wq_stride, wp_stride;  /* element strides of w for (output-side, reduction-side) channel: forward (Cin*9, 9), data gradient (9, Cin*9) */
  int reverse;               /* data gradient: taps reversed */
  int owns_absmax;           /* this job computes *w_absmax (one job per layer) */
} e2e_mm_pack_job_t;
typedef struct {
  int kind;                  /* 0 none, 1 normalised source (or its max-pool), 2 transposed conv of a normalised source, 3 measured word,
                              * 4 max |w| over N floats, 5 max over channels [C, C + wks) of sum_{o, tap} |w[o, c, tap]|, w [wCout][N][9] */
  int C;                     /* channels of the normalised tensor */
  long long N;               /* voxels per instance of the normalised tensor */
  const float* gamma;        /* [C] instnorm.weight of its producer */
  const float* beta;         /* [C] instnorm.bias */
  const float* w;            /* kind 2: transposed-conv weight [C, wCout, wks] */
  int wCout, wks;
  const unsigned* word;      /* kind 3 */
} e2e_range_src_t;
typedef struct {
  e2e_range_src_t src[3];    /* the concat sources of one conv (reference unetpp_d.py:453-478) */
  unsigned* out;             /* bit pattern of the bound of |x| over all of them */
} e2e_range_job_t;
long long e2e_conv133_mm_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw);
/* jobs: DEVICE table; max_elems >= the largest ceil(Q/32)*ceil(P/16)*9*512 of the table */
int e2e_conv133_mm_pack(const e2e_mm_pack_job_t* jobs, int njobs, long long max_elems, void* stream);
/* jobs: DEVICE table, one per range word (a conv's input planes; a transposed conv's activations, weights, dy factor);
 * ws: e2e_conv133_input_ranges_ws_bytes(njobs) bytes of scratch (two launches: partial maxima, fold) */
long long e2e_conv133_input_ranges_ws_bytes(int njobs);
int e2e_conv133_input_ranges(const e2e_range_job_t* jobs, int njobs, void* ws, void* stream);
/* *word = bit pattern of max |x| over n floats (a non-finite element gives +Inf) */
int e2e_absmax_word(const float* x, long long n, unsigned* word, void* stream);
int e2e_conv133_fwd_mm(const e2e_in_chan_t* chans, int Cin, const void* wpk, const unsigned* w_absmax, const float* bias,
                       const unsigned* x_absmax, float* y, double* part, int B, int Cout, int Di, int Hi, int Wi, void* stream);
int e2e_conv133_dgrad_mm(const float* dy, const unsigned* dy_absmax, const void* wpk_t, const unsigned* w_absmax,
                         const e2e_out_chan_t* outs, int B, int Cin, int Cout, int Di, int Hi, int Wi, void* stream);

/* ---- K6b: 1x3x3 convolution, weight gradient (dense: also for dead kernels, because the
 * reference's clip_grad_norm_ runs over all gradients, nnUNetTrainer_simple.py:573) ----
 *   dw   [Cout,Cin,1,3,3] (overwritten)
 *   ws   workspace of e2e_conv133_wgrad_ws_bytes(...) bytes
 */
long long e2e_conv133_wgrad_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw);
/*   dy_absmax  NULL, or the device word e2e_in_lrelu_bwd(dy_absmax) left behind for this dy (bit pattern of max |dy|).  With it,
 *              the shapes served by the matrix-pipe kernel run on fp16 two-piece operands (three products per fp32 product, dy
 *              pre-scaled by the power of two that puts its maximum in [2^14, 2^15)); without it on bf16 three-piece operands
 *              (six products, no range assumption).  Both are fp32-exact to the last two bits of an operand. */
/*   x_absmax   NULL, or the word e2e_conv133_input_ranges / e2e_absmax_word produced for this conv's input planes: the power-of-two
 *              scale of the activations in the fp16 two-piece kernel (NULL: the fixed 2^3 of round 5, |x| > 8188 -> Inf) */
int e2e_conv133_wgrad(const e2e_in_chan_t* chans, const float* dy, float* dw, void* ws, int B, int Cin,
                      int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw, const unsigned* dy_absmax,
                      const unsigned* x_absmax, void* stream);

/* Diagnostic (the numerics gate of the split-operand matrix paths, tests/test_gpu_ops.py): D[32][32] = A[32][K] Bt[32][K]^T through the
 * split functions and product orders of the matrix-pipe kernels.  mode 0: bf16 three-piece operands, six products; 1: fp16 two-piece
 * operands, three products, Bt pre-scaled by the power of two derived from *absmax_b (NULL: unscaled) exactly as dy is in
 * e2e_conv133_wgrad; 2: fp32-input MFMA (an fp32 FMA chain).  K % 16 == 0.  No reference counterpart (torch computes in fp32). */
int e2e_diag_split_gemm(const float* A, const float* Bt, float* D, int K, int mode, const unsigned* absmax_b, void* stream);

/* Diagnostic: the shader clock the hot kernels ran at.  Workgroup 0 of every launch of family 0 (K1m, conv133_mm) / 1 (the
 * matrix-pipe K1w, conv133_wgrad v5) adds its s_memtime span (shader-clock cycles) and its s_memrealtime span (100 MHz) to a
 * device-side pair; this call synchronises the device and returns *mhz = cycles / time over all launches since the last reset
 * (0 if none) and *busy_ms = that time (may be NULL).  bench.py prices roofline.frac_at_measured_clock with it.  No reference
 * counterpart. */
int e2e_diag_kernel_clock(int family, double* mhz, double* busy_ms, int reset);

/* ---- K2: InstanceNorm statistics finalize --------------------------------------------
 * Replaces: nn.InstanceNorm3d(eps, affine, instance statistics) (unetpp_d.py:99,111):
 * combines the per-tile partials (Chan) in fp64 and emits per-(n,c)
 *   scale = gamma * rstd, shift = beta - mean * gamma * rstd   (consumed on load), mean, rstd.
 */
int e2e_in_stats_finalize(const double* part, int np, const float* gamma, const float* beta, float eps,
                          float* scale, float* shift, float* mean, float* rstd, int B, int C, void* stream);

/* ---- K7: InstanceNorm + LeakyReLU backward -------------------------------------------
 * Given dz = dL/d(lrelu(IN(y))) and the saved pre-norm y, overwrite dz with dy = dL/dy and
 * produce dgamma, dbeta (accumulated over the batch) and dbias = sum(dy).
 *   sums  workspace of e2e_in_lrelu_bwd_ws_doubles(B, C) doubles (s1 = sum du, s2 = sum du*xhat per (n, c); the per-block
 *         records of the first pass; the per-block sum dy records of the apply pass); no state survives a call (round 6: every
 *         sum is formed from per-block records in a fixed order -- no atomics, no zeroing launch, bit-reproducible gradients)
 *   tile_sums  NULL: the first pass (s1, s2) runs here.  Otherwise the per-tile records [B][C][np][2] that the last writers of
 *              dz have produced (e2e_in_sum_chan_t): they are added up in a fixed order and only the apply pass runs
 *   dy_absmax  NULL, or one device word that receives the bit pattern of max |dy| over the tensor (consumed by e2e_conv133_wgrad)
 *   scale, shift  the arrays e2e_in_stats_finalize produced for this tensor: the LeakyReLU branch of an element is decided from
 *              u = fma(y, scale, shift) -- bit for bit the value every forward consumer formed (normalise-on-load), so the backward
 *              differentiates exactly the function the forward evaluated (until round 5 it was re-derived as gamma * xhat + beta,
 *              whose sign differs for |u| ~ 1e-7: found by the same-branch gradient check of tests/test_gpu_configs.py)
 */
long long e2e_in_lrelu_bwd_ws_doubles(int B, int C);
int e2e_in_lrelu_bwd(float* dz_dy, const float* y, const float* mean, const float* rstd, const float* scale, const float* shift,
                     const float* gamma, float slope, float* dgamma, float* dbeta, float* dbias, float* sums,
                     int B, int C, long long spatial, const double* tile_sums, int np, unsigned* dy_absmax, void* stream);

/* ---- K3: transposed convolution, kernel == stride in {1,2}^3, no bias ------------------
 * Replaces: nn.ConvTranspose3d(Cin,Cout,k,k,bias=False) (unetpp_d.py:521-522).
 *   x   [B,Cin,D,H,W] pre-norm producer output + (scale,shift) [B,Cin] (NULL => raw), slope
 *   w   [Cin,Cout,kd,kh,kw];  live [Cout, ceil(Cin/32)] bits or NULL
 *   y   [B,Cout,D*kd,H*kh,W*kw]
 * Operand ranges (round 6): the matrix-pipe kernels (kw == 2, W % 32 == 0 ...) run on fp16 TWO-piece operands -- three matrix
 * products per fp32 product instead of the six of the bf16 three-piece form -- when the caller hands over the device words that
 * bound their operands (e2e_conv133_input_ranges: x_absmax = kind 1 of the source, w_absmax = kind 4 of w, dy_bound_b = kind 5 of
 * the conv that consumes y, dy_bound_a = that conv's dy_absmax word of e2e_in_lrelu_bwd; the bound of dy is their product); with
 * NULL words (or E2E_CT_H2=0) they keep the bf16 form, which needs no range. */
int e2e_convT_fwd(const float* x, const float* scale, const float* shift, float slope, const float* w,
                  const unsigned* live, float* y, int B, int Cin, int Cout, int D, int H, int W, int kd,
                  int kh, int kw, const unsigned* x_absmax, const unsigned* w_absmax, void* stream);
/* dgrad: dx (gradient w.r.t. the *post-activation* input) [B,Cin,D,H,W]; accumulate != 0 adds.
 * live_t [Cin, ceil(Cout/32)] or NULL. */
int e2e_convT_dgrad(const float* dy, const float* w, const unsigned* live_t, float* dx, int accumulate, int B,
                    int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw, const unsigned* w_absmax,
                    const unsigned* dy_bound_a, const unsigned* dy_bound_b, void* stream);
/* wgrad (dense): dw [Cin,Cout,kd,kh,kw] overwritten; ws of e2e_convT_wgrad_ws_bytes bytes. */
long long e2e_convT_wgrad_ws_bytes(int B, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw);
int e2e_convT_wgrad(const float* x, const float* scale, const float* shift, float slope, const float* dy,
                    float* dw, void* ws, int B, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw,
                    const unsigned* x_absmax, const unsigned* dy_bound_a, const unsigned* dy_bound_b, void* stream);

/* ---- K4: max pooling, kernel == stride ------------------------------------------------
 * Replaces: nn.MaxPool3d(k) (unetpp_d.py:523-524) on a normalise-on-load source. */
int e2e_maxpool_fwd(const float* x, const float* scale, const float* shift, float slope, float* y, int B, int C,
                    int D, int H, int W, int kd, int kh, int kw, void* stream);
/* dx [B,C,D,H,W] (w.r.t. the post-activation input; first maximum in scan order wins, as ATen). */
/* mean / rstd / tile_sums (all NULL: plain pooling backward): when this launch writes dx LAST it also forms the per-block records
 * of the source's InstanceNorm-backward sums (see e2e_in_sum_chan_t; records [B*C][e2e_maxpool_bwd_num_records(..)][2], consumed
 * by e2e_in_lrelu_bwd(tile_sums, np)); e2e_maxpool_bwd_num_records returns 0 for shapes that cannot carry them.            */
int e2e_maxpool_bwd_num_records(int D, int H, int W, int kd, int kh, int kw);
int e2e_maxpool_bwd(const float* x, const float* scale, const float* shift, float slope, const float* dy,
                    float* dx, int accumulate, int B, int C, int D, int H, int W, int kd, int kh, int kw,
                    const float* mean, const float* rstd, double* tile_sums, void* stream);

/* ---- K5: 1x1x1 segmentation head, no bias ----------------------------------------------
 * Replaces: nn.Conv3d(C,K,1,bias=False) (unetpp_d.py:394-401, used :480-483). */
int e2e_head1x1_fwd(const float* x, const float* scale, const float* shift, float slope, const float* w,
                    float* logits, int B, int C, int K, long long spatial, void* stream);
int e2e_head1x1_dgrad(const float* dlogits, const float* w, float* dx, int accumulate, int B, int C, int K,
                      long long spatial, void* stream);
long long e2e_head1x1_wgrad_ws_bytes(int B, int C, int K, long long spatial);
int e2e_head1x1_wgrad(const float* x, const float* scale, const float* shift, float slope,
                      const float* dlogits, float* dw, void* ws, int B, int C, int K, long long spatial,
                      void* stream);

/* ---- K8: softmax + soft-Dice + cross-entropy, forward and gradient ----------------------
 * Replaces: DC_and_CE_loss (dice_loss.py:302-359, :156-192, :100-153, crossentropy.py:4-12) for one
 * deep-supervision scale; MultipleOutputLoss2 weights (deep_supervision.py:31-43) enter as `weight`.
 *   target  [B,1,spatial] float labels;  acc [B,K,3]+[1] fp64 workspace (tp,fp,fn ; ce sum) zeroed by caller
 *   loss_out: *loss_out += weight * (ce + dice)   (device scalar, fp32)
 *   dlogits [B,K,spatial] = weight * dL/dlogits
 */
long long e2e_loss_ws_bytes(int B, int K);
int e2e_dc_ce_reduce(const float* logits, const float* target, void* acc, int B, int K, long long spatial,
                     void* stream);
int e2e_dc_ce_grad(const float* logits, const float* target, const void* acc, float weight, int batch_dice,
                   float smooth, float* dlogits, float* loss_out, int B, int K, long long spatial, void* stream);
/* dlogits may be NULL: loss value only (validation batches, nnUNetTrainer_simple.py:980-988).  A label outside [0, K)
 * turns the loss NaN (torch's CrossEntropyLoss raises).
 * Data-parallel batch dice (reference nnUNetTrainerV2_DDP.py:263-268 all-gathers the per-sample dice numerators and
 * denominators): e2e_dc_ce_fold_batch folds acc's [B][K][3] rows into row 0 (rows 1.. zeroed); the host all-reduces the
 * 3*K doubles of row 0 over the ranks (RCCL) between e2e_dc_ce_reduce and e2e_dc_ce_grad(batch_dice = 1).           */
int e2e_dc_ce_fold_batch(void* acc, int B, int K, void* stream);
/* Online evaluation of a validation batch (nnUNetTrainer_simple.py:373-405): hard tp / fp / fn voxel counts of
 * argmax(softmax(logits)) against the target per class, summed over the batch.  counts [K][3] int64 (zeroed by callee). */
int e2e_online_eval_counts(const float* logits, const float* target, long long* counts, int B, int K,
                           long long spatial, void* stream);

/* Deep-supervision targets (the loss-side end of the input feed): reference
 * e2enet/training/data_augmentation/downsampling.py:87-107 (downsample_seg_for_ds_transform2, order 0) resizes the label
 * map to every deep-supervision scale with batchgenerators' resize_segmentation -> skimage.transform.resize(order=0,
 * mode="edge", anti_aliasing=False), which in scikit-image 0.19.3 is scipy.ndimage.zoom(order=0, mode="nearest",
 * grid_mode=True): out[j] = in[floor((j + 0.5) * in_size / out_size)] per axis.  The host computes the three index vectors
 * in float64 exactly like scipy's zoom; this kernel gathers.  seg [BC][D][H][W] -> out [BC][d][h][w].                 */
int e2e_ds_target_gather(const float* seg, float* out, const int* idx_d, const int* idx_h, const int* idx_w,
                         int BC, int D, int H, int W, int d, int h, int w, void* stream);


/* ---- K10: clip_grad_norm_ + SGD(nesterov) + DSFF mask, multi-tensor ---------------------
 * Replaces: torch.nn.utils.clip_grad_norm_(params, 12) + torch.optim.SGD.step (nnUNetTrainer_simple.py:573-574,
 * :369-370) + Masking.apply_mask (core_channel.py:427-434).
 *   table: device array of n entries {param, grad, momentum, mask-or-NULL, numel} */
typedef struct {
  float* param;
  const float* grad;
  float* momentum;
  const float* mask;
  long long numel;
} e2e_param_t;
int e2e_grad_sqnorm(const e2e_param_t* table, int n, double* sq_out /* device, zeroed by callee */, void* stream);
int e2e_sgd_clip_mask_step(const e2e_param_t* table, int n, const double* sq_norm, float max_norm, float lr,
                           float weight_decay, float momentum, int nesterov, int first_step, void* stream);
int e2e_apply_mask(const e2e_param_t* table, int n, void* stream);

/* ---- K9: DSFF kernel statistics ---------------------------------------------------------
 * Replaces: the three chained torch.sum(|w|, dim=-1) of kernel_death (core_channel.py:652-655; the association
 * order is reproduced bit for bit), torch.sort + threshold (:658-661) as an exact k-th order statistic
 * (radix select), and the liveness bit tables consumed by K1/K3/K6.
 *   w [R, Cc, kd,kh,kw] -> l1 [R*Cc] */
int e2e_dsff_kernel_l1(const float* w, float* l1, int R, int Cc, int kd, int kh, int kw, void* stream);
/* kth: value of the k-th smallest (0-based) of n non-negative floats; ws >= 4096 bytes */
int e2e_dsff_kth_value(const float* v, int n, int k, float* out, void* ws, void* stream);
/* kmask [R*Cc] u8: kmask &= !(l1 <= *thr) */
int e2e_dsff_death(const float* l1, const float* thr, unsigned char* kmask, int n, void* stream);
/* growth_mode='gradient' (Masking.kernel_grad_growth, core_channel.py:771-790): score [R*Cc*kd] = sum_kh(sum_kw |grad|) of
 * dead kernels (0 for live ones; two chained sums, the kernel's depth extent stays), grad [R,Cc,kd,kh,kw] scaled by the
 * clip_grad_norm_ coefficient max_norm / (sqrt(*sq_norm) + 1e-6) clamped to 1 (sq_norm NULL: taken as it is);
 * then kmask[i / kd] = 1 where score[i] > *thr (thr = the (num_growth)-th largest score: e2e_dsff_kth_value) */
int e2e_dsff_grad_score(const float* grad, const double* sq_norm, float max_norm, const unsigned char* kmask, float* score,
                        int R, int Cc, int kd, int kh, int kw, void* stream);
int e2e_dsff_grow_above(const float* score, const float* thr, unsigned char* kmask, int R, int Cc, int kd, void* stream);
/* expand a kernel-granular u8 map to the fp32 element mask [R,Cc,ks] and to liveness bit tables:
 * bits_rows [R, ceil(Cc/32)] and bits_cols [Cc, ceil(R/32)] (either may be NULL) */
int e2e_dsff_expand(const unsigned char* kmask, float* mask, unsigned* bits_rows, unsigned* bits_cols, int R,
                    int Cc, int ks, void* stream);
/* liveness "quad words" consumed by e2e_conv133_fwd / e2e_conv133_dgrad (4 output planes x 8 input planes per word,
 * input-plane-major so that the kernel walks the live kernels of one staged input plane back to back):
 *   quads_rows [ceil(R/4), ceil(Cc/8)], bit (c % 8) * 4 + r % 4 = kmask[r][c]   (forward: R = Cout, Cc = Cin)
 *   quads_cols [ceil(Cc/4), ceil(R/8)], bit (r % 8) * 4 + c % 4 = kmask[r][c]   (data gradient)
 * either may be NULL */
int e2e_dsff_expand_quads(const unsigned char* kmask, unsigned* quads_rows, unsigned* quads_cols, int R, int Cc,
                          void* stream);
/* kernel map from weights (inference: live <=> any nonzero tap) */
int e2e_dsff_kmask_from_weights(const float* w, unsigned char* kmask, int R, int Cc, int ks, void* stream);

/* ---- K11: sliding-window aggregation ------------------------------------------------------
 * Replaces: _internal_maybe_mirror_and_pred_3D accumulation (neural_network.py:529-563) and the overlap-add /
 * normalise / argmax of _internal_predict_3D_3Dconv_tiled (neural_network.py:383-407).            */
/* dst[n,c,flip(x)] = src  (torch.flip over the set bits of `axes`: bit0=X(dim2), bit1=Y, bit2=Z)   */
int e2e_flip3d(const float* src, float* dst, int NC, int X, int Y, int Z, int axes, void* stream);
/* result (+)= w * flip(softmax_c(logits));  first != 0 overwrites */
int e2e_softmax_flip_acc(const float* logits, float* result, float w, int first, int K, int X, int Y, int Z,
                         int axes, void* stream);
/* the same with the network's `inference_apply_nonlin` named (neural_network.py:531-560 applies whatever the attribute
 * holds, the constructor default at :80 is the identity): nonlin 0 = identity, 1 = softmax over the classes (the call
 * above), 2 = sigmoid */
int e2e_nonlin_flip_acc(const float* logits, float* result, float w, int first, int K, int X, int Y, int Z,
                        int axes, int nonlin, void* stream);
/* agg[:, x0:x0+px, ...] += patch * gauss (gauss NULL => 1); cnt[x..] += gauss (or 1); patch NULL: the weight map cnt
 * only (a tile whose probabilities another rank adds into ITS partial volume, parallel.run_tiles_partial) */
int e2e_sw_accumulate(const float* patch, const float* gauss, float* agg, float* cnt, int K, int X, int Y, int Z,
                      int px, int py, int pz, int x0, int y0, int z0, void* stream);
/* probs = agg[crop]/cnt[crop]; seg = argmax_k (first max), int64 */
int e2e_sw_finalize_argmax(const float* agg, const float* cnt, float* probs, long long* seg, int K, int X, int Y,
                           int Z, int cx0, int cy0, int cz0, int CX, int CY, int CZ, void* stream);

/* ---- N1: fold ensembling and segmentation export on device --------------------------------------------------
 * Replaces: the per-fold softmax accumulation and average of predict_cases (e2enet/inference/predict.py:282-296; a
 * multi-GB device->host copy per fold in the reference) and transpose_backward + argmax / region thresholds + crop-box
 * placement into the uint8 volume of save_segmentation_nifti_from_softmax (e2enet/inference/segmentation_export.py:
 * 118-136, predict.py:298-301), and the resampling to the original grid (:84-104).  The NIfTI writer stays on the host.
 *   dst (+)= src over n floats (first != 0 overwrites); n_folds > 0 additionally divides by n_folds (float32 division) */
int e2e_ensemble_accumulate(float* dst, const float* src, long long n, int first, int n_folds, void* stream);
/*   probs [K, ...] with class stride kstride; output voxel (a,b,c) of the TRANSPOSED [A,B,C] frame reads source offset
 *   a*sa + b*sb + c*sc; seg is the zero-initialised uint8 volume [OA,OB,OC] of the original (uncropped) size, written at
 *   offset (a0,b0,c0) and clipped to it; regions (device int[n_regions]) != NULL selects the region-threshold rule   */
int e2e_export_argmax_u8(const float* probs, unsigned char* seg, int K, long long kstride, int A, int B, int C,
                         long long sa, long long sb, long long sc, int OA, int OB, int OC, int a0, int b0, int c0,
                         const int* regions, int n_regions, void* stream);
/*   Softmax volume resampled to another grid (order 1; nearest along one optional low-resolution axis) -- replaces
 *   resample_data_or_seg(is_seg=False, order=1, order_z=0) as called by save_segmentation_nifti_from_softmax
 *   (e2enet/inference/segmentation_export.py:84-104, e2enet/preprocessing/preprocessing.py:113-202; skimage resize =
 *   scipy.ndimage.zoom(order=1, mode='nearest', grid_mode=True) on float64, result cast to float32).
 *   src: [K] volumes of the frame [A,B,C] read with strides (sa,sb,sc) (folds transpose_backward), class stride kstride;
 *   dst: contiguous [K,OA,OB,OC]; lowres_axis: -1 = trilinear, 0..2 = nearest along that axis, bilinear in the plane.   */
int e2e_resample_linear(const float* src, float* dst, int K, long long kstride, int A, int B, int C, long long sa,
                        long long sb, long long sc, int OA, int OB, int OC, int lowres_axis, void* stream);


/* ---- K1d: dense 1x3x3 convolution on the bf16 matrix pipe (fp32-exact three-piece operands) --------------------------
 * Same operator as e2e_conv133_fwd / e2e_conv133_dgrad (unetpp_d.py:45-59, :453-478, :93/:108 and their autograd) for layers
 * without DSFF sparsity to exploit (the encoder; masked layers at high density -- `live` / `live_t` are the quad words of
 * e2e_conv133_fwd / e2e_conv133_dgrad, null for a dense layer: pruned kernels are packed as zeros whatever `w` holds):
 * stride (1,1,1), Wi % 32 == 0, Hi % 16 == 0, >= 16 channels on both sides.  e2e_conv133_dense_ws_bytes returns the
 * workspace (packed weights) these calls need, 0 when the shape is not served (use the e2e_conv133_* entry points then).
 * `part` as e2e_conv133_fwd (the partial records are those of its 16 x 32 tiles).                                        */
long long e2e_conv133_dense_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw);
int e2e_conv133_fwd_dense(const e2e_in_chan_t* chans, int Cin, const float* w, const float* bias, const unsigned* live, float* y,
                          double* part, int B, int Cout, int Di, int Hi, int Wi, void* ws, long long ws_bytes, void* stream);
int e2e_conv133_dgrad_dense(const float* dy, const float* w, const unsigned* live_t, const e2e_out_chan_t* outs, int B, int Cin,
                            int Cout, int Di, int Hi, int Wi, void* ws, long long ws_bytes, void* stream);

/* ---- N3: training input feed on the device ------------------------------------------------------------------------
 * Replaces the spatial and intensity transforms of get_moreDA_augmentation
 * (e2enet/training/data_augmentation/data_augmentation_moreDA.py:66-111; batchgenerators 0.24 transforms executed by 24 CPU
 * worker processes in the reference).  All tensors float32 [B, C, D, H, W] contiguous; parameters are drawn on the host.
 *   e2e_aug_spatial: SpatialTransform (:66-79) as one affine gather: source coordinate = A (o - (size_out - 1) / 2) + t per
 *     sample (mat: B x 12 doubles, row-major 3 x 4); data cval 0, order 3 (the reference's order_data, :43) for the samples
 *     flagged in `cubic` (B ints; NULL = all) when `coef` -- the B-spline coefficient image of `data`, three
 *     e2e_aug_bspline_prefilter_axis passes -- is given, order 1 otherwise; seg (may be NULL) order `order_seg` (0: nearest;
 *     1: batchgenerators' per-label linear interpolation thresholded at 0.5), outside the volume cval_seg (order 0) / 0 (order 1)
 *   e2e_aug_bspline_prefilter_axis: scipy.ndimage.spline_filter1d(order 3, mode 'mirror') along `axis` of nvol volumes
 *     [D, H, W], src -> dst (may alias): what scipy's map_coordinates / zoom run in front of an order-3 interpolation
 *   e2e_aug_stats: per (sample, channel) min, max, mean, std (ddof 0) as 4 doubles; ws of e2e_aug_stats_ws_bytes(nbc) bytes
 *   e2e_aug_pointwise: op 1 GaussianNoise (:85), 2 BrightnessMultiplicative (:88), 3 ContrastAugmentation (:96), 4 / 5 the
 *     power and retain_stats steps of GammaTransform (:102-109); prm: nbc x 8 doubles, prm[0] == 0 leaves the channel untouched
 *   e2e_aug_blur_axis: one axis of GaussianBlurTransform (:86-87) = scipy gaussian_filter1d (truncate 4, 'reflect');
 *     wts: nbc x 16 floats (radius, w[0..radius])
 *   e2e_aug_lowres: SimulateLowResolutionTransform (:97-100): nearest down to lo_shape (nbc x 3 ints, 0 = untouched), linear up
 *   e2e_aug_lowres_down / e2e_aug_lowres_up3: the same with the reference's order_upsample = 3 (:99), one volume per call: the
 *     low-resolution volume materialised edge-padded by `pad` (scipy: 12) -> [prefilter, 3 axes] -> cubic up-sampling clipped to
 *     minmax[0..1] (skimage resize clip=True: the low-resolution volume's range, e.g. from e2e_aug_stats)
 *   e2e_aug_finish: MaskTransform (:117-119; use_mask: C ints or NULL) and RemoveLabelTransform(-1, 0) (:121)               */
int e2e_aug_spatial(const float* data, const float* coef, const int* cubic, const float* seg, float* out_data, float* out_seg,
                    const double* mat, int B, int C, int CS, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int order_seg,
                    float cval_seg, void* stream);
int e2e_aug_bspline_prefilter_axis(const float* src, float* dst, int nvol, int D, int H, int W, int axis, void* stream);
long long e2e_aug_stats_ws_bytes(int nbc);
int e2e_aug_stats(const float* x, double* stats, double* ws, int nbc, long long vol, void* stream);
int e2e_aug_pointwise(float* x, const double* prm, int op, int nbc, long long vol, unsigned long long seed, void* stream);
int e2e_aug_blur_axis(const float* src, float* dst, const float* wts, int nbc, int D, int H, int W, int axis, void* stream);
int e2e_aug_lowres(const float* src, float* dst, const int* lo_shape, int nbc, int D, int H, int W, void* stream);
int e2e_aug_lowres_down(const float* src, float* dst, int D, int H, int W, int ld, int lh, int lw, int pad, void* stream);
int e2e_aug_lowres_up3(const float* coef, float* dst, const double* minmax, int D, int H, int W, int ld, int lh, int lw, int pad,
                       void* stream);
int e2e_aug_finish(float* data, float* seg, const int* use_mask, int B, int C, int CS, long long vol, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* E2E_HIP_H */
