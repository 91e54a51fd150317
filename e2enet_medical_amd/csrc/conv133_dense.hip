// K1d: dense 1x3x3 convolution (forward and stride-1 data gradient) on the bf16 matrix pipe with fp32-exact operands (gfx950).
//
// Reference semantics: as conv133.hip (unetpp_d.py:45-59 depth shift, :453-478 concat, :93/:108 Conv3d k(1,3,3), autograd of the
// same for the data gradient).  Serves the layers the DSFF masks do not thin out: the encoder (conv_blocks_context, never
// masked, core_channel.py:324) and masked layers at high density, where the kernel-granular sparse walk of conv133_kernel
// has nothing to skip: dense 32 -> 32 @128^3 runs at 85 TFLOP/s there (54 % of the fp32 vector peak, VALU bound).  A dense
// layer IS a GEMM (M = pixels, N = out channels, K = in channels x 9 taps), so it goes to the matrix cores -- with the operand
// treatment of conv133_wgrad_bf3.hip: every fp32 value is split without error into three bf16 pieces and a product is rebuilt
// from the six leading cross terms (error class of an fp32 FMA; tools/scratch/bf3_numerics.hip).  DSFF-dead kernels are exact
// zeros in the weight tensor (apply_mask, core_channel.py:427-434), so a masked layer computed densely gives the same sums.
//
// Geometry.  Workgroup = 256 threads = one 16 x 32 tile (conv133_kernel's T32 tile, so the InstanceNorm partial records
// line up) of one depth slice x 32 out channels, walked as two 8-row halves.  Per half and 16-channel chunk the halo'd input
// (10 x 34 pixels) is staged into LDS channel-fastest, [piece][pixel][16 ch] bf16 with a 48-byte pixel stride (odd multiple of
// 16: conflict-free ds_read_b128 fragments), converted on the way (normalise-on-load, split); staging loads of the next chunk
// are in flight in registers during the matrix phase.  v_mfma_f32_32x32x16_bf16 with A = 32 pixels of a row x 16 channels
// (LDS), B = 16 channels x 32 out channels of one tap (pre-split, packed weights; the chunk's 27 fragments are copied to LDS
// once per workgroup and chunk), D[pixel][out channel]: a lane owns one out channel, so the InstanceNorm sums are per-lane
// register sums.
// Wave w owns rows 2w, 2w+1 of the half (2 x 16 accumulator registers, 108 matrix instructions per chunk).  Two workgroups per
// CU (2 x 77 KB LDS): one stages while the other multiplies.  Epilogue through LDS (transposed to pixel-fastest rows):
// forward: bias, coalesced stores, (count, mean, M2) per wave and half, combined in fp64 to the tile's partial record;
// data gradient: the scatter epilogue of conv133_kernel (un-shift on store, zero-fill, accumulate).
//
// Staging loads are aligned float4s over columns w0 - 4 .. w0 + 35 (one lane = one (row, 4-column group) x 8 channels); the plane
// pointers and normalise-on-load coefficients come from a per-workgroup LDS table built once (round 3: read through the
// descriptor table with dependent scalar loads in every staging step, each matrix phase started ~4000 cycles late;
// s_memtime stamps of a DENSE_DIAG=4 build: a chunk's matrix phase alone 4300 cycles, the staging commit 3000-3500).
//
// Shapes: stride (1,1,1), Cin <= 128 (the LDS table), W % 32 == 0, H % 16 == 0, H > 16 (the patch sizes of the BASELINE configs at the levels this kernel is
// dispatched for); everything else stays on conv133_kernel.
#include "e2e_common.h"
#include <cstdlib>

#ifndef DENSE_DIAG
#define DENSE_DIAG 0     // diagnostic builds only (make DEFS=-DDENSE_DIAG=n): 1 no matrix phase, 2 no input loads, 3 no split, 4 s_memtime stamps per phase (printf)
#endif

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef const u32x4_t __attribute__((address_space(1)))* gu4_p;
typedef const f32x4_t __attribute__((address_space(1)))* gf4_p;

constexpr int SUBH = 8, TW = 32, XR = SUBH + 2, XC = TW + 2;
constexpr int PXB = 48;                          // bytes per staged pixel and piece: 16 channels bf16 + 16 (odd multiple of 16)
constexpr int SPL = XR * XC * PXB;               // 16 320 B per piece
constexpr int XBYTES = 3 * SPL;                  // 48 960 B
constexpr int XG = 10;                           // 16-byte column groups per halo'd row: columns w0 - 4 .. w0 + 35 (aligned float4 loads)
constexpr int NITEM = XR * XG;                   // 100 (row, group) items per channel half: one lane each, two waves per half
constexpr int WBYTES = 27 * 1024;                // one chunk of packed weights: 9 taps x 3 pieces x 32 out channels x 16 ch bf16
constexpr int WRND = (WBYTES / 16 + 255) / 256;  // 16-byte units per thread and chunk
constexpr int CTAB_MAX = 128;                    // forward: input planes with an entry in the LDS channel table
constexpr int CTAB_ENT = 24;                     // bytes per entry: plane pointer of (n, d - s) 8, scale 4, shift 4, slope 4, valid 4
static_assert(2 * (XBYTES + WBYTES + 8 * 32 * 2 * 4 + CTAB_MAX * CTAB_ENT) <= 163840, "two workgroups per CU");
constexpr int SPITCH = 36;                       // floats per (row, out channel) of the epilogue staging: 32 pixels + 4
static_assert(SUBH * 32 * SPITCH * 4 <= XBYTES, "the epilogue staging (8 rows x 32 channels x 32 pixels fp32) fits the input image");

struct DenseParams {
  const e2e_in_chan_t* chans;     // forward: P input planes
  const float* xin;               // data gradient: dy [B, P, D, H, W]
  const unsigned short* wpk;      // packed weights [qblock][chunk][tap][piece][32 q][16 ch] bf16
  const float* bias;
  float* y;
  double* part;
  const e2e_out_chan_t* outs;     // data gradient: Q destination planes
  int P, Q, B, D, H, W;
  int nchunks, qblocks, tiles_x, tiles_y, tiles_per_n, total, padded_total;
};

template <class T>
__device__ __forceinline__ T load_uniform(const T* ptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const T __attribute__((address_space(4)))*>((unsigned long long)ptr);
#else
  return *ptr;
#endif
}

__device__ __forceinline__ void split1(float v, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, v);
  const float r1 = v - __builtin_bit_cast(float, h & 0xffff0000u);            // exact
  m = __builtin_bit_cast(unsigned, r1);
  const float r2 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);           // exact, <= 8 significant bits
  l = __builtin_bit_cast(unsigned, r2);
}

// ---- weights: fp32 [Q][P][9] (strides wq, wp; reversed taps for the data gradient) -> packed three-piece bf16 -------------------
// `quads` (null = dense layer): the DSFF liveness quad words of this direction, word [q / 4][p / 8], bit (p % 8) * 4 + q % 4
// (e2e_dsff_expand_quads).  A pruned (q, p) kernel is packed as zeros: the matrix-pipe kernel multiplies whole tiles, and its
// result must not depend on pruned weights being exact zeros in `w`.
__global__ __launch_bounds__(256) void pack_weights_bf3_kernel(const float* __restrict__ w, const unsigned* __restrict__ quads,
                                                               unsigned short* __restrict__ wpk, int P, int Q,
                                                               int wq_stride, int wp_stride, int reverse, int nchunks, int qblocks) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = (long long)qblocks * nchunks * 9 * 32 * 16;
  if (idx >= n) return;
  const int k = (int)(idx % 16), ql = (int)((idx / 16) % 32), tap = (int)((idx / 512) % 9);
  const int ch = (int)((idx / (512 * 9)) % nchunks), qb = (int)(idx / ((long long)512 * 9 * nchunks));
  const int q = qb * 32 + ql, pp = ch * 16 + k;
  float v = 0.f;
  if (q < Q && pp < P) {
    const bool alive = quads == nullptr || ((quads[(long long)(q >> 2) * ((P + 7) >> 3) + (pp >> 3)] >> (((pp & 7) << 2) + (q & 3))) & 1u);
    if (alive) v = w[(long long)q * wq_stride + (long long)pp * wp_stride + (reverse ? 8 - tap : tap)];
  }
  unsigned h, m, l;
  split1(v, h, m, l);
  // the two 8-channel halves of rows 16..31 are swapped: with lane -> (row fq = lane & 31, half lane >> 5) fixed by the MFMA
  // operand layout, the 16 lanes of a ds_read_b128 group then fall on 16 distinct bank quads of the 32-byte rows
  const long long base = ((((long long)qb * nchunks + ch) * 9 + tap) * 3) * 512 + ql * 16 + (k ^ ((ql >> 4) << 3));
  wpk[base] = (unsigned short)(h >> 16);
  wpk[base + 512] = (unsigned short)(m >> 16);
  wpk[base + 1024] = (unsigned short)(l >> 16);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv133_dense_kernel(DenseParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[XBYTES];
  __shared__ __attribute__((aligned(16))) unsigned char wlds[WBYTES];      // the chunk's weight fragments [tap][piece][32 q][16 ch]
  __shared__ float red[8][32][2];                          // (mean, M2) of 64 values per (half, wave) and out channel
  // forward: per input plane the resolved plane pointer (batch item, shifted depth) and the normalise-on-load coefficients,
  // built once per workgroup.  (Read through the descriptor table with scalar loads in every staging step, the compiler
  // serialised sixteen dependent scalar round trips in front of each matrix phase: ~4000 of a chunk's ~8000 cycles.)
  __shared__ __attribute__((aligned(8))) unsigned char ctab[MODE == 0 ? CTAB_MAX * CTAB_ENT : 8];

#if DENSE_DIAG == 4
  unsigned long long stamp[24];
  int ns = 0;
#define STAMP() do { if (ns < 24) stamp[ns++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int item = e2e::xcd_remap(blockIdx.x, p.padded_total);
  if (item >= p.total) return;
  const int qb = item % p.qblocks;
  item /= p.qblocks;
  const int n = item / p.tiles_per_n, tile_in_n = item - n * p.tiles_per_n;
  const int tx = tile_in_n % p.tiles_x, ty = (tile_in_n / p.tiles_x) % p.tiles_y, d = tile_in_n / (p.tiles_x * p.tiles_y);
  const int h0 = ty * 16, w0 = tx * TW;
  const long long plane = (long long)p.H * p.W;

  // ---- staging geometry: wave -> (channel half, item range); lane -> one (halo row, 4-column group) item.  Staging loads are
  // aligned float4s (w0 - 4 + 4 g): the vector-memory address path takes a wave instruction at a fixed cost whatever its
  // width (measured: a chunk's 24 dword loads per lane held the issuing wave ~2500 cycles inside the matrix phase), so the
  // same bytes go in a third of the instructions; the three columns loaded beyond either halo edge are dropped. ----
  const int shalf = wave & 1;
  const int sitem = (wave >> 1) * 64 + lane;
  const bool s_act = sitem < NITEM;
  const int s_row = (s_act ? sitem : 0) / XG, s_g = (s_act ? sitem : 0) - s_row * XG;
  const int s_pix0 = s_row * XC + 4 * s_g - 3;           // halo'd pixel index of the group's first column (may be < row start)
  int s_off = -1;                                        // element offset inside a plane, -1 = outside the image / inactive lane
  auto set_geometry = [&](int sub) {
    const int hi = h0 + sub * SUBH - 1 + s_row, wi = w0 - 4 + 4 * s_g;
    const bool ok = s_act && (unsigned)hi < (unsigned)p.H && wi >= 0 && wi < p.W;
    s_off = ok ? hi * p.W + wi : -1;
  };
  f32x4_t xv[8];
  unsigned pvalid = 0;                                    // bit j: channel j of the half in flight is a real plane at a valid depth
  auto prefetch_x = [&](int c) {
    gfloat_p base[8];
    pvalid = 0;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned char* e = ctab + (c * 16 + shalf * 8 + j) * CTAB_ENT;
        base[j] = (gfloat_p)(*reinterpret_cast<const unsigned long long*>(e));
        pvalid |= (*reinterpret_cast<const unsigned*>(e + 20) & 1u) << j;
      }
      } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = c * 16 + shalf * 8 + j;
        const bool v = ch < p.P;
        pvalid |= (v ? 1u : 0u) << j;
        base[j] = (gfloat_p)(p.xin + (((long long)n * p.P + (v ? ch : 0)) * p.D + d) * plane);
      }
    }
    const int off = s_off >= 0 ? s_off : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      xv[j] = DENSE_DIAG == 2 ? f32x4_t{1.f, 1.f, 1.f, 1.f} : *(gf4_p)(unsigned long long)(base[j] + off);
  };
  auto commit_x = [&](int c) {
    if (!s_act) return;
    const bool inside = s_off >= 0;
    float pa[8], pb[8], psl[8];
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned char* e = ctab + (c * 16 + shalf * 8 + j) * CTAB_ENT;
        const float2 ab = *reinterpret_cast<const float2*>(e + 8);
        pa[j] = ab.x; pb[j] = ab.y;
        psl[j] = *reinterpret_cast<const float*>(e + 16);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int hc = 4 * s_g - 3 + k;                     // halo column of this pixel
      if (hc < 0 || hc >= XC) continue;
      unsigned hh[8], mm[8], ll[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = xv[j][k];
        if (MODE == 0) {
          const float u = fmaf(v, pa[j], pb[j]);
          v = fmaxf(u, u * psl[j]);                      // LeakyReLU, 0 <= slope <= 1 (the engine's contract); slope 1 = identity
        }
        v = (((pvalid >> j) & 1u) && inside) ? v : 0.f;
        if (DENSE_DIAG == 3) { hh[j] = mm[j] = ll[j] = __builtin_bit_cast(unsigned, v); }
        else split1(v, hh[j], mm[j], ll[j]);
      }
      unsigned char* dst = lds + (s_pix0 + k) * PXB + shalf * 16;
      // v_perm_b32 0x07060302: (S1 >> 16) | (S0 & 0xffff0000): two bf16 pieces per word, element 0 in the low half
      *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{__builtin_amdgcn_perm(hh[1], hh[0], 0x07060302u), __builtin_amdgcn_perm(hh[3], hh[2], 0x07060302u),
                                                 __builtin_amdgcn_perm(hh[5], hh[4], 0x07060302u), __builtin_amdgcn_perm(hh[7], hh[6], 0x07060302u)};
      *reinterpret_cast<u32x4_t*>(dst + SPL) = u32x4_t{__builtin_amdgcn_perm(mm[1], mm[0], 0x07060302u), __builtin_amdgcn_perm(mm[3], mm[2], 0x07060302u),
                                                       __builtin_amdgcn_perm(mm[5], mm[4], 0x07060302u), __builtin_amdgcn_perm(mm[7], mm[6], 0x07060302u)};
      *reinterpret_cast<u32x4_t*>(dst + 2 * SPL) = u32x4_t{__builtin_amdgcn_perm(ll[1], ll[0], 0x07060302u), __builtin_amdgcn_perm(ll[3], ll[2], 0x07060302u),
                                                           __builtin_amdgcn_perm(ll[5], ll[4], 0x07060302u), __builtin_amdgcn_perm(ll[7], ll[6], 0x07060302u)};
    }
  };

  const int fq = lane & 31, fh8 = lane >> 5;
  // weight fragment (B operand): out channel fq, channels 8 fh8 .. + 7 of the chunk, from the LDS copy of the chunk's packed
  // weights (all four waves use the same fragments; straight from global each wave waited for L2 at every tap)
  const unsigned char* const wfbase = wlds + fq * 32 + ((fh8 ^ (fq >> 4)) << 4);
  auto wfrag = [&](int tap, int s) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(wfbase + (tap * 3 + s) * 1024); };
  u32x4_t vw[WRND];
  auto prefetch_w = [&](int c) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wpk) + ((long long)qb * p.nchunks + c) * WBYTES;
#pragma unroll
    for (int i = 0; i < WRND; ++i) {
      const int off = (i * 256 + tid) * 16;
      vw[i] = *(gu4_p)(unsigned long long)(src + (off < WBYTES ? off : 0));
    }
  };
  auto commit_w = [&]() {
#pragma unroll
    for (int i = 0; i < WRND; ++i) {
      const int off = (i * 256 + tid) * 16;
      if (off < WBYTES) *reinterpret_cast<u32x4_t*>(wlds + off) = vw[i];
    }
  };
  const float bq = (MODE == 0 && p.bias != nullptr && qb * 32 + fq < p.Q) ? p.bias[qb * 32 + fq] : 0.f;

  STAMP();
  set_geometry(0);
  prefetch_w(0);
  if (MODE == 0) {
    for (int ch = tid; ch < p.nchunks * 16; ch += 256) {
      const e2e_in_chan_t cd = p.chans[ch < p.P ? ch : 0];
      const int din = d - cd.dshift;
      const bool valid = ch < p.P && (unsigned)din < (unsigned)p.D;
      float a = 1.f, b = 0.f, sl = 1.f;
      if (cd.scale != nullptr) {
        a = cd.scale[(long long)n * cd.ab_nstride];
        b = cd.shift[(long long)n * cd.ab_nstride];
        sl = cd.slope;
      }
      unsigned char* e = ctab + ch * CTAB_ENT;
      *reinterpret_cast<unsigned long long*>(e) = (unsigned long long)(cd.ptr + (long long)n * cd.nstride + (long long)(valid ? din : 0) * plane);
      *reinterpret_cast<float2*>(e + 8) = float2{a, b};
      *reinterpret_cast<float*>(e + 16) = sl;
      *reinterpret_cast<unsigned*>(e + 20) = valid ? 1u : 0u;
    }
    __syncthreads();
  }
  prefetch_x(0);
  for (int sub = 0; sub < 2; ++sub) {
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    for (int c = 0; c < p.nchunks; ++c) {
      STAMP();
      commit_w();
      commit_x(c);
      STAMP();
      __syncthreads();
      STAMP();
      bf16x8 wb[2][3];                                    // ring over taps: [tap & 1][piece]
#pragma unroll
      for (int s = 0; s < 3; ++s) wb[0][s] = wfrag(0, s);
      // A fragment: pixel (row 2 wave + r + kh, column fq + kw), channels 8 fh8 .. + 7.  The 18 (tap, row) steps are software
      // pipelined: the three pieces of step i + 1 are read from LDS before the six matrix instructions of step i are issued
      // (two waves per SIMD do not hide a ds_read -> mfma dependency by themselves).
      const unsigned char* abase = lds + ((2 * wave) * XC + fq) * PXB + fh8 * 16;
      bf16x8 af[2][3];
#pragma unroll
      for (int s = 0; s < 3; ++s) af[0][s] = *reinterpret_cast<const bf16x8*>(abase + s * SPL);
#pragma unroll
      for (int step = 0; step < (DENSE_DIAG == 1 ? 1 : 18); ++step) {
        const int tap = step >> 1, r = step & 1;
        if (r == 0 && tap + 1 < 9) {
#pragma unroll
          for (int s = 0; s < 3; ++s) wb[(tap + 1) & 1][s] = wfrag(tap + 1, s);
        }
        if (step == 0) {
          // staging loads of the next (half, chunk): in flight during this whole matrix phase
          if (c + 1 < p.nchunks) { prefetch_w(c + 1); prefetch_x(c + 1); }
          else if (sub == 0) { set_geometry(1); prefetch_w(0); prefetch_x(0); }
        }
        if (step + 1 < 18) {
          const int nt = (step + 1) >> 1, nr = (step + 1) & 1;
#pragma unroll
          for (int s = 0; s < 3; ++s)
            af[(step + 1) & 1][s] = *reinterpret_cast<const bf16x8*>(abase + ((nr + nt / 3) * XC + nt % 3) * PXB + s * SPL);
        }
        f32x16 a = acc[r];
        // small terms first: lo*hi, mid*mid, hi*lo, then mid*hi, hi*mid, then hi*hi
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step & 1][2], wb[tap & 1][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step & 1][1], wb[tap & 1][1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step & 1][0], wb[tap & 1][2], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step & 1][1], wb[tap & 1][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step & 1][0], wb[tap & 1][1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step & 1][0], wb[tap & 1][0], a, 0, 0, 0);
        acc[r] = a;
      }
      STAMP();
      __syncthreads();
      STAMP();
    }

    // ---- epilogue of this half.  D layout of v_mfma_f32_32x32x16: column (out channel) = lane & 31, row (pixel of the tile
    // row) = (i & 3) + 8 (i >> 2) + 4 (lane >> 5).  LDS staging: [tile row 8][out channel 32][pixel 32 + 4 pad] fp32 (the pad
    // spreads the 32 out channels of a ds_write_b128 over the banks: unpadded, all lanes of a write hit one bank group). ----
    float* const stg = reinterpret_cast<float*>(lds);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[r][i] += bq;
        psum += acc[r][i];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4_t*>(stg + ((2 * wave + r) * 32 + fq) * SPITCH + 4 * fh8 + 8 * g) =
            f32x4_t{acc[r][4 * g], acc[r][4 * g + 1], acc[r][4 * g + 2], acc[r][4 * g + 3]};
    }
    if (MODE == 0 && p.part != nullptr) {
      // (count 64, mean, M2) of this wave's two rows per out channel: the two lane halves hold 32 values each
      const float mean = (psum + __shfl_xor(psum, 32, 64)) * (1.f / 64.f);
      float m2 = 0.f;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float dl = acc[r][i] - mean;
          m2 = fmaf(dl, dl, m2);
        }
      m2 += __shfl_xor(m2, 32, 64);
      if (fh8 == 0) { red[sub * 4 + wave][fq][0] = mean; red[sub * 4 + wave][fq][1] = m2; }
    }
    // read this wave's own two rows back pixel-fastest (only LDS ordering inside the wave is needed) and store / scatter them
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int seg = 8 * k + (lane >> 3), px4 = (lane & 7) * 4;
      const int rl = seg >> 5, ql = seg & 31;
      const int q = qb * 32 + ql;
      const int oh = h0 + sub * SUBH + 2 * wave + rl;
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(stg + ((2 * wave + rl) * 32 + ql) * SPITCH + px4);
      if (q >= p.Q) continue;
      const long long po = (long long)oh * p.W + w0 + px4;
      if (MODE == 0) {
        *reinterpret_cast<f32x4_t*>(p.y + (((long long)n * p.Q + q) * p.D + d) * plane + po) = v;
      } else {
        // data gradient: the gradient of virtual-concat channel q at (shifted) depth d goes to depth d - s(q) of its source;
        // slices that receive nothing are zero-filled by the workgroups of the out-of-range depths (conv133_kernel's rule)
        const e2e_out_chan_t oc = p.outs[q];
        if (oc.ptr == nullptr) continue;
        int dd = d - oc.dshift;
        bool zero_fill = false;
        if (dd < 0) {
          const int lo = p.D - oc.dshift > 0 ? p.D - oc.dshift : 0;
          dd = lo + d;
          zero_fill = true;
        } else if (dd >= p.D) {
          const int lo = p.D + oc.dshift > 0 ? p.D + oc.dshift : 0;
          dd = d - lo;
          zero_fill = true;
        }
        if (zero_fill && oc.accumulate) continue;
        f32x4_t* dst = reinterpret_cast<f32x4_t*>(oc.ptr + (long long)n * oc.nstride + (long long)dd * plane + po);
        f32x4_t o = zero_fill ? f32x4_t{0.f, 0.f, 0.f, 0.f} : v;
        if (!zero_fill && oc.accumulate) { const f32x4_t old = *dst; o += old; }
        *dst = o;
      }
    }
    STAMP();
    __syncthreads();                                       // the staging area is the next half's input image
  }
#if DENSE_DIAG == 4
  if (MODE == 0 && tid == 64 && (blockIdx.x == 4001 || blockIdx.x == 4002 || blockIdx.x == 7000)) {
#define DD(i) (unsigned)(stamp[i] - stamp[i - 1])
    printf("WG %d | pre %u | c0: commit %u bar %u mma %u bar %u | c1: %u %u %u %u | epi %u | c0: %u %u %u %u %u | c1: %u %u %u %u | epi %u\n", blockIdx.x,
           DD(1), DD(2), DD(3), DD(4), DD(5), DD(7), DD(8), DD(9), DD(10), DD(11), DD(12), DD(13), DD(14), DD(15), DD(16), DD(18), DD(19), DD(20), DD(21), DD(22));
  }
#endif

  if (MODE == 0 && p.part != nullptr && tid < 32 && qb * 32 + tid < p.Q) {
    // Chan combination of the eight (count 64, mean, M2) records of this tile, fp64
    double cn = 0.0, cm = 0.0, c2 = 0.0;
    for (int k = 0; k < 8; ++k) {
      const double bn = 64.0, bm = (double)red[k][tid][0], b2 = (double)red[k][tid][1];
      const double tot = cn + bn, dl = bm - cm;
      cm += dl * (bn / tot);
      c2 += b2 + dl * dl * (cn * bn / tot);
      cn = tot;
    }
    double* pp = p.part + (((long long)n * p.Q + qb * 32 + tid) * p.tiles_per_n + tile_in_n) * 3;
    pp[0] = cn; pp[1] = cm; pp[2] = c2;
  }
}

inline bool dense_knob() {
  static const int v = getenv("E2E_CONV_DENSE") ? atoi(getenv("E2E_CONV_DENSE")) : 1;
  return v != 0;
}

}  // namespace

// shapes the dense kernel serves; 0 bytes = not eligible
extern "C" long long e2e_conv133_dense_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw) {
  (void)B;
  if (!dense_knob()) return 0;
  if (sd != 1 || sh != 1 || sw != 1) return 0;
  if (Wi % 32 != 0 || Hi % 16 != 0 || Hi <= 16 || Di < 1) return 0;   // (planes of conv133_kernel's 16 x 32 tile class: the partial records line up)
  if (Cin < 16 || Cout < 16) return 0;                    // thin layers (the 4-modal input): padding would dominate
  if (Cin > CTAB_MAX) return 0;                           // the forward kernel's LDS channel table
  const long long cmax = Cin > Cout ? Cin : Cout;
  const long long blocks = e2e::cdiv((int)cmax, 32) * (long long)e2e::cdiv((int)cmax, 16);
  return blocks * 9 * 3 * 512 * 2;                        // packed weights of either direction
}

static int dense_launch(int mode, const e2e_in_chan_t* chans, const float* xin, const float* w, const unsigned* quads, const float* bias, float* y, double* part,
                        const e2e_out_chan_t* outs, int B, int P, int Q, int D, int H, int W, int wq_stride, int wp_stride,
                        void* ws, long long ws_bytes, hipStream_t st) {
  DenseParams p{};
  p.chans = chans; p.xin = xin; p.bias = bias; p.y = y; p.part = part; p.outs = outs;
  p.P = P; p.Q = Q; p.B = B; p.D = D; p.H = H; p.W = W;
  p.nchunks = e2e::cdiv(P, 16);
  p.qblocks = e2e::cdiv(Q, 32);
  const long long need = (long long)p.qblocks * p.nchunks * 9 * 3 * 512 * 2;
  E2E_REQUIRE(ws != nullptr && ws_bytes >= need, "conv133 dense: workspace too small (%lld < %lld bytes)", ws_bytes, need);
  p.wpk = reinterpret_cast<const unsigned short*>(ws);
  p.tiles_x = W / TW; p.tiles_y = H / 16;
  p.tiles_per_n = D * p.tiles_y * p.tiles_x;
  p.total = B * p.tiles_per_n * p.qblocks;
  p.padded_total = (p.total + 7) & ~7;
  const long long nel = (long long)p.qblocks * p.nchunks * 9 * 512;
  hipLaunchKernelGGL(pack_weights_bf3_kernel, dim3((unsigned)e2e::cdivll(nel, 256)), dim3(256), 0, st, w, quads, reinterpret_cast<unsigned short*>(ws),
                     P, Q, wq_stride, wp_stride, mode == 1 ? 1 : 0, p.nchunks, p.qblocks);
  e2e::note_kernel("conv133_dense_bf3<mode=%d> wgs=%d chunks=%d", mode, p.padded_total, p.nchunks);
  if (mode == 0) hipLaunchKernelGGL((conv133_dense_kernel<0>), dim3(p.padded_total), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((conv133_dense_kernel<1>), dim3(p.padded_total), dim3(256), 0, st, p);
  return e2e::check_launch("conv133_dense_kernel");
}

extern "C" int e2e_conv133_fwd_dense(const e2e_in_chan_t* chans, int Cin, const float* w, const float* bias, const unsigned* live, float* y, double* part,
                                     int B, int Cout, int Di, int Hi, int Wi, void* ws, long long ws_bytes, void* stream) {
  E2E_REQUIRE(chans && w && y, "conv133_fwd_dense: null pointer");
  E2E_REQUIRE(e2e_conv133_dense_ws_bytes(B, Cin, Cout, Di, Hi, Wi, 1, 1, 1) > 0, "conv133_fwd_dense: shape not served (stride 1, W %% 32 == 0, H %% 16 == 0, >= 16 channels)");
  return dense_launch(0, chans, nullptr, w, live, bias, y, part, nullptr, B, Cin, Cout, Di, Hi, Wi, Cin * 9, 9, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int e2e_conv133_dgrad_dense(const float* dy, const float* w, const unsigned* live_t, const e2e_out_chan_t* outs, int B, int Cin, int Cout, int Di,
                                       int Hi, int Wi, void* ws, long long ws_bytes, void* stream) {
  E2E_REQUIRE(dy && w && outs, "conv133_dgrad_dense: null pointer");
  E2E_REQUIRE(e2e_conv133_dense_ws_bytes(B, Cin, Cout, Di, Hi, Wi, 1, 1, 1) > 0, "conv133_dgrad_dense: shape not served");
  // the forward kernel with transposed, tap-reversed weights: its "input planes" are dy's Cout channels, its output planes the
  // Cin virtual-concat channels (weight element [q = c][p = o][tap] = w[o][c][8 - tap])
  return dense_launch(1, nullptr, dy, w, live_t, nullptr, nullptr, nullptr, outs, B, Cout, Cin, Di, Hi, Wi, 9, Cin * 9, ws, ws_bytes, (hipStream_t)stream);
}
