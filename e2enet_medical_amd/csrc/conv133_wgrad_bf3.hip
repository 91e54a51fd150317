// K6b', dense weight gradient of the 1x3x3 convolution on the bf16 matrix pipe with fp32-exact operands (gfx950).
//
//   dW[o, c, kh, kw] = sum_{n, d, h, w} dy[n, o, d, h, w] * xs[n, c, d*sd, h + kh - 1, w + kw - 1]          (stride-1 planes)
//   xs = depth-shifted virtual concat of the producers' lrelu(IN(.)) outputs (unetpp_d.py:45-59, :453-478); autograd of
//   unetpp_d.py:93/:108 (nnUNetTrainer_simple.py:572).
//
// Why: the fp32-input MFMA runs at the fp32 vector rate (64 FLOP/clk/SIMD); v_mfma_f32_16x16x32_bf16 runs at 16x that.
// Every fp32 operand is split WITHOUT error into three bf16 pieces, a = a_hi + a_mid + a_lo (8 + 8 + 8 significant bits,
// truncation splits, each remainder exact), once, while the tile is staged into LDS; a product a*b is rebuilt from the six
// leading cross terms hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi (each exact in the fp32 accumulator's product), the
// dropped terms are <= 2^-23 |a b|: the same class as the one rounding an fp32 FMA makes per term
// (tools/scratch/bf3_numerics.hip measures both against fp64).  6 matrix instructions per 16 fp32 k-steps instead of 16.
//
// Geometry (all variants).  One workgroup = 32 out x 32 in channels x 9 taps over a run of 4 x 32-pixel tiles of one batch item,
// two LDS images of 77 KB ([piece][channel][rows] bf16, channel stride 400 B = 25 x 16: the 16 lanes of a ds_read_b128 group hit
// 16 distinct bank quads), one barrier per tile.  GEMM per tile and tap: M = out channel, N = in channel, K = the 32 pixels of a
// tile row as two 16-pixel K blocks (v_mfma_f32_32x32x16_bf16).
//   * B fragments (input): x[c][row r + kh][16 px]: 8 consecutive bf16 per lane = one aligned ds_read_b128; no column halo:
//   * the tap's column shift is put on the dy side: dW[.,kw] = sum_w' dy[w' - kw + 1] x[w'].  The A fragment of kw = 1 is an
//     aligned ds_read_b128 of dy; kw = 0 / 2 are the same 8 elements moved by one bf16, built in registers with
//     v_alignbit_b32 from the aligned read plus the two neighbouring dwords (no shifted copies in LDS);
//   * the registers of tile t+1 are converted and committed while the matrix instructions of tile t run, the loads of tile
//     t+2 (v5: t+3 for the input) are in flight; straight-line code around every load (exact vmcnt distances);
//   * the row waves' accumulators are added through LDS in a fixed tree at the end; one slab per workgroup, reduced over the
//     chunks by wgrad_slab_reduce_kernel (deterministic).
// History: v1 (16x16x32 MFMA, eight waves), v2 (twelve waves that stage and multiply in lockstep) and v4 (four matrix waves + four
// staging waves) were measured against this form in rounds 3 / 4 (profiles/r04_kbench.txt, r04_pmc_sq_bf3.txt: 0.905 / 0.865 /
// 0.83 ms on 64 -> 32 @128^3 x 2) and removed in round 5; what is left is v5 in two operand formats (NPC below).
#include "e2e_common.h"
#include <cstdlib>

// raw buffer loads (the clang builtin __builtin_amdgcn_raw_buffer_load_b128 of this toolchain lowers to a splatted dword load)
__device__ float __attribute__((ext_vector_type(4))) llvm_raw_buffer_load_v4f32(int __attribute__((ext_vector_type(4))) rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float llvm_raw_buffer_load_f32(int __attribute__((ext_vector_type(4))) rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef const f32x4_t __attribute__((address_space(1)))* gf4_p;
typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int TH = 4, TW = 32;
constexpr int XROWS = TH + 2;
constexpr int YROWB = 96;                       // dy row: 48 bf16 (tile columns at 8..39, halo at 7 and 40)
constexpr int XROWB = 64;

// fp32 -> (hi, mid, lo) bf16 pieces of four values, packed pairwise (element 0 in the low half of word 0)
__device__ __forceinline__ void split4(const float (&v)[4], u32x2_t& hi, u32x2_t& mid, u32x2_t& lo) {
  unsigned u[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u[j] = __builtin_bit_cast(unsigned, v[j]);
    const float r1 = v[j] - __builtin_bit_cast(float, u[j] & 0xffff0000u);          // exact
    m[j] = __builtin_bit_cast(unsigned, r1);
    const float r2 = r1 - __builtin_bit_cast(float, m[j] & 0xffff0000u);            // exact, <= 8 significant bits
    l[j] = __builtin_bit_cast(unsigned, r2);
  }
  // v_perm_b32: bytes {S0, S1}; 0x07060302 = (S1 >> 16) | (S0 & 0xffff0000)
  hi = u32x2_t{__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u)};
  mid = u32x2_t{__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
  lo = u32x2_t{__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
}

// fp16 two-piece split (round 5): v * 2^k = hi + lo + r with hi = rn16(v), lo = rn16(v - hi) (the subtraction is exact), 11 + 11
// significant bits: |r| <= 2^-23 |v| while lo is a normal fp16 (|v| >= 2^-3 in scaled units), half a subnormal step (2^-25) below.
// v_cvt_pk_f16_f32, two v_cvt_f32_f16 (one SDWA), v_pk_add_f32, v_cvt_pk_f16_f32: 5 instructions per two values (bf16x3: 11).
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4h(const float (&v)[4], u32x2_t& hi, u32x2_t& lo) {
  const f16x2_t h0 = __builtin_convertvector((f32x2_t{v[0], v[1]}), f16x2_t), h1 = __builtin_convertvector((f32x2_t{v[2], v[3]}), f16x2_t);
  const f16x2_t l0 = __builtin_convertvector((f32x2_t{v[0] - (float)h0[0], v[1] - (float)h0[1]}), f16x2_t);
  const f16x2_t l1 = __builtin_convertvector((f32x2_t{v[2] - (float)h1[0], v[3] - (float)h1[1]}), f16x2_t);
  hi = u32x2_t{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
  lo = u32x2_t{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
}

// Channel stride 400 B (= 25 x 16, odd): the 16 lanes of a ds_read_b128 group (one 8-element k group, 16 channels) hit 16
// distinct bank quads.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CSTR2 = 400, SSTR2 = 32 * CSTR2;              // channel stride, piece stride (32 channels)
static_assert(XROWS * XROWB + 16 == CSTR2 && TH * YROWB + 16 == CSTR2 && (CSTR2 / 16) % 2 == 1, "channel stride");

// ---- v5: one wave per SIMD, the staging stream dealt into the matrix-instruction gaps ------------------------------------------
// v4's two streams are arbitrated by the hardware and both end up issue-bound on their shared SIMD (matrix pipe 66 % busy).  Here
// the workgroup is four waves (one per SIMD, 512 registers): wave r owns tile row r with all nine taps AND a quarter of the
// staging.  A tile is six phases of 18 matrix instructions; each phase carries a fixed share of the other work, spread over its
// gaps by sched_group_barrier (one matrix instruction, then up to five others):
//   phase 0: request the input pieces of tile t+2 (registers of the set that held tile t)
//   phases 1-4: convert and commit tile t+1 (requested during tile t-1) into the image not being read
//   phase 5: barrier first (all commits done, every read of this image landed), then the dy requests of tile t+2 and the first
//            fragment reads of tile t+1
// plus, in every phase, the fragment reads of the next phase.  Same images, split, order per accumulator and reduction tree as
// v2 / v4 (bit-identical results).
#define WG5_GAP1() __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x0b6, WG5_FILL, 0);
#define WG5_GAP6() WG5_GAP1() WG5_GAP1() WG5_GAP1() WG5_GAP1() WG5_GAP1() WG5_GAP1()
#ifndef WG5_FILL
#define WG5_FILL 5
#endif
#ifndef WG5H_FILL
#define WG5H_FILL 10               // fp16x2 form: half the matrix instructions, the same stream to deal between them
#endif
#define WG5H_GAP1() __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x0b6, WG5H_FILL, 0);
#define WG5H_GAP9() WG5H_GAP1() WG5H_GAP1() WG5H_GAP1() WG5H_GAP1() WG5H_GAP1() WG5H_GAP1() WG5H_GAP1() WG5H_GAP1() WG5H_GAP1()
#ifndef WG5_DIAG
#define WG5_DIAG 0                 // timing-only builds: 1 no conversion / commit, 2 no global requests, 4 no fragment reads
#endif
// G = 0: 4 x 32-pixel tiles (planes at least 32 wide; the two K blocks of a wave are the halves of its tile row).
// G = 1: 8 x 16-pixel tiles for planes 16..31 wide (verdict item 2: these ran on fp32 MFMA): the two K blocks of wave r are the tile
//        rows 2r and 2r+1; dy rows of 48 B (pixels at entries 8..23, halos at 7 and 24 -- entry 24 shares the unused head of the
//        next row), input rows of 32 B, the same 400 B channel stride: same images, same pipeline, 5 + 4 + 2 pieces per wave.
// NPC = 3: bf16 three-piece operands, six products per fp32 product.  NPC = 2 (round 5): fp16 two-piece operands, THREE products
//        (lo*hi, hi*lo, hi*hi through v_mfma_f32_32x32x16_f16): 54 matrix instructions per tile instead of 108, two thirds of the
//        LDS image, 5 instead of 11 conversion instructions per value pair.  fp16 has 5 exponent bits: dy (magnitudes of 1e-7 at
//        full resolution) is pre-scaled by the exact power of two 2^k that puts max |dy| in [2^14, 2^15), max |dy| being what the
//        producer of dy recorded (p.dy_absmax, e2e_in_lrelu_bwd); the slab is un-scaled on store.  The input side (activations
//        after InstanceNorm + LeakyReLU, transposed-conv outputs) is pre-scaled by the power of two that puts a BOUND of |x| over
//        the conv's input planes (p.x_absmax: e2e_conv133_input_ranges, from the parameters) in [2^14, 2^15) -- round 6; rounds 5's
//        fixed 2^XSH = 8 (still what a NULL word means) turned |x| > 8188 into Inf -- folded into the lane's (scale, shift) pair;
//        the lo piece of values 2^11 below the bound's scale is a subnormal fp16 (absolute error 2^-25 of the scaled range there
//        instead of relative 2^-23: the negative half of every LeakyReLU output lives around 0.01).  Error against fp64 (same probe, K = 256 .. 262144, activation x heavy-tailed
//        1e-7 gradients, all-positive operands): at or below the bf16x3 form and the fp32 FMA chain in every row; round-to-nearest
//        splits (truncating ones are biased: 4e-4 of the result at K = 262144 with one-signed operands).
constexpr int XSH = 3;
__device__ unsigned long long g_wg_clock[2];     // as g_mm_clock (conv133_mm.hip)
template <int G, int NPC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv133_wgrad_bf3v5_kernel(e2e::WgBf3Params p) {
  constexpr int XBn = NPC * SSTR2, BUFn = 2 * XBn;          // input image, one (input + dy) image
  static_assert(2 * 144 * 64 * 4 <= 2 * BUFn, "the reduction regions fit the images");
  constexpr int GTH = G ? 8 : 4, GTW = G ? 16 : 32;         // tile
  constexpr int NX = G ? 5 : 6, NH = G ? 2 : 1;             // input / halo pieces per wave (dy: 4)
  constexpr int XIT = G ? 40 : 48, XQS = G ? 2 : 3;         // items (rows x quads) per input channel, log2 quads per row
  constexpr int GXROWB = G ? 32 : 64, GYROWB = G ? 48 : 96; // bytes per staged input / dy row
  static_assert((GTH + 2) * GXROWB + 16 <= CSTR2 && GTH * GYROWB + 16 <= CSTR2, "channel stride");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFn];
  const int segs = p.segs;
  const int n = blockIdx.x / segs, seg = blockIdx.x - n * segs;
  const int cg = blockIdx.y % p.cblocks, ob = blockIdx.y / p.cblocks;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_lo = seg * p.tiles_per_chunk;
  int tile_hi = tile_lo + p.tiles_per_chunk;
  if (tile_hi > p.tiles_per_n) tile_hi = p.tiles_per_n;
  const int ntiles = tile_hi - tile_lo;
  const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // fp16x2: dy scale 2^k from the recorded max |dy| (biased exponent E: k = 141 - E puts the max in [2^14, 2^15)); k clamped so
  // that 2^k and 2^-(k + XSH) are normal numbers; a zero / denormal max takes E = 1, Inf / NaN propagate
  // The input side: 2^kx from the bound of |x| the caller derived for this conv's input planes (p.x_absmax, round 6; without it the
  // fixed 2^XSH).  The slab is un-scaled by the two exact factors 2^-k and 2^-kx (the sum of the exponents may leave the fp32 range).
  float ysc = 1.f, unsc = 1.f, unsc_x = 1.f, xsc = 1.f;
  if constexpr (NPC == 2) {
    int k = 0, kx = XSH;
    if (p.dy_absmax != nullptr) {
      int E = (int)((__builtin_nontemporal_load(p.dy_absmax) >> 23) & 0xffu);
      E = E < 1 ? 1 : E;
      k = 141 - E;
      k = k > 120 ? 120 : (k < -120 ? -120 : k);
    }
    if (p.x_absmax != nullptr) {
      int E = (int)((__builtin_nontemporal_load(p.x_absmax) >> 23) & 0xffu);
      E = E < 1 ? 1 : E;
      kx = 141 - E;
      kx = kx > 110 ? 110 : (kx < -110 ? -110 : kx);
    }
    ysc = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
    unsc = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
    xsc = __builtin_bit_cast(float, (unsigned)(127 + kx) << 23);
    unsc_x = __builtin_bit_cast(float, (unsigned)(127 - kx) << 23);
  }

  if (ntiles > 0) {
    const int sw = wave, wr = wave;
    const int obase = ob * 32, cbase = cg * 32;
    const int in_plane = p.Hi * p.Wi;                         // (host: Di * Hi * Wi < 2^29, Cout * Do * Hi * Wi < 2^29)
    // ---- staging share of this wave (as bf3v4_stage) ----
    gfloat_p xbase[NX];
    float xa[NX], xb[NX], xsl[NX];
    int xdsh[NX], xro[NX], xgc[NX], xoff[NX], xdst[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int item = j * 64 + lane;
      const int chl = item / XIT, rem = item - chl * XIT;
      const int x_r = rem >> XQS, x_q = rem & ((1 << XQS) - 1);
      const int c = cbase + sw * 8 + chl;
      const bool val = c < p.Cin;
      const e2e_in_chan_t* chd = p.chans + (val ? c : 0);
      xdsh[j] = val ? chd->dshift : (1 << 30);
      xbase[j] = (gfloat_p)(chd->ptr + (long long)n * chd->nstride);
      xa[j] = 1.f; xb[j] = 0.f; xsl[j] = 1.f;
      if (val && chd->scale != nullptr) {
        xa[j] = chd->scale[(long long)n * chd->ab_nstride];
        xb[j] = chd->shift[(long long)n * chd->ab_nstride];
        xsl[j] = chd->slope;
      }
      if constexpr (NPC == 2) { xa[j] *= xsc; xb[j] *= xsc; }   // LeakyReLU commutes with a positive scale
      xro[j] = x_r - 1;
      xgc[j] = 4 * x_q;
      xoff[j] = (x_r - 1) * p.Wi + 4 * x_q - (val ? chd->dshift : 0) * in_plane;
      xdst[j] = (sw * 8 + chl) * CSTR2 + x_r * GXROWB + x_q * 8;
    }
    const int y_grp = lane & 31, y_r = y_grp >> XQS, y_q = y_grp & ((1 << XQS) - 1);
    const int h_side = lane & 1, h_r = (lane >> 1) & (GTH - 1);   // halo round k: channel (lane / (2 GTH)) + (8 / NH) k
    const unsigned long long dya = (unsigned long long)(p.dy + (long long)n * p.Cout * p.Do * in_plane);
    const i32x4_t dyr = {__builtin_amdgcn_readfirstlane((int)dya), __builtin_amdgcn_readfirstlane((int)(dya >> 32) & 0xffff),
                         __builtin_amdgcn_readfirstlane(p.Cout * p.Do * in_plane * 4), 0x00020000};
    const i32x4_t dyh = {__builtin_amdgcn_readfirstlane((int)(dya - 4)), __builtin_amdgcn_readfirstlane((int)((dya - 4) >> 32) & 0xffff),
                         __builtin_amdgcn_readfirstlane(p.Cout * p.Do * in_plane * 4 + 4), 0x00020000};
    int yoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = obase + sw * 8 + 2 * j + (lane >> 5);
      yoff[j] = o < p.Cout ? ((o * p.Do) * in_plane + y_r * p.Wi + 4 * y_q) * 4 : (int)0x80000000;
    }
    int hoff[NH], hdst[NH];
#pragma unroll
    for (int k2 = 0; k2 < NH; ++k2) {
      const int h_ch = lane / (2 * GTH) + (8 / NH) * k2;
      const int ho_ = obase + sw * 8 + h_ch;
      hoff[k2] = ho_ < p.Cout ? ((ho_ * p.Do) * in_plane + h_r * p.Wi + (h_side ? GTW + 1 : 0)) * 4 : (int)0x80000000;
      hdst[k2] = (sw * 8 + h_ch) * CSTR2 + h_r * GYROWB + (h_side ? 8 + GTW : 7) * 2;
    }
    const int ydst = y_r * GYROWB + (8 + 4 * y_q) * 2;

    struct Pos { int tx, ty, d; };
    auto advance = [&](Pos& q, bool go) {
      const int tx = q.tx + 1;
      const bool wx = tx == p.tiles_x;
      const int ty = q.ty + (wx ? 1 : 0);
      const bool wy = ty == p.tiles_y;
      q.tx = go ? (wx ? 0 : tx) : q.tx;
      q.ty = go ? (wy ? 0 : ty) : q.ty;
      q.d = go ? q.d + (wy ? 1 : 0) : q.d;
    };
    f32x4_t vx[2][NX], vy[2][4];
    float vh[2][NH];
    float ta[2][NX], tb[2][NX];
    auto request_x = [&](const int rs, const Pos& q) {
      const int h0 = q.ty * GTH, w0 = q.tx * GTW;
      const int dd = q.d * p.sd;
      const int S = dd * in_plane + h0 * p.Wi + w0;
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const bool ok = (unsigned)(h0 + xro[j]) < (unsigned)p.Hi && xgc[j] < p.Wi - w0 && (unsigned)(dd - xdsh[j]) < (unsigned)p.Di;
        const unsigned off = ok ? (unsigned)(S + xoff[j]) : 0u;
        vx[rs][j] = *reinterpret_cast<gf4_p>(xbase[j] + off);
        ta[rs][j] = ok ? xa[j] : 0.f;
        tb[rs][j] = ok ? xb[j] : 0.f;
      }
    };
    auto request_y = [&](const int rs, const Pos& q) {
      const int h0 = q.ty * GTH, w0 = q.tx * GTW;
      const int sy = (q.d * in_plane + h0 * p.Wi + w0) * 4;
      const bool rowok = h0 + y_r < p.Hi && 4 * y_q < p.Wi - w0;
#pragma unroll
      for (int j = 0; j < 4; ++j) vy[rs][j] = llvm_raw_buffer_load_v4f32(dyr, rowok ? yoff[j] : (int)0x80000000, sy, 0);
      const bool hok = h0 + h_r < p.Hi && (unsigned)(w0 + (h_side ? GTW : -1)) < (unsigned)p.Wi;
#pragma unroll
      for (int k2 = 0; k2 < NH; ++k2) vh[rs][k2] = llvm_raw_buffer_load_f32(dyh, hok ? hoff[k2] : (int)0x80000000, sy, 0);
    };
    auto commit_x = [&](const int rs, const int j, unsigned char* img) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = fmaf(vx[rs][j][e], ta[rs][j], tb[rs][j]);
        v[e] = fmaxf(u, u * xsl[j]);                          // LeakyReLU with 0 <= slope <= 1 (the engine's contract)
      }
      unsigned char* dst = img + xdst[j];
      if constexpr (NPC == 3) {
        u32x2_t hi2, mid2, lo2;
        split4(v, hi2, mid2, lo2);
        *reinterpret_cast<u32x2_t*>(dst) = hi2;
        *reinterpret_cast<u32x2_t*>(dst + SSTR2) = mid2;
        *reinterpret_cast<u32x2_t*>(dst + 2 * SSTR2) = lo2;
      } else {
        u32x2_t hi2, lo2;
        split4h(v, hi2, lo2);
        *reinterpret_cast<u32x2_t*>(dst) = hi2;
        *reinterpret_cast<u32x2_t*>(dst + SSTR2) = lo2;
      }
    };
    auto commit_y = [&](const int rs, const int j, unsigned char* img) {
      const int ol = sw * 8 + 2 * j + (lane >> 5);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = NPC == 2 ? vy[rs][j][e] * ysc : vy[rs][j][e];
      unsigned char* dst = img + XBn + ol * CSTR2 + ydst;
      if constexpr (NPC == 3) {
        u32x2_t hi2, mid2, lo2;
        split4(v, hi2, mid2, lo2);
        *reinterpret_cast<u32x2_t*>(dst) = hi2;
        *reinterpret_cast<u32x2_t*>(dst + SSTR2) = mid2;
        *reinterpret_cast<u32x2_t*>(dst + 2 * SSTR2) = lo2;
      } else {
        u32x2_t hi2, lo2;
        split4h(v, hi2, lo2);
        *reinterpret_cast<u32x2_t*>(dst) = hi2;
        *reinterpret_cast<u32x2_t*>(dst + SSTR2) = lo2;
      }
    };
    auto commit_h = [&](const int rs, const int k2, unsigned char* img) {
      unsigned char* dst = img + XBn + hdst[k2];
      if constexpr (NPC == 3) {
        const float v = vh[rs][k2];
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const float r1 = v - __builtin_bit_cast(float, u & 0xffff0000u);
        const unsigned m = __builtin_bit_cast(unsigned, r1);
        const float r2 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
        const unsigned l = __builtin_bit_cast(unsigned, r2);
        *reinterpret_cast<unsigned short*>(dst) = (unsigned short)(u >> 16);
        *reinterpret_cast<unsigned short*>(dst + SSTR2) = (unsigned short)(m >> 16);
        *reinterpret_cast<unsigned short*>(dst + 2 * SSTR2) = (unsigned short)(l >> 16);
      } else {
        const float v = vh[rs][k2] * ysc;
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        *reinterpret_cast<_Float16*>(dst) = h;
        *reinterpret_cast<_Float16*>(dst + SSTR2) = l;
      }
    };
    // ---- matrix side (as bf3v4_mma) ----
    const int fr = lane & 31, fh8 = lane >> 5;
    // K block `half` of this wave: G = 0 the column half of tile row wr (+32 B), G = 1 tile row 2 wr + half (+ one row)
    constexpr int AH = G ? GYROWB : 32, BH = 32;
    const int a_off = XBn + fr * CSTR2 + (G ? 2 : 1) * wr * GYROWB + (8 + 8 * fh8) * 2;
    const int b_off = fr * CSTR2 + (G ? 2 : 1) * wr * GXROWB + 8 * fh8 * 2;
    u32x4_t r_an[NPC];
    unsigned r_prev[NPC], r_next[NPC];
    bf16x8 bq[2][NPC];
    auto read_a = [&](const unsigned char* img, int half) {
#pragma unroll
      for (int s = 0; s < NPC; ++s) {
        const unsigned char* ap = img + a_off + half * AH + s * SSTR2;
        r_an[s] = *reinterpret_cast<const u32x4_t*>(ap);
        r_prev[s] = *reinterpret_cast<const unsigned*>(ap - 4);
        r_next[s] = *reinterpret_cast<const unsigned*>(ap + 16);
      }
    };
    auto read_b = [&](const unsigned char* img, int half, int kh, bf16x8 (&b)[NPC]) {
#pragma unroll
      for (int s = 0; s < NPC; ++s) b[s] = *reinterpret_cast<const bf16x8*>(img + b_off + kh * GXROWB + half * BH + s * SSTR2);
    };

    // ---- prologue: tile 0 committed, tile 1 in the registers of set 1 ----
    Pos far;
    far.tx = tile_lo % p.tiles_x;
    const int tq = tile_lo / p.tiles_x;
    far.ty = tq % p.tiles_y;
    far.d = tq / p.tiles_y;
    request_x(0, far);
    request_y(0, far);
#pragma unroll
    for (int j = 0; j < NX; ++j) commit_x(0, j, lds);
#pragma unroll
    for (int j = 0; j < 4; ++j) commit_y(0, j, lds);
#pragma unroll
    for (int k2 = 0; k2 < NH; ++k2) commit_h(0, k2, lds);
    advance(far, ntiles > 1);
    request_x(1, far);
    request_y(1, far);
    advance(far, ntiles > 2);
    request_x(0, far);                                        // tile 2: its dy follows in phase 0 of tile 0
    __syncthreads();
    read_a(lds, 0);
    read_b(lds, 0, 0, bq[0]);

    auto tile = [&](const int t, const int rs) {               // rs: the register set that holds tile t+1
      const unsigned char* const img = lds + (t & 1) * BUFn;
      unsigned char* const imgn = lds + ((t & 1) ^ 1) * BUFn;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8 afr[3][NPC];                                   // [kw][piece]
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int ph = half * 3 + kh, cur = ph & 1;
          __builtin_amdgcn_sched_barrier(0);
          if (kh == 0) {                                      // (inside the phase: the shifts are dealt into its gaps too)
#pragma unroll
            for (int s = 0; s < NPC; ++s) {
              const u32x4_t an = r_an[s];
              const u32x4_t k0 = u32x4_t{__builtin_amdgcn_alignbit(an[1], an[0], 16), __builtin_amdgcn_alignbit(an[2], an[1], 16),
                                         __builtin_amdgcn_alignbit(an[3], an[2], 16), __builtin_amdgcn_alignbit(r_next[s], an[3], 16)};
              const u32x4_t k2 = u32x4_t{__builtin_amdgcn_alignbit(an[0], r_prev[s], 16), __builtin_amdgcn_alignbit(an[1], an[0], 16),
                                         __builtin_amdgcn_alignbit(an[2], an[1], 16), __builtin_amdgcn_alignbit(an[3], an[2], 16)};
              afr[0][s] = __builtin_bit_cast(bf16x8, k0);
              afr[1][s] = __builtin_bit_cast(bf16x8, an);
              afr[2][s] = __builtin_bit_cast(bf16x8, k2);
            }
          }
          if (ph == 5) {
            __syncthreads();                                  // tile t+1 committed; every read of this image has landed
            if (t + 1 < ntiles && !(WG5_DIAG & 4)) {
              read_a(imgn, 0);
              read_b(imgn, 0, 0, bq[cur ^ 1]);
            }
            // the set of tile t+1 is free (committed in phases 1-4): the input of tile t+3 goes there, a tile and a half ahead of
            // its conversion; past the end the last tile is requested and staged again
            advance(far, t + 3 < ntiles);
            if (!(WG5_DIAG & 2)) request_x(rs, far);
          } else {
            if (!(WG5_DIAG & 4)) {
              if (kh < 2) read_b(img, half, kh + 1, bq[cur ^ 1]);
              else read_b(img, 1, 0, bq[cur ^ 1]);
              if (ph == 1) read_a(img, 1);
            }
            if (ph == 0) {
              if (!(WG5_DIAG & 2)) request_y(rs ^ 1, far);     // dy of tile t+2 (its input was requested in phase 5 of tile t-1)
            } else if (WG5_DIAG & 1) {
            } else if (ph == 1) {
              commit_x(rs, 0, imgn); commit_x(rs, 1, imgn);
            } else if (ph == 2) {
              commit_x(rs, 2, imgn); commit_x(rs, 3, imgn); commit_h(rs, 0, imgn);
            } else if (ph == 3) {
              commit_x(rs, 4, imgn);
              if constexpr (G == 0) commit_x(rs, 5, imgn); else commit_h(rs, 1, imgn);
              commit_y(rs, 0, imgn);
            } else {
              commit_y(rs, 1, imgn); commit_y(rs, 2, imgn); commit_y(rs, 3, imgn);
            }
          }
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            f32x16 a = acc[kh * 3 + kw];
            if constexpr (NPC == 3) {
              a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][2], bq[cur][0], a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][1], bq[cur][1], a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bq[cur][2], a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][1], bq[cur][0], a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bq[cur][1], a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bq[cur][0], a, 0, 0, 0);
            } else {                                          // small terms first: lo*hi, hi*lo, then hi*hi
              a = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[kw][1]), __builtin_bit_cast(f16x8, bq[cur][0]), a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[kw][0]), __builtin_bit_cast(f16x8, bq[cur][1]), a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[kw][0]), __builtin_bit_cast(f16x8, bq[cur][0]), a, 0, 0, 0);
            }
            acc[kh * 3 + kw] = a;
          }
          if constexpr (NPC == 3) { WG5_GAP6() WG5_GAP6() WG5_GAP6() } else { WG5H_GAP9() }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    for (int t = 0; t < ntiles; t += 2) {
      tile(t, 1);
      if (t + 1 < ntiles) tile(t + 1, 0);
    }
  }

  // ---- sum of the four waves through LDS, fixed tree: (0 + 2) + (1 + 3) ----
  __syncthreads();
  float* const red = reinterpret_cast<float*>(lds);           // regions of 144 x 64 floats
  auto put = [&](int region) {
    float* dst = red + region * (144 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[(t * 16 + i) * 64] = acc[t][i];
  };
  auto add = [&](int region) {
    const float* src = red + region * (144 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] += src[(t * 16 + i) * 64];
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) add(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave == 0) {
    add(0);
    float* sp = p.slab + (long long)blockIdx.x * p.Cout * p.Cin * 9;
    const int c = cg * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int o = ob * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (o < p.Cout && c < p.Cin) {
        float* dst = sp + ((long long)o * p.Cin + c) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) dst[t] = NPC == 2 ? acc[t][i] * unsc * unsc_x : acc[t][i];
      }
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    atomicAdd(&g_wg_clock[0], __builtin_readcyclecounter() - clk_c0);
    atomicAdd(&g_wg_clock[1], __builtin_amdgcn_s_memrealtime() - clk_r0);
  }
}

// ---- diagnostic: one 32 x 32 output block of a GEMM through the SAME split functions and product orders as the kernels above -----
// D[m][n] = sum_k A[m][k] Bt[n][k]; mode 0 bf16 three-piece / six products, 1 fp16 two-piece / three products (Bt scaled from
// *absmax_b like dy), 2 the fp32-input MFMA (an fp32 FMA chain).  One wave; the test holds each against an fp64 evaluation.
template <int MODE>
__global__ __launch_bounds__(64) void diag_split_gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ D, int K,
                                                             const unsigned* __restrict__ absmax_b) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float ysc = 1.f, unsc = 1.f;
  if (MODE == 1 && absmax_b != nullptr) {
    int E = (int)((*absmax_b >> 23) & 0xffu);
    E = E < 1 ? 1 : E;
    int k = 141 - E;
    k = k > 120 ? 120 : (k < -120 ? -120 : k);
    ysc = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
    unsc = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
  }
  if (MODE == 2) {
    for (int k0 = 0; k0 < K; k0 += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(long long)r * K + k0 + h], Bt[(long long)r * K + k0 + h], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 16) {
      u32x4_t a[3], b[3];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float va[4], vb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          va[e] = A[(long long)r * K + k0 + 8 * h + 4 * q + e];
          vb[e] = Bt[(long long)r * K + k0 + 8 * h + 4 * q + e] * ysc;
        }
        u32x2_t p0, p1, p2;
        if (MODE == 0) {
          split4(va, p0, p1, p2);
          a[0][2 * q] = p0[0]; a[0][2 * q + 1] = p0[1]; a[1][2 * q] = p1[0]; a[1][2 * q + 1] = p1[1]; a[2][2 * q] = p2[0]; a[2][2 * q + 1] = p2[1];
          split4(vb, p0, p1, p2);
          b[0][2 * q] = p0[0]; b[0][2 * q + 1] = p0[1]; b[1][2 * q] = p1[0]; b[1][2 * q + 1] = p1[1]; b[2][2 * q] = p2[0]; b[2][2 * q + 1] = p2[1];
        } else {
          split4h(va, p0, p1);
          a[0][2 * q] = p0[0]; a[0][2 * q + 1] = p0[1]; a[1][2 * q] = p1[0]; a[1][2 * q + 1] = p1[1];
          split4h(vb, p0, p1);
          b[0][2 * q] = p0[0]; b[0][2 * q + 1] = p0[1]; b[1][2 * q] = p1[0]; b[1][2 * q + 1] = p1[1];
        }
      }
      if (MODE == 0) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[2]), __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, b[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i] * unsc;
}

}  // namespace

void e2e::wgrad_clock_read(unsigned long long out[2], bool reset) {
  const unsigned long long z[2] = {0, 0};
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_clock), sizeof(z));
  if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_clock), z, sizeof(z));
}

extern "C" int e2e_diag_split_gemm(const float* A, const float* Bt, float* D, int K, int mode, const unsigned* absmax_b, void* stream) {
  E2E_REQUIRE(A && Bt && D && K > 0 && K % 16 == 0 && mode >= 0 && mode <= 2, "diag_split_gemm: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(diag_split_gemm_kernel<0>, dim3(1), dim3(64), 0, st, A, Bt, D, K, absmax_b);
  else if (mode == 1) hipLaunchKernelGGL(diag_split_gemm_kernel<1>, dim3(1), dim3(64), 0, st, A, Bt, D, K, absmax_b);
  else hipLaunchKernelGGL(diag_split_gemm_kernel<2>, dim3(1), dim3(64), 0, st, A, Bt, D, K, absmax_b);
  return e2e::check_launch("diag_split_gemm_kernel");
}

namespace e2e {

int launch_wgrad_bf3(const WgBf3Params& p, int nchunks, int pairs, hipStream_t st) {
  // the caller checked: 32-bit element offsets inside one batch item's channel block / dy block
  if (p.geom == 1) {                                         // 8 x 16 tiles (planes 16..31 wide)
    if (p.h2) {
      hipLaunchKernelGGL((conv133_wgrad_bf3v5_kernel<1, 2>), dim3(nchunks, pairs), dim3(256), 0, st, p);
      return check_launch("conv133_wgrad_h2v5_kernel<1>");
    }
    hipLaunchKernelGGL((conv133_wgrad_bf3v5_kernel<1, 3>), dim3(nchunks, pairs), dim3(256), 0, st, p);
    return check_launch("conv133_wgrad_bf3v5_kernel<1>");
  }
  if (p.h2) {
    hipLaunchKernelGGL((conv133_wgrad_bf3v5_kernel<0, 2>), dim3(nchunks, pairs), dim3(256), 0, st, p);
    return check_launch("conv133_wgrad_h2v5_kernel");
  }
  hipLaunchKernelGGL((conv133_wgrad_bf3v5_kernel<0, 3>), dim3(nchunks, pairs), dim3(256), 0, st, p);
  return check_launch("conv133_wgrad_bf3v5_kernel");
}

}  // namespace e2e
