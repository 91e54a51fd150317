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
// Variants (E2E_WG_BF3): 5 = v5 (default: one wave per SIMD, staging dealt into the matrix-instruction gaps), 4 = v4 (four
// matrix waves + four staging waves), 2 = v2 (round 3: twelve waves that do both).  Same split, order per accumulator and
// reduction tree: bit-identical results.  (v1, 16x16x32 MFMA with eight waves, was removed in round 4.)
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

// ---- v2 (round 3): v_mfma_f32_32x32x16_bf16 and twelve waves ------------------------------------------------------------------
// An MFMA holds the SIMD's vector issue for 8 cycles whatever its shape, so the 16x16x32 form spends twice the issue slots
// of the 32x32x16 form on the same FLOPs (1728 of the 3456 cycles a tile takes at matrix rate).  Wave (r, kh) owns tile row r and kernel row kh: both 16-pixel K blocks of the row,
// the three taps (kh, 0..2), 3 x 16 accumulator registers, 36 matrix instructions per tile; twelve waves = three per SIMD
// (168 registers).  Staging roles: waves 0-7 convert four input channels each, waves 8-11 eight dy channels each (four float4
// rounds and one halo round); the two roles are two instantiations of the tile loop (no branch around a load inside it).
// Channel stride 400 B (= 25 x 16, odd): the 16 lanes of a ds_read_b128 group (one 8-element k group, 16 channels) hit 16
// distinct bank quads.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CSTR2 = 400, SSTR2 = 32 * CSTR2, XB2 = 3 * SSTR2, BUF2 = 2 * XB2;
static_assert(XROWS * XROWB + 16 == CSTR2 && TH * YROWB + 16 == CSTR2 && (CSTR2 / 16) % 2 == 1, "channel stride (v2)");
static_assert(2 * BUF2 <= 163840 && 6 * 48 * 64 * 4 <= 2 * BUF2, "LDS budget (v2)");

template <int ROLE>
__device__ __forceinline__ void bf3v2_body(const e2e::WgBf3Params& p, unsigned char* lds, f32x16 (&acc)[3], int n, int seg, int cg, int ob,
                                           int wave, int lane) {
  constexpr int NP = ROLE == 0 ? 4 : 5;                       // staging pieces of this wave per tile
  const int wr = wave & 3, wkh = wave >> 2;                 // tile row, kernel row
  const int obase = ob * 32, cbase = cg * 32;
  const long long in_plane = (long long)p.Hi * p.Wi;
  const int tile_lo = seg * p.tiles_per_chunk;
  int tile_hi = tile_lo + p.tiles_per_chunk;
  if (tile_hi > p.tiles_per_n) tile_hi = p.tiles_per_n;
  auto decode = [&](int tile, int& d0, int& h0, int& w0) {
    const int tx = tile % p.tiles_x;
    const int t = tile / p.tiles_x;
    const int ty = t % p.tiles_y;
    d0 = t / p.tiles_y;
    h0 = ty * TH;
    w0 = tx * TW;
  };
  // ---- role 0: four input channels of this wave (descriptors wave-uniform) ----
  gfloat_p xbase[4];
  float xa[4], xb[4], xsl[4];
  int xdsh[4];
  bool xval[4];
  if constexpr (ROLE == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = cbase + wave * 4 + k;
      xval[k] = c < p.Cin;
      const e2e_in_chan_t chd = p.chans[xval[k] ? c : 0];
      xdsh[k] = chd.dshift;
      xbase[k] = (gfloat_p)(chd.ptr + (long long)n * chd.nstride);
      xa[k] = 1.f; xb[k] = 0.f; xsl[k] = 1.f;
      if (xval[k] && chd.scale != nullptr) {
        xa[k] = chd.scale[(long long)n * chd.ab_nstride];
        xb[k] = chd.shift[(long long)n * chd.ab_nstride];
        xsl[k] = chd.slope;
      }
    }
  }
  const int xl = lane < 48 ? lane : 47;
  const int x_r = xl >> 3, x_q = xl & 7;
  // ---- role 1: eight dy channels of this wave: lane -> (channel lane / 32 + 2 it, row, group); halo: lane -> (channel, row, side) ----
  const int yw = (wave - 8) * 8;
  const int y_ch = lane >> 5, y_grp = lane & 31;
  const int y_r = y_grp >> 3, y_q = y_grp & 7;
  const int h_ch = lane >> 3, h_r = (lane >> 1) & 3, h_side = lane & 1;

  f32x4_t vs[4];
  float vh = 0.f;
  auto prefetch_piece = [&](int s, int d0, int h0, int w0) {
    if constexpr (ROLE == 0) {
      const int hi = h0 - 1 + x_r, gc = w0 + 4 * x_q;
      const int din = d0 * p.sd - xdsh[s];
      const bool ok = (unsigned)hi < (unsigned)p.Hi && gc < p.Wi && xval[s] && (unsigned)din < (unsigned)p.Di;
      vs[s] = *reinterpret_cast<gf4_p>(xbase[s] + (ok ? (long long)din * in_plane + (long long)hi * p.Wi + gc : 0));
    } else {
      if (s < 4) {
        const int ho = h0 + y_r, wo = w0 + 4 * y_q;
        const int o = obase + yw + y_ch + 2 * s;
        const bool ok = ho < p.Hi && wo < p.Wi && o < p.Cout;
        const long long off = ok ? (((long long)n * p.Cout + o) * p.Do + d0) * in_plane + (long long)ho * p.Wi + wo : 0;
        vs[s] = *reinterpret_cast<gf4_p>((gfloat_p)p.dy + off);
      } else {
        const int ho = h0 + h_r, wo = h_side ? w0 + TW : w0 - 1;
        const int o = obase + yw + h_ch;
        const bool ok = ho < p.Hi && (unsigned)wo < (unsigned)p.Wi && o < p.Cout;
        const long long off = ok ? (((long long)n * p.Cout + o) * p.Do + d0) * in_plane + (long long)ho * p.Wi + wo : 0;
        vh = ((gfloat_p)p.dy)[off];
      }
    }
  };
  auto commit_piece = [&](int s, int buf, int d0, int h0, int w0) {
    unsigned char* const xs = lds + buf * BUF2;
    unsigned char* const ys = xs + XB2;
    if constexpr (ROLE == 0) {
      const int hi = h0 - 1 + x_r, gc = w0 + 4 * x_q;
      const int din = d0 * p.sd - xdsh[s];
      const bool ok = (unsigned)hi < (unsigned)p.Hi && gc < p.Wi && xval[s] && (unsigned)din < (unsigned)p.Di;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = fmaf(vs[s][j], xa[s], xb[s]);
        const float t = fmaxf(u, u * xsl[s]);                 // LeakyReLU with 0 <= slope <= 1 (the engine's contract)
        v[j] = ok ? t : 0.f;
      }
      u32x2_t hi2, mid2, lo2;
      split4(v, hi2, mid2, lo2);
      unsigned char* dst = xs + (wave * 4 + s) * CSTR2 + x_r * XROWB + x_q * 8;
      *reinterpret_cast<u32x2_t*>(dst) = hi2;
      *reinterpret_cast<u32x2_t*>(dst + SSTR2) = mid2;
      *reinterpret_cast<u32x2_t*>(dst + 2 * SSTR2) = lo2;
    } else {
      if (s < 4) {
        const int ho = h0 + y_r, wo = w0 + 4 * y_q;
        const int ol = yw + y_ch + 2 * s;
        const bool ok = ho < p.Hi && wo < p.Wi && obase + ol < p.Cout;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ok ? vs[s][j] : 0.f;
        u32x2_t hi2, mid2, lo2;
        split4(v, hi2, mid2, lo2);
        unsigned char* dst = ys + ol * CSTR2 + y_r * YROWB + (8 + 4 * y_q) * 2;
        *reinterpret_cast<u32x2_t*>(dst) = hi2;
        *reinterpret_cast<u32x2_t*>(dst + SSTR2) = mid2;
        *reinterpret_cast<u32x2_t*>(dst + 2 * SSTR2) = lo2;
      } else {
        const int ho = h0 + h_r, wo = h_side ? w0 + TW : w0 - 1;
        const int ol = yw + h_ch;
        const bool ok = ho < p.Hi && (unsigned)wo < (unsigned)p.Wi && obase + ol < p.Cout;
        const float v = ok ? vh : 0.f;
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const float r1 = v - __builtin_bit_cast(float, u & 0xffff0000u);
        const unsigned m = __builtin_bit_cast(unsigned, r1);
        const float r2 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
        const unsigned l = __builtin_bit_cast(unsigned, r2);
        unsigned char* dst = ys + ol * CSTR2 + h_r * YROWB + (h_side ? 40 : 7) * 2;
        *reinterpret_cast<unsigned short*>(dst) = (unsigned short)(u >> 16);
        *reinterpret_cast<unsigned short*>(dst + SSTR2) = (unsigned short)(m >> 16);
        *reinterpret_cast<unsigned short*>(dst + 2 * SSTR2) = (unsigned short)(l >> 16);
      }
    }
  };

  if (tile_lo >= tile_hi) return;
  int d0, h0, w0;
  decode(tile_lo, d0, h0, w0);
#pragma unroll
  for (int s = 0; s < NP; ++s) prefetch_piece(s, d0, h0, w0);
#pragma unroll
  for (int s = 0; s < NP; ++s) commit_piece(s, 0, d0, h0, w0);
  int nd0 = d0, nh0 = h0, nw0 = w0;
  if (tile_lo + 1 < tile_hi) decode(tile_lo + 1, nd0, nh0, nw0);
#pragma unroll
  for (int s = 0; s < NP; ++s) prefetch_piece(s, nd0, nh0, nw0);
  int fd0 = nd0, fh0 = nh0, fw0 = nw0;
  __syncthreads();

  const int fr = lane & 31, fh8 = lane >> 5;
  const int a_off = XB2 + fr * CSTR2 + wr * YROWB + (8 + 8 * fh8) * 2;          // + 32 bytes per K block (half)
  const int b_off = fr * CSTR2 + (wr + wkh) * XROWB + 8 * fh8 * 2;

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int buf = (tile - tile_lo) & 1;
    decode(tile + 2 < tile_hi ? tile + 2 : tile_hi - 1, fd0, fh0, fw0);
    const unsigned char* const img = lds + buf * BUF2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      bf16x8 afr[3][3], bfr[3];                             // A: [kw][piece]
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const unsigned char* ap = img + a_off + half * 32 + s * SSTR2;
        const u32x4_t an = *reinterpret_cast<const u32x4_t*>(ap);
        const unsigned aprev = *reinterpret_cast<const unsigned*>(ap - 4);
        const unsigned anext = *reinterpret_cast<const unsigned*>(ap + 16);
        const u32x4_t k0 = u32x4_t{__builtin_amdgcn_alignbit(an[1], an[0], 16), __builtin_amdgcn_alignbit(an[2], an[1], 16),
                                   __builtin_amdgcn_alignbit(an[3], an[2], 16), __builtin_amdgcn_alignbit(anext, an[3], 16)};
        const u32x4_t k2 = u32x4_t{__builtin_amdgcn_alignbit(an[0], aprev, 16), __builtin_amdgcn_alignbit(an[1], an[0], 16),
                                   __builtin_amdgcn_alignbit(an[2], an[1], 16), __builtin_amdgcn_alignbit(an[3], an[2], 16)};
        afr[0][s] = __builtin_bit_cast(bf16x8, k0);
        afr[1][s] = __builtin_bit_cast(bf16x8, an);
        afr[2][s] = __builtin_bit_cast(bf16x8, k2);
        bfr[s] = *reinterpret_cast<const bf16x8*>(img + b_off + half * 32 + s * SSTR2);
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int g = half * 3 + kw;
        __builtin_amdgcn_sched_barrier(0);
        f32x16 a = acc[kw];
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][2], bfr[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][1], bfr[1], a, 0, 0, 0);
        if (g < NP) commit_piece(g < NP ? g : 0, buf ^ 1, nd0, nh0, nw0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bfr[2], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][1], bfr[0], a, 0, 0, 0);
        if (g >= 1 && g - 1 < NP) prefetch_piece(g >= 1 && g - 1 < NP ? g - 1 : 0, fd0, fh0, fw0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bfr[1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bfr[0], a, 0, 0, 0);
        acc[kw] = a;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    nd0 = fd0; nh0 = fh0; nw0 = fw0;
    __syncthreads();
  }
}

__global__ __launch_bounds__(768, 3) void conv133_wgrad_bf3v2_kernel(e2e::WgBf3Params p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF2];
  const int segs = p.segs;
  const int n = blockIdx.x / segs, seg = blockIdx.x - n * segs;
  const int cg = blockIdx.y % p.cblocks, ob = blockIdx.y / p.cblocks;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // staging roles by wave, MFMA roles by (wave & 3, wave >> 2): waves 8-11 are (row, kh = 2)
  if (wave < 8) bf3v2_body<0>(p, lds, acc, n, seg, cg, ob, wave, lane);
  else bf3v2_body<1>(p, lds, acc, n, seg, cg, ob, wave, lane);

  // ---- sum of the four row waves of each kernel row through LDS, fixed tree: (0 + 2) + (1 + 3) ----
  const int wr = wave & 3, wkh = wave >> 2;
  __syncthreads();
  float* const red = reinterpret_cast<float*>(lds);         // regions of 48 x 64 floats
  auto put = [&](int region) {
    float* dst = red + region * (48 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[(t * 16 + i) * 64] = acc[t][i];
  };
  auto add = [&](int region) {
    const float* src = red + region * (48 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] += src[(t * 16 + i) * 64];
  };
  if (wr >= 2) put(wkh * 2 + (wr - 2));
  __syncthreads();
  if (wr < 2) add(wkh * 2 + wr);
  __syncthreads();
  if (wr == 1) put(wkh * 2);
  __syncthreads();
  if (wr == 0) {
    add(wkh * 2);
    // C/D layout of v_mfma_f32_32x32x16: column (in channel) = lane & 31, row (out channel) = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    float* sp = p.slab + (long long)blockIdx.x * p.Cout * p.Cin * 9;
    const int c = cg * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int o = ob * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (o < p.Cout && c < p.Cin) {
        float* dst = sp + ((long long)o * p.Cin + c) * 9 + wkh * 3;
#pragma unroll
        for (int t = 0; t < 3; ++t) dst[t] = acc[t][i];
      }
    }
  }
}

// ---- v4: matrix waves and staging waves (producer / consumer) ---------------------------------------------------------------
// v2's twelve waves all do both jobs in lockstep behind one barrier: the three waves of a SIMD read their fragments together,
// convert together and issue matrix instructions together, so the phases add up (timing with phases switched off: matrix
// instructions alone 0.56 ms, + fragment reads 0.74, + conversion 0.90 on 64->32 @128^3 x 2).  tools/scratch/mfma_overlap.hip:
// the matrix instructions of one wave and the vector instructions of ANOTHER wave of the same SIMD do overlap.  So: eight waves,
// two per SIMD, 256 registers each.
//   * waves 0-3 (one per SIMD): matrix wave r owns tile row r, both 16-pixel K blocks, all nine taps (9 x 16 accumulator
//     registers), 108 matrix instructions per tile and nothing else but its fragment reads: dy row r once per K block (three
//     v2 waves read it), x rows r .. r+2 one kernel row at a time, double-buffered in registers and requested one phase
//     (18 matrix instructions) ahead; the dy words of the next K block two phases ahead.
//   * waves 4-7: staging wave s loads, normalises, splits and commits 8 input channels (6 rounds, lane -> (channel, row,
//     quad)), 8 dy channels (4 rounds) and their halo (1 round) of tile t+1 while the matrix waves work on tile t, and has the
//     loads of tile t+2 in flight.
//   * one barrier per tile; the matrix waves take it in front of their LAST phase (all reads of the tile's image have landed,
//     18 matrix instructions still to issue) and request the next tile's first fragments right behind it.
// Same LDS images, split and summation order per accumulator as v2 (bit-identical results).
#ifndef E2E_WG4_STAGE_PRIO
#define E2E_WG4_STAGE_PRIO 0
#endif
#ifdef E2E_CONV_DEBUG
__device__ unsigned long long g_wg4_stamps[8];   // matrix wave 0: [0] loop, [1] barrier; staging wave 4: [2] loop, [3] barrier, [4] request, [5] convert; [7] workgroups
#define WG4_T() __builtin_readcyclecounter()
#endif
__device__ __forceinline__ void bf3v4_mma(unsigned char* lds, f32x16 (&acc)[9], int ntiles, int wr, int lane) {
  const int fr = lane & 31, fh8 = lane >> 5;
  const int a_off = XB2 + fr * CSTR2 + wr * YROWB + (8 + 8 * fh8) * 2;          // + 32 bytes per K block
  const int b_off = fr * CSTR2 + wr * XROWB + 8 * fh8 * 2;                      // + kh rows
  u32x4_t r_an[3];
  unsigned r_prev[3], r_next[3];
  bf16x8 bq[2][3];
  auto read_a = [&](const unsigned char* img, int half) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const unsigned char* ap = img + a_off + half * 32 + s * SSTR2;
      r_an[s] = *reinterpret_cast<const u32x4_t*>(ap);
      r_prev[s] = *reinterpret_cast<const unsigned*>(ap - 4);
      r_next[s] = *reinterpret_cast<const unsigned*>(ap + 16);
    }
  };
  auto read_b = [&](const unsigned char* img, int half, int kh, bf16x8 (&b)[3]) {
#pragma unroll
    for (int s = 0; s < 3; ++s) b[s] = *reinterpret_cast<const bf16x8*>(img + b_off + kh * XROWB + half * 32 + s * SSTR2);
  };
  __syncthreads();                                            // image of the first tile committed
#ifdef E2E_CONV_DEBUG
  unsigned long long s_bar = 0;
  const unsigned long long s_t0 = WG4_T();
#endif
  read_a(lds, 0);
  read_b(lds, 0, 0, bq[0]);
  for (int t = 0; t < ntiles; ++t) {
    const unsigned char* const img = lds + (t & 1) * BUF2;
    const unsigned char* const imgn = lds + ((t & 1) ^ 1) * BUF2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      bf16x8 afr[3][3];                                       // [kw][piece]
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const u32x4_t an = r_an[s];
        const u32x4_t k0 = u32x4_t{__builtin_amdgcn_alignbit(an[1], an[0], 16), __builtin_amdgcn_alignbit(an[2], an[1], 16),
                                   __builtin_amdgcn_alignbit(an[3], an[2], 16), __builtin_amdgcn_alignbit(r_next[s], an[3], 16)};
        const u32x4_t k2 = u32x4_t{__builtin_amdgcn_alignbit(an[0], r_prev[s], 16), __builtin_amdgcn_alignbit(an[1], an[0], 16),
                                   __builtin_amdgcn_alignbit(an[2], an[1], 16), __builtin_amdgcn_alignbit(an[3], an[2], 16)};
        afr[0][s] = __builtin_bit_cast(bf16x8, k0);
        afr[1][s] = __builtin_bit_cast(bf16x8, an);
        afr[2][s] = __builtin_bit_cast(bf16x8, k2);
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ph = half * 3 + kh, cur = ph & 1;
        __builtin_amdgcn_sched_barrier(0);
        if (ph == 5) {
#ifdef E2E_CONV_DEBUG
          asm volatile("s_waitcnt lgkmcnt(0)");
          const unsigned long long s_b0 = WG4_T();
          __syncthreads();
          s_bar += WG4_T() - s_b0;
#else
          __syncthreads();                                    // every read of this image has landed; tile t+1 is committed
#endif
          if (t + 1 < ntiles) {
            read_a(imgn, 0);
            read_b(imgn, 0, 0, bq[cur ^ 1]);
          }
        } else {
          if (kh < 2) read_b(img, half, kh + 1, bq[cur ^ 1]);
          else read_b(img, 1, 0, bq[cur ^ 1]);
          if (ph == 1) read_a(img, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          f32x16 a = acc[kh * 3 + kw];
          // small terms first: lo*hi, mid*mid, hi*lo, then mid*hi, hi*mid, then hi*hi
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][2], bq[cur][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][1], bq[cur][1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bq[cur][2], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][1], bq[cur][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bq[cur][1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bq[cur][0], a, 0, 0, 0);
          acc[kh * 3 + kw] = a;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#ifdef E2E_CONV_DEBUG
  if (wr == 0 && lane == 0) { atomicAdd(&g_wg4_stamps[0], WG4_T() - s_t0); atomicAdd(&g_wg4_stamps[1], s_bar); atomicAdd(&g_wg4_stamps[7], 1ull); }
#endif
  if (ntiles & 1) __syncthreads();                            // the staging waves run their tiles in pairs
}

// The staging wave is ONE instruction stream per SIMD (a lone wave issues a vector instruction every ~5 cycles at best), so its
// length per tile is what the kernel runs at once the matrix waves are fed: tile coordinates advance by counters (no
// divisions), every address is a per-lane constant plus a per-tile scalar, dy comes through a buffer descriptor (lanes outside
// the plane read zeros: no predicate at commit time), the input's zero fill rides in the per-lane (scale, shift) pair.
__device__ __forceinline__ void bf3v4_stage(const e2e::WgBf3Params& p, unsigned char* lds, int n, int tile_lo, int ntiles, int cg, int ob,
                                            int sw, int lane) {
  const int obase = ob * 32, cbase = cg * 32;
  const int in_plane = p.Hi * p.Wi;                           // (host: Di * Hi * Wi < 2^29, Cout * Do * Hi * Wi < 2^29)
  __builtin_amdgcn_s_setprio(E2E_WG4_STAGE_PRIO);             // (A/B knob: priorities 0..3 measured equal, tools/scratch/wg_prio.sh)
  // ---- input: round j, lane l -> item 64 j + l = (channel of this wave, row, quad); descriptors per lane ----
  gfloat_p xbase[6];
  float xa[6], xb[6], xsl[6];
  int xdsh[6], xro[6], xgc[6], xoff[6], xdst[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int item = j * 64 + lane;
    const int chl = item / 48, rem = item - chl * 48;
    const int x_r = rem >> 3, x_q = rem & 7;
    const int c = cbase + sw * 8 + chl;
    const bool val = c < p.Cin;
    const e2e_in_chan_t* chd = p.chans + (val ? c : 0);
    xdsh[j] = val ? chd->dshift : (1 << 30);                  // an absent channel fails the depth test of every tile
    xbase[j] = (gfloat_p)(chd->ptr + (long long)n * chd->nstride);
    xa[j] = 1.f; xb[j] = 0.f; xsl[j] = 1.f;
    if (val && chd->scale != nullptr) {
      xa[j] = chd->scale[(long long)n * chd->ab_nstride];
      xb[j] = chd->shift[(long long)n * chd->ab_nstride];
      xsl[j] = chd->slope;
    }
    xro[j] = x_r - 1;
    xgc[j] = 4 * x_q;
    xoff[j] = (x_r - 1) * p.Wi + 4 * x_q - (val ? chd->dshift : 0) * in_plane;
    xdst[j] = (sw * 8 + chl) * CSTR2 + x_r * XROWB + x_q * 8;
  }
  // ---- dy: round j, lane l -> (channel 2 j + l / 32 of this wave, row, quad); halo: lane -> (channel, row, side) ----
  const int y_grp = lane & 31, y_r = y_grp >> 3, y_q = y_grp & 7;
  const int h_ch = lane >> 3, h_r = (lane >> 1) & 3, h_side = lane & 1;
  // buffer descriptor of this batch item's dy: base, stride 0, bytes, raw 32-bit format; a lane whose offset has bit 31 set is
  // out of range and reads zeros
  const unsigned long long dya = (unsigned long long)(p.dy + (long long)n * p.Cout * p.Do * in_plane);
  const i32x4_t dyr = {__builtin_amdgcn_readfirstlane((int)dya), __builtin_amdgcn_readfirstlane((int)(dya >> 32) & 0xffff),
                       __builtin_amdgcn_readfirstlane(p.Cout * p.Do * in_plane * 4), 0x00020000};
  int yoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = obase + sw * 8 + 2 * j + (lane >> 5);
    yoff[j] = o < p.Cout ? ((o * p.Do) * in_plane + y_r * p.Wi + 4 * y_q) * 4 : (int)0x80000000;
  }
  // the halo's descriptor starts one float earlier: the left neighbour of column 0 of row 0 would be a negative lane offset
  const i32x4_t dyh = {__builtin_amdgcn_readfirstlane((int)(dya - 4)), __builtin_amdgcn_readfirstlane((int)((dya - 4) >> 32) & 0xffff),
                       __builtin_amdgcn_readfirstlane(p.Cout * p.Do * in_plane * 4 + 4), 0x00020000};
  const int ho_ = obase + sw * 8 + h_ch;
  const int hoff = ho_ < p.Cout ? ((ho_ * p.Do) * in_plane + h_r * p.Wi + (h_side ? TW + 1 : 0)) * 4 : (int)0x80000000;

  struct Pos { int tx, ty, d; };
  auto advance = [&](Pos& q, bool go) {                       // branch-free: scalar selects only (one basic block per tile)
    const int tx = q.tx + 1;
    const bool wx = tx == p.tiles_x;
    const int ty = q.ty + (wx ? 1 : 0);
    const bool wy = ty == p.tiles_y;
    q.tx = go ? (wx ? 0 : tx) : q.tx;
    q.ty = go ? (wy ? 0 : ty) : q.ty;
    q.d = go ? q.d + (wy ? 1 : 0) : q.d;
  };

  // two register sets: the loads of tile t+2 are issued BEFORE tile t+1 is converted (a full tile of latency cover)
  f32x4_t vx[2][6], vy[2][4];
  float vh[2] = {0.f, 0.f};
  float ta[2][6], tb[2][6];                                   // (scale, shift) of the lane for the tile in registers; (0, 0) outside
  auto prefetch = [&](const int rs, const Pos& q) {
    const int h0 = q.ty * TH, w0 = q.tx * TW;
    const int dd = q.d * p.sd;
    const int S = dd * in_plane + h0 * p.Wi + w0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const bool ok = (unsigned)(h0 + xro[j]) < (unsigned)p.Hi && xgc[j] < p.Wi - w0 && (unsigned)(dd - xdsh[j]) < (unsigned)p.Di;
      const unsigned off = ok ? (unsigned)(S + xoff[j]) : 0u;
      vx[rs][j] = *reinterpret_cast<gf4_p>(xbase[j] + off);
      ta[rs][j] = ok ? xa[j] : 0.f;
      tb[rs][j] = ok ? xb[j] : 0.f;
    }
    const int sy = (q.d * in_plane + h0 * p.Wi + w0) * 4;
    const bool rowok = h0 + y_r < p.Hi && 4 * y_q < p.Wi - w0;
#pragma unroll
    for (int j = 0; j < 4; ++j) vy[rs][j] = llvm_raw_buffer_load_v4f32(dyr, rowok ? yoff[j] : (int)0x80000000, sy, 0);
    const bool hok = h0 + h_r < p.Hi && (unsigned)(w0 + (h_side ? TW : -1)) < (unsigned)p.Wi;
    vh[rs] = llvm_raw_buffer_load_f32(dyh, hok ? hoff : (int)0x80000000, sy, 0);
  };
  auto commit = [&](const int rs, int buf) {
    unsigned char* const xs = lds + buf * BUF2;
    unsigned char* const ys = xs + XB2;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = fmaf(vx[rs][j][e], ta[rs][j], tb[rs][j]);
        v[e] = fmaxf(u, u * xsl[j]);                          // LeakyReLU with 0 <= slope <= 1 (the engine's contract)
      }
      u32x2_t hi2, mid2, lo2;
      split4(v, hi2, mid2, lo2);
      unsigned char* dst = xs + xdst[j];
      *reinterpret_cast<u32x2_t*>(dst) = hi2;
      *reinterpret_cast<u32x2_t*>(dst + SSTR2) = mid2;
      *reinterpret_cast<u32x2_t*>(dst + 2 * SSTR2) = lo2;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ol = sw * 8 + 2 * j + (lane >> 5);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = vy[rs][j][e];
      u32x2_t hi2, mid2, lo2;
      split4(v, hi2, mid2, lo2);
      unsigned char* dst = ys + ol * CSTR2 + y_r * YROWB + (8 + 4 * y_q) * 2;
      *reinterpret_cast<u32x2_t*>(dst) = hi2;
      *reinterpret_cast<u32x2_t*>(dst + SSTR2) = mid2;
      *reinterpret_cast<u32x2_t*>(dst + 2 * SSTR2) = lo2;
    }
    {
      const int ol = sw * 8 + h_ch;
      const float v = vh[rs];
      const unsigned u = __builtin_bit_cast(unsigned, v);
      const float r1 = v - __builtin_bit_cast(float, u & 0xffff0000u);
      const unsigned m = __builtin_bit_cast(unsigned, r1);
      const float r2 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
      const unsigned l = __builtin_bit_cast(unsigned, r2);
      unsigned char* dst = ys + ol * CSTR2 + h_r * YROWB + (h_side ? 40 : 7) * 2;
      *reinterpret_cast<unsigned short*>(dst) = (unsigned short)(u >> 16);
      *reinterpret_cast<unsigned short*>(dst + SSTR2) = (unsigned short)(m >> 16);
      *reinterpret_cast<unsigned short*>(dst + 2 * SSTR2) = (unsigned short)(l >> 16);
    }
  };

  Pos far;
  far.tx = tile_lo % p.tiles_x;
  const int tq = tile_lo / p.tiles_x;
  far.ty = tq % p.tiles_y;
  far.d = tq / p.tiles_y;
  prefetch(0, far);
  commit(0, 0);
  advance(far, ntiles > 1);
  prefetch(1, far);
  __syncthreads();                                            // image of the first tile committed
#ifdef E2E_CONV_DEBUG
  unsigned long long s_bar = 0, s_req = 0, s_cvt = 0;
  const unsigned long long s_t0 = WG4_T();
#define WG4_SEG(acc_, body_) { const unsigned long long s_a = WG4_T(); body_; acc_ += WG4_T() - s_a; }
#else
#define WG4_SEG(acc_, body_) { body_; }
#endif
  for (int t = 0; t < ntiles; t += 2) {
    // registers of set 1 hold tile t+1: request tile t+2 into set 0, then convert tile t+1 into the image the matrix waves are
    // not reading.  Past the end the last tile is staged again into the image nobody reads (no branch around the loads; an odd
    // run ends with one barrier the matrix waves answer behind their loop).
    advance(far, t + 2 < ntiles);
    WG4_SEG(s_req, prefetch(0, far));
    __builtin_amdgcn_sched_barrier(0);                        // the requests stay in front of the conversion
    WG4_SEG(s_cvt, commit(1, 1));
    WG4_SEG(s_bar, __syncthreads());
    advance(far, t + 3 < ntiles);
    WG4_SEG(s_req, prefetch(1, far));
    __builtin_amdgcn_sched_barrier(0);
    WG4_SEG(s_cvt, commit(0, 0));
    WG4_SEG(s_bar, __syncthreads());
  }
#ifdef E2E_CONV_DEBUG
  if (sw == 0 && lane == 0) { atomicAdd(&g_wg4_stamps[2], WG4_T() - s_t0); atomicAdd(&g_wg4_stamps[3], s_bar); atomicAdd(&g_wg4_stamps[4], s_req); atomicAdd(&g_wg4_stamps[5], s_cvt); }
#endif
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv133_wgrad_bf3v4_kernel(e2e::WgBf3Params p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF2];
  const int segs = p.segs;
  const int n = blockIdx.x / segs, seg = blockIdx.x - n * segs;
  const int cg = blockIdx.y % p.cblocks, ob = blockIdx.y / p.cblocks;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_lo = seg * p.tiles_per_chunk;
  int tile_hi = tile_lo + p.tiles_per_chunk;
  if (tile_hi > p.tiles_per_n) tile_hi = p.tiles_per_n;
  const int ntiles = tile_hi - tile_lo;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  if (ntiles > 0) {
    if (wave < 4) bf3v4_mma(lds, acc, ntiles, wave, lane);
    else bf3v4_stage(p, lds, n, tile_lo, ntiles, cg, ob, wave - 4, lane);
  }

  // ---- sum of the four matrix waves through LDS, fixed tree: (0 + 2) + (1 + 3) ----
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
  if (wave == 2 || wave == 3) put(wave - 2);
  __syncthreads();
  if (wave < 2) add(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave == 0) {
    add(0);
    // C/D layout of v_mfma_f32_32x32x16: column (in channel) = lane & 31, row (out channel) = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    float* sp = p.slab + (long long)blockIdx.x * p.Cout * p.Cin * 9;
    const int c = cg * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int o = ob * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (o < p.Cout && c < p.Cin) {
        float* dst = sp + ((long long)o * p.Cin + c) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) dst[t] = acc[t][i];
      }
    }
  }
}

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
#ifndef WG5_DIAG
#define WG5_DIAG 0                 // timing-only builds: 1 no conversion / commit, 2 no global requests, 4 no fragment reads
#endif
// G = 0: 4 x 32-pixel tiles (planes at least 32 wide; the two K blocks of a wave are the halves of its tile row).
// G = 1: 8 x 16-pixel tiles for planes 16..31 wide (verdict item 2: these ran on fp32 MFMA): the two K blocks of wave r are the tile
//        rows 2r and 2r+1; dy rows of 48 B (pixels at entries 8..23, halos at 7 and 24 -- entry 24 shares the unused head of the
//        next row), input rows of 32 B, the same 400 B channel stride: same images, same pipeline, 5 + 4 + 2 pieces per wave.
template <int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv133_wgrad_bf3v5_kernel(e2e::WgBf3Params p) {
  constexpr int GTH = G ? 8 : 4, GTW = G ? 16 : 32;         // tile
  constexpr int NX = G ? 5 : 6, NH = G ? 2 : 1;             // input / halo pieces per wave (dy: 4)
  constexpr int XIT = G ? 40 : 48, XQS = G ? 2 : 3;         // items (rows x quads) per input channel, log2 quads per row
  constexpr int GXROWB = G ? 32 : 64, GYROWB = G ? 48 : 96; // bytes per staged input / dy row
  static_assert((GTH + 2) * GXROWB + 16 <= CSTR2 && GTH * GYROWB + 16 <= CSTR2, "channel stride");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF2];
  const int segs = p.segs;
  const int n = blockIdx.x / segs, seg = blockIdx.x - n * segs;
  const int cg = blockIdx.y % p.cblocks, ob = blockIdx.y / p.cblocks;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_lo = seg * p.tiles_per_chunk;
  int tile_hi = tile_lo + p.tiles_per_chunk;
  if (tile_hi > p.tiles_per_n) tile_hi = p.tiles_per_n;
  const int ntiles = tile_hi - tile_lo;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

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
      u32x2_t hi2, mid2, lo2;
      split4(v, hi2, mid2, lo2);
      unsigned char* dst = img + xdst[j];
      *reinterpret_cast<u32x2_t*>(dst) = hi2;
      *reinterpret_cast<u32x2_t*>(dst + SSTR2) = mid2;
      *reinterpret_cast<u32x2_t*>(dst + 2 * SSTR2) = lo2;
    };
    auto commit_y = [&](const int rs, const int j, unsigned char* img) {
      const int ol = sw * 8 + 2 * j + (lane >> 5);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = vy[rs][j][e];
      u32x2_t hi2, mid2, lo2;
      split4(v, hi2, mid2, lo2);
      unsigned char* dst = img + XB2 + ol * CSTR2 + ydst;
      *reinterpret_cast<u32x2_t*>(dst) = hi2;
      *reinterpret_cast<u32x2_t*>(dst + SSTR2) = mid2;
      *reinterpret_cast<u32x2_t*>(dst + 2 * SSTR2) = lo2;
    };
    auto commit_h = [&](const int rs, const int k2, unsigned char* img) {
      const float v = vh[rs][k2];
      const unsigned u = __builtin_bit_cast(unsigned, v);
      const float r1 = v - __builtin_bit_cast(float, u & 0xffff0000u);
      const unsigned m = __builtin_bit_cast(unsigned, r1);
      const float r2 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
      const unsigned l = __builtin_bit_cast(unsigned, r2);
      unsigned char* dst = img + XB2 + hdst[k2];
      *reinterpret_cast<unsigned short*>(dst) = (unsigned short)(u >> 16);
      *reinterpret_cast<unsigned short*>(dst + SSTR2) = (unsigned short)(m >> 16);
      *reinterpret_cast<unsigned short*>(dst + 2 * SSTR2) = (unsigned short)(l >> 16);
    };
    // ---- matrix side (as bf3v4_mma) ----
    const int fr = lane & 31, fh8 = lane >> 5;
    // K block `half` of this wave: G = 0 the column half of tile row wr (+32 B), G = 1 tile row 2 wr + half (+ one row)
    constexpr int AH = G ? GYROWB : 32, BH = 32;
    const int a_off = XB2 + fr * CSTR2 + (G ? 2 : 1) * wr * GYROWB + (8 + 8 * fh8) * 2;
    const int b_off = fr * CSTR2 + (G ? 2 : 1) * wr * GXROWB + 8 * fh8 * 2;
    u32x4_t r_an[3];
    unsigned r_prev[3], r_next[3];
    bf16x8 bq[2][3];
    auto read_a = [&](const unsigned char* img, int half) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const unsigned char* ap = img + a_off + half * AH + s * SSTR2;
        r_an[s] = *reinterpret_cast<const u32x4_t*>(ap);
        r_prev[s] = *reinterpret_cast<const unsigned*>(ap - 4);
        r_next[s] = *reinterpret_cast<const unsigned*>(ap + 16);
      }
    };
    auto read_b = [&](const unsigned char* img, int half, int kh, bf16x8 (&b)[3]) {
#pragma unroll
      for (int s = 0; s < 3; ++s) b[s] = *reinterpret_cast<const bf16x8*>(img + b_off + kh * GXROWB + half * BH + s * SSTR2);
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
      const unsigned char* const img = lds + (t & 1) * BUF2;
      unsigned char* const imgn = lds + ((t & 1) ^ 1) * BUF2;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8 afr[3][3];                                     // [kw][piece]
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int ph = half * 3 + kh, cur = ph & 1;
          __builtin_amdgcn_sched_barrier(0);
          if (kh == 0) {                                      // (inside the phase: the shifts are dealt into its gaps too)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
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
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][2], bq[cur][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][1], bq[cur][1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bq[cur][2], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][1], bq[cur][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bq[cur][1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kw][0], bq[cur][0], a, 0, 0, 0);
            acc[kh * 3 + kw] = a;
          }
          WG5_GAP6() WG5_GAP6() WG5_GAP6()
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
        for (int t = 0; t < 9; ++t) dst[t] = acc[t][i];
      }
    }
  }
}

}  // namespace

namespace e2e {

int launch_wgrad_bf3(const WgBf3Params& p, int nchunks, int pairs, hipStream_t st) {
  static const int variant = getenv("E2E_WG_BF3") ? atoi(getenv("E2E_WG_BF3")) : 5;       // 5: v5 (default), 4: v4, 2: v2 (A/B; 0 = the fp32-MFMA kernels, decided by the caller)
  // v4 addresses with 32-bit element offsets inside one batch item's channel block / dy block
  const bool fits32 = (long long)p.Di * p.Hi * p.Wi < (1ll << 29) && (long long)p.Cout * p.Do * p.Hi * p.Wi < (1ll << 29);
  if (p.geom == 1) {                                         // 8 x 16 tiles (planes 16..31 wide): v5 only
    hipLaunchKernelGGL(conv133_wgrad_bf3v5_kernel<1>, dim3(nchunks, pairs), dim3(256), 0, st, p);
    return check_launch("conv133_wgrad_bf3v5_kernel<1>");
  }
  if (variant >= 5 && fits32) {
    hipLaunchKernelGGL(conv133_wgrad_bf3v5_kernel<0>, dim3(nchunks, pairs), dim3(256), 0, st, p);
    return check_launch("conv133_wgrad_bf3v5_kernel");
  }
  if (variant == 4 && fits32) {
    hipLaunchKernelGGL(conv133_wgrad_bf3v4_kernel, dim3(nchunks, pairs), dim3(512), 0, st, p);
#ifdef E2E_CONV_DEBUG
    if (getenv("E2E_WG_STAMPS")) {
      (void)hipStreamSynchronize(st);
      unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wg4_stamps), sizeof(h));
      const double w = h[7] ? (double)h[7] : 1.0, tl = (double)p.tiles_per_chunk;
      fprintf(stderr, "[wgrad bf3v4 %d->%d] per tile, counter ticks: matrix wave loop %.0f (barrier wait %.0f) | staging wave loop %.0f: request %.0f convert %.0f barrier wait %.0f\n",
              p.Cin, p.Cout, h[0] / w / tl, h[1] / w / tl, h[2] / w / tl, h[4] / w / tl, h[5] / w / tl, h[3] / w / tl);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg4_stamps), z, sizeof(z));
    }
#endif
    return check_launch("conv133_wgrad_bf3v4_kernel");
  }
  hipLaunchKernelGGL(conv133_wgrad_bf3v2_kernel, dim3(nchunks, pairs), dim3(768), 0, st, p);
  return check_launch("conv133_wgrad_bf3v2_kernel");
}

}  // namespace e2e
