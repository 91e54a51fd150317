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
// Geometry.  One workgroup = 32 out x 32 in channels x 9 taps over a run of 4 x 32-pixel tiles of one batch item
// (512 threads, two LDS images of 78 KB, one barrier per tile).  GEMM per tile and tap: M = out channel, N = in channel,
// K = the 32 pixels of a tile row (v_mfma_f32_16x16x32_bf16).  Wave (r, oh) owns tile row r and the out-channel half oh:
// one K block per tile, 16 out x 32 in channels x 9 taps = 18 accumulator tiles of 4 registers, 108 matrix instructions
// per tile.  (The 32x32x16 shape needs 144 accumulator registers per wave for the same block and spilled at two waves per
// SIMD; one wave per SIMD halves the VALU issue rate the staging conversion needs.)
//   * B fragments (input): x[c][row r + kh][32 px]: 8 consecutive bf16 per lane = one aligned ds_read_b128; no column halo:
//   * the tap's column shift is put on the dy side: dW[.,kw] = sum_w' dy[w' - kw + 1] x[w'].  The A fragment of kw = 1 is an
//     aligned ds_read_b128 of dy; kw = 0 / 2 are the same 8 elements moved by one bf16, built in registers with
//     v_alignbit_b32 from the aligned read plus the two neighbouring dwords (no shifted copies in LDS);
//   * channel stride 416 B (= 26 x 16, 26 = 2 mod 4): the 16 lanes of every ds_read_b128 group (16 channels x 4 k groups
//     in the hardware's lane grouping) hit 16 distinct bank quads;
//   * staging as in conv133_wgrad_v3: a wave stages 4 input and 4 dy channels (descriptors wave-uniform), the registers of
//     tile t+1 are converted and committed piece by piece between the matrix instructions of tile t, the loads of tile t+2
//     are requested one piece behind, all in straight-line code (exact vmcnt distances);
//   * the eight waves' accumulators are added through LDS in a fixed tree at the end; one slab per workgroup, reduced over
//     the chunks by wgrad_slab_reduce_kernel (deterministic).
#include "e2e_common.h"
#include <cstdlib>

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef const f32x4_t __attribute__((address_space(1)))* gf4_p;

constexpr int TH = 4, TW = 32;
constexpr int XROWS = TH + 2;
constexpr int CSTR = 416;                       // bytes per (split, channel): 6 rows x 64 B (x) or 4 rows x 96 B (dy), + 32
constexpr int SSTR = 32 * CSTR;                 // bytes per split
constexpr int XBYTES = 3 * SSTR, YBYTES = 3 * SSTR;
constexpr int BUF = XBYTES + YBYTES;            // 79 872 B per image
constexpr int YROWB = 96;                       // dy row: 48 bf16 (tile columns at 8..39, halo at 7 and 40)
constexpr int XROWB = 64;
constexpr int NPIECE = 7;                       // staging pieces per wave and tile: 4 input channels, 2 dy rounds, 1 dy halo round
static_assert(XROWS * XROWB + 32 == CSTR && TH * YROWB + 32 == CSTR && (CSTR / 16) % 4 == 2, "channel stride");
static_assert(2 * BUF <= 163840 && 4 * 72 * 64 * 4 <= 2 * BUF, "LDS budget; the four reduction regions fit the two images");

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

__global__ __launch_bounds__(512, 2) void conv133_wgrad_bf3_kernel(e2e::WgBf3Params p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

  const int segs = p.segs;
  const int n = blockIdx.x / segs, seg = blockIdx.x - n * segs;
  const int cg = blockIdx.y % p.cblocks, ob = blockIdx.y / p.cblocks;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 3, woh = wave >> 2;                // tile row (K block) and out-channel half of this wave
  const int obase = ob * 32, cbase = cg * 32;
  const long long in_plane = (long long)p.Hi * p.Wi;       // (stride 1 in the plane: Ho == Hi, Wo == Wi)

  f32x4_t acc[2][9];                                        // [in-channel half][tap]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

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

  // ---- wave-uniform descriptors of this wave's 4 input channels -------------------------------------------------------
  gfloat_p xbase[4];
  float xa[4], xb[4], xsl[4];
  int xdsh[4];
  bool xval[4];
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

  // ---- per-lane staging geometry ------------------------------------------------------------------------------------
  const int xl = lane < 48 ? lane : 47;                    // 6 rows x 8 float4 groups per input channel (idle lanes: harmless copies)
  const int x_r = xl >> 3, x_q = xl & 7;
  const int y_ch = lane >> 5, y_grp = lane & 31;           // dy: lane -> (channel lane / 32 + 2 it, row grp / 8, group grp % 8)
  const int y_r = y_grp >> 3, y_q = y_grp & 7;
  const int hl = lane & 31;                                // dy halo: lane -> (channel hl / 8, row (hl / 2) % 4, side hl % 2)
  const int h_ch = hl >> 3, h_r = (hl >> 1) & 3, h_side = hl & 1;

  f32x4_t vx[4], vy[2];
  float vh;
  auto prefetch_piece = [&](int s, int d0, int h0, int w0) {
    if (s < 4) {
      const int hi = h0 - 1 + x_r, gc = w0 + 4 * x_q;
      const int din = d0 * p.sd - xdsh[s];
      const bool ok = (unsigned)hi < (unsigned)p.Hi && gc < p.Wi && xval[s] && (unsigned)din < (unsigned)p.Di;
      vx[s] = *reinterpret_cast<gf4_p>(xbase[s] + (ok ? (long long)din * in_plane + (long long)hi * p.Wi + gc : 0));
    } else if (s < 6) {
      const int it = s - 4;
      const int ho = h0 + y_r, wo = w0 + 4 * y_q;
      const int o = obase + wave * 4 + y_ch + 2 * it;
      const bool ok = ho < p.Hi && wo < p.Wi && o < p.Cout;
      const long long off = ok ? (((long long)n * p.Cout + o) * p.Do + d0) * in_plane + (long long)ho * p.Wi + wo : 0;
      vy[it] = *reinterpret_cast<gf4_p>((gfloat_p)p.dy + off);
    } else {
      const int ho = h0 + h_r, wo = h_side ? w0 + TW : w0 - 1;
      const int o = obase + wave * 4 + h_ch;
      const bool ok = ho < p.Hi && (unsigned)wo < (unsigned)p.Wi && o < p.Cout;
      const long long off = ok ? (((long long)n * p.Cout + o) * p.Do + d0) * in_plane + (long long)ho * p.Wi + wo : 0;
      vh = ((gfloat_p)p.dy)[off];
    }
  };
  auto commit_piece = [&](int s, int buf, int d0, int h0, int w0) {
    unsigned char* const xs = lds + buf * BUF;
    unsigned char* const ys = xs + XBYTES;
    if (s < 4) {
      const int hi = h0 - 1 + x_r, gc = w0 + 4 * x_q;
      const int din = d0 * p.sd - xdsh[s];
      const bool ok = (unsigned)hi < (unsigned)p.Hi && gc < p.Wi && xval[s] && (unsigned)din < (unsigned)p.Di;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = e2e::in_act(vx[s][j], xa[s], xb[s], xsl[s]);
        v[j] = ok ? t : 0.f;
      }
      u32x2_t hi2, mid2, lo2;
      split4(v, hi2, mid2, lo2);
      unsigned char* dst = xs + (wave * 4 + s) * CSTR + x_r * XROWB + x_q * 8;
      *reinterpret_cast<u32x2_t*>(dst) = hi2;
      *reinterpret_cast<u32x2_t*>(dst + SSTR) = mid2;
      *reinterpret_cast<u32x2_t*>(dst + 2 * SSTR) = lo2;
    } else if (s < 6) {
      const int it = s - 4;
      const int ho = h0 + y_r, wo = w0 + 4 * y_q;
      const int ol = wave * 4 + y_ch + 2 * it;
      const bool ok = ho < p.Hi && wo < p.Wi && obase + ol < p.Cout;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = ok ? vy[it][j] : 0.f;
      u32x2_t hi2, mid2, lo2;
      split4(v, hi2, mid2, lo2);
      unsigned char* dst = ys + ol * CSTR + y_r * YROWB + (8 + 4 * y_q) * 2;
      *reinterpret_cast<u32x2_t*>(dst) = hi2;
      *reinterpret_cast<u32x2_t*>(dst + SSTR) = mid2;
      *reinterpret_cast<u32x2_t*>(dst + 2 * SSTR) = lo2;
    } else {
      const int ho = h0 + h_r, wo = h_side ? w0 + TW : w0 - 1;
      const int ol = wave * 4 + h_ch;
      const bool ok = ho < p.Hi && (unsigned)wo < (unsigned)p.Wi && obase + ol < p.Cout;
      const float v = ok ? vh : 0.f;
      const unsigned u = __builtin_bit_cast(unsigned, v);
      const float r1 = v - __builtin_bit_cast(float, u & 0xffff0000u);
      const unsigned m = __builtin_bit_cast(unsigned, r1);
      const float r2 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
      const unsigned l = __builtin_bit_cast(unsigned, r2);
      unsigned char* dst = ys + ol * CSTR + h_r * YROWB + (h_side ? 40 : 7) * 2;
      *reinterpret_cast<unsigned short*>(dst) = (unsigned short)(u >> 16);
      *reinterpret_cast<unsigned short*>(dst + SSTR) = (unsigned short)(m >> 16);
      *reinterpret_cast<unsigned short*>(dst + 2 * SSTR) = (unsigned short)(l >> 16);
    }
  };

  if (tile_lo < tile_hi) {
    int d0, h0, w0;
    decode(tile_lo, d0, h0, w0);
#pragma unroll
    for (int s = 0; s < NPIECE; ++s) prefetch_piece(s, d0, h0, w0);
#pragma unroll
    for (int s = 0; s < NPIECE; ++s) commit_piece(s, 0, d0, h0, w0);
    int nd0 = d0, nh0 = h0, nw0 = w0;
    if (tile_lo + 1 < tile_hi) decode(tile_lo + 1, nd0, nh0, nw0);
#pragma unroll
    for (int s = 0; s < NPIECE; ++s) prefetch_piece(s, nd0, nh0, nw0);
    int fd0 = nd0, fh0 = nh0, fw0 = nw0;
    __syncthreads();

    const int fr = lane & 15, fk = lane >> 4;
    // fragment addresses inside an image (bytes): A = dy[o = 16 woh + fr][row wr][8 + 8 fk ...], B = x[c = 16 j + fr][row wr + kh][8 fk ...]
    const int a_off = XBYTES + (woh * 16 + fr) * CSTR + wr * YROWB + (8 + 8 * fk) * 2;
    const int b_off = fr * CSTR + wr * XROWB + 8 * fk * 2;

    for (int tile = tile_lo; tile < tile_hi; ++tile) {
      const int buf = (tile - tile_lo) & 1;
      decode(tile + 2 < tile_hi ? tile + 2 : tile_hi - 1, fd0, fh0, fw0);
      const unsigned char* const img = lds + buf * BUF;
      // ---- A fragments: the aligned 8 elements and their two neighbouring words, three pieces each ----
      bf16x8 afr[3][3];                                     // [kw][piece]
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const unsigned char* ap = img + a_off + s * SSTR;
        const u32x4_t an = *reinterpret_cast<const u32x4_t*>(ap);
        const unsigned aprev = *reinterpret_cast<const unsigned*>(ap - 4);
        const unsigned anext = *reinterpret_cast<const unsigned*>(ap + 16);
        const u32x4_t k0 = u32x4_t{__builtin_amdgcn_alignbit(an[1], an[0], 16), __builtin_amdgcn_alignbit(an[2], an[1], 16),
                                   __builtin_amdgcn_alignbit(an[3], an[2], 16), __builtin_amdgcn_alignbit(anext, an[3], 16)};   // dy[w' + 1]
        const u32x4_t k2 = u32x4_t{__builtin_amdgcn_alignbit(an[0], aprev, 16), __builtin_amdgcn_alignbit(an[1], an[0], 16),
                                   __builtin_amdgcn_alignbit(an[2], an[1], 16), __builtin_amdgcn_alignbit(an[3], an[2], 16)};   // dy[w' - 1]
        afr[0][s] = __builtin_bit_cast(bf16x8, k0);
        afr[1][s] = __builtin_bit_cast(bf16x8, an);
        afr[2][s] = __builtin_bit_cast(bf16x8, k2);
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        bf16x8 bfr[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int s = 0; s < 3; ++s) bfr[j][s] = *reinterpret_cast<const bf16x8*>(img + b_off + j * 16 * CSTR + kh * XROWB + s * SSTR);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int g = kh * 3 + kw;
          __builtin_amdgcn_sched_barrier(0);
          f32x4_t a0 = acc[0][g], a1 = acc[1][g];
          // small terms first: lo*hi, mid*mid, hi*lo, then mid*hi, hi*mid, then hi*hi; the two in-channel halves alternate
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][2], bfr[0][0], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][2], bfr[1][0], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][1], bfr[0][1], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][1], bfr[1][1], a1, 0, 0, 0);
          // staging in the shadow of the matrix instructions, one piece per tap, unconditionally (past the end of the chunk
          // the last tile is staged again into the image nobody reads): a branch here would make the compiler drain vmcnt
          if (g < NPIECE) commit_piece(g < NPIECE ? g : 0, buf ^ 1, nd0, nh0, nw0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][0], bfr[0][2], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][0], bfr[1][2], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][1], bfr[0][0], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][1], bfr[1][0], a1, 0, 0, 0);
          if (g >= 1 && g - 1 < NPIECE) prefetch_piece(g >= 1 && g - 1 < NPIECE ? g - 1 : 0, fd0, fh0, fw0);   // registers committed one tap ago
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][0], bfr[0][1], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][0], bfr[1][1], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][0], bfr[0][0], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kw][0], bfr[1][0], a1, 0, 0, 0);
          acc[0][g] = a0; acc[1][g] = a1;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      nd0 = fd0; nh0 = fh0; nw0 = fw0;
      __syncthreads();
    }
  }

  // ---- sum of the four row waves of each out-channel half through LDS, fixed tree: (0 + 2) + (1 + 3) ----------------------
  __syncthreads();
  float* const red = reinterpret_cast<float*>(lds);         // regions of 72 x 64 floats
  auto put = [&](int region) {
    float* dst = red + region * (72 * 64) + lane;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[((j * 9 + t) * 4 + i) * 64] = acc[j][t][i];
  };
  auto add = [&](int region) {
    const float* src = red + region * (72 * 64) + lane;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][t][i] += src[((j * 9 + t) * 4 + i) * 64];
  };
  if (wr >= 2) put(woh * 2 + (wr - 2));
  __syncthreads();
  if (wr < 2) add(woh * 2 + wr);
  __syncthreads();
  if (wr == 1) put(woh * 2);
  __syncthreads();
  if (wr == 0) {
    add(woh * 2);
    // C/D layout of v_mfma_f32_16x16x32: column (in channel) = lane & 15, row (out channel) = 4 (lane >> 4) + i
    float* sp = p.slab + (long long)blockIdx.x * p.Cout * p.Cin * 9;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = cbase + 16 * j + (lane & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = obase + woh * 16 + 4 * (lane >> 4) + i;
        if (o < p.Cout && c < p.Cin) {
          float* dst = sp + ((long long)o * p.Cin + c) * 9;
#pragma unroll
          for (int t = 0; t < 9; ++t) dst[t] = acc[j][t][i];
        }
      }
    }
  }
}

// ---- v2: the same tile, split and LDS pipeline with v_mfma_f32_32x32x16_bf16 and twelve waves ----------------------------------
// An MFMA holds the SIMD's vector issue for 8 cycles whatever its shape, so the 16x16x32 form spends twice the issue slots
// of the 32x32x16 form on the same FLOPs (1728 of the 3456 cycles a tile takes at matrix rate): with the conversion VALU on
// top, v1 is bound by vector issue.  Here wave (r, kh) owns tile row r and kernel row kh: both 16-pixel K blocks of the row,
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

}  // namespace

namespace e2e {

int launch_wgrad_bf3(const WgBf3Params& p, int nchunks, int pairs, hipStream_t st) {
  static const int variant = getenv("E2E_WG_BF3") ? atoi(getenv("E2E_WG_BF3")) : 2;       // 2: v2 (default), 1: v1 (A/B)
  if (variant == 1) {
    hipLaunchKernelGGL(conv133_wgrad_bf3_kernel, dim3(nchunks, pairs), dim3(512), 0, st, p);
    return check_launch("conv133_wgrad_bf3_kernel");
  }
  hipLaunchKernelGGL(conv133_wgrad_bf3v2_kernel, dim3(nchunks, pairs), dim3(768), 0, st, p);
  return check_launch("conv133_wgrad_bf3v2_kernel");
}

}  // namespace e2e
