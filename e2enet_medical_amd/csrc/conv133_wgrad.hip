// K6b: dense weight gradient of the 1x3x3 convolution on the fp32 MFMA (gfx950).
//
//   dW[o, c, kh, kw] = sum_{n, d, h, w} dy[n, o, d, h, w] * xs[n, c, d*sd, h*sh + kh - 1, w*sw + kw - 1]
//   xs = depth-shifted virtual concat of the producers' lrelu(IN(.)) outputs (unetpp_d.py:45-59, :453-478)
//
// Why dense: the reference clips the global gradient norm over ALL parameters including DSFF-dead kernels
// (nnUNetTrainer_simple.py:573, core_channel.py:431 masks only .data), so dead-kernel gradients are part of the
// result.  Why MFMA: this is a true GEMM, M = Cout, N = Cin*9, K = voxels (~2M); v_mfma_f32_16x16x4_f32 is an exact
// fp32 fma chain (bit-compatible with a plain fp32 accumulation) and keeps the whole reduction inside the matrix
// pipe, no cross-lane reduction.
//
// One workgroup = (32 out channels) x (32 in channels) x 9 taps over a chunk of pixel tiles; wave (oh, ch) owns
// the 16 x 16 sub-block for all 9 taps (36 accumulator registers).  dy and the halo'd input tile are staged in LDS
// with channel strides == 2 (mod 32) so that the A/B fragment reads (16 channels x 4 consecutive pixels) hit 32
// distinct banks.  Per-chunk partial sums go to a slab, reduced in fixed order by a second kernel (deterministic).
#include "e2e_common.h"
#include <cstdlib>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct WgParams {
  const e2e_in_chan_t* chans;
  const float* dy;
  float* slab;
  int B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, sd;
  int tiles_x, tiles_y, tiles_d, tiles_per_n;
  long long total_tiles;
  int tiles_per_chunk;
  int cblocks;
  int cblocks_segs;   // v3: chunks per batch item
  int dbg;     // diagnostic: bit0 = skip staging, bit1 = skip the MFMA phase (timing splits only; results are garbage)
};

constexpr int pad_mod32_2(int v) {
  while (v % 32 != 2) ++v;
  return v;
}

template <int SH, int SW, int ND, int TH, int TW>
struct WCfg {
  static constexpr int TP = ND * TH * TW;
  static_assert(TW % 4 == 0, "k-steps of 4 pixels stay inside a row");
  static constexpr int IH = (TH - 1) * SH + 3, IW = (TW - 1) * SW + 3;
  static constexpr int PITCH = IW;
  static constexpr int CS = pad_mod32_2(ND * IH * PITCH);
  static constexpr int OS = pad_mod32_2(TP);
  static constexpr int LDS_FLOATS = 32 * CS + 32 * OS;
};

template <int SH, int SW, int ND, int TH, int TW>
__global__ __launch_bounds__(256) void conv133_wgrad_kernel(WgParams p) {
  using C = WCfg<SH, SW, ND, TH, TW>;
  __shared__ float lds[C::LDS_FLOATS];
  float* xs = lds;
  float* ys = lds + 32 * C::CS;

  const int chunk = blockIdx.x;
  const int cb = blockIdx.y % p.cblocks, ob = blockIdx.y / p.cblocks;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ch = wave & 1, oh = wave >> 1;
  const long long in_plane = (long long)p.Hi * p.Wi;
  const long long out_plane = (long long)p.Ho * p.Wo;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int ti = 0; ti < p.tiles_per_chunk; ++ti) {
    const long long tile = (long long)chunk * p.tiles_per_chunk + ti;
    if (tile >= p.total_tiles) break;
    const int n = (int)((unsigned)tile / (unsigned)p.tiles_per_n);
    int t = (int)tile - n * p.tiles_per_n;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int td = t / p.tiles_y;
    const int d0 = td * ND, h0 = ty * TH, w0 = tx * TW;
    const int hbase = h0 * SH - 1, wbase = w0 * SW - 1;

    // ---- stage the input tile: 32 channels x ND slices x IH x IW (transform on load, zero padded) ----
    for (int cl = wave; cl < 32; cl += 4) {          // one wave per channel -> channel descriptor is wave-uniform
      const int c = cb * 32 + cl;
      const bool cv = c < p.Cin;
      e2e_in_chan_t chd = p.chans[cv ? c : 0];
      float a = 1.f, b = 0.f, sl = 1.f;
      if (cv && chd.scale != nullptr) {
        a = chd.scale[(long long)n * chd.ab_nstride];
        b = chd.shift[(long long)n * chd.ab_nstride];
        sl = chd.slope;
      }
      gfloat_p base = (gfloat_p)(chd.ptr + (long long)n * chd.nstride);
      for (int e = lane; e < ND * C::IH * C::IW; e += 64) {
        const int nd = e / (C::IH * C::IW);
        const int rem = e - nd * (C::IH * C::IW);
        const int r = rem / C::IW, cc = rem - r * C::IW;
        const int hi = hbase + r, wi = wbase + cc;
        const int dq = d0 + nd;
        const int din = dq * p.sd - chd.dshift;
        const bool ok = cv && dq < p.Do && (unsigned)din < (unsigned)p.Di && (unsigned)hi < (unsigned)p.Hi &&
                        (unsigned)wi < (unsigned)p.Wi;
        const long long off = ok ? (long long)din * in_plane + (long long)hi * p.Wi + wi : 0;
        float v = base[off];
        v = e2e::in_act(v, a, b, sl);
        xs[cl * C::CS + nd * (C::IH * C::PITCH) + r * C::PITCH + cc] = ok ? v : 0.f;
      }
    }
    // ---- stage dy: 32 channels x TP pixels ----
    for (int idx = tid; idx < 32 * C::TP; idx += 256) {
      const int ol = idx / C::TP, pi = idx - ol * C::TP;
      const int nd = pi / (TH * TW);
      const int rem = pi - nd * (TH * TW);
      const int r = rem / TW, col = rem - r * TW;
      const int o = ob * 32 + ol;
      const int dq = d0 + nd, ho = h0 + r, wo = w0 + col;
      float v = 0.f;
      if (o < p.Cout && dq < p.Do && ho < p.Ho && wo < p.Wo)
        v = p.dy[(((long long)n * p.Cout + o) * p.Do + dq) * out_plane + (long long)ho * p.Wo + wo];
      ys[ol * C::OS + pi] = v;
    }
    __syncthreads();

    // ---- MFMA: k = 4 consecutive pixels of a row; 9 taps share the A fragment ----
    const int li = lane & 15, lk = lane >> 4;
    const float* ap = ys + (oh * 16 + li) * C::OS + lk;
    const float* bp = xs + (ch * 16 + li) * C::CS + lk * SW;
    for (int nd = 0; nd < ND; ++nd) {
      for (int r = 0; r < TH; ++r) {
        const float* apr = ap + (nd * TH + r) * TW;
        const float* bpr = bp + nd * (C::IH * C::PITCH) + (r * SH) * C::PITCH;
#pragma unroll
        for (int cq = 0; cq < TW / 4; ++cq) {
          const float a = apr[cq * 4];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
              acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bpr[kh * C::PITCH + cq * 4 * SW + kw],
                                                                      acc[kh * 3 + kw], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // D[i = o][j = c]: col = lane & 15 -> c, row = (lane >> 4) * 4 + reg -> o
  float* sp = p.slab + (long long)chunk * p.Cout * p.Cin * 9;
  const int c = cb * 32 + ch * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = ob * 32 + oh * 16 + (lane >> 4) * 4 + r;
    if (o < p.Cout && c < p.Cin) {
      float* dst = sp + ((long long)o * p.Cin + c) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) dst[t] = acc[t][r];
    }
  }
}

// ---- v2: stride-1, rows that are multiples of 4 floats ---------------------------------------------------------
// Same math and MFMA schedule as above, but the load stage is software pipelined: the global loads of tile t+1
// (aligned float4 groups of the halo'd input rows and of the dy rows) are issued into registers right after the
// barrier that publishes tile t, so they are in flight during the whole MFMA phase of tile t (7-8 us), and are
// committed to LDS (normalise-on-load applied) when that phase is over.  One workgroup covers NCB blocks of 32 input
// channels so that the dy tile is staged once for all of them (NCB * 4 waves).  Each wave stages 8 input channels
// and 32 / (4 NCB) dy channels; the channel descriptors are wave-uniform and live in scalar registers for as long
// as the batch item does not change.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const f32x4_t __attribute__((address_space(1)))* gf4_p;

template <int ND, int TH, int TW, int NCB>
struct W2Cfg {
  static constexpr int TP = ND * TH * TW;
  static_assert(TP == 256, "one dy channel = 64 float4 groups = one wave-instruction");
  static constexpr int IH = TH + 2, IW = TW + 2;
  static constexpr int NQ = TW / 4 + 2;                       // float4 groups per input row (starts 4 left of the tile)
  // LDS rows hold the float4 groups whole (tile column -1 sits at row index 3): every group is stored with two
  // unconditional ds_write_b64, no per-element predicates; fragment reads are b32, so the +3 offset is free
  // (NCB == 1 keeps the compact pitch: 80 KB of LDS -> two workgroups per CU, which is worth more there)
  static constexpr bool PADDED = NCB == 2;
  static constexpr int PITCH = PADDED ? 4 * NQ : IW;
  static constexpr int COL0 = PADDED ? 3 : 0;                 // row index of tile column -1
  static constexpr int CS = pad_mod32_2(ND * IH * PITCH);
  static constexpr int OS = pad_mod32_2(TP);
  static constexpr int NT = 256 * NCB;
  static constexpr int GPC = ND * IH * NQ;                    // groups per input channel
  static constexpr int ITX = (GPC + 63) / 64;                 // wave iterations per input channel
  static constexpr int XCW = 8;                               // input channels staged per wave
  static constexpr int YCW = 32 / (4 * NCB);                  // dy channels staged per wave
  static constexpr int LDS_FLOATS = NCB * 32 * CS + 32 * OS;
};

template <int ND, int TH, int TW, int NCB>
__global__ __launch_bounds__(256 * NCB) void conv133_wgrad_v2_kernel(WgParams p) {
  using C = W2Cfg<ND, TH, TW, NCB>;
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  float* xs = lds;
  float* ys = lds + NCB * 32 * C::CS;

  const int chunk = blockIdx.x;
  const int cgroups = p.cblocks;                      // groups of NCB * 32 input channels
  const int cg = blockIdx.y % cgroups, ob = blockIdx.y / cgroups;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cbl = wave >> 2, ch = wave & 1, oh = (wave >> 1) & 1;
  const long long in_plane = (long long)p.Hi * p.Wi;
  const long long out_plane = (long long)p.Ho * p.Wo;
  const int cbase = cg * NCB * 32;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long tile_lo = (long long)chunk * p.tiles_per_chunk;
  long long tile_hi = tile_lo + p.tiles_per_chunk;
  if (tile_hi > p.total_tiles) tile_hi = p.total_tiles;

  auto decode = [&](long long tile, int& n, int& d0, int& h0, int& w0) {
    n = (int)((unsigned)tile / (unsigned)p.tiles_per_n);      // (fewer than 2^31 tiles: 32-bit division)
    int t = (int)tile - n * p.tiles_per_n;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    d0 = (t / p.tiles_y) * ND;
    h0 = ty * TH;
    w0 = tx * TW;
  };

  // ---- wave-uniform descriptors of this wave's 8 input channels (scalar registers) ----
  gfloat_p xbase[C::XCW];
  float xa[C::XCW], xb[C::XCW], xsl[C::XCW];
  int xdsh[C::XCW];
  bool xval[C::XCW];
  auto load_desc = [&](int n) {
#pragma unroll
    for (int k = 0; k < C::XCW; ++k) {
      const int c = cbase + wave * C::XCW + k;
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
  };

  // ---- per-lane geometry of the float4 groups (independent of the channel) ----
  int g_lrow[C::ITX], g_q[C::ITX], g_nd[C::ITX], g_r[C::ITX];
#pragma unroll
  for (int it = 0; it < C::ITX; ++it) {
    int g = lane + 64 * it;
    if (g >= C::GPC) g = C::GPC - 1;
    const int nd = g / (C::IH * C::NQ);
    const int rem = g - nd * (C::IH * C::NQ);
    const int r = rem / C::NQ;
    g_nd[it] = nd; g_r[it] = r; g_q[it] = rem - r * C::NQ; g_lrow[it] = nd * C::IH + r;
  }
  // dy: lane -> float4 group of the 256-pixel tile
  const int y_pi = lane * 4;
  const int y_nd = y_pi / (TH * TW);
  const int y_rem = y_pi - y_nd * (TH * TW);
  const int y_r = y_rem / TW, y_col = y_rem - y_r * TW;

  f32x4_t vx[C::XCW][C::ITX], vy[C::YCW];
  auto prefetch = [&](int n, int d0, int h0, int w0) {
#pragma unroll
    for (int it = 0; it < C::ITX; ++it) {
      const int hi = h0 - 1 + g_r[it], gc = w0 - 4 + 4 * g_q[it];
      const int dq = d0 + g_nd[it];
      const bool lane_ok = lane + 64 * it < C::GPC && dq < p.Do && (unsigned)hi < (unsigned)p.Hi && gc >= 0 && gc + 3 < p.Wi;
      const long long lane_off = (long long)hi * p.Wi + gc;
#pragma unroll
      for (int k = 0; k < C::XCW; ++k) {
        const int din = dq * p.sd - xdsh[k];
        const bool ok = lane_ok && xval[k] && (unsigned)din < (unsigned)p.Di && !(p.dbg & 4);
        vx[k][it] = *reinterpret_cast<gf4_p>(xbase[k] + (ok ? (long long)din * in_plane + lane_off : 0));
      }
    }
    const int dq = d0 + y_nd, ho = h0 + y_r, wo = w0 + y_col;
    const bool lane_ok = dq < p.Do && ho < p.Ho && wo + 3 < p.Wo;
#pragma unroll
    for (int k = 0; k < C::YCW; ++k) {
      const int o = ob * 32 + wave * C::YCW + k;
      const bool ok = lane_ok && o < p.Cout && !(p.dbg & 4);
      const long long off = ok ? (((long long)n * p.Cout + o) * p.Do + dq) * out_plane + (long long)ho * p.Wo + wo : 0;
      vy[k] = *reinterpret_cast<gf4_p>((gfloat_p)p.dy + off);
    }
  };
  auto commit = [&](int d0, int h0, int w0) {
#pragma unroll
    for (int it = 0; it < C::ITX; ++it) {
      if (lane + 64 * it >= C::GPC) continue;
      const int hi = h0 - 1 + g_r[it], gc = w0 - 4 + 4 * g_q[it];
      const int dq = d0 + g_nd[it];
      const bool lane_ok = dq < p.Do && (unsigned)hi < (unsigned)p.Hi && gc >= 0 && gc + 3 < p.Wi;
#pragma unroll
      for (int k = 0; k < C::XCW; ++k) {
        const int din = dq * p.sd - xdsh[k];
        const bool ok = lane_ok && xval[k] && (unsigned)din < (unsigned)p.Di;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = e2e::in_act(vx[k][it][j], xa[k], xb[k], xsl[k]);
          v[j] = ok ? t : 0.f;
        }
        float* row = xs + (wave * C::XCW + k) * C::CS + g_lrow[it] * C::PITCH;
        if (C::PADDED) {
          float2* dst = reinterpret_cast<float2*>(row + 4 * g_q[it]);
          dst[0] = make_float2(v[0], v[1]);
          dst[1] = make_float2(v[2], v[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int lc = 4 * g_q[it] + j - 3;
            if ((unsigned)lc < (unsigned)C::IW) row[lc] = v[j];
          }
        }
      }
    }
    const int dq = d0 + y_nd, ho = h0 + y_r, wo = w0 + y_col;
    const bool lane_ok = dq < p.Do && ho < p.Ho && wo + 3 < p.Wo;
#pragma unroll
    for (int k = 0; k < C::YCW; ++k) {
      const int ol = wave * C::YCW + k;
      const bool ok = lane_ok && ob * 32 + ol < p.Cout;
      float2* dst = reinterpret_cast<float2*>(ys + ol * C::OS + y_pi);       // OS is even: 8-byte aligned
      dst[0] = ok ? make_float2(vy[k][0], vy[k][1]) : make_float2(0.f, 0.f);
      dst[1] = ok ? make_float2(vy[k][2], vy[k][3]) : make_float2(0.f, 0.f);
    }
  };

  if (tile_lo < tile_hi) {
    int n, d0, h0, w0;
    decode(tile_lo, n, d0, h0, w0);
    load_desc(n);
    if (!(p.dbg & 1)) prefetch(n, d0, h0, w0);
    for (long long tile = tile_lo; tile < tile_hi; ++tile) {
      if (!(p.dbg & 1)) commit(d0, h0, w0);
      int nn = n, nd0 = d0, nh0 = h0, nw0 = w0;
      const bool more = tile + 1 < tile_hi;
      if (more) {
        decode(tile + 1, nn, nd0, nh0, nw0);
        if (nn != n) load_desc(nn);                       // rare: the chunk crosses a batch item
      }
      __syncthreads();
      if (more && !(p.dbg & 1)) prefetch(nn, nd0, nh0, nw0);              // in flight during the MFMA phase below

      // MFMA phase.  The A/B fragments of k-step s+1 are read from LDS before the 9 MFMAs of k-step s are issued
      // (explicit double registers + sched_barrier), so the matrix pipe never waits on an LDS round trip.
      const int li = lane & 15, lk = lane >> 4;
      const float* ap = ys + (oh * 16 + li) * C::OS + lk;
      const float* bp = xs + (cbl * 32 + ch * 16 + li) * C::CS + lk + C::COL0;
      float a_cur = ap[0];
      float b_cur[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) b_cur[kh * 3 + kw] = bp[kh * C::PITCH + kw];
      for (int row = 0; row < ((p.dbg & 2) ? 0 : ND * TH); ++row) {
        const int nd = row / TH, r = row - nd * TH;
        const float* apr = ap + row * TW;
        const float* bpr = bp + nd * (C::IH * C::PITCH) + r * C::PITCH;
        // first k-step of the next row (a harmless read past the tile after the last row: still inside the LDS array)
        const int row1 = row + 1;
        const int nd1 = row1 / TH, r1 = row1 - nd1 * TH;
        const float* apn = ap + row1 * TW;
        const float* bpn = bp + nd1 * (C::IH * C::PITCH) + r1 * C::PITCH;
#pragma unroll
        for (int cq = 0; cq < TW / 4; ++cq) {
          float a_nxt, b_nxt[9];
          if (cq + 1 < TW / 4) {
            a_nxt = apr[(cq + 1) * 4];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) b_nxt[kh * 3 + kw] = bpr[kh * C::PITCH + (cq + 1) * 4 + kw];
          } else {
            a_nxt = apn[0];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) b_nxt[kh * 3 + kw] = bpn[kh * C::PITCH + kw];
          }
#ifndef E2E_WG_NOSB
          __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
#ifndef E2E_WG_NOSB
          __builtin_amdgcn_sched_barrier(0);
#endif
          a_cur = a_nxt;
#pragma unroll
          for (int t = 0; t < 9; ++t) b_cur[t] = b_nxt[t];
        }
      }
      __syncthreads();
      n = nn; d0 = nd0; h0 = nh0; w0 = nw0;
    }
  }

  float* sp = p.slab + (long long)chunk * p.Cout * p.Cin * 9;
  const int c = cbase + cbl * 32 + ch * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = ob * 32 + oh * 16 + (lane >> 4) * 4 + r;
    if (o < p.Cout && c < p.Cin) {
      float* dst = sp + ((long long)o * p.Cin + c) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) dst[t] = acc[t][r];
    }
  }
}

// ---- v3: large stride-1 planes (rows >= 32, multiples of 4 floats): double-buffered LDS ------------------------------
// v2 alternates a staging phase (commit the prefetched registers to LDS) and an MFMA phase separated by two barriers,
// and with one 8-wave workgroup per CU nothing else runs while it stages: the matrix pipe idles ~40 % of the time.
// v3 keeps two LDS images (tile 4 x 32, 77 KB each for 64 input channels): while the MFMAs consume image t & 1, the
// registers holding tile t+1 are committed to the other image piece by piece between the k-steps, piece s of tile t+2
// is requested one k-step after piece s of tile t+1 left its registers (straight-line code, no branch: see the loop),
// and one barrier per tile publishes the new image.  A chunk never crosses a batch item, so the channel descriptors
// are loaded once per workgroup.  Phase split on 64->32 @128^3 (E2E_WG_DBG builds of round 2): MFMAs alone 1.07 ms
// (144 TFLOP/s), + commits 1.19, + the ten loads per lane in one block before the barrier 1.37; with the loads spread
// over the k-steps 1.29.
template <int NCB, int NOB, int NKB = 1>
struct W3Cfg {
  static_assert(NCB * NOB * NKB == 2, "8 waves: 2 x 2 sub-blocks of 16 x 16 for two 32 x 32 blocks, or for the two row halves of one");
  static constexpr int TH = 4, TW = 32, TP = TH * TW;
  static constexpr int IH = TH + 2;
  static constexpr int NQ = TW / 4 + 2;                       // float4 groups per input row (starts 4 left of the tile)
  static constexpr int PITCH = 4 * NQ;                        // groups stored whole: tile column -1 sits at row index 3
  static constexpr int COL0 = 3;
  static constexpr int CS = IH * PITCH + 2;                   // 242 = 2 x odd: 16 channels x 2 k-lanes hit 32 distinct banks
  static constexpr int OS = TP + 2;                           // 130 = 2 x odd
  static_assert((CS / 2) % 2 == 1 && (OS / 2) % 2 == 1, "channel strides must be 2 x odd");
  static constexpr int NT = 512;
  static constexpr int GPC = IH * NQ;                         // 60 groups per input channel: one wave-instruction
  static_assert(GPC <= 64, "one float4 group per lane and channel");
  static constexpr int XCW = NCB * 32 / 8;                    // input channels staged per wave
  static constexpr int YIT = NOB * 32 * (TP / 4) / NT;        // dy float4 groups per thread
  static constexpr int YCW = NOB * 32 / 8;                    // dy channels staged per wave
  static constexpr int XBUF = NCB * 32 * CS, YBUF = NOB * 32 * OS;
  static constexpr int LDS_FLOATS = 2 * (XBUF + YBUF);
  static constexpr int PIECES = XCW + YIT;
  static constexpr int ROWS = TH / NKB;                       // tile rows per wave group
  static_assert(PIECES <= ROWS * TW / 4, "one commit piece per k-step");
};

template <int NCB, int NOB, int NKB>
__global__ __launch_bounds__(512) void conv133_wgrad_v3_kernel(WgParams p) {
  using C = W3Cfg<NCB, NOB, NKB>;
  __shared__ __attribute__((aligned(16))) float lds3[C::LDS_FLOATS + 8];     // + slack for the read-ahead after the last k-step
  float* const xs0 = lds3;
  float* const ys0 = lds3 + 2 * C::XBUF;

  // chunk = (batch item, run of tiles inside it)
  const int segs = p.cblocks_segs;
  const int n = blockIdx.x / segs, seg = blockIdx.x - n * segs;
  const int cgroups = p.cblocks;
  const int cg = blockIdx.y % cgroups, ob = blockIdx.y / cgroups;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cbl = NCB == 2 ? wave >> 2 : 0, obl = NOB == 2 ? wave >> 2 : 0, ch = wave & 1, oh = (wave >> 1) & 1;
  const int kq = NKB == 2 ? wave >> 2 : 0;              // Cin <= 32 and Cout <= 32: the two wave groups split the tile rows
  const int obase = ob * NOB * 32;
  const long long in_plane = (long long)p.Hi * p.Wi;
  const long long out_plane = (long long)p.Ho * p.Wo;
  const int cbase = cg * NCB * 32;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tile_lo = seg * p.tiles_per_chunk;
  int tile_hi = tile_lo + p.tiles_per_chunk;
  if (tile_hi > p.tiles_per_n) tile_hi = p.tiles_per_n;

  auto decode = [&](int tile, int& d0, int& h0, int& w0) {
    const int tx = tile % p.tiles_x;
    const int t = tile / p.tiles_x;
    const int ty = t % p.tiles_y;
    d0 = t / p.tiles_y;
    h0 = ty * C::TH;
    w0 = tx * C::TW;
  };

  // ---- wave-uniform descriptors of this wave's 8 input channels (scalar registers, fixed for the workgroup) ----
  gfloat_p xbase[C::XCW];
  float xa[C::XCW], xb[C::XCW], xsl[C::XCW];
  int xdsh[C::XCW];
  bool xval[C::XCW];
#pragma unroll
  for (int k = 0; k < C::XCW; ++k) {
    const int c = cbase + wave * C::XCW + k;
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

  // ---- per-lane geometry ----
  const int g_r = (lane < C::GPC ? lane : C::GPC - 1) / C::NQ;
  const int g_q = (lane < C::GPC ? lane : C::GPC - 1) - g_r * C::NQ;
  // dy: a wave stages YCW channels x 32 groups = YIT rounds of 64 lanes: lane -> (channel lane / 32 + 2 it, group lane % 32)
  const int y_grp = lane & 31, y_kl = lane >> 5;
  const int y_r = y_grp >> 3, y_col = (y_grp & 7) * 4;

  f32x4_t vx[C::XCW], vy[C::YIT];
  // request piece s of a tile (s < XCW: input channel s of this wave, else dy round s - XCW) into its registers.  Inside the
  // MFMA loop the pieces of tile t+2 are requested one per k-step, each right after the registers' previous content (tile
  // t+1) went to LDS: ten back-to-back 16-byte loads per lane from all eight waves (80 KB) keep the texture addresser busy
  // for > 1 k cycles with every wave stuck in issue order behind them -- measured 13 % of the kernel when they sat in
  // one block in front of the barrier.
  auto prefetch_piece = [&](int s, int d0, int h0, int w0) {
    if (s < C::XCW) {
      const int k = s;
      const int hi = h0 - 1 + g_r, gc = w0 - 4 + 4 * g_q;
      const bool lane_ok = (unsigned)hi < (unsigned)p.Hi && gc >= 0 && gc + 3 < p.Wi;
      const long long lane_off = (long long)hi * p.Wi + gc;
      const int din = d0 * p.sd - xdsh[k];
      const bool ok = lane_ok && xval[k] && (unsigned)din < (unsigned)p.Di;
      vx[k] = *reinterpret_cast<gf4_p>(xbase[k] + (ok ? (long long)din * in_plane + lane_off : 0));
    } else {
      const int it = s - C::XCW;
      const int ho = h0 + y_r, wo = w0 + y_col;
      const bool yok = ho < p.Ho && wo + 3 < p.Wo;
      const int o = obase + wave * C::YCW + y_kl + 2 * it;
      const bool ok = yok && o < p.Cout;
      const long long off = ok ? (((long long)n * p.Cout + o) * p.Do + d0) * out_plane + (long long)ho * p.Wo + wo : 0;
      vy[it] = *reinterpret_cast<gf4_p>((gfloat_p)p.dy + off);
    }
  };
  auto prefetch = [&](int d0, int h0, int w0) {
#pragma unroll
    for (int s = 0; s < C::PIECES; ++s) prefetch_piece(s, d0, h0, w0);
  };
  // commit piece s of the registers (tile geometry d0, h0, w0) into image `buf`: s < XCW -> input channel s, else dy round
  auto commit_piece = [&](int s, int buf, int d0, int h0, int w0) {
    if (s < C::XCW) {
      // (lanes beyond the last group hold a clamped copy of it: same address, same value -- no divergent branch here, so
      //  that the in-loop staging below stays straight-line code and its loads get exact vmcnt distances)
      const int k = s;
      const int hi = h0 - 1 + g_r, gc = w0 - 4 + 4 * g_q;
      const int din = d0 * p.sd - xdsh[k];
      const bool ok = (unsigned)hi < (unsigned)p.Hi && gc >= 0 && gc + 3 < p.Wi && xval[k] && (unsigned)din < (unsigned)p.Di;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = e2e::in_act(vx[k][j], xa[k], xb[k], xsl[k]);
        v[j] = ok ? t : 0.f;
      }
      float2* dst = reinterpret_cast<float2*>(xs0 + buf * C::XBUF + (wave * C::XCW + k) * C::CS + g_r * C::PITCH + 4 * g_q);
      dst[0] = make_float2(v[0], v[1]);
      dst[1] = make_float2(v[2], v[3]);
    } else {
      const int it = s - C::XCW;
      const int ho = h0 + y_r, wo = w0 + y_col;
      const int ol = wave * C::YCW + y_kl + 2 * it;
      const bool ok = ho < p.Ho && wo + 3 < p.Wo && obase + ol < p.Cout;
      float2* dst = reinterpret_cast<float2*>(ys0 + buf * C::YBUF + ol * C::OS + y_grp * 4);
      dst[0] = ok ? make_float2(vy[it][0], vy[it][1]) : make_float2(0.f, 0.f);
      dst[1] = ok ? make_float2(vy[it][2], vy[it][3]) : make_float2(0.f, 0.f);
    }
  };

  if (tile_lo < tile_hi) {
    int d0, h0, w0;
    decode(tile_lo, d0, h0, w0);
    prefetch(d0, h0, w0);
#pragma unroll
    for (int s = 0; s < C::PIECES; ++s) commit_piece(s, 0, d0, h0, w0);
    int nd0 = d0, nh0 = h0, nw0 = w0;
    if (tile_lo + 1 < tile_hi) {
      decode(tile_lo + 1, nd0, nh0, nw0);
      prefetch(nd0, nh0, nw0);
    }
    int fd0 = nd0, fh0 = nh0, fw0 = nw0;               // tile + 2: requested piece by piece inside the loop
    __syncthreads();

    const int li = lane & 15, lk = lane >> 4;
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
      const int buf = (tile - tile_lo) & 1;
      decode(tile + 2 < tile_hi ? tile + 2 : tile_hi - 1, fd0, fh0, fw0);   // registers hold tile + 1 -> image buf ^ 1 during this phase
      const float* ap = ys0 + buf * C::YBUF + (obl * 32 + oh * 16 + li) * C::OS + lk + kq * C::ROWS * C::TW;
      const float* bp = xs0 + buf * C::XBUF + (cbl * 32 + ch * 16 + li) * C::CS + lk + C::COL0 + kq * C::ROWS * C::PITCH;
      float a_cur = ap[0];
      float b_cur[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) b_cur[kh * 3 + kw] = bp[kh * C::PITCH + kw];
#pragma unroll
      for (int row = 0; row < C::ROWS; ++row) {
        const float* apr = ap + row * C::TW;
        const float* bpr = bp + row * C::PITCH;
#pragma unroll
        for (int cq = 0; cq < C::TW / 4; ++cq) {
          float a_nxt, b_nxt[9];
          // fragments of the next k-step (a harmless read past the tile after the last one: still inside the LDS image)
          const int nrow = cq + 1 < C::TW / 4 ? row : row + 1, ncq = cq + 1 < C::TW / 4 ? cq + 1 : 0;
          a_nxt = ap[nrow * C::TW + ncq * 4];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) b_nxt[kh * 3 + kw] = bp[(nrow + kh) * C::PITCH + ncq * 4 + kw];
          (void)apr; (void)bpr;
          const int s = row * (C::TW / 4) + cq;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
            // staging in the shadow of the MFMAs, one piece per k-step, unconditionally (past the end of the chunk the
            // last tile is simply staged again into the image nobody reads): any branch here makes the compiler wait
            // for vmcnt(0) at every piece, i.e. for the load it issued one k-step earlier
            if (t == 0 && s < C::PIECES) commit_piece(s < C::PIECES ? s : 0, buf ^ 1, nd0, nh0, nw0);
            if (t == 4 && s >= 1 && s - 1 < C::PIECES) prefetch_piece(s >= 1 && s - 1 < C::PIECES ? s - 1 : 0, fd0, fh0, fw0);   // registers committed one k-step ago
          }
          __builtin_amdgcn_sched_barrier(0);
          a_cur = a_nxt;
#pragma unroll
          for (int t = 0; t < 9; ++t) b_cur[t] = b_nxt[t];
        }
      }
      nd0 = fd0; nh0 = fh0; nw0 = fw0;
      if (!(p.dbg & 32)) __syncthreads();
    }
  }

  float* sp = p.slab + ((long long)blockIdx.x * NKB + kq) * p.Cout * p.Cin * 9;
  const int c = cbase + cbl * 32 + ch * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = obase + obl * 32 + oh * 16 + (lane >> 4) * 4 + r;
    if (o < p.Cout && c < p.Cin) {
      float* dst = sp + ((long long)o * p.Cin + c) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) dst[t] = acc[t][r];
    }
  }
}

// ---- network input layer: Cin <= 4 (BraTS: 4 modalities, CT: 1), stride 1, large planes ------------------------------
// A 32 x 32 channel block would spend 7/8 of its MFMAs on padding.  Here the 16 columns of a B fragment are
// (4 channels) x (4 taps): the 9 taps take 3 MFMAs per k-step instead of 9, the four waves split the 32 output
// channels (2) and the rows of the 8 x 32 tile (2), and only the real input channels are staged.  The two row halves
// write separate slabs (chunk 2 b + half), summed by the slab reduction like any other chunk.
struct WSCfg {
  static constexpr int TH = 8, TW = 32, TP = TH * TW;
  static constexpr int IH = TH + 2, NQ = TW / 4 + 2, PITCH = 4 * NQ, COL0 = 3;
  static constexpr int CS = IH * PITCH + 8;                  // 408: channel c starts 8 c banks further
  static constexpr int OS = TP + 2;                          // 258 = 2 x odd
  static constexpr int GPC = IH * NQ;                        // 100 float4 groups per channel
  static constexpr int LDS_FLOATS = 4 * CS + 32 * OS;
};

__global__ __launch_bounds__(256) void conv133_wgrad_smallc_kernel(WgParams p) {
  using C = WSCfg;
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS + 8];
  float* xs = lds;
  float* ys = lds + 4 * C::CS;

  const int segs = p.cblocks_segs;
  const int n = blockIdx.x / segs, seg = blockIdx.x - n * segs;
  const int ob = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int oh = wave & 1, kq = wave >> 1;
  const long long in_plane = (long long)p.Hi * p.Wi;
  const long long out_plane = (long long)p.Ho * p.Wo;

  f32x4 acc[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tile_lo = seg * p.tiles_per_chunk;
  int tile_hi = tile_lo + p.tiles_per_chunk;
  if (tile_hi > p.tiles_per_n) tile_hi = p.tiles_per_n;
  auto decode = [&](int tile, int& d0, int& h0, int& w0) {
    const int tx = tile % p.tiles_x;
    const int t = tile / p.tiles_x;
    const int ty = t % p.tiles_y;
    d0 = t / p.tiles_y;
    h0 = ty * C::TH;
    w0 = tx * C::TW;
  };

  // wave w stages input channel w (if it exists): descriptor in scalar registers
  const bool xval = wave < p.Cin;
  const e2e_in_chan_t chd = p.chans[xval ? wave : 0];
  const gfloat_p xbase = (gfloat_p)(chd.ptr + (long long)n * chd.nstride);
  float xa = 1.f, xb = 0.f, xsl = 1.f;
  if (xval && chd.scale != nullptr) {
    xa = chd.scale[(long long)n * chd.ab_nstride];
    xb = chd.shift[(long long)n * chd.ab_nstride];
    xsl = chd.slope;
  }
  const int xdsh = chd.dshift;

  // dy: 32 channels x 64 float4 groups = 8 per thread: thread -> (channel tid / 64 + 4 it, group lane)
  const int y_pi = lane * 4;
  const int y_r = y_pi / C::TW, y_col = y_pi - y_r * C::TW;
  f32x4_t vx[2], vy[8];
  auto prefetch = [&](int d0, int h0, int w0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int g = lane + 64 * it;
      if (g >= C::GPC) g = C::GPC - 1;
      const int r = g / C::NQ, q = g - r * C::NQ;
      const int hi = h0 - 1 + r, gc = w0 - 4 + 4 * q;
      const int din = d0 * p.sd - xdsh;
      const bool ok = xval && (unsigned)hi < (unsigned)p.Hi && gc >= 0 && gc + 3 < p.Wi && (unsigned)din < (unsigned)p.Di;
      vx[it] = *reinterpret_cast<gf4_p>(xbase + (ok ? (long long)din * in_plane + (long long)hi * p.Wi + gc : 0));
    }
    const int ho = h0 + y_r, wo = w0 + y_col;
    const bool yok = ho < p.Ho && wo + 3 < p.Wo;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int o = ob * 32 + wave + 4 * it;
      const bool ok = yok && o < p.Cout;
      const long long off = ok ? (((long long)n * p.Cout + o) * p.Do + d0) * out_plane + (long long)ho * p.Wo + wo : 0;
      vy[it] = *reinterpret_cast<gf4_p>((gfloat_p)p.dy + off);
    }
  };
  auto commit = [&](int d0, int h0, int w0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int g = lane + 64 * it;
      if (g >= C::GPC) continue;
      const int r = g / C::NQ, q = g - r * C::NQ;
      const int hi = h0 - 1 + r, gc = w0 - 4 + 4 * q;
      const int din = d0 * p.sd - xdsh;
      const bool ok = xval && (unsigned)hi < (unsigned)p.Hi && gc >= 0 && gc + 3 < p.Wi && (unsigned)din < (unsigned)p.Di;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = e2e::in_act(vx[it][j], xa, xb, xsl);
        v[j] = ok ? t : 0.f;
      }
      float2* dst = reinterpret_cast<float2*>(xs + wave * C::CS + r * C::PITCH + 4 * q);
      dst[0] = make_float2(v[0], v[1]);
      dst[1] = make_float2(v[2], v[3]);
    }
    const int ho = h0 + y_r, wo = w0 + y_col;
    const bool yok = ho < p.Ho && wo + 3 < p.Wo;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int ol = wave + 4 * it;
      const bool ok = yok && ob * 32 + ol < p.Cout;
      float2* dst = reinterpret_cast<float2*>(ys + ol * C::OS + y_pi);
      dst[0] = ok ? make_float2(vy[it][0], vy[it][1]) : make_float2(0.f, 0.f);
      dst[1] = ok ? make_float2(vy[it][2], vy[it][3]) : make_float2(0.f, 0.f);
    }
  };

  // B-fragment column li = (channel li & 3, tap 4 m + (li >> 2)); taps >= 9 compute garbage columns that are never stored
  const int li = lane & 15, lk = lane >> 4;
  int boff[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    int tap = 4 * m + (li >> 2);
    if (tap > 8) tap = 8;
    boff[m] = (li & 3) * C::CS + (tap / 3) * C::PITCH + (tap % 3) + lk + C::COL0;
  }

  if (tile_lo < tile_hi) {
    int d0, h0, w0;
    decode(tile_lo, d0, h0, w0);
    prefetch(d0, h0, w0);
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
      commit(d0, h0, w0);
      __syncthreads();
      if (tile + 1 < tile_hi) {
        decode(tile + 1, d0, h0, w0);
        prefetch(d0, h0, w0);
      }
      const float* ap = ys + (oh * 16 + li) * C::OS + lk + kq * (C::TH / 2) * C::TW;
      const float* bp = xs + kq * (C::TH / 2) * C::PITCH;
      float a_cur = ap[0], b_cur[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) b_cur[m] = bp[boff[m]];
#pragma unroll
      for (int row = 0; row < C::TH / 2; ++row) {
#pragma unroll
        for (int cq = 0; cq < C::TW / 4; ++cq) {
          const int nrow = cq + 1 < C::TW / 4 ? row : row + 1, ncq = cq + 1 < C::TW / 4 ? cq + 1 : 0;
          const float a_nxt = ap[nrow * C::TW + ncq * 4];           // (read-ahead past the last k-step stays inside LDS)
          float b_nxt[3];
#pragma unroll
          for (int m = 0; m < 3; ++m) b_nxt[m] = bp[nrow * C::PITCH + ncq * 4 + boff[m]];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 3; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b_cur[m], acc[m], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          a_cur = a_nxt;
#pragma unroll
          for (int m = 0; m < 3; ++m) b_cur[m] = b_nxt[m];
        }
      }
      __syncthreads();
    }
  }

  // D[i = o][j = column]: column = lane & 15 -> (c, tap), row = (lane >> 4) * 4 + reg -> o
  float* sp = p.slab + ((long long)blockIdx.x * 2 + kq) * p.Cout * p.Cin * 9;
  const int c = li & 3;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int tap = 4 * m + (li >> 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = ob * 32 + oh * 16 + (lane >> 4) * 4 + r;
      if (o < p.Cout && c < p.Cin && tap < 9) sp[((long long)o * p.Cin + c) * 9 + tap] = acc[m][r];
    }
  }
}

// ---- stride (2,2) in the plane ("convolutional pooling" convs), rows that are multiples of 4 floats ------------------
// Same pipeline as v2.  Output tile 4 x 16; the 9 x 33 input patch it needs is staged per channel as rows of
// [20 even columns | 20 odd columns] so that the B fragments (4 consecutive output pixels = input columns 2 apart)
// are unit-stride LDS reads: tap kw = 0 -> odd[wo + 1], kw = 1 -> even[wo + 2], kw = 2 -> odd[wo + 2].
template <int DUMMY>
struct WS2Cfg {
  static constexpr int TH = 4, TW = 16, TP = 64;
  static constexpr int IH = 2 * TH + 1;                       // 9 input rows
  static constexpr int NQ = (2 * TW + 8) / 4;                 // 10 float4 groups per input row, from column 2*w0 - 4
  static constexpr int HALF = 2 * NQ;                         // 20 even (odd) columns per row
  static constexpr int PITCH = 2 * HALF;
  static constexpr int CS = pad_mod32_2(IH * PITCH);
  static constexpr int OS = pad_mod32_2(TP);
  static constexpr int GPC = IH * NQ;
  static constexpr int ITX = (GPC + 63) / 64;
  static constexpr int XCW = 8, YCW = 8;
  static constexpr int LDS_FLOATS = 32 * CS + 32 * OS;
};

__global__ __launch_bounds__(256) void conv133_wgrad_s2_kernel(WgParams p) {
  using C = WS2Cfg<0>;
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  float* xs = lds;
  float* ys = lds + 32 * C::CS;

  const int chunk = blockIdx.x;
  const int cb = blockIdx.y % p.cblocks, ob = blockIdx.y / p.cblocks;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = wave & 1, oh = wave >> 1;
  const long long in_plane = (long long)p.Hi * p.Wi;
  const long long out_plane = (long long)p.Ho * p.Wo;
  const int cbase = cb * 32;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long tile_lo = (long long)chunk * p.tiles_per_chunk;
  long long tile_hi = tile_lo + p.tiles_per_chunk;
  if (tile_hi > p.total_tiles) tile_hi = p.total_tiles;

  auto decode = [&](long long tile, int& n, int& d0, int& h0, int& w0) {
    n = (int)((unsigned)tile / (unsigned)p.tiles_per_n);      // (fewer than 2^31 tiles: 32-bit division)
    int t = (int)tile - n * p.tiles_per_n;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    d0 = t / p.tiles_y;
    h0 = ty * C::TH;
    w0 = tx * C::TW;
  };

  gfloat_p xbase[C::XCW];
  float xa[C::XCW], xb[C::XCW], xsl[C::XCW];
  int xdsh[C::XCW];
  bool xval[C::XCW];
  auto load_desc = [&](int n) {
#pragma unroll
    for (int k = 0; k < C::XCW; ++k) {
      const int c = cbase + wave * C::XCW + k;
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
  };

  int g_r[C::ITX], g_q[C::ITX];
#pragma unroll
  for (int it = 0; it < C::ITX; ++it) {
    int g = lane + 64 * it;
    if (g >= C::GPC) g = C::GPC - 1;
    g_r[it] = g / C::NQ;
    g_q[it] = g - g_r[it] * C::NQ;
  }
  // dy: 8 channels x 16 float4 groups per wave = 2 iterations; lane -> (channel lane / 16 + 4 it, group lane % 16)
  const int y_grp = lane & 15, y_kl = lane >> 4;
  const int y_pi = y_grp * 4;
  const int y_r = y_pi / C::TW, y_col = y_pi - y_r * C::TW;

  f32x4_t vx[C::XCW][C::ITX], vy[2];
  auto prefetch = [&](int n, int d0, int h0, int w0) {
#pragma unroll
    for (int it = 0; it < C::ITX; ++it) {
      const int hi = 2 * h0 - 1 + g_r[it], gc = 2 * w0 - 4 + 4 * g_q[it];
      const bool lane_ok = lane + 64 * it < C::GPC && (unsigned)hi < (unsigned)p.Hi && gc >= 0 && gc + 3 < p.Wi;
      const long long lane_off = (long long)hi * p.Wi + gc;
#pragma unroll
      for (int k = 0; k < C::XCW; ++k) {
        const int din = d0 * p.sd - xdsh[k];
        const bool ok = lane_ok && xval[k] && (unsigned)din < (unsigned)p.Di;
        vx[k][it] = *reinterpret_cast<gf4_p>(xbase[k] + (ok ? (long long)din * in_plane + lane_off : 0));
      }
    }
    const int ho = h0 + y_r, wo = w0 + y_col;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int o = ob * 32 + wave * C::YCW + y_kl + 4 * it;
      const bool ok = o < p.Cout && ho < p.Ho && wo + 3 < p.Wo;
      const long long off = ok ? (((long long)n * p.Cout + o) * p.Do + d0) * out_plane + (long long)ho * p.Wo + wo : 0;
      vy[it] = *reinterpret_cast<gf4_p>((gfloat_p)p.dy + off);
    }
  };
  auto commit = [&](int d0, int h0, int w0) {
#pragma unroll
    for (int it = 0; it < C::ITX; ++it) {
      if (lane + 64 * it >= C::GPC) continue;
      const int hi = 2 * h0 - 1 + g_r[it], gc = 2 * w0 - 4 + 4 * g_q[it];
      const bool lane_ok = (unsigned)hi < (unsigned)p.Hi && gc >= 0 && gc + 3 < p.Wi;
#pragma unroll
      for (int k = 0; k < C::XCW; ++k) {
        const int din = d0 * p.sd - xdsh[k];
        const bool ok = lane_ok && xval[k] && (unsigned)din < (unsigned)p.Di;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = e2e::in_act(vx[k][it][j], xa[k], xb[k], xsl[k]);
          v[j] = ok ? t : 0.f;
        }
        float* row = xs + (wave * C::XCW + k) * C::CS + g_r[it] * C::PITCH;
        *reinterpret_cast<float2*>(row + 2 * g_q[it]) = make_float2(v[0], v[2]);              // even columns
        *reinterpret_cast<float2*>(row + C::HALF + 2 * g_q[it]) = make_float2(v[1], v[3]);    // odd columns
      }
    }
    const int ho = h0 + y_r, wo = w0 + y_col;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ol = wave * C::YCW + y_kl + 4 * it;
      const bool ok = ob * 32 + ol < p.Cout && ho < p.Ho && wo + 3 < p.Wo;
      float2* dst = reinterpret_cast<float2*>(ys + ol * C::OS + y_pi);
      dst[0] = ok ? make_float2(vy[it][0], vy[it][1]) : make_float2(0.f, 0.f);
      dst[1] = ok ? make_float2(vy[it][2], vy[it][3]) : make_float2(0.f, 0.f);
    }
  };

  if (tile_lo < tile_hi) {
    int n, d0, h0, w0;
    decode(tile_lo, n, d0, h0, w0);
    load_desc(n);
    prefetch(n, d0, h0, w0);
    for (long long tile = tile_lo; tile < tile_hi; ++tile) {
      commit(d0, h0, w0);
      int nn = n, nd0 = d0, nh0 = h0, nw0 = w0;
      const bool more = tile + 1 < tile_hi;
      if (more) {
        decode(tile + 1, nn, nd0, nh0, nw0);
        if (nn != n) load_desc(nn);
      }
      __syncthreads();
      if (more) prefetch(nn, nd0, nh0, nw0);

      const int li = lane & 15, lk = lane >> 4;
      const float* ap = ys + (oh * 16 + li) * C::OS + lk;
      const float* bp = xs + (ch * 16 + li) * C::CS + lk;
#pragma unroll
      for (int r = 0; r < C::TH; ++r) {
#pragma unroll
        for (int cq = 0; cq < C::TW / 4; ++cq) {
          const float a = ap[r * C::TW + cq * 4];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const float* brow = bp + (2 * r + kh) * C::PITCH + cq * 4;
            acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, brow[C::HALF + 1], acc[kh * 3 + 0], 0, 0, 0);
            acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, brow[2], acc[kh * 3 + 1], 0, 0, 0);
            acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, brow[C::HALF + 2], acc[kh * 3 + 2], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      n = nn; d0 = nd0; h0 = nh0; w0 = nw0;
    }
  }

  float* sp = p.slab + (long long)chunk * p.Cout * p.Cin * 9;
  const int c = cbase + ch * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = ob * 32 + oh * 16 + (lane >> 4) * 4 + r;
    if (o < p.Cout && c < p.Cin) {
      float* dst = sp + ((long long)o * p.Cin + c) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) dst[t] = acc[t][r];
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                                long long numel, int nchunks) {
  // out[e] = sum_k slab[k][e]: 4 waves x 4 independent running sums per element, combined in a fixed order
  // (deterministic); the serial one-thread-per-element loop over up to 512 slabs was latency bound
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < numel) {
    for (int k = w; k < nchunks; k += 16) {
      s0 += slab[(long long)k * numel + e];
      if (k + 4 < nchunks) s1 += slab[(long long)(k + 4) * numel + e];
      if (k + 8 < nchunks) s2 += slab[(long long)(k + 8) * numel + e];
      if (k + 12 < nchunks) s3 += slab[(long long)(k + 12) * numel + e];
    }
  }
  part[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && e < numel) out[e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

struct TileSel {
  int nd, th, tw;
};
inline TileSel pick(int Ho, int Wo, bool strided) {
  const int m = Ho < Wo ? Ho : Wo;
  if (!strided) {
    if (m > 16) return {1, 8, 32};
    if (m > 8) return {1, 16, 16};
    if (m > 4) return {4, 8, 8};
    return {16, 4, 4};
  }
  if (m > 8) return {1, 8, 16};
  if (m > 4) return {2, 8, 8};
  return {8, 4, 4};
}

inline void plan(WgParams& p, TileSel ts, int pairs, int* nchunks, int target_wgs = 1024) {
  p.tiles_x = e2e::cdiv(p.Wo, ts.tw);
  p.tiles_y = e2e::cdiv(p.Ho, ts.th);
  p.tiles_d = e2e::cdiv(p.Do, ts.nd);
  p.tiles_per_n = p.tiles_d * p.tiles_y * p.tiles_x;
  p.total_tiles = (long long)p.tiles_per_n * p.B;
  long long want = target_wgs / (pairs > 0 ? pairs : 1);
  if (want < 1) want = 1;
  long long tpc = e2e::cdivll(p.total_tiles, want);
  if (tpc < 4) tpc = 4;
  if (tpc > p.total_tiles) tpc = p.total_tiles;
  p.tiles_per_chunk = (int)tpc;
  *nchunks = (int)e2e::cdivll(p.total_tiles, tpc);
}

template <int SH, int SW, int ND, int TH, int TW>
int launch(const WgParams& p, int nchunks, int pairs, hipStream_t st) {
  hipLaunchKernelGGL((conv133_wgrad_kernel<SH, SW, ND, TH, TW>), dim3(nchunks, pairs), dim3(256), 0, st, p);
  return e2e::check_launch("conv133_wgrad_kernel");
}

template <int ND, int TH, int TW, int NCB>
int launch_v2(const WgParams& p, int nchunks, int pairs, hipStream_t st) {
  hipLaunchKernelGGL((conv133_wgrad_v2_kernel<ND, TH, TW, NCB>), dim3(nchunks, pairs), dim3(256 * NCB), 0, st, p);
  return e2e::check_launch("conv133_wgrad_v2_kernel");
}

// v2 applies to stride-1 (in plane) convs whose rows are multiples of 4 floats
// (planes whose smaller side is <= 4 get the (16, 4, 4) tile, which only the v1 kernel implements)
inline bool use_v2(int Hi, int Wi, int sh, int sw) { return sh == 1 && sw == 1 && (Wi % 4) == 0 && Wi > 4 && Hi > 4; }
inline int v2_ncb(int Cin, int Ho, int Wo) {      // 8x8 planes: the (4,8,8) tile only fits one channel block in LDS
  const int m = Ho < Wo ? Ho : Wo;
  return (Cin > 32 && m > 8) ? 2 : 1;
}

// v3 (double-buffered 4 x 32 tiles): stride-1 planes at least 32 wide whose rows are multiples of 4 floats
inline bool use_v3(int Cin, int Hi, int Wi, int sh, int sw) {
  return sh == 1 && sw == 1 && (Wi % 4) == 0 && Wi >= 32 && Hi > 16;
}
// block shape of a v3 workgroup: 32 out x 64 in channels, or 64 out x 32 in.  The wide-out shape stages less (the
// halo'd input tile is the expensive half) and pads fewer input channels (Cin = 160: three 64-blocks waste a sixth of
// the MFMAs, five 32-blocks none): 160 -> 64 @64^3 runs at 122 instead of 97 TFLOP/s.
inline bool v3_wide_out(int Cin, int Cout) {
  (void)Cin;
  return Cout >= 64 && (Cout % 64) <= 0;
}
inline bool v3_ksplit(int Cin, int Cout) { return Cin <= 32 && Cout <= 32; }   // one 32 x 32 block: split the tile rows
inline int v3_pairs(int Cin, int Cout) {
  if (v3_ksplit(Cin, Cout)) return 1;
  return v3_wide_out(Cin, Cout) ? e2e::cdiv(Cin, 32) * e2e::cdiv(Cout, 64) : e2e::cdiv(Cin, 64) * e2e::cdiv(Cout, 32);
}
// chunks never cross a batch item: `segs` runs of tiles_per_chunk tiles per item; returns the number of chunks
inline int plan_v3(WgParams& p, int pairs) {
  p.tiles_x = e2e::cdiv(p.Wo, 32);
  p.tiles_y = e2e::cdiv(p.Ho, 4);
  p.tiles_d = p.Do;
  p.tiles_per_n = p.tiles_d * p.tiles_y * p.tiles_x;
  p.total_tiles = (long long)p.tiles_per_n * p.B;
  const int target = 256;
  long long want = target / (pairs > 0 ? pairs : 1);        // one workgroup per CU, equal work each; fewer chunks = fewer slabs to reduce
  if (want < p.B) want = p.B;
  int segs = (int)(want / p.B);
  int tpc = e2e::cdiv(p.tiles_per_n, segs);
  if (tpc < 8) tpc = 8;
  if (tpc > p.tiles_per_n) tpc = p.tiles_per_n;
  segs = e2e::cdiv(p.tiles_per_n, tpc);
  p.tiles_per_chunk = tpc;
  p.cblocks_segs = segs;
  return segs * p.B;
}

// conv133_wgrad_bf3.hip (bf16 / fp16 matrix pipe, fp32-exact split operands): the shapes v3 serves; E2E_WG_BF3=0 keeps the
// fp32-MFMA kernels (forced-path tests, A/B runs).  32 x 32 channel blocks, chunks planned like v3's.
inline bool use_bf3(int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw) {
  static const int on = getenv("E2E_WG_BF3") ? atoi(getenv("E2E_WG_BF3")) : 1;
  const long long Do = (Di - 1) / sd + 1;                    // the kernel addresses with 32-bit element offsets inside one batch item
  const bool fits32 = (long long)Di * Hi * Wi < (1ll << 29) && (long long)Cout * Do * Hi * Wi < (1ll << 29);
  return on && Cin > 4 && fits32 && use_v3(Cin, Hi, Wi, sh, sw);
}
// fp16 two-piece operands (round 5) where the caller hands over max |dy|; E2E_WG_H2=0 keeps the bf16 three-piece form (A/B runs)
inline int wg_h2_env() { static const int v = getenv("E2E_WG_H2") ? atoi(getenv("E2E_WG_H2")) : 1; return v; }
inline int bf3_pairs(int Cin, int Cout) { return e2e::cdiv(Cin, 32) * e2e::cdiv(Cout, 32); }
// planes 16..31 voxels wide: the same kernel on 8 x 16-pixel tiles (conv133_wgrad_bf3v5_kernel<1>; round 4, they ran on the fp32
// MFMA v2 kernel); E2E_WG_BF3=0 keeps v2.  32-bit element offsets inside one batch item (as v5).
inline bool use_bf3_w16(int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw) {
  static const int bf3 = getenv("E2E_WG_BF3") ? atoi(getenv("E2E_WG_BF3")) : 1;
  const long long Do = (Di - 1) / sd + 1;
  return bf3 && Cin > 4 && sh == 1 && sw == 1 && (Wi % 4) == 0 && Wi >= 16 && Wi < 32 && Hi >= 8 &&
         (long long)Di * Hi * Wi < (1ll << 29) && (long long)Cout * Do * Hi * Wi < (1ll << 29);
}
inline int plan_w16(WgParams& p, int pairs) {
  p.tiles_x = e2e::cdiv(p.Wo, 16);
  p.tiles_y = e2e::cdiv(p.Ho, 8);
  p.tiles_d = p.Do;
  p.tiles_per_n = p.tiles_d * p.tiles_y * p.tiles_x;
  p.total_tiles = (long long)p.tiles_per_n * p.B;
  long long want = 256 / (pairs > 0 ? pairs : 1);            // one workgroup per CU
  if (want < p.B) want = p.B;
  int segs = (int)(want / p.B);
  int tpc = e2e::cdiv(p.tiles_per_n, segs);
  if (tpc < 4) tpc = 4;
  if (tpc > p.tiles_per_n) tpc = p.tiles_per_n;
  segs = e2e::cdiv(p.tiles_per_n, tpc);
  p.tiles_per_chunk = tpc;
  p.cblocks_segs = segs;
  return segs * p.B;
}

// network input layer (Cin <= 4): stride 1, rows multiples of 4 floats, planes at least one 8 x 32 tile
inline bool use_smallc(int Cin, int Hi, int Wi, int sh, int sw) {
  return Cin <= 4 && sh == 1 && sw == 1 && (Wi % 4) == 0 && Wi >= 32 && Hi >= 8;
}
inline int plan_smallc(WgParams& p, int pairs) {        // returns the number of workgroup chunks (slabs = 2 x that)
  p.tiles_x = e2e::cdiv(p.Wo, 32);
  p.tiles_y = e2e::cdiv(p.Ho, 8);
  p.tiles_d = p.Do;
  p.tiles_per_n = p.tiles_d * p.tiles_y * p.tiles_x;
  p.total_tiles = (long long)p.tiles_per_n * p.B;
  long long want = 1024 / (pairs > 0 ? pairs : 1);          // 4 workgroups per CU
  if (want < p.B) want = p.B;
  int segs = (int)(want / p.B);
  int tpc = e2e::cdiv(p.tiles_per_n, segs);
  if (tpc < 4) tpc = 4;
  if (tpc > p.tiles_per_n) tpc = p.tiles_per_n;
  segs = e2e::cdiv(p.tiles_per_n, tpc);
  p.tiles_per_chunk = tpc;
  p.cblocks_segs = segs;
  return segs * p.B;
}

// stride-(2,2) pipelined kernel: needs 16-byte aligned input rows and output planes at least one tile wide
inline bool use_s2(int Wi, int Wo, int sh, int sw) { return sh == 2 && sw == 2 && (Wi % 4) == 0 && (Wo % 4) == 0 && Wo >= 16; }
inline int s2_chunks(long long total_tiles, int pairs, int* tpc_out) {
  long long want = 512 / (pairs > 0 ? pairs : 1);
  if (want < 1) want = 1;
  long long tpc = e2e::cdivll(total_tiles, want);
  if (tpc < 8) tpc = 8;
  if (tpc > total_tiles) tpc = total_tiles;
  *tpc_out = (int)tpc;
  return (int)e2e::cdivll(total_tiles, tpc);
}

template <int SH, int SW>
int dispatch_strided(const WgParams& p, TileSel ts, int nchunks, int pairs, hipStream_t st) {
  if (ts.nd == 1) return launch<SH, SW, 1, 8, 16>(p, nchunks, pairs, st);
  if (ts.nd == 2) return launch<SH, SW, 2, 8, 8>(p, nchunks, pairs, st);
  return launch<SH, SW, 8, 4, 4>(p, nchunks, pairs, st);
}

}  // namespace

extern "C" long long e2e_conv133_wgrad_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh,
                                                int sw) {
  WgParams p{};
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.sd = sd;
  p.Do = (Di - 1) / sd + 1; p.Ho = (Hi - 1) / sh + 1; p.Wo = (Wi - 1) / sw + 1;
  int nchunks;
  if (use_s2(Wi, p.Wo, sh, sw)) {
    const int pairs = e2e::cdiv(Cin, 32) * e2e::cdiv(Cout, 32);
    p.tiles_x = e2e::cdiv(p.Wo, 16); p.tiles_y = e2e::cdiv(p.Ho, 4); p.tiles_d = p.Do;
    p.tiles_per_n = p.tiles_d * p.tiles_y * p.tiles_x;
    p.total_tiles = (long long)p.tiles_per_n * B;
    nchunks = s2_chunks(p.total_tiles, pairs, &p.tiles_per_chunk);
  } else if (use_smallc(Cin, Hi, Wi, sh, sw)) {
    nchunks = 2 * plan_smallc(p, e2e::cdiv(Cout, 32));
  } else if (use_bf3(Cin, Cout, Di, Hi, Wi, sd, sh, sw)) {
    nchunks = plan_v3(p, bf3_pairs(Cin, Cout));
  } else if (use_v3(Cin, Hi, Wi, sh, sw)) {
    nchunks = plan_v3(p, v3_pairs(Cin, Cout)) * (v3_ksplit(Cin, Cout) ? 2 : 1);
  } else if (use_bf3_w16(Cin, Cout, Di, Hi, Wi, sd, sh, sw)) {
    nchunks = plan_w16(p, bf3_pairs(Cin, Cout));
  } else if (use_v2(Hi, Wi, sh, sw)) {
    const int pairs = e2e::cdiv(Cin, 32 * v2_ncb(Cin, p.Ho, p.Wo)) * e2e::cdiv(Cout, 32);
    plan(p, pick(p.Ho, p.Wo, false), pairs, &nchunks, 512);
  } else {
    const int pairs = e2e::cdiv(Cin, 32) * e2e::cdiv(Cout, 32);
    plan(p, pick(p.Ho, p.Wo, sh != 1 || sw != 1), pairs, &nchunks);
  }
  return (long long)nchunks * Cout * Cin * 9 * (long long)sizeof(float);
}

extern "C" int e2e_conv133_wgrad(const e2e_in_chan_t* chans, const float* dy, float* dw, void* ws, int B, int Cin,
                                 int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw, const unsigned* dy_absmax,
                                 const unsigned* x_absmax, void* stream) {
  E2E_REQUIRE(chans && dy && dw && ws, "conv133_wgrad: null pointer");
  E2E_REQUIRE((sd == 1 || sd == 2) && (sh == 1 || sh == 2) && (sw == 1 || sw == 2), "conv133_wgrad: stride must be 1 or 2");
  hipStream_t st = (hipStream_t)stream;
  WgParams p{};
  p.dbg = 0;
  p.chans = chans; p.dy = dy; p.slab = reinterpret_cast<float*>(ws);
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.sd = sd;
  p.Do = (Di - 1) / sd + 1; p.Ho = (Hi - 1) / sh + 1; p.Wo = (Wi - 1) / sw + 1;
  const bool strided = sh != 1 || sw != 1;
  const TileSel ts = pick(p.Ho, p.Wo, strided);
  int nchunks;
  int rc;
  const long long numel = (long long)Cout * Cin * 9;
  if (use_s2(Wi, p.Wo, sh, sw)) {
    p.cblocks = e2e::cdiv(Cin, 32);
    const int pairs = p.cblocks * e2e::cdiv(Cout, 32);
    p.tiles_x = e2e::cdiv(p.Wo, 16); p.tiles_y = e2e::cdiv(p.Ho, 4); p.tiles_d = p.Do;
    p.tiles_per_n = p.tiles_d * p.tiles_y * p.tiles_x;
    p.total_tiles = (long long)p.tiles_per_n * B;
    nchunks = s2_chunks(p.total_tiles, pairs, &p.tiles_per_chunk);
    e2e::note_kernel("conv133_wgrad_s2 chunks=%d pairs=%d", nchunks, pairs);
    hipLaunchKernelGGL(conv133_wgrad_s2_kernel, dim3(nchunks, pairs), dim3(256), 0, st, p);
    rc = e2e::check_launch("conv133_wgrad_s2_kernel");
    if (rc != E2E_OK) return rc;
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)e2e::cdivll(numel, 64)), dim3(256), 0, st, p.slab, dw, numel,
                       nchunks);
    return e2e::check_launch("wgrad_slab_reduce_kernel");
  }
  if (use_smallc(Cin, Hi, Wi, sh, sw)) {
    const int pairs = e2e::cdiv(Cout, 32);
    const int wgs = plan_smallc(p, pairs);
    nchunks = 2 * wgs;
    e2e::note_kernel("conv133_wgrad_smallc chunks=%d pairs=%d", wgs, pairs);
    hipLaunchKernelGGL(conv133_wgrad_smallc_kernel, dim3(wgs, pairs), dim3(256), 0, st, p);
    rc = e2e::check_launch("conv133_wgrad_smallc_kernel");
    if (rc != E2E_OK) return rc;
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)e2e::cdivll(numel, 64)), dim3(256), 0, st, p.slab, dw, numel,
                       nchunks);
    return e2e::check_launch("wgrad_slab_reduce_kernel");
  }
  if (use_bf3(Cin, Cout, Di, Hi, Wi, sd, sh, sw)) {
    const int pairs = bf3_pairs(Cin, Cout);
    nchunks = plan_v3(p, pairs);
    e2e::WgBf3Params q{};
    q.chans = chans; q.dy = dy; q.slab = p.slab;
    q.B = B; q.Cin = Cin; q.Cout = Cout; q.Di = Di; q.Hi = Hi; q.Wi = Wi; q.Do = p.Do; q.sd = sd;
    q.tiles_x = p.tiles_x; q.tiles_y = p.tiles_y; q.tiles_per_n = p.tiles_per_n; q.tiles_per_chunk = p.tiles_per_chunk;
    q.segs = p.cblocks_segs; q.cblocks = e2e::cdiv(Cin, 32);
    q.h2 = wg_h2_env() && dy_absmax != nullptr; q.dy_absmax = dy_absmax; q.x_absmax = x_absmax;
    e2e::note_kernel("conv133_wgrad_%s chunks=%d pairs=%d", q.h2 ? "h2" : "bf3", nchunks, pairs);
    rc = e2e::launch_wgrad_bf3(q, nchunks, pairs, st);
    if (rc != E2E_OK) return rc;
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)e2e::cdivll(numel, 64)), dim3(256), 0, st, p.slab, dw, numel,
                       nchunks);
    return e2e::check_launch("wgrad_slab_reduce_kernel");
  }
  if (use_v3(Cin, Hi, Wi, sh, sw)) {
    const bool wide = v3_wide_out(Cin, Cout);
    p.cblocks = wide ? e2e::cdiv(Cin, 32) : e2e::cdiv(Cin, 64);
    const int pairs = v3_pairs(Cin, Cout);
    nchunks = plan_v3(p, pairs);
    e2e::note_kernel("conv133_wgrad_v3<%s> chunks=%d pairs=%d", v3_ksplit(Cin, Cout) ? "1,1,2" : (wide ? "1,2,1" : "2,1,1"), nchunks, pairs);
    if (v3_ksplit(Cin, Cout)) {
      p.cblocks = 1;
      hipLaunchKernelGGL((conv133_wgrad_v3_kernel<1, 1, 2>), dim3(nchunks, pairs), dim3(512), 0, st, p);
      nchunks *= 2;                                      // two row-half slabs per workgroup
    } else if (wide) hipLaunchKernelGGL((conv133_wgrad_v3_kernel<1, 2, 1>), dim3(nchunks, pairs), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv133_wgrad_v3_kernel<2, 1, 1>), dim3(nchunks, pairs), dim3(512), 0, st, p);
    rc = e2e::check_launch("conv133_wgrad_v3_kernel");
    if (rc != E2E_OK) return rc;
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)e2e::cdivll(numel, 64)), dim3(256), 0, st, p.slab, dw, numel,
                       nchunks);
    return e2e::check_launch("wgrad_slab_reduce_kernel");
  }
  if (use_bf3_w16(Cin, Cout, Di, Hi, Wi, sd, sh, sw)) {
    const int pairs = bf3_pairs(Cin, Cout);
    nchunks = plan_w16(p, pairs);
    e2e::WgBf3Params q{};
    q.chans = chans; q.dy = dy; q.slab = p.slab;
    q.B = B; q.Cin = Cin; q.Cout = Cout; q.Di = Di; q.Hi = Hi; q.Wi = Wi; q.Do = p.Do; q.sd = sd;
    q.tiles_x = p.tiles_x; q.tiles_y = p.tiles_y; q.tiles_per_n = p.tiles_per_n; q.tiles_per_chunk = p.tiles_per_chunk;
    q.segs = p.cblocks_segs; q.cblocks = e2e::cdiv(Cin, 32); q.geom = 1;
    q.h2 = wg_h2_env() && dy_absmax != nullptr; q.dy_absmax = dy_absmax; q.x_absmax = x_absmax;
    e2e::note_kernel("conv133_wgrad_%sw16 chunks=%d pairs=%d", q.h2 ? "h2" : "bf3", nchunks, pairs);
    rc = e2e::launch_wgrad_bf3(q, nchunks, pairs, st);
    if (rc != E2E_OK) return rc;
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)e2e::cdivll(numel, 64)), dim3(256), 0, st, p.slab, dw, numel,
                       nchunks);
    return e2e::check_launch("wgrad_slab_reduce_kernel");
  }
  if (use_v2(Hi, Wi, sh, sw)) {
    const int ncb = v2_ncb(Cin, p.Ho, p.Wo);
    p.cblocks = e2e::cdiv(Cin, 32 * ncb);
    const int pairs = p.cblocks * e2e::cdiv(Cout, 32);
    plan(p, ts, pairs, &nchunks, 512);
    e2e::note_kernel("conv133_wgrad_v2<%d,%d,%d,%d> chunks=%d pairs=%d", ts.nd == 1 ? 1 : 4, ts.nd == 1 ? (ts.tw == 32 ? 8 : 16) : 8, ts.nd == 1 ? ts.tw : 8, ts.nd == 1 ? ncb : 1, nchunks, pairs);
    if (ts.nd == 1 && ts.tw == 32) rc = ncb == 2 ? launch_v2<1, 8, 32, 2>(p, nchunks, pairs, st) : launch_v2<1, 8, 32, 1>(p, nchunks, pairs, st);
    else if (ts.nd == 1) rc = ncb == 2 ? launch_v2<1, 16, 16, 2>(p, nchunks, pairs, st) : launch_v2<1, 16, 16, 1>(p, nchunks, pairs, st);
    else rc = launch_v2<4, 8, 8, 1>(p, nchunks, pairs, st);
    if (rc != E2E_OK) return rc;
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)e2e::cdivll(numel, 64)), dim3(256), 0, st, p.slab, dw, numel,
                       nchunks);
    return e2e::check_launch("wgrad_slab_reduce_kernel");
  }
  p.cblocks = e2e::cdiv(Cin, 32);
  const int pairs = p.cblocks * e2e::cdiv(Cout, 32);
  plan(p, ts, pairs, &nchunks);
  e2e::note_kernel("conv133_wgrad_v1<%d,%d,%d,%d,%d> chunks=%d pairs=%d", sh, sw, ts.nd, ts.th, ts.tw, nchunks, pairs);
  if (!strided) {
    if (ts.nd == 1 && ts.tw == 32) rc = launch<1, 1, 1, 8, 32>(p, nchunks, pairs, st);
    else if (ts.nd == 1) rc = launch<1, 1, 1, 16, 16>(p, nchunks, pairs, st);
    else if (ts.nd == 4) rc = launch<1, 1, 4, 8, 8>(p, nchunks, pairs, st);
    else rc = launch<1, 1, 16, 4, 4>(p, nchunks, pairs, st);
  } else if (sh == 2 && sw == 2) {
    rc = dispatch_strided<2, 2>(p, ts, nchunks, pairs, st);
  } else if (sh == 1 && sw == 2) {
    rc = dispatch_strided<1, 2>(p, ts, nchunks, pairs, st);
  } else {
    rc = dispatch_strided<2, 1>(p, ts, nchunks, pairs, st);
  }
  if (rc != E2E_OK) return rc;
  hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)e2e::cdivll(numel, 64)), dim3(256), 0, st, p.slab, dw, numel,
                     nchunks);
  return e2e::check_launch("wgrad_slab_reduce_kernel");
}
