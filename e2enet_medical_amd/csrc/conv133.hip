// K1 / K6a: LDS-tiled direct 1x3x3 convolution for gfx950 (forward and stride-1 data gradient).
//
// Reference semantics: unetpp_d.py:45-59 (depth shift), :453-478 (concat), :93/:108 (Conv3d k(1,3,3)),
// and autograd of the same for the data gradient.
//
// Design (MI355X-first):
//   * one workgroup = one (batch item, output depth slice, TH x TW spatial tile, group of OCG output planes);
//   * the input tile of CK input planes (+1 halo) is staged through LDS once and shared by all OCG output
//     planes; the depth shift is an index offset in the load stage (plane p is read at depth d*sd - s(p)),
//     the concat is a per-plane pointer table, the producer's InstanceNorm+LeakyReLU is applied while staging;
//     with float4 staging (STG 1) every wave stages whole planes: descriptor in scalar registers, scalar-base
//     global loads prefetched one chunk ahead in registers; the chunk's weights go to LDS through a raw buffer
//     descriptor (out-of-range offsets read 0);
//   * one wave owns OPW output planes; a lane owns a PH x PW micro-tile of each (register accumulators);
//   * DSFF sparsity: "quad words" (4 output planes x 8 input planes, input-plane-major; e2e_dsff_expand_quads):
//     the wave walks the live input planes of its output planes with scalar bit ops, reads the neighbourhood
//     rows of a plane once and applies them to the 1..4 output planes that consume it; dead (out,in) kernels
//     cost nothing;
//   * epilogue (fwd): bias, store, per-tile (count, mean, M2) partial for InstanceNorm;
//     epilogue (dgrad): un-shift on store into the per-channel destination (scatter back through the concat).
#include "e2e_common.h"
#include <cstdlib>
#include <cstdio>
#include <type_traits>

// phase-split diagnostics (tools/kbench.py): build with -DE2E_CONV_DEBUG, then E2E_CONV_DBG=1|2|4 at run time
#ifdef E2E_CONV_DEBUG
#define CDBG(bit) ((p.dbg & (bit)) != 0)
// in-kernel phase stamps (s_memtime), summed over waves: [0] prologue [1] commit [2] barrier 1 [3] prefetch issue
// [4] live-kernel walk [5] barrier 2 [6] epilogue [7] waves; E2E_CONV_DBG=8 prints the per-wave averages after each launch
__device__ unsigned long long g_conv_stamps[1024 * 8];
#define STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define STAMP_ADD(i, a, b) st_acc[i] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(i, a, b)
#define CDBG(bit) false
#endif

namespace {

struct ConvParams {
  const e2e_in_chan_t* chans;   // P input planes (null: plain tensor `xin`, used by the data gradient)
  const float* xin;             // [B, P, Ds, Hs, Ws] when chans == null
  const float* w;
  const float* bias;            // fwd only (may be null)
  const unsigned* live;         // quad words [ceil(Q/4)][ceil(P/8)], bit (p % 8) * 4 + q % 4, or null (dense)
  float* y;                     // fwd
  double* part;                 // fwd, may be null: per-tile (count, mean, M2) of the stored values, fp64
  const e2e_out_chan_t* outs;   // dgrad
  int P, Q;                     // input planes, output planes
  int wq_stride, wp_stride;     // element strides of the weight tensor for (q, p)
  int live_words;
  int B, Di, Hi, Wi, Do, Ho, Wo, sd;
  int Ds, Hs, Ws;               // dgrad: dims of the dy tensor (== Di.. when not dilated)
  int tiles_x, tiles_y, tiles_per_n;   // tiles_per_n = Do * tiles_y * tiles_x
  int groups;                          // ceil(Q / OCG)
  int total;                           // B * tiles_per_n * groups (logical work items)
  int padded_total;
  int items_per_wg;
  int ksplit;                          // fwd: the input-plane chunks of a tile are split over ksplit workgroups (deep levels)
  float* kpart;                        // ... which store raw partial sums here: [ksplit][B][Q][Do][Ho][Wo]
  int dbg;                             // diagnostics (E2E_CONV_DBG): 1 = no staging, 2 = no FMA phase, 4 = no barriers
};

// wave-uniform description of one staged input plane, kept in LDS (double buffered per chunk)
struct PlaneDesc {
  gfloat_p base;      // plane origin for this (n, depth); always dereferenceable
  float a, b, slope;  // normalise-on-load coefficients
  int valid;
};

template <int MODE, int SH, int SW, int DH, int DW, int TH, int TW, int LY, int LX, int OPW, int NW, int CK, int STG>
struct Cfg {
  static_assert(LY * LX == 64, "one wave covers the tile");
  static_assert(MODE == 1 || (DH == 1 && DW == 1), "dilation is a data-gradient feature");
  // MODE 2 = sub-pixel data gradient of a stride-(2,2) conv: the LDS tile holds dy at its native resolution
  // (TH/2+1 x TW/2+1 for a TH x TW tile of dx) and only the parity-matching taps are applied
  static constexpr bool SUB = MODE == 2;
  static constexpr int PH = TH / LY, PW = TW / LX;
  static_assert(!SUB || (PH % 2 == 0 && PW % 2 == 0 && SH == 1 && SW == 1), "sub-pixel mode needs even micro-tiles");
  static constexpr int IH = SUB ? TH / 2 + 1 : (TH - 1) * SH + 3, IW = SUB ? TW / 2 + 1 : (TW - 1) * SW + 3;
  // lane column / row offset in the LDS tile; the column step decides the widest aligned read of a neighbourhood row
  static constexpr int LSTEP = SUB ? PW / 2 : PW * SW;
  static constexpr int LROW = SUB ? PH / 2 : PH * SH;
  static constexpr int VEC = (LSTEP % 4 == 0) ? 4 : (LSTEP % 2 == 0 ? 2 : 1);
    static constexpr int pitch_for(int iw) {
    int p = (iw + 3) & ~3;
    // VEC 4: consecutive lane rows are PH*SH tile rows apart; (PH*SH*pitch) % 64 == 32 puts the four 16-lane groups
    // of a ds_read_b128 on disjoint bank quarters.  VEC 2: the same idea for the two lane rows of a ds_read_b64 group.
    if (VEC == 4) { while ((LROW * p) % 64 != 32 && p < iw + 64) p += 4; }
    else if (VEC == 2) {
      // a 32-lane ds_read_b64 group = 32 / LX lane rows of LX lanes x 2 banks: row r must start at bank 2 LX r (mod 64)
      p = (iw + 1) & ~1;
      int q = p;
      while ((LROW * q) % 64 != (2 * LX) % 64 && q < iw + 64) q += 2;
      if (q < iw + 64) p = q;
      else { p = iw; while (p % 8 != 4) ++p; }
    }
    else p = iw;
    return p;
  }
  // (STG 1 writes whole float4 groups: the row pitch must also cover the last group's overhang, see commit())
  static constexpr int NQ0 = SUB ? (IW + 3) / 4 : (IW + 3 + 3) / 4;
  static constexpr int WMIN = STG ? 4 * NQ0 - (SUB ? 0 : 3) : IW;
  static constexpr int PITCH = pitch_for(WMIN > IW ? WMIN : IW);
  static constexpr int CHS = IH * PITCH;
  static constexpr int NT = NW * 64;
  static constexpr int OCG = OPW * NW;
  static constexpr int NR = SUB ? PH / 2 + 1 : (PH - 1) * SH + 3, NC = SUB ? PW / 2 + 1 : (PW - 1) * SW + 3;
  static constexpr int LDS_FLOATS = CK * CHS;
  // staging units: STG 1 = aligned float4 groups of a tile row (tile column origin is a multiple of 4, so the
  // group starting 4 columns left of it covers the halo), STG 0 = single elements
  static constexpr int NQ = SUB ? (IW + 3) / 4 : (IW + 3 + 3) / 4;      // SUB: groups start at the (aligned) tile origin
  static constexpr int UNITS = STG ? CK * IH * NQ : CK * IH * IW;
  static constexpr int NU = (UNITS + NT - 1) / NT;
};

// read NC consecutive floats starting at an address aligned to VEC floats
template <int NC, int VEC>
__device__ __forceinline__ void load_row(const float* __restrict__ src, float* __restrict__ dst) {
  int c = 0;
  if (VEC == 4) {
#pragma unroll
    for (; c + 4 <= NC; c += 4) {
      const float4 v = *reinterpret_cast<const float4*>(src + c);
      dst[c] = v.x; dst[c + 1] = v.y; dst[c + 2] = v.z; dst[c + 3] = v.w;
    }
  }
  if (VEC >= 2) {
#pragma unroll
    for (; c + 2 <= NC; c += 2) {
      const float2 v = *reinterpret_cast<const float2*>(src + c);
      dst[c] = v.x; dst[c + 1] = v.y;
    }
  }
#pragma unroll
  for (; c < NC; ++c) dst[c] = src[c];
}

// wave-uniform read-only operands (liveness words, bias, destination descriptors) go through the scalar cache: a vector
// load of them would sit in the same vmcnt queue as the staged planes and make the wave wait for those too
template <class T>
__device__ __forceinline__ T load_uniform(const T* ptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const T __attribute__((address_space(4)))*>((unsigned long long)ptr);
#else
  return *ptr;
#endif
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const f32x4_t __attribute__((address_space(1)))* gfloat4_p;

template <int MODE, int SH, int SW, int DH, int DW, int TH, int TW, int LY, int LX, int OPW, int NW, int CK, int STG, int MINW, int PIPE, int PERSIST>
__global__ __launch_bounds__(NW * 64, MINW) void conv133_kernel(ConvParams p) {
  using C = Cfg<MODE, SH, SW, DH, DW, TH, TW, LY, LX, OPW, NW, CK, STG>;
  static_assert((OPW == 4 || OPW == 8) && CK % 8 == 0 && CK <= 16, "liveness quad words: 4 output planes x 8 input planes per word");
  constexpr int NQD = OPW / 4;                                 // quad rows (of 4 output planes) per wave
  constexpr int WPAD = 12;                                     // 9 taps padded to 3 x 16 bytes
  // weights of one chunk for this workgroup's OCG output planes.  Two thread mappings:
  //  WRND: a thread keeps one (input plane, tap) unit and steps through the output planes QPR at a time -- one vector
  //        register for the global offset, one for the LDS address, everything else is scalar / immediate;
  //  else: unit u = tid + i * NT, decoded per round (used when the rounds of WRND would be more loads than that).
  // Out-of-range rows (planes beyond Q, the tail of the last chunk) read 0 through the buffer descriptor or are never
  // used (their liveness bits are cleared); idle threads write into the 3 pad floats of a kernel slot: no branches.
  constexpr int KU = CK * 9;
  constexpr int QPR = C::NT / KU;
  constexpr int NRW = QPR > 0 ? (C::OCG + (QPR > 0 ? QPR : 1) - 1) / (QPR > 0 ? QPR : 1) : 1 << 20;
  constexpr int WUNITS = C::OCG * KU;
  constexpr int NUW0 = (WUNITS + C::NT - 1) / C::NT;
  constexpr bool WRND = NRW <= NUW0;
  constexpr int NUW = WRND ? NRW : NUW0;
  constexpr int WROWS = WRND ? NRW * QPR : C::OCG;             // kernel-slot rows of `wl` (WRND: the last round overhangs)
  __shared__ __attribute__((aligned(16))) float lds_raw[C::LDS_FLOATS + 4];   // 4 guard floats: see the group writes of commit()
  __shared__ __attribute__((aligned(16))) float wl[WROWS * CK * WPAD];
  float* const lds = lds_raw + 4;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.Q * p.P * 36, 0x00020000);
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  PlaneDesc* tab = reinterpret_cast<PlaneDesc*>(dyn_lds);      // [P]: rebuilt when the (batch item, depth slice) changes

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int lx = lane % LX, ly = lane / LX;
  const long long in_plane = (long long)p.Hi * p.Wi;
  const long long src_plane = (long long)p.Hs * p.Ws;
  const float* lane_tp = lds + (ly * C::LROW) * C::PITCH + lx * C::LSTEP;
  const long long out_plane = (long long)p.Ho * p.Wo;

  // persistent workgroup: a run of consecutive logical work items (tile-major, output-plane group fastest).  The plane
  // descriptor table is rebuilt only when the (batch item, depth slice) changes, and the first chunk of the next item
  // is requested before the epilogue of the current one, so that a tile's start-up (dependent descriptor loads, first
  // chunk latency: 0.14 of 0.86 ms on 64->32 @128^3 when every tile was its own workgroup) is paid once per run.
  const int wg = e2e::xcd_remap(blockIdx.x, gridDim.x);
  const int item_lo = PERSIST ? wg * p.items_per_wg : wg;
  const int item_hi = PERSIST ? (item_lo + p.items_per_wg < p.total ? item_lo + p.items_per_wg : p.total)
                              : (item_lo < p.total ? item_lo + 1 : item_lo);
  if (item_lo >= item_hi) return;

  struct Item { int g, n, d, tile_in_n, h0, w0, ks; };
  auto decode = [&](int item) {
    Item it;
    it.ks = item % p.ksplit;              // split fastest: the parts of one tile run side by side
    item /= p.ksplit;
    it.g = item % p.groups;
    int t = item / p.groups;
    it.n = t / p.tiles_per_n;
    t -= it.n * p.tiles_per_n;
    it.tile_in_n = t;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    it.d = t / p.tiles_y;
    it.h0 = ty * TH;
    it.w0 = tx * TW;
    return it;
  };
  // data gradient of a depth-strided conv: only every sd-th slice of the (shifted) input received anything
  // (split-K: part ks of a tile takes the chunks [ks * per, (ks + 1) * per) of the ceil(P / CK) chunks)
  const int chunks_all = (p.P + CK - 1) / CK;
  const int chunks_per = (chunks_all + p.ksplit - 1) / p.ksplit;
  auto chunk_lo = [&](const Item& it) { return it.ks * chunks_per; };
  auto chunks_of = [&](const Item& it) {
    if ((MODE != 0) && (it.d % p.sd != 0)) return 0;
    const int lo = it.ks * chunks_per;
    const int left = chunks_all - lo;
    return left < 0 ? 0 : (left < chunks_per ? left : chunks_per);
  };

  // ---- descriptor of plane `pl` for the (n, d) of an item ------------------------------------------------------
  auto make_desc = [&](int pl, int n, int d) {
    PlaneDesc ds;
    ds.a = 1.f; ds.b = 0.f; ds.slope = 1.f; ds.valid = 0;
    if (MODE == 0) {
      ds.base = (gfloat_p)p.chans[0].ptr;
      const e2e_in_chan_t ch = p.chans[pl];
      const int din = d * p.sd - ch.dshift;
      if ((unsigned)din < (unsigned)p.Di) {
        ds.valid = 1;
        ds.base = (gfloat_p)(ch.ptr + (long long)n * ch.nstride + (long long)din * in_plane);
        if (ch.scale != nullptr) {
          ds.a = ch.scale[(long long)n * ch.ab_nstride];
          ds.b = ch.shift[(long long)n * ch.ab_nstride];
          ds.slope = ch.slope;
        }
      }
    } else {
      ds.valid = 1;
      ds.base = (gfloat_p)(p.xin + (((long long)n * p.P + pl) * p.Ds + d / p.sd) * src_plane);
    }
    return ds;
  };
  int tab_n = -1, tab_d = -1;
  // (all waves must be past their last read of the old table: callers sit behind the second barrier of a chunk loop)
  auto ensure_table = [&](const Item& it) {
    if (it.n != tab_n || it.d != tab_d) {
      for (int pl = tid; pl < p.P; pl += C::NT) tab[pl] = make_desc(pl, it.n, it.d);
      tab_n = it.n; tab_d = it.d;
      __syncthreads();
    }
  };

  // ---- staging geometry of an item ---------------------------------------------------------------------------
  // STG 1 (aligned float4 groups): every wave stages PPW whole planes of the chunk, so the plane descriptor is
  // wave-uniform (scalar registers, scalar-base global loads, no per-chunk address arithmetic in vector registers);
  // a lane owns NUP groups (row r, group q) of each of its planes.  All four elements of a group are written: the ones
  // outside the tile land in the pad columns of the row before / the same row (PITCH >= row + 3), never read back.
  // STG 0 (single elements, strided / dilated sources): units are spread over the whole workgroup.
  constexpr int PPW = CK / NW;
  constexpr int UPP = C::IH * C::NQ;
  constexpr int NUP = STG ? (UPP + 63) / 64 : 1;
  static_assert(!STG || CK % NW == 0, "a wave stages whole planes");
  static_assert(!STG || C::PITCH >= 4 * C::NQ - (C::SUB ? 0 : 3), "group writes stay inside the row pitch");
  int su_k[STG ? 1 : C::NU], su_lds[STG ? NUP : C::NU], su_goff[STG ? NUP : C::NU], su_mask[STG ? 1 : C::NU];
  bool su_ok[NUP];
  auto set_geometry = [&](const Item& it) {
    const int h0 = it.h0, w0 = it.w0;
    const int hbase = h0 * SH - 1, wbase = w0 * SW - 1;   // input tile origin (incl. halo)
    if constexpr (STG) {
#pragma unroll
      for (int i = 0; i < NUP; ++i) {
        int u = lane + 64 * i;
        if (u >= UPP) u = UPP - 1;                                  // idle lanes of the last round: harmless duplicates (skipped at commit)
        const int r = u / C::NQ, q = u - r * C::NQ;
        const int hi = C::SUB ? h0 / 2 + r : hbase + r;
        const int gc = C::SUB ? w0 / 2 + 4 * q : w0 * SW - 4 + 4 * q;
        const int lc0 = C::SUB ? 4 * q : 4 * q - 3;                 // tile column of the group's first element
        const int srcH = MODE == 0 ? p.Hi : p.Hs, srcW = MODE == 0 ? p.Wi : p.Ws;
        const bool ok = (unsigned)hi < (unsigned)srcH && gc >= 0 && gc + 3 < srcW;
        su_lds[i] = r * C::PITCH + lc0;
        su_goff[i] = ok ? (hi * srcW + gc) * 4 : 0;                 // byte offset inside the plane
        su_ok[i] = ok;
      }
    } else {
#pragma unroll
      for (int i = 0; i < C::NU; ++i) {
        const int u = tid + i * C::NT;
        int k = u / (C::IH * C::IW);
        const int rem = u - k * (C::IH * C::IW);
        const int r = rem / C::IW, cc = rem - r * C::IW;
        const int hi = C::SUB ? h0 / 2 + r : hbase + r, wi = C::SUB ? w0 / 2 + cc : wbase + cc;
        bool ok = u < C::UNITS && hi >= 0 && wi >= 0;
        int off = 0;
        if (C::SUB) {
          ok = ok && hi < p.Hs && wi < p.Ws;
          off = hi * p.Ws + wi;
        } else if (DH == 1 && DW == 1) {
          ok = ok && hi < p.Hi && wi < p.Wi;
          off = hi * p.Wi + wi;
        } else {   // dilated source: only positions that are multiples of the stride carry a value
          const int hs = hi / DH, wsrc = wi / DW;
          ok = ok && (hi - hs * DH) == 0 && (wi - wsrc * DW) == 0 && hs < p.Hs && wsrc < p.Ws;
          off = hs * p.Ws + wsrc;
        }
        if (u >= C::UNITS) k = 0;
        su_k[i] = k;
        su_lds[i] = k * C::CHS + r * C::PITCH + cc;
        su_goff[i] = ok ? off : -1;
        su_mask[i] = u < C::UNITS ? 1 : 0;
      }
    }
  };

  // ---- weight units of this thread (item-invariant; the item's plane group and the chunk enter as a scalar offset) ----
  unsigned wu_off[WRND ? 1 : NUW];
  int wu_lds[WRND ? 1 : NUW];
  if constexpr (WRND) {
    const bool act = tid < QPR * KU;
    const int ql = tid / KU, rem = tid - ql * KU;
    const int cl = rem / 9, kk = rem - cl * 9;
    wu_off[0] = act ? (unsigned)(ql * p.wq_stride + cl * p.wp_stride + (MODE == 1 ? 8 - kk : kk)) * 4u : 0x80000000u;
    wu_lds[0] = act ? (ql * CK + cl) * WPAD + kk : (tid - QPR * KU) * WPAD + 9;
  } else {
#pragma unroll
    for (int i = 0; i < NUW; ++i) {
      const int u = tid + i * C::NT;
      const int ql = u / KU;
      const int rem = u - ql * KU;
      const int cl = rem / 9, kk = rem - cl * 9;
      const bool ok = u < WUNITS;
      wu_off[i] = ok ? (unsigned)(ql * p.wq_stride + cl * p.wp_stride + (MODE == 1 ? 8 - kk : kk)) * 4u : 0x80000000u;
      wu_lds[i] = ok ? (ql * CK + cl) * WPAD + kk : ((u - WUNITS) % (C::OCG * CK)) * WPAD + 9;
    }
  }

  // ---- staging: issue the global loads of one chunk into registers (prefetch), commit them to LDS later -----
  float vw[NUW];
  f32x4_t v4[STG ? PPW : 1][NUP];
  float v1[STG ? 1 : C::NU];
  float pd_a[PPW], pd_b[PPW], pd_slope[PPW];     // wave-uniform descriptors of the planes in flight (STG 1)
  bool pd_ok[PPW];
  auto prefetch = [&](int c0, int qgroup) {
    if constexpr (STG) {
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int pl = c0 + wave * PPW + j;
        const PlaneDesc ds = tab[pl < p.P ? pl : p.P - 1];
        const unsigned long long bb = (unsigned long long)ds.base;
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bb), bhi = __builtin_amdgcn_readfirstlane((unsigned)(bb >> 32));
        const char __attribute__((address_space(1)))* base =
            (const char __attribute__((address_space(1)))*)(((unsigned long long)bhi << 32) | blo);
        pd_a[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ds.a)));
        pd_b[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ds.b)));
        pd_slope[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ds.slope)));
        pd_ok[j] = pl < p.P && __builtin_amdgcn_readfirstlane(ds.valid) != 0;
        if (pd_ok[j]) {
#pragma unroll
          for (int i = 0; i < NUP; ++i) v4[j][i] = *reinterpret_cast<gfloat4_p>(base + (unsigned)su_goff[i]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < C::NU; ++i) {
        const int pl = c0 + su_k[i];
        const PlaneDesc ds = tab[pl < p.P ? pl : p.P - 1];
        const bool ok = su_goff[i] >= 0 && pl < p.P && ds.valid;
        v1[i] = ds.base[ok ? su_goff[i] : 0];
      }
    }
    const unsigned coff = (unsigned)(qgroup * p.wq_stride + c0 * p.wp_stride) * 4u;
    if (!CDBG(16)) {
#pragma unroll
      for (int i = 0; i < NUW; ++i) {
        const unsigned off = WRND ? wu_off[0] + coff + (unsigned)(i * QPR * p.wq_stride) * 4u : wu_off[i] + coff;
        vw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, (int)off, 0, 0));
      }
    }
  };
  // ---- straight-line variant of the request (occupancy-tuned 16x32 configurations): the walk below is unrolled over the
  // CK plane slots of a chunk and slot cl issues request step cl -- unconditionally, so that the code has no branch
  // around a load (a branch there makes the wait-count pass drain vmcnt to 0 at every use; measured in the weight
  // gradient and in an earlier version of this loop).  Without a next chunk the steps read one dummy line / an
  // out-of-range buffer offset.  Why not one block after the barrier: all eight waves queue up in the texture addresser
  // and none of them can start its walk before its last load is accepted (in-order issue).
  constexpr bool UNROLL = STG && MINW >= 3 && PPW == 1 && OPW == 4;
  constexpr int NSTEP = STG ? PPW * NUP + NUW : 0;
  unsigned pf_blo = 0, pf_bhi = 0, pf_coff = 0;
  auto request_begin = [&](int c0, int qgroup) {
    if constexpr (UNROLL) {
      const int pl = c0 + wave;
      const PlaneDesc ds = tab[pl < p.P ? pl : p.P - 1];
      const unsigned long long bb = (unsigned long long)ds.base;
      pf_blo = __builtin_amdgcn_readfirstlane((unsigned)bb);
      pf_bhi = __builtin_amdgcn_readfirstlane((unsigned)(bb >> 32));
      pd_a[0] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ds.a)));
      pd_b[0] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ds.b)));
      pd_slope[0] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ds.slope)));
      pd_ok[0] = pl < p.P && __builtin_amdgcn_readfirstlane(ds.valid) != 0;
      pf_coff = (unsigned)(qgroup * p.wq_stride + c0 * p.wp_stride) * 4u;
    }
  };
  auto request_step = [&](int k, bool live) {              // k: compile-time; live: wave-uniform
    if constexpr (UNROLL) {
      if (k < NUP) {
        const char __attribute__((address_space(1)))* base =
            (const char __attribute__((address_space(1)))*)(((unsigned long long)pf_bhi << 32) | pf_blo);
        const unsigned off = live && pd_ok[0] ? (unsigned)su_goff[k] : 0u;      // (plane bases are always dereferenceable)
        v4[0][k] = *reinterpret_cast<gfloat4_p>(base + off);
      } else if (k < NSTEP) {
        const int i = k - NUP;
        const unsigned off = WRND ? wu_off[0] + pf_coff + (unsigned)(i * QPR * p.wq_stride) * 4u : wu_off[WRND ? 0 : i] + pf_coff;
        vw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, (int)(live && !CDBG(16) ? off : 0x80000000u), 0, 0));
      }
    }
  };

  auto commit = [&](int c0) {
    if constexpr (STG) {
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        float* plane = lds + (wave * PPW + j) * C::CHS;
#pragma unroll
        for (int i = 0; i < NUP; ++i) {
          if ((i + 1) * 64 > UPP && lane + 64 * i >= UPP) continue;
          float* dst = plane + su_lds[i];
          if (!pd_ok[j]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = 0.f;
          } else if (MODE == 0) {
            const float ae = su_ok[i] ? pd_a[j] : 0.f, be = su_ok[i] ? pd_b[j] : 0.f;   // out-of-image groups stage zeros
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = e2e::in_act(v4[j][i][e], ae, be, pd_slope[j]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = su_ok[i] ? v4[j][i][e] : 0.f;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < C::NU; ++i) {
        if (su_mask[i] == 0) continue;
        const int pl = c0 + su_k[i];
        const PlaneDesc ds = tab[pl < p.P ? pl : p.P - 1];
        const bool ok = su_goff[i] >= 0 && pl < p.P && ds.valid;
        float val = v1[i];
        if (MODE == 0) val = e2e::in_act(val, ds.a, ds.b, ds.slope);
        lds[su_lds[i]] = ok ? val : 0.f;
      }
    }
    if (!CDBG(16)) {
#pragma unroll
      for (int i = 0; i < NUW; ++i) wl[(WRND ? wu_lds[0] + i * QPR * CK * WPAD : wu_lds[i])] = vw[i];
    }
  };

  // ---- one live (output plane a, input plane cl) kernel: neighbourhood rows from LDS (shared by the wave's output
  // planes that consume the same input plane), 9 weights from LDS (broadcast reads), 9 x PH x PW FMAs ----
  constexpr int NCL = (C::VEC == 4) ? ((C::NC + 3) & ~3) : C::NC;     // VEC 4: whole float4s (conflict-free), over-read into the row pad
  static_assert(C::VEC != 4 || (LX - 1) * C::LSTEP + NCL <= C::PITCH, "row over-read stays inside the pitch");
  auto issue = [&](int cl, float (&nb)[C::NR][NCL]) {
    const float* tp = lane_tp + cl * C::CHS;
#pragma unroll
    for (int r = 0; r < C::NR; ++r) load_row<NCL, C::VEC>(tp + r * C::PITCH, nb[r]);
  };
  auto load_w = [&](int a, int cl, float (&wk)[9]) {
    const float* wp = wl + ((wave * OPW + a) * CK + cl) * WPAD;
    const float4 w0v = *reinterpret_cast<const float4*>(wp);
    const float4 w1v = *reinterpret_cast<const float4*>(wp + 4);
    wk[0] = w0v.x; wk[1] = w0v.y; wk[2] = w0v.z; wk[3] = w0v.w;
    wk[4] = w1v.x; wk[5] = w1v.y; wk[6] = w1v.z; wk[7] = w1v.w;
    wk[8] = wp[8];
  };
  // (v_pk_fma_f32 was tried here: on gfx950 a wave64 v_fma_f32 already issues in 2 cycles, so packing the FMAs buys
  //  nothing -- measured equal)
  auto apply = [&](float (&ac)[C::PH][C::PW], const float (&nb)[C::NR][NCL], const float (&wk)[9]) {
#pragma unroll
    for (int i = 0; i < C::PH; ++i)
#pragma unroll
      for (int j = 0; j < C::PW; ++j)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            if (!C::SUB) {
              ac[i][j] = fmaf(wk[kh * 3 + kw], nb[i * SH + kh][j * SW + kw], ac[i][j]);
            } else {
              // dx[2a + pi][2b + pj] += dy[(2a + pi + 1 - kh) / 2][(2b + pj + 1 - kw) / 2] * w[kh][kw] for matching parities:
              // even output index -> centre tap only; odd -> the two outer taps (tap 0 reads the next dy sample)
              const int pi = i & 1, pj = j & 1;
              const bool hit = (pi == 0 ? kh == 1 : kh != 1) && (pj == 0 ? kw == 1 : kw != 1);
              if (hit) ac[i][j] = fmaf(wk[kh * 3 + kw], nb[(i >> 1) + ((pi == 1 && kh == 0) ? 1 : 0)][(j >> 1) + ((pj == 1 && kw == 0) ? 1 : 0)], ac[i][j]);
            }
          }
  };

  // liveness of this wave's 4 output planes x the chunk's CK input planes: quad words (bit = (plane % 8) * 4 + output % 4),
  // input-plane-major, so consecutive set bits that share an input plane reuse the neighbourhood registers
  const int nw8 = (p.P + 7) >> 3;
  auto chunk_mask = [&](int c0, int j, int qbase) -> unsigned long long {      // j: quad row of this wave
    unsigned long long m = 0;
    const int qb = qbase + 4 * j;
    if (c0 >= p.P || qb >= p.Q) return m;
#pragma unroll
    for (int wd = 0; wd < CK / 8; ++wd) {
      const int wi = (c0 >> 3) + wd;
      unsigned word = 0u;
      if (wi < nw8) word = p.live != nullptr ? load_uniform(p.live + (long long)(qb >> 2) * nw8 + wi) : 0xffffffffu;
      m |= (unsigned long long)word << (32 * wd);                  // (scalar load: already wave-uniform)
    }
    const int remain = p.P - c0;
    if (remain < CK) m &= (1ull << (remain * 4)) - 1ull;
    const int qleft = p.Q - qb;                          // planes beyond Q (last group): clear their bit in every nibble
    if (qleft < 4) m &= 0x1111111111111111ull * ((1ull << qleft) - 1ull);
    return m;
  };

  // Two-level summation.  One sequential fp32 FMA chain over up to 896 x 9 products is 2-6x noisier than the blocked
  // summation of a CPU conv (measured against fp64, tools/scratch/conv_err.py), and the network amplifies that noise 3-5x per
  // level: with single chains the logits of BASELINE configs 1 and 5 sat 2-4x further from an fp64 evaluation than the fp32
  // CPU path does (profiles/r03_parity.json, "default" vs "twolvl_all").  Every tile shape therefore flushes the chunk's
  // partial sum (CK x 9 terms) into an outer accumulator after each chunk: +32 VGPRs in the 16x32 tile (128, still four
  // waves per SIMD), +1.7 % on the conv walk, +0.6 % on the 128^3 training step.  The small tiles of the deep levels (sums
  // over 320..896 planes in front of InstanceNorms over 8..175 voxels) keep the outer accumulator in fp64: 4 or 16 values
  // per wave, free in these latency-bound launches.  A single-chunk layer (PERSIST: Cin <= CK) has nothing to flush.
  constexpr bool TWOLVL = (PERSIST == 0) || (C::PH * C::PW <= 4);
#ifdef E2E_ACC2_F32
  using acc2_t = float;
#else
  using acc2_t = typename std::conditional<(C::PH * C::PW <= 4), double, float>::type;
#endif

  // ================= the run of items =============================================================================
  Item cur = decode(item_lo);
  bool requested = false;        // the first chunk of `cur` is already in flight (requested under the previous item)
  for (int item = item_lo;; ++item) {
#ifdef E2E_CONV_DEBUG
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(t_begin);
    const int n = cur.n, d = cur.d, h0 = cur.h0, w0 = cur.w0, tile_in_n = cur.tile_in_n;
    const int qgroup = cur.g * C::OCG;
    const int qbase = qgroup + wave * OPW;
    const int nchunks = chunks_of(cur);
    const int cbase = chunk_lo(cur) * CK;           // first input plane of this item's chunk window
    const bool has_next = PERSIST && item + 1 < item_hi;
    Item nxt = cur;
    if (has_next) nxt = decode(item + 1);
    if (!requested) {             // first item of the run, another (batch item, depth slice), or the previous item had no chunks
      set_geometry(cur);
      if (nchunks > 0) {
        ensure_table(cur);
        if (!CDBG(1)) prefetch(cbase, qgroup);
      }
    }
    requested = false;

    float acc[OPW][C::PH][C::PW];
#pragma unroll
    for (int a = 0; a < OPW; ++a)
#pragma unroll
      for (int i = 0; i < C::PH; ++i)
#pragma unroll
        for (int j = 0; j < C::PW; ++j) acc[a][i][j] = 0.f;
    acc2_t acc2[TWOLVL ? OPW : 1][TWOLVL ? C::PH : 1][TWOLVL ? C::PW : 1];
    if constexpr (TWOLVL) {
#pragma unroll
      for (int a = 0; a < OPW; ++a)
#pragma unroll
        for (int i = 0; i < C::PH; ++i)
#pragma unroll
          for (int j = 0; j < C::PW; ++j) acc2[a][i][j] = (acc2_t)0;
    }

    // epilogue operands, requested before the main loop so that their (dependent) global loads are long done when the
    // epilogue starts: in-kernel stamps showed the data-gradient epilogue at 7-11 k cycles per wave, eight serialised round
    // trips (destination descriptor -> old value, per output plane)
    e2e_out_chan_t ocs[MODE != 0 ? OPW : 1];
    float bqs[MODE == 0 ? OPW : 1];
#pragma unroll
    for (int a = 0; a < OPW; ++a) {
      const int q = qbase + a < p.Q ? qbase + a : p.Q - 1;
      if (MODE == 0) bqs[a] = p.bias ? load_uniform(p.bias + q) : 0.f;
      else if (p.ksplit > 1) {   // split-K part: raw sums into the workspace as a plain [n][Q][Do] tensor; conv133_dsum_kernel scatters
        e2e_out_chan_t oc{};
        const long long vol = (long long)p.Do * p.Ho * p.Wo;
        oc.ptr = p.kpart + ((long long)cur.ks * p.B * p.Q + q) * vol;
        oc.nstride = (long long)p.Q * vol;
        oc.dshift = 0;
        oc.accumulate = 0;
        ocs[a] = oc;
      } else ocs[a] = load_uniform(p.outs + q);
    }

    unsigned long long m_cur[NQD];
#pragma unroll
    for (int j = 0; j < NQD; ++j) m_cur[j] = nchunks > 0 ? chunk_mask(cbase, j, qbase) : 0ull;
    STAMP(t_pro);
    STAMP_ADD(0, t_begin, t_pro);

    for (int ci = 0; ci < nchunks; ++ci) {
      const int c0 = cbase + ci * CK;
      unsigned long long m_next[NQD];                      // scalar loads, in flight during commit
#pragma unroll
      for (int j = 0; j < NQD; ++j) m_next[j] = ci + 1 < nchunks ? chunk_mask(c0 + CK, j, qbase) : 0ull;
      STAMP(t0);
      if (!CDBG(1)) commit(c0);
      STAMP(t1);
      if (!CDBG(4)) __syncthreads();
      STAMP(t2);
      // one request site for "next chunk of this item" and "first chunk of the next item" (same descriptor table): the
      // staging registers must have one home, or the compiler copies them and waits for the loads right here
      int pf_c0 = c0 + CK, pf_qg = qgroup;
      bool pf = ci + 1 < nchunks;
      if (!pf && has_next && nxt.n == n && nxt.d == d && !CDBG(128)) {
        set_geometry(nxt);          // travels under this item's last walk and epilogue
        pf_c0 = chunk_lo(nxt) * CK; pf_qg = nxt.g * C::OCG;
        pf = true; requested = true;
      }
      if constexpr (UNROLL) {
        pf = pf && !CDBG(1);
        request_begin(pf ? pf_c0 : c0, pf ? pf_qg : qgroup);
      } else {
        if (pf && !CDBG(1)) prefetch(pf_c0, pf_qg);         // in flight while this chunk is computed
      }
      STAMP(t3);

      // walk the chunk's live input planes (nibbles of the quad mask); the neighbourhood rows of a plane are read once
      // and shared by the 1..4 output planes of this wave that consume it
      if constexpr (UNROLL) {
        const unsigned long long mw = CDBG(2) ? 0ull : m_cur[0];
#pragma unroll
        for (int cl = 0; cl < CK; ++cl) {
#pragma unroll
          for (int k = cl; k < NSTEP; k += CK) request_step(k, pf);
          const unsigned nib = (unsigned)(mw >> (cl * 4)) & 15u;
          if (nib) {
            float nb[C::NR][NCL];
            issue(cl, nb);
#pragma unroll
            for (int a = 0; a < OPW; ++a) {
              if (nib & (1u << a)) {
                float wk[9];
                load_w(a, cl, wk);
                apply(acc[a], nb, wk);
              }
            }
          }
        }
      } else {
      unsigned long long m = 0;
#pragma unroll
      for (int j = 0; j < NQD; ++j) m |= m_cur[j];
      if (CDBG(2)) m = 0;
      while (m) {
        const int cl = __builtin_ctzll(m) >> 2;
        m &= ~(15ull << (cl * 4));
        unsigned nib = 0;
#pragma unroll
        for (int j = 0; j < NQD; ++j) nib |= ((unsigned)(m_cur[j] >> (cl * 4)) & 15u) << (4 * j);
        float nb[C::NR][NCL];
        issue(cl, nb);
#pragma unroll
        for (int a = 0; a < OPW; ++a) {
          if (nib & (1u << a)) {
            float wk[9];
            load_w(a, cl, wk);
            apply(acc[a], nb, wk);
          }
        }
      }
      }
#pragma unroll
      for (int j = 0; j < NQD; ++j) m_cur[j] = m_next[j];
      if constexpr (TWOLVL) {
#pragma unroll
        for (int a = 0; a < OPW; ++a)
#pragma unroll
          for (int i = 0; i < C::PH; ++i)
#pragma unroll
            for (int j = 0; j < C::PW; ++j) { acc2[a][i][j] += (acc2_t)acc[a][i][j]; acc[a][i][j] = 0.f; }
      }
      STAMP(t4);
      if (!CDBG(4)) __syncthreads();
      STAMP(t5);
      STAMP_ADD(1, t0, t1); STAMP_ADD(2, t1, t2); STAMP_ADD(3, t2, t3); STAMP_ADD(4, t3, t4); STAMP_ADD(5, t4, t5);
    }
    STAMP(t_epi);
    if constexpr (TWOLVL) {
#pragma unroll
      for (int a = 0; a < OPW; ++a)
#pragma unroll
        for (int i = 0; i < C::PH; ++i)
#pragma unroll
          for (int j = 0; j < C::PW; ++j) acc[a][i][j] = (float)acc2[a][i][j];
    }

    // ---------------- epilogue ------------------------------------------------------------------------------
    const int oh0 = h0 + ly * C::PH, ow0 = w0 + lx * C::PW;
    float psum[OPW];                                      // fwd: per-lane sums of the stored values
#pragma unroll
    for (int a = 0; a < OPW; ++a) psum[a] = 0.f;
    float4* vdst[MODE != 0 ? OPW : 1][C::PH];             // dgrad, float4 rows: destination (null = nothing to store) and mode
    int vmode[MODE != 0 ? OPW : 1];
    if (MODE != 0) {
#pragma unroll
      for (int a = 0; a < OPW; ++a) {
        vmode[a] = 0;
#pragma unroll
        for (int i = 0; i < C::PH; ++i) vdst[a][i] = nullptr;
      }
    }
#pragma unroll
    for (int a = 0; a < OPW; ++a) {
      const int q = qbase + a;
      if (q >= p.Q || CDBG(64)) continue;
      if (MODE == 0) {
        const float bq = p.ksplit > 1 ? 0.f : bqs[a];          // split-K parts store raw sums: bias, statistics in conv133_ksum_kernel
        float* yp = (p.ksplit > 1 ? p.kpart + (long long)cur.ks * p.B * p.Q * p.Do * out_plane : p.y) +
                    (((long long)n * p.Q + q) * p.Do + d) * out_plane;
        float s = 0.f;
        const bool vec_store = (C::PW == 4) && (p.Wo % 4 == 0);      // lane rows are 16-byte aligned
        const bool vec2_store = (C::PW == 2) && (p.Wo % 2 == 0);     // ... or 8-byte aligned pairs
#pragma unroll
        for (int i = 0; i < C::PH; ++i) {
          const int oh = oh0 + i;
#pragma unroll
          for (int j = 0; j < C::PW; ++j) {
            const int ow = ow0 + j;
            const float val = acc[a][i][j] + bq;
            acc[a][i][j] = val;
            if (oh < p.Ho && ow < p.Wo) {
              if (!vec_store && !vec2_store) yp[(long long)oh * p.Wo + ow] = val;
              s += val;
            }
          }
          if (vec_store && oh < p.Ho && ow0 < p.Wo)
            *reinterpret_cast<float4*>(yp + (long long)oh * p.Wo + ow0) =
                make_float4(acc[a][i][0], acc[a][i][1 % C::PW], acc[a][i][2 % C::PW], acc[a][i][3 % C::PW]);
          if (vec2_store && oh < p.Ho && ow0 < p.Wo)
            *reinterpret_cast<float2*>(yp + (long long)oh * p.Wo + ow0) = make_float2(acc[a][i][0], acc[a][i][1 % C::PW]);
        }
        psum[a] = s;
      } else {
        // dgrad: gradient of virtual-concat channel q at (shifted) depth d goes to depth d - s(q) of its source
        const e2e_out_chan_t oc = ocs[a];
        if (oc.ptr == nullptr) continue;
        int dd = d - oc.dshift;
        bool zero_fill = false;
        if (dd < 0) {              // s > 0: slices [max(D-s,0), D) receive nothing -> this workgroup zero-fills one
          const int lo = p.Do - oc.dshift > 0 ? p.Do - oc.dshift : 0;
          dd = lo + d;
          zero_fill = true;
        } else if (dd >= p.Do) {   // s < 0: slices [0, min(-s, D)) receive nothing
          const int lo = p.Do + oc.dshift > 0 ? p.Do + oc.dshift : 0;
          dd = d - lo;
          zero_fill = true;
        }
        if (zero_fill && oc.accumulate) continue;
        float* xp = oc.ptr + (long long)n * oc.nstride + (long long)dd * out_plane;
        const bool vec_store = (C::PW == 4) && (p.Wo % 4 == 0);
        const bool vec2_store = (C::PW == 2) && (p.Wo % 2 == 0);
#pragma unroll
        for (int i = 0; i < C::PH; ++i) {
          const int oh = oh0 + i;
          if (vec2_store) {
            if (oh < p.Ho && ow0 < p.Wo) {
              float2* dst = reinterpret_cast<float2*>(xp + (long long)oh * p.Wo + ow0);
              float2 val = make_float2(acc[a][i][0], acc[a][i][1 % C::PW]);
              if (zero_fill) val = make_float2(0.f, 0.f);
              else if (oc.accumulate) { const float2 o = *dst; val.x += o.x; val.y += o.y; }
              *dst = val;
            }
            continue;
          }
          if (vec_store) {      // handled below: all old values are requested first, then added and stored
            if (oh < p.Ho && ow0 < p.Wo) {
              vdst[a][i] = reinterpret_cast<float4*>(xp + (long long)oh * p.Wo + ow0);
              vmode[a] = zero_fill ? 2 : (oc.accumulate ? 1 : 0);
            }
            continue;
          }
#pragma unroll
          for (int j = 0; j < C::PW; ++j) {
            const int ow = ow0 + j;
            if (oh < p.Ho && ow < p.Wo) {
              float* dst = xp + (long long)oh * p.Wo + ow;
              if (zero_fill) *dst = 0.f;
              else if (oc.accumulate) *dst += acc[a][i][j];
              else *dst = acc[a][i][j];
            }
          }
        }
      }
    }
    if (MODE != 0 && C::PW == 4) {
      float4 oldv[OPW][C::PH];
#pragma unroll
      for (int a = 0; a < OPW; ++a)
#pragma unroll
        for (int i = 0; i < C::PH; ++i) {
          oldv[a][i] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (vdst[a][i] != nullptr && vmode[a] == 1) oldv[a][i] = *vdst[a][i];
        }
#pragma unroll
      for (int a = 0; a < OPW; ++a)
#pragma unroll
        for (int i = 0; i < C::PH; ++i) {
          if (vdst[a][i] == nullptr) continue;
          float4 val = make_float4(acc[a][i][0], acc[a][i][1 % C::PW], acc[a][i][2 % C::PW], acc[a][i][3 % C::PW]);
          if (vmode[a] == 2) val = make_float4(0.f, 0.f, 0.f, 0.f);
          val.x += oldv[a][i].x; val.y += oldv[a][i].y; val.z += oldv[a][i].z; val.w += oldv[a][i].w;
          *vdst[a][i] = val;
        }
    }
    if (MODE == 0 && p.part != nullptr && !CDBG(32)) {
      // per-tile (count, mean, M2) partials of the 4 output planes: the count is known from the tile geometry, the four
      // sums (and then the four M2) are reduced side by side so their cross-lane steps overlap.  The record is fp64.  The
      // small tiles (deep levels: InstanceNorms over 8..175 voxels) also FORM it in fp64: an fp32 rounding of a plane's mean
      // is an error common to every voxel of the plane, which the 9 taps of the next conv add up linearly -- as large as the
      // conv's own rounding noise there (tools/scratch/node_err.py: 1.3-1.4x the CPU path's noise at the 5x7x5 nodes).
      const int vr = p.Ho - h0 < TH ? p.Ho - h0 : TH, vc = p.Wo - w0 < TW ? p.Wo - w0 : TW;
      const float tcnt = (float)(vr * vc);
      constexpr bool STAT64 = (C::PH * C::PW <= 4);
      using st_t = typename std::conditional<STAT64, double, float>::type;
      st_t mean[OPW], m2[OPW];
      if constexpr (STAT64) {
#pragma unroll
        for (int a = 0; a < OPW; ++a) {
          double sd = 0.0;
#pragma unroll
          for (int i = 0; i < C::PH; ++i)
#pragma unroll
            for (int j = 0; j < C::PW; ++j)
              if (oh0 + i < p.Ho && ow0 + j < p.Wo) sd += (double)acc[a][i][j];
          mean[a] = sd;
        }
#pragma unroll
        for (int a = 0; a < OPW; ++a) mean[a] = e2e::wave_sum_dpp_d(mean[a]) / (double)tcnt;
      } else {
#pragma unroll
        for (int a = 0; a < OPW; ++a) mean[a] = e2e::wave_sum_dpp(psum[a]) / tcnt;
      }
#pragma unroll
      for (int a = 0; a < OPW; ++a) {
        st_t t = 0;
#pragma unroll
        for (int i = 0; i < C::PH; ++i)
#pragma unroll
          for (int j = 0; j < C::PW; ++j) {
            const int oh = oh0 + i, ow = ow0 + j;
            if (oh < p.Ho && ow < p.Wo) {
              const st_t dlt = (st_t)acc[a][i][j] - mean[a];
              if constexpr (STAT64) t = fma(dlt, dlt, t);
              else t = fmaf(dlt, dlt, t);
            }
          }
        m2[a] = t;
      }
#pragma unroll
      for (int a = 0; a < OPW; ++a) {
        if constexpr (STAT64) m2[a] = e2e::wave_sum_dpp_d(m2[a]);
        else m2[a] = e2e::wave_sum_dpp(m2[a]);
      }
      if (lane < OPW && qbase + lane < p.Q) {
        st_t mm = mean[0], vv = m2[0];
#pragma unroll
        for (int a = 1; a < OPW; ++a)
          if (lane == a) { mm = mean[a]; vv = m2[a]; }
        double* pp = p.part + (((long long)n * p.Q + qbase + lane) * p.tiles_per_n + tile_in_n) * 3;
        pp[0] = (double)tcnt;
        pp[1] = (double)mm;
        pp[2] = (double)vv;
      }
    }
    STAMP(t_end);
    STAMP_ADD(6, t_epi, t_end);
    STAMP_ADD(7, t_end - 1, t_end);
#ifdef E2E_CONV_DEBUG
    if (lane == 0)
      for (int i = 0; i < 8; ++i) atomicAdd(&g_conv_stamps[(blockIdx.x & 1023) * 8 + i], st_acc[i]);
#endif
    if (!has_next) break;
    cur = nxt;
  }
}

// ---- split-K tail (deep levels): y = bias + sum over the parts (fixed order), per-tile (count, mean, M2) partials ----
// The deep levels have tiny planes and hundreds of input planes: a tile's workgroup walks 40-112 chunks one after the
// other (a chain of load and LDS latencies, 130-160 us for a few MFLOP) while most CUs have nothing to do.  Splitting the
// chunks over ksplit workgroups shortens the chain; this kernel adds the parts up.  One wave per (n, q, tile).
__global__ __launch_bounds__(64) void conv133_ksum_kernel(const float* __restrict__ kpart, const float* __restrict__ bias,
                                                          float* __restrict__ y, double* __restrict__ part, int ksplit, int B,
                                                          int Q, int Do, int Ho, int Wo, int TH, int TW, int tiles_x,
                                                          int tiles_y) {
  const int tiles_per_n = Do * tiles_y * tiles_x;
  int t = blockIdx.x;
  const int tile_in_n = t % tiles_per_n;
  t /= tiles_per_n;
  const int q = t % Q, n = t / Q;
  int u = tile_in_n;
  const int tx = u % tiles_x;
  u /= tiles_x;
  const int ty = u % tiles_y, d = u / tiles_y;
  const int h0 = ty * TH, w0 = tx * TW;
  const int vr = Ho - h0 < TH ? Ho - h0 : TH, vc = Wo - w0 < TW ? Wo - w0 : TW;
  const long long plane = (long long)Ho * Wo;
  const long long base = (((long long)n * Q + q) * Do + d) * plane;
  const long long kstride = (long long)B * Q * Do * plane;
  const float bq = bias ? bias[q] : 0.f;
  const int lane = threadIdx.x;
  const int npx = vr * vc;
  double s = 0.0;
  for (int i = lane; i < npx; i += 64) {
    const int r = i / vc, c = i - r * vc;
    const long long off = base + (long long)(h0 + r) * Wo + (w0 + c);
    double vs = 0.0;                    // (the parts are two-level sums themselves; their sum is formed in fp64)
    for (int k = 0; k < ksplit; ++k) vs += (double)kpart[(long long)k * kstride + off];
    const float v = (float)(vs + (double)bq);
    y[off] = v;
    s += (double)v;
  }
  const double tcnt = (double)npx;
  const double mean = e2e::wave_sum_dpp_d(s) / tcnt;      // fp64 statistics: see the epilogue of conv133_kernel
  double m2 = 0.0;
  for (int i = lane; i < npx; i += 64) {
    const int r = i / vc, c = i - r * vc;
    const double dlt = (double)y[base + (long long)(h0 + r) * Wo + (w0 + c)] - mean;     // (this lane's own store above)
    m2 = fma(dlt, dlt, m2);
  }
  m2 = e2e::wave_sum_dpp_d(m2);
  if (lane == 0 && part != nullptr) {
    double* pp = part + (((long long)n * Q + q) * tiles_per_n + tile_in_n) * 3;
    pp[0] = tcnt;
    pp[1] = mean;
    pp[2] = m2;
  }
}

// ---- split-K tail of the data gradient: sum of the parts (fixed order), then the scatter epilogue of conv133_kernel
// (un-shift on store, zero-fill of the slices that receive nothing, accumulate flag).  One wave per (n, q, tile).
__global__ __launch_bounds__(64) void conv133_dsum_kernel(const float* __restrict__ kpart, const e2e_out_chan_t* __restrict__ outs,
                                                          int ksplit, int B, int Q, int Do, int Ho, int Wo, int TH, int TW,
                                                          int tiles_x, int tiles_y) {
  const int tiles_per_n = Do * tiles_y * tiles_x;
  int t = blockIdx.x;
  const int tile_in_n = t % tiles_per_n;
  t /= tiles_per_n;
  const int q = t % Q, n = t / Q;
  int u = tile_in_n;
  const int tx = u % tiles_x;
  u /= tiles_x;
  const int ty = u % tiles_y, d = u / tiles_y;
  const e2e_out_chan_t oc = outs[q];
  if (oc.ptr == nullptr) return;
  int dd = d - oc.dshift;
  bool zero_fill = false;
  if (dd < 0) {
    const int lo = Do - oc.dshift > 0 ? Do - oc.dshift : 0;
    dd = lo + d;
    zero_fill = true;
  } else if (dd >= Do) {
    const int lo = Do + oc.dshift > 0 ? Do + oc.dshift : 0;
    dd = d - lo;
    zero_fill = true;
  }
  if (zero_fill && oc.accumulate) return;
  const int h0 = ty * TH, w0 = tx * TW;
  const int vr = Ho - h0 < TH ? Ho - h0 : TH, vc = Wo - w0 < TW ? Wo - w0 : TW;
  const long long plane = (long long)Ho * Wo;
  const long long src = (((long long)n * Q + q) * Do + d) * plane;
  const long long kstride = (long long)B * Q * Do * plane;
  float* xp = oc.ptr + (long long)n * oc.nstride + (long long)dd * plane;
  const int npx = vr * vc;
  for (int i = threadIdx.x; i < npx; i += 64) {
    const int r = i / vc, c = i - r * vc;
    const long long po = (long long)(h0 + r) * Wo + (w0 + c);
    float v = 0.f;
    if (!zero_fill) {
      double vs = 0.0;
      for (int k = 0; k < ksplit; ++k) vs += (double)kpart[(long long)k * kstride + src + po];
      v = (float)vs;
      if (oc.accumulate) v += xp[po];
    }
    xp[po] = v;
  }
}

// ---- strided data gradient (encoder "convolutional pooling" convs, 5 layers, dense): gather form ------------
// dx[c][di][hi][wi] = sum_o sum_{kh,kw : (hi+1-kh) % sh == 0, (wi+1-kw) % sw == 0} dy[o][ds][(hi+1-kh)/sh][(wi+1-kw)/sw] w[o][c][kh][kw]
// where the shifted depth ds*sd = di + s(c).
__global__ __launch_bounds__(256) void conv133_dgrad_strided_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                    const e2e_out_chan_t* outs, int B, int Cin, int Cout,
                                                                    int Di, int Hi, int Wi, int Do, int Ho, int Wo, int sd,
                                                                    int sh, int sw) {
  const long long plane = (long long)Hi * Wi;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  const int n = blockIdx.z;
  if (idx >= (long long)Di * plane) return;
  const e2e_out_chan_t oc = outs[c];
  if (oc.ptr == nullptr) return;
  const int di = (int)(idx / plane);
  const int rem = (int)(idx - (long long)di * plane);
  const int hi = rem / Wi, wi = rem - hi * Wi;
  const int dsft = di + oc.dshift;     // depth in the shifted tensor
  float acc = 0.f;
  if (dsft >= 0 && dsft < Di && dsft % sd == 0) {
    const int dz = dsft / sd;
    if (dz < Do) {
      for (int kh = 0; kh < 3; ++kh) {
        const int th = hi + 1 - kh;
        if (th < 0 || th % sh) continue;
        const int ho = th / sh;
        if (ho >= Ho) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int tw = wi + 1 - kw;
          if (tw < 0 || tw % sw) continue;
          const int wo = tw / sw;
          if (wo >= Wo) continue;
          const float* dyp = dy + (((long long)n * Cout) * Do + dz) * Ho * Wo + (long long)ho * Wo + wo;
          const float* wp = w + (long long)c * 9 + kh * 3 + kw;
          const long long ostride = (long long)Do * Ho * Wo;
          float s = 0.f;
          for (int o = 0; o < Cout; ++o) s = fmaf(dyp[o * ostride], wp[(long long)o * Cin * 9], s);
          acc += s;
        }
      }
    }
  }
  float* dst = oc.ptr + (long long)n * oc.nstride + idx;
  if (oc.accumulate) *dst += acc;
  else *dst = acc;
}

template <int MODE, int SH, int SW, int DH, int DW, int TH, int TW, int LY, int LX, int OPW, int NW, int CK, int STG, int MINW, int PIPE, int PERSIST>
int launch_cfg_impl(ConvParams p, hipStream_t st);

// persistent grid: 4 workgroup slots' worth per CU (256 CUs x 2 resident workgroups x 2 rounds); E2E_CONV_WGS overrides (A/B)
inline int persist_knob() {
  static const int v = getenv("E2E_CONV_PERSIST") ? atoi(getenv("E2E_CONV_PERSIST")) : 0;
  return v;
}
inline int persist_wgs() {
  static const int v = getenv("E2E_CONV_WGS") ? atoi(getenv("E2E_CONV_WGS")) : 1024;
  return v > 8 ? v : 8;
}

template <int MODE, int SH, int SW, int DH, int DW, int TH, int TW, int LY, int LX, int OPW, int NW, int CK, int STG, int MINW = 1, int PIPE = 0>
int launch_cfg(ConvParams p, hipStream_t st) {
  // thin layers (a single input chunk, e.g. the 4-modal network input): per-tile start-up dominates -> persistent
  // workgroups that walk runs of consecutive items and request the next item's planes under the current epilogue.
  // For multi-chunk layers the run loop was measured SLOWER than one workgroup per item (0.91 vs 0.83 ms on 64->32
  // @128^3: the hardware dispatcher hides the start-up of the next workgroup behind the running ones, and runs that
  // start together stay in phase); E2E_CONV_PERSIST=1 forces it for A/B runs.  (Element staging keeps one item per
  // workgroup: the run loop would spill there.)
  if constexpr (STG == 1) {
    const long long total = (long long)p.B * p.Do * e2e::cdiv(p.Ho, TH) * e2e::cdiv(p.Wo, TW) * e2e::cdiv(p.Q, Cfg<MODE, SH, SW, DH, DW, TH, TW, LY, LX, OPW, NW, CK, STG>::OCG);
    if ((p.P <= CK || persist_knob()) && total > persist_wgs())
      return launch_cfg_impl<MODE, SH, SW, DH, DW, TH, TW, LY, LX, OPW, NW, CK, STG, MINW, PIPE, 1>(p, st);
  }
  return launch_cfg_impl<MODE, SH, SW, DH, DW, TH, TW, LY, LX, OPW, NW, CK, STG, MINW, PIPE, 0>(p, st);
}

template <int MODE, int SH, int SW, int DH, int DW, int TH, int TW, int LY, int LX, int OPW, int NW, int CK, int STG, int MINW, int PIPE, int PERSIST>
int launch_cfg_impl(ConvParams p, hipStream_t st) {
  using C = Cfg<MODE, SH, SW, DH, DW, TH, TW, LY, LX, OPW, NW, CK, STG>;
  p.tiles_x = e2e::cdiv(p.Wo, TW);
  p.tiles_y = e2e::cdiv(p.Ho, TH);
  p.tiles_per_n = p.Do * p.tiles_y * p.tiles_x;
  p.groups = e2e::cdiv(p.Q, C::OCG);
  if (p.ksplit < 1 || PERSIST) p.ksplit = 1;
  double* const part_out = p.part;
  if (p.ksplit > 1) p.part = nullptr;                 // raw partial sums: statistics come from conv133_ksum_kernel
  p.total = p.B * p.tiles_per_n * p.groups * p.ksplit;
  p.padded_total = (p.total + 7) & ~7;
  static const int dbg_knob = getenv("E2E_CONV_DBG") ? atoi(getenv("E2E_CONV_DBG")) : 0;
  p.dbg = dbg_knob;
  int wgs = PERSIST ? persist_wgs() : p.total;
  if (wgs > p.total) wgs = p.total;
  p.items_per_wg = e2e::cdiv(p.total, wgs);
  wgs = (e2e::cdiv(p.total, p.items_per_wg) + 7) & ~7;
  e2e::note_kernel("conv133_kernel<mode=%d,s=%dx%d,dil=%dx%d,tile=%dx%d,opw=%d,nw=%d,ck=%d,stg=%d,persist=%d> wgs=%d ksplit=%d", MODE, SH, SW, DH, DW,
                   TH, TW, OPW, NW, CK, STG, PERSIST, wgs, p.ksplit);
  hipLaunchKernelGGL((conv133_kernel<MODE, SH, SW, DH, DW, TH, TW, LY, LX, OPW, NW, CK, STG, MINW, PIPE, PERSIST>), dim3(wgs),
                     dim3(C::NT), (size_t)p.P * sizeof(PlaneDesc), st, p);
  if (p.ksplit > 1 && MODE == 0)
    hipLaunchKernelGGL(conv133_ksum_kernel, dim3(p.B * p.Q * p.tiles_per_n), dim3(64), 0, st, p.kpart, p.bias, p.y, part_out,
                       p.ksplit, p.B, p.Q, p.Do, p.Ho, p.Wo, TH, TW, p.tiles_x, p.tiles_y);
  if (p.ksplit > 1 && MODE != 0)
    hipLaunchKernelGGL(conv133_dsum_kernel, dim3(p.B * p.Q * p.tiles_per_n), dim3(64), 0, st, p.kpart, p.outs, p.ksplit, p.B,
                       p.Q, p.Do, p.Ho, p.Wo, TH, TW, p.tiles_x, p.tiles_y);
#ifdef E2E_CONV_DEBUG
  if (p.dbg & 8) {
    hipStreamSynchronize(st);
    static unsigned long long hh[1024 * 8];
    hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_conv_stamps), sizeof(hh));
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 1024 * 8; ++i) h[i & 7] += hh[i];
    const double w = h[7] ? (double)h[7] : 1.0;
    fprintf(stderr, "[conv133 MODE %d P %d Q %d] per-wave cycles(100MHz ticks): pro %.0f commit %.0f bar1 %.0f prefetch %.0f walk %.0f bar2 %.0f epi %.0f (waves %.0f)\n",
            MODE, p.P, p.Q, h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w, w);
    static unsigned long long zz[1024 * 8];
    hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), zz, sizeof(zz));
  }
#endif
  return e2e::check_launch("conv133_kernel");
}

// sub-pixel data gradient (stride (2,2) convs); dx planes of 8x8 and smaller keep the zero-dilated MODE 1 path
int launch_sub(const ConvParams& p, int kind, hipStream_t st);


inline int opw8_knob() {
  return 0;
}

// stride-1 tiles; STG = 1 (aligned float4 staging) needs rows that are multiples of 4 floats
template <int MODE, int DH, int DW>
int launch_s1(const ConvParams& p, int kind, hipStream_t st) {
  const bool vec = (DH == 1 && DW == 1) && (p.Wi % 4 == 0);
  switch (kind) {
    case 0:
      // 16x32 tile, 97 VGPRs -> two 512-thread workgroups per CU (measured best; see DESIGN.md §5 for the variants
      // that were measured against it)
      if (vec && opw8_knob() == 1 && p.P > 8) return launch_cfg<MODE, 1, 1, 1, 1, 16, 32, 8, 8, 8, 4, 8, 1, 3, 0>(p, st);   // 4 fat waves
      if (vec && opw8_knob() == 2 && p.P > 8) return launch_cfg<MODE, 1, 1, 1, 1, 16, 32, 8, 8, 4, 4, 8, 1, 4, 0>(p, st);   // 4 waves, 16 output planes: 4 workgroups per CU
      return vec ? launch_cfg<MODE, 1, 1, 1, 1, 16, 32, 8, 8, 4, 8, 8, 1, 4, 0>(p, st)
                 : launch_cfg<MODE, 1, 1, DH, DW, 16, 32, 8, 8, 4, 8, 8, 0, 4, 0>(p, st);
    case 1:
      return vec ? launch_cfg<MODE, 1, 1, 1, 1, 16, 16, 8, 8, 4, 8, 16, 1>(p, st)
                 : launch_cfg<MODE, 1, 1, DH, DW, 16, 16, 8, 8, 4, 8, 16, 0>(p, st);
    default:
      return vec ? launch_cfg<MODE, 1, 1, 1, 1, 8, 8, 8, 8, 4, 8, 16, 1>(p, st)
                 : launch_cfg<MODE, 1, 1, DH, DW, 8, 8, 8, 8, 4, 8, 16, 0>(p, st);
  }
}

// tile selection shared by the launcher and e2e_conv133_num_partials
enum TileKind { T32 = 0, T16 = 1, T8 = 2, T16x32S = 3, T8S = 4 };
inline TileKind pick_tile(int Ho, int Wo, int sh, int sw) {
  const bool strided = (sh != 1 || sw != 1);
  const int m = Ho < Wo ? Ho : Wo;
  if (strided) return m > 8 ? T16x32S : T8S;
  if (m > 16) return T32;
  if (m > 8) return T16;
  return T8;
}
inline void tile_dims(TileKind k, int& th, int& tw) {
  switch (k) {
    case T32: th = 16; tw = 32; break;
    case T16: th = 16; tw = 16; break;
    case T8: th = 8; tw = 8; break;
    case T16x32S: th = 16; tw = 32; break;
    default: th = 8; tw = 8; break;
  }
}



int launch_sub(const ConvParams& p, int kind, hipStream_t st) {
  const bool vec = (p.Ws % 4) == 0;
  if (kind == 0)
    return vec ? launch_cfg<2, 1, 1, 1, 1, 16, 32, 8, 8, 4, 8, 8, 1, 4, 0>(p, st)
               : launch_cfg<2, 1, 1, 1, 1, 16, 32, 8, 8, 4, 8, 8, 0, 4, 0>(p, st);
  return vec ? launch_cfg<2, 1, 1, 1, 1, 16, 16, 8, 8, 4, 8, 16, 1>(p, st)
             : launch_cfg<2, 1, 1, 1, 1, 16, 16, 8, 8, 4, 8, 16, 0>(p, st);
}

}  // namespace

extern "C" int e2e_conv133_num_partials(int Do, int Ho, int Wo, int sh, int sw) {
  int th, tw;
  tile_dims(pick_tile(Ho, Wo, sh, sw), th, tw);
  return Do * e2e::cdiv(Ho, th) * e2e::cdiv(Wo, tw);
}

// split-K plan of the forward (deep levels only: planes no larger than a 16 x 16 tile): number of parts, 1 = off
static int plan_fwd_ksplit(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw) {
  static const int knob = getenv("E2E_CONV_KSPLIT") ? atoi(getenv("E2E_CONV_KSPLIT")) : -1;   // 0/1 = off, n = force n (A/B)
  const int Do = (Di - 1) / sd + 1, Ho = (Hi - 1) / sh + 1, Wo = (Wi - 1) / sw + 1;
  const TileKind k = pick_tile(Ho, Wo, sh, sw);
  if (k == T32 || k == T16x32S) return 1;
  int th, tw;
  tile_dims(k, th, tw);
  const int chunks = e2e::cdiv(Cin, 16);                   // (these configurations stage 16 planes per chunk)
  const long long base = (long long)B * Do * e2e::cdiv(Ho, th) * e2e::cdiv(Wo, tw) * e2e::cdiv(Cout, 32);
  if (knob == 0 || knob == 1) return 1;
  long long ks = knob > 1 ? knob : 1024 / (base > 0 ? base : 1);
  if (ks > 8) ks = 8;
  if (ks > chunks / 2) ks = chunks / 2;
  return ks < 2 ? 1 : (int)ks;
}

extern "C" long long e2e_conv133_fwd_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw) {
  const int ks = plan_fwd_ksplit(B, Cin, Cout, Di, Hi, Wi, sd, sh, sw);
  if (ks <= 1) return 0;
  const long long Do = (Di - 1) / sd + 1, Ho = (Hi - 1) / sh + 1, Wo = (Wi - 1) / sw + 1;
  return (long long)ks * B * Cout * Do * Ho * Wo * 4;
}

static int conv133_fwd_impl(const e2e_in_chan_t* chans, int Cin, const float* w, const float* bias, const unsigned* live, float* y,
                            double* part, int B, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw, float* ws,
                            long long ws_bytes, void* stream) {
  E2E_REQUIRE(chans && w && y, "conv133_fwd: null pointer");
  E2E_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Di > 0 && Hi > 0 && Wi > 0, "conv133_fwd: bad dims");
  E2E_REQUIRE((sd == 1 || sd == 2) && (sh == 1 || sh == 2) && (sw == 1 || sw == 2), "conv133_fwd: stride must be 1 or 2");
  ConvParams p{};
  p.chans = chans; p.xin = nullptr; p.w = w; p.bias = bias; p.live = live; p.y = y; p.part = part; p.outs = nullptr;
  p.P = Cin; p.Q = Cout; p.wq_stride = Cin * 9; p.wp_stride = 9; p.live_words = e2e::cdiv(Cin, 32);
  p.B = B; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.sd = sd;
  p.Do = (Di - 1) / sd + 1; p.Ho = (Hi - 1) / sh + 1; p.Wo = (Wi - 1) / sw + 1;
  p.Ds = Di; p.Hs = Hi; p.Ws = Wi;
  p.ksplit = 1; p.kpart = nullptr;
  if (ws != nullptr) {
    const long long need = e2e_conv133_fwd_ws_bytes(B, Cin, Cout, Di, Hi, Wi, sd, sh, sw);
    if (need > 0) {
      E2E_REQUIRE(ws_bytes >= need, "conv133_fwd: workspace too small (%lld < %lld bytes)", ws_bytes, need);
      p.ksplit = plan_fwd_ksplit(B, Cin, Cout, Di, Hi, Wi, sd, sh, sw);
      p.kpart = ws;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const TileKind k = pick_tile(p.Ho, p.Wo, sh, sw);
  if (sh == 1 && sw == 1) return launch_s1<0, 1, 1>(p, k == T32 ? 0 : (k == T16 ? 1 : 2), st);
  // strided variants ("convolutional pooling", 5 layers): element staging with register prefetch
  if (sh == 2 && sw == 2) {
    if (k == T16x32S && (Wi % 4) == 0) return launch_cfg<0, 2, 2, 1, 1, 16, 32, 4, 16, 4, 8, 8, 1>(p, st);   // float4 staging
    if (k == T16x32S) return launch_cfg<0, 2, 2, 1, 1, 16, 32, 4, 16, 4, 8, 8, 0>(p, st);
    return launch_cfg<0, 2, 2, 1, 1, 8, 8, 8, 8, 4, 8, 16, 0>(p, st);
  }
  if (sh == 1 && sw == 2) {
    if (k == T16x32S) return launch_cfg<0, 1, 2, 1, 1, 16, 32, 4, 16, 4, 8, 8, 0>(p, st);
    return launch_cfg<0, 1, 2, 1, 1, 8, 8, 8, 8, 4, 8, 16, 0>(p, st);
  }
  if (k == T16x32S) return launch_cfg<0, 2, 1, 1, 1, 16, 32, 4, 16, 4, 8, 8, 0>(p, st);
  return launch_cfg<0, 2, 1, 1, 1, 8, 8, 8, 8, 4, 8, 16, 0>(p, st);
}

extern "C" int e2e_conv133_fwd(const e2e_in_chan_t* chans, int Cin, const float* w, const float* bias,
                               const unsigned* live, float* y, double* part, int B, int Cout, int Di, int Hi, int Wi,
                               int sd, int sh, int sw, void* stream) {
  return conv133_fwd_impl(chans, Cin, w, bias, live, y, part, B, Cout, Di, Hi, Wi, sd, sh, sw, nullptr, 0, stream);
}

extern "C" int e2e_conv133_fwd_splitk(const e2e_in_chan_t* chans, int Cin, const float* w, const float* bias,
                                      const unsigned* live, float* y, double* part, int B, int Cout, int Di, int Hi, int Wi,
                                      int sd, int sh, int sw, float* ws, long long ws_bytes, void* stream) {
  return conv133_fwd_impl(chans, Cin, w, bias, live, y, part, B, Cout, Di, Hi, Wi, sd, sh, sw, ws, ws_bytes, stream);
}

// split-K plan of the data gradient (deep levels; the tiled paths only)
static int plan_dgrad_ksplit(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw) {
  static const int knob = getenv("E2E_CONV_KSPLIT") ? atoi(getenv("E2E_CONV_KSPLIT")) : -1;
  if (knob == 0 || knob == 1) return 1;
  const bool tiled = (sh == 1 && sw == 1) || (sh == 2 && sw == 2);
  if (!tiled) return 1;
  const TileKind k = pick_tile(Hi, Wi, 1, 1);
  if (k == T32) return 1;
  int th, tw;
  tile_dims(k, th, tw);
  const int chunks = e2e::cdiv(Cout, 16);
  const long long base = (long long)B * Di * e2e::cdiv(Hi, th) * e2e::cdiv(Wi, tw) * e2e::cdiv(Cin, 32);
  long long ks = knob > 1 ? knob : 1024 / (base > 0 ? base : 1);
  if (ks > 8) ks = 8;
  if (ks > chunks / 2) ks = chunks / 2;
  return ks < 2 ? 1 : (int)ks;
}

extern "C" long long e2e_conv133_dgrad_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw) {
  const int ks = plan_dgrad_ksplit(B, Cin, Cout, Di, Hi, Wi, sd, sh, sw);
  return ks <= 1 ? 0 : (long long)ks * B * Cin * Di * Hi * Wi * 4;
}

static int conv133_dgrad_impl(const float* dy, const float* w, const unsigned* live_t, const e2e_out_chan_t* outs,
                              int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw, float* ws,
                              long long ws_bytes, void* stream) {
  E2E_REQUIRE(dy && w && outs, "conv133_dgrad: null pointer");
  E2E_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Di > 0 && Hi > 0 && Wi > 0, "conv133_dgrad: bad dims");
  E2E_REQUIRE((sd == 1 || sd == 2) && (sh == 1 || sh == 2) && (sw == 1 || sw == 2), "conv133_dgrad: stride must be 1 or 2");
  hipStream_t st = (hipStream_t)stream;
  const int Do = (Di - 1) / sd + 1, Ho = (Hi - 1) / sh + 1, Wo = (Wi - 1) / sw + 1;
  const bool tiled = (sh == 1 && sw == 1) || (sh == 2 && sw == 2);
  if (!tiled) {     // in-plane anisotropic strides: rare, plain gather kernel
    const long long per = (long long)Di * Hi * Wi;
    dim3 grid((unsigned)e2e::cdivll(per, 256), Cin, B);
    e2e::note_kernel("conv133_dgrad_strided_gather");
    hipLaunchKernelGGL(conv133_dgrad_strided_kernel, grid, dim3(256), 0, st, dy, w, outs, B, Cin, Cout, Di, Hi, Wi, Do,
                       Ho, Wo, sd, sh, sw);
    return e2e::check_launch("conv133_dgrad_strided_kernel");
  }
  // The forward kernel with transposed, tap-reversed weights.  Its "input planes" are dy's channels (a plain tensor,
  // no shift; for a strided conv dy is read zero-dilated: only positions that are multiples of the stride carry a
  // value, and only every sd-th depth slice is non-zero), its output planes are the virtual-concat input channels
  // (un-shift on store).
  ConvParams p{};
  p.chans = nullptr; p.xin = dy; p.w = w; p.bias = nullptr; p.live = live_t; p.y = nullptr; p.part = nullptr; p.outs = outs;
  p.P = Cout; p.Q = Cin; p.wq_stride = 9; p.wp_stride = Cin * 9; p.live_words = e2e::cdiv(Cout, 32);
  p.B = B; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.sd = sd; p.Do = Di; p.Ho = Hi; p.Wo = Wi;
  p.Ds = Do; p.Hs = Ho; p.Ws = Wo;
  p.ksplit = 1; p.kpart = nullptr;
  if (ws != nullptr) {
    const long long need = e2e_conv133_dgrad_ws_bytes(B, Cin, Cout, Di, Hi, Wi, sd, sh, sw);
    if (need > 0) {
      E2E_REQUIRE(ws_bytes >= need, "conv133_dgrad: workspace too small (%lld < %lld bytes)", ws_bytes, need);
      p.ksplit = plan_dgrad_ksplit(B, Cin, Cout, Di, Hi, Wi, sd, sh, sw);
      p.kpart = ws;
    }
  }
  const TileKind k = pick_tile(Hi, Wi, 1, 1);
  const int kind = k == T32 ? 0 : (k == T16 ? 1 : 2);
  if (sh == 1) return launch_s1<1, 1, 1>(p, kind, st);
  if (kind <= 1 && (Hi % 2) == 0 && (Wi % 2) == 0) {     // sub-pixel form: weights are read un-reversed
    p.wq_stride = 9; p.wp_stride = Cin * 9;
    return launch_sub(p, kind, st);
  }
  return launch_s1<1, 2, 2>(p, kind, st);
}

extern "C" int e2e_conv133_dgrad(const float* dy, const float* w, const unsigned* live_t, const e2e_out_chan_t* outs,
                                 int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw,
                                 void* stream) {
  return conv133_dgrad_impl(dy, w, live_t, outs, B, Cin, Cout, Di, Hi, Wi, sd, sh, sw, nullptr, 0, stream);
}

extern "C" int e2e_conv133_dgrad_splitk(const float* dy, const float* w, const unsigned* live_t, const e2e_out_chan_t* outs,
                                        int B, int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw, float* ws,
                                        long long ws_bytes, void* stream) {
  return conv133_dgrad_impl(dy, w, live_t, outs, B, Cin, Cout, Di, Hi, Wi, sd, sh, sw, ws, ws_bytes, stream);
}
