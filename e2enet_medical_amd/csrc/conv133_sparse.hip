// K1s: the DSFF-masked 1x3x3 convolution (forward and stride-1 data gradient) on a LOAD-BALANCED execution plan.
//
// Reference semantics: unetpp_d.py:45-59 (depth shift), :453-478 (concat), :93/:108 (Conv3d k(1,3,3) of a DSFF-masked weight,
// core_channel.py:427-434) and autograd of the same for the data gradient -- the operator of conv133.hip, for the layers that
// dominate the step (planes wider than 16 voxels, W % 4 == 0, stride 1, more than one 8-plane chunk, a DSFF kernel map).
//
// Why a second kernel.  In conv133_kernel a wave owns 4 output planes and the eight waves of a workgroup walk the live kernels
// of one 8-plane chunk between two barriers.  With a random kernel map at density 0.2 a wave has Binomial(32, 0.2) live kernels
// per chunk (6.4 +- 2.3) and every barrier waits for the slowest of eight: s_memtime stamps of the shipping kernel
// (profiles/r04_k1_phases.txt) put 18 % of a wave's life into that barrier and 47 % into the walk itself.  The ORDER in which
// input planes are chunked and the assignment of output planes to waves are free, so the host picks them per 32-plane group
// such that every (chunk, wave) cell carries nearly the same work (e2e_conv133_sparse_plan: longest-processing-time assignment
// + pair swaps; sum over chunks of the slowest wave drops from 1.48x to 1.15x of the mean).  What follows from the plan:
//   * weights are read from a PACKED copy that holds the LIVE kernels only, in the order the waves walk them:
//     [group][chunk][kmax kernel slots][12 floats], wave w's list starting at slot woff[group][chunk][w]
//     (e2e_conv133_sparse_pack: one launch for all layers of a network, after every optimizer step).  A chunk's block (2-4 KB at
//     density 0.2 instead of 12 KB) goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), double
//     buffered; the walk reads a wave's kernels one after the other;
//   * the LDS that the dense weight image would take holds a SECOND image of the staged planes: a wave commits the planes of
//     chunk c + 1 right after its walk of chunk c, so there is ONE barrier per chunk and the waves that finish a chunk early
//     spend the wait for the slowest on their share of the staging instead of idling;
//   * the plane table, the destination table and the liveness words arrive in plan order (built by the caller);
//   * a neighbourhood row is read as ds_read_b128 + ds_read_b64 (the compiler merged the halo halves of two rows into one
//     ds_read2_b64, whose accesses are banked mod 32 and collide 2-way: SQ_LDS_BANK_CONFLICT was 39 % of the LDS cycles);
//   * the two-level summation flushes after `flush_every` chunks (a chain of ~18 products: every chunk at density >= 0.2, fewer
//     flushes for sparser maps; the noise against fp64 stays below the torch-CPU conv's);
//   * an accumulating data gradient loads the old dx values in its PROLOGUE into the outer accumulators (they are free until the
//     first flush): no read-modify-write round trip in the epilogue;
//   * a data gradient that is the LAST writer of a gradient buffer also forms the InstanceNorm-backward sums of that buffer's
//     producer in its epilogue (e2e_in_sum_chan_t): this kernel is VALU bound, the extra read of y rides on idle HBM bandwidth,
//     and the producer's e2e_in_lrelu_bwd drops its first streaming pass over (dz, y).
// Everything else (16 x 32 tile, 8 waves x 4 output planes, 2 x 4 micro-tile per lane, register-prefetched float4 plane staging
// with normalise-on-load, quad-nibble walk unrolled over the eight plane slots, fp64 statistics records) is conv133_kernel's.
#include "e2e_common.h"
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

// phase stamps (s_memtime), diagnostic build -DE2E_CONV_DEBUG with E2E_CONV_DBG=8: [0] prologue (incl. first chunk) [1] commit of the
// next chunk + load wait [2] the barrier [3] request setup [4] walk [5] flush [6] epilogue [7] waves
#ifdef E2E_CONV_DEBUG
__device__ unsigned long long g_sparse_stamps[1024 * 8];
#define SSTAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define SSTAMP_ADD(i, a, b) st_acc[i] += (b) - (a)
#else
#define SSTAMP(var)
#define SSTAMP_ADD(i, a, b)
#endif

namespace {

constexpr int TH = 16, TW = 32, LX = 8, PH = 2, PW = 4, OPW = 4, NW = 8, CK = 8, OCG = 32;
constexpr int IH = TH + 2, IW = TW + 2;
constexpr int NQ = (IW + 3 + 3) / 4;             // float4 groups per staged row (the one 4 columns left of the tile covers the halo)
constexpr int PITCH = 48;                         // (2 * PITCH) % 64 == 32: the lane groups of a ds_read_b128 fall on disjoint bank quarters
constexpr int CHS = IH * PITCH;
constexpr int UPP = IH * NQ, NUP = (UPP + 63) / 64;
constexpr int WSLOT = 12;                         // 9 taps padded to 3 x 16 bytes
constexpr int WCHUNK = OCG * CK * WSLOT;          // floats of one chunk's weight block (12 KB)
constexpr int NR = PH + 2, NCL = 6;

struct SparseParams {
  const e2e_in_chan_t* chans;   // MODE 0: [groups][ppad] plane descriptors in plan order (ptr == null: empty slot)
  const float* xin;             // MODE 1: dy [B, P, D, H, W]
  const int* pslot;             // MODE 1: [groups][ppad] dy channel of each chunk slot (-1: empty)
  const float* wpk;             // [groups][nchunks][kmax][12]: the live kernels of a chunk, wave by wave in walk order
  const unsigned* quads;        // [groups][8 waves][nchunks]: bit cl * 4 + a
  const int* woff;              // [groups][nchunks][8]: first kernel slot of each wave's list
  const int* qslot;             // MODE 0: [groups][32] output plane of slot wave * 4 + a (-1: empty)
  const e2e_out_chan_t* outs;   // MODE 1: [groups][32] destinations in plan order (ptr == null: nothing to store)
  const e2e_in_sum_chan_t* insum;   // MODE 1, optional: [groups][32] fused InstanceNorm-backward sums of the channels written last here
  const float* bias;
  float* y;
  double* part;
  int P, Q, B, D, H, W;
  int nchunks, ppad, groups, flush_every, kmax, uw;      // uw: 16-byte units of a chunk's weight block per wave
  int dbg;                                               // diagnostic build: 1 no walk, 2 no plane loads, 4 no stores
  int tiles_x, tiles_y, tiles_per_n, total, padded_total;
};

struct PlaneDesc {
  gfloat_p base;
  float a, b, slope;
  int valid;
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef const f32x4_t __attribute__((address_space(1)))* gfloat4_p;
typedef const volatile f32x2_t __attribute__((address_space(3)))* lds_v2_p;      // (volatile: not merged into ds_read2_b64)
typedef const f32x4_t __attribute__((address_space(3)))* lds_v4_p;
typedef const float __attribute__((address_space(3)))* lds_f_p;

template <class T>
__device__ __forceinline__ T load_uniform(const T* ptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const T __attribute__((address_space(4)))*>((unsigned long long)ptr);      // scalar cache
#else
  return *ptr;
#endif
}

template <int MODE>
__global__ __launch_bounds__(NW * 64, 4) void conv133_sparse_kernel(SparseParams p) {
  constexpr int IMG = CK * CHS;                         // one image of a chunk's planes
  __shared__ __attribute__((aligned(16))) float lds_raw[2 * IMG];
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  float* const wl = reinterpret_cast<float*>(dyn_lds);              // 2 x kmax x 12 floats
  PlaneDesc* tab = reinterpret_cast<PlaneDesc*>(dyn_lds + (size_t)2 * p.kmax * WSLOT * 4);
  float* const lds = lds_raw;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int lx = lane % LX, ly = lane / LX;
  const long long plane = (long long)p.H * p.W;
  const float* lane_tp = lds + (ly * PH) * PITCH + lx * PW;

  int item = e2e::xcd_remap(blockIdx.x, p.padded_total);
  if (item >= p.total) return;
#ifdef E2E_CONV_DEBUG
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  SSTAMP(t_begin);
  const int g = item % p.groups;
  item /= p.groups;
  const int n = item / p.tiles_per_n, tile_in_n = item - n * p.tiles_per_n;
  const int tx = tile_in_n % p.tiles_x, ty = (tile_in_n / p.tiles_x) % p.tiles_y, d = tile_in_n / (p.tiles_x * p.tiles_y);
  const int h0 = ty * TH, w0 = tx * TW;

  // ---- staging geometry: a wave stages one whole plane of each chunk; a lane loads the NUP CONSECUTIVE aligned float4 groups
  // 3 lane .. 3 lane + 2 of the plane's IH x NQ groups (group (r, q) = image row r, global columns w0 - 4 + 4 q ..).  The LDS image
  // is shifted by one column against those groups (its column 0 is the left halo, so that every lane's neighbourhood rows start
  // 16-byte aligned): image unit (r, q) = the last element of group q and the first three of group q + 1 -- the lane's own next
  // group, or, for its last one, the first group of lane + 1 (one cross-lane move per element).  Written as aligned
  // ds_write_b128; the straight copy of the global groups needed four ds_write_b32 per group at a lane stride of four words:
  // 4-way bank conflicts, 96 instead of 24 LDS-array cycles per wave and chunk, a third of this kernel's LDS time.
  int su_lds[NUP], su_goff[NUP];
  bool su_ok[NUP], su_wr[NUP];
#pragma unroll
  for (int i = 0; i < NUP; ++i) {
    const int u0 = 3 * lane + i;
    const int u = u0 < UPP ? u0 : UPP - 1;                // (idle lanes: harmless duplicate loads, no writes)
    const int r = u / NQ, q = u - r * NQ;
    const int hi = h0 - 1 + r, gc = w0 - 4 + 4 * q;
    const bool ok = (unsigned)hi < (unsigned)p.H && gc >= 0 && gc + 3 < p.W;
    su_lds[i] = r * PITCH + 4 * q;
    su_goff[i] = ok ? (hi * p.W + gc) * 4 : 0;
    su_ok[i] = ok;
    su_wr[i] = u0 < UPP && q < NQ - 1;
  }

  // ---- staging: planes through registers (prefetched one chunk ahead), weights by LDS-DMA into the other weight buffer ---------
  f32x4_t v4[NUP];
  float pd_a = 1.f, pd_b = 0.f, pd_slope = 1.f;
  bool pd_ok = false;
  unsigned pf_blo = 0, pf_bhi = 0;
  const int wstride = p.kmax * WSLOT;                    // floats of a chunk's weight block
  const float* wblock = p.wpk + (long long)g * p.nchunks * wstride;
  auto request_begin = [&](int c) {
    const PlaneDesc ds = tab[c * CK + wave];
    const unsigned long long bb = (unsigned long long)ds.base;
    pf_blo = __builtin_amdgcn_readfirstlane((unsigned)bb);
    pf_bhi = __builtin_amdgcn_readfirstlane((unsigned)(bb >> 32));
    pd_a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ds.a)));
    pd_b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ds.b)));
    pd_slope = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ds.slope)));
    pd_ok = __builtin_amdgcn_readfirstlane(ds.valid) != 0;
  };
  auto request_plane = [&](int k, bool live) {           // k: compile-time
    const char __attribute__((address_space(1)))* base =
        (const char __attribute__((address_space(1)))*)(((unsigned long long)pf_bhi << 32) | pf_blo);
#ifdef E2E_CONV_DEBUG
    const unsigned off = live && pd_ok && !(p.dbg & 2) ? (unsigned)su_goff[k] : 0u;
#else
    const unsigned off = live && pd_ok ? (unsigned)su_goff[k] : 0u;
#endif
    v4[k] = *reinterpret_cast<gfloat4_p>(base + off);
  };
  // a chunk's weight block = 3 kmax float4 units; wave w moves units [uw w, uw w + uw) with up to two LDS-DMA instructions
  // (LDS destination = wave-uniform base + 16 * lane)
  auto request_weights = [&](int c, int half, bool live) {
    if (!live || half * 64 >= p.uw) return;               // (wave-uniform)
    const int u = wave * p.uw + half * 64 + lane;
    const float* src = wblock + (long long)c * wstride + u * 4;
    float* dst = wl + (c & 1) * wstride + (wave * p.uw + half * 64) * 4;
    if (half * 64 + lane < p.uw && u < 3 * p.kmax)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto commit = [&](int img) {
    float* pl = lds + img * IMG + wave * CHS;
    float t[NUP][4];
#pragma unroll
    for (int i = 0; i < NUP; ++i) {
      const float ae = su_ok[i] ? pd_a : 0.f, be = su_ok[i] ? pd_b : 0.f;   // out-of-image groups stage zeros
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!pd_ok) t[i][e] = 0.f;
        else if (MODE == 0) t[i][e] = e2e::in_act(v4[i][e], ae, be, pd_slope);
        else t[i][e] = su_ok[i] ? v4[i][e] : 0.f;
      }
    }
    float nx[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) nx[e] = __shfl_down(t[0][e], 1, 64);        // the first group of lane + 1 follows this lane's last
#pragma unroll
    for (int i = 0; i < NUP; ++i) {
      if (!su_wr[i]) continue;
      const float4 val = i + 1 < NUP ? make_float4(t[i][3], t[(i + 1) % NUP][0], t[(i + 1) % NUP][1], t[(i + 1) % NUP][2])
                                     : make_float4(t[i][3], nx[0], nx[1], nx[2]);
      *reinterpret_cast<float4*>(pl + su_lds[i]) = val;
    }
  };

  // ---- the first chunk is requested before anything else: this wave's plane descriptor comes straight from the plan tables on
  // the scalar path, the loads are in flight while the descriptor table of the remaining chunks is built (the prologue was 11 %
  // of a wave's life in the forward, 21 % in a 4-chunk data gradient: three dependent round trips -- table, barrier, first loads)
  {
    gfloat_p base = (gfloat_p)p.wpk;                      // always dereferenceable
    bool ok = false;
    if (MODE == 0) {
      const e2e_in_chan_t ch = load_uniform(p.chans + (long long)g * p.ppad + wave);
      const int din = d - ch.dshift;
      if (ch.ptr != nullptr && (unsigned)din < (unsigned)p.D) {
        ok = true;
        base = (gfloat_p)(ch.ptr + (long long)n * ch.nstride + (long long)din * plane);
        if (ch.scale != nullptr) {
          pd_a = load_uniform(ch.scale + (long long)n * ch.ab_nstride);
          pd_b = load_uniform(ch.shift + (long long)n * ch.ab_nstride);
          pd_slope = ch.slope;
        }
      }
    } else {
      const int c = load_uniform(p.pslot + (long long)g * p.ppad + wave);
      if (c >= 0) {
        ok = true;
        base = (gfloat_p)(p.xin + (((long long)n * p.P + c) * p.D + d) * plane);
      }
    }
    const unsigned long long bb = (unsigned long long)base;
    pf_blo = (unsigned)bb;
    pf_bhi = (unsigned)(bb >> 32);
    pd_ok = ok;
  }
#pragma unroll
  for (int k = 0; k < NUP; ++k) request_plane(k, true);
  request_weights(0, 0, true);
  request_weights(0, 1, true);

  // ---- plane table of this (group, batch item, depth slice) ------------------------------------------------------------------
  for (int pl = tid; pl < p.ppad; pl += NW * 64) {
    PlaneDesc ds;
    ds.a = 1.f; ds.b = 0.f; ds.slope = 1.f; ds.valid = 0;
    ds.base = (gfloat_p)p.wpk;                                    // always dereferenceable
    if (MODE == 0) {
      const e2e_in_chan_t ch = p.chans[(long long)g * p.ppad + pl];
      const int din = d - ch.dshift;
      if (ch.ptr != nullptr && (unsigned)din < (unsigned)p.D) {
        ds.valid = 1;
        ds.base = (gfloat_p)(ch.ptr + (long long)n * ch.nstride + (long long)din * plane);
        if (ch.scale != nullptr) {
          ds.a = ch.scale[(long long)n * ch.ab_nstride];
          ds.b = ch.shift[(long long)n * ch.ab_nstride];
          ds.slope = ch.slope;
        }
      }
    } else {
      const int c = p.pslot[(long long)g * p.ppad + pl];
      if (c >= 0) {
        ds.valid = 1;
        ds.base = (gfloat_p)(p.xin + (((long long)n * p.P + c) * p.D + d) * plane);
      }
    }
    tab[pl] = ds;
  }

  // ---- epilogue operands, requested up front --------------------------------------------------------------------------------
  const int slot0 = g * OCG + wave * OPW;
  int qphys[MODE == 0 ? OPW : 1];
  float bqs[MODE == 0 ? OPW : 1];
  e2e_out_chan_t ocs[MODE != 0 ? OPW : 1];
#pragma unroll
  for (int a = 0; a < OPW; ++a) {
    if (MODE == 0) {
      qphys[a] = load_uniform(p.qslot + slot0 + a);
      bqs[a] = (p.bias != nullptr && qphys[a] >= 0) ? load_uniform(p.bias + qphys[a]) : 0.f;
    } else {
      ocs[a] = load_uniform(p.outs + slot0 + a);
    }
  }

  // ---- accumulators; an accumulating data gradient starts its outer accumulators from the old dx values -----------------------
  // Accumulators as aligned register PAIRS over adjacent output columns: a tap with an even column offset (kw = 0, 2) is one
  // v_pk_fma_f32 per pair (the weight is broadcast by op_sel, the two neighbourhood values are an aligned pair of the row's
  // ds_read registers), kw = 1 (odd offset: no aligned pair) stays two v_fma_f32: 48 instead of 72 VALU instructions per kernel.
  // A packed FMA costs ~4 SIMD cycles at any occupancy, a plain one 2.3 with two waves issuing but 4.3 when a wave issues alone
  // (tools/scratch/fma_rate.hip) -- which in this kernel (four waves per SIMD, each in an FMA burst a third of its life) is the
  // common case.
  f32x2_t accp[OPW][PH][PW / 2], acc2p[OPW][PH][PW / 2];
  const int oh0 = h0 + ly * PH, ow0 = w0 + lx * PW;
#pragma unroll
  for (int a = 0; a < OPW; ++a)
#pragma unroll
    for (int i = 0; i < PH; ++i)
#pragma unroll
      for (int j = 0; j < PW / 2; ++j) { accp[a][i][j] = f32x2_t{0.f, 0.f}; acc2p[a][i][j] = f32x2_t{0.f, 0.f}; }
  // data gradient: gradient of virtual-concat channel q at (shifted) depth d goes to depth d - s(q) of its source; the slices that
  // receive nothing are zero-filled by the workgroups of the slices that fall outside (conv133_kernel's rule).  Wave-uniform.
  // mode: 0 store, 1 accumulate (the old values are the initial outer accumulators), 2 zero fill, 3 nothing to do, 4 nothing to
  // store but the slice holds final values of earlier writers (an accumulating launch whose depth falls outside: dd = that slice)
  int dd_out = 0;
  auto destination = [&](const e2e_out_chan_t& oc, int& mode) -> float* {
    mode = 3;
    dd_out = 0;
    if (oc.ptr == nullptr) return nullptr;
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
    dd_out = dd;
    if (zero_fill && oc.accumulate) { mode = 4; return nullptr; }
    mode = zero_fill ? 2 : (oc.accumulate ? 1 : 0);
    return oc.ptr + (long long)n * oc.nstride + (long long)dd * plane;
  };
  if (MODE != 0) {
#pragma unroll
    for (int a = 0; a < OPW; ++a) {
      int mode;
      const float* xp = destination(ocs[a], mode);
      if (mode != 1) continue;
#pragma unroll
      for (int i = 0; i < PH; ++i) {
        const int oh = oh0 + i;
        if (oh < p.H && ow0 < p.W) {
          const float4 o = *reinterpret_cast<const float4*>(xp + (long long)oh * p.W + ow0);
          acc2p[a][i][0] = f32x2_t{o.x, o.y}; acc2p[a][i][1] = f32x2_t{o.z, o.w};
        }
      }
    }
  }

  // (no barrier for the plane table: it is first read after the barrier behind the first chunk's commit)

  const unsigned* qrow = p.quads + ((long long)g * NW + wave) * p.nchunks;
  const int* orow = p.woff + (long long)g * p.nchunks * NW + wave;
  unsigned m_cur = load_uniform(qrow);
  int o_cur = load_uniform(orow);
  commit(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA share of the first weight block has landed
  __syncthreads();
  SSTAMP(t_pro);
  SSTAMP_ADD(0, t_begin, t_pro);

  int until_flush = p.flush_every;
  for (int c = 0; c < p.nchunks; ++c) {
    const bool more = c + 1 < p.nchunks;
    const unsigned m_next = more ? load_uniform(qrow + c + 1) : 0u;
    const int o_next = more ? load_uniform(orow + (c + 1) * NW) : 0;
    SSTAMP(t2);
    request_begin(more ? c + 1 : c);
    // this wave's kernels of the chunk, one after the other (LDS byte address, wave-uniform)
    unsigned wq = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)(wl + (c & 1) * wstride + o_cur * WSLOT);
    const float* tp0 = lane_tp + (c & 1) * IMG;
    SSTAMP(t3);
#pragma unroll
    for (int cl = 0; cl < CK; ++cl) {
      if (cl < NUP) request_plane(cl, more);
      else if (cl < NUP + 2) request_weights(c + 1, cl - NUP, more);
#ifdef E2E_CONV_DEBUG
      const unsigned nib = (p.dbg & 1) ? 0u : (m_cur >> (cl * 4)) & 15u;
#else
      const unsigned nib = (m_cur >> (cl * 4)) & 15u;
#endif
      if (nib) {
        float nb[NR][NCL];
        const float* tp = tp0 + cl * CHS;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          // (the halo pair as its own ds_read_b64: merged with the next row's into a ds_read2_b64 -- what the compiler does with
          //  plain loads -- the two accesses are banked mod 32 and collide 2-way on top of twice the base cost)
          const float4 lo = *reinterpret_cast<const float4*>(tp + r * PITCH);
          const f32x2_t hi = *(lds_v2_p)(tp + r * PITCH + 4);
          nb[r][0] = lo.x; nb[r][1] = lo.y; nb[r][2] = lo.z; nb[r][3] = lo.w;
          nb[r][4] = hi[0]; nb[r][5] = hi[1];
        }
#pragma unroll
        for (int a = 0; a < OPW; ++a) {
          if (nib & (1u << a)) {
            const lds_f_p wp = (lds_f_p)(unsigned long long)wq;
            const f32x4_t w0v = *(lds_v4_p)wp;
            const f32x4_t w1v = *(lds_v4_p)(wp + 4);
            const float wk[9] = {w0v[0], w0v[1], w0v[2], w0v[3], w1v[0], w1v[1], w1v[2], w1v[3], wp[8]};
            wq += WSLOT * 4;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              // taps in the order kw = 0, 1, 2 (the summation order of every kernel since round 1): the even column offsets as
              // packed pairs, the odd one lane by lane
              auto pair_tap = [&](int kw) {
                const f32x2_t wv = {wk[kh * 3 + kw], wk[kh * 3 + kw]};
#pragma unroll
                for (int i = 0; i < PH; ++i)
#pragma unroll
                  for (int j = 0; j < PW / 2; ++j) {
                    const f32x2_t bv = {nb[i + kh][2 * j + kw], nb[i + kh][2 * j + kw + 1]};
                    accp[a][i][j] = __builtin_elementwise_fma(wv, bv, accp[a][i][j]);
                  }
              };
              pair_tap(0);
#pragma unroll
              for (int i = 0; i < PH; ++i)
#pragma unroll
                for (int j = 0; j < PW; ++j)
                  accp[a][i][j >> 1][j & 1] = fmaf(wk[kh * 3 + 1], nb[i + kh][j + 1], accp[a][i][j >> 1][j & 1]);
              pair_tap(2);
            }
          }
        }
      }
    }
    m_cur = m_next;
    o_cur = o_next;
    SSTAMP(t4);
    if (--until_flush == 0 || !more) {
      until_flush = p.flush_every;
#pragma unroll
      for (int a = 0; a < OPW; ++a)
#pragma unroll
        for (int i = 0; i < PH; ++i)
#pragma unroll
          for (int j = 0; j < PW / 2; ++j) { acc2p[a][i][j] += accp[a][i][j]; accp[a][i][j] = f32x2_t{0.f, 0.f}; }
    }
    SSTAMP(t5);
    if (more) {
      commit((c + 1) & 1);                                // into the other image: nobody reads it before the barrier below
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ... and this wave's share of the next weight block has landed
      SSTAMP(t6);
      __syncthreads();
      SSTAMP(t7);
      SSTAMP_ADD(1, t5, t6); SSTAMP_ADD(2, t6, t7);
    }
    SSTAMP_ADD(3, t2, t3); SSTAMP_ADD(4, t3, t4); SSTAMP_ADD(5, t4, t5);
  }
  SSTAMP(t_epi);

  // ---- epilogue --------------------------------------------------------------------------------------------------------------
  float acc2[OPW][PH][PW];
#pragma unroll
  for (int a = 0; a < OPW; ++a)
#pragma unroll
    for (int i = 0; i < PH; ++i)
#pragma unroll
      for (int j = 0; j < PW; ++j) acc2[a][i][j] = acc2p[a][i][j >> 1][j & 1];
  if (MODE == 0) {
    float psum[OPW];
    const long long out_n = (long long)p.Q * p.D * plane;
#pragma unroll
    for (int a = 0; a < OPW; ++a) {
      psum[a] = 0.f;
      if (qphys[a] < 0) continue;
      float* yp = p.y + (long long)n * out_n + ((long long)qphys[a] * p.D + d) * plane;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < PH; ++i) {
        const int oh = oh0 + i;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
          acc2[a][i][j] += bqs[a];
          if (oh < p.H && ow0 + j < p.W) s += acc2[a][i][j];
        }
#ifdef E2E_CONV_DEBUG
        if (p.dbg & 4) continue;
#endif
        if (oh < p.H && ow0 < p.W)
          *reinterpret_cast<float4*>(yp + (long long)oh * p.W + ow0) = make_float4(acc2[a][i][0], acc2[a][i][1], acc2[a][i][2], acc2[a][i][3]);
      }
      psum[a] = s;
    }
    if (p.part != nullptr) {
      const int vr = p.H - h0 < TH ? p.H - h0 : TH, vc = p.W - w0 < TW ? p.W - w0 : TW;
      const float tcnt = (float)(vr * vc);
      float mean[OPW], m2[OPW];
#pragma unroll
      for (int a = 0; a < OPW; ++a) mean[a] = e2e::wave_sum_dpp(psum[a]) / tcnt;
#pragma unroll
      for (int a = 0; a < OPW; ++a) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < PH; ++i)
#pragma unroll
          for (int j = 0; j < PW; ++j)
            if (oh0 + i < p.H && ow0 + j < p.W) {
              const float dlt = acc2[a][i][j] - mean[a];
              t = fmaf(dlt, dlt, t);
            }
        m2[a] = t;
      }
#pragma unroll
      for (int a = 0; a < OPW; ++a) m2[a] = e2e::wave_sum_dpp(m2[a]);
      if (lane < OPW) {
        float mm = mean[0], vv = m2[0];
        int qq = qphys[0];
#pragma unroll
        for (int a = 1; a < OPW; ++a)
          if (lane == a) { mm = mean[a]; vv = m2[a]; qq = qphys[a]; }
        if (qq >= 0) {
          double* pp = p.part + (((long long)n * p.Q + qq) * p.tiles_per_n + tile_in_n) * 3;
          pp[0] = (double)tcnt;
          pp[1] = (double)mm;
          pp[2] = (double)vv;
        }
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < OPW; ++a) {
      int mode;
      float* xp = destination(ocs[a], mode);
      if (mode == 3) continue;
      const int dd = dd_out;
      if (mode != 4) {
#pragma unroll
        for (int i = 0; i < PH; ++i) {
          const int oh = oh0 + i;
          if (oh >= p.H || ow0 >= p.W) continue;
          float4 val = make_float4(acc2[a][i][0], acc2[a][i][1], acc2[a][i][2], acc2[a][i][3]);
          if (mode == 2) val = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(xp + (long long)oh * p.W + ow0) = val;
        }
      }
      // ---- fused first pass of the InstanceNorm + LeakyReLU backward (instnorm.hip: in_bwd_reduce_kernel) for channels whose
      // gradient buffer this launch writes last: the values just stored (or, in an untouched slice, found) are final.  One
      // record per (channel, destination slice, tile), written by exactly one wave of the launch.
      if (p.insum == nullptr) continue;
      const e2e_in_sum_chan_t sc = load_uniform(p.insum + slot0 + a);
      if (sc.y == nullptr) continue;
      double* rec = sc.part + (long long)n * sc.part_nstride + (((long long)dd * p.tiles_y + ty) * p.tiles_x + tx) * 2;
      if (mode == 2) {                                      // a zero-filled slice adds nothing
        if (lane == 0) { rec[0] = 0.0; rec[1] = 0.0; }
        continue;
      }
      const long long ci = (long long)n * sc.ab_nstride;
      const float ca = load_uniform(sc.scale + ci), cb = load_uniform(sc.shift + ci);
      const float mu = load_uniform(sc.mean + ci), rs = load_uniform(sc.rstd + ci);
      const float* yp = sc.y + (long long)n * sc.nstride + (long long)dd * plane;
      const float* op = ocs[a].ptr + (long long)n * ocs[a].nstride + (long long)dd * plane;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < PH; ++i) {
        const int oh = oh0 + i;
        if (oh >= p.H || ow0 >= p.W) continue;
        const float4 yv = *reinterpret_cast<const float4*>(yp + (long long)oh * p.W + ow0);
        float4 dzv = make_float4(acc2[a][i][0], acc2[a][i][1], acc2[a][i][2], acc2[a][i][3]);
        if (mode == 4) dzv = *reinterpret_cast<const float4*>(op + (long long)oh * p.W + ow0);
        const float ys[4] = {yv.x, yv.y, yv.z, yv.w}, dz[4] = {dzv.x, dzv.y, dzv.z, dzv.w};
#pragma unroll
        for (int j = 0; j < PW; ++j) {
          const float u = fmaf(ca, ys[j], cb);
          const float du = u > 0.f ? dz[j] : dz[j] * sc.slope;
          s1 += du;
          s2 = fmaf(du, (ys[j] - mu) * rs, s2);
        }
      }
      s1 = e2e::wave_sum_dpp(s1);
      s2 = e2e::wave_sum_dpp(s2);
      if (lane == 0) { rec[0] = (double)s1; rec[1] = (double)s2; }
    }
  }
#ifdef E2E_CONV_DEBUG
  SSTAMP(t_end);
  SSTAMP_ADD(6, t_epi, t_end);
  st_acc[7] = 1;
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_sparse_stamps[(blockIdx.x & 1023) * 8 + i], st_acc[i]);
#endif
}

// ---- weight packing: one launch for a table of (layer, direction) jobs --------------------------------------------------------
// thread = (group, chunk, output slot, plane slot, tap); a live kernel's 12-float slot lands at woff[wave] + its rank in the wave's
// walk order (ascending bit index of the liveness word); pruned kernels are not stored at all
__global__ __launch_bounds__(256) void sparse_pack_kernel(const e2e_sparse_pack_job_t* __restrict__ jobs) {
  const e2e_sparse_pack_job_t jb = jobs[blockIdx.y];
  const long long total = (long long)jb.groups * jb.nchunks * WCHUNK;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int t = (int)(idx % WSLOT), cl = (int)((idx / WSLOT) % CK), ql = (int)((idx / (WSLOT * CK)) % OCG);
    const long long gc = idx / WCHUNK;
    const int c = (int)(gc % jb.nchunks), g = (int)(gc / jb.nchunks);
    const int wave = ql >> 2, bit = cl * 4 + (ql & 3);
    const unsigned word = jb.quads[((long long)g * NW + wave) * jb.nchunks + c];
    if (!((word >> bit) & 1u)) continue;
    const int q = jb.qslot[g * OCG + ql];
    const int pp = jb.pslot[((long long)g * jb.nchunks + c) * CK + cl];
    const int slot = jb.woff[(gc) * NW + wave] + __popc(word & ((1u << bit) - 1u));
    float v = 0.f;
    if (t < 9 && q >= 0 && pp >= 0) v = jb.w[(long long)q * jb.wq_stride + (long long)pp * jb.wp_stride + (jb.reverse ? 8 - t : t)];
    jb.wpk[(gc * jb.kmax + slot) * WSLOT + t] = v;
  }
}

inline bool sparse_knob() {
  static const int v = getenv("E2E_CONV_SPARSE2") ? atoi(getenv("E2E_CONV_SPARSE2")) : 1;
  return v != 0;
}

int sparse_launch(int mode, SparseParams p, hipStream_t st) {
  p.tiles_x = e2e::cdiv(p.W, TW);
  p.tiles_y = e2e::cdiv(p.H, TH);
  p.tiles_per_n = p.D * p.tiles_y * p.tiles_x;
  p.groups = e2e::cdiv(p.Q, OCG);
  p.nchunks = e2e::cdiv(p.P, CK);
  p.ppad = p.nchunks * CK;
  p.total = p.B * p.tiles_per_n * p.groups;
  p.padded_total = (p.total + 7) & ~7;
  if (p.flush_every < 1) p.flush_every = 1;
#ifdef E2E_CONV_DEBUG
  static const int dbg_knob = getenv("E2E_CONV_DBG") ? atoi(getenv("E2E_CONV_DBG")) : 0;
  p.dbg = dbg_knob;
#endif
  p.uw = e2e::cdiv(3 * p.kmax, NW);
  size_t dyn = (size_t)2 * p.kmax * WSLOT * 4 + (size_t)p.ppad * sizeof(PlaneDesc);
  // static 55 KB + dynamic: beyond the default 64 KB for dense maps / many planes.  Set before every launch of the mode (cheap): a
  // process-wide "already set" flag is a data race under concurrent callers and leaves other devices of the process without it
  E2E_REQUIRE(hipFuncSetAttribute(mode == 0 ? (const void*)conv133_sparse_kernel<0> : (const void*)conv133_sparse_kernel<1>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) == hipSuccess,
              "conv133_sparse: cannot raise the dynamic LDS limit");
  E2E_REQUIRE(dyn <= 100 * 1024 && p.uw <= 128, "conv133_sparse: %zu bytes of dynamic LDS (kmax %d, %d planes) not served", dyn, p.kmax, p.ppad);
  e2e::note_kernel("conv133_sparse_kernel<mode=%d> wgs=%d groups=%d chunks=%d flush=%d kmax=%d", mode, p.padded_total, p.groups, p.nchunks, p.flush_every, p.kmax);
  if (mode == 0) hipLaunchKernelGGL((conv133_sparse_kernel<0>), dim3(p.padded_total), dim3(NW * 64), dyn, st, p);
  else hipLaunchKernelGGL((conv133_sparse_kernel<1>), dim3(p.padded_total), dim3(NW * 64), dyn, st, p);
#ifdef E2E_CONV_DEBUG
  if (p.dbg & 8) {
    (void)hipStreamSynchronize(st);
    static unsigned long long hh[1024 * 8], zz[1024 * 8];
    (void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_sparse_stamps), sizeof(hh));
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 1024 * 8; ++i) h[i & 7] += hh[i];
    const double w = h[7] ? (double)h[7] : 1.0;
    fprintf(stderr, "[conv133_sparse MODE %d P %d Q %d] per-wave cycles: pro %.0f commit %.0f barrier %.0f prefetch %.0f walk %.0f flush %.0f epi %.0f (waves %.0f)\n",
            mode, p.P, p.Q, h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w, w);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sparse_stamps), zz, sizeof(zz));
  }
#endif
  return e2e::check_launch("conv133_sparse_kernel");
}

}  // namespace

extern "C" int e2e_conv133_sparse_eligible(int Cin, int Cout, int Di, int Hi, int Wi, int sd, int sh, int sw) {
  if (!sparse_knob()) return 0;
  if (sd != 1 || sh != 1 || sw != 1) return 0;
  if (Wi % 4 != 0 || Hi <= 16 || Wi <= 16 || Di < 1) return 0;      // the 16 x 32 tile class of conv133_kernel, float4 staging
  if (Cin <= CK || Cout <= CK) return 0;                             // both directions need more than one chunk
  if ((long long)e2e::cdiv(Cin > Cout ? Cin : Cout, CK) * CK * (long long)sizeof(PlaneDesc) > 40 * 1024) return 0;
  return 1;
}

extern "C" long long e2e_conv133_sparse_wpk_floats(int P, int Q, int kmax) {
  return (long long)e2e::cdiv(Q, OCG) * e2e::cdiv(P, CK) * kmax * WSLOT;
}

extern "C" int e2e_conv133_sparse_plan(const unsigned char* kmask, int R, int Cc, int transpose, int* qslot, int* pslot,
                                       unsigned* quads, int* woff, int* kmax, int* flush_every) {
  E2E_REQUIRE(kmask && qslot && pslot && quads && woff && kmax && R > 0 && Cc > 0, "conv133_sparse_plan: bad arguments");
  const int Q = transpose ? Cc : R, P = transpose ? R : Cc;
  const int groups = e2e::cdiv(Q, OCG), nchunks = e2e::cdiv(P, CK);
  const int NWG = groups * NW;                           // waves of all groups: they share ONE chunking of the input planes
  auto alive = [&](int q, int pp) -> unsigned char { return transpose ? kmask[(long long)pp * Cc + q] : kmask[(long long)q * Cc + pp]; };
  long long nlive = 0;
  // -- output planes to waves, per 32-plane group: longest-processing-time first on the planes' live counts
  for (int g = 0; g < groups; ++g) {
    const int q0 = g * OCG, nq = std::min(OCG, Q - q0);
    std::vector<int> cnt(nq, 0), order(nq);
    for (int i = 0; i < nq; ++i) {
      for (int pp = 0; pp < P; ++pp) cnt[i] += alive(q0 + i, pp);
      order[i] = i;
      nlive += cnt[i];
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });
    int wload[NW] = {0}, wfill[NW] = {0};
    int* qs = qslot + g * OCG;
    for (int s = 0; s < OCG; ++s) qs[s] = -1;
    for (int i : order) {
      int best = -1;
      for (int w = 0; w < NW; ++w)
        if (wfill[w] < OPW && (best < 0 || wload[w] < wload[best])) best = w;
      qs[best * OPW + wfill[best]++] = q0 + i;
      wload[best] += cnt[i];
    }
  }
  // -- input planes to chunks.  The groups of a layer are separate workgroups that run side by side on the same tile: with one
  // common plane order they request the same planes at the same time and share them through L2 (per-group orders were measured:
  // 160 -> 64 @64^3 forward 0.387 -> 0.424 ms, the input was fetched once per group).  Cost of a (chunk, wave) cell: live
  // kernels + 0.35 per visited plane; objective: sum over chunks and groups of the slowest wave of the group.
  std::vector<float> cell((size_t)NWG * P, 0.f);          // [wave of any group][plane]
  for (int g = 0; g < groups; ++g)
    for (int w = 0; w < NW; ++w)
      for (int pp = 0; pp < P; ++pp) {
        int k = 0;
        for (int a = 0; a < OPW; ++a) {
          const int q = qslot[g * OCG + w * OPW + a];
          if (q >= 0) k += alive(q, pp);
        }
        cell[(size_t)(g * NW + w) * P + pp] = k ? (float)k + 0.35f : 0.f;
      }
  std::vector<int> chunk_of(P, -1), fill(nchunks, 0);
  std::vector<float> load((size_t)nchunks * NWG, 0.f);
  auto chunk_cost = [&](int c) {
    float tot = 0.f;
    for (int g = 0; g < groups; ++g) {
      float m = 0.f;
      for (int w = 0; w < NW; ++w) m = std::max(m, load[(size_t)c * NWG + g * NW + w]);
      tot += m;
    }
    return tot;
  };
  auto add = [&](int c, int pp, float sign) {
    for (int w = 0; w < NWG; ++w) load[(size_t)c * NWG + w] += sign * cell[(size_t)w * P + pp];
  };
  std::vector<int> porder(P);
  std::vector<float> pw(P, 0.f);
  for (int pp = 0; pp < P; ++pp) {
    for (int w = 0; w < NWG; ++w) pw[pp] += cell[(size_t)w * P + pp];
    porder[pp] = pp;
  }
  std::stable_sort(porder.begin(), porder.end(), [&](int a, int b) { return pw[a] > pw[b]; });
  for (int pp : porder) {                                  // greedy: the chunk whose cost grows least (ties: the lighter chunk)
    int best = -1;
    float bcost = 0.f, bsum = 0.f;
    for (int c = 0; c < nchunks; ++c) {
      if (fill[c] >= CK) continue;
      const float before = chunk_cost(c);
      add(c, pp, 1.f);
      const float grow = chunk_cost(c) - before;
      float sm = 0.f;
      for (int w = 0; w < NWG; ++w) sm += load[(size_t)c * NWG + w];
      add(c, pp, -1.f);
      if (best < 0 || grow < bcost || (grow == bcost && sm < bsum)) { best = c; bcost = grow; bsum = sm; }
    }
    chunk_of[pp] = best;
    fill[best]++;
    add(best, pp, 1.f);
  }
  // pair swaps between chunks (fixed pseudo-random sequence: the plan is a pure function of the kernel map)
  unsigned long long rng = 0x9e3779b97f4a7c15ull ^ ((unsigned long long)P * 1315423911ull + (unsigned long long)Q);
  auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng >> 33); };
  const int tries = nchunks > 1 ? 400 * nchunks : 0;
  for (int t = 0; t < tries; ++t) {
    const int p1 = (int)(next() % (unsigned)P), p2 = (int)(next() % (unsigned)P);
    const int c1 = chunk_of[p1], c2 = chunk_of[p2];
    if (c1 == c2) continue;
    const float before = chunk_cost(c1) + chunk_cost(c2);
    add(c1, p1, -1.f); add(c2, p2, -1.f); add(c1, p2, 1.f); add(c2, p1, 1.f);
    if (chunk_cost(c1) + chunk_cost(c2) < before) { chunk_of[p1] = c2; chunk_of[p2] = c1; }
    else { add(c1, p2, -1.f); add(c2, p1, -1.f); add(c1, p1, 1.f); add(c2, p2, 1.f); }
  }
  // -- emit: planes of a chunk in ascending order, the same table for every group; liveness words per (group, wave, chunk)
  std::vector<int> ps0((size_t)nchunks * CK, -1), at(nchunks, 0);
  for (int pp = 0; pp < P; ++pp) { const int c = chunk_of[pp]; ps0[(size_t)c * CK + at[c]++] = pp; }
  for (int g = 0; g < groups; ++g) {
    int* ps = pslot + (long long)g * nchunks * CK;
    for (int i = 0; i < nchunks * CK; ++i) ps[i] = ps0[i];
    for (int w = 0; w < NW; ++w)
      for (int c = 0; c < nchunks; ++c) {
        unsigned word = 0u;
        for (int cl = 0; cl < CK; ++cl) {
          const int pp = ps0[(size_t)c * CK + cl];
          if (pp < 0) continue;
          for (int a = 0; a < OPW; ++a) {
            const int q = qslot[g * OCG + w * OPW + a];
            if (q >= 0 && alive(q, pp)) word |= 1u << (cl * 4 + a);
          }
        }
        quads[((long long)g * NW + w) * nchunks + c] = word;
      }
  }
  // -- the live kernels of a chunk are stored wave by wave: first slot of every wave's list, and the largest chunk (block size)
  int km = 1;
  for (int g = 0; g < groups; ++g)
    for (int c = 0; c < nchunks; ++c) {
      int at_slot = 0;
      for (int w = 0; w < NW; ++w) {
        woff[((long long)g * nchunks + c) * NW + w] = at_slot;
        at_slot += __builtin_popcount(quads[((long long)g * NW + w) * nchunks + c]);
      }
      km = std::max(km, at_slot);
    }
  *kmax = km;
  if (flush_every != nullptr) {
    // flush the chunk accumulators into the outer ones once a chain holds ~18 products: measured against fp64
    // (tools/scratch/sparse_err.py, 64 -> 32 at density 0.2) chains of ~58 products left the output 1.9e-7 rms from exact, ~29
    // products 1.5e-7, the torch-CPU conv 1.5e-7, a flush after every chunk 1.3e-7; the engine is to stay below the CPU path's noise
    // (and the flushes cost nothing measurable: 0.777 / 0.763 ms at an interval of 4 / 2 chunks)
    const double per_plane_chunk = Q > 0 ? (double)nlive / ((double)Q * nchunks) : (double)CK;   // live kernels per output plane and chunk
    int f = per_plane_chunk > 0.0 ? (int)(2.0 / per_plane_chunk) : nchunks;
    *flush_every = f < 1 ? 1 : (f > nchunks ? nchunks : f);
  }
  return E2E_OK;
}

extern "C" int e2e_conv133_sparse_pack(const e2e_sparse_pack_job_t* jobs, int njobs, long long max_floats, void* stream) {
  E2E_REQUIRE(jobs && njobs > 0 && max_floats > 0, "conv133_sparse_pack: bad arguments");
  long long blocks = e2e::cdivll(max_floats, 256 * 4);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sparse_pack_kernel, dim3((unsigned)blocks, njobs), dim3(256), 0, (hipStream_t)stream, jobs);
  return e2e::check_launch("sparse_pack_kernel");
}

extern "C" int e2e_conv133_fwd_sparse(const e2e_in_chan_t* chans_plan, int Cin, const float* wpk, const float* bias, const unsigned* quads,
                                      const int* woff, int kmax, const int* qslot, int flush_every, float* y, double* part, int B,
                                      int Cout, int Di, int Hi, int Wi, void* stream) {
  E2E_REQUIRE(chans_plan && wpk && quads && woff && kmax > 0 && qslot && y, "conv133_fwd_sparse: bad arguments");
  E2E_REQUIRE(e2e_conv133_sparse_eligible(Cin, Cout, Di, Hi, Wi, 1, 1, 1), "conv133_fwd_sparse: shape not served");
  SparseParams p{};
  p.chans = chans_plan; p.wpk = wpk; p.bias = bias; p.quads = quads; p.woff = woff; p.kmax = kmax; p.qslot = qslot; p.y = y; p.part = part;
  p.P = Cin; p.Q = Cout; p.B = B; p.D = Di; p.H = Hi; p.W = Wi; p.flush_every = flush_every;
  return sparse_launch(0, p, (hipStream_t)stream);
}

extern "C" int e2e_conv133_dgrad_sparse(const float* dy, const float* wpk_t, const unsigned* quads_t, const int* woff_t, int kmax_t,
                                        const int* pslot_t, const e2e_out_chan_t* outs_plan, const e2e_in_sum_chan_t* insum_plan,
                                        int flush_every, int B, int Cin, int Cout, int Di, int Hi, int Wi, void* stream) {
  E2E_REQUIRE(dy && wpk_t && quads_t && woff_t && kmax_t > 0 && pslot_t && outs_plan, "conv133_dgrad_sparse: bad arguments");
  E2E_REQUIRE(e2e_conv133_sparse_eligible(Cin, Cout, Di, Hi, Wi, 1, 1, 1), "conv133_dgrad_sparse: shape not served");
  SparseParams p{};
  p.xin = dy; p.wpk = wpk_t; p.quads = quads_t; p.woff = woff_t; p.kmax = kmax_t; p.pslot = pslot_t; p.outs = outs_plan; p.insum = insum_plan;
  p.P = Cout; p.Q = Cin; p.B = B; p.D = Di; p.H = Hi; p.W = Wi; p.flush_every = flush_every;
  return sparse_launch(1, p, (hipStream_t)stream);
}
