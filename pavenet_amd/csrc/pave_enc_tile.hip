// pave_enc_tile.hip -- encoder multi-scale deformable attention (T = 1, M = 8, D = 32, L = 4,
// P = 4) with the sampled value rows staged in LDS.  gfx950 (CDNA4, wave64) only.
//
// Replaces, per encoder layer, softmax + sampling-location arithmetic + ms_deformable_im2col
// (third_party/mmcv/mmcv/ops/multi_scale_deform_attn.py:373-404,
//  .../csrc/common/cuda/ms_deform_attn_cuda_kernel.cuh:200-254).
//
// Why LDS: one (query, head) gathers 16 points x 4 corners x 128 B.  Straight from global memory
// that is 1.46 GB of 128-byte requests per frame-layer through the texture-address path
// (64 B / clk / CU): a 1.04 ms floor for the 28-frame bench launch however well the caches hit.
// Neighbouring queries sample neighbouring pixels, so a workgroup takes an 8 x 8-pixel image
// TILE (its 64 level-0 queries and the 16 + 4 + 1 queries of the coarser levels that sit on the
// same image region) of ONE head, copies the four level windows around the tile into LDS with
// LDS-DMA (global_load_lds_dwordx4, per-lane source rows, no VGPR round trip) and gathers from
// LDS at 256 B / clk / CU.  Corners that fall outside the window (large offsets) are fetched from
// global memory in a second, predicated pass -- results never depend on the window size.
//
// Work layout: 4 lanes x 8 channels per (query, head) "pair", 16 pairs per wave.  Lane k of a
// pair prepares the 16 corner descriptors (weight x attention weight, LDS row address) of LEVEL
// k's four points and keeps them in VGPRs; the gather loop broadcasts them inside the quad with
// DPP quad_perm operands fused into the consuming v_add / v_fma (no LDS traffic, no extra
// instruction for the address).  LDS bank conflicts are designed out: a ds_read_b128 is served
// per 16-lane group = 4 pairs x 64 B; each pair has a (row-parity, chunk-set) role so that the
// four 64-byte pieces always fall on the four bank quarters: the two x-neighbours of a bilinear
// footprint sit in rows of opposite parity, and a pair with parity role 1 simply visits them in
// the other order.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "pave_hip.h"
#include "pave_internal.h"
#include "pave_enc_math.h"

namespace {

constexpr int kHeads = 8;
constexpr int kRowBytes = 1024;  // one token: 8 heads x 32 channels fp32

struct TileParams {
  const float* value;  // [F, S, 8, 32]
  const float* proj;   // [F*S, proj_stride]: offsets [8][4][4][2], then logits [8][4][4]
  const float* ref;    // [F*S, 4, 2]
  float* out;          // [F*S, 256]
  int S;
  int proj_stride;
  int out_f16;         // out is fp16 [F*S, 256] (fp16 operand mode: the rows only feed output_proj's MFMA)
  int nx, ny;          // tiles per frame
  int n_blocks;
  int Hs[4], Ws[4], St[4];
  // per-level constants each lane fetches by its level k = lane & 3 (one 32-byte kernarg row):
  // H, W, first token, window side, first LDS row of the window, 1/W, 1/H (float bits), unused
  int tab[4][8];
  // window origin shift per (head, level): (dx, dy) in level pixels.  A head's sampling points sit
  // around reference + its learnt mean offset (the reference initialises them on a ray, 1..4 px
  // out, MO:227-240), so the LDS window of (tile, head) is centred there instead of on the tile.
  int shift[kHeads][4][2];
};

__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int per = nb >> 3, rem = nb & 7;
  const int x = b & 7, idx = b >> 3;
  return x * per + min(x, rem) + idx;
}

// quad broadcast of lane Q's value (VOP DPP quad_perm:[Q,Q,Q,Q]); hipcc folds it into the user
template <int Q>
__device__ __forceinline__ int qbi(int x) {
  return __builtin_amdgcn_update_dpp(0, x, Q * 0x55, 0xf, 0xf, true);
}
template <int Q>
__device__ __forceinline__ float qbf(float x) {
  return __builtin_bit_cast(float, qbi<Q>(__builtin_bit_cast(int, x)));
}
using pave_enc::quad_max;
using pave_enc::quad_sum;
using pave_enc::div_by;

template <int W0, int W1, int W2, int W3>
struct WinGeom {
  static constexpr int B0 = 0, B1 = W0 * W0, B2 = B1 + W1 * W1, B3 = B2 + W2 * W2;
  static constexpr int kRows = B3 + W3 * W3;
  // two all-zero rows behind the windows: kZ is even, kZ + 1 odd
  static constexpr int kZ = (kRows + 1) & ~1;
  static constexpr int kLdsBytes = (kZ + 2) * 128;
};

// LDS-DMA of one level's window, one window row (<= 16 pixels) at a time: 8 pixels x 128 B per
// wave instruction, lane = (pixel, 16-byte chunk).  Window pixels beyond the map are filled from
// the clamped neighbour (their corners always carry weight 0).  `voff_lo / voff_hi`: per-lane
// byte offset of pixel ox + (lane >> 3) (+ 8) inside a map row, incl. the lane's chunk.
template <int WW, int NW>
__device__ __forceinline__ void stage_level(char* lds_level, const char* vhead, int wave, int lane,
                                            int ox, int oy, int H, int W, int st) {
  static_assert(WW <= 16, "one window row = at most two DMA instructions");
  const int px_lo = lane >> 3, px_hi = px_lo + 8;
  const unsigned cofs = (lane & 7) * 16;
  const unsigned xlo = (unsigned)min(max(ox + px_lo, 0), W - 1) * kRowBytes + cofs;
  const unsigned xhi = (unsigned)min(max(ox + px_hi, 0), W - 1) * kRowBytes + cofs;
  for (int wy = wave; wy < WW; wy += NW) {
    const int y = min(max(oy + wy, 0), H - 1);                 // scalar
    const unsigned rowb = (unsigned)(st + y * W) * kRowBytes;  // scalar
    char* dst = lds_level + wy * (WW * 128);
    if (px_lo < WW)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(vhead + (rowb + xlo)),
          (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    if (WW > 8 && px_hi < WW)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(vhead + (rowb + xhi)),
          (__attribute__((address_space(3))) void*)(dst + 1024), 16, 0, 0);
  }
}

// Corner descriptors of one bilinear point, in VISITING order: slots 0, 1 = the two x-neighbours
// of the upper row, slots 2, 3 = of the lower row; inside a row the pair visits first the pixel
// whose LDS row parity equals its parity role.
struct PointDesc {
  float w[4];  // bilinear weight x attention weight; 0 for a corner outside the map
  int a[4];    // LDS byte address of the 128-byte row (a zero row if the point is not in LDS)
  int tok00;   // token of the footprint's upper-left pixel (second pass)
};

// returns the point's 3 flag bits: 1 = footprint not (entirely) inside the window -> second
// pass, 2 = upper row visited right-to-left, 4 = lower row visited right-to-left
__device__ __forceinline__ int make_point(PointDesc& d, float px, float py, float aw, int H, int W,
                                          int st, int ox, int oy, int WW, int wbase, int zrow,
                                          int par) {
  // NaN / Inf-safe clamp: fmaxf / fminf return the non-NaN operand, so a NaN position becomes
  // -2 (every corner outside the map), exactly what the reference's range test yields
  px = fminf(fmaxf(px, -2.f), (float)W + 1.f);
  py = fminf(fmaxf(py, -2.f), (float)H + 1.f);
  const float fx = floorf(px), fy = floorf(py);
  const int x0 = (int)fx, y0 = (int)fy;
  float lx = px - fx, ly = py - fy;
  float hx = 1.f - lx, hy = 1.f - ly;
  // corner validity == the reference's (h_im > -1 && w_im > -1 && h_im < H && w_im < W) plus
  // its per-corner range tests (ms_deform_attn_cuda_kernel.cuh:36-60)
  hx = (unsigned)x0 < (unsigned)W ? hx : 0.f;
  lx = (unsigned)(x0 + 1) < (unsigned)W ? lx : 0.f;
  hy = (unsigned)y0 < (unsigned)H ? hy * aw : 0.f;
  ly = (unsigned)(y0 + 1) < (unsigned)H ? ly * aw : 0.f;
  const int dx0 = x0 - ox, dy0 = y0 - oy;
  const bool inwin = (unsigned)dx0 < (unsigned)(WW - 1) && (unsigned)dy0 < (unsigned)(WW - 1);
  // outside the window: the zero rows (zrow even, zrow + 1 odd, "window width" 0)
  const int r00 = inwin ? __mul24(dy0, WW) + (dx0 + wbase) : zrow;
  const int wws = inwin ? WW : 0;
  const int e = (r00 ^ par) & 1, eb = e ^ (wws & 1);
  const float xf = e ? lx : hx, xs = e ? hx : lx;
  const float xfb = eb ? lx : hx, xsb = eb ? hx : lx;
  d.w[0] = hy * xf;
  d.w[1] = hy * xs;
  d.w[2] = ly * xfb;
  d.w[3] = ly * xsb;
  const int r10 = r00 + wws;
  d.a[0] = (r00 + e) << 7;
  d.a[1] = (r00 + 1 - e) << 7;
  d.a[2] = (r10 + eb) << 7;
  d.a[3] = (r10 + 1 - eb) << 7;
  d.tok00 = __mul24(y0, W) + (x0 + st);
  return (inwin ? 0 : 1) | (e << 1) | (eb << 2);
}

__device__ __forceinline__ void fma4(float4& acc, float w, const float4& v) {
  acc.x = fmaf(w, v.x, acc.x);
  acc.y = fmaf(w, v.y, acc.y);
  acc.z = fmaf(w, v.z, acc.z);
  acc.w = fmaf(w, v.w, acc.w);
}

// the 16 corners of level LVL: descriptors come from lane LVL of each quad
template <int LVL>
__device__ __forceinline__ void gather_level(float4& accA, float4& accB, const char* lds,
                                             const PointDesc (&d)[4], int offA, int offB) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float w = qbf<LVL>(d[i].w[c]);
      const int a = qbi<LVL>(d[i].a[c]);
      const float4 vA = *reinterpret_cast<const float4*>(lds + (a + offA));
      const float4 vB = *reinterpret_cast<const float4*>(lds + (a + offB));
      fma4(accA, w, vA);
      fma4(accB, w, vB);
    }
  }
}

// second pass: points whose footprint is not in the LDS window, straight from global memory.
// Each quad (= one (query, head) pair) walks ITS OWN far points -- a 16-bit mask, lane k's four
// points at bits 4k .. 4k + 3, lowest first, i.e. in (level, point) order -- so a wave runs
// max-over-its-pairs rounds instead of one round per (level, point) that ANY pair needs.  The
// round's descriptor (token, 4 weights, visiting-order bits) is selected by point in the owning
// lane and broadcast inside the quad with ds_bpermute (the source lane is data dependent).
__device__ __forceinline__ int sel4i(int x0, int x1, int x2, int x3, int q) {
  const int lo = (q & 1) ? x1 : x0, hi = (q & 1) ? x3 : x2;
  return (q & 2) ? hi : lo;
}
__device__ __forceinline__ float sel4f(float x0, float x1, float x2, float x3, int q) {
  const float lo = (q & 1) ? x1 : x0, hi = (q & 1) ? x3 : x2;
  return (q & 2) ? hi : lo;
}
// one far point of the quad: descriptor of entry j (lane j >> 2, point j & 3) broadcast, the 8 loads
struct FarPoint {
  float w[4];
  float4 a[4], b[4];
};
__device__ __forceinline__ void far_fetch(FarPoint& f, bool act, int j, const char* vlane,
                                          const PointDesc (&d)[4], int flags, int quad_base, int W0,
                                          int W1, int W2, int W3, int S) {
  const int src = j >> 2, pt = j & 3;
  const int addr = quad_base + (src << 2);
  // the owning lane's descriptor of point pt (every lane selects from its own, lane src's counts)
  const int t00 = __builtin_amdgcn_ds_bpermute(
      addr, sel4i(d[0].tok00, d[1].tok00, d[2].tok00, d[3].tok00, pt));
  const int fb = __builtin_amdgcn_ds_bpermute(addr, (flags >> (3 * pt)) & 7);
#pragma unroll
  for (int c = 0; c < 4; ++c)
    f.w[c] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(
                 addr, __builtin_bit_cast(int, sel4f(d[0].w[c], d[1].w[c], d[2].w[c], d[3].w[c], pt))));
  const int W = sel4i(W0, W1, W2, W3, src);   // lane src <-> level src
  if (act) {
    const int e = (fb >> 1) & 1, eb = (fb >> 2) & 1;
    // a corner outside the map has weight 0 and may have any token: clamp, never mask by value
    const int t0 = min(max(t00 + e, 0), S - 1), t1 = min(max(t00 + 1 - e, 0), S - 1);
    const int t2 = min(max(t00 + W + eb, 0), S - 1), t3 = min(max(t00 + W + 1 - eb, 0), S - 1);
    const int t[4] = {t0, t1, t2, t3};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4* r = reinterpret_cast<const float4*>(vlane + (size_t)t[c] * kRowBytes);
      f.a[c] = r[0];
      f.b[c] = r[1];
    }
  }
}
__device__ __forceinline__ void far_accumulate(float4& accA, float4& accB, const FarPoint& f) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    fma4(accA, f.w[c], f.a[c]);
    fma4(accB, f.w[c], f.b[c]);
  }
}

template <int PER_ROUND>
__device__ __forceinline__ void far_quad_loop(float4& accA, float4& accB, const char* vlane,
                                              const PointDesc (&d)[4], int flags, int lane,
                                              int W0, int W1, int W2, int W3, int S) {
  const int fm = (flags & 1) | ((flags >> 2) & 2) | ((flags >> 4) & 4) | ((flags >> 6) & 8);
  int M = qbi<0>(fm) | (qbi<1>(fm) << 4) | (qbi<2>(fm) << 8) | (qbi<3>(fm) << 12);
  const int quad_base = (lane & ~3) << 2;   // byte address of the quad's lane 0 for ds_bpermute
  while (__builtin_amdgcn_ballot_w64(M != 0) != 0ull) {
    FarPoint f[PER_ROUND];
    bool act[PER_ROUND];
#pragma unroll
    for (int u = 0; u < PER_ROUND; ++u) {   // all loads of the round in flight before the first FMA
      act[u] = M != 0;
      const int j = act[u] ? __builtin_ctz(M) : 0;
      M &= M - 1;
      far_fetch(f[u], act[u], j, vlane, d, flags, quad_base, W0, W1, W2, W3, S);
    }
#pragma unroll
    for (int u = 0; u < PER_ROUND; ++u)     // (level, point) order within the pair, as always
      if (act[u]) far_accumulate(accA, accB, f[u]);
  }
}

// TILE = 8: 64 + 16 + 4 + 1 = 85 queries -> 96 pair slots = 6 waves of 16 pairs
// ABL: timing-only ablations for tools/ (1: no window staging, 2: no gather loop); 0 in the product
constexpr int FAR_PER_ROUND = 1;   // far points of a pair fetched per round of the second pass (2: spills at 96 VGPRs, 2-3x slower)

template <int W0, int W1, int W2, int W3, int MB0, int MB1, int MB2, int MB3, int WPE, int ABL = 0,
          bool PREP = false>
__global__ __launch_bounds__(384, WPE) void enc_tile_kernel(const TileParams p) {
  using G = WinGeom<W0, W1, W2, W3>;
  constexpr int kWaves = 6;
  __shared__ __attribute__((aligned(1024))) char lds[G::kLdsBytes];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = lane & 3;       // level whose points this lane prepares; channel octet it owns
  const int pr = lane >> 2;     // pair within the wave
  const int par = (pr >> 1) & 1, setr = (pr >> 2) & 1;  // bank roles (see header)
  // second pass reads chunks in memory order: accA always holds the chunk at offA
  const int offA = k * 32 + setr * 16, offB = k * 32 + (setr ^ 1) * 16;

  const int lb = xcd_remap(blockIdx.x, p.n_blocks);
  const int head = lb & 7;
  const int tiles = p.nx * p.ny;
  const int tile = (lb >> 3) % tiles, frame = (lb >> 3) / tiles;
  // tiles in bands of 4 tile rows, column-major inside a band: the ~12 tiles an XCD has in flight
  // form a 4 x 3 patch, so the halo rows shared with the tiles above / below are still in its L2
  // (row-major order puts vertical neighbours 21 tiles = 8.7 MB of windows apart)
  constexpr int kBand = 4;
  const int band = tile / (kBand * p.nx), in_band = tile - band * (kBand * p.nx);
  const int band_rows = min(kBand, p.ny - band * kBand);
  const int tx = in_band / band_rows, ty = band * kBand + (in_band - tx * band_rows);

  // ---- which query this pair is: waves 0-3 level 0, wave 4 level 1, wave 5 levels 2, 3
  int qy, qx, qH, qW, qS;
  bool valid = true;
  if (wave < 4) {
    qy = ty * 8 + wave * 2 + (pr >> 3), qx = tx * 8 + (pr & 7);
    qH = p.Hs[0], qW = p.Ws[0], qS = p.St[0];
  } else if (wave == 4) {
    qy = ty * 4 + (pr >> 2), qx = tx * 4 + (pr & 3);
    qH = p.Hs[1], qW = p.Ws[1], qS = p.St[1];
  } else {
    const bool l2 = pr < 4;
    qy = l2 ? ty * 2 + (pr >> 1) : ty, qx = l2 ? tx * 2 + (pr & 1) : tx;
    qH = l2 ? p.Hs[2] : p.Hs[3], qW = l2 ? p.Ws[2] : p.Ws[3], qS = l2 ? p.St[2] : p.St[3];
    valid = pr < 5;
  }
  valid = valid && qy < qH && qx < qW;
  const unsigned ubase = (unsigned)frame * (unsigned)p.S;
  const unsigned unit = ubase + (valid ? (unsigned)(qS + qy * qW + qx) : 0u);

  // ---- projections of (unit, head, level k): 4 offset pairs + 4 logits, reference point
  const float* row = p.proj + (size_t)unit * (unsigned)p.proj_stride + head * 32 + k * 8;
  const float4 of01 = *reinterpret_cast<const float4*>(row);
  const float4 of23 = *reinterpret_cast<const float4*>(row + 4);
  const float4 lg = *reinterpret_cast<const float4*>(row + (kHeads - head) * 32 + head * 16 - k * 4);
  const float2 rf = PREP ? make_float2(0.f, 0.f)
                         : *reinterpret_cast<const float2*>(p.ref + (size_t)unit * 8 + k * 2);
  const int4* tab = reinterpret_cast<const int4*>(
      (const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TileParams, tab)) + k * 2;
  const int4 tabA = tab[0], tabB = tab[1];
  const int* shp = reinterpret_cast<const int*>(
      (const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TileParams, shift)) + head * 8;
  const int2 my_shift = *reinterpret_cast<const int2*>(shp + k * 2);   // lane k <-> level k

  // ---- stage the four windows (LDS-DMA) and the two zero rows
  const int ox0 = tx * 8 - MB0 + shp[0], oy0 = ty * 8 - MB0 + shp[1];
  const int ox1 = tx * 4 - MB1 + shp[2], oy1 = ty * 4 - MB1 + shp[3];
  const int ox2 = tx * 2 - MB2 + shp[4], oy2 = ty * 2 - MB2 + shp[5];
  const int ox3 = tx - MB3 + shp[6], oy3 = ty - MB3 + shp[7];
  const char* vhead = reinterpret_cast<const char*>(p.value) + (size_t)ubase * kRowBytes + head * 128;
  if (!(ABL & 1)) {
    stage_level<W0, kWaves>(lds + G::B0 * 128, vhead, wave, lane, ox0, oy0, p.Hs[0], p.Ws[0], p.St[0]);
    stage_level<W1, kWaves>(lds + G::B1 * 128, vhead, wave, lane, ox1, oy1, p.Hs[1], p.Ws[1], p.St[1]);
    stage_level<W2, kWaves>(lds + G::B2 * 128, vhead, wave, lane, ox2, oy2, p.Hs[2], p.Ws[2], p.St[2]);
    stage_level<W3, kWaves>(lds + G::B3 * 128, vhead, wave, lane, ox3, oy3, p.Hs[3], p.Ws[3], p.St[3]);
  }
  if (threadIdx.x < 64) reinterpret_cast<float*>(lds + G::kZ * 128)[threadIdx.x] = 0.f;

  // ---- my level's constants (lane k <-> level k): one vector load of the kernarg table row
  // instead of seven 4-way select chains
  static_assert(MB0 == MB1 && MB1 == MB2 && MB2 == MB3, "one window margin for all levels");
  const int ox = ((tx * 8) >> k) - MB0 + my_shift.x, oy = ((ty * 8) >> k) - MB0 + my_shift.y;
  const int H = tabA.x, W = tabA.y, st = tabA.z, ww = tabA.w, wbase = tabB.x;
  const float fW = (float)W, fH = (float)H;
  const float rW = __int_as_float(tabB.y), rH = __int_as_float(tabB.z);

  // ---- softmax over the 16 logits of (unit, head): 4 per lane, quad reduction
  // (PREP: the producer -- the merged projection GEMM's epilogue -- has run the softmax and the
  // location arithmetic with the very same code, pave_enc_math.h: `proj` holds the attention
  // weights and the level pixel coordinates)
  float ee[4], inv_sum;
  if (PREP) {
    ee[0] = lg.x, ee[1] = lg.y, ee[2] = lg.z, ee[3] = lg.w, inv_sum = 1.f;
  } else {
    pave_enc::softmax16(lg.x, lg.y, lg.z, lg.w, ee, inv_sum);
  }
  if (!valid) inv_sum = 0.f;  // idle slots carry weight 0 through the DPP steps

  // ---- corner descriptors of my 4 points
  PointDesc d[4];
  int flags = 0;
  {
    const float ofx[4] = {of01.x, of01.z, of23.x, of23.z}, ofy[4] = {of01.y, of01.w, of23.y, of23.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // MO:381-384, cuda_kernel.cuh:233-234
      const float px = PREP ? ofx[i] : pave_enc::pixel_coord(rf.x, ofx[i], fW, rW);
      const float py = PREP ? ofy[i] : pave_enc::pixel_coord(rf.y, ofy[i], fH, rH);
      const float aw = PREP ? (valid ? ee[i] : 0.f) : ee[i] * inv_sum;
      flags |= make_point(d[i], px, py, aw, H, W, st, ox, oy, ww, wbase, G::kZ, par) << (3 * i);
    }
  }
  if (!valid) flags &= ~0x249;  // no second pass for idle slots
  const bool any_far = (flags & 0x249) != 0;
  // descriptors are final here: keep hipcc from sinking their arithmetic into the gather loop
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(d[i].w[c]), "+v"(d[i].a[c]));
    asm volatile("" : "+v"(d[i].tok00));
  }
  asm volatile("" : "+v"(flags));
  __syncthreads();  // (emits s_waitcnt vmcnt(0): the windows have landed)

  float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = accA;
  if (!(ABL & 2)) {
    gather_level<0>(accA, accB, lds, d, offA, offB);
    gather_level<1>(accA, accB, lds, d, offA, offB);
    gather_level<2>(accA, accB, lds, d, offA, offB);
    gather_level<3>(accA, accB, lds, d, offA, offB);
  } else {  // keep the descriptors alive
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) accA.x += d[i].w[c] + __int_as_float(d[i].a[c]);
  }

  if (__builtin_amdgcn_ballot_w64(any_far) != 0ull) {  // rare with trained-size offsets
    // memory order inside the lane's 32 bytes: chunk 2k (flo), then 2k + 1 (fhi)
    float4 flo = make_float4(0.f, 0.f, 0.f, 0.f), fhi = flo;
    const char* vlane = vhead + k * 32;
    far_quad_loop<FAR_PER_ROUND>(flo, fhi, vlane, d, flags, lane, p.Ws[0], p.Ws[1], p.Ws[2], p.Ws[3], p.S);
    const float4 fa = setr ? fhi : flo, fb = setr ? flo : fhi;  // accA holds the chunk at offA
    accA.x += fa.x, accA.y += fa.y, accA.z += fa.z, accA.w += fa.w;
    accB.x += fb.x, accB.y += fb.y, accB.z += fb.z, accB.w += fb.w;
  }
  if (valid && p.out_f16) {   // (wave-uniform flag) the lane's two 4-channel chunks as 8 bytes each
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    char* o = reinterpret_cast<char*>(p.out) + ((size_t)unit * 256 + head * 32) * 2;
    *reinterpret_cast<h4*>(o + (offA >> 1)) = __builtin_convertvector(f4{accA.x, accA.y, accA.z, accA.w}, h4);
    *reinterpret_cast<h4*>(o + (offB >> 1)) = __builtin_convertvector(f4{accB.x, accB.y, accB.z, accB.w}, h4);
  } else if (valid) {
    char* o = reinterpret_cast<char*>(p.out + (size_t)unit * 256 + head * 32);
    *reinterpret_cast<float4*>(o + offA) = accA;
    *reinterpret_cast<float4*>(o + offB) = accB;
  }
}

}  // namespace

template <int ABL>
static int enc_tile_launch(const float* value, const float* proj, const float* ref, float* out,
                           int n_frames, int S, const int* levels_hw, int proj_stride, int variant,
                           const int* window_shift, void* stream) {
  // variant bit 2 (value 4): `proj` is PREPARED (pave_gemm_bf16x3_encproj_f32): attention weights and
  // level pixel coordinates instead of logits and offsets; ref is not read
  // every unsupported combination is refused HERE, before anything is enqueued
  if (variant < 0 || (variant & ~15) != 0)
    return pave_internal_fail(PAVE_E_ARG, "enc_deform_attn_tile: variant is a 4-bit mask (1 = wide windows, 4 = prepared "
                                          "input, 8 = fp16 output)");
  const bool prepared = (variant & 4) != 0;
  const bool out_f16 = (variant & 8) != 0;
  variant &= 3;
  if (prepared && variant != 0)
    return pave_internal_fail(PAVE_E_UNSUPPORTED, "enc_deform_attn_tile: prepared input with the default windows only");
  if (!value || !proj || (!ref && !prepared) || !out || !levels_hw)
    return pave_internal_fail(PAVE_E_ARG, "enc_deform_attn_tile: null pointer");
  if (n_frames <= 0 || S <= 0) return pave_internal_fail(PAVE_E_ARG, "enc_deform_attn_tile: sizes must be positive");
  if (proj_stride < kHeads * 16 * 3)
    return pave_internal_fail(PAVE_E_ARG, "enc_deform_attn_tile: proj_stride too small");
  if ((long long)S * kRowBytes >= (1ll << 31))
    return pave_internal_fail(PAVE_E_ARG, "enc_deform_attn_tile: one value slab must be < 2 GiB");
  TileParams p{};
  p.value = value;
  p.proj = proj;
  p.ref = ref;
  p.out = out;
  p.S = S;
  p.proj_stride = proj_stride;
  p.out_f16 = out_f16 ? 1 : 0;
  if (window_shift)
    for (int i = 0; i < kHeads * 8; ++i) {
      if (window_shift[i] < -64 || window_shift[i] > 64)
        return pave_internal_fail(PAVE_E_ARG, "enc_deform_attn_tile: |window_shift| <= 64");
      (&p.shift[0][0][0])[i] = window_shift[i];
    }
  long long start = 0;
  for (int l = 0; l < 4; ++l) {
    if (levels_hw[2 * l] <= 0 || levels_hw[2 * l + 1] <= 0)
      return pave_internal_fail(PAVE_E_ARG, "enc_deform_attn_tile: bad level size");
    p.Hs[l] = levels_hw[2 * l];
    p.Ws[l] = levels_hw[2 * l + 1];
    p.St[l] = (int)start;
    start += (long long)p.Hs[l] * p.Ws[l];
  }
  if (start != S) return pave_internal_fail(PAVE_E_ARG, "enc_deform_attn_tile: levels do not add up to S");
  p.nx = (p.Ws[0] + 7) / 8;
  p.ny = (p.Hs[0] + 7) / 8;
  for (int l = 1; l < 4; ++l)  // every coarser token must belong to exactly one 8x8 image tile
    if (p.Hs[l] > (8 >> l) * p.ny || p.Ws[l] > (8 >> l) * p.nx)
      return pave_internal_fail(PAVE_E_UNSUPPORTED,
                                "enc_deform_attn_tile: level sizes are not a halving pyramid");
  const long long nb = (long long)n_frames * p.nx * p.ny * kHeads;
  if (nb >= (1ll << 31)) return pave_internal_fail(PAVE_E_ARG, "enc_deform_attn_tile: grid too large");
  p.n_blocks = (int)nb;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  auto fill = [&](auto geom, int w0, int w1, int w2, int w3) {
    using G = decltype(geom);
    const int ww[4] = {w0, w1, w2, w3}, wb[4] = {G::B0, G::B1, G::B2, G::B3};
    for (int l = 0; l < 4; ++l) {
      const float rw = 1.f / (float)p.Ws[l], rh = 1.f / (float)p.Hs[l];
      int irw, irh;
      memcpy(&irw, &rw, 4);
      memcpy(&irh, &rh, 4);
      const int row[8] = {p.Hs[l], p.Ws[l], p.St[l], ww[l], wb[l], irw, irh, 0};
      memcpy(p.tab[l], row, sizeof(row));
    }
  };
  if (variant == 1) {  // +-4 px windows, 2 workgroups per CU
    fill(WinGeom<16, 12, 10, 9>{}, 16, 12, 10, 9);
    hipLaunchKernelGGL((enc_tile_kernel<16, 12, 10, 9, 4, 4, 4, 4, 1, ABL>), dim3((unsigned)nb), dim3(384), 0, st, p);
  } else {             // -4 .. +3 px windows (52 KB), 3 workgroups per CU
    fill(WinGeom<14, 10, 8, 7>{}, 14, 10, 8, 7);
    if (prepared)
      hipLaunchKernelGGL((enc_tile_kernel<14, 10, 8, 7, 3, 3, 3, 3, 5, ABL, true>), dim3((unsigned)nb), dim3(384), 0, st, p);
    else
      hipLaunchKernelGGL((enc_tile_kernel<14, 10, 8, 7, 3, 3, 3, 3, 5, ABL>), dim3((unsigned)nb), dim3(384), 0, st, p);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

extern "C" int pave_enc_deform_attn_tile_f32(const float* value, const float* proj,
                                              const float* ref, float* out, int n_frames, int S,
                                              const int* levels_hw, int proj_stride, int variant,
                                              const int* window_shift, void* stream) {
  return enc_tile_launch<0>(value, proj, ref, out, n_frames, S, levels_hw, proj_stride, variant,
                            window_shift, stream);
}

#ifdef PAVE_DIAG
// Timing-only ablations for tools/bench_kernels.py (-DPAVE_DIAG build only, outputs are wrong):
// ablate 1 = no window staging, 2 = no gather loop, 3 = neither.
extern "C" int pave_diag_enc_tile_ablate(const float* value, const float* proj, const float* ref,
                                         float* out, int n_frames, int S, const int* levels_hw,
                                         int proj_stride, int variant, int ablate, void* stream) {
  switch (ablate) {
    case 1: return enc_tile_launch<1>(value, proj, ref, out, n_frames, S, levels_hw, proj_stride, variant, nullptr, stream);
    case 2: return enc_tile_launch<2>(value, proj, ref, out, n_frames, S, levels_hw, proj_stride, variant, nullptr, stream);
    case 3: return enc_tile_launch<3>(value, proj, ref, out, n_frames, S, levels_hw, proj_stride, variant, nullptr, stream);
    default: return enc_tile_launch<0>(value, proj, ref, out, n_frames, S, levels_hw, proj_stride, variant, nullptr, stream);
  }
}
#endif  // PAVE_DIAG
