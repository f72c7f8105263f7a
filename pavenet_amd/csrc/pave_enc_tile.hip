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
#include <stdint.h>

#include "pave_hip.h"
#include "pave_internal.h"

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
  int nx, ny;          // tiles per frame
  int n_blocks;
  int Hs[4], Ws[4], St[4];
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
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                   0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true)));  // quad_perm:[1,0,3,2]
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                   0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true)));  // quad_perm:[2,3,0,1]
  return v;
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1,
                                                             0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e,
                                                             0xf, 0xf, true));
  return v;
}

template <int W0, int W1, int W2, int W3>
struct WinGeom {
  static constexpr int pad8(int r) { return (r + 7) & ~7; }
  static constexpr int R0 = pad8(W0 * W0), R1 = pad8(W1 * W1), R2 = pad8(W2 * W2),
                       R3 = pad8(W3 * W3);
  static constexpr int B0 = 0, B1 = R0, B2 = R0 + R1, B3 = R0 + R1 + R2;
  static constexpr int kRows = R0 + R1 + R2 + R3;  // multiple of 8
  // two all-zero rows (one of each parity) live in a region's pad rows when one has room
  // (the DMA never writes pad rows), else behind the last region
  static constexpr int zero_pair() {
    const int b[4] = {B0, B1, B2, B3}, r[4] = {R0, R1, R2, R3}, w[4] = {W0, W1, W2, W3};
    for (int l = 0; l < 4; ++l)
      if (r[l] - w[l] * w[l] >= 2) return b[l] + w[l] * w[l];
    return kRows;
  }
  static constexpr int kZ = zero_pair();
  static constexpr int kZeroEven = (kZ & 1) ? kZ + 1 : kZ, kZeroOdd = (kZ & 1) ? kZ : kZ + 1;
  static constexpr int kLdsBytes = (kZ == kRows ? kRows + 2 : kRows) * 128;
};

// LDS-DMA of one level's window: 8 rows (1 KiB) per wave instruction; lane = (row, 16-byte chunk)
template <int WW, int RPAD, int NW>
__device__ __forceinline__ void stage_level(char* lds_level, const char* vhead, int wave, int lane,
                                            int ox, int oy, int H, int W, int st) {
  constexpr int kGroups = RPAD / 8;
  for (int g = wave; g < kGroups; g += NW) {
    const int r = g * 8 + (lane >> 3);
    const int wy = r / WW, wx = r - wy * WW;
    // window rows beyond the map hold a clamped neighbour (their corners carry weight 0 and are
    // redirected to the zero rows anyway); pad rows of the region are not written
    const int x = min(max(ox + wx, 0), W - 1), y = min(max(oy + wy, 0), H - 1);
    const char* src = vhead + (size_t)(st + y * W + x) * kRowBytes + (lane & 7) * 16;
    if (r < WW * WW)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(lds_level + g * 1024), 16, 0, 0);
  }
}

// one bilinear point of the owner lane -> 4 (weight, LDS byte address, fallback token)
struct PointDesc {
  float w[4];
  int a[4];
  int fb[4];  // global token index of a corner outside the window, -1 otherwise
};

__device__ __forceinline__ void make_point(PointDesc& d, float px, float py, float aw, int H, int W,
                                           int st, int ox, int oy, int WW, int wbase, int par,
                                           int zero_even, int zero_odd) {
  // identical arithmetic to make_corners() of pave_kernels.hip (the direct kernels)
  const bool inside = (py > -1.f) && (px > -1.f) && (py < (float)H) && (px < (float)W);
  const float fy = floorf(py), fx = floorf(px);
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + 1, x1 = x0 + 1;
  const float ly = py - fy, lx = px - fx;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const bool y0ok = inside && (y0 >= 0), y1ok = inside && (y1 <= H - 1);
  const bool x0ok = (x0 >= 0), x1ok = (x1 <= W - 1);
  float w[4];
  w[0] = (y0ok && x0ok) ? hy * hx * aw : 0.f;
  w[1] = (y0ok && x1ok) ? hy * lx * aw : 0.f;
  w[2] = (y1ok && x0ok) ? ly * hx * aw : 0.f;
  w[3] = (y1ok && x1ok) ? ly * lx * aw : 0.f;
  const int dx0 = x0 - ox, dy0 = y0 - oy;
  const int r00 = wbase + __mul24(dy0, WW) + dx0;
  int a[4], fb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int dy = dy0 + (c >> 1), dx = dx0 + (c & 1);
    const int row = wbase + __mul24(dy, WW) + dx;
    const bool in_win = (unsigned)dy < (unsigned)WW && (unsigned)dx < (unsigned)WW;
    const bool live = w[c] != 0.f;
    // the x-neighbours of a footprint have rows of opposite parity; a masked or out-of-window
    // corner reads the all-zero row of the parity its slot expects
    const int want_odd = (r00 ^ c ^ ((c >> 1) & WW)) & 1;  // parity `row` would have
    const int zrow = want_odd ? zero_odd : zero_even;
    a[c] = ((live && in_win) ? row : zrow) * 128;
    fb[c] = (live && !in_win) ? st + (y0 + (c >> 1)) * W + (x0 + (c & 1)) : -1;
  }
  // visiting order of each x-pair: slot 0 reads the row whose parity equals the pair's role
  const bool swap = ((r00 ^ par) & 1) != 0;          // top pair (c = 0, 1)
  const bool swap2 = (((r00 + WW) ^ par) & 1) != 0;  // bottom pair (c = 2, 3)
  d.w[0] = swap ? w[1] : w[0];
  d.w[1] = swap ? w[0] : w[1];
  d.a[0] = swap ? a[1] : a[0];
  d.a[1] = swap ? a[0] : a[1];
  d.fb[0] = swap ? fb[1] : fb[0];
  d.fb[1] = swap ? fb[0] : fb[1];
  d.w[2] = swap2 ? w[3] : w[2];
  d.w[3] = swap2 ? w[2] : w[3];
  d.a[2] = swap2 ? a[3] : a[2];
  d.a[3] = swap2 ? a[2] : a[3];
  d.fb[2] = swap2 ? fb[3] : fb[2];
  d.fb[3] = swap2 ? fb[2] : fb[3];
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

template <int LVL>
__device__ __forceinline__ void fallback_level(float4& accA, float4& accB, const char* vlane,
                                               const PointDesc (&d)[4], int offA, int offB) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int tok = qbi<LVL>(d[i].fb[c]);
      if (__builtin_amdgcn_ballot_w64(tok >= 0) != 0ull) {
        const float w = qbf<LVL>(d[i].w[c]);
        if (tok >= 0) {
          const char* src = vlane + (size_t)tok * kRowBytes;
          fma4(accA, w, *reinterpret_cast<const float4*>(src + offA));
          fma4(accB, w, *reinterpret_cast<const float4*>(src + offB));
        }
      }
    }
  }
}

// TILE = 8: 64 + 16 + 4 + 1 = 85 queries -> 96 pair slots = 6 waves of 16 pairs
template <int W0, int W1, int W2, int W3, int MB0, int MB1, int MB2, int MB3, int WPE>
__global__ __launch_bounds__(384, WPE) void enc_tile_kernel(const TileParams p) {
  using G = WinGeom<W0, W1, W2, W3>;
  constexpr int kWaves = 6;
  __shared__ __attribute__((aligned(1024))) char lds[G::kLdsBytes];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = lane & 3;       // level whose points this lane prepares; channel octet it owns
  const int pr = lane >> 2;     // pair within the wave
  const int par = (pr >> 1) & 1, setr = (pr >> 2) & 1;  // bank roles (see header)
  const int offA = k * 32 + setr * 16, offB = k * 32 + (setr ^ 1) * 16;

  const int lb = xcd_remap(blockIdx.x, p.n_blocks);
  const int head = lb & 7;
  const int tiles = p.nx * p.ny;
  const int tile = (lb >> 3) % tiles, frame = (lb >> 3) / tiles;
  const int ty = tile / p.nx, tx = tile - ty * p.nx;

  // ---- which query this pair is (slot -> level, position inside the tile)
  const int slot = wave * 16 + pr;
  int ql, qy, qx;
  if (slot < 64) {
    ql = 0, qy = ty * 8 + (slot >> 3), qx = tx * 8 + (slot & 7);
  } else if (slot < 80) {
    ql = 1, qy = ty * 4 + ((slot - 64) >> 2), qx = tx * 4 + ((slot - 64) & 3);
  } else if (slot < 84) {
    ql = 2, qy = ty * 2 + ((slot - 80) >> 1), qx = tx * 2 + ((slot - 80) & 1);
  } else {
    ql = 3, qy = ty, qx = tx;
  }
  const int qH = ql == 0 ? p.Hs[0] : ql == 1 ? p.Hs[1] : ql == 2 ? p.Hs[2] : p.Hs[3];
  const int qW = ql == 0 ? p.Ws[0] : ql == 1 ? p.Ws[1] : ql == 2 ? p.Ws[2] : p.Ws[3];
  const int qS = ql == 0 ? p.St[0] : ql == 1 ? p.St[1] : ql == 2 ? p.St[2] : p.St[3];
  const bool valid = slot < 85 && qy < qH && qx < qW;
  const long long fbase = (long long)frame * p.S;
  const long long unit = fbase + (valid ? qS + qy * qW + qx : 0);

  // ---- my level's constants (lane k <-> level k)
  const int H = k == 0 ? p.Hs[0] : k == 1 ? p.Hs[1] : k == 2 ? p.Hs[2] : p.Hs[3];
  const int W = k == 0 ? p.Ws[0] : k == 1 ? p.Ws[1] : k == 2 ? p.Ws[2] : p.Ws[3];
  const int st = k == 0 ? p.St[0] : k == 1 ? p.St[1] : k == 2 ? p.St[2] : p.St[3];
  const int ox0 = tx * 8 - MB0, oy0 = ty * 8 - MB0, ox1 = tx * 4 - MB1, oy1 = ty * 4 - MB1;
  const int ox2 = tx * 2 - MB2, oy2 = ty * 2 - MB2, ox3 = tx - MB3, oy3 = ty - MB3;
  const int ox = k == 0 ? ox0 : k == 1 ? ox1 : k == 2 ? ox2 : ox3;
  const int oy = k == 0 ? oy0 : k == 1 ? oy1 : k == 2 ? oy2 : oy3;
  const int wbase = k == 0 ? G::B0 : k == 1 ? G::B1 : k == 2 ? G::B2 : G::B3;
  const int ww = k == 0 ? W0 : k == 1 ? W1 : k == 2 ? W2 : W3;

  // ---- projections of (unit, head, level k): 4 offsets pairs + 4 logits, reference point
  const float* row = p.proj + unit * p.proj_stride;
  const float4 of01 = *reinterpret_cast<const float4*>(row + head * 32 + k * 8);
  const float4 of23 = *reinterpret_cast<const float4*>(row + head * 32 + k * 8 + 4);
  const float4 lg = *reinterpret_cast<const float4*>(row + kHeads * 32 + head * 16 + k * 4);
  const float2 rf = *reinterpret_cast<const float2*>(p.ref + unit * 8 + k * 2);

  // ---- stage the four windows (LDS-DMA) and the two zero rows
  const char* vframe = reinterpret_cast<const char*>(p.value) + fbase * kRowBytes;
  const char* vhead = vframe + head * 128;
  stage_level<W0, G::R0, kWaves>(lds + G::B0 * 128, vhead, wave, lane, ox0, oy0, p.Hs[0], p.Ws[0], p.St[0]);
  stage_level<W1, G::R1, kWaves>(lds + G::B1 * 128, vhead, wave, lane, ox1, oy1, p.Hs[1], p.Ws[1], p.St[1]);
  stage_level<W2, G::R2, kWaves>(lds + G::B2 * 128, vhead, wave, lane, ox2, oy2, p.Hs[2], p.Ws[2], p.St[2]);
  stage_level<W3, G::R3, kWaves>(lds + G::B3 * 128, vhead, wave, lane, ox3, oy3, p.Hs[3], p.Ws[3], p.St[3]);
  if (threadIdx.x < 64) {
    reinterpret_cast<float*>(lds + G::kZeroEven * 128)[threadIdx.x & 31] = 0.f;
    reinterpret_cast<float*>(lds + G::kZeroOdd * 128)[threadIdx.x & 31] = 0.f;
  }

  // ---- softmax over the 16 logits of (unit, head): 4 per lane, quad reduction
  const float mx = quad_max(fmaxf(fmaxf(lg.x, lg.y), fmaxf(lg.z, lg.w)));
  const float e0 = expf(lg.x - mx), e1 = expf(lg.y - mx), e2 = expf(lg.z - mx),
              e3 = expf(lg.w - mx);
  const float inv_sum = 1.f / quad_sum((e0 + e1) + (e2 + e3));

  // ---- corner descriptors of my 4 points
  PointDesc d[4];
  {
    const float fW = (float)W, fH = (float)H;
    const float ofx[4] = {of01.x, of01.z, of23.x, of23.z}, ofy[4] = {of01.y, of01.w, of23.y, of23.w};
    const float ee[4] = {e0, e1, e2, e3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float lx = rf.x + ofx[i] / fW, ly = rf.y + ofy[i] / fH;  // MO:381-384
      const float px = lx * fW - 0.5f, py = ly * fH - 0.5f;           // cuda_kernel.cuh:233-234
      make_point(d[i], px, py, ee[i] * inv_sum, H, W, st, ox, oy, ww, wbase, par, G::kZeroEven,
                 G::kZeroOdd);
    }
  }
  bool any_fb = false;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) any_fb |= d[i].fb[c] >= 0;
  if (!valid) {  // idle slots: weights 0 on the zero rows (they still take part in the DPP steps)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d[i].w[c] = 0.f;
        d[i].fb[c] = -1;
      }
    any_fb = false;
  }
  __syncthreads();  // (emits s_waitcnt vmcnt(0): the windows have landed)

  float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = accA;
  gather_level<0>(accA, accB, lds, d, offA, offB);
  gather_level<1>(accA, accB, lds, d, offA, offB);
  gather_level<2>(accA, accB, lds, d, offA, offB);
  gather_level<3>(accA, accB, lds, d, offA, offB);

  if (__builtin_amdgcn_ballot_w64(any_fb) != 0ull) {  // rare with trained-size offsets
    const char* vlane = vhead;
    fallback_level<0>(accA, accB, vlane, d, offA, offB);
    fallback_level<1>(accA, accB, vlane, d, offA, offB);
    fallback_level<2>(accA, accB, vlane, d, offA, offB);
    fallback_level<3>(accA, accB, vlane, d, offA, offB);
  }
  if (valid) {
    char* o = reinterpret_cast<char*>(p.out + unit * 256 + head * 32);
    *reinterpret_cast<float4*>(o + offA) = accA;
    *reinterpret_cast<float4*>(o + offB) = accB;
  }
}

}  // namespace

extern "C" int pave_enc_deform_attn_tile_f32(const float* value, const float* proj,
                                              const float* ref, float* out, int n_frames, int S,
                                              const int* levels_hw, int proj_stride, int variant,
                                              void* stream) {
  if (!value || !proj || !ref || !out || !levels_hw)
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
  if (variant == 1) {  // +-4 px windows, 2 workgroups per CU
    hipLaunchKernelGGL((enc_tile_kernel<16, 12, 10, 9, 4, 4, 4, 4, 1>), dim3((unsigned)nb), dim3(384), 0, st, p);
  } else {             // -4 .. +3 px windows (52 KB), 3 workgroups per CU
    hipLaunchKernelGGL((enc_tile_kernel<14, 10, 8, 7, 3, 3, 3, 3, 1>), dim3((unsigned)nb), dim3(384), 0, st, p);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}
