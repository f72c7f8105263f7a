// pave_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the PAVE-Net
// forward hot path + their C ABI (include/pave_hip.h).
//
// Written for MI355X only: 64-lane wavefronts, 8 lanes x float4 = one 128-byte
// value row per attention head, LDS-staged sampling descriptors, XCD-aware
// block -> unit mapping.  No CUDA compatibility layer.
//
// Semantics restated from (zgspose/PAVENet):
//   third_party/mmcv/mmcv/ops/csrc/common/cuda/ms_deform_attn_cuda_kernel.cuh:17-64,200-254
//   third_party/mmcv/mmcv/ops/multi_scale_deform_attn.py:305-412, 1388-1587
//   opera/models/utils/transformer.py:1644-1863
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pave_hip.h"
#include "pave_internal.h"

namespace {

constexpr int kWave = 64;
constexpr int kHeads = 8;      // fused kernels: M
constexpr int kDim = 32;       // fused kernels: D
constexpr int kRowFloats = kHeads * kDim;  // 256 floats = 1 KiB per token
constexpr int kMaxLevels = 8;

thread_local char g_err[256] = "";

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

// ---------------------------------------------------------------------------
// [R1] generic forward: arbitrary M, D, L, P; scalar_t = float | double.
// One thread per output scalar (b, q, m, c), exactly the reference's work
// decomposition, used for shapes the vector kernel does not cover (D % 4 != 0,
// fp64).  Arithmetic order follows ms_deform_attn_cuda_kernel.cuh:200-254.
// ---------------------------------------------------------------------------
template <typename scalar_t>
__global__ __launch_bounds__(256) void msda_fwd_scalar_kernel(
    const long long n, const scalar_t* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const scalar_t* __restrict__ loc,
    const scalar_t* __restrict__ attw, const int S, const int M, const int D, const int L,
    const int Lq, const int P, scalar_t* __restrict__ out) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < n;
       index += (long long)gridDim.x * blockDim.x) {
    long long tmp = index;
    const int c = (int)(tmp % D);
    tmp /= D;
    const long long sampling_index = tmp;  // (b*Lq + q)*M + m
    const int m = (int)(tmp % M);
    tmp /= M;
    tmp /= Lq;
    const long long b = tmp;
    long long wptr = sampling_index * L * P;
    long long lptr = wptr << 1;
    const long long row = (long long)M * D;
    const scalar_t* vb = value + b * S * row;
    scalar_t col = 0;
    for (int l = 0; l < L; ++l) {
      const long long start = lsi[l];
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const scalar_t* vl = vb + start * row;
      for (int p = 0; p < P; ++p) {
        const scalar_t loc_w = loc[lptr], loc_h = loc[lptr + 1];
        const scalar_t weight = attw[wptr];
        const scalar_t h_im = loc_h * H - (scalar_t)0.5;
        const scalar_t w_im = loc_w * W - (scalar_t)0.5;
        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
          const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
          const int h_high = h_low + 1, w_high = w_low + 1;
          const scalar_t lh = h_im - h_low, lw = w_im - w_low;
          const scalar_t hh = 1 - lh, hw = 1 - lw;
          scalar_t v1 = 0, v2 = 0, v3 = 0, v4 = 0;
          const long long base = (long long)m * D + c;
          if (h_low >= 0 && w_low >= 0) v1 = vl[((long long)h_low * W + w_low) * row + base];
          if (h_low >= 0 && w_high <= W - 1) v2 = vl[((long long)h_low * W + w_high) * row + base];
          if (h_high <= H - 1 && w_low >= 0) v3 = vl[((long long)h_high * W + w_low) * row + base];
          if (h_high <= H - 1 && w_high <= W - 1)
            v4 = vl[((long long)h_high * W + w_high) * row + base];
          const scalar_t w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
          col += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * weight;
        }
        wptr += 1;
        lptr += 2;
      }
    }
    out[index] = col;
  }
}

// ---------------------------------------------------------------------------
// Bilinear corner descriptor shared by the vector kernels: 4 weights (already
// multiplied by the attention weight, 0 for a corner outside the map) and 4
// byte offsets of the corner's 128-byte head row relative to the frame slab.
// ---------------------------------------------------------------------------
struct Corners {
  float w[4];
  int o[4];
};

// px, py: pixel-space sample position (loc * size - 0.5).  `rowbytes` = M*D*4.
__device__ __forceinline__ Corners make_corners(float px, float py, int H, int W, int start,
                                                int head_byte, int rowbytes, float aw) {
  Corners c;
  const bool inside = (py > -1.f) && (px > -1.f) && (py < (float)H) && (px < (float)W);
  const float fy = floorf(py), fx = floorf(px);
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + 1, x1 = x0 + 1;
  const float ly = py - fy, lx = px - fx;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const bool y0ok = inside && (y0 >= 0), y1ok = inside && (y1 <= H - 1);
  const bool x0ok = (x0 >= 0), x1ok = (x1 <= W - 1);
  c.w[0] = (y0ok && x0ok) ? hy * hx * aw : 0.f;
  c.w[1] = (y0ok && x1ok) ? hy * lx * aw : 0.f;
  c.w[2] = (y1ok && x0ok) ? ly * hx * aw : 0.f;
  c.w[3] = (y1ok && x1ok) ? ly * lx * aw : 0.f;
  // clamp so that a masked corner still reads a nearby, valid (cached) row
  const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
  const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
  c.o[0] = (start + cy0 * W + cx0) * rowbytes + head_byte;
  c.o[1] = (start + cy0 * W + cx1) * rowbytes + head_byte;
  c.o[2] = (start + cy1 * W + cx0) * rowbytes + head_byte;
  c.o[3] = (start + cy1 * W + cx1) * rowbytes + head_byte;
  return c;
}

__device__ __forceinline__ float4 ld16(const char* base, unsigned off) {
  return *reinterpret_cast<const float4*>(base + off);
}

__device__ __forceinline__ void fma4(float4& acc, float w, const float4& v) {
  acc.x = fmaf(w, v.x, acc.x);
  acc.y = fmaf(w, v.y, acc.y);
  acc.z = fmaf(w, v.z, acc.z);
  acc.w = fmaf(w, v.w, acc.w);
}

// ---------------------------------------------------------------------------
// [R1] vector forward, fp32, D % 4 == 0, G = D/4 lanes per (b, q, m) group, G a
// power of two <= 64.  Each lane owns 4 channels and walks the L*P points of its
// group; loc / weight reads are same-address broadcasts inside the group.
// ---------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void msda_fwd_vec_kernel(
    const long long ngroups, const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc,
    const float* __restrict__ attw, const int S, const int M, const int D, const int L,
    const int Lq, const int P, float* __restrict__ out) {
  int Hs[kMaxLevels], Ws[kMaxLevels], St[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) {
    if (l < L) {
      Hs[l] = (int)shapes[2 * l];
      Ws[l] = (int)shapes[2 * l + 1];
      St[l] = (int)lsi[l];
    } else {
      Hs[l] = Ws[l] = 1;
      St[l] = 0;
    }
  }
  const int rowbytes = M * D * 4;
  const long long slab = (long long)S * rowbytes;
  constexpr int kGroupsPerBlock = 256 / G;
  const int sub = threadIdx.x % G;
  for (long long g = (long long)blockIdx.x * kGroupsPerBlock + threadIdx.x / G; g < ngroups;
       g += (long long)gridDim.x * kGroupsPerBlock) {
    const int m = (int)(g % M);
    const long long b = g / ((long long)M * Lq);
    const char* vb = reinterpret_cast<const char*>(value) + b * slab + sub * 16;
    const float* lp = loc + g * L * P * 2;
    const float* wp = attw + g * L * P;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l) {
      if (l < L) {
        const int H = Hs[l], W = Ws[l];
        for (int p = 0; p < P; ++p) {
          const float2 xy = *reinterpret_cast<const float2*>(lp);
          const float aw = *wp;
          lp += 2;
          wp += 1;
          const Corners c =
              make_corners(xy.x * W - 0.5f, xy.y * H - 0.5f, H, W, St[l], m * D * 4, rowbytes, aw);
          const float4 v0 = ld16(vb, c.o[0]), v1 = ld16(vb, c.o[1]);
          const float4 v2 = ld16(vb, c.o[2]), v3 = ld16(vb, c.o[3]);
          fma4(acc, c.w[0], v0);
          fma4(acc, c.w[1], v1);
          fma4(acc, c.w[2], v2);
          fma4(acc, c.w[3], v3);
        }
      }
    }
    *reinterpret_cast<float4*>(out + g * D + sub * 4) = acc;
  }
}

// ---------------------------------------------------------------------------
// Fused T-frame deformable attention, M = 8 x D = 32.
//
// Wave layout: lane = (head h = lane>>3, sub j = lane&7).  The 8 lanes of a head
// own the head's 32 channels as float4 each, so one corner fetch is one 128-byte
// row per head and a wave instruction gathers 8 rows.
//
// Work is cut into ITEMS of <= 8*PPL sampling points per head:
//   GRID: item = frame t, 16 points = 4 levels x 4 points (slot i -> l = i>>2)
//   POSE: item = (frame t, level l), K keypoints
// Lane j of a head prepares the corner descriptors of points j*PPL .. j*PPL+PPL-1
// of the item (location arithmetic + softmax weight), stages them in LDS, and
// all 8 lanes of the head then walk the item's points reading the descriptors as
// LDS broadcasts.
//
// Softmax: every wave first makes one online (max, sum-exp) pass over ALL
// T*L*P logits of its unit (8 lanes/head + 3 xor-shuffles), so its items carry
// final normalised weights; waves of one unit just add their partial rows
// through LDS at the end.
// ---------------------------------------------------------------------------
enum { kGrid = 0, kPose = 1 };

struct FusedParams {
  const float* value;
  const int64_t* shapes;
  const int64_t* lsi;
  const float* proj;
  const float* ref;
  const int32_t* unit_clip;
  const int32_t* order;
  const int32_t* frame_table;  // slab of (clip, t) = frame_table[clip * T + t] (NULL: clip * T + t)
  float* out;
  float* stat_max;
  float* stat_sum;
  int n_units;
  int units_per_clip;
  int T;
  int S;
  int L;
  int P;  // points per level (GRID: 4; POSE: K)
  int ref_L;  // level rows behind every reference entry: L, or 1 = one row shared by the L levels
  int n_slabs;  // frame slabs behind `value`: every slab index is clamped into [0, n_slabs)
  int proj_stride;
  int n_blocks_logical;
};

__device__ __forceinline__ float group8_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1));
  v = fmaxf(v, __shfl_xor(v, 2));
  v = fmaxf(v, __shfl_xor(v, 4));
  return v;
}
__device__ __forceinline__ float group8_min(float v) {
  v = fminf(v, __shfl_xor(v, 1));
  v = fminf(v, __shfl_xor(v, 2));
  v = fminf(v, __shfl_xor(v, 4));
  return v;
}
__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

// XCD-aware bijective remap: hardware deals blocks round-robin over the 8 XCDs
// (block b -> XCD b % 8); give each XCD one contiguous run of logical blocks so
// that the units it processes (ordered by image band on the host) share its L2.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int per = nb >> 3, rem = nb & 7;
  const int x = b & 7, idx = b >> 3;
  return x * per + min(x, rem) + idx;
}

template <int MODE, int PPL, int WQ>
__global__ __launch_bounds__(256) void fused_deform_attn_kernel(const FusedParams p) {
  constexpr int kWavesPerBlock = 4;
  constexpr int kUnitsPerBlock = kWavesPerBlock / WQ;
  constexpr int kSlots = 8 * PPL;
  constexpr int kHeadStride = kSlots * 8 + 8;  // floats; +8 pad => heads on distinct banks
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kHeads * kHeadStride];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int h = lane >> 3, j = lane & 7;
  const int wq = wave % WQ;  // which of the unit's WQ waves
  const int logical_block = xcd_remap(blockIdx.x, p.n_blocks_logical);
  const int slot_unit = logical_block * kUnitsPerBlock + wave / WQ;
  const bool active = slot_unit < p.n_units;
  const int unit = active ? (p.order ? p.order[slot_unit] : slot_unit) : 0;

  const int L = p.L, P = p.P, T = p.T;
  const int LP = L * P;
  const int rowbytes = kRowFloats * 4;
  float* my_lds = lds + (wave * kHeads + h) * kHeadStride;

  // level table (uniform -> SGPRs)
  int Hs[4], Ws[4], St[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const int ll = l < L ? l : 0;
    Hs[l] = (int)p.shapes[2 * ll];
    Ws[l] = (int)p.shapes[2 * ll + 1];
    St[l] = (int)p.lsi[ll];
  }

  const float* row = p.proj + (long long)unit * p.proj_stride;
  const float* off_row = row;                            // [T][8][LP][2]
  const float* logit_row = row + (long long)T * kHeads * LP * 2;  // [T][8][LP]
  const int n_items = (MODE == kGrid) ? T : T * L;
  const int n_items_run = active ? n_items : 0;  // idle waves still join the barriers
  const int pts = (MODE == kGrid) ? 16 : P;  // valid slots per item

  // ---- pass 1: online softmax statistics over all T*LP logits of (unit, head)
  float mx = -INFINITY, sm = 0.f;
  for (int it = 0; it < n_items; ++it) {
    const int t = (MODE == kGrid) ? it : it / L;
    const int lp0 = (MODE == kGrid) ? 0 : (it % L) * P;
    const float* lg = logit_row + (t * kHeads + h) * LP + lp0;
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
      const int i = j * PPL + s;
      if (i < pts) {
        const float x = lg[i];
        const float nm = fmaxf(mx, x);
        sm = sm * expf(mx - nm) + expf(x - nm);
        mx = nm;
      }
    }
  }
  {
    const float gm = group8_max(mx);
    sm = (mx == -INFINITY) ? 0.f : sm * expf(mx - gm);
    sm = group8_sum(sm);
    mx = gm;
  }
  const float inv_sum = 1.f / sm;

  const int clip = p.unit_clip ? p.unit_clip[unit] : unit / p.units_per_clip;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- pass 2: this wave's items
  for (int it = wq; it < n_items_run; it += WQ) {
    const int t = (MODE == kGrid) ? it : it / L;
    const int lvl = (MODE == kGrid) ? 0 : it % L;
    const int lp0 = (MODE == kGrid) ? 0 : lvl * P;
    const float* lg = logit_row + (t * kHeads + h) * LP + lp0;
    const float* of = off_row + ((long long)(t * kHeads + h) * LP + lp0) * 2;
    // (clamped: a frame table / unit_clip entry outside the value tensor reads a wrong slab, never
    // memory outside it)
    const int slab = min(max(p.frame_table ? p.frame_table[clip * T + t] : clip * T + t, 0), p.n_slabs - 1);
    const char* frame = reinterpret_cast<const char*>(p.value) +
                        (long long)slab * p.S * rowbytes + j * 16;

    float rx[PPL], ry[PPL];
    float whx = 0.f, why = 0.f;
    if (MODE == kPose) {
      // ref [n_clips, T, Q, L, 2K]; unit = clip*Q + q
      const int q = unit - clip * p.units_per_clip;
      const float* rp =
          p.ref + ((((long long)clip * T + t) * p.units_per_clip + q) * p.ref_L + (p.ref_L == 1 ? 0 : lvl)) * (2 * P);
      float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
#pragma unroll
      for (int s = 0; s < PPL; ++s) {
        const int i = j * PPL + s;
        if (i < pts) {
          const float2 r = *reinterpret_cast<const float2*>(rp + 2 * i);
          rx[s] = r.x;
          ry[s] = r.y;
          xmin = fminf(xmin, r.x);
          xmax = fmaxf(xmax, r.x);
          ymin = fminf(ymin, r.y);
          ymax = fmaxf(ymax, r.y);
        } else {
          rx[s] = ry[s] = 0.f;
        }
      }
      xmin = group8_min(xmin);
      xmax = group8_max(xmax);
      ymin = group8_min(ymin);
      ymax = group8_max(ymax);
      whx = fmaxf(xmax - xmin, 1e-4f);
      why = fmaxf(ymax - ymin, 1e-4f);
    }

    // descriptors of my PPL points -> LDS
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
      const int i = j * PPL + s;
      Corners c;
      if (i < pts) {
        const int l = (MODE == kGrid) ? (i >> 2) : lvl;
        // select level constants without dynamic register indexing
        const int H = l == 0 ? Hs[0] : l == 1 ? Hs[1] : l == 2 ? Hs[2] : Hs[3];
        const int W = l == 0 ? Ws[0] : l == 1 ? Ws[1] : l == 2 ? Ws[2] : Ws[3];
        const int st = l == 0 ? St[0] : l == 1 ? St[1] : l == 2 ? St[2] : St[3];
        const float2 o = *reinterpret_cast<const float2*>(of + 2 * i);
        const float aw = expf(lg[i] - mx) * inv_sum;
        float lx, ly;
        if (MODE == kGrid) {
          // ref [T, n_units, L, 2]
          const float2 r = *reinterpret_cast<const float2*>(
              p.ref + (((long long)t * p.n_units + unit) * p.ref_L + (p.ref_L == 1 ? 0 : l)) * 2);
          lx = r.x + o.x / (float)W;
          ly = r.y + o.y / (float)H;
        } else {
          lx = rx[s] + o.x * whx * 0.5f;
          ly = ry[s] + o.y * why * 0.5f;
        }
        c = make_corners(lx * (float)W - 0.5f, ly * (float)H - 0.5f, H, W, st, h * kDim * 4,
                         rowbytes, aw);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          c.w[k] = 0.f;
          c.o[k] = h * kDim * 4;
        }
      }
      float4* dst = reinterpret_cast<float4*>(my_lds + i * 8);
      dst[0] = make_float4(c.w[0], c.w[1], c.w[2], c.w[3]);
      dst[1] = make_float4(__int_as_float(c.o[0]), __int_as_float(c.o[1]),
                           __int_as_float(c.o[2]), __int_as_float(c.o[3]));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // gather: all 8 lanes of the head walk the item's points
#pragma unroll 4
    for (int i = 0; i < pts; ++i) {
      const float4* src = reinterpret_cast<const float4*>(my_lds + i * 8);
      const float4 w = src[0];
      const float4 o = src[1];
      const float4 v0 = ld16(frame, (unsigned)__float_as_int(o.x));
      const float4 v1 = ld16(frame, (unsigned)__float_as_int(o.y));
      const float4 v2 = ld16(frame, (unsigned)__float_as_int(o.z));
      const float4 v3 = ld16(frame, (unsigned)__float_as_int(o.w));
      fma4(acc, w.x, v0);
      fma4(acc, w.y, v1);
      fma4(acc, w.z, v2);
      fma4(acc, w.w, v3);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- combine the WQ partial rows of a unit
  if (WQ > 1) {
    __syncthreads();  // every wave is done with its descriptor region
    float4* red = reinterpret_cast<float4*>(lds);  // [wave][64] float4 = 4 KiB
    red[wave * 64 + lane] = acc;
    __syncthreads();
    if (wq == 0) {
#pragma unroll
      for (int k = 1; k < WQ; ++k) {
        const float4 o = red[(wave + k) * 64 + lane];
        acc.x += o.x;
        acc.y += o.y;
        acc.z += o.z;
        acc.w += o.w;
      }
    }
  }
  if (active && wq == 0) {
    *reinterpret_cast<float4*>(p.out + (long long)unit * kRowFloats + lane * 4) = acc;
    if (p.stat_max && j == 0) {
      p.stat_max[(long long)unit * kHeads + h] = mx;
      p.stat_sum[(long long)unit * kHeads + h] = sm;
    }
  }
}


// ---------------------------------------------------------------------------
// Greedy OKS-NMS on the device (replaces the NumPy loop + host sync of
// opera/models/dense_heads/videopose_head_mul_frames.py:1624-1665).
// One workgroup per clip.  Precisions follow the NumPy code: squared distances
// and areas in float32, the division chain / exp / mean in float64.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void oks_nms_kernel(const float* __restrict__ kpts,
                                                      const float* __restrict__ scores,
                                                      const double* __restrict__ sigmas,
                                                      const double thresh, int* __restrict__ keep,
                                                      int* __restrict__ order_out, const int N,
                                                      const int K, const int staged) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* var_s = reinterpret_cast<double*>(smem);     // [K] when `staged`: (2 sigma_k)^2
  int* order = reinterpret_cast<int*>(smem + (staged ? K * sizeof(double) : 0));   // [N]
  int* dead = order + N;                               // [N]
  float* area = reinterpret_cast<float*>(dead + N);    // [N]
  float* kxy = area + N;                               // [N][K][2] when `staged`
  const int b = blockIdx.x;
  const float* kp = kpts + (long long)b * N * K * 3;
  const float* sc = scores + (long long)b * N;
  // the clip's key points once into LDS (the launcher stages them while they fit): the suppression sweep below is
  // N dependent steps, each of which otherwise waits for its poses' coordinates from L2
  if (staged) {
    for (int t = threadIdx.x; t < N * K; t += blockDim.x) {
      kxy[2 * t] = kp[3 * t];
      kxy[2 * t + 1] = kp[3 * t + 1];
    }
    for (int k = threadIdx.x; k < K; k += blockDim.x) var_s[k] = (sigmas[k] * 2.0) * (sigmas[k] * 2.0);
  }
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int k = 0; k < K; ++k) {
      const float x = kp[(j * K + k) * 3], y = kp[(j * K + k) * 3 + 1];
      xmin = fminf(xmin, x);
      xmax = fmaxf(xmax, x);
      ymin = fminf(ymin, y);
      ymax = fmaxf(ymax, y);
    }
    area[j] = (xmax - xmin) * (ymax - ymin);
    // descending score; ties: larger index first (reverse of a stable ascending sort).  A NaN
    // score sorts as +inf (numpy's argsort puts NaNs last, the reference reverses that order):
    // every comparison with a raw NaN is false, which would leave `order` a non-permutation and
    // send the suppression loop to arbitrary addresses.
    const float s0 = sc[j], s = (s0 != s0) ? INFINITY : s0;
    int rank = 0;
    for (int i = 0; i < N; ++i) {
      const float t0 = sc[i], t = (t0 != t0) ? INFINITY : t0;
      rank += (t > s) || (t == s && i > j);
    }
    order[rank] = j;
    dead[j] = 0;
  }
  __syncthreads();
  for (int ii = 0; ii < N; ++ii) {
    const int i = order[ii];
    if (!dead[i]) {  // uniform: read from LDS after the barrier
      for (int jj = ii + 1 + threadIdx.x; jj < N; jj += blockDim.x) {
        const int j = order[jj];
        if (dead[j]) continue;
        const double denom = (double)((area[i] + area[j]) / 2.f) + 2.220446049250313e-16;
        double acc = 0.0;
        for (int k = 0; k < K; ++k) {
          float dx, dy;
          if (staged) {
            dx = kxy[(j * K + k) * 2] - kxy[(i * K + k) * 2];
            dy = kxy[(j * K + k) * 2 + 1] - kxy[(i * K + k) * 2 + 1];
          } else {
            dx = kp[(j * K + k) * 3] - kp[(i * K + k) * 3];
            dy = kp[(j * K + k) * 3 + 1] - kp[(i * K + k) * 3 + 1];
          }
          const float d2 = dx * dx + dy * dy;
          const double var = staged ? var_s[k] : (sigmas[k] * 2.0) * (sigmas[k] * 2.0);
          const double e = (double)d2 / var / denom / 2.0;
          acc += exp(-e);
        }
        if (acc / (double)K > thresh) dead[j] = 1;
      }
    }
    __syncthreads();
  }
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    keep[(long long)b * N + j] = dead[j] ? 0 : 1;
    order_out[(long long)b * N + j] = order[j];
  }
}

// ---------------------------------------------------------------------------
// Fused row epilogues (HBM-bound glue around the library GEMMs / convolutions).
//
// bias_act_rows:      y[r, c] = act(x[r, c] + bias[c] (+ res[r, c])), in place allowed.
//                     One pass (1-2 reads + 1 write) instead of the 2-3 passes of separate
//                     bias / residual-add / ReLU kernels.  C % 4 == 0; rows are NHWC pixels
//                     or tokens.
// bias_add_layernorm: y[r, :] = LayerNorm(x[r, :] (+ bias) (+ res[r, :])) * gamma + beta.
//                     One wave per row (C <= 1024, C % 4 == 0): float4 loads, two-pass
//                     mean / variance in registers, wave64 xor-shuffle reductions.
// ---------------------------------------------------------------------------
// fuse_sum:           y[n, h, w, :] = act(sum_k src_k[n, h >> s_k, w >> s_k, :]) in the order k = 0 .. 3:
//                     HRNet's fuse layer (hrnet.py:197-214: y += fuse_layers[i][j](x[j]) over the
//                     branches, nearest-neighbour up-sampling of the coarser ones, ReLU) as ONE pass
//                     instead of an up-sampling kernel and an add per term.
struct FuseSrc {
  const float* p[4];
  int sh[4];
};
__global__ __launch_bounds__(256) void fuse_sum_kernel(const FuseSrc src, float* __restrict__ y,
                                                       const long long n4, const int H, const int W,
                                                       const int c4, const int relu) {
  float4* y4 = reinterpret_cast<float4*>(y);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long pix = i / c4;
    const int c = (int)(i - pix * c4);
    const long long row = pix / W;
    const int w = (int)(pix - row * W);
    const long long n = row / H;
    const int h = (int)(row - n * H);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    bool first = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!src.p[k]) continue;
      const int sh = src.sh[k];
      const long long si = ((n * (H >> sh) + (h >> sh)) * (W >> sh) + (w >> sh)) * c4 + c;
      const float4 t = reinterpret_cast<const float4*>(src.p[k])[si];
      if (first) {
        v = t;      // (0 + t == t: the reference starts its sum at 0)
        first = false;
      } else {
        v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
      }
    }
    if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    y4[i] = v;
  }
}

// x[rows[i], :] = values (or zeros) for the listed rows only: the reference's padding-mask semantics on a
// projected value matrix without a pass over the whole matrix (a padded 800 x 1333 image masks ~1 % of the tokens)
__global__ __launch_bounds__(256) void fill_rows_kernel(float* __restrict__ x, const long long ld,
                                                        const int* __restrict__ rows, const long long n4,
                                                        const int c4, const long long total_rows,
                                                        const float* __restrict__ values) {
  const float4* v4 = reinterpret_cast<const float4*>(values);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c4;
    const int c = (int)(i - r * c4);
    const long long row = rows[r];
    if (row < 0 || row >= total_rows) continue;     // (an index outside the matrix is never dereferenced)
    const float4 v = values ? v4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(x + row * ld + 4 * c) = v;
  }
}

// dst[r, 0:W] = src[r, 0:W], dst[r, W:pitch] = 0: image rows re-laid at a 16-byte aligned pitch (one lane per
// 16-byte chunk of dst; the source rows start at any 4-byte boundary, so they are read dword by dword -- the
// four loads of a lane are consecutive addresses and neighbouring lanes continue them: full lines either way)
__global__ __launch_bounds__(256) void repitch_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           const long long rows, const int W, const int pitch4) {
  const long long n4 = rows * pitch4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / pitch4;
    const int c = 4 * (int)(i - r * pitch4);
    const float* s = src + r * W + c;
    float4 v;
    v.x = c < W ? s[0] : 0.f;
    v.y = c + 1 < W ? s[1] : 0.f;
    v.z = c + 2 < W ? s[2] : 0.f;
    v.w = c + 3 < W ? s[3] : 0.f;
    reinterpret_cast<float4*>(dst)[i] = v;
  }
}

__global__ __launch_bounds__(256) void bias_act_rows_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ res,
                                                            float* __restrict__ y,
                                                            const long long n4, const int c4,
                                                            const int relu) {
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* b4 = reinterpret_cast<const float4*>(bias);
  const float4* r4 = reinterpret_cast<const float4*>(res);
  float4* y4 = reinterpret_cast<float4*>(y);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    float4 v = x4[i];
    if (bias) {
      const float4 b = b4[i % c4];
      v.x += b.x;
      v.y += b.y;
      v.z += b.z;
      v.w += b.w;
    }
    if (res) {
      const float4 r = r4[i];
      v.x += r.x;
      v.y += r.y;
      v.z += r.z;
      v.w += r.w;
    }
    if (relu) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    }
    y4[i] = v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int VPL>  // float4 vectors per lane: C = 256 * VPL
__global__ __launch_bounds__(256) void bias_add_layernorm_kernel(
    const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ res,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
    const long long rows, const int C, const float eps, const float* __restrict__ pos,
    const long long pos_rows, float* __restrict__ y_plus) {
  // optional second output y_plus[r] = y[r] + pos[r % pos_rows]: the next layer's
  // `query + query_pos` (MO:353-354) written in the same pass
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const int c4 = C >> 2;
  const float inv_c = 1.f / (float)C;
  for (long long r = wave; r < rows; r += nwaves) {
    const float4* x4 = reinterpret_cast<const float4*>(x + r * C);
    const float4* r4 = res ? reinterpret_cast<const float4*>(res + r * C) : nullptr;
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int i = lane + k * 64;
      if (i < c4) {
        float4 t = x4[i];
        if (bias) {
          const float4 b = reinterpret_cast<const float4*>(bias)[i];
          t.x += b.x;
          t.y += b.y;
          t.z += b.z;
          t.w += b.w;
        }
        if (r4) {
          const float4 q = r4[i];
          t.x += q.x;
          t.y += q.y;
          t.z += q.z;
          t.w += q.w;
        }
        v[k] = t;
        s += (t.x + t.y) + (t.z + t.w);
      } else {
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float mean = wave_sum(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int i = lane + k * 64;
      if (i < c4) {
        const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
    float4* y4 = reinterpret_cast<float4*>(y + r * C);
    // 32-bit modulo (rows < 2^31 is checked on the host): a 64-bit one costs ~100 instructions
    const float4* p4 = y_plus ? reinterpret_cast<const float4*>(
                                    pos + (long long)((unsigned)r % (unsigned)pos_rows) * C)
                              : nullptr;
    float4* yp4 = y_plus ? reinterpret_cast<float4*>(y_plus + r * C) : nullptr;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int i = lane + k * 64;
      if (i < c4) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[i];
        const float4 b = reinterpret_cast<const float4*>(beta)[i];
        float4 o;
        o.x = (v[k].x - mean) * rstd * g.x + b.x;
        o.y = (v[k].y - mean) * rstd * g.y + b.y;
        o.z = (v[k].z - mean) * rstd * g.z + b.z;
        o.w = (v[k].w - mean) * rstd * g.w + b.w;
        y4[i] = o;
        if (yp4) {
          const float4 pp = p4[i];
          yp4[i] = make_float4(o.x + pp.x, o.y + pp.y, o.z + pp.z, o.w + pp.w);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Encoder (GRID, T = 1), head-major work split.  The kernel above gives a wave ONE query and all
// 8 heads: its 64 corner rows per point set lie in 8 different 128-byte head rows per pixel, so
// the ~32 queries resident on a CU touch ~100 KB of value rows -- three times the 32-KB L1
// (measured: 59 % L1 hit rate, 10 TB/s of L2 -> L1 traffic).  Here a wave takes 8 QUERIES of one
// head (lane = (query slot lane>>3, sub lane&7)) and a workgroup 32 neighbouring queries of that
// head, so the rows a workgroup gathers are one head's rows of a small pixel neighbourhood
// (~12 KB) and repeat across its queries and points while they are in L1.
// Logical block = (patch of 32 units, head); same FusedParams / outputs as the kernel above.
// ---------------------------------------------------------------------------
// kHmPatches: patches of 32 units a workgroup walks (same head); UNR: gather-loop unroll
template <int kHmPatches, int UNR>
__global__ __launch_bounds__(256) void enc_head_major_kernel(const FusedParams p) {
  constexpr int kSlots = 16;
  constexpr int kGroupStride = kSlots * 8 + 8;  // floats
  __shared__ __attribute__((aligned(16))) float lds[4 * 8 * kGroupStride];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 3, j = lane & 7;
  const int lb = xcd_remap(blockIdx.x, p.n_blocks_logical);
  const int head = lb & 7, pgroup = lb >> 3;
  const int L = p.L, LP = L * 4;  // 4 points per level, LP <= 16
  const int rowbytes = kRowFloats * 4;
  float* my_lds = lds + (wave * 8 + g) * kGroupStride;

  int Hs[4], Ws[4], St[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const int ll = l < L ? l : 0;
    Hs[l] = (int)p.shapes[2 * ll];
    Ws[l] = (int)p.shapes[2 * ll + 1];
    St[l] = (int)p.lsi[ll];
  }
  const int i0 = j * 2;                 // this lane prepares points i0, i0 + 1 (same level)
  const int lvl = i0 >> 2;
  const int H = lvl == 0 ? Hs[0] : lvl == 1 ? Hs[1] : lvl == 2 ? Hs[2] : Hs[3];
  const int W = lvl == 0 ? Ws[0] : lvl == 1 ? Ws[1] : lvl == 2 ? Ws[2] : Ws[3];
  const int st = lvl == 0 ? St[0] : lvl == 1 ? St[1] : lvl == 2 ? St[2] : St[3];
  const bool pts_ok = i0 < LP;

  // Two-deep software pipeline over the workgroup's patches: the unit index of patch it + 2
  // and the projections / reference point of patch it + 1 are loaded while patch it gathers.
  struct Pre {
    float2 lg, rf;
    float4 of;
  };
  auto slot_of = [&](int it) { return (pgroup * kHmPatches + it) * 32 + wave * 8 + g; };
  auto unit_of = [&](int it) {
    const int s = slot_of(it);
    return (it < kHmPatches && s < p.n_units) ? (p.order ? p.order[s] : s) : 0;
  };
  auto prefetch = [&](int unit) {
    Pre q;
    const float* row = p.proj + (long long)unit * p.proj_stride;
    const int ii = pts_ok ? i0 : 0;
    q.lg = *reinterpret_cast<const float2*>(row + kHeads * LP * 2 + head * LP + ii);
    q.of = *reinterpret_cast<const float4*>(row + head * LP * 2 + 2 * ii);
    q.rf = *reinterpret_cast<const float2*>(
        p.ref + ((long long)unit * p.ref_L + ((pts_ok && p.ref_L != 1) ? lvl : 0)) * 2);
    return q;
  };
  int unit = unit_of(0);
  int unit_n = unit_of(1);
  Pre cur = prefetch(unit);
  for (int it = 0; it < kHmPatches; ++it) {
    const int unit_nn = unit_of(it + 2);
    const Pre nxt = prefetch(unit_n);
    const bool active = slot_of(it) < p.n_units;
    // softmax statistics over the LP logits of (unit, head): 2 per lane
    const float x0 = pts_ok ? cur.lg.x : -INFINITY, x1 = pts_ok ? cur.lg.y : -INFINITY;
    const float mx = group8_max(fmaxf(x0, x1));
    const float e0 = pts_ok ? expf(x0 - mx) : 0.f, e1 = pts_ok ? expf(x1 - mx) : 0.f;
    const float sm = group8_sum(e0 + e1);
    const float inv_sum = 1.f / sm;
    const int clip = p.unit_clip ? p.unit_clip[unit] : unit / p.units_per_clip;
    const char* frame =
        reinterpret_cast<const char*>(p.value) + (long long)clip * p.S * rowbytes + j * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Corners c;
      if (pts_ok) {
        const float ox = s ? cur.of.z : cur.of.x, oy = s ? cur.of.w : cur.of.y;
        const float aw = (s ? e1 : e0) * inv_sum;
        const float lx = cur.rf.x + ox / (float)W, ly = cur.rf.y + oy / (float)H;
        c = make_corners(lx * (float)W - 0.5f, ly * (float)H - 0.5f, H, W, st, head * kDim * 4,
                         rowbytes, aw);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          c.w[k] = 0.f;
          c.o[k] = head * kDim * 4;
        }
      }
      float4* dst = reinterpret_cast<float4*>(my_lds + (i0 + s) * 8);
      dst[0] = make_float4(c.w[0], c.w[1], c.w[2], c.w[3]);
      dst[1] = make_float4(__int_as_float(c.o[0]), __int_as_float(c.o[1]), __int_as_float(c.o[2]),
                           __int_as_float(c.o[3]));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll UNR
    for (int i = 0; i < kSlots; ++i) {
      const float4* src = reinterpret_cast<const float4*>(my_lds + i * 8);
      const float4 w = src[0];
      const float4 o = src[1];
      const float4 v0 = ld16(frame, (unsigned)__float_as_int(o.x));
      const float4 v1 = ld16(frame, (unsigned)__float_as_int(o.y));
      const float4 v2 = ld16(frame, (unsigned)__float_as_int(o.z));
      const float4 v3 = ld16(frame, (unsigned)__float_as_int(o.w));
      fma4(acc, w.x, v0);
      fma4(acc, w.y, v1);
      fma4(acc, w.z, v2);
      fma4(acc, w.w, v3);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // descriptors consumed before the next patch overwrites
    if (active) {
      *reinterpret_cast<float4*>(p.out + (long long)unit * kRowFloats + head * kDim + j * 4) = acc;
      if (p.stat_max && j == 0) {
        p.stat_max[(long long)unit * kHeads + head] = mx;
        p.stat_sum[(long long)unit * kHeads + head] = sm;
      }
    }
    unit = unit_n;
    unit_n = unit_nn;
    cur = nxt;
  }
}

template <int PATCHES, int UNR>
int launch_enc_head_major_t(const FusedParams& p0, hipStream_t stream) {
  FusedParams p = p0;
  const int npatch = (p.n_units + 31) / 32;
  const int nb = ((npatch + PATCHES - 1) / PATCHES) * 8;
  p.n_blocks_logical = nb;
  hipLaunchKernelGGL((enc_head_major_kernel<PATCHES, UNR>), dim3(nb), dim3(256), 0, stream, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int launch_enc_head_major(const FusedParams& p, hipStream_t stream) {
  // 2 patches per workgroup, gather loop unrolled by 4: the best of {2,4,8,16} x {2,4} on the
  // bench workload (1.51-1.64 ms; all within 8 %)
  return launch_enc_head_major_t<2, 4>(p, stream);
}

template <int MODE, int PPL, int WQ>
int launch_fused(const FusedParams& p0, hipStream_t stream) {
  FusedParams p = p0;
  constexpr int kUnitsPerBlock = 4 / WQ;
  const int nb = (p.n_units + kUnitsPerBlock - 1) / kUnitsPerBlock;
  p.n_blocks_logical = nb;
  hipLaunchKernelGGL((fused_deform_attn_kernel<MODE, PPL, WQ>), dim3(nb), dim3(256), 0, stream, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

template <typename scalar_t>
int msda_forward_impl(const scalar_t* value, const int64_t* shapes, const int64_t* lsi,
                      const scalar_t* loc, const scalar_t* attw, scalar_t* out, int bs, int S,
                      int M, int D, int L, int Lq, int P, int im2col_step, void* stream) {
  if (!value || !shapes || !lsi || !loc || !attw || !out)
    return fail(PAVE_E_ARG, "ms_deform_attn_forward: null pointer");
  if (bs <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0 || im2col_step <= 0)
    return fail(PAVE_E_ARG, "ms_deform_attn_forward: sizes must be positive");
  if (L > kMaxLevels) return fail(PAVE_E_ARG, "ms_deform_attn_forward: more than 8 levels");
  const int step = bs < im2col_step ? bs : im2col_step;
  if (bs % step != 0)
    return fail(PAVE_E_STEP, "ms_deform_attn_forward: batch must be divisible by im2col_step");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long n = (long long)bs * Lq * M * D;
  if ((long long)S * M * D * (long long)sizeof(scalar_t) >= (1ll << 31))
    return fail(PAVE_E_ARG, "ms_deform_attn_forward: one value slab must be < 2 GiB");
  bool launched = false;
  if constexpr (sizeof(scalar_t) == 4) {
    const int G = D / 4;
    const long long ngroups = (long long)bs * Lq * M;
    if (D % 4 == 0 && (G & (G - 1)) == 0 && G <= 64) {
      const float* v = reinterpret_cast<const float*>(value);
      const float* lc = reinterpret_cast<const float*>(loc);
      const float* aw = reinterpret_cast<const float*>(attw);
      float* o = reinterpret_cast<float*>(out);
#define PAVE_LAUNCH_VEC(GG)                                                                     \
  {                                                                                             \
    const long long per = 256 / GG;                                                             \
    long long nb = (ngroups + per - 1) / per;                                                   \
    if (nb > 65536 * 4) nb = 65536 * 4;                                                         \
    hipLaunchKernelGGL((msda_fwd_vec_kernel<GG>), dim3((unsigned)nb), dim3(256), 0, st, ngroups, \
                       v, shapes, lsi, lc, aw, S, M, D, L, Lq, P, o);                           \
    launched = true;                                                                            \
  }
      switch (G) {
        case 1: PAVE_LAUNCH_VEC(1) break;
        case 2: PAVE_LAUNCH_VEC(2) break;
        case 4: PAVE_LAUNCH_VEC(4) break;
        case 8: PAVE_LAUNCH_VEC(8) break;
        case 16: PAVE_LAUNCH_VEC(16) break;
        case 32: PAVE_LAUNCH_VEC(32) break;
        case 64: PAVE_LAUNCH_VEC(64) break;
      }
#undef PAVE_LAUNCH_VEC
    }
  }
  if (!launched) {
    long long nb = (n + 255) / 256;
    if (nb > 65536 * 4) nb = 65536 * 4;
    hipLaunchKernelGGL((msda_fwd_scalar_kernel<scalar_t>), dim3((unsigned)nb), dim3(256), 0, st, n,
                       value, shapes, lsi, loc, attw, S, M, D, L, Lq, P, out);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

// ---------------------------------------------------------------------------
// [R1] backward (ms_deform_attn_backward): restates the col2im family of
// ms_deform_attn_cuda_kernel.cuh:66-198, 256-801 with one decomposition for every D:
// a group of G lanes (fp32, D % 4 == 0: G = D/4 lanes x 4 channels; otherwise G = 1 lane looping
// over the channels) owns one (b, q, m).  grad_value is scattered with float atomics (the only
// cross-group sum); the channel sums of grad_sampling_loc / grad_attn_weight are reduced with
// xor-shuffles inside the group and written once, so those two outputs need no atomics and no
// shared-memory variants per D.
// ---------------------------------------------------------------------------
template <typename scalar_t, int G, int CPL>  // CPL channels per lane per step (4 or 1)
__global__ __launch_bounds__(256) void msda_bwd_kernel(
    const long long ngroups, const scalar_t* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const scalar_t* __restrict__ loc,
    const scalar_t* __restrict__ attw, const scalar_t* __restrict__ gout, const int S,
    const int M, const int D, const int L, const int Lq, const int P,
    scalar_t* __restrict__ gvalue, scalar_t* __restrict__ gloc, scalar_t* __restrict__ gattw) {
  constexpr int kGroupsPerBlock = 256 / G;
  const int sub = threadIdx.x % G;
  const long long row = (long long)M * D;
  for (long long g = (long long)blockIdx.x * kGroupsPerBlock + threadIdx.x / G; g < ngroups;
       g += (long long)gridDim.x * kGroupsPerBlock) {
    const int m = (int)(g % M);
    const long long b = g / ((long long)M * Lq);
    const scalar_t* vb = value + b * S * row + (long long)m * D;
    scalar_t* gvb = gvalue + b * S * row + (long long)m * D;
    const scalar_t* go = gout + g * D;
    for (int l = 0; l < L; ++l) {
      const long long start = lsi[l];
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      for (int p = 0; p < P; ++p) {
        const long long pi = (g * L + l) * P + p;
        const scalar_t loc_w = loc[2 * pi], loc_h = loc[2 * pi + 1];
        const scalar_t weight = attw[pi];
        const scalar_t h_im = loc_h * H - (scalar_t)0.5, w_im = loc_w * W - (scalar_t)0.5;
        scalar_t g_w = 0, g_x = 0, g_y = 0;
        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
          const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
          const int h_high = h_low + 1, w_high = w_low + 1;
          const scalar_t lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
          const bool ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= W - 1;
          const bool ok3 = h_high <= H - 1 && w_low >= 0, ok4 = h_high <= H - 1 && w_high <= W - 1;
          const long long o1 = (start + (long long)h_low * W + w_low) * row;
          const long long o2 = (start + (long long)h_low * W + w_high) * row;
          const long long o3 = (start + (long long)h_high * W + w_low) * row;
          const long long o4 = (start + (long long)h_high * W + w_high) * row;
          const scalar_t w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
          for (int c0 = sub * CPL; c0 < D; c0 += G * CPL) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
              const int c = c0 + k;
              const scalar_t top = go[c];
              const scalar_t tgv = top * weight;  // d out / d bilinear
              const scalar_t v1 = ok1 ? vb[o1 + c] : (scalar_t)0, v2 = ok2 ? vb[o2 + c] : (scalar_t)0;
              const scalar_t v3 = ok3 ? vb[o3 + c] : (scalar_t)0, v4 = ok4 ? vb[o4 + c] : (scalar_t)0;
              if (ok1) atomicAdd(gvb + o1 + c, w1 * tgv);
              if (ok2) atomicAdd(gvb + o2 + c, w2 * tgv);
              if (ok3) atomicAdd(gvb + o3 + c, w3 * tgv);
              if (ok4) atomicAdd(gvb + o4 + c, w4 * tgv);
              g_w += top * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
              g_x += tgv * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);  // d/dw of the bilinear form
              g_y += tgv * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);
            }
          }
        }
#pragma unroll
        for (int o = G >> 1; o > 0; o >>= 1) {
          g_w += __shfl_xor(g_w, o);
          g_x += __shfl_xor(g_x, o);
          g_y += __shfl_xor(g_y, o);
        }
        if (sub == 0) {
          gattw[pi] = g_w;
          gloc[2 * pi] = g_x * W;
          gloc[2 * pi + 1] = g_y * H;
        }
      }
    }
  }
}

template <typename scalar_t>
int msda_backward_impl(const scalar_t* value, const int64_t* shapes, const int64_t* lsi,
                       const scalar_t* loc, const scalar_t* attw, const scalar_t* gout,
                       scalar_t* gvalue, scalar_t* gloc, scalar_t* gattw, int bs, int S, int M,
                       int D, int L, int Lq, int P, int im2col_step, void* stream) {
  if (!value || !shapes || !lsi || !loc || !attw || !gout || !gvalue || !gloc || !gattw)
    return fail(PAVE_E_ARG, "ms_deform_attn_backward: null pointer");
  if (bs <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0 || im2col_step <= 0)
    return fail(PAVE_E_ARG, "ms_deform_attn_backward: sizes must be positive");
  const int step = bs < im2col_step ? bs : im2col_step;
  if (bs % step != 0)
    return fail(PAVE_E_STEP, "ms_deform_attn_backward: batch must be divisible by im2col_step");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long ngroups = (long long)bs * Lq * M;
  bool done = false;
#define PAVE_BWD(T_, G_, C_)                                                                     \
  {                                                                                               \
    long long nb = (ngroups + (256 / G_) - 1) / (256 / G_);                                       \
    if (nb > 65536 * 4) nb = 65536 * 4;                                                           \
    hipLaunchKernelGGL((msda_bwd_kernel<T_, G_, C_>), dim3((unsigned)nb), dim3(256), 0, st,      \
                       ngroups, value, shapes, lsi, loc, attw, gout, S, M, D, L, Lq, P, gvalue,   \
                       gloc, gattw);                                                              \
    done = true;                                                                                  \
  }
  if constexpr (sizeof(scalar_t) == 4) {
    if (D % 4 == 0) {
      const int G = D / 4;
      if (G == 8) PAVE_BWD(float, 8, 4)
      else if (G == 16) PAVE_BWD(float, 16, 4)
      else if (G == 4) PAVE_BWD(float, 4, 4)
      else if (G == 2) PAVE_BWD(float, 2, 4)
      else if (G == 1) PAVE_BWD(float, 1, 4)
    }
  }
  if (!done) PAVE_BWD(scalar_t, 1, 1)
#undef PAVE_BWD
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

// ---------------------------------------------------------------------------
// Input pipeline on the device: for T frames at once,
// keep-ratio bilinear resize (cv2.INTER_LINEAR float semantics: half-pixel centres, edge
// clamp), BGR->RGB, (x - mean) / std, zero pad to the batch canvas, HWC -> CHW.
// Restates mmdet Resize / Normalize / Pad / MulImageToTensor
// (configs/_base_/datasets/posetrack17_video_keypoint.py:71-84, mmcv/image/geometric.py
// imresize + photometric.py imnormalize).  One thread per output pixel (3 channels).
// ---------------------------------------------------------------------------
template <typename src_t>
__global__ __launch_bounds__(256) void preprocess_frames_kernel(
    const src_t* __restrict__ src, float* __restrict__ dst, const int T, const int H0,
    const int W0, const int Hn, const int Wn, const int Hp, const int Wp, const float m0,
    const float m1, const float m2, const float s0, const float s1, const float s2,
    const int to_rgb) {
#pragma clang fp contract(off)   // every product and sum below is rounded on its own (no FMA)
  const long long n = (long long)T * Hp * Wp;
  // OpenCV's float INTER_LINEAR (mmcv.imresize -> cv2.resize on the to_float32 image,
  // mmcv/image/geometric.py:63-107) in its published arithmetic order: inv_scale = dst / src and
  // scale = 1 / inv_scale in DOUBLE, source coordinate (dx + 0.5) * scale - 0.5 in double, cast to
  // float, floor, fraction in float, border clamp with the fraction zeroed; the horizontal pass
  // (a * (1 - fx) + b * fx) then the vertical one, each product and sum rounded to float on its
  // own (no fused multiply-add: contraction is switched off for this function).
  const double scx = 1.0 / ((double)Wn / (double)W0), scy = 1.0 / ((double)Hn / (double)H0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const int y = (int)((i / Wp) % Hp);
    const int t = (int)(i / ((long long)Wp * Hp));
    float c[3] = {0.f, 0.f, 0.f};
    if (x < Wn && y < Hn) {
      float fx = (float)(((double)x + 0.5) * scx - 0.5), fy = (float)(((double)y + 0.5) * scy - 0.5);
      int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
      fx = fx - (float)x0;
      fy = fy - (float)y0;
      if (x0 < 0) { x0 = 0; fx = 0.f; }
      if (x0 >= W0 - 1) { x0 = W0 - 1; fx = 0.f; }
      if (y0 < 0) { y0 = 0; fy = 0.f; }
      if (y0 >= H0 - 1) { y0 = H0 - 1; fy = 0.f; }
      const int x1 = min(x0 + 1, W0 - 1), y1 = min(y0 + 1, H0 - 1);
      const float gx = 1.f - fx, gy = 1.f - fy;
      const src_t* f = src + (long long)t * H0 * W0 * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float a = (float)f[((long long)y0 * W0 + x0) * 3 + k];
        const float b = (float)f[((long long)y0 * W0 + x1) * 3 + k];
        const float cc = (float)f[((long long)y1 * W0 + x0) * 3 + k];
        const float d = (float)f[((long long)y1 * W0 + x1) * 3 + k];
        const float t0 = a * gx, t1 = b * fx, b0 = cc * gx, b1 = d * fx;
        const float top = t0 + t1, bot = b0 + b1;
        const float u0 = top * gy, u1 = bot * fy;
        c[k] = u0 + u1;
      }
      if (to_rgb) {
        const float tmp = c[0];
        c[0] = c[2];
        c[2] = tmp;
      }
      c[0] = (c[0] - m0) * s0;   // mmcv.imnormalize: subtract, then multiply
      c[1] = (c[1] - m1) * s1;
      c[2] = (c[2] - m2) * s2;
    }
    const long long plane = (long long)Hp * Wp;
    float* o = dst + (long long)t * 3 * plane + (long long)y * Wp + x;
    o[0] = c[0];
    o[plane] = c[1];
    o[2 * plane] = c[2];
  }
}

// ---------------------------------------------------------------------------
// KSxKS convolution (KS = 3: pad 1, stride 1|2;  KS = 1: a row GEMM), NHWC fp32, as an implicit
// GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD) with
// bias (+residual) (+ReLU) in the epilogue.
//   C[M, Cout] = A[M, KS*KS*Cin] * B[KS*KS*Cin, Cout],  M = N*Ho*Wo;  A is gathered on the fly:
//   for tap (ky, kx) and channel slab c0 the 32-channel segment of input pixel
//   (oy*s + ky - pad, ox*s + kx - pad) is 128 contiguous bytes (zero outside the image).
// Block = 256 threads, tile 128 (pixels) x BN (64 | 128 output channels), K-slab 32.
// LDS: A[128][33] (row stride 33 floats: the 32 lanes of an MFMA operand read 32 different
// rows at one k -> conflict-free) and B[32][BN].  Wave tile: 64x64 (BN=128: 2x2 waves) or
// 32x64 (BN=64: 4x1 waves) as 32x32 accumulator tiles.
// Grid is 1-D with the Cout tile fastest, so the blocks sharing an A tile run back to back.
// Measured (tools/bench_conv.py, 28 frames): 87-100 TFLOP/s on the R-50 3x3 shapes, MIOpen's
// searched fp32 kernels reach 110-128 there, so the model keeps MIOpen for 3x3; the KS = 1
// instance with the residual epilogue replaces hipBLASLt GEMM + a separate bias/ReLU pass on the
// bandwidth-bound 1x1 expansions of ResNet layer1/layer2 (K = 64 / 128).
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kConvSmemFloats = 128 * 68;  // >= operand tiles (128*33 + 32*128) and epilogue chunk

template <int BN, int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_nhwc_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* residual, float* y, const int N, const int H, const int W, const int Cin,
    const int Cout, const int Ho, const int Wo, const int stride, const int relu,
    const float* __restrict__ a_bias, const float* __restrict__ x2, const int Cin2) {
  // KS == 1 extras: the K axis may continue into a second row matrix x2 [M, Cin2] (the
  // Bottleneck's downsample branch shares the accumulator), and the first source may get
  // relu(x + a_bias[k]) applied on load (the producer's folded-BN bias + ReLU).
  constexpr int PAD = KS / 2, TAPS = KS * KS;
  constexpr int BM = 128, BK = 32, AST = 33;
  constexpr int TM = (BN == 128) ? 2 : 1, TN = 2;
  constexpr int WM = TM * 32, WN = TN * 32;
  __shared__ __attribute__((aligned(16))) float smem[kConvSmemFloats];
  float* As = smem;                  // [BM][AST]
  float* Bs = smem + BM * AST;       // [BK][BN]   (BM*AST*4 is a multiple of 16)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (BN == 128) ? (wave >> 1) * WM : wave * WM;
  const int wn0 = (BN == 128) ? (wave & 1) * WN : 0;
  const long long M = (long long)N * Ho * Wo;
  const int ntiles = Cout / BN;
  const long long m0 = (long long)(blockIdx.x / ntiles) * BM;
  const int n0 = (blockIdx.x % ntiles) * BN;

  // the 4 pixels whose 16-byte segment `seg` this thread stages per K-slab
  const int seg = tid & 7;
  int iy0[4], ix0[4];
  long long nbase[4];
  bool pvalid[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long long gm = m0 + (tid >> 3) + q * 32;
    pvalid[q] = gm < M;
    const unsigned g = pvalid[q] ? (unsigned)gm : 0u;  // host guarantees M < 2^31
    const unsigned gy = g / (unsigned)Wo;
    const int ox = (int)(g - gy * (unsigned)Wo);
    const int n = (int)(gy / (unsigned)Ho);
    const int oy = (int)(gy - (unsigned)n * (unsigned)Ho);
    iy0[q] = oy * stride - PAD;
    ix0[q] = ox * stride - PAD;
    nbase[q] = (long long)n * H * W;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lrow = lane & 31, lk = lane >> 5;
  constexpr int BV = (BK * BN) / (256 * 4);  // float4 per thread for B
  const int cslabs = Cin / BK;
  const int nslabs = TAPS * cslabs + (KS == 1 ? Cin2 / BK : 0);
  float4 av[4], bv[BV], abv = make_float4(0.f, 0.f, 0.f, 0.f);
  bool abv_on = false;
  // global -> registers for one K-slab (tap, 32 channels): issued one slab ahead so that the
  // loads are in flight while the MFMAs of the current slab run
#define PAVE_CONV_LOAD_SLAB(slab_)                                                              \
  {                                                                                             \
    /* KS == 1: "tap" 0 = first source, 1 = second source (its slabs follow the first's) */      \
    const int tap_ = (KS == 1) ? ((slab_) >= cslabs ? 1 : 0) : (slab_) / cslabs;                 \
    const int c0_ = ((slab_) - tap_ * cslabs) * BK;                                              \
    const int ky_ = (KS == 1) ? 0 : tap_ / KS, kx_ = (KS == 1) ? 0 : tap_ - ky_ * KS;            \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
      const int iy = iy0[q] + ky_, ix = ix0[q] + kx_;                                            \
      const bool ok = pvalid[q] && iy >= 0 && iy < H && ix >= 0 && ix < W;                       \
      const long long pix_ = nbase[q] + (long long)iy * W + ix;                                  \
      const float* src_ = !ok ? x                                                                \
                          : (KS == 1 && tap_ > 0) ? x2 + pix_ * Cin2 + c0_ + seg * 4             \
                                                  : x + pix_ * Cin + c0_ + seg * 4;              \
      const float4 t_ = *reinterpret_cast<const float4*>(src_);                                  \
      av[q] = ok ? t_ : make_float4(0.f, 0.f, 0.f, 0.f);                                         \
    }                                                                                           \
    /* A-side bias of this slab; applied when the slab is written to LDS, so that the loads */  \
    /* above stay in flight across the MFMA loop (rows >= M get relu(bias): never stored)   */  \
    abv_on = KS == 1 && a_bias && tap_ == 0;                                                     \
    if (abv_on) abv = *reinterpret_cast<const float4*>(a_bias + c0_ + seg * 4);                  \
    const float* wrow_ = w + ((long long)tap_ * Cin + c0_) * Cout + n0; /* rows of [K(+K2), N] */ \
    _Pragma("unroll") for (int q = 0; q < BV; ++q) {                                            \
      const int idx = tid + q * 256; /* float4 index in the [32][BN] tile */                     \
      const int kr = idx / (BN / 4), nc = (idx - kr * (BN / 4)) * 4;                             \
      bv[q] = *reinterpret_cast<const float4*>(wrow_ + (long long)kr * Cout + nc);               \
    }                                                                                           \
  }
  PAVE_CONV_LOAD_SLAB(0)
  for (int slab = 0; slab < nslabs; ++slab) {
    __syncthreads();  // previous slab fully consumed
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* d = As + ((tid >> 3) + q * 32) * AST + seg * 4;
      float4 t = av[q];
      if (abv_on) {
        t.x = fmaxf(t.x + abv.x, 0.f);
        t.y = fmaxf(t.y + abv.y, 0.f);
        t.z = fmaxf(t.z + abv.z, 0.f);
        t.w = fmaxf(t.w + abv.w, 0.f);
      }
      d[0] = t.x;
      d[1] = t.y;
      d[2] = t.z;
      d[3] = t.w;
    }
#pragma unroll
    for (int q = 0; q < BV; ++q) *reinterpret_cast<float4*>(Bs + (tid + q * 256) * 4) = bv[q];
    __syncthreads();
    if (slab + 1 < nslabs) PAVE_CONV_LOAD_SLAB(slab + 1)
    // ---- 16 k-steps of 2
#pragma unroll 4
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(wm0 + i * 32 + lrow) * AST + kk * 2 + lk];
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) b[jn] = Bs[(kk * 2 + lk) * BN + wn0 + jn * 32 + lrow];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[jn], acc[i][jn], 0, 0, 0);
    }
  }
  // ---- epilogue.  The accumulators (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) +
  // 4*(lane>>5)) go through LDS in chunks of CR rows so that bias / residual / ReLU / store run
  // on float4 with whole BN*4-byte row segments per wave (4-byte stores straight from the MFMA
  // layout reach only half the HBM write rate).
  constexpr int CST = BN + 4;             // chunk row stride (floats), keeps float4 alignment
  constexpr int CR = (BN == 128) ? 64 : 128;
  constexpr int RV = BN / 4;              // float4 per chunk row
  constexpr int RPP = 256 / RV;           // rows per pass of the block
  constexpr int NPASS = CR / RPP;         // = 8
  static_assert(CR * CST <= kConvSmemFloats, "epilogue chunk must fit the staging buffer");
  float* Cs = smem;
  const int g = wm0 / WM;                 // wave row group
  const int c4 = tid % RV;
  const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + n0 + c4 * 4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __syncthreads();  // operand tiles (i = 0) / previous chunk (i > 0) fully consumed
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        Cs[lr * CST + wn0 + jn * 32 + lrow] = acc[i][jn][r];
      }
    __syncthreads();
    float4 res[NPASS];
    if (residual) {  // all loads in flight before the first store
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int lr = ps * RPP + tid / RV;
        const long long gm = m0 + (lr >> 5) * WM + i * 32 + (lr & 31);
        res[ps] = gm < M ? *reinterpret_cast<const float4*>(residual + gm * Cout + n0 + c4 * 4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int lr = ps * RPP + tid / RV;
      const long long gm = m0 + (lr >> 5) * WM + i * 32 + (lr & 31);
      if (gm < M) {
        float4 v = *reinterpret_cast<const float4*>(Cs + lr * CST + c4 * 4);
        v.x += b4.x;
        v.y += b4.y;
        v.z += b4.z;
        v.w += b4.w;
        if (residual) {
          v.x += res[ps].x;
          v.y += res[ps].y;
          v.z += res[ps].z;
          v.w += res[ps].w;
        }
        if (relu) {
          v.x = fmaxf(v.x, 0.f);
          v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f);
          v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<float4*>(y + gm * Cout + n0 + c4 * 4) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// ResNet stem tail in one pass: y = maxpool3x3/s2/p1(relu(x + bias)) on an NHWC map
// (= relu(max(x) + bias): bias is per channel and ReLU is monotonic).  One thread = one
// output pixel x 4 channels; the 16 lanes of a 64-channel pixel read 256 contiguous bytes.
// Replaces a bias/ReLU pass (read + write of the 400x672 map) + MaxPool2d (resnet.py:640-645).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_relu_maxpool_kernel(
    const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y,
    const int N, const int H, const int W, const int C, const int Ho, const int Wo) {
  const int c4 = C >> 2;
  const long long total = (long long)N * Ho * Wo * c4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    const long long pix = i / c4;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const long long n = pix / ((long long)Wo * Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const float4 v =
            *reinterpret_cast<const float4*>(x + ((n * H + iy) * W + ix) * C + c);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
    m.x = fmaxf(m.x + b.x, 0.f);
    m.y = fmaxf(m.y + b.y, 0.f);
    m.z = fmaxf(m.z + b.z, 0.f);
    m.w = fmaxf(m.w + b.w, 0.f);
    *reinterpret_cast<float4*>(y + pix * C + c) = m;
  }
}

// ---------------------------------------------------------------------------
// GroupNorm over an NHWC map (the ChannelMapper neck: conv -> GN(32), necks/channel_mapper.py:
// 90-100, mmcv ConvModule), written straight into a row-strided destination (the transformer's
// flattened multi-level buffer).  Three launches, all deterministic: (1) per (image, row chunk)
// partial sums / sums of squares per group, accumulated in fp64; (2) per image: mean / rstd per
// group -> per-channel scale a = rstd * gamma and shift b = beta - mean * a; (3) y = x * a + b.
// HBM-bound: the map is read twice and written once.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(
    const float* __restrict__ x, double* __restrict__ partial, const int HW, const int C,
    const int G, const int nchunks) {
  __shared__ double sh[256][2];
  const int tpr = C >> 2;                 // threads per row (one float4 each)
  const int rpp = 256 / tpr;              // rows per pass
  const int c4 = threadIdx.x % tpr, r0 = threadIdx.x / tpr;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int rows_per_chunk = (HW + nchunks - 1) / nchunks;
  const int rbeg = chunk * rows_per_chunk;
  const int rend = min(HW, rbeg + rows_per_chunk);
  const float* xb = x + (long long)n * HW * C + c4 * 4;
  double s = 0.0, q = 0.0;
  if (r0 < rpp)
    for (int r = rbeg + r0; r < rend; r += rpp) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (long long)r * C);
      s += (double)((v.x + v.y) + (v.z + v.w));
      q += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
    }
  sh[threadIdx.x][0] = (r0 < rpp) ? s : 0.0;
  sh[threadIdx.x][1] = (r0 < rpp) ? q : 0.0;
  __syncthreads();
  if (threadIdx.x < G) {                  // fixed summation order: bit-reproducible
    const int tpg = (C / G) >> 2;         // float4 lanes per group
    double ts = 0.0, tq = 0.0;
    for (int r = 0; r < rpp; ++r)
      for (int j = 0; j < tpg; ++j) {
        const int t = r * tpr + threadIdx.x * tpg + j;
        ts += sh[t][0];
        tq += sh[t][1];
      }
    double* o = partial + (((long long)n * nchunks + chunk) * G + threadIdx.x) * 2;
    o[0] = ts;
    o[1] = tq;
  }
}

__global__ __launch_bounds__(256) void groupnorm_finalize_kernel(
    const double* __restrict__ partial, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ ab, const int HW, const int C, const int G,
    const int nchunks, const float eps) {
  __shared__ float mean_s[256], rstd_s[256];
  const int n = blockIdx.x;
  for (int g = threadIdx.x; g < G; g += 256) {
    // (16 chunks' sums in flight per step, added in chunk order: the dependent-load form spent a memory latency
    // per chunk -- 50 us at 128 chunks)
    double ts = 0.0, tq = 0.0;
    for (int ch0 = 0; ch0 < nchunks; ch0 += 16) {
      double a[16], b[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double* o = partial + (((long long)n * nchunks + min(ch0 + j, nchunks - 1)) * G + g) * 2;
        a[j] = o[0], b[j] = o[1];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (ch0 + j < nchunks) ts += a[j], tq += b[j];
    }
    const double cnt = (double)HW * (double)(C / G);
    const double mean = ts / cnt;
    double var = tq / cnt - mean * mean;  // fp64: no visible cancellation at fp32 data
    var = var < 0.0 ? 0.0 : var;
    mean_s[g] = (float)mean;
    rstd_s[g] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / (C / G);
    const float a = rstd_s[g] * gamma[c];
    ab[((long long)n * 2) * C + c] = a;
    ab[((long long)n * 2 + 1) * C + c] = beta[c] - mean_s[g] * a;
  }
}

__global__ __launch_bounds__(256) void groupnorm_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ ab, float* __restrict__ y,
    const long long y_batch_stride, const int N, const int HW, const int C) {
  const int tpr = C >> 2;
  const long long total = (long long)N * HW * tpr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % tpr);
    const long long row = i / tpr;
    const int n = (int)(row / HW);
    const long long r = row - (long long)n * HW;
    const float4 v = *reinterpret_cast<const float4*>(x + row * C + c4 * 4);
    const float4 a = *reinterpret_cast<const float4*>(ab + ((long long)n * 2) * C + c4 * 4);
    const float4 b = *reinterpret_cast<const float4*>(ab + ((long long)n * 2 + 1) * C + c4 * 4);
    float4 o;
    o.x = fmaf(v.x, a.x, b.x);
    o.y = fmaf(v.y, a.y, b.y);
    o.z = fmaf(v.z, a.z, b.z);
    o.w = fmaf(v.w, a.w, b.w);
    *reinterpret_cast<float4*>(y + (long long)n * y_batch_stride + r * C + c4 * 4) = o;
  }
}

// The same three passes over up to four maps at once (the neck's levels share N, C and G and differ in their
// HW): one launch per pass instead of one per pass and level -- the small levels' launches are latency, not
// work.  Every block finds its level from the cumulative block counts and then runs the single-level body on
// it (same chunks, same summation order: the values are those of pave_groupnorm_nhwc_f32 bit for bit).
struct GnLevels {
  const float* x[4]; const float* gamma[4]; const float* beta[4]; float* y[4];
  long long y_batch_stride[4];
  long long partial_off[4];   // doubles
  int HW[4], nchunks[4];
  int chunk_end[4];           // cumulative chunk counts (stats grid)
  int apply_end[4];           // cumulative block counts (apply grid)
  float eps[4];
  int nlev;
};
__global__ __launch_bounds__(256) void groupnorm_levels_stats_kernel(const GnLevels p, double* __restrict__ partial,
                                                                    const int C, const int G) {
  __shared__ double sh[256][2];
  int l = 0;
  while (l + 1 < p.nlev && (int)blockIdx.x >= p.chunk_end[l]) ++l;
  const int chunk = (int)blockIdx.x - (l ? p.chunk_end[l - 1] : 0);
  const int HW = p.HW[l], nchunks = p.nchunks[l];
  const int tpr = C >> 2;
  const int rpp = 256 / tpr;
  const int c4 = threadIdx.x % tpr, r0 = threadIdx.x / tpr;
  const int n = blockIdx.y;
  const int rows_per_chunk = (HW + nchunks - 1) / nchunks;
  const int rbeg = chunk * rows_per_chunk;
  const int rend = min(HW, rbeg + rows_per_chunk);
  const float* xb = p.x[l] + (long long)n * HW * C + c4 * 4;
  double s = 0.0, q = 0.0;
  if (r0 < rpp)
    for (int r = rbeg + r0; r < rend; r += rpp) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (long long)r * C);
      s += (double)((v.x + v.y) + (v.z + v.w));
      q += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
    }
  sh[threadIdx.x][0] = (r0 < rpp) ? s : 0.0;
  sh[threadIdx.x][1] = (r0 < rpp) ? q : 0.0;
  __syncthreads();
  if (threadIdx.x < G) {
    const int tpg = (C / G) >> 2;
    double ts = 0.0, tq = 0.0;
    for (int r = 0; r < rpp; ++r)
      for (int j = 0; j < tpg; ++j) {
        const int t = r * tpr + threadIdx.x * tpg + j;
        ts += sh[t][0];
        tq += sh[t][1];
      }
    double* o = partial + p.partial_off[l] + (((long long)n * nchunks + chunk) * G + threadIdx.x) * 2;
    o[0] = ts;
    o[1] = tq;
  }
}
// grid (N, levels): ab [level][N][2][C]
__global__ __launch_bounds__(256) void groupnorm_levels_finalize_kernel(const GnLevels p,
                                                                       const double* __restrict__ partial,
                                                                       float* __restrict__ ab, const int N,
                                                                       const int C, const int G) {
  __shared__ float mean_s[256], rstd_s[256];
  const int n = blockIdx.x, l = blockIdx.y;
  const int HW = p.HW[l], nchunks = p.nchunks[l];
  for (int g = threadIdx.x; g < G; g += 256) {
    double ts = 0.0, tq = 0.0;
    const double* pl = partial + p.partial_off[l];
    for (int ch0 = 0; ch0 < nchunks; ch0 += 16) {   // (as groupnorm_finalize_kernel: chunk order, 16 loads in flight)
      double a[16], b[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double* o = pl + (((long long)n * nchunks + min(ch0 + j, nchunks - 1)) * G + g) * 2;
        a[j] = o[0], b[j] = o[1];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (ch0 + j < nchunks) ts += a[j], tq += b[j];
    }
    const double cnt = (double)HW * (double)(C / G);
    const double mean = ts / cnt;
    double var = tq / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    mean_s[g] = (float)mean;
    rstd_s[g] = (float)(1.0 / sqrt(var + (double)p.eps[l]));
  }
  __syncthreads();
  float* abl = ab + (long long)l * N * 2 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / (C / G);
    const float a = rstd_s[g] * p.gamma[l][c];
    abl[((long long)n * 2) * C + c] = a;
    abl[((long long)n * 2 + 1) * C + c] = p.beta[l][c] - mean_s[g] * a;
  }
}
__global__ __launch_bounds__(256) void groupnorm_levels_apply_kernel(const GnLevels p, const float* __restrict__ ab,
                                                                    const int N, const int C) {
  int l = 0;
  while (l + 1 < p.nlev && (int)blockIdx.x >= p.apply_end[l]) ++l;
  const int b0 = l ? p.apply_end[l - 1] : 0;
  const long long blk = (long long)blockIdx.x - b0, nblk = p.apply_end[l] - b0;
  const int HW = p.HW[l];
  const int tpr = C >> 2;
  const long long total = (long long)N * HW * tpr;
  const float* x = p.x[l];
  float* y = p.y[l];
  const float* abl = ab + (long long)l * N * 2 * C;
  for (long long i = blk * 256 + threadIdx.x; i < total; i += nblk * 256) {
    const int c4 = (int)(i % tpr);
    const long long row = i / tpr;
    const int n = (int)(row / HW);
    const long long r = row - (long long)n * HW;
    const float4 v = *reinterpret_cast<const float4*>(x + row * C + c4 * 4);
    const float4 a = *reinterpret_cast<const float4*>(abl + ((long long)n * 2) * C + c4 * 4);
    const float4 b = *reinterpret_cast<const float4*>(abl + ((long long)n * 2 + 1) * C + c4 * 4);
    float4 o;
    o.x = fmaf(v.x, a.x, b.x);
    o.y = fmaf(v.y, a.y, b.y);
    o.z = fmaf(v.z, a.z, b.z);
    o.w = fmaf(v.w, a.w, b.w);
    *reinterpret_cast<float4*>(y + (long long)n * p.y_batch_stride[l] + r * C + c4 * 4) = o;
  }
}

// out = sigmoid(tmp + inverse_sigmoid(ref)): the reference-point update after every decoder layer
// (OT:6733-6735, MT:865-866 with mmdet's inverse_sigmoid, eps = 1e-5) -- seven elementwise
// launches of the PyTorch formulation in one; same operation order in fp32.
__global__ __launch_bounds__(256) void ref_update_kernel(const float* __restrict__ tmp,
                                                         const float* __restrict__ ref,
                                                         float* __restrict__ out, const long long n,
                                                         const float eps) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (long long)gridDim.x * 256) {
    float x = fminf(fmaxf(ref[i], 0.f), 1.f);
    const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
    const float v = tmp[i] + logf(x1 / x2);
    out[i] = 1.f / (1.f + expf(-v));
  }
}

// HRNet stem conv1: 3x3 / stride 2 / pad 1 convolution of the NCHW image batch, 3 -> 64 channels,
// folded BatchNorm bias + ReLU, NHWC output (third_party/mmdetection/mmdet/models/backbones/
// hrnet.py:549-556: conv1 -> norm1 -> relu).  27 input taps per output pixel: far too thin for the
// matrix cores (K = 27) and HBM-bound anyway (1.93 GB written per 28 frames), so a lane owns one
// output pixel and all 64 channels -- 27 x 64 fp32 FMAs with the weights as scalar operands (one
// s_load burst per tap, uniform over the wave), 16 float4 stores of the pixel's 256-byte row.
// Taps are accumulated in (channel, ky, kx) order.
__global__ __launch_bounds__(256) void conv3x3s2_c3_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ w,   // [27][64]
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           const int N, const int H, const int W,
                                                           const int Ho, const int Wo, const int relu) {
  const long long total = (long long)N * Ho * Wo;
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= total) return;
  const int ox = (int)(pix % Wo);
  const long long t = pix / Wo;
  const int oy = (int)(t % Ho), n = (int)(t / Ho);
  float in[27];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        in[(c * 3 + ky) * 3 + kx] = ok ? x[(((long long)n * 3 + c) * H + iy) * W + ix] : 0.f;
      }
  float acc[64];
#pragma unroll
  for (int co = 0; co < 64; ++co) acc[co] = bias ? bias[co] : 0.f;
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int co = 0; co < 64; ++co) acc[co] = fmaf(in[k], w[k * 64 + co], acc[co]);
  float4* o = reinterpret_cast<float4*>(y + pix * 64);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    o[q] = v;
  }
}

}  // namespace

int pave_internal_fail(int code, const char* msg) { return fail(code, msg); }

extern "C" {

int pave_abi_version(void) { return PAVE_ABI_VERSION; }
const char* pave_last_error(void) { return g_err; }

int pave_ms_deform_attn_forward_f32(const float* value, const int64_t* spatial_shapes,
                                    const int64_t* level_start, const float* sampling_loc,
                                    const float* attn_weight, float* out, int bs, int S, int M,
                                    int D, int L, int Lq, int P, int im2col_step, void* stream) {
  return msda_forward_impl<float>(value, spatial_shapes, level_start, sampling_loc, attn_weight,
                                  out, bs, S, M, D, L, Lq, P, im2col_step, stream);
}

int pave_ms_deform_attn_forward_f64(const double* value, const int64_t* spatial_shapes,
                                    const int64_t* level_start, const double* sampling_loc,
                                    const double* attn_weight, double* out, int bs, int S, int M,
                                    int D, int L, int Lq, int P, int im2col_step, void* stream) {
  return msda_forward_impl<double>(value, spatial_shapes, level_start, sampling_loc, attn_weight,
                                   out, bs, S, M, D, L, Lq, P, im2col_step, stream);
}

int pave_deform_attn_grid_fused_f32(const float* value, const int64_t* spatial_shapes,
                                    const int64_t* level_start, const float* proj,
                                    const float* ref, const int32_t* unit_clip,
                                    const int32_t* order, float* out, float* stat_max,
                                    float* stat_sum, int n_units, int units_per_clip, int n_clips,
                                    int T, int S, int L, int P, int proj_stride,
                                    const int32_t* frame_table, int n_slabs, int ref_levels, void* stream) {
  if (!value || !spatial_shapes || !level_start || !proj || !ref || !out)
    return fail(PAVE_E_ARG, "deform_attn_grid_fused: null pointer");
  if (ref_levels != L && ref_levels != 1)
    return fail(PAVE_E_ARG, "deform_attn_grid_fused: ref_levels = L (a row per level) or 1 (one row for all levels)");
  if (frame_table ? n_slabs <= 0 : (n_slabs != 0 && n_slabs != n_clips * T))
    return fail(PAVE_E_ARG, "deform_attn_grid_fused: n_slabs = the frame slabs behind value (> 0 with a frame table; "
                            "0 or n_clips * T without)");
  if (n_units <= 0 || T <= 0 || S <= 0 || n_clips <= 0 || units_per_clip <= 0)
    return fail(PAVE_E_ARG, "deform_attn_grid_fused: sizes must be positive");
  if (L != 4 || P != 4)
    return fail(PAVE_E_ARG, "deform_attn_grid_fused: only L = 4, P = 4 is built");
  if ((stat_max == nullptr) != (stat_sum == nullptr))
    return fail(PAVE_E_ARG, "deform_attn_grid_fused: stat_max / stat_sum must both be given");
  if (proj_stride < T * kHeads * L * P * 3)
    return fail(PAVE_E_ARG, "deform_attn_grid_fused: proj_stride too small");
  if ((long long)S * kRowFloats * 4 >= (1ll << 31))
    return fail(PAVE_E_ARG, "deform_attn_grid_fused: one value slab must be < 2 GiB");
  if (!unit_clip && (long long)n_clips * units_per_clip < n_units)
    return fail(PAVE_E_ARG, "deform_attn_grid_fused: n_units exceeds n_clips * units_per_clip");
  FusedParams p{};
  p.value = value;
  p.shapes = spatial_shapes;
  p.lsi = level_start;
  p.proj = proj;
  p.ref = ref;
  p.unit_clip = unit_clip;
  p.order = order;
  p.frame_table = frame_table;
  p.n_slabs = frame_table ? n_slabs : n_clips * T;
  p.out = out;
  p.stat_max = stat_max;
  p.stat_sum = stat_sum;
  p.n_units = n_units;
  p.units_per_clip = units_per_clip;
  p.T = T;
  p.S = S;
  p.L = L;
  p.P = P;
  p.ref_L = ref_levels;
  p.proj_stride = proj_stride;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (T == 1 && P == 4 && L <= 4) {
    // head-major split; its float4 / float2 projection loads need 16-byte aligned rows, other
    // row strides take the per-query form of the same kernel below
    if (proj_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(proj) & 15) == 0)
      return launch_enc_head_major(p, st);
  }
  if (T == 1) return launch_fused<kGrid, 2, 1>(p, st);
  if (T == 2) return launch_fused<kGrid, 2, 2>(p, st);
  return launch_fused<kGrid, 2, 4>(p, st);
}

int pave_deform_attn_pose_fused_f32(const float* value, const int64_t* spatial_shapes,
                                    const int64_t* level_start, const float* proj,
                                    const float* ref, float* out, float* stat_max,
                                    float* stat_sum, int n_clips, int Q, int T, int S, int L,
                                    int K, int proj_stride, const int32_t* frame_table, int n_slabs,
                                    int ref_levels, void* stream) {
  if (!value || !spatial_shapes || !level_start || !proj || !ref || !out)
    return fail(PAVE_E_ARG, "deform_attn_pose_fused: null pointer");
  if (ref_levels != L && ref_levels != 1)
    return fail(PAVE_E_ARG, "deform_attn_pose_fused: ref_levels = L (a row per level) or 1 (one row for all levels)");
  if (frame_table ? n_slabs <= 0 : (n_slabs != 0 && n_slabs != n_clips * T))
    return fail(PAVE_E_ARG, "deform_attn_pose_fused: n_slabs = the frame slabs behind value (> 0 with a frame table; "
                            "0 or n_clips * T without)");
  if (n_clips <= 0 || Q <= 0 || T <= 0 || S <= 0 || K <= 0)
    return fail(PAVE_E_ARG, "deform_attn_pose_fused: sizes must be positive");
  if (L < 1 || L > 4) return fail(PAVE_E_ARG, "deform_attn_pose_fused: 1 <= L <= 4 levels");
  if (K > 24) return fail(PAVE_E_ARG, "deform_attn_pose_fused: at most 24 keypoints");
  if ((stat_max == nullptr) != (stat_sum == nullptr))
    return fail(PAVE_E_ARG, "deform_attn_pose_fused: stat_max / stat_sum must both be given");
  if (proj_stride < T * kHeads * L * K * 3)
    return fail(PAVE_E_ARG, "deform_attn_pose_fused: proj_stride too small");
  if ((long long)S * kRowFloats * 4 >= (1ll << 31))
    return fail(PAVE_E_ARG, "deform_attn_pose_fused: one value slab must be < 2 GiB");
  FusedParams p{};
  p.value = value;
  p.shapes = spatial_shapes;
  p.lsi = level_start;
  p.proj = proj;
  p.ref = ref;
  p.unit_clip = nullptr;
  p.order = nullptr;
  p.frame_table = frame_table;
  p.n_slabs = frame_table ? n_slabs : n_clips * T;
  p.out = out;
  p.stat_max = stat_max;
  p.stat_sum = stat_sum;
  p.n_units = n_clips * Q;
  p.units_per_clip = Q;
  p.T = T;
  p.S = S;
  p.L = L;
  p.P = K;
  p.ref_L = ref_levels;
  p.proj_stride = proj_stride;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (K <= 16) return launch_fused<kPose, 2, 4>(p, st);
  return launch_fused<kPose, 3, 4>(p, st);
}

int pave_oks_nms_f32(const float* kpts, const float* scores, const double* sigmas, double thresh,
                     int32_t* keep, int32_t* order, int n_clips, int N, int K, void* stream) {
  if (!kpts || !scores || !sigmas || !keep || !order)
    return fail(PAVE_E_ARG, "oks_nms: null pointer");
  if (n_clips <= 0 || N <= 0 || K <= 0) return fail(PAVE_E_ARG, "oks_nms: sizes must be positive");
  if (N > 4096) return fail(PAVE_E_ARG, "oks_nms: at most 4096 poses per clip");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  size_t shmem = (size_t)N * (2 * sizeof(int) + sizeof(float));
  const size_t stage = (size_t)N * K * 2 * sizeof(float) + (size_t)K * sizeof(double);
  const int staged = shmem + stage <= 48 * 1024;   // (100 poses x 15 key points: 12 KB)
  if (staged) shmem += stage;
  hipLaunchKernelGGL(oks_nms_kernel, dim3(n_clips), dim3(256), shmem, st, kpts, scores, sigmas,
                     thresh, keep, order, N, K, staged);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_fill_rows_f32(float* x, long long ld, long long total_rows, const int* rows, long long n_rows,
                       const float* values, int C, void* stream) {
  if (!x || (!rows && n_rows > 0)) return fail(PAVE_E_ARG, "fill_rows: null pointer");
  if (n_rows < 0 || total_rows <= 0 || C <= 0 || (C & 3) || ld < C || (ld & 3))
    return fail(PAVE_E_ARG, "fill_rows: C and ld must be positive multiples of 4, ld >= C");
  if (n_rows == 0) return PAVE_OK;
  const long long n4 = n_rows * (C >> 2);
  long long nb = (n4 + 255) / 256;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, ld, rows, n4, C >> 2, total_rows, values);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_repitch_rows_f32(const float* src, float* dst, long long rows, int W, int pitch, void* stream) {
  if (!src || !dst) return fail(PAVE_E_ARG, "repitch_rows: null pointer");
  if (rows <= 0 || W <= 0 || pitch < W || (pitch & 3) || (reinterpret_cast<uintptr_t>(dst) & 15))
    return fail(PAVE_E_ARG, "repitch_rows: pitch >= W, pitch % 4 == 0, dst 16-byte aligned");
  const long long n4 = rows * (pitch >> 2);
  long long nb = (n4 + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(repitch_rows_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     src, dst, rows, W, pitch >> 2);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_bias_act_rows_f32(const float* x, const float* bias, const float* res, float* y,
                           long long rows, int C, int relu, void* stream) {
  if (!x || !y) return fail(PAVE_E_ARG, "bias_act_rows: null pointer");
  if (rows <= 0 || C <= 0 || (C & 3)) return fail(PAVE_E_ARG, "bias_act_rows: C must be a positive multiple of 4");
  const long long n4 = rows * (C >> 2);
  long long nb = (n4 + 255) / 256;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(bias_act_rows_kernel, dim3((unsigned)nb), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, bias, res, y, n4, C >> 2, relu);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_fuse_sum_nhwc_f32(const float* s0, int sh0, const float* s1, int sh1, const float* s2, int sh2,
                           const float* s3, int sh3, float* y, int N, int H, int W, int C, int relu,
                           void* stream) {
  if (!y || !s0) return fail(PAVE_E_ARG, "fuse_sum: null pointer (the first source is mandatory)");
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3))
    return fail(PAVE_E_ARG, "fuse_sum: bad sizes (C must be a positive multiple of 4)");
  const float* ps[4] = {s0, s1, s2, s3};
  const int shs[4] = {sh0, sh1, sh2, sh3};
  FuseSrc src;
  for (int k = 0; k < 4; ++k) {
    src.p[k] = ps[k];
    src.sh[k] = ps[k] ? shs[k] : 0;
    if (ps[k] && (shs[k] < 0 || shs[k] > 8 || (H & ((1 << shs[k]) - 1)) || (W & ((1 << shs[k]) - 1))))
      return fail(PAVE_E_ARG, "fuse_sum: a source's up-sampling factor 2^s must divide H and W (0 <= s <= 8)");
  }
  const long long n4 = (long long)N * H * W * (C >> 2);
  long long nb = (n4 + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(fuse_sum_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     src, y, n4, H, W, C >> 2, relu);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_bias_add_layernorm_f32(const float* x, const float* bias, const float* res,
                                const float* gamma, const float* beta, float* y, long long rows,
                                int C, float eps, void* stream) {
  return pave_bias_add_layernorm_pos_f32(x, bias, res, gamma, beta, y, nullptr, 0, nullptr, rows, C,
                                         eps, stream);
}

int pave_bias_add_layernorm_pos_f32(const float* x, const float* bias, const float* res,
                                    const float* gamma, const float* beta, float* y,
                                    const float* pos, long long pos_rows, float* y_plus,
                                    long long rows, int C, float eps, void* stream) {
  if (!x || !y || !gamma || !beta) return fail(PAVE_E_ARG, "bias_add_layernorm: null pointer");
  if (rows >= (1ll << 31)) return fail(PAVE_E_ARG, "bias_add_layernorm: rows must be < 2^31");
  if ((y_plus != nullptr) != (pos != nullptr) || (pos && pos_rows <= 0))
    return fail(PAVE_E_ARG, "bias_add_layernorm: pos, pos_rows > 0 and y_plus go together");
  if (rows <= 0 || C <= 0 || (C & 3) || C > 3072)
    return fail(PAVE_E_ARG, "bias_add_layernorm: C must be a multiple of 4, <= 3072");
  long long nb = (rows + 3) / 4;
  if (nb > 256 * 16) nb = 256 * 16;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int vpl = ((C >> 2) + 63) / 64;
#define PAVE_LN(V)                                                                              \
  hipLaunchKernelGGL((bias_add_layernorm_kernel<V>), dim3((unsigned)nb), dim3(256), 0, st, x,   \
                     bias, res, gamma, beta, y, rows, C, eps, pos, pos_rows, y_plus)
  switch (vpl) {
    case 1: PAVE_LN(1); break;
    case 2: PAVE_LN(2); break;
    case 3: PAVE_LN(3); break;
    case 4: PAVE_LN(4); break;
    case 5: case 6: PAVE_LN(6); break;      // (Swin-L: 1536-wide rows; its patch-merging norm: 3072)
    default: PAVE_LN(12); break;
  }
#undef PAVE_LN
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_ms_deform_attn_backward_f32(const float* value, const int64_t* spatial_shapes,
                                     const int64_t* level_start, const float* sampling_loc,
                                     const float* attn_weight, const float* grad_output,
                                     float* grad_value, float* grad_sampling_loc,
                                     float* grad_attn_weight, int bs, int S, int M, int D, int L,
                                     int Lq, int P, int im2col_step, void* stream) {
  return msda_backward_impl<float>(value, spatial_shapes, level_start, sampling_loc, attn_weight,
                                   grad_output, grad_value, grad_sampling_loc, grad_attn_weight,
                                   bs, S, M, D, L, Lq, P, im2col_step, stream);
}

int pave_ms_deform_attn_backward_f64(const double* value, const int64_t* spatial_shapes,
                                     const int64_t* level_start, const double* sampling_loc,
                                     const double* attn_weight, const double* grad_output,
                                     double* grad_value, double* grad_sampling_loc,
                                     double* grad_attn_weight, int bs, int S, int M, int D, int L,
                                     int Lq, int P, int im2col_step, void* stream) {
  return msda_backward_impl<double>(value, spatial_shapes, level_start, sampling_loc, attn_weight,
                                    grad_output, grad_value, grad_sampling_loc, grad_attn_weight,
                                    bs, S, M, D, L, Lq, P, im2col_step, stream);
}

int pave_preprocess_frames(const void* src, int src_is_u8, float* dst, int T, int H0, int W0,
                           int Hn, int Wn, int Hp, int Wp, const float* mean, const float* std,
                           int to_rgb, void* stream) {
  if (!src || !dst || !mean || !std) return fail(PAVE_E_ARG, "preprocess_frames: null pointer");
  if (T <= 0 || H0 <= 0 || W0 <= 0 || Hn <= 0 || Wn <= 0 || Hp < Hn || Wp < Wn)
    return fail(PAVE_E_ARG, "preprocess_frames: bad sizes");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long n = (long long)T * Hp * Wp;
  long long nb = (n + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  const float s0 = (float)(1.0 / (double)std[0]), s1 = (float)(1.0 / (double)std[1]),
              s2 = (float)(1.0 / (double)std[2]);
  if (src_is_u8)
    hipLaunchKernelGGL((preprocess_frames_kernel<unsigned char>), dim3((unsigned)nb), dim3(256), 0,
                       st, static_cast<const unsigned char*>(src), dst, T, H0, W0, Hn, Wn, Hp, Wp,
                       mean[0], mean[1], mean[2], s0, s1, s2, to_rgb);
  else
    hipLaunchKernelGGL((preprocess_frames_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st,
                       static_cast<const float*>(src), dst, T, H0, W0, Hn, Wn, Hp, Wp, mean[0],
                       mean[1], mean[2], s0, s1, s2, to_rgb);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_conv3x3_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int N,
                          int H, int W, int Cin, int Cout, int stride, int relu, void* stream) {
  if (!x || !w || !y) return fail(PAVE_E_ARG, "conv3x3_nhwc: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2))
    return fail(PAVE_E_ARG, "conv3x3_nhwc: bad sizes (stride 1 or 2)");
  if (Cin % 32 != 0 || Cout % 64 != 0)
    return fail(PAVE_E_ARG, "conv3x3_nhwc: Cin %% 32 == 0 and Cout %% 64 == 0 required");
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const long long M = (long long)N * Ho * Wo;
  const int bn = Cout % 128 == 0 ? 128 : 64;
  const long long gx = ((M + 127) / 128) * (Cout / bn);
  if (gx >= (1ll << 31)) return fail(PAVE_E_ARG, "conv3x3_nhwc: grid too large");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bn == 128)
    hipLaunchKernelGGL((conv_nhwc_kernel<128, 3>), dim3((unsigned)gx), dim3(256), 0, st, x, w, bias,
                       (const float*)nullptr, y, N, H, W, Cin, Cout, Ho, Wo, stride, relu,
                       (const float*)nullptr, (const float*)nullptr, 0);
  else
    hipLaunchKernelGGL((conv_nhwc_kernel<64, 3>), dim3((unsigned)gx), dim3(256), 0, st, x, w, bias,
                       (const float*)nullptr, y, N, H, W, Cin, Cout, Ho, Wo, stride, relu,
                       (const float*)nullptr, (const float*)nullptr, 0);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_rows_gemm_bias_res_act_f32(const float* a, const float* a_bias, const float* a2,
                                    const float* w, const float* bias, const float* residual,
                                    float* out, long long M, int K, int K2, int Nc, int relu,
                                    void* stream) {
  if (!a || !w || !out) return fail(PAVE_E_ARG, "rows_gemm: null pointer");
  if (M <= 0 || K <= 0 || K2 < 0 || Nc <= 0 || M >= (1ll << 31))
    return fail(PAVE_E_ARG, "rows_gemm: bad sizes (0 < M < 2^31)");
  if ((K2 > 0) != (a2 != nullptr)) return fail(PAVE_E_ARG, "rows_gemm: a2 and K2 go together");
  if (K % 32 != 0 || K2 % 32 != 0 || Nc % 64 != 0)
    return fail(PAVE_E_ARG, "rows_gemm: K, K2 %% 32 == 0 and N %% 64 == 0 required");
  const int bn = Nc % 128 == 0 ? 128 : 64;
  const long long gx = ((M + 127) / 128) * (Nc / bn);
  if (gx >= (1ll << 31)) return fail(PAVE_E_ARG, "rows_gemm: grid too large");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // a 1x1 convolution over a 1 x M "image"
  if (bn == 128)
    hipLaunchKernelGGL((conv_nhwc_kernel<128, 1>), dim3((unsigned)gx), dim3(256), 0, st, a, w, bias,
                       residual, out, 1, 1, (int)M, K, Nc, 1, (int)M, 1, relu, a_bias, a2, K2);
  else
    hipLaunchKernelGGL((conv_nhwc_kernel<64, 1>), dim3((unsigned)gx), dim3(256), 0, st, a, w, bias,
                       residual, out, 1, 1, (int)M, K, Nc, 1, (int)M, 1, relu, a_bias, a2, K2);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_bias_relu_maxpool_nhwc_f32(const float* x, const float* bias, float* y, int N, int H,
                                    int W, int C, void* stream) {
  if (!x || !bias || !y) return fail(PAVE_E_ARG, "bias_relu_maxpool: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0)
    return fail(PAVE_E_ARG, "bias_relu_maxpool: bad sizes (C %% 4 == 0)");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * Ho * Wo * (C / 4);
  const long long nb = std::min<long long>((total + 255) / 256, 256 * 64);
  hipLaunchKernelGGL(bias_relu_maxpool_kernel, dim3((unsigned)nb), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, bias, y, N, H, W, C, Ho, Wo);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_groupnorm_nhwc_f32(const float* x, const float* gamma, const float* beta, float* y,
                            long long y_batch_stride, int N, int HW, int C, int G, float eps,
                            double* partial, int nchunks, float* ab, void* stream) {
  if (!x || !gamma || !beta || !y || !partial || !ab)
    return fail(PAVE_E_ARG, "groupnorm_nhwc: null pointer");
  if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || nchunks <= 0 || nchunks > 65535 || N > 65535)
    return fail(PAVE_E_ARG, "groupnorm_nhwc: sizes must be positive (N, nchunks <= 65535)");
  if (C % 4 != 0 || C > 1024 || C % G != 0 || (C / G) % 4 != 0 || G > 256 || 256 % (C / 4) != 0)
    return fail(PAVE_E_UNSUPPORTED,
                "groupnorm_nhwc: C %% 4 == 0, (C / G) %% 4 == 0, C / 4 divides 256, G <= 256");
  if (y_batch_stride < (long long)HW * C)
    return fail(PAVE_E_ARG, "groupnorm_nhwc: y_batch_stride smaller than one image");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(groupnorm_stats_kernel, dim3((unsigned)nchunks, (unsigned)N), dim3(256), 0, st,
                     x, partial, HW, C, G, nchunks);
  hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3((unsigned)N), dim3(256), 0, st, partial, gamma,
                     beta, ab, HW, C, G, nchunks, eps);
  const long long total = (long long)N * HW * (C / 4);
  const long long nb = std::min<long long>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(groupnorm_apply_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, ab, y,
                     y_batch_stride, N, HW, C);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_groupnorm_levels_nhwc_f32(const pave_gn_level* levels, int nlev, int N, int C, int G, double* partial,
                                   float* ab, void* stream) {
  if (!levels || !partial || !ab) return fail(PAVE_E_ARG, "groupnorm_levels: null pointer");
  if (nlev <= 0 || nlev > 4) return fail(PAVE_E_ARG, "groupnorm_levels: 1 .. 4 levels per call");
  if (N <= 0 || N > 65535 || C <= 0 || G <= 0)
    return fail(PAVE_E_ARG, "groupnorm_levels: sizes must be positive (N <= 65535)");
  if (C % 4 != 0 || C > 1024 || C % G != 0 || (C / G) % 4 != 0 || G > 256 || 256 % (C / 4) != 0)
    return fail(PAVE_E_UNSUPPORTED,
                "groupnorm_levels: C %% 4 == 0, (C / G) %% 4 == 0, C / 4 divides 256, G <= 256");
  GnLevels p{};
  p.nlev = nlev;
  long long off = 0, chunks = 0, blocks = 0;
  for (int l = 0; l < nlev; ++l) {
    const pave_gn_level& v = levels[l];
    if (!v.x || !v.gamma || !v.beta || !v.y) return fail(PAVE_E_ARG, "groupnorm_levels: null pointer in a level");
    if (v.HW <= 0 || v.nchunks <= 0 || v.nchunks > 65535)
      return fail(PAVE_E_ARG, "groupnorm_levels: HW and nchunks must be positive (nchunks <= 65535)");
    if (v.y_batch_stride < (long long)v.HW * C)
      return fail(PAVE_E_ARG, "groupnorm_levels: y_batch_stride smaller than one image");
    p.x[l] = v.x, p.gamma[l] = v.gamma, p.beta[l] = v.beta, p.y[l] = v.y;
    p.y_batch_stride[l] = v.y_batch_stride;
    p.HW[l] = v.HW, p.nchunks[l] = v.nchunks, p.eps[l] = v.eps;
    p.partial_off[l] = off;
    off += (long long)N * v.nchunks * G * 2;
    chunks += v.nchunks;
    p.chunk_end[l] = (int)chunks;
    const long long total = (long long)N * v.HW * (C / 4);
    blocks += std::min<long long>((total + 255) / 256, 256 * 32);   // (per level what the one-level entry launches)
    p.apply_end[l] = (int)blocks;
  }
  if (chunks > 65535 * 4ll || blocks >= (1ll << 31)) return fail(PAVE_E_ARG, "groupnorm_levels: grid too large");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(groupnorm_levels_stats_kernel, dim3((unsigned)chunks, (unsigned)N), dim3(256), 0, st, p, partial,
                     C, G);
  hipLaunchKernelGGL(groupnorm_levels_finalize_kernel, dim3((unsigned)N, (unsigned)nlev), dim3(256), 0, st, p, partial,
                     ab, N, C, G);
  hipLaunchKernelGGL(groupnorm_levels_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, ab, N, C);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_conv3x3s2_c3_nchw_f32(const float* x, const float* w_taps, const float* bias, float* y, int N,
                               int H, int W, int relu, void* stream) {
  if (!x || !w_taps || !y) return fail(PAVE_E_ARG, "conv3x3s2_c3: null pointer");
  if (N <= 0 || H <= 0 || W <= 0) return fail(PAVE_E_ARG, "conv3x3s2_c3: sizes must be positive");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)N * Ho * Wo;
  if ((total + 255) / 256 >= (1ll << 31)) return fail(PAVE_E_ARG, "conv3x3s2_c3: grid too large");
  hipLaunchKernelGGL(conv3x3s2_c3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, w_taps, bias, y, N, H, W, Ho, Wo, relu);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

int pave_ref_update_f32(const float* tmp, const float* ref, float* out, long long n, float eps,
                        void* stream) {
  if (!tmp || !ref || !out) return fail(PAVE_E_ARG, "ref_update: null pointer");
  if (n <= 0) return fail(PAVE_E_ARG, "ref_update: n must be positive");
  const long long nb = std::min<long long>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(ref_update_kernel, dim3((unsigned)nb), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), tmp, ref, out, n, eps);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

}  // extern "C"
