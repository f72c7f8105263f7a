// pave_decoder.hip -- the small kernels of the decoders and the head (gfx950, wave64): what sits between
// the GEMM launches after the encoder, each replacing a run of library / elementwise launches:
//   pave_mha_core_f32            scaled-dot-product core of the decoders' self-attention
//   pave_topk_rows_f32           proposal / score top-k (one launch; torch.topk: up to 22)
//   pave_gather_frame_poses_f32  the selected queries' poses of all T frames, frame-major
//   pave_ref_update_frames_f32   reference-point update read from the grouped per-frame MLP output
//   pave_pose_finalize_f32       post-processing of the refined poses (pixels, box, RLE confidence)
//
// pave_mha_core_f32
// replaces the scaled-dot-product core of nn.MultiheadAttention as the reference uses it in both
// decoders (third_party/mmcv/mmcv/cnn/bricks/transformer.py:406-551: q = k = x + pos, v = x,
// 8 heads of 32 channels, no masks, no dropout at inference): 300 pose queries per clip in the
// pose decoder, 15 joint queries per pose in the joint decoder.  The q | k | v projection and
// out_proj + identity + LayerNorm are launches of the split GEMM (pave_gemm_dma.hip).
//
// Work layout: one workgroup per (query chunk, head, sequence).  The head's K and V rows
// ([L, 32] fp32 each) are copied once into LDS with rows padded to 36 dwords, so that the four
// lanes of a quad -- which walk four DIFFERENT keys j, j + 1, j + 2, j + 3 at the same channel
// chunk -- read four different bank groups (36 j mod 64 = 0, 36, 8, 44), and the 16 quads of a
// wave, which read the SAME key, broadcast.  A query is owned by LQ = 4 or 16 consecutive lanes:
// lane r takes the keys j = r (mod LQ) with an online softmax over groups of four of its keys (one
// rescale of the 32 accumulators per group), and the LQ partial (max, sum, row) states are
// merged with a shuffle butterfly at the end (16 keys at pitch 36: 16 different bank groups too).  Everything is fp32 FMA: the whole stage is
// ~25 MFLOP per (clip, head) and latency-bound, the matrix cores have nothing to add here.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pave_hip.h"
#include "pave_internal.h"

namespace {

constexpr int kD = 32;        // channels per head
constexpr int kPad = 36;      // LDS row pitch in dwords (see header)
constexpr int kGroup = 4;     // keys per online-softmax group and lane

// one butterfly step of the partial-row merge: lanes r and r ^ ST exchange halves of the N * 2 live
// channels (the lane with bit ST set keeps the upper half); register indices are compile-time
template <int ST, int N>
__device__ __forceinline__ void merge_step(float (&o)[kD], const int r) {
  const bool hi = (r & ST) != 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float keep = hi ? o[N + i] : o[i], send = hi ? o[i] : o[N + i];
    o[i] = keep + __shfl_xor(send, ST, 64);
  }
}

struct MhaParams {
  const float* qkv;   // [n_seq * L, ld]: q at column 0, k at column E, v at column 2 E (E = H * 32)
  float* out;         // [n_seq * L, E]
  int L, H, ld;
  float scale;        // 1 / sqrt(32)
};

// NW waves per workgroup; a query is owned by LQ consecutive lanes (64 / LQ queries per wave):
// LQ = 4 for short sequences (the joint decoder's 15 queries), LQ = 16 for the pose decoder's 300
// (19 keys per lane instead of 75: the stage is latency-bound, so the serial chain per lane counts)
template <int NW, int LQ>
__global__ __launch_bounds__(NW * 64) void mha_core_kernel(const MhaParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // K rows, then V rows
  constexpr int QW = 64 / LQ;                                   // queries per wave
  const int L = p.L, E = p.H * kD;
  const int head = blockIdx.y, seq = blockIdx.z;
  float* ks = lds;
  float* vs = lds + (size_t)L * kPad;
  const float* base = p.qkv + (size_t)seq * L * p.ld + head * kD;
  // ---- stage K_h and V_h: 8 lanes x 16 B per row; the loads of a batch are all in flight before
  // the first LDS store (one L2 round trip per batch, not per chunk)
  constexpr int SU = 5;
  for (int i0 = threadIdx.x; i0 < L * 8; i0 += SU * NW * 64) {
    float4 kreg[SU], vreg[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = min(i0 + u * NW * 64, L * 8 - 1);   // (clamped: loaded anyway, stored if in range)
      const float* src = base + (size_t)(i >> 3) * p.ld + (i & 7) * 4;
      kreg[u] = *reinterpret_cast<const float4*>(src + E);
      vreg[u] = *reinterpret_cast<const float4*>(src + 2 * E);
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = i0 + u * NW * 64;
      if (i < L * 8) {
        *reinterpret_cast<float4*>(ks + (i >> 3) * kPad + (i & 7) * 4) = kreg[u];
        *reinterpret_cast<float4*>(vs + (i >> 3) * kPad + (i & 7) * 4) = vreg[u];
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & (LQ - 1);
  const int qi = blockIdx.x * (NW * QW) + wave * QW + lane / LQ;
  const bool valid = qi < L;
  float q[kD];
  {
    const float* qrow = base + (size_t)(valid ? qi : 0) * p.ld;
#pragma unroll
    for (int c = 0; c < kD; c += 4) {
      const float4 t = *reinterpret_cast<const float4*>(qrow + c);
      q[c] = t.x * p.scale, q[c + 1] = t.y * p.scale, q[c + 2] = t.z * p.scale, q[c + 3] = t.w * p.scale;
    }
  }
  __syncthreads();

  float m = -INFINITY, l = 0.f;
  float o[kD];
#pragma unroll
  for (int c = 0; c < kD; ++c) o[c] = 0.f;
  // lane r walks the keys j = r + LQ t; a group = kGroup consecutive t
  for (int j0 = r; j0 < L; j0 += LQ * kGroup) {
    float s[kGroup];
    int jj[kGroup];
#pragma unroll
    for (int g = 0; g < kGroup; ++g) {
      const int j = j0 + LQ * g;
      jj[g] = j < L ? j : L - 1;          // clamped row, score forced to -inf below
      const float* kr = ks + jj[g] * kPad;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < kD; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(kr + c);
        acc = fmaf(q[c], t.x, acc);
        acc = fmaf(q[c + 1], t.y, acc);
        acc = fmaf(q[c + 2], t.z, acc);
        acc = fmaf(q[c + 3], t.w, acc);
      }
      s[g] = j < L ? acc : -INFINITY;
    }
    float mn = m;
#pragma unroll
    for (int g = 0; g < kGroup; ++g) mn = fmaxf(mn, s[g]);
    // (s[0] is always a real key, so mn is finite from the first group on)
    const float alpha = __expf(m - mn);   // exp(-inf) = 0 on the first group
    l *= alpha;
#pragma unroll
    for (int c = 0; c < kD; ++c) o[c] *= alpha;
#pragma unroll
    for (int g = 0; g < kGroup; ++g) {
      const float pg = __expf(s[g] - mn);
      l += pg;
      const float* vr = vs + jj[g] * kPad;
#pragma unroll
      for (int c = 0; c < kD; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(vr + c);
        o[c] = fmaf(pg, t.x, o[c]);
        o[c + 1] = fmaf(pg, t.y, o[c + 1]);
        o[c + 2] = fmaf(pg, t.z, o[c + 2]);
        o[c + 3] = fmaf(pg, t.w, o[c + 3]);
      }
    }
    m = mn;
  }
  // ---- merge the LQ partial states of a query (a lane without keys: m = -inf, l = 0, weight 0):
  // butterfly over the lane bits; the channel range a lane keeps halves at every step, so lane r
  // ends with the 32 / LQ channels from (32 / LQ) r on
  float M = m;
#pragma unroll
  for (int st = LQ / 2; st >= 1; st >>= 1) M = fmaxf(M, __shfl_xor(M, st, 64));
  const float w = (m == -INFINITY) ? 0.f : __expf(m - M);
  float lsum = l * w;
#pragma unroll
  for (int st = LQ / 2; st >= 1; st >>= 1) lsum += __shfl_xor(lsum, st, 64);
  const float inv = 1.f / lsum;
#pragma unroll
  for (int c = 0; c < kD; ++c) o[c] *= w;
  if constexpr (LQ == 16) {
    merge_step<8, 16>(o, r);
    merge_step<4, 8>(o, r);
    merge_step<2, 4>(o, r);
    merge_step<1, 2>(o, r);
  } else {
    static_assert(LQ == 4, "4 or 16 lanes per query");
    merge_step<2, 16>(o, r);
    merge_step<1, 8>(o, r);
  }
  constexpr int CW = kD / LQ;   // channels per lane: 8 (LQ 4) | 2 (LQ 16)
  if (valid) {
    float* orow = p.out + ((size_t)seq * L + qi) * E + head * kD + r * CW;
    if (CW == 8) {
      *reinterpret_cast<float4*>(orow) = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
      *reinterpret_cast<float4*>(orow + 4) = make_float4(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv);
    } else {
      *reinterpret_cast<float2*>(orow) = make_float2(o[0] * inv, o[1] * inv);
    }
  }
}

template <int NW, int LQ>
int launch_mha(const MhaParams& p, int n_seq, hipStream_t st) {
  const size_t smem = (size_t)2 * p.L * kPad * sizeof(float);
  auto kern = mha_core_kernel<NW, LQ>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return pave_internal_fail(PAVE_E_LAUNCH, "mha_core: cannot raise the dynamic LDS limit");
    attr_set = true;
  }
  constexpr int QB = NW * (64 / LQ);   // queries per workgroup
  const dim3 grid((unsigned)((p.L + QB - 1) / QB), (unsigned)p.H, (unsigned)n_seq);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

}  // namespace

extern "C" int pave_mha_core_f32(const float* qkv, float* out, int n_seq, int L, int H, int ld,
                                 void* stream) {
  if (!qkv || !out) return pave_internal_fail(PAVE_E_ARG, "mha_core: null pointer");
  if (n_seq <= 0 || L <= 0 || H <= 0 || n_seq > 65535 || H > 65535)
    return pave_internal_fail(PAVE_E_ARG, "mha_core: sizes must be positive (n_seq, H <= 65535)");
  if (ld < 3 * H * kD || ld % 4 != 0 || (reinterpret_cast<uintptr_t>(qkv) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15))
    return pave_internal_fail(PAVE_E_ARG, "mha_core: ld >= 3 * H * 32, ld %% 4 == 0, 16-byte aligned pointers");
  if ((size_t)2 * L * kPad * sizeof(float) > 160 * 1024)
    return pave_internal_fail(PAVE_E_UNSUPPORTED, "mha_core: K and V of one head must fit in LDS (L <= 568)");
  MhaParams p{qkv, out, L, H, ld, 0.17677669529663687f /* 1 / sqrt(32) */};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (L <= 16) return launch_mha<1, 4>(p, n_seq, st);
  if (L <= 64) return launch_mha<4, 16>(p, n_seq, st);
  return launch_mha<8, 16>(p, n_seq, st);
}

// ---------------------------------------------------------------------------
// Row-wise top-k (k <= 1024 of n <= 32768 per row), one launch: the 300-of-S proposal selection
// (opera/models/utils/transformer.py:21383-21385) and the N-of-300 score selection
// (opera/models/dense_heads/videopose_head_mul_frames.py:1416) -- torch.topk's multi-block radix
// path is ~22 launches for the former.  One 1024-thread workgroup per row: the row is copied to
// LDS once (coalesced), the k-th largest value is found by a 4 x 8-bit radix select over
// order-preserving keys, the k winners are compacted in index order (ties at the k-th value: the
// lowest indices, whatever the launch geometry) and sorted by (value descending, index
// ascending) with a bitonic network in LDS.  NaN ranks above +inf, as in torch.topk.
// ---------------------------------------------------------------------------
namespace {

constexpr int kTopkThreads = 1024;
constexpr int kTopkMaxN = 32768;
constexpr int kTopkMaxK = 1024;

__device__ __forceinline__ unsigned order_key(float x) {
  const unsigned b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

struct TopkParams {
  const float* x;   // element (r, i) at x[r ld + i cs]
  float* values;    // [rows, k] or null
  long long* index; // [rows, k]
  int n, k, ld, cs;
};

__global__ __launch_bounds__(kTopkThreads) void topk_rows_kernel(const TopkParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  unsigned* keys = reinterpret_cast<unsigned*>(tsm);                          // [n]
  unsigned long long* cand = reinterpret_cast<unsigned long long*>(keys + ((p.n + 1) & ~1));  // [sort size]
  __shared__ unsigned hist[256];
  __shared__ unsigned wtot[4];
  __shared__ unsigned sel_prefix, sel_need;
  __shared__ unsigned long long wsum[kTopkThreads / 64];
  const int tid = threadIdx.x, n = p.n, k = p.k;
  const int lane = tid & 63, wave = tid >> 6;
  const float* row = p.x + (size_t)blockIdx.x * p.ld;
  for (int i = tid; i < n; i += kTopkThreads) keys[i] = order_key(row[(size_t)i * p.cs]);
  if (tid == 0) sel_prefix = 0u, sel_need = (unsigned)k;
  __syncthreads();
  // ---- radix select, most significant byte first: after the pass over byte b, `prefix` holds the
  // top (4 - b) bytes of the k-th largest key and `need` how many keys with exactly that prefix
  // are still to be taken
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = sel_prefix, need = sel_need;
    const unsigned hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    // (logits share their sign / exponent byte: the first pass would pile every lane's atomic on two
    // or three bins.  A wave first counts its four most frequent digits with ballots -- one atomic
    // each -- and only the lanes left over add individually)
    for (int ib = wave * 64; ib < n; ib += kTopkThreads) {
      const int i = ib + lane;
      const unsigned kk = i < n ? keys[i] : 0u;
      const bool match = i < n && (kk & hi_mask) == prefix;
      const unsigned bin = (kk >> shift) & 255u;
      unsigned long long rem = __ballot(match);
      for (int it = 0; it < 4 && rem != 0ull; ++it) {
        const int first = __ffsll((long long)rem) - 1;
        const unsigned b = (unsigned)__shfl((int)bin, first, 64);
        const unsigned long long m = __ballot(match && bin == b) & rem;
        if (lane == first) atomicAdd(&hist[b], (unsigned)__popcll(m));
        rem &= ~m;
      }
      if ((rem >> lane) & 1ull) atomicAdd(&hist[bin], 1u);
    }
    __syncthreads();
    // thread t < 256 owns bin 255 - t (descending); inclusive scan of the counts over t: the bin
    // where the running count first reaches `need` is the next byte of the k-th key
    unsigned c = 0, incl = 0;
    if (tid < 256) {
      c = hist[255 - tid];
      incl = c;
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      if (lane == 63) wtot[wave] = incl;
    }
    __syncthreads();
    if (tid < 256) {
      for (int w = 0; w < wave; ++w) incl += wtot[w];
      if (incl >= need && incl - c < need) {     // exactly one thread
        sel_prefix = prefix | ((unsigned)(255 - tid) << shift);
        sel_need = need - (incl - c);
      }
    }
    __syncthreads();
  }
  const unsigned kth = sel_prefix, need_eq = sel_need;   // need_eq >= 1 keys equal to kth are taken
  // ---- ordered compaction: thread t owns the contiguous index range [t c, (t + 1) c)
  const int chunk = (n + kTopkThreads - 1) / kTopkThreads;
  const int i0 = tid * chunk, i1 = min(n, i0 + chunk);
  unsigned n_gt = 0, n_eq = 0;
  for (int i = i0; i < i1; ++i) {
    const unsigned kk = keys[i];
    n_gt += kk > kth;
    n_eq += kk == kth;
  }
  // block-wide exclusive scans of (n_gt, n_eq), packed into one 64-bit sum per thread
  unsigned long long v = ((unsigned long long)n_gt << 32) | n_eq, incl = v;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  int ssz = 32;                                          // sort size: power of two >= k
  while (ssz < k) ssz <<= 1;
  for (int i = tid; i < ssz; i += kTopkThreads) cand[i] = 0ull;   // padding sorts last
  __syncthreads();
  unsigned long long base = 0, tot = 0;
  for (int w = 0; w < kTopkThreads / 64; ++w) {
    const unsigned long long t = wsum[w];
    if (w < wave) base += t;
    tot += t;
  }
  const unsigned long long excl = base + incl - v;
  const unsigned total_gt = (unsigned)(tot >> 32);        // == k - need_eq
  unsigned o_gt = (unsigned)(excl >> 32), o_eq = (unsigned)excl;
  for (int i = i0; i < i1; ++i) {
    const unsigned kk = keys[i];
    const unsigned long long item = ((unsigned long long)kk << 32) | (0xffffffffu - (unsigned)i);
    if (kk > kth) cand[o_gt++] = item;
    else if (kk == kth) {
      if (o_eq < need_eq) cand[total_gt + o_eq] = item;
      ++o_eq;
    }
  }
  __syncthreads();
  // ---- bitonic sort of the ssz slots, descending (composite: key, then lower index first)
  for (int size = 2; size <= ssz; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int j = tid ^ stride;
      if (tid < ssz && j > tid) {
        const unsigned long long a = cand[tid], b = cand[j];
        const bool desc = (tid & size) == 0;
        if (desc ? a < b : a > b) cand[tid] = b, cand[j] = a;
      }
      __syncthreads();
    }
  }
  if (tid < k) {
    const unsigned long long item = cand[tid];
    const unsigned idx = 0xffffffffu - (unsigned)item;
    p.index[(size_t)blockIdx.x * k + tid] = (long long)idx;
    if (p.values) p.values[(size_t)blockIdx.x * k + tid] = row[(size_t)idx * p.cs];
  }
}

}  // namespace

extern "C" int pave_topk_rows_f32(const float* x, float* values, long long* index, int rows, int n,
                                  int ld, int cs, int k, void* stream) {
  if (!x || !index) return pave_internal_fail(PAVE_E_ARG, "topk_rows: null pointer");
  if (rows <= 0 || n <= 0 || k <= 0 || k > n || cs <= 0 || (rows > 1 && (long long)ld < (long long)(n - 1) * cs + 1))
    return pave_internal_fail(PAVE_E_ARG, "topk_rows: rows, n, cs > 0, 0 < k <= n, ld >= (n - 1) cs + 1");
  if (n > kTopkMaxN || k > kTopkMaxK)
    return pave_internal_fail(PAVE_E_UNSUPPORTED, "topk_rows: n <= 32768 and k <= 1024");
  const size_t smem = (size_t)((n + 1) & ~1) * 4 + (size_t)kTopkMaxK * 8;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(topk_rows_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
      return pave_internal_fail(PAVE_E_LAUNCH, "topk_rows: cannot raise the dynamic LDS limit");
    attr_set = true;
  }
  TopkParams p{x, values, index, n, k, ld, cs};
  hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)rows), dim3(kTopkThreads), smem,
                     reinterpret_cast<hipStream_t>(stream), p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

// ---------------------------------------------------------------------------
// Selected poses of all frames in one gather (videopose_head_mul_frames.py:1419-1427, 610: the
// reference gathers the N selected queries from the centre-frame key points and from every
// auxiliary frame's, then concatenates them frame-major):
//   poses [B, T * Q, C] (frame t at rows [t Q, (t + 1) Q)), index [B, N] -> out [T, B * N, C]
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void gather_frame_poses_kernel(const float* __restrict__ poses,
                                                                 const long long* __restrict__ index,
                                                                 float* __restrict__ out, int B, int T,
                                                                 int Q, int N, int C) {
  const long long total = (long long)T * B * N * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long r = i / C;
    const int bn = (int)(r % ((long long)B * N)), t = (int)(r / ((long long)B * N));
    const int b = bn / N;
    long long q = index[bn];
    q = q < 0 ? 0 : (q >= Q ? Q - 1 : q);   // (a bad index never becomes an out-of-range read)
    out[i] = poses[((long long)b * T * Q + (long long)t * Q + q) * C + c];
  }
}

// ---------------------------------------------------------------------------
// Post-processing of the refined poses, one launch (videopose_head_mul_frames.py:1440-1490 +
// get_p, :1531-1535): key points to pixels (x img_w, y img_h), clamp to the image, optional
// division by the scale factor, bounding box = min / max over the key points, RLE confidence
// p = 0.7 (1 - exp(-0.2 / sigma_x)) (1 - exp(-0.2 / sigma_y)), kpt <- kpt p^5 / (p^5 + 1e-10),
// key-point score = pose score x p.  Same operations in the same order as the tensor
// expressions they replace (expf / powf / IEEE division as ATen's kernels use them).
//   kpts, sigmas [B, N, K, 2]; scores [B, N]; wh, sf [B, 2]
//   -> det_kpts [B, N, K, 3], det_bboxes [B, N, 5]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pose_finalize_kernel(const float* __restrict__ kpts,
                                                           const float* __restrict__ sigmas,
                                                           const float* __restrict__ scores,
                                                           const float* __restrict__ wh,
                                                           const float* __restrict__ sf, float* det_kpts,
                                                           float* det_bboxes, int N, int K, int rescale,
                                                           int sigma_ld) {
  const int pose = blockIdx.x, b = pose / N, j = threadIdx.x;
  const bool act = j < K;
  const float w = wh[b * 2], h = wh[b * 2 + 1];
  float x = 0.f, y = 0.f, sx = 1.f, sy = 1.f;
  if (act) {
    const float2 kp = *reinterpret_cast<const float2*>(kpts + ((size_t)pose * K + j) * 2);
    const float* sp = sigmas + ((size_t)pose * K + j) * sigma_ld;   // (sigma_ld = 2: dense rows)
    const float2 sg = make_float2(sp[0], sp[1]);
    x = fminf(fmaxf(kp.x * w, 0.f), w);
    y = fminf(fmaxf(kp.y * h, 0.f), h);
    if (rescale) x = x / sf[b * 2], y = y / sf[b * 2 + 1];
    sx = sg.x, sy = sg.y;
  }
  float x1 = act ? x : INFINITY, y1 = act ? y : INFINITY, x2 = act ? x : -INFINITY,
        y2 = act ? y : -INFINITY;
  for (int o = 32; o > 0; o >>= 1) {
    x1 = fminf(x1, __shfl_xor(x1, o, 64));
    y1 = fminf(y1, __shfl_xor(y1, o, 64));
    x2 = fmaxf(x2, __shfl_xor(x2, o, 64));
    y2 = fmaxf(y2, __shfl_xor(y2, o, 64));
  }
  const float score = scores[pose];
  if (act) {
    const float px = 1.f - expf(-(0.2f / sx)), py = 1.f - expf(-(0.2f / sy));
    const float p = (px * py) * 0.7f;
    const float p5 = powf(p, 5.f);
    float* o = det_kpts + ((size_t)pose * K + j) * 3;
    o[0] = (x * p5) / (p5 + 1e-10f);
    o[1] = (y * p5) / (p5 + 1e-10f);
    o[2] = score * p;
  }
  if (j == 0) {
    float* bb = det_bboxes + (size_t)pose * 5;
    bb[0] = x1, bb[1] = y1, bb[2] = x2, bb[3] = y2, bb[4] = score;
  }
}
}  // namespace

extern "C" int pave_gather_frame_poses_f32(const float* poses, const long long* index, float* out, int B,
                                           int T, int Q, int N, int C, void* stream) {
  if (!poses || !index || !out) return pave_internal_fail(PAVE_E_ARG, "gather_frame_poses: null pointer");
  if (B <= 0 || T <= 0 || Q <= 0 || N <= 0 || C <= 0)
    return pave_internal_fail(PAVE_E_ARG, "gather_frame_poses: sizes must be positive");
  const long long total = (long long)T * B * N * C;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(gather_frame_poses_kernel, dim3(blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), poses, index, out, B, T, Q, N, C);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

extern "C" int pave_pose_finalize_f32(const float* kpts, const float* sigmas, const float* scores,
                                      const float* wh, const float* sf, float* det_kpts,
                                      float* det_bboxes, int B, int N, int K, int rescale, int sigma_ld,
                                      void* stream) {
  if (!kpts || !sigmas || !scores || !wh || !det_kpts || !det_bboxes || (rescale && !sf))
    return pave_internal_fail(PAVE_E_ARG, "pose_finalize: null pointer");
  if (B <= 0 || N <= 0 || K <= 0 || K > 64 || sigma_ld < 2)
    return pave_internal_fail(PAVE_E_ARG, "pose_finalize: B, N > 0, 0 < K <= 64, sigma_ld >= 2");
  hipLaunchKernelGGL(pose_finalize_kernel, dim3((unsigned)(B * N)), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), kpts, sigmas, scores, wh, sf, det_kpts,
                     det_bboxes, N, K, rescale, sigma_ld);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

// ---------------------------------------------------------------------------
// Reference-point update straight from the grouped per-frame MLP output (one launch instead of a
// layout copy + the update):  y [R, T * op] (row r, frame t's `o` outputs at columns [t op, t op + o))
//   out[(r / G) T G + t G + r % G][c] = sigmoid(y[r][t op + c] + inverse_sigmoid(ref[same]))
// G = queries per clip (pose decoder: rows frame-major inside a clip, OT:6728-6735) or G = R (joint
// decoder: frame-major over all rows, MT:861-866).  Same fp32 operation order as pave_ref_update_f32.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void ref_update_frames_kernel(const float* __restrict__ y,
                                                                const float* __restrict__ ref,
                                                                float* __restrict__ out, int R, int T,
                                                                int op, int o, int G, float eps) {
  const long long n = (long long)R * T * o;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (long long)gridDim.x * 256) {
    const int c = (int)(i % o);
    const long long rt = i / o;
    const int t = (int)(rt % T), r = (int)(rt / T);
    const long long orow = (long long)(r / G) * T * G + (long long)t * G + r % G;
    const float x = fminf(fmaxf(ref[orow * o + c], 0.f), 1.f);
    const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
    const float v = y[(long long)r * T * op + (long long)t * op + c] + logf(x1 / x2);
    out[orow * o + c] = 1.f / (1.f + expf(-v));
  }
}
}  // namespace

extern "C" int pave_ref_update_frames_f32(const float* y, const float* ref, float* out, int R, int T,
                                          int op, int o, int G, float eps, void* stream) {
  if (!y || !ref || !out) return pave_internal_fail(PAVE_E_ARG, "ref_update_frames: null pointer");
  if (R <= 0 || T <= 0 || o <= 0 || op < o || G <= 0 || R % G != 0)
    return pave_internal_fail(PAVE_E_ARG, "ref_update_frames: R, T, o > 0, op >= o, R %% G == 0");
  const long long n = (long long)R * T * o;
  const unsigned nb = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(ref_update_frames_kernel, dim3(nb), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), y, ref, out, R, T, op, o, G, eps);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

// ---------------------------------------------------------------------------
// Two-stage query initialisation behind the proposal top-k (opera/models/utils/transformer.py:21386-21418),
// two launches instead of thirteen elementwise / gather launches:
//
// pave_gather_rows_add_f32:  rows[n, q, :] = src[n, index[n, q], :]  (the top-k rows of output_memory: `tgt`)
//                            sum[n, q, :]  = rows[n, q, :] + add[q, :]  (`query = tgt + query`, optional)
// pave_proposal_refs_f32:    kpt[n q, c] += props[n, index[n, q], c & 1]  (in place: the un-activated key points of
//                            the selected tokens, x at even and y at odd columns, + their proposal's logit position)
//                            refs[n, t Q + q, c] = sigmoid(kpt[n q, c])  for t = 0 .. T - 1  (`.sigmoid().repeat(1, T, 1)`)
// An index outside [0, S) is never dereferenced (the row is zero-filled / the proposal term is dropped).
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void gather_rows_add_kernel(const float* __restrict__ src,
                                                              const long long src_batch,
                                                              const long long* __restrict__ index,
                                                              const float* __restrict__ add,
                                                              float* __restrict__ rows, float* __restrict__ sum,
                                                              const int n, const int Q, const int S, const int C4) {
  const long long total = (long long)n * Q * C4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    const long long r = i / C4;
    const int q = (int)(r % Q), b = (int)(r / Q);
    const long long s = index[r];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= 0 && s < S) v = reinterpret_cast<const float4*>(src + (long long)b * src_batch + s * (C4 * 4ll))[c];
    reinterpret_cast<float4*>(rows)[i] = v;
    if (sum) {
      const float4 a = reinterpret_cast<const float4*>(add)[(long long)q * C4 + c];
      reinterpret_cast<float4*>(sum)[i] = make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
    }
  }
}

__global__ __launch_bounds__(256) void proposal_refs_kernel(float* __restrict__ kpt, const int ld,
                                                            const float* __restrict__ props,
                                                            const long long props_batch,
                                                            const long long* __restrict__ index,
                                                            float* __restrict__ refs, const int n, const int Q,
                                                            const int S, const int K2, const int T) {
  const long long total = (long long)n * Q * K2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % K2);
    const long long r = i / K2;
    const int q = (int)(r % Q), b = (int)(r / Q);
    const long long s = index[r];
    float v = kpt[r * ld + c];
    if (s >= 0 && s < S) v += props[(long long)b * props_batch + s * 2 + (c & 1)];
    kpt[r * ld + c] = v;
    const float sg = 1.f / (1.f + expf(-v));
    for (int t = 0; t < T; ++t) refs[(((long long)b * T + t) * Q + q) * K2 + c] = sg;
  }
}
}  // namespace

extern "C" int pave_gather_rows_add_f32(const float* src, long long src_batch_stride, const long long* index,
                                        const float* add, float* rows, float* sum, int n, int Q, int S, int C,
                                        void* stream) {
  if (!src || !index || !rows || (sum && !add))
    return pave_internal_fail(PAVE_E_ARG, "gather_rows_add: null pointer (sum needs add)");
  if (n <= 0 || Q <= 0 || S <= 0 || C <= 0 || C % 4 != 0 || src_batch_stride < 0 || src_batch_stride % 4 != 0)
    return pave_internal_fail(PAVE_E_ARG, "gather_rows_add: n, Q, S > 0, C %% 4 == 0, batch stride %% 4 == 0");
  const long long total = (long long)n * Q * (C / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(gather_rows_add_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src,
                     src_batch_stride, index, add, rows, sum, n, Q, S, C / 4);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

extern "C" int pave_proposal_refs_f32(float* kpt, int ld, const float* props, long long props_batch_stride,
                                      const long long* index, float* refs, int n, int Q, int S, int K2, int T,
                                      void* stream) {
  if (!kpt || !props || !index || !refs) return pave_internal_fail(PAVE_E_ARG, "proposal_refs: null pointer");
  if (n <= 0 || Q <= 0 || S <= 0 || K2 <= 0 || K2 % 2 != 0 || T <= 0 || ld < K2 || props_batch_stride < 0)
    return pave_internal_fail(PAVE_E_ARG, "proposal_refs: n, Q, S, T > 0, K2 even, ld >= K2");
  const long long total = (long long)n * Q * K2;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(proposal_refs_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), kpt, ld,
                     props, props_batch_stride, index, refs, n, Q, S, K2, T);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

// pave_merge_softmax_partials_f32
// The frame-sharded T-frame attentions (BASELINE configs[4]: frame t on rank t % G) leave on every rank the
// all-gathered [G, U, C + 2 H] buffer of per-rank partial rows (normalised by the rank's OWN sum) followed by the
// per-head (max logit, sum exp) of the rank's frames; the exact row of a softmax over all ranks' logits is
//   w_g = ssum_g exp(smax_g - max_g smax_g),   out = sum_g rows_g w_g / sum_g w_g
// (a rank with ssum = 0 -- no frames -- drops out, its row is not read).  One launch instead of the ~10
// elementwise launches of the torch formulation on the latency-bound decoder tail.
namespace {
__global__ __launch_bounds__(256) void merge_softmax_partials_kernel(const float* __restrict__ parts,
                                                                      float* __restrict__ out, const int G,
                                                                      const int U, const int C, const int H) {
  const int c4n = C >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)U * c4n) return;
  const int u = (int)(i / c4n), c = (int)(i - (long long)u * c4n) * 4;
  const int h = c / (C / H);
  const int ld = C + 2 * H;
  float m = -INFINITY;
  for (int g = 0; g < G; ++g) {
    const float* row = parts + ((long long)g * U + u) * ld;
    if (row[C + H + h] > 0.f) m = fmaxf(m, row[C + h]);
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float wsum = 0.f;
  for (int g = 0; g < G; ++g) {
    const float* row = parts + ((long long)g * U + u) * ld;
    const float s = row[C + H + h];
    if (!(s > 0.f)) continue;
    const float w = s * expf(row[C + h] - m);
    if (!(w > 0.f)) continue;     // a rank whose weight underflows contributes nothing -- not 0 x its row (a
                                  // non-finite row there would give NaN: the host formulation masks w == 0 too)
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    acc.x = fmaf(v.x, w, acc.x), acc.y = fmaf(v.y, w, acc.y), acc.z = fmaf(v.z, w, acc.z), acc.w = fmaf(v.w, w, acc.w);
    wsum += w;
  }
  const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
  *reinterpret_cast<float4*>(out + (long long)u * C + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}
}  // namespace

extern "C" int pave_merge_softmax_partials_f32(const float* parts, float* out, int G, int U, int C, int H,
                                               void* stream) {
  if (!parts || !out) return pave_internal_fail(PAVE_E_ARG, "merge_softmax_partials: null pointer");
  if (G <= 0 || U <= 0 || C <= 0 || H <= 0 || C % H != 0 || (C / H) % 4 != 0 || (2 * H) % 4 != 0)
    return pave_internal_fail(PAVE_E_ARG, "merge_softmax_partials: C / H channels per head, a multiple of 4; H even");
  const long long n = (long long)U * (C >> 2);
  hipLaunchKernelGGL(merge_softmax_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), parts, out, G, U, C, H);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

// pave_swin_window_attn_f32
// The (shifted-)window multi-head self-attention core of a Swin block (ShiftWindowMSA.forward + WindowMSA.forward,
// third_party/mmdetection/mmdet/models/backbones/swin.py:22-126, 128-286) on the UN-partitioned token map:
//   qkv [B, H, W, 3 C] = the block's qkv Linear applied per token (a Linear commutes with every permutation of
//   the rows), out [B, H, W, C] = what `window_reverse` + the reverse roll + the crop return, ready for `proj`.
// One wave per (window, head): lane i < ws^2 owns query token i of the window.  The reference pads the map to a
// multiple of the window (zeros BEFORE the qkv Linear: a pad token's q | k | v is the Linear's bias), rolls it by
// -shift, cuts windows, adds the relative-position bias and -- for shifted windows -- the -100 mask between
// tokens of different roll regions, and undoes all of it afterwards; here every lane computes the source pixel of
// its token ((window row * ws + iy + shift) mod H_pad, ...), takes the bias vector for a pad pixel, derives its
// mask region from the shifted coordinates (swin.py:213-231: three bands per axis) and writes its output row
// straight to its source pixel -- no pad / roll / partition / reverse copies, no mask tensor.
// K and V of the head sit in LDS (rows padded to 36 dwords: conflict-free 16-byte stores; the key loop reads one
// row per step, broadcast to the whole wave); two passes over the ws^2 keys (scores kept in registers):
// max, then exp / sum / PV.  head dim 32 (every Swin variant), ws^2 <= 64 (window 7: 49 tokens).
namespace {
template <int WS>
__global__ __launch_bounds__(64) void swin_window_attn_kernel(
    const float* __restrict__ qkv, const float* __restrict__ bias_t, const float* __restrict__ pad_qkv,
    float* __restrict__ out, const int B, const int H, const int W, const int C, const int heads,
    const int shift, const float scale) {
  constexpr int N = WS * WS;
  __shared__ __attribute__((aligned(16))) float ks[N * kPad];
  __shared__ __attribute__((aligned(16))) float vs[N * kPad];
  __shared__ int regs[N];
  const int lane = threadIdx.x;
  const int nwx = (W + WS - 1) / WS, nwy = (H + WS - 1) / WS;
  const int Hp = nwy * WS, Wp = nwx * WS;
  int bid = blockIdx.x;
  const int h = bid % heads;
  bid /= heads;
  const int wx = bid % nwx;
  bid /= nwx;
  const int wy = bid % nwy;
  const int b = bid / nwy;
  const bool act = lane < N;
  const int iy = act ? lane / WS : 0, ix = act ? lane % WS : 0;
  const int ys = wy * WS + iy, xs = wx * WS + ix;          // coordinates in the rolled, padded map
  int y = ys + shift, x = xs + shift;                        // source pixel (roll by -shift)
  if (y >= Hp) y -= Hp;
  if (x >= Wp) x -= Wp;
  const bool real = act && y < H && x < W;
  const float* row = real ? qkv + (((long long)b * H + y) * W + x) * 3 * C : pad_qkv;
  int reg = 0;
  if (shift > 0) {
    const int rh = ys < Hp - WS ? 0 : (ys < Hp - shift ? 1 : 2);
    const int rw = xs < Wp - WS ? 0 : (xs < Wp - shift ? 1 : 2);
    reg = rh * 3 + rw;
  }
  float q[kD];
  if (act) {
#pragma unroll
    for (int c = 0; c < kD; c += 4) {
      const float4 t = *reinterpret_cast<const float4*>(row + h * kD + c);
      q[c] = t.x * scale, q[c + 1] = t.y * scale, q[c + 2] = t.z * scale, q[c + 3] = t.w * scale;
      *reinterpret_cast<float4*>(ks + lane * kPad + c) = *reinterpret_cast<const float4*>(row + C + h * kD + c);
      *reinterpret_cast<float4*>(vs + lane * kPad + c) = *reinterpret_cast<const float4*>(row + 2 * C + h * kD + c);
    }
    regs[lane] = reg;
  }
  __syncthreads();
  if (!act) return;
  // pass 1: scores (bias_t [heads][key j][query i]: consecutive lanes read consecutive addresses)
  const float* bt = bias_t + (long long)h * N * N + lane;
  float s[N];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < kD; c += 4) {
      const float4 k4 = *reinterpret_cast<const float4*>(ks + j * kPad + c);
      d = fmaf(q[c], k4.x, d), d = fmaf(q[c + 1], k4.y, d), d = fmaf(q[c + 2], k4.z, d), d = fmaf(q[c + 3], k4.w, d);
    }
    d += bt[j * N];
    if (shift > 0 && regs[j] != reg) d += -100.f;
    s[j] = d;
    m = fmaxf(m, d);
  }
  // pass 2: softmax weights and the weighted sum of the values
  float acc[kD];
#pragma unroll
  for (int c = 0; c < kD; ++c) acc[c] = 0.f;
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float p = expf(s[j] - m);
    l += p;
#pragma unroll
    for (int c = 0; c < kD; c += 4) {
      const float4 v4 = *reinterpret_cast<const float4*>(vs + j * kPad + c);
      acc[c] = fmaf(p, v4.x, acc[c]), acc[c + 1] = fmaf(p, v4.y, acc[c + 1]);
      acc[c + 2] = fmaf(p, v4.z, acc[c + 2]), acc[c + 3] = fmaf(p, v4.w, acc[c + 3]);
    }
  }
  if (!real) return;                 // a pad token: attended to by its window, never written back (the crop)
  const float inv = 1.f / l;
  float* o = out + (((long long)b * H + y) * W + x) * C + h * kD;
#pragma unroll
  for (int c = 0; c < kD; c += 4)
    *reinterpret_cast<float4*>(o + c) = make_float4(acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv);
}

// The same attention core on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 products, 64 FLOP / clk / SIMD)
// -- the shipped form.  Why: in the per-lane form above every lane walks all 49 keys, i.e. 16 broadcast
// ds_read_b128 per key and wave -- 6.4 k cycles of the CU's ONE LDS pipe per wave, which bounds that form while the
// vector and matrix pipes idle (packed FMAs changed nothing, scalar loads of the wave-uniform K / V rows are one L2
// round trip per key: DESIGN.md section 9).  Here the window's 49 tokens are padded to 64 and the wave computes
//   S^T [key, query] = K Q^T   (2 x 2 tiles of 32 x 32, 16 K-steps: 64 MFMAs.  Step s contracts channel s -- supplied
//                               by the lanes of half 0 -- and channel 16 + s -- half 1 --, so lane (row r, half)
//                               needs 16 CONSECUTIVE channels of its two tokens' q and k: four float4 loads per
//                               row straight into registers, no staging, no LDS read in the loop)
// -- TRANSPOSED, so that a query's scores lie along the accumulator's ROW axis: lane (query, half) holds 32 of its
// 64 keys in registers, the other half-wave the rest, and max / sum are 31 in-lane operations and one cross-half
// exchange.  The probabilities never leave the registers: the accumulator layout of S^T (lane = query column,
// register r of key tile kt = key 32 kt + 8 (r / 4) + 4 half + r % 4) IS a B operand of
//   O^T [channel, query] = V^T P^T  (K axis = keys, taken in exactly that register order: step (kt, r) contracts
//                                    the keys 32 kt + 8 (r / 4) + r % 4 (half 0) and + 4 (half 1); the A operand
//                                    reads V [that key][channel = lane % 32] from LDS, the only staged tensor)
// 64 more MFMAs.  Relative-position bias (all 32 values of a lane requested before the first is used), shift mask
// (regions of the 64 tokens in LDS) and -inf for the 15 pad keys are added to the S^T registers; outputs are four
// float4 stores per lane and query tile.  9.5 KB of LDS, 158 + 64 registers: two waves per SIMD.
// tools/swin_attn_ab.py, 3 frames of 800 x 1344, us per launch (per-lane form -> this): stage 1 267 -> 221,
// stage 2 149 -> 117, stage 3 75.6 -> 60.4, stage 4 46.3 -> 35.5.  (Built and measured on the way: Q / K / V all
// staged in LDS, a lane per token: 253 / 136 / 77 / 43; the same with 8 lanes per 128-byte piece and the output
// transposed through LDS: 317 / 165 / 84 / 47; bounded to three waves per SIMD: 36 bytes of scratch, 244 / 129 /
// 63 / 38.)
template <int WS>
__global__ __launch_bounds__(64) void swin_window_attn_mfma_kernel(
    const float* __restrict__ qkv, const float* __restrict__ bias_t, const float* __restrict__ pad_qkv,
    float* __restrict__ out, const int B, const int H, const int W, const int C, const int heads,
    const int shift, const float scale) {
  constexpr int N = WS * WS;
  static_assert(N <= 64, "one wave per window");
  typedef float f16v __attribute__((ext_vector_type(16)));
  __shared__ __attribute__((aligned(16))) float vs[64 * kPad];
  __shared__ int regs[64];
  const int lane = threadIdx.x;
  const int r32 = lane & 31, hf = lane >> 5;
  const int nwx = (W + WS - 1) / WS, nwy = (H + WS - 1) / WS;
  const int Hp = nwy * WS, Wp = nwx * WS;
  int bid = blockIdx.x;
  const int h = bid % heads;
  bid /= heads;
  const int wx = bid % nwx;
  bid /= nwx;
  const int wy = bid % nwy;
  const int b = bid / nwy;
  // token `t` of the window: source pixel (the roll by -shift), pad flag, roll region
  auto token = [&](const int t, bool& real, int& reg) -> long long {
    const int iy = t / WS, ix = t - iy * WS;
    const int ys = wy * WS + iy, xs = wx * WS + ix;
    int y = ys + shift, x = xs + shift;
    if (y >= Hp) y -= Hp;
    if (x >= Wp) x -= Wp;
    real = t < N && y < H && x < W;
    reg = 0;
    if (shift > 0) {
      const int rh = ys < Hp - WS ? 0 : (ys < Hp - shift ? 1 : 2);
      const int rw = xs < Wp - WS ? 0 : (xs < Wp - shift ? 1 : 2);
      reg = rh * 3 + rw;
    }
    return ((long long)b * H + y) * W + x;
  };
  // V of token `lane` (zeros for the 15 pad slots of the 64-token tile) and its roll region go to LDS; the Q and K
  // operands come straight from the rows: K-step s of S^T contracts channel s (supplied by the lanes of half 0)
  // and channel 16 + s (half 1), so lane (row r, half) needs 16 CONSECUTIVE channels of its two tokens' q and k
  // -- four float4 loads per row, nothing staged, no LDS read in the loop
  float qr[2][16], kr[2][16];
  {
    bool real;
    int reg;
    const long long pix = token(lane, real, reg);
    const float* row = (real ? qkv + pix * 3 * C : pad_qkv) + 2 * C + h * kD;
    const float on = lane < N ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < kD; c += 4) {
      float4 tv = *reinterpret_cast<const float4*>(row + c);
      tv.x *= on, tv.y *= on, tv.z *= on, tv.w *= on;
      *reinterpret_cast<float4*>(vs + lane * kPad + c) = tv;
    }
    regs[lane] = reg;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = 32 * tt + r32;
      const long long px = token(t, real, reg);
      const float* rq = (real ? qkv + px * 3 * C : pad_qkv) + h * kD + 16 * hf;
      const float onq = t < N ? 1.f : 0.f, qscale = onq * scale;
#pragma unroll
      for (int c = 0; c < 16; c += 4) {
        const float4 tq = *reinterpret_cast<const float4*>(rq + c);
        const float4 tk = *reinterpret_cast<const float4*>(rq + C + c);
        qr[tt][c] = tq.x * qscale, qr[tt][c + 1] = tq.y * qscale, qr[tt][c + 2] = tq.z * qscale, qr[tt][c + 3] = tq.w * qscale;
        kr[tt][c] = tk.x * onq, kr[tt][c + 1] = tk.y * onq, kr[tt][c + 2] = tk.z * onq, kr[tt][c + 3] = tk.w * onq;
      }
    }
  }
  __syncthreads();
  // ---- S^T = K Q^T
  f16v st[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[a][q][r] = 0.f;
#pragma unroll
  for (int sstep = 0; sstep < kD / 2; ++sstep) {
    st[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[0][sstep], qr[0][sstep], st[0][0], 0, 0, 0);
    st[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[0][sstep], qr[1][sstep], st[0][1], 0, 0, 0);
    st[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[1][sstep], qr[0][sstep], st[1][0], 0, 0, 0);
    st[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[1][sstep], qr[1][sstep], st[1][1], 0, 0, 0);
  }
  // ---- bias, mask, softmax over the keys of each query column (qt: the lane's query 32 qt + r32)
  float inv[2];
  const float pen = shift > 0 ? -100.f : 0.f;      // (without a shift every token is in region 0)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int query = 32 * qt + r32;
    const bool qok = query < N;
    const int regq = regs[query];
    const float* bt = bias_t + (long long)h * N * N + (qok ? query : 0);
    // (all 32 bias values of the lane requested before the first is used: the loads are unconditional -- a pad
    // key reads the last row and is overwritten with -inf below -- so that they are ONE round trip, not 32)
    float bv[2][16];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kt + 8 * (r >> 2) + 4 * hf + (r & 3);
        bv[kt][r] = bt[(key < N ? key : N - 1) * N];
      }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kt + 8 * (r >> 2) + 4 * hf + (r & 3);
        float d = st[kt][qt][r] + bv[kt][r];
        d += regs[key] != regq ? pen : 0.f;
        d = key < N ? d : -INFINITY;
        st[kt][qt][r] = d;
        m = fmaxf(m, d);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = expf(st[kt][qt][r] - m);
        st[kt][qt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 32, 64);
    inv[qt] = 1.f / l;
  }
  // ---- O^T = V^T P^T
  f16v ot[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[q][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * kt + 8 * (r >> 2) + 4 * hf + (r & 3);
      const float a = vs[key * kPad + r32];
      ot[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, st[kt][0][r], ot[0], 0, 0, 0);
      ot[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, st[kt][1][r], ot[1], 0, 0, 0);
    }
  // ---- the lane's two queries: channels 8 g + 4 half + 0 .. 3 of register group g (through LDS to whole
  // 128-byte pieces measured slower: 253 -> 317 us at stage 1)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    bool real;
    int reg;
    const long long pix = token(32 * qt + r32, real, reg);
    if (!real) continue;           // pad slot or pad token: attended to by its window, never written back
    float* o = out + pix * C + h * kD + 4 * hf;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(o + 8 * g) =
          make_float4(ot[qt][4 * g] * inv[qt], ot[qt][4 * g + 1] * inv[qt], ot[qt][4 * g + 2] * inv[qt],
                      ot[qt][4 * g + 3] * inv[qt]);
  }
}
}  // namespace

extern "C" int pave_swin_window_attn_f32(const float* qkv, const float* bias_t, const float* pad_qkv, float* out,
                                         int B, int H, int W, int C, int heads, int window, int shift,
                                         float scale, void* stream) {
  if (!qkv || !bias_t || !pad_qkv || !out) return pave_internal_fail(PAVE_E_ARG, "swin_window_attn: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || heads <= 0 || C != heads * kD)
    return pave_internal_fail(PAVE_E_ARG, "swin_window_attn: C == heads * 32 (head dim 32) required");
  if (window != 7) return pave_internal_fail(PAVE_E_UNSUPPORTED, "swin_window_attn: window size 7 is built");
  if (shift < 0 || shift >= window) return pave_internal_fail(PAVE_E_ARG, "swin_window_attn: 0 <= shift < window");
  const long long nb = (long long)B * ((H + window - 1) / window) * ((W + window - 1) / window) * heads;
  if (nb >= (1ll << 31)) return pave_internal_fail(PAVE_E_ARG, "swin_window_attn: grid too large");
  if (pave_internal_diag_variant() == 18)   // the per-lane form (tests compare the two)
    hipLaunchKernelGGL((swin_window_attn_kernel<7>), dim3((unsigned)nb), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), qkv, bias_t, pad_qkv, out, B, H, W, C, heads, shift,
                       scale);
  else
    hipLaunchKernelGGL((swin_window_attn_mfma_kernel<7>), dim3((unsigned)nb), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), qkv, bias_t, pad_qkv, out, B, H, W, C, heads, shift,
                       scale);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}
