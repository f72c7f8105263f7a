// libpave_hip.so -- fp32 row GEMM on the bf16 matrix cores by 3-way operand splitting.
//
// gfx950 multiplies fp32 on the MFMA at 157 TFLOP/s (v_mfma_f32_32x32x2_f32) but bf16 at
// 2.5 PFLOP/s (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  An fp32 value splits EXACTLY into
// three bf16 values by truncation (8 + 8 + 8 significand bits):
//     a0 = hi16(a), r1 = a - a0, a1 = hi16(r1), a2 = r1 - a1            a = a0 + a1 + a2
// (the subtractions are exact in fp32), products of bf16 pairs are exact in the MFMA's fp32
// datapath, and
//     a*b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + [a1b2 + a2b1 + a2b2]
// where the bracket is <= 2^-23 |ab| -- below fp32's own rounding of the product sum.  Six bf16
// MFMAs therefore give a GEMM with fp32-level accuracy (tests/test_ops_gpu.py measures it
// against fp64 next to the native fp32 MFMA) at 2.5 PF / 6 = 417 TFLOP/s peak, 2.7x the fp32
// MFMA rate.  Weights are split once on the host side (pave_split_bf16x3_f32); activations are
// split while they are staged into LDS, so HBM traffic is that of an fp32 GEMM.
//
//   out[M, N] = act(A'[M, K] * W[N, K]^T + bias[N] + residual[M, N]),
//   A' = a_bias ? relu(A + a_bias[K]) : A
//
// Block = 256 threads = 2 x 2 waves, block tile (2*TM*32) x (2*TN*32), K slab 16 (one MFMA
// k-step).  LDS holds the three bf16 planes of the A and W slabs, double buffered, rows padded to
// 48 bytes so that the ds_read_b128 of an MFMA operand (lane -> row l&31, k half l>>5) is
// bank-conflict free.  The epilogue goes through LDS so that bias / residual / ReLU / store run
// on float4 with whole row segments per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pave_hip.h"
#include "pave_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned load

constexpr int BK = 16;          // K slab = one MFMA k-step
constexpr int RST = 48;         // LDS row stride in bytes: 32 B of bf16 + 16 B pad (the 16 lanes
                                // of a ds_read_b128 phase then fall on 16 distinct bank quads)

// bijection on the low 3 bits: 0 1 2 3 4 5 6 7 -> 0 2 4 6 1 3 5 7
__device__ __forceinline__ int stage_row_perm(int r) {
  return (r & ~7) | ((r & 3) << 1) | ((r >> 2) & 1);
}

__device__ __forceinline__ unsigned hi16(float x) { return __float_as_uint(x) & 0xffff0000u; }
// pack the bf16 (= high halves) of two fp32 bit patterns: lo -> bits 0..15, hi -> bits 16..31
__device__ __forceinline__ unsigned pack_hi(unsigned lo, unsigned hi) {
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// two fp32 -> two bf16 (or fp16), round to nearest even
template <bool F16 = false>
__device__ __forceinline__ unsigned pack_rne(float lo, float hi) {
  const f32x2 v = {lo, hi};
  if (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// P-term bf16 split of 4 consecutive fp32 -> P x (4 bf16 = 8 bytes).  All terms but the last
// are truncations (exact residuals); the last is rounded to nearest (for P = 3 it is exact
// either way: the residual has at most 8 significant bits).
template <int P, bool F16>
__device__ __forceinline__ void split4(const float4 v, uint2 (&p)[P]) {
  float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int t = 0; t < P - 1; ++t) {
    unsigned h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h[i] = hi16(x[i]);
      x[i] -= __uint_as_float(h[i]);
    }
    p[t] = make_uint2(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]));
  }
  p[P - 1] = make_uint2(pack_rne<F16>(x[0], x[1]), pack_rne<F16>(x[2], x[3]));
}

// 3x3 / pad 1 convolution as an implicit GEMM: row m = output pixel (n, oy, ox), K axis =
// (tap, cin) with a 16-wide slab inside one tap; H == 0 means a plain row GEMM.
struct ConvGeom {
  int H, W, Cin, Ho, Wo, stride;
};

// Epilogue options of the row GEMM: the residual may be a [res_rows, N] table indexed by
// row % res_rows (a per-token term shared by all frames), and the output may be cut at column
// nsplit into two dense matrices: out [M, nsplit] and out2 [M, N - nsplit] (nsplit % BN == 0).
struct OutSplit {
  float* out2;
  int nsplit;
  int res_rows;
};

// LayerNorm over the output row fused into the epilogue (LNORM forms: the block tile spans the
// whole row, N == BN): out = LN(acc + bias + residual) * gamma + beta.
struct LnArgs {
  const float* gamma;
  const float* beta;
  float eps;
};

// Software pipeline (one wave per SIMD, so nothing else hides latency):
//   global loads run two slabs ahead (registers), LDS is double buffered with ONE barrier per
//   slab, and the operand fragments of slab s+1 are read from LDS between the two halves of
//   slab s's MFMAs, into a second fragment register set.
template <int TM, int TN, bool ABIAS, int P, bool F16, int CONV, int WGN, bool LNORM>
__device__ __forceinline__ void gemm_split_body(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const int relu,
    const float* __restrict__ a_bias, const ConvGeom g, const OutSplit os, const LnArgs ln) {
  constexpr int NT = 128 * WGN, NW = 2 * WGN;                  // threads, waves (2 x WGN grid)
  constexpr int BM = 2 * TM * 32, BN = WGN * TN * 32;
  constexpr int A_PLANE = BM * RST, W_PLANE = BN * RST;        // bytes
  constexpr int BUF = P * (A_PLANE + W_PLANE);                 // one LDS buffer
  constexpr int EPI = NW * 32 * (TN * 32 + 4) * 4 + (LNORM ? 2 * BM * WGN * 4 : 0);
  constexpr int SMEM = (2 * BUF > EPI) ? 2 * BUF : EPI;
  constexpr int AROWS = NT / 4;           // A rows staged per pass (4 threads per row)
  constexpr int APASS = BM / AROWS;       // float4 loads of A per thread per slab
  constexpr int WN = P * BN * 2;          // uint4 loads of W per slab (2 per row-plane)
  constexpr int WV = (WN + NT - 1) / NT;  // per thread (the last round may be partial)
  static_assert(BM % AROWS == 0 && APASS >= 1, "A staging: whole passes");
  static_assert(!LNORM || (CONV == 0 && !F16), "LayerNorm epilogue: plain row GEMM only");
  static_assert(!(CONV != 0 && ABIAS), "the convolution forms have no A-side bias");
  static_assert(!F16 || P == 1, "fp16 operands: single plane only");
  static_assert(TM % 2 == 0, "the MFMAs of a slab are issued in two halves of TM / 2 row tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int lr = lane & 31, kh = lane >> 5;
  const int ntiles = N / BN;
  // XCD-aware bijective remap: the hardware deals workgroups round-robin over the 8 XCDs; give
  // each XCD a contiguous run of logical tiles (column tile fastest) so that the column tiles of
  // one row tile run side by side on ONE XCD and its A rows cross HBM -> L2 once.
  const int nb = gridDim.x, per = nb >> 3, rem = nb & 7;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int lb = xcd * per + (xcd < rem ? xcd : rem) + idx;
  const long long m0 = (long long)(lb / ntiles) * BM;
  const int n0 = (lb % ntiles) * BN;

  // ---- staging roles
  // A: thread -> (row = tid>>2 + 64*q, 16-byte segment seg = tid&3 of the row's 64-byte slab)
  const int a_seg = tid & 3;
  // LDS stores are banked mod 32 dwords and serviced per 16 (b64) / 8 (b128) CONTIGUOUS lanes, i.e.
  // per 4 consecutive staging rows; at the 12-dword row stride rows r and r + 3 share 4 banks.
  // Staging rows are therefore permuted inside each block of 8 so that the 4 rows of a store
  // group are 2 apart (offsets 0, 24, 16, 8 mod 32: conflict-free); global addresses follow.
  const int a_row = stage_row_perm(tid >> 2);
  const float* a_ptr[APASS];
  int iy0[APASS], ix0[APASS];   // CONV: top-left input pixel of the row's 3x3 window
#pragma unroll
  for (int q = 0; q < APASS; ++q) {
    long long r = m0 + a_row + AROWS * q;
    if (r >= M) r = M - 1;  // clamp: rows past M are computed on stand-in data, never stored
    if (CONV != 0) {
      const unsigned ur = (unsigned)r, gy = ur / (unsigned)g.Wo;
      const int ox = (int)(ur - gy * (unsigned)g.Wo);
      const int n = (int)(gy / (unsigned)g.Ho);
      const int oy = (int)(gy - (unsigned)n * (unsigned)g.Ho);
      const int pad = CONV == 2 ? 3 : 1;
      iy0[q] = oy * g.stride - pad;
      ix0[q] = ox * g.stride - pad;
      a_ptr[q] = A + (long long)n * g.H * g.W * g.Cin + (CONV == 2 ? 0 : a_seg * 4);   // image base
    } else if (g.H > 0) {
      // 1x1 convolution with a stride on the NHWC map (ResNet downsample): row = output pixel,
      // its A row is the input pixel (oy * stride, ox * stride) -- no strided-slice copy
      const unsigned ur = (unsigned)r, gy = ur / (unsigned)g.Wo;
      const int ox = (int)(ur - gy * (unsigned)g.Wo);
      const int n = (int)(gy / (unsigned)g.Ho);
      const int oy = (int)(gy - (unsigned)n * (unsigned)g.Ho);
      a_ptr[q] = A + (((long long)n * g.H + oy * g.stride) * g.W + ox * g.stride) * K + a_seg * 4;
    } else {
      a_ptr[q] = A + r * K + a_seg * 4;
    }
  }
  // W operand, slab-major [K/16][3][N][16] bf16 (host layout: one slab of a column tile is 3
  // contiguous 4-KiB runs): uint4 index v = tid + NT*q -> plane, row, 16-byte half
  const uint16_t* w_ptr[WV];
  int w_dst[WV];
#pragma unroll
  for (int q = 0; q < WV; ++q) {
    const int v = (tid + NT * q) < WN ? tid + NT * q : 0;   // spare threads of a partial round
    const int seg = v & 1, row = stage_row_perm((v >> 1) % BN), plane = v / (2 * BN);   // repeat item 0
    w_ptr[q] = Wp + ((long long)plane * N + n0 + row) * 16 + seg * 8;   // + slab * P*N*16
    w_dst[q] = P * A_PLANE + plane * W_PLANE + row * RST + seg * 16;
  }
  const long long w_slab = (long long)P * N * 16;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // global -> register sets.  A set holds a PAIR of slabs (32 k): the two float4 a thread loads
  // per row are the halves of one 128-byte line, so every line crosses L2 -> L1 once.  Two sets
  // alternate; a pair is loaded 3 slabs before its first use.
  f32x4 ra0[2][APASS], ra1[2][APASS];
  u32x4 rw0[2][WV], rw1[2][WV];
  f32x4 rb0[2], rb1[2];
  const int nslabs = K / BK;
  const int npairs = nslabs / 2;
  auto gload = [&](int pair, f32x4 (&ra)[2][APASS], u32x4 (&rw)[2][WV], f32x4 (&rb)[2]) {
    const int pp = pair < npairs ? pair : npairs - 1;  // past the end: repeat (never consumed)
    const int k0 = pp * 2 * BK;
    if (CONV == 2) {
      // 7x7 / stride 2 / pad 3 stem on the NCHW image: K axis = (c, ky, kx padded 7 -> 8), i.e. 24
      // rows of 8 (21 real); a thread's float4 is one half of such a row: 4 consecutive input
      // pixels of channel c, row iy -- an unaligned 16-byte load (per-element guarded at the edges)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k4 = k0 + h * BK + a_seg * 4;
        const int rowid = k4 >> 3, xh = (k4 >> 2) & 1;
        const int c = (rowid * 37) >> 8;          // rowid / 7 for rowid < 32
        const int ky = rowid - 7 * c;
#pragma unroll
        for (int q = 0; q < APASS; ++q) {
          const int iy = iy0[q] + ky, ix = ix0[q] + 4 * xh;
          const bool rowok = rowid < 21 && iy >= 0 && iy < g.H;
          const float* src = a_ptr[q] + ((long long)(c * g.H + iy) * g.W + ix);
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
          if (rowok && ix >= 0 && ix + 4 <= g.W) {
            t = *reinterpret_cast<const f32x4_u*>(src);
          } else if (rowok) {
            if (ix + 0 >= 0 && ix + 0 < g.W) t.x = src[0];
            if (ix + 1 >= 0 && ix + 1 < g.W) t.y = src[1];
            if (ix + 2 >= 0 && ix + 2 < g.W) t.z = src[2];
            if (ix + 3 >= 0 && ix + 3 < g.W) t.w = src[3];
          }
          if (xh) t.w = 0.f;                      // kx = 7: the pad column of the 8-wide row
          ra[h][q] = t;
        }
      }
    } else if (CONV == 1) {
      const int tap = k0 / g.Cin, c0 = k0 - tap * g.Cin;   // a pair (32 k) lies inside one tap
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int q = 0; q < APASS; ++q) {
        const int iy = iy0[q] + ky, ix = ix0[q] + kx;
        const bool ok = iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        const float* src = a_ptr[q] + (ok ? ((long long)iy * g.W + ix) * g.Cin + c0 : 0);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(src);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(src + BK);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ra[0][q] = ok ? t0 : z;   // zero padding
        ra[1][q] = ok ? t1 : z;
      }
    } else {
#pragma unroll
      for (int q = 0; q < APASS; ++q) {
        ra[0][q] = *reinterpret_cast<const f32x4*>(a_ptr[q] + k0);
        ra[1][q] = *reinterpret_cast<const f32x4*>(a_ptr[q] + k0 + BK);
      }
    }
#pragma unroll
    for (int q = 0; q < WV; ++q) {
      rw[0][q] = *reinterpret_cast<const u32x4*>(w_ptr[q] + (long long)(2 * pp) * w_slab);
      rw[1][q] = *reinterpret_cast<const u32x4*>(w_ptr[q] + (long long)(2 * pp + 1) * w_slab);
    }
    if (ABIAS) {
      rb[0] = *reinterpret_cast<const f32x4*>(a_bias + k0 + a_seg * 4);
      rb[1] = *reinterpret_cast<const f32x4*>(a_bias + k0 + BK + a_seg * 4);
    }
  };
  auto stage = [&](unsigned char* buf, const f32x4 (&av)[APASS], const u32x4 (&wv)[WV],
                   const f32x4 abv) {  // registers -> LDS (A split on the way)
#pragma unroll
    for (int q = 0; q < APASS; ++q) {
      float4 t = make_float4(av[q].x, av[q].y, av[q].z, av[q].w);
      if (ABIAS) {
        t.x = fmaxf(t.x + abv.x, 0.f);
        t.y = fmaxf(t.y + abv.y, 0.f);
        t.z = fmaxf(t.z + abv.z, 0.f);
        t.w = fmaxf(t.w + abv.w, 0.f);
      }
      uint2 pl[P];
      split4<P, F16>(t, pl);
      const int d = (a_row + AROWS * q) * RST + a_seg * 8;
#pragma unroll
      for (int t2 = 0; t2 < P; ++t2) *reinterpret_cast<uint2*>(buf + t2 * A_PLANE + d) = pl[t2];
    }
#pragma unroll
    for (int q = 0; q < WV; ++q) *reinterpret_cast<u32x4*>(buf + w_dst[q]) = wv[q];
  };
  const int a_rd = (wm * TM * 32 + lr) * RST + kh * 16;
  const int w_rd = P * A_PLANE + (wn * TN * 32 + lr) * RST + kh * 16;
  constexpr int TH = TM / 2;  // row tiles per MFMA half-phase
  // fragment reads: A row tiles [i0, i0 + TH) of a buffer, and all W column tiles
  auto fread_a = [&](const unsigned char* buf, u32x4 (&fa)[P][TH], int i0) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int i = 0; i < TH; ++i)
        fa[p][i] = *reinterpret_cast<const u32x4*>(buf + p * A_PLANE + a_rd + (i0 + i) * 32 * RST);
  };
  auto fread_w = [&](const unsigned char* buf, u32x4 (&fb)[P][TN]) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[p][j] = *reinterpret_cast<const u32x4*>(buf + p * W_PLANE + w_rd + j * 32 * RST);
  };
  // products kept: all (pa, pb) with pa + pb < P  (P = 3: six, error <= 2^-23; P = 2: three,
  // ~2^-16; P = 1: plain bf16), smallest terms first
  auto mma = [&](const u32x4 (&fa)[P][TH], const u32x4 (&fb)[P][TN], int i0) {
    // products outermost, tiles innermost: consecutive MFMAs write different accumulators (the
    // per-accumulator order of the six products is unchanged)
#pragma unroll
    for (int o = P - 1; o >= 0; --o)       // order o = pa + pb
#pragma unroll
      for (int pa = 0; pa <= o; ++pa)
#pragma unroll
        for (int i = 0; i < TH; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const f32x16 c = acc[i0 + i][j];
            acc[i0 + i][j] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                       __builtin_bit_cast(f16x8, fa[pa][i]),
                                       __builtin_bit_cast(f16x8, fb[o - pa][j]), c, 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                       __builtin_bit_cast(bf16x8, fa[pa][i]),
                                       __builtin_bit_cast(bf16x8, fb[o - pa][j]), c, 0, 0, 0);
          }
  };

  // fragment registers: the two A halves of the current slab, the current and the next W set
  u32x4 faL[P][TH], faH[P][TH], fbA[P][TN], fbB[P][TN];
  unsigned char* buf0 = smem;
  unsigned char* buf1 = smem + BUF;
  // prologue: pairs 0 and 1 in registers, slab 0 staged, its first fragments read
  gload(0, ra0, rw0, rb0);
  gload(1, ra1, rw1, rb1);
  stage(buf0, ra0[0], rw0[0], rb0[0]);
  __syncthreads();
  fread_a(buf0, faL, 0);
  fread_w(buf0, fbA);

  // One slab = two half-phases of TH*TN*6 MFMAs.  Phase 1 (low row tiles) covers the LDS reads
  // of the high tiles' fragments and the staging of slab s+1 (STAGE_); phase 2 (high tiles)
  // covers the reads of slab s+1's first fragments.  No branch in the body.
#define PAVE_SLAB(bufc, bufn, fbc, fbn, STAGE_, LOAD_) \
  {                                                    \
    fread_a(bufc, faH, TH);                            \
    STAGE_;                                            \
    LOAD_;                                             \
    mma(faL, fbc, 0);                                  \
    __syncthreads(); /* bufn complete everywhere */    \
    fread_a(bufn, faL, 0);                             \
    fread_w(bufn, fbn);                                \
    mma(faH, fbc, TH);                                 \
  }
  // 4 slabs per trip (K % 64 == 0): pair P = 2t in set 0, pair P+1 in set 1.  A set is
  // reloaded with the pair two ahead right after its second slab has been staged.
  for (int s = 0; s < nslabs; s += 4) {
    const int pr = s >> 1;
    PAVE_SLAB(buf0, buf1, fbA, fbB, stage(buf1, ra0[1], rw0[1], rb0[1]), gload(pr + 2, ra0, rw0, rb0))
    PAVE_SLAB(buf1, buf0, fbB, fbA, stage(buf0, ra1[0], rw1[0], rb1[0]), (void)0)
    PAVE_SLAB(buf0, buf1, fbA, fbB, stage(buf1, ra1[1], rw1[1], rb1[1]), gload(pr + 3, ra1, rw1, rb1))
    PAVE_SLAB(buf1, buf0, fbB, fbA, stage(buf0, ra0[0], rw0[0], rb0[0]), (void)0)
  }
#undef PAVE_SLAB

  // ---- epilogue: per wave, 32-row chunks of its (TM*32) x (TN*32) tile through LDS
  constexpr int CW = TN * 32;            // chunk width (floats)
  constexpr int CST = CW + 4;            // row stride, keeps float4 alignment
  constexpr int RV = CW / 4;             // float4 per chunk row
  constexpr int RPP = 64 / RV;           // rows per pass of the wave
  constexpr int NPASS = 32 / RPP;
  static_assert(NW * 32 * CST * 4 <= SMEM, "epilogue chunks must fit the operand buffers");
  __syncthreads();  // operand tiles fully consumed by every wave
  float* Cs = reinterpret_cast<float*>(smem) + wave * 32 * CST;
  const int c4 = lane % RV;
  const int ncol = n0 + wn * CW + c4 * 4;
  const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + ncol)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
  // output segment of this column tile (block-uniform)
  const bool seg2 = os.out2 != nullptr && n0 >= os.nsplit;
  float* const obase = seg2 ? os.out2 : out;
  const int ldo = os.out2 == nullptr ? N : (seg2 ? N - os.nsplit : os.nsplit);
  const int ocol = seg2 ? ncol - os.nsplit : ncol;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[((r & 3) + 8 * (r >> 2) + 4 * kh) * CST + j * 32 + lr] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4 res[NPASS];
    if (residual) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const long long gm = m0 + wm * TM * 32 + i * 32 + ps * RPP + lane / RV;
        const long long rr = os.res_rows ? (long long)((unsigned)gm % (unsigned)os.res_rows) : gm;
        res[ps] = gm < M ? *reinterpret_cast<const float4*>(residual + rr * N + ncol)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if constexpr (LNORM) {
      // v = acc + bias + residual for this wave's 64-column share of 32 rows; the row statistics
      // are completed across the WGN waves of the row through LDS (two passes: mean, then the
      // centred sum of squares -- no E[x^2] - mean^2 cancellation)
      float* st1 = reinterpret_cast<float*>(smem) + NW * 32 * CST;   // [BM][WGN] partial sums
      float* st2 = st1 + BM * WGN;                                     // [BM][WGN] partial M2
      const float4 g4 = *reinterpret_cast<const float4*>(ln.gamma + ncol);
      const float4 be4 = *reinterpret_cast<const float4*>(ln.beta + ncol);
      float4 v[NPASS];
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int lrow = ps * RPP + lane / RV;
        const long long gm = m0 + wm * TM * 32 + i * 32 + lrow;
        float4 t = *reinterpret_cast<const float4*>(Cs + lrow * CST + c4 * 4);
        t.x += b4.x, t.y += b4.y, t.z += b4.z, t.w += b4.w;
        if (residual) t.x += res[ps].x, t.y += res[ps].y, t.z += res[ps].z, t.w += res[ps].w;
        if (gm >= M) t = make_float4(0.f, 0.f, 0.f, 0.f);
        v[ps] = t;
        float sm = (t.x + t.y) + (t.z + t.w);
#pragma unroll
        for (int o = RV / 2; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        if (c4 == 0) st1[(wm * TM * 32 + i * 32 + lrow) * WGN + wn] = sm;
      }
      __syncthreads();
      float mean[NPASS];
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int brow = wm * TM * 32 + i * 32 + ps * RPP + lane / RV;
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < WGN; ++w) sm += st1[brow * WGN + w];
        mean[ps] = sm * (1.f / (float)BN);
        float4& t = v[ps];
        t.x -= mean[ps], t.y -= mean[ps], t.z -= mean[ps], t.w -= mean[ps];
        float q = (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
#pragma unroll
        for (int o = RV / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        if (c4 == 0) st2[brow * WGN + wn] = q;
      }
      __syncthreads();
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int lrow = ps * RPP + lane / RV;
        const int brow = wm * TM * 32 + i * 32 + lrow;
        const long long gm = m0 + brow;
        float q = 0.f;
#pragma unroll
        for (int w = 0; w < WGN; ++w) q += st2[brow * WGN + w];
        const float rstd = rsqrtf(q * (1.f / (float)BN) + ln.eps);
        if (gm < M) {
          float4 t = v[ps];
          t.x = fmaf(t.x * rstd, g4.x, be4.x);
          t.y = fmaf(t.y * rstd, g4.y, be4.y);
          t.z = fmaf(t.z * rstd, g4.z, be4.z);
          t.w = fmaf(t.w * rstd, g4.w, be4.w);
          *reinterpret_cast<float4*>(out + gm * N + ncol) = t;
        }
      }
    } else
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int lrow = ps * RPP + lane / RV;
      const long long gm = m0 + wm * TM * 32 + i * 32 + lrow;
      if (gm < M) {
        float4 v = *reinterpret_cast<const float4*>(Cs + lrow * CST + c4 * 4);
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
        *reinterpret_cast<float4*>(obase + gm * ldo + ocol) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// P = 3 needs the whole register file of a SIMD (one wave each); with fewer planes two workgroups
// fit a CU, which the bandwidth-bound P = 1 form wants.
template <int TM, int TN, bool ABIAS, int P, bool F16, int CONV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16x3_kernel(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const int relu,
    const float* __restrict__ a_bias, const ConvGeom g, const OutSplit os, const LnArgs ln) {
  gemm_split_body<TM, TN, ABIAS, P, F16, CONV, 2, false>(A, Wp, bias, residual, out, M, K, N, relu,
                                                         a_bias, g, os, ln);
}
template <int TM, int TN, bool ABIAS, int P, bool F16, int CONV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16x3_kernel_occ2(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const int relu,
    const float* __restrict__ a_bias, const ConvGeom g, const OutSplit os, const LnArgs ln) {
  gemm_split_body<TM, TN, ABIAS, P, F16, CONV, 2, false>(A, Wp, bias, residual, out, M, K, N, relu,
                                                         a_bias, g, os, ln);
}
// 128 x 256 block tile, 8 waves (2 x 4), one workgroup per CU at two waves per SIMD: the A tile
// is split (VALU) and staged once for twice the MFMA work of the 128 x 128 form, and with N == 256
// the block owns whole output rows, so LayerNorm can run in the epilogue (LNORM).
template <int TM, int TN, bool ABIAS, int P, int CONV, bool LNORM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16x3_kernel_w8(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const int relu,
    const float* __restrict__ a_bias, const ConvGeom g, const OutSplit os, const LnArgs ln) {
  gemm_split_body<TM, TN, ABIAS, P, false, CONV, 4, LNORM>(A, Wp, bias, residual, out, M, K, N, relu,
                                                           a_bias, g, os, ln);
}

// fp32 [n] -> nplanes bf16 planes [nplanes][n]: truncation terms, the last one rounded to
// nearest even (nplanes = 3: x = p0 + p1 + p2 exactly)
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ x,
                                                           uint16_t* __restrict__ planes,
                                                           const long long n, const int nplanes,
                                                           const int f16) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (long long)gridDim.x * 256) {
    float v = x[i];
    for (int t = 0; t < nplanes - 1; ++t) {
      const unsigned h = hi16(v);
      planes[t * n + i] = (uint16_t)(h >> 16);
      v -= __uint_as_float(h);
    }
    planes[(nplanes - 1) * n + i] =
        (uint16_t)((f16 ? pack_rne<true>(v, 0.f) : pack_rne<false>(v, 0.f)) & 0xffffu);
  }
}

#ifdef PAVE_DIAG
int g_diag_variant = 0;
#else
constexpr int g_diag_variant = 0;   // the shipped library has no kernel-form global
#endif
                         // -DPAVE_DIAG build only (pave_diag_gemm_variant): 2 = the 256-row tile forms,
                         // 3 = 128 x 256 / 8-wave tiles wherever N % 256 == 0, 4 = never,
                         // 9 = first-generation kernels for every 3-plane form (A/B against
                         // the LDS-DMA generation of pave_gemm_dma.hip, the default),
                         // 5 = 3x3 form with 64-bit lane addresses (not buffer-addressed),
                         // 6 = no split-K plan,
                         // 13 / 14 = LayerNorm-epilogue GEMM: always the 8-wave / the wide form,
                         // 8 = LDS-DMA generation without its wide tile form, 7 = wide
                         // tile form wherever it applies (default: from 512 tiles up),
                         // 18 = Swin window attention on the per-lane (LDS broadcast) form instead of the
                         // fp32-MFMA form (pave_decoder.hip),
                         // 15 / 16 = two row tiles per wave (256-row blocks) for the 64- / 96-column
                         // tile forms and the ResNet layer1 chain: never / wherever the form exists,
                         // 17 = no half-tail form (33 .. 48 output columns of a 3x3 as zero-padded 32x32x16
                         // products instead of 16x16x32 products over slab pairs),
                         // 19 = the small-row selection of rounds 4 - 5 (no K-split small-row form: the forms the
                         // bit-equality tests compare), 20 = the wide GEMM capped at one block per CU (40 KiB of
                         // unused dynamic LDS; tools/coresidency_probe.py)

// Shapes that take the 128 x 256, 8-wave tile (measured per shape, tools/bench_gemm_shapes.py)
bool use_w8(long long M, int K, int N) {
  if (N % 256 != 0 || g_diag_variant == 4 || g_diag_variant == 2) return false;
  if (g_diag_variant == 3) return true;
  (void)M;
  (void)K;
  return false;
}

template <int TM, int TN, bool ABIAS, int P, bool F16, int CONV, bool OCC2 = (P == 1), int WGN = 2,
          bool LNORM = false>
int launch_gemm(const float* a, const uint16_t* w, const float* bias, const float* residual,
                float* out, long long M, int K, int N, int relu, const float* a_bias,
                hipStream_t st, const ConvGeom g = ConvGeom{0, 0, 0, 0, 0, 0},
                const OutSplit os = OutSplit{nullptr, 0, 0},
                const LnArgs ln = LnArgs{nullptr, nullptr, 0.f}) {
  constexpr int NT = 128 * WGN, NW = 2 * WGN;
  constexpr int BM = 2 * TM * 32, BN = WGN * TN * 32;
  constexpr int EPI = NW * 32 * (TN * 32 + 4) * 4 + (LNORM ? 2 * BM * WGN * 4 : 0);
  constexpr int SMEM = (2 * P * (BM + BN) * RST > EPI) ? 2 * P * (BM + BN) * RST : EPI;
  static_assert(WGN == 2 || (WGN == 4 && !F16), "wave grids: 2 x 2 or 2 x 4");
  const long long gx = ((M + BM - 1) / BM) * (N / BN);
  if (gx >= (1ll << 31)) return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3: grid too large");
  using kern_t = void (*)(const float*, const uint16_t*, const float*, const float*, float*, int, int,
                          int, int, const float*, ConvGeom, OutSplit, LnArgs);
  kern_t kern;
  if constexpr (WGN == 4) kern = gemm_bf16x3_kernel_w8<TM, TN, ABIAS, P, CONV, LNORM>;
  else if constexpr (OCC2) kern = gemm_bf16x3_kernel_occ2<TM, TN, ABIAS, P, F16, CONV>;
  else kern = gemm_bf16x3_kernel<TM, TN, ABIAS, P, F16, CONV>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return pave_internal_fail(PAVE_E_LAUNCH, "gemm_bf16x3: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(NT), SMEM, st, a, w, bias, residual, out,
                     (int)M, K, N, relu, a_bias, g, os, ln);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

}  // namespace

extern "C" {

int pave_split_bf16x3_f32(const float* x, void* planes, long long n, int nplanes, void* stream) {
  const int f16 = nplanes == PAVE_PLANES_FP16;
  if (f16) nplanes = 1;
  if (!x || !planes || n <= 0 || nplanes < 1 || nplanes > 3)
    return pave_internal_fail(PAVE_E_ARG, "split_bf16x3: bad argument (1 <= nplanes <= 3)");
  const long long nb = (n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536;
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)nb), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, static_cast<uint16_t*>(planes), n,
                     nplanes, f16);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

static int gemm_split_entry(const float* a, const float* a_bias, const void* w_planes,
                            const float* bias, const float* residual, float* out, long long M,
                            int K, int N, int relu, int nplanes, void* stream, OutSplit os);
// nplanes of the C ABI -> the LDS-DMA generation's operand planes (3 | 1 = fp16), 0 = not one of its modes
static inline int q_planes(int nplanes) { return nplanes == 3 ? 3 : (nplanes == PAVE_PLANES_FP16 ? 1 : 0); }

int pave_gemm_bf16x3_encproj_f32(const float* a, const void* w_planes, const float* table,
                                 long long table_rows, const float* value_bias, const float* ref,
                                 const int* levels_hw, float* value, float* samp, long long M, int K,
                                 int nplanes, void* stream) {
  if (!a || !w_planes || !table || !ref || !levels_hw || !value || !samp)
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_encproj: null pointer");
  if (M <= 0 || M >= (1ll << 31) || table_rows <= 0 || table_rows >= (1ll << 31))
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_encproj: bad sizes (0 < M, table_rows < 2^31)");
  if (!q_planes(nplanes)) return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_encproj: nplanes must be 3 or PAVE_PLANES_FP16");
  return pave_internal_gemm_encproj(a, w_planes, table, table_rows, value_bias, ref, levels_hw, value, samp, M, K,
                                    stream, q_planes(nplanes));
}

int pave_gemm_bf16x3_f32(const float* a, const float* a_bias, const void* w_planes,
                         const float* bias, const float* residual, float* out, long long M, int K,
                         int N, int relu, int nplanes, void* stream) {
  return gemm_split_entry(a, a_bias, w_planes, bias, residual, out, M, K, N, relu, nplanes, stream,
                          OutSplit{nullptr, 0, 0});
}

int pave_gemm_bf16x3_ex_f32(const float* a, const float* a_bias, const void* w_planes,
                            const float* bias, const float* residual, long long residual_rows,
                            float* out, float* out2, int n_split, long long M, int K, int N,
                            int relu, int nplanes, void* stream) {
  if (residual_rows < 0 || residual_rows >= (1ll << 31) || (residual_rows > 0 && !residual))
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_ex: bad residual_rows");
  if (out2 && (n_split <= 0 || n_split >= N || n_split % 128 != 0))
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_ex: 0 < n_split < N, n_split %% 128 == 0");
  if (out2 && residual == out)
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_ex: in-place residual needs one output");
  return gemm_split_entry(a, a_bias, w_planes, bias, residual, out, M, K, N, relu, nplanes, stream,
                          OutSplit{out2, out2 ? n_split : 0,
                                   residual_rows >= M ? 0 : (int)residual_rows});
}

static int gemm_split_entry(const float* a, const float* a_bias, const void* w_planes,
                            const float* bias, const float* residual, float* out, long long M,
                            int K, int N, int relu, int nplanes, void* stream, OutSplit os) {
  if (!a || !w_planes || !out) return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3: null pointer");
  if (relu < 0 || relu > 3 || (relu >= 2 && (!q_planes(nplanes) || g_diag_variant == 9)))
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3: relu = 0 | 1 | 2 | 3 (2 = exact GELU, 3 = sigmoid: 3 planes / fp16 only)");
  if (M <= 0 || K <= 0 || N <= 0 || M >= (1ll << 31))
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3: bad sizes (0 < M < 2^31)");
  // 3 planes (the exact split): the LDS-DMA generation (pave_gemm_dma.hip) takes K %% 32 == 0 and any
  // N %% 4 == 0 -- the weight planes then carry roundup(N, 64) rows (zero rows beyond N), out / bias /
  // residual have N columns.  The kernels of this file keep the 1- / 2-plane and fp16 modes.
  if (q_planes(nplanes) && g_diag_variant != 9 && K % 32 == 0 && K >= 64 && N % 4 == 0 &&
      (N % 64 == 0 || !os.out2))
    return pave_internal_gemm_q(a, a_bias, w_planes, bias, residual, os.res_rows, out, os.out2, os.nsplit,
                                M, K, (N + 63) / 64 * 64, relu, 0, 0, 0, 0, 0, 0, 0, stream, nullptr, N, 1, 0,
                                q_planes(nplanes));
  if (K % 64 != 0 || (N % 128 != 0 && !(N == 64 && nplanes == 3 && !os.out2)))
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3: K %% 64 == 0 and N %% 128 == 0 (or N == 64 "
                                          "with 3 planes) required");
  if ((nplanes < 1 || nplanes > 3) && nplanes != PAVE_PLANES_FP16)
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3: nplanes must be 1, 2, 3 or PAVE_PLANES_FP16");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint16_t* w = static_cast<const uint16_t*>(w_planes);
  if (N == 64) {  // 128 x 64 tiles (the ResNet layer1 1x1 reductions): HBM-bound, A read once
    const ConvGeom gz{0, 0, 0, 0, 0, 0};
    if (a_bias)
      return launch_gemm<2, 1, true, 3, false, 0, true>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, gz, os);
    return launch_gemm<2, 1, false, 3, false, 0, true>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, gz, os);
  }
  // P = 3: 128 x 128 tiles, two workgroups (8 waves) per CU -- one workgroup's operand split
  // (VALU) and LDS traffic overlap the other's MFMAs: 12-25 % faster than the 256 x 128 tile at
  // one wave per SIMD on every shape of the model (tools/bench_gemm_shapes.py)
  const ConvGeom g0{0, 0, 0, 0, 0, 0};
  // 128 x 256 tile on 8 waves where the shape allows it and it wins (see use_w8)
  if (nplanes == 3 && use_w8(M, K, N) && (!os.out2 || os.nsplit % 256 == 0)) {
    if (a_bias)
      return launch_gemm<2, 2, true, 3, false, false, true, 4>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g0, os);
    return launch_gemm<2, 2, false, 3, false, false, true, 4>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g0, os);
  }
#define PAVE_GO(AB, P) \
  return (P == 3 && g_diag_variant != 2) \
      ? launch_gemm<2, 2, AB, P, false, false, true>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g0, os) \
      : launch_gemm<4, 2, AB, P, false, false>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g0, os)
  if (nplanes == PAVE_PLANES_FP16) {
    if (a_bias) return launch_gemm<4, 2, true, 1, true, false>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g0, os);
    return launch_gemm<4, 2, false, 1, true, false>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g0, os);
  }
  if (a_bias) {
    if (nplanes == 3) PAVE_GO(true, 3);
    if (nplanes == 2) PAVE_GO(true, 2);
    PAVE_GO(true, 1);
  }
  if (nplanes == 3) PAVE_GO(false, 3);
  if (nplanes == 2) PAVE_GO(false, 2);
  PAVE_GO(false, 1);
#undef PAVE_GO
}

// Split-K form of the plain row GEMM (few row tiles, K >= 2048: ResNet layer4's 1x1 reductions and the neck's
// C5 lateral on a one-clip batch).  The plan is pave_internal_splitk_plan's -- the 3x3 form's.
long long pave_gemm_splitk_workspace_bytes(long long M, int K, int N) {
  if (M <= 0 || M >= (1ll << 31) || K < 64 || K % 32 != 0 || N <= 0 || N % 4 != 0 || g_diag_variant == 9) return 0;
  int parts, per;
  pave_internal_splitk_plan(M, K, (N + 63) / 64 * 64, &parts, &per);
  return parts > 1 ? (long long)parts * M * N * 4 : 0;
}

int pave_gemm_bf16x3_splitk_f32(const float* a, const void* w_planes, const float* bias, const float* residual,
                                float* out, long long M, int K, int N, int relu, int nplanes, void* workspace,
                                long long workspace_bytes, void* stream) {
  if (!a || !w_planes || !out || !workspace || (relu != 0 && relu != 1))
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_splitk: null pointer (or relu not 0 | 1)");
  if (!q_planes(nplanes)) return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_splitk: nplanes must be 3 or PAVE_PLANES_FP16");
  const long long need = pave_gemm_splitk_workspace_bytes(M, K, N);
  if (need == 0 || workspace_bytes < need)
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_splitk: shape has no split-K plan (use pave_gemm_bf16x3_f32) or "
                                          "the workspace is smaller than pave_gemm_splitk_workspace_bytes");
  const int Np = (N + 63) / 64 * 64;
  int parts, per;
  pave_internal_splitk_plan(M, K, Np, &parts, &per);
  float* ws = static_cast<float*>(workspace);
  const int rc = pave_internal_gemm_q(a, nullptr, w_planes, nullptr, nullptr, 0, ws, nullptr, 0, M, K, Np, 0, 0, 0, 0,
                                      0, 0, 0, 0, stream, nullptr, N, parts, per, q_planes(nplanes));
  if (rc != PAVE_OK) return rc;
  return pave_internal_splitk_reduce(ws, parts, M, N, bias, residual, relu, out, stream);
}

int pave_gemm_fp16_act_f32(const void* a, int a_is_f16, const void* w_plane, const float* bias,
                           const float* residual, const float* gamma, const float* beta, float eps, void* out,
                           int out_is_f16, long long M, int K, int N, int relu, void* stream) {
  if (!a || !w_plane || !out) return pave_internal_fail(PAVE_E_ARG, "gemm_fp16_act: null pointer");
  if (M <= 0 || M >= (1ll << 31) || K <= 0 || N <= 0)
    return pave_internal_fail(PAVE_E_ARG, "gemm_fp16_act: bad sizes (0 < M < 2^31)");
  return pave_internal_gemm_f16act(a, a_is_f16 != 0, w_plane, bias, residual, gamma, beta, eps, out, out_is_f16 != 0,
                                   M, K, N, relu, stream);
}

int pave_gemm_bf16x3_cat_f32(const float* a, long long K1, const float* a2, const void* w_planes,
                             const float* bias, const float* residual, float* out, long long M, int K,
                             int N, int relu, int nplanes, void* stream) {
  if (!a || !a2 || !w_planes || !out) return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_cat: null pointer");
  if (M <= 0 || M >= (1ll << 31) || K <= 0 || N <= 0)
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_cat: bad sizes (0 < M < 2^31)");
  if (K % 32 != 0 || K < 64 || N % 64 != 0 || K1 <= 0 || K1 >= K || K1 % 16 != 0)
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_cat: K %% 32 == 0, N %% 64 == 0, 0 < K1 < K, K1 %% 16 == 0");
  if (!q_planes(nplanes)) return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_cat: nplanes must be 3 or PAVE_PLANES_FP16");
  return pave_internal_gemm_q(a, nullptr, w_planes, bias, residual, 0, out, nullptr, 0, M, K, N, relu, 4,
                              0, 0, (int)K1, 0, 0, 0, stream, a2, 0, 1, 0, q_planes(nplanes));
}

int pave_gemm_bf16x3_grouped_f32(const float* a, long long lda, const void* w_planes, const float* bias,
                                 float* out, long long M, int K, int N, int group_n, int relu,
                                 int nplanes, void* stream) {
  if (!a || !w_planes || !out) return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_grouped: null pointer");
  if (M <= 0 || M >= (1ll << 31) || K <= 0 || N <= 0 || group_n <= 0)
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_grouped: bad sizes (0 < M < 2^31)");
  if (K % 32 != 0 || K < 64 || N % group_n != 0 || group_n % 64 != 0 ||
      lda < (long long)(N / group_n) * K || lda % 4 != 0 || lda >= (1ll << 23))
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_grouped: K %% 32 == 0, group_n %% 64 == 0, N %% group_n == 0, "
                                          "groups * K <= lda < 2^23 (a tile's 127 rows x lda x 4 B is a 32-bit lane offset), lda %% 4 == 0");
  // column tiles must not straddle a group: 128-wide tiles need group_n %% 128 == 0
  const int np = (N % 128 == 0 && group_n % 128 == 0) ? N : -N;
  if (!q_planes(nplanes)) return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_grouped: nplanes must be 3 or PAVE_PLANES_FP16");
  return pave_internal_gemm_q(a, nullptr, w_planes, bias, nullptr, 0, out, nullptr, 0, M, K, np, relu, 0,
                              (int)lda, group_n, 0, 0, 0, 0, stream, nullptr, 0, 1, 0, q_planes(nplanes));
}

int pave_gemm_bf16x3_ln_f32(const float* a, const void* w_planes, const float* bias,
                            const float* residual, const float* gamma, const float* beta, float eps,
                            float* out, long long M, int K, int N, int nplanes, void* stream) {
  if (!a || !w_planes || !out || !gamma || !beta)
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_ln: null pointer");
  if (M <= 0 || K <= 0 || M >= (1ll << 31))
    return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_ln: bad sizes (0 < M < 2^31)");
  if (K % 64 != 0 || N != 256)
    return pave_internal_fail(PAVE_E_UNSUPPORTED, "gemm_bf16x3_ln: K %% 64 == 0 and N == 256 required");
  if (!q_planes(nplanes)) return pave_internal_fail(PAVE_E_ARG, "gemm_bf16x3_ln: nplanes must be 3 or PAVE_PLANES_FP16");
  if (g_diag_variant != 9 || nplanes != 3)
    return pave_internal_gemm_q_ln(a, w_planes, bias, residual, gamma, beta, eps, out, M, K, N, stream,
                                   q_planes(nplanes));
  return launch_gemm<2, 2, false, 3, false, false, true, 4, true>(
      a, static_cast<const uint16_t*>(w_planes), bias, residual, out, M, K, N, 0, nullptr,
      reinterpret_cast<hipStream_t>(stream), ConvGeom{0, 0, 0, 0, 0, 0}, OutSplit{nullptr, 0, 0},
      LnArgs{gamma, beta, eps});
}

#ifdef PAVE_DIAG
void pave_diag_gemm_variant(int v) { g_diag_variant = v; }  // not part of the C ABI (tests/, tools/)
#endif
}
#ifdef PAVE_DIAG
int pave_internal_diag_variant() { return g_diag_variant; }
#endif
extern "C" {

int pave_conv3x3_split_f32(const float* x, const void* w_planes, const float* bias,
                           const float* residual, float* y, int N, int H, int W, int Cin, int Cout,
                           int stride, int relu, int nplanes, void* stream) {
  if (!x || !w_planes || !y) return pave_internal_fail(PAVE_E_ARG, "conv3x3_split: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || (relu != 0 && relu != 1))
    return pave_internal_fail(PAVE_E_ARG, "conv3x3_split: bad sizes (stride 1 or 2; relu 0 | 1)");
  const bool padded = Cin % 64 != 0 || Cout % 64 != 0;   // zero-padded weight planes: 3-plane DMA kernel only
  if (Cin <= 0 || Cout <= 0 || Cin % 16 != 0 || Cout % 4 != 0 ||
      ((padded || residual) && (!q_planes(nplanes) || g_diag_variant == 9)))
    return pave_internal_fail(PAVE_E_ARG, "conv3x3_split: Cin %% 16 == 0 and Cout %% 4 == 0 (3 planes; the other "
                                          "modes: Cin, Cout %% 64 == 0, no residual) required");
  if ((nplanes < 1 || nplanes > 3) && nplanes != PAVE_PLANES_FP16)
    return pave_internal_fail(PAVE_E_ARG, "conv3x3_split: nplanes must be 1, 2, 3 or PAVE_PLANES_FP16");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long long M = (long long)N * Ho * Wo;
  if (M >= (1ll << 31) || (long long)N * H * W * Cin >= (1ll << 40))
    return pave_internal_fail(PAVE_E_ARG, "conv3x3_split: tensor too large");
  if (q_planes(nplanes) && g_diag_variant != 9)   // K = 9 Cin padded to a multiple of 32, Cout to one of 64
    return pave_internal_gemm_q(x, nullptr, w_planes, bias, residual, 0, y, nullptr, 0, M,
                                (9 * Cin + 31) / 32 * 32, (Cout + 63) / 64 * 64, relu, 1, H, W, Cin, Ho,
                                Wo, stride, stream, nullptr, Cout, 1, 0, q_planes(nplanes));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint16_t* w = static_cast<const uint16_t*>(w_planes);
  const ConvGeom g{H, W, Cin, Ho, Wo, stride};
  const int K = 9 * Cin;
#define PAVE_CV(TN_, P_, F_) \
  return (P_ == 3 && g_diag_variant != 2) \
      ? launch_gemm<2, TN_, false, P_, F_, true, true>(x, w, bias, nullptr, y, M, K, Cout, relu, nullptr, st, g) \
      : launch_gemm<4, TN_, false, P_, F_, true>(x, w, bias, nullptr, y, M, K, Cout, relu, nullptr, st, g)
  if (Cout % 128 == 0) {
    if (nplanes == PAVE_PLANES_FP16) PAVE_CV(2, 1, true);
    if (nplanes == 3) PAVE_CV(2, 3, false);
    if (nplanes == 2) PAVE_CV(2, 2, false);
    PAVE_CV(2, 1, false);
  }
  if (nplanes == PAVE_PLANES_FP16) PAVE_CV(1, 1, true);
  if (nplanes == 3) PAVE_CV(1, 3, false);
  if (nplanes == 2) PAVE_CV(1, 2, false);
  PAVE_CV(1, 1, false);
#undef PAVE_CV
}

long long pave_conv3x3_splitk_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2) ||
      Cin % 16 != 0 || Cout % 4 != 0 || g_diag_variant == 9)
    return 0;
  const long long M = (long long)N * ((H - 1) / stride + 1) * ((W - 1) / stride + 1);
  int parts, per;
  pave_internal_splitk_plan(M, (9 * Cin + 31) / 32 * 32, (Cout + 63) / 64 * 64, &parts, &per);
  return parts > 1 ? (long long)parts * M * Cout * 4 : 0;
}

int pave_conv3x3_splitk_f32(const float* x, const void* w_planes, const float* bias,
                            const float* residual, float* y, int N, int H, int W, int Cin, int Cout,
                            int stride, int relu, void* workspace, long long workspace_bytes,
                            int nplanes, void* stream) {
  if (!x || !w_planes || !y || !workspace || (relu != 0 && relu != 1))
    return pave_internal_fail(PAVE_E_ARG, "conv3x3_splitk: null pointer (or relu not 0 | 1)");
  const long long need = pave_conv3x3_splitk_workspace_bytes(N, H, W, Cin, Cout, stride);
  if (need == 0 || workspace_bytes < need)
    return pave_internal_fail(PAVE_E_ARG, "conv3x3_splitk: shape has no split-K plan (use pave_conv3x3_split_f32) "
                                          "or the workspace is smaller than pave_conv3x3_splitk_workspace_bytes");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long long M = (long long)N * Ho * Wo;
  if ((long long)N * H * W * Cin >= (1ll << 40))
    return pave_internal_fail(PAVE_E_ARG, "conv3x3_splitk: tensor too large");
  const int Kp = (9 * Cin + 31) / 32 * 32, Np = (Cout + 63) / 64 * 64;
  int parts, per;
  pave_internal_splitk_plan(M, Kp, Np, &parts, &per);
  float* ws = static_cast<float*>(workspace);
  if (!q_planes(nplanes)) return pave_internal_fail(PAVE_E_ARG, "conv3x3_splitk: nplanes must be 3 or PAVE_PLANES_FP16");
  const int rc = pave_internal_gemm_q(x, nullptr, w_planes, nullptr, nullptr, 0, ws, nullptr, 0, M, Kp, Np, 0,
                                      1, H, W, Cin, Ho, Wo, stride, stream, nullptr, Cout, parts, per,
                                      q_planes(nplanes));
  if (rc != PAVE_OK) return rc;
  return pave_internal_splitk_reduce(ws, parts, M, Cout, bias, residual, relu, y, stream);
}

int pave_conv1x1_strided_split_f32(const float* x, const void* w_planes, const float* bias, float* y,
                                   int N, int H, int W, int Cin, int Cout, int stride, int relu,
                                   int nplanes, void* stream) {
  if (!x || !w_planes || !y) return pave_internal_fail(PAVE_E_ARG, "conv1x1_strided_split: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || stride < 1 || stride > 4)
    return pave_internal_fail(PAVE_E_ARG, "conv1x1_strided_split: bad sizes (1 <= stride <= 4)");
  if (Cin % 64 != 0 || Cout % 128 != 0 || Cin <= 0 || Cout <= 0)
    return pave_internal_fail(PAVE_E_ARG, "conv1x1_strided_split: Cin %% 64 == 0 and Cout %% 128 == 0 required");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long long M = (long long)N * Ho * Wo;
  if (M >= (1ll << 31) || (long long)N * H * W * Cin >= (1ll << 40))
    return pave_internal_fail(PAVE_E_ARG, "conv1x1_strided_split: tensor too large");
  // (the LDS-DMA kernel addresses the strided pixels with 32-bit byte offsets from x)
  if (!q_planes(nplanes)) return pave_internal_fail(PAVE_E_ARG, "conv1x1_strided_split: nplanes must be 3 or PAVE_PLANES_FP16");
  if ((g_diag_variant != 9 || nplanes != 3) && (long long)N * H * W * Cin * 4 < (1ll << 32))
    return pave_internal_gemm_q(x, nullptr, w_planes, bias, nullptr, 0, y, nullptr, 0, M, Cin, Cout, relu,
                                3, H, W, Cin, Ho, Wo, stride, stream, nullptr, 0, 1, 0, q_planes(nplanes));
  if (nplanes != 3) return pave_internal_fail(PAVE_E_UNSUPPORTED, "conv1x1_strided_split: fp16 operands need a map below 4 GiB");
  const ConvGeom g{H, W, Cin, Ho, Wo, stride};
  return launch_gemm<2, 2, false, 3, false, 0, true>(x, static_cast<const uint16_t*>(w_planes), bias,
                                                     nullptr, y, M, Cin, Cout, relu, nullptr,
                                                     reinterpret_cast<hipStream_t>(stream), g);
}

int pave_conv7x7s2_nchw_split_f32(const float* x, const void* w_planes, const float* bias, float* y,
                                  int N, int H, int W, int row_pitch, int Cout, int relu, int nplanes,
                                  void* stream) {
  if (!x || !w_planes || !y) return pave_internal_fail(PAVE_E_ARG, "conv7x7s2_nchw_split: null pointer");
  if (N <= 0 || H <= 0 || W < 8 || (relu != 0 && relu != 1))
    return pave_internal_fail(PAVE_E_ARG, "conv7x7s2_nchw_split: bad sizes (W >= 8; relu 0 | 1)");
  if (row_pitch == 0) row_pitch = W;
  if (row_pitch < W || (row_pitch != W && ((row_pitch & 3) || (reinterpret_cast<uintptr_t>(x) & 15))))
    return pave_internal_fail(PAVE_E_ARG, "conv7x7s2_nchw_split: row_pitch >= W; a pitch beyond W must be a "
                                          "multiple of 4 on a 16-byte aligned image (pave_repitch_rows_f32)");
  if (Cout != 64)
    return pave_internal_fail(PAVE_E_UNSUPPORTED, "conv7x7s2_nchw_split: Cout == 64 (the ResNet / HRNet stem)");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long M = (long long)N * Ho * Wo;
  if (M >= (1ll << 31) || (long long)N * 3 * H * row_pitch >= (1ll << 40))
    return pave_internal_fail(PAVE_E_ARG, "conv7x7s2_nchw_split: tensor too large");
  // w_planes holds two layouts of the same weights (ops.split_stem7x7_weight): 12 slabs
  // (c, ky, kx | pad) for the kernel of this file, then 11 slabs (c, ky, kx + 1) for the LDS-window
  // kernel of pave_gemm_dma.hip, which needs 16-byte aligned image rows
  if (!q_planes(nplanes)) return pave_internal_fail(PAVE_E_ARG, "conv7x7s2_nchw_split: nplanes must be 3 or PAVE_PLANES_FP16");
  if ((g_diag_variant != 9 || nplanes != 3 || row_pitch != W) && row_pitch % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    return pave_internal_stem7x7_q(x, static_cast<const uint16_t*>(w_planes) + 12 * q_planes(nplanes) * 64 * 16, bias, y,
                                   N, H, W, row_pitch, relu, stream, q_planes(nplanes));
  if (nplanes != 3)
    return pave_internal_fail(PAVE_E_UNSUPPORTED, "conv7x7s2_nchw_split: fp16 operands need W %% 4 == 0 (or a repitched image) and a 16-byte aligned image");
  const ConvGeom g{H, W, 3, Ho, Wo, 2};
  return launch_gemm<2, 1, false, 3, false, 2, true>(x, static_cast<const uint16_t*>(w_planes), bias,
                                                     nullptr, y, M, 192, Cout, relu, nullptr,
                                                     reinterpret_cast<hipStream_t>(stream), g);
}

}  // extern "C"
