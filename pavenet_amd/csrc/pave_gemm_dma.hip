// libpave_hip.so -- the 3-plane split GEMM (see pave_gemm_split.hip for the arithmetic), second
// generation: EVERY global -> LDS byte moves by LDS-DMA (global_load_lds_dwordx4: no VGPR round
// trip, no ds_write), the activation tile lands in LDS as raw fp32 and is split into its three
// bf16 planes by the wave that consumes it, at operand-fetch time.
//
// Why (timing ablations of the first-generation kernel, one binary, FFN2 shape): removing the
// global -> VGPR loads of the loop bought 20 %, removing the ds_write staging 14 %, the VALU split
// itself only 3 %.  The loads and the LDS stores were the cost, not the split.
//
// Block = 4 WN waves: wave (wm, wn) owns rows [32 wm, 32 wm + 32) x columns [128 wn, 128 wn + 32 TN)
// (TN accumulator tiles); WN = 1 (256 threads, two blocks per CU) or WN = 2 (512 threads, the block
// spans N = 256 whole rows: LayerNorm in the epilogue).  K slab = 16 = one
// v_mfma_f32_32x32x16_bf16 k-step.  LDS is a ring of 3 stages; a stage holds
//     A raw  [128 rows][4 chunks of 16 B]       chunk c of row r at slot c ^ ((r >> 2) & 3)
//     W      [3 planes][BN rows][2 halves]      half h of row r at slot h ^ ((r >> 3) & 1)
// The slot permutations make the ds_read_b128 of the MFMA operands (lane -> row l & 31, k half
// l >> 5) bank-conflict free without padding (a DMA instruction writes 1 KiB of CONSECUTIVE LDS
// bytes; the permutation is applied on the per-lane SOURCE address).
// Pipeline per slab s: wait until the DMA of slab s + 1 has landed (vmcnt counts in issue order:
// only the DMAs of slab s + 2 may still be in flight) and all LDS reads of this wave have returned,
// barrier, issue the DMA of slab s + 3 into the stage slab s occupied, read the operand fragments
// of slab s + 1, run the 6 TN MFMAs of slab s from registers while the split of slab s + 1 runs
// on the VALU.  One barrier per slab, two slabs of load latency covered.  The residual rows of the
// epilogue are fetched before the last slab's MFMAs.
// (A persistent form -- 2 blocks per CU walking an XCD-local tile list, the next tile's first three
// slabs issued before the epilogue -- was built and measured 7-10 % SLOWER on every shape: the tile
// loop costs 40 more VGPRs (spills around the epilogue) and, vmcnt being one in-order counter, the
// next tile's first wait also drains the epilogue's stores.  One tile per block stays.
// A v_mfma_f32_16x16x32_bf16 form (32-k stages, ring of 2, W fragments read quarter by quarter,
// one barrier per 32 k) was built too: a timing-only swap of the MFMA shape had shown +4-5 %
// (higher clock at equal pipe cycles), the real kernel ran equal at K >= 512 and 13 % slower at
// K = 256 (224 live VGPRs, spills in the tail steps, a two-stage prologue).  Dropped.)
// The DMA instructions are inline assembly (the compiler's wait-count pass would otherwise fence
// every LDS read behind the youngest DMA); every wait on them is an explicit vmcnt here.  LDS
// reads stay ordinary loads, so their lgkmcnt waits are the compiler's.  Results are bit-identical
// to the first-generation kernel (same products, same order per accumulator).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "pave_hip.h"
#include "pave_internal.h"
#include "pave_enc_math.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int QBM = 128;   // rows per block
constexpr int QNS = 3;     // ring stages
constexpr int QCST = 36;   // epilogue chunk (32 rows x 32 columns per wave) row stride in floats

struct QConv {   // convolution forms: NHWC geometry
  int H, W, Cin, Ho, Wo, stride;
  unsigned rcp;  // 3x3 form: ceil(2^32 / (Cin / 16)) (0 when Cin == 16): slab -> tap by multiply-high
};
struct QOut {    // epilogue options: second output from column nsplit on, row-periodic residual,
  float* out2;   // real output width n_real <= N (N = the width of the zero-padded weight planes:
  int nsplit;    // out / bias / residual have n_real columns, the columns beyond are not stored)
  int res_rows;
  int n_real;
  int ks_slabs;  // split-K (grid.y parts): K slabs per part (even, >= 4), 0 = whole K.  Part y
};               // writes its raw partial sums to out + y M n_real (bias / residual / relu unset)
struct QLn {     // LayerNorm over the output row (LNORM forms, N == block width)
  const float* gamma;
  const float* beta;
  float eps;
};

struct QEpi {    // encoder projection epilogues (EPI 1 / 2): what the sampler would otherwise compute
  const float* ref;              // [M, 4, 2] reference points of the rows (normalised x, y per level)
  float fW[4], fH[4], rW[4], rH[4];   // level sizes and their reciprocals (as the sampler's launcher)
};

__device__ __forceinline__ unsigned hi16(float x) { return __float_as_uint(x) & 0xffff0000u; }
__device__ __forceinline__ unsigned pack_hi(unsigned lo, unsigned hi) {
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_rne(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_rne_f16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

// 8 consecutive fp32 (two 16-byte chunks) -> the three bf16 planes of an MFMA A operand
// (4 dwords each): truncation, truncation, exact remainder.
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, u32x4 (&pl)[3]) {
  float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    unsigned h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      h[i] = hi16(x[i]);
      x[i] -= __uint_as_float(h[i]);
    }
    pl[t] = u32x4{pack_hi(h[0], h[1]), pack_hi(h[2], h[3]), pack_hi(h[4], h[5]), pack_hi(h[6], h[7])};
  }
  pl[2] = u32x4{pack_rne(x[0], x[1]), pack_rne(x[2], x[3]), pack_rne(x[4], x[5]), pack_rne(x[6], x[7])};
}

// one LDS-DMA instruction: 64 lanes x 16 B from sbase + voff (per lane) to LDS bytes
// [lds_addr, lds_addr + 1024) in lane order
__device__ __forceinline__ void dma16(unsigned voff, const void* sbase, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}
// the same with a full 64-bit per-lane address (3x3 form: a lane may point at the zero chunk)
__device__ __forceinline__ void dma16_flat(const void* vaddr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(vaddr), "s"(lds_addr)
               : "memory");
}
// the same through a buffer resource (base, size): a lane whose byte offset lies beyond the size
// reads ZEROS -- the zero padding of the 3x3 form costs one select per lane instead of a second
// address
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_buf(unsigned voff, i32x4 rsrc, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
               :
               : "v"(voff), "s"(rsrc), "s"(lds_addr)
               : "memory");
}
// every DMA older than the NV youngest vector-memory operations of this wave has landed, every LDS
// read of this wave has returned; then the workgroup barrier
#define PAVE_QWAIT(NV) \
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NV) : "memory")
// LDS-only barrier of the epilogue (no vector-memory wait)
#define PAVE_QBAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// sum over the 32 lanes of a half wave, result in every lane: DPP butterflies inside the rows of 16
// (quad xor 1, xor 2, half-row mirror, row mirror), then one exchange between the two rows
__device__ __forceinline__ float half32_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v + __shfl_xor(v, 16, 64);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
// torch.sigmoid's expression (ATen: 1 / (1 + exp(-x))), as the reference-point kernels use it
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// max(x, 0) as the single v_max_f32 fmaxf ends in (the compiler puts a canonicalising v_max in front of it)
__device__ __forceinline__ float relu_max(float x) {
  float y;
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}

__device__ const uint4 g_zero_chunk[4] = {};   // source of out-of-image taps (zero padding)

// In-kernel clock of the split GEMM class (MI355X_MICROARCH.md, "DVFS give-back" item 6): delta s_memtime (shader
// clock) / delta s_memrealtime (100 MHz) around a workgroup's whole body, summed over every 64th workgroup of a
// launch -- the -DPAVE_DIAG build only (bench.py's clock pass, tools/): no stamp executes in the shipped library,
// and the values go to counters of their own that no kernel reads.
#ifdef PAVE_DIAG
__device__ unsigned long long g_clock_acc[8][2];   // [kernel kind][shader ticks, 100 MHz ticks]
struct ClockStamp {
  unsigned long long t0, r0;
  bool on;
  int kind;
  __device__ __forceinline__ explicit ClockStamp(const int k) : kind(k) {
    on = (blockIdx.x & 63) == 0 && blockIdx.y == 0;
    if (on) {
      t0 = __builtin_amdgcn_s_memtime();
      r0 = __builtin_amdgcn_s_memrealtime();
    }
  }
  __device__ __forceinline__ void end() const {
    if (on && threadIdx.x == 0) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      atomicAdd(&g_clock_acc[kind][0], t1 - t0);
      atomicAdd(&g_clock_acc[kind][1], r1 - r0);
    }
  }
};
// Phase stagger probe (tools/stagger_probe.py): the two workgroups that share a CU start a launch together and,
// tiles taking equal time, stay in step -- main loops (matrix pipe) together, epilogues (HBM) together.  With
// g_diag_stagger = n > 0 the first-round workgroups (blockIdx.x < 512) that sit in an ODD slot of their CU sleep
// n x 8 128 shader clocks before they start, so that the pair runs out of phase.  Slot = bit 0 of the wave id on the
// SIMD (mode 0), of the CU's thread-group id (mode 1), or of blockIdx.x (mode 2: a control -- neighbours in
// blockIdx.x sit on different XCDs); g_diag_stagger = mode * 1000 + n.  g_diag_hwid[b] = HW_ID of wave 0 of block b.
__device__ int g_diag_stagger = 0;
__device__ unsigned g_diag_hwid[1024];
__device__ __forceinline__ void diag_stagger() {
  const int v = g_diag_stagger;
  if (v <= 0 || blockIdx.y != 0 || blockIdx.x >= 1024) return;
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) g_diag_hwid[blockIdx.x] = hw;
  if (blockIdx.x >= 512) return;
  const int mode = v / 1000, n = v % 1000;
  const unsigned bit = mode == 0 ? hw & 1u : (mode == 1 ? (hw >> 16) & 1u : blockIdx.x & 1u);
  if (bit)
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}
#define PAVE_CLOCK_BEGIN(kind) diag_stagger(); const ClockStamp pave_cs(kind)
#define PAVE_CLOCK_END() pave_cs.end()
// timing-only traffic probes (tools/ffn_traffic_probe.py; the results are WRONG on purpose): g_diag_stagger = -1: the
// epilogue's stores are dropped (out-of-range buffer offsets); -2: every row tile of a plain-row launch reads the A
// rows of tile 0 (they stay in L2); -3: both
#define PAVE_PROBE_DROP_STORES() (g_diag_stagger == -1 || g_diag_stagger == -3 || g_diag_stagger == -5)
#define PAVE_PROBE_CACHED_A() (g_diag_stagger == -2 || g_diag_stagger == -3)
// -4: the layer1 chain without its workgroup fences / barriers between the bodies (a body then reads rows its
// neighbours may not have written yet: wrong values, same instruction stream otherwise); -5: -4 and -1
#define PAVE_PROBE_NO_CHAIN_SYNC() (g_diag_stagger == -4 || g_diag_stagger == -5)
// -6: the epilogue's 16-byte stores carry the non-temporal hint (aux bit 1 = nt on gfx942 / gfx950): same values
#define PAVE_PROBE_NT_STORES() (g_diag_stagger == -6)
// -7: the A operand's three planes are the raw fp32 bits (no vector arithmetic for the split; wrong values)
#define PAVE_PROBE_FREE_SPLIT() (g_diag_stagger == -7)
#else
#define PAVE_CLOCK_BEGIN(kind)
#define PAVE_CLOCK_END()
#define PAVE_PROBE_DROP_STORES() false
#define PAVE_PROBE_CACHED_A() false
#define PAVE_PROBE_NO_CHAIN_SYNC() false
#define PAVE_PROBE_NT_STORES() false
#define PAVE_PROBE_FREE_SPLIT() false
#endif

// KIND: 0 = plain rows A [M, K] (row stride g.H floats if g.H > 0; GROUPED when g.W > 0: the N axis
//           is cut into groups of g.W columns, group i multiplies columns [i K, (i + 1) K) of A by its
//           own [g.W, K] weight -- T per-frame Linears of one layer as ONE launch);  1 = 3x3 / pad 1 implicit GEMM over (tap, cin) of an NHWC map
//           smaller than 4 GiB (buffer-addressed: out-of-image taps are out-of-range lanes);  2 = the same for larger maps (64-bit lane addresses, a zero chunk for the padding);
//       3 = rows = strided pixels of an NHWC map (1x1 convolution with a stride);
//       4 = rows [A | A2]: the first g.Cin columns of the K axis come from A [M, g.Cin], the rest
//           from A2 [M, K - g.Cin] (two GEMMs sharing one accumulator: conv3 + downsample)
// WIDE (TN = 8, WN = 1): the wave owns 32 rows x 256 columns -- one split of the A rows and one A
// DMA feed 48 MFMAs instead of 24.  The 128 accumulator registers leave room for only a QUARTER
// of a slab's W fragments (2 column tiles x 3 planes) twice over, so the fragments are read
// quarter by quarter just ahead of their MFMAs, the ring has 2 stages (2 x 32 KiB: two blocks per
// CU) and the barrier sits between the third and the fourth quarter: by then every fragment of
// the slab is in registers, its stage takes the DMA of slab s + 2 (one slab = 48 MFMAs per wave of
// latency cover), and the fourth quarter's MFMAs run over the reads and the split of slab s + 1.
// Per accumulator the products keep the order of the narrow form: results are bit-identical.
// EPI (epilogue transform of the stored values, plain row forms only; pave_enc_math.h):
//   1 = the columns are sampling offsets [head][level][point][x, y] of the encoder's deformable
//       attention: stored as level pixel coordinates (ref + off / size) * size - 0.5;
//   2 = the columns are attention logits [head][16]: stored as their softmax over each 16
// RM (narrow form, WN = 1): row tiles per wave.  RM = 2: the block is 256 rows, wave wm owns rows
// [64 wm, 64 wm + 64) = two 32-row tiles that share every W fragment -- 12 TN MFMAs per slab behind the same
// wait, barrier and W fragment reads as the 6 TN of RM = 1 (the 64-column-tile forms, TN = 2, are bound by
// that per-slab overhead, not by the matrix pipe).  Per accumulator the products keep their order:
// bit-identical to RM = 1.
// PL: operand planes.  3 = the exact bf16 split (six products per tile);  1 = fp16 operands ("fp16 MFMA
// projections", BASELINE configs[4]): ONE plane of fp16 weights in the same slab-major layout, the raw fp32
// activation rows converted (round to nearest even) where the 3-plane form splits them, one
// v_mfma_f32_32x32x16_f16 per tile and slab -- every row source, tile form and epilogue of this file as it is.
// HT ("half tail", TN = 2, 33 .. 48 real output columns in 64-row weight planes: HRNet-w48's 48-channel branch):
// the second column tile is only 16 columns wide, so it is not run as zero-padded 32x32x16 products (a quarter of
// the launch's MFMA cycles multiplying zeros) but as v_mfma_f32_16x16x32_bf16 over PAIRS of K slabs: its A
// operands (16 rows x 32 k) are built from the two slabs' 32x32x16 operands with v_permlane32_swap +
// v_permlane16_swap in place (once the main tile's MFMAs of the second slab are issued nobody else needs them;
// tools/microbench/permlane_swap.hip), its B operand is one 16-byte piece per lane, lanes 0-31 from the first
// slab's stage, 32-63 from the second's.  6 + 3 MFMA-equivalents per slab instead of 12.  The tail columns
// accumulate 32 k per instruction: equal to the padded form up to fp32 summation order, not bit for bit.
// IO (fp16 mode, plain rows): bit 0 = the A rows are fp16 in memory (a tensor that is only ever a GEMM operand --
// the FFN hidden -- kept as what the MFMA consumes: half the bytes, no conversion, the same values), bit 1 = the
// output is stored as fp16 (bias + ReLU in fp32 first).
// (A form whose A operand arrives ALREADY SPLIT -- the map as [pixel][3 planes][Cin] bf16 written by the producer,
// its stage holding operand-ready planes in the W planes' layout, no vector arithmetic in the loop -- was built for
// the 3x3 form in round 6, bit-identical, and measured 16 - 30 % SLOWER than splitting at operand fetch (HRNet-w48's
// 48 -> 48 at 28 x 200 x 336: 645 -> 749 us, 96 -> 96: 438 -> 530, 64 -> 64: 820 -> 1 071): 12 instead of 8 A DMA
// instructions per slab, 54 KB of LDS (two blocks per CU instead of three at 64-column tiles).  What bounds these
// launches is the A-side data movement, not the VALU split.  Removed; commit 1c032d8 has it, docs/HISTORY.md the
// numbers.)
template <int TN, int WN, int KIND, bool ABIAS, bool LNORM, bool WIDE = false, int EPI = 0, int RM = 1, int PL = 3,
          bool HT = false, int IO = 0>
__device__ __forceinline__ void gemm_q_body(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const int relu,
    const float* __restrict__ a_bias, const QConv g, const QOut os, const QLn ln,
    const float* __restrict__ A2 = nullptr, const int tile_m0 = -1, const int tile_n0 = 0,
    const QEpi* epi = nullptr) {
  // (tile_m0 >= 0: the caller names the tile -- kernels that run several bodies per block)
  constexpr int NWAVE = 4 * WN;
  constexpr int BN = WN * TN * 32;           // block width
  constexpr int BM = QBM * RM;               // rows per block
  constexpr bool AH = (IO & 1) != 0, OH = (IO & 2) != 0;
  static_assert(!AH || (PL == 1 && KIND == 0 && !ABIAS && RM == 1), "fp16 A rows: fp16 operand mode, plain rows");
  static_assert(!OH || (PL == 1 && !LNORM && EPI == 0 && !HT && RM == 1), "fp16 output: plain epilogue");
  constexpr int A_STAGE = AH ? BM * 32 : BM * 64;   // raw fp32: BM rows x 64 B (fp16 rows: 32 B)
  static_assert(RM == 1 || (RM == 2 && WN == 1 && !WIDE && !LNORM && EPI == 0), "two row tiles per wave: narrow form");
  static_assert(PL == 3 || PL == 1, "operand planes: 3 (bf16 split) or 1 (fp16)");
  constexpr int W_STAGE = PL * BN * 32;      // PL planes x BN rows x 32 B
  constexpr int STAGE = A_STAGE + W_STAGE;
  constexpr int NA = A_STAGE / 1024;         // DMA instructions per slab: A (8)
  constexpr int NWI = W_STAGE / 1024;        //                            W (6 | 12 | 24)
  constexpr int NDI = NA + NWI;
  constexpr int QMAX = (NDI + NWAVE - 1) / NWAVE;   // per wave and slab: at most / at least
  constexpr int QMIN = NDI / NWAVE;
  constexpr int QA = (NA + NWAVE - 1) / NWAVE;      // A instructions per wave: 2 (WN 1) | 1 (WN 2)
  constexpr int NS = WIDE ? 2 : QNS;                // ring stages
  constexpr int EPI_OFF = 0;                        // per-wave epilogue chunks reuse the ring
  constexpr int STAT_OFF = NS * STAGE;              // LayerNorm row statistics (WN > 1)
  constexpr int TQ = WIDE ? 2 : (HT ? 1 : TN);      // column tiles per W fragment set (HT: the full tile only)
  static_assert(!HT || (TN == 2 && WN == 1 && RM == 1 && !WIDE && !LNORM && EPI == 0 && PL == 3),
                "half-tail form: 64-column planes, one row tile per wave, 3 planes");
  static_assert(!WIDE || (TN == 8 && WN == 1 && !ABIAS), "wide form: 32 x 256 per wave");
  constexpr int ABOFF = STAT_OFF + (LNORM ? 2 * QBM * WN * 4 : 0);   // a_bias vector
  static_assert(NA % NWAVE == 0, "A DMA instructions divide evenly over the waves");
  static_assert(!LNORM || KIND == 0, "LayerNorm epilogue: plain row GEMM only");
  static_assert(KIND >= 0 && KIND <= 4, "row source");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, kh = lane >> 5;
  const int ntiles = N / BN;
  const int ttot = ((M + BM - 1) / BM) * ntiles;
  // XCD-aware bijective tile order (the hardware deals blocks round-robin over the 8 XCDs; each
  // XCD gets a contiguous run of logical tiles, column tile fastest, so the column tiles of one
  // row tile run side by side on ONE XCD and its A rows cross HBM -> L2 once)
  const int per = ttot >> 3, rem = ttot & 7;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // split-K: this block's part of the K axis
  const int s0 = os.ks_slabs > 0 ? (int)blockIdx.y * os.ks_slabs : 0;
  const int nslabs = os.ks_slabs > 0 ? min(os.ks_slabs, K / 16 - s0) : K / 16;
  const long long w_slab = (long long)PL * N * 32;  // bytes per K slab of the weight planes
  const unsigned char* const w_base = reinterpret_cast<const unsigned char*>(Wp);

  // ---- DMA roles of this wave: instructions d = wave + NWAVE q;  q < QA: A rows, else W rows
  int m0 = 0, n0 = 0;
  unsigned a_voff[QA];       // KIND 0 / 3 / 4: byte offset of the lane's chunk from the slab base
  unsigned a2_voff[QA];      // KIND 4: the same inside A2
  const unsigned char* a2_base = reinterpret_cast<const unsigned char*>(A2);
  int a_iy0[QA], a_ix0[QA];  // KIND 2: top-left input pixel of the lane's output pixel
  const float* a_img[QA];    // KIND 2: image base + chunk offset
  unsigned a_mask[QA];       // KIND 1: bit t = tap t of the lane's pixel lies inside the image
  i32x4 a_rsrc;              // KIND 1: buffer resource of the whole map
  if (KIND == 1) {
    const unsigned long long ab = reinterpret_cast<unsigned long long>(A);
    a_rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ab);
    a_rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(ab >> 32) & 0xffff);
    // bytes of the whole map (the launcher takes this form below 4 GiB only)
    a_rsrc.z = __builtin_amdgcn_readfirstlane(
        (int)((unsigned)(M / (g.Ho * g.Wo)) * (unsigned)g.H * (unsigned)g.W * (unsigned)g.Cin * 4u));
    a_rsrc.w = 0x00020000;
  }
  unsigned w_voff[QMAX - QA];
  const unsigned char* a_base = reinterpret_cast<const unsigned char*>(A);

  auto setup_tile = [&](const int t) {
    const int xcd = t & 7, idx = t >> 3;
    const int lb = xcd * per + (xcd < rem ? xcd : rem) + idx;
    m0 = (lb / ntiles) * BM;
    n0 = (lb % ntiles) * BN;
    if (tile_m0 >= 0) m0 = tile_m0, n0 = tile_n0;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int d = wave + NWAVE * q;
      // (fp16 rows: an instruction covers 32 rows x two 16-byte halves, half h of row r at slot h ^ ((r >> 3) & 1))
      const int r = AH ? d * 32 + (lane >> 1) : d * 16 + (lane >> 2);
      const int c = AH ? (lane & 1) ^ ((r >> 3) & 1) : (lane & 3) ^ ((r >> 2) & 3);
      long long gm = (long long)m0 + r;
      if (gm >= M) gm = M - 1;   // rows past M: stand-in data, never stored
      if (KIND == 0 && AH) {
        a_voff[q] = (unsigned)(((gm - m0) * (g.H > 0 ? g.H : K) + c * 8) * 2);
      } else if (KIND == 0) {
        a_voff[q] = (unsigned)(((gm - m0) * (g.H > 0 ? g.H : K) + c * 4) * 4);
      } else if (KIND == 4) {
        a_voff[q] = (unsigned)(((gm - m0) * g.Cin + c * 4) * 4);
        a2_voff[q] = (unsigned)(((gm - m0) * (K - g.Cin) + c * 4) * 4);
      } else {
        const unsigned ur = (unsigned)gm, gy = ur / (unsigned)g.Wo;
        const int ox = (int)(ur - gy * (unsigned)g.Wo);
        const int n = (int)(gy / (unsigned)g.Ho);
        const int oy = (int)(gy - (unsigned)n * (unsigned)g.Ho);
        if (KIND == 3) {
          a_voff[q] = (unsigned)(((((long long)n * g.H + oy * g.stride) * g.W + ox * g.stride) * K + c * 4) * 4);
        } else if (KIND == 1) {
          const int iy0 = oy * g.stride - 1, ix0 = ox * g.stride - 1;
          // byte offset of tap (0, 0), chunk c (mod 2^32: may lie before the map for a padded tap)
          a_voff[q] = (unsigned)(((((long long)n * g.H + iy0) * g.W + ix0) * g.Cin + c * 4) * 4);
          unsigned mk = 0;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int iy = iy0 + t / 3, ix = ix0 + t % 3;
            mk |= (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) ? (1u << t) : 0u;
          }
          a_mask[q] = mk;
        } else {
          a_iy0[q] = oy * g.stride - 1;
          a_ix0[q] = ox * g.stride - 1;
          a_img[q] = A + (long long)n * g.H * g.W * g.Cin + c * 4;
        }
      }
    }
    if (KIND == 0 && AH)
      a_base = reinterpret_cast<const unsigned char*>(A) +
               ((long long)m0 * (g.H > 0 ? g.H : K) + (g.W > 0 ? (long long)(n0 / g.W) * K : 0)) * 2;
    else if (KIND == 0)
      a_base = reinterpret_cast<const unsigned char*>(A + (long long)(PAVE_PROBE_CACHED_A() ? 0 : m0) * (g.H > 0 ? g.H : K) +
                                                      (g.W > 0 ? (long long)(n0 / g.W) * K : 0));
    if (KIND == 4) {
      a_base = reinterpret_cast<const unsigned char*>(A + (long long)m0 * g.Cin);
      a2_base = reinterpret_cast<const unsigned char*>(A2 + (long long)m0 * (K - g.Cin));
    }
#pragma unroll
    for (int q = QA; q < QMAX; ++q) {
      const int j = wave + NWAVE * (q - QA);         // W instruction: plane j / (NWI/PL), 32 rows
      const int p = j / (NWI / PL), row = (j % (NWI / PL)) * 32 + (lane >> 1);
      const int h = (lane & 1) ^ ((row >> 3) & 1);
      w_voff[q - QA] = (unsigned)((((long long)p * N + n0 + row) * 32 + h * 16));
    }
  };
  auto issue = [&](const int rel, const int stage) {
    const int slab = rel + s0;
    const unsigned sl = lds0 + stage * STAGE;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const unsigned dst = sl + (wave + NWAVE * q) * 1024;
      if (KIND == 1) {
        // (scalar) a slab lies inside one tap: tap = slab / (Cin / 16) by reciprocal multiply
        const int tap = g.rcp ? (int)__umulhi((unsigned)slab, g.rcp) : slab;
        const int c0 = (slab - tap * (g.Cin >> 4)) * 16;
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;   // (tap < 16)
        const unsigned toff = (unsigned)(((ky * g.W + kx) * g.Cin + c0) * 4);
        // (tap >= 9: the zero slab that pads K = 9 Cin to a multiple of 32 -- no mask bit)
        const bool ok = (a_mask[q] >> tap) & 1u;
        dma16_buf(ok ? a_voff[q] + toff : 0xffffff00u, a_rsrc, dst);   // (beyond any map < 4 GiB - 64 KiB)
      } else if (KIND == 2) {
        const int k0 = slab * 16;
        const int tap = k0 / g.Cin, c0 = k0 - tap * g.Cin;   // (scalar) a slab lies inside one tap
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = a_iy0[q] + ky, ix = a_ix0[q] + kx;
        // (tap >= 9: the zero slab that pads K = 9 Cin to a multiple of 32)
        const bool ok = tap < 9 && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        const float* src = ok ? a_img[q] + ((long long)iy * g.W + ix) * g.Cin + c0
                              : reinterpret_cast<const float*>(g_zero_chunk);
        dma16_flat(src, dst);
      } else if (KIND == 4) {
        const int s1 = g.Cin >> 4;   // (scalar) slabs of the first source
        if (slab < s1) dma16(a_voff[q], a_base + (long long)slab * 64, dst);
        else dma16(a2_voff[q], a2_base + (long long)(slab - s1) * 64, dst);
      } else {
        dma16(a_voff[q], a_base + (long long)slab * (AH ? 32 : 64), dst);
      }
    }
#pragma unroll
    for (int q = QA; q < QMAX; ++q) {
      const int j = wave + NWAVE * (q - QA);
      if (NWI % NWAVE == 0 || j < NWI)   // (wave-uniform; always true unless BN = 64)
        dma16(w_voff[q - QA], w_base + slab * w_slab, sl + A_STAGE + j * 1024);
    }
  };

  // ---- operand fragment addresses (bytes inside a stage)
  const int sw = (lr >> 2) & 3;
  const int a_rd0 = AH ? (wm * 32 + lr) * 32 + ((kh ^ ((lr >> 3) & 1)) * 16)
                       : (wm * 32 * RM + lr) * 64 + (((2 * kh) ^ sw) * 16);   // (+ 2 KiB per further row tile)
  const int a_rd1 = a_rd0 ^ 16;
  const int w_rd = A_STAGE + (wn * TN * 32 + lr) * 32 + ((kh ^ ((lr >> 3) & 1)) * 16);

  f32x16 acc[RM * TN];     // [row tile][column tile]
  f32x4 raw[RM][2];        // the lane's 8 fp32 of the next slab, per row tile
  u32x4 apl[2][RM][PL];    // A planes: [set][row tile][plane]
  u32x4 wf[2][PL][TQ];     // W fragments: [set][plane][column tile (of the quarter, WIDE)]
  const float* const ab_lds = reinterpret_cast<const float*>(smem + ABOFF);

  auto read_raw = [&](const int stage) {
    const unsigned char* st = smem + stage * STAGE;
#pragma unroll
    for (int rt = 0; rt < RM; ++rt) {
      raw[rt][0] = *reinterpret_cast<const f32x4*>(st + a_rd0 + rt * 2048);
      if constexpr (!AH) raw[rt][1] = *reinterpret_cast<const f32x4*>(st + a_rd1 + rt * 2048);
    }
  };
  // column tiles [TQ quarter, TQ quarter + TQ) of the stage's W planes
  auto read_wq = [&](const int stage, const int quarter, const int set) {
    const unsigned char* st = smem + stage * STAGE;
#pragma unroll
    for (int p = 0; p < PL; ++p)
#pragma unroll
      for (int j = 0; j < TQ; ++j)
        wf[set][p][j] =
            *reinterpret_cast<const u32x4*>(st + w_rd + (p * BN + (quarter * TQ + j) * 32) * 32);
  };
  // HT: the 16-column tail's B pieces -- row 32 + (lane & 15) of every plane, k half (lane >> 4) & 1 -- of
  // the slab in `stage` (every lane reads; which half of the wave keeps them is the caller's business)
  u32x4 wtA[PL], wtB[PL], wt[PL];   // first slab of the pair | second | merged operand
  f32x4 acct[2];                    // tail accumulators: rows 0-15, 16-31 (16x16 layout)
  const int w_rdt = A_STAGE + (32 + (lane & 15)) * 32 + (((((lane >> 4) & 1)) ^ ((lane >> 3) & 1)) * 16);
  auto read_tail = [&](const int stage, u32x4 (&dst)[PL]) {
    const unsigned char* st = smem + stage * STAGE;
#pragma unroll
    for (int p = 0; p < PL; ++p) dst[p] = *reinterpret_cast<const u32x4*>(st + w_rdt + p * BN * 32);
  };
  auto read_frags = [&](const int stage, const int set) {
    read_raw(stage);
    read_wq(stage, 0, set);
    if constexpr (HT) {
      if (set == 0) read_tail(stage, wtA);   // an even slab: the first of its pair
      else read_tail(stage, wtB);            // the second: lanes 32-63 carry its k blocks
    }
  };
  // (after the first MFMA group of the step: the reads have landed under it)
  auto merge_tail = [&]() {
    if constexpr (HT) {
#pragma unroll
      for (int p = 0; p < PL; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) wt[p][i] = lane < 32 ? wtA[p][i] : wtB[p][i];
    }
  };
  // HT, after the main tile's MFMAs of the pair's second slab: the tail tile's 12 MFMAs of the pair
  auto tail_pair = [&]() {
    if constexpr (HT) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int p = 0; p < PL; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // [X0 X1 X2 X3], [Y0 Y1 Y2 Y3] (rows of 16 lanes: k half x row half) -> [X0 X2 Y0 Y2], [X1 X3 Y1 Y3]
          u32x2 r = __builtin_amdgcn_permlane32_swap(apl[0][0][p][i], apl[1][0][p][i], false, false);
          r = __builtin_amdgcn_permlane16_swap(r.x, r.y, false, false);
          apl[0][0][p][i] = r.x, apl[1][0][p][i] = r.y;
        }
#pragma unroll
      for (int o = 2; o >= 0; --o)
#pragma unroll
        for (int pa = 0; pa <= o; ++pa)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
            acct[hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, apl[hh][0][pa]), __builtin_bit_cast(bf16x8, wt[o - pa]), acct[hh], 0, 0, 0);
    }
  };
  auto split_raw = [&](const int slab, const int set) {
#pragma unroll
   for (int rt = 0; rt < RM; ++rt) {
    f32x4 lo = raw[rt][0], hi = raw[rt][1];
    if (ABIAS) {   // A' = relu(A + a_bias[k]) (the previous BatchNorm + ReLU), a_bias staged in LDS
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(ab_lds + slab * 16 + kh * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(ab_lds + slab * 16 + kh * 8 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lo[i] = fmaxf(lo[i] + b0[i], 0.f);
        hi[i] = fmaxf(hi[i] + b1[i], 0.f);
      }
    }
    if constexpr (AH) {      // the lane's 8 halves ARE the operand
      apl[set][rt][0] = __builtin_bit_cast(u32x4, lo);
    } else if constexpr (PL == 3) {
      if (PAVE_PROBE_FREE_SPLIT()) {
        apl[set][rt][0] = __builtin_bit_cast(u32x4, lo);
        apl[set][rt][1] = __builtin_bit_cast(u32x4, hi);
        apl[set][rt][2] = __builtin_bit_cast(u32x4, lo);
      } else
      split8(lo, hi, apl[set][rt]);
    } else {
      apl[set][rt][0] = u32x4{pack_rne_f16(lo.x, lo.y), pack_rne_f16(lo.z, lo.w), pack_rne_f16(hi.x, hi.y),
                              pack_rne_f16(hi.z, hi.w)};
    }
   }
  };
  // the products of order o = pa + pb (o = 2, 1, 0: smallest terms first), column tiles innermost
  auto mma = [&](const int set, const int o) {
    if constexpr (PL == 1) {   // fp16 operands: the slab's one product per tile (scheduled where o = 0 sits)
      if (o != 0) return;
#pragma unroll
      for (int rt = 0; rt < RM; ++rt)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
          acc[rt * TN + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
              __builtin_bit_cast(f16x8, apl[set][rt][0]), __builtin_bit_cast(f16x8, wf[set][0][j]),
              acc[rt * TN + j], 0, 0, 0);
    } else {
#pragma unroll
    for (int pa = 0; pa <= o; ++pa)
#pragma unroll
      for (int rt = 0; rt < RM; ++rt)
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
          acc[rt * TN + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, apl[set][rt][pa]), __builtin_bit_cast(bf16x8, wf[set][o - pa][j]),
              acc[rt * TN + j], 0, 0, 0);
        }
    }
  };
  // WIDE: all six products of one quarter (A planes of set aset, W fragments of set wset)
  auto mma_q = [&](const int aset, const int wset, const int quarter) {
    if constexpr (PL == 1) {
#pragma unroll
      for (int j = 0; j < TQ; ++j)
        acc[quarter * TQ + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            __builtin_bit_cast(f16x8, apl[aset][0][0]), __builtin_bit_cast(f16x8, wf[wset][0][j]),
            acc[quarter * TQ + j], 0, 0, 0);
    } else
#pragma unroll
    for (int o = 2; o >= 0; --o)
#pragma unroll
      for (int pa = 0; pa <= o; ++pa)
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
          acc[quarter * TQ + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, apl[aset][0][pa]),
              __builtin_bit_cast(bf16x8, wf[wset][o - pa][j]), acc[quarter * TQ + j], 0, 0, 0);
        }
  };

  // ---- epilogue state
  constexpr int NPS = 4;                 // passes per accumulator tile: 8 rows x 8 float4 each
  const int erow = lane >> 3, ec4 = lane & 7;
  constexpr int NPR = NPS * RM;          // 8-row passes over the wave's rows
  constexpr int NT = RM * TN;            // accumulator tiles of the wave: t = row tile * TN + column tile
  constexpr int RB = WIDE ? 4 : TN;      // residual tiles in registers (WIDE, RM > 1: rotating)
  float4 resv[RB][NPS];
  // Every global access of the epilogue goes through a buffer resource over THIS TILE's rows (output,
  // full residual) or over the whole row-periodic table: one 32-bit lane offset per pass, the column
  // tile as a scalar offset, rows past M and columns past n_real get the offset kOut (>= any range
  // below 2 GiB: the hardware drops the store / returns zeros) -- no 64-bit address per access and
  // no predication around it.
  constexpr unsigned kOut = 0x80000000u;
  using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
  unsigned rro[NPR];                     // residual byte offset of the lane's row per pass (+ its 16-byte column)
  __amdgpu_buffer_rsrc_t rrs;
  auto residual_rows = [&](const int em0) {
    // row-periodic table (row m adds residual[m % res_rows]): one modulo per lane, the passes
    // step the row by 8 with a wrap (res_rows >= 32: at most one wrap per step)
    const bool table = os.res_rows != 0;
    const int rows_left = min(BM, M - em0);
    rrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(residual) + (table ? 0ll : (long long)em0 * os.n_real), 0,
        (table ? os.res_rows : rows_left) * os.n_real * 4, 0x00020000);
    const int lrow0 = wm * 32 * RM + erow;
    const bool stepwise = os.res_rows >= 32 * RM;
    const unsigned rbase = table ? (unsigned)(em0 + lrow0) % (unsigned)os.res_rows : 0u;
#pragma unroll
    for (int ps = 0; ps < NPR; ++ps) {
      unsigned rr = (unsigned)(lrow0 + ps * 8);
      if (table) {
        if (stepwise) {
          rr = rbase + ps * 8;
          if (rr >= (unsigned)os.res_rows) rr -= (unsigned)os.res_rows;
        } else {
          rr = (unsigned)(em0 + lrow0 + ps * 8) % (unsigned)os.res_rows;
        }
      }
      rro[ps] = lrow0 + ps * 8 < rows_left ? rr * (unsigned)(os.n_real * 4) + ec4 * 16 : kOut;
    }
  };
  auto prefetch_residual_tile = [&](const int en0, const int t, const int buf) {
    const int rt = t / TN, j = t % TN;
    const int ncol0 = en0 + wn * TN * 32 + j * 32;   // wave-uniform
    const bool colok = WIDE || ncol0 + ec4 * 4 < os.n_real;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps)
      resv[buf][ps] = __builtin_bit_cast(
          float4, __builtin_amdgcn_raw_buffer_load_b128(rrs, colok ? rro[rt * NPS + ps] : kOut, ncol0 * 4, 0));
  };
  auto prefetch_residual = [&](const int em0, const int en0) {
    if (!residual) return;
    residual_rows(em0);
#pragma unroll
    for (int j = 0; j < RB; ++j) prefetch_residual_tile(en0, j, j);
  };

  if (ABIAS) {
    float* ab = reinterpret_cast<float*>(smem + ABOFF);
    for (int i = tid; i < K; i += 64 * NWAVE) ab[i] = a_bias[i];
  }
  setup_tile(blockIdx.x);
  {
    const int em0 = m0, en0 = n0;
    if constexpr (!WIDE) {
      // ---- three slabs in flight
      issue(0, 0);
      issue(1, 1);
      issue(2, 2);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      acct[0] = f32x4{0.f, 0.f, 0.f, 0.f}, acct[1] = acct[0];
      PAVE_QWAIT(2 * QMIN);                  // slab 0 has landed everywhere
      read_frags(0, 0);
      split_raw(0, 0);

      // ---- main loop, 6 slabs per trip (ring stage = slab % 3, register set = slab & 1).  FULL
      // steps (no slab-count tests: one basic block, so the split of slab s + 1 is scheduled
      // between the MFMAs of slab s) run while three more slabs follow; the guarded form runs up
      // to the last slab but one.  nslabs is even, so the last slab's operands sit in set 1.
#define PAVE_QSTEP(I, FULL)                                                    \
  if (FULL || s + I < nslabs - 1) {                                            \
    constexpr int cur = (I) & 1, nxt = cur ^ 1;                                \
    const int sl = s + I;                                                      \
    if (FULL || sl + 2 < nslabs) PAVE_QWAIT(QMIN); else PAVE_QWAIT(0);         \
    if (FULL || sl + 3 < nslabs) issue(sl + 3, (I) % 3);                       \
    read_frags(((I) + 1) % 3, nxt);                                            \
    __builtin_amdgcn_sched_barrier(0);                                         \
    mma(cur, 2);                                                               \
    if (HT && cur == 0) merge_tail();                                          \
    if (!(HT && cur == 1)) split_raw(sl + 1, nxt);                             \
    mma(cur, 1);                                                               \
    mma(cur, 0);                                                               \
    if (HT && cur == 1) {   /* the pair is complete: its tail tile, then the split that overwrites set 0 */ \
      tail_pair();                                                             \
      split_raw(sl + 1, nxt);                                                  \
    }                                                                          \
  }
      int s = 0;
      for (; s + 9 <= nslabs; s += 6) {
        PAVE_QSTEP(0, true)
        PAVE_QSTEP(1, true)
        PAVE_QSTEP(2, true)
        PAVE_QSTEP(3, true)
        PAVE_QSTEP(4, true)
        PAVE_QSTEP(5, true)
      }
      for (; s < nslabs - 1; s += 6) {
        PAVE_QSTEP(0, false)
        PAVE_QSTEP(1, false)
        PAVE_QSTEP(2, false)
        PAVE_QSTEP(3, false)
        PAVE_QSTEP(4, false)
        PAVE_QSTEP(5, false)
      }
      // ---- last slab: every wave has read the last stage (the epilogue reuses the ring); the
      // residual rows are fetched under the last MFMAs
      PAVE_QWAIT(0);
      prefetch_residual(em0, en0);
      mma(1, 2);
      mma(1, 1);
      mma(1, 0);
      tail_pair();
#undef PAVE_QSTEP
    } else {
      // ---- two slabs in flight
      issue(0, 0);
      issue(1, 1);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      PAVE_QWAIT(QMIN);                      // slab 0 has landed everywhere
      read_frags(0, 0);
      split_raw(0, 0);
      // ---- one slab: quarters 0..2 from the stage (stage = A plane set = slab & 1), then the
      // barrier (slab s + 1 landed, the stage free for slab s + 2), quarter 3 over the first
      // reads and the split of slab s + 1.  W sets alternate per quarter: 0 1 0 1.
#define PAVE_WQ012(cur)                                                        \
  read_wq(cur, 1, 1);                                                          \
  mma_q(cur, 0, 0);                                                            \
  read_wq(cur, 2, 0);                                                          \
  mma_q(cur, 1, 1);                                                            \
  read_wq(cur, 3, 1);                                                          \
  mma_q(cur, 0, 2);                                                            \
  PAVE_QWAIT(0);
#define PAVE_WSTEP(I)                                                          \
  {                                                                            \
    constexpr int cur = (I) & 1, nxt = cur ^ 1;                                \
    PAVE_WQ012(cur)                                                            \
    issue(s + I + 2, cur);                                                     \
    read_frags(nxt, 0);                                                        \
    __builtin_amdgcn_sched_barrier(0);                                         \
    mma_q(cur, 1, 3);                                                          \
    split_raw(s + I + 1, nxt);                                                 \
  }
      // (nslabs is even and >= 4: the loop leaves exactly the last two slabs, written out below
      // without slab-count tests)
      for (int s = 0; s + 4 <= nslabs; s += 2) {
        PAVE_WSTEP(0)
        PAVE_WSTEP(1)
      }
      PAVE_WQ012(0)
      read_frags(1, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma_q(0, 1, 3);
      split_raw(nslabs - 1, 1);
      PAVE_WQ012(1)   // (every wave has read the last stage: the epilogue reuses the ring)
      if constexpr (!LNORM) prefetch_residual(em0, en0);
      mma_q(1, 1, 3);
#undef PAVE_WQ012
#undef PAVE_WSTEP
    }

    // ---- epilogue: one 32 x 32 accumulator tile at a time through the wave's own LDS chunk,
    // float4 row segments (one full 128-byte line per row and pass)
    float* Cs = reinterpret_cast<float*>(smem + EPI_OFF) + wave * 32 * QCST;
    const bool seg2 = os.out2 != nullptr && en0 >= os.nsplit;
    float* const obase = seg2 ? os.out2
                              : out + (os.ks_slabs > 0 ? (long long)blockIdx.y * M * os.n_real : 0);
    const int ldo = os.out2 == nullptr ? os.n_real : (seg2 ? N - os.nsplit : os.nsplit);
    const int csh = seg2 ? os.nsplit : 0;
    // the tile's output rows behind a buffer resource (see resv above) that starts at the wave's first
    // column: rows past M fall outside it.  The column tile is an immediate offset and the scalar offset
    // stays 0: with a REGISTER there the compiler does not put the wait state between a 16-byte store and a
    // VALU write of its data registers (GCNHazardRecognizer: "no hazard with an SGPR offset"), and on gfx950
    // rows stored that way came out with the next instruction's value in them (tools/debug_encproj.py).
    const int orows = min(BM, M - em0);
    const int ocol0 = en0 + wn * TN * 32 - csh;
    constexpr int OB = OH ? 2 : 4;   // bytes per stored element
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(obase) + ((long long)em0 * ldo + ocol0) * OB, 0, (orows * ldo - ocol0) * OB,
        0x00020000);
    unsigned oro[NPR];
    int late = 0;   // an opaque zero made HERE: the lane offsets below are not computed ahead of the main
    asm volatile("" : "+v"(late));   // loop (where every register is taken)
#pragma unroll
    for (int ps = 0; ps < NPR; ++ps) {
      const int lrow = wm * 32 * RM + ps * 8 + erow + late;
      oro[ps] = (lrow < orows && !PAVE_PROBE_DROP_STORES()) ? (unsigned)(lrow * ldo * OB + ec4 * 4 * OB) : kOut;
    }
    constexpr bool LNW = LNORM && WIDE;   // LayerNorm computed on the accumulator layout (below)
    if constexpr (LNW) {
      __builtin_amdgcn_sched_barrier(0);   // (nothing of the epilogue is hoisted into the last slab)
      // The wave owns 32 whole rows of the N = 256 output: lane (lr, kh) holds, for r = 0..15, row
      // (r & 3) + 8 (r >> 2) + 4 kh at the columns lr + 32 j.  v = acc + bias + identity IN PLACE
      // (the identity read as 128-byte row segments: 32 lanes x 4 B), row statistics = in-lane sums
      // over the 8 tiles + an all-reduce over the 32 lanes of the half wave (two-pass: mean, then
      // centred squares), normalised in place -- no second register image of the tile, no LDS sweep,
      // no workgroup barrier; gamma / beta ride the ordinary store pass below.
      // identity rows through a buffer resource over [M, BN] (< 4 GiB, the launcher's condition): ONE
      // 32-bit lane offset, the row / tile part of the address is a scalar offset, and rows past M are
      // out-of-range reads (zeros: the hardware compares lane offset + scalar offset with num_records,
      // without wrap-around -- tools/microbench/buffer_range.hip) -- no 64-bit address per element
      const __amdgpu_buffer_rsrc_t idr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(residual ? residual : out), 0, residual ? (int)((unsigned)M * (unsigned)(BN * 4)) : 0,
          0x00020000);
      // three tiles of identity values in flight (48 registers), not more: the lane offset is made to
      // depend on the tile added last, so the compiler cannot hoist every tile's loads to the front
      // (it did: 128 more live registers, spills once anything else of the epilogue needed one)
#ifndef PAVE_IDD
#define PAVE_IDD 3
#endif
      constexpr int IDD = PAVE_IDD;
      float idv[IDD][16];
      int id_voff = ((wm * 32 + 4 * kh) * BN + lr) * 4;
      auto load_identity = [&](const int j) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          // (the column tile rides the instruction's immediate offset: 16 scalar offsets serve all tiles)
          idv[j % IDD][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
              idr, id_voff + j * 128, (em0 + (r & 3) + 8 * (r >> 2)) * (BN * 4) + en0 * 4, 0));
      };
#pragma unroll
      for (int j = 0; j < IDD - 1; ++j) load_identity(j);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float bj = bias ? bias[en0 + j * 32 + lr] : 0.f;
        if (j + IDD - 1 < TN) load_identity(j + IDD - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = (acc[j][r] + bj) + idv[j % IDD][r];
        asm volatile("" : "+v"(id_voff) : "v"(acc[j][0]), "v"(acc[j][1]), "v"(acc[j][2]), "v"(acc[j][3]),
                     "v"(acc[j][4]), "v"(acc[j][5]), "v"(acc[j][6]), "v"(acc[j][7]), "v"(acc[j][8]), "v"(acc[j][9]),
                     "v"(acc[j][10]), "v"(acc[j][11]), "v"(acc[j][12]), "v"(acc[j][13]), "v"(acc[j][14]),
                     "v"(acc[j][15]));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) sm += acc[j][r];
        const float mean = half32_sum(sm) * (1.f / (float)BN);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[j][r] -= mean;
          q = fmaf(acc[j][r], acc[j][r], q);
        }
        const float rs = rsqrtf(half32_sum(q) * (1.f / (float)BN) + ln.eps);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j][r] *= rs;
      }
    }
    if constexpr (!LNORM || LNW) {
      float2 erf[NPS];          // EPI 1: reference point (x, y) of the lane's level per pass
      float eW = 0.f, eH = 0.f, erW = 0.f, erH = 0.f;
      if constexpr (EPI == 1) {
        const int lvl = ec4 >> 1;   // 32 offset columns per head = 4 levels x 4 points x (x, y)
        eW = epi->fW[lvl], eH = epi->fH[lvl], erW = epi->rW[lvl], erH = epi->rH[lvl];
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
          const long long gm = (long long)em0 + wm * 32 + ps * 8 + erow;
          erf[ps] = gm < M ? *reinterpret_cast<const float2*>(epi->ref + gm * 8 + lvl * 2)
                           : make_float2(0.f, 0.f);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int rt = t / TN, j = t % TN;
        const int ncol0 = en0 + wn * TN * 32 + j * 32;   // wave-uniform
        const int ncol = ncol0 + ec4 * 4;
        const bool colok = WIDE || ncol < os.n_real;   // (wide form: N == n_real, the launcher's condition)
        const float4 b4 = (!LNW && bias && colok) ? *reinterpret_cast<const float4*>(bias + ncol)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), be4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (LNW) {
          g4 = *reinterpret_cast<const float4*>(ln.gamma + ncol);
          be4 = *reinterpret_cast<const float4*>(ln.beta + ncol);
        }
        if (HT && t == 1) {   // 16x16 layout: col = lane & 15, row = 4 (lane >> 4) + r (+ 16 for the second half)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(hh * 16 + 4 * (lane >> 4) + r) * QCST + (lane & 15)] = acct[hh][r];
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[((r & 3) + 8 * (r >> 2) + 4 * kh) * QCST + lr] = acc[t][r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (LNW) {
          // (one pass at a time: the 128 accumulators are live until their tile is stored)
#pragma unroll
          for (int ps = 0; ps < NPS; ++ps) {
            float4 v = *reinterpret_cast<const float4*>(Cs + (ps * 8 + erow) * QCST + ec4 * 4);
            v.x = fmaf(v.x, g4.x, be4.x), v.y = fmaf(v.y, g4.y, be4.y);
            v.z = fmaf(v.z, g4.z, be4.z), v.w = fmaf(v.w, g4.w, be4.w);
            if (PAVE_PROBE_NT_STORES())
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, oro[ps] + j * 128, 0, 2);
            else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, oro[ps] + j * 128, 0, 0);
          }
        } else {
          // the tile's four passes side by side: the wave-uniform options (residual, ReLU) are ONE branch
          // per tile each (the empty asm keeps them branches: as selects they cost up to 3 instructions
          // per value whether the option is on or not)
          float4 v[NPS];
#pragma unroll
          for (int ps = 0; ps < NPS; ++ps) {
            v[ps] = *reinterpret_cast<const float4*>(Cs + (ps * 8 + erow) * QCST + ec4 * 4);
            v[ps].x += b4.x, v[ps].y += b4.y, v[ps].z += b4.z, v[ps].w += b4.w;
          }
          if (residual) {
            asm volatile("" ::);
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
              const float4 rv = resv[t % RB][ps];
              v[ps].x += rv.x, v[ps].y += rv.y, v[ps].z += rv.z, v[ps].w += rv.w;
            }
          }
          if (relu == 1) {
            asm volatile("" ::);
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps)
              v[ps].x = relu_max(v[ps].x), v[ps].y = relu_max(v[ps].y), v[ps].z = relu_max(v[ps].z),
              v[ps].w = relu_max(v[ps].w);
          } else if (relu == 2) {     // exact GELU (nn.GELU: x Phi(x)): the Swin block's FFN activation
            asm volatile("" ::);
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps)
              v[ps].x = gelu_erf(v[ps].x), v[ps].y = gelu_erf(v[ps].y), v[ps].z = gelu_erf(v[ps].z),
              v[ps].w = gelu_erf(v[ps].w);
          } else if (relu == 3) {     // sigmoid (the heads' sigma branches)
            asm volatile("" ::);
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps)
              v[ps].x = sigmoid_f(v[ps].x), v[ps].y = sigmoid_f(v[ps].y), v[ps].z = sigmoid_f(v[ps].z),
              v[ps].w = sigmoid_f(v[ps].w);
          }
#pragma unroll
          for (int ps = 0; ps < NPS; ++ps) {
            if constexpr (EPI == 1) {   // two points (x, y) of one level
              v[ps].x = pave_enc::pixel_coord(erf[ps].x, v[ps].x, eW, erW);
              v[ps].y = pave_enc::pixel_coord(erf[ps].y, v[ps].y, eH, erH);
              v[ps].z = pave_enc::pixel_coord(erf[ps].x, v[ps].z, eW, erW);
              v[ps].w = pave_enc::pixel_coord(erf[ps].y, v[ps].w, eH, erH);
            }
            if constexpr (EPI == 2) {   // the quad holds the 16 logits of (row, head)
              float e[4], inv;
              pave_enc::softmax16(v[ps].x, v[ps].y, v[ps].z, v[ps].w, e, inv);
              v[ps].x = e[0] * inv, v[ps].y = e[1] * inv, v[ps].z = e[2] * inv, v[ps].w = e[3] * inv;
            }
            if constexpr (OH) {
              typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
              const u32x2s hv = {pack_rne_f16(v[ps].x, v[ps].y), pack_rne_f16(v[ps].z, v[ps].w)};
              __builtin_amdgcn_raw_buffer_store_b64(hv, ors, (colok ? oro[rt * NPS + ps] : kOut) + j * 64, 0, 0);
            } else if (PAVE_PROBE_NT_STORES())
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[ps]), ors,
                                                   (colok ? oro[rt * NPS + ps] : kOut) + j * 128, 0, 2);
            else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[ps]), ors,
                                                   (colok ? oro[rt * NPS + ps] : kOut) + j * 128, 0, 0);
          }
        }
        if ((WIDE || RM > 1) && !LNW && residual && t + RB < NT) prefetch_residual_tile(en0, t + RB, t % RB);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      // LayerNorm over the N = BN columns of a row: v = acc + bias + residual kept in registers,
      // row sums completed across the 8 lanes of a row segment, the TN tiles and the WN waves of
      // the row (through LDS); two passes: mean, then the centred sum of squares
      float* st1 = reinterpret_cast<float*>(smem + STAT_OFF);   // [QBM][WN]
      float* st2 = st1 + QBM * WN;
      float4 v[TN][NPS];
      float rsum[NPS];
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) rsum[ps] = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int ncol = en0 + wn * TN * 32 + j * 32 + ec4 * 4;
        const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + ncol)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[((r & 3) + 8 * (r >> 2) + 4 * kh) * QCST + lr] = acc[j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
          const int lrow = ps * 8 + erow;
          const long long gm = (long long)em0 + wm * 32 + lrow;
          float4 x = *reinterpret_cast<const float4*>(Cs + lrow * QCST + ec4 * 4);
          x.x += b4.x, x.y += b4.y, x.z += b4.z, x.w += b4.w;
          if (residual) {
            const float4 rv = resv[j % RB][ps];
            x.x += rv.x, x.y += rv.y, x.z += rv.z, x.w += rv.w;
          }
          if (gm >= M) x = make_float4(0.f, 0.f, 0.f, 0.f);
          v[j][ps] = x;
          rsum[ps] += (x.x + x.y) + (x.z + x.w);
        }
        if (WIDE && residual && j + RB < TN) prefetch_residual_tile(en0, j + RB, j % RB);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      // (WN == 1, the wide form: the wave owns whole rows -- the 8 lanes of a row segment hold the
      // row's totals after the shuffles, no exchange through LDS and no workgroup barrier)
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        float sm = rsum[ps];
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        rsum[ps] = sm;
        if (WN > 1 && ec4 == 0) st1[(wm * 32 + ps * 8 + erow) * WN + wn] = sm;
      }
      if constexpr (WN > 1) PAVE_QBAR();
      float rstd[NPS];
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int brow = wm * 32 + ps * 8 + erow;
        float sm = WN > 1 ? 0.f : rsum[ps];
        if constexpr (WN > 1) {
#pragma unroll
          for (int w = 0; w < WN; ++w) sm += st1[brow * WN + w];
        }
        const float mean = sm * (1.f / (float)BN);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float4& x = v[j][ps];
          x.x -= mean, x.y -= mean, x.z -= mean, x.w -= mean;
          q += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        if (WN > 1 && ec4 == 0) st2[brow * WN + wn] = q;
        rstd[ps] = q;
      }
      if constexpr (WN > 1) PAVE_QBAR();
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int brow = wm * 32 + ps * 8 + erow;
        float q = WN > 1 ? 0.f : rstd[ps];
        if constexpr (WN > 1) {
#pragma unroll
          for (int w = 0; w < WN; ++w) q += st2[brow * WN + w];
        }
        rstd[ps] = rsqrtf(q * (1.f / (float)BN) + ln.eps);
      }
      // column tile outermost: gamma / beta of one tile live at a time
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int ncol = en0 + wn * TN * 32 + j * 32 + ec4 * 4;
        const float4 g4 = *reinterpret_cast<const float4*>(ln.gamma + ncol);
        const float4 be4 = *reinterpret_cast<const float4*>(ln.beta + ncol);
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
          const long long gm = (long long)em0 + wm * 32 + ps * 8 + erow;
          float4 x = v[j][ps];
          x.x = fmaf(x.x * rstd[ps], g4.x, be4.x);
          x.y = fmaf(x.y * rstd[ps], g4.y, be4.y);
          x.z = fmaf(x.z * rstd[ps], g4.z, be4.z);
          x.w = fmaf(x.w * rstd[ps], g4.w, be4.w);
          if (gm < M) *reinterpret_cast<float4*>(out + gm * N + ncol) = x;
        }
      }
    }
  }
}

// (64-column tiles, TN = 2: 136 VGPRs and 42 KB of LDS -- three blocks per CU; these launches are
// issue-bound, not MFMA-bound, and take the extra wave per SIMD)
// (RM = 2: 256-row blocks, a wave owns two row tiles; ~200 VGPRs and 66 KB of LDS -- two blocks per CU)
template <int TN, int KIND, bool ABIAS, int RM = 1, int PL = 3, bool HT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((TN == 2 && RM == 1) ? 3 : 2, (TN == 2 && RM == 1) ? 3 : 2))) void gemm_q_kernel(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const int relu,
    const float* __restrict__ a_bias, const QConv g, const QOut os, const float* __restrict__ A2) {
  PAVE_CLOCK_BEGIN(0);
  gemm_q_body<TN, 1, KIND, ABIAS, false, false, 0, RM, PL, HT>(A, Wp, bias, residual, out, M, K, N, relu, a_bias, g,
                                                               os, QLn{nullptr, nullptr, 0.f}, A2);
  PAVE_CLOCK_END();
}
// the wide form: 128 x 256 block on 4 waves, 32 x 256 per wave, ring of 2
template <int KIND, int PL = 3, int IO = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_w_kernel(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const int relu,
    const QConv g, const QOut os, const float* __restrict__ A2) {
  PAVE_CLOCK_BEGIN(1);
  gemm_q_body<8, 1, KIND, false, false, true, 0, 1, PL, false, IO>(A, Wp, bias, residual, out, M, K, N, relu, nullptr,
                                                                   g, os, QLn{nullptr, nullptr, 0.f}, A2);
  PAVE_CLOCK_END();
}
// mixed tiles for N % 256 == 128 (the encoder's merged projection, N = 640): the first N / 256 column
// tiles of a row tile run the wide body, its last 128 columns the narrow one; the blocks of a row
// tile stay neighbours on one XCD
template <int PL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wn_kernel(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const int relu,
    const QConv g, const QOut os) {
  const int nw = N / 256, ntl = nw + 1;
  const int ttot = ((M + QBM - 1) / QBM) * ntl;
  const int per = ttot >> 3, rem = ttot & 7;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int lb = xcd * per + (xcd < rem ? xcd : rem) + idx;
  const int row = lb / ntl, c = lb - row * ntl;
  const QLn ln0{nullptr, nullptr, 0.f};
  PAVE_CLOCK_BEGIN(2);
  if (c < nw)
    gemm_q_body<8, 1, 0, false, false, true, 0, 1, PL>(A, Wp, bias, residual, out, M, K, N, relu, nullptr, g, os, ln0,
                                                       nullptr, row * QBM, c * 256);
  else
    gemm_q_body<4, 1, 0, false, false, false, 0, 1, PL>(A, Wp, bias, residual, out, M, K, N, relu, nullptr, g, os, ln0,
                                                        nullptr, row * QBM, nw * 256);
  PAVE_CLOCK_END();
}
// The encoder layer's merged projection (N = 640 = value 256 | sampling offsets 256 | attention
// logits 128, multi_scale_deform_attn.py:357-384) with the sampler's per-(query, head) arithmetic in
// the epilogue: tile 0 (value) is stored as it is, tile 1 as level pixel coordinates, the narrow
// tail as softmaxed attention weights -- with the code the sampling kernel itself uses
// (pave_enc_math.h), so pave_enc_deform_attn_tile_f32 in its `prepared` mode returns the same bits.
// These waves are ~25 % VALU-active; the sampler is VALU-bound.
template <int PL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wn_enc_kernel(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* residual,
    const float* __restrict__ value_bias, float* out, const int M, const int K, const QOut os, const QEpi epi) {
  constexpr int N = 640, ntl = 3;
  const int ttot = ((M + QBM - 1) / QBM) * ntl;
  const int per = ttot >> 3, rem = ttot & 7;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int lb = xcd * per + (xcd < rem ? xcd : rem) + idx;
  const int row = lb / ntl, c = lb - row * ntl;
  const QLn ln0{nullptr, nullptr, 0.f};
  const QConv g{0, 0, 0, 0, 0, 0, 0u};
  PAVE_CLOCK_BEGIN(3);
  if (c == 0)   // (value columns: one bias row instead of 256 table columns per row, when the caller has it)
    gemm_q_body<8, 1, 0, false, false, true, 0, 1, PL>(A, Wp, value_bias, value_bias ? nullptr : residual, out, M, K, N, 0,
                                                       nullptr, g, os, ln0, nullptr, row * QBM, 0);
  else if (c == 1)
    gemm_q_body<8, 1, 0, false, false, true, 1, 1, PL>(A, Wp, nullptr, residual, out, M, K, N, 0, nullptr, g, os,
                                                       ln0, nullptr, row * QBM, 256, &epi);
  else
    gemm_q_body<4, 1, 0, false, false, false, 2, 1, PL>(A, Wp, nullptr, residual, out, M, K, N, 0, nullptr, g, os,
                                                        ln0, nullptr, row * QBM, 512, &epi);
  PAVE_CLOCK_END();
}
// ---------------------------------------------------------------------------
// ResNet Bottleneck (64-channel stage) from its 3x3 convolution on, chained with the NEXT block's
// conv1 -- three GEMMs of ONE 128-pixel row tile back to back in one workgroup:
//   c2  = relu(conv3x3(c1) + b2)                         [M, 64]    K = 576   (narrow form)
//   out = relu([c2 | a2] @ W3^T + b3 + residual)         [M, 256]   K = 64 | 64 + k2 (wide form)
//   c1n = relu(out @ W1n^T + b1n)                        [M, 64 | 128]   K = 256  (narrow form)
// (third_party/mmdetection/mmdet/models/backbones/resnet.py:263-300 Bottleneck.forward, conv2 ->
// conv3 -> + identity | downsample -> relu, then the next block's conv1).  The tile's c2 and out
// rows go through global memory (written, re-read by LDS-DMA after a workgroup barrier: they come
// back from L2), so the three row GEMMs are the unchanged bodies above.  Why one launch: on the
// 1.9 M-pixel layer1 maps conv3 + identity is HBM-bound (4.3 GB per launch), the 3x3 and conv1 are
// MFMA- / issue-bound; chained, the workgroups of a CU sit in different phases and the two resources
// overlap, and conv1 reads `out` from L2 instead of HBM.
// ---------------------------------------------------------------------------
struct ChainArgs {
  const float* c1; const uint16_t* w2; const float* b2; float* c2;
  const uint16_t* w3; const float* b3; const float* residual; const float* a2; float* out;
  const uint16_t* w1n; const float* b1n; float* c1n;
  int M, H, W, K3, k1;
};
__device__ __forceinline__ void chain_sync() {
  // The tile's rows are written and re-read by waves of ONE workgroup = one CU: its stores have
  // reached L2 (write-through L1, vmcnt(0)) and the CU's own L1 stays coherent with its stores,
  // so workgroup scope is enough.  (Agent scope writes back / invalidates the whole L2 of the
  // XCD per tile: measured 3x slower.)
  if (PAVE_PROBE_NO_CHAIN_SYNC()) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// HAS_A = false: the launch starts at conv3 (the 3x3 was a launch of its own: at three blocks per CU
// the 64-column-tile 3x3 runs faster alone than as the first phase of a two-blocks-per-CU chain);
// CN = outputs of the next conv1 (0: none; 64 | 128: narrow form)
// RMC = 2: the workgroup carries a 256-row tile -- the 64-column bodies (3x3, a 64-output conv1) as ONE body
// with two row tiles per wave, the 256- / 128-column bodies as two 128-row halves back to back
template <bool HAS_A, int KIND2, int CN, int RMC = 1, int PL = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck_chain_kernel(
    const ChainArgs p) {
  const QLn ln0{nullptr, nullptr, 0.f};
  // the block's row tile (XCD-aware order over the row tiles), the same for every body
  constexpr int CBM = QBM * RMC;
  const int ttot = (p.M + CBM - 1) / CBM;
  const int per = ttot >> 3, rem = ttot & 7;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tm0 = (xcd * per + (xcd < rem ? xcd : rem) + idx) * CBM;
  PAVE_CLOCK_BEGIN(4);
  if constexpr (HAS_A) {
    gemm_q_body<2, 1, 1, false, false, false, 0, RMC, PL>(p.c1, p.w2, p.b2, nullptr, p.c2, p.M, 576, 64, 1, nullptr,
                                                      QConv{p.H, p.W, 64, p.H, p.W, 1, 1u << 30},
                                                      QOut{nullptr, 0, 0, 64, 0}, ln0, nullptr, tm0, 0);
    chain_sync();
  }
#pragma unroll
  for (int h = 0; h < RMC; ++h) {
    if (h > 0) PAVE_QBAR();   // (every wave is done with the previous half's epilogue chunks in the ring)
    if (tm0 + h * QBM < p.M)
      gemm_q_body<8, 1, KIND2, false, false, true, 0, 1, PL>(p.c2, p.w3, p.b3, p.residual, p.out, p.M, p.K3, 256, 1,
                                                   nullptr, QConv{0, 0, p.k1, 0, 0, 0},
                                                   QOut{nullptr, 0, 0, 256, 0}, ln0, p.a2, tm0 + h * QBM, 0);
  }
  if constexpr (CN > 0) {
    chain_sync();
    if constexpr (CN == 64 && RMC == 2) {
      gemm_q_body<2, 1, 0, false, false, false, 0, 2, PL>(
          p.out, p.w1n, p.b1n, nullptr, p.c1n, p.M, 256, CN, 1, nullptr, QConv{0, 0, 0, 0, 0, 0},
          QOut{nullptr, 0, 0, CN, 0}, ln0, nullptr, tm0, 0);
    } else {
#pragma unroll
      for (int h = 0; h < RMC; ++h) {
        if (h > 0) PAVE_QBAR();
        if (tm0 + h * QBM < p.M)
          gemm_q_body<(CN > 0 ? CN / 32 : 2), 1, 0, false, false, false, 0, 1, PL>(
              p.out, p.w1n, p.b1n, nullptr, p.c1n, p.M, 256, CN, 1, nullptr, QConv{0, 0, 0, 0, 0, 0},
              QOut{nullptr, 0, 0, CN, 0}, ln0, nullptr, tm0 + h * QBM, 0);
      }
    }
  }
  PAVE_CLOCK_END();
}

// 128 x 256 block on 8 waves (two per SIMD, one block per CU), narrow form: the block owns whole
// rows of an N = 256 output, LayerNorm runs in the epilogue.  (The wide form with this epilogue --
// a wave owning whole rows, no statistics exchange -- was built twice: with the 8 x 16 result
// registers next to the draining accumulators it spilled 45 dwords per lane and ran 1-5 % slower;
// with three sweeps of the accumulators through the LDS chunk (sum, centred squares, normalise)
// it ran EQUAL, 1.736 vs 1.743 ms at K = 1024 and 0.612 vs 0.615 ms at K = 256: these launches sit
// on a mixed HBM / MFMA bound that the tile form does not move.  Not kept.)
template <int PL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_q_ln_kernel(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const QLn ln) {
  // (A first-tile stagger of every other CU -- half a tile of s_sleep, so that main loops and the
  // HBM-heavy LayerNorm epilogues of different CUs interleave instead of running in lockstep -- was
  // measured: 647 -> 729 us at K = 256, equal at K = 1024.  Not kept.)
  PAVE_CLOCK_BEGIN(5);
  gemm_q_body<4, 2, 0, false, true, false, 0, 1, PL>(A, Wp, bias, residual, out, M, K, N, 0, nullptr,
                                                     QConv{0, 0, 0, 0, 0, 0}, QOut{nullptr, 0, 0, N, 0}, ln);
  PAVE_CLOCK_END();
}

// LayerNorm epilogue on the wide form: 4 waves, a wave owns 32 whole rows of the N = 256 output (no
// statistics exchange between waves, no workgroup barrier in the epilogue), two blocks per CU.
// Round 3 built this with a second register image of the tile (spills) and with LDS sweeps (equal to
// the 8-wave form); here bias + identity are added and the statistics taken IN the accumulator
// registers (see LNW in gemm_q_body), which costs no registers and no LDS pass.
template <int PL, int IO = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_w_ln_kernel(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const QLn ln) {
  PAVE_CLOCK_BEGIN(6);
  gemm_q_body<8, 1, 0, false, true, true, 0, 1, PL, false, IO>(A, Wp, bias, residual, out, M, K, N, 0, nullptr,
                                                               QConv{0, 0, 0, 0, 0, 0}, QOut{nullptr, 0, 0, N, 0}, ln);
  PAVE_CLOCK_END();
}

// ---------------------------------------------------------------------------
// ResNet / HRNet stem: 7x7 / stride 2 / pad 3 convolution of the NCHW image batch, 3 -> 64
// channels, NHWC output (third_party/mmdetection/mmdet/models/backbones/resnet.py:607-611 conv1).
// Block = 3 waves = 96 output pixels of R output rows; the 3 x (2 R + 5) input rows of its
// window are copied ONCE into LDS by LDS-DMA (aligned 16-byte chunks, coalesced; chunks outside
// the image come from a zero chunk = the zero padding) and every wave builds its MFMA A operands
// from there: K axis = (c, ky, kx' = 0..7) where kx' = kx + 1 is the tap shifted by one so that
// the lane's 8 taps start at the even window column 2 p (8-byte aligned ds_read_b64); tap kx' = 0
// is not part of the convolution and is ZEROED in the operand (not merely weighted by zero: a
// NaN pixel must poison only its own windows).  K = 22 rows x 8 = 176 = 11 slabs of 16 (row 21 is
// a zero row).  The W fragments (64 output channels: 2 column tiles x 3 planes) are read straight
// from global memory / L2 into registers one slab ahead -- no LDS ring, no barrier in the loop.
// ---------------------------------------------------------------------------
constexpr int SBM = 96;                 // output pixels per block
constexpr int SWC = 50;                 // window chunks (16 B) per row: 2 * 95 + 8 = 198 floats -> 200
constexpr int SNSLAB = 11;

// R output rows per block (rows R b .. R b + R - 1): a wave's 32 pixels of every row share each W fragment
// -- 12 R MFMAs per fetched slab of weight planes, 1 / R of the L2 traffic for them (with one row per block
// 78 400 blocks read the 67 KB of planes once each: ~14 TB/s of L2 reads) -- and the rows' windows overlap
// (2 R + 5 window rows per channel instead of 7 R).  K row (c, ky) of output row r sits at window row
// (2 R + 5) c + ky + 2 r; the products of an accumulator keep their order whatever R is: R = 1, 2, 3, 4 are
// bit-identical.  Measured (28 x 800 x 1344, tools/lib_ab.py): R = 1 1 139 us, R = 2 969, R = 3 1 073,
// R = 4 1 036 (228 registers: two waves per SIMD) -- R = 2 (136 registers, 22 KB of LDS) is the launcher's.
template <int R>
struct StemRows {
  static constexpr int WPC = 2 * R + 5;           // window rows per channel
  static constexpr int ROWS = 3 * WPC + 1;        // + one zero row
  static constexpr int WIN = ((ROWS * SWC + 63) / 64) * 1024;
};

template <int R, int PL = 3>
__global__ __launch_bounds__(192) void stem7x7_qr_kernel(
    const float* __restrict__ x, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ y, const int H, const int W, const int Ho, const int Wo, const int relu) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  PAVE_CLOCK_BEGIN(7);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, kh = lane >> 5;
  const int segs = (Wo + SBM - 1) / SBM;
  constexpr int WPC = StemRows<R>::WPC, ZROW = 3 * WPC;
  const int Hp = (Ho + R - 1) / R;      // row groups
  const int seg = blockIdx.x % segs;
  const int oy = R * ((blockIdx.x / segs) % Hp);
  const int n = blockIdx.x / (segs * Hp);
  const int ox0 = seg * SBM;
  const int xs = 2 * ox0 - 4;           // image column of window column 0 (a multiple of 4)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  constexpr int NCH = StemRows<R>::ROWS * SWC;
  constexpr int NINS = StemRows<R>::WIN / 1024;    // (R = 2: 22 instructions over 3 waves)
#pragma unroll
  for (int q = 0; q < (NINS + 2) / 3; ++q) {
    const int ins = wave + 3 * q;
    if (ins < NINS) {
      const int ci = ins * 64 + lane;
      const int row = ci / SWC, cc = ci - row * SWC;
      const int c = row / WPC, wy = row - WPC * c;
      const int iy = 2 * oy - 3 + wy, ix = xs + 4 * cc;
      const bool ok = ci < NCH && row < ZROW && iy >= 0 && iy < H && ix >= 0 && ix + 4 <= W;
      const float* src = ok ? x + (((long long)n * 3 + c) * H + iy) * W + ix
                            : reinterpret_cast<const float*>(g_zero_chunk);
      dma16_flat(src, lds0 + ins * 1024);
    }
  }
  const uint16_t* wl = Wp + ((long long)lr * 16 + kh * 8);
  u32x4 wf[2][PL][2];
  auto load_w = [&](const int slab, const int set) {
#pragma unroll
    for (int p = 0; p < PL; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wf[set][p][j] = *reinterpret_cast<const u32x4*>(wl + ((slab * PL + p) * 64 + 32 * j) * 16);
  };
  load_w(0, 0);
  const int a_px = (wave * 32 + lr) * 8;
  u32x4 apl[R][PL];
  // A operand of output row r, slab s: K row 2 s + kh = (c, ky) -> window row WPC c + ky + 2 r (21 -> the zero row)
  auto read_split = [&](const int slab, const int r) {
    const int k0 = 2 * slab, k1 = 2 * slab + 1;
    const int w0 = k0 >= 21 ? ZROW : WPC * (k0 / 7) + k0 % 7 + 2 * r;
    const int w1 = k1 >= 21 ? ZROW : WPC * (k1 / 7) + k1 % 7 + 2 * r;
    const unsigned char* p = smem + a_px + (kh ? w1 : w0) * (SWC * 16);
    const f32x2 v0 = *reinterpret_cast<const f32x2*>(p);
    const f32x2 v1 = *reinterpret_cast<const f32x2*>(p + 8);
    const f32x2 v2 = *reinterpret_cast<const f32x2*>(p + 16);
    const f32x2 v3 = *reinterpret_cast<const f32x2*>(p + 24);
    if constexpr (PL == 3)
      split8(f32x4{0.f, v0.y, v1.x, v1.y}, f32x4{v2.x, v2.y, v3.x, v3.y}, apl[r]);   // tap kx' = 0: zeroed
    else
      apl[r][0] = u32x4{pack_rne_f16(0.f, v0.y), pack_rne_f16(v1.x, v1.y), pack_rne_f16(v2.x, v2.y),
                        pack_rne_f16(v3.x, v3.y)};
  };
  f32x16 acc[R][2];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][j][i] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // the window has landed everywhere
#pragma unroll
  for (int s = 0; s < SNSLAB; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    if (s + 1 < SNSLAB) load_w(s + 1, nxt);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      read_split(s, r);
      if constexpr (PL == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
              __builtin_bit_cast(f16x8, apl[r][0]), __builtin_bit_cast(f16x8, wf[cur][0][j]), acc[r][j], 0, 0, 0);
      } else {
#pragma unroll
      for (int o = 2; o >= 0; --o)
#pragma unroll
        for (int pa = 0; pa <= o; ++pa)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, apl[r][pa]), __builtin_bit_cast(bf16x8, wf[cur][o - pa][j]),
                acc[r][j], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the window
  float* Cs = reinterpret_cast<float*>(smem) + wave * 32 * QCST;
  const int erow = lane >> 3, ec4 = lane & 7;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float* const orow = y + ((long long)n * Ho + oy + r) * Wo * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = j * 32 + ec4 * 4;
      const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 16; ++i) Cs[((i & 3) + 8 * (i >> 2) + 4 * kh) * QCST + lr] = acc[r][j][i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int lrow = ps * 8 + erow;
        const int ox = ox0 + wave * 32 + lrow;
        float4 v = *reinterpret_cast<const float4*>(Cs + lrow * QCST + ec4 * 4);
        v.x += b4.x, v.y += b4.y, v.z += b4.z, v.w += b4.w;
        if (relu) {
          v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
        }
        if (ox < Wo && oy + r < Ho) *reinterpret_cast<float4*>(orow + (long long)ox * 64 + col) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  PAVE_CLOCK_END();
}

template <int TN, int KIND, bool ABIAS, int RM = 1, int PL = 3, bool HT = false>
int launch_q(const float* a, const uint16_t* w, const float* bias, const float* residual, float* out,
             long long M, int K, int N, int relu, const float* a_bias, hipStream_t st, const QConv g,
             const QOut os, const float* a2 = nullptr, int ksplit = 1) {
  constexpr int BN = TN * 32;
  constexpr int BM = QBM * RM;
  constexpr int STAGE = BM * 64 + PL * BN * 32;
  // (the per-wave epilogue chunks reuse the ring: 4 x 32 rows x QCST floats)
  constexpr int EPIB = 4 * 32 * QCST * 4;
  const int smem = (QNS * STAGE > EPIB ? QNS * STAGE : EPIB) + (ABIAS ? K * 4 : 0);
  const long long gx = ((M + BM - 1) / BM) * (N / BN);
  if (gx >= (1ll << 31)) return pave_internal_fail(PAVE_E_ARG, "gemm_q: grid too large");
  auto kern = gemm_q_kernel<TN, KIND, ABIAS, RM, PL, HT>;
  static int attr_smem = 0;
  if (smem > attr_smem) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return pave_internal_fail(PAVE_E_LAUNCH, "gemm_q: cannot raise dynamic LDS limit");
    attr_smem = smem;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ksplit), dim3(256), smem, st, a, w, bias,
                     residual, out, (int)M, K, N, relu, a_bias, g, os, a2);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

// split-K second pass: out = act(sum over the parts IN ORDER + bias + residual), float4 per lane
__global__ __launch_bounds__(256) void splitk_reduce_kernel(
    const float* __restrict__ ws, const int parts, const long long total4, const int n4,
    const float* __restrict__ bias, const float* __restrict__ residual, const int relu,
    float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const float4* w4 = reinterpret_cast<const float4*>(ws);
  float4 v = w4[i];
  for (int p = 1; p < parts; ++p) {
    const float4 t = w4[p * total4 + i];
    v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
  }
  if (bias) {
    const float4 b = reinterpret_cast<const float4*>(bias)[i % n4];
    v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
  }
  if (residual) {
    const float4 r = reinterpret_cast<const float4*>(residual)[i];
    v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
  }
  if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
  reinterpret_cast<float4*>(out)[i] = v;
}

// ---------------------------------------------------------------------------
// Small-row form: the decoders' and heads' Linears have a few hundred to ~1 200 rows (300 queries per
// clip, 15 joint queries per pose) -- 10 row tiles of the 128-row kernels for 256 CUs, each walking its
// whole K axis behind one barrier per slab: latency, not throughput (17 us at K = 256, 66 us at
// K = 1024).  Here a WAVE owns a 32-row x 32 TN-column tile and walks K on its own: the A fragment
// (8 consecutive fp32 of the lane's row) and the W fragments (the lane's 16-byte piece of each plane)
// come straight from global memory / L2 into registers, PF slabs ahead -- no LDS, no barrier, no
// workgroup-level dependence at all -- so a launch is hundreds to thousands of independent waves
// spread over every CU.  Same six products per slab in the same order per accumulator as the tile
// kernels: results are bit-identical to them (and to a batch large enough to take the tile kernels).
// Forms: plain rows (row stride lda), grouped columns (group i = columns [i K, (i + 1) K) of A),
// bias, full or row-periodic residual, ReLU, zero-padded weight planes (n_real <= N).  The shipped
// selection uses TN = 1 (32 x 32 per wave) for outputs up to 512 columns.
// ---------------------------------------------------------------------------
template <int TN, int PF, int PL = 3>
__global__ __launch_bounds__(256) void gemm_s_kernel(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const int relu,
    const int lda, const int group_n, const int res_rows, const int n_real) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 31, kh = lane >> 5;
  const int ntn = (N + 32 * TN - 1) / (32 * TN);
  const int t = blockIdx.x * 4 + wave;                 // the four waves of a block: neighbouring
  const int tm = t / ntn, tn = t - tm * ntn;           // column tiles of one row tile (A rows via L1)
  if (tm * 32 >= M) return;                            // (wave-uniform; no barrier anywhere)
  const int m0 = tm * 32, n0 = tn * 32 * TN;
  const int nslabs = K >> 4;
  const int arow = min(m0 + lr, M - 1);                // rows past M: stand-in data, never stored
  const float* ap = A + (long long)arow * lda + (group_n > 0 ? (n0 / group_n) * K : 0) + kh * 8;
  // plane row of column n: 16 bf16 = 32 bytes; the lane's half kh
  const uint16_t* wp = Wp + ((long long)(n0 + lr) * 16 + kh * 8);
  const long long w_plane = (long long)N * 16, w_slab = PL * w_plane;   // in 16-bit elements
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f32x4 raw[PF][2];
  u32x4 wf[PF][PL][TN];
  auto load = [&](const int slab, const int set) {
    raw[set][0] = *reinterpret_cast<const f32x4*>(ap + slab * 16);
    raw[set][1] = *reinterpret_cast<const f32x4*>(ap + slab * 16 + 4);
#pragma unroll
    for (int p = 0; p < PL; ++p)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        // (column tiles past the planes' N rows -- N % (32 TN) != 0 -- re-read the last tile: not stored)
        const long long jo = (n0 + j * 32 < N) ? (long long)j * 32 * 16 : 0;
        wf[set][p][j] = *reinterpret_cast<const u32x4*>(wp + slab * w_slab + p * w_plane + jo);
      }
  };
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u < nslabs) load(u, u);
  for (int s = 0; s < nslabs; s += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (s + u < nslabs) {
        u32x4 apl[PL];
        if constexpr (PL == 1) {
          apl[0] = u32x4{pack_rne_f16(raw[u][0].x, raw[u][0].y), pack_rne_f16(raw[u][0].z, raw[u][0].w),
                         pack_rne_f16(raw[u][1].x, raw[u][1].y), pack_rne_f16(raw[u][1].z, raw[u][1].w)};
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                __builtin_bit_cast(f16x8, apl[0]), __builtin_bit_cast(f16x8, wf[u][0][j]), acc[j], 0, 0, 0);
        } else {
        split8(raw[u][0], raw[u][1], apl);
        // the products of order o = pa + pb: o = 2, 1, 0 (smallest terms first), as the tile kernels
#pragma unroll
        for (int o = 2; o >= 0; --o)
#pragma unroll
          for (int pa = 0; pa <= o; ++pa)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, apl[pa]), __builtin_bit_cast(bf16x8, wf[u][o - pa][j]),
                  acc[j], 0, 0, 0);
        }
        if (s + u + PF < nslabs) load(s + u + PF, u);
      }
    }
  }
  // ---- epilogue from the accumulator layout: lane (lr, kh), register r = row (r & 3) + 8 (r >> 2) +
  // 4 kh, column lr of tile j; a row's 32 columns are one 128-byte segment
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + j * 32 + lr;
    const bool colok = col < n_real;
    const float bj = (bias && colok) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (row < M && colok) {
        float v = acc[j][r] + bj;
        if (residual) {
          const int rr = res_rows > 0 ? (int)((unsigned)row % (unsigned)res_rows) : row;
          v += residual[(long long)rr * n_real + col];
        }
        if (relu == 1) v = fmaxf(v, 0.f);
        else if (relu == 2) v = gelu_erf(v);
        else if (relu == 3) v = sigmoid_f(v);
        out[(long long)row * n_real + col] = v;
      }
    }
  }
}

template <int TN, int PF, int PL = 3>
int launch_s(const float* a, const uint16_t* w, const float* bias, const float* residual, float* out,
             long long M, int K, int N, int relu, int lda, int group_n, int res_rows, int n_real,
             hipStream_t st) {
  const long long tiles = ((M + 31) / 32) * ((N + 32 * TN - 1) / (32 * TN));
  hipLaunchKernelGGL((gemm_s_kernel<TN, PF, PL>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a, w, bias,
                     residual, out, (int)M, K, N, relu, lda, group_n, res_rows, n_real);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}
// ---------------------------------------------------------------------------
// Small-row form with the K axis split over the four waves of a block (round 6).  The form above gives a wave
// a 32 x 32 tile and the WHOLE K axis: 64 dependent slab steps at K = 1024, each behind its own L2 round trip --
// 26 - 33 us for the decoders' FFN2 (300 .. 1 200 x 1024 x 256) on a chip that is idle but for 80 - 300 waves.
// Here a BLOCK owns the 32 x 32 tile and wave w walks slabs [w q, (w + 1) q), q = ceil(K / 16 / 4): a quarter of
// the dependent steps, four times the waves in flight; the four partial tiles meet in LDS (16 KiB) and are added
// IN ORDER ((p0 + p1) + p2) + p3 -- deterministic, run to run and whatever else the chip is doing -- by all four
// waves, wave w finishing rows 8 w .. 8 w + 7 of the tile (bias, full or row-periodic residual, activation,
// zero-padded planes: the epilogue of the form above).  Per wave the six products per slab keep the tile kernels'
// order; the sum over the four K quarters is a different association of the same fp32 terms, so the form equals
// the others to rounding (1e-6 relative), not bit for bit -- as every split-K launch of this file.
// Taken for plain / grouped row GEMMs of up to kSkTiles 32 x 32 tiles with K >= 512 (the decoders' FFN2, the
// 512-wide layers of the key-point branch MLPs, the layer4 1x1 reductions of a one-clip batch).
// ---------------------------------------------------------------------------
template <int PF, int PL = 3>
__global__ __launch_bounds__(256) void gemm_sk_kernel(
    const float* __restrict__ A, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
    const float* residual, float* out, const int M, const int K, const int N, const int relu,
    const int lda, const int group_n, const int res_rows, const int n_real) {
  __shared__ float part[4][16][64];                    // [wave][accumulator register][lane]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 31, kh = lane >> 5;
  const int ntn = (N + 31) / 32;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
  const int m0 = tm * 32, n0 = tn * 32;
  const int nslabs = K >> 4;
  const int per = (nslabs + 3) >> 2;
  const int sb = wave * per, se = min(nslabs, sb + per);       // this wave's slabs (wave-uniform)
  const int arow = min(m0 + lr, M - 1);                // rows past M: stand-in data, never stored
  const float* ap = A + (long long)arow * lda + (group_n > 0 ? (n0 / group_n) * K : 0) + kh * 8;
  const uint16_t* wp = Wp + ((long long)(n0 + lr) * 16 + kh * 8);
  const long long w_plane = (long long)N * 16, w_slab = PL * w_plane;   // in 16-bit elements
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // the epilogue's operands -- bias of the lane's column, identity of its four output elements -- are fetched NOW,
  // beside the first slabs (an element is read by the thread that writes it: in-place residuals stay correct);
  // behind the LDS exchange they were one more dependent memory round trip of a 7 - 10 us launch
  const int col = n0 + lr;
  const bool colok = col < n_real;
  const float bj = (bias && colok) ? bias[col] : 0.f;
  float rv[4] = {0.f, 0.f, 0.f, 0.f};
  if (residual && colok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * wave + i;
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (row < M) {
        const int rr = res_rows > 0 ? (int)((unsigned)row % (unsigned)res_rows) : row;
        rv[i] = residual[(long long)rr * n_real + col];
      }
    }
  }
  f32x4 raw[PF][2];
  u32x4 wf[PF][PL];
  auto load = [&](const int slab, const int set) {
    raw[set][0] = *reinterpret_cast<const f32x4*>(ap + slab * 16);
    raw[set][1] = *reinterpret_cast<const f32x4*>(ap + slab * 16 + 4);
#pragma unroll
    for (int p = 0; p < PL; ++p)
      wf[set][p] = *reinterpret_cast<const u32x4*>(wp + slab * w_slab + p * w_plane);
  };
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (sb + u < se) load(sb + u, u);
  for (int s = sb; s < se; s += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (s + u < se) {
        u32x4 apl[PL];
        if constexpr (PL == 1) {
          apl[0] = u32x4{pack_rne_f16(raw[u][0].x, raw[u][0].y), pack_rne_f16(raw[u][0].z, raw[u][0].w),
                         pack_rne_f16(raw[u][1].x, raw[u][1].y), pack_rne_f16(raw[u][1].z, raw[u][1].w)};
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, apl[0]),
                                                       __builtin_bit_cast(f16x8, wf[u][0]), acc, 0, 0, 0);
        } else {
          split8(raw[u][0], raw[u][1], apl);
#pragma unroll
          for (int o = 2; o >= 0; --o)
#pragma unroll
            for (int pa = 0; pa <= o; ++pa)
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, apl[pa]),
                                                            __builtin_bit_cast(bf16x8, wf[u][o - pa]), acc, 0, 0, 0);
        }
        if (s + u + PF < se) load(s + u + PF, u);
      }
    }
  }
  // ---- the four partial tiles meet in LDS; wave w finishes registers 4 w .. 4 w + 3 = rows 8 w + (0 .. 3) + 4 kh
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * wave + i;
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    float v = ((part[0][r][lane] + part[1][r][lane]) + part[2][r][lane]) + part[3][r][lane];
    if (row < M && colok) {
      v += bj;
      if (residual) v += rv[i];
      if (relu == 1) v = fmaxf(v, 0.f);
      else if (relu == 2) v = gelu_erf(v);
      else if (relu == 3) v = sigmoid_f(v);
      out[(long long)row * n_real + col] = v;
    }
  }
}

template <int PL = 3>
int launch_sk(const float* a, const uint16_t* w, const float* bias, const float* residual, float* out,
              long long M, int K, int N, int relu, int lda, int group_n, int res_rows, int n_real,
              hipStream_t st) {
  const long long tiles = ((M + 31) / 32) * ((N + 31) / 32);
  hipLaunchKernelGGL((gemm_sk_kernel<4, PL>), dim3((unsigned)tiles), dim3(256), 0, st, a, w, bias, residual, out,
                     (int)M, K, N, relu, lda, group_n, res_rows, n_real);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}
// Taken from K = 512 on, up to kSkTiles blocks (tools/small_gemm_ksplit_ab.py, us per launch behind a busy stream,
// K-split | one-wave form | tile kernels): at K = 256 a wave's whole walk is 16 slabs and the split only adds blocks
// (1 200 x 256 x 2 688: 22 | 15 | 15), at K = 1 024 it is what the launch waits for (300 .. 1 200 x 1024 x 256:
// < 15 | 24 | 34).
// (Linear + identity + LayerNorm at few rows as ONE launch -- a 1 024-thread block owning 32 whole rows, 8 column
// tiles x 2 K halves, LayerNorm from LDS -- was built in round 6 and measured SLOWER than the K-split GEMM + LayerNorm
// pass it was meant to replace: 300 x 256 x 256 13.8 against 7.6 us, 300 x 1024 x 256 41.8 against 11.4, 1 200 x 1024 x 256
// 41.9 against 19.2 (tools/small_gemm_ksplit_ab.py at commit 188af1e: M / 32 blocks are 10 - 38 blocks for 256 CUs
// and a wave walks half of K alone).  Removed; the two-launch form stays.)
constexpr long long kSkTiles = 4096;     // 32 x 32 tiles (= blocks) up to which the K-split small-row form is taken
constexpr long long kSkTiles256 = 1280;   // ... and at K = 256 .. 511 (4 - 7 slabs per wave): only where the blocks fit one round
// ... and only up to 2 048 rows: a block re-reads its A rows once per 32 output columns, which a few hundred
// query rows do from L2 for free and a 3 150-pixel layer4 map of a one-clip batch does not (3 150 x 2048 x 512 took
// 93 us on this form in the step's trace, 45 on the tile kernels).
constexpr long long kSkRows = 2048;
inline bool small_rows_ksplit_form(long long M, int N, int K) {
  const long long tiles = ((M + 31) / 32) * (((long long)N + 31) / 32);
  return M <= kSkRows && ((K >= 512 && tiles <= kSkTiles) || (K >= 256 && tiles <= kSkTiles256));
}

// Which launches take the small-row form: fewer than 64 tiles of 128 x 128 (and < 8 192 rows).  tools/
// small_vs_tile.py (us, small-row | tile kernels): 1 200 x 1024 x 256: 26 | 35, 1 200 x 512 x 512: 16 | 21,
// 3 150 x 1536 x 256: 42 | 50 -- 2 100 x 2048 x 512: 86 | 65, 3 150 x 2048 x 512: 92 | 69, 6 000 x 1024 x 256: 47 | 36,
// 8 000 x 2048 x 512: 167 | 83 (until round 5 the rule was rows alone, < 8 192: a one-clip batch's layer4
// convolutions sat on the wrong side of it).  The forms are bit-identical.
constexpr long long kSmallRows = 8192;
inline bool small_rows_form(long long M, int N) {
  return M < kSmallRows && ((M + 127) / 128) * ((N + 127) / 128) < 64;
}

constexpr int W_SMEM = 2 * (QBM * 64 + 3 * 256 * 32);   // wide form: ring of 2 x 32 KiB
template <int PL>
constexpr int w_smem() { return 2 * (QBM * 64 + PL * 256 * 32); }   // (fp16: 2 x 16 KiB; >= the epilogue chunks)
// the mixed-tile kernels (wide tiles + a 128-column narrow tail with its ring of 3): the larger of the two
template <int PL>
constexpr int wn_smem() {
  return w_smem<PL>() > QNS * (QBM * 64 + PL * 128 * 32) ? w_smem<PL>() : QNS * (QBM * 64 + PL * 128 * 32);
}

template <int KIND, int PL = 3>
int launch_w(const float* a, const uint16_t* w, const float* bias, const float* residual, float* out,
             long long M, int K, int N, int relu, hipStream_t st, const QConv g, const QOut os,
             const float* a2 = nullptr, int ksplit = 1) {
  const long long gx = ((M + QBM - 1) / QBM) * (N / 256);
  if (gx >= (1ll << 31)) return pave_internal_fail(PAVE_E_ARG, "gemm_w: grid too large");
  auto kern = gemm_w_kernel<KIND, PL>;
  // (diag variant 20, tools/coresidency_probe.py: 40 KiB more dynamic LDS than the kernel uses, so that ONE block
  // fits a CU instead of two -- what a GEMM that leaves room for a co-resident sampler block would look like)
#ifdef PAVE_DIAG
  constexpr int kPad = 40 * 1024;
#else
  constexpr int kPad = 0;
#endif
  const int smem = w_smem<PL>() + (pave_internal_diag_variant() == 20 ? kPad : 0);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, w_smem<PL>() + kPad) != hipSuccess)
      return pave_internal_fail(PAVE_E_LAUNCH, "gemm_w: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ksplit), dim3(256), smem, st, a, w, bias,
                     residual, out, (int)M, K, N, relu, g, os, a2);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

}  // namespace

// Split-K plan of an [M, Kp] x [Np, Kp] product: how many parts the K axis is cut into (1 = none)
// and the slabs per part.  Worth it when the tiles alone leave most of the 512 block slots of the
// chip empty and K is very long; every part gets an even number >= 64 of slabs.
void pave_internal_splitk_plan(long long M, int Kp, int Np, int* ksplit, int* ks_slabs) {
  const long long tiles = ((M + QBM - 1) / QBM) * (Np % 256 == 0 ? Np / 256 : (Np + 127) / 128);
  const int nsl = Kp / 16;
  *ksplit = 1, *ks_slabs = 0;
  // Few tiles and a very long K (K >= 8192: the ChannelMapper's 3x3 on C5): 8 parts.
  // WHAT THE PLAN DEPENDS ON: the K range and the TILE-COUNT CLASS of the launch (< 16, < 128, < 200 tiles), and
  // the tile count comes from the batch.  The summation order of an output is therefore a function of the batch
  // composition: a clip's values are bit-identical run to run and across batches of the same class, and differ in
  // the last bits between classes (a one-clip batch against the 28-frame bench batch) -- as they do between the
  // row-GEMM forms that are chosen by tile count (small-row form, wide form, 256-row chain tiles).  The parity
  // tests compare compositions at 1e-4 relative; nothing in the package relies on cross-composition bit equality.
  if (tiles >= 200 || nsl < 128 || pave_internal_diag_variant() == 6) return;
  // 2 048 <= K < 8 192 (the ChannelMapper's extra level behind HRNet-w48: 3x3 / stride 2 on 384 channels,
  // K = 3 456; ResNet layer3 / layer4's 3x3 on a one-clip batch): 4 parts
  // below 128 tiles (tools/conv_small_batch.py: 256 -> 256 on 3 x 50 x 84, 99 tiles, 113 -> 101 us;
  // on 6 x 50 x 84, 197 tiles, 173 -> 179: not there)
  // (K >= 8 192 on fewer than 16 tiles -- that level of a one-clip batch: 819 pixels = 7 tiles -- 32 parts)
  const int want = nsl < 512 ? 4 : (tiles < 16 ? 32 : 8);
  if (nsl < 512 && tiles >= 128) return;
  int per = ((nsl + want - 1) / want + 1) & ~1;
  int parts;
  for (;; per += 2) {   // the last part keeps >= 4 slabs (nsl and per are even: its size is even)
    parts = (nsl + per - 1) / per;
    if (nsl - (parts - 1) * per >= 4) break;
  }
  if (parts < 2) return;
  *ksplit = parts, *ks_slabs = per;
}
int pave_internal_splitk_reduce(const float* ws, int parts, long long M, int n, const float* bias,
                                const float* residual, int relu, float* out, void* stream) {
  const long long total4 = M * n / 4;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), ws, parts, total4, n / 4, bias, residual, relu,
                     out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

// Internal entries (pave_gemm_split.hip dispatches here).  kind as the kernel's KIND; the geometry
// is ignored for kind 0.
template <int PL>
static int gemm_q_dispatch(const float* a, const float* a_bias, const void* w_planes, const float* bias,
                           const float* residual, long long residual_rows, float* out, float* out2,
                           int n_split, long long M, int K, int N, int relu, int kind, int H, int W,
                           int Cin, int Ho, int Wo, int stride, void* stream, const float* a2, int n_real,
                           int ksplit, int ks_slabs) {
  const QConv g{H, W, Cin, Ho, Wo, stride,
                (kind == 1 && Cin > 16) ? (unsigned)(((1ull << 32) + (Cin >> 4) - 1) / (unsigned)(Cin >> 4)) : 0u};
  if (relu < 0 || relu > 3)
    return pave_internal_fail(PAVE_E_ARG, "gemm_q: activation 0 (none), 1 (ReLU), 2 (GELU) or 3 (sigmoid)");
  const bool narrow = N < 0;   // (grouped rows with 64-column groups: 64-wide tiles)
  if (narrow) N = -N;
  if (n_real <= 0) n_real = N;
  if (n_real > N || n_real % 4 != 0 || (out2 && n_real != N))
    return pave_internal_fail(PAVE_E_ARG, "gemm_q: n_real %% 4 == 0, n_real <= N (== N with two outputs)");
  if (ksplit > 1 && (bias || residual || out2 || a_bias || relu || ks_slabs < 4 || ks_slabs % 2 != 0 ||
                     (long long)(ksplit - 1) * ks_slabs + 4 > K / 16 || (long long)ksplit * ks_slabs < K / 16))
    return pave_internal_fail(PAVE_E_ARG, "gemm_q: split-K parts are raw partial sums of >= 4 slabs");
  if (ksplit < 1) ksplit = 1;
  const QOut os{out2, out2 ? n_split : 0, residual_rows >= M ? 0 : (int)residual_rows, n_real,
                ksplit > 1 ? ks_slabs : 0};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint16_t* w = static_cast<const uint16_t*>(w_planes);
  if (K % 32 != 0 || K < 64) return pave_internal_fail(PAVE_E_ARG, "gemm_q: K %% 32 == 0, K >= 64");
  // the epilogue reads a row-periodic table / writes a tile's rows through 32-bit buffer offsets
  if ((os.res_rows != 0 && (long long)os.res_rows * n_real * 4 >= (1ll << 31)) || (long long)N * 512 >= (1ll << 31))
    return pave_internal_fail(PAVE_E_ARG, "gemm_q: row-periodic table of >= 2 GiB, or rows of >= 16 MiB");
  // the row forms address a tile's A bytes with a 32-bit lane offset: (rows - 1) * row length * 4 + 64
  if (kind != 1 && K >= (1 << 23))
    return pave_internal_fail(PAVE_E_ARG, "gemm_q: K < 2^23 (32-bit lane offsets inside a 128-row tile)");
  if (a_bias && (kind != 0 || K > 8192))
    return pave_internal_fail(PAVE_E_ARG, "gemm_q: a_bias with the plain row form, K <= 8192");
  if (kind == 1 && Cin % 16 != 0) return pave_internal_fail(PAVE_E_ARG, "gemm_q: 3x3 form needs Cin %% 16 == 0");
  if (kind == 4 && (!a2 || Cin <= 0 || Cin >= K || Cin % 16 != 0))
    return pave_internal_fail(PAVE_E_ARG, "gemm_q: two-source rows need a2 and 0 < K1 < K, K1 %% 16 == 0");
  // few rows (the decoders' / heads' Linears): the small-row form -- a wave per 32-row tile, no LDS, no
  // barrier; the shipped selection only (any diag variant keeps the tile kernels, which the
  // form-equality tests compare it with)
  // Narrow outputs only (N <= 512: the 256-wide Linears in front of a LayerNorm, the heads' 2 .. 30
  // output columns): wider ones re-read the A rows once per 32-column tile and measured SLOWER than the
  // tile kernels (64- and 128-column wave tiles: 1200 x 256 x 10080 45 -> 73 us, the grouped branch MLPs
  // 38 -> 76 us), so they keep those.  Measured with it: 1200 x 1024 x 256 + LayerNorm 62 -> 35 us,
  // 1200 x 256 x 256 + LayerNorm 24 -> 19 us.
  // ... and, from round 6 on, with the K axis split over the block's four waves wherever the launch is at most
  // kSkTiles 32 x 32 tiles, at ANY output width (diag variant 19: the forms as they were, for A/B and the tests
  // that compare the one-wave form with the tile kernels bit for bit)
  if (kind == 0 && small_rows_ksplit_form(M, N, K) && !a_bias && !out2 && ksplit == 1 &&
      pave_internal_diag_variant() == 0)
    return launch_sk<PL>(a, w, bias, residual, out, M, K, N, relu, H > 0 ? H : K, W > 0 ? W : 0, os.res_rows,
                         n_real, st);
  if (kind == 0 && small_rows_form(M, N) && N <= 512 && !a_bias && !out2 && ksplit == 1 &&
      (pave_internal_diag_variant() == 0 || pave_internal_diag_variant() == 19))
    return launch_s<1, 4, PL>(a, w, bias, residual, out, M, K, N, relu, H > 0 ? H : K, W > 0 ? W : 0, os.res_rows,
                          n_real, st);
  // 3x3 form: buffer-addressed below 4 GiB of map (a lane's byte offset is 32 bits wide)
  // (diag variant 5: the 64-bit lane-address form everywhere, for A/B)
  const bool big3 = kind == 1 && ((M / ((long long)Ho * Wo)) * H * W * Cin * 4 >= (1ll << 32) - 65536 ||
                                  pave_internal_diag_variant() == 5);
#define PAVE_QGO(TN_)                                                                               \
  if (kind == 0) {                                                                                  \
    if (a_bias) return launch_q<TN_, 0, true, 1, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os); \
    return launch_q<TN_, 0, false, 1, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os, nullptr, ksplit); \
  }                                                                                                 \
  if (kind == 1 && big3) return launch_q<TN_, 2, false, 1, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os, nullptr, ksplit); \
  if (kind == 1) return launch_q<TN_, 1, false, 1, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os, nullptr, ksplit); \
  if (kind == 4) return launch_q<TN_, 4, false, 1, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os, a2, ksplit); \
  return launch_q<TN_, 3, false, 1, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os, nullptr, ksplit)
  // the wide form (32 x 256 per wave): whole 256-column tiles, about one tile per block slot
  // of the chip (2 blocks x 256 CUs) -- below that the narrow form's twice as many tiles fill
  // the CUs better
  const int dv = pave_internal_diag_variant();
  const long long wtiles = ((M + QBM - 1) / QBM) * (N / 256) * ksplit;
  if (N % 256 == 0 && n_real == N && !narrow && !a_bias && dv != 8 && (wtiles >= 400 || dv == 7) &&
      !(kind == 0 && W > 0 && W % 256 != 0) && (!out2 || n_split % 256 == 0)) {
    if (kind == 0) return launch_w<0, PL>(a, w, bias, residual, out, M, K, N, relu, st, g, os, nullptr, ksplit);
    if (kind == 1 && big3) return launch_w<2, PL>(a, w, bias, residual, out, M, K, N, relu, st, g, os, nullptr, ksplit);
    if (kind == 1) return launch_w<1, PL>(a, w, bias, residual, out, M, K, N, relu, st, g, os, nullptr, ksplit);
    if (kind == 4) return launch_w<4, PL>(a, w, bias, residual, out, M, K, N, relu, st, g, os, a2, ksplit);
    return launch_w<3, PL>(a, w, bias, residual, out, M, K, N, relu, st, g, os, nullptr, ksplit);
  }
  if (kind == 0 && N % 256 == 128 && N >= 384 && n_real == N && !narrow && !a_bias && dv != 8 && W == 0 && ksplit == 1 &&
      (((M + QBM - 1) / QBM) * (N / 256 + 1) >= 400 || dv == 7) && (!out2 || n_split % 256 == 0)) {
    const long long gx = ((M + QBM - 1) / QBM) * (N / 256 + 1);
    if (gx >= (1ll << 31)) return pave_internal_fail(PAVE_E_ARG, "gemm_wn: grid too large");
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wn_kernel<PL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, wn_smem<PL>()) != hipSuccess)
        return pave_internal_fail(PAVE_E_LAUNCH, "gemm_wn: cannot raise dynamic LDS limit");
      attr_set = true;
    }
    hipLaunchKernelGGL(gemm_wn_kernel<PL>, dim3((unsigned)gx), dim3(256), wn_smem<PL>(), st, a, w, bias, residual, out,
                       (int)M, K, N, relu, g, os);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
    return PAVE_OK;
  }
  // 65..96 real outputs in 128-row planes (HRNet's 96-channel branch): three column tiles instead of four
  if (N == 128 && n_real <= 96 && n_real > 64 && !narrow && !out2 && pave_internal_diag_variant() != 8 &&
      ksplit == 1 && (kind == 1 || kind == 0) && !a_bias) {
    if (kind == 1 && big3) return launch_q<3, 2, false, 1, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os);
    if (kind == 1 && dv != 15 && (((M + 255) / 256) >= 1024 || dv == 16) && (os.res_rows == 0 || os.res_rows >= 64))
      return launch_q<3, 1, false, 2, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os);
    if (kind == 1) return launch_q<3, 1, false, 1, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os);
    return launch_q<3, 0, false, 1, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os);
  }
  if (N % 128 == 0 && !narrow) { PAVE_QGO(4); }
  // 64-column tiles with two row tiles per wave (256-row blocks): only on request (diag variant 16).  Measured
  // at 28 x 200 x 336 pixels (tools/rm_ab.py, profiles/r05_rm_ab.txt): 3x3 48 -> 48 + identity 628 -> 653 us,
  // 64 -> 64 801 -> 817, conv1 256 -> 64 470 -> 478 -- as launches of their own these forms lose more from two
  // blocks per CU (196 registers) instead of three than they gain from sharing the W fragments; the 96-column
  // 3x3 (two blocks per CU either way: 442 -> 428 us) and the layer1 chain (-3 ... -4 %) take the form.
  if (N % 64 == 0 && (kind == 0 || (kind == 1 && !big3)) && !a_bias && ksplit == 1 && dv == 16 &&
      (os.res_rows == 0 || os.res_rows >= 64) && (long long)N * 1024 < (1ll << 31)) {
    if (kind == 1) return launch_q<2, 1, false, 2, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os);
    return launch_q<2, 0, false, 2, PL>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os);
  }
  // 33 .. 48 real outputs in 64-row planes (HRNet-w48's 48-channel 3x3 branch, 154 launches of a T = 7 x 4 step):
  // the 16-column tail on v_mfma_f32_16x16x32_bf16 over slab pairs instead of zero-padded 32x32x16 products
  // (HT above).  The 3x3 form only: it takes the tile kernels at every size, so a clip's values do not depend on
  // the batch it is in.  (diag variant 17: never -- the padded form, for A/B and the tolerance test)
  if constexpr (PL == 3) {
    if (N == 64 && n_real > 32 && n_real <= 48 && kind == 1 && !big3 && !a_bias && !out2 && ksplit == 1 &&
        dv != 17 && dv != 15 && dv != 16)   // (15 / 16: the A/B of the padded one- / two-row-tile forms)
      return launch_q<2, 1, false, 1, 3, true>(a, w, bias, residual, out, M, K, N, relu, a_bias, st, g, os);
  }
  if (N % 64 == 0) { PAVE_QGO(2); }
#undef PAVE_QGO
  return pave_internal_fail(PAVE_E_ARG, "gemm_q: N %% 64 == 0 required");
}

int pave_internal_gemm_q(const float* a, const float* a_bias, const void* w_planes, const float* bias,
                         const float* residual, long long residual_rows, float* out, float* out2,
                         int n_split, long long M, int K, int N, int relu, int kind, int H, int W,
                         int Cin, int Ho, int Wo, int stride, void* stream, const float* a2, int n_real,
                         int ksplit, int ks_slabs, int planes) {
  // planes: 3 = the exact bf16 split, 1 = one plane of fp16 operands (PAVE_PLANES_FP16 at the C ABI)
  if (planes == 1)
    return gemm_q_dispatch<1>(a, a_bias, w_planes, bias, residual, residual_rows, out, out2, n_split, M, K, N, relu,
                              kind, H, W, Cin, Ho, Wo, stride, stream, a2, n_real, ksplit, ks_slabs);
  if (planes != 3) return pave_internal_fail(PAVE_E_ARG, "gemm_q: 3 bf16 planes or 1 fp16 plane");
  return gemm_q_dispatch<3>(a, a_bias, w_planes, bias, residual, residual_rows, out, out2, n_split, M, K, N, relu,
                            kind, H, W, Cin, Ho, Wo, stride, stream, a2, n_real, ksplit, ks_slabs);
}

// merged encoder projection with the sampler's softmax / location arithmetic in the epilogue
template <int PL>
static int gemm_encproj_go(const float* a, const void* w_planes, const float* table, long long table_rows,
                           const float* value_bias, const float* ref, const int* levels_hw, float* value,
                           float* samp, long long M, int K, void* stream) {
  if (K % 32 != 0 || K < 64 || K >= (1 << 23)) return pave_internal_fail(PAVE_E_ARG, "gemm_encproj: K %% 32 == 0, 64 <= K < 2^23");
  QEpi epi{};
  epi.ref = ref;
  for (int l = 0; l < 4; ++l) {
    if (levels_hw[2 * l] <= 0 || levels_hw[2 * l + 1] <= 0)
      return pave_internal_fail(PAVE_E_ARG, "gemm_encproj: bad level size");
    epi.fH[l] = (float)levels_hw[2 * l], epi.fW[l] = (float)levels_hw[2 * l + 1];
    epi.rH[l] = 1.f / (float)levels_hw[2 * l], epi.rW[l] = 1.f / (float)levels_hw[2 * l + 1];
  }
  if (table_rows < M && table_rows * 640 * 4 >= (1ll << 31))
    return pave_internal_fail(PAVE_E_ARG, "gemm_encproj: row-periodic table of >= 2 GiB");
  const QOut os{samp, 256, table_rows >= M ? 0 : (int)table_rows, 640, 0};
  const long long gx = ((M + QBM - 1) / QBM) * 3;
  if (gx >= (1ll << 31)) return pave_internal_fail(PAVE_E_ARG, "gemm_encproj: grid too large");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wn_enc_kernel<PL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, wn_smem<PL>()) != hipSuccess)
      return pave_internal_fail(PAVE_E_LAUNCH, "gemm_encproj: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_wn_enc_kernel<PL>, dim3((unsigned)gx), dim3(256), wn_smem<PL>(),
                     reinterpret_cast<hipStream_t>(stream), a, static_cast<const uint16_t*>(w_planes), table,
                     value_bias, value, (int)M, K, os, epi);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}
int pave_internal_gemm_encproj(const float* a, const void* w_planes, const float* table, long long table_rows,
                               const float* value_bias, const float* ref, const int* levels_hw, float* value,
                               float* samp, long long M, int K, void* stream, int planes) {
  if (planes == 1)
    return gemm_encproj_go<1>(a, w_planes, table, table_rows, value_bias, ref, levels_hw, value, samp, M, K, stream);
  if (planes != 3) return pave_internal_fail(PAVE_E_ARG, "gemm_encproj: 3 bf16 planes or 1 fp16 plane");
  return gemm_encproj_go<3>(a, w_planes, table, table_rows, value_bias, ref, levels_hw, value, samp, M, K, stream);
}

template <int PL>
static int gemm_q_ln_go(const float* a, const void* w_planes, const float* bias, const float* residual,
                        const float* gamma, const float* beta, float eps, float* out, long long M,
                        int K, int N, void* stream) {
  if (K % 32 != 0 || K < 64 || N != 256)
    return pave_internal_fail(PAVE_E_UNSUPPORTED, "gemm_q_ln: K %% 32 == 0, K >= 64 and N == 256 required");
  if (small_rows_form(M, N) && (pave_internal_diag_variant() == 0 || pave_internal_diag_variant() == 19)) {
    // few rows: the small-row GEMM (bias + identity in its epilogue), then LayerNorm in place -- two
    // launches of a few microseconds instead of 10 row tiles walking K behind barriers
    const int st1 = (small_rows_ksplit_form(M, N, K) && pave_internal_diag_variant() == 0)
        ? launch_sk<PL>(a, static_cast<const uint16_t*>(w_planes), bias, residual, out, M, K, N, 0, K, 0, 0, N,
                        reinterpret_cast<hipStream_t>(stream))
        : launch_s<1, 4, PL>(a, static_cast<const uint16_t*>(w_planes), bias, residual, out, M, K, N, 0, K,
                                   0, 0, N, reinterpret_cast<hipStream_t>(stream));
    if (st1 != PAVE_OK) return st1;
    return pave_bias_add_layernorm_f32(out, nullptr, nullptr, gamma, beta, out, M, N, eps, stream);
  }
  constexpr int STAGE = QBM * 64 + PL * 256 * 32;
  constexpr int smem = QNS * STAGE + 2 * QBM * 2 * 4;
  auto kern = gemm_q_ln_kernel<PL>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return pave_internal_fail(PAVE_E_LAUNCH, "gemm_q_ln: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  const long long gx = (M + QBM - 1) / QBM;
  // From two tiles per CU slot on: the wide form (4 waves, a wave owns 32 whole rows; LayerNorm on the
  // accumulator layout, two blocks per CU) -- 602 -> 548 us at K = 256, 1 715 -> 1 606 us at K = 1024 for
  // 625 044 rows (tools/ln_ab.py).  Few tiles (the decoders' M = 1 200) keep the 8-wave block: twice the
  // waves per tile.  (diag variant 13: always the 8-wave form, 14: always the wide form.)  The identity
  // rows are read through a buffer resource: M * 1024 bytes < 4 GiB.
  const int dvl = pave_internal_diag_variant();
  if (dvl != 13 && (gx >= 512 || dvl == 14) && M * 1024ll < (1ll << 32)) {
    static bool attr_w = false;
    if (!attr_w) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w_ln_kernel<PL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, w_smem<PL>()) != hipSuccess)
        return pave_internal_fail(PAVE_E_LAUNCH, "gemm_w_ln: cannot raise dynamic LDS limit");
      attr_w = true;
    }
    hipLaunchKernelGGL(gemm_w_ln_kernel<PL>, dim3((unsigned)gx), dim3(256), w_smem<PL>(),
                       reinterpret_cast<hipStream_t>(stream), a, static_cast<const uint16_t*>(w_planes),
                       bias, residual, out, (int)M, K, N, QLn{gamma, beta, eps});
    const hipError_t ew = hipGetLastError();
    if (ew != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(ew));
    return PAVE_OK;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), smem,
                     reinterpret_cast<hipStream_t>(stream), a, static_cast<const uint16_t*>(w_planes),
                     bias, residual, out, (int)M, K, N, QLn{gamma, beta, eps});
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}
int pave_internal_gemm_q_ln(const float* a, const void* w_planes, const float* bias, const float* residual,
                            const float* gamma, const float* beta, float eps, float* out, long long M,
                            int K, int N, void* stream, int planes) {
  if (planes == 1) return gemm_q_ln_go<1>(a, w_planes, bias, residual, gamma, beta, eps, out, M, K, N, stream);
  if (planes != 3) return pave_internal_fail(PAVE_E_ARG, "gemm_q_ln: 3 bf16 planes or 1 fp16 plane");
  return gemm_q_ln_go<3>(a, w_planes, bias, residual, gamma, beta, eps, out, M, K, N, stream);
}

// fp16 mode with fp16 ACTIVATIONS around a launch (wide tile forms only): a_f16 -- the A rows are fp16 [M, K];
// gamma == nullptr: out = act(a W^T + bias) as fp32 or (out_f16) fp16 [M, N], N % 256 == 0;  gamma != nullptr:
// out = LayerNorm(a W^T + bias + residual) fp32, N == 256 (the accumulator-layout LayerNorm form).
int pave_internal_gemm_f16act(const void* a, int a_f16, const void* w_plane, const float* bias, const float* residual,
                              const float* gamma, const float* beta, float eps, void* out, int out_f16, long long M,
                              int K, int N, int relu, void* stream) {
  if (K % 32 != 0 || K < 64 || N % 256 != 0 || K >= (1 << 23))
    return pave_internal_fail(PAVE_E_ARG, "gemm_fp16_act: K %% 32 == 0 (64 <= K < 2^23), N %% 256 == 0");
  if (gamma && (N != 256 || out_f16 || !beta || M * 1024ll >= (1ll << 32)))
    return pave_internal_fail(PAVE_E_ARG, "gemm_fp16_act: the LayerNorm form has N == 256, fp32 output, M < 2^22");
  if (!gamma && (residual || relu < 0 || relu > 2))
    return pave_internal_fail(PAVE_E_ARG, "gemm_fp16_act: the plain form takes bias + activation only");
  if ((long long)N * 512 >= (1ll << 31)) return pave_internal_fail(PAVE_E_ARG, "gemm_fp16_act: rows of >= 16 MiB");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint16_t* w = static_cast<const uint16_t*>(w_plane);
  const float* af = static_cast<const float*>(a);
  float* of = static_cast<float*>(out);
  const QConv g0{0, 0, 0, 0, 0, 0, 0u};
  const long long gx = ((M + QBM - 1) / QBM) * (N / 256);
  if (gx >= (1ll << 31)) return pave_internal_fail(PAVE_E_ARG, "gemm_fp16_act: grid too large");
#define PAVE_F16_GO(KERN, ...)                                                                        \
  {                                                                                                   \
    static bool attr = false;                                                                         \
    if (!attr) {                                                                                      \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              w_smem<1>()) != hipSuccess)                                             \
        return pave_internal_fail(PAVE_E_LAUNCH, "gemm_fp16_act: cannot raise dynamic LDS limit");    \
      attr = true;                                                                                    \
    }                                                                                                 \
    hipLaunchKernelGGL(KERN, dim3((unsigned)gx), dim3(256), w_smem<1>(), st, __VA_ARGS__);            \
  }
  if (gamma) {
    const QLn ln{gamma, beta, eps};
    if (a_f16) PAVE_F16_GO((gemm_w_ln_kernel<1, 1>), af, w, bias, residual, of, (int)M, K, N, ln)
    else PAVE_F16_GO((gemm_w_ln_kernel<1, 0>), af, w, bias, residual, of, (int)M, K, N, ln)
  } else {
    const QOut os{nullptr, 0, 0, N, 0};
    const int io = (a_f16 ? 1 : 0) | (out_f16 ? 2 : 0);
    if (io == 0) PAVE_F16_GO((gemm_w_kernel<0, 1, 0>), af, w, bias, nullptr, of, (int)M, K, N, relu, g0, os, nullptr)
    else if (io == 1) PAVE_F16_GO((gemm_w_kernel<0, 1, 1>), af, w, bias, nullptr, of, (int)M, K, N, relu, g0, os, nullptr)
    else if (io == 2) PAVE_F16_GO((gemm_w_kernel<0, 1, 2>), af, w, bias, nullptr, of, (int)M, K, N, relu, g0, os, nullptr)
    else PAVE_F16_GO((gemm_w_kernel<0, 1, 3>), af, w, bias, nullptr, of, (int)M, K, N, relu, g0, os, nullptr)
  }
#undef PAVE_F16_GO
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

template <bool HAS_A, int KIND2, int CN, int RMC = 1, int PL = 3>
static int launch_chain(const ChainArgs& p, hipStream_t st) {
  auto kern = bottleneck_chain_kernel<HAS_A, KIND2, CN, RMC, PL>;
  // (the 256-row 64-column body: ring of 3 x 22 KiB)
  constexpr int smem = RMC == 2 ? (3 * (2 * QBM * 64 + 3 * 64 * 32) > W_SMEM ? 3 * (2 * QBM * 64 + 3 * 64 * 32) : W_SMEM) : W_SMEM;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return pave_internal_fail(PAVE_E_LAUNCH, "bottleneck_chain: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  constexpr int CBM = QBM * RMC;
  hipLaunchKernelGGL(kern, dim3((unsigned)((p.M + CBM - 1) / CBM)), dim3(256), smem, st, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}

template <int PL>
static int bottleneck_chain_go(const float* c1, const void* w2_planes, const float* b2, float* c2,
                               const void* w3_planes, const float* b3, const float* residual,
                               const float* a2, int k2, float* out, const void* w1n_planes,
                               const float* b1n, float* c1n, int cn, int N, int H, int W,
                               void* stream) {
  if (!c2 || !w3_planes || !out || (c1 != nullptr) != (w2_planes != nullptr))
    return pave_internal_fail(PAVE_E_ARG, "bottleneck_chain: null pointer (c1 and w2_planes: both or neither)");
  // (the 3x3 body addresses the c1 map through a buffer resource: below 4 GiB)
  if (N <= 0 || H <= 0 || W <= 0 || (long long)N * H * W >= (1ll << 31) ||
      (c1 && (long long)N * H * W * 64 * 4 >= (1ll << 32) - 65536))
    return pave_internal_fail(PAVE_E_ARG, "bottleneck_chain: bad sizes (N*H*W < 2^31; with the 3x3 inside "
                                          "the launch the c1 map must stay below 4 GiB)");
  if ((a2 != nullptr) != (k2 > 0) || (a2 && (k2 % 32 != 0 || k2 > 960)) || (a2 && residual))
    return pave_internal_fail(PAVE_E_ARG, "bottleneck_chain: a2 [M, k2] (k2 %% 32 == 0) replaces the residual");
  if ((w1n_planes != nullptr) != (cn > 0) || (w1n_planes && (!c1n || (cn != 64 && cn != 128))))
    return pave_internal_fail(PAVE_E_ARG, "bottleneck_chain: next conv1 with 64 or 128 output channels, or none");
  if (pave_internal_diag_variant() == 9)
    return pave_internal_fail(PAVE_E_UNSUPPORTED, "bottleneck_chain: LDS-DMA generation only");
  ChainArgs p{c1, static_cast<const uint16_t*>(w2_planes), b2, c2,
              static_cast<const uint16_t*>(w3_planes), b3, residual, a2, out,
              static_cast<const uint16_t*>(w1n_planes), b1n, c1n,
              N * H * W, H, W, 64 + k2, 64};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define PAVE_CHAIN_GO(HA)                                              \
  if (a2) {                                                            \
    if (cn == 64) return launch_chain<HA, 4, 64, 1, PL>(p, st);               \
    if (cn == 128) return launch_chain<HA, 4, 128, 1, PL>(p, st);             \
    return launch_chain<HA, 4, 0, 1, PL>(p, st);                              \
  }                                                                    \
  if (cn == 64) return launch_chain<HA, 0, 64, 1, PL>(p, st);                 \
  if (cn == 128) return launch_chain<HA, 0, 128, 1, PL>(p, st);               \
  return launch_chain<HA, 0, 0, 1, PL>(p, st)
  // from two 256-row tiles per block slot of the chip on: 256-row tiles (the 3x3 and a 64-output conv1 with two
  // row tiles per wave); diag variant 15: never, 16: always
  const int dvc = pave_internal_diag_variant();
  if (c1 && cn > 0 && dvc != 15 && (p.M >= 256 * 1024 || dvc == 16)) {
    if (a2) {
      if (cn == 64) return launch_chain<true, 4, 64, 2, PL>(p, st);
      return launch_chain<true, 4, 128, 2, PL>(p, st);
    }
    if (cn == 64) return launch_chain<true, 0, 64, 2, PL>(p, st);
    return launch_chain<true, 0, 128, 2, PL>(p, st);
  }
  if (c1) { PAVE_CHAIN_GO(true); }
  PAVE_CHAIN_GO(false);
#undef PAVE_CHAIN_GO
}

#ifdef PAVE_DIAG
// In-kernel clock counters of the split GEMM class (-DPAVE_DIAG build only; not part of the C ABI).
// pave_diag_clock_read: out[2 k] = shader-clock ticks, out[2 k + 1] = 100 MHz ticks of kernel kind k
// (0 gemm_q, 1 gemm_w, 2 gemm_wn, 3 gemm_wn_enc, 4 bottleneck_chain, 5 gemm_q_ln, 6 gemm_w_ln, 7 stem);
// both synchronise the device.
extern "C" int pave_diag_clock_reset(void) {
  unsigned long long z[16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_clock_acc), z, sizeof(z)) == hipSuccess ? PAVE_OK : PAVE_E_LAUNCH;
}
extern "C" int pave_diag_set_stagger(int v) {
  if (hipDeviceSynchronize() != hipSuccess) return PAVE_E_LAUNCH;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_stagger), &v, sizeof(v)) == hipSuccess ? PAVE_OK : PAVE_E_LAUNCH;
}
extern "C" int pave_diag_hwid_read(unsigned* out1024) {
  if (!out1024) return PAVE_E_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return PAVE_E_LAUNCH;
  return hipMemcpyFromSymbol(out1024, HIP_SYMBOL(g_diag_hwid), 1024 * sizeof(unsigned)) == hipSuccess
             ? PAVE_OK : PAVE_E_LAUNCH;
}
extern "C" int pave_diag_clock_read(unsigned long long* out16) {
  if (!out16) return PAVE_E_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return PAVE_E_LAUNCH;
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_clock_acc), 16 * sizeof(unsigned long long)) == hipSuccess
             ? PAVE_OK : PAVE_E_LAUNCH;
}
#endif

extern "C" int pave_bottleneck_chain_f32(const float* c1, const void* w2_planes, const float* b2, float* c2,
                                         const void* w3_planes, const float* b3, const float* residual,
                                         const float* a2, int k2, float* out, const void* w1n_planes,
                                         const float* b1n, float* c1n, int cn, int N, int H, int W,
                                         int nplanes, void* stream) {
  if (nplanes == PAVE_PLANES_FP16)
    return bottleneck_chain_go<1>(c1, w2_planes, b2, c2, w3_planes, b3, residual, a2, k2, out, w1n_planes, b1n, c1n,
                                  cn, N, H, W, stream);
  if (nplanes != 3) return pave_internal_fail(PAVE_E_ARG, "bottleneck_chain: nplanes must be 3 or PAVE_PLANES_FP16");
  return bottleneck_chain_go<3>(c1, w2_planes, b2, c2, w3_planes, b3, residual, a2, k2, out, w1n_planes, b1n, c1n,
                                cn, N, H, W, stream);
}

// Stem: w_stem = the 11-slab (c, ky, kx' = kx + 1) planes [11][3][64][16]; requires pitch % 4 == 0 and
// a 16-byte aligned image base (LDS-DMA chunks); the caller falls back to the first-generation
// kernel otherwise.
int pave_internal_stem7x7_q(const float* x, const void* w_stem, const float* bias, float* y, int N,
                            int H, int W, int pitch, int relu, void* stream, int planes) {
  // pitch > W: rows at a 16-byte aligned pitch with ZERO pad columns (pave_repitch_rows_f32) -- the kernel
  // addresses and range-checks its window chunks against the pitch, the stored columns are those of the real W
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long gx = (long long)N * Ho * ((Wo + SBM - 1) / SBM);
  if (gx >= (1ll << 31)) return pave_internal_fail(PAVE_E_ARG, "stem7x7: grid too large");
#ifndef PAVE_STEM_ROWS
#define PAVE_STEM_ROWS 2
#endif
  constexpr int R = PAVE_STEM_ROWS;
  const long long g2 = (long long)N * ((Ho + R - 1) / R) * ((Wo + SBM - 1) / SBM);
  if (planes == 1)
    hipLaunchKernelGGL((stem7x7_qr_kernel<R, 1>), dim3((unsigned)g2), dim3(192), StemRows<R>::WIN,
                       reinterpret_cast<hipStream_t>(stream), x, static_cast<const uint16_t*>(w_stem), bias,
                       y, H, pitch, Ho, Wo, relu);
  else
    hipLaunchKernelGGL((stem7x7_qr_kernel<R, 3>), dim3((unsigned)g2), dim3(192), StemRows<R>::WIN,
                       reinterpret_cast<hipStream_t>(stream), x, static_cast<const uint16_t*>(w_stem), bias,
                       y, H, pitch, Ho, Wo, relu);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pave_internal_fail(PAVE_E_LAUNCH, hipGetErrorString(e));
  return PAVE_OK;
}
