/*
 * pave_hip.h -- C ABI of libpave_hip.so: the MI355X (gfx950) device side of the
 * PAVE-Net forward hot path.
 *
 * Every entry point takes plain device pointers + sizes + a hipStream_t passed
 * as void* (no torch / ATen types), returns 0 on success and a negative
 * PAVE_E_* code on failure (pave_last_error() has the message), launches on the
 * given stream and never synchronises or allocates.
 *
 * Reference interfaces these replace (paths relative to zgspose/PAVENet):
 *   [R1] third_party/mmcv/mmcv/ops/csrc/pytorch/pybind.cpp:160-173,737-748
 *        Tensor ms_deform_attn_forward(value, spatial_shapes, level_start_index,
 *                                      sampling_loc, attn_weight, im2col_step)
 *        -> third_party/mmcv/mmcv/ops/csrc/pytorch/cuda/ms_deform_attn_cuda.cu:209-277
 *        -> kernel third_party/mmcv/mmcv/ops/csrc/common/cuda/ms_deform_attn_cuda_kernel.cuh:200-254
 *   [R2] third_party/mmcv/mmcv/ops/multi_scale_deform_attn.py:305-412
 *        MultiScaleDeformableAttention.forward: softmax over L*P, loc = ref + off/(W,H),
 *        then [R1]  (encoder self-attention, hot loop #1)
 *   [R3] opera/models/utils/transformer.py:1644-1863 (T=3) / 2738-3117 (T=5)
 *        MulFramesMultiScaleDeformablePoseAttentionNumFrames{3,5}.forward: per-frame
 *        softmax + Z_t re-weighting (== joint softmax over T*L*K), pose-box scaled
 *        offsets, T x [R1]  (pose decoder, hot loop #2)
 *   [R4] third_party/mmcv/mmcv/ops/multi_scale_deform_attn.py:1388-1587 (T=3) / 1590-1981 (T=5)
 *        MulFramesMultiScaleDeformableAttentionNumFrames{3,5}.forward: same fusion with
 *        grid offsets, T x [R1]  (joint decoder, hot loop #3)
 */
#ifndef PAVE_HIP_H_
#define PAVE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAVE_OK 0
#define PAVE_E_ARG (-1)     /* bad argument (null pointer, non-positive size, unsupported shape) */
#define PAVE_E_LAUNCH (-2)  /* hipLaunchKernel reported an error */
#define PAVE_E_STEP (-3)    /* batch %% im2col_step != 0, as the reference asserts */
#define PAVE_E_UNSUPPORTED (-4) /* valid arguments, but not a shape this entry point's kernel covers */

/* ABI version; bumped on any signature change (pavenet_amd/native.py checks it at load). */
#define PAVE_ABI_VERSION 20
int pave_abi_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char* pave_last_error(void);

/*
 * [R1] Multi-scale deformable attention, forward, fp32 / fp64.
 *   value          [bs, S, M, D]
 *   spatial_shapes [L, 2] int64 (h, w), DEVICE memory   (as the reference tensor)
 *   level_start    [L]    int64,        DEVICE memory
 *   sampling_loc   [bs, Lq, M, L, P, 2] (x, y) normalised to [0,1]
 *   attn_weight    [bs, Lq, M, L, P]
 *   out            [bs, Lq, M*D]  (fully overwritten; need not be zeroed)
 * im2col_step only reproduces the reference's divisibility check
 * (ms_deform_attn_cuda.cu:242-245): min(bs, step) must divide bs.
 * Two kernels behind it: D %% 4 == 0 fp32 takes msda_fwd_vec_kernel (G lanes x 4 channels per (query, head),
 * float4 corner loads); every other case -- fp64, D not a multiple of 4 (the reference's gradcheck list has 30, 71
 * and 1025) -- takes msda_fwd_scalar_kernel, the COMPATIBILITY FALLBACK of this entry point: one thread per output
 * element, i.e. the reference kernel's own work decomposition (ms_deform_attn_cuda_kernel.cuh:200-254), kept for
 * coverage of the drop-in signature and not a CDNA4 design.  The PAVE-Net forward path never calls either: it runs
 * on the fused launches below (pave_enc_deform_attn_tile_f32, pave_deform_attn_*_fused_f32).
 */
int pave_ms_deform_attn_forward_f32(const float* value, const int64_t* spatial_shapes,
                                    const int64_t* level_start, const float* sampling_loc,
                                    const float* attn_weight, float* out, int bs, int S, int M,
                                    int D, int L, int Lq, int P, int im2col_step, void* stream);
int pave_ms_deform_attn_forward_f64(const double* value, const int64_t* spatial_shapes,
                                    const int64_t* level_start, const double* sampling_loc,
                                    const double* attn_weight, double* out, int bs, int S, int M,
                                    int D, int L, int Lq, int P, int im2col_step, void* stream);

/*
 * Fused T-frame deformable attention (M = 8 heads x D = 32 channels).
 *
 * One launch replaces, for all T frames: softmax / Z_t arithmetic, sampling
 * location arithmetic, T calls of [R1] and the cross-frame fusion.  The softmax
 * is the numerically stabilised joint softmax over all T*L*P logits of a
 * (unit, head), which is algebraically what [R3]/[R4] compute with an
 * un-stabilised exp.
 *
 * "unit" = one query row: encoder token (b, q) / pose query (b, q) / joint query (n, k).
 *
 *   value      [n_clips*T, S, 8, 32]  projected (and padding-masked) memory; frame t of
 *                                     clip c is slab c*T + t
 *   proj       [n_units, proj_stride] raw outputs of the offset / logit Linears for one unit:
 *                offsets at [t][m][l][p][2] starting at column 0,
 *                logits  at [t][m][l][p]    starting at column T*8*L*P*2
 *   unit_clip  [n_units] int32 clip index of each unit, or NULL => unit / units_per_clip
 *   order      [n_units] int32 permutation giving the processing order of units (XCD/L2
 *              locality), or NULL => identity
 *   frame_table [n_clips*T] int32, or NULL: the value slab of frame t of clip c is
 *              frame_table[c*T + t] instead of c*T + t -- `value` is then a per-frame cache
 *              [n_cached_frames, S, 8, 32] shared by overlapping clips (streaming windows of a video,
 *              opera/datasets/posetrack_video_pose.py:578-623: no per-window copy / re-projection).
 *              n_slabs = n_cached_frames (> 0 with a table; 0 or n_clips*T without): the entries are slab
 *              indices read on the device; every slab index is CLAMPED into [0, n_slabs) there, so a bad
 *              table reads a wrong frame of `value`, never memory outside it (a correct result still needs
 *              0 <= frame_table[i] < n_slabs: pavenet_amd/streaming.py checks the host list the table is
 *              made from, FrameSlabs.covers)
 *   out        [n_units, 256]
 *   stat_max, stat_sum  [n_units, 8] or NULL: per-head max logit and sum(exp(logit-max)) over the
 *              frames this call saw (for merging frame-sharded partial results)
 *
 * GRID form ([R2] with T = 1, [R4]): L = 4 levels x P = 4 points;
 *   ref [T, n_units, L, 2] (already multiplied by valid ratios);  loc = ref + off / (W_l, H_l)
 * POSE form ([R3]): L levels x P = K keypoints (K <= 24);
 *   ref [n_clips, T, Q, L, 2K] with n_units = n_clips*Q;  wh = clamp(max-min over K, 1e-4);
 *   loc = ref_k + off * wh * 0.5
 * ref_levels = L, or 1: ref is [T, n_units, 1, 2] / [n_clips, T, Q, 1, 2K] and every level reads the same
 *   row (un-padded batches: all valid ratios are 1, the reference broadcasts `reference_points[:, :, None]`
 *   over the levels, OT:6712-6720 / MT:845-856 -- no materialised copy).
 */
int pave_deform_attn_grid_fused_f32(const float* value, const int64_t* spatial_shapes,
                                    const int64_t* level_start, const float* proj,
                                    const float* ref, const int32_t* unit_clip,
                                    const int32_t* order, float* out, float* stat_max,
                                    float* stat_sum, int n_units, int units_per_clip, int n_clips,
                                    int T, int S, int L, int P, int proj_stride,
                                    const int32_t* frame_table, int n_slabs, int ref_levels, void* stream);

int pave_deform_attn_pose_fused_f32(const float* value, const int64_t* spatial_shapes,
                                    const int64_t* level_start, const float* proj,
                                    const float* ref, float* out, float* stat_max,
                                    float* stat_sum, int n_clips, int Q, int T, int S, int L,
                                    int K, int proj_stride, const int32_t* frame_table, int n_slabs,
                                    int ref_levels, void* stream);

/*
 * The encoder layer's merged projection (value_proj | sampling_offsets | attention_weights as ONE
 * N = 640 GEMM, third_party/mmcv/mmcv/ops/multi_scale_deform_attn.py:357-384) with the sampler's
 * per-(query, head) arithmetic done in the GEMM epilogue:
 *   a [M, K] fp32;  w_planes = the 3-plane split of the [640, K] row-concatenated weight;
 *   table [table_rows, 640]: row m adds table[m % table_rows] (bias + positional term);
 *   value_bias: NULL, or 256 floats added to the value columns INSTEAD of the table's first 256
 *     columns, which are then not read (value_proj has no positional term: its table columns are
 *     one bias row repeated);
 *   ref [M, 4, 2] reference points (normalised x, y per level);  levels_hw [4][2] (h, w), HOST memory
 *   -> value [M, 256];  samp [M, 384] = level PIXEL coordinates [8 heads][4 levels][4 points][x, y]
 *      ((ref + off / (w, h)) * (w, h) - 0.5), then softmaxed attention weights [8][16]
 * i.e. `proj` of pave_enc_deform_attn_tile_f32 in its prepared form (variant | 4): same bits as
 * feeding that kernel the raw projections (one definition of the arithmetic, csrc/pave_enc_math.h).
 * nplanes: 3 or PAVE_PLANES_FP16 (below).
 */
int pave_gemm_bf16x3_encproj_f32(const float* a, const void* w_planes, const float* table,
                                 long long table_rows, const float* value_bias, const float* ref,
                                 const int* levels_hw, float* value, float* samp, long long M, int K,
                                 int nplanes, void* stream);

/*
 * HRNet stem conv1 (third_party/mmdetection/mmdet/models/backbones/hrnet.py:549-556): 3x3 / stride 2
 * / pad 1 convolution of the NCHW image batch x [N, 3, H, W], 3 -> 64 channels, + bias (the folded
 * BatchNorm) + optional ReLU -> y [N, Ho, Wo, 64] NHWC, Ho = (H - 1) / 2 + 1.
 *   w_taps [27][64]: weight[co][c][ky][kx] at w_taps[(c * 3 + ky) * 3 + kx][co]
 */
int pave_conv3x3s2_c3_nchw_f32(const float* x, const float* w_taps, const float* bias, float* y, int N,
                               int H, int W, int relu, void* stream);

/*
 * Scaled-dot-product core of the decoders' self-attention (replaces what nn.MultiheadAttention
 * runs between its in- and out-projection, third_party/mmcv/mmcv/cnn/bricks/transformer.py:
 * 406-551 as the reference's decoder layers call it: no masks, no dropout, 32 channels per head).
 *   qkv [n_seq * L, ld]: row = (sequence, position); q at column 0, k at column H*32, v at 2*H*32
 *   out [n_seq * L, H*32] = softmax(q k^T / sqrt(32)) v per (sequence, head)
 * One head's K and V ([L, 32] each) are staged in LDS: L <= 568.
 */
int pave_mha_core_f32(const float* qkv, float* out, int n_seq, int L, int H, int ld, void* stream);

/*
 * Row-wise top-k in one launch: element (r, i) of x at x[r ld + i cs], i < n; k <= 1024,
 * n <= 32768 -> index [rows, k] int64 sorted by (value descending, index ascending), values
 * [rows, k] (may be NULL).  NaN ranks above +inf (torch.topk's order).  Replaces torch.topk for the
 * proposal selection (opera/models/utils/transformer.py:21383-21385) and the score selection
 * (opera/models/dense_heads/videopose_head_mul_frames.py:1416).
 */
int pave_topk_rows_f32(const float* x, float* values, long long* index, int rows, int n, int ld, int cs,
                       int k, void* stream);

/*
 * The N selected queries of every frame in one gather (videopose_head_mul_frames.py:1419-1427, 610):
 * poses [B, T*Q, C] (frame t at rows [t Q, (t+1) Q)), index [B, N] int64 (clamped to [0, Q))
 * -> out [T, B*N, C] frame-major.
 */
int pave_gather_frame_poses_f32(const float* poses, const long long* index, float* out, int B, int T,
                                int Q, int N, int C, void* stream);

/*
 * Post-processing of the refined poses in one launch (videopose_head_mul_frames.py:1440-1490,
 * get_p :1531-1535): pixels = clamp(kpt * (w, h), 0, (w, h)) [/ scale factor], bounding box =
 * min / max over the K key points, p = 0.7 (1 - exp(-0.2 / sigma_x)) (1 - exp(-0.2 / sigma_y)),
 * kpt <- kpt p^5 / (p^5 + 1e-10), key-point score = pose score * p.
 *   kpts, sigmas [B, N, K, 2]; scores [B, N]; wh, sf [B, 2] (sf only read when rescale != 0)
 *   sigma_ld: floats between consecutive (x, y) rows of sigmas (2 = dense; 4 = the sigma branch's output as its
 *   GEMM leaves it, padded to the 4-column grid)
 *   -> det_kpts [B, N, K, 3] (x, y, score), det_bboxes [B, N, 5] (x1, y1, x2, y2, score);  K <= 64
 */
int pave_pose_finalize_f32(const float* kpts, const float* sigmas, const float* scores, const float* wh,
                           const float* sf, float* det_kpts, float* det_bboxes, int B, int N, int K,
                           int rescale, int sigma_ld, void* stream);

/*
 * Reference-point update of the decoders read straight from the grouped per-frame MLP output
 * (opera/models/utils/transformer.py:6728-6735, mmdet/models/utils/transformer.py:861-866):
 *   y [R, T*op]: row r, frame t's o outputs at columns [t op, t op + o);  ref, out [R*T, o] with row
 *   (r / G) T G + t G + r % G  (G = queries per clip: frame-major inside a clip; G = R: frame-major
 *   over all rows);  out = sigmoid(y + inverse_sigmoid(ref)), eps as mmdet's inverse_sigmoid.
 */
int pave_ref_update_frames_f32(const float* y, const float* ref, float* out, int R, int T, int op, int o,
                               int G, float eps, void* stream);

/*
 * Two-stage query initialisation behind the proposal top-k (opera/models/utils/transformer.py:21386-21418), two
 * launches for the reference's gather / repeat / strided add / sigmoid sequence:
 *   pave_gather_rows_add_f32: rows[n, q, :] = src[n, index[n, q], :] (src [n, S, C] with a batch stride in elements,
 *     index [n, Q] int64: `tgt`, the selected rows of output_memory) and, when sum != NULL,
 *     sum[n, q, :] = rows[n, q, :] + add[q, :] (`query = tgt + query`; add [Q, C]).  C %% 4 == 0.
 *   pave_proposal_refs_f32: kpt [n*Q, ld] (its first K2 = 2 K columns; in place) += props[n, index[n, q], c & 1]
 *     (props [n, S, 2], batch stride in elements, 0 = one table for every clip; +inf marks an invalid proposal) and
 *     refs [n, T*Q, K2] = sigmoid(kpt) repeated for the T frames (`topk_kpts_unact.sigmoid().repeat(1, T, 1)`).
 * An index outside [0, S) is never dereferenced: the gathered row is zero, the proposal term is dropped.
 */
int pave_gather_rows_add_f32(const float* src, long long src_batch_stride, const long long* index, const float* add,
                             float* rows, float* sum, int n, int Q, int S, int C, void* stream);
int pave_proposal_refs_f32(float* kpt, int ld, const float* props, long long props_batch_stride,
                           const long long* index, float* refs, int n, int Q, int S, int K2, int T, void* stream);

/*
 * Greedy OKS-NMS, one launch for n_clips clips (replaces oks_nms / oks_iou,
 * opera/models/dense_heads/videopose_head_mul_frames.py:1624-1665, a NumPy loop behind a
 * device->host sync in the reference).
 *   kpts   [n_clips, N, K, 3] (x, y, score) pixels;  scores [n_clips, N];  sigmas [K] double, DEVICE
 *   keep   [n_clips, N] int32: 1 = kept;  order [n_clips, N] int32: indices by descending score
 * A pose is suppressed when its OKS with an earlier kept pose is > thresh.
 */
int pave_oks_nms_f32(const float* kpts, const float* scores, const double* sigmas, double thresh,
                     int32_t* keep, int32_t* order, int n_clips, int N, int K, void* stream);

/*
 * Fused row epilogues around the library GEMMs / convolutions (HBM-bound, one pass).
 *   bias_act_rows:      y[r,c] = act(x[r,c] + bias[c] + res[r,c]);  bias / res may be NULL,
 *                       y may alias x; relu != 0 applies max(.,0).  Replaces the separate
 *                       bias, residual-add and ReLU kernels behind conv / Linear
 *                       (mmdet resnet.py Bottleneck.forward, mmcv FFN).
 *   bias_add_layernorm: y[r,:] = LayerNorm(x[r,:] + bias + res[r,:]) * gamma + beta,
 *                       C % 4 == 0, C <= 3072 (the 'attn/ffn -> + identity -> norm' step of
 *                       mmcv BaseTransformerLayer.forward, bricks/transformer.py:1316-1353).
 */
/*
 * x[rows[i], 0:C] = values[0:C] (values == NULL: zeros) for i < n_rows; x [total_rows, ld] row-major, `rows`
 * int32 indices on the DEVICE (an index outside [0, total_rows) is skipped, never dereferenced).
 * The padding mask of the reference on a projected value matrix -- masked tokens are 0 when the mask follows
 * value_proj (third_party/mmcv/mmcv/ops/multi_scale_deform_attn.py:369-371) and value_proj.bias when it
 * precedes it (opera/models/utils/transformer.py:1706-1711, multi_scale_deform_attn.py:1454-1458) -- applied
 * to the listed rows only instead of a masked_fill pass over the whole memory (an 800 x 1333 image in an
 * 800 x 1344 batch masks ~1 % of the tokens).
 */
int pave_fill_rows_f32(float* x, long long ld, long long total_rows, const int* rows, long long n_rows,
                       const float* values, int C, void* stream);
int pave_bias_act_rows_f32(const float* x, const float* bias, const float* res, float* y,
                           long long rows, int C, int relu, void* stream);
/*
 * HRNet fuse layer (third_party/mmdetection/mmdet/models/backbones/hrnet.py:197-214:
 * `y = 0; for j: y += x[j] if i == j else fuse_layers[i][j](x[j])`, `relu(y)`), NHWC fp32, one pass:
 *   y[n, h, w, :] = act(s0[n, h >> sh0, w >> sh0, :] + s1[...] + s2[...] + s3[...])   (in this order)
 * source k is a [N, H >> sh_k, W >> sh_k, C] map -- the nearest-neighbour nn.Upsample(scale_factor =
 * 2^sh_k) that ends the coarser branches' fuse path is folded into the read; s1..s3 may be NULL.
 * y may alias a source with sh == 0.  C %% 4 == 0, 2^sh_k divides H and W.
 */
int pave_fuse_sum_nhwc_f32(const float* s0, int sh0, const float* s1, int sh1, const float* s2, int sh2,
                           const float* s3, int sh3, float* y, int N, int H, int W, int C, int relu,
                           void* stream);
int pave_bias_add_layernorm_f32(const float* x, const float* bias, const float* res,
                                const float* gamma, const float* beta, float* y, long long rows,
                                int C, float eps, void* stream);
/* As above, plus y_plus[r,:] = y[r,:] + pos[r %% pos_rows, :] in the same pass: the next
 * encoder layer's `query + query_pos` (mmcv multi_scale_deform_attn.py:353-354). */
int pave_bias_add_layernorm_pos_f32(const float* x, const float* bias, const float* res,
                                    const float* gamma, const float* beta, float* y,
                                    const float* pos, long long pos_rows, float* y_plus,
                                    long long rows, int C, float eps, void* stream);

/*
 * Encoder deformable attention ([R2]: mmcv MultiScaleDeformableAttention.forward, MO:373-404,
 * T = 1, M = 8, D = 32, L = 4, P = 4) with the value rows staged in LDS per 8 x 8-pixel image
 * tile and head (pavenet_amd/csrc/pave_enc_tile.hip).  Same inputs, outputs and results as
 * pave_deform_attn_grid_fused_f32 with T = 1 and no unit_clip / order; corners outside a tile's
 * LDS window are fetched from global memory, so results do not depend on the window size.
 *   value [n_frames, S, 8, 32]; proj [n_frames*S, proj_stride] (offsets [8][4][4][2], then
 *   logits [8][4][4]); ref [n_frames*S, 4, 2]; out [n_frames*S, 256]
 *   levels_hw   HOST array of 8 ints (h0, w0, ..., h3, w3), levels in flattening order
 *   variant     a bit mask: 0 = windows of -4 .. +3 px (3 workgroups per CU), 1 = -4 .. +4 px (1 per CU);
 *               | 4 = `proj` is PREPARED (pave_gemm_bf16x3_encproj_f32: pixel coordinates + attention weights;
 *               ref is not read; default windows only);  | 8 = `out` is fp16 [n_frames*S, 256] (fp16 operand
 *               mode: the rows only feed output_proj's MFMA, pave_gemm_fp16_act_f32 -- the same values at half
 *               the bytes).  Other bits / combinations: PAVE_E_ARG / PAVE_E_UNSUPPORTED before anything is enqueued
 *   window_shift  NULL, or a HOST array [8 heads][4 levels][2] of (dx, dy) in level pixels that
 *               moves the LDS window of (tile, head, level): a head's points sit around
 *               reference + its mean learnt offset (the reference initialises them on a ray 1..4 px
 *               out, MO:227-240), so centring the window there keeps them in LDS.  Speed only.
 * Returns PAVE_E_UNSUPPORTED (nothing launched) unless every level l satisfies
 * H_l <= (8 >> l) * ceil(H_0 / 8) and W_l likewise (a halving pyramid): callers then use
 * pave_deform_attn_grid_fused_f32.
 */
int pave_enc_deform_attn_tile_f32(const float* value, const float* proj, const float* ref,
                                  float* out, int n_frames, int S, const int* levels_hw,
                                  int proj_stride, int variant, const int* window_shift,
                                  void* stream);

/*
 * [R1] backward: void ms_deform_attn_backward(value, spatial_shapes, level_start_index,
 * sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc, grad_attn_weight,
 * im2col_step)  (pybind.cpp:167-173, 743-748; ms_deform_attn_cuda.cu:279-351;
 * kernels ms_deform_attn_cuda_kernel.cuh:66-198, 256-801).
 *   grad_output [bs, Lq, M*D];  grad_value [bs, S, M, D] is ACCUMULATED into (caller zeroes it,
 *   as MO:72 does);  grad_sampling_loc [bs, Lq, M, L, P, 2] and grad_attn_weight
 *   [bs, Lq, M, L, P] are fully overwritten.
 */
int pave_ms_deform_attn_backward_f32(const float* value, const int64_t* spatial_shapes,
                                     const int64_t* level_start, const float* sampling_loc,
                                     const float* attn_weight, const float* grad_output,
                                     float* grad_value, float* grad_sampling_loc,
                                     float* grad_attn_weight, int bs, int S, int M, int D, int L,
                                     int Lq, int P, int im2col_step, void* stream);
int pave_ms_deform_attn_backward_f64(const double* value, const int64_t* spatial_shapes,
                                     const int64_t* level_start, const double* sampling_loc,
                                     const double* attn_weight, const double* grad_output,
                                     double* grad_value, double* grad_sampling_loc,
                                     double* grad_attn_weight, int bs, int S, int M, int D, int L,
                                     int Lq, int P, int im2col_step, void* stream);

/*
 * Device input pipeline for T frames of one clip (mmdet Resize(keep_ratio) -> Normalize(to_rgb)
 * -> Pad -> MulImageToTensor; configs/_base_/datasets/posetrack17_video_keypoint.py:71-84).
 *   src   [T, H0, W0, 3] HWC, BGR, uint8 (src_is_u8 = 1) or float32, DEVICE
 *   dst   [T, 3, Hp, Wp] float32: the (Hn, Wn) resized + normalised image top-left, zeros elsewhere
 *   mean, std  HOST float[3] (in the order of the channels AFTER the optional BGR->RGB swap)
 */
int pave_preprocess_frames(const void* src, int src_is_u8, float* dst, int T, int H0, int W0,
                           int Hn, int Wn, int Hp, int Wp, const float* mean, const float* std,
                           int to_rgb, void* stream);

/*
 * 3x3 convolution, NHWC fp32, pad 1, stride 1 or 2, bias (+ReLU) fused: implicit GEMM on the
 * exact-fp32 MFMA.  Used for the ResNet / HRNet 3x3 convolutions with BatchNorm folded in
 * (mmdet resnet.py Bottleneck.conv2 / BasicBlock, hrnet.py).
 *   x [N, H, W, Cin];  w [3, 3, Cin, Cout] (tap-major, Cout innermost);  bias [Cout] or NULL;
 *   y [N, Ho, Wo, Cout], Ho = (H - 1)/stride + 1.  Cin %% 32 == 0, Cout %% 64 == 0.
 */
int pave_conv3x3_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int N,
                          int H, int W, int Cin, int Cout, int stride, int relu, void* stream);

/*
 * Row GEMM with the whole Bottleneck tail in its prologue / epilogue (fp32 MFMA, exact fp32):
 *   A1 = a_bias ? relu(a[M, K] + a_bias[K]) : a            (conv2's folded bn2 + ReLU on load)
 *   out[M, N] = act([A1 | a2[M, K2]] * w[K + K2, N] + bias[N] + residual[M, N])
 * = mmdet resnet.py:264-283 `bn2 -> relu -> conv3 -> bn3 -> (+ downsample(x) | + identity) ->
 * relu` on the NHWC map (BatchNorm folded into w / bias; the downsample 1x1 convolution is the
 * second K range, w = [W3; Wd]).  a_bias, a2 (with K2 = 0), bias, residual may be NULL;
 * `residual` may alias `out` (every element is read before it is written by the same lane).
 * K %% 32 == 0, K2 %% 32 == 0, N %% 64 == 0, M < 2^31.
 */
int pave_rows_gemm_bias_res_act_f32(const float* a, const float* a_bias, const float* a2,
                                    const float* w, const float* bias, const float* residual,
                                    float* out, long long M, int K, int K2, int N, int relu,
                                    void* stream);

/*
 * ResNet stem tail in one pass (resnet.py:640-645 `norm1 -> relu -> maxpool` with BN folded):
 *   y[N, Ho, Wo, C] = maxpool3x3/s2/p1(relu(x[N, H, W, C] + bias[C])),  Ho = (H - 1)/2 + 1.
 */
int pave_bias_relu_maxpool_nhwc_f32(const float* x, const float* bias, float* y, int N, int H,
                                    int W, int C, void* stream);

/*
 * fp32 row GEMM on the bf16 matrix cores by operand splitting into `nplanes` bf16 terms:
 *   nplanes = 3: exact split, 6 bf16 MFMAs per product tile, error <= 2^-23 relative per product
 *                (fp32-level);  2: 3 MFMAs, ~2^-16;  1: plain bf16 operands;  PAVE_PLANES_FP16:
 *                plain fp16 operands (BASELINE config 5's "fp16 MFMA projections") -- always
 *                with fp32 accumulation and fp32 in/out.
 *   out[M, N] = act(A'[M, K] * W[N, K]^T + bias[N] + residual[M, N]),
 *   A' = a_bias ? relu(a + a_bias[K]) : a;  act: relu = 0 none | 1 ReLU | 2 exact GELU x Phi(x) (nn.GELU, the
 *   activation of the Swin block's FFN, third_party/mmdetection/mmdet/models/backbones/swin.py:330-341) |
 *   3 sigmoid 1 / (1 + exp(-x)) (the heads' sigma branches, videopose_head_mul_frames.py:533, 652);  2 and 3:
 *   3 planes / fp16 only
 * = nn.Linear (mmcv FFN / projections, bricks/transformer.py:1046-1120) with the residual and
 * activation of its caller in the epilogue.  `w_planes` = the weight [N, K] split once by
 * pave_split_bf16x3_f32 into `nplanes` bf16 planes and re-laid slab-major [K/16][nplanes][N][16]
 * (one 16-wide K slab of a column tile contiguous; pavenet_amd.ops.split_weight_bf16x3).
 * residual may alias out.  M < 2^31.
 *   3 planes: K %% 32 == 0 (K >= 64), N %% 4 == 0; for N %% 64 != 0 the weight planes carry
 *   roundup(N, 64) rows (zero rows beyond N) while out / bias / residual have N columns.
 *   PAVE_PLANES_FP16: the same kernels and shape rules as 3 planes (one fp16 plane [K/16][1][N][16], the
 *   fp32 activation rows converted to fp16, round to nearest even, where the 3-plane form splits them;
 *   one v_mfma_f32_32x32x16_f16 per tile and 16-wide K slab).
 *   1 / 2 bf16 planes: K %% 64 == 0, N %% 128 == 0 (first-generation kernels).
 * Every entry point below that takes `nplanes` accepts 3 or PAVE_PLANES_FP16 (pave_gemm_bf16x3_f32,
 * _ex_f32 and pave_conv3x3_split_f32 also 1 and 2).
 */
#define PAVE_PLANES_FP16 16 /* nplanes value: ONE plane of fp16 (not bf16) operands */
int pave_gemm_bf16x3_f32(const float* a, const float* a_bias, const void* w_planes,
                         const float* bias, const float* residual, float* out, long long M, int K,
                         int N, int relu, int nplanes, void* stream);

/*
 * fp16 operand mode with fp16 ACTIVATIONS around a launch (BASELINE configs[4]; wide tile forms, N %% 256 == 0):
 * a tensor that is only ever a GEMM operand -- the hidden activation between the two Linears of an FFN
 * (third_party/mmcv/mmcv/cnn/bricks/transformer.py:1046-1120) -- is stored as the fp16 values the next launch's
 * MFMA consumes anyway: the same results as fp32 storage (the consumer would round it to fp16 at operand fetch),
 * half the bytes on an HBM-bound chain.
 *   a [M, K] fp32, or fp16 when a_is_f16;  w_plane = ONE fp16 plane [K/16][1][N][16] (PAVE_PLANES_FP16);
 *   gamma == NULL:  out[M, N] = act(a W^T + bias), fp32 or (out_is_f16) fp16;  relu as pave_gemm_bf16x3_f32;
 *   gamma != NULL:  out[M, 256] = LayerNorm(a W^T + bias + residual) * gamma + beta, fp32 (N == 256; residual fp32,
 *                   may be NULL or alias out).
 */
int pave_gemm_fp16_act_f32(const void* a, int a_is_f16, const void* w_plane, const float* bias,
                           const float* residual, const float* gamma, const float* beta, float eps, void* out,
                           int out_is_f16, long long M, int K, int N, int relu, void* stream);

/*
 * Same GEMM with two epilogue options (the encoder layer's `value_proj | sampling_offsets |
 * attention_weights` Linears of third_party/mmcv/mmcv/ops/multi_scale_deform_attn.py:355-379 run
 * as ONE launch over the layer input):
 *   residual_rows > 0: `residual` is a [residual_rows, N] table and row m adds residual[m %
 *     residual_rows] -- a per-token term shared by all frames (e.g. (query_pos @ W^T + b) when the
 *     positional encoding is the same for every frame); 0: residual is [M, N] as above.
 *   out2 != NULL: the product is cut at column n_split (n_split %% 128 == 0) into two dense
 *     matrices, out [M, n_split] and out2 [M, N - n_split].
 */
int pave_gemm_bf16x3_ex_f32(const float* a, const float* a_bias, const void* w_planes,
                            const float* bias, const float* residual, long long residual_rows,
                            float* out, float* out2, int n_split, long long M, int K, int N,
                            int relu, int nplanes, void* stream);

/*
 * out[i] = sigmoid(tmp[i] + inverse_sigmoid(ref[i])), inverse_sigmoid(x) = log(max(clamp(x, 0, 1),
 * eps) / max(1 - clamp(x, 0, 1), eps)): the per-layer reference-point update of the pose / joint
 * decoders (opera/models/utils/transformer.py:6733-6735,
 * third_party/mmdetection/mmdet/models/utils/transformer.py:865-866) as one launch.
 */
int pave_ref_update_f32(const float* tmp, const float* ref, float* out, long long n, float eps,
                        void* stream);

/*
 * GroupNorm of an NHWC map x [N, HW, C] (G groups of C/G consecutive channels), y = GN(x) * gamma +
 * beta written to y + n * y_batch_stride + (hw * C + c): the destination may be a slice of a larger
 * [N, S, C] token buffer (the neck's conv -> GN levels land directly in the transformer's flattened
 * multi-level feature, third_party/mmdetection/mmdet/models/necks/channel_mapper.py:90-100 +
 * opera/models/utils/transformer.py:21312-21331).  Statistics are accumulated in fp64 in a fixed
 * order (bit-reproducible).  Scratch supplied by the caller: partial [N * nchunks * G * 2] doubles
 * (nchunks = number of row chunks the statistics pass is cut into), ab [N * 2 * C] floats.
 * C %% 4 == 0, (C / G) %% 4 == 0, C / 4 divides 256.
 */
int pave_groupnorm_nhwc_f32(const float* x, const float* gamma, const float* beta, float* y,
                            long long y_batch_stride, int N, int HW, int C, int G, float eps,
                            double* partial, int nchunks, float* ab, void* stream);

/*
 * The same GroupNorm over up to four maps of one N, C and G at once -- the levels of the neck
 * (necks/channel_mapper.py:90-100: one ConvModule per level, each with its own GroupNorm(32)) -- as THREE launches
 * instead of three per level; per level the values are those of pave_groupnorm_nhwc_f32 with the same nchunks,
 * bit for bit.  Scratch: partial [N * G * 2 * sum(nchunks)] doubles, ab [nlev * N * 2 * C] floats.
 */
typedef struct pave_gn_level {
  const float* x;              /* [N, HW, C] dense */
  const float* gamma;          /* [C] */
  const float* beta;           /* [C] */
  float* y;                    /* row n at y + n * y_batch_stride */
  long long y_batch_stride;    /* floats, >= HW * C */
  int HW;
  int nchunks;                 /* row chunks of the statistics pass */
  float eps;
} pave_gn_level;
int pave_groupnorm_levels_nhwc_f32(const pave_gn_level* levels, int nlev, int N, int C, int G, double* partial,
                                   float* ab, void* stream);

/*
 * out[M, N] = act([a | a2] @ W^T + bias + residual): two row matrices a [M, K1] and a2 [M, K - K1]
 * share one K axis and one accumulator (3 bf16 planes) -- the ResNet Bottleneck tail with a
 * stride-1 downsample, relu(conv3(y) + downsample(x)) = relu([y | x] @ [W3 | Wd]^T + b3 + bd)
 * (third_party/mmdetection/mmdet/models/backbones/resnet.py:264-283), in ONE launch with no
 * concatenation copy.  w_planes = the planes of the [N, K] row-concatenated weight.
 * K %% 32 == 0, N %% 64 == 0, 0 < K1 < K, K1 %% 16 == 0.  residual may be NULL or alias out.
 */
int pave_gemm_bf16x3_cat_f32(const float* a, long long K1, const float* a2, const void* w_planes,
                             const float* bias, const float* residual, float* out, long long M, int K,
                             int N, int relu, int nplanes, void* stream);

/*
 * Grouped row GEMM (3 bf16 planes): the N axis is cut into N / group_n groups; group i computes
 *   out[:, i group_n : (i + 1) group_n] = act(a[:, i K : (i + 1) K] @ W_i^T + bias_i),
 * a [M, lda] row-major (lda >= groups * K), W_i [group_n, K]; w_planes = the planes of the [N, K]
 * row-concatenation of the W_i.  One launch for the T per-frame Linears of one layer of the
 * reference's per-frame key-point / regression branches (pre_pre_ / pre_ / "" / next_ / next_next_
 * kpt_branches, opera/models/utils/transformer.py:6728-6740;
 * third_party/mmdetection/mmdet/models/utils/transformer.py:860-875).
 * K %% 32 == 0, group_n %% 64 == 0, lda %% 4 == 0.
 */
int pave_gemm_bf16x3_grouped_f32(const float* a, long long lda, const void* w_planes, const float* bias,
                                 float* out, long long M, int K, int N, int group_n, int relu,
                                 int nplanes, void* stream);

/*
 * out[M, 256] = LayerNorm(a @ W^T + bias + residual) * gamma + beta  (3 bf16 planes, N == 256): the
 * attention / FFN output Linear, its residual add and the post-norm of a BaseTransformerLayer
 * (third_party/mmcv/mmcv/cnn/bricks/transformer.py:1316-1353) in ONE launch -- a 128 x 256 block
 * tile owns whole rows, the row statistics are completed across its waves through LDS (two-pass
 * mean / centred variance).  residual may be NULL or alias out.  K %% 64 == 0.
 */
int pave_gemm_bf16x3_ln_f32(const float* a, const void* w_planes, const float* bias,
                            const float* residual, const float* gamma, const float* beta, float eps,
                            float* out, long long M, int K, int N, int nplanes, void* stream);

/*
 * 3x3 / pad 1 / stride 1|2 convolution, NHWC fp32 in and out, as an implicit GEMM through the same
 * split-operand kernel (K axis = (ky, kx, cin)), bias (+ReLU) in the epilogue:
 *   x [N, H, W, Cin];  w_planes = the weight [Cout, 3, 3, Cin] (i.e. [Cout, 9*Cin] rows) split and
 *   re-laid like pave_gemm_bf16x3_f32's operand;  y [N, Ho, Wo, Cout], Ho = (H - 1)/stride + 1.
 * Replaces a ResNet / HRNet 3x3 nn.Conv2d + folded BatchNorm (+ identity) (+ReLU)
 * (third_party/mmdetection/mmdet/models/backbones/resnet.py:53-98 BasicBlock, hrnet.py:183-260)
 * in the split / 16-bit GEMM modes.
 *   3 planes: Cin %% 16 == 0, Cout %% 4 == 0; the weight planes are ZERO-PADDED to
 *   roundup(9 Cin, 32) columns and roundup(Cout, 64) rows (HRNet's 48- / 96-channel branches);
 *   residual [N, Ho, Wo, Cout] (may be NULL or alias y) is added before the ReLU.
 *   PAVE_PLANES_FP16: as 3 planes.  1 / 2 bf16 planes: Cin %% 64 == 0, Cout %% 64 == 0, residual == NULL.
 */
int pave_conv3x3_split_f32(const float* x, const void* w_planes, const float* bias,
                           const float* residual, float* y, int N, int H, int W, int Cin, int Cout,
                           int stride, int relu, int nplanes, void* stream);

/*
 * The same convolution (3 planes) with the K axis cut into parts -- for maps with FEW output pixels
 * and many input channels, where the 128-row tiles alone leave most of the chip idle (the
 * ChannelMapper's extra level: 3x3 / stride 2, 2048 -> 256 on the C5 map,
 * third_party/mmdetection/mmdet/models/necks/channel_mapper.py:84-97).  Each part is computed by
 * its own workgroups into `workspace` ([parts][N Ho Wo][Cout] fp32), a second launch adds the
 * parts IN ORDER (deterministic), then bias, residual and ReLU.
 *   pave_conv3x3_splitk_workspace_bytes: the workspace the shape needs; 0 = the shape has no
 *   split-K plan (enough tiles or a short K): call pave_conv3x3_split_f32.
 *   pave_conv3x3_splitk_f32: workspace_bytes >= that value; the workspace is free for reuse in
 *   stream order after the call.
 */
long long pave_conv3x3_splitk_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride);
int pave_conv3x3_splitk_f32(const float* x, const void* w_planes, const float* bias,
                            const float* residual, float* y, int N, int H, int W, int Cin, int Cout,
                            int stride, int relu, void* workspace, long long workspace_bytes,
                            int nplanes, void* stream);

/*
 * The same split-K plan for the plain row GEMM out[M, N] = act(a[M, K] W^T + bias + residual) (relu = 0 | 1): few
 * row tiles and K >= 2048 -- ResNet layer4's 1x1 reductions (resnet.py:264-271) and the ChannelMapper's C5 lateral
 * on a one-clip batch.  pave_gemm_splitk_workspace_bytes = 0: the shape has no plan, use pave_gemm_bf16x3_f32.
 */
long long pave_gemm_splitk_workspace_bytes(long long M, int K, int N);
int pave_gemm_bf16x3_splitk_f32(const float* a, const void* w_planes, const float* bias, const float* residual,
                                float* out, long long M, int K, int N, int relu, int nplanes, void* workspace,
                                long long workspace_bytes, void* stream);

/*
 * ResNet Bottleneck of the 64-channel stage from its 3x3 convolution on, CHAINED with the next
 * block's conv1, in ONE launch (third_party/mmdetection/mmdet/models/backbones/resnet.py:263-300
 * Bottleneck.forward: conv2 + bn2 + relu -> conv3 + bn3 -> + identity | downsample(x) -> relu; then
 * the following block's conv1 + bn1 + relu), BatchNorms folded, NHWC fp32, M = N H W pixels:
 *   c2  = relu(conv3x3_pad1(c1 [N, H, W, 64]; w2_planes) + b2)            -> c2  [M, 64] (scratch)
 *         c1 == NULL and w2_planes == NULL: the launch starts at conv3 and c2 [M, 64] is its INPUT
 *         (the 3x3 -- pave_conv3x3_split_f32 -- was a launch of its own)
 *   out = relu([c2 | a2] @ W3^T + b3 + residual)                          -> out [M, 256]
 *         a2 [M, k2] (k2 %% 32 == 0): the block input of a stride-1 downsample block, W3 = the
 *         [256, 64 + k2] row-concatenated conv3 | downsample weight, b3 = both biases; else
 *         residual [M, 256] = the block input (may alias out), a2 == NULL, k2 == 0
 *   c1n = relu(out @ W1n^T + b1n)                                         -> c1n [M, cn], cn = 64 | 128
 *         (w1n_planes == NULL, cn == 0: the launch stops after `out`)
 * w2_planes as pave_conv3x3_split_f32's (3 planes), w3_planes / w1n_planes as
 * pave_gemm_bf16x3_f32's (3 planes).  Every value equals what pave_conv3x3_split_f32,
 * pave_gemm_bf16x3_cat_f32 / pave_gemm_bf16x3_f32 give launched one after the other (same
 * kernels bodies, same tiles) -- the chain exists for time: a workgroup carries one 128-pixel tile
 * through the three GEMMs, so the HBM-bound conv3 + identity phase of one workgroup overlaps the
 * MFMA-bound phases of its neighbours and conv1 reads `out` back from L2.
 */
int pave_bottleneck_chain_f32(const float* c1, const void* w2_planes, const float* b2, float* c2,
                              const void* w3_planes, const float* b3, const float* residual,
                              const float* a2, int k2, float* out, const void* w1n_planes,
                              const float* b1n, float* c1n, int cn, int N, int H, int W, int nplanes,
                              void* stream);

/*
 * 1x1 convolution with a stride on an NHWC map (the ResNet downsample branch,
 * third_party/mmdetection/mmdet/models/backbones/resnet.py:258-262 `self.downsample(x)`): the row
 * GEMM pave_gemm_bf16x3_f32 (3 planes) whose A row m is the input pixel (oy * stride, ox * stride)
 * of image n -- no strided-slice copy.  x [N, H, W, Cin], y [N, Ho, Wo, Cout], Ho = (H - 1)/stride + 1.
 */
int pave_conv1x1_strided_split_f32(const float* x, const void* w_planes, const float* bias, float* y,
                                   int N, int H, int W, int Cin, int Cout, int stride, int relu,
                                   int nplanes, void* stream);

/*
 * The ResNet / HRNet stem convolution (7x7, stride 2, pad 3, 3 -> 64 channels; folded BatchNorm as
 * `bias`; third_party/mmdetection/mmdet/models/backbones/resnet.py:632-640) read straight from the
 * NCHW fp32 image batch x [N, 3, H, W], as an implicit GEMM on the 3-plane split scheme.
 * w_planes = 23 K-slabs [23][3][64][16] bf16 holding the SAME weights in two K layouts:
 *   slabs 0..11:  K = (c, ky, kx) with kx padded 7 -> 8 and (c, ky) padded 21 -> 24 rows (192) --
 *                 the per-lane window-load kernel (any W);
 *   slabs 12..22: K = (c, ky, kx') with kx' = kx + 1 (tap 0 unused) and (c, ky) padded 21 -> 22 rows
 *                 (176) -- the kernel that stages the block's input window in LDS by LDS-DMA,
 *                 taken when W %% 4 == 0 and x is 16-byte aligned.
 * y [N, Ho, Wo, 64] NHWC, Ho = (H - 1) / 2 + 1.
 * nplanes = PAVE_PLANES_FP16: w_planes = [23][1][64][16] fp16 (the same two layouts, one plane); only the
 * LDS-window kernel exists for it (W %% 4 == 0, x 16-byte aligned; PAVE_E_UNSUPPORTED otherwise).
 * row_pitch (ABI 19): elements between two image rows (0 or W: dense rows).  row_pitch > W is the layout
 * pave_repitch_rows_f32 writes: row_pitch %% 4 == 0, x 16-byte aligned, columns W .. row_pitch - 1 of every row
 * ZERO (they are the convolution's right-hand zero padding) -- an image of ANY width, e.g. the 750 x 1333 frames
 * of the reference's PoseTrack test pipeline (configs/_base_/datasets/posetrack17_video_keypoint.py:68-81,
 * size_divisor = 1), then takes the LDS-window kernel; Wo = (W - 1) / 2 + 1 is that of the real width.
 */
int pave_conv7x7s2_nchw_split_f32(const float* x, const void* w_planes, const float* bias, float* y,
                                  int N, int H, int W, int row_pitch, int Cout, int relu, int nplanes,
                                  void* stream);

/*
 * dst[r, 0:W] = src[r, 0:W], dst[r, W:pitch] = 0 for r < rows: dense fp32 rows re-laid at a row pitch that is a
 * multiple of 4 elements (16-byte aligned rows for the LDS-DMA stem above).  pitch %% 4 == 0, pitch >= W, dst
 * 16-byte aligned, src any alignment; one pass, ~2 x the image bytes.
 */
int pave_repitch_rows_f32(const float* src, float* dst, long long rows, int W, int pitch, void* stream);

/*
 * The (shifted-)window multi-head self-attention core of a Swin block -- ShiftWindowMSA.forward around
 * WindowMSA.forward, third_party/mmdetection/mmdet/models/backbones/swin.py:22-126, 128-286 -- on the
 * UN-partitioned token map (pad to a multiple of the window, roll by -shift, window partition, relative-position
 * bias, the -100 mask between roll regions, softmax, PV, window reverse, reverse roll and crop are all index
 * arithmetic inside the kernel; no copy of the map, no mask tensor):
 *   qkv [B, H, W, 3 C] fp32 = the block's qkv Linear per token (q | k | v, each [heads][32]);
 *   bias_t [heads, ws^2 (key j), ws^2 (query i)] = relative_position_bias_table[relative_position_index],
 *     TRANSPOSED per head (lanes = queries read consecutive addresses);
 *   pad_qkv [3 C] = the qkv Linear's bias: what a zero pad token projects to (the reference pads BEFORE qkv);
 *   out [B, H, W, C]: the attention output at every real pixel, the input of `proj`.
 * scale = head_dim^-0.5 (or qk_scale).  Built for window 7, head dim 32 (C == heads * 32): every Swin variant
 * of the reference's configs.
 */
int pave_swin_window_attn_f32(const float* qkv, const float* bias_t, const float* pad_qkv, float* out, int B,
                              int H, int W, int C, int heads, int window, int shift, float scale, void* stream);

/*
 * Exact merge of G partial attention rows of a frame-sharded T-frame attention (pavenet_amd/dist.py; SURVEY
 * 8e: "all-gather of pose-query logits"): parts [G, U, C + 2 H] = per rank the softmax-weighted row over its
 * frames (normalised by its own sum) followed by the per-head max logit and sum(exp(logit - max)) the fused
 * kernels emit (stat_max, stat_sum) -> out [U, C], the row of ONE softmax over all ranks' logits.  A rank
 * with sum = 0 (no frames) drops out.  C / H channels per head (a multiple of 4).
 */
int pave_merge_softmax_partials_f32(const float* parts, float* out, int G, int U, int C, int H, void* stream);

/* x[n] fp32 -> planes[nplanes][n] bf16: truncation terms, the last rounded to nearest even
 * (nplanes = 3: x = p0 + p1 + p2 exactly). */
int pave_split_bf16x3_f32(const float* x, void* planes, long long n, int nplanes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PAVE_HIP_H_ */
