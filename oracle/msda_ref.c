/*
 * oracle/msda_ref.c -- TEST INFRASTRUCTURE (the checker, never the product).
 *
 * Plain-C CPU restatement of the reference's multi-scale deformable attention
 * forward.  Follows, line for line in arithmetic order:
 *   third_party/mmcv/mmcv/ops/csrc/common/cuda/ms_deform_attn_cuda_kernel.cuh
 *     :17-64   ms_deform_attn_im2col_bilinear   (4-corner fetch, zero outside)
 *     :200-254 ms_deformable_im2col_gpu_kernel  (loop over levels/points,
 *              h_im = loc_h*H - 0.5, in-bounds test at :241)
 * The reference kernel itself is CUDA-only (no CPU implementation is registered
 * for ms_deform_attn in csrc/pytorch/cpu/), so it cannot be compiled here with
 * gcc; this restatement is pinned instead against the reference's own PyTorch
 * implementation (multi_scale_deform_attn.py:92-149) through tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.
 */
#include <math.h>
#include <stdint.h>

#define DEFINE_MSDA(NAME, T, FLOOR)                                                             \
  static T NAME##_bilinear(const T* data, int height, int width, int nheads, int channels,     \
                           T h, T w, int m, int c) {                                            \
    const int h_low = (int)FLOOR(h);                                                            \
    const int w_low = (int)FLOOR(w);                                                            \
    const int h_high = h_low + 1;                                                               \
    const int w_high = w_low + 1;                                                               \
    const T lh = h - h_low;                                                                     \
    const T lw = w - w_low;                                                                     \
    const T hh = 1 - lh, hw = 1 - lw;                                                           \
    const int64_t w_stride = (int64_t)nheads * channels;                                        \
    const int64_t h_stride = width * w_stride;                                                  \
    const int64_t h_low_off = h_low * h_stride;                                                 \
    const int64_t h_high_off = h_low_off + h_stride;                                            \
    const int64_t w_low_off = w_low * w_stride;                                                 \
    const int64_t w_high_off = w_low_off + w_stride;                                            \
    const int64_t base = (int64_t)m * channels + c;                                             \
    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                           \
    if (h_low >= 0 && w_low >= 0) v1 = data[h_low_off + w_low_off + base];                      \
    if (h_low >= 0 && w_high <= width - 1) v2 = data[h_low_off + w_high_off + base];            \
    if (h_high <= height - 1 && w_low >= 0) v3 = data[h_high_off + w_low_off + base];           \
    if (h_high <= height - 1 && w_high <= width - 1) v4 = data[h_high_off + w_high_off + base]; \
    const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                             \
    return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                                             \
  }                                                                                             \
  void NAME(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,            \
            const T* attw, T* out, int bs, int S, int M, int D, int L, int Lq, int P) {         \
    const int64_t n = (int64_t)bs * Lq * M * D;                                                 \
    for (int64_t index = 0; index < n; ++index) {                                               \
      int64_t tmp = index;                                                                      \
      const int c_col = (int)(tmp % D);                                                         \
      tmp /= D;                                                                                 \
      const int64_t sampling_index = tmp;                                                       \
      const int m_col = (int)(tmp % M);                                                         \
      tmp /= M;                                                                                 \
      tmp /= Lq;                                                                                \
      const int64_t b_col = tmp;                                                                \
      int64_t wptr = sampling_index * L * P;                                                    \
      int64_t lptr = wptr << 1;                                                                 \
      const int64_t qid_stride = (int64_t)M * D;                                                \
      const int64_t init = b_col * S * qid_stride;                                              \
      T col = 0;                                                                                \
      for (int l = 0; l < L; ++l) {                                                             \
        const int64_t start = lsi[l];                                                           \
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                           \
        const T* vptr = value + init + start * qid_stride;                                      \
        for (int p = 0; p < P; ++p) {                                                           \
          const T loc_w = loc[lptr], loc_h = loc[lptr + 1];                                     \
          const T weight = attw[wptr];                                                          \
          const T h_im = loc_h * H - (T)0.5;                                                    \
          const T w_im = loc_w * W - (T)0.5;                                                    \
          if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)                                   \
            col += NAME##_bilinear(vptr, H, W, M, D, h_im, w_im, m_col, c_col) * weight;        \
          wptr += 1;                                                                            \
          lptr += 2;                                                                            \
        }                                                                                       \
      }                                                                                         \
      out[index] = col;                                                                         \
    }                                                                                           \
  }

DEFINE_MSDA(oracle_msda_forward_f32, float, floorf)
DEFINE_MSDA(oracle_msda_forward_f64, double, floor)
