/* Arithmetic shared by the encoder sampling kernel (pave_enc_tile.hip) and the epilogue of the merged
   projection GEMM that can run it on the sampler's behalf (pave_gemm_dma.hip): the softmax over the
   16 logits of a (query, head) -- 4 lanes of a quad x 4 logits -- and sampling location -> level
   pixel coordinates (multi_scale_deform_attn.py:373-404, ms_deform_attn_cuda_kernel.cuh:233-234).
   ONE definition, so that both sides produce the same bits. */
#ifndef PAVE_ENC_MATH_H_
#define PAVE_ENC_MATH_H_
#include <hip/hip_runtime.h>

namespace pave_enc {

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
__device__ __forceinline__ float div_by(float x, float d, float rd) {
  // x / d with rd = 1 / d: one residual correction of the product (correctly rounded up to rare
  // half-way cases), 3 instructions instead of the ~10 of an IEEE division
  const float q = x * rd;
  return fmaf(fmaf(-q, d, x), rd, q);
}
// softmax over the 16 logits a quad holds (4 per lane): lane's 4 weights, un-normalised e's and 1/sum
__device__ __forceinline__ void softmax16(float l0, float l1, float l2, float l3, float (&e)[4],
                                          float& inv_sum) {
  const float mx = quad_max(fmaxf(fmaxf(l0, l1), fmaxf(l2, l3)));
  e[0] = __expf(l0 - mx), e[1] = __expf(l1 - mx), e[2] = __expf(l2 - mx), e[3] = __expf(l3 - mx);
  inv_sum = __builtin_amdgcn_rcpf(quad_sum((e[0] + e[1]) + (e[2] + e[3])));  // 1 ulp
}
// level pixel coordinate of a sampling point: (ref + off / size) * size - 0.5
__device__ __forceinline__ float pixel_coord(float ref, float off, float size, float rsize) {
  return fmaf(ref + div_by(off, size, rsize), size, -0.5f);
}

}  // namespace pave_enc
#endif /* PAVE_ENC_MATH_H_ */
