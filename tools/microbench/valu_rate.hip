// VALU issue-rate probe for gfx950: cycles per wave64 instruction per SIMD for v_fma_f32,
// v_pk_fma_f32 and a DPP v_add_u32, at 1 / 2 / 4 / 5 waves per SIMD.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void probe(float* out, int iters) {
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  f2 q0 = p0, q1 = p1, q2 = p2, q3 = p3;
  int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
  const float w = 1.0001f;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
      asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                   "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
    } else if (KIND == 1) {
      asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                   "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(p0));
    } else if (KIND == 2) {
      asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_u32_dpp %1, %2, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_u32_dpp %2, %3, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_u32_dpp %3, %4, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_u32_dpp %4, %5, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_u32_dpp %5, %6, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_u32_dpp %6, %7, %6 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_u32_dpp %7, %0, %7 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
    } else {
      asm volatile("v_add_u32 %0, %1, %0\n v_add_u32 %1, %2, %1\n v_add_u32 %2, %3, %2\n v_add_u32 %3, %4, %3\n"
                   "v_add_u32 %4, %5, %4\n v_add_u32 %5, %6, %5\n v_add_u32 %6, %7, %6\n v_add_u32 %7, %0, %7\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y +
                                               q0.x + q1.x + q2.x + q3.x + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}
template <int KIND>
void run(const char* name, float* out) {
  const int iters = 20000;
  for (int wps : {1, 2, 4, 5, 8}) {
    dim3 grid(256), block(256 * wps > 1024 ? 1024 : 256 * wps);
    int blocks = 256 * ((256 * wps + 1023) / 1024);
    if (256 * wps > 1024) { grid = dim3(blocks); }
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    probe<KIND><<<grid, block>>>(out, 100);
    hipEventRecord(s);
    probe<KIND><<<grid, block>>>(out, iters);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double waves_per_simd = (double)grid.x * block.x / 64 / 1024;
    const double instr_per_simd = waves_per_simd * iters * 8.0;
    printf("%-14s waves/SIMD %.1f: %.3f ms -> %.2f ns per instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name,
           waves_per_simd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  }
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
  run<0>("v_fma_f32", out);
  run<1>("v_pk_fma_f32", out);
  run<2>("v_add_u32_dpp", out);
  run<3>("v_add_u32", out);
  return 0;
}
