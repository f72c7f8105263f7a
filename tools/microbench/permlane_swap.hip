// What v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950) do to two registers, as the builtins return them:
// prints, for r = __builtin_amdgcn_permlane32_swap(a, b) and q = __builtin_amdgcn_permlane16_swap(a, b), the
// source (register, lane) of every lane of r.x, r.y, q.x, q.y -- the half-tail GEMM form builds its 16x16x32 A
// operands from two 32x32x16 operands with these two instructions.
//   hipcc --offload-arch=gfx950 tools/microbench/permlane_swap.hip -o /tmp/pls && /tmp/pls
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  const unsigned a = l, b = 100 + l;
  u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  u2 q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  u2 c = __builtin_amdgcn_permlane16_swap(r.x, r.y, false, false);   // the composition the kernel uses
  out[l] = r.x, out[64 + l] = r.y, out[128 + l] = q.x, out[192 + l] = q.y, out[256 + l] = c.x, out[320 + l] = c.y;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 384 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[384];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"permlane32_swap(a,b).x", "permlane32_swap(a,b).y", "permlane16_swap(a,b).x",
                          "permlane16_swap(a,b).y", "16(32(a,b)).x", "16(32(a,b)).y"};
  for (int i = 0; i < 6; ++i) {
    printf("%-24s rows of 16 lanes:", names[i]);
    for (int r = 0; r < 4; ++r) {
      const unsigned v = h[i * 64 + r * 16];
      printf("  %c%u", v >= 100 ? 'b' : 'a', (v % 100) / 16);
      for (int j = 1; j < 16; ++j)
        if (h[i * 64 + r * 16 + j] != v + j) printf("(!)");
    }
    printf("\n");
  }
  return 0;
}
