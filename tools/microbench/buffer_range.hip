// Which offsets does the gfx950 raw-buffer range check see?  num_records = 1024 bytes over a 64 KiB
// allocation filled with (index + 1); a load that the hardware takes as out of range returns 0.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/buffer_range tools/microbench/buffer_range.hip && /tmp/buffer_range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const unsigned* buf, unsigned* out, int nrec) {
  const __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, nrec, 0x00020000);
  const int cases[8][2] = {{0, 0}, {0, 2048}, {512, 768}, {1020, 0}, {1024, 0}, {0, 1020}, {0, 1024}, {(int)0x80000000, 64}};
  for (int c = 0; c < 8; ++c) {
    int voff = cases[c][0], soff = cases[c][1];
    asm volatile("" : "+v"(voff));
    const int s = __builtin_amdgcn_readfirstlane(soff);
    out[c] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, s, 0);
  }
}

int main() {
  const int n = 16384;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = i + 1;
  unsigned *d, *o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 64);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d, o, 1024);
  unsigned r[8];
  hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
  const char* what[8] = {"voff 0, soff 0 (in range)", "voff 0, soff 2048 (soffset > num_records)",
                         "voff 512, soff 768 (sum > num_records, each below)", "voff 1020, soff 0 (last dword)",
                         "voff 1024, soff 0 (first dword past the end)", "voff 0, soff 1020", "voff 0, soff 1024",
                         "voff 0x80000000, soff 64"};
  for (int c = 0; c < 8; ++c) printf("%-55s -> %u%s\n", what[c], r[c], r[c] ? "" : "   (out of range: zero)");
  return 0;
}
