/* Shared between the translation units of libpave_hip.so (not part of the C ABI). */
#ifndef PAVE_INTERNAL_H_
#define PAVE_INTERNAL_H_
int pave_internal_fail(int code, const char* msg); /* records pave_last_error(), returns code */
// pave_gemm_dma.hip: the LDS-DMA generation of the 3-plane split GEMM (rows / 3x3 / strided rows)
int pave_internal_gemm_q(const float* a, const float* a_bias, const void* w_planes, const float* bias,
                         const float* residual, long long residual_rows, float* out, float* out2,
                         int n_split, long long M, int K, int N, int relu, int kind, int H, int W,
                         int Cin, int Ho, int Wo, int stride, void* stream, const float* a2 = nullptr,
                         int n_real = 0, int ksplit = 1, int ks_slabs = 0, int planes = 3);
// (planes: 3 = the exact bf16 split, 1 = one plane of fp16 operands)
// split-K (few output rows, long K): plan, and the ordered sum of the parts + bias / residual / ReLU
void pave_internal_splitk_plan(long long M, int Kp, int Np, int* ksplit, int* ks_slabs);
int pave_internal_splitk_reduce(const float* ws, int parts, long long M, int n, const float* bias,
                                const float* residual, int relu, float* out, void* stream);
int pave_internal_gemm_q_ln(const float* a, const void* w_planes, const float* bias, const float* residual,
                            const float* gamma, const float* beta, float eps, float* out, long long M,
                            int K, int N, void* stream, int planes = 3);
int pave_internal_gemm_encproj(const float* a, const void* w_planes, const float* table, long long table_rows,
                               const float* value_bias, const float* ref, const int* levels_hw, float* value,
                               float* samp, long long M, int K, void* stream, int planes = 3);
/* Kernel-form override for A/B runs and the form-equality tests.  Only the -DPAVE_DIAG build
   (lib/libpave_hip_diag.so, loaded by tests/ and tools/ through native.diag_build()) has the
   process-global and its setter pave_diag_gemm_variant(); in the shipped library the form
   selection is a compile-time constant. */
#ifdef PAVE_DIAG
int pave_internal_diag_variant();
#else
static inline int pave_internal_diag_variant() { return 0; }
#endif
int pave_internal_gemm_f16act(const void* a, int a_f16, const void* w_plane, const float* bias, const float* residual,
                              const float* gamma, const float* beta, float eps, void* out, int out_f16, long long M,
                              int K, int N, int relu, void* stream);
int pave_internal_stem7x7_q(const float* x, const void* w_stem, const float* bias, float* y, int N,
                            int H, int W, int pitch, int relu, void* stream, int planes = 3);
#endif /* PAVE_INTERNAL_H_ */
