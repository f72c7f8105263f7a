// integration/ms_deform_attn_pave.cpp -- the reference-side binding of libpave_hip.so.
//
// This is the ONE translation unit a maintainer of zgspose/PAVENet's vendored mmcv adds
// (as third_party/mmcv/mmcv/ops/csrc/pytorch/hip/ms_deform_attn_pave.cpp, linked with
// -lpave_hip) to make `mmcv._ext.ms_deform_attn_forward / _backward`
// (csrc/pytorch/pybind.cpp:737-748 -> csrc/pytorch/ms_deform_attn.cpp:15-46 ->
// DISPATCH_DEVICE_IMPL) run on the MI355X kernels.  It replaces the registrations of
// csrc/pytorch/cuda/cudabind.cpp:736-765; on PyTorch-ROCm the device type is still `CUDA`.
//
// It is compiled (syntax + types, against the installed torch headers and the reference's own
// pytorch_cpp_helper.hpp / pytorch_device_registry.hpp) by
// tests/test_host_cpu.py::test_integration_binding_compiles, so INTEGRATION.md cannot rot.
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>  // the current stream on ROCm builds

#include "pave_hip.h"
#include "pytorch_cpp_helper.hpp"
#include "pytorch_device_registry.hpp"

namespace {

struct Dims {
  int bs, S, M, D, L, Lq, P;
};

Dims check_and_dims(const Tensor& value, const Tensor& spatial_shapes,
                    const Tensor& level_start_index, const Tensor& sampling_loc,
                    const Tensor& attn_weight) {
  // the pre-conditions of ms_deform_attn_cuda.cu:215-230
  AT_ASSERTM(value.is_contiguous(), "value tensor has to be contiguous");
  AT_ASSERTM(spatial_shapes.is_contiguous(), "spatial_shapes tensor has to be contiguous");
  AT_ASSERTM(level_start_index.is_contiguous(), "level_start_index tensor has to be contiguous");
  AT_ASSERTM(sampling_loc.is_contiguous(), "sampling_loc tensor has to be contiguous");
  AT_ASSERTM(attn_weight.is_contiguous(), "attn_weight tensor has to be contiguous");
  AT_ASSERTM(value.is_cuda(), "value must be a CUDA tensor");
  AT_ASSERTM(spatial_shapes.is_cuda(), "spatial_shapes must be a CUDA tensor");
  AT_ASSERTM(level_start_index.is_cuda(), "level_start_index must be a CUDA tensor");
  AT_ASSERTM(sampling_loc.is_cuda(), "sampling_loc must be a CUDA tensor");
  AT_ASSERTM(attn_weight.is_cuda(), "attn_weight must be a CUDA tensor");
  Dims d;
  d.bs = (int)value.size(0), d.S = (int)value.size(1), d.M = (int)value.size(2);
  d.D = (int)value.size(3), d.L = (int)spatial_shapes.size(0);
  d.Lq = (int)sampling_loc.size(1), d.P = (int)sampling_loc.size(4);
  return d;
}

void* current_stream() {
  return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
}

}  // namespace

Tensor ms_deform_attn_pave_forward(const Tensor& value, const Tensor& spatial_shapes,
                                   const Tensor& level_start_index, const Tensor& sampling_loc,
                                   const Tensor& attn_weight, const int im2col_step) {
  const Dims d = check_and_dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight);
  auto out = at::empty({d.bs, d.Lq, d.M * d.D}, value.options());  // fully overwritten
  int rc;
  if (value.scalar_type() == at::kFloat) {
    rc = pave_ms_deform_attn_forward_f32(
        value.data_ptr<float>(), spatial_shapes.data_ptr<int64_t>(),
        level_start_index.data_ptr<int64_t>(), sampling_loc.data_ptr<float>(),
        attn_weight.data_ptr<float>(), out.data_ptr<float>(), d.bs, d.S, d.M, d.D, d.L, d.Lq, d.P,
        im2col_step, current_stream());
  } else {
    AT_ASSERTM(value.scalar_type() == at::kDouble, "ms_deform_attn_forward: float or double");
    rc = pave_ms_deform_attn_forward_f64(
        value.data_ptr<double>(), spatial_shapes.data_ptr<int64_t>(),
        level_start_index.data_ptr<int64_t>(), sampling_loc.data_ptr<double>(),
        attn_weight.data_ptr<double>(), out.data_ptr<double>(), d.bs, d.S, d.M, d.D, d.L, d.Lq,
        d.P, im2col_step, current_stream());
  }
  TORCH_CHECK(rc == PAVE_OK, "ms_deform_attn_forward: ", pave_last_error());
  return out;
}

void ms_deform_attn_pave_backward(const Tensor& value, const Tensor& spatial_shapes,
                                  const Tensor& level_start_index, const Tensor& sampling_loc,
                                  const Tensor& attn_weight, const Tensor& grad_output,
                                  Tensor& grad_value, Tensor& grad_sampling_loc,
                                  Tensor& grad_attn_weight, const int im2col_step) {
  const Dims d = check_and_dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight);
  AT_ASSERTM(grad_output.is_contiguous(), "grad_output tensor has to be contiguous");
  AT_ASSERTM(grad_output.is_cuda(), "grad_output must be a CUDA tensor");
  // grad_value is ACCUMULATED into the caller's zeroed tensor (MO:72-86); the other two are
  // overwritten
  int rc;
  if (value.scalar_type() == at::kFloat) {
    rc = pave_ms_deform_attn_backward_f32(
        value.data_ptr<float>(), spatial_shapes.data_ptr<int64_t>(),
        level_start_index.data_ptr<int64_t>(), sampling_loc.data_ptr<float>(),
        attn_weight.data_ptr<float>(), grad_output.data_ptr<float>(), grad_value.data_ptr<float>(),
        grad_sampling_loc.data_ptr<float>(), grad_attn_weight.data_ptr<float>(), d.bs, d.S, d.M,
        d.D, d.L, d.Lq, d.P, im2col_step, current_stream());
  } else {
    AT_ASSERTM(value.scalar_type() == at::kDouble, "ms_deform_attn_backward: float or double");
    rc = pave_ms_deform_attn_backward_f64(
        value.data_ptr<double>(), spatial_shapes.data_ptr<int64_t>(),
        level_start_index.data_ptr<int64_t>(), sampling_loc.data_ptr<double>(),
        attn_weight.data_ptr<double>(), grad_output.data_ptr<double>(),
        grad_value.data_ptr<double>(), grad_sampling_loc.data_ptr<double>(),
        grad_attn_weight.data_ptr<double>(), d.bs, d.S, d.M, d.D, d.L, d.Lq, d.P, im2col_step,
        current_stream());
  }
  TORCH_CHECK(rc == PAVE_OK, "ms_deform_attn_backward: ", pave_last_error());
}

// declared in csrc/pytorch/ms_deform_attn.cpp:15-39
Tensor ms_deform_attn_impl_forward(const Tensor& value, const Tensor& spatial_shapes,
                                   const Tensor& level_start_index, const Tensor& sampling_loc,
                                   const Tensor& attn_weight, const int im2col_step);
void ms_deform_attn_impl_backward(const Tensor& value, const Tensor& spatial_shapes,
                                  const Tensor& level_start_index, const Tensor& sampling_loc,
                                  const Tensor& attn_weight, const Tensor& grad_output,
                                  Tensor& grad_value, Tensor& grad_sampling_loc,
                                  Tensor& grad_attn_weight, const int im2col_step);

REGISTER_DEVICE_IMPL(ms_deform_attn_impl_forward, CUDA, ms_deform_attn_pave_forward);
REGISTER_DEVICE_IMPL(ms_deform_attn_impl_backward, CUDA, ms_deform_attn_pave_backward);
