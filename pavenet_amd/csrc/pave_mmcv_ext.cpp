// pavenet_amd/_ext -- the pybind module form of the drop-in boundary.
//
// Exports `ms_deform_attn_forward` / `ms_deform_attn_backward` with the argument list AND keyword
// names of `mmcv._ext` (third_party/mmcv/mmcv/ops/csrc/pytorch/pybind.cpp:160-173 declarations,
// :737-748 `m.def(... py::arg("value"), py::arg("value_spatial_shapes"), ...)`), so that
// `mmcv/ops/multi_scale_deform_attn.py:47-53,76-86` (`ext_module.ms_deform_attn_forward(value,
// value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
// im2col_step=...)`) runs on it unchanged.  Each function checks what
// csrc/pytorch/cuda/ms_deform_attn_cuda.cu:215-245 asserts (contiguity, device, batch %
// im2col_step), guards the device (ms_deform_attn.cpp:43), takes torch's current stream and calls
// the C ABI of libpave_hip.so (include/pave_hip.h); kernel errors raise (the reference only
// printf's them, ms_deform_attn_cuda.cu:41-44).  Uses the installed torch headers only.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/extension.h>

#include "pave_hip.h"

namespace {

using at::Tensor;

struct Dims {
  int bs, S, M, D, L, Lq, P;
};

void need(const Tensor& t, const char* name, at::ScalarType st) {
  TORCH_CHECK(t.is_contiguous(), name, " tensor has to be contiguous");
  TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");
  TORCH_CHECK(t.scalar_type() == st, name, " has the wrong dtype");
}

Dims dims_of(const Tensor& value, const Tensor& shapes, const Tensor& lsi, const Tensor& loc,
             const Tensor& aw) {
  TORCH_CHECK(value.scalar_type() == at::kFloat || value.scalar_type() == at::kDouble,
              "ms_deform_attn: value must be float32 or float64");
  need(value, "value", value.scalar_type());
  need(shapes, "spatial_shapes", at::kLong);
  need(lsi, "level_start_index", at::kLong);
  need(loc, "sampling_loc", value.scalar_type());
  need(aw, "attn_weight", value.scalar_type());
  TORCH_CHECK(value.dim() == 4 && loc.dim() == 6 && aw.dim() == 5 && shapes.dim() == 2 &&
                  lsi.dim() == 1,
              "ms_deform_attn: bad tensor ranks");
  Dims d;
  d.bs = (int)value.size(0), d.S = (int)value.size(1), d.M = (int)value.size(2);
  d.D = (int)value.size(3), d.L = (int)shapes.size(0);
  d.Lq = (int)loc.size(1), d.P = (int)loc.size(4);
  TORCH_CHECK(loc.size(0) == d.bs && loc.size(2) == d.M && loc.size(3) == d.L && loc.size(5) == 2,
              "ms_deform_attn: sampling_loc shape mismatch");
  TORCH_CHECK(aw.size(0) == d.bs && aw.size(1) == d.Lq && aw.size(2) == d.M && aw.size(3) == d.L &&
                  aw.size(4) == d.P,
              "ms_deform_attn: attn_weight shape mismatch");
  TORCH_CHECK(shapes.size(1) == 2 && lsi.size(0) == d.L,
              "ms_deform_attn: spatial_shapes / level_start_index shape mismatch");
  return d;
}

void* stream_now() { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

Tensor ms_deform_attn_forward(const Tensor& value, const Tensor& spatial_shapes,
                              const Tensor& level_start_index, const Tensor& sampling_loc,
                              const Tensor& attn_weight, const int im2col_step) {
  const Dims d = dims_of(value, spatial_shapes, level_start_index, sampling_loc, attn_weight);
  const c10::DeviceGuard guard(value.device());
  Tensor out = at::empty({d.bs, d.Lq, d.M * d.D}, value.options());
  if (out.numel() == 0) return out;
  int rc;
  if (value.scalar_type() == at::kFloat)
    rc = pave_ms_deform_attn_forward_f32(
        value.data_ptr<float>(), spatial_shapes.data_ptr<int64_t>(),
        level_start_index.data_ptr<int64_t>(), sampling_loc.data_ptr<float>(),
        attn_weight.data_ptr<float>(), out.data_ptr<float>(), d.bs, d.S, d.M, d.D, d.L, d.Lq, d.P,
        im2col_step, stream_now());
  else
    rc = pave_ms_deform_attn_forward_f64(
        value.data_ptr<double>(), spatial_shapes.data_ptr<int64_t>(),
        level_start_index.data_ptr<int64_t>(), sampling_loc.data_ptr<double>(),
        attn_weight.data_ptr<double>(), out.data_ptr<double>(), d.bs, d.S, d.M, d.D, d.L, d.Lq,
        d.P, im2col_step, stream_now());
  TORCH_CHECK(rc == PAVE_OK, "ms_deform_attn_forward: ", pave_last_error());
  return out;
}

void ms_deform_attn_backward(const Tensor& value, const Tensor& spatial_shapes,
                             const Tensor& level_start_index, const Tensor& sampling_loc,
                             const Tensor& attn_weight, const Tensor& grad_output,
                             Tensor& grad_value, Tensor& grad_sampling_loc,
                             Tensor& grad_attn_weight, const int im2col_step) {
  const Dims d = dims_of(value, spatial_shapes, level_start_index, sampling_loc, attn_weight);
  const auto st = value.scalar_type();
  need(grad_output, "grad_output", st);
  need(grad_value, "grad_value", st);
  need(grad_sampling_loc, "grad_sampling_loc", st);
  need(grad_attn_weight, "grad_attn_weight", st);
  TORCH_CHECK(grad_output.numel() == (int64_t)d.bs * d.Lq * d.M * d.D &&
                  grad_value.sizes() == value.sizes() &&
                  grad_sampling_loc.sizes() == sampling_loc.sizes() &&
                  grad_attn_weight.sizes() == attn_weight.sizes(),
              "ms_deform_attn_backward: gradient shapes mismatch");
  if (grad_output.numel() == 0) return;
  const c10::DeviceGuard guard(value.device());
  int rc;
  if (st == at::kFloat)
    rc = pave_ms_deform_attn_backward_f32(
        value.data_ptr<float>(), spatial_shapes.data_ptr<int64_t>(),
        level_start_index.data_ptr<int64_t>(), sampling_loc.data_ptr<float>(),
        attn_weight.data_ptr<float>(), grad_output.data_ptr<float>(), grad_value.data_ptr<float>(),
        grad_sampling_loc.data_ptr<float>(), grad_attn_weight.data_ptr<float>(), d.bs, d.S, d.M,
        d.D, d.L, d.Lq, d.P, im2col_step, stream_now());
  else
    rc = pave_ms_deform_attn_backward_f64(
        value.data_ptr<double>(), spatial_shapes.data_ptr<int64_t>(),
        level_start_index.data_ptr<int64_t>(), sampling_loc.data_ptr<double>(),
        attn_weight.data_ptr<double>(), grad_output.data_ptr<double>(),
        grad_value.data_ptr<double>(), grad_sampling_loc.data_ptr<double>(),
        grad_attn_weight.data_ptr<double>(), d.bs, d.S, d.M, d.D, d.L, d.Lq, d.P, im2col_step,
        stream_now());
  TORCH_CHECK(rc == PAVE_OK, "ms_deform_attn_backward: ", pave_last_error());
}

}  // namespace

PYBIND11_MODULE(_ext, m) {
  m.doc() = "mmcv._ext-compatible ms_deform_attn entry points on libpave_hip.so (MI355X)";
  m.def("ms_deform_attn_forward", &ms_deform_attn_forward, "forward function of ms_deform_attn",
        py::arg("value"), py::arg("value_spatial_shapes"), py::arg("value_level_start_index"),
        py::arg("sampling_locations"), py::arg("attention_weights"), py::arg("im2col_step"));
  m.def("ms_deform_attn_backward", &ms_deform_attn_backward, "backward function of ms_deform_attn",
        py::arg("value"), py::arg("value_spatial_shapes"), py::arg("value_level_start_index"),
        py::arg("sampling_locations"), py::arg("attention_weights"), py::arg("grad_output"),
        py::arg("grad_value"), py::arg("grad_sampling_loc"), py::arg("grad_attn_weight"),
        py::arg("im2col_step"));
  m.def("pave_abi_version", []() { return pave_abi_version(); });
}
