// PyTorch-ROCm custom-op registration of the hot path: torch.ops.osi.* (TORCH_LIBRARY), the form BASELINE.json's north_star names
// ("surfaced through PyTorch-ROCm custom ops") for the arithmetic the reference runs at openset_imagenet/train.py:132-139:
//
//   torch.ops.osi.resnet50_forward    model.py:28-39   (logits, features = model(images))
//   torch.ops.osi.resnet50_backward   train.py:138     (j.backward() through the network, stage range for the DP bucket schedule)
//   torch.ops.osi.resnet50_grads_ready  config/train.yaml:18,35-39 (hand a finished stage's gradients to the communication stream)
//   torch.ops.osi.loss_fwd_bwd        losses.py:16-29, train.py:343-347 (the three losses + objectosphere term, value and gradient)
//   torch.ops.osi.adam_step / sgd_step    train.py:139, 356-359
//   torch.ops.osi.stage_canvas        train.py:259-263 after decode + resize (crop, flip, ToTensor, NHWC4)
//   torch.ops.osi.softmax / confidence_accumulate    train.py:177, metrics.py:8-42
//
// This file contains no arithmetic: every op validates its tensor arguments (device, dtype, contiguity, sizes), takes the CURRENT
// HIP stream of the tensors' device inside the op, and forwards raw pointers to the C ABI of libosi_hip.so (include/osi.h), which
// stays the boundary a non-Python host binds. Tensor arguments keep their storage alive for the duration of the call and the
// dispatcher / profiler see the ops by name; stream and lifetime safety no longer depend on the caller passing data_ptr()s.
#include <torch/library.h>
#include <ATen/ATen.h>
// PyTorch-ROCm keeps the device type spelled "cuda"; these are its own HIP stream / guard classes for that spelling
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <optional>
#include <string>
#include <tuple>

#include "../../include/osi.h"

namespace {

using at::Tensor;

void ok(int code, const char* what) {
    TORCH_CHECK(code == OSI_OK, "libosi_hip ", what, " failed: ", osi_strerror(code), " (code ", code, ")");
}

// align = 16 for the tensors the kernels read with 16-byte vector accesses (the arenas, images, the workspace); the small per-row
// tensors (logits, targets, features, class weights, gradients of the head) are read element-wise and only need their natural
// alignment, so a contiguous row slice such as logits[3:] with an odd class count is accepted
void need(const Tensor& t, at::ScalarType dt, const char* name, uintptr_t align = 16) {
    TORCH_CHECK(t.defined(), name, ": undefined tensor");
    TORCH_CHECK(t.is_cuda(), "osi::", name, " must live on the GPU: the MI355X build has no CPU path");
    TORCH_CHECK(t.scalar_type() == dt, "osi::", name, " has dtype ", t.scalar_type(), ", expected ", dt);
    TORCH_CHECK(t.is_contiguous(), "osi::", name, " must be contiguous");
    TORCH_CHECK((reinterpret_cast<uintptr_t>(t.data_ptr()) & (align - 1)) == 0, "osi::", name, " must be ", align, "-byte aligned");
}
constexpr uintptr_t NATURAL = 4;   // element-wise readers: fp32 / int64 / fp64 tensors from the allocator or row slices of them

osi_stream_t stream_of(const Tensor& t) {
    return (osi_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

osi_resnet50_t handle(int64_t h) {
    TORCH_CHECK(h != 0, "osi: null executor handle");
    return reinterpret_cast<osi_resnet50_t>(static_cast<uintptr_t>(h));
}

const float* fptr(const std::optional<Tensor>& t) { return t.has_value() && t->defined() ? t->data_ptr<float>() : nullptr; }

// image: fp32 NCHW [B,3,H,W] (the reference's batch), fp32 NHWC4 [B,H,W,4] (bound in place) or uint8 [B,H,W,3] (+ flip flags)
std::tuple<Tensor, Tensor> resnet50_forward(int64_t net, const Tensor& params, Tensor buffers, Tensor nbt, const Tensor& image,
                                            const std::optional<Tensor>& flip, Tensor workspace, int64_t fc_dim, int64_t out_features,
                                            bool training) {
    need(params, at::kFloat, "params"); need(buffers, at::kFloat, "buffers"); need(nbt, at::kLong, "num_batches_tracked");
    need(workspace, at::kByte, "workspace");
    TORCH_CHECK(image.dim() == 4, "osi::resnet50_forward: image must be 4-D");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(params.device());
    osi_resnet50_t h = handle(net);
    TORCH_CHECK((size_t)workspace.numel() >= osi_resnet50_workspace_bytes(h), "osi::resnet50_forward: workspace too small");
    TORCH_CHECK((size_t)params.numel() == osi_resnet50_param_floats(h), "osi::resnet50_forward: parameter arena size mismatch");
    TORCH_CHECK((size_t)buffers.numel() == osi_resnet50_buffer_floats(h), "osi::resnet50_forward: buffer arena size mismatch");
    TORCH_CHECK(nbt.numel() == osi_resnet50_num_bn(h), "osi::resnet50_forward: num_batches_tracked size mismatch");
    osi_stream_t st = stream_of(params);
    const int64_t B = image.size(0);
    {   // the executor was created for ONE geometry: a batch of another shape would read / write outside its workspace
        int eb = 0, eh = 0, ew = 0;
        ok(osi_resnet50_geometry(h, &eb, &eh, &ew), "osi_resnet50_geometry");
        const bool chw = image.scalar_type() != at::kByte && !(image.size(3) == 4 && image.size(1) != 3);
        const int64_t ih = chw ? image.size(2) : image.size(1), iw = chw ? image.size(3) : image.size(2);
        TORCH_CHECK(B == eb && ih == eh && iw == ew, "osi::resnet50_forward: image batch ", B, "x", ih, "x", iw,
                    " does not match the executor's geometry ", eb, "x", eh, "x", ew);
    }
    const float* img = nullptr;
    if (image.scalar_type() == at::kByte) {
        need(image, at::kByte, "image"); TORCH_CHECK(image.size(3) == 3, "uint8 image batch must be [B,H,W,3]");
        const unsigned char* fl = nullptr;
        if (flip.has_value() && flip->defined()) { need(*flip, at::kByte, "flip"); TORCH_CHECK(flip->numel() == B); fl = flip->data_ptr<unsigned char>(); }
        ok(osi_resnet50_stage_input_u8(h, image.data_ptr<unsigned char>(), fl, workspace.data_ptr(), st), "osi_resnet50_stage_input_u8");
    } else if (image.size(3) == 4 && image.size(1) != 3) {
        need(image, at::kFloat, "image");
        ok(osi_resnet50_bind_input_nhwc4(h, image.data_ptr<float>()), "osi_resnet50_bind_input_nhwc4");
    } else {
        need(image, at::kFloat, "image"); TORCH_CHECK(image.size(1) == 3, "fp32 image batch must be [B,3,H,W]");
        img = image.data_ptr<float>();
    }
    Tensor logits = at::empty({B, out_features}, params.options());
    Tensor features = at::empty({B, fc_dim}, params.options());
    ok(osi_resnet50_forward(h, params.data_ptr<float>(), buffers.data_ptr<float>(), (long long*)nbt.data_ptr<int64_t>(), img,
                            workspace.data_ptr(), logits.data_ptr<float>(), features.data_ptr<float>(), training ? 1 : 0, st),
       "osi_resnet50_forward");
    return {logits, features};
}

void resnet50_backward(int64_t net, const Tensor& params, Tensor grads, Tensor workspace, const Tensor& dlogits,
                       const std::optional<Tensor>& dfeatures, int64_t stage_lo, int64_t stage_hi) {
    need(params, at::kFloat, "params"); need(grads, at::kFloat, "grads"); need(workspace, at::kByte, "workspace");
    need(dlogits, at::kFloat, "dlogits", NATURAL);
    if (dfeatures.has_value() && dfeatures->defined()) need(*dfeatures, at::kFloat, "dfeatures", NATURAL);
    TORCH_CHECK(grads.numel() == params.numel(), "osi::resnet50_backward: gradient arena size mismatch");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(params.device());
    ok(osi_resnet50_backward(handle(net), params.data_ptr<float>(), grads.data_ptr<float>(), workspace.data_ptr(),
                             dlogits.data_ptr<float>(), fptr(dfeatures), (int)stage_lo, (int)stage_hi, stream_of(params)),
       "osi_resnet50_backward");
}

// Data parallel: the stream `waiter` (a raw hipStream_t handle: torch.cuda.Stream.cuda_stream of the communication stream) waits for the
// gradients of the backward stages enqueued so far on the CURRENT stream and on the executor's side stream; the current stream waits for nothing
void resnet50_grads_ready(int64_t net, const Tensor& grads, int64_t waiter) {
    need(grads, at::kFloat, "grads");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(grads.device());
    ok(osi_resnet50_grads_ready(handle(net), stream_of(grads), reinterpret_cast<osi_stream_t>(static_cast<uintptr_t>(waiter))),
       "osi_resnet50_grads_ready");
}

// returns (loss [], dlogits [B,C] or empty, dfeatures [B,F] or empty)
std::tuple<Tensor, Tensor, Tensor> loss_fwd_bwd(int64_t mode, const Tensor& logits, const Tensor& target, double unk_weight,
                                                int64_t ignore_index, const std::optional<Tensor>& class_weights,
                                                const std::optional<Tensor>& features, double xi, double alpha, bool need_grad) {
    need(logits, at::kFloat, "logits", NATURAL); need(target, at::kLong, "target", NATURAL);
    TORCH_CHECK(logits.dim() == 2 && target.dim() == 1 && target.size(0) == logits.size(0), "expected logits [B, C] and target [B]");
    const int B = (int)logits.size(0), C = (int)logits.size(1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(logits.device());
    Tensor loss = at::empty({}, logits.options());
    Tensor dlogits = need_grad ? at::empty_like(logits) : at::empty({0}, logits.options());
    Tensor dfeat = at::empty({0}, logits.options());
    int F = 0;
    if (class_weights.has_value() && class_weights->defined()) { need(*class_weights, at::kFloat, "class_weights", NATURAL); TORCH_CHECK(class_weights->numel() == C); }
    if (features.has_value() && features->defined()) {
        need(*features, at::kFloat, "features", NATURAL); TORCH_CHECK(features->dim() == 2 && features->size(0) == B);
        F = (int)features->size(1);
        dfeat = at::empty_like(*features);
    }
    ok(osi_loss_fwd_bwd((int)mode, logits.data_ptr<float>(), (const long long*)target.data_ptr<int64_t>(), B, C, (float)unk_weight,
                        (long long)ignore_index, fptr(class_weights), fptr(features), F, (float)xi, (float)alpha, loss.data_ptr<float>(),
                        need_grad ? dlogits.data_ptr<float>() : nullptr, F ? dfeat.data_ptr<float>() : nullptr, stream_of(logits)),
       "osi_loss_fwd_bwd");
    return {loss, dlogits, dfeat};
}

void adam_step(Tensor params, const Tensor& grads, Tensor exp_avg, Tensor exp_avg_sq, double lr, double beta1, double beta2, double eps,
               int64_t step, double grad_scale) {
    need(params, at::kFloat, "params"); need(grads, at::kFloat, "grads"); need(exp_avg, at::kFloat, "exp_avg"); need(exp_avg_sq, at::kFloat, "exp_avg_sq");
    TORCH_CHECK(grads.numel() == params.numel() && exp_avg.numel() == params.numel() && exp_avg_sq.numel() == params.numel());
    c10::hip::HIPGuardMasqueradingAsCUDA guard(params.device());
    ok(osi_adam_step(params.data_ptr<float>(), grads.data_ptr<float>(), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(),
                     (size_t)params.numel(), lr, beta1, beta2, eps, (long long)step, (float)grad_scale, stream_of(params)), "osi_adam_step");
}

void sgd_step(Tensor params, const Tensor& grads, Tensor momentum_buffer, double lr, double momentum, bool first, double grad_scale) {
    need(params, at::kFloat, "params"); need(grads, at::kFloat, "grads"); need(momentum_buffer, at::kFloat, "momentum_buffer");
    TORCH_CHECK(grads.numel() == params.numel() && momentum_buffer.numel() == params.numel());
    c10::hip::HIPGuardMasqueradingAsCUDA guard(params.device());
    ok(osi_sgd_step(params.data_ptr<float>(), grads.data_ptr<float>(), momentum_buffer.data_ptr<float>(), (size_t)params.numel(),
                    (float)lr, (float)momentum, first ? 1 : 0, (float)grad_scale, stream_of(params)), "osi_sgd_step");
}

Tensor stage_canvas(const Tensor& canvas, const std::optional<Tensor>& crop_xy, const std::optional<Tensor>& flip, int64_t H, int64_t W) {
    need(canvas, at::kByte, "canvas");
    TORCH_CHECK(canvas.dim() == 4 && canvas.size(3) == 3, "canvas must be uint8 [B,Hc,Wc,3]");
    const int B = (int)canvas.size(0);
    const int* cp = nullptr; const unsigned char* fl = nullptr;
    if (crop_xy.has_value() && crop_xy->defined()) { need(*crop_xy, at::kInt, "crop_xy", NATURAL); TORCH_CHECK(crop_xy->numel() == 2 * B); cp = crop_xy->data_ptr<int>(); }
    if (flip.has_value() && flip->defined()) { need(*flip, at::kByte, "flip", 1); TORCH_CHECK(flip->numel() == B); fl = flip->data_ptr<unsigned char>(); }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(canvas.device());
    Tensor out = at::empty({B, H, W, 4}, canvas.options().dtype(at::kFloat));
    ok(osi_u8_crop_flip_to_nhwc4(canvas.data_ptr<unsigned char>(), cp, fl, out.data_ptr<float>(), B, (int)canvas.size(1), (int)canvas.size(2),
                                 (int)H, (int)W, stream_of(canvas)), "osi_u8_crop_flip_to_nhwc4");
    return out;
}

Tensor softmax(const Tensor& logits) {
    need(logits, at::kFloat, "logits", NATURAL); TORCH_CHECK(logits.dim() == 2);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(logits.device());
    Tensor out = at::empty_like(logits);
    ok(osi_softmax(logits.data_ptr<float>(), out.data_ptr<float>(), (int)logits.size(0), (int)logits.size(1), stream_of(logits)), "osi_softmax");
    return out;
}

void confidence_accumulate(const Tensor& logits, const Tensor& target, double offset, int64_t unknown_class, int64_t last_valid_class,
                           Tensor acc4) {
    need(logits, at::kFloat, "logits", NATURAL); need(target, at::kLong, "target", NATURAL); need(acc4, at::kDouble, "acc4", NATURAL);
    TORCH_CHECK(logits.dim() == 2 && target.numel() == logits.size(0) && acc4.numel() == 4);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(logits.device());
    ok(osi_confidence_accumulate(logits.data_ptr<float>(), (const long long*)target.data_ptr<int64_t>(), (int)logits.size(0),
                                 (int)logits.size(1), (float)offset, (long long)unknown_class, (int)last_valid_class,
                                 acc4.data_ptr<double>(), stream_of(logits)), "osi_confidence_accumulate");
}

}  // namespace

TORCH_LIBRARY(osi, m) {
    m.def("resnet50_forward(int net, Tensor params, Tensor(a!) buffers, Tensor(b!) nbt, Tensor image, Tensor? flip, Tensor(c!) workspace, "
          "int fc_dim, int out_features, bool training) -> (Tensor, Tensor)");
    m.def("resnet50_backward(int net, Tensor params, Tensor(a!) grads, Tensor(b!) workspace, Tensor dlogits, Tensor? dfeatures, "
          "int stage_lo, int stage_hi) -> ()");
    m.def("resnet50_grads_ready(int net, Tensor grads, int waiter_stream) -> ()");
    m.def("loss_fwd_bwd(int mode, Tensor logits, Tensor target, float unk_weight, int ignore_index, Tensor? class_weights, "
          "Tensor? features, float xi, float alpha, bool need_grad) -> (Tensor, Tensor, Tensor)");
    m.def("adam_step(Tensor(a!) params, Tensor grads, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float lr, float beta1, float beta2, "
          "float eps, int step, float grad_scale) -> ()");
    m.def("sgd_step(Tensor(a!) params, Tensor grads, Tensor(b!) momentum_buffer, float lr, float momentum, bool first, float grad_scale) -> ()");
    m.def("stage_canvas(Tensor canvas, Tensor? crop_xy, Tensor? flip, int H, int W) -> Tensor");   // crop corners are read-only (clamped in registers)
    m.def("softmax(Tensor logits) -> Tensor");
    m.def("confidence_accumulate(Tensor logits, Tensor target, float offset, int unknown_class, int last_valid_class, Tensor(a!) acc4) -> ()");
}

// The ops are MI355X-only: registered for the HIP device key (spelled CUDA in PyTorch-ROCm); a CPU tensor finds no kernel and raises.
TORCH_LIBRARY_IMPL(osi, CUDA, m) {
    m.impl("resnet50_forward", &resnet50_forward);
    m.impl("resnet50_backward", &resnet50_backward);
    m.impl("resnet50_grads_ready", &resnet50_grads_ready);
    m.impl("loss_fwd_bwd", &loss_fwd_bwd);
    m.impl("adam_step", &adam_step);
    m.impl("sgd_step", &sgd_step);
    m.impl("stage_canvas", &stage_canvas);
    m.impl("softmax", &softmax);
    m.impl("confidence_accumulate", &confidence_accumulate);
}
