// dmel_torch.cpp -- the torch-registered form of the boundary: TORCH_LIBRARY(dmel, ...) over the C ABI of include/dmel.h.
//
// the reference's boundary is the nn.Module MelSpectrogramLayer (models.py:14-56); what replaces its
// body is a torch extension with
//     dmel::forward(x, lambd, plan, flags, eps, want_tangent, lambd_sync, out_bf16) -> (out, tangent_buf)
//     dmel::backward(grad_out, tangent_buf, plan)                                      -> d lambd
//     dmel::mel_spectrogram(x, lambd, plan, flags, eps, lambd_sync, out_bf16)          -> out      (differentiable in lambd)
//     dmel::mel_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate)                     -> (n_freqs, n_mels) fp32
// This file holds no arithmetic: it allocates outputs with at::empty on x's device, takes the current HIP stream
// (forward on the caller's thread, backward on the autograd engine's thread: models.py:33, train.py:47) and calls
// libdmel_hip.so.  With lambd_sync = false lambd never leaves the device (dmel_forward_dev): nothing in a training
// step waits for the host, and the step can be captured into a HIP graph.  Built in-tree by build.py with g++ (no device
// code here), loaded with torch.ops.load_library.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <string>
#include <tuple>

#include "../../include/dmel.h"

namespace {

inline dmel_plan* as_plan(int64_t h)
{
    TORCH_CHECK(h != 0, "dmel: plan handle is 0 (closed or never created)");
    dmel_plan* p = reinterpret_cast<dmel_plan*>(static_cast<intptr_t>(h));
    // the handle travels as an integer: refuse anything that is not a live plan of this process instead of dereferencing it
    TORCH_CHECK(dmel_plan_is_live(p) == 1, "dmel: ", h, " is not the handle of a live plan (destroyed, or not a plan at all)");
    return p;
}

inline void check(dmel_status st)
{
    TORCH_CHECK(st == DMEL_OK, "libdmel_hip: status ", (int)st, ": ", dmel_last_error());
}

inline void* stream_of(const at::Tensor& t)
{
    return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

constexpr int64_t kAlign = 256;     // the scratch block sits behind the tangent in one allocation, 256-byte aligned

// (out, tangent_buf): tangent_buf is a flat fp32 tensor holding the tangent (out.numel() values, empty when no tangent is
// wanted) followed, 256-byte aligned, by the scratch block of this call (dmel_scratch_bytes): the ticket and partials of
// the reduction in dmel::backward live there, so nothing plan-owned is shared between steps on different streams.
std::tuple<at::Tensor, at::Tensor> dmel_forward_op(const at::Tensor& x, const at::Tensor& lambd, int64_t plan_h, int64_t flags,
                                                   double eps, bool want_tangent, bool lambd_sync, bool out_bf16)
{
    TORCH_CHECK(x.is_cuda(), "dmel::forward: x must be on the GPU (no CPU fallback)");
    // DMEL_FLAG_X_INDIRECT (dmel_amd.SlotInput): x is a (batch, n_points) VIEW with zero strides over a pointer cell -- its shape is the
    // batch's, its data pointer the cell's; the kernel reads the batch's address from the cell when it runs
    const bool indirect = (flags & DMEL_FLAG_X_INDIRECT) != 0;
    TORCH_CHECK(x.dim() == 2 && x.scalar_type() == at::kFloat && (x.is_contiguous() || indirect), "dmel::forward: x must be a contiguous (batch, n_points) float32 tensor");
    TORCH_CHECK(!indirect || (!lambd_sync && !(flags & DMEL_FLAG_FULL_WINDOW)), "dmel::forward: a pointer-slot input needs lambd on the device and optimized=True");
    TORCH_CHECK(lambd.numel() == 1 && lambd.scalar_type() == at::kFloat, "dmel::forward: lambd must hold one float32");
    dmel_plan* plan = as_plan(plan_h);
    dmel_plan_info info{};
    const c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0);
    dmel_config cfg{};
    check(dmel_plan_get_config(plan, &cfg));
    TORCH_CHECK(x.size(1) == cfg.n_points, "dmel::forward: input has ", x.size(1), " points, the plan was built for n_points=", cfg.n_points);
    const int64_t T = cfg.n_points / cfg.hop_length + 1, M = cfg.n_mels;
    at::Tensor out = at::empty({B, 1, M, T}, x.options().dtype(out_bf16 ? at::kBFloat16 : at::kFloat));      // models.py:36
    const int64_t count = out.numel();
    const int64_t tan_bytes = want_tangent ? (count * 4 + kAlign - 1) / kAlign * kAlign : 0;
    const int64_t scr_bytes = (int64_t)dmel_scratch_bytes(plan, (int32_t)B);
    at::Tensor buf = at::empty({(tan_bytes + scr_bytes + kAlign) / 4}, x.options().dtype(at::kFloat));
    // the allocator hands out 512-byte aligned blocks; keep the arithmetic explicit anyway
    char* base = reinterpret_cast<char*>(buf.data_ptr<float>());
    const int64_t lead = (kAlign - (reinterpret_cast<uintptr_t>(base) % kAlign)) % kAlign;
    TORCH_CHECK(lead == 0, "dmel::forward: unexpected allocation alignment");
    float* tangent = want_tangent ? reinterpret_cast<float*>(base) : nullptr;
    void* scratch = base + tan_bytes;
    const uint32_t f = (uint32_t)flags | (out_bf16 ? DMEL_FLAG_OUT_BF16 : 0u);
    if (B > 0) {
        if (lambd_sync) {
            TORCH_CHECK(lambd.is_cuda() || lambd.is_cpu(), "dmel::forward: lambd on an unsupported device");
            const float lam = lambd.item<float>();                       // the one host read (time_frequency.py:39)
            check(dmel_forward_scratch(plan, x.data_ptr<float>(), (int32_t)B, lam, f, eps, out.data_ptr(), tangent, scratch, stream_of(x)));
        } else if (f & DMEL_FLAG_FULL_WINDOW) {
            // optimized=False: n_fft = 2 n_points whatever lambd is -- the kernels take the window's width from the device value
            TORCH_CHECK(lambd.device() == x.device(), "dmel::forward: lambd is on ", lambd.device(), " but x is on ", x.device());
            check(dmel_forward_dev_fixed(plan, x.data_ptr<float>(), (int32_t)B, lambd.data_ptr<float>(), 0, f, eps, out.data_ptr(), tangent,
                                         scratch, stream_of(x)));
        } else {
            TORCH_CHECK(lambd.device() == x.device(), "dmel::forward: lambd is on ", lambd.device(), " but x is on ", x.device());
            check(dmel_forward_dev(plan, x.data_ptr<float>(), (int32_t)B, lambd.data_ptr<float>(), f, eps, out.data_ptr(), tangent, scratch,
                                   stream_of(x)));
        }
    }
    (void)info;
    return {out, buf};
}

at::Tensor dmel_backward_impl(const at::Tensor& grad_out, const at::Tensor& tangent_buf, int64_t plan_h, bool scalar_shape)
{
    TORCH_CHECK(grad_out.is_cuda() && tangent_buf.is_cuda(), "dmel::backward: tensors must be on the GPU");
    dmel_plan* plan = as_plan(plan_h);
    const c10::DeviceGuard guard(grad_out.device());
    at::Tensor g = grad_out;
    const bool bf16 = g.scalar_type() == at::kBFloat16;                 // the gradient of a bf16 output is read as it is
    if (!bf16 && g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
    if (!g.is_contiguous()) g = g.contiguous();
    const int64_t count = g.numel();
    const int64_t tan_bytes = (count * 4 + kAlign - 1) / kAlign * kAlign;
    TORCH_CHECK(tangent_buf.numel() * 4 >= tan_bytes + (int64_t)dmel_scratch_bytes(plan, 1),
                "dmel::backward: tangent buffer does not belong to a forward of this shape");
    // allocated in lambd's own shape: AccumulateGrad then keeps it as it is (a reshaped view would be copied by one more kernel)
    at::Tensor dl = scalar_shape ? at::empty({}, g.options().dtype(at::kFloat)) : at::empty({1}, g.options().dtype(at::kFloat));
    char* base = reinterpret_cast<char*>(tangent_buf.data_ptr<float>());
    check(dmel_backward_scratch(plan, g.data_ptr(), bf16 ? DMEL_DTYPE_BF16 : DMEL_DTYPE_F32, reinterpret_cast<const float*>(base), count,
                                /*accumulate=*/0, dl.data_ptr<float>(), base + tan_bytes, stream_of(g)));
    return dl;
}

at::Tensor dmel_backward_op(const at::Tensor& grad_out, const at::Tensor& tangent_buf, int64_t plan_h)
{
    return dmel_backward_impl(grad_out, tangent_buf, plan_h, false);
}

at::Tensor dmel_mel_fbanks_op(int64_t n_freqs, double f_min, double f_max, int64_t n_mels, int64_t sample_rate)
{
    at::Tensor fb = at::empty({n_freqs, n_mels}, at::TensorOptions().dtype(at::kFloat));
    check(dmel_mel_fbanks_host((int32_t)n_freqs, f_min, f_max, (int32_t)n_mels, (int32_t)sample_rate, fb.data_ptr<float>()));
    return fb;
}

// A reference on the plan that lives as long as the autograd graph does: the op takes the plan as an integer, and the Python
// object that created it (capi.Plan, held by the layer) may be collected before backward runs -- `y = make_layer()(x);
// y.backward(g)` or `del net; loss.backward()`.  The reference travels in ctx->saved_data as a one-byte CPU tensor whose
// deleter releases it (dmel_plan_retain / dmel_plan_release, include/dmel.h); a plain torch module (models.py:33-56) has no
// such hazard and neither has this one.
at::Tensor plan_reference(dmel_plan* plan)
{
    check(dmel_plan_retain(plan));
    static char anchor = 0;
    return at::from_blob(&anchor, {1}, [plan](void*) { (void)dmel_plan_release(plan); }, at::TensorOptions().dtype(at::kByte));
}

// forward carries d out / d lambd (one trainable scalar: forward mode), backward is one dot product (train.py:47)
struct DmelFn : public torch::autograd::Function<DmelFn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& lambd, int64_t plan_h,
                              int64_t flags, double eps, bool want_tangent, bool lambd_sync, bool out_bf16)
    {
        auto r = dmel_forward_op(x, lambd, plan_h, flags, eps, want_tangent, lambd_sync, out_bf16);
        ctx->saved_data["plan"] = plan_h;
        if (want_tangent) ctx->saved_data["plan_ref"] = plan_reference(as_plan(plan_h));      // backward dereferences the plan
        ctx->saved_data["want"] = want_tangent;
        ctx->saved_data["lam_dim"] = (int64_t)lambd.dim();
        ctx->saved_data["lam_dtype"] = (int64_t)lambd.scalar_type();
        if (want_tangent) ctx->save_for_backward({std::get<1>(r)});
        return std::get<0>(r);
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads)
    {
        at::Tensor dl;
        if (ctx->saved_data["want"].toBool() && grads[0].defined()) {
            const auto saved = ctx->get_saved_variables();
            dl = dmel_backward_impl(grads[0], saved[0], ctx->saved_data["plan"].toInt(), ctx->saved_data["lam_dim"].toInt() == 0);
        }
        return {at::Tensor(), dl, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor dmel_mel_spectrogram_op(const at::Tensor& x, const at::Tensor& lambd, int64_t plan_h, int64_t flags, double eps,
                                   bool lambd_sync, bool out_bf16)
{
    const bool want = at::GradMode::is_enabled() && lambd.requires_grad();
    return DmelFn::apply(x, lambd, plan_h, flags, eps, want, lambd_sync, out_bf16);
}

}  // namespace

TORCH_LIBRARY(dmel, m)
{
    m.def("forward(Tensor x, Tensor lambd, int plan, int flags, float eps, bool want_tangent, bool lambd_sync, bool out_bf16) -> (Tensor, Tensor)");
    m.def("backward(Tensor grad_out, Tensor tangent_buf, int plan) -> Tensor");
    m.def("mel_spectrogram(Tensor x, Tensor lambd, int plan, int flags, float eps, bool lambd_sync, bool out_bf16) -> Tensor");
    m.def("mel_fbanks(int n_freqs, float f_min, float f_max, int n_mels, int sample_rate) -> Tensor");
}

TORCH_LIBRARY_IMPL(dmel, CUDA, m)
{
    m.impl("forward", dmel_forward_op);
    m.impl("backward", dmel_backward_op);
}

TORCH_LIBRARY_IMPL(dmel, CompositeImplicitAutograd, m)
{
    m.impl("mel_spectrogram", dmel_mel_spectrogram_op);
    m.impl("mel_fbanks", dmel_mel_fbanks_op);
}
