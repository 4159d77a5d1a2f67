// The autograd node of GE2ELoss.forward in C++ (libge2e_torch.so): torch.ops.ge2e_amd.loss(e, w, b, eps, eps_cos, variant, impl).
//
// What it replaces: functional._GE2ELossFunction, a Python torch.autograd.Function.  The eager module step -- forward +
// loss.backward() of ONE (N, M, D) batch, the reference's training step s3:19-30 / s4:196-200 -- is host-bound (DESIGN.md
// 5a): the device needs ~33 us for its four operations, the Python path 90-200 us of dispatch around them.  This node does
// the same two C-ABI calls (ge2e_loss_fwd_bwd in forward, ge2e_scale_grads in backward) without leaving C++.
// Plumbing only: no arithmetic here, the kernels live in libge2e_hip.so (include/ge2e_hip.h), which this library links.
#include <torch/library.h>
#include <torch/autograd.h>
#include <ATen/ATen.h>
// ROCm torch keeps DeviceType::CUDA for its devices ("masquerading"): the guard / stream accessors of that flavour
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPGraphsC10Utils.h>

#include <map>
#include <mutex>
#include <utility>

#include "ge2e_hip.h"

namespace {

void check(int code, const char* what) {
    TORCH_CHECK(code == 0, what, " failed: [", code, "] ", ge2e_strerror(code));
}

// Per-(device, stream) workspace, like functional._workspace_for: reused only by launches on the same stream (which the
// stream serialises), bypassed while the stream is being captured (the graph's private pool owns that allocation), at most
// eight entries.  ge2e_workspace_init once per allocation: the first call on it already runs the team kernel.
struct WsEntry { at::Tensor ws; uint64_t used; };
struct WsCache {
    std::mutex mu;
    std::map<std::pair<int, void*>, WsEntry> m;
    uint64_t tick = 0;
};
WsCache& ws_cache() { static WsCache c; return c; }

at::Tensor new_workspace(size_t need, const at::Device& dev, hipStream_t stream) {
    at::Tensor ws = at::empty({(int64_t)(need < 256 ? 256 : need)}, at::TensorOptions().dtype(at::kByte).device(dev));
    check(ge2e_workspace_init(ws.data_ptr(), (size_t)ws.numel(), stream), "ge2e_workspace_init");
    return ws;
}
at::Tensor workspace_for(size_t need, const at::Device& dev, hipStream_t stream) {
    if (c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None) return new_workspace(need, dev, stream);
    WsCache& c = ws_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    const auto key = std::make_pair((int)dev.index(), (void*)stream);
    auto it = c.m.find(key);
    if (it != c.m.end() && (size_t)it->second.ws.numel() >= need) {
        it->second.used = ++c.tick;
        return it->second.ws;
    }
    if (it == c.m.end() && c.m.size() >= 8) {          // least recently used out (not the smallest key: that is the default stream's)
        auto lru = c.m.begin();
        for (auto j = c.m.begin(); j != c.m.end(); ++j)
            if (j->second.used < lru->second.used) lru = j;
        c.m.erase(lru);
    }
    at::Tensor ws = new_workspace(need, dev, stream);
    c.m[key] = WsEntry{ws, ++c.tick};
    return ws;
}
// the workspace this node would use for `like`'s device on the current stream (an empty tensor when there is none yet):
// diagnostics -- functional.workspace_fallback_count reads TeamCtl.fallbacks from it
at::Tensor cached_workspace(const at::Tensor& like) {
    TORCH_CHECK(like.is_cuda(), "cached_workspace: a ROCm device tensor names the device");
    hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(like.device().index()).stream();
    WsCache& c = ws_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.m.find(std::make_pair((int)like.device().index(), (void*)stream));
    return it == c.m.end() ? at::empty({0}, at::TensorOptions().dtype(at::kByte).device(like.device())) : it->second.ws;
}

// one ge2e_loss_fwd_bwd launch on the current stream; need: also dE, dw, db (sc = loss | dw | db, [3][B], else [1][B])
struct Launched { at::Tensor sc, dE; bool squeeze; };
Launched launch_forward(const at::Tensor& e, const at::Tensor& w, const at::Tensor& b, double eps, double eps_cos, int64_t variant,
                        int64_t impl, bool need) {
        TORCH_CHECK(e.is_cuda(), "embeddings are on ", e.device(), ": the GE2E HIP path needs a ROCm device tensor (no CPU fallback exists)");
        TORCH_CHECK(e.dim() == 3 || e.dim() == 4, "embeddings must be (N,M,D) or (B,N,M,D)");
        TORCH_CHECK(e.is_contiguous(), "embeddings must be contiguous (the reference uses .view(), s3:49-52)");
        TORCH_CHECK(e.scalar_type() == at::kFloat, "embeddings must be float32 at this boundary");
        for (const at::Tensor* t : {&w, &b}) {
            TORCH_CHECK(t->is_cuda() && t->device() == e.device(), "w / b must be on the embeddings' device");
            TORCH_CHECK(t->scalar_type() == at::kFloat && t->numel() == 1, "w / b must be float32 scalar tensors");
        }
        const bool squeeze = e.dim() == 3;
        const at::Tensor e4 = squeeze ? e.unsqueeze(0) : e;
        const int B = (int)e4.size(0), N = (int)e4.size(1), M = (int)e4.size(2), D = (int)e4.size(3);
        c10::hip::HIPGuardMasqueradingAsCUDA guard(e.device());
        hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(e.device().index()).stream();
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(e.device());
        at::Tensor sc = at::empty({need ? 3 : 1, B}, f32);          // loss | dw | db
        at::Tensor dE = need ? at::empty_like(e4) : at::Tensor();
        const size_t wsb = ge2e_workspace_bytes(B, N, M, D, (int)variant, (int)impl);
        at::Tensor ws = workspace_for(wsb, e.device(), stream);
        float* scp = sc.data_ptr<float>();
        check(ge2e_loss_fwd_bwd(e4.data_ptr<float>(), B, N, M, D, w.data_ptr<float>(), b.data_ptr<float>(), (float)eps_cos,
                                (float)eps, (int)variant, (int)impl, scp, nullptr, need ? dE.data_ptr<float>() : nullptr,
                                need ? scp + B : nullptr, need ? scp + 2 * B : nullptr, ws.data_ptr(), (size_t)ws.numel(),
                                stream),
              "ge2e_loss_fwd_bwd");
        return Launched{sc, dE, squeeze};
}

struct GE2ELossNode : public torch::autograd::Function<GE2ELossNode> {
    // (reached through the Autograd dispatch key only: at least one input wants a gradient)
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& e, const at::Tensor& w, const at::Tensor& b,
                              double eps, double eps_cos, int64_t variant, int64_t impl) {
        const Launched r = launch_forward(e, w, b, eps, eps_cos, variant, impl, true);
        ctx->saved_data["squeeze"] = r.squeeze;
        ctx->saved_data["w_sizes"] = w.sizes().vec();       // one element each, but any shape: the gradient comes back in it
        ctx->saved_data["b_sizes"] = b.sizes().vec();
        ctx->save_for_backward({r.dE, r.sc});
        at::Tensor loss = r.sc[0];
        return r.squeeze ? loss[0] : loss;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor& dE = saved[0];
        const at::Tensor& sc = saved[1];
        at::Tensor g = grads[0];
        if (g.scalar_type() != at::kFloat || !g.is_contiguous()) g = g.to(at::kFloat).contiguous();
        const int B = (int)dE.size(0), N = (int)dE.size(1), M = (int)dE.size(2), D = (int)dE.size(3);
        const bool need_e = ctx->needs_input_grad(0), need_w = ctx->needs_input_grad(1), need_b = ctx->needs_input_grad(2);
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dE.device());
        hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dE.device().index()).stream();
        at::Tensor gE = need_e ? at::empty_like(dE) : at::Tensor();
        at::Tensor gwb = (need_w || need_b) ? at::empty({2}, dE.options()) : at::Tensor();
        const float* scp = sc.data_ptr<float>();
        // one launch: gE = g dE, gw = sum g dw, gb = sum g db (out of place: a retained graph may run again)
        check(ge2e_scale_grads(dE.data_ptr<float>(), scp + B, scp + 2 * B, g.data_ptr<float>(), (int)g.numel(), B, N, M, D,
                               need_e ? gE.data_ptr<float>() : nullptr, need_w ? gwb.data_ptr<float>() : nullptr,
                               need_b ? gwb.data_ptr<float>() + 1 : nullptr, stream),
              "ge2e_scale_grads");
        if (need_e && ctx->saved_data["squeeze"].toBool()) gE = gE[0];
        at::Tensor gw, gb;
        if (need_w) gw = gwb[0].reshape(ctx->saved_data["w_sizes"].toIntVector());
        if (need_b) gb = gwb[1].reshape(ctx->saved_data["b_sizes"].toIntVector());
        return {gE, gw, gb, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor ge2e_loss_autograd(const at::Tensor& e, const at::Tensor& w, const at::Tensor& b, double eps, double eps_cos,
                              int64_t variant, int64_t impl) {
    if (!(at::GradMode::is_enabled() && (e.requires_grad() || w.requires_grad() || b.requires_grad()))) {
        at::AutoDispatchBelowADInplaceOrView below;      // nothing to differentiate: the plain kernel, forward only
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("ge2e_amd::loss", "")
                             .typed<at::Tensor(const at::Tensor&, const at::Tensor&, const at::Tensor&, double, double, int64_t, int64_t)>();
        return op.call(e, w, b, eps, eps_cos, variant, impl);
    }
    return GE2ELossNode::apply(e, w, b, eps, eps_cos, variant, impl);
}
at::Tensor ge2e_loss_forward_only(const at::Tensor& e, const at::Tensor& w, const at::Tensor& b, double eps, double eps_cos,
                                  int64_t variant, int64_t impl) {
    const Launched r = launch_forward(e, w, b, eps, eps_cos, variant, impl, false);
    at::Tensor loss = r.sc[0];
    return r.squeeze ? loss[0] : loss;
}

}  // namespace

TORCH_LIBRARY(ge2e_amd, m) {
    m.def("loss(Tensor e, Tensor w, Tensor b, float eps, float eps_cos, int variant, int impl) -> Tensor");
    m.def("cached_workspace(Tensor like) -> Tensor", cached_workspace);
}
TORCH_LIBRARY_IMPL(ge2e_amd, Autograd, m) { m.impl("loss", ge2e_loss_autograd); }
TORCH_LIBRARY_IMPL(ge2e_amd, CUDA, m) { m.impl("loss", ge2e_loss_forward_only); }
TORCH_LIBRARY_IMPL(ge2e_amd, CPU, m) { m.impl("loss", ge2e_loss_forward_only); }     // raises: no CPU path
