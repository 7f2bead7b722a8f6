// Host side of QuantizeLinear's hot path as a C++ autograd node (PyTorch extension `_fq_node`, optional accelerator of utils_quant.py).
//
// What it replaces: utils_quant._PairNode (a Python torch.autograd.Function) for the straight-line case -- a contiguous weight and
// input of one dtype, clip [-2, 2] (models/utils_quant.py:198,245), mask backward.  Why: a Python Function's backward runs on the
// autograd engine's device thread, where it has to take the GIL and runs cache-cold -- 25-40 us per node on the hosts of this pool
// (profiles/r05_host_overhead_ab.txt), more than the 32 us the backward KERNEL takes on the metric tensors.  This
// node's backward is ~3 us of C++: the in-place guard on reference counts, two allocations at most, one call through the C ABI.
//
// What it does not do: decide anything.  Which arithmetic, whether operands pair, what is remembered for sibling projections stays
// in utils_quant.py; every case outside the straight line (a gradient of another dtype, a missing gradient, a backward that is
// itself recorded -- create_graph=True --, a launch the entry point declines) is handed back to utils_quant._PairNode.backward under
// the GIL, so those semantics have ONE implementation.  The kernels are reached through the function pointers of the C ABI
// (include/llmqat_fakequant.h) that utils_quant hands over at import: this file links against PyTorch, not against the HIP library.
#include <torch/extension.h>
#include <torch/csrc/autograd/anomaly_mode.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <atomic>
#include <deque>
#include <mutex>

#include "../../include/llmqat_fakequant.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

decltype(&fq_sym_fwd_pair) g_fwd_pair = nullptr;
decltype(&fq_ste_bwd_mask_pair) g_bwd_pair = nullptr;
decltype(&fq_sym_fwd_multi) g_fwd_multi = nullptr;
decltype(&fq_ste_bwd_mask) g_bwd_one = nullptr;
decltype(&fq_ste_bwd_mask_wide) g_bwd_wide = nullptr;
decltype(&fq_last_error) g_last_error = nullptr;
PyObject* g_slow_backward = nullptr;   // utils_quant._pair_backward_from_cpp (leaked on purpose: never released without the GIL)
PyObject* g_slow_backward_one = nullptr;   // utils_quant._one_backward_from_cpp
std::atomic<bool> g_inplace{true};     // utils_quant._INPLACE_WGRAD

// ---- counters (merged into llm_qat_amd.stats() and into the tests' launch counts) ----------------------------------------------------
enum Counter { C_FWD_LAUNCH, C_FWD_WEIGHT, C_BWD_LAUNCH, C_BWD_ONE, C_BWD_ONE_WIDE, C_SLOW_BACKWARD, C_INPLACE_TAKEN, C_REFUSE_UNCALIBRATED, C_REFUSE_CXX_REFS, C_REFUSE_STORAGE,
               C_REFUSE_ANOMALY, C_REFUSE_STORAGE_REFS, C_REFUSE_BASE_REFS, C_COUNT };
const char* const kCounterNames[C_COUNT] = {"cpp_pair_forward", "cpp_weight_forward", "cpp_pair_backward", "cpp_one_backward", "cpp_one_backward_wide", "cpp_slow_backward", "inplace_taken",
                                            "inplace_refused:uncalibrated", "inplace_refused:cxx_refs", "inplace_refused:storage",
                                            "inplace_refused:anomaly", "inplace_refused:storage_refs", "inplace_refused:base_refs"};
std::atomic<int64_t> g_counters[C_COUNT];
inline void count(Counter c) { g_counters[c].fetch_add(1, std::memory_order_relaxed); }

// ---- backward epochs -----------------------------------------------------------------------------------------------------------------
// utils_quant remembers things per forward thread (a fake-quantized activation for the sibling projections, a pending V of the K/V hooks)
// and must let go of them when a backward over that thread's graph starts.  The Python nodes do that themselves; these nodes cannot touch
// Python state without the GIL, so a backward bumps a counter cell that utils_quant compares on the thread's next call (`_state()`): what
// was remembered under an older epoch goes then.  One exception is worth the GIL: a PENDING V RESULT NOBODY ASKED FOR (a wrong guess of
// the K/V speculation, once per call signature) pins a tensor and its side buffer -- the thread raises the cell's second word while one is
// pending, and the first backward that sees it calls utils_quant._forget_from_cpp under the GIL.  One cell (two words) per forward thread
// state, recycled: a bump that reaches a recycled cell only invalidates, which is safe.
struct Cell { std::atomic<int64_t> epoch{0}, pending{0}; };
struct Cells {
    std::mutex m;
    std::deque<Cell> cells;   // (a deque never moves its elements)
    std::vector<Cell*> free;
} g_cells;
PyObject* g_forget = nullptr;   // utils_quant._forget_from_cpp(cell handle)

int64_t epoch_new() {
    std::lock_guard<std::mutex> lock(g_cells.m);
    Cell* c;
    if (!g_cells.free.empty()) {
        c = g_cells.free.back();
        g_cells.free.pop_back();
        c->pending.store(0);
    } else {
        g_cells.cells.emplace_back();
        c = &g_cells.cells.back();
    }
    return reinterpret_cast<int64_t>(c);
}

void epoch_free(int64_t h) {
    std::lock_guard<std::mutex> lock(g_cells.m);
    g_cells.free.push_back(reinterpret_cast<Cell*>(h));
}

inline void backward_began(int64_t h) {
    Cell* c = reinterpret_cast<Cell*>(h);
    c->epoch.fetch_add(1, std::memory_order_release);
    if (c->pending.load(std::memory_order_relaxed) != 0 && c->pending.exchange(0) != 0 && g_forget != nullptr) {
        pybind11::gil_scoped_acquire gil;
        pybind11::reinterpret_borrow<pybind11::object>(g_forget)(h);
    }
}

// ---- the in-place guard (utils_quant._inplace_ok, on C++ reference counts) -------------------------------------------------------------
// May the weight's gradient be masked where it stands?  Only if this node provably holds the only handle on it: it IS its storage
// (contiguous, offset 0, nothing before or after), no other TensorImpl / Python object / storage alias refers to it -- reference
// counts compared with what a gradient nobody else holds showed in the SAME code path at import (calibrate(): F.linear's wgrad, which
// arrives as a view of a temporary, and a plain fresh tensor) -- and anomaly mode is off.  Anything else takes the copying launch.
struct Baseline { bool set = false; int64_t use = 0, storage = 0, base_use = 0; };
Baseline g_base[2];   // [is_view]
bool g_probe_armed = false;

struct Counts { bool view; int64_t use, storage, base_use; };

inline Counts counts_of(const at::Tensor& g) {
    Counts c{g.is_view(), (int64_t)g.use_count(), (int64_t)g.storage().use_count(), 0};
    if (c.view) c.base_use = (int64_t)g._base().use_count();
    return c;
}

bool inplace_ok(const at::Tensor& g) {
    const Counts c = counts_of(g);
    const Baseline& b = g_base[c.view ? 1 : 0];
    if (!b.set) { count(C_REFUSE_UNCALIBRATED); return false; }
    if (c.use > b.use) { count(C_REFUSE_CXX_REFS); return false; }
    if (!(g.is_contiguous() && g.storage_offset() == 0 && (int64_t)g.storage().nbytes() == g.numel() * (int64_t)g.element_size())) {
        count(C_REFUSE_STORAGE);
        return false;
    }
    if (torch::autograd::AnomalyMode::is_enabled()) { count(C_REFUSE_ANOMALY); return false; }
    if (c.view && c.base_use > b.base_use) { count(C_REFUSE_BASE_REFS); return false; }
    if (c.storage > b.storage) { count(C_REFUSE_STORAGE_REFS); return false; }
    count(C_INPLACE_TAKEN);
    return true;
}

// ---- the node ------------------------------------------------------------------------------------------------------------------------
struct PairArgs {   // not a tensor type: Function::apply takes only `weight` and `input` for the node's inputs
    at::Tensor wq, xq, side_w, side_x;
    bool view_x, need_w, need_x;
    int64_t rows_w, rows_x, cols, code, epoch;
};

inline int dtype_code(c10::ScalarType t) {
    return t == at::kFloat ? FQ_DTYPE_F32 : t == at::kBFloat16 ? FQ_DTYPE_BF16 : t == at::kHalf ? FQ_DTYPE_F16 : -1;
}

[[noreturn]] void launch_failed(const char* what, int rc) {
    TORCH_CHECK(false, "llm_qat_amd: ", what, " failed (rc ", rc, "): ", g_last_error ? g_last_error() : "");
}

struct FqPairNode : public torch::autograd::Function<FqPairNode> {
    static variable_list forward(AutogradContext* ctx, const at::Tensor& weight, const at::Tensor& input, const PairArgs* a) {
        ctx->set_materialize_grads(false);             // a gradient that never arrived is answered with None (utils_quant: _no_gradient)
        ctx->save_for_backward({a->side_w, a->side_x});   // saved tensors (either may be undefined): visible to saved-tensor hooks
        ctx->saved_data["m"] = std::vector<int64_t>{a->rows_w, a->rows_x, a->cols, a->code, (int64_t)weight.scalar_type(), a->epoch,
                                                    (int64_t)a->need_w, (int64_t)a->need_x};
        at::Tensor wq = a->wq;
        at::Tensor xq = a->view_x ? a->xq.view_as(a->xq) : a->xq;   // a sibling's data: this node gets a tensor of its own over it
        if (!(a->need_w && a->need_x)) ctx->mark_non_differentiable({a->need_w ? xq : wq});
        return {wq, xq};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto m = ctx->saved_data["m"].toIntVector();
        const int64_t rows_w = m[0], rows_x = m[1], cols = m[2], code = m[3];
        const auto dtype = (c10::ScalarType)m[4];
        if (code < 0) {   // calibrate(): what the reference counts of a gradient nobody else holds look like here
            if (g_probe_armed && grads[0].defined()) {
                const Counts c = counts_of(grads[0]);
                g_base[c.view ? 1 : 0] = Baseline{true, c.use, c.storage, c.base_use};
            }
            return {grads[0], grads[1], at::Tensor()};
        }
        backward_began(m[5]);
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &gw = grads[0], &gx = grads[1], &side_w = saved[0], &side_x = saved[1];
        if (gw.defined() && gx.defined() && side_w.defined() && side_x.defined() && gw.is_cuda() && gx.is_cuda() && gw.scalar_type() == dtype &&
            gx.scalar_type() == dtype &&
            !at::GradMode::is_enabled() && gw.is_contiguous() && gx.is_contiguous() &&
            ((reinterpret_cast<uintptr_t>(gw.data_ptr()) | reinterpret_cast<uintptr_t>(gx.data_ptr())) & 15) == 0) {
            const bool inplace = g_inplace.load(std::memory_order_relaxed) && inplace_ok(gw);
            at::Tensor ow = inplace ? gw : at::empty_like(gw);
            at::Tensor ox = at::empty_like(gx);
            auto* pw = static_cast<uint8_t*>(side_w.data_ptr());
            auto* px = static_cast<uint8_t*>(side_x.data_ptr());
            c10::DeviceGuard guard(gw.device());
            void* stream = c10::hip::getCurrentHIPStream(gw.device().index()).stream();
            const int rc = g_bwd_pair(gw.data_ptr(), ow.data_ptr(), rows_w, reinterpret_cast<const float*>(pw), pw + rows_w * 8, gx.data_ptr(),
                                      ox.data_ptr(), rows_x, reinterpret_cast<const float*>(px), px + rows_x * 8, cols, -2.0f, 2.0f, (int)code, stream);
            if (rc == FQ_OK) {
                count(C_BWD_LAUNCH);
                return {ow, ox, at::Tensor()};   // (one entry per argument of forward: the third is not a tensor)
            }
            if (rc != FQ_ERR_UNSUPPORTED) launch_failed("fq_ste_bwd_mask_pair", rc);
        }
        // everything else: utils_quant._PairNode.backward decides, as it does for the Python node (never in place from here)
        count(C_SLOW_BACKWARD);
        pybind11::gil_scoped_acquire gil;
        pybind11::object out = pybind11::reinterpret_borrow<pybind11::object>(g_slow_backward)(
            gw.defined() ? pybind11::cast(gw) : pybind11::none(), gx.defined() ? pybind11::cast(gx) : pybind11::none(),
            side_w.defined() ? pybind11::cast(side_w) : pybind11::none(), side_x.defined() ? pybind11::cast(side_x) : pybind11::none(), rows_w, rows_x, cols,
            code, m[6] != 0, m[7] != 0);
        auto t = out.cast<pybind11::tuple>();
        variable_list res(3);
        for (int i = 0; i < 2; i++)
            if (!t[i].is_none()) res[i] = t[i].cast<at::Tensor>();
        return res;
    }
};

// ---- the one-tensor node ---------------------------------------------------------------------------------------------------------------
// utils_quant._PrecomputedAct in C++: a node over ONE tensor that a launch has already fake-quantized (K and V of the KV-cache hooks,
// models/modeling_llama_quant.py:320-327, which share a forward launch but never a node).  Forward launches nothing; backward is that
// tensor's own STE launch -- fq_ste_bwd_mask, or fq_ste_bwd_mask_wide behind the reference's fp32 result under autocast (fp32 gradient in,
// input-dtype gradient out).  Never in place: these are gradients of tensors a caller can see.
struct OneArgs {
    at::Tensor y, side;
    int64_t rows, cols, epoch;
    double lo, hi;
};

struct FqOneNode : public torch::autograd::Function<FqOneNode> {
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const OneArgs* a) {
        ctx->set_materialize_grads(false);
        ctx->save_for_backward({a->side});
        ctx->saved_data["m"] = std::vector<int64_t>{a->rows, a->cols, (int64_t)x.scalar_type(), a->epoch, (int64_t)(a->y.scalar_type() != x.scalar_type())};
        ctx->saved_data["c"] = std::vector<double>{a->lo, a->hi};
        return a->y;   // the launch's own fresh result (not an input of this node): it becomes the node's output as it is, as in the reference
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const at::Tensor& g = grads[0];
        if (!g.defined()) return {at::Tensor(), at::Tensor()};
        const auto m = ctx->saved_data["m"].toIntVector();
        const auto c = ctx->saved_data["c"].toDoubleVector();
        const int64_t rows = m[0], cols = m[1];
        const auto dtype = (c10::ScalarType)m[2];
        const bool wide = m[4] != 0;
        backward_began(m[3]);
        const auto saved = ctx->get_saved_variables();
        const at::Tensor& side = saved[0];
        const int code = dtype_code(dtype);
        if (side.defined() && g.is_cuda() && code >= 0 && g.scalar_type() == (wide ? at::kFloat : dtype) && !at::GradMode::is_enabled() && g.is_contiguous() &&
            g.numel() == rows * cols && (reinterpret_cast<uintptr_t>(g.data_ptr()) & 15) == 0) {
            at::Tensor gx = at::empty(g.sizes(), g.options().dtype(dtype));
            auto* sp = static_cast<uint8_t*>(side.data_ptr());
            c10::DeviceGuard guard(g.device());
            void* stream = c10::hip::getCurrentHIPStream(g.device().index()).stream();
            const int rc = wide ? g_bwd_wide(g.data_ptr(), gx.data_ptr(), rows, reinterpret_cast<const float*>(sp), sp + rows * 8, nullptr, nullptr, 0, nullptr,
                                             nullptr, cols, (float)c[0], (float)c[1], code, stream)
                                : g_bwd_one(g.data_ptr(), gx.data_ptr(), rows, cols, (float)c[0], (float)c[1], reinterpret_cast<const float*>(sp), sp + rows * 8,
                                            (size_t)(side.numel() - rows * 8), code, stream);
            if (rc == FQ_OK) {
                count(wide ? C_BWD_ONE_WIDE : C_BWD_ONE);
                return {gx, at::Tensor()};
            }
            if (rc != FQ_ERR_UNSUPPORTED) launch_failed(wide ? "fq_ste_bwd_mask_wide" : "fq_ste_bwd_mask", rc);
        }
        count(C_SLOW_BACKWARD);
        pybind11::gil_scoped_acquire gil;
        pybind11::object out = pybind11::reinterpret_borrow<pybind11::object>(g_slow_backward_one)(
            pybind11::cast(g), side.defined() ? pybind11::cast(side) : pybind11::none(), rows, cols, c[0], c[1], (int64_t)code, wide);
        variable_list res(2);
        if (!out.is_none()) res[0] = out.cast<at::Tensor>();
        return res;
    }
};

at::Tensor one_node(const at::Tensor& x, const at::Tensor& y, const at::Tensor& side, int64_t rows, int64_t cols, double lo, double hi, int64_t epoch) {
    TORCH_CHECK(epoch != 0, "llm_qat_amd: one_node needs its forward thread's epoch cell");
    OneArgs a{y, side, rows, cols, epoch, lo, hi};
    return FqOneNode::apply(x, &a);
}

// ---- what utils_quant calls ----------------------------------------------------------------------------------------------------------
void bind(int64_t fwd_pair, int64_t bwd_pair, int64_t fwd_multi, int64_t bwd_one, int64_t bwd_wide, int64_t last_error, pybind11::object slow_backward,
          pybind11::object slow_backward_one, pybind11::object forget) {
    g_forget = forget.release().ptr();
    g_bwd_one = reinterpret_cast<decltype(g_bwd_one)>(bwd_one);
    g_bwd_wide = reinterpret_cast<decltype(g_bwd_wide)>(bwd_wide);
    g_slow_backward_one = slow_backward_one.release().ptr();
    g_fwd_pair = reinterpret_cast<decltype(g_fwd_pair)>(fwd_pair);
    g_fwd_multi = reinterpret_cast<decltype(g_fwd_multi)>(fwd_multi);
    g_bwd_pair = reinterpret_cast<decltype(g_bwd_pair)>(bwd_pair);
    g_last_error = reinterpret_cast<decltype(g_last_error)>(last_error);
    g_slow_backward = slow_backward.release().ptr();
}

// One node over results that exist already (the weight's own launch + an activation a sibling projection has fake-quantized).
std::pair<at::Tensor, at::Tensor> pair_node(const at::Tensor& weight, const at::Tensor& input, const at::Tensor& wq, const at::Tensor& xq,
                                            const c10::optional<at::Tensor>& side_w, const c10::optional<at::Tensor>& side_x, int64_t rows_w,
                                            int64_t rows_x, int64_t cols, int64_t code, bool view_x, int64_t epoch) {
    const bool grad = at::GradMode::is_enabled();
    TORCH_CHECK(epoch != 0, "llm_qat_amd: pair_node needs its forward thread's epoch cell");
    if (!(grad && (weight.requires_grad() || input.requires_grad()))) return {wq, xq};
    PairArgs a{wq, xq, side_w.value_or(at::Tensor()), side_x.value_or(at::Tensor()), view_x, grad && weight.requires_grad(),
               grad && input.requires_grad(), rows_w, rows_x, cols, code, epoch};
    auto out = FqPairNode::apply(weight, input, &a);
    return {out[0], out[1]};
}

// Both operands of a QuantizeLinear in one launch + their node: ops.pair_forward_planned and _PairNode.apply in one call.
// -> (wq, xq, side_x or None), or None where the entry point declines the shape (the caller then takes the general path).
pybind11::object pair_forward(const at::Tensor& weight, const at::Tensor& input, int64_t code, int64_t cols, int64_t rows_w, int64_t rows_x, int64_t mw,
                              int64_t mx, int64_t w_bits, int64_t a_bits, bool need_w, bool need_x, bool ac, int64_t sem, int64_t epoch) {
    TORCH_CHECK(g_fwd_pair != nullptr, "llm_qat_amd: _fq_node is not bound to the kernel library");
    TORCH_CHECK(epoch != 0, "llm_qat_amd: pair_forward needs its forward thread's epoch cell");
    TORCH_CHECK(weight.is_contiguous() && input.is_contiguous() && weight.scalar_type() == input.scalar_type() && dtype_code(weight.scalar_type()) == code &&
                    weight.numel() == rows_w * cols && input.numel() == rows_x * cols && weight.device() == input.device(),
                "llm_qat_amd: pair_forward called with operands its plan does not describe");
    at::Tensor wq = at::empty_like(weight), xq = at::empty_like(input), side_w, side_x;
    uint8_t *pw = nullptr, *px = nullptr;
    const auto bytes = weight.options().dtype(at::kByte);
    if (need_w) {
        side_w = at::empty({rows_w * 8 + mw}, bytes);
        pw = static_cast<uint8_t*>(side_w.data_ptr());
    }
    if (need_x) {
        side_x = at::empty({rows_x * 8 + mx}, bytes);
        px = static_cast<uint8_t*>(side_x.data_ptr());
    }
    int rc;
    {
        c10::DeviceGuard guard(weight.device());
        void* stream = c10::hip::getCurrentHIPStream(weight.device().index()).stream();
        rc = g_fwd_pair(weight.data_ptr(), wq.data_ptr(), rows_w, (int)w_bits, reinterpret_cast<float*>(pw), pw ? pw + rows_w * 8 : nullptr, need_w ? (size_t)mw : 0,
                        input.data_ptr(), xq.data_ptr(), rows_x, (int)a_bits, reinterpret_cast<float*>(px), px ? px + rows_x * 8 : nullptr, need_x ? (size_t)mx : 0,
                        cols, (int)code, (int)sem, ac ? 1 : 0, -2.0f, 2.0f, stream);
    }
    if (rc == FQ_ERR_UNSUPPORTED) return pybind11::none();
    if (rc != FQ_OK) launch_failed("fq_sym_fwd_pair", rc);
    count(C_FWD_LAUNCH);
    if (need_w || need_x) {
        PairArgs a{wq, xq, side_w, side_x, false, need_w, need_x, rows_w, rows_x, cols, code, epoch};
        auto out = FqPairNode::apply(weight, input, &a);
        wq = out[0];
        xq = out[1];
    }
    return pybind11::make_tuple(wq, xq, side_x.defined() ? pybind11::cast(side_x) : pybind11::none());
}

// A sibling projection whose input another projection has already fake-quantized (q/k/v, gate/up): the weight's own launch (the one-tensor
// form of the same entry point family) + the node over it and the remembered activation.  -> (wq, xq') or None where the entry point
// declines the shape (the caller then takes ops.weight_forward's general path).
pybind11::object weight_forward_node(const at::Tensor& weight, const at::Tensor& input, const at::Tensor& xq, const c10::optional<at::Tensor>& side_x,
                                     int64_t code, int64_t cols, int64_t rows_w, int64_t rows_x, int64_t mw, int64_t w_bits, bool need_w, bool need_x, bool ac,
                                     int64_t sem, int64_t epoch) {
    TORCH_CHECK(g_fwd_multi != nullptr, "llm_qat_amd: _fq_node is not bound to the kernel library");
    TORCH_CHECK(epoch != 0, "llm_qat_amd: weight_forward_node needs its forward thread's epoch cell");
    TORCH_CHECK(weight.is_contiguous() && dtype_code(weight.scalar_type()) == code && weight.numel() == rows_w * cols && xq.numel() == rows_x * cols &&
                    xq.scalar_type() == weight.scalar_type() && weight.device() == xq.device() && (!need_x || side_x.has_value()),
                "llm_qat_amd: weight_forward_node called with operands its plan does not describe");
    at::Tensor wq = at::empty_like(weight), side_w;
    uint8_t* pw = nullptr;
    if (need_w) {
        side_w = at::empty({rows_w * 8 + mw}, weight.options().dtype(at::kByte));
        pw = static_cast<uint8_t*>(side_w.data_ptr());
    }
    int rc;
    {
        c10::DeviceGuard guard(weight.device());
        void* stream = c10::hip::getCurrentHIPStream(weight.device().index()).stream();
        const fq_fwd_tensor t{weight.data_ptr(), wq.data_ptr(), rows_w, (int)w_bits, reinterpret_cast<float*>(pw), pw ? pw + rows_w * 8 : nullptr,
                              need_w ? (size_t)mw : 0};
        rc = g_fwd_multi(1, &t, cols, (int)code, (int)sem, ac ? 1 : 0, -2.0f, 2.0f, stream);
    }
    if (rc == FQ_ERR_UNSUPPORTED) return pybind11::none();
    if (rc != FQ_OK) launch_failed("fq_sym_fwd_multi", rc);
    count(C_FWD_WEIGHT);
    if (!(need_w || need_x)) return pybind11::make_tuple(wq, xq);
    PairArgs a{wq, xq, side_w, need_x ? *side_x : at::Tensor(), true, need_w, need_x, rows_w, rows_x, cols, code, epoch};
    auto out = FqPairNode::apply(weight, input, &a);
    return pybind11::make_tuple(out[0], out[1]);
}

// Reference counts of a gradient nobody else holds, learned in this node's own backward (see inplace_ok): utils_quant runs, once at
// import and on tiny CPU tensors, F.linear over a probe node (-> a view of a temporary) and an elementwise product (-> a plain tensor).
std::pair<at::Tensor, at::Tensor> probe_node(const at::Tensor& weight, const at::Tensor& input) {
    PairArgs a{weight * 1.0, input * 1.0, at::Tensor(), at::Tensor(), false, true, true, 0, 0, 0, -1, 0};
    auto out = FqPairNode::apply(weight, input, &a);
    return {out[0], out[1]};
}

void arm_probe(bool on) { g_probe_armed = on; }

// Measurement stand-in for F.linear (bench.py api_path, tools/api_path_probe.py): a "GEMM" that launches nothing -- forward returns an
// uninitialised output, backward fresh uninitialised gradients nobody else holds.  In C++ so that the gradients reach the fake-quant node
// the way F.linear's own do: a tensor a Python function returns keeps its Python wrapper, which the in-place guard counts as a holder.
struct NoGemm : public torch::autograd::Function<NoGemm> {
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const at::Tensor& w) {
        ctx->save_for_backward({x, w});
        auto shape = x.sizes().vec();
        shape.back() = w.size(0);
        return at::empty(shape, x.options());
    }
    static variable_list backward(AutogradContext* ctx, variable_list) {
        const auto saved = ctx->get_saved_variables();
        return {at::empty_like(saved[0]), at::empty_like(saved[1])};
    }
};

at::Tensor no_gemm_linear(const at::Tensor& x, const at::Tensor& w) { return NoGemm::apply(x, w); }

pybind11::dict baselines() {
    pybind11::dict d;
    for (int v = 0; v < 2; v++)
        if (g_base[v].set) d[v ? "view" : "plain"] = pybind11::make_tuple(g_base[v].use, g_base[v].storage, g_base[v].base_use);
    return d;
}

pybind11::dict counters(bool reset) {
    pybind11::dict d;
    for (int i = 0; i < C_COUNT; i++) {
        const int64_t v = reset ? g_counters[i].exchange(0) : g_counters[i].load();
        if (v) d[kCounterNames[i]] = v;
    }
    return d;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "llm_qat_amd: QuantizeLinear's operand pair as a C++ autograd node over the C ABI of libllmqat_fakequant.so";
    m.def("bind", &bind);
    m.def("pair_forward", &pair_forward);
    m.def("pair_node", &pair_node);
    m.def("one_node", &one_node);
    m.def("weight_forward_node", &weight_forward_node);
    m.def("probe_node", &probe_node);
    m.def("arm_probe", &arm_probe);
    m.def("no_gemm_linear", &no_gemm_linear);
    m.def("baselines", &baselines);
    m.def("counters", &counters, pybind11::arg("reset") = false);
    m.def("set_inplace", [](bool on) { g_inplace.store(on); });
    m.def("current_stream", [](int64_t device) { return reinterpret_cast<int64_t>(c10::hip::getCurrentHIPStream((c10::DeviceIndex)device).stream()); },
          "the raw handle of the stream the node would launch on (tests compare it with torch's)");
    m.def("epoch_new", &epoch_new);
    m.def("epoch_free", &epoch_free);
    m.attr("abi_version") = FQ_ABI_VERSION;
}
