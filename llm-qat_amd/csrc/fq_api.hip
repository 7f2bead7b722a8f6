// fq_api.hip -- host side of the C ABI declared in include/llmqat_fakequant.h:
// argument validation, kernel selection, launches.  No allocation, no synchronisation.
#include "../../include/llmqat_fakequant.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "fq_launch.h"

using namespace fq;

namespace {
thread_local char g_err[320] = "";
}

namespace fq {
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int ok() {
    g_err[0] = 0;
    return FQ_OK;
}
hipError_t& launch_status() {
    static thread_local hipError_t st = hipSuccess;
    return st;
}
}  // namespace fq

namespace {

struct MaskArgs {
    void* mask = nullptr;
    size_t bytes = 0;
    float lo = 0.f, hi = 0.f;
};

// fq_rows_view (elements) -> RowPitch (bytes).  A view that describes contiguous rows stays off.  false: not representable.
bool set_pitch(RowPitch& p, const fq_rows_view* v, int64_t rows, int64_t cols, int esize) {
    p = RowPitch{};
    if (!v || v->n_inner <= 0) return true;
    if (rows > 0x7FFFFFFF || v->n_inner > 0x7FFFFFFF || v->stride_outer < 0 || v->stride_inner < 0) return false;
    if (v->stride_inner == cols && (v->n_inner >= rows || v->stride_outer == v->n_inner * cols)) return true;   // contiguous after all
    p.outer = v->stride_outer * esize;
    p.inner = v->stride_inner * esize;
    p.n_inner = (uint32_t)v->n_inner;
    p.on = 1;
    return true;
}

template <bool ASYM>
int rowwise(const void* x, void* y, int32_t* idx, float* scale, float* bounds, int64_t rows, int64_t cols, int bits, int dtype,
            int sem, void* ws, size_t wsb, void* stream, const MaskArgs* mk = nullptr, const fq_rows_view* xv = nullptr,
            const fq_rows_view* yv = nullptr) {
    if (dtype < 0 || dtype > FQ_DTYPE_F64) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (bits < 1 || bits > 31) return fail(FQ_ERR_BITS, "num_bits=%d outside [1, 31]", bits);
    if (sem != FQ_SEM_CPU_EAGER && sem != FQ_SEM_DEVICE_EAGER) return fail(FQ_ERR_ARG, "unknown semantics code %d", sem);
    if (rows < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape rows=%lld cols=%lld", (long long)rows, (long long)cols);
    if (rows == 0 || cols == 0) return ok();  // empty tensor: nothing to do (the reference returns an empty tensor)
    if (!x || !y) return fail(FQ_ERR_NULL, "x / y must not be NULL");
    if (x == y) return fail(FQ_ERR_ARG, "in-place (y == x) is not supported");
    if (dtype == FQ_DTYPE_F64) {  // correctness path in double arithmetic; no training-mode side buffers
        if (bounds || mk) return fail(FQ_ERR_DTYPE, "float64 tensors: row bounds / STE mask are not produced (use fq_ste_bwd, the reference's data flow)");
        if ((xv && xv->n_inner > 0) || (yv && yv->n_inner > 0)) return fail(FQ_ERR_UNSUPPORTED, "float64 tensors: contiguous rows only");
        return launch_f64_rowwise(ASYM, x, y, idx, scale, rows, cols, bits, sem, (hipStream_t)stream);
    }
    const Consts c = make_consts(bits, dtype, sem);
    RowArgs a{x, y, idx, scale, bounds, rows, cols, c.sym, c.asym, nullptr, 0, 0.f, 0.f, 0u, rows, 0, {}};
    if (!set_pitch(a.xp, xv, rows, cols, esize_of(dtype)) || !set_pitch(a.yp, yv, rows, cols, esize_of(dtype)))
        return fail(FQ_ERR_SHAPE, "fq_rows_view: negative stride, or more than 2^31 - 1 rows");
    if (mk) {
        if (!mk->mask || !bounds) return fail(FQ_ERR_NULL, "train-mode forward needs row_bounds_out and mask_out");
        const int64_t mrw = mask_row_words(cols, esize_of(dtype));
        if (!mrw) return fail(FQ_ERR_UNSUPPORTED, "shape not served by the STE-mask path (see fq_ste_mask_bytes)");
        if (mk->bytes < (size_t)rows * mrw * 8) return fail(FQ_ERR_WORKSPACE, "mask buffer too small: need %zu bytes", (size_t)rows * mrw * 8);
        a.mask = (uint64_t*)mk->mask;
        a.mask_row_words = mrw;
        a.lo = host_rb(mk->lo, dtype);
        a.hi = host_rb(mk->hi, dtype);
        a.clipk = ste_clip_key(a.lo, a.hi, dtype);
    }
    hipStream_t st = (hipStream_t)stream;
    // reciprocal-multiply instead of IEEE divide (only honoured for bf16).  Sym: valid for every bit width, because
    // the bin index is rint() of a bf16 value and so has an 8-bit significand itself; Asym: the divisor 2^bits-1 must
    // have one too, i.e. bits <= 8.
    const bool fast = ASYM ? bits <= 8 : true;
    switch (dtype) {
        case FQ_DTYPE_F32: return launch_rowwise<F32>(ASYM, fast, a, ws, wsb, st);
        case FQ_DTYPE_F16: return launch_rowwise<F16>(ASYM, fast, a, ws, wsb, st);
        default: return launch_rowwise<BF16>(ASYM, fast, a, ws, wsb, st);
    }
}

}  // namespace

#define FQ_API __attribute__((visibility("default")))

extern "C" {

FQ_API int fq_version(void) { return FQ_ABI_VERSION; }

FQ_API const char* fq_build_info(void) {
    static char buf[160];
    snprintf(buf, sizeof(buf), "llmqat_fakequant abi %d, gfx950 (CDNA4, wave64), HIP %d.%d, -ffp-contract=off", FQ_ABI_VERSION,
             HIP_VERSION_MAJOR, HIP_VERSION_MINOR);
    return buf;
}

FQ_API const char* fq_last_error(void) { return g_err; }

FQ_API size_t fq_rowwise_workspace_bytes(int64_t rows, int64_t cols, int dtype) {
    (void)dtype;
    if (rows <= 0 || cols <= WS_COLS_THRESHOLD) return 0;
    return (size_t)rows * 8;
}

FQ_API int fq_sym_fwd(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem, float* row_bounds_out,
                      void* workspace, size_t workspace_bytes, void* stream) {
    return rowwise<false>(x, y, nullptr, nullptr, row_bounds_out, rows, cols, bits, dtype, sem, workspace, workspace_bytes, stream);
}
FQ_API int fq_asym_fwd(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem, float* row_bounds_out,
                       void* workspace, size_t workspace_bytes, void* stream) {
    return rowwise<true>(x, y, nullptr, nullptr, row_bounds_out, rows, cols, bits, dtype, sem, workspace, workspace_bytes, stream);
}
FQ_API int fq_sym_fwd_debug(const void* x, void* y, int32_t* idx_out, float* scale_out, int64_t rows, int64_t cols, int bits,
                            int dtype, int sem, void* workspace, size_t workspace_bytes, void* stream) {
    return rowwise<false>(x, y, idx_out, scale_out, nullptr, rows, cols, bits, dtype, sem, workspace, workspace_bytes, stream);
}
FQ_API int fq_asym_fwd_debug(const void* x, void* y, int32_t* idx_out, float* scale_out, int64_t rows, int64_t cols, int bits,
                             int dtype, int sem, void* workspace, size_t workspace_bytes, void* stream) {
    return rowwise<true>(x, y, idx_out, scale_out, nullptr, rows, cols, bits, dtype, sem, workspace, workspace_bytes, stream);
}

FQ_API size_t fq_ste_mask_bytes(int64_t rows, int64_t cols, int dtype) {
    if (dtype < 0 || dtype > 2 || rows <= 0) return 0;
    return (size_t)rows * (size_t)mask_row_words(cols, esize_of(dtype)) * 8;
}

FQ_API int fq_sym_fwd_train(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem, float lo, float hi,
                            float* row_bounds_out, void* mask_out, size_t mask_bytes, void* stream) {
    MaskArgs mk;
    mk.mask = mask_out; mk.bytes = mask_bytes; mk.lo = lo; mk.hi = hi;
    return rowwise<false>(x, y, nullptr, nullptr, row_bounds_out, rows, cols, bits, dtype, sem, nullptr, 0, stream, &mk);
}
FQ_API int fq_asym_fwd_train(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem, float lo, float hi,
                             float* row_bounds_out, void* mask_out, size_t mask_bytes, void* stream) {
    MaskArgs mk;
    mk.mask = mask_out; mk.bytes = mask_bytes; mk.lo = lo; mk.hi = hi;
    return rowwise<true>(x, y, nullptr, nullptr, row_bounds_out, rows, cols, bits, dtype, sem, nullptr, 0, stream, &mk);
}

FQ_API int fq_sym_fwd_autocast(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem, int wide_out, float lo, float hi,
                               float* row_bounds_out, void* mask_out, size_t mask_bytes, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (dtype != FQ_DTYPE_BF16 && dtype != FQ_DTYPE_F16)
        return fail(FQ_ERR_DTYPE, "autocast arithmetic applies to bf16 / fp16 tensors (fp32 tensors are unaffected by autocast)");
    if (bits < 1 || bits > 31) return fail(FQ_ERR_BITS, "num_bits=%d outside [1, 31]", bits);
    if (sem != FQ_SEM_CPU_EAGER && sem != FQ_SEM_DEVICE_EAGER) return fail(FQ_ERR_ARG, "unknown semantics code %d", sem);
    if (rows < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape");
    if (rows == 0 || cols == 0) return ok();
    if (!x || !y) return fail(FQ_ERR_NULL, "x / y must not be NULL");
    if (x == y) return fail(FQ_ERR_ARG, "in-place (y == x) is not supported");
    const Consts c = make_consts(bits, dtype, sem);  // under autocast `sem` touches `max + 1e-6` only (everything behind it is fp32)
    RowArgs a{x, y, nullptr, nullptr, row_bounds_out, rows, cols, c.sym, c.asym, nullptr, 0, 0.f, 0.f, 0u, rows, 0, {}};
    if (mask_out) {
        if (!row_bounds_out) return fail(FQ_ERR_NULL, "a mask needs row_bounds_out too");
        const int64_t mrw = mask_row_words(cols, 2);
        if (!mrw) return fail(FQ_ERR_UNSUPPORTED, "shape not served by the STE-mask path (see fq_ste_mask_bytes)");
        if (mask_bytes < (size_t)rows * mrw * 8) return fail(FQ_ERR_WORKSPACE, "mask buffer too small: need %zu bytes", (size_t)rows * mrw * 8);
        a.mask = (uint64_t*)mask_out;
        a.mask_row_words = mrw;
        a.lo = host_rb(lo, dtype);
        a.hi = host_rb(hi, dtype);
        a.clipk = ste_clip_key(a.lo, a.hi, dtype);
    }
    hipStream_t st = (hipStream_t)stream;
    return dtype == FQ_DTYPE_BF16 ? launch_sym_autocast<BF16>(wide_out != 0, a, workspace, workspace_bytes, st)
                                  : launch_sym_autocast<F16>(wide_out != 0, a, workspace, workspace_bytes, st);
}

static int sym_fwd_multi_impl(int n, const fq_fwd_tensor* t, const fq_rows_view* const* xv, const fq_rows_view* const* yv, int64_t cols, int dtype,
                              int sem, int autocast, float lo, float hi, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (n < 1 || n > 1 + MAX_MORE || !t) return fail(FQ_ERR_ARG, "a launch takes 1 to %d tensors", 1 + MAX_MORE);
    if (sem != FQ_SEM_CPU_EAGER && sem != FQ_SEM_DEVICE_EAGER) return fail(FQ_ERR_ARG, "unknown semantics code %d", sem);
    if (autocast < 0 || autocast > 2) return fail(FQ_ERR_ARG, "autocast must be 0, 1 or 2");
    if (autocast && dtype == FQ_DTYPE_F32) return fail(FQ_ERR_DTYPE, "autocast arithmetic applies to bf16 / fp16 tensors only");
    if (cols <= 0) return fail(FQ_ERR_SHAPE, "multi-tensor launch needs non-empty tensors");
    const int64_t mrw = mask_row_words(cols, esize_of(dtype));
    if (!mrw) return fail(FQ_ERR_UNSUPPORTED, "shape not served by the register kernels (see fq_ste_mask_bytes)");
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (t[i].bits < 1 || t[i].bits > 31) return fail(FQ_ERR_BITS, "num_bits outside [1, 31]");
        if (t[i].rows <= 0) return fail(FQ_ERR_SHAPE, "multi-tensor launch needs non-empty tensors");
        if (!t[i].x || !t[i].y) return fail(FQ_ERR_NULL, "multi-tensor launch: x / y of every tensor required");
        if (t[i].mask && !t[i].row_bounds) return fail(FQ_ERR_NULL, "a mask needs its row_bounds too");
        if (t[i].mask && t[i].mask_bytes < (size_t)t[i].rows * mrw * 8) return fail(FQ_ERR_WORKSPACE, "mask buffer too small");
        total += t[i].rows;
    }
    const Consts c0 = make_consts(t[0].bits, dtype, sem);
    RowArgs a{t[0].x, t[0].y, nullptr, nullptr, t[0].row_bounds, total, cols, c0.sym, c0.asym, (uint64_t*)t[0].mask, mrw, host_rb(lo, dtype),
              host_rb(hi, dtype), ste_clip_key(host_rb(lo, dtype), host_rb(hi, dtype), dtype), t[0].rows, n - 1, {}};
    int64_t begin = t[0].rows;
    for (int i = 1; i < n; ++i) {
        a.more[i - 1] = TensorSlot{begin, t[i].x, t[i].y, t[i].row_bounds, (uint64_t*)t[i].mask, make_consts(t[i].bits, dtype, sem).sym.qmax, {}, {}};
        begin += t[i].rows;
    }
    {   // rows that do not follow one another (the _v entry point); the results of an autocast = 2 launch are fp32
        const int xe = esize_of(dtype), ye = autocast == 2 ? 4 : xe;
        bool okv = true;
        for (int i = 0; i < n; ++i) {
            RowPitch& px = i ? a.more[i - 1].xp : a.xp;
            RowPitch& py = i ? a.more[i - 1].yp : a.yp;
            okv = okv && set_pitch(px, xv ? xv[i] : nullptr, t[i].rows, cols, xe) && set_pitch(py, yv ? yv[i] : nullptr, t[i].rows, cols, ye);
        }
        if (!okv) return fail(FQ_ERR_SHAPE, "fq_rows_view: negative stride, or more than 2^31 - 1 rows");
    }
    hipStream_t st = (hipStream_t)stream;
    if (autocast) {
        const bool wide = autocast == 2;  // the y are fp32; masks (if any) in the wide layout, for fq_ste_bwd_mask_wide
        return dtype == FQ_DTYPE_BF16 ? launch_sym_autocast<BF16>(wide, a, nullptr, 0, st) : launch_sym_autocast<F16>(wide, a, nullptr, 0, st);
    }
    switch (dtype) {
        case FQ_DTYPE_F32: return launch_rowwise<F32>(false, true, a, nullptr, 0, st);
        case FQ_DTYPE_F16: return launch_rowwise<F16>(false, true, a, nullptr, 0, st);
        default: return launch_rowwise<BF16>(false, true, a, nullptr, 0, st);
    }
}

FQ_API int fq_sym_fwd_multi(int n, const fq_fwd_tensor* t, int64_t cols, int dtype, int sem, int autocast, float lo, float hi, void* stream) {
    return sym_fwd_multi_impl(n, t, nullptr, nullptr, cols, dtype, sem, autocast, lo, hi, stream);
}

FQ_API int fq_sym_fwd_multi_v(int n, const fq_fwd_tensor_v* tv, int64_t cols, int dtype, int sem, int autocast, float lo, float hi, void* stream) {
    if (n < 1 || n > 1 + MAX_MORE || !tv) return fail(FQ_ERR_ARG, "a launch takes 1 to %d tensors", 1 + MAX_MORE);
    fq_fwd_tensor t[1 + MAX_MORE];
    const fq_rows_view *xv[1 + MAX_MORE], *yv[1 + MAX_MORE];
    for (int i = 0; i < n; ++i) {
        t[i] = fq_fwd_tensor{tv[i].x, tv[i].y, tv[i].rows, tv[i].bits, tv[i].row_bounds, tv[i].mask, tv[i].mask_bytes};
        xv[i] = &tv[i].xv;
        yv[i] = &tv[i].yv;
    }
    return sym_fwd_multi_impl(n, t, xv, yv, cols, dtype, sem, autocast, lo, hi, stream);
}

FQ_API int fq_rowwise_fwd_v(int asym, const void* x, const fq_rows_view* xv, void* y, const fq_rows_view* yv, int64_t rows, int64_t cols, int bits,
                            int dtype, int sem, float lo, float hi, float* row_bounds_out, void* mask_out, size_t mask_bytes, void* stream) {
    MaskArgs mk;
    mk.mask = mask_out; mk.bytes = mask_bytes; mk.lo = lo; mk.hi = hi;
    const MaskArgs* m = mask_out ? &mk : nullptr;
    return asym ? rowwise<true>(x, y, nullptr, nullptr, row_bounds_out, rows, cols, bits, dtype, sem, nullptr, 0, stream, m, xv, yv)
                : rowwise<false>(x, y, nullptr, nullptr, row_bounds_out, rows, cols, bits, dtype, sem, nullptr, 0, stream, m, xv, yv);
}

FQ_API int fq_sym_fwd_pair(const void* x0, void* y0, int64_t rows0, int bits0, float* row_bounds0, void* mask0, size_t mask_bytes0,
                           const void* x1, void* y1, int64_t rows1, int bits1, float* row_bounds1, void* mask1, size_t mask_bytes1,
                           int64_t cols, int dtype, int sem, int autocast, float lo, float hi, void* stream) {
    const fq_fwd_tensor t[2] = {{x0, y0, rows0, bits0, row_bounds0, mask0, mask_bytes0}, {x1, y1, rows1, bits1, row_bounds1, mask1, mask_bytes1}};
    return fq_sym_fwd_multi(2, t, cols, dtype, sem, autocast, lo, hi, stream);
}

static int ste_bwd_mask_multi_impl(int n, const fq_bwd_tensor* t, const fq_rows_view* const* gv, const fq_rows_view* const* ov, int64_t cols, float lo,
                                   float hi, int dtype, int wide_grad, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (wide_grad && dtype == FQ_DTYPE_F32) return fail(FQ_ERR_DTYPE, "fp32-gradient STE backward: the input dtype must be bf16 / fp16");
    if (n < 1 || n > 1 + MAX_MORE || !t) return fail(FQ_ERR_ARG, "a launch takes 1 to %d tensors", 1 + MAX_MORE);
    if (cols <= 0) return fail(FQ_ERR_SHAPE, "multi-tensor launch needs non-empty tensors");
    if (!mask_row_words(cols, esize_of(dtype))) return fail(FQ_ERR_UNSUPPORTED, "shape not served by the STE-mask path");
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (t[i].rows <= 0) return fail(FQ_ERR_SHAPE, "multi-tensor launch needs non-empty tensors");
        if (!t[i].g || !t[i].gx || !t[i].row_bounds || !t[i].mask) return fail(FQ_ERR_NULL, "multi-tensor launch: all buffers required");
        total += t[i].rows;
    }
    lo = host_rb(lo, dtype);
    hi = host_rb(hi, dtype);
    SteLaunch L{};
    L.n = n;
    for (int i = 0; i < n; ++i) {
        L.t[i] = SteSlot{t[i].g, t[i].gx, t[i].row_bounds, (const uint64_t*)t[i].mask, t[i].rows, 0, 0, {}, {}};
        const int oe = esize_of(dtype), ge = wide_grad ? 4 : oe;   // behind a fp32-result forward the gradient is fp32
        if (!set_pitch(L.t[i].gp, gv ? gv[i] : nullptr, t[i].rows, cols, ge) || !set_pitch(L.t[i].op, ov ? ov[i] : nullptr, t[i].rows, cols, oe))
            return fail(FQ_ERR_SHAPE, "fq_rows_view: negative stride, or more than 2^31 - 1 rows");
        if (t[i].gx == (const void*)t[i].g && memcmp(&L.t[i].gp, &L.t[i].op, sizeof(RowPitch)) != 0)
            return fail(FQ_ERR_ARG, "an in-place tensor (gx == g) needs equal views");
    }
    hipStream_t st = (hipStream_t)stream;
    if (wide_grad) return dtype == FQ_DTYPE_BF16 ? launch_ste_mask_wide<BF16>(L, cols, lo, hi, st) : launch_ste_mask_wide<F16>(L, cols, lo, hi, st);
    switch (dtype) {
        case FQ_DTYPE_F32: return launch_ste_mask<F32>(L, cols, lo, hi, st);
        case FQ_DTYPE_F16: return launch_ste_mask<F16>(L, cols, lo, hi, st);
        default: return launch_ste_mask<BF16>(L, cols, lo, hi, st);
    }
}

FQ_API int fq_ste_bwd_mask_multi(int n, const fq_bwd_tensor* t, int64_t cols, float lo, float hi, int dtype, int wide_grad, void* stream) {
    return ste_bwd_mask_multi_impl(n, t, nullptr, nullptr, cols, lo, hi, dtype, wide_grad, stream);
}

FQ_API int fq_ste_bwd_mask_multi_v(int n, const fq_bwd_tensor_v* tv, int64_t cols, float lo, float hi, int dtype, int wide_grad, void* stream) {
    if (n < 1 || n > 1 + MAX_MORE || !tv) return fail(FQ_ERR_ARG, "a launch takes 1 to %d tensors", 1 + MAX_MORE);
    fq_bwd_tensor t[1 + MAX_MORE];
    const fq_rows_view *gv[1 + MAX_MORE], *ov[1 + MAX_MORE];
    for (int i = 0; i < n; ++i) {
        t[i] = fq_bwd_tensor{tv[i].g, tv[i].gx, tv[i].rows, tv[i].row_bounds, tv[i].mask};
        gv[i] = &tv[i].gv;
        ov[i] = &tv[i].gxv;
    }
    return ste_bwd_mask_multi_impl(n, t, gv, ov, cols, lo, hi, dtype, wide_grad, stream);
}

FQ_API int fq_ste_bwd_mask_pair(const void* g0, void* gx0, int64_t rows0, const float* row_bounds0, const void* mask0,
                                const void* g1, void* gx1, int64_t rows1, const float* row_bounds1, const void* mask1,
                                int64_t cols, float lo, float hi, int dtype, void* stream) {
    const fq_bwd_tensor t[2] = {{g0, gx0, rows0, row_bounds0, mask0}, {g1, gx1, rows1, row_bounds1, mask1}};
    return fq_ste_bwd_mask_multi(2, t, cols, lo, hi, dtype, 0, stream);
}

FQ_API int fq_ste_bwd_mask_wide(const void* g0, void* gx0, int64_t rows0, const float* row_bounds0, const void* mask0,
                                const void* g1, void* gx1, int64_t rows1, const float* row_bounds1, const void* mask1,
                                int64_t cols, float lo, float hi, int dtype, void* stream) {
    if (dtype != FQ_DTYPE_BF16 && dtype != FQ_DTYPE_F16) return fail(FQ_ERR_DTYPE, "fp32-gradient STE backward: the input dtype must be bf16 / fp16");
    if (rows0 < 0 || rows1 < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape");
    if ((rows0 == 0 && rows1 == 0) || cols == 0) return ok();
    if (rows0 == 0) return fail(FQ_ERR_SHAPE, "a single tensor goes first (rows1 = 0)");
    const fq_bwd_tensor t[2] = {{g0, gx0, rows0, row_bounds0, mask0}, {g1, gx1, rows1, row_bounds1, mask1}};
    return fq_ste_bwd_mask_multi(rows1 ? 2 : 1, t, cols, lo, hi, dtype, 1, stream);
}

FQ_API int fq_ste_bwd_mask(const void* g, void* gx, int64_t rows, int64_t cols, float lo, float hi, const float* row_bounds,
                           const void* mask, size_t mask_bytes, int dtype, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (rows < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape");
    if (rows == 0 || cols == 0) return ok();
    if (!g || !gx || !row_bounds || !mask) return fail(FQ_ERR_NULL, "g / gx / row_bounds / mask must not be NULL");
    const int64_t mrw = mask_row_words(cols, esize_of(dtype));
    if (!mrw) return fail(FQ_ERR_UNSUPPORTED, "shape not served by the STE-mask path (see fq_ste_mask_bytes)");
    if (mask_bytes < (size_t)rows * mrw * 8) return fail(FQ_ERR_WORKSPACE, "mask buffer too small");
    const fq_bwd_tensor t{g, gx, rows, row_bounds, mask};
    return fq_ste_bwd_mask_multi(1, &t, cols, lo, hi, dtype, 0, stream);
}

FQ_API int fq_w12_fwd(const void* w, const void* scale, void* out, int64_t rows, int64_t cols, int w_bits, int scale_per_row,
                      int dtype, void* stream) {
    if (dtype < 0 || dtype > FQ_DTYPE_F64) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (w_bits != 1 && w_bits != 2) return fail(FQ_ERR_BITS, "w_bits=%d: this entry point serves 1 and 2", w_bits);
    if (rows < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape");
    if (rows == 0 || cols == 0) return ok();
    if (!w || !scale || !out) return fail(FQ_ERR_NULL, "w / scale / out must not be NULL");
    if (dtype == FQ_DTYPE_F64) return launch_f64_w12(w, scale, out, rows, cols, w_bits, scale_per_row, (hipStream_t)stream);
    // clip_val = 1 - 1e-2 (utils_quant.py:217): a Python double; the clamp compares in fp32 on the device and in the
    // tensor dtype on the CPU -- both give the same results (DESIGN.md "Numerics"), fp32 is used here
    const float cv = (float)(1.0 - 1e-2);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FQ_DTYPE_F32: return launch_w12<F32>(w, scale, out, rows, cols, w_bits, scale_per_row, cv, st);
        case FQ_DTYPE_F16: return launch_w12<F16>(w, scale, out, rows, cols, w_bits, scale_per_row, cv, st);
        default: return launch_w12<BF16>(w, scale, out, rows, cols, w_bits, scale_per_row, cv, st);
    }
}

FQ_API int fq_w12_fwd_rows(const void* w, void* out, void* scale_out, int64_t rows, int64_t cols, int w_bits, int dtype, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (w_bits != 1 && w_bits != 2) return fail(FQ_ERR_BITS, "w_bits=%d: this entry point serves 1 and 2", w_bits);
    if (rows < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape");
    if (rows == 0 || cols == 0) return ok();
    if (!w || !out) return fail(FQ_ERR_NULL, "w / out must not be NULL");
    const float cv = (float)(1.0 - 1e-2);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FQ_DTYPE_F32: return launch_w12_rows<F32>(w, out, scale_out, rows, cols, w_bits, cv, st);
        case FQ_DTYPE_F16: return launch_w12_rows<F16>(w, out, scale_out, rows, cols, w_bits, cv, st);
        default: return launch_w12_rows<BF16>(w, out, scale_out, rows, cols, w_bits, cv, st);
    }
}

FQ_API int fq_ste_bwd(const void* g, const void* x, void* gx, int64_t n, float lo, float hi, int dtype, void* stream) {
    if (dtype < 0 || dtype > FQ_DTYPE_F64) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (n < 0) return fail(FQ_ERR_SHAPE, "negative n");
    if (n == 0) return ok();
    if (!g || !x || !gx) return fail(FQ_ERR_NULL, "g / x / gx must not be NULL");
    if (dtype == FQ_DTYPE_F64) return launch_f64_ste(g, x, gx, n, lo, hi, (hipStream_t)stream);  // the clip is a float32 tensor (:198,:245): its values as doubles
    lo = host_rb(lo, dtype);  // the reference compares in the tensor dtype (utils_quant.py:85-86)
    hi = host_rb(hi, dtype);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FQ_DTYPE_F32: return launch_ste<F32>(g, x, gx, n, lo, hi, st);
        case FQ_DTYPE_F16: return launch_ste<F16>(g, x, gx, n, lo, hi, st);
        default: return launch_ste<BF16>(g, x, gx, n, lo, hi, st);
    }
}

FQ_API int fq_ste_bwd_rows(const void* g, const void* x, void* gx, int64_t rows, int64_t cols, float lo, float hi,
                           const float* row_bounds, int dtype, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (rows < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape");
    if (rows == 0 || cols == 0) return ok();
    if (!g || !x || !gx || !row_bounds) return fail(FQ_ERR_NULL, "g / x / gx / row_bounds must not be NULL");
    lo = host_rb(lo, dtype);
    hi = host_rb(hi, dtype);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FQ_DTYPE_F32: return launch_ste_rows<F32>(g, x, gx, rows, cols, lo, hi, row_bounds, st);
        case FQ_DTYPE_F16: return launch_ste_rows<F16>(g, x, gx, rows, cols, lo, hi, row_bounds, st);
        default: return launch_ste_rows<BF16>(g, x, gx, rows, cols, lo, hi, row_bounds, st);
    }
}

FQ_API int fq_ste_bwd_v(const void* g, const fq_rows_view* gv, const void* x, const fq_rows_view* xv, void* gx, const fq_rows_view* gxv, int64_t rows,
                        int64_t cols, float lo, float hi, const float* row_bounds, int dtype, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (rows < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape");
    if (rows == 0 || cols == 0) return ok();
    if (!g || !x || !gx) return fail(FQ_ERR_NULL, "g / x / gx must not be NULL");
    StePitch3 p{};
    const int es = esize_of(dtype);
    if (!set_pitch(p.g, gv, rows, cols, es) || !set_pitch(p.x, xv, rows, cols, es) || !set_pitch(p.o, gxv, rows, cols, es))
        return fail(FQ_ERR_SHAPE, "fq_rows_view: negative stride, or more than 2^31 - 1 rows");
    lo = host_rb(lo, dtype);
    hi = host_rb(hi, dtype);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FQ_DTYPE_F32: return launch_ste_rows<F32>(g, x, gx, rows, cols, lo, hi, row_bounds, st, p);
        case FQ_DTYPE_F16: return launch_ste_rows<F16>(g, x, gx, rows, cols, lo, hi, row_bounds, st, p);
        default: return launch_ste_rows<BF16>(g, x, gx, rows, cols, lo, hi, row_bounds, st, p);
    }
}

}  // extern "C"
