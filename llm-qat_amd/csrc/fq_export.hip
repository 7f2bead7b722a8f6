// fq_export.hip -- host side of the export / scale pre-pass entry points (include/llmqat_fakequant.h):
// fq_sym_export, fq_asym_export, fq_sym_row_scales, fq_export_bins_bytes.  No allocation, no synchronisation.
#include "../../include/llmqat_fakequant.h"

#include <hip/hip_runtime.h>

#include "fq_export.h"
#include "fq_launch.h"

using namespace fq;

namespace {

template <int DT, bool ASYM, bool NTL> void launch_export_reg(const ExportArgs& a, int64_t nvec, hipStream_t st) {
#define R(TPR, V)                                                                                                              \
    case V:                                                                                                                    \
        FQ_LAUNCHK((row_export_kernel<DT, TPR, V, ASYM, NTL>), dim3((unsigned)(TPR == 64 ? (a.rows + 3) / 4 : a.rows)), \
                           dim3(TPR == 64 ? 256 : TPR), 0, st, a);                                                             \
        break;
    // launch shapes of the forward kernel (fq_dtype_impl.h launch_reg): same loads, same reduction
    if (nvec <= 192) {
        switch ((int)((nvec + 63) / 64)) { R(64, 1) R(64, 2) R(64, 3) }
    } else if (nvec <= 384) {
        switch ((int)((nvec + 127) / 128)) { R(128, 2) R(128, 3) }
    } else if (nvec <= 768) {
        switch ((int)((nvec + 255) / 256)) { R(256, 2) R(256, 3) }
    } else if (nvec <= 4096) {
        switch ((int)((nvec + 511) / 512)) { R(512, 2) R(512, 3) R(512, 4) case 5: R(512, 6) case 7: R(512, 8) }
    } else {
        switch ((int)((nvec + 1023) / 1024)) { case 5: R(1024, 6) case 7: R(1024, 8) }
    }
#undef R
}

template <int DT, bool ASYM> int export_t(const ExportArgs& a, hipStream_t st) {
    using T = Ty<DT>;
    constexpr int EPV = 16 / T::ESIZE;
    begin_launches();
    if (a.rows > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows=%lld exceeds the grid limit", (long long)a.rows);
    const int64_t nvec = a.cols / EPV;
    // packed stores need the bins row to start on the store's natural boundary: EPV elements -> EPV*cbits/8 bytes
    const int64_t vb = a.container == BINS_INT4 ? EPV / 2 : a.container == BINS_INT8 ? EPV : a.container == BINS_INT16 ? EPV * 2 : 1;
    const bool vec_ok = aligned16(a.x) && a.cols % EPV == 0 && nvec <= REG_MAX_VEC &&
                        (a.container == BINS_NONE || (reinterpret_cast<uintptr_t>(a.bins) % vb == 0));
    if (vec_ok) {
        if (a.rows * a.cols * T::ESIZE >= NT_LOAD_MIN_BYTES) launch_export_reg<DT, ASYM, true>(a, nvec, st);
        else launch_export_reg<DT, ASYM, false>(a, nvec, st);
    } else {
        if (a.mask) return fail(FQ_ERR_UNSUPPORTED, "STE mask: rows must be 16-byte aligned and fit the register kernels (see fq_ste_mask_bytes)");
        FQ_LAUNCHK((row_export_generic_kernel<DT, ASYM>), dim3((unsigned)a.rows), dim3(256), 0, st, a);
    }
    return launch_result();
}

template <bool ASYM>
int export_entry(const void* x, void* bins, float* scales, int32_t* overflow, int64_t rows, int64_t cols, int bits, int container, int dtype,
                 int sem, int autocast, float lo, float hi, float* bounds, void* mask, size_t mask_bytes, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (bits < 1 || bits > 31) return fail(FQ_ERR_BITS, "num_bits=%d outside [1, 31]", bits);
    if (sem != FQ_SEM_CPU_EAGER && sem != FQ_SEM_DEVICE_EAGER) return fail(FQ_ERR_ARG, "unknown semantics code %d", sem);
    if (container < BINS_NONE || container > BINS_INT16) return fail(FQ_ERR_ARG, "unknown bins container %d", container);
    if (autocast && (ASYM || dtype == FQ_DTYPE_F32)) return fail(FQ_ERR_DTYPE, "autocast arithmetic applies to SymQuantizer on bf16 / fp16 tensors only");
    if (rows < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape rows=%lld cols=%lld", (long long)rows, (long long)cols);
    if (rows == 0 || cols == 0) return ok();
    if (!x) return fail(FQ_ERR_NULL, "x must not be NULL");
    if (container != BINS_NONE && !bins) return fail(FQ_ERR_NULL, "bins_out must not be NULL for this container");
    if (container == BINS_NONE && !scales && !bounds && !mask) return fail(FQ_ERR_NULL, "nothing to produce: scales_out, row_bounds_out and mask_out are all NULL");
    const Consts c = make_consts(bits, dtype, sem);
    ExportArgs a{};
    a.x = x;
    a.bins = bins;
    a.scales = scales;
    a.overflow = overflow;
    a.bounds = bounds;
    a.rows = rows;
    a.cols = cols;
    a.row_bytes = export_row_bytes(cols, container);
    a.sym = c.sym;
    a.asym = c.asym;
    a.container = container;
    a.autocast = autocast ? 1 : 0;
    const int cb = container == BINS_INT4 ? 4 : container == BINS_INT8 ? 8 : 16;
    a.cmin = ASYM ? 0.f : -(float)(1 << (cb - 1));
    a.cmax = ASYM ? (float)((1 << cb) - 1) : (float)((1 << (cb - 1)) - 1);
    if (mask) {
        if (!bounds) return fail(FQ_ERR_NULL, "a mask needs row_bounds_out too");
        const int64_t mrw = mask_row_words(cols, esize_of(dtype));
        if (!mrw) return fail(FQ_ERR_UNSUPPORTED, "shape not served by the STE-mask path (see fq_ste_mask_bytes)");
        if (mask_bytes < (size_t)rows * mrw * 8) return fail(FQ_ERR_WORKSPACE, "mask buffer too small: need %zu bytes", (size_t)rows * mrw * 8);
        a.mask = (uint64_t*)mask;
        a.mask_row_words = mrw;
        a.lo = host_rb(lo, dtype);
        a.hi = host_rb(hi, dtype);
        a.clipk = ste_clip_key(a.lo, a.hi, dtype);
    }
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FQ_DTYPE_F32: return export_t<F32, ASYM>(a, st);
        case FQ_DTYPE_F16: return export_t<F16, ASYM>(a, st);
        default: return export_t<BF16, ASYM>(a, st);
    }
}

}  // namespace

#define FQ_API __attribute__((visibility("default")))

extern "C" {

FQ_API size_t fq_export_bins_bytes(int64_t rows, int64_t cols, int container) {
    if (rows <= 0 || cols <= 0 || container <= BINS_NONE || container > BINS_INT16) return 0;
    return (size_t)rows * (size_t)export_row_bytes(cols, container);
}

FQ_API int fq_sym_export(const void* x, void* bins_out, float* scales_out, int32_t* overflow_out, int64_t rows, int64_t cols, int bits,
                         int container, int dtype, int sem, int autocast, void* stream) {
    if (container == BINS_NONE) return fail(FQ_ERR_ARG, "fq_sym_export needs a bins container (scales only: fq_sym_row_scales)");
    return export_entry<false>(x, bins_out, scales_out, overflow_out, rows, cols, bits, container, dtype, sem, autocast, 0.f, 0.f, nullptr, nullptr, 0, stream);
}

FQ_API int fq_asym_export(const void* x, void* bins_out, float* scales_out, int32_t* overflow_out, int64_t rows, int64_t cols, int bits,
                          int container, int dtype, int sem, void* stream) {
    if (container == BINS_NONE) return fail(FQ_ERR_ARG, "fq_asym_export needs a bins container");
    return export_entry<true>(x, bins_out, scales_out, overflow_out, rows, cols, bits, container, dtype, sem, 0, 0.f, 0.f, nullptr, nullptr, 0, stream);
}

FQ_API int fq_sym_row_scales(const void* x, float* scales_out, int64_t rows, int64_t cols, int bits, int dtype, int sem, int autocast, float lo,
                             float hi, float* row_bounds_out, void* mask_out, size_t mask_bytes, void* stream) {
    return export_entry<false>(x, nullptr, scales_out, nullptr, rows, cols, bits, BINS_NONE, dtype, sem, autocast, lo, hi, row_bounds_out, mask_out,
                               mask_bytes, stream);
}

}  // extern "C"
