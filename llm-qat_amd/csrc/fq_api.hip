// fq_api.hip -- host side of the C ABI declared in include/llmqat_fakequant.h:
// argument validation, kernel selection, launches.  No allocation, no synchronisation.
#include "../../include/llmqat_fakequant.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "fq_kernels.h"

using namespace fq;

namespace {

thread_local char g_err[320] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
inline int ok() {
    g_err[0] = 0;
    return FQ_OK;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- host-side round-to-dtype for the launch-uniform scalars (1e-6, 1e-8, clip bounds) ----
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
float host_rb(float v, int dt) {
    if (dt == FQ_DTYPE_BF16) {
        uint32_t u = f2u(v);
        if ((u & 0x7FFFFFFFu) > 0x7F800000u) return u2f(0x7FC00000u);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
        return u2f(u);
    }
    if (dt == FQ_DTYPE_F16) {
        uint32_t x = f2u(v), sign = x & 0x80000000u;
        x &= 0x7FFFFFFFu;
        if (x > 0x7F800000u) return u2f(0x7FC00000u);
        if (x >= 0x477FF000u) return u2f(sign | 0x7F800000u);  // rounds to +-inf in fp16
        if (x < 0x38800000u) {                                  // fp16 subnormal range: quantum 2^-24
            const float q = 5.9604644775390625e-08f;
            float k = __builtin_rintf(u2f(x) / q);              // x/q < 1024 is exact (q is a power of two); rintf = RNE
            return u2f(f2u(k * q) | sign);
        }
        uint32_t odd = (x >> 13) & 1u;
        x = (x + 0xFFFu + odd) & 0xFFFFE000u;
        return u2f(sign | x);
    }
    return v;
}

struct Consts {
    SymConst sym;
    AsymConst asym;
};
Consts make_consts(int bits, int dt, int sem) {
    Consts c;
    const float c6 = 1e-6f, c8 = 1e-8f;
    c.sym.qmax = (float)(double)((1u << (bits - 1)) - 1u);
    c.sym.c6 = sem == FQ_SEM_CPU_EAGER ? host_rb(c6, dt) : c6;
    const double S = (double)(bits >= 32 ? 4294967295.0 : (double)((1ull << bits) - 1ull));
    c.asym.S = (float)S;
    c.asym.invS = 1.0f / (float)S;
    c.asym.c8 = sem == FQ_SEM_CPU_EAGER ? host_rb(c8, dt) : c8;
    c.asym.mul_inv = sem == FQ_SEM_DEVICE_EAGER ? 1 : 0;
    return c;
}

// ---- kernel selection -------------------------------------------------------------------
constexpr int64_t REG_MAX_VEC = 1024 * 8;   // longest row (in 16-byte vectors) the register kernels hold
constexpr int64_t GENERIC_MAX_COLS = 32768; // longest row the scalar-load kernel sweeps
constexpr int64_t WS_COLS_THRESHOLD = 32768;

#define FQ_LAUNCH(kern, grid, block, st, ...) hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3(block), 0, st, __VA_ARGS__)

template <int DT, int TPR, bool ASYM, bool FAST>
void launch_reg(const RowArgs& a, int vpt, hipStream_t st) {
    const int64_t grid = TPR == 64 ? (a.rows + 3) / 4 : a.rows;
    constexpr int BLOCK = TPR == 64 ? 256 : TPR;
    switch (vpt) {
#define C(V) \
    case V: FQ_LAUNCH((row_reg_kernel<DT, TPR, V, ASYM, FAST>), grid, BLOCK, st, a); break;
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8)
#undef C
        default: break;
    }
}

template <int DT, bool ASYM, bool FAST>
int launch_rowwise_t(RowArgs a, void* ws, size_t wsb, hipStream_t st) {
    using T = Ty<DT>;
    constexpr int EPV = 16 / T::ESIZE;
    const bool vec_ok = aligned16(a.x) && aligned16(a.y) && (a.cols % EPV == 0);
    const int64_t nvec = a.cols / EPV;
    bool two_pass = false, two_pass_vec = false;
    if (vec_ok && nvec <= REG_MAX_VEC) {
        if (a.rows > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows=%lld exceeds the grid limit", (long long)a.rows);
        if (nvec <= 128) launch_reg<DT, 64, ASYM, FAST>(a, (int)((nvec + 63) / 64), st);
        else if (nvec <= 2048) launch_reg<DT, 256, ASYM, FAST>(a, (int)((nvec + 255) / 256), st);
        else launch_reg<DT, 1024, ASYM, FAST>(a, (int)((nvec + 1023) / 1024), st);
    } else if (vec_ok) {
        two_pass = two_pass_vec = true;
    } else if (a.cols <= GENERIC_MAX_COLS) {
        if (a.rows > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows=%lld exceeds the grid limit", (long long)a.rows);
        if (a.cols <= 1024) FQ_LAUNCH((row_generic_kernel<DT, 64, ASYM>), (a.rows + 3) / 4, 256, st, a);
        else FQ_LAUNCH((row_generic_kernel<DT, 256, ASYM>), a.rows, 256, st, a);
    } else {
        two_pass = true;
    }
    if (two_pass) {
        if (!ws || wsb < (size_t)a.rows * 8) return fail(FQ_ERR_WORKSPACE, "two-pass path needs %zu workspace bytes, got %zu", (size_t)a.rows * 8, wsb);
        const int64_t ch = two_pass_vec ? tp_chunk_elems<DT, true>() : tp_chunk_elems<DT, false>();
        const int64_t chunks = (a.cols + ch - 1) / ch;
        if (a.rows * chunks > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows*chunks exceeds the grid limit");
        if (hipMemsetAsync(ws, 0, (size_t)a.rows * 8, st) != hipSuccess) return fail(FQ_ERR_LAUNCH, "hipMemsetAsync failed");
        uint32_t* w = (uint32_t*)ws;
        if (two_pass_vec) {
            FQ_LAUNCH((stats_kernel<DT, ASYM, true>), a.rows * chunks, TP_THREADS, st, a, w, chunks);
            FQ_LAUNCH((apply_kernel<DT, ASYM, FAST, true>), a.rows * chunks, TP_THREADS, st, a, (const uint32_t*)w, chunks);
        } else {
            FQ_LAUNCH((stats_kernel<DT, ASYM, false>), a.rows * chunks, TP_THREADS, st, a, w, chunks);
            FQ_LAUNCH((apply_kernel<DT, ASYM, FAST, false>), a.rows * chunks, TP_THREADS, st, a, (const uint32_t*)w, chunks);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FQ_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return ok();
}

template <bool ASYM>
int rowwise(const void* x, void* y, int32_t* idx, float* scale, float* bounds, int64_t rows, int64_t cols, int bits, int dtype,
            int sem, void* ws, size_t wsb, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (bits < (ASYM ? 1 : 2) || bits > 31) return fail(FQ_ERR_BITS, "num_bits=%d outside [%d, 31]", bits, ASYM ? 1 : 2);
    if (sem != FQ_SEM_CPU_EAGER && sem != FQ_SEM_DEVICE_EAGER) return fail(FQ_ERR_ARG, "unknown semantics code %d", sem);
    if (rows < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape rows=%lld cols=%lld", (long long)rows, (long long)cols);
    if (rows == 0 || cols == 0) return ok();  // empty tensor: nothing to do (the reference returns an empty tensor)
    if (!x || !y) return fail(FQ_ERR_NULL, "x / y must not be NULL");
    if (x == y) return fail(FQ_ERR_ARG, "in-place (y == x) is not supported");
    const Consts c = make_consts(bits, dtype, sem);
    RowArgs a{x, y, idx, scale, bounds, rows, cols, c.sym, c.asym};
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FQ_DTYPE_F32: return launch_rowwise_t<F32, ASYM, false>(a, ws, wsb, st);
        case FQ_DTYPE_F16: return launch_rowwise_t<F16, ASYM, false>(a, ws, wsb, st);
        default:
            // bf16, Sym, <= 8 bits: the final divide is replaced by a multiply with the row's
            // reciprocal -- provably bit-identical after the bf16 rounding (DESIGN.md "Numerics").
            if (!ASYM && bits <= 8) return launch_rowwise_t<BF16, ASYM, !ASYM>(a, ws, wsb, st);
            return launch_rowwise_t<BF16, ASYM, false>(a, ws, wsb, st);
    }
}

template <int DT>
int ste_t(const void* g, const void* x, void* gx, int64_t n, float lo, float hi, hipStream_t st) {
    using T = Ty<DT>;
    constexpr int EPV = 16 / T::ESIZE;
    constexpr int UNR = 4;
    if (aligned16(g) && aligned16(x) && aligned16(gx) && n % EPV == 0) {
        const int64_t nvec = n / EPV;
        const int64_t grid = (nvec + STE_THREADS * UNR - 1) / (STE_THREADS * UNR);
        if (grid > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "n too large");
        FQ_LAUNCH((ste_vec_kernel<DT, UNR>), grid, STE_THREADS, st, (const uint4*)g, (const uint4*)x, (uint4*)gx, nvec, lo, hi);
    } else {
        int64_t grid = (n + STE_THREADS - 1) / STE_THREADS;
        if (grid > 8192) grid = 8192;
        FQ_LAUNCH((ste_scalar_kernel<DT>), grid, STE_THREADS, st, g, x, gx, n, lo, hi);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FQ_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return ok();
}

template <int DT>
int ste_rows_t(const void* g, const void* x, void* gx, int64_t rows, int64_t cols, float lo, float hi, const float* bounds,
               hipStream_t st) {
    using T = Ty<DT>;
    constexpr int EPV = 16 / T::ESIZE;
    constexpr int UNR = 4;
    if (!(aligned16(g) && aligned16(x) && aligned16(gx) && cols % EPV == 0))
        return ste_t<DT>(g, x, gx, rows * cols, lo, hi, st);  // odd layout: plain path, same result
    const int64_t ch = (int64_t)STE_THREADS * UNR * EPV;
    const int64_t chunks = (cols + ch - 1) / ch;
    if (rows * chunks > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows*chunks exceeds the grid limit");
    FQ_LAUNCH((ste_rows_kernel<DT, UNR>), rows * chunks, STE_THREADS, st, g, x, gx, cols, chunks, bounds, lo, hi);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FQ_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return ok();
}

}  // namespace

extern "C" {

int fq_version(void) { return FQ_ABI_VERSION; }

const char* fq_build_info(void) {
    static char buf[160];
    snprintf(buf, sizeof(buf), "llmqat_fakequant abi %d, gfx950 (CDNA4, wave64), HIP %d.%d, -ffp-contract=off", FQ_ABI_VERSION,
             HIP_VERSION_MAJOR, HIP_VERSION_MINOR);
    return buf;
}

const char* fq_last_error(void) { return g_err; }

size_t fq_rowwise_workspace_bytes(int64_t rows, int64_t cols, int dtype) {
    (void)dtype;
    if (rows <= 0 || cols <= WS_COLS_THRESHOLD) return 0;
    return (size_t)rows * 8;
}

int fq_sym_fwd(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem, float* row_bounds_out,
               void* workspace, size_t workspace_bytes, void* stream) {
    return rowwise<false>(x, y, nullptr, nullptr, row_bounds_out, rows, cols, bits, dtype, sem, workspace, workspace_bytes, stream);
}
int fq_asym_fwd(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem, float* row_bounds_out,
                void* workspace, size_t workspace_bytes, void* stream) {
    return rowwise<true>(x, y, nullptr, nullptr, row_bounds_out, rows, cols, bits, dtype, sem, workspace, workspace_bytes, stream);
}
int fq_sym_fwd_debug(const void* x, void* y, int32_t* idx_out, float* scale_out, int64_t rows, int64_t cols, int bits, int dtype,
                     int sem, void* workspace, size_t workspace_bytes, void* stream) {
    return rowwise<false>(x, y, idx_out, scale_out, nullptr, rows, cols, bits, dtype, sem, workspace, workspace_bytes, stream);
}
int fq_asym_fwd_debug(const void* x, void* y, int32_t* idx_out, float* scale_out, int64_t rows, int64_t cols, int bits, int dtype,
                      int sem, void* workspace, size_t workspace_bytes, void* stream) {
    return rowwise<true>(x, y, idx_out, scale_out, nullptr, rows, cols, bits, dtype, sem, workspace, workspace_bytes, stream);
}

int fq_ste_bwd(const void* g, const void* x, void* gx, int64_t n, float lo, float hi, int dtype, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (n < 0) return fail(FQ_ERR_SHAPE, "negative n");
    if (n == 0) return ok();
    if (!g || !x || !gx) return fail(FQ_ERR_NULL, "g / x / gx must not be NULL");
    lo = host_rb(lo, dtype);  // the reference compares in the tensor dtype (utils_quant.py:85-86)
    hi = host_rb(hi, dtype);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FQ_DTYPE_F32: return ste_t<F32>(g, x, gx, n, lo, hi, st);
        case FQ_DTYPE_F16: return ste_t<F16>(g, x, gx, n, lo, hi, st);
        default: return ste_t<BF16>(g, x, gx, n, lo, hi, st);
    }
}

int fq_ste_bwd_rows(const void* g, const void* x, void* gx, int64_t rows, int64_t cols, float lo, float hi, const float* row_bounds,
                    int dtype, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(FQ_ERR_DTYPE, "unknown dtype code %d", dtype);
    if (rows < 0 || cols < 0) return fail(FQ_ERR_SHAPE, "negative shape");
    if (rows == 0 || cols == 0) return ok();
    if (!g || !x || !gx || !row_bounds) return fail(FQ_ERR_NULL, "g / x / gx / row_bounds must not be NULL");
    lo = host_rb(lo, dtype);
    hi = host_rb(hi, dtype);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FQ_DTYPE_F32: return ste_rows_t<F32>(g, x, gx, rows, cols, lo, hi, row_bounds, st);
        case FQ_DTYPE_F16: return ste_rows_t<F16>(g, x, gx, rows, cols, lo, hi, row_bounds, st);
        default: return ste_rows_t<BF16>(g, x, gx, rows, cols, lo, hi, row_bounds, st);
    }
}

}  // extern "C"
