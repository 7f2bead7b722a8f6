// fq_f64.hip -- float64 tensors.  The reference has no dtype restriction (models/utils_quant.py runs on whatever it is given);
// no BASELINE config uses float64, so this is a CORRECTNESS path, not a tuned one: one workgroup per row, two sweeps with
// element loads (the second is cache-hot), every op in double exactly as ATen computes it on a float64 tensor -- the recipes of
// fq_device.h without the round-to-dtype steps.  Served: SymQuantizer / AsymQuantizer forward (utils_quant.py:37-74, :96-149),
// their STE backward re-reading x (:77-87, :152-162) and the elementwise part of the 1-/2-bit weight branches (:202-242).
// Not served for float64 (FQ_ERR_DTYPE): the training-mode side buffers (bounds / STE mask), multi-tensor launches, export, the
// fused GEMM -- the Python shim uses the reference's data flow (saved input) for float64 tensors.
#include "../../include/llmqat_fakequant.h"
#include "fq_launch.h"

namespace fq {

constexpr int F64_THREADS = 256;

__device__ __forceinline__ double block_max_f64(double v, bool& nan, double* lds, int* lds_nan) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v = fmax(v, __shfl_xor(v, o, 64));   // fmax ignores NaN; NaN is carried by the flag
        const int other = __shfl_xor((int)nan, o, 64);   // (unconditionally: `nan || shfl(...)` would short-circuit the cross-lane op away
        nan = nan | (other != 0);                        //  in exactly the lanes whose flag the others need)
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        lds[wave] = v;
        lds_nan[wave] = nan;
    }
    __syncthreads();
    double r = lds[0];
    bool n = lds_nan[0] != 0;
#pragma unroll
    for (int i = 1; i < F64_THREADS / 64; ++i) {
        r = fmax(r, lds[i]);
        n = n || lds_nan[i] != 0;
    }
    nan = n;
    __syncthreads();
    return r;
}

__device__ __forceinline__ int32_t idx_i32_f64(double q) {  // same coding as the oracle / fixtures
    if (q != q) return INT32_MIN;
    if (q >= 2.0e9) return isinf(q) ? INT32_MAX : 2000000000;
    if (q <= -2.0e9) return isinf(q) ? -INT32_MAX : -2000000000;
    return (int32_t)q;
}

// utils_quant.py:50-72 (Sym) / :110-147 (Asym)
template <bool ASYM>
__global__ __launch_bounds__(F64_THREADS) void row_f64_kernel(const double* __restrict__ x, double* __restrict__ y, int32_t* __restrict__ idx,
                                                              float* __restrict__ scale, int64_t cols, double qs, int mul_inv) {
    __shared__ double lds[F64_THREADS / 64];
    __shared__ int lds_nan[F64_THREADS / 64];
    const int64_t row = blockIdx.x, base = row * cols;
    const int t = threadIdx.x;
    if constexpr (!ASYM) {
        double m = 0.0;
        bool nan = false;
        for (int64_t c = t; c < cols; c += F64_THREADS) {
            const double v = x[base + c];
            nan = nan || v != v;
            m = fmax(m, fabs(v));
        }
        m = block_max_f64(m, nan, lds, lds_nan);
        if (nan) m = __longlong_as_double(0x7FF8000000000000ll);   // torch.max propagates NaN
        const double s = (1.0 / (m + 1e-6)) * qs;                   // :71  `int / Tensor` = reciprocal() * int
        const double t2 = s + 1e-6;
        if (t == 0 && scale) scale[row] = (float)s;
        for (int64_t c = t; c < cols; c += F64_THREADS) {
            const double q = rint(x[base + c] * s);                 // :72  torch.round: half to even
            if (idx) idx[base + c] = idx_i32_f64(q);
            y[base + c] = q / t2;
        }
    } else {
        double mx = -__longlong_as_double(0x7FF0000000000000ll), nmn = mx;  // running max of x and of -x
        bool nan = false;
        for (int64_t c = t; c < cols; c += F64_THREADS) {
            const double v = x[base + c];
            nan = nan || v != v;
            mx = fmax(mx, v);
            nmn = fmax(nmn, -v);
        }
        bool nan2 = nan;
        mx = block_max_f64(mx, nan, lds, lds_nan);
        nmn = block_max_f64(nmn, nan2, lds, lds_nan);
        double mn = -nmn;
        if (nan) mx = mn = __longlong_as_double(0x7FF8000000000000ll);
        const double al = mx - mn, a = al + 1e-8, invS = 1.0 / qs;
        if (t == 0 && scale) {
            scale[2 * row] = (float)al;
            scale[2 * row + 1] = (float)mn;
        }
        for (int64_t c = t; c < cols; c += F64_THREADS) {
            const double n = (x[base + c] - mn) / a;                // :144
            const double q = rint(n * qs);                          // :146
            if (idx) idx[base + c] = idx_i32_f64(q);
            const double w = mul_inv ? q * invS : q / qs;           // .div(python int): true division on CPU, * (1/S) in ATen's GPU kernel
            y[base + c] = w * a + mn;                               // :147
        }
    }
}

// utils_quant.py:83-87
__global__ __launch_bounds__(256) void ste_f64_kernel(const double* __restrict__ g, const double* __restrict__ x, double* __restrict__ gx, int64_t n,
                                                      double lo, double hi) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double v = x[i];
        gx[i] = (v >= hi || v <= lo) ? 0.0 : g[i];
    }
}

// utils_quant.py:203-242, elementwise part (forward value of the detach trick)
template <int WBITS>
__global__ __launch_bounds__(256) void w12_f64_kernel(const double* __restrict__ w, const double* __restrict__ scale, double* __restrict__ out, int64_t rows,
                                                      int64_t cols, int scale_per_row) {
    const int64_t n = rows * cols, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double sc = scale[scale_per_row ? i / cols : 0], wv = w[i];
        const double t = wv / sc;
        double q;
        if constexpr (WBITS == 1) {
            q = sc * ((t > 0.0) ? 1.0 : (t < 0.0) ? -1.0 : 0.0);
        } else {
            const double cv = 1.0 - 1e-2;
            const double c = (t != t) ? t : fmin(fmax(t, -cv), cv);
            q = sc * (rint(c * 2.0 - 0.5) + 0.5) / 2.0;
        }
        out[i] = (q - wv) + wv;
    }
}

int launch_f64_rowwise(bool asym, const void* x, void* y, int32_t* idx, float* scale, int64_t rows, int64_t cols, int bits, int sem, hipStream_t st) {
    begin_launches();
    if (rows > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows=%lld exceeds the grid limit", (long long)rows);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) return fail(FQ_ERR_UNSUPPORTED, "float64 tensors must be 8-byte aligned");
    if (asym) {
        const double S = (double)((1ull << bits) - 1ull);
        FQ_LAUNCHK(row_f64_kernel<true>, dim3((unsigned)rows), dim3(F64_THREADS), 0, st, (const double*)x, (double*)y, idx, scale, cols, S,
                           sem == FQ_SEM_DEVICE_EAGER ? 1 : 0);
    } else {
        const double qmax = (double)((1u << (bits - 1)) - 1u);
        FQ_LAUNCHK(row_f64_kernel<false>, dim3((unsigned)rows), dim3(F64_THREADS), 0, st, (const double*)x, (double*)y, idx, scale, cols, qmax, 0);
    }
    return launch_result();
}

int launch_f64_ste(const void* g, const void* x, void* gx, int64_t n, float lo, float hi, hipStream_t st) {
    begin_launches();
    int64_t grid = (n + 255) / 256;
    if (grid > 16384) grid = 16384;
    FQ_LAUNCHK(ste_f64_kernel, dim3((unsigned)grid), dim3(256), 0, st, (const double*)g, (const double*)x, (double*)gx, n, (double)lo, (double)hi);
    return launch_result();
}

int launch_f64_w12(const void* w, const void* scale, void* out, int64_t rows, int64_t cols, int w_bits, int scale_per_row, hipStream_t st) {
    begin_launches();
    int64_t grid = (rows * cols + 255) / 256;
    if (grid > 16384) grid = 16384;
    if (w_bits == 1) FQ_LAUNCHK(w12_f64_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, (const double*)w, (const double*)scale, (double*)out, rows, cols, scale_per_row);
    else FQ_LAUNCHK(w12_f64_kernel<2>, dim3((unsigned)grid), dim3(256), 0, st, (const double*)w, (const double*)scale, (double*)out, rows, cols, scale_per_row);
    return launch_result();
}

}  // namespace fq
