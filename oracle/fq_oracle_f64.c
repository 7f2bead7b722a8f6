/*
 * fq_oracle_f64.c -- CPU restatement of LLM-QAT's fake-quantization arithmetic for FLOAT64 tensors.
 *
 * TEST INFRASTRUCTURE ONLY (as fq_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch
 * anything under oracle/).
 *
 * Parity status: PINNED by tests/golden/f64.npz -- 50 cases produced by importing the real reference
 * (models/utils_quant.py) on CPU with float64 tensors (tests/golden/make_golden_f64.py, torch 2.10.0).
 *
 * The reference has no dtype restriction; on a float64 tensor every ATen op computes in double, once rounded, so the recipes
 * are the ones of fq_oracle.c without the round-to-dtype steps.  Python scalars (1e-6, 1e-8, 0.99, 2**bits - 1) are doubles.
 * Reference lines (all /root/reference/models/utils_quant.py):
 *   SymQuantizer.forward :37-74 (s = (1 / (max|x| + 1e-6)) * qmax :71 -- `int / Tensor` is reciprocal() * int --, round / div :72)
 *   AsymQuantizer.forward :96-149;  backward of both :77-87, :152-162 (the clip is a float32 tensor :198,:245: its values as doubles)
 *   QuantizeLinear W1 / W2 :202-242
 * `sem` as in fq_oracle.c: 0 = CPU eager (`.div(python int)` is a true division), 1 = device eager (ATen's GPU kernel multiplies by
 * the reciprocal of a scalar divisor, in double).  Only AsymQuantizer's `.div(s)` :146 is affected.
 */
#include <math.h>
#include <stdint.h>

static inline double nanmax64(double a, double b) { return (a != a) ? a : (b != b) ? b : (a > b ? a : b); }
static inline double nanmin64(double a, double b) { return (a != a) ? a : (b != b) ? b : (a < b ? a : b); }
static inline int32_t idx64_i32(double r) { /* same coding as make_golden.py: NaN -> INT32_MIN, +-Inf -> +-INT32_MAX, clamp at 2e9 */
    if (r != r) return INT32_MIN;
    if (r >= 2.0e9) return isinf(r) ? INT32_MAX : 2000000000;
    if (r <= -2.0e9) return isinf(r) ? -INT32_MAX : -2000000000;
    return (int32_t)r;
}

/* utils_quant.py:50-72 */
int fqo64_sym_fwd(const double* x, double* y, int32_t* idx, double* scale, int64_t rows, int64_t cols, int bits) {
    if (bits < 1 || bits > 31) return -2;
    const double qmax = (double)((1u << (bits - 1)) - 1u);
    for (int64_t r = 0; r < rows; ++r) {
        const double* xr = x + r * cols;
        double m = fabs(xr[0]);
        for (int64_t c = 1; c < cols; ++c) m = nanmax64(m, fabs(xr[c]));
        const double s = (1.0 / (m + 1e-6)) * qmax;     /* :71 */
        const double t2 = s + 1e-6;
        if (scale) scale[r] = s;
        for (int64_t c = 0; c < cols; ++c) {
            const double q = nearbyint(xr[c] * s);       /* :72 torch.round: half to even */
            if (idx) idx[r * cols + c] = idx64_i32(q);
            y[r * cols + c] = q / t2;
        }
    }
    return 0;
}

/* utils_quant.py:110-147 */
int fqo64_asym_fwd(const double* x, double* y, int32_t* idx, double* alpha, double* beta, int64_t rows, int64_t cols, int bits, int sem) {
    if (bits < 1 || bits > 31) return -2;
    const double S = (double)((1ull << bits) - 1ull), invS = 1.0 / S;
    for (int64_t r = 0; r < rows; ++r) {
        const double* xr = x + r * cols;
        double mx = xr[0], mn = xr[0];
        for (int64_t c = 1; c < cols; ++c) {
            mx = nanmax64(mx, xr[c]);
            mn = nanmin64(mn, xr[c]);
        }
        if (mx != mx || mn != mn) mx = mn = NAN;
        const double al = mx - mn, a = al + 1e-8;
        if (alpha) alpha[r] = al;
        if (beta) beta[r] = mn;
        for (int64_t c = 0; c < cols; ++c) {
            const double n = (xr[c] - mn) / a;          /* :144 */
            const double q = nearbyint(n * S);           /* :146 */
            if (idx) idx[r * cols + c] = idx64_i32(q);
            const double w = sem ? q * invS : q / S;
            y[r * cols + c] = w * a + mn;                /* :147 */
        }
    }
    return 0;
}

/* utils_quant.py:83-87; lo / hi are float32 values (the clip tensor's dtype) compared as doubles */
int fqo64_ste_bwd(const double* g, const double* x, double* gx, int64_t n, float lo, float hi) {
    for (int64_t i = 0; i < n; ++i) gx[i] = (x[i] >= (double)hi || x[i] <= (double)lo) ? 0.0 : g[i];
    return 0;
}

/* utils_quant.py:202-242, forward value of `q.detach() - w.detach() + w`; scale_in = the mean-|w| scaling factor (rows values, or one
 * when rows == 1), as the reference's own torch.mean produced it; NULL: this file's sequential double sum (order dependent: fallback) */
int fqo64_w12_fwd(const double* w, double* q, double* scale_out, const double* scale_in, int64_t rows, int64_t cols, int w_bits) {
    if (w_bits != 1 && w_bits != 2) return -2;
    for (int64_t r = 0; r < rows; ++r) {
        const double* wr = w + r * cols;
        double sc;
        if (scale_in) sc = scale_in[r];
        else {
            double acc = 0.0;
            for (int64_t c = 0; c < cols; ++c) acc += fabs(wr[c]);
            sc = acc / (double)cols;
            if (w_bits == 2) sc = 2.0 * sc;
        }
        if (scale_out) scale_out[r] = sc;
        for (int64_t c = 0; c < cols; ++c) {
            const double t = wr[c] / sc;
            double v;
            if (w_bits == 1) {
                v = sc * ((t > 0.0) ? 1.0 : (t < 0.0) ? -1.0 : 0.0);   /* torch.sign(NaN) = 0 */
            } else {
                const double cv = 1.0 - 1e-2;
                const double cl = (t != t) ? t : (t < -cv ? -cv : (t > cv ? cv : t));   /* torch.clamp propagates NaN */
                v = sc * (nearbyint(cl * 2.0 - 0.5) + 0.5) / 2.0;
            }
            q[r * cols + c] = (v - wr[c]) + wr[c];
        }
    }
    return 0;
}
