// fq_launch.h -- host-side launch layer shared by the per-dtype translation units.
// The kernels are instantiated per element type in fq_f32.hip / fq_bf16.hip / fq_f16.hip so the
// library builds in parallel; fq_api.hip holds the extern "C" entry points.
#pragma once
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <utility>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/llmqat_fakequant.h"
#include "fq_kernels.h"

namespace fq {

#define FQ_HIDDEN __attribute__((visibility("hidden")))

// error plumbing (defined in fq_api.hip): set the thread-local message, return the code
FQ_HIDDEN int fail(int code, const char* fmt, ...);
FQ_HIDDEN int ok();

// Launch status.  hipLaunchKernelGGL returns nothing, and hipGetLastError() reports -- and clears -- the calling thread's LAST error,
// whoever caused it: rounds 1-3 drained that slot before every launch (hiding other libraries' failures) and read it afterwards.
// Since round 4 the library never touches the slot: every kernel goes through hipLaunchKernel(), whose RETURN VALUE is the status of
// that launch alone; the first failure of an entry point's launches is kept (per thread) and reported by launch_result().  A HIP
// error that an earlier launch or another library left pending is neither cleared nor mistaken for ours.
FQ_HIDDEN hipError_t& launch_status();   // thread-local, defined in fq_api.hip
inline void begin_launches() { launch_status() = hipSuccess; }
template <typename... P, size_t... I>
inline hipError_t launch_with(void (*kernel)(P...), dim3 grid, dim3 block, hipStream_t st, std::tuple<P...>& params, std::index_sequence<I...>) {
    void* args[] = {(void*)&std::get<I>(params)...};
    return hipLaunchKernel((const void*)kernel, grid, block, args, 0, st);
}
// arguments are converted to the kernel's own parameter types first (hipLaunchKernel takes them by address)
template <typename... P, typename... A> inline void launch(void (*kernel)(P...), dim3 grid, dim3 block, hipStream_t st, A&&... a) {
    static_assert(sizeof...(P) == sizeof...(A), "argument count does not match the kernel's parameter list");
    std::tuple<P...> params{static_cast<P>(std::forward<A>(a))...};
    const hipError_t e = launch_with(kernel, grid, block, st, params, std::index_sequence_for<P...>{});
    if (e != hipSuccess && launch_status() == hipSuccess) launch_status() = e;
}
inline int launch_result() {
    const hipError_t e = launch_status();
    return e == hipSuccess ? ok() : fail(FQ_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
}
// (drop-in for the hipLaunchKernelGGL call shape; the kernels use no dynamic LDS)
#define FQ_LAUNCHK(kern, grid, block, shmem, st, ...) ::fq::launch(kern, grid, block, st, __VA_ARGS__)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// pitched rows (RowPitch): every row start must keep the alignment the vector kernels assume
inline bool pitch_aligned(const RowPitch& p, int64_t mask) { return !p.on || (((p.outer | p.inner) & mask) == 0); }
inline bool pitches_aligned(const RowArgs& a, int64_t xmask, int64_t ymask) {
    bool ok = pitch_aligned(a.xp, xmask) && pitch_aligned(a.yp, ymask);
    for (int i = 0; i < a.n_more; ++i) ok = ok && pitch_aligned(a.more[i].xp, xmask) && pitch_aligned(a.more[i].yp, ymask);
    return ok;
}
inline bool any_pitch(const RowArgs& a) {
    bool on = a.xp.on || a.yp.on;
    for (int i = 0; i < a.n_more; ++i) on = on || a.more[i].xp.on || a.more[i].yp.on;
    return on;
}
// the further tensors of a multi-tensor launch: alignment of their x / y, and the rows of the largest tensor (its size
// decides the cache policy of the launch)
inline bool more_aligned(const RowArgs& a, uintptr_t xmask, uintptr_t ymask) {
    for (int i = 0; i < a.n_more; ++i)
        if ((reinterpret_cast<uintptr_t>(a.more[i].x) & xmask) || (reinterpret_cast<uintptr_t>(a.more[i].y) & ymask)) return false;
    return true;
}
// the kernels compare a row against every slot's first row (pick_tensor): unused slots must never match
inline void seal_slots(RowArgs& a) {
    for (int i = a.n_more; i < MAX_MORE; ++i) {
        a.more[i] = TensorSlot{};
        a.more[i].row_begin = INT64_MAX;
    }
}
inline bool any_mask(const RowArgs& a) {
    bool m = a.mask != nullptr;
    for (int i = 0; i < a.n_more; ++i) m = m || a.more[i].mask;
    return m;
}
inline int64_t largest_rows(const RowArgs& a) {
    if (!a.n_more) return a.rows;
    int64_t big = a.rows0;
    for (int i = 0; i < a.n_more; ++i) {
        const int64_t end = i + 1 < a.n_more ? a.more[i + 1].row_begin : a.rows;
        if (end - a.more[i].row_begin > big) big = end - a.more[i].row_begin;
    }
    return big;
}
inline int64_t largest_rows(const SteLaunch& L) {
    int64_t big = 0;
    for (int i = 0; i < L.n; ++i)
        if (L.t[i].rows > big) big = L.t[i].rows;
    return big;
}

// Cache policy of the 16-byte streams, from tools/kbench on MI355X (every tensor is touched once per launch):
//  * stores: always non-temporal.  NT stores win at every size measured from 4 MiB up (16 MB forward 7.3 -> 5.9 us, 45 MB 17.9 -> 14.2 us,
//    90 MB 34.3 -> 31.9 us); below that a launch is latency-bound either way, so since round 4 there is no cacheable-store flavour
//    of the kernels (a third fewer instantiations).
//  * loads: non-temporal only from 72 MiB up.  Below that the input was typically produced by the kernel just
//    before and still sits in the 256 MiB Infinity Cache, where plain loads are faster (16 MB: 5.9 vs 6.1 us,
//    45 MB: 14.2 vs 16.1 us, 67 MB: 20.4 vs 22.2 us); the big MLP weights come from HBM, where NT loads are faster
//    (84 MB: 27.8 vs 28.8 us, 90 MB: 30.3 vs 31.3 us, 113 MB: 37.0 vs 38.0 us).
// Tuning override (read once per process, MiB): LLMQAT_FQ_NT_LOAD_MIN_MB.
inline int64_t nt_env_mib(const char* name, int64_t dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt << 20;
    const long long mb = atoll(v);
    return (mb < 0 ? dflt : (int64_t)mb) << 20;
}
inline int64_t nt_load_min_bytes() {
    static const int64_t v = nt_env_mib("LLMQAT_FQ_NT_LOAD_MIN_MB", 72);
    return v;
}
#define NT_LOAD_MIN_BYTES (::fq::nt_load_min_bytes())

constexpr int64_t REG_MAX_VEC = 1024 * 8;    // longest row (in 16-byte vectors) the register kernels hold
constexpr int64_t GENERIC_MAX_COLS = 32768;  // longest row the scalar-load kernel sweeps
constexpr int64_t WS_COLS_THRESHOLD = 32768; // rows longer than this may take the two-pass path

// ---- host-side round-to-dtype for the launch-uniform scalars (1e-6, 1e-8, clip bounds) ----
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline float host_rb(float v, int dt) {
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
inline Consts make_consts(int bits, int dt, int sem) {
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

inline int esize_of(int dtype) { return dtype == FQ_DTYPE_F64 ? 8 : dtype == FQ_DTYPE_F32 ? 4 : 2; }

// float64 tensors (fq_f64.hip): a correctness path in double arithmetic
FQ_HIDDEN int launch_f64_rowwise(bool asym, const void* x, void* y, int32_t* idx, float* scale, int64_t rows, int64_t cols, int bits, int sem, hipStream_t st);
FQ_HIDDEN int launch_f64_ste(const void* g, const void* x, void* gx, int64_t n, float lo, float hi, hipStream_t st);
FQ_HIDDEN int launch_f64_w12(const void* w, const void* scale, void* out, int64_t rows, int64_t cols, int w_bits, int scale_per_row, hipStream_t st);


template <int DT> FQ_HIDDEN int launch_rowwise(bool asym, bool fast, RowArgs a, void* ws, size_t wsb, hipStream_t st);
// Sym under CUDA-autocast arithmetic (16-bit tensors only): wide = fp32 output, else rounded once to the tensor dtype
template <int DT> FQ_HIDDEN int launch_sym_autocast(bool wide, RowArgs a, void* ws, size_t wsb, hipStream_t st);
template <int DT> FQ_HIDDEN int launch_ste(const void* g, const void* x, void* gx, int64_t n, float lo, float hi, hipStream_t st);
// STE backward from (bounds, mask) for the L.n tensors of one launch (g / gx / bounds / mask / rows filled in by the caller;
// blk_begin / inplace are set here)
template <int DT> FQ_HIDDEN int launch_ste_mask(SteLaunch L, int64_t cols, float lo, float hi, hipStream_t st);
// the same behind a fp32-result forward: the g are fp32, the gx have the (16-bit) dtype DT
template <int DT> FQ_HIDDEN int launch_ste_mask_wide(SteLaunch L, int64_t cols, float lo, float hi, hipStream_t st);
template <int DT>
FQ_HIDDEN int launch_w12(const void* w, const void* scale, void* out, int64_t rows, int64_t cols, int w_bits, int scale_per_row,
                         float cv, hipStream_t st);
template <int DT>
FQ_HIDDEN int launch_w12_rows(const void* w, void* out, void* scale_out, int64_t rows, int64_t cols, int w_bits, float cv, hipStream_t st);
// words (uint64) of STE mask per row -- a plain bitmap, one bit per element, rows padded to 8 bytes; 0 if the shape is not
// served by the mask path (the recording kernels are the register-resident ones: whole 16-byte vectors, rows that fit)
inline int64_t mask_row_words(int64_t cols, int esize) {
    const int epv = 16 / esize;
    if (cols <= 0 || cols % epv) return 0;
    const int64_t nvec = cols / epv;
    if (nvec > REG_MAX_VEC) return 0;
    return (cols + 63) / 64;
}
// bits of an fp16-representable float as IEEE half (host side; v has been through host_rb)
inline uint32_t half_bits(float v) {
    const uint32_t x = f2u(v), sign = (x >> 16) & 0x8000u, a = x & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return sign | 0x7E00u;
    if (a == 0x7F800000u) return sign | 0x7C00u;
    if (a < 0x38800000u) return sign | (uint32_t)__builtin_rintf(u2f(a) / 5.9604644775390625e-08f);  // subnormal: multiples of 2^-24
    return sign | (((a >> 23) - 112u) << 10) | ((a >> 13) & 0x3FFu);
}
// Integer form of the STE predicate for 16-bit tensors (fq_kernels.h ste_flags16_*): applicable when lo == -hi and hi >= 0
// (|x| >= hi as an unsigned compare of the magnitude bits); lo / hi already rounded to the dtype.  0 = compare as floats.
inline uint32_t ste_clip_key(float lo, float hi, int dtype) {
    if (dtype == FQ_DTYPE_F32 || !(lo == -hi) || !(hi >= 0.0f)) return 0;
    const uint32_t t = dtype == FQ_DTYPE_BF16 ? (f2u(hi) >> 16) & 0x7FFFu : half_bits(hi) & 0x7FFFu;  // <= 0x7F80 / 0x7C00 (+inf)
    const uint32_t k = 0x8000u - t;
    return k | (k << 16);
}
template <int DT>
FQ_HIDDEN int launch_ste_rows(const void* g, const void* x, void* gx, int64_t rows, int64_t cols, float lo, float hi,
                              const float* bounds, hipStream_t st, const StePitch3& pitch = StePitch3{});

}  // namespace fq
