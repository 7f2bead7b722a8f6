// fq_launch.h -- host-side launch layer shared by the per-dtype translation units.
// The kernels are instantiated per element type in fq_f32.hip / fq_bf16.hip / fq_f16.hip so the
// library builds in parallel; fq_api.hip holds the extern "C" entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "fq_kernels.h"

namespace fq {

#define FQ_HIDDEN __attribute__((visibility("hidden")))

// error plumbing (defined in fq_api.hip): set the thread-local message, return the code
FQ_HIDDEN int fail(int code, const char* fmt, ...);
FQ_HIDDEN int ok();

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Tensors at least this large cannot stay in the 32 MiB of L2 anyway: stream them with the
// non-temporal policy (+7..9 % on the 90 MB metric tensor).  Smaller ones keep the default
// policy so the consumer (the GEMM that follows) can still find them in L2 / Infinity Cache.
constexpr int64_t NT_MIN_BYTES = 32ll << 20;

constexpr int64_t REG_MAX_VEC = 1024 * 8;    // longest row (in 16-byte vectors) the register kernels hold
constexpr int64_t GENERIC_MAX_COLS = 32768;  // longest row the scalar-load kernel sweeps
constexpr int64_t WS_COLS_THRESHOLD = 32768; // rows longer than this may take the two-pass path

template <int DT> FQ_HIDDEN int launch_rowwise(bool asym, bool fast, RowArgs a, void* ws, size_t wsb, hipStream_t st);
template <int DT> FQ_HIDDEN int launch_ste(const void* g, const void* x, void* gx, int64_t n, float lo, float hi, hipStream_t st);
template <int DT>
FQ_HIDDEN int launch_ste_mask(const void* g, void* gx, int64_t rows, int64_t cols, float lo, float hi, const float* bounds,
                              const uint64_t* mask, hipStream_t st);
template <int DT>
FQ_HIDDEN int launch_w12(const void* w, const void* scale, void* out, int64_t rows, int64_t cols, int w_bits, int scale_per_row,
                         float cv, hipStream_t st);
// words (uint64) of STE mask per row; 0 if the shape is not served by the mask path
inline int64_t mask_row_words(int64_t cols, int esize) {
    const int epv = 16 / esize;
    if (cols <= 0 || cols % epv) return 0;
    const int64_t nvec = cols / epv;
    if (nvec > REG_MAX_VEC) return 0;
    return (nvec + 63) / 64 * epv;
}
template <int DT>
FQ_HIDDEN int launch_ste_rows(const void* g, const void* x, void* gx, int64_t rows, int64_t cols, float lo, float hi,
                              const float* bounds, hipStream_t st);

}  // namespace fq
