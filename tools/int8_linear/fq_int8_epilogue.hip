// Epilogue of the integer consumer of the export format (tools/int8_linear/int8_linear.py; SURVEY §8 f4b: "emitting real int8/int4 +
// scales for inference export" -- this is what reads them).  NOT part of the product package.
//
//   out[m, n] = bf16( float(acc[m, n]) * (rx[m] * rw[n]) )      acc = int32 result of the int8 x int8 GEMM over the exported bins,
//                                                               rx = 1 / t2 of activation row m, rw = 1 / t2 of weight row n
//
// (t2 = s + 1e-6 is the divisor of the reference's `.div(s + 1e-6)`, models/utils_quant.py:72: bin / t2 is the fake-quant value before
// its rounding to the tensor dtype.)  HBM-bound: reads 4 B, writes 2 B per element, one pass; 16-byte loads, 8-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FQ_EXPORT extern "C" __attribute__((visibility("default")))

static __device__ __forceinline__ uint32_t bf16_rne(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// one workgroup row-slice: blockIdx.y = m, each thread 4 consecutive n
__global__ __launch_bounds__(256) void int8_epilogue_kernel(const int32_t* __restrict__ acc, const float* __restrict__ sx, const float* __restrict__ sw,
                                                            uint16_t* __restrict__ out, int64_t n, int sx_stride, int sw_stride) {
    const int64_t m = blockIdx.y;
    const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= n) return;
    const float rx = 1.0f / sx[m * sx_stride + 1];      // scales rows are {s, t2}: t2 at [1]
    const int4 a = *reinterpret_cast<const int4*>(acc + m * n + c);
    const float f0 = rx * (1.0f / sw[(c + 0) * sw_stride + 1]);
    const float f1 = rx * (1.0f / sw[(c + 1) * sw_stride + 1]);
    const float f2 = rx * (1.0f / sw[(c + 2) * sw_stride + 1]);
    const float f3 = rx * (1.0f / sw[(c + 3) * sw_stride + 1]);
    uint2 o;
    o.x = bf16_rne((float)a.x * f0) | (bf16_rne((float)a.y * f1) << 16);
    o.y = bf16_rne((float)a.z * f2) | (bf16_rne((float)a.w * f3) << 16);
    *reinterpret_cast<uint2*>(out + m * n + c) = o;
}

// acc [m, n] int32, sx [m, 2] / sw [n, 2] float32 {s, t2} rows as fq_sym_export writes them, out [m, n] bf16.  n % 4 == 0.
FQ_EXPORT int fq_int8_epilogue(const void* acc, const void* sx, const void* sw, void* out, int64_t m, int64_t n, void* stream) {
    if (!acc || !sx || !sw || !out || m <= 0 || n <= 0 || (n & 3) || m > 65535) return -1;
    dim3 grid((unsigned)((n / 4 + 255) / 256), (unsigned)m);
    const int32_t* a = (const int32_t*)acc;
    const float *x = (const float*)sx, *w = (const float*)sw;
    uint16_t* o = (uint16_t*)out;
    int stride = 2;
    void* args[] = {&a, &x, &w, &o, &n, &stride, &stride};
    // the launch's own return value: the thread's sticky error slot is neither read nor cleared (as in the product library)
    return hipLaunchKernel((const void*)int8_epilogue_kernel, grid, dim3(256), args, 0, (hipStream_t)stream) == hipSuccess ? 0 : -2;
}
