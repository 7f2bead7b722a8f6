// fq_qlinear.hip -- EXPERIMENT, NOT PART OF THE PRODUCT LIBRARY (moved out of llm-qat_amd/csrc in round 4: measured slower than the
// unfused product path -- fake-quant at the HBM roofline + hipBLASLt -- on every LLaMA shape; it is kept here, with
// its tests and its bench, as the measured no-go it is).  Built by tools/qlinear/qlinear.py into tools/qlinear/libfq_qlinear_exp.so.
//
// QuantizeLinear's no-grad forward with the fake-quant applied WHILE THE GEMM LOADS ITS
// OPERANDS (models/utils_quant.py:250  `F.linear(input_, weight)`  fed by :195-201 and :244-248).
//
//     out[tokens, out] = fq(x)[tokens, in] . fq(W)[out, in]^T          bf16 in, fp32 accumulate (MFMA), bf16 out
//
// With the per-row scale terms {s, t2} known (fq_sym_row_scales: one read of the tensor), `round(x * s) / t2` is a pure
// elementwise map, so the quantized operand never has to exist in HBM: the global -> LDS staging of each tile runs the same
// `sym_dword` chain as fq_sym_fwd (bit-identical values, checked by dumping the staged tiles) between its
// global_load_dwordx4 and its ds_write_b128.  LDS-DMA cannot be used for a quantized operand (the data must pass through
// VGPRs); an operand that arrives already quantized (the activation a sibling projection shares) is staged unchanged.
//
// Kernel: 256 (tokens) x 128 (out features) x 64 (K) tiles, 512 threads = 8 waves as 2 (out) x 4 (tokens), each wave
// 2 x 2 tiles of v_mfma_f32_32x32x16_bf16 with W as the A operand and x as the B operand (so the accumulator holds, per
// lane, 4 consecutive OUTPUT columns of one token row: 8-byte stores).  LDS: two stages x (16 KiB W + 32 KiB x), rows of
// 128 B with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 -- conflict-free for ds_read_b128's 16-lane groups
// and for the row-contiguous ds_write_b128.  One barrier per K-step; two register sets keep the global loads of tiles t+1
// and t+2 in flight while tile t is multiplied, then tile t+1 is quantized into the other LDS stage.
// Grid: one block per tile, tile index remapped so that the blocks of one XCD (blockIdx % 8) cover a compact
// 8 x 4 patch of tiles (they share W / x panels in that XCD's L2).  Speed only; any placement is correct.
#include "../../include/llmqat_fakequant.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../llm-qat_amd/csrc/fq_launch.h"

using namespace fq;

// the product's error plumbing lives in fq_api.hip; this stand-alone library carries its own copy of the two functions
namespace {
thread_local char g_qerr[320] = "";
}
namespace fq {
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_qerr, sizeof(g_qerr), fmt, ap);
    va_end(ap);
    return code;
}
int ok() {
    g_qerr[0] = 0;
    return FQ_OK;
}
hipError_t& launch_status() {
    static thread_local hipError_t st = hipSuccess;
    return st;
}
}  // namespace fq

namespace {

constexpr int QL_BM = 256;   // token rows per block (x tile)
constexpr int QL_BN = 128;   // output features per block (W tile)
constexpr int QL_BK = 64;    // K per step: 128-byte rows in LDS
constexpr int QL_THREADS = 512;
constexpr int QL_W_BYTES = QL_BN * QL_BK * 2;          // 16 KiB
constexpr int QL_X_BYTES = QL_BM * QL_BK * 2;          // 32 KiB
constexpr int QL_STAGE = QL_W_BYTES + QL_X_BYTES;      // 48 KiB

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

struct QLArgs {
    const uint16_t* x;     // [M][K]
    const uint16_t* w;     // [N][K]
    uint16_t* out;         // [M][N]
    const float* xs;       // [M][2] = {s, t2} (QA != 0)
    const float* ws;       // [N][2]           (QW != 0)
    int M, N, K;
    uint16_t* dump_x;      // optional: the x tile values as staged (written by the blocks of the first W panel)
    uint16_t* dump_w;      // optional: the W tile values as staged (written by the blocks of the first x panel)
    int tiles_m, tiles_n;
};

struct RowQ {
    float s, t2, rinv;
};

// Q: 0 = stage unchanged, 1 = SymQuantizer arithmetic in bf16 (fq_sym_fwd), 2 = its autocast arithmetic (fq_sym_fwd_autocast,
// result rounded once to bf16)
template <int Q> __device__ __forceinline__ uint4 quant_vec(uint4 v, const RowQ& q) {
    if constexpr (Q == 0) {
        return v;
    } else if constexpr (Q == 1) {
        SymRow r;
        r.s = q.s, r.t2 = q.t2, r.rinv = q.rinv, r.mk = true;
        return make_uint4(sym_dword<BF16, true>(v.x, r, nullptr), sym_dword<BF16, true>(v.y, r, nullptr), sym_dword<BF16, true>(v.z, r, nullptr),
                          sym_dword<BF16, true>(v.w, r, nullptr));
    } else {
        SymRow r;
        r.s = q.s, r.t2 = q.t2, r.rinv = q.rinv, r.mk = true;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float f[2];
            Ty<BF16>::unpack(w[d], f);
            f[0] = sym_elem_autocast(f[0], r);
            f[1] = sym_elem_autocast(f[1], r);
            o[d] = Ty<BF16>::pack(f);
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4); }

// ABL (ablation, timing builds only -- results are garbage): 0 = the kernel; 1 = no MFMA (staging pipeline alone);
// 2 = no staging (LDS reads + MFMA alone, tile 0 re-used)
template <int QA, int QW, bool DUMP, int ABL>
__global__ __launch_bounds__(QL_THREADS, 2) void qlinear_kernel(QLArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * QL_STAGE];
    const int t = threadIdx.x;
    // tile of this block: XCD-aware (blocks b, b+8, ... share an XCD: give them consecutive tiles, m fastest)
    const int ntiles = a.tiles_m * a.tiles_n;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
    const int tm = tile % a.tiles_m, tn = tile / a.tiles_m;
    const int m0 = tm * QL_BM, n0 = tn * QL_BN;
    const int K = a.K;

    // ---- staging map: thread t owns 16-byte chunk (t & 7) of rows (t >> 3) + 64 i
    const int chunk = t & 7, srow = t >> 3;
    const uint16_t* wp[2];
    const uint16_t* xp[4];
    RowQ wq[2], xq[4];
    int wr[2], xr[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = n0 + srow + 64 * i;
        r = r < a.N ? r : a.N - 1;   // clamp: tail rows are computed on a copy of the last row and never stored
        wr[i] = r;
        wp[i] = a.w + (int64_t)r * K + chunk * 8;
        if constexpr (QW != 0) {
            wq[i].s = a.ws[2 * r];
            wq[i].t2 = a.ws[2 * r + 1];
            wq[i].rinv = 1.0f / wq[i].t2;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = m0 + srow + 64 * i;
        r = r < a.M ? r : a.M - 1;
        xr[i] = r;
        xp[i] = a.x + (int64_t)r * K + chunk * 8;
        if constexpr (QA != 0) {
            xq[i].s = a.xs[2 * r];
            xq[i].t2 = a.xs[2 * r + 1];
            xq[i].rinv = 1.0f / xq[i].t2;
        }
    }
    struct Regs {
        uint4 w[2], x[4];
    };
    auto gload = [&](Regs& r, int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) r.w[i] = *(const uint4*)(wp[i] + kt * QL_BK);
#pragma unroll
        for (int i = 0; i < 4; ++i) r.x[i] = *(const uint4*)(xp[i] + kt * QL_BK);
    };
    auto stage = [&](int buf, int kt, const Regs& r) {
        const uint4* rw = r.w;
        const uint4* rx = r.x;
        char* sw = smem + buf * QL_STAGE;
        char* sx = sw + QL_W_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint4 q = quant_vec<QW>(rw[i], wq[i]);
            *(uint4*)(sw + lds_off(srow + 64 * i, chunk)) = q;
            if constexpr (DUMP) {
                if (a.dump_w && tm == 0 && n0 + srow + 64 * i < a.N) *(uint4*)(a.dump_w + (int64_t)wr[i] * K + kt * QL_BK + chunk * 8) = q;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 q = quant_vec<QA>(rx[i], xq[i]);
            *(uint4*)(sx + lds_off(srow + 64 * i, chunk)) = q;
            if constexpr (DUMP) {
                if (a.dump_x && tn == 0 && m0 + srow + 64 * i < a.M) *(uint4*)(a.dump_x + (int64_t)xr[i] * K + kt * QL_BK + chunk * 8) = q;
            }
        }
    };

    // ---- MFMA map: wave = (wn, wm); lane l: r = l & 31 (row of the 32-row fragment), h = l >> 5 (which 8 of the 16 k)
    const int lane = t & 63, wave = t >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int buf) {
        const char* sw = smem + buf * QL_STAGE;
        const char* sx = sw + QL_W_BYTES;
#pragma unroll
        for (int ks = 0; ks < QL_BK / 16; ++ks) {
            const int c = ks * 2 + fh;
            uint4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *(const uint4*)(sw + lds_off(wn * 64 + i * 32 + fr, c));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *(const uint4*)(sx + lds_off(wm * 64 + j * 32 + fr, c));
            if constexpr (ABL == 1) {
                asm volatile("" ::"v"(fa[0].x), "v"(fa[1].x), "v"(fb[0].x), "v"(fb[1].x));
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[i]), __builtin_bit_cast(bf16x8_t, fb[j]),
                                                                            acc[i][j], 0, 0, 0);
            }
        }
    };

    const int nk = K / QL_BK;
    Regs ra, rb;   // two register sets: the global loads of tiles t+1 AND t+2 are in flight while tile t is multiplied
    gload(ra, 0);
    stage(0, 0, ra);
    __syncthreads();
    if (nk > 1) gload(ra, 1);
    for (int kt = 0; kt < nk; kt += 2) {
        // even step: LDS stage 0 holds tile kt, `ra` tile kt+1; tile kt+2 starts its flight into `rb`
        if constexpr (ABL != 2) {
            if (kt + 2 < nk) gload(rb, kt + 2);
        }
        compute(0);
        if constexpr (ABL != 2) {
            if (kt + 1 < nk) stage(1, kt + 1, ra);
        }
        __syncthreads();
        if (kt + 1 >= nk) break;
        // odd step: stage 1 holds tile kt+1, `rb` tile kt+2; tile kt+3 starts its flight into `ra`
        if constexpr (ABL != 2) {
            if (kt + 3 < nk) gload(ra, kt + 3);
        }
        compute(ABL == 2 ? 0 : 1);
        if constexpr (ABL != 2) {
            if (kt + 2 < nk) stage(0, kt + 2, rb);
        }
        __syncthreads();
    }

    // ---- epilogue: acc[i][j][reg]: out column n = n0 + wn*64 + i*32 + 8*(reg>>2) + 4*fh + (reg&3), token m = m0 + wm*64 + j*32 + fr
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 64 + j * 32 + fr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * fh;
                const float lo2[2] = {acc[i][j][4 * g], acc[i][j][4 * g + 1]}, hi2[2] = {acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                const uint2 o = make_uint2(Ty<BF16>::pack(lo2), Ty<BF16>::pack(hi2));
                if (m < a.M && n < a.N) *(uint2*)(a.out + (int64_t)m * a.N + n) = o;   // N % 4 == 0 (host-checked)
            }
        }
    }
}

template <int QA, int QW> int launch_ql(const QLArgs& a, int ablation, hipStream_t st) {
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(QL_THREADS);
    const bool dump = a.dump_x || a.dump_w;
    if (ablation == 1) FQ_LAUNCHK((qlinear_kernel<QA, QW, false, 1>), grid, block, 0, st, a);
    else if (ablation == 2) FQ_LAUNCHK((qlinear_kernel<QA, QW, false, 2>), grid, block, 0, st, a);
    else if (dump) FQ_LAUNCHK((qlinear_kernel<QA, QW, true, 0>), grid, block, 0, st, a);
    else FQ_LAUNCHK((qlinear_kernel<QA, QW, false, 0>), grid, block, 0, st, a);
    return launch_result();
}

}  // namespace

#define FQ_API __attribute__((visibility("default")))

extern "C" {

FQ_API const char* fq_qlinear_last_error(void) { return g_qerr; }

/*
 * out[tokens, out] = fq(x)[tokens, in] . fq(W)[out, in]^T   (models/utils_quant.py:195-201, :244-250), no backward.
 *   x_scales / w_scales  per-row {s, t2} of the product's fq_sym_row_scales (float[rows][2]); NULL: that operand is multiplied AS IS
 *   dtype      FQ_DTYPE_BF16 (operands and result);  autocast  1: the staged values follow fq_sym_fwd_autocast (rounded once to bf16)
 *   dump_x / dump_w   optional [tokens, in] / [out, in] bf16 buffers receiving the operand tiles exactly as staged for the MFMAs
 *   ablation   0; 1 / 2 are timing builds of the cost model (1: staging pipeline without MFMAs, 2: LDS reads + MFMAs without
 *              staging) whose result is garbage
 * FQ_ERR_UNSUPPORTED unless in_features % 64 == 0, out_features % 4 == 0, x / w 16-byte and out 8-byte aligned.
 */
FQ_API int fq_qlinear_fwd(const void* x, const float* x_scales, const void* w, const float* w_scales, void* out, int64_t tokens,
                          int64_t in_features, int64_t out_features, int dtype, int autocast, void* dump_x, void* dump_w, int ablation,
                          void* stream) {
    if (dtype != FQ_DTYPE_BF16) return fail(FQ_ERR_DTYPE, "fq_qlinear_fwd serves bf16 operands");
    if (tokens < 0 || in_features < 0 || out_features < 0) return fail(FQ_ERR_SHAPE, "negative shape");
    if (tokens == 0 || out_features == 0) return ok();
    if (!x || !w || !out) return fail(FQ_ERR_NULL, "x / w / out must not be NULL");
    if (in_features == 0 || in_features % QL_BK) return fail(FQ_ERR_UNSUPPORTED, "in_features=%lld must be a positive multiple of %d", (long long)in_features, QL_BK);
    if (out_features % 4) return fail(FQ_ERR_UNSUPPORTED, "out_features=%lld must be a multiple of 4", (long long)out_features);
    if (!aligned16(x) || !aligned16(w) || (reinterpret_cast<uintptr_t>(out) & 7u)) return fail(FQ_ERR_UNSUPPORTED, "x / w must be 16-byte, out 8-byte aligned");
    if (tokens > 0x7FFFFFFF || out_features > 0x7FFFFFFF || in_features > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "dimension exceeds int32");
    if (ablation < 0 || ablation > 2) return fail(FQ_ERR_ARG, "ablation must be 0, 1 or 2");
    QLArgs a{};
    a.x = (const uint16_t*)x;
    a.w = (const uint16_t*)w;
    a.out = (uint16_t*)out;
    a.xs = x_scales;
    a.ws = w_scales;
    a.M = (int)tokens, a.N = (int)out_features, a.K = (int)in_features;
    a.dump_x = (uint16_t*)dump_x;
    a.dump_w = (uint16_t*)dump_w;
    a.tiles_m = (int)((tokens + QL_BM - 1) / QL_BM);
    a.tiles_n = (int)((out_features + QL_BN - 1) / QL_BN);
    if ((int64_t)a.tiles_m * a.tiles_n > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "too many tiles");
    begin_launches();
    hipStream_t st = (hipStream_t)stream;
    const int q = autocast ? 2 : 1;
    const int qa = x_scales ? q : 0, qw = w_scales ? q : 0;
#define QL_CASE(A, W) \
    if (qa == A && qw == W) return launch_ql<A, W>(a, ablation, st);
    QL_CASE(0, 0) QL_CASE(0, 1) QL_CASE(1, 0) QL_CASE(1, 1) QL_CASE(0, 2) QL_CASE(2, 0) QL_CASE(2, 2)
#undef QL_CASE
    return fail(FQ_ERR_ARG, "unsupported operand arithmetic combination");
}

}  // extern "C"
