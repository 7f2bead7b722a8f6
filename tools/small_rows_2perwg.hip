// small_rows_2perwg.hip -- one bounded experiment (VERDICT r04 "next round" #8): the [2048,4096]-sized launches sit at 0.60-0.65 of the
// HBM roofline because a launch of 2048-4096 workgroups of one row each is mostly launch boundary + a single round of dependent
// load -> reduce -> store chains.  Does putting TWO rows into one 512-thread workgroup (two independent 256-thread halves; half the
// workgroups, one launch boundary amortised over twice the work per workgroup) help?
//   forward : the K4 + V4 pair launch on [2048,4096] bf16 x 2 in training mode (bounds + STE mask): product row_reg_kernel<BF16,256,2>
//             vs the same arithmetic with 2 (and 4) rows per workgroup
//   backward: the mask backward of one [2048,4096] bf16 activation gradient (a copying slot): product ste_mask_kernel<BF16,2> vs 2 rows
//             per 512-thread workgroup
// Every variant is compared with the product kernel bit for bit before it is timed; interleaved rounds on rotating buffers, medians.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I llm-qat_amd/csrc -o tools/small_rows_2perwg tools/small_rows_2perwg.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fq_kernels.h"

using namespace fq;

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

// ---- forward: RPB rows per workgroup of RPB * TPR threads; each group of TPR threads is one row of row_reg_kernel<BF16, TPR, VPT, Sym, FAST>
template <int TPR, int VPT, int RPB> __global__ __launch_bounds__(TPR* RPB) void fwd_rpb_kernel(RowArgs a) {
    using T = Ty<BF16>;
    constexpr int NW = TPR / 64;
    __shared__ uint32_t red[RPB][NW];
    const int part = threadIdx.x / TPR, t = threadIdx.x % TPR, wv = (threadIdx.x % TPR) >> 6;
    int64_t row = (int64_t)blockIdx.x * RPB + part;
    const bool live = row < a.rows;   // a dead part computes on a copy of the last row and stores nothing: every wave reaches the barrier
    row = live ? row : a.rows - 1;
    const void* xb = a.x;
    void* yb = a.y;
    float* bnd = a.bounds;
    uint64_t* msk = a.mask;
    SymConst symk = a.sym;
    if (row >= a.rows0) {
        xb = a.more[0].x, yb = a.more[0].y, bnd = a.more[0].bounds, msk = a.more[0].mask, symk.qmax = a.more[0].qmax;
        row -= a.more[0].row_begin;
    }
    const int nvec = (int)(a.cols / 8);
    const uint4* __restrict__ xr = (const uint4*)((const char*)xb + row * a.cols * 2);
    uint4* __restrict__ yr = (uint4*)((char*)yb + row * a.cols * 2);
    uint4 r[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        int v = t + i * TPR;
        v = v < nvec ? v : nvec - 1;
        r[i] = ld16<false>(&xr[v]);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        acc = T::absmax_acc(acc, r[i].x);
        acc = T::absmax_acc(acc, r[i].y);
        acc = T::absmax_acc(acc, r[i].z);
        acc = T::absmax_acc(acc, r[i].w);
    }
    const uint32_t w = wave_reduce<OpMaxU>(T::absmax_finish(acc));
    if ((threadIdx.x & 63) == 0) red[part][wv] = w;
    __syncthreads();
    uint32_t mb = red[part][0];
#pragma unroll
    for (int i = 1; i < NW; ++i) mb = OpMaxU::f(mb, red[part][i]);
    const float m = as_f(mb);
    const SymRow sr = sym_row<BF16>(m, symk);
    if (t == 0 && bnd && live) {
        bnd[2 * row] = m;
        bnd[2 * row + 1] = -m;
    }
    const bool want_mask = msk && !((m < a.hi) && (-m > a.lo));
    const bool sym_clip = a.lo == -a.hi;
    const uint32_t clipk = (m != m) ? 0u : a.clipk;
    uint8_t* mrow = (uint8_t*)(msk + row * a.mask_row_words);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = t + i * TPR;
        const uint32_t wd[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        float f[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float fd[2];
            T::unpack(wd[d], fd);
            f[2 * d] = fd[0];
            f[2 * d + 1] = fd[1];
        }
        if (want_mask) ste_mask_record<BF16>(mrow, v, v < nvec && live, r[i], f, a.lo, a.hi, sym_clip, clipk);
        uint32_t o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float fd[2] = {f[2 * d], f[2 * d + 1]};
            o[d] = sym_chain<BF16, true>(fd, sr, nullptr);
        }
        if (v < nvec && live) st16<true>(&yr[v], make_uint4(o[0], o[1], o[2], o[3]));
    }
}

// ---- backward: RPB rows per workgroup of RPB * STE_THREADS threads, one copying slot, one chunk per row (cols <= 256 * VPT * 8)
template <int VPT, int RPB>
__global__ __launch_bounds__(STE_THREADS* RPB) void bwd_rpb_kernel(const void* g, void* gx, const float* bounds, const uint64_t* mask, int64_t rows,
                                                                   int64_t nvec_row, int64_t mask_row_words, float lo, float hi) {
    const int part = threadIdx.x / STE_THREADS, t = threadIdx.x % STE_THREADS;
    const int64_t row = (int64_t)blockIdx.x * RPB + part;
    if (row >= rows) return;   // (no barrier in this kernel; whole waves leave together: STE_THREADS is a multiple of 64)
    ste_mask_chunk<BF16, VPT, false, true>(g, gx, (const uint8_t*)(mask + row * mask_row_words), (int)mask_row_words * 2, row, nvec_row, 0, (int)nvec_row, bounds,
                                           lo, hi, t, RowPitch{}, RowPitch{});
}

__global__ void fill_bf16(uint16_t* p, int64_t n, uint32_t seed, float scale, int outliers) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        float v = ((float)(h & 255) + (float)((h >> 8) & 255) + (float)((h >> 16) & 255) + (float)(h >> 24) - 510.0f) / 255.0f;
        if (outliers && (h % 1000u) == 0) v *= 20.0f;
        p[i] = __builtin_bit_cast(uint16_t, (__bf16)(v * scale));
    }
}

struct Set {
    void *k, *v, *yk, *yv, *mk, *mv, *g, *gx;
    float *bk, *bv;
};

int main(int argc, char** argv) {
    const int64_t rows = argc > 2 ? atoll(argv[1]) : 2048, cols = argc > 2 ? atoll(argv[2]) : 4096;
    const int rounds = argc > 3 ? atoi(argv[3]) : 7;
    constexpr int TPR = 256, VPT = 2;
    const int64_t n = rows * cols, nvec = cols / 8;
    if (cols % 8 || nvec > TPR * VPT || nvec <= TPR * (VPT - 1)) {
        fprintf(stderr, "this harness is built for the 256 x 2 launch shape (cols in (2048, 4096], cols %% 8 == 0)\n");
        return 2;
    }
    const size_t bytes = (size_t)n * 2, mrw = (cols + 63) / 64, mbytes = (size_t)rows * mrw * 8;
    const int NS = 12, IT = 200;   // 12 sets x 8 tensors x 16.8 MB: far beyond the Infinity Cache
    std::vector<Set> sets(NS);
    for (int s = 0; s < NS; ++s) {
        Set& q = sets[s];
        CK(hipMalloc(&q.k, bytes)); CK(hipMalloc(&q.v, bytes)); CK(hipMalloc(&q.yk, bytes)); CK(hipMalloc(&q.yv, bytes)); CK(hipMalloc(&q.g, bytes)); CK(hipMalloc(&q.gx, bytes));
        CK(hipMalloc(&q.mk, mbytes)); CK(hipMalloc(&q.mv, mbytes)); CK(hipMalloc((void**)&q.bk, rows * 8)); CK(hipMalloc((void**)&q.bv, rows * 8));
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (uint16_t*)q.k, n, 17u + s, 1.0f, 1);
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (uint16_t*)q.v, n, 99u + s, 1.0f, 1);
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (uint16_t*)q.g, n, 7u + s, 0.01f, 0);
    }
    CK(hipDeviceSynchronize());
    auto args = [&](const Set& q) {
        RowArgs a{};
        a.x = q.k; a.y = q.yk; a.bounds = q.bk; a.rows = 2 * rows; a.rows0 = rows; a.cols = cols;
        a.sym.qmax = 7.0f; a.sym.c6 = 9.98377799987793e-07f;
        a.mask = (uint64_t*)q.mk; a.mask_row_words = (int64_t)mrw; a.lo = -2.0f; a.hi = 2.0f; a.clipk = 0x40004000u;
        a.n_more = 1;
        a.more[0] = TensorSlot{rows, q.v, q.yv, q.bv, (uint64_t*)q.mv, 7.0f, {}, {}};
        for (int i = 1; i < MAX_MORE; ++i) { a.more[i] = TensorSlot{}; a.more[i].row_begin = INT64_MAX; }
        return a;
    };
    auto fwd = [&](int rpb, const Set& q) {
        RowArgs a = args(q);
        const int64_t total = 2 * rows;
        if (rpb == -1) hipLaunchKernelGGL((fwd_rpb_kernel<TPR, VPT, 1>), dim3((unsigned)total), dim3(TPR), 0, 0, a);   // the lean two-tensor forward, one row per workgroup
        else if (rpb == 1) hipLaunchKernelGGL((row_reg_kernel<BF16, TPR, VPT, false, true, false, true>), dim3((unsigned)total), dim3(TPR), 0, 0, a);
        else if (rpb == 2) hipLaunchKernelGGL((fwd_rpb_kernel<TPR, VPT, 2>), dim3((unsigned)((total + 1) / 2)), dim3(TPR * 2), 0, 0, a);
        else hipLaunchKernelGGL((fwd_rpb_kernel<TPR, VPT, 4>), dim3((unsigned)((total + 3) / 4)), dim3(TPR * 4), 0, 0, a);
    };
    auto bwd = [&](int rpb, const Set& q) {
        if (rpb == -1) {   // the lean kernel with ONE row per workgroup: separates "two rows per workgroup" from "no slot table, no in-place path"
            hipLaunchKernelGGL((bwd_rpb_kernel<VPT, 1>), dim3((unsigned)rows), dim3(STE_THREADS), 0, 0, q.g, q.gx, q.bk, (const uint64_t*)q.mk, rows, nvec, (int64_t)mrw, -2.0f, 2.0f);
        } else if (rpb == 1) {
            SteLaunch L{};
            L.n = 1;
            L.t[0] = SteSlot{q.g, q.gx, q.bk, (const uint64_t*)q.mk, rows, 0, 0, {}, {}};
            for (int i = 1; i < 1 + MAX_MORE; ++i) { L.t[i] = SteSlot{}; L.t[i].blk_begin = INT64_MAX; }
            hipLaunchKernelGGL((ste_mask_kernel<BF16, VPT, false, true>), dim3((unsigned)rows, 1), dim3(STE_THREADS), 0, 0, L, nvec, (int)nvec, (int64_t)mrw, -2.0f, 2.0f);
        } else if (rpb == 2) {
            hipLaunchKernelGGL((bwd_rpb_kernel<VPT, 2>), dim3((unsigned)((rows + 1) / 2)), dim3(STE_THREADS * 2), 0, 0, q.g, q.gx, q.bk, (const uint64_t*)q.mk, rows, nvec,
                               (int64_t)mrw, -2.0f, 2.0f);
        } else {
            hipLaunchKernelGGL((bwd_rpb_kernel<VPT, 4>), dim3((unsigned)((rows + 3) / 4)), dim3(STE_THREADS * 4), 0, 0, q.g, q.gx, q.bk, (const uint64_t*)q.mk, rows, nvec,
                               (int64_t)mrw, -2.0f, 2.0f);
        }
    };
    // ---- correctness against the product kernels on set 0
    std::vector<char> y0(bytes), y1(bytes), m0(mbytes), m1(mbytes);
    std::vector<float> b0(rows * 2), b1(rows * 2);
    fwd(1, sets[0]);
    CK(hipDeviceSynchronize());
    std::vector<char> yk0(bytes), yv0(bytes), gx0(bytes);
    CK(hipMemcpy(yk0.data(), sets[0].yk, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(yv0.data(), sets[0].yv, bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(m0.data(), sets[0].mk, mbytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(b0.data(), sets[0].bk, rows * 8, hipMemcpyDeviceToHost));
    bwd(1, sets[0]);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(gx0.data(), sets[0].gx, bytes, hipMemcpyDeviceToHost));
    for (int rpb : {2, 4}) {
        CK(hipMemset(sets[0].yk, 0, bytes)); CK(hipMemset(sets[0].yv, 0, bytes)); CK(hipMemset(sets[0].gx, 0, bytes));
        fwd(rpb, sets[0]);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        bool ok = true;
        CK(hipMemcpy(y1.data(), sets[0].yk, bytes, hipMemcpyDeviceToHost));
        ok = ok && !memcmp(yk0.data(), y1.data(), bytes);
        CK(hipMemcpy(y1.data(), sets[0].yv, bytes, hipMemcpyDeviceToHost));
        ok = ok && !memcmp(yv0.data(), y1.data(), bytes);
        CK(hipMemcpy(b1.data(), sets[0].bk, rows * 8, hipMemcpyDeviceToHost));
        ok = ok && !memcmp(b0.data(), b1.data(), rows * 8);
        CK(hipMemcpy(m1.data(), sets[0].mk, mbytes, hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < rows && ok; ++r)
            if (b0[2 * r] >= 2.0f && memcmp(m0.data() + r * mrw * 8, m1.data() + r * mrw * 8, (cols + 7) / 8)) ok = false;
        bwd(rpb, sets[0]);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        CK(hipMemcpy(y1.data(), sets[0].gx, bytes, hipMemcpyDeviceToHost));
        ok = ok && !memcmp(gx0.data(), y1.data(), bytes);
        printf("%d rows per workgroup: forward and backward %s\n", rpb, ok ? "bit-identical to the product kernels" : "MISMATCH");
        if (!ok) return 1;
    }
    // ---- timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rp[4] = {1, 2, 4, -1};
    std::vector<float> uf[4], ub[4];
    auto timed = [&](auto&& fn) {
        for (int i = 0; i < 24; ++i) fn(sets[i % NS]);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < IT; ++i) fn(sets[i % NS]);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / IT;
    };
    for (int r = 0; r < rounds; ++r)
        for (int vi = 0; vi < 4; ++vi) {
            const int v = (r & 1) ? 3 - vi : vi;
            uf[v].push_back(timed([&](const Set& q) { fwd(rp[v], q); }));
            ub[v].push_back(timed([&](const Set& q) { bwd(rp[v], q); }));
        }
    auto med = [](std::vector<float>& x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    const double fb = 2.0 * 2.0 * bytes + 2.0 * n / 8, bb = 2.0 * bytes + n / 8.0;
    printf("[%lld, %lld] bf16, %d rounds x %d launches on %d rotating buffer sets, median us (min .. max):\n", (long long)rows, (long long)cols, rounds, IT, NS);
    const float f1 = med(uf[0]), b1m = med(ub[0]);
    for (int v = 0; v < 4; ++v) {
        const float f = med(uf[v]), b = med(ub[v]);
        if (rp[v] < 0) {
            printf("  1 row / workgroup, the LEAN kernels (forward: two tensors, no slot loop; backward: one slot, no in-place path, no LDS): pair forward %6.2f us (%.2f .. %.2f) "
                   "vs product %.3f   |   A8 mask backward %6.2f us (%.2f .. %.2f)  frac %.3f  vs product %.3f\n", f, uf[v].front(), uf[v].back(), f / f1, b,
                   ub[v].front(), ub[v].back(), bb / (b * 1e-6) / 8e12, b / b1m);
            continue;
        }
        printf("  %d row(s) / workgroup: K4 + V4 pair forward (training mode) %6.2f us (%.2f .. %.2f)  frac %.3f  vs product %.3f   |   A8 mask backward %6.2f us (%.2f .. %.2f)  frac %.3f  vs product %.3f\n",
               rp[v], f, uf[v].front(), uf[v].back(), fb / (f * 1e-6) / 8e12, f / f1, b, ub[v].front(), ub[v].back(), bb / (b * 1e-6) / 8e12, b / b1m);
    }
    return 0;
}
