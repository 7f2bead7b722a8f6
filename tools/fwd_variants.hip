// fwd_variants.hip -- A/B harness for the metric forward: does the Sym forward's row load go faster
//   (1) through LDS-DMA (global_load_lds_dwordx4: the row lands in LDS without passing VGPRs, read back with ds_read_b128),
//   (2) software-pipelined over rows (a workgroup issues row r+1's 16-byte loads before it reduces / stores row r), or
//   (3) both (row r+1's LDS-DMA in flight in a second LDS buffer while row r is processed)
// than the product's one-row-per-workgroup register kernel (variant 0 = row_reg_kernel itself, called with the same RowArgs)?
// Same arithmetic (the product's device functions), same side outputs (row bounds + STE mask), same NT policy, one box, interleaved
// rounds.  Every variant's output is compared with variant 0's bit for bit before it is timed.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I llm-qat_amd/csrc -o tools/fwd_variants tools/fwd_variants.hip
//   ./tools/fwd_variants [rows cols rounds]        (default 4096 11008 5: the W4 + A8 pair launch of the metric step)
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

struct Sel {
    const void* xb;
    void* yb;
    float* bnd;
    uint64_t* msk;
    SymConst symk;
    int64_t row;
};
// which tensor of the launch a row belongs to (as row_reg_kernel does it)
__device__ __forceinline__ Sel select_tensor(const RowArgs& a, int64_t row) {
    Sel s{a.x, a.y, a.bounds, a.mask, a.sym, row};
    if (row >= a.rows0) {
        int64_t rbase = 0;
#pragma unroll
        for (int i = 0; i < MAX_MORE; ++i) {
            if (i < a.n_more && row >= a.more[i].row_begin) {
                rbase = a.more[i].row_begin;
                s.xb = a.more[i].x;
                s.yb = a.more[i].y;
                s.bnd = a.more[i].bounds;
                s.msk = a.more[i].mask;
                s.symk.qmax = a.more[i].qmax;
            }
        }
        s.row = row - rbase;
    }
    return s;
}

// reduce -> scale -> bounds -> (mask) -> round -> dequant of one row held as r[VPT]: the body of row_reg_kernel<BF16, .., Sym, FAST>, split
// into a compute phase (results + mask bytes in registers) and a store phase, so that a pipelined variant can wait for the NEXT
// row's loads between the two (loads are then never queued behind this row's stores on the in-order VM counter)
// block max through LDS with a RAW barrier: __syncthreads() would fence, and with an LDS-DMA in flight hipcc turns that fence into
// s_waitcnt vmcnt(0) -- draining the next row's prefetch at this row's reduction
template <int NW> __device__ __forceinline__ uint32_t block_max_raw(uint32_t v, uint32_t* lds) {
    const uint32_t w = wave_reduce<OpMaxU>(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = w;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    uint32_t r = lds[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = OpMaxU::f(r, lds[i]);
    return r;
}

template <int VPT> struct RowOut {
    uint4 o[VPT];
    uint32_t mbyte[VPT];
    bool want_mask;
};
template <int TPR, int VPT>
__device__ __forceinline__ void compute_row(const RowArgs& a, const Sel& s, const uint4 (&r)[VPT], int t, int nvec, uint32_t* red, RowOut<VPT>& out) {
    using T = Ty<BF16>;
    constexpr int NW = TPR / 64;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        acc = T::absmax_acc(acc, r[i].x);
        acc = T::absmax_acc(acc, r[i].y);
        acc = T::absmax_acc(acc, r[i].z);
        acc = T::absmax_acc(acc, r[i].w);
    }
    const float m = as_f(block_max_raw<NW>(T::absmax_finish(acc), red));
    const SymRow sr = sym_row<BF16>(m, s.symk);
    if (t == 0 && s.bnd) {
        s.bnd[2 * s.row] = m;
        s.bnd[2 * s.row + 1] = -m;
    }
    out.want_mask = s.msk && !((m < a.hi) && (-m > a.lo));
    const bool sym_clip = a.lo == -a.hi;
    const uint32_t clipk = (m != m) ? 0u : a.clipk;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const uint32_t w[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        float f[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float fd[2];
            T::unpack(w[d], fd);
            f[2 * d] = fd[0];
            f[2 * d + 1] = fd[1];
        }
        out.mbyte[i] = 0;
        if (out.want_mask) out.mbyte[i] = clipk ? ste_flags16_vec(r[i], clipk) : (sym_clip ? ste_flags_f<8, true>(f, a.lo, a.hi) : ste_flags_f<8, false>(f, a.lo, a.hi));
        uint32_t o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float fd[2] = {f[2 * d], f[2 * d + 1]};
            o[d] = sym_chain<BF16, true>(fd, sr, nullptr);
        }
        out.o[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
template <int TPR, int VPT> __device__ __forceinline__ void store_row(const RowArgs& a, const Sel& s, int t, int nvec, const RowOut<VPT>& out) {
    uint8_t* mrow = (uint8_t*)(s.msk + s.row * a.mask_row_words);
    uint4* __restrict__ yr = (uint4*)((char*)s.yb + s.row * a.cols * 2);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = t + i * TPR;
        if (v < nvec) {
            if (out.want_mask) mrow[v] = (uint8_t)out.mbyte[i];
            st16<true>(&yr[v], out.o[i]);
        }
    }
}
template <int TPR, int VPT>
__device__ __forceinline__ void process_row(const RowArgs& a, const Sel& s, const uint4 (&r)[VPT], int t, int nvec, uint32_t* red) {
    RowOut<VPT> out;
    compute_row<TPR, VPT>(a, s, r, t, nvec, red, out);
    store_row<TPR, VPT>(a, s, t, nvec, out);
}

// LDS-DMA of this thread's VPT vectors of a row: wave w's instruction i writes 1 KiB at lds + (i * TPR + w * 64) * 16 (lane-linear)
template <int TPR, int VPT>
__device__ __forceinline__ void dma_row(const uint4* __restrict__ xr, int t, int nvec, uint4* lds) {
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        int v = t + i * TPR;
        v = v < nvec ? v : nvec - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(&xr[v]),
                                         (__attribute__((address_space(3))) void*)(lds + i * TPR + (t & ~63)), 16, 0, 2 /* nt */);
    }
}
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// the lane reads back what its own lane's DMA wrote (no barrier needed).  In asm: hipcc drains vmcnt(0) before a ds_read it can
// associate with a pending LDS-DMA (cdna_hip_programming.md §5 "Pipelining across barriers"), which would also wait for the NEXT row
template <int TPR, int VPT> __device__ __forceinline__ void lds_row(const uint4* lds, int t, uint4 (&r)[VPT]) {
    static_assert(VPT == 3 && TPR * 16 * 2 < 65536, "three ds_read_b128 with immediate offsets");
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(lds + t);
    u32x4_t a, b, c;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:%4\n\tds_read_b128 %2, %3 offset:%5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(addr), "n"(TPR * 16), "n"(TPR * 32) : "memory");
    r[0] = make_uint4(a.x, a.y, a.z, a.w);
    r[1] = make_uint4(b.x, b.y, b.z, b.w);
    r[2] = make_uint4(c.x, c.y, c.z, c.w);
}

// MODE 1: LDS-DMA, one row per workgroup.  MODE 2: registers, rows_per_wg rows per workgroup, next row's loads issued before this row's
// reduce / stores.  MODE 3: LDS-DMA, rows_per_wg rows per workgroup, next row's DMA in flight in the other LDS buffer.
template <int TPR, int VPT, int MODE>
__global__ __launch_bounds__(TPR) void fwd_variant_kernel(RowArgs a, int rows_per_wg) {
    constexpr int NW = TPR / 64;
    __shared__ uint32_t red[2][NW];
    __shared__ uint4 stage[(MODE == 1 ? 1 : MODE == 3 ? 2 : 0) * TPR * VPT + 1];
    const int t = threadIdx.x;
    const int nvec = (int)(a.cols / 8);
    if constexpr (MODE == 1) {
        const Sel s = select_tensor(a, blockIdx.x);
        const uint4* xr = (const uint4*)((const char*)s.xb + s.row * a.cols * 2);
        dma_row<TPR, VPT>(xr, t, nvec, stage);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 r[VPT];
        lds_row<TPR, VPT>(stage, t, r);
        process_row<TPR, VPT>(a, s, r, t, nvec, red[0]);
    } else if constexpr (MODE == 2) {
        // workgroup b owns rows b, b + grid, b + 2 grid, ... (consecutive workgroups stream consecutive rows at any moment)
        int64_t row = blockIdx.x;
        Sel s = select_tensor(a, row);
        uint4 cur[VPT], nxt[VPT];
        {
            const uint4* xr = (const uint4*)((const char*)s.xb + s.row * a.cols * 2);
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                int v = t + i * TPR;
                v = v < nvec ? v : nvec - 1;
                cur[i] = ld16<true>(&xr[v]);
            }
        }
        for (int k = 0; k < rows_per_wg; ++k) {
            const int64_t nrow = row + gridDim.x;
            const bool more = (k + 1 < rows_per_wg) && nrow < a.rows;
            Sel sn = s;
            if (more) {
                sn = select_tensor(a, nrow);
                const uint4* xr = (const uint4*)((const char*)sn.xb + sn.row * a.cols * 2);
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    int v = t + i * TPR;
                    v = v < nvec ? v : nvec - 1;
                    nxt[i] = ld16<true>(&xr[v]);
                }
            }
            RowOut<VPT> out;
            compute_row<TPR, VPT>(a, s, cur, t, nvec, red[k & 1], out);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {   // the next row's loads are the only VM operations outstanding here: wait for them BEFORE this row's stores queue up
#pragma unroll
                for (int i = 0; i < VPT; ++i) cur[i] = nxt[i];
            }
            __builtin_amdgcn_sched_barrier(0);
            store_row<TPR, VPT>(a, s, t, nvec, out);
            if (!more) break;
            s = sn;
            row = nrow;
        }
    } else {
        int64_t row = blockIdx.x;
        Sel s = select_tensor(a, row);
        dma_row<TPR, VPT>((const uint4*)((const char*)s.xb + s.row * a.cols * 2), t, nvec, stage);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int k = 0; k < rows_per_wg; ++k) {
            const int64_t nrow = row + gridDim.x;
            const bool more = (k + 1 < rows_per_wg) && nrow < a.rows;
            Sel sn = s;
            if (more) {
                sn = select_tensor(a, nrow);
                // the other buffer was last READ by this very lane one iteration ago (program order, lgkmcnt(0) behind the reads): free
                dma_row<TPR, VPT>((const uint4*)((const char*)sn.xb + sn.row * a.cols * 2), t, nvec, stage + ((k + 1) & 1) * TPR * VPT);
            }
            uint4 r[VPT];
            lds_row<TPR, VPT>(stage + (k & 1) * TPR * VPT, t, r);   // this row's DMA was waited for before the previous row's stores
            RowOut<VPT> out;
            compute_row<TPR, VPT>(a, s, r, t, nvec, red[k & 1], out);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the next row has landed (nothing younger is outstanding yet)
            __builtin_amdgcn_sched_barrier(0);
            store_row<TPR, VPT>(a, s, t, nvec, out);
            if (!more) break;
            s = sn;
            row = nrow;
        }
    }
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
    void *w, *a, *yw, *ya, *mw, *ma;
    float *bw, *ba;
};

int main(int argc, char** argv) {
    const int64_t rows = argc > 2 ? atoll(argv[1]) : 4096, cols = argc > 2 ? atoll(argv[2]) : 11008;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    constexpr int TPR = 512, VPT = 3;
    const int64_t n = rows * cols, nvec = cols / 8;
    if (cols % 8 || nvec > TPR * VPT || nvec <= TPR * (VPT - 1)) {
        fprintf(stderr, "this harness is built for the 512 x 3 launch shape (cols in (8192, 12288], cols %% 8 == 0)\n");
        return 2;
    }
    const size_t bytes = (size_t)n * 2, mrw = (cols + 63) / 64, mbytes = (size_t)rows * mrw * 8;
    const int NS = 4, IT = 100;
    std::vector<Set> sets(NS);
    for (int s = 0; s < NS; ++s) {
        Set& q = sets[s];
        CK(hipMalloc(&q.w, bytes)); CK(hipMalloc(&q.a, bytes)); CK(hipMalloc(&q.yw, bytes)); CK(hipMalloc(&q.ya, bytes));
        CK(hipMalloc(&q.mw, mbytes)); CK(hipMalloc(&q.ma, mbytes)); CK(hipMalloc((void**)&q.bw, rows * 8)); CK(hipMalloc((void**)&q.ba, rows * 8));
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (uint16_t*)q.w, n, 17u + s, 0.02f, 0);
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (uint16_t*)q.a, n, 99u + s, 1.0f, 1);
    }
    void *cy_w, *cy_a, *cm_a;   // variant 0's outputs of set 0, for the bit-for-bit check
    CK(hipMalloc(&cy_w, bytes)); CK(hipMalloc(&cy_a, bytes)); CK(hipMalloc(&cm_a, mbytes));
    CK(hipDeviceSynchronize());

    auto args = [&](const Set& q) {
        RowArgs a{};
        a.x = q.w; a.y = q.yw; a.bounds = q.bw; a.rows = 2 * rows; a.rows0 = rows; a.cols = cols;
        a.sym.qmax = 7.0f; a.sym.c6 = 9.98377799987793e-07f;
        a.mask = (uint64_t*)q.mw; a.mask_row_words = (int64_t)mrw; a.lo = -2.0f; a.hi = 2.0f; a.clipk = 0x40004000u;   // 0x8000 - bits(2.0 bf16 = 0x4000)
        a.n_more = 1;
        a.more[0] = TensorSlot{rows, q.a, q.ya, q.ba, (uint64_t*)q.ma, 127.0f};
        for (int i = 1; i < MAX_MORE; ++i) { a.more[i] = TensorSlot{}; a.more[i].row_begin = INT64_MAX; }
        return a;
    };
    struct Var { const char* name; int mode, rpw; };
    const Var vars[] = {{"0 product row_reg_kernel<512,3> (one row / WG)", 0, 1},
                        {"1 LDS-DMA row load, one row / WG", 1, 1},
                        {"2 pipelined rows, registers, 2 rows / WG", 2, 2},
                        {"2 pipelined rows, registers, 4 rows / WG", 2, 4},
                        {"2 pipelined rows, registers, 8 rows / WG", 2, 8},
                        {"3 LDS-DMA + pipelined, 2 rows / WG", 3, 2},
                        {"3 LDS-DMA + pipelined, 4 rows / WG", 3, 4},
                        {"3 LDS-DMA + pipelined, 8 rows / WG", 3, 8}};
    const int NV = sizeof(vars) / sizeof(vars[0]);
    auto launch = [&](const Var& v, const Set& q) {
        RowArgs a = args(q);
        const int64_t total = 2 * rows;
        if (v.mode == 0) hipLaunchKernelGGL((row_reg_kernel<BF16, TPR, VPT, false, true, true, true>), dim3((unsigned)total), dim3(TPR), 0, 0, a);
        else {
            const unsigned grid = (unsigned)((total + v.rpw - 1) / v.rpw);
            if (v.mode == 1) hipLaunchKernelGGL((fwd_variant_kernel<TPR, VPT, 1>), dim3((unsigned)total), dim3(TPR), 0, 0, a, 1);
            else if (v.mode == 2) hipLaunchKernelGGL((fwd_variant_kernel<TPR, VPT, 2>), dim3(grid), dim3(TPR), 0, 0, a, v.rpw);
            else hipLaunchKernelGGL((fwd_variant_kernel<TPR, VPT, 3>), dim3(grid), dim3(TPR), 0, 0, a, v.rpw);
        }
    };
    // ---- correctness: every variant == variant 0 on set 0 (values of both tensors, the A8 tensor's mask)
    launch(vars[0], sets[0]);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(cy_w, sets[0].yw, bytes, hipMemcpyDeviceToDevice)); CK(hipMemcpy(cy_a, sets[0].ya, bytes, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(cm_a, sets[0].ma, mbytes, hipMemcpyDeviceToDevice));
    std::vector<char> h0(bytes), h1(bytes), m0(mbytes), m1(mbytes);
    std::vector<float> b0(rows * 2), b1(rows * 2);
    CK(hipMemcpy(b0.data(), sets[0].ba, rows * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(m0.data(), cm_a, mbytes, hipMemcpyDeviceToHost));
    for (int vi = 1; vi < NV; ++vi) {
        CK(hipMemset(sets[0].yw, 0, bytes)); CK(hipMemset(sets[0].ya, 0, bytes));
        launch(vars[vi], sets[0]);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        bool ok = true;
        CK(hipMemcpy(h0.data(), cy_w, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), sets[0].yw, bytes, hipMemcpyDeviceToHost));
        ok = ok && !memcmp(h0.data(), h1.data(), bytes);
        CK(hipMemcpy(h0.data(), cy_a, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), sets[0].ya, bytes, hipMemcpyDeviceToHost));
        ok = ok && !memcmp(h0.data(), h1.data(), bytes);
        CK(hipMemcpy(b1.data(), sets[0].ba, rows * 8, hipMemcpyDeviceToHost));
        ok = ok && !memcmp(b0.data(), b1.data(), rows * 8);
        CK(hipMemcpy(m1.data(), sets[0].ma, mbytes, hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < rows && ok; ++r)   // the mask is defined only for rows whose bounds reach the clip
            if (b0[2 * r] >= 2.0f && memcmp(m0.data() + r * mrw * 8, m1.data() + r * mrw * 8, (cols + 7) / 8)) ok = false;
        printf("variant %-52s %s\n", vars[vi].name, ok ? "bit-identical to the product kernel" : "MISMATCH");
        if (!ok) return 1;
    }
    // ---- timing: interleaved rounds, median of rounds
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> us(NV);
    for (int r = 0; r < rounds; ++r)
        for (int vi = 0; vi < NV; ++vi) {
            for (int i = 0; i < 20; ++i) launch(vars[vi], sets[i % NS]);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < IT; ++i) launch(vars[vi], sets[i % NS]);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            us[vi].push_back(ms * 1e3f / IT);
        }
    const double alg = 2.0 * 2.0 * bytes;   // read x + write y of both tensors
    printf("pair forward W4 + A8 on [%lld, %lld] bf16 x 2 (%.1f MB algorithmic), %d rounds x %d launches, median us:\n", (long long)rows, (long long)cols, alg / 1e6, rounds, IT);
    for (int vi = 0; vi < NV; ++vi) {
        std::sort(us[vi].begin(), us[vi].end());
        const float med = us[vi][us[vi].size() / 2];
        printf("  %-52s %7.2f us  %6.0f GB/s  frac %.3f   (min %.2f max %.2f)  vs product %.3f\n", vars[vi].name, med, alg / (med * 1e-6) / 1e9,
               alg / (med * 1e-6) / 1e9 / 8000.0, us[vi].front(), us[vi].back(), med / us[0][us[0].size() / 2]);
    }
    return 0;
}
