// kbench.hip -- standalone tuning harness for the fake-quant kernels (not part of the product).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I llm-qat_amd/csrc -o tools/kbench tools/kbench.hip
//   ./tools/kbench [rows cols]
//
// Times kernel variants with HIP events on rotating buffer sets (defeats the 256 MiB Infinity
// Cache) and prints achieved algorithmic GB/s.  Used to pick launch shapes; the winners are
// moved into llm-qat_amd/csrc/fq_api.hip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt(const uint4* p) {
    u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_nt(uint4* p, uint4 v) {
    u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (u32x4*)p);
}
// ---------------------------------------------------------------- reference streaming kernels
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t nvec) {
    const int64_t v0 = (int64_t)blockIdx.x * (256 * UNR) + threadIdx.x;
    uint4 r[UNR];
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
        int64_t v = v0 + (int64_t)i * 256;
        v = v < nvec ? v : nvec - 1;
        if constexpr (NT) r[i] = ld_nt(&in[v]);
        else r[i] = in[v];
    }
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
        const int64_t v = v0 + (int64_t)i * 256;
        if (v < nvec) {
            if constexpr (NT) st_nt(&out[v], r[i]);
            else out[v] = r[i];
        }
    }
}

// read R vectors, write 1 (the export kernels' byte ratio: int8 = 2:1, int4 = 4:1): what does the device sustain?
template <int R, bool NT>
__global__ __launch_bounds__(256) void shrink_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t nout) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t oc = o < nout ? o : nout - 1;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < R; ++k) {   // R input planes, each read fully coalesced
        const uint4 v = NT ? ld_nt(&in[(int64_t)k * nout + oc]) : in[(int64_t)k * nout + oc];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (o < nout) {
        if constexpr (NT) st_nt(&out[o], acc);
        else out[o] = acc;
    }
}

// The export kernels' STRUCTURE without their arithmetic: one row per block, the row loaded once into registers (VPT x 16 B per
// thread, NT), a block-wide max through LDS (one barrier -- the stores depend on every load of the row), then 16 / R bytes per
// vector slot written with NT stores (R = 2: 8 B as int8 bins, R = 4: 4 B as int4 bins).  Against shrink_kernel (same bytes, no row
// structure) this isolates what the load -> reduce -> store dependency of a row-resident block costs for these byte ratios.
template <int R, int TPR, int VPT>
__global__ __launch_bounds__(TPR) void rowshrink_kernel(const uint4* __restrict__ in, uint32_t* __restrict__ out, int nvec_row) {
    __shared__ uint32_t red[TPR / 64];
    const int t = threadIdx.x;
    const uint4* xr = in + (int64_t)blockIdx.x * nvec_row;
    uint4 r[VPT];
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        int v = t + i * TPR;
        v = v < nvec_row ? v : nvec_row - 1;
        r[i] = ld_nt(&xr[v]);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const uint32_t m = (r[i].x | r[i].y | r[i].z | r[i].w) & 0x7FFF7FFFu;
        acc = acc > m ? acc : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)acc, o, 64);
        acc = acc > other ? acc : other;
    }
    if ((t & 63) == 0) red[t >> 6] = acc;
    __syncthreads();
    uint32_t m = red[0];
#pragma unroll
    for (int w = 1; w < TPR / 64; ++w) m = m > red[w] ? m : red[w];
    constexpr int DW = 4 / R;  // dwords stored per slot
    uint32_t* orow = out + (int64_t)blockIdx.x * nvec_row * DW;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = t + i * TPR;
        if (v >= nvec_row) continue;
        if constexpr (R == 4) {
            __builtin_nontemporal_store((r[i].x ^ r[i].y ^ r[i].z ^ r[i].w) + m, orow + v);
        } else {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            u32x2 w = {(r[i].x ^ r[i].y) + m, (r[i].z ^ r[i].w) + m};
            __builtin_nontemporal_store(w, (u32x2*)orow + v);
        }
    }
}

// grid-stride copy: fixed grid, each block walks tiles
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void copy_gs_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t nvec) {
    const int64_t tile = 256 * UNR;
    for (int64_t base = (int64_t)blockIdx.x * tile; base < nvec; base += (int64_t)gridDim.x * tile) {
        uint4 r[UNR];
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            int64_t v = base + threadIdx.x + (int64_t)i * 256;
            v = v < nvec ? v : nvec - 1;
            if constexpr (NT) r[i] = ld_nt(&in[v]);
            else r[i] = in[v];
        }
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const int64_t v = base + threadIdx.x + (int64_t)i * 256;
            if (v < nvec) {
                if constexpr (NT) st_nt(&out[v], r[i]);
                else out[v] = r[i];
            }
        }
    }
}

// copy with an XCD-contiguous block remap: XCD k (= blockIdx % 8 under round-robin dispatch) streams the k-th eighth
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void copy_xcd_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t nvec) {
    const int64_t nb = gridDim.x, per = (nb + 7) / 8;
    const int64_t b = (int64_t)(blockIdx.x % 8) * per + blockIdx.x / 8;
    if (b >= nb) return;
    const int64_t v0 = b * (256 * UNR) + threadIdx.x;
    uint4 r[UNR];
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
        int64_t v = v0 + (int64_t)i * 256;
        v = v < nvec ? v : nvec - 1;
        if constexpr (NT) r[i] = ld_nt(&in[v]);
        else r[i] = in[v];
    }
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
        const int64_t v = v0 + (int64_t)i * 256;
        if (v < nvec) {
            if constexpr (NT) st_nt(&out[v], r[i]);
            else out[v] = r[i];
        }
    }
}

template <int UNR>
__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ in, uint32_t* __restrict__ out, int64_t nvec) {
    const int64_t v0 = (int64_t)blockIdx.x * (256 * UNR) + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
        int64_t v = v0 + (int64_t)i * 256;
        v = v < nvec ? v : nvec - 1;
        uint4 r = in[v];
        acc |= r.x ^ r.y ^ r.z ^ r.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int UNR>
__global__ __launch_bounds__(256) void write_kernel(uint4* __restrict__ out, int64_t nvec) {
    const int64_t v0 = (int64_t)blockIdx.x * (256 * UNR) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
        const int64_t v = v0 + (int64_t)i * 256;
        if (v < nvec) out[v] = make_uint4((uint32_t)v, 1, 2, 3);
    }
}

// 8 bytes per lane (the fp32-result kernels read 16-bit rows / write 16-bit gradients in 8-byte half-vectors): PMC calibration
__global__ __launch_bounds__(256) void read8_kernel(const uint2* __restrict__ in, uint32_t* __restrict__ out, int64_t n8) {
    const int64_t v0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t v = v0 + (int64_t)i * 256;
        v = v < n8 ? v : n8 - 1;
        const uint2 r = in[v];
        acc |= r.x ^ r.y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void write8_kernel(uint2* __restrict__ out, int64_t n8) {
    const int64_t v0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t v = v0 + (int64_t)i * 256;
        if (v < n8) out[v] = make_uint2((uint32_t)v, 7);
    }
}

// ---------------------------------------------------------------- harness
struct Bufs {
    std::vector<void*> x, y, g, gx;
    std::vector<float*> bounds;
};

static float time_it(const std::function<void(int)>& fn, int iters, int warm = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) fn(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fn(i);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / iters;
}

static void report(const char* name, double bytes, float ms) {
    printf("%-44s %8.2f us  %8.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e9 / 80.0);
    fflush(stdout);
}

__global__ void fill_bf16(uint16_t* p, int64_t n, uint32_t seed, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        // sum of 4 bytes -> roughly bell shaped in [-2, 2] * scale
        float v = ((float)(h & 255) + (float)((h >> 8) & 255) + (float)((h >> 16) & 255) + (float)(h >> 24) - 510.0f) / 255.0f;
        p[i] = __builtin_bit_cast(uint16_t, (__bf16)(v * scale));
    }
}

template <int TPR, int VPT, bool FAST, bool NTL, bool NTS>
static void launch_sym(const void* x, void* y, float* bounds, int64_t rows, int64_t cols, int bits) {
    RowArgs a{};
    a.x = x; a.y = y; a.idx = nullptr; a.scale = nullptr; a.bounds = bounds; a.rows = rows; a.rows0 = rows; a.cols = cols;
    a.sym.qmax = (float)((1 << (bits - 1)) - 1);
    a.sym.c6 = 9.98377799987793e-07f;
    for (int i = 0; i < MAX_MORE; ++i) a.more[i].row_begin = INT64_MAX;  // single tensor: no slot may match (pick_tensor)
    const int64_t grid = TPR == 64 ? (rows + 3) / 4 : rows;
    hipLaunchKernelGGL((row_reg_kernel<BF16, TPR, VPT, false, FAST, NTL, NTS>), dim3((unsigned)grid), dim3(TPR == 64 ? 256 : TPR), 0, 0, a);
}

template <int TPR, bool NTL, bool NTS>
static bool launch_sym_any(int vpt, const void* x, void* y, float* bounds, int64_t rows, int64_t cols) {
    switch (vpt) {
#define V(N) case N: launch_sym<TPR, N, true, NTL, NTS>(x, y, bounds, rows, cols, 4); return true;
        V(1) V(2) V(3) V(4) V(5) V(6) V(7) V(8)
#undef V
        default: return false;
    }
}

int main(int argc, char** argv) {
    const int64_t rows = argc > 2 ? atoll(argv[1]) : 4096, cols = argc > 2 ? atoll(argv[2]) : 11008;
    const int64_t n = rows * cols;
    const size_t bytes = (size_t)n * 2;
    const int NS = 4, IT = 60;
    Bufs b;
    for (int s = 0; s < NS; ++s) {
        void *x, *y, *g, *gx;
        float* bd;
        CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&g, bytes)); CK(hipMalloc(&gx, bytes));
        CK(hipMalloc(&bd, rows * 2 * sizeof(float)));
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (uint16_t*)x, n, 17u + s, 0.02f);
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (uint16_t*)g, n, 99u + s, 0.001f);
        b.x.push_back(x); b.y.push_back(y); b.g.push_back(g); b.gx.push_back(gx); b.bounds.push_back(bd);
    }
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs, clock %d MHz, mem clock %d MHz, bus %d bits | tensor %lld x %lld bf16 (%.1f MB)\n", prop.name,
           prop.multiProcessorCount, prop.clockRate / 1000, prop.memoryClockRate / 1000, prop.memoryBusWidth, (long long)rows,
           (long long)cols, bytes / 1e6);
    const int64_t nvec = n / 8;

    // ---- ceilings
#define COPY(U, NT)                                                                                                                 \
    report("copy<UNR=" #U ",NT=" #NT ">", 2.0 * bytes, time_it([&](int i) {                                                        \
               hipLaunchKernelGGL((copy_kernel<U, NT>), dim3((unsigned)((nvec + 256 * U - 1) / (256 * U))), dim3(256), 0, 0,       \
                                  (const uint4*)b.x[i % NS], (uint4*)b.y[i % NS], nvec);                                            \
           }, IT));
    COPY(1, false) COPY(2, false) COPY(4, false) COPY(8, false) COPY(4, true) COPY(8, true)
#define COPYGS(U, NT, G)                                                                                                            \
    report("copy_gs<UNR=" #U ",NT=" #NT ",grid=" #G ">", 2.0 * bytes, time_it([&](int i) {                                          \
               hipLaunchKernelGGL((copy_gs_kernel<U, NT>), dim3(G), dim3(256), 0, 0, (const uint4*)b.x[i % NS], (uint4*)b.y[i % NS], nvec); \
           }, IT));
    COPYGS(4, false, 2048) COPYGS(4, false, 4096) COPYGS(8, false, 2048) COPYGS(4, true, 2048) COPYGS(2, false, 4096)
    report("copy_xcd<UNR=4,NT=true>", 2.0 * bytes, time_it([&](int i) {
               const int64_t nb = (nvec + 1023) / 1024;
               hipLaunchKernelGGL((copy_xcd_kernel<4, true>), dim3((unsigned)(((nb + 7) / 8) * 8)), dim3(256), 0, 0, (const uint4*)b.x[i % NS], (uint4*)b.y[i % NS], nvec);
           }, IT));
    report("copy_xcd<UNR=1,NT=true>", 2.0 * bytes, time_it([&](int i) {
               const int64_t nb = (nvec + 255) / 256;
               hipLaunchKernelGGL((copy_xcd_kernel<1, true>), dim3((unsigned)(((nb + 7) / 8) * 8)), dim3(256), 0, 0, (const uint4*)b.x[i % NS], (uint4*)b.y[i % NS], nvec);
           }, IT));
    COPY(1, true) COPY(2, true)
    {   // does the relative placement of the read and the write stream matter? (same-bank / same-channel lock-step)
        char* big;
        CK(hipMalloc((void**)&big, 2 * bytes + (64u << 20)));
        const size_t offs[] = {0, 256, 4096, 65536, 1u << 20, (1u << 20) + 4096 + 256, 8u << 20, (32u << 20) + 12345 * 16};
        for (size_t off : offs) {
            char name[96];
            snprintf(name, sizeof name, "copy<UNR=1,NT=true> dst = src_end + %zu B", off);
            uint4* dst = (uint4*)(big + bytes + off);
            report(name, 2.0 * bytes, time_it([&](int i) {
                       hipLaunchKernelGGL((copy_kernel<1, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const uint4*)big, dst, nvec);
                   }, IT));
        }
        CK(hipFree(big));
    }
#define SHRINK(R)                                                                                                                   \
    report("shrink<read " #R " : write 1, NT>", (1.0 + 1.0 / R) * bytes, time_it([&](int i) {                                      \
               hipLaunchKernelGGL((shrink_kernel<R, true>), dim3((unsigned)((nvec / R + 255) / 256)), dim3(256), 0, 0,             \
                                  (const uint4*)b.x[i % NS], (uint4*)b.y[i % NS], nvec / R);                                        \
           }, IT));
    SHRINK(2) SHRINK(4)
    uint32_t* sink;
    CK(hipMalloc(&sink, 64));
    report("read-only 8 B/lane", 1.0 * bytes, time_it([&](int i) {
               hipLaunchKernelGGL(read8_kernel, dim3((unsigned)((2 * nvec + 1023) / 1024)), dim3(256), 0, 0, (const uint2*)b.x[i % NS], sink, 2 * nvec);
           }, IT));
    report("write-only 8 B/lane", 1.0 * bytes, time_it([&](int i) {
               hipLaunchKernelGGL(write8_kernel, dim3((unsigned)((2 * nvec + 1023) / 1024)), dim3(256), 0, 0, (uint2*)b.y[i % NS], 2 * nvec);
           }, IT));
    if (argc > 3 && std::string(argv[3]) == "placement") {
        // does the DISTANCE between the read stream and the write stream matter (same channel / bank phase)?  src = big, dst = big + D
        for (int s = 0; s < NS; ++s)
            printf("separate allocations, set %d: x %p  y %p  (y - x) = %lld B = %.3f MiB, mod 2 MiB = %lld\n", s, b.x[s], b.y[s],
                   (long long)((char*)b.y[s] - (char*)b.x[s]), ((char*)b.y[s] - (char*)b.x[s]) / 1048576.0,
                   (long long)(((char*)b.y[s] - (char*)b.x[s]) & ((1 << 21) - 1)));
        char* big;
        CK(hipMalloc((void**)&big, (size_t)640 << 20));
        CK(hipMemset(big, 1, (size_t)640 << 20));
        const long long MiB = 1 << 20;
        const long long ds[] = {86 * MiB, 96 * MiB, 128 * MiB, 128 * MiB + 256, 128 * MiB + 4096, 128 * MiB + 65536, 129 * MiB, 130 * MiB, 136 * MiB, 160 * MiB, 192 * MiB,
                                256 * MiB, 256 * MiB + 4096, 257 * MiB, 384 * MiB, 512 * MiB};
        for (long long d : ds) {
            char name[96];
            snprintf(name, sizeof name, "copy<UNR=1,NT=true> dst = src + %.4f MiB", d / 1048576.0);
            report(name, 2.0 * bytes, time_it([&](int i) {
                       hipLaunchKernelGGL((copy_kernel<1, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const uint4*)big, (uint4*)(big + d), nvec);
                   }, IT));
        }
        // is it the SIZE of the allocation (larger physical fragments -> fewer translation misses)?  two separate hipMallocs of S each,
        // only the first 86 MiB of each used
        for (long long S : {88 * MiB, 96 * MiB, 128 * MiB, 192 * MiB, 256 * MiB, 512 * MiB, 1024 * MiB}) {
            char *a, *c;
            CK(hipMalloc((void**)&a, (size_t)S));
            CK(hipMalloc((void**)&c, (size_t)S));
            CK(hipMemset(a, 1, (size_t)S));
            CK(hipMemset(c, 1, (size_t)S));
            char name[128];
            snprintf(name, sizeof name, "copy<UNR=1,NT=true> two hipMallocs of %lld MiB (a %p c %p)", S / MiB, (void*)a, (void*)c);
            report(name, 2.0 * bytes, time_it([&](int i) {
                       hipLaunchKernelGGL((copy_kernel<1, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const uint4*)a, (uint4*)c, nvec);
                   }, IT));
            CK(hipFree(a));
            CK(hipFree(c));
        }
        // constant vs random data x single pair vs 4 rotating pairs, all freshly allocated 88 MiB buffers
        for (int rnd = 0; rnd < 2; ++rnd) {
            char *pa[4], *pc[4];
            for (int k = 0; k < 4; ++k) {
                CK(hipMalloc((void**)&pa[k], (size_t)88 << 20));
                CK(hipMalloc((void**)&pc[k], (size_t)88 << 20));
                if (rnd) {
                    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (uint16_t*)pa[k], n, 1234u + k, 0.02f);
                    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (uint16_t*)pc[k], n, 4321u + k, 0.02f);
                } else {
                    CK(hipMemset(pa[k], 1, (size_t)88 << 20));
                    CK(hipMemset(pc[k], 1, (size_t)88 << 20));
                }
            }
            CK(hipDeviceSynchronize());
            char name[128];
            snprintf(name, sizeof name, "copy NT, fresh 88 MiB buffers, %s data, SINGLE pair", rnd ? "random" : "constant");
            report(name, 2.0 * bytes, time_it([&](int i) {
                       hipLaunchKernelGGL((copy_kernel<1, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const uint4*)pa[0], (uint4*)pc[0], nvec);
                   }, IT));
            snprintf(name, sizeof name, "copy NT, fresh 88 MiB buffers, %s data, 4 ROTATING pairs", rnd ? "random" : "constant");
            report(name, 2.0 * bytes, time_it([&](int i) {
                       hipLaunchKernelGGL((copy_kernel<1, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const uint4*)pa[i % 4], (uint4*)pc[i % 4], nvec);
                   }, IT));
            snprintf(name, sizeof name, "copy plain (no NT), fresh buffers, %s data, 4 ROTATING pairs", rnd ? "random" : "constant");
            report(name, 2.0 * bytes, time_it([&](int i) {
                       hipLaunchKernelGGL((copy_kernel<1, false>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const uint4*)pa[i % 4], (uint4*)pc[i % 4], nvec);
                   }, IT));
            for (int k = 0; k < 4; ++k) { CK(hipFree(pa[k])); CK(hipFree(pc[k])); }
        }
        {   // one allocation carved into 8 rotating 88 MiB slots (what a pre-reserved arena would give), rotating like bench.py
            char* arena;
            CK(hipMalloc((void**)&arena, (size_t)(8 * 88) << 20));
            CK(hipMemset(arena, 1, (size_t)(8 * 88) << 20));
            report("copy<UNR=1,NT=true> ONE 704 MiB arena, 4 rotating (src, dst) slot pairs", 2.0 * bytes, time_it([&](int i) {
                       const int k = i % 4;
                       hipLaunchKernelGGL((copy_kernel<1, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0,
                                          (const uint4*)(arena + (size_t)(2 * k) * (88 << 20)), (uint4*)(arena + (size_t)(2 * k + 1) * (88 << 20)), nvec);
                   }, IT));
            CK(hipFree(arena));
        }
        // and the same copy between the separately allocated rotating sets (what bench.py's buffers look like)
        report("copy<UNR=1,NT=true> separate allocations, rotating", 2.0 * bytes, time_it([&](int i) {
                   hipLaunchKernelGGL((copy_kernel<1, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const uint4*)b.x[i % NS], (uint4*)b.y[i % NS], nvec);
               }, IT));
        report("copy<UNR=1,NT=true> separate allocations, set 0 only", 2.0 * bytes, time_it([&](int i) {
                   hipLaunchKernelGGL((copy_kernel<1, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const uint4*)b.x[0], (uint4*)b.y[0], nvec);
               }, IT));
        return 0;
    }
    if (argc > 3 && std::string(argv[3]) == "ceilings") {
        if (cols % 8 == 0 && cols / 8 <= 512 * 3 && cols / 8 > 512 * 2) {   // the export kernels' launch shape for 11008-column rows
            report("rowshrink<read 2 : write 1, 512 x 3> (export structure, no arithmetic)", 1.5 * bytes, time_it([&](int i) {
                       hipLaunchKernelGGL((rowshrink_kernel<2, 512, 3>), dim3((unsigned)rows), dim3(512), 0, 0, (const uint4*)b.x[i % NS], (uint32_t*)b.y[i % NS], (int)(cols / 8));
                   }, IT));
            report("rowshrink<read 4 : write 1, 512 x 3> (export structure, no arithmetic)", 1.25 * bytes, time_it([&](int i) {
                       hipLaunchKernelGGL((rowshrink_kernel<4, 512, 3>), dim3((unsigned)rows), dim3(512), 0, 0, (const uint4*)b.x[i % NS], (uint32_t*)b.y[i % NS], (int)(cols / 8));
                   }, IT));
            report("rowshrink<read 4 : write 1, 256 x 6> (export structure, no arithmetic)", 1.25 * bytes, time_it([&](int i) {
                       hipLaunchKernelGGL((rowshrink_kernel<4, 256, 6>), dim3((unsigned)rows), dim3(256), 0, 0, (const uint4*)b.x[i % NS], (uint32_t*)b.y[i % NS], (int)(cols / 8));
                   }, IT));
        }
        report("read-only<UNR=4>", 1.0 * bytes, time_it([&](int i) {
                   hipLaunchKernelGGL((read_kernel<4>), dim3((unsigned)((nvec + 1023) / 1024)), dim3(256), 0, 0, (const uint4*)b.x[i % NS], sink, nvec);
               }, IT));
        report("write-only<UNR=4>", 1.0 * bytes, time_it([&](int i) {
                   hipLaunchKernelGGL((write_kernel<4>), dim3((unsigned)((nvec + 1023) / 1024)), dim3(256), 0, 0, (uint4*)b.y[i % NS], nvec);
               }, IT));
        return 0;
    }
    report("read-only<UNR=4>", 1.0 * bytes, time_it([&](int i) {
               hipLaunchKernelGGL((read_kernel<4>), dim3((unsigned)((nvec + 1023) / 1024)), dim3(256), 0, 0, (const uint4*)b.x[i % NS], sink, nvec);
           }, IT));
    report("read-only<UNR=8>", 1.0 * bytes, time_it([&](int i) {
               hipLaunchKernelGGL((read_kernel<8>), dim3((unsigned)((nvec + 2047) / 2048)), dim3(256), 0, 0, (const uint4*)b.x[i % NS], sink, nvec);
           }, IT));
    report("write-only<UNR=4>", 1.0 * bytes, time_it([&](int i) {
               hipLaunchKernelGGL((write_kernel<4>), dim3((unsigned)((nvec + 1023) / 1024)), dim3(256), 0, 0, (uint4*)b.y[i % NS], nvec);
           }, IT));

    // ---- sym forward variants (need cols/8 <= TPR*VPT)
    const int64_t nv_row = cols / 8;
#define SYM(TPR, VPT, FAST, NTL, NTS)                                                                                   \
    if (nv_row <= (int64_t)TPR * VPT && nv_row > (int64_t)TPR * (VPT - 1))                                              \
        report("sym_fwd<TPR=" #TPR ",VPT=" #VPT ",FAST=" #FAST ",NTL=" #NTL ",NTS=" #NTS ">", 2.0 * bytes,              \
               time_it([&](int i) { launch_sym<TPR, VPT, FAST, NTL, NTS>(b.x[i % NS], b.y[i % NS], b.bounds[i % NS], rows, cols, 4); }, IT));
    SYM(256, 6, true, false, false) SYM(256, 6, true, true, false) SYM(256, 6, true, false, true) SYM(256, 6, true, true, true)
    SYM(512, 3, true, false, false) SYM(512, 3, true, true, true) SYM(1024, 2, true, true, true) SYM(256, 6, false, true, true)
    SYM(256, 2, true, false, false) SYM(256, 2, true, true, true) SYM(256, 2, true, true, false) SYM(256, 2, true, false, true) SYM(128, 4, true, true, true) SYM(64, 8, true, true, true) SYM(512, 1, true, true, true)
    SYM(256, 7, true, true, true) SYM(512, 4, true, true, true)
    // block sizes that are not powers of two: fewer idle vector slots (11008 cols = 1376 vectors: 704 x 2 = 1408 vs 512 x 3 = 1536;
    // 13824 cols = 1728 vectors: 576 x 3 exactly vs 512 x 4 = 2048; 5120 cols = 640 vectors: 320 x 2 exactly vs 256 x 3 = 768)
    SYM(704, 2, true, true, true) SYM(704, 2, true, false, true) SYM(576, 3, true, true, true) SYM(320, 2, true, true, true) SYM(320, 2, true, false, true)
    SYM(256, 3, true, true, true) SYM(256, 3, true, false, true) SYM(896, 2, true, true, true) SYM(448, 4, true, true, true) SYM(640, 3, true, true, true)
    if (argc > 3 && std::string(argv[3]) == "shapes") return 0;

    // ---- full launch-shape sweep: every TPR whose VPT = ceil(nvec/TPR) fits in 1..8
    if (argc > 3) {
#define SWEEP(TPR)                                                                                                         \
    {                                                                                                                      \
        const int vpt = (int)((nv_row + TPR - 1) / TPR);                                                                   \
        if (vpt >= 1 && vpt <= 8) {                                                                                        \
            char nm[96];                                                                                                   \
            snprintf(nm, sizeof nm, "sweep sym_fwd TPR=%d VPT=%d NT/NT", TPR, vpt);                                        \
            report(nm, 2.0 * bytes, time_it([&](int i) { launch_sym_any<TPR, true, true>(vpt, b.x[i % NS], b.y[i % NS], b.bounds[i % NS], rows, cols); }, IT)); \
            snprintf(nm, sizeof nm, "sweep sym_fwd TPR=%d VPT=%d plain/NT", TPR, vpt);                                     \
            report(nm, 2.0 * bytes, time_it([&](int i) { launch_sym_any<TPR, false, true>(vpt, b.x[i % NS], b.y[i % NS], b.bounds[i % NS], rows, cols); }, IT)); \
        }                                                                                                                  \
    }
        SWEEP(64) SWEEP(128) SWEEP(256) SWEEP(512) SWEEP(1024)
        return 0;
    }

    // ---- STE backward variants
#define STE(U, NT)  STE2(U, NT, NT)
#define STE2(U, NTL, NTS)                                                                                                           \
    report("ste_vec<UNR=" #U ",NTL=" #NTL ",NTS=" #NTS ">", 3.0 * bytes, time_it([&](int i) {                                       \
               hipLaunchKernelGGL((ste_vec_kernel<BF16, U, NTL, NTS>), dim3((unsigned)((nvec + STE_THREADS * U - 1) / (STE_THREADS * U))), \
                                  dim3(STE_THREADS), 0, 0, (const uint4*)b.g[i % NS], (const uint4*)b.x[i % NS], (uint4*)b.gx[i % NS], nvec, -2.0f, 2.0f); \
           }, IT));
    STE(1, false) STE(1, true) STE2(1, false, true) STE2(1, true, false) STE(2, true) STE(4, true)
    // row-bounds variant: bounds were written by the sym launches above (weights: every row safe)
    {
        const int64_t chunks = (nv_row + 256 * 8 - 1) / (256 * 8);
        const int cv = (int)((nv_row + chunks - 1) / chunks);
        const int vpt = (cv + 255) / 256;
        printf("ste_rows: chunks=%lld cv=%d vpt=%d\n", (long long)chunks, cv, vpt);
#define STER(V, NT, LO, HI, LABEL)                                                                                                    \
    if (vpt == V)                                                                                                                     \
        report("ste_rows<VPT=" #V ",NT=" #NT "> " LABEL, 3.0 * bytes, time_it([&](int i) {                                            \
                   hipLaunchKernelGGL((ste_rows_kernel<BF16, V, NT, NT>), dim3((unsigned)(rows * chunks)), dim3(STE_THREADS), 0, 0, b.g[i % NS], \
                                      b.x[i % NS], b.gx[i % NS], nv_row, chunks, cv, b.bounds[i % NS], LO, HI, StePitch3{});                       \
               }, IT));
#define STERM(V, LO, HI, LABEL)                                                                                                      \
    if (vpt == V)                                                                                                                     \
        report("ste_rows<VPT=" #V ",NTL=0,NTS=1> " LABEL, 3.0 * bytes, time_it([&](int i) {                                          \
                   hipLaunchKernelGGL((ste_rows_kernel<BF16, V, false, true>), dim3((unsigned)(rows * chunks)), dim3(STE_THREADS), 0, 0, b.g[i % NS], \
                                      b.x[i % NS], b.gx[i % NS], nv_row, chunks, cv, b.bounds[i % NS], LO, HI, StePitch3{});                       \
               }, IT));
        STERM(6, -2.0f, 2.0f, "all rows safe") STERM(6, -1e-3f, 1e-3f, "no row safe") STERM(2, -2.0f, 2.0f, "all rows safe") STERM(2, -1e-3f, 1e-3f, "no row safe")
        STER(6, false, -2.0f, 2.0f, "all rows safe") STER(6, true, -2.0f, 2.0f, "all rows safe")
        STER(6, false, -1e-3f, 1e-3f, "no row safe") STER(6, true, -1e-3f, 1e-3f, "no row safe")
        // forced small chunks: cv = 256 / 512 vectors per block
#define STERC(V, CV, LO, HI, LABEL)                                                                                                   \
    {                                                                                                                                 \
        const int64_t ch2 = (nv_row + CV - 1) / CV;                                                                                   \
        report("ste_rows<VPT=" #V ",NT=1,cv=" #CV "> " LABEL, 3.0 * bytes, time_it([&](int i) {                                        \
                   hipLaunchKernelGGL((ste_rows_kernel<BF16, V, true, true>), dim3((unsigned)(rows * ch2)), dim3(STE_THREADS), 0, 0, b.g[i % NS], \
                                      b.x[i % NS], b.gx[i % NS], nv_row, ch2, CV, b.bounds[i % NS], LO, HI, StePitch3{});                          \
               }, IT));                                                                                                               \
    }
        STERC(1, 256, -1e-3f, 1e-3f, "no row safe") STERC(2, 512, -1e-3f, 1e-3f, "no row safe") STERC(3, 768, -1e-3f, 1e-3f, "no row safe")
        STERC(1, 256, -2.0f, 2.0f, "all rows safe") STERC(2, 512, -2.0f, 2.0f, "all rows safe") STERC(3, 768, -2.0f, 2.0f, "all rows safe")
        STER(2, false, -2.0f, 2.0f, "all rows safe") STER(2, true, -2.0f, 2.0f, "all rows safe")
        STER(2, false, -1e-3f, 1e-3f, "no row safe") STER(2, true, -1e-3f, 1e-3f, "no row safe")
    }
    return 0;
}
