// fq_qlinear_direct.hip -- EXPERIMENT (round 5), NOT PART OF THE PRODUCT LIBRARY.  Second, structurally different attempt at
// SURVEY §8 f4a (fake-quant fused into the GEMM prologue; models/utils_quant.py:250 `F.linear(input_, weight)` fed by :195-201):
//
//     out[tokens, out] = xq[tokens, in] . fq(W)[out, in]^T          bf16 in, fp32 accumulate (MFMA), bf16 out
//
// The round-2 kernel (fq_qlinear.hip) staged BOTH operands global -> VGPR -> LDS and lost to hipBLASLt on the staging alone
// (DESIGN.md §10).  Here the quantized operand never touches LDS:
//   * W is the A operand of v_mfma_f32_32x32x16_bf16, whose fragment is 8 consecutive k of ONE row per lane = 16 contiguous bytes
//     of the K-major [out, in] weight: a lane's global_load_dwordx4 IS its MFMA operand (after the sym_dword chain has run on it in
//     registers when W is quantized on load).  The k index inside a 64-wide sub-tile is re-mapped so that lane half h owns
//     k in [32 h, 32 h + 32): the four loads of a sub-tile cover one full 128-byte line per row pair of lanes.  Each wave owns 32
//     output rows, so every W element is loaded exactly once per block (8 waves = 256 output features per block).
//   * x (already fake-quantized once by the standalone kernel: it is shared by q/k/v and gate/up anyway) rides LDS-DMA
//     (global_load_lds_dwordx4): no VGPR round trip, no ds_write.  The LDS image is lane-linear per wave-instruction (8 rows x 128 B),
//     so the bank-conflict swizzle (16-byte chunk index ^ (row >> 1) & 7) is applied to the SOURCE address and to the ds_read address.
//     The x fragment reads use the same k re-mapping as the W loads.
//   * block = 128 tokens x 256 out, K-step BK (64 or 128), 512 threads = 8 waves laid out 1 x 8 along `out`; each wave 4 MFMA tiles
//     (4 x 32 tokens) = 64 accumulator VGPRs.  256 blocks for down_proj [2048,11008].[4096,11008]^T: one per CU.
//   * three LDS stages and three W register sets; every VMEM operation is inline asm (hipcc counts none of them), one counted
//     s_waitcnt vmcnt(N) + one raw s_barrier per K-step; tile t+2 is issued right after the barrier of step t.
#include "../../include/llmqat_fakequant.h"

#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../llm-qat_amd/csrc/fq_launch.h"

using namespace fq;

namespace {

template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

constexpr int DQ_BM = 128;   // tokens per block
constexpr int DQ_BN = 256;   // output features per block: 8 waves x 32
constexpr int DQ_THREADS = 512;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // a 128-bit register operand (HIP's u32x4 is a struct: no "+v")

struct DQArgs {
    const uint16_t* x;   // [M][K]  (tokens)
    const uint16_t* w;   // [N][K]  (output features)
    uint16_t* out;       // [M][N]
    const float* ws;     // [N][2] = {s, t2}  (QW != 0)
    int M, N, K;
    int tiles_m, tiles_n;
};

// CP: cache policy of the W loads -- 0 plain, 1 `nt`, 2 `sc1` (the latter two do not allocate in the CU's 32 KiB vector L1)
template <int OFF, int CP> __device__ __forceinline__ void gload16(u32x4& dst, const char* p) {
    if constexpr (CP == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
    else if constexpr (CP == 2) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
// LDS-DMA: 64 lanes x 16 bytes -> LDS [lds_dst, lds_dst + 1024), lane-linear.  M0 is written in the statement that reads it.
__device__ __forceinline__ void glds16(const char* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// the x fragment reads are inline asm too: hipcc's scheduler otherwise sinks every ds_read next to its MFMA (lgkmcnt(0) before each
// one) whatever order the source gives them.  Volatile asm statements keep their order; the wait names the registers it makes valid
// ("+v"), so no MFMA that consumes them can be scheduled above it (cdna guide 5.7, form ii).
template <int OFF> __device__ __forceinline__ void lds_read16(u32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// QW: 0 = W multiplied as it is, 1 = SymQuantizer arithmetic in bf16 (fq_sym_fwd), 2 = its autocast arithmetic rounded once to bf16
template <int QW> __device__ __forceinline__ u32x4 quant_w(u32x4 v, const SymRow& r) {
    if constexpr (QW == 0) {
        return v;
    } else if constexpr (QW == 1) {
        return u32x4{sym_dword<BF16, true>(v.x, r, nullptr), sym_dword<BF16, true>(v.y, r, nullptr), sym_dword<BF16, true>(v.z, r, nullptr),
                          sym_dword<BF16, true>(v.w, r, nullptr)};
    } else {
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
        return u32x4{o[0], o[1], o[2], o[3]};
    }
}

// One K-step of one wave: (optionally) fake-quantize the W fragments in place, then NS = BK / 16 k-steps of 4 MFMAs; the x fragments
// (4 token tiles = 4 ds_read_b128 per k-step) run PD k-steps ahead of the MFMAs that consume them; LDS returns in order.
// la[s]: this lane's LDS byte address (stage included) of token-tile 0, sub-tile 0, for step s of a sub-tile.
template <int BK, int QW, int ABL, int PD>
__device__ __forceinline__ void dq_compute(u32x4 (&wd)[BK / 16], f32x16_t (&acc)[4], const uint32_t (&la)[4], const SymRow& wq) {
    constexpr int NS = BK / 16;
    if constexpr (QW != 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) wd[i] = quant_w<QW>(wd[i], wq);
    }
    u32x4 xb[PD + 1][4];
    auto reads = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        constexpr int u = ks >> 2, s = ks & 3, o = u * (DQ_BM * 128);
        lds_read16<o>(xb[ks % (PD + 1)][0], la[s]);
        lds_read16<o + 4096>(xb[ks % (PD + 1)][1], la[s]);
        lds_read16<o + 8192>(xb[ks % (PD + 1)][2], la[s]);
        lds_read16<o + 12288>(xb[ks % (PD + 1)][3], la[s]);
    };
    auto body = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        if constexpr (ks + PD < NS) reads(std::integral_constant<int, ks + PD>{});
        constexpr int ahead = (NS - 1 - ks) < PD ? (NS - 1 - ks) : PD;   // k-steps issued after this one
        u32x4(&xc)[4] = xb[ks % (PD + 1)];
        const u32x4 wk = wd[ks];   // (named outside the asm: clang does not capture a variable first used as an asm operand of a generic lambda)
        wait_lgkm<4 * ahead>();
        __builtin_amdgcn_sched_barrier(0);   // nothing moves across: hipcc otherwise sinks the MFMAs into one cluster, every fragment stays live, spills
        if constexpr (ABL == 1) {
            // (whole registers: with only one component named, hipcc treats the other three as dead, overlaps the tuples and hands
            // them to other values -- address registers among them -- while the loads into them are still in flight: round 5's first
            // run of this ablation ended in a memory access fault)
            asm volatile("" ::"v"(xc[0]), "v"(xc[1]), "v"(xc[2]), "v"(xc[3]), "v"(wk));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wk), __builtin_bit_cast(bf16x8_t, xc[j]), acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    static_for<0, (PD < NS ? PD : NS)>(reads);
    static_for<0, NS>(body);
}

// The same K-step on v_mfma_f32_16x16x32_bf16 (SH = 16): the wave's 32 output rows are two A fragments (rows 16 a + lane % 16), the
// 128 tokens eight B fragments; lane group g = lane / 16 holds k = 32 ks + 8 g .. + 8 of k-step ks, so ONE W load instruction covers
// 16 rows x 64 contiguous bytes (the 32x32x16 form: 32 rows x 2 x 16 bytes: twice the cache lines per instruction for the same bytes).
// wd[2 ks + a]: W fragment of k-step ks, row fragment a.  A read group = 4 token fragments of one k-step (ks, jh) -> 8 MFMAs.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int BK, int QW, int ABL, int PD>
__device__ __forceinline__ void dq_compute16(u32x4 (&wd)[BK / 16], f32x4_t (&acc)[16], const uint32_t (&la)[2], const SymRow (&wq)[2]) {
    constexpr int NS = BK / 16;   // read groups per K-step: 4 per 64-k sub-tile
    if constexpr (QW != 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) wd[i] = quant_w<QW>(wd[i], wq[i & 1]);
    }
    u32x4 xb[PD + 1][4];
    auto reads = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int u = q >> 2, ks = (q >> 1) & 1, jh = q & 1, o = u * (DQ_BM * 128) + jh * 4 * 2048;
        lds_read16<o>(xb[q % (PD + 1)][0], la[ks]);
        lds_read16<o + 2048>(xb[q % (PD + 1)][1], la[ks]);
        lds_read16<o + 4096>(xb[q % (PD + 1)][2], la[ks]);
        lds_read16<o + 6144>(xb[q % (PD + 1)][3], la[ks]);
    };
    auto body = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int u = q >> 2, ks = (q >> 1) & 1, jh = q & 1;
        if constexpr (q + PD < NS) reads(std::integral_constant<int, q + PD>{});
        constexpr int ahead = (NS - 1 - q) < PD ? (NS - 1 - q) : PD;
        u32x4(&xc)[4] = xb[q % (PD + 1)];
        const u32x4 wa = wd[4 * u + 2 * ks], wb = wd[4 * u + 2 * ks + 1];
        wait_lgkm<4 * ahead>();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL == 1) {
            asm volatile("" ::"v"(xc[0]), "v"(xc[1]), "v"(xc[2]), "v"(xc[3]), "v"(wa), "v"(wb));
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc[4 * jh + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xc[jj]), acc[4 * jh + jj], 0, 0, 0);
                acc[8 + 4 * jh + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb), __builtin_bit_cast(bf16x8_t, xc[jj]), acc[8 + 4 * jh + jj], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    static_for<0, (PD < NS ? PD : NS)>(reads);
    static_for<0, NS>(body);
}

// ABL (timing builds, results are garbage): 0 = the kernel; 1 = no MFMA (loads + LDS reads only); 2 = no W loads (x path + MFMA);
// 3 = no x LDS-DMA (W loads + LDS reads of stale data + MFMA)
// SH: 32 = v_mfma_f32_32x32x16_bf16 (lane half h owns 64 contiguous bytes of its row per sub-tile), 16 = v_mfma_f32_16x16x32_bf16
template <int SH, int BK, int QW, int ABL, int CP = 0> __global__ __launch_bounds__(DQ_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void qdirect_kernel(DQArgs a) {
    constexpr int SUB = BK / 64;             // 64-k sub-tiles per K-step: each an [128 tokens][128 B] LDS image
    constexpr int STAGE = DQ_BM * BK * 2;    // bytes of one LDS stage
    constexpr int NW = 4 * SUB;              // W loads per lane per K-step
    constexpr int NX = 2 * SUB;              // LDS-DMA instructions per wave per K-step (16 * SUB per block)
    constexpr int OPS = (ABL == 2 ? 0 : NW) + (ABL == 3 ? 0 : NX);
    constexpr int PD = 2;                    // x fragment prefetch depth in k-steps
    __shared__ __attribute__((aligned(1024))) char smem[3 * STAGE];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    // tile of this block.  Blocks b, b + 8, ... share an XCD (and its L2): give each XCD a compact 8 (token tiles) x 4 (out tiles)
    // patch when the grid allows it -- 8 x 2.8 MB of x panels + 4 x 5.6 MB of W panels per XCD at K = 11008.
    int tm, tn;
    if ((a.tiles_m & 7) == 0 && (a.tiles_n & 3) == 0 && (((a.tiles_m >> 3) * (a.tiles_n >> 2)) & 7) == 0) {
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;          // l: index among this XCD's blocks
        const int per_patch = 32, patch = (l / per_patch) * 8 + xcd;  // patches are dealt round-robin to the XCDs
        const int li = l % per_patch;
        const int pm = a.tiles_m >> 3;
        tm = (patch % pm) * 8 + (li & 7);
        tn = (patch / pm) * 4 + (li >> 3);
    } else {
        tm = blockIdx.x % a.tiles_m, tn = blockIdx.x / a.tiles_m;
    }
    const int m0 = tm * DQ_BM, n0 = tn * DQ_BN;
    const int K = a.K;

    // ---- W: SH 32: this lane's row, its 64-byte half of every 128-byte sub-tile segment; SH 16: two rows (fragments a = 0, 1), 16 bytes
    //      at 16 g of each 64-byte k-step
    constexpr int NR = SH == 32 ? 1 : 2;
    const int lc = lane & 15, lg = lane >> 4;
    const char* wsrc[NR];
    SymRow wq[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        int wrow = n0 + wave * 32 + (SH == 32 ? fr : 16 * r + lc);
        wrow = wrow < a.N ? wrow : a.N - 1;   // clamp: tail rows are computed on a copy of the last row and never stored
        wsrc[r] = (const char*)a.w + ((int64_t)wrow * K) * 2 + (SH == 32 ? 64 * fh : 16 * lg);
        wq[r] = SymRow{};
        if constexpr (QW != 0) {
            wq[r].s = a.ws[2 * wrow];
            wq[r].t2 = a.ws[2 * wrow + 1];
            wq[r].rinv = 1.0f / wq[r].t2;
            wq[r].mk = true;
        }
    }
    // ---- x by LDS-DMA: wave-instruction qq = wave + 8 i covers rows 8 (qq % 16) .. + 7 of sub-tile qq / 16; lane -> row, physical chunk
    const char* xsrc[NX];
    uint32_t xdst[NX];
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int qq = wave + 8 * i, u = qq >> 4, rg = qq & 15;
        const int row = 8 * rg + (lane >> 3), pc = lane & 7;
        const int c = (pc ^ (row >> 1)) & 7;   // the logical chunk that lives at physical chunk pc of this row
        int xr = m0 + row;
        xr = xr < a.M ? xr : a.M - 1;
        xsrc[i] = (const char*)a.x + ((int64_t)xr * K) * 2 + 128 * u + 16 * c;
        xdst[i] = lds0 + u * (DQ_BM * 128) + rg * 1024;
    }
    // ---- x fragment reads.  SH 32: lane (fr, fh), token tile j, sub-tile u, step s: row 32 j + fr, logical chunk 4 fh + s;
    //      SH 16: lane (lc, lg), token tile j, k-step ks: row 16 j + lc, logical chunk 4 ks + lg.  (row >> 1) & 7 does not depend on j.
    constexpr int NCO = SH == 32 ? 4 : 2;
    int co[NCO];
#pragma unroll
    for (int s = 0; s < NCO; ++s)
        co[s] = SH == 32 ? fr * 128 + ((((4 * fh + s) ^ (fr >> 1)) & 7) << 4) : lc * 128 + ((((4 * s + lg) ^ (lc >> 1)) & 7) << 4);

    using acc_t = std::conditional_t<SH == 32, f32x16_t, f32x4_t>;
    constexpr int NACC = SH == 32 ? 4 : 16, NEL = SH == 32 ? 16 : 4;
    acc_t acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int e = 0; e < NEL; ++e) acc[j][e] = 0.f;
    u32x4 w0[NW], w1[NW], w2[NW];

    const int nk = K / BK;
    auto issue = [&](u32x4(&wd)[NW], int stage, int tile) {
        const int64_t koff = (int64_t)tile * (BK * 2);
        if constexpr (ABL != 2) {
#pragma unroll
            for (int u = 0; u < SUB; ++u) {
                if constexpr (SH == 32) {
                    const char* p = wsrc[0] + koff + 128 * u;
                    gload16<0, CP>(wd[4 * u + 0], p);
                    gload16<16, CP>(wd[4 * u + 1], p);
                    gload16<32, CP>(wd[4 * u + 2], p);
                    gload16<48, CP>(wd[4 * u + 3], p);
                } else {
                    const char* pa = wsrc[0] + koff + 128 * u;
                    const char* pb = wsrc[NR - 1] + koff + 128 * u;
                    gload16<0, CP>(wd[4 * u + 0], pa);    // k-step 0, rows 0..15
                    gload16<0, CP>(wd[4 * u + 1], pb);    //           rows 16..31
                    gload16<64, CP>(wd[4 * u + 2], pa);   // k-step 1
                    gload16<64, CP>(wd[4 * u + 3], pb);
                }
            }
        }
        if constexpr (ABL != 3) {
#pragma unroll
            for (int i = 0; i < NX; ++i) glds16(xsrc[i] + koff, xdst[i] + stage * STAGE);
        }
    };
    auto compute = [&](u32x4(&wd)[NW], int stage) {
        uint32_t la[NCO];
#pragma unroll
        for (int s = 0; s < NCO; ++s) la[s] = lds0 + (uint32_t)(stage * STAGE) + (uint32_t)co[s];
        if constexpr (SH == 32) dq_compute<BK, QW, ABL, PD>(wd, acc, la, wq[0]);
        else dq_compute16<BK, QW, ABL, PD>(wd, acc, la, wq);
    };
    // one K-step: tile t is complete once only the OPS operations of tile t + 1 are still in flight; the barrier makes every wave's
    // LDS-DMA of tile t visible and proves that every wave has finished reading the stage tile t + 2 is about to overwrite
    // (it held tile t - 1).  Past the last tile the prefetch re-loads tile nk - 1 into stages / registers nobody reads again, so the
    // wait count is the same constant in every step.
#define DQ_STEP(WCUR, SCUR, WNXT, SNXT, T)                \
    {                                                     \
        wait_vm<OPS>();                                   \
        __builtin_amdgcn_s_barrier();                     \
        __builtin_amdgcn_sched_barrier(0);                \
        issue(WNXT, SNXT, (T) + 2 < nk ? (T) + 2 : nk - 1); \
        compute(WCUR, SCUR);                              \
    }
    issue(w0, 0, 0);
    issue(w1, 1, nk > 1 ? 1 : 0);
    const int nfull = nk / 3 * 3;   // whole triples in a loop without exits, the remaining one or two steps behind it
    int kt = 0;
    for (; kt < nfull; kt += 3) {
        DQ_STEP(w0, 0, w2, 2, kt)
        DQ_STEP(w1, 1, w0, 0, kt + 1)
        DQ_STEP(w2, 2, w1, 1, kt + 2)
    }
    if (kt < nk) {
        DQ_STEP(w0, 0, w2, 2, kt)
        if (kt + 1 < nk) DQ_STEP(w1, 1, w0, 0, kt + 1)
    }
#undef DQ_STEP
    wait_vm<0>();   // the surplus prefetches: no LDS-DMA may still be in flight when the block's LDS is handed on
    // ... and their register destinations stay allocated until here: to hipcc an asm output nobody reads is dead at once, and it would
    // hand the register to another value while the load is still in flight
#pragma unroll
    for (int i = 0; i < NW; ++i) asm volatile("" ::"v"(w0[i]), "v"(w1[i]), "v"(w2[i]));

    // ---- epilogue.  SH 32: acc[j][reg]: out column n = n0 + wave*32 + 8*(reg>>2) + 4*fh + (reg&3), token m = m0 + 32 j + fr
    //      SH 16: acc[8 a + j][reg]: n = n0 + wave*32 + 16 a + 4 lg + reg, m = m0 + 16 j + lc
    if constexpr (SH == 32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + 32 * j + fr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wave * 32 + 8 * g + 4 * fh;
                const float lo2[2] = {acc[j][4 * g], acc[j][4 * g + 1]}, hi2[2] = {acc[j][4 * g + 2], acc[j][4 * g + 3]};
                const uint2 o = make_uint2(Ty<BF16>::pack(lo2), Ty<BF16>::pack(hi2));
                if (m < a.M && n < a.N) *(uint2*)(a.out + (int64_t)m * a.N + n) = o;   // N % 4 == 0 (host-checked)
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = m0 + 16 * j + lc;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int n = n0 + wave * 32 + 16 * r + 4 * lg;
                const float lo2[2] = {acc[8 * r + j][0], acc[8 * r + j][1]}, hi2[2] = {acc[8 * r + j][2], acc[8 * r + j][3]};
                const uint2 o = make_uint2(Ty<BF16>::pack(lo2), Ty<BF16>::pack(hi2));
                if (m < a.M && n < a.N) *(uint2*)(a.out + (int64_t)m * a.N + n) = o;
            }
        }
    }
}

template <int SH, int BK, int QW> int launch_dq(const DQArgs& a, int ablation, hipStream_t st) {
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(DQ_THREADS);
    if (ablation == 1) FQ_LAUNCHK((qdirect_kernel<SH, BK, QW, 1>), grid, block, 0, st, a);
    else if (ablation == 2) FQ_LAUNCHK((qdirect_kernel<SH, BK, QW, 2>), grid, block, 0, st, a);
    else if (ablation == 3) FQ_LAUNCHK((qdirect_kernel<SH, BK, QW, 3>), grid, block, 0, st, a);
    else if (ablation == 10) FQ_LAUNCHK((qdirect_kernel<SH, BK, QW, 0, 1>), grid, block, 0, st, a);   // W loads `nt`
    else if (ablation == 20) FQ_LAUNCHK((qdirect_kernel<SH, BK, QW, 0, 2>), grid, block, 0, st, a);   // W loads `sc1`
    else if (ablation == 11) FQ_LAUNCHK((qdirect_kernel<SH, BK, QW, 1, 1>), grid, block, 0, st, a);   // no MFMA, W loads `nt`
    else if (ablation == 21) FQ_LAUNCHK((qdirect_kernel<SH, BK, QW, 1, 2>), grid, block, 0, st, a);   // no MFMA, W loads `sc1`
    else FQ_LAUNCHK((qdirect_kernel<SH, BK, QW, 0>), grid, block, 0, st, a);
    return launch_result();
}

}  // namespace

#define FQ_API __attribute__((visibility("default")))

extern "C" {

/*
 * out[tokens, out] = x[tokens, in] . fq(W)[out, in]^T   (models/utils_quant.py:195-201, :250), no backward.  x is multiplied as it is
 * (the caller fake-quantizes it once with the product's fq_sym_fwd); W is fake-quantized in registers on its way into the MFMA
 * when w_scales (per-row {s, t2} of fq_sym_row_scales) is given.
 *   bk        K per step: 64 or 128;  mfma  32 | 16: the MFMA shape (32x32x16 | 16x16x32);  autocast  1: the W values follow fq_sym_fwd_autocast (rounded once to bf16)
 *   ablation  0; 1..3 are timing builds whose result is garbage (1: no MFMA, 2: no W loads, 3: no x LDS-DMA);
 *             10 / 20: the kernel with `nt` / `sc1` W loads (correct results), 11 / 21: those without MFMA
 * FQ_ERR_UNSUPPORTED unless in_features % bk == 0, out_features % 4 == 0, x / w 16-byte and out 8-byte aligned.
 */
FQ_API int fq_qlinear_direct_fwd(const void* x, const void* w, const float* w_scales, void* out, int64_t tokens, int64_t in_features,
                                 int64_t out_features, int bk, int mfma, int autocast, int ablation, void* stream) {
    if (tokens < 0 || in_features < 0 || out_features < 0) return fail(FQ_ERR_SHAPE, "negative shape");
    if (tokens == 0 || out_features == 0) return ok();
    if (!x || !w || !out) return fail(FQ_ERR_NULL, "x / w / out must not be NULL");
    if (bk != 64 && bk != 128) return fail(FQ_ERR_ARG, "bk must be 64 or 128");
    if (mfma != 32 && mfma != 16) return fail(FQ_ERR_ARG, "mfma must be 32 (v_mfma_f32_32x32x16_bf16) or 16 (v_mfma_f32_16x16x32_bf16)");
    if (in_features < 3 * bk || in_features % bk) return fail(FQ_ERR_UNSUPPORTED, "in_features=%lld must be a multiple of %d and at least 3 K-steps", (long long)in_features, bk);
    if (out_features % 4) return fail(FQ_ERR_UNSUPPORTED, "out_features=%lld must be a multiple of 4", (long long)out_features);
    if (!aligned16(x) || !aligned16(w) || (reinterpret_cast<uintptr_t>(out) & 7u)) return fail(FQ_ERR_UNSUPPORTED, "x / w must be 16-byte, out 8-byte aligned");
    if (tokens > 0x7FFFFFFF || out_features > 0x7FFFFFFF || in_features > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "dimension exceeds int32");
    if (!((ablation >= 0 && ablation <= 3) || ablation == 10 || ablation == 20 || ablation == 11 || ablation == 21)) return fail(FQ_ERR_ARG, "ablation must be 0..3, 10, 11, 20 or 21");
    DQArgs a{};
    a.x = (const uint16_t*)x;
    a.w = (const uint16_t*)w;
    a.out = (uint16_t*)out;
    a.ws = w_scales;
    a.M = (int)tokens, a.N = (int)out_features, a.K = (int)in_features;
    a.tiles_m = (int)((tokens + DQ_BM - 1) / DQ_BM);
    a.tiles_n = (int)((out_features + DQ_BN - 1) / DQ_BN);
    if ((int64_t)a.tiles_m * a.tiles_n > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "too many tiles");
    begin_launches();
    hipStream_t st = (hipStream_t)stream;
    const int qw = w_scales ? (autocast ? 2 : 1) : 0;
#define DQ_CASE(S, B, Q) \
    if (mfma == S && bk == B && qw == Q) return launch_dq<S, B, Q>(a, ablation, st);
    DQ_CASE(32, 64, 0) DQ_CASE(32, 64, 1) DQ_CASE(32, 64, 2) DQ_CASE(32, 128, 0) DQ_CASE(32, 128, 1)
    DQ_CASE(16, 64, 0) DQ_CASE(16, 64, 1) DQ_CASE(16, 64, 2) DQ_CASE(16, 128, 0) DQ_CASE(16, 128, 1)
#undef DQ_CASE
    return fail(FQ_ERR_ARG, "unsupported combination");
}

}  // extern "C"
