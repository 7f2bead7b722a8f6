// fq_export.h -- the integer side of the fake-quant forward: per-row scale terms and, on request, the
// packed integer bins the reference rounds to (models/utils_quant.py:71-72  `torch.round(input * s)`; AsymQuantizer
// :144-146), for inference export and as the pre-pass of the fused QuantizeLinear GEMM.
//
//   row_export_kernel          register-resident rows (the row_reg_kernel data flow without the dequantized store):
//                              read x once (2 B/elem bf16), write bins (1 or 0.5 B/elem, non-temporal) + 8 B/row of
//                              scales; container = NONE makes it the scale pre-pass (`fq_sym_row_scales`): 2 B/elem read only.
//   row_export_generic_kernel  any width / alignment (element loads, two sweeps), correctness path.
//
// Containers saturate; overflow[row] counts the elements that did not fit (or were NaN).  The reference has no clamp, so
// in bf16 an 8-bit row can hold the bin +128 (and a 16-bit row bins beyond 32767); the count makes that explicit instead of
// silently changing a value.  A row whose top bin fits skips the counting entirely (the bin is monotone in |x| / in x).
#pragma once
#include <type_traits>

#include "fq_kernels.h"

namespace fq {

enum : int { BINS_NONE = 0, BINS_INT4 = 1, BINS_INT8 = 2, BINS_INT16 = 3 };

struct ExportArgs {
    const void* x;
    void* bins;          // packed bins, row stride = export_row_bytes(cols, container); NULL with BINS_NONE
    float* scales;       // [rows][2]: Sym {s, t2 = s + 1e-6};  Asym {alpha + 1e-8, beta}
    int32_t* overflow;   // [rows] elements saturated by the container (optional)
    float* bounds;       // optional [rows][2], as row_reg_kernel
    uint64_t* mask;      // optional STE bit mask (the row bitmap of fq_kernels.h)
    int64_t mask_row_words;
    float lo, hi;
    uint32_t clipk;      // integer form of the clip (16-bit tensors), as RowArgs::clipk
    int64_t rows, cols;
    int64_t row_bytes;   // bins row stride in bytes
    SymConst sym;
    AsymConst asym;
    int container;
    int autocast;        // Sym on 16-bit tensors: the reference's arithmetic under CUDA autocast (fp32 behind the reciprocal)
    float cmin, cmax;    // container range as floats: Sym signed [-2^(b-1), 2^(b-1)-1], Asym unsigned [0, 2^b - 1]
};

__host__ __device__ inline int64_t export_row_bytes(int64_t cols, int container) {
    return container == BINS_INT4 ? (cols + 1) / 2 : container == BINS_INT8 ? cols : container == BINS_INT16 ? cols * 2 : 0;
}

// bin value (an integer-valued float, possibly NaN / Inf) -> saturated integer; `bad` = it did not fit
__device__ __forceinline__ int sat_bin(float q, float cmin, float cmax, bool& bad) {
    bad = !(q >= cmin && q <= cmax);  // true for NaN
    if (q != q) return 0;
    return (int)__builtin_fminf(__builtin_fmaxf(q, cmin), cmax);
}

// the row's top bin: Sym rint(rb(m * s)) for m = max|x| (the chain is odd and monotone), Asym the bin of the max element
template <int DT> __device__ __forceinline__ float sym_top_bin(float m, const SymRow& r, bool autocast) {
    return autocast ? __builtin_rintf(m * r.s) : __builtin_rintf(Ty<DT>::rb(m * r.s));
}
template <int DT> __device__ __forceinline__ float asym_bin(float x, const AsymRow& r, const AsymConst& k) {
    using T = Ty<DT>;
    const float d = T::rb(x - r.mn);
    const float n = T::rb(r.mk ? div_exact(d, r.a, r.ra) : d / r.a);
    return __builtin_rintf(T::rb(n * k.S));
}
template <int DT> __device__ __forceinline__ float sym_bin(float x, const SymRow& r, bool autocast) {
    return autocast ? __builtin_rintf(x * r.s) : __builtin_rintf(Ty<DT>::rb(x * r.s));
}

// The elementwise half of row_export_kernel: STE mask (scale pre-pass only), bins, packing, stores; returns this lane's count of bins
// that did not fit.  CONT / AC / MASK >= 0 fix the container, the autocast arithmetic and "no mask" at COMPILE time, -1 reads them
// from the arguments.  They are the same for every block of a launch, but as run-time values they cost the 512 x 3 kernel 7 % on the
// metric tensor (a dozen scalar branches per vector slot: profiles/r03_ab_export_constexpr_flags.txt), so the kernel calls the
// specialised bodies for the deployment formats -- Sym, no autocast, int4 / int8 -- and the general one for everything else.
template <int DT, int TPR, int VPT, bool ASYM, int CONT, int AC, int MASK>
__device__ __forceinline__ uint32_t export_row_body(const ExportArgs& a, const uint4 (&r)[VPT], int64_t row, int t, int nvec, const SymRow& sr,
                                                    const AsymRow& ar, float ub, float lb, float top, bool& count) {
    using T = Ty<DT>;
    constexpr int EPV = 16 / T::ESIZE;
    const bool ac = AC < 0 ? a.autocast != 0 : AC != 0;
    const bool want_mask = MASK == 0 ? false : (a.mask && !((ub < a.hi) && (lb > a.lo)));  // block-uniform
    const bool sym_clip = a.lo == -a.hi;
    const uint32_t clipk = (ub != ub) ? 0u : a.clipk;
    uint8_t* mrow = (uint8_t*)(a.mask + row * a.mask_row_words);
    const int cont = CONT < 0 ? a.container : CONT;
    // Sym: only the positive side can exceed a signed container whose top bin is cmax + 1 (-128 fits int8, +128 does not)
    count = cont != BINS_NONE && !(top <= a.cmax);  // block-uniform; true for a NaN row
    if (cont == BINS_NONE && !want_mask) return 0u;  // the scale pre-pass: nothing elementwise to do
    // How a row's bins become integers (block-uniform):
    //   0  the row's top bin fits the container, so every bin does and none is NaN: the integer comes out of ONE add --
    //      p + 1.5 * 2^23 rounds p to an integer (half to even, like torch.round) and leaves its two's complement in the low
    //      mantissa bits (|p| < 2^22 holds for every container);
    //   1  the top bin is finite but does not fit (a bf16 8-bit row whose top bin is +128, 16-bit bins into int8, ...): p is an
    //      ordinary number everywhere in the row, so v_med3_f32 clamps the sum behind the same add and a compare counts what moved;
    //   2  NaN / Inf / huge rows: saturate and count element by element (sat_bin).
    const int row_mode = !count ? 0 : (top < 4194304.0f ? 1 : 2);
    // int4: nibbles are built in offset binary (bin - cmin in 0..15, so neighbours cannot borrow from each other) with shift-adds
    // on the raw sums and flipped to two's complement by one XOR per stored dword.  The offset rides in the rounding constant (an
    // even integer: ties still go to even).
    const int ibias = (cont == BINS_INT4 && !ASYM) ? 8 : 0;  // = -cmin: Sym containers are signed, Asym ones start at 0 (export_entry)
    const float magic = 12582912.0f + (float)ibias;
    const uint32_t flip = ibias ? 0x88888888u : 0u;
    constexpr uint32_t MAGIC_U = 0x4B400000u;  // bits of 1.5 * 2^23
    char* brow = (char*)a.bins + row * a.row_bytes;
    uint32_t nbad = 0;  // per lane; summed over the wave after the loop
    uint32_t pk[VPT][4] = {};  // this lane's packed bins per slot (EPV * container bits / 32 dwords used)
    // The slot loop, once per way of a row (the specialised bodies) or with the way as a run-time value (the general body): the
    // way is block-uniform, so the specialised bodies branch on it once instead of two or three times per slot.
    auto slots = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const int mode = MODE < 0 ? row_mode : MODE;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = t + i * TPR;
            const uint32_t w[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
            float f[EPV];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float fd[T::EPD];
                T::unpack(w[d], fd);
#pragma unroll
                for (int k = 0; k < T::EPD; ++k) f[d * T::EPD + k] = fd[k];
            }
            if (want_mask) ste_mask_record<DT>(mrow, v, v < nvec, r[i], f, a.lo, a.hi, sym_clip, clipk);
            if (cont == BINS_NONE) continue;
            uint32_t qu[EPV];  // MAGIC_U + ibias + bin
            if (mode < 2) {
                // p = what torch.round() sees, dword by dword so that the multiplies, roundings and the final add pair up
                // (v_pk_mul_f32, ONE v_cvt_pk_bf16_f32 per pair, v_pk_add_f32); the autocast switch is taken once per slot
                if constexpr (ASYM) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float g[T::EPD];
#pragma unroll
                        for (int k = 0; k < T::EPD; ++k) g[k] = f[d * T::EPD + k] - ar.mn;
                        T::round_dt(g);
#pragma unroll
                        for (int k = 0; k < T::EPD; ++k) g[k] = ar.mk ? div_exact(g[k], ar.a, ar.ra) : g[k] / ar.a;
                        T::round_dt(g);
#pragma unroll
                        for (int k = 0; k < T::EPD; ++k) g[k] = g[k] * a.asym.S;
                        T::round_dt(g);
#pragma unroll
                        for (int k = 0; k < T::EPD; ++k) f[d * T::EPD + k] = g[k];
                    }
                } else if (ac) {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) f[e] = f[e] * sr.s;
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float g[T::EPD];
#pragma unroll
                        for (int k = 0; k < T::EPD; ++k) g[k] = f[d * T::EPD + k] * sr.s;
                        T::round_dt(g);
#pragma unroll
                        for (int k = 0; k < T::EPD; ++k) f[d * T::EPD + k] = g[k];
                    }
                }
#pragma unroll
                for (int e = 0; e < EPV; ++e) f[e] = f[e] + magic;  // = magic + bin, exactly
                if (mode == 1) {  // clamp the ROUNDED value (7.3 fits int4, 7.6 does not) -- integers below 2^24 compare exactly as floats
                    const float rlo = magic + a.cmin, rhi = magic + a.cmax;
                    uint32_t moved = 0;
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        const float c = __builtin_amdgcn_fmed3f(f[e], rlo, rhi);
                        moved += (c != f[e]) ? 1u : 0u;
                        f[e] = c;
                    }
                    nbad += v < nvec ? moved : 0u;
                }
#pragma unroll
                for (int e = 0; e < EPV; ++e) qu[e] = as_u(f[e]);
            } else {
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float b = ASYM ? asym_bin<DT>(f[e], ar, a.asym) : sym_bin<DT>(f[e], sr, ac);
                    bool bad;
                    qu[e] = MAGIC_U + (uint32_t)(sat_bin(b, a.cmin, a.cmax, bad) + ibias);
                    nbad += (bad && v < nvec) ? 1u : 0u;  // (a ballot + popcount per element made these rows ~10x slower than the others)
                }
            }
            // pack this slot's bins; the stores follow the loop
            if (cont == BINS_INT8) {
#pragma unroll
                for (int d = 0; d < EPV / 4; ++d) {  // v_perm_b32: low bytes of four dwords into one
                    const uint32_t lo = __builtin_amdgcn_perm(qu[4 * d + 1], qu[4 * d], 0x0c0c0400u);
                    const uint32_t hi = __builtin_amdgcn_perm(qu[4 * d + 3], qu[4 * d + 2], 0x04000c0cu);
                    pk[i][d] = lo | hi;
                }
            } else if (cont == BINS_INT4) {
                // (hi << 4) + lo per pair: the low byte holds two offset nibbles, everything above it is the constant's junk;
                // (t1 << 8) + t0 puts two such bytes into a clean low half
                uint32_t h[EPV / 4];
#pragma unroll
                for (int d = 0; d < EPV / 4; ++d) {
                    const uint32_t t0 = (qu[4 * d + 1] << 4) + qu[4 * d];
                    const uint32_t t1 = (qu[4 * d + 3] << 4) + qu[4 * d + 2];
                    h[d] = (t1 << 8) + t0;
                }
                if constexpr (EPV == 8) pk[i][0] = __builtin_amdgcn_perm(h[1], h[0], 0x05040100u) ^ flip;
                else pk[i][0] = (h[0] ^ flip) & 0xFFFFu;
            } else {  // BINS_INT16
#pragma unroll
                for (int d = 0; d < EPV / 2; ++d) pk[i][d] = __builtin_amdgcn_perm(qu[2 * d + 1], qu[2 * d], 0x05040100u);
            }
        }
    };
    if constexpr (CONT >= 0) {
        if (row_mode == 0) slots(std::integral_constant<int, 0>{});
        else if (row_mode == 1) slots(std::integral_constant<int, 1>{});
        else slots(std::integral_constant<int, 2>{});
    } else {
        slots(std::integral_constant<int, -1>{});
    }
    // ---- stores.  A lane's packed vector is EPV * bits / 8 bytes: 16 only for 16-bit elements into int16, else 8, 4 or 2.  Round 3
    // built the alternative -- lane pairs / quads trade packed dwords over two / four slots so that every storing lane writes 16
    // bytes -- and measured it against these plain stores on one box (profiles/r03_ab_export_store_width.txt): int4 22.22 vs 22.23 us,
    // int8 25.1-25.6 vs 25.7-27.4 us.  The store width is not what keeps these kernels above their streaming ceilings, so the simple
    // form stays.
    if (cont != BINS_NONE) {
        const int vb = cont == BINS_INT4 ? EPV / 2 : cont == BINS_INT8 ? EPV : 2 * EPV;  // bytes per vector
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = t + i * TPR;
            if (v >= nvec) continue;
            if (vb == 16) st16<true>((uint4*)brow + v, make_uint4(pk[i][0], pk[i][1], pk[i][2], pk[i][3]));
            else if (vb == 8) st8<true>((uint2*)brow + v, make_uint2(pk[i][0], pk[i][1]));
            else if (vb == 4) __builtin_nontemporal_store(pk[i][0], (uint32_t*)brow + v);
            else __builtin_nontemporal_store((uint16_t)pk[i][0], (uint16_t*)brow + v);
        }
    }
    return nbad;
}

template <int DT, int TPR, int VPT, bool ASYM, bool NTL>
__global__ __launch_bounds__(TPR == 64 ? 256 : TPR) void row_export_kernel(ExportArgs a) {
    using T = Ty<DT>;
    constexpr int EPV = 16 / T::ESIZE;
    constexpr int NW = TPR / 64;
    __shared__ uint32_t red[3][NW > 1 ? NW : 1];
    __shared__ uint32_t cnt_lds[NW > 1 ? NW : 1];

    int64_t row;
    int t;
    if constexpr (TPR == 64) {
        row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        t = threadIdx.x & 63;
        if (row >= a.rows) return;  // wave-uniform; NW == 1: no barrier below
    } else {
        row = blockIdx.x;
        t = threadIdx.x;
    }
    const int nvec = (int)(a.cols / EPV);
    const uint4* __restrict__ xr = (const uint4*)((const char*)a.x + row * a.cols * T::ESIZE);
    uint4 r[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        int v = t + i * TPR;
        v = v < nvec ? v : nvec - 1;
        r[i] = ld16<NTL>(&xr[v]);
    }

    SymRow sr;
    AsymRow ar;
    float ub, lb, top;
    const bool ac = a.autocast != 0;
    if constexpr (!ASYM) {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            acc = T::absmax_acc(acc, r[i].x);
            acc = T::absmax_acc(acc, r[i].y);
            acc = T::absmax_acc(acc, r[i].z);
            acc = T::absmax_acc(acc, r[i].w);
        }
        const float m = as_f(block_reduce<OpMaxU, NW>(T::absmax_finish(acc), red[0]));
        sr = ac ? sym_row_autocast<DT>(m, a.sym) : sym_row<DT>(m, a.sym);
        ub = m;
        lb = -m;
        top = sym_top_bin<DT>(m, sr, ac);
        if (t == 0 && a.scales) {
            a.scales[2 * row] = sr.s;
            a.scales[2 * row + 1] = sr.t2;
        }
    } else {
        float mx, mn;
        if constexpr (T::ESIZE == 2) {  // min and max on the raw bits (order-preserving 16-bit keys), one reduction
            MinMaxKeys mk;
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                mk.acc(r[i].x);
                mk.acc(r[i].y);
                mk.acc(r[i].z);
                mk.acc(r[i].w);
            }
            minmax_from_keys<DT>(block_reduce<OpPkMaxU16, NW>(mk.word(), red[0]), mx, mn);
        } else {
            MinMax mm;
            {
                float f0[T::EPD];
                T::unpack(r[0].x, f0);
                mm.mx = mm.mn = f0[0];
                mm.absacc = 0;
            }
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                minmax_acc<DT>(mm, r[i].x);
                minmax_acc<DT>(mm, r[i].y);
                minmax_acc<DT>(mm, r[i].z);
                minmax_acc<DT>(mm, r[i].w);
            }
            uint32_t nb = T::absmax_finish(mm.absacc), umx = as_u(mm.mx), umn = as_u(mm.mn);
            block_reduce3<OpMaxU, OpMaxF, OpMinF, NW>(nb, umx, umn, red);
            mx = as_f(umx), mn = as_f(umn);
            if (absbits_is_nan(nb)) mx = mn = as_f(0x7FC00000u);  // torch.max/min propagate NaN
        }
        ar = asym_row<DT>(mx, mn, a.asym);
        ub = mx;
        lb = mn;
        top = asym_bin<DT>(mx, ar, a.asym);
        if (t == 0 && a.scales) {
            a.scales[2 * row] = ar.a;
            a.scales[2 * row + 1] = ar.mn;
        }
    }
    if (t == 0 && a.bounds) {
        a.bounds[2 * row] = ub;
        a.bounds[2 * row + 1] = lb;
    }

    uint32_t nbad;  // per lane; summed over the wave below
    bool count;     // block-uniform: the row's top bin does not fit the container
    if constexpr (!ASYM) {
        const bool plain = !a.mask && !ac;  // the same for every block of the launch
        if (plain && a.container == BINS_INT4) nbad = export_row_body<DT, TPR, VPT, false, BINS_INT4, 0, 0>(a, r, row, t, nvec, sr, ar, ub, lb, top, count);
        else if (plain && a.container == BINS_INT8) nbad = export_row_body<DT, TPR, VPT, false, BINS_INT8, 0, 0>(a, r, row, t, nvec, sr, ar, ub, lb, top, count);
        else nbad = export_row_body<DT, TPR, VPT, false, -1, -1, -1>(a, r, row, t, nvec, sr, ar, ub, lb, top, count);
    } else {
        nbad = export_row_body<DT, TPR, VPT, true, -1, 0, -1>(a, r, row, t, nvec, sr, ar, ub, lb, top, count);
    }
    if (a.overflow) {
        if (!count) {
            if (t == 0) a.overflow[row] = 0;
        } else if constexpr (NW == 1) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nbad += (uint32_t)__shfl_xor((int)nbad, o, 64);
            if (t == 0) a.overflow[row] = (int32_t)nbad;
        } else {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nbad += (uint32_t)__shfl_xor((int)nbad, o, 64);
            if ((t & 63) == 0) cnt_lds[t >> 6] = nbad;
            __syncthreads();  // `count` is block-uniform
            if (t == 0) {
                uint32_t s = 0;
#pragma unroll
                for (int i = 0; i < NW; ++i) s += cnt_lds[i];
                a.overflow[row] = (int32_t)s;
            }
        }
    }
}

// Any width / alignment: 256 threads sweep the row twice (second sweep is cache-hot); each thread packs element PAIRS so
// that an int4 byte has one writer.  No STE mask on this path (bounds only).
template <int DT, bool ASYM> __global__ __launch_bounds__(256) void row_export_generic_kernel(ExportArgs a) {
    using T = Ty<DT>;
    constexpr int NW = 4;
    __shared__ uint32_t red[3][NW];
    __shared__ uint32_t cnt_lds[NW];
    const int64_t row = blockIdx.x;
    const int t = threadIdx.x;
    const int64_t base = row * a.cols, cols = a.cols;
    const bool ac = a.autocast != 0;
    SymRow sr;
    AsymRow ar;
    float ub, lb;
    if constexpr (!ASYM) {
        uint32_t acc = 0;
        for (int64_t c = t; c < cols; c += 256) {
            const uint32_t b = as_u(T::load1(a.x, base + c)) & 0x7FFFFFFFu;
            acc = acc > b ? acc : b;
        }
        const float m = as_f(block_reduce<OpMaxU, NW>(acc, red[0]));
        sr = ac ? sym_row_autocast<DT>(m, a.sym) : sym_row<DT>(m, a.sym);
        ub = m;
        lb = -m;
        if (t == 0 && a.scales) {
            a.scales[2 * row] = sr.s;
            a.scales[2 * row + 1] = sr.t2;
        }
    } else {
        const float first = T::load1(a.x, base);
        float mx = first, mn = first;
        uint32_t acc = 0;
        for (int64_t c = t; c < cols; c += 256) {
            const float v = T::load1(a.x, base + c);
            mx = __builtin_fmaxf(mx, v);
            mn = __builtin_fminf(mn, v);
            const uint32_t b = as_u(v) & 0x7FFFFFFFu;
            acc = acc > b ? acc : b;
        }
        uint32_t nb = acc, umx = as_u(mx), umn = as_u(mn);
        block_reduce3<OpMaxU, OpMaxF, OpMinF, NW>(nb, umx, umn, red);
        mx = as_f(umx), mn = as_f(umn);
        if (absbits_is_nan(nb)) mx = mn = as_f(0x7FC00000u);
        ar = asym_row<DT>(mx, mn, a.asym);
        ub = mx;
        lb = mn;
        if (t == 0 && a.scales) {
            a.scales[2 * row] = ar.a;
            a.scales[2 * row + 1] = ar.mn;
        }
    }
    if (t == 0 && a.bounds) {
        a.bounds[2 * row] = ub;
        a.bounds[2 * row + 1] = lb;
    }
    const int cont = a.container;
    uint32_t nbad = 0;
    if (cont != BINS_NONE) {
        char* brow = (char*)a.bins + row * a.row_bytes;
        for (int64_t c = 2 * (int64_t)t; c < cols; c += 512) {
            int q[2] = {0, 0};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (c + e < cols) {
                    const float x = T::load1(a.x, base + c + e);
                    const float b = ASYM ? asym_bin<DT>(x, ar, a.asym) : sym_bin<DT>(x, sr, ac);
                    bool bad;
                    q[e] = sat_bin(b, a.cmin, a.cmax, bad);
                    nbad += bad ? 1u : 0u;
                }
            }
            if (cont == BINS_INT4) {
                ((uint8_t*)brow)[c >> 1] = (uint8_t)((q[0] & 0xF) | ((q[1] & 0xF) << 4));
            } else if (cont == BINS_INT8) {
                ((int8_t*)brow)[c] = (int8_t)q[0];
                if (c + 1 < cols) ((int8_t*)brow)[c + 1] = (int8_t)q[1];
            } else {
                ((int16_t*)brow)[c] = (int16_t)q[0];
                if (c + 1 < cols) ((int16_t*)brow)[c + 1] = (int16_t)q[1];
            }
        }
    }
    if (a.overflow) {
        // per-thread counts -> block sum (ballot-free: counts differ per lane)
        uint32_t s = nbad;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += (uint32_t)__shfl_xor((int)s, o, 64);
        if ((t & 63) == 0) cnt_lds[t >> 6] = s;
        __syncthreads();
        if (t == 0) a.overflow[row] = (int32_t)(cnt_lds[0] + cnt_lds[1] + cnt_lds[2] + cnt_lds[3]);
    }
}

}  // namespace fq
