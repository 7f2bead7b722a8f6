// fq_device.h -- device-side building blocks of the gfx950 fake-quant kernels.
//
// Written for CDNA4 only: 64-lane wavefronts, DPP row operations + v_readlane for the
// wave-level reductions, v_cvt_pk_bf16_f32 for bf16 rounding, v_pk_max_u16 for the packed
// |x| max.  No portability layer.
//
// Numerics contract (DESIGN.md "Numerics"): the reference computes every intermediate in
// the tensor dtype, one rounding per ATen op.  Every helper that is named after a
// reference op therefore ends in a round-to-dtype; the build uses -ffp-contract=off so
// no mul/add pair is ever fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fq {

enum : int { F32 = 0, BF16 = 1, F16 = 2 };

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

// ------------------------------------------------------------------------------------
// 16-byte global accesses.  NT = non-temporal ("streaming") cache policy: every tensor on this
// path is touched exactly once per launch, so lines need not be retained in L2 / Infinity Cache.
// Measured on MI355X (tools/kbench): a 90 MB -> 90 MB copy runs 5.55 TB/s plain, 5.97 TB/s NT.
// ------------------------------------------------------------------------------------
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld16(const uint4* p) {
    if constexpr (NT) {
        u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
template <bool NT> __device__ __forceinline__ void st16(uint4* p, uint4 v) {
    if constexpr (NT) {
        u32x4_t w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, (u32x4_t*)p);
    } else {
        *p = v;
    }
}

// ------------------------------------------------------------------------------------
// dtype traits.  A "dword" is the 32-bit register unit: 1 fp32 element or 2 16-bit ones.
// ------------------------------------------------------------------------------------
template <int DT> struct Ty;

template <> struct Ty<F32> {
    static constexpr int ESIZE = 4, EPD = 1;
    static constexpr uint32_t ABS_MASK = 0x7FFFFFFFu;
    __device__ static __forceinline__ void unpack(uint32_t w, float (&f)[1]) { f[0] = as_f(w); }
    __device__ static __forceinline__ uint32_t pack(const float (&f)[1]) { return as_u(f[0]); }
    __device__ static __forceinline__ void round_dt(float (&)[1]) {}
    __device__ static __forceinline__ float rb(float v) { return v; }
    // packed |x| max on raw bits (sign-magnitude order; NaN patterns sort above Inf, so the
    // integer max propagates NaN exactly like torch.max)
    __device__ static __forceinline__ uint32_t absmax_acc(uint32_t acc, uint32_t w) {
        uint32_t a = w & ABS_MASK;
        return acc > a ? acc : a;
    }
    __device__ static __forceinline__ uint32_t absmax_finish(uint32_t acc) { return acc; }  // -> fp32 bits of max|x|
    __device__ static __forceinline__ float load1(const void* p, int64_t i) { return ((const float*)p)[i]; }
    __device__ static __forceinline__ void store1(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
};

template <> struct Ty<BF16> {
    static constexpr int ESIZE = 2, EPD = 2;
    static constexpr uint32_t ABS_MASK = 0x7FFF7FFFu;
    __device__ static __forceinline__ void unpack(uint32_t w, float (&f)[2]) {
        f[0] = as_f(w << 16);
        f[1] = as_f(w & 0xFFFF0000u);
    }
    __device__ static __forceinline__ uint32_t pack(const float (&f)[2]) {  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
        f32x2_t v;
        v.x = f[0];
        v.y = f[1];
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    }
    __device__ static __forceinline__ void round_dt(float (&f)[2]) { unpack(pack(f), f); }
    __device__ static __forceinline__ float rb(float v) { return (float)(__bf16)v; }
    __device__ static __forceinline__ uint32_t absmax_acc(uint32_t acc, uint32_t w) {  // v_and + v_pk_max_u16
        u16x2_t a = __builtin_bit_cast(u16x2_t, acc), b = __builtin_bit_cast(u16x2_t, w & ABS_MASK);
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(a, b));
    }
    __device__ static __forceinline__ uint32_t absmax_finish(uint32_t acc) {
        uint32_t lo = acc & 0xFFFFu, hi = acc >> 16;
        return (lo > hi ? lo : hi) << 16;
    }
    __device__ static __forceinline__ float load1(const void* p, int64_t i) { return as_f((uint32_t)((const uint16_t*)p)[i] << 16); }
    __device__ static __forceinline__ void store1(void* p, int64_t i, float v) {
        ((uint16_t*)p)[i] = __builtin_bit_cast(uint16_t, (__bf16)v);
    }
};

template <> struct Ty<F16> {
    static constexpr int ESIZE = 2, EPD = 2;
    static constexpr uint32_t ABS_MASK = 0x7FFF7FFFu;
    __device__ static __forceinline__ void unpack(uint32_t w, float (&f)[2]) {
        f16x2_t h = __builtin_bit_cast(f16x2_t, w);
        f[0] = (float)h.x;
        f[1] = (float)h.y;
    }
    __device__ static __forceinline__ uint32_t pack(const float (&f)[2]) {  // 2x v_cvt_f16_f32 (RNE) + pack
        f16x2_t h;
        h.x = (_Float16)f[0];
        h.y = (_Float16)f[1];
        return __builtin_bit_cast(uint32_t, h);
    }
    __device__ static __forceinline__ void round_dt(float (&f)[2]) { unpack(pack(f), f); }
    __device__ static __forceinline__ float rb(float v) { return (float)(_Float16)v; }
    __device__ static __forceinline__ uint32_t absmax_acc(uint32_t acc, uint32_t w) { return Ty<BF16>::absmax_acc(acc, w); }
    __device__ static __forceinline__ uint32_t absmax_finish(uint32_t acc) {
        uint32_t lo = acc & 0xFFFFu, hi = acc >> 16;
        uint16_t m = (uint16_t)(lo > hi ? lo : hi);
        return as_u((float)__builtin_bit_cast(_Float16, m));
    }
    __device__ static __forceinline__ float load1(const void* p, int64_t i) {
        return (float)__builtin_bit_cast(_Float16, ((const uint16_t*)p)[i]);
    }
    __device__ static __forceinline__ void store1(void* p, int64_t i, float v) {
        ((uint16_t*)p)[i] = __builtin_bit_cast(uint16_t, (_Float16)v);
    }
};

// ------------------------------------------------------------------------------------
// order-preserving float <-> uint key (for atomicMax-based cross-block min/max)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t okey(float v) {
    uint32_t u = as_u(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float okey_inv(uint32_t k) { return as_f((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// ------------------------------------------------------------------------------------
// wave64 reduction for IDEMPOTENT ops (max / min): 4 DPP steps make every 16-lane row
// uniform, 4 v_readlane + scalar ops finish.  All 64 lanes must be active.
// ------------------------------------------------------------------------------------
struct OpMaxU {
    __device__ static __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a > b ? a : b; }
};
struct OpMaxF {  // v_max_f32: ignores NaN operands; NaN is tracked separately by the callers
    __device__ static __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return as_u(__builtin_fmaxf(as_f(a), as_f(b))); }
};
struct OpMinF {
    __device__ static __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return as_u(__builtin_fminf(as_f(a), as_f(b))); }
};

template <int CTRL> __device__ __forceinline__ uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}

template <class Op> __device__ __forceinline__ uint32_t wave_reduce(uint32_t v) {
    v = Op::f(v, dpp<0xB1>(v));   // quad_perm:[1,0,3,2]
    v = Op::f(v, dpp<0x4E>(v));   // quad_perm:[2,3,0,1]
    v = Op::f(v, dpp<0x141>(v));  // row_half_mirror
    v = Op::f(v, dpp<0x140>(v));  // row_mirror  -> each 16-lane row uniform
    uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
    uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
    uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return Op::f(Op::f(a, b), Op::f(c, d));
}

// block reduction over NW waves through LDS (one barrier); result uniform in every thread
template <class Op, int NW> __device__ __forceinline__ uint32_t block_reduce(uint32_t v, uint32_t* lds /*[NW]*/) {
    uint32_t w = wave_reduce<Op>(v);
    if constexpr (NW == 1) return w;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wave] = w;
    __syncthreads();
    uint32_t r = lds[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = Op::f(r, lds[i]);
    return r;
}

// three reductions behind ONE barrier (AsymQuantizer: NaN detector, max, min); lds is [3][NW]
template <class Op0, class Op1, class Op2, int NW>
__device__ __forceinline__ void block_reduce3(uint32_t& v0, uint32_t& v1, uint32_t& v2, uint32_t (*lds)[NW > 1 ? NW : 1]) {
    v0 = wave_reduce<Op0>(v0);
    v1 = wave_reduce<Op1>(v1);
    v2 = wave_reduce<Op2>(v2);
    if constexpr (NW > 1) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            lds[0][wave] = v0;
            lds[1][wave] = v1;
            lds[2][wave] = v2;
        }
        __syncthreads();
        v0 = lds[0][0];
        v1 = lds[1][0];
        v2 = lds[2][0];
#pragma unroll
        for (int i = 1; i < NW; ++i) {
            v0 = Op0::f(v0, lds[0][i]);
            v1 = Op1::f(v1, lds[1][i]);
            v2 = Op2::f(v2, lds[2][i]);
        }
    }
}

// ------------------------------------------------------------------------------------
// per-row scale terms
// ------------------------------------------------------------------------------------
struct SymConst {   // launch-uniform
    float qmax;     // 2^(bits-1)-1 as fp32 (ATen keeps a Python scalar in fp32 for mul)
    float c6;       // the "+1e-6": rounded to the dtype (CPU-eager) or plain fp32 (device-eager)
};
struct SymRow {
    float s, t2, rinv;
    bool mk;  // t2 lies where div_exact() is proven equal to the IEEE quotient (wave-uniform)
};
// x / b for a per-row (or launch-wide) divisor b WITHOUT the IEEE-divide sequence (~10 VALU ops + v_div_scale /
// v_div_fmas / v_div_fixup): Markstein's correction of a reciprocal multiply.  With r = RN(1/b),
//     q0 = RN(x * r) ; e = fma(-q0, b, x) (exact) ; q = RN(q0 + e * r)
// is the correctly rounded quotient -- bit-identical to the division, signed zeros included (`e == 0` keeps the sign
// of a zero numerator; NaN propagates; an infinite numerator returns q0 = +-inf).  Preconditions, both enforced by the callers: r is a NORMAL number (b <= 2^100)
// and e does not underflow (|x| >= 2^-101); the quantizers guarantee the second whenever the quotient can still
// influence a bin (divisor >= 2^-60: a quotient below 2^-32 rounds to bin 0 for every bit width).
// Checked against the division in tests/c_host/markstein_check.c (10^8 quotients).
__device__ __forceinline__ bool div_exact_ok(float b) { return b >= 0x1p-60f && b <= 0x1p100f; }  // false for NaN
__device__ __forceinline__ float div_exact(float x, float b, float r) {
    const float q0 = x * r;
    const float e = __builtin_fmaf(-q0, b, x);
    // an infinite numerator (fp16 bins beyond 65504 round to inf) gives q0 = +-inf = the quotient, but e = NaN
    return (e == 0.0f || __builtin_fabsf(q0) == __builtin_inff()) ? q0 : __builtin_fmaf(e, r, q0);
}
// utils_quant.py:71-72:  s = reciprocal(max + 1e-6) * qmax ;  divisor = s + 1e-6
template <int DT> __device__ __forceinline__ SymRow sym_row(float m, SymConst k) {
    using T = Ty<DT>;
    SymRow r;
    const float t1 = T::rb(m + k.c6);
    const float rc = T::rb(1.0f / t1);  // IEEE divide
    r.s = T::rb(rc * k.qmax);
    r.t2 = T::rb(r.s + k.c6);
    r.rinv = 1.0f / r.t2;  // bf16 FAST path: plain multiply (exactness argument: DESIGN.md); fp16: div_exact()
    r.mk = div_exact_ok(r.t2);
    return r;
}

// ---- SymQuantizer as it executes under CUDA autocast (torch.autocast("cuda", bf16|fp16), LLM-QAT's --bf16 run) ----
// `reciprocal` is on autocast's fp32 list, so  s = qmax / (max + 1e-6)  comes back as an fp32 tensor and every op
// after it is promoted to fp32:  t1 = rb(max + 1e-6)  (the add still runs in the tensor dtype; the GPU keeps the
// scalar in fp32: c6 = 1e-6f unless the `sem` knob asks for the CPU's rounded scalar),  s = (1/t1) * qmax,  idx = round(x * s),  y = idx / (s + 1e-6)  -- all fp32, output fp32.
// (measured on MI355X: tools/autocast_probe.py).  AC: 0 = no autocast, 1 = autocast, result rounded once to the
// tensor dtype (what F.linear's autocast cast does to it next), 2 = autocast, fp32 result as the reference returns it.
template <int DT> __device__ __forceinline__ SymRow sym_row_autocast(float m, SymConst k) {
    SymRow r;
    const float t1 = Ty<DT>::rb(m + k.c6);  // k.c6: 1e-6f (device-eager, what a real autocast run computes) or rb(1e-6) (`sem` knob)
    r.s = (1.0f / t1) * k.qmax;
    r.t2 = r.s + 1e-6f;
    r.rinv = 1.0f / r.t2;
    r.mk = true;  // t2 = s + 1e-6 with s in [0, 2^31 * 1e6]: always inside div_exact()'s range (or NaN, which propagates)
    return r;
}
__device__ __forceinline__ float div_by_row(float a, const SymRow& r) { return div_exact(a, r.t2, r.rinv); }
__device__ __forceinline__ float sym_elem_autocast(float x, const SymRow& r) { return div_by_row(__builtin_rintf(x * r.s), r); }

struct AsymConst {
    float S;      // 2^bits - 1 as fp32
    float invS;   // fp32 1/S   (device-eager: x.div(python_scalar) == x * (1/S))
    float c8;     // the "+1e-8"
    int mul_inv;  // 1 -> use invS
};
struct AsymRow {
    float mn, al, a, ra;
    bool mk;  // a lies where div_exact() equals the IEEE quotient (wave-uniform)
};
// utils_quant.py:116-124,:144: alpha = max - min ; beta = min ; a = alpha + 1e-8
template <int DT> __device__ __forceinline__ AsymRow asym_row(float mx, float mn, AsymConst k) {
    using T = Ty<DT>;
    AsymRow r;
    r.mn = mn;
    r.al = T::rb(mx - mn);
    r.a = T::rb(r.al + k.c8);
    r.ra = 1.0f / r.a;  // bf16 FAST path: plain multiply; otherwise div_exact()
    r.mk = div_exact_ok(r.a);
    return r;
}

__device__ __forceinline__ int32_t idx_i32(float q) {  // same coding as the oracle / fixtures
    if (q != q) return INT32_MIN;
    if (q >= 2.0e9f) return __builtin_isinf(q) ? INT32_MAX : 2000000000;
    if (q <= -2.0e9f) return __builtin_isinf(q) ? -INT32_MAX : -2000000000;
    return (int32_t)q;
}

// ------------------------------------------------------------------------------------
// element chains on one dword.  `idx` (if non-null) receives EPD bin indices.
// ------------------------------------------------------------------------------------
// utils_quant.py:72   output = round(input * s).div(s + 1e-6)
template <int DT, bool FAST> __device__ __forceinline__ uint32_t sym_chain(float (&f)[Ty<DT>::EPD], const SymRow& r, int32_t* idx) {
    using T = Ty<DT>;
#pragma unroll
    for (int e = 0; e < T::EPD; ++e) f[e] = f[e] * r.s;
    T::round_dt(f);
#pragma unroll
    for (int e = 0; e < T::EPD; ++e) f[e] = __builtin_rintf(f[e]);  // v_rndne_f32
    if (idx) {
#pragma unroll
        for (int e = 0; e < T::EPD; ++e) idx[e] = idx_i32(f[e]);
    }
#pragma unroll
    for (int e = 0; e < T::EPD; ++e) {
        if constexpr (FAST) f[e] = f[e] * r.rinv;
        else if constexpr (DT == F16) f[e] = r.mk ? div_exact(f[e], r.t2, r.rinv) : f[e] / r.t2;  // same bits, a third of the ops
        else f[e] = f[e] / r.t2;  // fp32 rows: the kernel is HBM-bound with the IEEE sequence as well
    }
    return T::pack(f);
}

// utils_quant.py:144-147
template <int DT, bool FAST> __device__ __forceinline__ uint32_t sym_dword(uint32_t w, const SymRow& r, int32_t* idx) {
    float f[Ty<DT>::EPD];
    Ty<DT>::unpack(w, f);
    return sym_chain<DT, FAST>(f, r, idx);
}

// FAST (bf16, bits <= 8): both IEEE divides become multiplies by a reciprocal.  Same argument as for Sym:
// numerator and denominator have 8-bit significands (d = rb(x-beta) and a; idx <= 255 and S = 2^b-1), so the
// exact quotient is never a bf16 rounding midpoint and lies >= 2^-17 (relative) away from one, while the
// reciprocal-multiply error is < 2^-22.  Quotients below the bf16 normal range all round to bin 0 either way.
template <int DT, bool FAST = false>
__device__ __forceinline__ uint32_t asym_chain(float (&f)[Ty<DT>::EPD], const AsymRow& r, const AsymConst& k, int32_t* idx) {
    using T = Ty<DT>;
#pragma unroll
    for (int e = 0; e < T::EPD; ++e) f[e] = f[e] - r.mn;  // input - beta
    T::round_dt(f);
#pragma unroll
    for (int e = 0; e < T::EPD; ++e)  // / (alpha + 1e-8)
        f[e] = FAST ? f[e] * r.ra : (r.mk ? div_exact(f[e], r.a, r.ra) : f[e] / r.a);
    T::round_dt(f);
#pragma unroll
    for (int e = 0; e < T::EPD; ++e) f[e] = f[e] * k.S;  // * s
    T::round_dt(f);
#pragma unroll
    for (int e = 0; e < T::EPD; ++e) f[e] = __builtin_rintf(f[e]);
    if (idx) {
#pragma unroll
        for (int e = 0; e < T::EPD; ++e) idx[e] = idx_i32(f[e]);
    }
#pragma unroll
    for (int e = 0; e < T::EPD; ++e)  // .div(s): S = 2^bits - 1 in [3, 2^31], integer numerators: always inside div_exact()'s range
        f[e] = (FAST || k.mul_inv) ? f[e] * k.invS : div_exact(f[e], k.S, k.invS);
    T::round_dt(f);
#pragma unroll
    for (int e = 0; e < T::EPD; ++e) f[e] = f[e] * r.a;  // * (alpha + 1e-8)
    T::round_dt(f);
#pragma unroll
    for (int e = 0; e < T::EPD; ++e) f[e] = f[e] + r.mn;  // + beta
    return T::pack(f);
}

// The two halves of that chain, for the lookup-table form of the register kernel (16-bit tensors, bits <= 8): everything
// up to the bin index depends on the element, everything after it only on (bin, row) -- at most 256 values per row.
template <int DT, bool FAST> __device__ __forceinline__ float asym_first_half(float x, const AsymRow& r, const AsymConst& k) {
    using T = Ty<DT>;
    const float d = T::rb(x - r.mn);
    const float n = T::rb(FAST ? d * r.ra : (r.mk ? div_exact(d, r.a, r.ra) : d / r.a));
    return __builtin_rintf(T::rb(n * k.S));
}
template <int DT, bool FAST> __device__ __forceinline__ float asym_second_half(float q, const AsymRow& r, const AsymConst& k) {
    using T = Ty<DT>;
    const float w = T::rb((FAST || k.mul_inv) ? q * k.invS : div_exact(q, k.S, k.invS));
    return T::rb(T::rb(w * r.a) + r.mn);
}

template <int DT, bool FAST = false>
__device__ __forceinline__ uint32_t asym_dword(uint32_t w, const AsymRow& r, const AsymConst& k, int32_t* idx) {
    float f[Ty<DT>::EPD];
    Ty<DT>::unpack(w, f);
    return asym_chain<DT, FAST>(f, r, k, idx);
}

// QuantizeLinear's 1-/2-bit weight branches (utils_quant.py:203-242), forward value INCLUDING the
// detach trick  weight = q.detach() - w.detach() + w  (two more roundings; gradient is the identity).
//   1 bit : q = sc * sign(w / sc)                                            sign(NaN) = sign(0) = 0, as torch.sign
//   2 bit : q = sc * (round(clamp(w / sc, -cv, cv) * 2 - 0.5) + 0.5) / 2     cv = 0.99 (clamp propagates NaN)
// w12_elem is the reference chain op for op (any scale: zero, NaN, infinite, subnormal); rsc = 1 / sc and mk = div_exact_ok(sc)
// are per row: w / sc then costs 3-4 VALU ops instead of the IEEE sequence, same bits (div_exact()'s second precondition,
// |w| >= 2^-100 or w == 0, is checked per element).  Rows with an ordinary scale take w12_dword below.
template <int DT, int WBITS> __device__ __forceinline__ float w12_elem(float w, float sc, float cv, float rsc, bool mk) {
    using T = Ty<DT>;
    const bool fast = mk && (__builtin_fabsf(w) >= 0x1p-100f || w == 0.0f) && !(__builtin_fabsf(w) == __builtin_inff());
    const float t = T::rb(fast ? div_exact(w, sc, rsc) : w / sc);
    float q;
    if constexpr (WBITS == 1) {
        const float sg = (t > 0.f) ? 1.f : (t < 0.f) ? -1.f : 0.f;
        q = sc * sg;  // sc * {-1, 0, 1}: exact, the reference's rounding of it is the identity
    } else {
        const float c = (t != t) ? t : __builtin_fminf(__builtin_fmaxf(t, -cv), cv);
        float u = T::rb(c) * 2.0f;  // |c| <= 0.99: doubling a dtype value is exact, rounding it again the identity
        u = T::rb(u - 0.5f);
        u = __builtin_rintf(u);     // in {-2, -1, 0, 1} (or NaN)
        u = u + 0.5f;               // {-1.5, -0.5, 0.5, 1.5}: exact in every dtype
        u = T::rb(sc * u);
        q = T::rb(u / 2.0f);
    }
    return T::rb(q - w) + w;  // rounded once more by the store
}

// The same chain on one dword (two 16-bit elements / one fp32) for a row whose scale is ORDINARY (mk: sc in [2^-60, 2^100],
// so rsc is normal and nothing below underflows), written on float2 so that the multiplies / adds / explicit fmas become
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, and with the steps that are provably the identity for such a row removed:
//  * bf16: w / sc == rb(w * rsc) after the rounding to bf16 (numerator and divisor have 8-bit significands: the quotient is
//    never within 2^-17 of a bf16 rounding midpoint, the reciprocal multiply errs by < 2^-22; DESIGN.md "Numerics", checked
//    exhaustively in tests/test_oracle_golden.py) wherever the quotient is a NORMAL bf16 value.  Below that the 2-bit chain
//    does not see the difference (|t| < 2^-26 vanishes in `2 t - 0.5`), the 1-bit chain does (sign(t) = 0 once t rounds to
//    zero): it sends |w * rsc| < 2^-100 (w != 0) to w12_elem.  fp16 keeps div_exact (always applicable: a non-zero fp16 value
//    is >= 2^-24); fp32 checks div_exact's precondition per element.
//  * t is NaN only if w is (sc is finite and positive), and then `+ w` makes the result NaN whatever came before: the clamp
//    needs no NaN select (v_med3_f32).
//  * rb(clamp(t, +-cv)) == clamp(t, +-rb(cv)) for a dtype-valued t (rounding is monotone, t is a fixed point): cvr = rb(cv).
//  * 2 c - 0.5 in one explicit fma (2 c is exact); (r + 0.5) * sc / 2 == fma(r, sc/2, sc/4) rounded once: r * (sc/2) and the sum
//    are exact reals before the single rounding, and halving commutes with rounding while nothing is subnormal (bf16, fp32:
//    sc >= 2^-60; fp16 tensors keep the reference's two steps, their sc * u can be subnormal).
struct W12Row {
    float sc, rsc, sch, scq, cvr;  // scale, 1 / sc, sc / 2, sc / 4, rb(cv)
    bool mk;
};
template <int DT> __device__ __forceinline__ W12Row w12_row(float sc, float cv) {
    W12Row r;
    r.sc = sc;
    r.rsc = 1.0f / sc;
    r.sch = 0.5f * sc;
    r.scq = 0.25f * sc;
    r.cvr = Ty<DT>::rb(cv);
    r.mk = div_exact_ok(sc);
    return r;
}
template <int DT> __device__ __forceinline__ f32x2_t rb2(f32x2_t v) {  // round both to the dtype (bf16: one v_cvt_pk_bf16_f32)
    float f[2] = {v.x, v.y};
    Ty<DT>::round_dt(f);
    f32x2_t o;
    o.x = f[0];
    o.y = f[1];
    return o;
}
// any scale / any element (the callers branch once per row or vector, wave-uniformly, never per dword)
template <int DT, int WBITS> __device__ __forceinline__ uint32_t w12_dword_any(uint32_t wbits, const W12Row& r, float cv) {
    using T = Ty<DT>;
    float f[T::EPD];
    T::unpack(wbits, f);
#pragma unroll
    for (int e = 0; e < T::EPD; ++e) f[e] = w12_elem<DT, WBITS>(f[e], r.sc, cv, r.rsc, r.mk);
    return T::pack(f);
}
// r.mk must hold.  `odd` is set when an element needs the reference chain instead (1-bit bf16: a quotient that may round to
// zero; fp32: an element outside div_exact's precondition): the caller then redoes the vector with w12_dword_any behind a
// wave-uniform branch (a per-lane select would make the compiler execute the IEEE divisions for every element).
template <int DT, int WBITS> __device__ __forceinline__ uint32_t w12_dword(uint32_t wbits, const W12Row& r, bool& odd) {
    using T = Ty<DT>;
    float f[T::EPD];
    T::unpack(wbits, f);
    if constexpr (T::EPD == 1) {
        const float w = f[0], aw = __builtin_fabsf(w);
        odd = odd || !((aw >= 0x1p-100f || w == 0.0f) && !(aw == __builtin_inff()));
        float t = div_exact(w, r.sc, r.rsc), q;
        if constexpr (WBITS == 1) q = (t > 0.f) ? r.sc : (t < 0.f) ? -r.sc : 0.f;
        else q = __builtin_fmaf(__builtin_rintf(__builtin_fmaf(__builtin_amdgcn_fmed3f(t, -r.cvr, r.cvr), 2.0f, -0.5f)), r.sch, r.scq);
        f[0] = (q - w) + w;
        return T::pack(f);
    } else {
        f32x2_t w2, t;
        w2.x = f[0];
        w2.y = f[1];
        const f32x2_t q0 = w2 * r.rsc;
        if constexpr (DT == BF16) {
            t = q0;
            if constexpr (WBITS == 1)  // sign(t) needs the exact quotient where t may round to zero
                odd = odd || (__builtin_fabsf(q0.x) < 0x1p-100f && w2.x != 0.0f) || (__builtin_fabsf(q0.y) < 0x1p-100f && w2.y != 0.0f);
        } else {  // fp16: Markstein-corrected quotient, the division bit for bit
            const f32x2_t e2 = __builtin_elementwise_fma(-q0, (f32x2_t){r.sc, r.sc}, w2);
            const f32x2_t c2 = __builtin_elementwise_fma(e2, (f32x2_t){r.rsc, r.rsc}, q0);
            t.x = (e2.x == 0.0f || __builtin_fabsf(q0.x) == __builtin_inff()) ? q0.x : c2.x;
            t.y = (e2.y == 0.0f || __builtin_fabsf(q0.y) == __builtin_inff()) ? q0.y : c2.y;
        }
        t = rb2<DT>(t);
        f32x2_t q;
        if constexpr (WBITS == 1) {
            q.x = (t.x > 0.f) ? r.sc : (t.x < 0.f) ? -r.sc : 0.f;
            q.y = (t.y > 0.f) ? r.sc : (t.y < 0.f) ? -r.sc : 0.f;
        } else {
            f32x2_t c;
            c.x = __builtin_amdgcn_fmed3f(t.x, -r.cvr, r.cvr);
            c.y = __builtin_amdgcn_fmed3f(t.y, -r.cvr, r.cvr);
            f32x2_t u = rb2<DT>(__builtin_elementwise_fma(c, (f32x2_t){2.0f, 2.0f}, (f32x2_t){-0.5f, -0.5f}));
            u.x = __builtin_rintf(u.x);
            u.y = __builtin_rintf(u.y);
            if constexpr (DT == BF16) {
                q = rb2<DT>(__builtin_elementwise_fma(u, (f32x2_t){r.sch, r.sch}, (f32x2_t){r.scq, r.scq}));
            } else {
                q = rb2<DT>((u + 0.5f) * r.sc);
                q = rb2<DT>(q * 0.5f);
            }
        }
        const f32x2_t d = rb2<DT>(q - w2) + w2;
        f[0] = d.x;
        f[1] = d.y;
        return T::pack(f);
    }
}
// one 16-byte vector: the fast chain, redone with the reference chain if any lane of the wave met an odd element
template <int DT, int WBITS> __device__ __forceinline__ uint4 w12_vec(const uint4& in, const W12Row& r, float cv) {
    const uint32_t w[4] = {in.x, in.y, in.z, in.w};
    uint32_t o[4];
    bool odd = false;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = w12_dword<DT, WBITS>(w[d], r, odd);
    if constexpr (DT == F32 || (DT == BF16 && WBITS == 1)) {
        if (__builtin_amdgcn_ballot_w64(odd) != 0) {  // wave-uniform, practically never taken
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = w12_dword_any<DT, WBITS>(w[d], r, cv);
        }
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// min/max/NaN accumulation for Asym on one dword
struct MinMax {
    float mx, mn;
    uint32_t absacc;  // packed |x| max: only consulted for "is there a NaN"
};
template <int DT> __device__ __forceinline__ void minmax_acc(MinMax& a, uint32_t w) {
    using T = Ty<DT>;
    float f[T::EPD];
    T::unpack(w, f);
#pragma unroll
    for (int e = 0; e < T::EPD; ++e) {
        a.mx = __builtin_fmaxf(a.mx, f[e]);
        a.mn = __builtin_fminf(a.mn, f[e]);
    }
    a.absacc = T::absmax_acc(a.absacc, w);
}
__device__ __forceinline__ bool absbits_is_nan(uint32_t fp32_abs_bits) { return fp32_abs_bits > 0x7F800000u; }

// ---- AsymQuantizer's row min and max for 16-bit tensors, on the raw bits -------------------------------------------------
// key = bits ^ (negative ? 0xFFFF : 0x8000) orders every bf16 / fp16 value as an unsigned 16-bit integer (-0 below +0, as
// v_min_f32 / v_max_f32 do; positive NaNs above +inf, negative NaNs below -inf), so min and max of both halves of a dword cost
// v_pk_ashrrev_i16 + v_bitop3 (or + xor) + v_pk_max_u16 + v_pk_min_u16: 2 VALU ops per element instead of 4 (unpack, v_max,
// v_min and the packed |x| max that only served to detect NaN), and ONE cross-lane reduction instead of three: the lane's max
// key and its inverted min key travel as the two halves of one dword under v_pk_max_u16.
struct OpPkMaxU16 {
    __device__ static __forceinline__ uint32_t f(uint32_t a, uint32_t b) {
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
    }
};
struct MinMaxKeys {
    uint32_t kmx = 0u, kmn = 0xFFFFFFFFu;  // running max / min key of both halves
    __device__ __forceinline__ void acc(uint32_t w) {
        typedef short s16x2_t __attribute__((ext_vector_type(2)));
        const uint32_t neg = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2_t, w) >> (short)15);  // 0xFFFF where the half is negative
        const uint32_t k = w ^ (neg | 0x80008000u);
        kmx = OpPkMaxU16::f(kmx, k);
        kmn = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, kmn), __builtin_bit_cast(u16x2_t, k)));
    }
    // (max key << 16) | ~(min key): reduce with OpPkMaxU16
    __device__ __forceinline__ uint32_t word() const {
        const uint32_t mx = (kmx & 0xFFFFu) > (kmx >> 16) ? (kmx & 0xFFFFu) : (kmx >> 16);
        const uint32_t mn = (kmn & 0xFFFFu) < (kmn >> 16) ? (kmn & 0xFFFFu) : (kmn >> 16);
        return (mx << 16) | (~mn & 0xFFFFu);
    }
};
// reduced word -> row max / min as fp32 (both NaN if the row holds a NaN: torch.max / torch.min propagate it)
template <int DT> __device__ __forceinline__ void minmax_from_keys(uint32_t word, float& mx, float& mn) {
    static_assert(Ty<DT>::ESIZE == 2, "16-bit tensors");
    constexpr uint32_t INF = DT == BF16 ? 0x7F80u : 0x7C00u;
    const uint32_t kx = word >> 16, kn = ~word & 0xFFFFu;
    if (kx > (INF | 0x8000u) || kn < (~(INF | 0x8000u) & 0xFFFFu)) {
        mx = mn = as_f(0x7FC00000u);
        return;
    }
    const uint32_t bx = kx >= 0x8000u ? kx ^ 0x8000u : ~kx & 0xFFFFu, bn = kn >= 0x8000u ? kn ^ 0x8000u : ~kn & 0xFFFFu;
    float f[2];
    Ty<DT>::unpack(bx | (bn << 16), f);
    mx = f[0];
    mn = f[1];
}

// STE mask on one dword (utils_quant.py:85-86): zero where x >= hi or x <= lo; NaN x passes
template <int DT> __device__ __forceinline__ uint32_t ste_dword(uint32_t g, uint32_t x, float lo, float hi) {
    using T = Ty<DT>;
    float f[T::EPD];
    T::unpack(x, f);
    if constexpr (T::EPD == 1) {
        return (f[0] >= hi || f[0] <= lo) ? 0u : g;
    } else {
        uint32_t keep = 0;
        if (!(f[0] >= hi || f[0] <= lo)) keep |= 0x0000FFFFu;
        if (!(f[1] >= hi || f[1] <= lo)) keep |= 0xFFFF0000u;
        return g & keep;
    }
}

}  // namespace fq
