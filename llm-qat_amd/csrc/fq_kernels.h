// fq_kernels.h -- the gfx950 kernels of the fake-quant hot path.
//
//   row_reg_kernel      one row per wave (TPR=64) or per workgroup (TPR=128..1024); the whole row
//                       is loaded ONCE with 16-byte global loads and stays in VGPRs across the
//                       reduce -> scale -> round -> dequant sequence.  HBM traffic = read x + write y.
//                       Training mode also records per-row bounds and a 1-bit/element STE mask.
//   row_generic_kernel  any alignment / odd widths: element loads, two sweeps (2nd sweep is L2-hot).
//   stats / apply       two-pass path for rows too long for registers (layerwise = one row).
//   ste_mask_kernel     STE backward from (bounds, mask): reads g (+ mask), never x.
//   ste_vec_kernel      STE backward re-reading x (the reference's data flow), pure streaming.
//   ste_rows_kernel     same, but skips the x read for rows whose recorded bounds lie inside the clip.
//   w12_kernel          elementwise part of QuantizeLinear's 1-/2-bit weight branches.
//
// Roofline for all of them: HBM bandwidth (<= ~10 VALU ops per element, no MFMA).
#pragma once
#include "fq_device.h"



namespace fq {

constexpr int MAX_MORE = 3;  // up to 4 tensors per launch (q/k/v weights + their shared input)

// Rows that do not follow one another in memory -- "last dim contiguous, rows strided": a slice / chunk() of the last dimension, the
// transpose(0, 1) of a 3-D tensor (round 5; the reference accepts any strides, utils_quant.py:37).  Row r of the [rows, cols] view is
// element (r / n_inner, r % n_inner) of a two-level index: it starts `(r / n_inner) * outer + (r % n_inner) * inner` BYTES behind the
// base pointer.  on == 0 (every model call site): row r starts at r * cols elements and the address arithmetic is what it always was.
// Block-uniform, so the division is scalar work once per block; rows < 2^31 in pitched mode (host-checked).
struct RowPitch {
    int64_t outer, inner;
    uint32_t n_inner;
    uint32_t on;
};
// PITCH is a template parameter of the streaming kernels: as a run-time (block-uniform) branch the pitch cost the contiguous launches
// 3-7 % at [2048,4096] and 0.4-1.6 % at [4096,11008] (the extra kernarg loads and scalar selects sit in every block's prologue:
// profiles/r05_ab_pitch_branch.txt), so contiguous tensors run the instantiations they always ran and pitched ones their own.
template <bool PITCH = true> __device__ __forceinline__ int64_t row_byte_off(int64_t row, int64_t row_bytes, const RowPitch& p) {
    if constexpr (!PITCH) {
        return row * row_bytes;
    } else {
        if (!p.on) return row * row_bytes;
        const uint32_t r = (uint32_t)row, q = r / p.n_inner;
        return (int64_t)q * p.outer + (int64_t)(r - q * p.n_inner) * p.inner;
    }
}

struct TensorSlot {
    int64_t row_begin;
    const void* x;
    void* y;
    float* bounds;
    uint64_t* mask;
    float qmax;
    RowPitch xp, yp;
};

struct RowArgs {
    const void* x;
    void* y;
    int32_t* idx;    // optional (debug): bin index per element
    float* scale;    // optional (debug): Sym s[rows] / Asym {alpha,beta}[rows]
    float* bounds;   // optional: {upper, lower} bound of the row's values, for ste_rows_kernel
    int64_t rows;
    int64_t cols;
    SymConst sym;
    AsymConst asym;
    // optional STE mask (register-resident kernels only): one bit per element, set where the
    // backward must zero the gradient (x >= hi || x <= lo).  Written only for rows whose bounds
    // do not already prove that nothing is clipped.  Layout: a plain bitmap per row, see ste_mask_record().
    uint64_t* mask;
    int64_t mask_row_words;
    float lo, hi;
    uint32_t clipk;  // integer form of the clip for 16-bit tensors (ste_flags16_*), 0 = compare as floats
    // optional FURTHER tensors of the same launch (register-resident Sym kernels only): same dtype and cols (so the same
    // launch shape), each with its own rows / bit width / outputs.  QuantizeLinear needs its weight [out, in] and its input
    // [tokens, in] fake-quantized at the same moment and both reduce over `in`; sibling projections (q/k/v, gate/up) add
    // their weights: one launch instead of up to four saves ~2.8 us of launch boundary each.  Rows [0, rows0) belong to
    // the first tensor, rows [more[i].row_begin, next begin) to further tensor i.
    int64_t rows0;
    int n_more;
    TensorSlot more[MAX_MORE];
    RowPitch xp, yp;   // of the first tensor (x / y); the further tensors carry their own
};

// STE bit mask: a plain bitmap per row.  Row r starts at mask + r * mask_row_words 64-bit words (mask_row_words =
// ceil(cols / 64)); bit (j % 8) of byte (j / 8) is the flag of element j: 1 = the backward zeroes the gradient
// (x >= hi || x <= lo; NaN compares false, i.e. passes the gradient, as in the reference).  One layout for every forward
// (16-byte vectors of 16-bit or fp32 elements, the fp32-result kernel's 8-byte half-vectors) and every backward, so any
// mask-consuming kernel can serve any mask-producing one.  Everything is lane-local: a lane derives the flags of its
// own vector and stores its own byte (bf16 / fp16: 8 elements = one byte; 4-element lanes -- fp32 vectors, half-vectors --
// hand their nibble to the even neighbour first), and the backward lane loads that byte back next to its gradient.
// Round 2 built the words with one __ballot per element position (8 v_cmp + 16 v_cndmask per vector to route the
// ballots to their storing lanes, and 16 v_readlane per vector in the backward); this costs 15 / 13 VALU ops per vector.
//
// Integer form of the predicate for 16-bit tensors (the usual clip: lo == -hi, hi >= 0, no NaN in the row): with
// a = bits & 0x7FFF (sign-magnitude order, |x| as an integer) and T = bits(hi),  |x| >= hi  <=>  a >= T  <=>  bit 15 of
// a + (0x8000 - T); both halves of a dword at once with one AND and one ADD (no carry crosses the halves: a + K <= 0xFFFF).
// v_perm_b32 gathers the four high bytes of two dwords, v_dot4_u32_u8 weighs the flag bits (bit 7 of each byte) into place.
// clipk = (0x8000 - T) in both halves; 0 = not applicable (host: fq_launch.h ste_clip_key).
__device__ __forceinline__ uint32_t ste_flags16_pair(uint32_t w0, uint32_t w1, uint32_t clipk) {   // 4 elements -> low nibble
    const uint32_t s0 = (w0 & 0x7FFF7FFFu) + clipk, s1 = (w1 & 0x7FFF7FFFu) + clipk;
    const uint32_t p = __builtin_amdgcn_perm(s1, s0, 0x07050301u) & 0x80808080u;   // bytes: s0.b1, s0.b3, s1.b1, s1.b3
    return __builtin_amdgcn_udot4(p, 0x08040201u, 0u, false) >> 7;
}
__device__ __forceinline__ uint32_t ste_flags16_vec(const uint4& w, uint32_t clipk) {   // 8 elements -> one byte
    const uint32_t s0 = (w.x & 0x7FFF7FFFu) + clipk, s1 = (w.y & 0x7FFF7FFFu) + clipk;
    const uint32_t s2 = (w.z & 0x7FFF7FFFu) + clipk, s3 = (w.w & 0x7FFF7FFFu) + clipk;
    const uint32_t p = __builtin_amdgcn_perm(s1, s0, 0x07050301u) & 0x80808080u;
    const uint32_t q = __builtin_amdgcn_perm(s3, s2, 0x07050301u) & 0x80808080u;
    return __builtin_amdgcn_udot4(q, 0x80402010u, __builtin_amdgcn_udot4(p, 0x08040201u, 0u, false), false) >> 7;
}
// general form (any clip, NaN rows, fp32 tensors): compare the unpacked values
template <int N, bool SYMCLIP> __device__ __forceinline__ uint32_t ste_flags_f(const float (&f)[N], float lo, float hi) {
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const bool z = SYMCLIP ? (__builtin_fabsf(f[e]) >= hi) : ((f[e] >= hi) || (f[e] <= lo));
        m |= z ? (1u << e) : 0u;
    }
    return m;
}
// nibbles of a lane pair -> the even lane's byte (the odd lane's return value is not used)
__device__ __forceinline__ uint32_t ste_nibble_pair(uint32_t nib) { return nib | (dpp<0xB1>(nib) << 4); }   // quad_perm:[1,0,3,2]

// Flags of this lane's 16-byte vector `v` of the row (f = its unpacked elements, raw = its dwords) into the row's bitmap.
// in_range: v < nvec (lanes holding a clamped duplicate of the last vector contribute nothing).
template <int DT> __device__ __forceinline__ void ste_mask_record(uint8_t* mrow, int v, bool in_range, const uint4& raw,
                                                                  const float (&f)[16 / Ty<DT>::ESIZE], float lo, float hi, bool sym_clip,
                                                                  uint32_t clipk /* 0 unless the integer form applies to this row */) {
    if constexpr (Ty<DT>::ESIZE == 2) {
        uint32_t m;
        if (clipk) m = ste_flags16_vec(raw, clipk);          // wave-uniform choice
        else m = sym_clip ? ste_flags_f<8, true>(f, lo, hi) : ste_flags_f<8, false>(f, lo, hi);
        // (gathering a quad's four bytes into one dword store from its first lane -- 16 storing lanes instead of 64 -- measured
        // 1 % slower on the [2048,4096] launches: profiles/r03_ab_mask_store_dword.txt)
        if (in_range) mrow[v] = (uint8_t)m;
    } else {
        uint32_t nib = sym_clip ? ste_flags_f<4, true>(f, lo, hi) : ste_flags_f<4, false>(f, lo, hi);
        nib = in_range ? nib : 0u;
        const uint32_t byte = ste_nibble_pair(nib);
        if (in_range && !(v & 1)) mrow[v >> 1] = (uint8_t)byte;
    }
}

// Backward side: this lane's flag byte -> its gradient vector with the flagged elements zeroed.
//   16-bit: m2 = m | m << 15 holds flag k at bits k and k + 15, so (m2 << (15 - 2d)) has the flags of dword d's two
//   halves at bits 15 and 31; v_pk_ashrrev_i16 by 15 smears each into its half; v_bfi clears: 3 VALU ops per dword.
template <int DT> __device__ __forceinline__ uint4 ste_mask_apply(const uint4& g, uint32_t m /* 8 (4 for fp32) flag bits */) {
    uint32_t w[4] = {g.x, g.y, g.z, g.w};
    if constexpr (Ty<DT>::EPD == 1) {
#pragma unroll
        for (int d = 0; d < 4; ++d) w[d] = ((m >> d) & 1u) ? 0u : w[d];
    } else {
        typedef short s16x2_t __attribute__((ext_vector_type(2)));
        const uint32_t m2 = m | (m << 15);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const s16x2_t t = __builtin_bit_cast(s16x2_t, m2 << (15 - 2 * d));
            const uint32_t z = __builtin_bit_cast(uint32_t, t >> (short)15);   // v_pk_ashrrev_i16
            w[d] &= ~z;
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// (Which tensor of a multi-tensor launch a row belongs to is decided by a short branch chain in the kernels below: a
// branch-free variant -- all slots' kernarg fields loaded up front, chosen with selects, as the mask backward does -- measured
// 2-4 % SLOWER on the [2048,4096] forward launches, profiles/r03_ab_forward_pick_tensor.txt: the single-tensor launch pays one
// compare today.)
// ------------------------------------------------------------------------------------
// Register-resident row kernel.
//   nvec = cols / elements-per-16B must satisfy nvec <= TPR * VPT.
//   Thread t owns vectors t, t+TPR, ...: each wave-instruction reads 1 KiB contiguous.
//   Out-of-range slots re-load the row's last vector (idempotent for max/min), so no load
//   sits behind a branch; only stores are predicated.
// ------------------------------------------------------------------------------------
template <int DT, int TPR, int VPT, bool ASYM, bool FAST, bool NTL = true, bool NTS = true, bool DBG = false, int AC = 0, bool PITCH = false>
__global__ __launch_bounds__(TPR == 64 ? 256 : TPR) void row_reg_kernel(RowArgs a) {
    using T = Ty<DT>;
    static_assert(AC == 0 || (AC == 1 && !ASYM && !DBG && T::ESIZE == 2), "autocast arithmetic: Sym on 16-bit tensors");
    constexpr int EPV = 16 / T::ESIZE;
    constexpr int NW = TPR / 64;
    __shared__ uint32_t red[3][NW > 1 ? NW : 1];
    // AsymQuantizer on fp16 tensors at <= 8 bits (FAST): the chain behind the bin index -- three of its six roundings and one
    // of its two exact divisions -- depends only on (bin, row), so each row tabulates its <= 256 dequantized values in LDS
    // once and the elementwise pass ends in a lookup.  Measured (tools/shape_sweep.py): fp16 [4096,11008] 43 -> 34 us,
    // [2048,11008] 24.0 -> 21.4 us.  bf16 keeps the arithmetic chain: its divisions are already reciprocal multiplies
    // (AFAST) and the eight 2-byte LDS reads per vector cost what the saved VALU gains (19.4 -> 20.3 us with the table).
    constexpr bool ALUT = ASYM && FAST && DT == F16 && !DBG;
    constexpr bool AFAST = FAST && DT == BF16;
    __shared__ uint16_t lut[ALUT ? (TPR == 64 ? 4 : 1) : 1][ALUT ? 256 : 1];

    int64_t row;
    int t;
    if constexpr (TPR == 64) {
        row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        t = threadIdx.x & 63;
        if (row >= a.rows) return;  // wave-uniform
    } else {
        row = blockIdx.x;
        t = threadIdx.x;
    }
    // which tensor of the launch this row belongs to (wave-uniform)
    const void* xb = a.x;
    void* yb = a.y;
    float* bnd = a.bounds;
    uint64_t* msk = a.mask;
    SymConst symk = a.sym;
    RowPitch xp = a.xp, yp = a.yp;
    if (row >= a.rows0) {
        int64_t rbase = 0;
#pragma unroll
        for (int i = 0; i < MAX_MORE; ++i) {  // static indices: scalar selects, the slots stay in SGPRs / kernarg loads
            if (i < a.n_more && row >= a.more[i].row_begin) {
                rbase = a.more[i].row_begin;
                xb = a.more[i].x;
                yb = a.more[i].y;
                bnd = a.more[i].bounds;
                msk = a.more[i].mask;
                symk.qmax = a.more[i].qmax;
                if constexpr (PITCH) {
                    xp = a.more[i].xp;
                    yp = a.more[i].yp;
                }
            }
        }
        row -= rbase;
    }
    const int nvec = (int)(a.cols / EPV);
    const uint4* __restrict__ xr = (const uint4*)((const char*)xb + row_byte_off<PITCH>(row, a.cols * T::ESIZE, xp));
    uint4* __restrict__ yr = (uint4*)((char*)yb + row_byte_off<PITCH>(row, a.cols * T::ESIZE, yp));

    uint4 r[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        int v = t + i * TPR;
        v = v < nvec ? v : nvec - 1;
        r[i] = ld16<NTL>(&xr[v]);
    }

    SymRow sr;
    AsymRow ar;
    float ub, lb;  // bounds of the row's values
    if constexpr (!ASYM) {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            acc = T::absmax_acc(acc, r[i].x);
            acc = T::absmax_acc(acc, r[i].y);
            acc = T::absmax_acc(acc, r[i].z);
            acc = T::absmax_acc(acc, r[i].w);
        }
        const uint32_t mbits = block_reduce<OpMaxU, NW>(T::absmax_finish(acc), red[0]);
        const float m = as_f(mbits);
        if constexpr (AC == 0) sr = sym_row<DT>(m, symk);
        else sr = sym_row_autocast<DT>(m, symk);
        ub = m;
        lb = -m;
        if (t == 0) {
            if (DBG && a.scale) a.scale[row] = sr.s;
            if (bnd) {
                bnd[2 * row] = m;
                bnd[2 * row + 1] = -m;
            }
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
        if (t == 0) {
            if (DBG && a.scale) {
                a.scale[2 * row] = ar.al;
                a.scale[2 * row + 1] = ar.mn;
            }
            if (bnd) {
                bnd[2 * row] = mx;
                bnd[2 * row + 1] = mn;
            }
        }
    }
    // A row with a NaN / Inf among {alpha + 1e-8, beta} has NaN bins: it takes the plain chain (wave-uniform).
    bool use_lut = false;
    if constexpr (ALUT) {
        // ... as does a row with alpha + 1e-8 == 0 (1e-8 rounds to 0 in fp16: a constant row divides 0 by 0)
        use_lut = (ar.a - ar.a == 0.0f) && (ar.mn - ar.mn == 0.0f) && ar.a > 0.0f;
        if (use_lut) {
            uint16_t* L = lut[TPR == 64 ? (threadIdx.x >> 6) : 0];
            const int nbins = (int)a.asym.S + 1;  // <= 256
            for (int i = t; i < nbins; i += TPR) {
                float y[2] = {asym_second_half<DT, AFAST>((float)i, ar, a.asym), 0.f};
                L[i] = (uint16_t)T::pack(y);
            }
        }
        // the row's own threads wrote the table: one wave (TPR == 64) sees its LDS writes in order; a workgroup needs the
        // barrier (use_lut is uniform over the row's threads, but the barrier is taken unconditionally)
        if constexpr (NW > 1) __syncthreads();
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }

    // Elementwise pass.  Rows that can actually be clipped also emit the STE bit mask for the backward.
    const bool want_mask = msk && !((ub < a.hi) && (lb > a.lo));  // wave-uniform
    const bool sym_clip = a.lo == -a.hi;
    const uint32_t clipk = (ub != ub) ? 0u : a.clipk;  // a row with a NaN compares as floats (NaN passes the gradient)
    uint8_t* mrow = (uint8_t*)(msk + row * a.mask_row_words);
    int32_t* idxr = (DBG && a.idx) ? a.idx + row * a.cols : nullptr;
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
        if constexpr (AC != 0) {  // fp32 arithmetic behind the reciprocal, as autocast makes the reference do
#pragma unroll
            for (int e = 0; e < EPV; ++e) f[e] = sym_elem_autocast(f[e], sr);
            uint32_t o[4];  // rounded once to the tensor dtype (the fp32 result has its own kernel below)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float fd[T::EPD];
#pragma unroll
                for (int k = 0; k < T::EPD; ++k) fd[k] = f[d * T::EPD + k];
                o[d] = T::pack(fd);
            }
            if (v < nvec) st16<NTS>(&yr[v], make_uint4(o[0], o[1], o[2], o[3]));
            continue;
        }
        uint32_t o[4];
        int32_t ib[EPV];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float fd[T::EPD];
#pragma unroll
            for (int k = 0; k < T::EPD; ++k) fd[k] = f[d * T::EPD + k];
            int32_t* ip = (DBG && idxr) ? ib + d * T::EPD : nullptr;
            if constexpr (!ASYM) o[d] = sym_chain<DT, FAST>(fd, sr, ip);
            else if constexpr (ALUT) {
                if (use_lut) {
                    const uint16_t* L = lut[TPR == 64 ? (threadIdx.x >> 6) : 0];
                    // p + 1.5 * 2^23: the (already integral) bin as an integer in the low mantissa bits
                    const uint32_t i0 = as_u(asym_first_half<DT, AFAST>(fd[0], ar, a.asym) + 12582912.0f) & 0xFFu;
                    const uint32_t i1 = as_u(asym_first_half<DT, AFAST>(fd[1], ar, a.asym) + 12582912.0f) & 0xFFu;
                    o[d] = (uint32_t)L[i0] | ((uint32_t)L[i1] << 16);
                } else {
                    o[d] = asym_chain<DT, AFAST>(fd, ar, a.asym, ip);
                }
            } else o[d] = asym_chain<DT, FAST && DT == BF16>(fd, ar, a.asym, ip);
        }
        if (v < nvec) {
            st16<NTS>(&yr[v], make_uint4(o[0], o[1], o[2], o[3]));
            if constexpr (DBG) {
                if (idxr) {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) idxr[(int64_t)v * EPV + e] = ib[e];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Autocast, fp32 result (AC == 2 semantics) with fully coalesced traffic: a 16-bit row is read in 8-byte
// half-vectors (4 elements per lane, 512 B contiguous per wave-instruction) so that each lane's 4 fp32 results
// are one 16-byte store and a wave-instruction writes 1 KiB contiguous.  (Keeping the 16-byte loads makes every
// store instruction touch half of each 128-byte line: 99 us instead of ~50 us on the 90 MB tensor.)
// Records row bounds and, on request, the STE mask (the row bitmap every kernel shares: a lane's 4 elements are one
// nibble, lane pairs assemble a byte).  Its usual consumer is ste_mask_wide_kernel, whose fp32 gradient is read with
// the same 4-elements-per-lane mapping.
// Serves two tensors per launch like row_reg_kernel (the K and V hooks: modeling_llama_quant.py:320-327).
// ------------------------------------------------------------------------------------
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ uint2 ld8(const uint2* p) {
    if constexpr (NT) {
        u32x2_t v = __builtin_nontemporal_load((const u32x2_t*)p);
        return make_uint2(v.x, v.y);
    } else {
        return *p;
    }
}
template <bool NT> __device__ __forceinline__ void st8(uint2* p, uint2 v) {
    if constexpr (NT) {
        u32x2_t w = {v.x, v.y};
        __builtin_nontemporal_store(w, (u32x2_t*)p);
    } else {
        *p = v;
    }
}

template <int DT, int TPR, int HPT, bool NTL, bool NTS, bool MASK, bool PITCH = false>
__global__ __launch_bounds__(TPR == 64 ? 256 : TPR) void row_reg_wide_kernel(RowArgs a) {
    using T = Ty<DT>;
    static_assert(T::ESIZE == 2, "16-bit input, fp32 output");
    constexpr int NW = TPR / 64;
    __shared__ uint32_t red[NW > 1 ? NW : 1];
    int64_t row;
    int t;
    if constexpr (TPR == 64) {
        row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        t = threadIdx.x & 63;
        if (row >= a.rows) return;
    } else {
        row = blockIdx.x;
        t = threadIdx.x;
    }
    const void* xb = a.x;  // which tensor of the launch this row belongs to (wave-uniform), as in row_reg_kernel
    void* yb = a.y;
    float* bnd = a.bounds;
    uint64_t* msk = a.mask;
    float qmax = a.sym.qmax;
    RowPitch xp = a.xp, yp = a.yp;
    if (row >= a.rows0) {
        int64_t rbase = 0;
#pragma unroll
        for (int i = 0; i < MAX_MORE; ++i) {
            if (i < a.n_more && row >= a.more[i].row_begin) {
                rbase = a.more[i].row_begin;
                xb = a.more[i].x;
                yb = a.more[i].y;
                bnd = a.more[i].bounds;
                msk = a.more[i].mask;
                qmax = a.more[i].qmax;
                if constexpr (PITCH) {
                    xp = a.more[i].xp;
                    yp = a.more[i].yp;
                }
            }
        }
        row -= rbase;
    }
    const int nh = (int)(a.cols / 4);
    const uint2* __restrict__ xr = (const uint2*)((const char*)xb + row_byte_off<PITCH>(row, a.cols * 2, xp));
    uint4* __restrict__ yr = (uint4*)((char*)yb + row_byte_off<PITCH>(row, a.cols * 4, yp));
    uint2 r[HPT];
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        int h = t + i * TPR;
        h = h < nh ? h : nh - 1;
        r[i] = ld8<NTL>(&xr[h]);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < HPT; ++i) acc = T::absmax_acc(T::absmax_acc(acc, r[i].x), r[i].y);
    const float m = as_f(block_reduce<OpMaxU, NW>(T::absmax_finish(acc), red));
    const SymRow sr = sym_row_autocast<DT>(m, SymConst{qmax, a.sym.c6});
    if (t == 0 && bnd) {
        bnd[2 * row] = m;
        bnd[2 * row + 1] = -m;
    }
    const bool want_mask = MASK && msk && !((m < a.hi) && (-m > a.lo));  // wave-uniform; NaN row: mask written, all bits 0
    const bool sym_clip = a.lo == -a.hi;
    const uint32_t clipk = (m != m) ? 0u : a.clipk;
    uint8_t* mrow = (uint8_t*)(msk + row * a.mask_row_words);
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        const int h = t + i * TPR;
        float f[4];
        {
            float f0[2], f1[2];
            T::unpack(r[i].x, f0);
            T::unpack(r[i].y, f1);
            f[0] = f0[0], f[1] = f0[1], f[2] = f1[0], f[3] = f1[1];
        }
        if (MASK && want_mask) {  // half-vector h = elements 4h .. 4h+3 = nibble (h & 1) of the bitmap's byte h / 2
            uint32_t nib;
            if (clipk) nib = ste_flags16_pair(r[i].x, r[i].y, clipk);
            else nib = sym_clip ? ste_flags_f<4, true>(f, a.lo, a.hi) : ste_flags_f<4, false>(f, a.lo, a.hi);
            nib = h < nh ? nib : 0u;
            const uint32_t byte = ste_nibble_pair(nib);
            if (h < nh && !(h & 1)) mrow[h >> 1] = (uint8_t)byte;
        }
        const uint4 o = make_uint4(as_u(sym_elem_autocast(f[0], sr)), as_u(sym_elem_autocast(f[1], sr)),
                                   as_u(sym_elem_autocast(f[2], sr)), as_u(sym_elem_autocast(f[3], sr)));
        if (h < nh) st16<NTS>(&yr[h], o);
    }
}

// ------------------------------------------------------------------------------------
// scalar element chains (generic / two-pass-unaligned paths)
// ------------------------------------------------------------------------------------
template <int DT, bool FAST> __device__ __forceinline__ float sym_elem(float x, const SymRow& r, int32_t* idx) {
    using T = Ty<DT>;
    float q = __builtin_rintf(T::rb(x * r.s));
    if (idx) *idx = idx_i32(q);
    return FAST ? q * r.rinv : q / r.t2;  // stored through T::store1 -> rounded to dtype
}
template <int DT> __device__ __forceinline__ float asym_elem(float x, const AsymRow& r, const AsymConst& k, int32_t* idx) {
    using T = Ty<DT>;
    float n = T::rb(T::rb(x - r.mn) / r.a);
    float q = __builtin_rintf(T::rb(n * k.S));
    if (idx) *idx = idx_i32(q);
    float w = T::rb(k.mul_inv ? q * k.invS : q / k.S);
    return T::rb(w * r.a) + r.mn;
}

// Any width / alignment; TPR threads sweep the row twice.
template <int DT, int TPR, bool ASYM, int AC = 0>
__global__ __launch_bounds__(TPR == 64 ? 256 : TPR) void row_generic_kernel(RowArgs a) {
    using T = Ty<DT>;
    static_assert(AC == 0 || (!ASYM && T::ESIZE == 2), "autocast arithmetic: Sym on 16-bit tensors");
    constexpr int NW = TPR / 64;
    __shared__ uint32_t red[3][NW > 1 ? NW : 1];
    int64_t row;
    int t;
    if constexpr (TPR == 64) {
        row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        t = threadIdx.x & 63;
        if (row >= a.rows) return;
    } else {
        row = blockIdx.x;
        t = threadIdx.x;
    }
    const int64_t base = row * a.cols;   // (the debug index output is always contiguous)
    const int64_t cols = a.cols;
    const int64_t xbase = row_byte_off(row, cols * T::ESIZE, a.xp) / T::ESIZE;                                       // in elements
    const int64_t ybase = row_byte_off(row, cols * (AC == 2 ? 4 : T::ESIZE), a.yp) / (AC == 2 ? 4 : T::ESIZE);
    const float first = T::load1(a.x, xbase);  // cols >= 1 guaranteed by the host
    SymRow sr;
    AsymRow ar;
    if constexpr (!ASYM) {
        uint32_t acc = 0;
        for (int64_t c = t; c < cols; c += TPR) {
            uint32_t b = as_u(T::load1(a.x, xbase + c)) & 0x7FFFFFFFu;
            acc = acc > b ? acc : b;
        }
        const float m = as_f(block_reduce<OpMaxU, NW>(acc, red[0]));
        if constexpr (AC == 0) sr = sym_row<DT>(m, a.sym);
        else sr = sym_row_autocast<DT>(m, a.sym);
        if (t == 0) {
            if (a.scale) a.scale[row] = sr.s;
            if (a.bounds) {
                a.bounds[2 * row] = m;
                a.bounds[2 * row + 1] = -m;
            }
        }
    } else {
        float mx = first, mn = first;
        uint32_t acc = 0;
        for (int64_t c = t; c < cols; c += TPR) {
            float v = T::load1(a.x, xbase + c);
            mx = __builtin_fmaxf(mx, v);
            mn = __builtin_fminf(mn, v);
            uint32_t b = as_u(v) & 0x7FFFFFFFu;
            acc = acc > b ? acc : b;
        }
        const uint32_t nb = block_reduce<OpMaxU, NW>(acc, red[0]);
        mx = as_f(block_reduce<OpMaxF, NW>(as_u(mx), red[1]));
        mn = as_f(block_reduce<OpMinF, NW>(as_u(mn), red[2]));
        if (absbits_is_nan(nb)) mx = mn = as_f(0x7FC00000u);
        ar = asym_row<DT>(mx, mn, a.asym);
        if (t == 0) {
            if (a.scale) {
                a.scale[2 * row] = ar.al;
                a.scale[2 * row + 1] = ar.mn;
            }
            if (a.bounds) {
                a.bounds[2 * row] = mx;
                a.bounds[2 * row + 1] = mn;
            }
        }
    }
    for (int64_t c = t; c < cols; c += TPR) {
        const float v = T::load1(a.x, xbase + c);
        int32_t* ip = a.idx ? a.idx + base + c : nullptr;
        float o;
        if constexpr (AC != 0) {
            o = sym_elem_autocast(v, sr);
            if constexpr (AC == 2) {
                ((float*)a.y)[ybase + c] = o;
                continue;
            }
        } else if constexpr (!ASYM) o = sym_elem<DT, false>(v, sr, ip);
        else o = asym_elem<DT>(v, ar, a.asym, ip);
        T::store1(a.y, ybase + c, o);
    }
}

// ------------------------------------------------------------------------------------
// Two-pass path: pass 1 reduces fixed-size chunks and merges them with atomicMax on
// order-preserving keys (ws[row] = {max key, ~min key}, zero-initialised by the host with
// hipMemsetAsync); pass 2 applies.  Sym uses ws[2*row] = fp32 bits of max|x|.
// ------------------------------------------------------------------------------------
constexpr int TP_THREADS = 256;
constexpr int TP_VPT = 4;                              // 16-byte vectors per thread per chunk (VEC path)
constexpr int TP_EPT = 16;                             // elements per thread per chunk (scalar path)

template <int DT, bool VEC> __host__ __device__ constexpr int tp_chunk_elems() {
    return VEC ? TP_THREADS * TP_VPT * (16 / Ty<DT>::ESIZE) : TP_THREADS * TP_EPT;
}

template <int DT, bool ASYM, bool VEC>
__global__ __launch_bounds__(TP_THREADS) void stats_kernel(RowArgs a, uint32_t* __restrict__ ws, int64_t chunks) {
    using T = Ty<DT>;
    constexpr int EPV = 16 / T::ESIZE;
    constexpr int CH = tp_chunk_elems<DT, VEC>();
    __shared__ uint32_t red[3][TP_THREADS / 64];
    const int64_t row = blockIdx.x / chunks;
    const int64_t c0 = (blockIdx.x % chunks) * CH;
    const int64_t base = row * a.cols;
    const int t = threadIdx.x;
    int64_t cend = c0 + CH;
    if (cend > a.cols) cend = a.cols;
    uint32_t acc = 0;
    float mx, mn;
    if constexpr (VEC) {
        const uint4* xr = (const uint4*)((const char*)a.x + (base + c0) * T::ESIZE);
        const int nvec = (int)((cend - c0) / EPV);  // >= 1
        uint4 r[TP_VPT];
#pragma unroll
        for (int i = 0; i < TP_VPT; ++i) {
            int v = t + i * TP_THREADS;
            v = v < nvec ? v : nvec - 1;
            r[i] = xr[v];
        }
        if constexpr (!ASYM) {
#pragma unroll
            for (int i = 0; i < TP_VPT; ++i) {
                acc = T::absmax_acc(acc, r[i].x);
                acc = T::absmax_acc(acc, r[i].y);
                acc = T::absmax_acc(acc, r[i].z);
                acc = T::absmax_acc(acc, r[i].w);
            }
            acc = T::absmax_finish(acc);
        } else {
            MinMax mm;
            float f0[T::EPD];
            T::unpack(r[0].x, f0);
            mm.mx = mm.mn = f0[0];
            mm.absacc = 0;
#pragma unroll
            for (int i = 0; i < TP_VPT; ++i) {
                minmax_acc<DT>(mm, r[i].x);
                minmax_acc<DT>(mm, r[i].y);
                minmax_acc<DT>(mm, r[i].z);
                minmax_acc<DT>(mm, r[i].w);
            }
            acc = T::absmax_finish(mm.absacc);
            mx = mm.mx;
            mn = mm.mn;
        }
    } else {
        mx = mn = T::load1(a.x, base + c0);
        for (int64_t c = c0 + t; c < cend; c += TP_THREADS) {
            float v = T::load1(a.x, base + c);
            uint32_t b = as_u(v) & 0x7FFFFFFFu;
            acc = acc > b ? acc : b;
            if constexpr (ASYM) {
                mx = __builtin_fmaxf(mx, v);
                mn = __builtin_fminf(mn, v);
            }
        }
    }
    acc = block_reduce<OpMaxU, TP_THREADS / 64>(acc, red[0]);
    if constexpr (!ASYM) {
        if (t == 0) atomicMax(&ws[2 * row], acc);
    } else {
        mx = as_f(block_reduce<OpMaxF, TP_THREADS / 64>(as_u(mx), red[1]));
        mn = as_f(block_reduce<OpMinF, TP_THREADS / 64>(as_u(mn), red[2]));
        if (t == 0) {
            const bool nan = absbits_is_nan(acc);
            atomicMax(&ws[2 * row], nan ? 0xFFFFFFFFu : okey(mx));
            atomicMax(&ws[2 * row + 1], nan ? 0xFFFFFFFFu : ~okey(mn));
        }
    }
}

template <int DT, bool ASYM, bool FAST, bool VEC>
__global__ __launch_bounds__(TP_THREADS) void apply_kernel(RowArgs a, const uint32_t* __restrict__ ws, int64_t chunks) {
    using T = Ty<DT>;
    constexpr int EPV = 16 / T::ESIZE;
    constexpr int CH = tp_chunk_elems<DT, VEC>();
    const int64_t row = blockIdx.x / chunks;
    const int64_t chunk = blockIdx.x % chunks;
    const int64_t c0 = chunk * CH;
    const int64_t base = row * a.cols;
    const int t = threadIdx.x;
    int64_t cend = c0 + CH;
    if (cend > a.cols) cend = a.cols;

    SymRow sr;
    AsymRow ar;
    if constexpr (!ASYM) {
        const float m = as_f(ws[2 * row]);
        sr = sym_row<DT>(m, a.sym);
        if (chunk == 0 && t == 0) {
            if (a.scale) a.scale[row] = sr.s;
            if (a.bounds) {
                a.bounds[2 * row] = m;
                a.bounds[2 * row + 1] = -m;
            }
        }
    } else {
        const uint32_t k0 = ws[2 * row], k1 = ws[2 * row + 1];
        float mx, mn;
        if (k0 == 0xFFFFFFFFu || k1 == 0xFFFFFFFFu) mx = mn = as_f(0x7FC00000u);
        else {
            mx = okey_inv(k0);
            mn = okey_inv(~k1);
        }
        ar = asym_row<DT>(mx, mn, a.asym);
        if (chunk == 0 && t == 0) {
            if (a.scale) {
                a.scale[2 * row] = ar.al;
                a.scale[2 * row + 1] = ar.mn;
            }
            if (a.bounds) {
                a.bounds[2 * row] = mx;
                a.bounds[2 * row + 1] = mn;
            }
        }
    }

    if constexpr (VEC) {
        const uint4* xr = (const uint4*)((const char*)a.x + (base + c0) * T::ESIZE);
        uint4* yr = (uint4*)((char*)a.y + (base + c0) * T::ESIZE);
        int32_t* idxr = a.idx ? a.idx + base + c0 : nullptr;
        const int nvec = (int)((cend - c0) / EPV);
        uint4 r[TP_VPT];
#pragma unroll
        for (int i = 0; i < TP_VPT; ++i) {
            int v = t + i * TP_THREADS;
            v = v < nvec ? v : nvec - 1;
            r[i] = xr[v];
        }
#pragma unroll
        for (int i = 0; i < TP_VPT; ++i) {
            const int v = t + i * TP_THREADS;
            uint4 o;
            int32_t ib[EPV];
            if constexpr (!ASYM) {
                o.x = sym_dword<DT, FAST && DT == BF16>(r[i].x, sr, idxr ? ib + 0 * T::EPD : nullptr);
                o.y = sym_dword<DT, FAST && DT == BF16>(r[i].y, sr, idxr ? ib + 1 * T::EPD : nullptr);
                o.z = sym_dword<DT, FAST && DT == BF16>(r[i].z, sr, idxr ? ib + 2 * T::EPD : nullptr);
                o.w = sym_dword<DT, FAST && DT == BF16>(r[i].w, sr, idxr ? ib + 3 * T::EPD : nullptr);
            } else {
                o.x = asym_dword<DT, FAST && DT == BF16>(r[i].x, ar, a.asym, idxr ? ib + 0 * T::EPD : nullptr);
                o.y = asym_dword<DT, FAST && DT == BF16>(r[i].y, ar, a.asym, idxr ? ib + 1 * T::EPD : nullptr);
                o.z = asym_dword<DT, FAST && DT == BF16>(r[i].z, ar, a.asym, idxr ? ib + 2 * T::EPD : nullptr);
                o.w = asym_dword<DT, FAST && DT == BF16>(r[i].w, ar, a.asym, idxr ? ib + 3 * T::EPD : nullptr);
            }
            if (v < nvec) {
                yr[v] = o;
                if (idxr) {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) idxr[(int64_t)v * EPV + e] = ib[e];
                }
            }
        }
    } else {
        for (int64_t c = c0 + t; c < cend; c += TP_THREADS) {
            const float v = T::load1(a.x, base + c);
            int32_t* ip = a.idx ? a.idx + base + c : nullptr;
            float o;
            if constexpr (!ASYM) o = sym_elem<DT, FAST && DT == BF16>(v, sr, ip);
            else o = asym_elem<DT>(v, ar, a.asym, ip);
            T::store1(a.y, base + c, o);
        }
    }
}

// Two-pass apply with the autocast arithmetic (rows too long for the single-pass kernels: layerwise under
// autocast).  Correctness path: element loads, one chunk of TP_THREADS * TP_EPT elements per workgroup.
template <int DT, bool WIDE>
__global__ __launch_bounds__(TP_THREADS) void apply_autocast_kernel(RowArgs a, const uint32_t* __restrict__ ws, int64_t chunks) {
    using T = Ty<DT>;
    constexpr int CH = TP_THREADS * TP_EPT;
    const int64_t row = blockIdx.x / chunks;
    const int64_t chunk = blockIdx.x % chunks;
    const int64_t base = row * a.cols;
    int64_t cend = (chunk + 1) * CH;
    if (cend > a.cols) cend = a.cols;
    const float m = as_f(ws[2 * row]);
    const SymRow sr = sym_row_autocast<DT>(m, a.sym);
    if (chunk == 0 && threadIdx.x == 0 && a.bounds) {
        a.bounds[2 * row] = m;
        a.bounds[2 * row + 1] = -m;
    }
    for (int64_t c = chunk * CH + threadIdx.x; c < cend; c += TP_THREADS) {
        const float o = sym_elem_autocast(T::load1(a.x, base + c), sr);
        if constexpr (WIDE) ((float*)a.y)[base + c] = o;
        else T::store1(a.y, base + c, o);
    }
}

// ------------------------------------------------------------------------------------
// STE backward (utils_quant.py:83-87).  VEC: n is a whole number of 16-byte vectors and all
// three pointers are 16-byte aligned; each thread moves UNR vectors of g and of x.
// ------------------------------------------------------------------------------------
constexpr int STE_THREADS = 256;

template <int DT, int UNR, bool NTL = true, bool NTS = true>
__global__ __launch_bounds__(STE_THREADS) void ste_vec_kernel(const uint4* __restrict__ g, const uint4* __restrict__ x,
                                                              uint4* __restrict__ gx, int64_t nvec, float lo, float hi) {
    const int64_t v0 = (int64_t)blockIdx.x * (STE_THREADS * UNR) + threadIdx.x;
    uint4 rg[UNR], rx[UNR];
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
        int64_t v = v0 + (int64_t)i * STE_THREADS;
        v = v < nvec ? v : nvec - 1;
        rg[i] = ld16<NTL>(&g[v]);
        rx[i] = ld16<NTL>(&x[v]);
    }
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
        const int64_t v = v0 + (int64_t)i * STE_THREADS;
        uint4 o;
        o.x = ste_dword<DT>(rg[i].x, rx[i].x, lo, hi);
        o.y = ste_dword<DT>(rg[i].y, rx[i].y, lo, hi);
        o.z = ste_dword<DT>(rg[i].z, rx[i].z, lo, hi);
        o.w = ste_dword<DT>(rg[i].w, rx[i].w, lo, hi);
        if (v < nvec) st16<NTS>(&gx[v], o);
    }
}

template <int DT>
__global__ __launch_bounds__(STE_THREADS) void ste_scalar_kernel(const void* __restrict__ g, const void* __restrict__ x,
                                                                 void* __restrict__ gx, int64_t n, float lo, float hi) {
    using T = Ty<DT>;
    const int64_t stride = (int64_t)gridDim.x * STE_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * STE_THREADS + threadIdx.x; i < n; i += stride) {
        const float v = T::load1(x, i);
        const bool masked = (v >= hi) || (v <= lo);
        if constexpr (T::ESIZE == 4) ((uint32_t*)gx)[i] = masked ? 0u : ((const uint32_t*)g)[i];
        else ((uint16_t*)gx)[i] = masked ? (uint16_t)0 : ((const uint16_t*)g)[i];
    }
}

// Row-aware STE: block b handles chunk (b % chunks) of row (b / chunks); a chunk is `cv` vectors
// (cv <= STE_THREADS * VPT; the host balances chunks so no block is nearly empty).  If the row's
// recorded bounds are strictly inside (lo, hi) no element can be masked: copy g, never touch x.
struct StePitch3 {
    RowPitch g, x, o;   // of grad_output, the forward's input and the result (RowPitch above)
};
template <int DT, int VPT, bool NTL = true, bool NTS = true, bool PITCH = false>
__global__ __launch_bounds__(STE_THREADS) void ste_rows_kernel(const void* __restrict__ g, const void* __restrict__ x,
                                                               void* __restrict__ gx, int64_t nvec_row, int64_t chunks, int cv,
                                                               const float* __restrict__ bounds, float lo, float hi, StePitch3 p) {
    const int64_t row = blockIdx.x / chunks;
    const int64_t vs = (blockIdx.x % chunks) * cv;
    const uint4* gr = (const uint4*)((const char*)g + row_byte_off<PITCH>(row, nvec_row * 16, p.g)) + vs;
    const uint4* xr = (const uint4*)((const char*)x + row_byte_off<PITCH>(row, nvec_row * 16, p.x)) + vs;
    uint4* or_ = (uint4*)((char*)gx + row_byte_off<PITCH>(row, nvec_row * 16, p.o)) + vs;
    const int64_t rem = nvec_row - vs;
    const int nvec = (int)(rem < cv ? rem : cv);
    bool safe = false;   // no recorded bounds (pitched tensors in the plain data flow): every row re-reads x
    if (bounds) {
        const float ub = bounds[2 * row], lb = bounds[2 * row + 1];
        safe = (ub < hi) && (lb > lo);  // false when a bound is NaN
    }
    const int t = threadIdx.x;
    uint4 rg[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        int v = t + i * STE_THREADS;
        v = v < nvec ? v : nvec - 1;
        rg[i] = ld16<NTL>(&gr[v]);
    }
    if (safe) {  // block-uniform
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = t + i * STE_THREADS;
            if (v < nvec) st16<NTS>(&or_[v], rg[i]);
        }
    } else {
        uint4 rx[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            int v = t + i * STE_THREADS;
            v = v < nvec ? v : nvec - 1;
            rx[i] = ld16<NTL>(&xr[v]);
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = t + i * STE_THREADS;
            uint4 o;
            o.x = ste_dword<DT>(rg[i].x, rx[i].x, lo, hi);
            o.y = ste_dword<DT>(rg[i].y, rx[i].y, lo, hi);
            o.z = ste_dword<DT>(rg[i].z, rx[i].z, lo, hi);
            o.w = ste_dword<DT>(rg[i].w, rx[i].w, lo, hi);
            if (v < nvec) st16<NTS>(&or_[v], o);
        }
    }
}


// 1-/2-bit weight branch, elementwise part (the per-row mean|w| scale is an input: it comes from the
// caller's own reduction so that the result stays bit-identical to the reference's).
// VEC: 16-byte vectors, one per thread; otherwise one element per thread (grid-stride).
template <int DT, int WBITS, bool VEC>
__global__ __launch_bounds__(256) void w12_kernel(const void* __restrict__ w, const void* __restrict__ scale, void* __restrict__ out,
                                                  int64_t rows, int64_t cols, int scale_per_row, float cv) {
    using T = Ty<DT>;
    constexpr int EPV = 16 / T::ESIZE;
    if constexpr (VEC) {
        const int64_t nvec_row = cols / EPV, nvec = rows * nvec_row;
        const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (v >= nvec) return;
        const W12Row wr = w12_row<DT>(T::load1(scale, scale_per_row ? v / nvec_row : 0), cv);
        const uint4 r = ((const uint4*)w)[v];
        const uint32_t in[4] = {r.x, r.y, r.z, r.w};
        uint4 o;
        if (__builtin_amdgcn_ballot_w64(!wr.mk) == 0) {  // every lane's row has an ordinary scale (a wave can straddle rows)
            o = w12_vec<DT, WBITS>(r, wr, cv);
        } else {
            uint32_t od[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) od[d] = w12_dword_any<DT, WBITS>(in[d], wr, cv);
            o = make_uint4(od[0], od[1], od[2], od[3]);
        }
        ((uint4*)out)[v] = o;
    } else {
        const int64_t n = rows * cols, stride = (int64_t)gridDim.x * 256;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const float sc = T::load1(scale, scale_per_row ? i / cols : 0);
            T::store1(out, i, w12_elem<DT, WBITS>(T::load1(w, i), sc, cv, 1.0f / sc, div_exact_ok(sc)));
        }
    }
}

// ------------------------------------------------------------------------------------
// The 1-/2-bit weight branch in ONE launch (read w, write out) with the per-row mean summed IN ATen'S OWN ORDER, so that the scale is
// bit-identical to what the reference computes on this device (`torch.mean(abs(w), dim=1, keepdim=True)`, utils_quant.py:205-209 /
// :219-224) and the one-launch path can be the default.  ATen's GPU reduction is a fixed tree for a given shape (third-party
// arithmetic, restated from torch 2.10's aten/src/ATen/native/cuda/Reduce.cuh: setReduceConfig, input_vectorized_thread_reduce_impl,
// block_x_reduce with its ROCm shuffle order, block_y_reduce, MeanOps::project; pinned by tests/test_gpu_features.py
// ::test_low_bit_fused_row_mean against live ATen: 0 differing rows).  For a contiguous [rows >= 8, cols] tensor with cols >= 256,
// cols % 4 == 0, reduced over its last dimension, ATen launches 512 threads as 8 waves of 64 lanes:
//   * elements are consumed in groups of 4 (ATen's input vector); ATen thread T owns groups T, T + S, T + 2S, ... and keeps FOUR
//     fp32 accumulators, one per position in the group, each summed sequentially from +0; then ((a0 + a1) + a2) + a3;
//   * cols <= 8128 (ATen's values_per_thread < 128): every wave reduces its OWN row, S = 64, T = lane;
//     cols >= 8129: the 8 waves share one row, S = 512, T = wave * 64 + lane;
//   * across the 64 lanes: v += shfl_down(v, 1), then 2, 4, ... 32 (increasing offsets on ROCm) -- lane 0 ends up with the
//     balanced tree over the lanes in order;  across the 8 waves (shared rows): ((y0 + y4) + (y2 + y6)) + ((y1 + y5) + (y3 + y7));
//   * mean = sum * float(1 / cols) (MeanOps::project: a multiply by the fp32 factor), rounded once to the tensor dtype.
// This kernel gives thread T exactly ATen thread T's elements: groups of 4 (8-byte accesses for 16-bit tensors, 16-byte for fp32),
// MAXG >= groups per thread held in registers; groups past the row's end are not loaded and add +0.
// (Rounds 2-3 had an opt-in kernel with its own summation order: <= 0.5 % of the rows of a [4096,11008] bf16 weight came out one
// bf16 ulp off ATen's, so it could not be the default; it is gone.)
// ------------------------------------------------------------------------------------
template <int DT> struct AtenGroup {   // bf16 / fp16: 4 elements = 8 bytes
    typedef uint2 type;
    template <bool NT> __device__ static __forceinline__ uint2 ld(const void* p, int64_t g) { return ld8<NT>((const uint2*)p + g); }
    template <bool NT> __device__ static __forceinline__ void st(void* p, int64_t g, uint2 v) { st8<NT>((uint2*)p + g, v); }
    __device__ static __forceinline__ void abs4(const uint2& v, float (&a)[4]) {
        float lo[2], hi[2];
        Ty<DT>::unpack(v.x, lo);
        Ty<DT>::unpack(v.y, hi);
        a[0] = __builtin_fabsf(lo[0]), a[1] = __builtin_fabsf(lo[1]), a[2] = __builtin_fabsf(hi[0]), a[3] = __builtin_fabsf(hi[1]);
    }
};
template <> struct AtenGroup<F32> {    // 4 elements = 16 bytes
    typedef uint4 type;
    template <bool NT> __device__ static __forceinline__ uint4 ld(const void* p, int64_t g) { return ld16<NT>((const uint4*)p + g); }
    template <bool NT> __device__ static __forceinline__ void st(void* p, int64_t g, uint4 v) { st16<NT>((uint4*)p + g, v); }
    __device__ static __forceinline__ void abs4(const uint4& v, float (&a)[4]) {
        a[0] = __builtin_fabsf(as_f(v.x)), a[1] = __builtin_fabsf(as_f(v.y)), a[2] = __builtin_fabsf(as_f(v.z)), a[3] = __builtin_fabsf(as_f(v.w));
    }
};

template <int DT, int WBITS, bool SHARED_ROW, int MAXG, bool NTL, bool NTS>
__global__ __launch_bounds__(512) void w12_row_aten_kernel(const void* __restrict__ w, void* __restrict__ out, void* __restrict__ scale_out,
                                                           int64_t rows, int64_t cols, float cv, float factor /* fp32 1 / cols */) {
    using T = Ty<DT>;
    using G = AtenGroup<DT>;
    typedef typename G::type group_t;
    __shared__ float ysum[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = SHARED_ROW ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 8 + wave;
    const int tid = SHARED_ROW ? (int)threadIdx.x : lane;     // ATen's thread index within the row's reduction
    constexpr int STEP = SHARED_ROW ? 512 : 64;
    if (!SHARED_ROW && row >= rows) return;                   // wave-uniform
    const int ngroups = (int)(cols / 4);
    const char* xr = (const char*)w + row * cols * T::ESIZE;
    char* yr = (char*)out + row * cols * T::ESIZE;
    group_t r[MAXG];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MAXG; ++k) {   // groups past the row's end are not loaded: they stay +0 and add nothing
        const int g = tid + k * STEP;
        r[k] = group_t{};
        if (g < ngroups) r[k] = G::template ld<NTL>(xr, g);
    }
#pragma unroll
    for (int k = 0; k < MAXG; ++k) {   // sequential in k, one accumulator per position in the group: ATen's value_list[0..3]
        float a[4];
        G::abs4(r[k], a);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = acc[i] + a[i];
    }
    float v = ((acc[0] + acc[1]) + acc[2]) + acc[3];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = v + __shfl_down(v, o, 64);   // block_x_reduce, ROCm order; lane 0 holds the wave's sum
    if constexpr (SHARED_ROW) {
        if (lane == 0) ysum[wave] = v;
        __syncthreads();
        v = ((ysum[0] + ysum[4]) + (ysum[2] + ysum[6])) + ((ysum[1] + ysum[5]) + (ysum[3] + ysum[7]));   // block_y_reduce
    } else {
        v = __shfl(v, 0, 64);
    }
    float sc = T::rb(v * factor);
    if constexpr (WBITS == 2) sc = T::rb(2.0f * sc);
    if (tid == 0 && scale_out) T::store1(scale_out, row, sc);
    const W12Row wr = w12_row<DT>(sc, cv);
#pragma unroll
    for (int k = 0; k < MAXG; ++k) {
        const int g = tid + k * STEP;
        group_t o;
        if constexpr (DT == F32) {
            o = wr.mk ? w12_vec<DT, WBITS>(r[k], wr, cv)
                      : make_uint4(w12_dword_any<DT, WBITS>(r[k].x, wr, cv), w12_dword_any<DT, WBITS>(r[k].y, wr, cv),
                                   w12_dword_any<DT, WBITS>(r[k].z, wr, cv), w12_dword_any<DT, WBITS>(r[k].w, wr, cv));
        } else {
            if (wr.mk) {   // row-uniform: an ordinary scale (every row of a real weight)
                bool odd = false;
                o.x = w12_dword<DT, WBITS>(r[k].x, wr, odd);
                o.y = w12_dword<DT, WBITS>(r[k].y, wr, odd);
                if constexpr (DT == BF16 && WBITS == 1) {
                    if (__builtin_amdgcn_ballot_w64(odd) != 0) {   // practically never taken: redo with the reference chain op for op
                        o.x = w12_dword_any<DT, WBITS>(r[k].x, wr, cv);
                        o.y = w12_dword_any<DT, WBITS>(r[k].y, wr, cv);
                    }
                }
            } else {       // zero / NaN / infinite / extreme scale
                o.x = w12_dword_any<DT, WBITS>(r[k].x, wr, cv);
                o.y = w12_dword_any<DT, WBITS>(r[k].y, wr, cv);
            }
        }
        if (g < ngroups) G::template st<NTS>(yr, g, o);
    }
}

// STE backward from the forward's bit mask: reads g (+ 1 bit/element of mask for rows that can be
// clipped), never x.  One launch serves up to 1 + MAX_MORE tensors of the same dtype and cols (a QuantizeLinear's weight and
// input, a sibling group).  Two kinds of slot:
//   copying (gx != g)   rows x chunks blocks, block b of the slot = chunk (b % chunks) of row (b / chunks), as ste_rows_kernel.
//   in place (gx == g)  a weight's gradient handed on by reference: rows whose bounds prove that nothing clips -- all of a
//                       weight's, in practice -- need nothing at all, so the slot gets one block per STE_THREADS rows: each thread
//                       reads one row's bounds, and the block then walks the (rare) clippable rows, masking them where they
//                       stand.  (Round 2 launched rows x chunks blocks that each read 8 bytes and left: 3.7 us of empty blocks
//                       for a q/k/v group.)  Meant for tensors whose rows rarely clip; an activation gradient belongs in a
//                       copying slot.
struct SteSlot {
    const void* g;
    void* gx;
    const float* bounds;
    const uint64_t* mask;
    int64_t rows;
    int64_t blk_begin;  // first block of this slot
    int inplace;
    RowPitch gp, op;    // rows of g / gx that do not follow one another (RowPitch above); an in-place slot: gp == op
};
struct SteLaunch {
    int n;
    SteSlot t[1 + MAX_MORE];
};

// The flag bits of a wave's SLOTS slots (slot i = the 64 lane-vectors of EPL elements each that the wave's lanes hold at
// t + i * STE_THREADS) with ONE dword load per lane and four (eight) slots -- issued next to the gradient loads, before the
// row's bounds have even arrived -- then one ds_bpermute_b32 per slot (the LDS crossbar, no memory access) to hand every lane
// the dword that holds its bits.  (A byte load per lane and slot costs a VMEM instruction each: 2-3 % slower on the
// [2048,4096] launches, profiles/r03_ab_mask_fetch.txt.)  Slot i's bits are a contiguous run of 64 * EPL / 8 bytes of the row
// bitmap starting at bit (first + (t & ~63) + i * STE_THREADS) * EPL.
// EPL = elements per lane and slot: 8 (16-bit vectors: a byte per lane) or 4 (fp32 vectors, half-vectors: a nibble).
// row_dwords: dwords of one bitmap row (reads are clamped to it; bits past the row's end belong to lanes that store nothing).
template <int EPL, int SLOTS> struct SteMaskHeld {
    static constexpr int RUN_DW = 64 * EPL / 32;          // dwords per slot run: 16 or 8
    static constexpr int PER_LOAD = 64 / RUN_DW;          // slots covered by one dword load per lane: 4 or 8
    static constexpr int NLOAD = (SLOTS + PER_LOAD - 1) / PER_LOAD;
    uint32_t held[NLOAD];
    __device__ __forceinline__ void load(const uint8_t* mrow, int64_t first, int row_dwords, int t) {
        const int lane = t & 63;
        const int64_t wave_first = first + (t - lane);  // multiple of 64 lane-vectors
        const uint32_t* md = (const uint32_t*)mrow;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int slot = k * PER_LOAD + lane / RUN_DW;
            int64_t dw = (wave_first + (int64_t)slot * STE_THREADS) * EPL / 32 + lane % RUN_DW;
            dw = dw < row_dwords ? dw : row_dwords - 1;
            held[k] = md[dw];
        }
    }
    // lane l's bits of a slot sit in dword (l * EPL / 32) of the run, at bit (l * EPL) % 32.  All 64 lanes must be active.
    __device__ __forceinline__ void bits(int t, uint32_t (&mb)[SLOTS]) const {
        const int lane = t & 63;
        const int src = (lane * EPL / 32) * 4, sh = (lane * EPL) % 32;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute(src + (i % PER_LOAD) * RUN_DW * 4, (int)held[i / PER_LOAD]);
            mb[i] = (w >> sh) & ((1u << EPL) - 1u);
        }
    }
};

// which tensor of the launch a block belongs to.  Branch-free on purpose: every kernarg field is loaded up front and chosen
// with scalar selects (a chain of `if (blockIdx >= begin_i) slot = t[i]` compiled to three dependent s_load / s_waitcnt /
// s_cbranch round trips in the prologue of every block).  Unused slots carry blk_begin = INT64_MAX (host).
// NS: how many slots the launch can have (4, or 2: a QuantizeLinear's weight + input, K + V -- the launch then loads and selects among
// half the kernarg fields; a one-tensor copying launch has a kernel of its own, ste_mask_one_kernel).
template <bool PITCH = false, int NS = 4> __device__ __forceinline__ SteSlot ste_pick_slot(const SteLaunch& L, int64_t b) {
    static_assert(MAX_MORE == 3 && (NS == 2 || NS == 4), "four slots, or the first two");
    SteSlot r;
    if constexpr (NS == 2) {
        const bool s1 = b >= L.t[1].blk_begin;
#define FQ_PICK2(f) r.f = s1 ? L.t[1].f : L.t[0].f
        FQ_PICK2(g);
        FQ_PICK2(gx);
        FQ_PICK2(bounds);
        FQ_PICK2(mask);
        FQ_PICK2(rows);
        FQ_PICK2(blk_begin);
        FQ_PICK2(inplace);
        if constexpr (PITCH) {
            FQ_PICK2(gp);
            FQ_PICK2(op);
        } else {
            r.gp = r.op = RowPitch{};
        }
#undef FQ_PICK2
        return r;
    }
    const int64_t b1 = L.t[1].blk_begin, b2 = L.t[2].blk_begin, b3 = L.t[3].blk_begin;
    const int s = (int)(b >= b1) + (int)(b >= b2) + (int)(b >= b3);
#define FQ_PICK(f) r.f = s == 0 ? L.t[0].f : s == 1 ? L.t[1].f : s == 2 ? L.t[2].f : L.t[3].f
    FQ_PICK(g);
    FQ_PICK(gx);
    FQ_PICK(bounds);
    FQ_PICK(mask);
    FQ_PICK(rows);
    FQ_PICK(blk_begin);
    FQ_PICK(inplace);
    if constexpr (PITCH) {
        FQ_PICK(gp);
        FQ_PICK(op);
    } else {
        r.gp = r.op = RowPitch{};
    }
#undef FQ_PICK
    return r;
}

// one chunk (cv vectors from vector vs) of one row.  `bounds` non-null: the row's {upper, lower} bounds are read HERE, after
// the gradient and mask loads have been issued, so that nothing waits on them (null: the caller knows the row can clip).
template <int DT, int VPT, bool NTL, bool NTS, bool PITCH = false>
__device__ __forceinline__ void ste_mask_chunk(const void* g, void* gx, const uint8_t* mrow, int mrow_dwords, int64_t row, int64_t nvec_row,
                                               int64_t vs, int cv, const float* bounds, float lo, float hi, int t, const RowPitch& gp,
                                               const RowPitch& op) {
    using T = Ty<DT>;
    const uint4* gr = (const uint4*)((const char*)g + row_byte_off<PITCH>(row, nvec_row * 16, gp)) + vs;
    uint4* or_ = (uint4*)((char*)gx + row_byte_off<PITCH>(row, nvec_row * 16, op)) + vs;
    const int64_t rem = nvec_row - vs;
    const int nvec = (int)(rem < cv ? rem : cv);
    uint4 rg[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        int v = t + i * STE_THREADS;
        v = v < nvec ? v : nvec - 1;
        rg[i] = ld16<NTL>(&gr[v]);
    }
    SteMaskHeld<16 / T::ESIZE, VPT> mh;
    mh.load(mrow, vs, mrow_dwords, t);  // unconditionally (a safe row's bitmap is unwritten memory of the same buffer: read, ignored)
    bool safe = false;
    if (bounds) {
        const float ub = bounds[2 * row], lb = bounds[2 * row + 1];
        safe = (ub < hi) && (lb > lo);  // false when a bound is NaN
    }
    if (safe) {  // block-uniform
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = t + i * STE_THREADS;
            if (v < nvec) st16<NTS>(&or_[v], rg[i]);
        }
    } else {
        uint32_t mb[VPT];
        mh.bits(t, mb);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = t + i * STE_THREADS;
            const uint4 o = ste_mask_apply<DT>(rg[i], mb[i]);
            if (v < nvec) st16<NTS>(&or_[v], o);
        }
    }
}

// grid: x = the slots' row blocks (a copying slot: one per row; an in-place slot: one per STE_THREADS rows), y = chunk of the row
// ONE copying tensor per launch, nothing else (round 5): the gradient of an activation, of K, of V -- [2048,4096]-sized launches that are
// mostly launch boundary.  Without the slot table (four slots' kernarg fields loaded and selected in every block's prologue), the
// in-place path and its LDS the same chunk body runs 10-11 % faster there (6.97 -> 6.23 us; tools/small_rows_2perwg.hip, which also shows
// that putting two or four rows into one workgroup adds nothing on top: profiles/r05_ab_small_rows_2perwg.txt).
template <int DT, int VPT, bool NTL = true, bool NTS = true>
__global__ __launch_bounds__(STE_THREADS) void ste_mask_one_kernel(const void* g, void* gx, const float* bounds, const uint64_t* mask, int64_t nvec_row, int cv,
                                                                   int64_t mask_row_words, float lo, float hi) {
    const int64_t row = blockIdx.x;
    ste_mask_chunk<DT, VPT, NTL, NTS, false>(g, gx, (const uint8_t*)(mask + row * mask_row_words), (int)mask_row_words * 2, row, nvec_row, (int64_t)blockIdx.y * cv, cv,
                                             bounds, lo, hi, (int)threadIdx.x, RowPitch{}, RowPitch{});
}

template <int DT, int VPT, bool NTL = true, bool NTS = true, bool PITCH = false, int NS = 4>
__global__ __launch_bounds__(STE_THREADS) void ste_mask_kernel(SteLaunch L, int64_t nvec_row, int cv, int64_t mask_row_words, float lo, float hi) {
    __shared__ uint64_t unsafe_rows[STE_THREADS / 64];
    const int t = threadIdx.x;
    const SteSlot sl = ste_pick_slot<PITCH, NS>(L, (int64_t)blockIdx.x);
    const int64_t local = (int64_t)blockIdx.x - sl.blk_begin;
    if (!sl.inplace) {
        ste_mask_chunk<DT, VPT, NTL, NTS, PITCH>(sl.g, sl.gx, (const uint8_t*)(sl.mask + local * mask_row_words), (int)mask_row_words * 2, local, nvec_row,
                                          (int64_t)blockIdx.y * cv, cv, sl.bounds, lo, hi, t, sl.gp, sl.op);
        return;
    }
    if (blockIdx.y) return;  // an in-place slot's blocks walk whole rows
    // in place: which of this block's STE_THREADS rows can clip at all?
    const int64_t r0 = local * STE_THREADS;
    bool unsafe = false;
    if (r0 + t < sl.rows) {
        const float ub = sl.bounds[2 * (r0 + t)], lb = sl.bounds[2 * (r0 + t) + 1];
        unsafe = !((ub < hi) && (lb > lo));
    }
    const uint64_t bal = __ballot(unsafe);
    if ((t & 63) == 0) unsafe_rows[t >> 6] = bal;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < STE_THREADS / 64; ++w) {  // block-uniform walk: every wave reaches the end
        uint64_t bits = unsafe_rows[w];
        while (bits) {
            const int64_t row = r0 + w * 64 + __builtin_ctzll(bits);
            bits &= bits - 1;
            const uint8_t* mrow = (const uint8_t*)(sl.mask + row * mask_row_words);
#pragma unroll 1
            for (int c = 0; c < (int)gridDim.y; ++c)
                ste_mask_chunk<DT, VPT, false, false, PITCH>(sl.g, sl.gx, mrow, (int)mask_row_words * 2, row, nvec_row, (int64_t)c * cv, cv, nullptr, lo, hi, t,
                                                      sl.gp, sl.op);
        }
    }
}

// STE backward of a fp32-result (autocast) forward: the gradient arrives in fp32 (the dtype of the forward's result),
// the input's gradient leaves in the input's 16-bit dtype -- the autograd engine's cast and the masking in one pass
// (read 4 B + write 2 B per element instead of a cast kernel followed by a 16-bit STE kernel).  A lane owns 4 elements
// (one 16-byte fp32 vector in, one 8-byte 16-bit half-vector out) = one nibble of the row bitmap.  ch is a multiple of 64.
// grid: x = rows of all slots, y = chunk of the row.
template <int DT, int HPT, bool NTL = true, bool NTS = true, bool PITCH = false, int NS = 4>
__global__ __launch_bounds__(STE_THREADS) void ste_mask_wide_kernel(SteLaunch L, int64_t nh_row, int ch, int64_t mask_row_words, float lo, float hi) {
    using T = Ty<DT>;
    static_assert(T::ESIZE == 2, "fp32 gradient in, 16-bit gradient out");
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    SteSlot sl;   // NS = 1: K's or V's gradient alone (each has its own autograd node): the first slot as it stands, nothing to pick
    if constexpr (NS == 1) sl = L.t[0];
    else sl = ste_pick_slot<PITCH, NS>(L, (int64_t)blockIdx.x);
    const int64_t row = (int64_t)blockIdx.x - sl.blk_begin;
    const int64_t hs = (int64_t)blockIdx.y * ch;
    const uint4* gr = (const uint4*)((const char*)sl.g + row_byte_off<PITCH>(row, nh_row * 16, sl.gp)) + hs;   // fp32 gradient: 16 bytes per lane
    uint2* or_ = (uint2*)((char*)sl.gx + row_byte_off<PITCH>(row, nh_row * 8, sl.op)) + hs;                    // 16-bit result: 8 bytes per lane
    const int64_t rem = nh_row - hs;
    const int nh = (int)(rem < ch ? rem : ch);
    const int t = threadIdx.x;
    uint4 rg[HPT];
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        int h = t + i * STE_THREADS;
        h = h < nh ? h : nh - 1;
        rg[i] = ld16<NTL>(&gr[h]);
    }
    SteMaskHeld<4, HPT> mh;
    mh.load((const uint8_t*)(sl.mask + row * mask_row_words), hs, (int)mask_row_words * 2, t);
    const float ub = sl.bounds[2 * row], lb = sl.bounds[2 * row + 1];
    const bool safe = (ub < hi) && (lb > lo);
    uint32_t mb[HPT];
    if (!safe) mh.bits(t, mb);  // block-uniform
    else {
#pragma unroll
        for (int i = 0; i < HPT; ++i) mb[i] = 0u;
    }
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        const int h = t + i * STE_THREADS;
        const float f0[2] = {as_f(rg[i].x), as_f(rg[i].y)}, f1[2] = {as_f(rg[i].z), as_f(rg[i].w)};
        uint32_t w0 = T::pack(f0), w1 = T::pack(f1);
        const uint32_t m2 = mb[i] | (mb[i] << 15);
        w0 &= ~__builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2_t, m2 << 15) >> (short)15);
        w1 &= ~__builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2_t, m2 << 13) >> (short)15);
        if (h < nh) st8<NTS>(&or_[h], make_uint2(w0, w1));
    }
}

}  // namespace fq
