// fq_dtype_impl.h -- kernel selection + launches for ONE element type (included by fq_<dtype>.hip).
#pragma once
#include "../../include/llmqat_fakequant.h"
#include "fq_launch.h"


namespace fq {

#define FQ_LAUNCH(kern, grid, block, st, ...) FQ_LAUNCHK(kern, dim3((unsigned)(grid)), dim3(block), 0, st, __VA_ARGS__)

// Which (threads per row, 16-byte vectors per thread) the model widths select (bf16 / fp16: 8 elements per vector):
//   4096 -> 512 vectors -> 256 x 2     5120 -> 640 -> 256 x 3     11008 -> 1376 -> 512 x 3     13824 -> 1728 -> 512 x 4
//   tiny-LLaMA (fp32, 4 per vector): 256 -> 64 x 1, 688 -> 64 x 3.      Every other width is served by the nearest shape that holds it.
// Round 4 pruned the instantiations nothing selects often enough to deserve its own code (27 MB / 100 s of build in round 3): stores
// are always non-temporal (two cache-policy flavours instead of three), 5 / 7 vectors per thread share the 6 / 8 kernels.
// Launch shape of the register-resident kernel, from tools/kbench on MI355X: 2-3 vectors per
// thread is the sweet spot (round 3 re-checked with block sizes that are not powers of two -- 704 x 2 for 11008 cols wastes 2 % of
// the vector slots instead of 10 %, 576 x 3 for 13824, 320 x 2 for 5120 -- and with 256 x 6 for mid-sized tensors: within +-2 % of
// the shapes below everywhere, profiles/r03_kbench_block_shapes.txt, r03_ab_mid_rows_256.txt) (11008 bf16 cols: 512 thr x 3 = 30.6 us, 256 x 6 = 31.1, 1024 x 2 = 32.2;
// 4096 cols: 256 x 2 = 6.1 us, 128 x 4 = 6.3, 64 x 8 = 6.9, 512 x 1 = 7.7).
template <int DT, bool ASYM, bool FAST, bool NTL, bool NTS, bool DBG, bool PITCH = false>
static void launch_reg(const RowArgs& a, int64_t nvec, hipStream_t st) {
#define R(TPR, V)                                                                                                   \
    case V:                                                                                                         \
        FQ_LAUNCH((row_reg_kernel<DT, TPR, V, ASYM, FAST, NTL, NTS, DBG, 0, PITCH>), (TPR == 64 ? (a.rows + 3) / 4 : a.rows), \
                  (TPR == 64 ? 256 : TPR), st, a);                                                                  \
        break;
    if (nvec <= 192) {
        switch ((int)((nvec + 63) / 64)) { R(64, 1) R(64, 2) R(64, 3) }
    } else if (nvec <= 384) {
        switch ((int)((nvec + 127) / 128)) { R(128, 2) R(128, 3) }
    } else if (nvec <= 768) {
        switch ((int)((nvec + 255) / 256)) { R(256, 2) R(256, 3) }
    } else if (nvec <= 4096) {   // (5 and 7 vectors per thread run as 6 and 8: no model width lands there, see the table above launch_reg)
        switch ((int)((nvec + 511) / 512)) { R(512, 2) R(512, 3) R(512, 4) case 5: R(512, 6) case 7: R(512, 8) }
    } else {
        switch ((int)((nvec + 1023) / 1024)) { case 5: R(1024, 6) case 7: R(1024, 8) }
    }
#undef R
}

template <int DT, int AC, bool NTL, bool NTS, bool PITCH = false>
static void launch_reg_ac(const RowArgs& a, int64_t nvec, hipStream_t st) {
#define R(TPR, V)                                                                                                            \
    case V:                                                                                                                  \
        FQ_LAUNCH((row_reg_kernel<DT, TPR, V, false, false, NTL, NTS, false, AC, PITCH>), (TPR == 64 ? (a.rows + 3) / 4 : a.rows),  \
                  (TPR == 64 ? 256 : TPR), st, a);                                                                           \
        break;
    if (nvec <= 192) {
        switch ((int)((nvec + 63) / 64)) { R(64, 1) R(64, 2) R(64, 3) }
    } else if (nvec <= 384) {
        switch ((int)((nvec + 127) / 128)) { R(128, 2) R(128, 3) }
    } else if (nvec <= 768) {
        switch ((int)((nvec + 255) / 256)) { R(256, 2) R(256, 3) }
    } else if (nvec <= 4096) {   // (5 and 7 vectors per thread run as 6 and 8: no model width lands there, see the table above launch_reg)
        switch ((int)((nvec + 511) / 512)) { R(512, 2) R(512, 3) R(512, 4) case 5: R(512, 6) case 7: R(512, 8) }
    } else {
        switch ((int)((nvec + 1023) / 1024)) { case 5: R(1024, 6) case 7: R(1024, 8) }
    }
#undef R
}

template <int DT, int TPR, bool NTL, bool NTS, bool PITCH = false>
static void launch_wide(const RowArgs& a, int hpt, hipStream_t st) {
    const int64_t grid = TPR == 64 ? (a.rows + 3) / 4 : a.rows;
    constexpr int BLOCK = TPR == 64 ? 256 : TPR;
    const bool mask = any_mask(a) || PITCH;  // the mask-recording code lives in its own instantiation (it costs the plain one 7 %); pitched rows: one flavour
    switch (hpt) {
#define H(N)                                                                                            \
    case N:                                                                                             \
        if (mask) FQ_LAUNCH((row_reg_wide_kernel<DT, TPR, N, NTL, NTS, true, PITCH>), grid, BLOCK, st, a);     \
        else if constexpr (!PITCH) FQ_LAUNCH((row_reg_wide_kernel<DT, TPR, N, NTL, NTS, false>), grid, BLOCK, st, a);         \
        break;
        H(1) H(2) H(3) H(4) H(5) H(6) case 7: H(8)
#undef H
        default: break;
    }
}

template <int DT, int AC>
static int sym_autocast_t(RowArgs a, void* ws, size_t wsb, hipStream_t st) {
    using T = Ty<DT>;
    begin_launches();
    seal_slots(a);
    if constexpr (T::ESIZE != 2) {
        return fail(FQ_ERR_DTYPE, "autocast arithmetic applies to bf16 / fp16 tensors only");
    } else {
        constexpr int EPV = 8;
        const bool pair = a.n_more > 0;
        const bool vec_ok = aligned16(a.x) && aligned16(a.y) && (a.cols % EPV == 0) && more_aligned(a, 15u, 15u) && pitches_aligned(a, 15, 15);
        const int64_t nvec = a.cols / EPV;
        const int64_t big_rows = largest_rows(a);
        const int64_t bytes = big_rows * a.cols * T::ESIZE;
        if (a.rows > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows=%lld exceeds the grid limit", (long long)a.rows);
        const bool pitched = any_pitch(a);
        if constexpr (AC == 2) {
            // fp32 result: 8-byte loads / 16-byte stores keep both streams fully coalesced
            const int64_t nh = a.cols / 4;
            const bool wide_ok = aligned16(a.y) && (reinterpret_cast<uintptr_t>(a.x) & 7u) == 0 && a.cols % 4 == 0 && nh <= 1024 * 8 &&
                                 more_aligned(a, 7u, 15u) && pitches_aligned(a, 7, 15);
            if (wide_ok) {
                const bool ntl = bytes >= NT_LOAD_MIN_BYTES;
#define W(TPR)                                                                                          \
    {                                                                                                   \
        const int hpt = (int)((nh + TPR - 1) / TPR);                                                    \
        if (pitched) launch_wide<DT, TPR, false, true, true>(a, hpt, st);                               \
        else if (ntl) launch_wide<DT, TPR, true, true>(a, hpt, st);                                     \
        else launch_wide<DT, TPR, false, true>(a, hpt, st);                                             \
    }
                if (nh <= 512) W(64) else if (nh <= 2048) W(256) else W(1024)
#undef W
                return launch_result();
            }
            if (pair || a.mask)
                return fail(FQ_ERR_UNSUPPORTED, "fp32-result forward with STE mask / second tensor: rows must be 8-byte aligned, cols %% 4 == 0, cols <= 32768");
            // other shapes: scalar-load kernel or two passes (bounds only)
            if (a.cols <= GENERIC_MAX_COLS) {
                if (a.cols <= 1024) FQ_LAUNCH((row_generic_kernel<DT, 64, false, AC>), (a.rows + 3) / 4, 256, st, a);
                else FQ_LAUNCH((row_generic_kernel<DT, 256, false, AC>), a.rows, 256, st, a);
                return launch_result();
            }
        } else {
            if (pair && !(vec_ok && nvec <= REG_MAX_VEC))
                return fail(FQ_ERR_UNSUPPORTED, "pair launch: rows must be 16-byte aligned and fit the register kernels");
        }
        if (AC == 1 && vec_ok && nvec <= REG_MAX_VEC) {
            if constexpr (AC == 1) {
                if (pitched) launch_reg_ac<DT, AC, false, true, true>(a, nvec, st);   // rows that do not follow one another: their own instantiations
                else if (bytes >= NT_LOAD_MIN_BYTES) launch_reg_ac<DT, AC, true, true>(a, nvec, st);
                else launch_reg_ac<DT, AC, false, true>(a, nvec, st);
            }
        } else if (a.mask) {
            return fail(FQ_ERR_UNSUPPORTED, "STE-mask forward needs 16-byte aligned rows that fit the register kernels");
        } else if (a.cols <= GENERIC_MAX_COLS) {
            if (a.cols <= 1024) FQ_LAUNCH((row_generic_kernel<DT, 64, false, AC>), (a.rows + 3) / 4, 256, st, a);
            else FQ_LAUNCH((row_generic_kernel<DT, 256, false, AC>), a.rows, 256, st, a);
        } else {  // very long rows (layerwise): two passes -- |x| max per row through atomics, then apply
            if (pitched) return fail(FQ_ERR_UNSUPPORTED, "rows that do not follow one another: not served by the two-pass path");
            if (!ws || wsb < (size_t)a.rows * 8)
                return fail(FQ_ERR_WORKSPACE, "two-pass path needs %zu workspace bytes, got %zu", (size_t)a.rows * 8, wsb);
            if (hipMemsetAsync(ws, 0, (size_t)a.rows * 8, st) != hipSuccess) return fail(FQ_ERR_LAUNCH, "hipMemsetAsync failed");
            uint32_t* w = (uint32_t*)ws;
            const bool svec = aligned16(a.x) && a.cols % EPV == 0;
            const int64_t sch = svec ? tp_chunk_elems<DT, true>() : tp_chunk_elems<DT, false>();
            const int64_t schunks = (a.cols + sch - 1) / sch, ach = (int64_t)TP_THREADS * TP_EPT, achunks = (a.cols + ach - 1) / ach;
            if (a.rows * schunks > 0x7FFFFFFF || a.rows * achunks > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows*chunks exceeds the grid limit");
            if (svec) FQ_LAUNCH((stats_kernel<DT, false, true>), a.rows * schunks, TP_THREADS, st, a, w, schunks);
            else FQ_LAUNCH((stats_kernel<DT, false, false>), a.rows * schunks, TP_THREADS, st, a, w, schunks);
            FQ_LAUNCH((apply_autocast_kernel<DT, AC == 2>), a.rows * achunks, TP_THREADS, st, a, (const uint32_t*)w, achunks);
        }
        return launch_result();
    }
}

template <int DT> int launch_sym_autocast(bool wide, RowArgs a, void* ws, size_t wsb, hipStream_t st) {
    return wide ? sym_autocast_t<DT, 2>(a, ws, wsb, st) : sym_autocast_t<DT, 1>(a, ws, wsb, st);
}

template <int DT, bool ASYM, bool FAST>
static int rowwise_t(RowArgs a, void* ws, size_t wsb, hipStream_t st) {
    using T = Ty<DT>;
    begin_launches();
    seal_slots(a);
    constexpr int EPV = 16 / T::ESIZE;
    const bool pair = a.n_more > 0;  // several tensors in one launch: register kernels only
    const bool vec_ok = aligned16(a.x) && aligned16(a.y) && (a.cols % EPV == 0) && more_aligned(a, 15u, 15u) && pitches_aligned(a, 15, 15);
    const int64_t nvec = a.cols / EPV;
    const int64_t big_rows = largest_rows(a);
    const int64_t bytes = big_rows * a.cols * T::ESIZE;  // cache policy follows the larger tensor
    const bool pitched = any_pitch(a);
    const bool ntl = bytes >= NT_LOAD_MIN_BYTES;
    bool two_pass = false, two_pass_vec = false;
    if (pair && !(vec_ok && nvec <= REG_MAX_VEC)) return fail(FQ_ERR_UNSUPPORTED, "pair launch: rows must be 16-byte aligned and fit the register kernels");
    if (vec_ok && nvec <= REG_MAX_VEC) {
        if (a.rows > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows=%lld exceeds the grid limit", (long long)a.rows);
        // the diagnostic outputs (bin indices, scales) live in their own instantiation so the product
        // kernels carry none of that code; it runs the same arithmetic with the default cache policy
        if (a.idx || a.scale) launch_reg<DT, ASYM, FAST, false, false, true>(a, nvec, st);
        else if (pitched) launch_reg<DT, ASYM, FAST, false, true, false, true>(a, nvec, st);   // rows that do not follow one another: their own instantiations
        else if (ntl) launch_reg<DT, ASYM, FAST, true, true, false>(a, nvec, st);
        else launch_reg<DT, ASYM, FAST, false, true, false>(a, nvec, st);
    } else if (a.mask) {
        return fail(FQ_ERR_UNSUPPORTED, "STE-mask forward needs 16-byte aligned rows that fit the register kernels");
    } else if (vec_ok && !pitched) {
        two_pass = two_pass_vec = true;
    } else if (a.cols <= GENERIC_MAX_COLS) {
        if (a.rows > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows=%lld exceeds the grid limit", (long long)a.rows);
        if (a.cols <= 1024) FQ_LAUNCH((row_generic_kernel<DT, 64, ASYM>), (a.rows + 3) / 4, 256, st, a);
        else FQ_LAUNCH((row_generic_kernel<DT, 256, ASYM>), a.rows, 256, st, a);
    } else {
        two_pass = true;
    }
    if (two_pass) {
        if (pitched) return fail(FQ_ERR_UNSUPPORTED, "rows that do not follow one another: not served by the two-pass path");
        if (!ws || wsb < (size_t)a.rows * 8)
            return fail(FQ_ERR_WORKSPACE, "two-pass path needs %zu workspace bytes, got %zu", (size_t)a.rows * 8, wsb);
        const int64_t ch = two_pass_vec ? tp_chunk_elems<DT, true>() : tp_chunk_elems<DT, false>();
        const int64_t chunks = (a.cols + ch - 1) / ch;
        if (a.rows * chunks > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows*chunks exceeds the grid limit");
        if (hipMemsetAsync(ws, 0, (size_t)a.rows * 8, st) != hipSuccess) return fail(FQ_ERR_LAUNCH, "hipMemsetAsync failed");
        uint32_t* w = (uint32_t*)ws;
        if (two_pass_vec) {
            FQ_LAUNCH((stats_kernel<DT, ASYM, true>), a.rows * chunks, TP_THREADS, st, a, w, chunks);
            FQ_LAUNCH((apply_kernel<DT, ASYM, FAST, true>), a.rows * chunks, TP_THREADS, st, a, (const uint32_t*)w, chunks);
        } else {
            FQ_LAUNCH((stats_kernel<DT, ASYM, false>), a.rows * chunks, TP_THREADS, st, a, w, chunks);
            FQ_LAUNCH((apply_kernel<DT, ASYM, FAST, false>), a.rows * chunks, TP_THREADS, st, a, (const uint32_t*)w, chunks);
        }
    }
    return launch_result();
}

template <int DT> int launch_rowwise(bool asym, bool fast, RowArgs a, void* ws, size_t wsb, hipStream_t st) {
    if constexpr (DT == BF16) {
        // bf16, <= 8 bits: the IEEE divides become multiplies with a reciprocal -- provably
        // bit-identical after the bf16 rounding (DESIGN.md "Numerics").
        if (fast) return asym ? rowwise_t<DT, true, true>(a, ws, wsb, st) : rowwise_t<DT, false, true>(a, ws, wsb, st);
    }
    if constexpr (DT == F16) {
        // fp16 AsymQuantizer at <= 8 bits: FAST selects the lookup-table form of the register kernel (exact arithmetic kept)
        if (fast && asym) return rowwise_t<DT, true, true>(a, ws, wsb, st);
    }
    return asym ? rowwise_t<DT, true, false>(a, ws, wsb, st) : rowwise_t<DT, false, false>(a, ws, wsb, st);
}

template <int DT> int launch_ste(const void* g, const void* x, void* gx, int64_t n, float lo, float hi, hipStream_t st) {
    using T = Ty<DT>;
    begin_launches();
    constexpr int EPV = 16 / T::ESIZE;
    if (aligned16(g) && aligned16(x) && aligned16(gx) && n % EPV == 0) {
        const int64_t nvec = n / EPV;
        const int64_t grid = (nvec + STE_THREADS - 1) / STE_THREADS;  // one vector per thread measured best (kbench)
        if (grid > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "n too large");
        const int64_t bytes = n * T::ESIZE;
        if (bytes >= NT_LOAD_MIN_BYTES)
            FQ_LAUNCH((ste_vec_kernel<DT, 1, true, true>), grid, STE_THREADS, st, (const uint4*)g, (const uint4*)x, (uint4*)gx, nvec, lo, hi);
        else
            FQ_LAUNCH((ste_vec_kernel<DT, 1, false, true>), grid, STE_THREADS, st, (const uint4*)g, (const uint4*)x, (uint4*)gx, nvec, lo, hi);
    } else {
        int64_t grid = (n + STE_THREADS - 1) / STE_THREADS;
        if (grid > 8192) grid = 8192;
        FQ_LAUNCH((ste_scalar_kernel<DT>), grid, STE_THREADS, st, g, x, gx, n, lo, hi);
    }
    return launch_result();
}

template <int DT>
int launch_ste_rows(const void* g, const void* x, void* gx, int64_t rows, int64_t cols, float lo, float hi, const float* bounds,
                    hipStream_t st, const StePitch3& pitch) {
    using T = Ty<DT>;
    begin_launches();
    constexpr int EPV = 16 / T::ESIZE;
    const bool pitched = pitch.g.on || pitch.x.on || pitch.o.on;
    if (!(aligned16(g) && aligned16(x) && aligned16(gx) && cols % EPV == 0 && pitch_aligned(pitch.g, 15) && pitch_aligned(pitch.x, 15) && pitch_aligned(pitch.o, 15))) {
        if (pitched) return fail(FQ_ERR_UNSUPPORTED, "rows that do not follow one another: the STE backward needs 16-byte aligned rows of whole vectors");
        return launch_ste<DT>(g, x, gx, rows * cols, lo, hi, st);  // odd layout: plain path, same result
    }
    if (pitched && rows > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows exceed the grid limit");
    const int64_t nvec_row = cols / EPV;
    // balanced chunks: every block of a row gets the same number of vectors (<= 256 x 8)
    const int64_t chunks = (nvec_row + STE_THREADS * 8 - 1) / (STE_THREADS * 8);
    const int cv = (int)((nvec_row + chunks - 1) / chunks);
    const int vpt = (cv + STE_THREADS - 1) / STE_THREADS;
    if (rows * chunks > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows*chunks exceeds the grid limit");
    const int64_t bytes = rows * cols * T::ESIZE;
    const bool ntl = bytes >= NT_LOAD_MIN_BYTES;
#define S(V)                                                                                                                              \
    case V:                                                                                                                               \
        if (pitched) FQ_LAUNCH((ste_rows_kernel<DT, V, false, true, true>), rows * chunks, STE_THREADS, st, g, x, gx, nvec_row, chunks, cv, bounds, lo, hi, pitch);  \
        else if (ntl) FQ_LAUNCH((ste_rows_kernel<DT, V, true, true>), rows * chunks, STE_THREADS, st, g, x, gx, nvec_row, chunks, cv, bounds, lo, hi, pitch);        \
        else FQ_LAUNCH((ste_rows_kernel<DT, V, false, true>), rows * chunks, STE_THREADS, st, g, x, gx, nvec_row, chunks, cv, bounds, lo, hi, pitch);           \
        break;
    switch (vpt) { S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) }
#undef S
    return launch_result();
}

// block layout of a mask-backward launch (grid x): copying slots get one block per row, in-place slots one per STE_THREADS rows;
// grid y = chunks of a row.  Unused slots: blk_begin = INT64_MAX (ste_pick_slot compares against all four).
inline int64_t ste_layout(SteLaunch& L, bool allow_inplace) {
    int64_t blk = 0;
    for (int i = 0; i < 1 + MAX_MORE; ++i) {
        if (i >= L.n) {
            L.t[i] = SteSlot{};
            L.t[i].blk_begin = INT64_MAX;
            continue;
        }
        L.t[i].blk_begin = blk;
        L.t[i].inplace = allow_inplace && L.t[i].gx == (const void*)L.t[i].g;
        blk += L.t[i].inplace ? (L.t[i].rows + STE_THREADS - 1) / STE_THREADS : L.t[i].rows;
    }
    return blk;
}
#define FQ_LAUNCH2(kern, gx_, gy_, block, st, ...) FQ_LAUNCHK(kern, dim3((unsigned)(gx_), (unsigned)(gy_)), dim3(block), 0, st, __VA_ARGS__)

template <int DT> int launch_ste_mask(SteLaunch L, int64_t cols, float lo, float hi, hipStream_t st) {
    using T = Ty<DT>;
    begin_launches();
    constexpr int EPV = 16 / T::ESIZE;
    const int64_t mrw = mask_row_words(cols, T::ESIZE);
    if (!mrw) return fail(FQ_ERR_UNSUPPORTED, "STE-mask backward: shape not served");
    for (int i = 0; i < L.n; ++i)
        if (!(aligned16(L.t[i].g) && aligned16(L.t[i].gx) && (reinterpret_cast<uintptr_t>(L.t[i].mask) & 7u) == 0 && pitch_aligned(L.t[i].gp, 15) &&
              pitch_aligned(L.t[i].op, 15)))
            return fail(FQ_ERR_UNSUPPORTED, "STE-mask backward: alignment not served (g / gx rows 16 bytes, mask 8 bytes)");
    const int64_t nvec_row = cols / EPV;
    const int64_t chunks = (nvec_row + STE_THREADS * 8 - 1) / (STE_THREADS * 8);
    int cv = (int)((nvec_row + chunks - 1) / chunks);
    cv = (cv + 63) / 64 * 64;  // whole waves per slot
    const int vpt = (cv + STE_THREADS - 1) / STE_THREADS;
    const int64_t grid = ste_layout(L, true);
    if (grid > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows exceed the grid limit");
    int64_t big_rows = 0;  // cache policy follows the largest tensor that is actually streamed
    for (int i = 0; i < L.n; ++i)
        if (!L.t[i].inplace && L.t[i].rows > big_rows) big_rows = L.t[i].rows;
    const int64_t bytes = big_rows * cols * T::ESIZE;
    const bool ntl = bytes >= NT_LOAD_MIN_BYTES;
    bool pitched = false;
    for (int i = 0; i < L.n; ++i) pitched = pitched || L.t[i].gp.on || L.t[i].op.on;
    if (L.n == 1 && !L.t[0].inplace && !pitched) {   // one copying tensor: its own lean kernel (fq_kernels.h, ste_mask_one_kernel)
        const SteSlot& t0 = L.t[0];
#define S1(V)                                                                                                                                            \
    case V:                                                                                                                                              \
        if (ntl) FQ_LAUNCH2((ste_mask_one_kernel<DT, V, true, true>), grid, chunks, STE_THREADS, st, t0.g, t0.gx, t0.bounds, t0.mask, nvec_row, cv, mrw, lo, hi);   \
        else FQ_LAUNCH2((ste_mask_one_kernel<DT, V, false, true>), grid, chunks, STE_THREADS, st, t0.g, t0.gx, t0.bounds, t0.mask, nvec_row, cv, mrw, lo, hi);      \
        break;
        switch (vpt) { S1(1) S1(2) S1(3) S1(4) S1(5) S1(6) S1(7) S1(8) }
#undef S1
        return launch_result();
    }
    if (L.n <= 2 && !pitched) {   // two tensors (a QuantizeLinear's weight + input): the slot pick looks at two slots instead of four
#define S2(V)                                                                                                                  \
    case V:                                                                                                                    \
        if (ntl) FQ_LAUNCH2((ste_mask_kernel<DT, V, true, true, false, 2>), grid, chunks, STE_THREADS, st, L, nvec_row, cv, mrw, lo, hi);   \
        else FQ_LAUNCH2((ste_mask_kernel<DT, V, false, true, false, 2>), grid, chunks, STE_THREADS, st, L, nvec_row, cv, mrw, lo, hi);      \
        break;
        switch (vpt) { S2(1) S2(2) S2(3) S2(4) S2(5) S2(6) S2(7) S2(8) }
#undef S2
        return launch_result();
    }
#define S(V)                                                                                                               \
    case V:                                                                                                                \
        if (pitched) FQ_LAUNCH2((ste_mask_kernel<DT, V, false, true, true>), grid, chunks, STE_THREADS, st, L, nvec_row, cv, mrw, lo, hi);   \
        else if (ntl) FQ_LAUNCH2((ste_mask_kernel<DT, V, true, true>), grid, chunks, STE_THREADS, st, L, nvec_row, cv, mrw, lo, hi);        \
        else FQ_LAUNCH2((ste_mask_kernel<DT, V, false, true>), grid, chunks, STE_THREADS, st, L, nvec_row, cv, mrw, lo, hi);           \
        break;
    switch (vpt) { S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) }
#undef S
    return launch_result();
}

template <int DT> int launch_ste_mask_wide(SteLaunch L, int64_t cols, float lo, float hi, hipStream_t st) {
    using T = Ty<DT>;
    begin_launches();
    if constexpr (T::ESIZE != 2) {
        return fail(FQ_ERR_DTYPE, "fp32-gradient STE backward applies to bf16 / fp16 inputs only");
    } else {
        const int64_t mrw = mask_row_words(cols, T::ESIZE);
        auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
        if (!mrw || cols > 32768) return fail(FQ_ERR_UNSUPPORTED, "fp32-gradient STE backward: shape not served");
        for (int i = 0; i < L.n; ++i)
            if (!(aligned16(L.t[i].g) && al8(L.t[i].gx) && al8(L.t[i].mask) && pitch_aligned(L.t[i].gp, 15) && pitch_aligned(L.t[i].op, 7)))
                return fail(FQ_ERR_UNSUPPORTED, "fp32-gradient STE backward: alignment not served");
        const int64_t nh_row = cols / 4;
        const int64_t chunks = (nh_row + STE_THREADS * 8 - 1) / (STE_THREADS * 8);
        int ch = (int)((nh_row + chunks - 1) / chunks);
        ch = (ch + 63) / 64 * 64;
        const int hpt = (ch + STE_THREADS - 1) / STE_THREADS;
        const int64_t grid = ste_layout(L, false);
        if (grid > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows exceed the grid limit");
        const int64_t bytes = largest_rows(L) * cols * 4;  // the fp32 gradient is the larger stream
        const bool ntl = bytes >= NT_LOAD_MIN_BYTES;
        bool pitched = false;
        for (int i = 0; i < L.n; ++i) pitched = pitched || L.t[i].gp.on || L.t[i].op.on;
        if (!ntl && !pitched && L.n <= 2) {   // the K / V-sized launches: one or two slots instead of a four-slot pick in every block's prologue
#define SN(V)                                                                                                                                  \
    case V:                                                                                                                                    \
        if (L.n == 1) FQ_LAUNCH2((ste_mask_wide_kernel<DT, V, false, true, false, 1>), grid, chunks, STE_THREADS, st, L, nh_row, ch, mrw, lo, hi);        \
        else FQ_LAUNCH2((ste_mask_wide_kernel<DT, V, false, true, false, 2>), grid, chunks, STE_THREADS, st, L, nh_row, ch, mrw, lo, hi);                 \
        break;
            switch (hpt) { SN(1) SN(2) SN(3) SN(4) SN(5) SN(6) SN(7) SN(8) }
#undef SN
            return launch_result();
        }
#define S(V)                                                                                                                  \
    case V:                                                                                                                   \
        if (pitched) FQ_LAUNCH2((ste_mask_wide_kernel<DT, V, false, true, true>), grid, chunks, STE_THREADS, st, L, nh_row, ch, mrw, lo, hi);   \
        else if (ntl) FQ_LAUNCH2((ste_mask_wide_kernel<DT, V, true, true>), grid, chunks, STE_THREADS, st, L, nh_row, ch, mrw, lo, hi);        \
        else FQ_LAUNCH2((ste_mask_wide_kernel<DT, V, false, true>), grid, chunks, STE_THREADS, st, L, nh_row, ch, mrw, lo, hi);           \
        break;
        switch (hpt) { S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) }
#undef S
        return launch_result();
    }
}

template <int DT>
int launch_w12(const void* w, const void* scale, void* out, int64_t rows, int64_t cols, int w_bits, int scale_per_row, float cv,
               hipStream_t st) {
    using T = Ty<DT>;
    begin_launches();
    constexpr int EPV = 16 / T::ESIZE;
    const bool vec = aligned16(w) && aligned16(out) && cols % EPV == 0;
    int64_t grid = vec ? (rows * (cols / EPV) + 255) / 256 : (rows * cols + 255) / 256;
    if (!vec && grid > 16384) grid = 16384;
    if (grid > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "tensor too large");
    if (w_bits == 1) {
        if (vec) FQ_LAUNCH((w12_kernel<DT, 1, true>), grid, 256, st, w, scale, out, rows, cols, scale_per_row, cv);
        else FQ_LAUNCH((w12_kernel<DT, 1, false>), grid, 256, st, w, scale, out, rows, cols, scale_per_row, cv);
    } else {
        if (vec) FQ_LAUNCH((w12_kernel<DT, 2, true>), grid, 256, st, w, scale, out, rows, cols, scale_per_row, cv);
        else FQ_LAUNCH((w12_kernel<DT, 2, false>), grid, 256, st, w, scale, out, rows, cols, scale_per_row, cv);
    }
    return launch_result();
}

// The one-launch 1-/2-bit branch with ATen's own summation order (w12_row_aten_kernel): served where that order is the restated one --
// contiguous rows >= 8, cols >= 256, cols % 4 == 0, 16-byte aligned tensors, rows that fit the kernel's registers.
template <int DT>
int launch_w12_rows(const void* w, void* out, void* scale_out, int64_t rows, int64_t cols, int w_bits, float cv, hipStream_t st) {
    using T = Ty<DT>;
    if (!(aligned16(w) && aligned16(out) && cols % 4 == 0 && cols >= 256 && rows >= 8))
        return fail(FQ_ERR_UNSUPPORTED, "one-launch 1-/2-bit branch: needs rows >= 8, cols >= 256, cols %% 4 == 0 and 16-byte aligned tensors");
    if (rows > 0x7FFFFFFF) return fail(FQ_ERR_SHAPE, "rows=%lld exceeds the grid limit", (long long)rows);
    const int64_t ngroups = cols / 4;
    const bool shared_row = cols >= 8129;                      // ATen: values_per_thread = ceil(cols / 64) >= 128 splits a row across the 8 waves
    const int64_t gpt = shared_row ? (ngroups + 511) / 512 : (ngroups + 63) / 64;   // groups per ATen thread
    constexpr int MAXG_OWN = DT == F32 ? 16 : 32;              // fp32 groups are 4 VGPRs each
    if (gpt > (shared_row ? 16 : MAXG_OWN) || (shared_row && (cols + 511) / 512 >= 256))
        return fail(FQ_ERR_UNSUPPORTED, "one-launch 1-/2-bit branch: cols=%lld outside the row lengths this kernel holds in registers", (long long)cols);
    begin_launches();
    const float factor = (float)rows / (float)(rows * cols);   // ATen: static_cast<float>(num_outputs) / numel
    const bool ntl = rows * cols * T::ESIZE >= NT_LOAD_MIN_BYTES;
    const int64_t grid = shared_row ? rows : (rows + 7) / 8;
#define K(SH, G, WB)                                                                                                              \
    {                                                                                                                             \
        if (ntl) FQ_LAUNCH((w12_row_aten_kernel<DT, WB, SH, G, true, true>), grid, 512, st, w, out, scale_out, rows, cols, cv, factor);   \
        else FQ_LAUNCH((w12_row_aten_kernel<DT, WB, SH, G, false, true>), grid, 512, st, w, out, scale_out, rows, cols, cv, factor);      \
    }
#define KW(SH, G)                    \
    {                                \
        if (w_bits == 1) K(SH, G, 1) \
        else K(SH, G, 2)             \
    }
    if (shared_row) {
        if (gpt <= 6) KW(true, 6) else if (gpt <= 8) KW(true, 8) else KW(true, 16)
    } else {
        if (gpt <= 16) KW(false, 16)
        else if constexpr (DT != F32) KW(false, 32)
    }
#undef KW
#undef K
    return launch_result();
}

#define FQ_INSTANTIATE(DT)                                                                                      \
    template int launch_rowwise<DT>(bool, bool, RowArgs, void*, size_t, hipStream_t);                           \
    template int launch_sym_autocast<DT>(bool, RowArgs, void*, size_t, hipStream_t);                                           \
    template int launch_ste<DT>(const void*, const void*, void*, int64_t, float, float, hipStream_t);           \
    template int launch_ste_rows<DT>(const void*, const void*, void*, int64_t, int64_t, float, float, const float*, hipStream_t, const StePitch3&); \
    template int launch_ste_mask<DT>(SteLaunch, int64_t, float, float, hipStream_t);                             \
    template int launch_ste_mask_wide<DT>(SteLaunch, int64_t, float, float, hipStream_t);                        \
    template int launch_w12<DT>(const void*, const void*, void*, int64_t, int64_t, int, int, float, hipStream_t);     \
    template int launch_w12_rows<DT>(const void*, void*, void*, int64_t, int64_t, int, float, hipStream_t);

}  // namespace fq
