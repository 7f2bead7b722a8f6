/* abi_smoke.c -- a plain C host of the C ABI (no Python, no torch): proves that include/llmqat_fakequant.h +
 * libllmqat_fakequant.so are usable from C with nothing but the HIP runtime, and checks the results against the
 * CPU oracle (linked in from oracle/fq_oracle.c -- this is a test).
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c_host/abi_smoke.c oracle/fq_oracle.c \
 *       -L llm-qat_amd -lllmqat_fakequant -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/llm-qat_amd -Wl,-rpath,/opt/rocm/lib -o abi_smoke
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "llmqat_fakequant.h"

int fqo_sym_fwd(const void*, void*, int32_t*, float*, int64_t, int64_t, int, int, int);
int fqo_ste_bwd(const void*, const void*, void*, int64_t, float, float, int);
int fqo_sym_fwd_autocast(const void* x, void* y, int32_t* idx, float* scale, int64_t rows, int64_t cols, int bits, int dt, int wide, int sem);
int fqo_export(const void* x, void* bins, float* scales, int32_t* overflow, int64_t rows, int64_t cols, int bits, int container, int dt, int sem,
               int asym, int autocast);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
#define FQ(x) do { int rc_ = (x); if (rc_ != FQ_OK) { fprintf(stderr, "%s:%d rc=%d %s\n", __FILE__, __LINE__, rc_, fq_last_error()); return 3; } } while (0)

static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }

int main(void) {
    const int64_t rows = 37, cols = 11008, n = rows * cols;
    printf("%s (abi %d)\n", fq_build_info(), fq_version());
    uint16_t *hx = malloc(n * 2), *hg = malloc(n * 2), *hy = malloc(n * 2), *hgx = malloc(n * 2), *oy = malloc(n * 2), *ogx = malloc(n * 2);
    uint32_t s = 1;
    for (int64_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        float v = ((int32_t)s / 2147483648.0f) * ((i / cols) % 3 == 0 ? 0.05f : 3.0f);
        hx[i] = bf16(v);
        s = s * 1664525u + 1013904223u;
        hg[i] = bf16((int32_t)s / 2147483648.0f);
    }
    void *dx, *dy, *dg, *dgx, *dmask;
    float* dbounds;
    const size_t mbytes = fq_ste_mask_bytes(rows, cols, FQ_DTYPE_BF16);
    CK(hipMalloc(&dx, n * 2)); CK(hipMalloc(&dy, n * 2)); CK(hipMalloc(&dg, n * 2)); CK(hipMalloc(&dgx, n * 2));
    CK(hipMalloc((void**)&dbounds, rows * 8)); CK(hipMalloc(&dmask, mbytes));
    CK(hipMemcpy(dx, hx, n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg, n * 2, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int bits = 4; bits <= 8; bits += 4) {
        /* reference data flow: forward, then a backward that re-reads x */
        FQ(fq_sym_fwd(dx, dy, rows, cols, bits, FQ_DTYPE_BF16, FQ_SEM_CPU_EAGER, NULL, NULL, 0, st));
        FQ(fq_ste_bwd(dg, dx, dgx, n, -2.0f, 2.0f, FQ_DTYPE_BF16, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hy, dy, n * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hgx, dgx, n * 2, hipMemcpyDeviceToHost));
        if (fqo_sym_fwd(hx, oy, NULL, NULL, rows, cols, bits, 1, 0) || fqo_ste_bwd(hg, hx, ogx, n, -2.0f, 2.0f, 1)) return 4;
        if (memcmp(hy, oy, n * 2) || memcmp(hgx, ogx, n * 2)) { fprintf(stderr, "bits %d: mismatch vs oracle (plain flow)\n", bits); return 5; }
        /* training data flow: bounds + STE bit mask, backward without x */
        CK(hipMemsetAsync(dy, 0, n * 2, st)); CK(hipMemsetAsync(dgx, 0, n * 2, st));
        FQ(fq_sym_fwd_train(dx, dy, rows, cols, bits, FQ_DTYPE_BF16, FQ_SEM_CPU_EAGER, -2.0f, 2.0f, dbounds, dmask, mbytes, st));
        FQ(fq_ste_bwd_mask(dg, dgx, rows, cols, -2.0f, 2.0f, dbounds, dmask, mbytes, FQ_DTYPE_BF16, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hy, dy, n * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hgx, dgx, n * 2, hipMemcpyDeviceToHost));
        if (memcmp(hy, oy, n * 2) || memcmp(hgx, ogx, n * 2)) { fprintf(stderr, "bits %d: mismatch vs oracle (training flow)\n", bits); return 6; }
    }
    if (fq_sym_fwd(dx, dy, rows, cols, 99, FQ_DTYPE_BF16, 0, NULL, NULL, 0, st) != FQ_ERR_BITS || !strstr(fq_last_error(), "num_bits")) return 7;

    /* ---- ABI 2 / 3 entry points, each against the oracle ---------------------------------------------------------------- */
    /* (a) fq_sym_fwd_multi + fq_ste_bwd_mask_multi: the tensor split in three (W4 rows 0..11, A8 rows 12..29, W4 rows 30..36) in ONE
     *     launch each way; the first slot's gradient in place (gx == g) */
    {
        const int64_t r0 = 12, r1 = 18, r2 = rows - 30;
        const int bitsv[3] = {4, 8, 4};
        const int64_t rv[3] = {r0, r1, r2}, ov[3] = {0, r0, r0 + r1};
        void* dgi;  /* a private copy of the first slot's gradient: masked where it stands */
        CK(hipMalloc(&dgi, r0 * cols * 2));
        CK(hipMemcpy(dgi, dg, r0 * cols * 2, hipMemcpyDeviceToDevice));
        fq_fwd_tensor ft[3];
        fq_bwd_tensor bt[3];
        size_t moff = 0;
        for (int i = 0; i < 3; ++i) {
            const size_t mb = fq_ste_mask_bytes(rv[i], cols, FQ_DTYPE_BF16);
            ft[i].x = (char*)dx + ov[i] * cols * 2; ft[i].y = (char*)dy + ov[i] * cols * 2; ft[i].rows = rv[i]; ft[i].bits = bitsv[i];
            ft[i].row_bounds = dbounds + 2 * ov[i]; ft[i].mask = (char*)dmask + moff; ft[i].mask_bytes = mb;
            bt[i].g = i == 0 ? dgi : (char*)dg + ov[i] * cols * 2; bt[i].gx = i == 0 ? dgi : (char*)dgx + ov[i] * cols * 2; bt[i].rows = rv[i];
            bt[i].row_bounds = ft[i].row_bounds; bt[i].mask = ft[i].mask;
            moff += mb;
        }
        if (moff > mbytes) return 8;
        CK(hipMemsetAsync(dy, 0, n * 2, st)); CK(hipMemsetAsync(dgx, 0, n * 2, st));
        FQ(fq_sym_fwd_multi(3, ft, cols, FQ_DTYPE_BF16, FQ_SEM_CPU_EAGER, 0, -2.0f, 2.0f, st));
        FQ(fq_ste_bwd_mask_multi(3, bt, cols, -2.0f, 2.0f, FQ_DTYPE_BF16, 0, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hy, dy, n * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hgx, dgx, n * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hgx, dgi, r0 * cols * 2, hipMemcpyDeviceToHost));   /* slot 0's result stands in its own gradient buffer */
        for (int i = 0; i < 3; ++i) {
            const int64_t off = ov[i] * cols;
            if (fqo_sym_fwd(hx + off, oy + off, NULL, NULL, rv[i], cols, bitsv[i], 1, 0)) return 4;
        }
        if (fqo_ste_bwd(hg, hx, ogx, n, -2.0f, 2.0f, 1)) return 4;
        if (memcmp(hy, oy, n * 2) || memcmp(hgx, ogx, n * 2)) { fprintf(stderr, "multi-tensor launch: mismatch vs oracle\n"); return 9; }
        CK(hipFree(dgi));
    }
    /* (b) fq_sym_export (int4 / int8 / int16 + scales + overflow counts) and fq_sym_row_scales */
    {
        const int conts[3] = {FQ_BINS_INT4, FQ_BINS_INT8, FQ_BINS_INT16};
        void* dbins;
        float* dsc;
        int32_t* dov;
        CK(hipMalloc(&dbins, n * 2)); CK(hipMalloc((void**)&dsc, rows * 8)); CK(hipMalloc((void**)&dov, rows * 4));
        uint8_t *hb = malloc(n * 2), *ob = malloc(n * 2);
        float *hs = malloc(rows * 8), *os = malloc(rows * 8);
        int32_t *hov = malloc(rows * 4), *oov = malloc(rows * 4);
        for (int bits = 4; bits <= 8; bits += 4)
            for (int c = 0; c < 3; ++c) {
                const size_t bb = fq_export_bins_bytes(rows, cols, conts[c]);
                FQ(fq_sym_export(dx, dbins, dsc, dov, rows, cols, bits, conts[c], FQ_DTYPE_BF16, FQ_SEM_CPU_EAGER, 0, st));
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(hb, dbins, bb, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, dsc, rows * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hov, dov, rows * 4, hipMemcpyDeviceToHost));
                memset(ob, 0, bb);
                if (fqo_export(hx, ob, os, oov, rows, cols, bits, conts[c], 1, 0, 0, 0)) return 4;
                if (memcmp(hb, ob, bb) || memcmp(hs, os, rows * 8) || memcmp(hov, oov, rows * 4)) {
                    fprintf(stderr, "export bits %d container %d: mismatch vs oracle\n", bits, conts[c]);
                    return 10;
                }
            }
        FQ(fq_sym_row_scales(dx, dsc, rows, cols, 8, FQ_DTYPE_BF16, FQ_SEM_CPU_EAGER, 0, -2.0f, 2.0f, NULL, NULL, 0, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hs, dsc, rows * 8, hipMemcpyDeviceToHost));
        if (memcmp(hs, os, rows * 8)) { fprintf(stderr, "fq_sym_row_scales: mismatch vs the oracle's export scales\n"); return 11; }
        /* (c) ABI 4: the autocast arithmetic (fp32 behind the reciprocal; models/utils_quant.py:71-72 under torch.autocast("cuda")) with
         *     both scalar policies, fp32 result and the result rounded once to bf16 */
        {
            void* dy32;
            CK(hipMalloc(&dy32, n * 4));
            float *hy32 = malloc(n * 4), *oy32 = malloc(n * 4);
            for (int sem = 0; sem <= 1; ++sem) {
                FQ(fq_sym_fwd_autocast(dx, dy32, rows, cols, 8, FQ_DTYPE_BF16, sem, 1, -2.0f, 2.0f, NULL, NULL, 0, NULL, 0, st));
                FQ(fq_sym_fwd_autocast(dx, dy, rows, cols, 4, FQ_DTYPE_BF16, sem, 0, -2.0f, 2.0f, NULL, NULL, 0, NULL, 0, st));
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(hy32, dy32, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy, dy, n * 2, hipMemcpyDeviceToHost));
                if (fqo_sym_fwd_autocast(hx, oy32, NULL, NULL, rows, cols, 8, 1, 1, sem) || fqo_sym_fwd_autocast(hx, oy, NULL, NULL, rows, cols, 4, 1, 0, sem)) return 4;
                if (memcmp(hy32, oy32, n * 4) || memcmp(hy, oy, n * 2)) { fprintf(stderr, "autocast arithmetic (sem %d): mismatch vs oracle\n", sem); return 12; }
            }
        }
    }
    printf("c host ok: %lld elements x 2 bit widths x 2 data flows, multi-tensor launches, export x 3 containers, row scales and the autocast arithmetic bit-equal to the oracle\n", (long long)n);
    return 0;
}
