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
    printf("c host ok: %lld elements x 2 bit widths x 2 data flows bit-equal to the oracle\n", (long long)n);
    return 0;
}
