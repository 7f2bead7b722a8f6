/* selftest.c -- drives every oracle entry point over ragged shapes and hostile values under
 * AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle sanitize`).  Test infrastructure.
 * GPU sanitizers are not available on the pool, so memory-safety checking happens on this CPU build. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int fqo_sym_fwd(const void*, void*, int32_t*, float*, int64_t, int64_t, int, int, int);
int fqo_asym_fwd(const void*, void*, int32_t*, float*, float*, int64_t, int64_t, int, int, int);
int fqo_ste_bwd(const void*, const void*, void*, int64_t, float, float, int);
int fqo_w12_fwd(const void*, void*, float*, const float*, int64_t, int64_t, int, int);
int fqo_sym_fwd_autocast(const void*, void*, int32_t*, float*, int64_t, int64_t, int, int, int, int);
int fqo_ste_bwd_wide(const float*, const void*, void*, int64_t, float, float, int);
int fqo_export(const void*, void*, float*, int32_t*, int64_t, int64_t, int, int, int, int, int, int);

static uint32_t rng = 12345u;
static uint32_t next(void) { rng = rng * 1664525u + 1013904223u; return rng; }

int main(void) {
    const int64_t shapes[][2] = {{1, 1}, {3, 7}, {5, 33}, {2, 255}, {4, 1024}, {1, 11008}, {7, 0}, {0, 9}};
    const float specials[] = {0.f, -0.f, INFINITY, -INFINITY, NAN, 1e-40f, 3e38f, -3e38f, 2.f, -2.f, 1e-7f};
    long calls = 0;
    for (unsigned si = 0; si < sizeof shapes / sizeof shapes[0]; ++si) {
        const int64_t rows = shapes[si][0], cols = shapes[si][1], n = rows * cols;
        for (int dt = 0; dt < 3; ++dt) {
            const size_t es = dt == 0 ? 4 : 2;
            /* exact-size heap blocks so any off-by-one trips ASan */
            void* x = malloc(n * es + 1), *y = malloc(n * es + 1), *g = malloc(n * es + 1), *gx = malloc(n * es + 1);
            int32_t* idx = malloc(n * sizeof(int32_t) + 1);
            float* s = malloc(rows * sizeof(float) + 1), *s2 = malloc(rows * sizeof(float) + 1);
            for (int64_t i = 0; i < n; ++i) {
                float v = (next() % 7 == 0) ? specials[next() % (sizeof specials / sizeof specials[0])] : ((int32_t)next() / 1.0e9f);
                if (dt == 0) { ((float*)x)[i] = v; ((float*)g)[i] = v * 0.5f; }
                else { uint32_t u; memcpy(&u, &v, 4); ((uint16_t*)x)[i] = (uint16_t)(dt == 1 ? u >> 16 : next()); ((uint16_t*)g)[i] = (uint16_t)next(); }
            }
            for (int bits = 1; bits <= 31; bits += (bits < 8 ? 1 : 7)) {
                for (int sem = 0; sem < 2; ++sem) {
                    if (bits >= 2) { if (fqo_sym_fwd(x, y, idx, s, rows, cols, bits, dt, sem)) return 2; ++calls; }
                    if (fqo_asym_fwd(x, y, idx, s, s2, rows, cols, bits, dt, sem)) return 3;
                    if (fqo_asym_fwd(x, y, NULL, NULL, NULL, rows, cols, bits, dt, sem)) return 4;
                    calls += 2;
                }
            }
            if (fqo_ste_bwd(g, x, gx, n, -2.f, 2.f, dt) || fqo_ste_bwd(g, x, gx, n, -0.3009f, 0.75f, dt)) return 5;
            if (dt != 0) {   /* the autocast arithmetic (16-bit tensors): fp32 and narrow results, both scalar policies; the fp32-gradient backward; export */
                float* y32 = malloc(n * 4 + 1), *g32 = malloc(n * 4 + 1), *sc2 = malloc(rows * 8 + 1);
                int32_t* ov = malloc(rows * 4 + 1);
                uint8_t* bins = malloc(n * 2 + 1);
                for (int64_t i = 0; i < n; ++i) g32[i] = (next() % 9 == 0) ? specials[next() % (sizeof specials / sizeof specials[0])] : ((int32_t)next() / 1.0e9f);
                for (int bits = 2; bits <= 31; bits += (bits < 8 ? 2 : 7))
                    for (int sem = 0; sem < 2; ++sem) {
                        if (fqo_sym_fwd_autocast(x, y32, idx, s, rows, cols, bits, dt, 1, sem) || fqo_sym_fwd_autocast(x, y, NULL, NULL, rows, cols, bits, dt, 0, sem)) return 9;
                        for (int cont = 1; cont <= 3; ++cont)
                            if (fqo_export(x, bins, sc2, ov, rows, cols, bits, cont, dt, sem, 0, 1)) return 10;
                        calls += 5;
                    }
                if (fqo_ste_bwd_wide(g32, x, gx, n, -2.f, 2.f, dt)) return 11;
                ++calls;
                free(y32); free(g32); free(sc2); free(ov); free(bins);
            }
            if (cols > 0) {
                if (fqo_w12_fwd(x, y, s, NULL, rows, cols, 1, dt) || fqo_w12_fwd(x, y, s, NULL, rows, cols, 2, dt)) return 6;
                if (fqo_w12_fwd(x, y, s2, s, rows, cols, 2, dt)) return 7;
            }
            calls += 5;
            free(x); free(y); free(g); free(gx); free(idx); free(s); free(s2);
        }
    }
    if (fqo_sym_fwd(NULL, NULL, NULL, NULL, 0, 0, 99, 0, 0) == 0 || fqo_ste_bwd(NULL, NULL, NULL, 0, 0, 0, 7) == 0) return 8; /* bad args rejected */
    printf("oracle selftest ok: %ld calls under ASan+UBSan\n", calls);
    return 0;
}
