/* A C (not C++, not Python) host of libfourq_amd.so: what a maintainer binding the library from another language does.
 * Reads a little test vector file written by tests/test_gpu_cabi.py (scalars, R1 points, expected R1 outputs computed by the
 * oracle), runs fourq_mul_endo_batch and fourq_dh_endo_batch through include/fourq_amd.h, compares bit for bit.
 *   cc -std=c99 -I include -o cabi_check tests/c/cabi_check.c -L fourq_amd -lfourq_amd            exit status 0 = all equal */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fourq_amd.h"

static int fail(const char *what, int rc, fourq_ctx *ctx) {
    fprintf(stderr, "%s: %s (%d) %s\n", what, fourq_strerror(rc), rc, ctx ? fourq_last_error(ctx) : "");
    return 2;
}

int main(int argc, char **argv) {
    if (argc != 2) { fprintf(stderr, "usage: cabi_check <vector file>\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1 || n == 0 || n > 4096) { fprintf(stderr, "bad header\n"); return 2; }
    uint64_t *scalars = malloc(n * 32), *points = malloc(n * 160), *want = malloc(n * 160), *got = malloc(n * 160);
    uint64_t *affine = malloc(n * 64), *want_dh = malloc(n * 64), *got_dh = malloc(n * 64);
    uint8_t *want_st = malloc(n), *got_st = malloc(n);
    if (fread(scalars, 32, n, f) != n || fread(points, 160, n, f) != n || fread(want, 160, n, f) != n ||
        fread(affine, 64, n, f) != n || fread(want_dh, 64, n, f) != n || fread(want_st, 1, n, f) != n) { fprintf(stderr, "short file\n"); return 2; }
    fclose(f);

    fourq_ctx *ctx = NULL;
    int rc = fourq_ctx_create(0, &ctx);
    if (rc != FOURQ_OK) return fail("fourq_ctx_create", rc, NULL);
    if ((rc = fourq_mul_endo_batch(ctx, scalars, points, got, (size_t)n)) != FOURQ_OK) return fail("fourq_mul_endo_batch", rc, ctx);
    if (memcmp(got, want, n * 160) != 0) { fprintf(stderr, "MUL_endo outputs differ from the oracle's\n"); return 1; }
    if ((rc = fourq_dh_endo_batch(ctx, scalars, affine, NULL, got_dh, got_st, (size_t)n)) != FOURQ_OK) return fail("fourq_dh_endo_batch", rc, ctx);
    if (memcmp(got_dh, want_dh, n * 64) != 0 || memcmp(got_st, want_st, n) != 0) { fprintf(stderr, "DH_endo outputs differ from the oracle's\n"); return 1; }
    /* pinned buffers and the transfer statistics, from C */
    void *pin = NULL;
    if ((rc = fourq_host_alloc(ctx, n * 160, &pin)) != FOURQ_OK) return fail("fourq_host_alloc", rc, ctx);
    if ((rc = fourq_mul_endo_batch(ctx, scalars, points, (uint64_t *)pin, (size_t)n)) != FOURQ_OK) return fail("fourq_mul_endo_batch (pinned out)", rc, ctx);
    fourq_host_stats st;
    if ((rc = fourq_ctx_host_stats(ctx, &st)) != FOURQ_OK) return fail("fourq_ctx_host_stats", rc, ctx);
    if (memcmp(pin, want, n * 160) != 0 || st.pinned_out != 1 || st.d2h_bytes != n * 160) { fprintf(stderr, "pinned path differs\n"); return 1; }
    /* round 5: copy durations only on request, and the in-kernel clock probe */
    if (st.h2d_ms != 0.0 || st.d2h_ms != 0.0) { fprintf(stderr, "copy durations reported although timing was not asked for\n"); return 1; }
    if ((rc = fourq_ctx_set_host_timing(ctx, 1)) != FOURQ_OK) return fail("fourq_ctx_set_host_timing", rc, ctx);
    if ((rc = fourq_mul_endo_batch(ctx, scalars, points, (uint64_t *)pin, (size_t)n)) != FOURQ_OK) return fail("fourq_mul_endo_batch (timed)", rc, ctx);
    if ((rc = fourq_ctx_host_stats(ctx, &st)) != FOURQ_OK) return fail("fourq_ctx_host_stats", rc, ctx);
    if (memcmp(pin, want, n * 160) != 0 || (st.d2h_bytes && !(st.h2d_ms > 0.0 && st.d2h_ms > 0.0))) { fprintf(stderr, "timed call differs\n"); return 1; }
    double mhz = 0, lo = 0, hi = 0, win = 0;
    int busy = -1;
    if (fourq_version() != FOURQ_ABI_VERSION) { fprintf(stderr, "library %d, header %d\n", fourq_version(), FOURQ_ABI_VERSION); return 1; }
    if ((rc = fourq_diag_clock(ctx, 2000, &mhz, &lo, &hi, &busy)) != FOURQ_OK) return fail("fourq_diag_clock", rc, ctx);
    if (!(lo > 100.0 && lo <= mhz && mhz <= hi && hi < 3000.0)) { fprintf(stderr, "implausible clock %.0f MHz (%.0f .. %.0f)\n", mhz, lo, hi); return 1; }
    if (busy != 0) { fprintf(stderr, "fourq_diag_clock says the idle stream was under load\n"); return 1; }
    if (fourq_diag_clock(ctx, 0, &mhz, NULL, NULL, NULL) != FOURQ_ERR_INVALID || fourq_diag_clock(NULL, 10, &mhz, NULL, NULL, NULL) != FOURQ_ERR_INVALID) {
        fprintf(stderr, "fourq_diag_clock accepts bad arguments\n"); return 1; }
    /* round 6: the bracket form -- two stamp launches on the context's stream around device-resident work, paired per CU */
    if ((rc = fourq_diag_clock_begin(ctx)) != FOURQ_OK) return fail("fourq_diag_clock_begin", rc, ctx);
    if (fourq_diag_clock_begin(ctx) != FOURQ_ERR_INVALID) { fprintf(stderr, "a second bracket was accepted\n"); return 1; }
    if ((rc = fourq_mul_endo_batch(ctx, scalars, points, (uint64_t *)pin, (size_t)n)) != FOURQ_OK || memcmp(pin, want, n * 160) != 0) return fail("fourq_mul_endo_batch inside a bracket", rc, ctx);
    if ((rc = fourq_diag_clock_stop(ctx)) != FOURQ_OK) return fail("fourq_diag_clock_stop", rc, ctx);
    if (fourq_diag_clock_stop(ctx) != FOURQ_ERR_INVALID) { fprintf(stderr, "a bracket was stopped twice\n"); return 1; }
    if ((rc = fourq_diag_clock_end(ctx, &mhz, &lo, &hi, &win)) != FOURQ_OK) return fail("fourq_diag_clock_end", rc, ctx);
    if (!(lo > 100.0 && lo <= mhz && mhz <= hi && hi < 3000.0 && win > 50.0 && win < 400000.0)) { fprintf(stderr, "implausible bracket clock %.0f MHz (%.0f .. %.0f) over %.0f us\n", mhz, lo, hi, win); return 1; }
    if (fourq_diag_clock_end(ctx, &mhz, NULL, NULL, NULL) != FOURQ_ERR_INVALID) { fprintf(stderr, "bracket ended twice\n"); return 1; }
    unsigned char small[48];
    memset(small, 0xEE, sizeof small);
    if ((rc = fourq_ctx_host_stats_sized(ctx, small, 40)) != FOURQ_OK || small[40] != 0xEE) { fprintf(stderr, "fourq_ctx_host_stats_sized wrote past the size it was given\n"); return 1; }
    fourq_host_free(ctx, pin);
    fourq_ctx_destroy(ctx);
    printf("cabi_check: %llu elements, MUL_endo and DH_endo bit-exact through the C ABI\n", (unsigned long long)n);
    return 0;
}
