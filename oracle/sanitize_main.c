/* TEST INFRASTRUCTURE ONLY.  Driver for the sanitizer build of the C oracle (make -C oracle sanitize): the oracle is the bulk checker of
 * every full-batch GPU test and of bench.py's parity gate, so undefined behaviour in its unsigned __int128 arithmetic would be a parity
 * hole nobody sees.  Reads one vector file written by tests/test_oracle_sanitize.py, runs every record through
 * fourq_oracle.c compiled with -fsanitize=address,undefined -fno-sanitize-recover=all, compares with the expected words.
 *
 * File: little-endian u64 words.  Header: magic 0x4f51524f55514652, number of records.  Record: op (0 mul, 1 dh, 2 decompose, 3 table),
 * kind (0 endo, 1 windowed), n, has_table, then the arrays in the order of the entry point's arguments, then the expected outputs
 * (dh: n x 8 output words followed by the n status bytes padded to whole words).
 * Exit code 0: every word equal.  A sanitizer report aborts the process (non-zero exit, text on stderr). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fourq_oracle.c"

static u64 *take(u64 **p, size_t words) { u64 *r = *p; *p += words; return r; }

int main(int argc, char **argv) {
    if (argc != 2) { fprintf(stderr, "usage: %s vectors.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    u64 *buf = malloc((size_t)bytes);
    if (!buf || fread(buf, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);
    u64 *p = buf;
    if (*take(&p, 1) != 0x4f51524f55514652ull) { fprintf(stderr, "bad magic\n"); return 2; }
    const u64 records = *take(&p, 1);
    size_t bad = 0, checked = 0;
    for (u64 r = 0; r < records; r++) {
        const u64 op = *take(&p, 1), kind = *take(&p, 1), n = *take(&p, 1), has_table = *take(&p, 1);
        if (op == 0) {
            const u64 *s = take(&p, 4 * n), *pts = has_table ? NULL : take(&p, 20 * n), *tab = has_table ? take(&p, 128) : NULL;
            const u64 *want = take(&p, 20 * n);
            u64 *got = malloc(20 * n * 8);
            fqo_mul_batch((int)kind, s, pts, tab, got, n);
            for (size_t i = 0; i < 20 * n; i++) bad += got[i] != want[i];
            checked += 20 * n;
            free(got);
        } else if (op == 1) {
            const u64 *s = take(&p, 4 * n), *pts = take(&p, 8 * n), *tab = has_table ? take(&p, 128) : NULL;
            const u64 *want = take(&p, 8 * n);
            const uint8_t *want_st = (const uint8_t *)take(&p, (n + 7) / 8);
            u64 *got = malloc(8 * n * 8);
            uint8_t *st = malloc(n);
            fqo_dh_batch((int)kind, s, pts, tab, got, st, n);
            for (size_t i = 0; i < 8 * n; i++) bad += got[i] != want[i];
            for (size_t i = 0; i < n; i++) bad += st[i] != want_st[i];
            checked += 9 * n;
            free(got); free(st);
        } else if (op == 2) {
            const u64 *s = take(&p, 4 * n), *want = take(&p, 4 * n);
            u64 *got = malloc(4 * n * 8);
            fqo_decompose_batch(s, got, n);
            for (size_t i = 0; i < 4 * n; i++) bad += got[i] != want[i];
            checked += 4 * n;
            free(got);
        } else if (op == 3) {
            const u64 *pt = take(&p, 20), *want = take(&p, 128);
            u64 got[128];
            if (kind == 0) fqo_table_endo(pt, got); else fqo_table_windowed(pt, got);
            for (size_t i = 0; i < 128; i++) bad += got[i] != want[i];
            checked += 128;
        } else {
            fprintf(stderr, "unknown op %llu\n", (unsigned long long)op);
            return 2;
        }
    }
    if ((char *)p != (char *)buf + bytes) { fprintf(stderr, "trailing bytes\n"); return 2; }
    free(buf);
    printf("sanitized oracle: %zu records, %zu words checked, %zu differ\n", (size_t)records, checked, bad);
    return bad ? 1 : 0;
}
