/* The library's HDF5 reader (jefferson-2.0_amd/csrc/jf_hdf5.c) under AddressSanitizer + UBSan + LeakSanitizer: every object of
 * the given files listed, looked up, read and asked for attributes; then the same over damaged copies (truncations, byte
 * flips, runs of 0xff = undefined addresses, bytes of the first 4 KB).  usage: hdf5_san_driver <scratch dir> <iterations> <file> ...
 * Test infrastructure (tests/test_sanitizers.py). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jf_hdf5.h"

static unsigned long long rs = 88172645463325252ULL;
static unsigned rnd(void) {
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return (unsigned)(rs >> 11);
}
static long opened = 0, reads = 0, reads_ok = 0, directed = 0;

static void exercise(const char *path) {
    jf_h5 *f;
    char err[256];
    if (jf_h5_open(path, &f, err, sizeof err)) return;
    opened++;
    char *names = NULL, v[64];
    if (jf_h5_list(f, jf_h5_root(f), &names) == 0) {
        char *save = NULL;
        for (char *n = strtok_r(names, "\n", &save); n; n = strtok_r(NULL, "\n", &save)) {
            uint64_t a;
            if (jf_h5_lookup(f, n, &a) != 0) continue;
            jf_h5_attr_str(f, a, "Type", v, sizeof v);
            jf_h5_attr_str(f, a, "Units", v, sizeof v);
            if (jf_h5_is_dataset(f, a) != 1) continue;
            int rank;
            uint64_t dims[JF_H5_MAXRANK];
            double *d = NULL;
            reads++;
            if (jf_h5_read_f64(f, a, &rank, dims, &d) == 0) reads_ok++;
            free(d);
        }
        free(names);
    }
    jf_h5_attr_str(f, jf_h5_root(f), "DataType", v, sizeof v);
    jf_h5_attr_str(f, jf_h5_root(f), "Extra3", v, sizeof v);
    uint64_t a;
    if (jf_h5_lookup(f, "nested/seven", &a) == 0) {
        int rank;
        uint64_t dims[JF_H5_MAXRANK];
        double *d = NULL;
        jf_h5_read_f64(f, a, &rank, dims, &d);
        free(d);
    }
    jf_h5_close(f);
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    char tmp[1024];
    snprintf(tmp, sizeof tmp, "%s/damaged.h5", argv[1]);
    const int iters = atoi(argv[2]);
    for (int k = 3; k < argc; k++) {
        exercise(argv[k]);
        FILE *fp = fopen(argv[k], "rb");
        if (!fp) return 3;
        fseek(fp, 0, SEEK_END);
        const long n = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        unsigned char *b = malloc((size_t)n), *m = malloc((size_t)n);
        if (!b || !m || n < 16 || fread(b, 1, (size_t)n, fp) != (size_t)n) return 3;
        fclose(fp);
        for (int it = 0; it < iters; it++) {
            memcpy(m, b, (size_t)n);
            long len = n;
            switch (it % 4) {
            case 0: len = (long)(rnd() % (unsigned long)n); break;
            case 1:
                for (int c = 1 + (int)(rnd() % 8); c > 0; c--) m[rnd() % (unsigned long)n] = (unsigned char)rnd();
                break;
            case 2:
                for (int c = 1 + (int)(rnd() % 3); c > 0; c--) memset(m + rnd() % (unsigned long)(n - 8), 0xff, 8);
                break;
            default: m[rnd() % (unsigned long)(n > 4096 ? 4096 : n)] = (unsigned char)rnd(); break;
            }
            fp = fopen(tmp, "wb");
            if (!fp) return 3;
            fwrite(m, 1, (size_t)len, fp);
            fclose(fp);
            exercise(tmp);
        }
        /* directed: every version-2 B-tree header of the file gets a record size smaller than what its record
         * type's callback reads, and its root pointed at a leaf that ends with the file -- the records' heap IDs would be
         * read past the buffer; the reader must refuse the tree */
        for (long o = 0; o + 34 <= n; o++) {
            if (memcmp(b + o, "BTHD", 4) != 0) continue;
            for (unsigned rec = 1; rec <= 7; rec += 2) {
                const unsigned nrec = 3;
                const long len = n + 6 + (long)(nrec * rec);
                unsigned char *x = calloc(1, (size_t)len);
                if (!x) return 3;
                memcpy(x, b, (size_t)n);
                memcpy(x + n, "BTLF", 4);
                x[n + 5] = b[o + 5]; /* the tree's own type */
                x[o + 10] = (unsigned char)rec, x[o + 11] = 0; /* record size */
                x[o + 12] = x[o + 13] = 0;                   /* depth 0: the root is a leaf */
                for (int k8 = 0; k8 < 8; k8++) x[o + 16 + k8] = (unsigned char)((unsigned long long)n >> (8 * k8));
                x[o + 24] = (unsigned char)nrec, x[o + 25] = 0;
                fp = fopen(tmp, "wb");
                if (!fp) return 3;
                fwrite(x, 1, (size_t)len, fp);
                fclose(fp);
                free(x);
                exercise(tmp);
                directed++;
            }
        }
        free(b);
        free(m);
    }
    printf("%ld directed B-tree cases; ", directed);
    printf("opened %ld files, %ld of %ld dataset reads succeeded\n", opened, reads_ok, reads);
    return opened > 0 && reads_ok > 0 ? 0 : 1;
}
