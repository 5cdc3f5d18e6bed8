/*
 * jf_hdf5.c -- reader for the subset of HDF5 that SOFA files use ("SOFA/other HRTF sets", the reference's
 * TODO FuturePlans.md:21).  Own code from the HDF5 File Format Specification 3.0; the only dependency is zlib's
 * uncompress() for deflated chunks.  Host-side file I/O like the WAV reader: nothing here runs per block.
 *
 * Understood
 *   superblock versions 0-3 (a user block in front of it: searched at 0, 512, 1024, ...), offsets and lengths of 2-8 bytes;
 *   object headers version 1 and 2 ("OHDR"), continuation blocks;
 *   groups: symbol tables (B-tree version 1 + local heap + "SNOD" nodes), compact link messages, dense link storage
 *     (fractal heap + version-2 B-tree name index, a root indirect block of direct blocks, B-tree depth <= 1) -- what
 *     netCDF-4's creation-order tracked groups use;
 *   datasets of integers / IEEE floats (1, 2, 4, 8 bytes, either byte order): compact, contiguous and chunked layouts
 *     (layout message versions 1-4; chunk indices: version-1 B-tree, single chunk, implicit, fixed array);
 *     filters deflate, shuffle, fletcher32 (stripped, not verified);
 *   string attributes, fixed-length or variable-length (global heap), in the header or in dense attribute storage.
 * Refused with a message: everything else (extensible-array and B-tree-2 chunk indices, i.e. unlimited dimensions under
 *   `libver=latest`; szip/n-bit/scale-offset and third-party filters; shared messages; filtered fractal heaps; external
 *   storage; virtual datasets).  Checksums are not verified.  Every access is bounds-checked against the file image: a
 *   damaged file yields an error, never a fault (tests/test_sofa.py: truncations and byte flips).
 */
#include "jf_hdf5.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define UNDEF UINT64_MAX
#define MAX_FILE_BYTES ((uint64_t)1 << 32)
#define MAX_ELEMENTS ((uint64_t)1 << 28)
#define MAX_MSGS 4096
#define MAX_LINKS 65536

struct jf_h5 {
    uint8_t *buf;
    uint64_t size;
    uint64_t base; /* where the superblock lies: addresses count from here */
    int so, sl;    /* size of offsets / of lengths */
    uint64_t root; /* root group's object header */
    char err[240];
};

static int fail(jf_h5 *f, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(f->err, sizeof f->err, fmt, ap);
    va_end(ap);
    return -1;
}

/* n bytes at file address addr, or NULL (error text set) */
static const uint8_t *at(jf_h5 *f, uint64_t addr, uint64_t n) {
    if (addr == UNDEF || addr > f->size || f->base > f->size - addr || n > f->size - addr - f->base) {
        fail(f, "HDF5: %llu bytes at address %llu lie outside the file (%llu bytes)", (unsigned long long)n,
             (unsigned long long)addr, (unsigned long long)f->size);
        return NULL;
    }
    return f->buf + f->base + addr;
}

static uint64_t rd(const uint8_t *p, int n) {
    uint64_t v = 0;
    for (int i = n - 1; i >= 0; i--) v = (v << 8) | p[i];
    return v;
}
static uint64_t rd_off(const jf_h5 *f, const uint8_t *p) {
    const uint64_t v = rd(p, f->so);
    return (f->so < 8 && v == (((uint64_t)1 << (8 * f->so)) - 1)) ? UNDEF : v;
}
static uint64_t rd_len(const jf_h5 *f, const uint8_t *p) { return rd(p, f->sl); }
static int log2_floor(uint64_t v) {
    int r = 0;
    while (v >>= 1) r++;
    return r;
}
static int enc_bytes(uint64_t limit) { return log2_floor(limit) / 8 + 1; } /* H5VM_limit_enc_size */

/* ---- object headers --------------------------------------------------------------------------------------------------- */

typedef struct {
    unsigned type, flags;
    const uint8_t *p;
    uint32_t size;
} h5_msg;
typedef struct {
    h5_msg m[MAX_MSGS];
    int n;
} h5_msgs;

static int push_msg(jf_h5 *f, h5_msgs *ms, unsigned type, unsigned flags, const uint8_t *p, uint32_t size) {
    if (ms->n >= MAX_MSGS) return fail(f, "HDF5: more than %d header messages", MAX_MSGS);
    ms->m[ms->n].type = type;
    ms->m[ms->n].flags = flags;
    ms->m[ms->n].p = p;
    ms->m[ms->n].size = size;
    ms->n++;
    return 0;
}

static int read_header(jf_h5 *f, uint64_t addr, h5_msgs *ms) {
    ms->n = 0;
    const uint8_t *p = at(f, addr, 16);
    if (!p) return -1;
    struct {
        uint64_t addr, len;
    } chunk[256];
    int n_chunks = 0, done = 0;
    if (memcmp(p, "OHDR", 4) == 0) {
        if (p[4] != 2) return fail(f, "HDF5: object header version %d", p[4]);
        const unsigned hf = p[5];
        uint64_t q = 6;
        if (hf & 0x20) q += 16;
        if (hf & 0x10) q += 4;
        const int szb = 1 << (hf & 3);
        const uint8_t *s = at(f, addr + q, (uint64_t)szb);
        if (!s) return -1;
        chunk[0].addr = addr + q + (uint64_t)szb;
        chunk[0].len = rd(s, szb);
        n_chunks = 1;
        const int mh = 4 + ((hf & 0x04) ? 2 : 0);
        while (done < n_chunks) {
            const uint64_t ca = chunk[done].addr, cl = chunk[done].len;
            done++;
            const uint8_t *c = at(f, ca, cl);
            if (!c) return -1;
            uint64_t o = 0;
            while (o + (uint64_t)mh <= cl) {
                const unsigned type = c[o];
                const uint32_t size = (uint32_t)rd(c + o + 1, 2);
                const unsigned flags = c[o + 3];
                if (size > cl - o - (uint64_t)mh) return fail(f, "HDF5: a header message runs past its chunk");
                const uint8_t *d = c + o + mh;
                if (type == 0x10) {
                    if (size < (uint32_t)(f->so + f->sl)) return fail(f, "HDF5: short continuation message");
                    if (n_chunks >= 256) return fail(f, "HDF5: too many header continuations");
                    const uint64_t a = rd_off(f, d), l = rd_len(f, d + f->so);
                    const uint8_t *k = at(f, a, l);
                    if (!k) return -1;
                    if (l < 8 || memcmp(k, "OCHK", 4) != 0) return fail(f, "HDF5: no OCHK at a continuation");
                    chunk[n_chunks].addr = a + 4;
                    chunk[n_chunks].len = l - 8; /* signature in front, checksum behind */
                    n_chunks++;
                } else if (type != 0) {
                    if (push_msg(f, ms, type, flags, d, size)) return -1;
                }
                o += (uint64_t)mh + size;
            }
        }
        return 0;
    }
    if (p[0] != 1) return fail(f, "HDF5: no object header at address %llu", (unsigned long long)addr);
    const unsigned n_msgs = (unsigned)rd(p + 2, 2);
    unsigned seen = 0;
    chunk[0].addr = addr + 16;
    chunk[0].len = rd(p + 8, 4);
    n_chunks = 1;
    while (done < n_chunks && seen < n_msgs) {
        const uint64_t ca = chunk[done].addr, cl = chunk[done].len;
        done++;
        const uint8_t *c = at(f, ca, cl);
        if (!c) return -1;
        uint64_t o = 0;
        while (o + 8 <= cl && seen < n_msgs) {
            const unsigned type = (unsigned)rd(c + o, 2);
            const uint32_t size = (uint32_t)rd(c + o + 2, 2);
            const unsigned flags = c[o + 4];
            if (size > cl - o - 8) return fail(f, "HDF5: a header message runs past its chunk");
            const uint8_t *d = c + o + 8;
            seen++;
            if (type == 0x10) {
                if (size < (uint32_t)(f->so + f->sl)) return fail(f, "HDF5: short continuation message");
                if (n_chunks >= 256) return fail(f, "HDF5: too many header continuations");
                chunk[n_chunks].addr = rd_off(f, d);
                chunk[n_chunks].len = rd_len(f, d + f->so);
                n_chunks++;
            } else if (type != 0) {
                if (push_msg(f, ms, type, flags, d, size)) return -1;
            }
            o += 8 + (uint64_t)size;
        }
    }
    return 0;
}

static const h5_msg *find_msg(const h5_msgs *ms, unsigned type) {
    for (int i = 0; i < ms->n; i++)
        if (ms->m[i].type == type) return &ms->m[i];
    return NULL;
}

/* ---- fractal heaps and version-2 B-trees (dense link / attribute storage) ------------------------------------------------ */

typedef struct {
    int id_len, width, off_bytes, len_bytes, cur_rows;
    uint64_t start, max_direct, root;
} h5_fheap;

static int fheap_open(jf_h5 *f, uint64_t addr, h5_fheap *h) {
    const uint64_t need = 4 + 1 + 2 + 2 + 1 + 4 + (uint64_t)f->sl + f->so + f->sl + f->so + 8 * (uint64_t)f->sl + 2 +
                          2 * (uint64_t)f->sl + 2 + 2 + f->so + 2;
    const uint8_t *p = at(f, addr, need);
    if (!p) return -1;
    if (memcmp(p, "FRHP", 4) != 0 || p[4] != 0) return fail(f, "HDF5: no fractal heap at address %llu", (unsigned long long)addr);
    h->id_len = (int)rd(p + 5, 2);
    if (rd(p + 7, 2) != 0) return fail(f, "HDF5: a filtered fractal heap is not supported");
    const uint64_t max_man = rd(p + 10, 4);
    const uint8_t *q = p + 14 + f->sl + f->so + f->sl + f->so + 8 * f->sl;
    h->width = (int)rd(q, 2);
    h->start = rd_len(f, q + 2);
    h->max_direct = rd_len(f, q + 2 + f->sl);
    const int heap_bits = (int)rd(q + 2 + 2 * f->sl, 2);
    h->root = rd_off(f, q + 2 + 2 * f->sl + 4);
    h->cur_rows = (int)rd(q + 2 + 2 * f->sl + 4 + f->so, 2);
    if (h->width < 1 || h->start < 16 || (h->start & (h->start - 1)) || h->max_direct < h->start ||
        (h->max_direct & (h->max_direct - 1)) || heap_bits < 8 || heap_bits > 64 || max_man == 0)
        return fail(f, "HDF5: fractal heap with an odd doubling table");
    h->off_bytes = (heap_bits + 7) / 8;
    const int a = (log2_floor(h->max_direct) + 7) / 8, b = enc_bytes(max_man);
    h->len_bytes = a < b ? a : b;
    if (1 + h->off_bytes + h->len_bytes > h->id_len) return fail(f, "HDF5: fractal heap IDs too short for the heap");
    return 0;
}

/* the bytes of a managed object */
static const uint8_t *fheap_object(jf_h5 *f, const h5_fheap *h, const uint8_t *id, uint64_t *len) {
    const unsigned kind = (id[0] >> 4) & 3;
    if ((id[0] >> 6) != 0 || kind != 0) {
        fail(f, "HDF5: a %s fractal-heap object is not supported", kind == 1 ? "huge" : kind == 2 ? "tiny" : "reserved");
        return NULL;
    }
    const uint64_t off = rd(id + 1, h->off_bytes);
    *len = rd(id + 1 + h->off_bytes, h->len_bytes);
    if (h->cur_rows == 0) { /* the root block is the one direct block */
        if (off > h->max_direct || *len > h->max_direct - off) {
            fail(f, "HDF5: heap object outside the root block");
            return NULL;
        }
        return at(f, h->root + off, *len);
    }
    uint64_t cum = 0;
    for (int r = 0; r < h->cur_rows && r < 64; r++) {
        const uint64_t size = r < 2 ? h->start : h->start << (r - 1);
        if (size > h->max_direct) break; /* rows of indirect blocks: not walked */
        for (int c = 0; c < h->width; c++, cum += size) {
            if (off >= cum + size) continue;
            if (*len > cum + size - off) {
                fail(f, "HDF5: heap object runs past its block");
                return NULL;
            }
            const uint64_t entry = h->root + 5 + (uint64_t)f->so + (uint64_t)h->off_bytes + (uint64_t)(r * h->width + c) * f->so;
            const uint8_t *e = at(f, entry, (uint64_t)f->so);
            const uint8_t *sig = at(f, h->root, 4);
            if (!e || !sig) return NULL;
            if (memcmp(sig, "FHIB", 4) != 0) {
                fail(f, "HDF5: no indirect block at the fractal heap's root");
                return NULL;
            }
            return at(f, rd_off(f, e) + (off - cum), *len);
        }
    }
    fail(f, "HDF5: fractal heap deeper than one indirect block is not supported");
    return NULL;
}

typedef int (*bt2_fn)(jf_h5 *f, const uint8_t *rec, void *ctx);

static int bt2_leaf(jf_h5 *f, uint64_t addr, unsigned nrec, unsigned rec_size, bt2_fn fn, void *ctx) {
    const uint8_t *p = at(f, addr, 6 + (uint64_t)nrec * rec_size);
    if (!p) return -1;
    if (memcmp(p, "BTLF", 4) != 0) return fail(f, "HDF5: no B-tree leaf at address %llu", (unsigned long long)addr);
    for (unsigned i = 0; i < nrec; i++)
        if (fn(f, p + 6 + (uint64_t)i * rec_size, ctx)) return -1;
    return 0;
}

/* `type` is the tree's record type the caller's callback understands (5: a group's links by name, 8: an object's
 * attributes by name) and `min_rec` the number of bytes that callback reads of a record: the file's own record size is
 * only trusted to be at least that (a record size of 1 walked the callbacks off the end of the file: ADVICE r05) */
static int bt2_walk(jf_h5 *f, uint64_t addr, unsigned type, unsigned min_rec, bt2_fn fn, void *ctx) {
    const uint8_t *p = at(f, addr, 16 + (uint64_t)f->so + 2 + (uint64_t)f->sl);
    if (!p) return -1;
    if (memcmp(p, "BTHD", 4) != 0 || p[4] != 0) return fail(f, "HDF5: no version-2 B-tree at address %llu", (unsigned long long)addr);
    if (p[5] != type) return fail(f, "HDF5: a version-2 B-tree of type %u where type %u belongs", (unsigned)p[5], type);
    const uint64_t node_size = rd(p + 6, 4);
    const unsigned rec_size = (unsigned)rd(p + 10, 2), depth = (unsigned)rd(p + 12, 2);
    if (rec_size < min_rec) return fail(f, "HDF5: B-tree records of %u bytes, shorter than the %u their type needs", rec_size, min_rec);
    const uint64_t root = rd_off(f, p + 16);
    const unsigned nrec = (unsigned)rd(p + 16 + f->so, 2);
    if (rec_size == 0 || node_size < 16) return fail(f, "HDF5: B-tree with empty records");
    if (root == UNDEF || nrec == 0) return 0;
    if (depth == 0) return bt2_leaf(f, root, nrec, rec_size, fn, ctx);
    if (depth > 1) return fail(f, "HDF5: version-2 B-tree of depth %u is not supported", depth);
    const int nb = enc_bytes((node_size - 10) / rec_size);
    const uint64_t ptr = (uint64_t)f->so + (uint64_t)nb;
    const uint8_t *q = at(f, root, 6 + (uint64_t)nrec * rec_size + (uint64_t)(nrec + 1) * ptr);
    if (!q) return -1;
    if (memcmp(q, "BTIN", 4) != 0) return fail(f, "HDF5: no B-tree node at address %llu", (unsigned long long)root);
    for (unsigned i = 0; i < nrec; i++)
        if (fn(f, q + 6 + (uint64_t)i * rec_size, ctx)) return -1;
    const uint8_t *c = q + 6 + (uint64_t)nrec * rec_size;
    for (unsigned i = 0; i <= nrec; i++, c += ptr)
        if (bt2_leaf(f, rd_off(f, c), (unsigned)rd(c + f->so, nb), rec_size, fn, ctx)) return -1;
    return 0;
}

/* ---- groups ------------------------------------------------------------------------------------------------------------- */

typedef struct {
    char *name;
    uint64_t addr;
} h5_link;
typedef struct {
    h5_link *l;
    int n, cap;
} h5_links;

static void free_links(h5_links *ls) {
    for (int i = 0; i < ls->n; i++) free(ls->l[i].name);
    free(ls->l);
    ls->l = NULL;
    ls->n = ls->cap = 0;
}

static int push_link(jf_h5 *f, h5_links *ls, const uint8_t *name, uint64_t len, uint64_t addr) {
    if (ls->n >= MAX_LINKS) return fail(f, "HDF5: more than %d links in a group", MAX_LINKS);
    if (ls->n == ls->cap) {
        const int cap = ls->cap ? 2 * ls->cap : 32;
        h5_link *l = (h5_link *)realloc(ls->l, sizeof(h5_link) * (size_t)cap);
        if (!l) return fail(f, "out of memory");
        ls->l = l;
        ls->cap = cap;
    }
    char *s = (char *)malloc(len + 1);
    if (!s) return fail(f, "out of memory");
    memcpy(s, name, len);
    s[len] = 0;
    ls->l[ls->n].name = s;
    ls->l[ls->n].addr = addr;
    ls->n++;
    return 0;
}

/* a link message (header message 0x06, or an object of a group's fractal heap) */
static int parse_link(jf_h5 *f, const uint8_t *p, uint64_t size, h5_links *out) {
    if (size < 3 || p[0] != 1) return fail(f, "HDF5: link message version %d", size ? p[0] : -1);
    const unsigned fl = p[1];
    uint64_t o = 2;
    unsigned type = 0;
    if (fl & 0x08) type = p[o++];
    if (fl & 0x04) o += 8;
    if (fl & 0x10) o += 1;
    const int nb = 1 << (fl & 3);
    if (o + (uint64_t)nb > size) return fail(f, "HDF5: short link message");
    const uint64_t len = rd(p + o, nb);
    o += (uint64_t)nb;
    if (len > size - o) return fail(f, "HDF5: a link's name runs past its message");
    if (type != 0) return 0; /* soft and external links: not followed */
    if (size - o - len < (uint64_t)f->so) return fail(f, "HDF5: short link message");
    return push_link(f, out, p + o, len, rd_off(f, p + o + len));
}

static int symtab_walk(jf_h5 *f, uint64_t node, const uint8_t *heap, uint64_t heap_size, h5_links *out, int guard) {
    if (guard > 16) return fail(f, "HDF5: group B-tree too deep");
    const uint8_t *p = at(f, node, 8 + 2 * (uint64_t)f->so);
    if (!p) return -1;
    if (memcmp(p, "TREE", 4) != 0 || p[4] != 0) return fail(f, "HDF5: no group B-tree node at address %llu", (unsigned long long)node);
    const unsigned level = p[5], n = (unsigned)rd(p + 6, 2);
    const uint64_t body = 8 + 2 * (uint64_t)f->so, step = (uint64_t)f->sl + f->so;
    const uint8_t *b = at(f, node + body, (uint64_t)n * step + f->sl);
    if (!b) return -1;
    for (unsigned i = 0; i < n; i++) {
        const uint64_t child = rd_off(f, b + (uint64_t)i * step + f->sl);
        if (level > 0) {
            if (symtab_walk(f, child, heap, heap_size, out, guard + 1)) return -1;
            continue;
        }
        const uint8_t *s = at(f, child, 8);
        if (!s) return -1;
        if (memcmp(s, "SNOD", 4) != 0) return fail(f, "HDF5: no symbol node at address %llu", (unsigned long long)child);
        const unsigned n_sym = (unsigned)rd(s + 6, 2);
        const uint64_t esz = 2 * (uint64_t)f->so + 24;
        const uint8_t *e = at(f, child + 8, (uint64_t)n_sym * esz);
        if (!e) return -1;
        for (unsigned k = 0; k < n_sym; k++, e += esz) {
            const uint64_t no = rd_off(f, e);
            if (no >= heap_size) return fail(f, "HDF5: a link's name lies outside the group's heap");
            const void *z = memchr(heap + no, 0, heap_size - no);
            if (!z) return fail(f, "HDF5: unterminated name in a group's heap");
            if (push_link(f, out, heap + no, (uint64_t)((const uint8_t *)z - (heap + no)), rd_off(f, e + f->so))) return -1;
        }
    }
    return 0;
}

typedef struct {
    const h5_fheap *heap;
    h5_links *out;
} dense_link_ctx;

static int dense_link_rec(jf_h5 *f, const uint8_t *rec, void *vctx) {
    dense_link_ctx *c = (dense_link_ctx *)vctx;
    uint64_t len = 0;
    const uint8_t *obj = fheap_object(f, c->heap, rec + 4, &len); /* hash (4), heap ID */
    return obj ? parse_link(f, obj, len, c->out) : -1;
}

static int group_links(jf_h5 *f, uint64_t addr, h5_links *out) {
    h5_msgs *ms = (h5_msgs *)malloc(sizeof(h5_msgs));
    if (!ms) return fail(f, "out of memory");
    int rc = read_header(f, addr, ms);
    for (int i = 0; rc == 0 && i < ms->n; i++) {
        const h5_msg *m = &ms->m[i];
        if (m->type == 0x11) {
            if (m->size < 2 * (uint32_t)f->so) {
                rc = fail(f, "HDF5: short symbol table message");
                break;
            }
            const uint64_t bt = rd_off(f, m->p), hp = rd_off(f, m->p + f->so);
            const uint8_t *h = at(f, hp, 8 + 2 * (uint64_t)f->sl + f->so);
            if (!h) {
                rc = -1;
                break;
            }
            if (memcmp(h, "HEAP", 4) != 0) {
                rc = fail(f, "HDF5: no local heap at address %llu", (unsigned long long)hp);
                break;
            }
            const uint64_t hs = rd_len(f, h + 8);
            const uint8_t *hd = at(f, rd_off(f, h + 8 + 2 * f->sl), hs);
            rc = hd ? symtab_walk(f, bt, hd, hs, out, 0) : -1;
        } else if (m->type == 0x06) {
            if (m->flags & 2) continue; /* shared: not a thing for links */
            rc = parse_link(f, m->p, m->size, out);
        } else if (m->type == 0x02) {
            if (m->size < 2 || m->p[0] != 0) {
                rc = fail(f, "HDF5: link info message version %d", m->size ? m->p[0] : -1);
                break;
            }
            uint64_t o = 2 + ((m->p[1] & 1) ? 8 : 0);
            if (m->size < o + 2 * (uint64_t)f->so) {
                rc = fail(f, "HDF5: short link info message");
                break;
            }
            const uint64_t heap = rd_off(f, m->p + o), bt = rd_off(f, m->p + o + f->so);
            if (heap == UNDEF || bt == UNDEF) continue; /* compact: the links are messages of this header */
            h5_fheap fh;
            dense_link_ctx ctx = {&fh, out};
            rc = fheap_open(f, heap, &fh);
            if (rc == 0) rc = bt2_walk(f, bt, 5, 4 + (unsigned)fh.id_len, dense_link_rec, &ctx); /* hash + heap ID */
        }
    }
    free(ms);
    return rc;
}

/* ---- datatypes, dataspaces --------------------------------------------------------------------------------------------- */

typedef struct {
    int cls;        /* 0 integer, 1 float, 3 string, 9 variable-length */
    uint32_t size;  /* bytes of an element */
    int big_endian, is_signed;
    int vlen_string;
} h5_type;

static int parse_type(jf_h5 *f, const uint8_t *p, uint64_t size, h5_type *t) {
    if (size < 8) return fail(f, "HDF5: short datatype message");
    memset(t, 0, sizeof *t);
    t->cls = p[0] & 15;
    t->size = (uint32_t)rd(p + 4, 4);
    t->big_endian = p[1] & 1;
    if (t->cls == 0) t->is_signed = (p[1] >> 3) & 1;
    if (t->cls == 4) t->cls = 0; /* a bitfield: read as an unsigned integer */
    if (t->cls == 1) {
        if (size < 20) return fail(f, "HDF5: short floating-point datatype");
        /* IEEE single or double: exponent and mantissa where they belong (properties: bit offset, precision, exponent
         * location and size, mantissa location and size, bias) */
        const unsigned prec = (unsigned)rd(p + 10, 2), eloc = p[12], esz = p[13], mloc = p[14], msz = p[15];
        const uint64_t bias = rd(p + 16, 4);
        const int f32 = t->size == 4 && prec == 32 && eloc == 23 && esz == 8 && mloc == 0 && msz == 23 && bias == 127;
        const int f64 = t->size == 8 && prec == 64 && eloc == 52 && esz == 11 && mloc == 0 && msz == 52 && bias == 1023;
        if (!f32 && !f64) return fail(f, "HDF5: a floating-point type that is neither IEEE single nor double");
        if (p[1] & 0x40) return fail(f, "HDF5: VAX byte order");
    }
    if (t->cls == 9) t->vlen_string = (p[1] & 15) == 1;
    return 0;
}

typedef struct {
    int rank;
    uint64_t dims[JF_H5_MAXRANK];
    uint64_t n; /* elements */
} h5_space;

static int parse_space(jf_h5 *f, const uint8_t *p, uint64_t size, h5_space *s) {
    if (size < 4) return fail(f, "HDF5: short dataspace message");
    memset(s, 0, sizeof *s);
    const int ver = p[0];
    s->rank = p[1];
    uint64_t o;
    if (ver == 1) o = 8;
    else if (ver == 2) {
        o = 4;
        if (p[3] == 2) { /* null dataspace: handed out as one dimension of extent 0, so that a caller's product of the
                          * extents is the element count (rank 0 is a scalar: one element) */
            s->rank = 1;
            s->dims[0] = 0;
            s->n = 0;
            return 0;
        }
    } else
        return fail(f, "HDF5: dataspace message version %d", ver);
    if (s->rank > JF_H5_MAXRANK) return fail(f, "HDF5: a dataspace of rank %d", s->rank);
    if (size < o + (uint64_t)s->rank * f->sl) return fail(f, "HDF5: short dataspace message");
    s->n = 1;
    for (int i = 0; i < s->rank; i++) {
        s->dims[i] = rd_len(f, p + o + (uint64_t)i * f->sl);
        if (s->dims[i] != 0 && s->n > MAX_ELEMENTS / s->dims[i]) return fail(f, "HDF5: a dataset of more than 2^28 elements");
        s->n *= s->dims[i];
    }
    return 0;
}

/* ---- filters ------------------------------------------------------------------------------------------------------------ */

typedef struct {
    int n;
    unsigned id[8], flags[8], cd0[8];
} h5_filters;

static int parse_filters(jf_h5 *f, const uint8_t *p, uint64_t size, h5_filters *fl) {
    memset(fl, 0, sizeof *fl);
    if (size < 2) return fail(f, "HDF5: short filter pipeline message");
    const int ver = p[0];
    fl->n = p[1];
    if (fl->n > 8) return fail(f, "HDF5: %d filters in a pipeline", fl->n);
    uint64_t o = ver == 1 ? 8 : 2;
    if (ver != 1 && ver != 2) return fail(f, "HDF5: filter pipeline message version %d", ver);
    for (int i = 0; i < fl->n; i++) {
        if (o + 4 > size) return fail(f, "HDF5: short filter pipeline message");
        const unsigned id = (unsigned)rd(p + o, 2);
        uint64_t name_len = 0;
        o += 2;
        if (ver == 1 || id >= 256) {
            name_len = rd(p + o, 2);
            o += 2;
        }
        if (o + 4 > size) return fail(f, "HDF5: short filter pipeline message");
        const unsigned flags = (unsigned)rd(p + o, 2), n_cd = (unsigned)rd(p + o + 2, 2);
        o += 4;
        if (ver == 1) name_len = (name_len + 7) & ~(uint64_t)7;
        o += name_len;
        if (o + 4 * (uint64_t)n_cd > size) return fail(f, "HDF5: short filter pipeline message");
        fl->id[i] = id;
        fl->flags[i] = flags;
        fl->cd0[i] = n_cd ? (unsigned)rd(p + o, 4) : 0;
        o += 4 * (uint64_t)n_cd;
        if (ver == 1 && (n_cd & 1)) o += 4;
    }
    return 0;
}

/* the chunk as stored -> the chunk's elements: *buf (malloc'd, *len bytes) is replaced filter by filter, last filter first */
static int unfilter(jf_h5 *f, const h5_filters *fl, uint32_t mask, uint64_t nominal, uint32_t elem, uint8_t **buf, uint64_t *len) {
    for (int i = fl->n - 1; i >= 0; i--) {
        if (mask & (1u << i)) continue;
        if (fl->id[i] == 3) { /* fletcher32: four bytes of checksum behind the data */
            if (*len < 4) return fail(f, "HDF5: a chunk shorter than its checksum");
            *len -= 4;
        } else if (fl->id[i] == 1) {
            uint8_t *out = (uint8_t *)malloc(nominal ? nominal : 1);
            if (!out) return fail(f, "out of memory");
            uLongf n = (uLongf)nominal;
            const int z = uncompress(out, &n, *buf, (uLong)*len);
            if (z != Z_OK) {
                free(out);
                return fail(f, "HDF5: a deflated chunk does not inflate (zlib %d)", z);
            }
            free(*buf);
            *buf = out;
            *len = n;
        } else if (fl->id[i] == 2) {
            const uint64_t es = fl->cd0[i] ? fl->cd0[i] : elem;
            if (es > 1 && *len >= es) {
                const uint64_t n = *len / es;
                uint8_t *out = (uint8_t *)malloc(*len);
                if (!out) return fail(f, "out of memory");
                for (uint64_t j = 0; j < es; j++)
                    for (uint64_t k = 0; k < n; k++) out[k * es + j] = (*buf)[j * n + k];
                memcpy(out + n * es, *buf + n * es, *len - n * es);
                free(*buf);
                *buf = out;
            }
        } else {
            return fail(f, "HDF5: filter %u is not supported (deflate, shuffle and fletcher32 are)", fl->id[i]);
        }
    }
    return 0;
}

/* ---- datasets ----------------------------------------------------------------------------------------------------------- */

typedef struct {
    jf_h5 *f;
    const h5_space *sp;
    const h5_filters *fl;
    uint32_t elem;
    uint64_t cdim[JF_H5_MAXRANK]; /* chunk dimensions in elements */
    uint64_t nominal;             /* bytes of a whole chunk */
    uint8_t *out;                 /* the dataset's elements, row-major */
} chunk_ctx;

/* one stored chunk (file address, stored bytes, filter mask) whose first element is at `off` of the dataset */
static int place_chunk(chunk_ctx *c, uint64_t addr, uint64_t stored, uint32_t mask, const uint64_t *off) {
    jf_h5 *f = c->f;
    const int rank = c->sp->rank;
    for (int d = 0; d < rank; d++)
        if (off[d] >= c->sp->dims[d] || off[d] % c->cdim[d]) return fail(f, "HDF5: a chunk outside its dataset");
    const uint8_t *src = at(f, addr, stored);
    if (!src) return -1;
    uint8_t *buf = NULL;
    uint64_t len = stored;
    if (c->fl->n) {
        buf = (uint8_t *)malloc(stored ? stored : 1);
        if (!buf) return fail(f, "out of memory");
        memcpy(buf, src, stored);
        if (unfilter(f, c->fl, mask, c->nominal, c->elem, &buf, &len)) {
            free(buf);
            return -1;
        }
        src = buf;
    }
    int rc = 0;
    if (len < c->nominal) rc = fail(f, "HDF5: a chunk of %llu bytes where %llu are due", (unsigned long long)len, (unsigned long long)c->nominal);
    if (rc == 0) {
        /* rows along the last dimension, clipped at the dataset's edges */
        uint64_t idx[JF_H5_MAXRANK] = {0}, take[JF_H5_MAXRANK];
        for (int d = 0; d < rank; d++) take[d] = c->sp->dims[d] - off[d] < c->cdim[d] ? c->sp->dims[d] - off[d] : c->cdim[d];
        const uint64_t row = take[rank - 1] * c->elem;
        for (;;) {
            uint64_t so = 0, dof = 0;
            for (int d = 0; d < rank - 1; d++) {
                so = so * c->cdim[d] + idx[d];
                dof = dof * c->sp->dims[d] + off[d] + idx[d];
            }
            so = so * c->cdim[rank - 1];
            dof = dof * c->sp->dims[rank - 1] + off[rank - 1];
            memcpy(c->out + dof * c->elem, src + so * c->elem, row);
            int d = rank - 2;
            while (d >= 0 && ++idx[d] == take[d]) idx[d--] = 0;
            if (d < 0) break;
        }
    }
    free(buf);
    return rc;
}

static int chunk_btree(chunk_ctx *c, uint64_t node, int guard) {
    jf_h5 *f = c->f;
    if (guard > 16) return fail(f, "HDF5: chunk B-tree too deep");
    const int rank = c->sp->rank;
    const uint8_t *p = at(f, node, 8 + 2 * (uint64_t)f->so);
    if (!p) return -1;
    if (memcmp(p, "TREE", 4) != 0 || p[4] != 1) return fail(f, "HDF5: no chunk B-tree node at address %llu", (unsigned long long)node);
    const unsigned level = p[5], n = (unsigned)rd(p + 6, 2);
    const uint64_t key = 8 + 8 * (uint64_t)(rank + 1), step = key + f->so;
    const uint8_t *b = at(f, node + 8 + 2 * (uint64_t)f->so, (uint64_t)n * step + key);
    if (!b) return -1;
    for (unsigned i = 0; i < n; i++, b += step) {
        const uint64_t child = rd_off(f, b + key);
        if (level > 0) {
            if (chunk_btree(c, child, guard + 1)) return -1;
            continue;
        }
        uint64_t off[JF_H5_MAXRANK];
        for (int d = 0; d < rank; d++) off[d] = rd(b + 8 + 8 * (uint64_t)d, 8);
        if (place_chunk(c, child, rd(b, 4), (uint32_t)rd(b + 4, 4), off)) return -1;
    }
    return 0;
}

/* chunk number k of the row-major chunk grid -> its first element */
static void chunk_offset(const chunk_ctx *c, uint64_t k, uint64_t *off) {
    for (int d = c->sp->rank - 1; d >= 0; d--) {
        const uint64_t n = (c->sp->dims[d] + c->cdim[d] - 1) / c->cdim[d];
        off[d] = (k % n) * c->cdim[d];
        k /= n;
    }
}

static int chunk_fixed_array(chunk_ctx *c, uint64_t addr, uint64_t n_chunks) {
    jf_h5 *f = c->f;
    const uint8_t *h = at(f, addr, 8 + (uint64_t)f->sl + f->so);
    if (!h) return -1;
    if (memcmp(h, "FAHD", 4) != 0 || h[4] != 0) return fail(f, "HDF5: no fixed array at address %llu", (unsigned long long)addr);
    const unsigned client = h[5], esz = h[6], page_bits = h[7];
    const uint64_t n = rd_len(f, h + 8), db = rd_off(f, h + 8 + f->sl);
    if (n < n_chunks) return fail(f, "HDF5: a chunk index shorter than the chunk grid");
    if (db == UNDEF) return 0; /* nothing written: fill values (zeros here) */
    const unsigned want = client ? (unsigned)f->so + 4 : (unsigned)f->so;
    if (client > 1 || esz < want || esz > want + 8 || page_bits > 30) return fail(f, "HDF5: fixed array with odd entries");
    const uint64_t page = (uint64_t)1 << page_bits;
    uint64_t o = db + 6 + (uint64_t)f->so;
    const uint8_t *d = at(f, db, 6 + (uint64_t)f->so);
    if (!d) return -1;
    if (memcmp(d, "FADB", 4) != 0) return fail(f, "HDF5: no fixed-array data block at address %llu", (unsigned long long)db);
    const int paged = n > page;
    const uint8_t *bitmap = NULL;
    if (paged) {
        const uint64_t n_pages = (n + page - 1) / page, bm = (n_pages + 7) / 8;
        bitmap = at(f, o, bm);
        if (!bitmap) return -1;
        o += bm + 4; /* the data block's own checksum lies in front of the pages */
    }
    for (uint64_t k = 0; k < n_chunks; k++) {
        uint64_t ea;
        if (paged) {
            const uint64_t pg = k / page;
            if (!(bitmap[pg / 8] & (0x80u >> (pg % 8)))) continue; /* page never written */
            ea = o + pg * (page * esz + 4) + (k % page) * esz;
        } else {
            ea = o + k * esz;
        }
        const uint8_t *e = at(f, ea, esz);
        if (!e) return -1;
        const uint64_t ca = rd_off(f, e);
        if (ca == UNDEF) continue;
        uint64_t off[JF_H5_MAXRANK];
        chunk_offset(c, k, off);
        const int nb = (int)esz - f->so - 4;
        if (place_chunk(c, ca, client ? rd(e + f->so, nb) : c->nominal, client ? (uint32_t)rd(e + f->so + nb, 4) : 0, off)) return -1;
    }
    return 0;
}

static double to_double(const uint8_t *p, const h5_type *t) {
    uint8_t b[8];
    for (uint32_t i = 0; i < t->size; i++) b[i] = t->big_endian ? p[t->size - 1 - i] : p[i];
    if (t->cls == 1) {
        if (t->size == 4) {
            float v;
            memcpy(&v, b, 4);
            return v;
        }
        double v;
        memcpy(&v, b, 8);
        return v;
    }
    uint64_t u = rd(b, (int)t->size);
    if (t->is_signed && t->size < 8 && (u >> (8 * t->size - 1))) u |= ~(uint64_t)0 << (8 * t->size);
    return t->is_signed ? (double)(int64_t)u : (double)u;
}

/* unfilter() and place_chunk() allocate a chunk's nominal size per stored chunk: a damaged chunk shape must not cost
 * gigabytes either (the bound of the dataset itself, below: deflate gains a factor of a thousand at the very most; chunks of
 * extendable datasets ARE wider than the dataset's current extent, so the extent is no bound) */
static int chunk_bound(jf_h5 *f, uint64_t nominal) {
    return nominal > 1024 * f->size + ((uint64_t)1 << 20) ? fail(f, "HDF5: chunks far larger than their file") : 0;
}

int jf_h5_read_f64(jf_h5 *f, uint64_t addr, int *rank, uint64_t dims[JF_H5_MAXRANK], double **data) {
    *data = NULL;
    h5_msgs *ms = (h5_msgs *)malloc(sizeof(h5_msgs));
    if (!ms) return fail(f, "out of memory");
    h5_type t;
    h5_space sp;
    h5_filters fl;
    uint8_t *raw = NULL;
    int rc = read_header(f, addr, ms);
    const h5_msg *mt = NULL, *msp = NULL, *ml = NULL, *mf = NULL;
    if (rc == 0) {
        mt = find_msg(ms, 0x03);
        msp = find_msg(ms, 0x01);
        ml = find_msg(ms, 0x08);
        mf = find_msg(ms, 0x0B);
        if (!mt || !msp || !ml) rc = fail(f, "HDF5: the object at address %llu is not a dataset", (unsigned long long)addr);
        else if ((mt->flags | msp->flags | ml->flags | (mf ? mf->flags : 0)) & 2) rc = fail(f, "HDF5: shared header messages are not supported");
    }
    if (rc == 0) rc = parse_type(f, mt->p, mt->size, &t);
    if (rc == 0 && !((t.cls == 0 || t.cls == 1) && (t.size == 1 || t.size == 2 || t.size == 4 || t.size == 8) && !(t.cls == 1 && t.size < 4)))
        rc = fail(f, "HDF5: a dataset of class %d, %u bytes per element, is not numeric", t.cls, t.size);
    if (rc == 0) rc = parse_space(f, msp->p, msp->size, &sp);
    memset(&fl, 0, sizeof fl);
    if (rc == 0 && mf) rc = parse_filters(f, mf->p, mf->size, &fl);
    /* (a damaged dataspace must not cost gigabytes: deflate gains a factor of a thousand at the very most) */
    if (rc == 0 && sp.n * t.size > 1024 * f->size + ((uint64_t)1 << 20)) rc = fail(f, "HDF5: a dataset far larger than its file");
    if (rc == 0) {
        raw = (uint8_t *)calloc(sp.n ? sp.n : 1, t.size);
        if (!raw) rc = fail(f, "out of memory");
    }
    const uint64_t bytes = rc == 0 ? sp.n * t.size : 0;
    if (rc == 0 && sp.n) {
        const uint8_t *p = ml->p;
        const uint64_t sz = ml->size;
        const int ver = sz ? p[0] : 0;
        /* versions 1 and 2 (files of HDF5 1.6.2 and older): dimensionality, class, five reserved bytes, the address (not
         * for compact data), the dimensions; version 3: class, then what the class needs */
        const int old = ver == 1 || ver == 2;
        const int cls = old ? (sz > 2 ? p[2] : -1) : (sz > 1 ? p[1] : -1);
        const int old_nd = old && sz > 1 ? p[1] : 0;
        const uint64_t old_dims = 8 + (cls != 0 ? (uint64_t)f->so : 0);
        if (sz < 2 || ver < 1 || ver > 4) {
            rc = fail(f, "HDF5: data layout message version %d", ver);
        } else if (old && (sz < old_dims + 4 * (uint64_t)old_nd || old_nd > JF_H5_MAXRANK + 1)) {
            rc = fail(f, "HDF5: short layout message");
        } else if (cls == 0) {
            const uint8_t *d = old ? p + old_dims + 4 * (uint64_t)old_nd + 4 : p + 4;
            const uint64_t have = old ? (sz >= old_dims + 4 * (uint64_t)old_nd + 4 ? rd(d - 4, 4) : 0) : (sz >= 4 ? rd(p + 2, 2) : 0);
            if ((uint64_t)(d - p) > sz || have > sz - (uint64_t)(d - p) || have < bytes) rc = fail(f, "HDF5: short compact dataset");
            else memcpy(raw, d, bytes);
        } else if (cls == 1) {
            if (!old && sz < 2 + (uint64_t)f->so + f->sl) rc = fail(f, "HDF5: short layout message");
            else {
                const uint64_t a = rd_off(f, old ? p + 8 : p + 2), l = old ? bytes : rd_len(f, p + 2 + f->so);
                if (a != UNDEF) { /* never written: fill values (zeros here) */
                    const uint8_t *src = l >= bytes ? at(f, a, bytes) : NULL;
                    if (l < bytes) rc = fail(f, "HDF5: contiguous storage shorter than the dataset");
                    else if (!src) rc = -1;
                    else memcpy(raw, src, bytes);
                }
            }
        } else if (cls == 2 && sp.rank >= 1) {
            chunk_ctx c;
            memset(&c, 0, sizeof c);
            c.f = f;
            c.sp = &sp;
            c.fl = &fl;
            c.elem = t.size;
            c.out = raw;
            uint64_t n_chunks = 1;
            if (ver <= 3) {
                const int nd = old ? old_nd : sz > 2 ? p[2] : 0;
                const uint8_t *pa = old ? p + 8 : p + 3, *pd = pa + f->so;
                if (nd != sp.rank + 1 || (uint64_t)(pd - p) + 4 * (uint64_t)nd > sz) rc = fail(f, "HDF5: chunked layout of %d dimensions for rank %d", nd, sp.rank);
                c.nominal = t.size;
                for (int d = 0; rc == 0 && d < sp.rank; d++) {
                    c.cdim[d] = rd(pd + 4 * (uint64_t)d, 4);
                    if (c.cdim[d] == 0 || c.nominal > MAX_FILE_BYTES / c.cdim[d]) rc = fail(f, "HDF5: odd chunk dimensions");
                    else c.nominal *= c.cdim[d];
                }
                if (rc == 0) rc = chunk_bound(f, c.nominal);
                if (rc == 0) {
                    const uint64_t bt = rd_off(f, pa);
                    if (bt != UNDEF) rc = chunk_btree(&c, bt, 0);
                }
            } else {
                const unsigned lf = sz > 2 ? p[2] : 0;
                const int nd = sz > 3 ? p[3] : 0, eb = sz > 4 ? p[4] : 0;
                if (nd != sp.rank + 1 || eb < 1 || eb > 8 || sz < 5 + (uint64_t)eb * nd + 1) rc = fail(f, "HDF5: chunked layout of %d dimensions for rank %d", nd, sp.rank);
                else if (lf & 1) rc = fail(f, "HDF5: partial edge chunks stored unfiltered are not supported");
                c.nominal = t.size;
                for (int d = 0; rc == 0 && d < sp.rank; d++) {
                    c.cdim[d] = rd(p + 5 + (uint64_t)eb * d, eb);
                    if (c.cdim[d] == 0 || c.nominal > MAX_FILE_BYTES / c.cdim[d]) rc = fail(f, "HDF5: odd chunk dimensions");
                    else {
                        c.nominal *= c.cdim[d];
                        n_chunks *= (sp.dims[d] + c.cdim[d] - 1) / c.cdim[d];
                    }
                }
                if (rc == 0) rc = chunk_bound(f, c.nominal);
                if (rc == 0) {
                    const uint8_t *q = p + 5 + (uint64_t)eb * nd;
                    uint64_t left = sz - (5 + (uint64_t)eb * nd);
                    const unsigned index = q[0];
                    q++, left--;
                    uint64_t off[JF_H5_MAXRANK] = {0};
                    if (index == 1) {
                        uint64_t stored = c.nominal;
                        uint32_t mask = 0;
                        if (lf & 2) {
                            if (left < (uint64_t)f->sl + 4) rc = fail(f, "HDF5: short layout message");
                            else {
                                stored = rd_len(f, q);
                                mask = (uint32_t)rd(q + f->sl, 4);
                                q += f->sl + 4, left -= (uint64_t)f->sl + 4;
                            }
                        } else {
                            memset(&fl, 0, sizeof fl); /* (a single chunk without the flag is stored unfiltered) */
                        }
                        if (rc == 0 && left < (uint64_t)f->so) rc = fail(f, "HDF5: short layout message");
                        if (rc == 0 && rd_off(f, q) != UNDEF) rc = place_chunk(&c, rd_off(f, q), stored, mask, off);
                    } else if (index == 2) {
                        if (left < (uint64_t)f->so) rc = fail(f, "HDF5: short layout message");
                        const uint64_t a = rc == 0 ? rd_off(f, q) : UNDEF;
                        for (uint64_t k = 0; rc == 0 && a != UNDEF && k < n_chunks; k++) {
                            chunk_offset(&c, k, off);
                            rc = place_chunk(&c, a + k * c.nominal, c.nominal, 0, off);
                        }
                    } else if (index == 3) {
                        if (left < 1 + (uint64_t)f->so) rc = fail(f, "HDF5: short layout message");
                        else if (rd_off(f, q + 1) != UNDEF) rc = chunk_fixed_array(&c, rd_off(f, q + 1), n_chunks);
                    } else {
                        rc = fail(f, "HDF5: chunk index type %u (a dataset with unlimited dimensions written with libver=latest) is not supported", index);
                    }
                }
            }
        } else {
            rc = fail(f, "HDF5: data layout class %d is not supported", cls);
        }
    }
    if (rc == 0) {
        double *out = (double *)calloc((size_t)(sp.n ? sp.n : 1), sizeof(double));
        if (!out) rc = fail(f, "out of memory");
        else {
            for (uint64_t i = 0; i < sp.n; i++) out[i] = to_double(raw + i * t.size, &t);
            *data = out;
            *rank = sp.rank;
            for (int d = 0; d < sp.rank; d++) dims[d] = sp.dims[d];
        }
    }
    free(raw);
    free(ms);
    return rc;
}

int jf_h5_is_dataset(jf_h5 *f, uint64_t addr) {
    h5_msgs *ms = (h5_msgs *)malloc(sizeof(h5_msgs));
    if (!ms) return fail(f, "out of memory");
    int rc = read_header(f, addr, ms);
    if (rc == 0) rc = find_msg(ms, 0x08) != NULL;
    free(ms);
    return rc;
}

/* ---- attributes --------------------------------------------------------------------------------------------------------- */

typedef struct {
    const char *name;
    char *out;
    size_t cap;
    int found;
    const h5_fheap *heap;
} attr_ctx;

/* an attribute message; ctx->found = 1 when it is the string attribute asked for */
static int parse_attr(jf_h5 *f, const uint8_t *p, uint64_t size, attr_ctx *ctx) {
    if (size < 8) return fail(f, "HDF5: short attribute message");
    const int ver = p[0];
    if (ver < 1 || ver > 3) return fail(f, "HDF5: attribute message version %d", ver);
    const uint64_t nsz = rd(p + 2, 2), tsz = rd(p + 4, 2), ssz = rd(p + 6, 2);
    uint64_t o = ver == 3 ? 9 : 8;
    const uint64_t pad = ver == 1 ? 7 : 0;
    const uint64_t n_at = o, t_at = n_at + ((nsz + pad) & ~pad), s_at = t_at + ((tsz + pad) & ~pad), d_at = s_at + ((ssz + pad) & ~pad);
    if (d_at > size || nsz == 0) return fail(f, "HDF5: an attribute's parts run past its message");
    if (strnlen((const char *)p + n_at, nsz) != strlen(ctx->name) || memcmp(p + n_at, ctx->name, strlen(ctx->name)) != 0) return 0;
    if (ver > 1 && (p[1] & 3)) return 0; /* shared datatype or dataspace: not read */
    h5_type t;
    h5_space sp;
    if (parse_type(f, p + t_at, tsz, &t) || parse_space(f, p + s_at, ssz, &sp)) return -1;
    if (sp.n == 0 || ctx->cap == 0) return 0;
    const uint8_t *d = p + d_at;
    const uint64_t left = size - d_at;
    uint64_t len = 0;
    const uint8_t *s = NULL;
    if (t.cls == 3) {
        if (left < t.size) return fail(f, "HDF5: an attribute's value runs past its message");
        s = d;
        len = t.size;
    } else if (t.cls == 9 && t.vlen_string) {
        if (left < 4 + (uint64_t)f->so + 4) return fail(f, "HDF5: an attribute's value runs past its message");
        len = rd(d, 4);
        const uint64_t col = rd_off(f, d + 4), index = rd(d + 4 + f->so, 4);
        const uint8_t *g = at(f, col, 8 + (uint64_t)f->sl);
        if (!g) return -1;
        if (memcmp(g, "GCOL", 4) != 0) return fail(f, "HDF5: no global heap collection at address %llu", (unsigned long long)col);
        const uint64_t csz = rd_len(f, g + 8);
        g = at(f, col, csz);
        if (!g) return -1;
        uint64_t q = 8 + (uint64_t)f->sl;
        while (q + 8 + (uint64_t)f->sl <= csz) {
            const uint64_t idx = rd(g + q, 2), osz = rd_len(f, g + q + 8);
            if (idx == 0) break;
            if (osz > csz - q - 8 - (uint64_t)f->sl) return fail(f, "HDF5: a global heap object runs past its collection");
            if (idx == index) {
                s = g + q + 8 + f->sl;
                if (len > osz) len = osz;
                break;
            }
            q += 8 + (uint64_t)f->sl + ((osz + 7) & ~(uint64_t)7);
        }
        if (!s) return fail(f, "HDF5: a string's global heap object is missing");
    } else {
        return 0;
    }
    uint64_t n = strnlen((const char *)s, len);
    while (n > 0 && s[n - 1] == ' ') n--; /* space-padded strings */
    if (n >= ctx->cap) n = ctx->cap - 1;
    memcpy(ctx->out, s, n);
    ctx->out[n] = 0;
    ctx->found = 1;
    return 0;
}

static int dense_attr_rec(jf_h5 *f, const uint8_t *rec, void *vctx) {
    attr_ctx *c = (attr_ctx *)vctx;
    if (c->found) return 0;
    if (((rec[0] >> 4) & 3) != 0) return 0; /* huge or tiny objects: attributes this reader does not need */
    uint64_t len = 0;
    const uint8_t *obj = fheap_object(f, c->heap, rec, &len);
    return obj ? parse_attr(f, obj, len, c) : -1;
}

int jf_h5_attr_str(jf_h5 *f, uint64_t addr, const char *name, char *out, size_t cap) {
    h5_msgs *ms = (h5_msgs *)malloc(sizeof(h5_msgs));
    if (!ms) return fail(f, "out of memory");
    attr_ctx ctx = {name, out, cap, 0, NULL};
    int rc = read_header(f, addr, ms);
    for (int i = 0; rc == 0 && !ctx.found && i < ms->n; i++) {
        const h5_msg *m = &ms->m[i];
        if (m->type == 0x0C && !(m->flags & 2)) {
            rc = parse_attr(f, m->p, m->size, &ctx);
        } else if (m->type == 0x15) {
            if (m->size < 2 || m->p[0] != 0) {
                rc = fail(f, "HDF5: attribute info message version %d", m->size ? m->p[0] : -1);
                break;
            }
            const uint64_t o = 2 + ((m->p[1] & 1) ? 2 : 0);
            if (m->size < o + 2 * (uint64_t)f->so) {
                rc = fail(f, "HDF5: short attribute info message");
                break;
            }
            const uint64_t heap = rd_off(f, m->p + o), bt = rd_off(f, m->p + o + f->so);
            if (heap == UNDEF || bt == UNDEF) continue;
            h5_fheap fh;
            ctx.heap = &fh;
            rc = fheap_open(f, heap, &fh);
            if (rc == 0) rc = bt2_walk(f, bt, 8, (unsigned)fh.id_len + 9, dense_attr_rec, &ctx); /* heap ID, flags, creation order, hash */
        }
    }
    free(ms);
    return rc ? rc : ctx.found ? 0 : 1;
}

/* ---- files -------------------------------------------------------------------------------------------------------------- */

static const uint8_t kSignature[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};

int jf_h5_open(const char *path, jf_h5 **out, char *err, size_t errlen) {
    *out = NULL;
    jf_h5 *f = (jf_h5 *)calloc(1, sizeof(jf_h5));
    FILE *fp = path ? fopen(path, "rb") : NULL;
    int rc = 0;
    if (!f) {
        if (fp) fclose(fp);
        snprintf(err, errlen, "out of memory");
        return -1;
    }
    if (!fp) rc = fail(f, "cannot open %s", path ? path : "(null)");
    if (rc == 0) {
        long n = -1;
        if (fseek(fp, 0, SEEK_END) == 0) n = ftell(fp);
        if (n < 0 || fseek(fp, 0, SEEK_SET) != 0) rc = fail(f, "cannot read %s", path);
        else if ((uint64_t)n > MAX_FILE_BYTES) rc = fail(f, "%s: larger than 4 GiB", path);
        else {
            f->size = (uint64_t)n;
            f->buf = (uint8_t *)malloc(f->size ? f->size : 1);
            if (!f->buf) rc = fail(f, "out of memory");
            else if (fread(f->buf, 1, f->size, fp) != f->size) rc = fail(f, "cannot read %s", path);
        }
    }
    if (fp) fclose(fp);
    if (rc == 0) {
        uint64_t b = 0;
        int found = 0;
        for (;;) {
            if (b + 8 <= f->size && memcmp(f->buf + b, kSignature, 8) == 0) {
                found = 1;
                break;
            }
            b = b ? 2 * b : 512;
            if (b >= f->size) break;
        }
        if (!found) rc = fail(f, "%s: not an HDF5 file (no signature)", path);
        f->base = b;
    }
    if (rc == 0) {
        f->so = f->sl = 8; /* for at() until the superblock says */
        const uint8_t *p = at(f, 0, 16);
        if (!p) rc = -1;
        else {
            const int ver = p[8];
            if (ver == 0 || ver == 1) {
                const int so = p[13], sl = p[14];
                const uint64_t o = ver == 0 ? 24 : 28;
                if (so < 2 || so > 8 || sl < 2 || sl > 8) rc = fail(f, "HDF5: offsets of %d and lengths of %d bytes", so, sl);
                else {
                    f->so = so, f->sl = sl;
                    const uint8_t *q = at(f, o, 4 * (uint64_t)so + 2 * (uint64_t)so);
                    if (!q) rc = -1;
                    else {
                        /* base address, free-space info, end of file, driver info; then the root group's symbol table entry */
                        const uint64_t base = rd_off(f, q);
                        f->root = rd_off(f, q + 4 * so + so);
                        if (base != 0 && base != UNDEF && base != f->base) rc = fail(f, "HDF5: a base address that is not the superblock's");
                    }
                }
            } else if (ver == 2 || ver == 3) {
                const int so = p[9], sl = p[10];
                if (so < 2 || so > 8 || sl < 2 || sl > 8) rc = fail(f, "HDF5: offsets of %d and lengths of %d bytes", so, sl);
                else {
                    f->so = so, f->sl = sl;
                    const uint8_t *q = at(f, 12, 4 * (uint64_t)so);
                    if (!q) rc = -1;
                    else f->root = rd_off(f, q + 3 * so);
                }
            } else {
                rc = fail(f, "HDF5: superblock version %d", ver);
            }
        }
    }
    if (rc) {
        snprintf(err, errlen, "%s", f->err);
        free(f->buf);
        free(f);
        return -1;
    }
    *out = f;
    return 0;
}

void jf_h5_close(jf_h5 *f) {
    if (!f) return;
    free(f->buf);
    free(f);
}

const char *jf_h5_error(const jf_h5 *f) { return f ? f->err : ""; }
uint64_t jf_h5_root(const jf_h5 *f) { return f->root; }

int jf_h5_lookup(jf_h5 *f, const char *path, uint64_t *addr) {
    uint64_t cur = f->root;
    const char *p = path;
    while (*p == '/') p++;
    while (*p) {
        const char *e = strchr(p, '/');
        const size_t n = e ? (size_t)(e - p) : strlen(p);
        h5_links ls = {NULL, 0, 0};
        if (group_links(f, cur, &ls)) {
            free_links(&ls);
            return -1;
        }
        int hit = 0;
        for (int i = 0; i < ls.n; i++)
            if (strlen(ls.l[i].name) == n && memcmp(ls.l[i].name, p, n) == 0) {
                cur = ls.l[i].addr;
                hit = 1;
                break;
            }
        free_links(&ls);
        if (!hit) return 1;
        p += n;
        while (*p == '/') p++;
    }
    *addr = cur;
    return 0;
}

int jf_h5_list(jf_h5 *f, uint64_t group, char **names) {
    *names = NULL;
    h5_links ls = {NULL, 0, 0};
    if (group_links(f, group, &ls)) {
        free_links(&ls);
        return -1;
    }
    size_t total = 1;
    for (int i = 0; i < ls.n; i++) total += strlen(ls.l[i].name) + 1;
    char *s = (char *)malloc(total);
    if (!s) {
        free_links(&ls);
        return fail(f, "out of memory");
    }
    size_t o = 0;
    for (int i = 0; i < ls.n; i++) {
        const size_t n = strlen(ls.l[i].name);
        memcpy(s + o, ls.l[i].name, n);
        o += n;
        s[o++] = '\n';
    }
    s[o] = 0;
    free_links(&ls);
    *names = s;
    return 0;
}
