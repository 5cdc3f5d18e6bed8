/*
 * jf_hdf5.h -- read-only access to the part of the HDF5 file format that SOFA files (netCDF-4 containers: AES69) are made of.
 * Own code written from the HDF5 File Format Specification (version 3.0); no HDF5 library is linked.  See jf_hdf5.c for what
 * is understood and what is refused.  Internal to the library: the C ABI is jf_sofa_* in include/jefferson.h.
 */
#ifndef JF_HDF5_H
#define JF_HDF5_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JF_H5_MAXRANK 8

typedef struct jf_h5 jf_h5;

/* 0, or -1 with a text in err */
int jf_h5_open(const char *path, jf_h5 **out, char *err, size_t errlen);
void jf_h5_close(jf_h5 *f);
const char *jf_h5_error(const jf_h5 *f);

/* "/Data.IR", "Data.IR", "group/dataset": hard links from the root group.  0 found, 1 no such object, -1 error. */
int jf_h5_lookup(jf_h5 *f, const char *path, uint64_t *addr);
/* names of a group's links, '\n'-separated, malloc'd (free()); addr = the group's object header (root: jf_h5_root) */
uint64_t jf_h5_root(const jf_h5 *f);
int jf_h5_list(jf_h5 *f, uint64_t group, char **names);
/* 1 if the object is a dataset (has a layout message), 0 if not, -1 error */
int jf_h5_is_dataset(jf_h5 *f, uint64_t addr);

/* A dataset of integers or IEEE floats of 1, 2, 4 or 8 bytes, either byte order, as doubles (malloc'd, row-major;
 * free()).  rank 0 = a scalar (one element).  0, or -1 with a text in jf_h5_error. */
int jf_h5_read_f64(jf_h5 *f, uint64_t addr, int *rank, uint64_t dims[JF_H5_MAXRANK], double **data);

/* A string attribute (fixed-length, or variable-length in the global heap; the first element of an array of them),
 * without its padding.  0 found, 1 no such attribute or not a string, -1 error. */
int jf_h5_attr_str(jf_h5 *f, uint64_t addr, const char *name, char *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
