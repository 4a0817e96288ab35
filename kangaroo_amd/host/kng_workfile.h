/*
 * kng_workfile.h -- work files in the reference's format (libkangaroo_host.so).
 *
 * SURVEY 8(f) row 3 / Appendix B.  Byte-compatible with Backup.cpp:368-407 (SaveHeader/SaveWork),
 * :497-552 (kangaroo section) and :139-231 (ReadHeader/LoadWork/FetchWalks): a file written here loads in
 * the reference program (-i, -winfo, -wcheck, -wm) and vice versa.
 *
 *   HEADW  u32 0xFA6A8001 | u32 version 0 | u32 dpSize | 32 B rangeStart | 32 B rangeEnd | 32 B key.x |
 *          32 B key.y | u64 totalCount | f64 totalTime | hash table (kng_dptable.h) | u64 nbKangaroo |
 *          nbKangaroo x { 32 B x, 32 B y, 32 B d }      d = TRUE distance mod n
 *   HEADK  u32 0xFA6A8002 | u32 version 0 | u64 nbKangaroo | nbKangaroo x 96 B   (kangaroo-only file)
 *
 * The reference writes the kangaroo section with three 32-byte fwrite calls per kangaroo
 * (Backup.cpp:532-534); here records are assembled in a large buffer and written in one call per chunk,
 * so a herd can be streamed straight from kng_get_kangaroos_range (6.4 GB per save in SURVEY config 5).
 */
#ifndef KNG_WORKFILE_H
#define KNG_WORKFILE_H

#include <stdint.h>

#include "kng_dptable.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KNGW_HEADW 0xFA6A8001u
#define KNGW_HEADK 0xFA6A8002u

typedef struct kngw_header {
    uint32_t magic;   /* KNGW_HEADW or KNGW_HEADK */
    uint32_t version; /* 0 */
    uint32_t dp_size; /* HEADW only, like everything below */
    uint32_t reserved;
    uint64_t range_start[4], range_end[4], key_x[4], key_y[4];
    uint64_t total_count; /* jumps performed so far */
    double total_seconds;
} kngw_header;

typedef struct kngw_file kngw_file;

/* ---- writer.  table may be NULL only for HEADK.  n_kangaroos = size of the kangaroo section that
 * kngw_put_kangaroos will fill (0: none, like -ws off).  Returns NULL on error (kngw_last_error). */
kngw_file *kngw_create(const char *path, const kngw_header *h, const kngt_table *table, uint64_t n_kangaroos);
/* append n kangaroos: x, y, d_true are n x 4 limbs each */
int kngw_put_kangaroos(kngw_file *f, const uint64_t *x, const uint64_t *y, const uint64_t *d_true, uint64_t n);
/* the same from / into n records of the file's own layout -- 96 bytes each: x[4], y[4], d_true[4] -- as the engine's
 * kng_snapshot_read delivers and kng_snapshot_write takes them (no per-kangaroo copy on the host) */
int kngw_put_records(kngw_file *f, const void *records, uint64_t n);
int kngw_get_records(kngw_file *f, void *records, uint64_t n);

/* ---- reader.  Fills *h; loads the hash table into table when the file has one and table != NULL
 * (skips it otherwise); *n_kangaroos = size of the kangaroo section. */
kngw_file *kngw_open(const char *path, kngw_header *h, kngt_table *table, uint64_t *n_kangaroos);
/* read the next n kangaroos */
int kngw_get_kangaroos(kngw_file *f, uint64_t *x, uint64_t *y, uint64_t *d_true, uint64_t n);

/* flushes and closes; for a writer also checks that exactly n_kangaroos were written.  0 or -1. */
int kngw_close(kngw_file *f);
const char *kngw_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
