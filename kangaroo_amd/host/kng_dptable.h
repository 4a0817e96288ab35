/*
 * kng_dptable.h -- distinguished-point table of the MI355X kangaroo host pipeline (libkangaroo_host.so).
 *
 * the consumer side of the DP drain.  Same observable behaviour and the same
 * serialised form as the reference's HashTable (HashTable.h:27-56, HashTable.cpp:75-100,262-307,375-396)
 * so that work files stay interchangeable with the reference program:
 *   - 2^18 buckets, bucket = x.limb2 & 0x3FFFF;
 *   - an entry is 32 bytes: x limbs 0-1, then |d| (126 bits) with bit 127 = sign, bit 126 = kangaroo type;
 *   - buckets are kept sorted by (x.limb1, x.limb0); adding an x that is already present returns
 *     DUPLICATE when the 128-bit d word is equal too, else COLLISION (the new entry is NOT stored);
 *   - the per-bucket "maxItem" word of the file follows the reference's allocation rule (16, then +4
 *     whenever nbItem >= maxItem-1 at the start of an Add) so a table fed with the same sequence
 *     serialises to the same bytes.
 * What is different is the storage: entries live contiguously (no malloc per entry, no pointer array),
 * a bucket that grows is split into 2^k sorted runs by the top bits of x.limb1 so that an insertion moves
 * ~0.5 KB at any table size (kng_dptable.cpp), and nothing in here is global -- buckets are independent,
 * so several consumer threads may add concurrently as long as each bucket is only touched by one of them
 * (the solver partitions buckets by index).
 * Plain C ABI, little-endian uint64 limbs.
 */
#ifndef KNG_DPTABLE_H
#define KNG_DPTABLE_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNGT_HASH_BITS 18
#define KNGT_BUCKETS (1u << KNGT_HASH_BITS)

#define KNGT_ADD_OK 0
#define KNGT_ADD_DUPLICATE 1
#define KNGT_ADD_COLLISION 2

typedef struct kngt_entry { /* the 32 bytes of the file format */
    uint64_t x[2];
    uint64_t d[2];
} kngt_entry;

typedef struct kngt_table kngt_table;

kngt_table *kngt_create(void);
void kngt_destroy(kngt_table *t);
void kngt_reset(kngt_table *t);

/* HashTable::Convert (HashTable.cpp:75-100): position + true distance mod n + type -> bucket and entry */
void kngt_encode(const uint64_t x[4], const uint64_t d_true[4], uint32_t type, uint32_t *bucket, kngt_entry *e);
/* the same straight from an engine record: device distance (128 bits; wild = odd kidx still carries +wild_offset,
 * GPUEngine.cu:406-411,672) -> entry, without the detour through a 256-bit distance mod n.  Equal to
 * kngt_encode(x, (d_dev - wild_offset) mod n or d_dev, kidx & 1) for every d_dev, wild_offset < 2^128. */
void kngt_encode_device(const uint64_t x[4], const uint64_t d_dev[2], const uint64_t wild_offset[2], uint64_t kidx,
                        uint32_t *bucket, kngt_entry *e);
/* HashTable::CalcDistAndType (HashTable.cpp:246-260): entry d word -> true distance mod n and type */
void kngt_decode(const uint64_t d_word[2], uint64_t d_true[4], uint32_t *type);

/* HashTable::Add(h, e) (HashTable.cpp:262-307).  On COLLISION *other receives the stored entry. */
int kngt_add_entry(kngt_table *t, uint32_t bucket, const kngt_entry *e, kngt_entry *other);
/* HashTable::Add(x, d, type): encode + add; on COLLISION other_d/other_type describe the stored kangaroo */
int kngt_add(kngt_table *t, const uint64_t x[4], const uint64_t d_true[4], uint32_t type, uint64_t other_d[4],
             uint32_t *other_type);

/* hint for batched adds: stage 0 touches the bucket header, 1 the run the key falls into, 2 the middle of that
 * run (call them a few entries ahead of kngt_add_entry; they read only what stage-1 callers already published) */
void kngt_prefetch(const kngt_table *t, uint32_t bucket, uint64_t x_limb1, int stage);

uint64_t kngt_count(const kngt_table *t);
/* bytes held by the table (entries, run headers, the bucket array) */
uint64_t kngt_memory_bytes(const kngt_table *t);
uint32_t kngt_bucket_count(const kngt_table *t, uint32_t bucket);
/* copies up to cap entries of a bucket (sorted order); returns the number copied */
uint32_t kngt_bucket_entries(const kngt_table *t, uint32_t bucket, kngt_entry *out, uint32_t cap);

/* HashTable::SaveTable / LoadTable (HashTable.cpp:375-396,436-461): per bucket u32 nbItem, u32 maxItem,
 * nbItem x 32 bytes.  Return 0, or -1 on a short read/write; kngt_read also rejects a bucket that announces more
 * entries than the file has bytes left or whose entries are not strictly ascending (a corrupt file must not hang
 * or poison the binary search). */
int kngt_write(const kngt_table *t, FILE *f);
int kngt_read(kngt_table *t, FILE *f);
/* bytes kngt_write will produce */
uint64_t kngt_serialised_size(const kngt_table *t);

#ifdef __cplusplus
}
#endif
#endif
