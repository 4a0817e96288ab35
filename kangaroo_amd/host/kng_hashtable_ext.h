// kng_hashtable_ext.h -- what HashTable_kng.cpp offers BESIDE the reference's `class HashTable` (HashTable.h:66-108).
//
// HashTable_kng.cpp is a link-time replacement for the reference's HashTable.o: it is compiled against the reference's own
// HashTable.h and defines every member of the class with the reference's behaviour (statuses, bucket order, kDist / kType,
// SaveTable / LoadTable bytes, the public E[] array).  A host that only knows HashTable.h gets a table whose insertion cost
// no longer grows with the table.  A host that also includes this header (SolveKeyGPU_kng.cpp, the link-time replacement for
// Kangaroo::SolveKeyGPU, Kangaroo.cpp:510-644) can hand over a whole launch's distinguished points in one call, from several
// threads at once, without the program's global ghMutex (Kangaroo.cpp:594-612).
#ifndef KNG_HASHTABLE_EXT_H
#define KNG_HASHTABLE_EXT_H

#include <stdint.h>

#include "kangaroo_hip.h" /* kng_dp_record: the 64-byte record the engine writes */

class HashTable;

/* One insertion that did not end as ADD_OK.  `index` = position of the point in the batch; `status` = ADD_DUPLICATE or
 * ADD_COLLISION (HashTable.h:33-35); for a collision `stored_d` is the d word of the entry ALREADY in the table, i.e. what
 * HashTable::Add would have decoded into kDist / kType (HashTable.cpp:286-288). */
typedef struct {
    uint32_t index;
    uint32_t status;
    uint64_t stored_d[2];
} kng_ht_event;

typedef struct {
    uint64_t entries;      /* GetNbItem() */
    uint64_t bytes_mapped; /* taken from the OS by the arenas */
    uint64_t bytes_touched;
    uint64_t merges;       /* (rounds 4-5: tail runs folded into their bucket's main run; always 0 since round 6) */
    uint64_t grows;        /* buckets re-split into four times as many runs */
    uint64_t bytes_recycled; /* (rounds 4-5; always 0 since round 6: freed runs go to the arena's free lists) */
    uint64_t lock_spins;   /* failed attempts on a stripe lock (contention between ingesting threads) */
} kng_ht_stats_t;

extern "C" {
/* Insert n engine records (kng_drain_view layout: x[4], DEVICE distance d[2], kidx).  The call does what
 * GPUEngine::Launch + HashTable::Convert + HashTable::Add do per point in the reference (GPUEngine.cu:668-674,
 * HashTable.cpp:85-113,262-307): odd kidx = wild, its true distance is d - wild_off mod n (|true distance| < 2^127), sign and
 * type go to bits 127 / 126 of the stored distance, bucket = x[2] & 0x3FFFF, key = x[0..1].  Thread-safe against itself and
 * against HashTable::Add on the same object (1024 bucket stripes, one spin lock each); it never touches kDist / kType.
 * Events beyond ev_cap are counted in *n_ev but not stored (the caller sizes ev for the whole batch to lose none).
 * Returns 0. */
int kng_ht_ingest(HashTable *ht, const kng_dp_record *recs, uint32_t n, const uint64_t wild_off[2], kng_ht_event *ev,
                  uint32_t ev_cap, uint32_t *n_ev);
/* The public array of the class, on request: E[h].items[0 .. nbItem) = pointers to the bucket's entries in ascending x for all
 * h, as the reference keeps it after every Add.  The entries live in sorted runs (kng_bucket.h), not behind a pointer array;
 * LoadTable builds these views for the reference's only readers of items[] (Check.cpp:47,88), this call builds them at any
 * other time, and the next insertion drops them all (items = NULL).  kng_ht_set_tail: rounds 4-5 only, does nothing. */
void kng_ht_normalize(HashTable *ht);
void kng_ht_set_tail(HashTable *ht, uint32_t tail);
/* The class has no destructor (HashTable.h:66-108): a host that creates and deletes tables calls this before `delete` -- the
 * table's memory and its slot in the registry (1024 slots) go back.  Not needed by the program, which has one table. */
void kng_ht_release(HashTable *ht);
void kng_ht_stats(HashTable *ht, kng_ht_stats_t *out);
}
#endif
