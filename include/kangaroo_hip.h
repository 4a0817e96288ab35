/*
 * kangaroo_hip.h -- C ABI of libkangaroo_hip.so, the MI355X (gfx950) kangaroo jump engine.
 *
 * This is the drop-in boundary for ONE hot path of JeanLucPons/Kangaroo: the per-herd random
 * walk that the reference implements in GPU/GPUEngine.cu + GPU/GPUCompute.h + GPU/GPUMath.h
 * behind `class GPUEngine` (GPU/GPUEngine.h:40-84).  Every entry point below names the
 * reference interface it replaces (file:line relative to the reference root).  The C++ class
 * `GPUEngine` with the reference's exact public surface is re-created over this ABI in
 * kangaroo_amd/host/GPUEngine.{h,cpp}; INTEGRATION.md shows how the unmodified reference
 * host code links against it.
 *
 * Conventions
 *   - plain pointers and sizes only; all big integers are little-endian uint64_t limbs:
 *     field elements / x / y = 4 limbs, device distances = 2 limbs (128 bit, GPUMath.h:119-121).
 *   - kangaroo index kIdx = position in the arrays given to kng_set_kangaroos
 *     (= ITEM.kIdx of GPUEngine.h:34-38; type = kIdx & 1, GPUEngine.cu:409).
 *   - distances at this level are DEVICE distances: the caller has already added the wild
 *     offset mod n for odd kIdx (GPUEngine.cu:406-411) and removes it again from what
 *     kng_drain / kng_get_kangaroos return (GPUEngine.cu:477,672).  The C++ class does that.
 *   - every function returns KNG_OK (0) or a negative KNG_E* code; kng_last_error() gives text.
 *     Nothing falls back to the CPU: without a usable gfx950 device kng_create fails.
 *   - one host thread per engine handle; distinct handles may be driven concurrently
 *     (one per GPU, Kangaroo.cpp:1041-1047).  Every entry point selects its own device.
 */
#ifndef KANGAROO_HIP_H
#define KANGAROO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNG_NB_JUMP 32   /* Constants.h:29  NB_JUMP                                    */
#define KNG_NB_RUN 64    /* Constants.h:35  NB_RUN: jumps per kangaroo per launch       */
#define KNG_GRP_SIZE 128 /* Constants.h:32  GPU_GRP_SIZE: herd = gridX*gridY*128         */

#define KNG_OK 0
#define KNG_E_NODEVICE (-1) /* no HIP device / bad device id (GPUEngine.cu:152-169)      */
#define KNG_E_ALLOC (-2)    /* device or pinned allocation failed (GPUEngine.cu:201-232) */
#define KNG_E_ARG (-3)      /* bad argument                                              */
#define KNG_E_STATE (-4)    /* call sequence error (e.g. launch before set_params)       */
#define KNG_E_HIP (-5)      /* HIP runtime error, see kng_last_error()                   */

typedef struct kng_engine kng_engine; /* opaque; replaces the private state of GPUEngine.h:66-82 */

/* One distinguished point; replaces the 56-byte device record of GPUMath.h:173-188 and the
 * host ITEM of GPUEngine.h:34-38 (d is still the DEVICE distance here). */
typedef struct {
    uint64_t x[4];
    uint64_t d[2];
    uint64_t kidx;
} kng_item;

/* ---- device discovery: GPUEngine.cu:280-308 (GetGridSize), :329-375 (PrintCudaInfo) -------- */
int kng_device_count(void);
/* name (<= name_cap bytes incl. NUL), compute units, total memory, gcn arch string */
int kng_device_info(int dev, char *name, size_t name_cap, int *cu_count, uint64_t *mem_bytes,
                    char *arch, size_t arch_cap);
/* NUMA node of the host the device hangs off (sysfs numa_node of its PCI address), or -1 when unknown: a host that drives
 * several GPUs (Kangaroo.cpp:1041-1047, one thread per GPU) keeps each GPU's thread and its pinned DP buffers on that node */
int kng_device_numa_node(int dev);
/* the two halves of that, usable on their own: the device's PCI address ("0000:c3:00.0", lower case) -- what ties a HIP
 * device index to the same card in rocm_smi / amd-smi, whose indices need not agree with HIP's under HIP_VISIBLE_DEVICES --
 * and the sysfs lookup of an address (no device needed; KNG_SYSFS_ROOT replaces "/sys") */
int kng_device_pci_bdf(int dev, char *bdf, size_t cap);
int kng_numa_node_of_bdf(const char *bdf);
/* free / total device memory right now (hipMemGetInfo): lets a host that re-creates its engine once per key
 * (Kangaroo.cpp:1021-1075, ctor :523, `delete gpu` :634) check that a create / destroy cycle gives everything back */
int kng_device_free_bytes(int dev, uint64_t *free_bytes, uint64_t *total_bytes);
/* fills *x / *y when <= 0 with the reference's defaults: x = 2*CU count, y = 128
 * (GPUEngine.cu:299-303; the per-SM core table has no AMD entry, so y falls back to 128) */
int kng_default_grid(int dev, int *x, int *y);

/* ---- lifetime: GPUEngine ctor/dtor, GPUEngine.cu:144-263 ------------------------------------ */
/* herd size = grid_x*grid_y*128 kangaroos; max_found = DP capacity per launch (Kangaroo.cpp:523) */
int kng_create(int dev, int grid_x, int grid_y, uint32_t max_found, kng_engine **out);
void kng_destroy(kng_engine *h); /* safe while a launch is in flight (Kangaroo.cpp:572-634) */

/* Raise the DP capacity per launch to at least `points` (never lowers it).  The reference fixes max_found in the program
 * (65536*2, Kangaroo.cpp:523) for a V100-sized herd; at the MI355X default grid and the DP size the program suggests for
 * eight GPUs one launch yields 262 144 points, of which that constant would drop half (GPUEngine.cu:641-648).  The class
 * shim calls this from SetParams with twice the expected yield of herd and mask.  Only between launches (nothing
 * outstanding, nothing undrained): KNG_E_STATE otherwise, capacity unchanged.  "max_found" reads the capacity back. */
int kng_reserve_points(kng_engine *h, uint32_t points);

uint64_t kng_nb_kangaroos(const kng_engine *h); /* GetNbThread()*GetGroupSize(), GPUEngine.cu:377 */
uint64_t kng_memory_bytes(const kng_engine *h); /* GetMemory(), GPUEngine.cu:266-268 (64-bit)  */

/* ---- parameters: SetParams, GPUEngine.cu:559-590 ---------------------------------------------- */
/* jd: [32][2], jx/jy: [32][4] limbs.  May be called while a launch is outstanding: the uploads are ordered on the walk
 * stream, so that launch finishes with the table and mask it started with, the call blocks until it has (like the
 * reference's cudaMemcpyToSymbol, :565-583), and the next launch uses the new parameters. */
int kng_set_params(kng_engine *h, uint64_t dp_mask, const uint64_t *jd, const uint64_t *jx,
                   const uint64_t *jy);

/* ---- herd state: SetKangaroos/GetKangaroos/SetKangaroo, GPUEngine.cu:381-538 ---------------- */
/* x,y: n x 4 limbs; d: n x 2 limbs; strides in uint64_t units between consecutive kangaroos
 * (4,4,2 for packed arrays; 5,5,5 when pointing into an array of reference `Int`). n must
 * equal kng_nb_kangaroos(). */
int kng_set_kangaroos(kng_engine *h, const uint64_t *x, size_t xs, const uint64_t *y, size_t ys,
                      const uint64_t *d, size_t ds, uint64_t n);
/* waits for an in-flight launch, then returns the state it left (GPUEngine.cu:452 blocks the
 * same way through the null stream) */
int kng_get_kangaroos(kng_engine *h, uint64_t *x, size_t xs, uint64_t *y, size_t ys, uint64_t *d,
                      size_t ds, uint64_t n);
/* The same for kangaroos first .. first+count-1 only (x, y, d address kangaroo `first` at index 0): lets a
 * caller stream a herd to or from a work file (Backup.cpp:525-546 / :211-231, 96 B per kangaroo) in chunks
 * through the engine's pinned staging buffer instead of holding 80 B x herd in host memory.  The herd counts
 * as loaded once a range ending at the last kangaroo has been set.  Copies are stream-ordered behind an
 * in-flight launch, like the whole-herd calls. */
int kng_set_kangaroos_range(kng_engine *h, uint64_t first, uint64_t count, const uint64_t *x, size_t xs,
                            const uint64_t *y, size_t ys, const uint64_t *d, size_t ds);
int kng_get_kangaroos_range(kng_engine *h, uint64_t first, uint64_t count, uint64_t *x, size_t xs, uint64_t *y,
                            size_t ys, uint64_t *d, size_t ds);
/* Create the whole herd ON THE DEVICE instead of uploading it (new; replaces Kangaroo::CreateHerd,
 * Kangaroo.cpp:670-738, + SetKangaroos).  Kangaroo i gets a device distance dd uniform in
 * [1, 2^range_power) (counter-based generator keyed by seed and i) and the point
 *     base_type(i) + dd*G + final_add,     type(i) = i & 1,
 * where table[w][v] = v*256^w*G for v = 1..255 (entry 0 unused), windows = ceil(range_power/8), and
 * base_tame = b*G, base_wild = K - (N/2)*G + b*G, final_add = -b*G for a random scalar b: the random
 * offset keeps every batched affine addition generic.  The host library computes these inputs
 * (kngh_herd_params); points are 8 limbs: x[4], y[4].  Read the herd back with kng_get_kangaroos. */
int kng_build_herd(kng_engine *h, int range_power, uint64_t seed, const uint64_t *table, uint32_t windows,
                   const uint64_t base_tame[8], const uint64_t base_wild[8], const uint64_t final_add[8]);
/* ---- work-file snapshot (new): the kangaroo section of a work file, Backup.cpp:525-546 (save: GPUEngine::GetKangaroos,
 *      GPUEngine.cu:443-500, into 3 x N `Int`, then three 32-byte fwrite calls per kangaroo, every GPU thread parked
 *      meanwhile, Kangaroo.cpp:617-626) and Backup.cpp:211-231 (restore: three 32-byte fread calls per kangaroo into
 *      3 x N `Int`, then SetKangaroos, GPUEngine.cu:381-441).
 * The records here ARE the file's bytes: 96 per kangaroo, {x[4], y[4], d[4]} limbs, d = the TRUE distance mod n -- for odd
 * kIdx the 256-bit `wild_offset` (reduced mod n; NULL = none, records then carry the zero-extended device distance) has
 * been subtracted mod n exactly as Int::ModSubK1order does (GPUEngine.cu:477).
 * kng_snapshot packs the whole herd into a second device buffer (96 B x herd, allocated on first use and counted by
 *   kng_memory_bytes) with one kernel on the walk stream: ordered behind a launch in flight and ahead of the next, it
 *   freezes the state between two launches and returns at once -- the walk goes on while the records are read.
 * kng_snapshot_read copies records first .. first+count-1 of the last snapshot to `dst` (pinned memory for speed) on the
 *   snapshot's own stream and blocks until they have landed.  THE ONE ENTRY POINT THAT MAY BE CALLED FROM ANOTHER HOST
 *   THREAD than the engine's own while that one launches, waits and drains; the caller only has to keep kng_snapshot /
 *   kng_snapshot_write / kng_snapshot_release / kng_destroy away until its reads are done.
 * kng_snapshot_write uploads records into the same buffer; kng_snapshot_restore turns records first .. first+count-1 into
 *   herd state (adds wild_offset mod n for odd kIdx; stream-ordered like kng_set_kangaroos_range; the herd counts as loaded
 *   once a range ending at the last kangaroo has been restored).  A distance that does not fit the 128-bit device
 *   distance after the offset is an error (KNG_E_ARG, *bad_index = the first such kangaroo; the reference truncates,
 *   GPUEngine.cu:410-411).
 * kng_snapshot_release gives the buffer back. */
int kng_snapshot(kng_engine *h, const uint64_t wild_offset[4]);
int kng_snapshot_read(kng_engine *h, uint64_t first, uint64_t count, void *dst);
int kng_snapshot_write(kng_engine *h, uint64_t first, uint64_t count, const void *src);
int kng_snapshot_restore(kng_engine *h, uint64_t first, uint64_t count, const uint64_t wild_offset[4], uint64_t *bad_index);
int kng_snapshot_release(kng_engine *h);
/* overwrite one kangaroo; stream-ordered after an in-flight launch, never blocks the host
 * (the reference issues ten blocking 8-byte copies, GPUEngine.cu:504-530) */
int kng_set_kangaroo(kng_engine *h, uint64_t kidx, const uint64_t x[4], const uint64_t y[4],
                     const uint64_t d[2]);

/* ---- the hot path: callKernel / Launch, GPUEngine.cu:540-557, :607-679 ------------------------ */
/* start KNG_NB_RUN jumps for every kangaroo, asynchronously.  At most one launch may be
 * outstanding (not yet waited for). */
int kng_launch(kng_engine *h);
/* 1 when a launch has been started and not yet waited for, else 0 */
int kng_outstanding(const kng_engine *h);
/* 1 when a waited launch still has its points in the engine (kng_drain / kng_drain_view not yet called), else 0 */
int kng_undrained(const kng_engine *h);
/* block until the outstanding launch has finished.  spin != 0 busy-waits, otherwise the host
 * thread sleeps on the completion event (the reference polls with 1 ms sleeps, :621-629). */
int kng_wait(kng_engine *h, int spin);
/* copy out the distinguished points of the most recently WAITED launch.  *n_items = number
 * stored (<= cap and <= max_found), *n_lost = points dropped because max_found was exceeded
 * (GPUEngine.cu:641-648).  May be called while the next launch is already running: DP buffers
 * are double-buffered. */
int kng_drain(kng_engine *h, kng_item *items, uint32_t cap, uint32_t *n_items, uint32_t *n_lost);
/* the same without the per-item copy: *records points at the engine's pinned landing buffer (64-byte records, the
 * layout the kernel writes: OutputDP of GPUMath.h:173-188 padded to four 16-byte stores) holding *n_items points.
 * For hosts that ingest ~10^5 points per launch (many GPUs share a small DP size, Kangaroo.cpp:980-993).
 * LIFETIME of the view: until the next kng_drain / kng_drain_view of this engine, and -- with "dp_ring" 1, where the
 * view IS the buffer the kernel wrote -- at most until the SECOND kng_launch after the kng_wait that completed its launch:
 * the two landing buffers alternate, so in the pipelined order  wait A, launch B, drain_view A, wait B, launch C  the
 * kernel of launch C writes the buffer view A points into.  Consume (or copy) a view before waiting for the next launch;
 * kng_drain copies and has no such limit. */
typedef struct kng_dp_record {
    uint64_t x[4];
    uint64_t d[2]; /* device distance: wild kangaroos (odd kidx) still carry +wildOffset */
    uint64_t kidx;
    uint64_t reserved;
} kng_dp_record;
int kng_drain_view(kng_engine *h, const kng_dp_record **records, uint32_t *n_items, uint32_t *n_lost);

/* ---- whole-run audit on the device (new; the reference's closest tools are -wcheck, Check.cpp:141-411, which re-derives
 *      every stored distinguished point from its distance on the CPU, and the final key check, Kangaroo.cpp:196-206) ----
 * A walk error is permanent for its kangaroo: the invariant  (x, y) = d*G (tame) / K + d*G (wild)  holds after every exact
 * jump and never again after an inexact one.  The audit recomputes that point from the 128-bit DEVICE distance dd alone,
 *     base_type + dd*G + final_add     (type = kidx & 1; inputs as for kng_build_herd, with the full 16 windows:
 *                                       kngh_herd_params(128, ...) of the host library),
 * with general arithmetic only (nothing of the scheduled loop's short forms), and compares it with what the walk left.
 * kng_audit_setup uploads the inputs once (table: [16][256][8] limbs).
 * kng_audit_herd checks every kangaroo of the herd in x AND y; no launch may be outstanding (it borrows the walk's
 *   product planes; 4 x 16 B x herd of scratch are allocated for the call).
 * kng_audit_points checks n 64-byte records {x, d, kidx, reserved}: reserved = 0 compares all 256 bits of x (what
 *   kng_drain_view returns), reserved = 1 only what a hash-table entry keeps of x (limbs 0-1 and the 18 bucket bits of
 *   limb 2, HashTable.h:27-56).  Runs on its own stream: allowed while a launch is outstanding.
 * *n_bad = number of mismatches; the first min(bad_cap, 1024 per 2^21 records) offending indices (kIdx resp. position in
 * `recs`) go to bad_idx (may be NULL).  A zero distance has no affine point and counts as a mismatch.
 * kng_get_option "audit_us" = kernel time of the last audit call. */
#define KNG_AUDIT_WINDOWS 16
int kng_audit_setup(kng_engine *h, const uint64_t *table, const uint64_t base_tame[8], const uint64_t base_wild[8],
                    const uint64_t final_add[8]);
int kng_audit_herd(kng_engine *h, uint64_t *n_bad, uint64_t *bad_idx, uint32_t bad_cap);
int kng_audit_points(kng_engine *h, const kng_dp_record *recs, uint64_t n, uint64_t *n_bad, uint64_t *bad_idx,
                     uint32_t bad_cap);

/* ---- measurement (new; the reference only has the host-side MK/s average, Thread.cpp:254-300) */
/* HIP-event duration (ms) of the walk kernel of the most recently waited launch, measured on
 * the stream the kernel ran on. */
int kng_last_kernel_ms(const kng_engine *h, float *ms);
/* tuning knobs, must be set before kng_set_kangaroos:
 *   "group"  kangaroos walked per lane (batch size of the Montgomery inverse), power of two
 *   "lanes"  alternatively the lane count itself (multiple of 64, need not divide the herd: waves then
 *            walk ceil or floor of herd/lanes kangaroos)
 *   "block"  threads per workgroup (multiple of 64)
 *   "share"  waves that share one modular inversion per jump: 8 = the eight waves of a 512-thread block, i.e. one inversion
 *            per CU; 4 = 256-thread blocks, one wave per SIMD, for herds too small to give every CU a 512-thread block (their
 *            launches are 64 serial inversions: latency, not throughput); -1 (default) = 4 when lanes < 512 x CUs, else 8.
 *            Reads back what a launch uses (8 or 4).  (Rounds 1-3 also carried 1 = every wave inverts for itself.)
 *   "dsplit" -1 (default): stream only the low word of the 128-bit distances through HBM when every jump
 *            distance given to kng_set_params is below 2^58 (ranges up to 115 bits; 2^50 with "asm" 0): the high word
 *            is then updated by an L2 atomic of the lanes whose low word carried;
 *            0 = never, 1 = whenever the table allows it (all high words zero).  Reads back 0/1 = in effect.
 *   "asm"    1 (default): the per-kangaroo loop runs as one scheduled asm statement (kng_walk_asm.h); 0 = the
 *            compiler-scheduled loop (also what herds beyond 2^28 kangaroos get).  Same results.
 *            2 = MEASUREMENT ONLY: the scheduled loop of the headline form (share 8, low-word streaming) with the global and LDS
 *            accesses of its per-kangaroo loop left out -- the integer-ALU ceiling of the kernel (bench.py roofline.alu_ceiling,
 *            SURVEY 8d (ii)).  WRONG results on purpose; reload the herd before walking it again.
 *   "dp_ring" 1 (default): the kernel writes its DP records straight into pinned, device-mapped host memory, one buffer
 *            per launch slot, the count landing last; 0 = device buffer + copy at drain time (rounds 1-2).  Only the
 *            buffers of the mode in use are allocated; switching releases the others and fails with KNG_E_STATE while
 *            a waited launch has not been drained
 *   "steps"  jumps per launch (default KNG_NB_RUN; only tests change it)
 * kng_get_option reads them back (also "lanes", "waves_per_cu", and "exact_exits": how many wave-iterations of the last
 * waited launch the scheduled loop handed to the general arithmetic -- its short forms flag a superset of the operands
 * they are not exact for). */
int kng_set_option(kng_engine *h, const char *key, int64_t value);
int kng_get_option(const kng_engine *h, const char *key, int64_t *value);

/* ---- device self-test of the 256-bit primitives (replaces the compiled-out GPU_CHECK kernel,
 *      GPUEngine.cu:43-92): r[i] = op(a[i], b[i]) on the GPU, n x 4 limbs each ----------------- */
#define KNG_OP_MODMUL 0 /* GPUMath.h:810-858  */
#define KNG_OP_MODSQR 1 /* GPUMath.h:909-1019 */
#define KNG_OP_MODSUB 2 /* GPUMath.h:476-494  */
#define KNG_OP_MODINV 3 /* GPUMath.h:700-803  */
int kng_test_fieldop(int dev, int op, const uint64_t *a, const uint64_t *b, uint64_t *r, uint64_t n);

/* ---- pinned host memory: AllocatePinnedMemory/FreePinnedMemory, GPUEngine.cu:311-327 ---------- */
void *kng_alloc_pinned(size_t size); /* NULL on failure */
void kng_free_pinned(void *p);

const char *kng_last_error(void); /* thread-local text of the last failure */
const char *kng_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KANGAROO_HIP_H */
