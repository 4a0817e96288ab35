// kng_savework.h -- what the two link-time replacements of the reference program share about work files:
// Backup_kng.cpp (Kangaroo::SaveWork / Kangaroo::FectchKangaroos, Backup.cpp:449-563 / :286-364, run by the main thread)
// and SolveKeyGPU_kng.cpp (the GPU threads).  The reference's headers have no member to hang this on, so it is a small
// process-wide registry keyed by the thread's TH_PARAM.
//
// Save protocol.  The reference parks every thread for the whole save: a GPU thread converts its herd into 3 x N `Int`
// (GPUEngine::GetKangaroos), sets isWaiting and blocks on saveMutex until the main thread has written the table and 96 bytes
// per kangaroo with three fwrite calls each (Backup.cpp:525-546).  Here a save is a GENERATION:
//   main thread   g = ++requested; saveRequest = true; waits for isWaiting of every thread (as the reference does)
//   GPU thread    sees the new generation at a launch boundary: kng_snapshot (the herd frozen as work-file records in a second
//                 device buffer, stream-ordered between two launches), next launch started at once, the drained points flushed
//                 into the table, its table threads put on hold for g (kng_ingest.h), isWaiting = true -- and it goes on
//                 walking; the points it finds meanwhile wait in the queue
//   main thread   writes header + table (the reference's own SaveWork(fileName, f, HEADW, ..) / HashTable::SaveTable), then
//                 finished = g: table threads resume by themselves; then streams every registered engine's snapshot
//                 (kng_snapshot_read into pinned memory, one fwrite per 24 MB) -- no Int[3N], no per-kangaroo calls
// A file therefore holds the table and the kangaroos of the same launch boundary, exactly as the reference's does, and the
// GPUs idle for the snapshot kernel (under a millisecond) instead of the whole save.
#ifndef KNG_SAVEWORK_H
#define KNG_SAVEWORK_H

#include <atomic>
#include <cstdint>
#include <string>

struct kng_engine;

namespace kng_save {

extern std::atomic<uint64_t> requested; // generation of the newest save request
extern std::atomic<uint64_t> finished;  // newest generation whose table section is on disk (or that was given up)

// a GPU thread announces the engine behind its TH_PARAM / withdraws it (blocks while the saver is reading its snapshot)
void attach(const void *th_param, kng_engine *eng);
void detach(const void *th_param);
// the engine's snapshot now holds the herd for this generation (kng_snapshot has been queued)
void snapshot_taken(const void *th_param, uint64_t generation);

// the kangaroos of a work file that belong to a GPU thread: where they are instead of 3 x N `Int` (FectchKangaroos ->
// SolveKeyGPU)
struct Restore {
  std::string file;
  uint64_t offset = 0; // byte offset of the thread's first 96-byte record
  uint64_t count = 0;  // records available (<= the thread's herd; the rest is created)
};
void plan_restore(const void *th_param, const Restore &r);
bool take_restore(const void *th_param, Restore &r);

} // namespace kng_save
#endif
