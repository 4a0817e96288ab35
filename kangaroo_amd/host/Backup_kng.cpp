// Backup_kng.cpp -- the kangaroo section of the reference program's work files without parking the GPUs:
// LINK-TIME replacements for
//     Kangaroo::SaveWork(uint64_t, double, TH_PARAM *, int)      Backup.cpp:449-563   (-ws -wi N)
//     Kangaroo::FectchKangaroos(TH_PARAM *)                      Backup.cpp:286-364   (-i file)
// compiled against the reference's own Kangaroo.h.  In the reference's Backup.o both symbols are made weak and get a second
// name (kng_ref_SaveWork / kng_ref_FectchKangaroos, oracle/Makefile), so these definitions bind and everything they do not
// improve on is DELEGATED to the original code instead of restated: client mode, kangaroos kept by a server, saves without
// kangaroos, KNG_REF_SAVE=1.  No reference source is edited; every other member of Backup.o (headers, LoadWork, FetchWalks,
// the 5-argument SaveWork that writes header + HashTable::SaveTable, -winfo / -wcheck / merging) is used as it is.
//
// What changes for `-ws` with a GPU (protocol: kng_savework.h).  The reference stops every GPU for the whole save: 2^23
// kangaroos -> 3 x 2^23 `Int` through GPUEngine::GetKangaroos (one mod-n subtraction per wild one, serial, 1 GB of host
// objects), then 25 million 32-byte fwrite calls per GPU from the main thread (Backup.cpp:525-546), the table before that.
// Here the herd is frozen on the device in the file's own byte layout (kng_snapshot: 96-byte records, wild offset already
// removed) between two launches, the walk goes on, and this file streams the records device -> pinned buffer -> one fwrite
// per 24 MB.  The bytes written are the bytes the reference writes for the same state (tests: oracle/saveprobe.cpp links both
// objects and compares files; KNG_SAVE_VERIFY=1 compares every streamed record with GPUEngine::GetKangaroos of the same
// launch boundary in the running program).  `-i`: the GPU threads' records are not read into `Int` arrays at all -- the
// section's offset is handed to SolveKeyGPU_kng.cpp, which uploads the file's bytes and unpacks them on the device.
#include <fcntl.h>
#include <pthread.h>
#include <unistd.h>

#include <cerrno>
#include <cinttypes>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

#include "Kangaroo.h"
#include "Timer.h"
#include "kangaroo_hip.h"
#include "kng_savework.h"

// the reference's own definitions, under their second names (Itanium ABI: `this` is the first argument)
extern "C" void kng_ref_SaveWork(Kangaroo *self, uint64_t totalCount, double totalTime, TH_PARAM *threads, int nbThread);
extern "C" void kng_ref_FectchKangaroos(Kangaroo *self, TH_PARAM *threads);

using namespace std;

// ---- the registry ---------------------------------------------------------------------------------------------------
namespace kng_save {
std::atomic<uint64_t> requested{0}, finished{0};
namespace {
struct Slot {
  kng_engine *eng = nullptr;
  uint64_t snap_gen = 0;
};
std::mutex g_lock;            // the maps
std::mutex g_stream_lock;     // held by the saver while it reads snapshots: detach() waits for it
std::map<const void *, Slot> g_slots;
std::map<const void *, Restore> g_restores;
} // namespace
void attach(const void *ph, kng_engine *eng) {
  std::lock_guard<std::mutex> l(g_lock);
  g_slots[ph].eng = eng;
  g_slots[ph].snap_gen = 0;
}
void detach(const void *ph) {
  std::lock_guard<std::mutex> s(g_stream_lock);
  std::lock_guard<std::mutex> l(g_lock);
  g_slots.erase(ph);
}
void snapshot_taken(const void *ph, uint64_t generation) {
  std::lock_guard<std::mutex> l(g_lock);
  auto it = g_slots.find(ph);
  if (it != g_slots.end()) it->second.snap_gen = generation;
}
void plan_restore(const void *ph, const Restore &r) {
  std::lock_guard<std::mutex> l(g_lock);
  g_restores[ph] = r;
}
bool take_restore(const void *ph, Restore &r) {
  std::lock_guard<std::mutex> l(g_lock);
  auto it = g_restores.find(ph);
  if (it == g_restores.end()) return false;
  r = it->second;
  g_restores.erase(it);
  return true;
}
static kng_engine *engine_with_snapshot(const void *ph, uint64_t generation) {
  std::lock_guard<std::mutex> l(g_lock);
  auto it = g_slots.find(ph);
  return (it != g_slots.end() && it->second.snap_gen == generation) ? it->second.eng : nullptr;
}
} // namespace kng_save

// ---- staging --------------------------------------------------------------------------------------------------------
static const uint64_t REC = 96;                 // bytes per kangaroo in a work file (Backup.cpp:532-534)
static const uint64_t STAGE_KANG = 1u << 18;    // 24 MB per piece
static uint8_t *stage_buffer() {                // pinned when the engine library can give it (main thread only)
  static uint8_t *buf = NULL;
  if (!buf) buf = (uint8_t *)kng_alloc_pinned(STAGE_KANG * REC);
  if (!buf) buf = (uint8_t *)malloc(STAGE_KANG * REC);
  return buf;
}

// ---- Kangaroo::SaveWork ------------------------------------------------------------------------------------------------
void Kangaroo::SaveWork(uint64_t totalCount, double totalTime, TH_PARAM *threads, int nbThread) {
  const uint64_t gen = ++kng_save::requested;

  if (clientMode || saveKangarooByServer || !saveKangaroo || getenv("KNG_REF_SAVE")) {
    // nothing to stream (or not to a file): the reference's code as it is.  GPU threads of SolveKeyGPU_kng.cpp still only
    // pause their table threads for it, and fill ph->px/py/distance when it will want them (KNG_REF_SAVE).
    kng_ref_SaveWork(this, totalCount, totalTime, threads, nbThread);
    kng_save::finished = gen;
    return;
  }

  LOCK(saveMutex);
  const double t0 = Timer::get_tick();

  // every thread at a save point (Backup.cpp:454-470)
  saveRequest = true;
  int timeout = wtimeout;
  while (!isWaiting(threads) && timeout > 0) {
    Timer::SleepMillis(20);
    timeout -= 20;
  }
  if (timeout <= 0) {
    if (!endOfSearch) ::printf("\nSaveWork timeout !\n");
    kng_save::finished = gen; // whoever did arrive goes on
    UNLOCK(saveMutex);
    return;
  }
  const double tParked = Timer::get_tick();

  string fileName = workFile;
  if (splitWorkfile) fileName = workFile + "_" + Timer::getTS();
  FILE *f = fopen(fileName.c_str(), "wb");
  if (f == NULL) {
    ::printf("\nSaveWork: Cannot open %s for writing\n", fileName.c_str());
    ::printf("%s\n", ::strerror(errno));
    saveRequest = false;
    kng_save::finished = gen;
    UNLOCK(saveMutex);
    return;
  }

  // header + table: the reference's own writer (Backup.cpp:395-407)
  SaveWork(fileName, f, HEADW, totalCount, totalTime);
  if (splitWorkfile) hashTable.Reset(); // (Backup.cpp:550-551; done here so that the table threads restart on the emptied table)
  const double tTable = Timer::get_tick();
  // the table is on its way to the disk: threads that only paused their table threads for it may insert again; the kangaroos
  // they froze for this generation stay where they are until the next request
  kng_save::finished = gen;

  uint64_t totalWalk = 0;
  for (int i = 0; i < nbThread; i++) totalWalk += threads[i].nbKangaroo;
  ::fwrite(&totalWalk, sizeof(uint64_t), 1, f);

  const bool verify = getenv("KNG_SAVE_VERIFY") != NULL;
  uint8_t *buf = stage_buffer();
  const uint64_t point = totalWalk / 16;
  uint64_t pointPrint = 0, streamed = 0, differ = 0;
  bool ok = buf != NULL;
  {
    std::lock_guard<std::mutex> reading(kng_save::g_stream_lock);
    for (int i = 0; ok && i < nbThread; i++) {
      kng_engine *eng = kng_save::engine_with_snapshot(&threads[i], gen);
      const uint64_t n = threads[i].nbKangaroo;
      if (eng == NULL && n && threads[i].px == NULL) {
        ::printf("\nSaveWork: thread %d has neither a snapshot nor its kangaroos in memory\n", i);
        ok = false;
        break;
      }
      for (uint64_t first = 0; ok && first < n; first += STAGE_KANG) {
        const uint64_t m = n - first < STAGE_KANG ? n - first : STAGE_KANG;
        if (eng) {
          if (kng_snapshot_read(eng, first, m, buf) != KNG_OK) {
            ::printf("\nSaveWork: GPU#%d: %s\n", threads[i].gpuId, kng_last_error());
            ok = false;
            break;
          }
          streamed += m;
          if (verify && threads[i].px) // GetKangaroos of the same launch boundary (SolveKeyGPU_kng.cpp fills the arrays in this mode)
            for (uint64_t k = 0; k < m; k++)
              differ += (memcmp(buf + k * REC, threads[i].px[first + k].bits64, 32) || memcmp(buf + k * REC + 32, threads[i].py[first + k].bits64, 32) ||
                         memcmp(buf + k * REC + 64, threads[i].distance[first + k].bits64, 32)) ? 1 : 0;
        } else { // CPU threads (and GPU threads run by the reference's own SolveKeyGPU): their arrays, Backup.cpp:530-536
          for (uint64_t k = 0; k < m; k++) {
            memcpy(buf + k * REC, threads[i].px[first + k].bits64, 32);
            memcpy(buf + k * REC + 32, threads[i].py[first + k].bits64, 32);
            memcpy(buf + k * REC + 64, threads[i].distance[first + k].bits64, 32);
          }
        }
        if (::fwrite(buf, REC, m, f) != m) {
          ::printf("\nSaveWork: write to %s failed: %s\n", fileName.c_str(), ::strerror(errno));
          ok = false;
          break;
        }
        for (pointPrint += m; point && pointPrint > point; pointPrint -= point) ::printf(".");
      }
    }
  }

  const uint64_t size = FTell(f);
  fclose(f);
  saveRequest = false;
  UNLOCK(saveMutex);

  const double t1 = Timer::get_tick();
  if (verify)
    ::printf("\nSaveWork_kng verify: %" PRIu64 " streamed kangaroos, %" PRIu64 " differ from GPUEngine::GetKangaroos\n", streamed, differ);
  if (getenv("KNG_STATS"))
    ::fprintf(stderr, "\nSaveWork_kng: threads at the save point after %.3f s, header + table %.3f s (table threads released), %" PRIu64
                      " kangaroos (%" PRIu64 " streamed from device snapshots) %.3f s%s\n",
              tParked - t0, tTable - tParked, totalWalk, streamed, t1 - tTable, ok ? "" : " -- FAILED, file incomplete");
  time_t now = time(NULL);
  ::printf("done [%.1f MB] [%s] %s", (double)size / (1024.0 * 1024.0), GetTimeStr(t1 - t0).c_str(), ctime(&now));
}

// ---- Kangaroo::FectchKangaroos -----------------------------------------------------------------------------------------
void Kangaroo::FectchKangaroos(TH_PARAM *threads) {
  if (saveKangarooByServer || clientMode || nbLoadedWalk <= 0 || nbGPUThread == 0 || fRead == NULL || getenv("KNG_REF_SAVE")) {
    kng_ref_FectchKangaroos(this, threads);
    return;
  }

  const double sFetch = Timer::get_tick();
  ::printf("Restoring");
  const uint64_t nbSaved = nbLoadedWalk;
  uint64_t created = 0;

  // CPU threads: into their arrays, with the reference's reader (Backup.cpp:309-318, :211-231)
  for (int i = 0; i < nbCPUThread; i++) {
    threads[i].px = new Int[CPU_GRP_SIZE];
    threads[i].py = new Int[CPU_GRP_SIZE];
    threads[i].distance = new Int[CPU_GRP_SIZE];
    FetchWalks(CPU_GRP_SIZE, threads[i].px, threads[i].py, threads[i].distance);
  }
  // GPU threads: no Int[3N] and no 3N fread calls -- the thread is told where its records are
  for (int i = 0; i < nbGPUThread; i++) {
    ::printf(".");
    const int id = nbCPUThread + i;
    kng_save::Restore r;
    r.file = inputFile;
    r.offset = FTell(fRead);
    r.count = (uint64_t)nbLoadedWalk < threads[id].nbKangaroo ? (uint64_t)nbLoadedWalk : threads[id].nbKangaroo;
    if (r.count) {
      FSeek(fRead, r.offset + r.count * REC);
      nbLoadedWalk -= (int64_t)r.count;
      kng_save::plan_restore(&threads[id], r);
    }
  }
  ::printf("Done\n");

  const double eFetch = Timer::get_tick();
  if (nbLoadedWalk != 0) ::printf("FectchKangaroos: Warning %.0f unhandled kangaroos !\n", (double)nbLoadedWalk);
  if (nbSaved < totalRW) created = totalRW - nbSaved;
  ::printf("FectchKangaroos: [2^%.2f kangaroos loaded] [%.0f created] [%s]\n", log2((double)nbSaved), (double)created, GetTimeStr(eFetch - sFetch).c_str());

  if (fRead) fclose(fRead);
}
