// SolveKeyGPU_kng.cpp -- `Kangaroo::SolveKeyGPU` (Kangaroo.cpp:510-644) for LINK-TIME replacement (VERDICT r4
// item 1).  Compiled against the reference's own Kangaroo.h; the reference's Kangaroo.o keeps every other member, its
// SolveKeyGPU symbol is made weak (objcopy -W, oracle/Makefile) so that this definition wins at link time.  No reference source
// is edited.  Needs HashTable_kng.o (the batch interface) and GPUEngine.cpp of this repo (kng_shim_engine) in the same link.
//
// What the reference's loop does per launch, on the GPU's own host thread, between two kernels (Kangaroo.cpp:572-615):
// Launch() -> one Int-pair ITEM per point -> ghMutex -> AddToTable per point -> SetKangaroo for same-herd collisions.  One
// MI355X at the program's own DP 14 hands that loop 32 768 points every 21 ms; with eight GPUs all eight threads queue on the
// one mutex.  Here the GPU thread only moves bytes: wait, start the next kernel, copy the launch's 64-byte records out of the
// engine's pinned ring (kng_drain_view) into a queue.  Table threads (4 per GPU by default, KNG_TABLE_THREADS, never more than
// the CPUs the process may use) take 8192-point chunks and insert them with kng_ht_ingest: no Int objects, no ghMutex, stripe
// locks inside the table.  What is not ADD_OK comes back as an event and is handled by the GPU thread exactly as the
// reference handles it -- under ghMutex: CollisionCheck for a tame/wild collision (it ends the search), a fresh kangaroo through
// CreateHerd + SetKangaroo for a collision inside a herd, collisionInSameHerd++.  The queue is bounded (64 launches' worth): a
// host that cannot keep up stalls the GPU thread instead of losing points, and the stall is counted (KNG_STATS=1 prints it).
//
// Kept from the reference, because other code depends on it: the banner lines, counters[thId], hasStarted / isRunning /
// isWaiting and the saveRequest handshake of SaveWork (Backup.cpp:454-563) -- before parking, every queued point is in the
// table, so a work file never misses a point that the saved kangaroos have already passed -- GetKangaroos into ph->px/py/distance,
// and client mode (SendToServer), which keeps the reference's lock-step shape: the table is not involved there.
#include <pthread.h>
#include <sched.h>

#include <condition_variable>
#include <cinttypes>
#include <cmath>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "Kangaroo.h"
#include "SECPK1/Random.h"
#include "Timer.h"
#include "kangaroo_hip.h"
#include "kng_cpus.h"
#include "kng_hashtable_ext.h"
#include "kng_host.h"
#include "kng_ingest.h"

#ifndef WITHGPU
#error "SolveKeyGPU_kng.cpp replaces the GPU path: build with -DWITHGPU, like the reference's gpu=1 target"
#endif

extern "C" kng_engine *kng_shim_engine(GPUEngine *g); // GPUEngine.cpp of this repo

using namespace std;

#ifndef safe_delete_array
#define safe_delete_array(x) \
  if (x) {                   \
    delete[] x;              \
    x = NULL;                \
  }
#endif

using namespace kng_ingest;

void Kangaroo::SolveKeyGPU(TH_PARAM *ph) {
  const int thId = ph->threadId;

  GPUEngine *gpu = new GPUEngine(ph->gridSizeX, ph->gridSizeY, ph->gpuId, 65536 * 2);

  if (keyIdx == 0) ::printf("GPU: %s (%.1f MB used)\n", gpu->deviceName.c_str(), gpu->GetMemory() / 1048576.0);

  const double t0 = Timer::get_tick();

#ifdef USE_SYMMETRY
  Int *wildOffset = &rangeWidthDiv4;
#else
  Int *wildOffset = &rangeWidthDiv2;
#endif
  gpu->SetWildOffset(wildOffset);
  gpu->SetParams(dMask, jumpDistance, jumpPointx, jumpPointy);

  // Kangaroo::CreateHerd on the host takes 20 s for the 2^23 kangaroos of one MI355X (one scalar multiplication each, per GPU
  // and per key of the input file).  When no work file hands the kangaroos in, the engine builds the herd itself
  // (kng_build_herd: SURVEY 8 f2; same law -- distances uniform in [0, 2^rangePower), wild ones shifted by -N/2, alternating
  // types from TAME -- from a seed drawn from the program's own generator, 6 ms of kernel time).  KNG_HOST_HERD=1 keeps the
  // reference's host path; the symmetry build keeps it too (its equivalence-class switch is not in the device kernel).
  bool onDevice = false;
#ifndef USE_SYMMETRY
  if (ph->px == NULL && !getenv("KNG_HOST_HERD")) {
    if (kng_engine *e = kng_shim_engine(gpu)) {
      if (keyIdx == 0) ::printf("SolveKeyGPU Thread GPU#%d: creating kangaroos...\n", ph->gpuId);
      LOCK(ghMutex); // the generator is shared (CreateHerd takes the same lock, Kangaroo.cpp:683)
      const uint64_t seed = ((uint64_t)rndl() << 32) ^ (uint64_t)rndl() ^ ((uint64_t)ph->gpuId << 56);
      UNLOCK(ghMutex);
      const uint32_t windows = (uint32_t)(rangePower + 7) / 8;
      vector<uint64_t> table((size_t)windows * 256 * 8);
      uint64_t bt[8], bw[8], fin[8], woff[4] = {wildOffset->bits64[0], wildOffset->bits64[1], wildOffset->bits64[2], wildOffset->bits64[3]};
      if (rangePower >= 1 && rangePower <= 128 &&
          kngh_herd_params(rangePower, woff, keyToSearch.x.bits64, keyToSearch.y.bits64, seed, table.data(), bt, bw, fin) == 0 &&
          kng_build_herd(e, rangePower, seed, table.data(), windows, bt, bw, fin) == KNG_OK) {
        onDevice = true;
        if (workFile.length() > 0 && saveKangaroo) { // SaveWork will ask for them (GetKangaroos fills the arrays)
          ph->px = new Int[ph->nbKangaroo];
          ph->py = new Int[ph->nbKangaroo];
          ph->distance = new Int[ph->nbKangaroo];
        }
      } else {
        ::fprintf(stderr, "SolveKeyGPU_kng: herd creation on the device failed (%s); creating the kangaroos on the host\n", kng_last_error());
      }
    }
  }
#endif
  if (!onDevice) {
    if (ph->px == NULL) {
      // no kangaroos loaded from a work file: create them, one block of GPU_GRP_SIZE per GPU thread, tame first
      if (keyIdx == 0) ::printf("SolveKeyGPU Thread GPU#%d: creating kangaroos...\n", ph->gpuId);
      const uint64_t nbThread = gpu->GetNbThread();
      ph->px = new Int[ph->nbKangaroo];
      ph->py = new Int[ph->nbKangaroo];
      ph->distance = new Int[ph->nbKangaroo];
      for (uint64_t i = 0; i < nbThread; i++)
        CreateHerd(GPU_GRP_SIZE, &(ph->px[i * GPU_GRP_SIZE]), &(ph->py[i * GPU_GRP_SIZE]), &(ph->distance[i * GPU_GRP_SIZE]), TAME);
    }
    gpu->SetKangaroos(ph->px, ph->py, ph->distance);
    if (workFile.length() == 0 || !saveKangaroo) {
      // nobody will ask for the kangaroos back
      safe_delete_array(ph->px);
      safe_delete_array(ph->py);
      safe_delete_array(ph->distance);
    }
  }

  gpu->callKernel();

  const double t1 = Timer::get_tick();

  if (keyIdx == 0) ::printf("SolveKeyGPU Thread GPU#%d: 2^%.2f kangaroos [%.1fs]\n", ph->gpuId, log2((double)ph->nbKangaroo), (t1 - t0));

  ph->hasStarted = true;

  kng_engine *eng = kng_shim_engine(gpu);

  if (clientMode || eng == NULL) {
    // Points go to a server, not to the table: the reference's loop as it is (Kangaroo.cpp:577-590).  Also the way out when
    // the engine could not be created: Launch() then reports the dead engine on every call, like the reference's would.
    vector<ITEM> dps, gpuFound;
    double lastSent = 0;
    while (!endOfSearch) {
      const bool ok = gpu->Launch(gpuFound);
      if (!clientMode) {
        if (!ok) break; // no engine: nothing will ever be found by this thread
        continue;
      }
      counters[thId] += ph->nbKangaroo * NB_RUN;
      dps.insert(dps.end(), gpuFound.begin(), gpuFound.end());
      const double now = Timer::get_tick();
      if (now - lastSent > SEND_PERIOD) {
        LOCK(ghMutex);
        SendToServer(dps, ph->threadId, ph->gpuId);
        UNLOCK(ghMutex);
        lastSent = now;
      }
      if (saveRequest && !endOfSearch) {
        if (saveKangaroo) gpu->GetKangaroos(ph->px, ph->py, ph->distance);
        ph->isWaiting = true;
        LOCK(saveMutex);
        ph->isWaiting = false;
        UNLOCK(saveMutex);
      }
    }
  } else {
    int tableThreads = 4;
    if (const char *e = getenv("KNG_TABLE_THREADS")) tableThreads = atoi(e);
    const int gpus = nbGPUThread > 0 ? nbGPUThread : 1;
    const int roomFor = ((int)(kng_effective_cpus() + 0.5) - gpus - nbCPUThread) / gpus; // the GPU threads and the program's CPU walkers come first
    if (tableThreads > roomFor) tableThreads = roomFor;
    if (tableThreads < 1) tableThreads = 1;
    const uint64_t off[2] = {wildOffset->bits64[0], wildOffset->bits64[1]};
    // points per launch decide how many chunks "64 launches" are
    int bits = 0;
    for (uint64_t mk = dMask; mk; mk &= mk - 1) bits++;
    const uint64_t perLaunch = bits >= 64 ? 0 : (ph->nbKangaroo * NB_RUN) >> bits;
    const size_t maxChunks = QUEUE_LAUNCHES * (size_t)(perLaunch / CHUNK + 1);
    Ingest ingest(&hashTable, off, tableThreads, maxChunks);

    vector<Event> events;
    uint64_t launches = 0, lostTotal = 0, nEvents = 0;
    double blocked = 0, waitGpu = 0;
    bool lostWarning = false, behindWarning = false;
    const double loop0 = Timer::get_tick();
    // KNG_STATS=1: one line when the loop ends; KNG_STATS=<seconds> (> 1): also a "(running)" line every so many seconds
    const double statsEvery = getenv("KNG_STATS") ? atof(getenv("KNG_STATS")) : 0.0;
    double statsNext = loop0 + statsEvery;
    auto report = [&](const char *state) {
      const double wall = Timer::get_tick() - loop0;
      const Ingest::Totals tt = ingest.totals();
      ::fprintf(stderr,
                "\nSolveKeyGPU_kng GPU#%d%s: %" PRIu64 " launches in %.3f s = %.1f MK/s; points %" PRIu64 " (lost %" PRIu64 "), events %" PRIu64
                "; GPU thread waited %.3f s for kernels, %.3f s for queue room; %d table threads busy %.3f s (%.0f ns/point), queue high water %zu of "
                "%zu chunks\n",
                ph->gpuId, state, launches, wall, wall > 0 ? (double)launches * (double)ph->nbKangaroo * NB_RUN / wall / 1e6 : 0.0, tt.points, lostTotal,
                nEvents, waitGpu, blocked, tableThreads, tt.busy_s, tt.points ? tt.busy_s / (double)tt.points * 1e9 : 0.0, tt.high_water,
                maxChunks);
    };

    while (!endOfSearch) {
      // the launch in flight, then the next one at once: from here on the GPU is busy again while the host works
      double tw = Timer::get_tick();
      if (kng_wait(eng, 0) != KNG_OK) {
        ::printf("GPUEngine: Launch: %s\n", kng_last_error());
        break;
      }
      waitGpu += Timer::get_tick() - tw;
      if (kng_launch(eng) != KNG_OK) {
        ::printf("GPUEngine: Kernel: %s\n", kng_last_error());
        break;
      }
      const kng_dp_record *recs = NULL;
      uint32_t nb = 0, lost = 0;
      if (kng_drain_view(eng, &recs, &nb, &lost) != KNG_OK) {
        ::printf("GPUEngine: Launch: %s\n", kng_last_error());
        break;
      }
      if (lost && !lostWarning) { // GPUEngine.cu:641-648
        ::printf("\nWarning, %u items lost\nHint: Search with less threads (-g) or increse dp (-d)\n", lost);
        lostWarning = true;
      }
      lostTotal += lost;
      launches++;
      counters[thId] += ph->nbKangaroo * NB_RUN;
      blocked += ingest.push(recs, nb); // the view is only good until the launch after next: copy now
      if (blocked > 1.0 && !behindWarning) {
        // nothing is lost -- the GPU waits -- but the user should know why the rate is below the kernel's (INTEGRATION.md has
        // the table of -d against table threads)
        ::printf("\nWarning, the distinguished-point table cannot keep up with GPU#%d (%d table threads): the GPU waits\n"
                 "Hint: increase dp (-d), or give the table more threads (KNG_TABLE_THREADS) if the machine has CPUs to spare\n",
                 ph->gpuId, tableThreads);
        behindWarning = true;
      }

      // what the table threads could not simply store (Kangaroo.cpp:594-612, AddToTable :306-314)
      ingest.take_events(events);
      if (!events.empty()) {
        LOCK(ghMutex);
        for (size_t g = 0; !endOfSearch && g < events.size(); g++) {
          const Event &ev = events[g];
          const uint32_t kType = (uint32_t)(ev.rec.kidx % 2);
          bool keep = false;
          if (ev.status == ADD_COLLISION) {
            // the distance GPUEngine::Launch would have handed over (GPUEngine.cu:668-674)
            Int dist;
            dist.SetInt32(0);
            dist.bits64[0] = ev.rec.d[0];
            dist.bits64[1] = ev.rec.d[1];
            if (kType == WILD) dist.ModSubK1order(wildOffset);
            int128_t stored;
            stored.i64[0] = ev.stored_d[0];
            stored.i64[1] = ev.stored_d[1];
            HashTable::CalcDistAndType(stored, &hashTable.kDist, &hashTable.kType); // what HashTable::Add leaves behind
            keep = CollisionCheck(&hashTable.kDist, hashTable.kType, &dist, kType);
          }
          if (!keep) {
            // collision inside one herd (or the same point twice): that kangaroo follows another one from now on, replace it
            Int px, py, d;
            CreateHerd(1, &px, &py, &d, kType, false);
            gpu->SetKangaroo(ev.rec.kidx, &px, &py, &d);
            collisionInSameHerd++;
          }
          nEvents++;
        }
        UNLOCK(ghMutex);
      }

      if (statsEvery > 1.0 && Timer::get_tick() >= statsNext) {
        report(" (running)");
        statsNext += statsEvery;
      }

      if (saveRequest && !endOfSearch) {
        ingest.flush(); // the table must hold every point the kangaroos have passed before either is written
        if (saveKangaroo) gpu->GetKangaroos(ph->px, ph->py, ph->distance);
        ph->isWaiting = true;
        LOCK(saveMutex);
        ph->isWaiting = false;
        UNLOCK(saveMutex);
      }
    }

    if (getenv("KNG_STATS")) report("");
  } // ~Ingest: table threads joined, whatever was still queued is dropped (the search is over)

  safe_delete_array(ph->px);
  safe_delete_array(ph->py);
  safe_delete_array(ph->distance);
  delete gpu;

  ph->isRunning = false;
}
