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
// isWaiting.  Client mode (SendToServer; the table is not involved there), a build with USE_SYMMETRY and an engine that could
// not be created are not restated here: the reference's own SolveKeyGPU, still present in Kangaroo.o under a second name
// (kng_ref_SolveKeyGPU, oracle/Makefile), runs them.
//
// Work files (protocol in kng_savework.h, file side in Backup_kng.cpp).  The reference parks the GPU for a whole
// save (Kangaroo.cpp:617-626: GetKangaroos into 3 x N Int, then blocked on saveMutex until the main thread has written table
// and kangaroos).  Here, at the launch boundary where the request is seen: kng_snapshot freezes the herd as work-file records
// on the device, the next kernel starts at once, the points drained so far are flushed into the table, the table threads go
// on hold until the table section is on disk, isWaiting is raised -- and the thread keeps walking; points found meanwhile wait
// in the queue.  The file holds table and kangaroos of the same launch boundary, like the reference's.  `-i`: the thread's
// records are uploaded as the file's bytes and unpacked on the device (kng_snapshot_write / kng_snapshot_restore).
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#include <condition_variable>
#include <cinttypes>
#include <cmath>
#include <cstring>
#include <deque>
#include <unordered_map>
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
#include "kng_placement.h"
#include "kng_savework.h"

#ifndef WITHGPU
#error "SolveKeyGPU_kng.cpp replaces the GPU path: build with -DWITHGPU, like the reference's gpu=1 target"
#endif

extern "C" kng_engine *kng_shim_engine(GPUEngine *g); // GPUEngine.cpp of this repo
// the reference's own Kangaroo::SolveKeyGPU (Kangaroo.cpp:510-644), kept in Kangaroo.o under a second name (oracle/Makefile)
extern "C" void kng_ref_SolveKeyGPU(Kangaroo *self, TH_PARAM *ph);

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
#ifdef USE_SYMMETRY
  kng_ref_SolveKeyGPU(this, ph); // the equivalence-class switch of that build is not in the device kernels
#else
  if (clientMode) { // points go to a server, not to the table: nothing here to gain (Kangaroo.cpp:577-590)
    kng_ref_SolveKeyGPU(this, ph);
    return;
  }
  const int thId = ph->threadId;

  // On a machine with several NUMA nodes this thread belongs next to its GPU: the pinned ring it reads (allocated by the
  // engine it is about to create: first touch) was written over that node's PCIe root.  KNG_TABLE_PIN=0 leaves it alone.
  {
    const char *mode = getenv("KNG_TABLE_PIN");
    if (!(mode && (!strcmp(mode, "0") || !strcmp(mode, "off")))) {
      const std::vector<cpu_set_t> nodes = kng_placement::numa_node_cpus();
      int usable = 0;
      for (const cpu_set_t &c : nodes) usable += CPU_COUNT(&c) > 0;
      const int node = kng_device_numa_node(ph->gpuId);
      if (usable > 1 && kng_placement::node_usable(nodes, node)) (void)kng_placement::pin_this_thread(nodes[(size_t)node], "GPU");
    }
  }

  GPUEngine *gpu = new GPUEngine(ph->gridSizeX, ph->gridSizeY, ph->gpuId, 65536 * 2);
  kng_engine *eng = kng_shim_engine(gpu);

  if (keyIdx == 0) ::printf("GPU: %s (%.1f MB used)\n", gpu->deviceName.c_str(), gpu->GetMemory() / 1048576.0);
  if (eng == NULL) { // the constructor has said why; this thread will never find anything
    delete gpu;
    ph->hasStarted = true;
    ph->isRunning = false;
    return;
  }

  const double t0 = Timer::get_tick();

  Int *wildOffset = &rangeWidthDiv2;
  const uint64_t woff[4] = {wildOffset->bits64[0], wildOffset->bits64[1], wildOffset->bits64[2], wildOffset->bits64[3]};
  gpu->SetWildOffset(wildOffset);
  gpu->SetParams(dMask, jumpDistance, jumpPointx, jumpPointy);

  // ---- the herd --------------------------------------------------------------------------------------------------------
  // Kangaroo::CreateHerd on the host takes 20 s for the 2^23 kangaroos of one MI355X (one scalar multiplication each, per GPU
  // and per key of the input file).  The engine builds the herd itself (kng_build_herd: SURVEY 8 f2; same law -- distances
  // uniform in [0, 2^rangePower), wild ones shifted by -N/2, alternating types from TAME -- from a seed drawn from the
  // program's own generator, 6 ms of kernel time).  KNG_HOST_HERD=1 keeps the reference's host path.
  auto deviceHerd = [&]() -> bool {
    LOCK(ghMutex); // the generator is shared (CreateHerd takes the same lock, Kangaroo.cpp:683)
    const uint64_t seed = ((uint64_t)rndl() << 32) ^ (uint64_t)rndl() ^ ((uint64_t)ph->gpuId << 56);
    UNLOCK(ghMutex);
    const uint32_t windows = (uint32_t)(rangePower + 7) / 8;
    vector<uint64_t> table((size_t)windows * 256 * 8);
    uint64_t bt[8], bw[8], fin[8];
    if (rangePower >= 1 && rangePower <= 128 &&
        kngh_herd_params(rangePower, woff, keyToSearch.x.bits64, keyToSearch.y.bits64, seed, table.data(), bt, bw, fin) == 0 &&
        kng_build_herd(eng, rangePower, seed, table.data(), windows, bt, bw, fin) == KNG_OK)
      return true;
    ::fprintf(stderr, "SolveKeyGPU_kng: herd creation on the device failed (%s); creating the kangaroos on the host\n", kng_last_error());
    return false;
  };
  // the 3 x N `Int` of the reference's interface: only the host-side herd, KNG_REF_SAVE and KNG_SAVE_VERIFY still use them
  auto needArrays = [&]() {
    Int **a[3] = {&ph->px, &ph->py, &ph->distance};
    for (Int **q : a)
      if (*q == NULL) *q = new Int[ph->nbKangaroo];
  };
  auto dropArrays = [&]() {
    Int **a[3] = {&ph->px, &ph->py, &ph->distance};
    for (Int **q : a) safe_delete_array(*q);
  };
  auto hostHerd = [&]() { // Kangaroo.cpp:529-541: one block of GPU_GRP_SIZE per GPU thread, tame first
    const uint64_t nbThread = gpu->GetNbThread();
    needArrays();
    for (uint64_t i = 0; i < nbThread; i++)
      CreateHerd(GPU_GRP_SIZE, &(ph->px[i * GPU_GRP_SIZE]), &(ph->py[i * GPU_GRP_SIZE]), &(ph->distance[i * GPU_GRP_SIZE]), TAME);
    gpu->SetKangaroos(ph->px, ph->py, ph->distance);
  };
  auto createHerd = [&]() {
    if (keyIdx == 0) ::printf("SolveKeyGPU Thread GPU#%d: creating kangaroos...\n", ph->gpuId);
    if (getenv("KNG_HOST_HERD") || !deviceHerd()) hostHerd();
  };
  // The thread's share of a work file (Backup_kng.cpp left its place instead of 3 x N Int): the file's bytes go to the device
  // in 24 MB pieces and one kernel turns them into herd state, wild offset added mod n (GPUEngine.cu:381-441 on the host).
  auto uploadRecords = [&](const kng_save::Restore &rs) -> bool {
    const uint64_t piece = 1u << 18;
    uint8_t *buf = (uint8_t *)kng_alloc_pinned(piece * 96);
    const bool pinned = buf != NULL;
    if (!buf) buf = (uint8_t *)malloc(piece * 96);
    const int fd = buf ? ::open(rs.file.c_str(), O_RDONLY) : -1;
    bool ok = fd >= 0;
    if (!ok) ::fprintf(stderr, "SolveKeyGPU_kng GPU#%d: cannot read the kangaroos of %s: %s\n", ph->gpuId, rs.file.c_str(), buf ? strerror(errno) : "out of memory");
    for (uint64_t first = 0; ok && first < rs.count; first += piece) {
      const uint64_t m = rs.count - first < piece ? rs.count - first : piece;
      size_t got = 0;
      while (got < m * 96) {
        const ssize_t r = ::pread(fd, buf + got, m * 96 - got, (off_t)(rs.offset + first * 96 + got));
        if (r <= 0) break;
        got += (size_t)r;
      }
      if (got != m * 96) {
        ::fprintf(stderr, "SolveKeyGPU_kng GPU#%d: %s is shorter than its header says\n", ph->gpuId, rs.file.c_str());
        ok = false;
      } else if (kng_snapshot_write(eng, first, m, buf) != KNG_OK) {
        ::fprintf(stderr, "SolveKeyGPU_kng GPU#%d: %s\n", ph->gpuId, kng_last_error());
        ok = false;
      }
    }
    uint64_t bad = 0;
    if (ok && kng_snapshot_restore(eng, 0, rs.count, woff, &bad) != KNG_OK) {
      ::fprintf(stderr, "SolveKeyGPU_kng GPU#%d: %s\n", ph->gpuId, kng_last_error());
      ok = false;
    }
    if (fd >= 0) ::close(fd);
    if (pinned) kng_free_pinned(buf);
    else free(buf);
    return ok;
  };

  if (ph->px != NULL) { // the reference's FectchKangaroos handed the kangaroos in (KNG_REF_SAVE=1)
    gpu->SetKangaroos(ph->px, ph->py, ph->distance);
  } else {
    kng_save::Restore rs;
    const bool plan = kng_save::take_restore(ph, rs);
    if (!plan || rs.count < ph->nbKangaroo) createHerd(); // all of it, or the part the file does not have (Backup.cpp:224-229)
    if (plan && !uploadRecords(rs)) {
      ::fprintf(stderr, "SolveKeyGPU_kng GPU#%d: the saved kangaroos are not used\n", ph->gpuId);
      createHerd();
    }
  }
  dropArrays(); // nobody asks for the kangaroos back through them any more (KNG_REF_SAVE / KNG_SAVE_VERIFY allocate them when needed)

  gpu->callKernel();

  const double t1 = Timer::get_tick();

  if (keyIdx == 0) ::printf("SolveKeyGPU Thread GPU#%d: 2^%.2f kangaroos [%.1fs]\n", ph->gpuId, log2((double)ph->nbKangaroo), (t1 - t0));

  kng_save::attach(ph, eng);
  uint64_t handled = kng_save::requested.load(); // save generations this thread has dealt with
  ph->isWaiting = false; // (a thread of the previous key may have ended at a save point: TH_PARAM is reused, Kangaroo.cpp:1041-1047)
  ph->hasStarted = true;

  {
    int tableThreads = 4;
    if (const char *e = getenv("KNG_TABLE_THREADS")) tableThreads = atoi(e);
    const int gpus = nbGPUThread > 0 ? nbGPUThread : 1;
    const int roomFor = ((int)(kng_effective_cpus() + 0.5) - gpus - nbCPUThread) / gpus; // the GPU threads and the program's CPU walkers come first
    if (tableThreads > roomFor) tableThreads = roomFor;
    if (tableThreads < 1) tableThreads = 1;
    // the table threads of ALL GPU threads form one pool, each owning its share of the buckets (kng_ingest.h): every GPU thread
    // computes the same size, whoever comes first creates it
    const int poolThreads = tableThreads * gpus;
    const uint64_t off[2] = {woff[0], woff[1]};
    // points per launch decide how many chunks "64 launches" are
    int bits = 0;
    for (uint64_t mk = dMask; mk; mk &= mk - 1) bits++;
    const uint64_t perLaunch = bits >= 64 ? 0 : (ph->nbKangaroo * NB_RUN) >> bits;
    size_t maxChunks = QUEUE_LAUNCHES * (size_t)(perLaunch / CHUNK + 1);
    if (maxChunks > QUEUE_MAX_CHUNKS) maxChunks = QUEUE_MAX_CHUNKS;
    // while a work file's table section is being written the points wait: room for them (default 4 GiB per GPU thread)
    size_t holdChunks = 8192;
    if (const char *e = getenv("KNG_SAVE_QUEUE_MB")) holdChunks = (size_t)(atof(e) * 1048576.0 / sizeof(Chunk)) + 1;
    Ingest ingest(&hashTable, off, poolThreads, maxChunks);
    if (keyIdx == 0) // what the table side can take, next to what the kernel will offer (INTEGRATION.md has the table of -d)
      ::printf("SolveKeyGPU Thread GPU#%d: 2^%.1f points per launch at DP %d into a pool of %d table thread%s (%d per GPU thread, "
               "KNG_TABLE_THREADS; one thread takes ~10 M points/s)%s\n",
               ph->gpuId, perLaunch ? log2((double)perLaunch) : 0.0, bits, ingest.threads(), ingest.threads() == 1 ? "" : "s", tableThreads,
               ingest.nodes_used() > 1 ? ", spread over the NUMA nodes (KNG_TABLE_PIN)" : "");

    const bool refSave = getenv("KNG_REF_SAVE") != NULL, verifySave = getenv("KNG_SAVE_VERIFY") != NULL;
    vector<Event> events;
    unordered_map<uint64_t, uint64_t> resetAt; // kIdx -> last launch that still walked the kangaroo a reset replaced
    uint64_t launches = 0, lostTotal = 0, nEvents = 0, staleEvents = 0, saves = 0;
    double blocked = 0, waitGpu = 0, savePoint = 0;
    bool lostWarning = false, behindWarning = false, saving = false;
    const double loop0 = Timer::get_tick();
    // KNG_STATS=1: one line when the loop ends; KNG_STATS=<seconds> (> 1): also a "(running)" line every so many seconds
    const double statsEvery = getenv("KNG_STATS") ? atof(getenv("KNG_STATS")) : 0.0;
    double statsNext = loop0 + statsEvery;
    auto report = [&](const char *state) {
      const double wall = Timer::get_tick() - loop0;
      const Ingest::Totals tt = ingest.totals();
      ::fprintf(stderr,
                "\nSolveKeyGPU_kng GPU#%d%s: %" PRIu64 " launches in %.3f s = %.1f MK/s; points %" PRIu64 " (lost %" PRIu64 "), events %" PRIu64
                " (+%" PRIu64 " stale); GPU thread waited %.3f s for kernels, %.3f s for queue room, %.3f s at %" PRIu64
                " save points; pool of %d table threads busy %.3f s with this GPU's points (%.0f ns/point), queue high water %zu of %zu chunks\n",
                ph->gpuId, state, launches, wall, wall > 0 ? (double)launches * (double)ph->nbKangaroo * NB_RUN / wall / 1e6 : 0.0, tt.points, lostTotal,
                nEvents, staleEvents, waitGpu, blocked, savePoint, saves, ingest.threads(), tt.busy_s, tt.points ? tt.busy_s / (double)tt.points * 1e9 : 0.0,
                tt.high_water, maxChunks);
    };

    while (!endOfSearch) {
      // the launch in flight, then the next one at once: from here on the GPU is busy again while the host works
      double tw = Timer::get_tick();
      if (kng_wait(eng, 0) != KNG_OK) {
        ::printf("GPUEngine: Launch: %s\n", kng_last_error());
        break;
      }
      waitGpu += Timer::get_tick() - tw;

      // a new save request, seen at a launch boundary: freeze the herd here, before the next kernel moves it
      const uint64_t req = kng_save::requested.load();
      const bool savePointNow = saveRequest && !endOfSearch && req != handled && !saving;
      const double ts = Timer::get_tick();
      if (savePointNow && saveKangaroo) {
        if (refSave || verifySave) { // the reference's way, for comparison: 3 x N Int through GPUEngine::GetKangaroos
          needArrays();
          gpu->GetKangaroos(ph->px, ph->py, ph->distance);
        }
        if (!refSave) {
          if (kng_snapshot(eng, woff) == KNG_OK) {
            kng_save::snapshot_taken(ph, req);
          } else { // (no room for the second buffer?)  This save goes the reference's way: SaveWork writes from the arrays
            ::fprintf(stderr, "SolveKeyGPU_kng GPU#%d: %s; saving through GetKangaroos\n", ph->gpuId, kng_last_error());
            if (!verifySave) {
              needArrays();
              gpu->GetKangaroos(ph->px, ph->py, ph->distance);
            }
          }
        }
      }

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
      const uint64_t waited = launches++; // number of the launch these points come from; launch `waited + 1` is running
      counters[thId] += ph->nbKangaroo * NB_RUN;
      blocked += ingest.push(recs, nb, waited); // the view is only good until the launch after next: copy now
      if (blocked > 1.0 && !behindWarning && !saving) {
        // nothing is lost -- the GPU waits -- but the user should know why the rate is below the kernel's (INTEGRATION.md has
        // the table of -d against table threads)
        ::printf("\nWarning, the distinguished-point table cannot keep up with GPU#%d (%d table threads): the GPU waits\n"
                 "Hint: increase dp (-d), or give the table more threads (KNG_TABLE_THREADS) if the machine has CPUs to spare\n",
                 ph->gpuId, ingest.threads());
        behindWarning = true;
      }

      if (savePointNow) {
        // the table must hold every point the frozen kangaroos have passed, and none they have not: everything drained so far
        // goes in, then the table threads stand still until the table section is written (they resume by themselves)
        ingest.flush();
        ingest.hold(req, &kng_save::finished, holdChunks);
        handled = req;
        saving = true;
        saves++;
        ph->isWaiting = true;
        savePoint += Timer::get_tick() - ts;
      } else if (saving && kng_save::finished.load() >= handled) {
        ph->isWaiting = false;
        saving = false;
      }

      // what the table threads could not simply store (Kangaroo.cpp:594-612, AddToTable :306-314)
      ingest.take_events(events);
      if (!events.empty()) {
        LOCK(ghMutex);
        for (size_t g = 0; !endOfSearch && g < events.size(); g++) {
          const Event &ev = events[g];
          const uint32_t kType = (uint32_t)(ev.rec.kidx % 2);
          // Events arrive late (the queue), the reference's arrive before the kangaroo moves again: a kangaroo that merged
          // into another path keeps producing that path's points until its replacement is on the device.  Points of launches
          // that still walked the replaced kangaroo say nothing about the new one.
          const auto was = resetAt.find(ev.rec.kidx);
          if (was != resetAt.end() && ev.rec.reserved <= was->second) {
            staleEvents++;
            continue;
          }
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
            gpu->SetKangaroo(ev.rec.kidx, &px, &py, &d); // stream-ordered behind the running launch `waited + 1`
            resetAt[ev.rec.kidx] = waited + 1;
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
    }

    ph->isWaiting = false;
    kng_save::detach(ph); // (waits while a saver is still reading this engine's snapshot)
    if (getenv("KNG_STATS")) report("");
  } // ~Ingest: table threads joined, whatever was still queued is dropped (the search is over)

  dropArrays();
  delete gpu;

  ph->isRunning = false;
#endif
}
