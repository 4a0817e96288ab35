// GPUEngine.cpp -- `class GPUEngine` (reference boundary GPU/GPUEngine.h:40-64) implemented over
// the C ABI of libkangaroo_hip.so.  Replaces the host half of GPU/GPUEngine.cu.
//
// Two build modes, one source:
//   default                 : stand-alone, with our GPUEngine.h / Int.h
//   -DKNG_REFERENCE_HEADER  : compiled against the REFERENCE's own GPU/GPUEngine.h and SECPK1/Int.h
//                             (-I<reference root>); this object then replaces GPU/GPUEngine.o in the
//                             reference's link line and the unmodified program runs on the MI355X.
//                             The reference header's private fields are CUDA-era buffers we do not
//                             need; the engine handle is parked in `inputKangaroo`.
// Errors: like the reference (GPUEngine.cu:144-253) a failure is reported on stderr and leaves the object unusable
// -- `initialised` cleared, every later call refused with a message, callKernel / callKernelAndWait / Launch return
// false -- but nothing in here ends the process: that decision belongs to the program that owns the object.
#ifdef KNG_REFERENCE_HEADER
#include "GPU/GPUEngine.h"
#include "kangaroo_hip.h"
#define ENGINE (*reinterpret_cast<kng_engine **>(&this->inputKangaroo))
#define ITEMBUF (*reinterpret_cast<kng_item **>(&this->outputItemPinned))
#else
#include "GPUEngine.h"
#define ENGINE (this->engine)
#define ITEMBUF (this->itemBuf)
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>

// The C handle behind a GPUEngine object, for hosts that drive the hot loop through the C ABI (SolveKeyGPU_kng.cpp: kng_wait /
// kng_launch / kng_drain_view instead of Launch's per-item copies) while leaving construction, SetParams, Set/GetKangaroos and
// SetKangaroo to the class.  The reference's header has no accessor to add one to, so the objects are kept in a small list.
static std::mutex g_engines_lock;
static std::vector<std::pair<GPUEngine *, kng_engine *>> g_engines;
static void remember_engine(GPUEngine *g, kng_engine *h) {
  std::lock_guard<std::mutex> l(g_engines_lock);
  for (auto &e : g_engines)
    if (e.first == g) {
      e.second = h;
      return;
    }
  g_engines.emplace_back(g, h);
}
static void forget_engine(GPUEngine *g) {
  std::lock_guard<std::mutex> l(g_engines_lock);
  for (size_t i = 0; i < g_engines.size(); i++)
    if (g_engines[i].first == g) {
      g_engines.erase(g_engines.begin() + (long)i);
      return;
    }
}
extern "C" kng_engine *kng_shim_engine(GPUEngine *g) {
  std::lock_guard<std::mutex> l(g_engines_lock);
  for (auto &e : g_engines)
    if (e.first == g) return e.second;
  return NULL;
}

static const size_t INT_STRIDE = sizeof(Int) / sizeof(uint64_t);

// report a failed C-ABI call and put the object out of service; evaluates to true on success
static bool in_service(bool initialised, const void *engine, const char *where) {
  if (initialised && engine) return true;
  fprintf(stderr, "GPUEngine: %s refused: the engine is out of service after an earlier error\n", where);
  return false;
}
#define usable(where) in_service(initialised, ENGINE, where)
#define KNG_MUST(call, where) \
  (!usable(where) ? false : ((call) == KNG_OK ? true : (fprintf(stderr, "GPUEngine: %s: %s\n", where, kng_last_error()), initialised = false, false)))

GPUEngine::GPUEngine(int nbThreadGroup, int nbThreadPerGroup, int gpuId, uint32_t maxFound) {
#ifdef KNG_REFERENCE_HEADER
  inputKangaroo = NULL;
  inputKangarooPinned = NULL;
  outputItem = NULL;
  outputItemPinned = NULL;
  jumpPinned = NULL;
  dpMask = 0;
#endif
  initialised = false;
  this->nbThreadPerGroup = nbThreadPerGroup;
  this->nbThread = nbThreadGroup * nbThreadPerGroup;
  this->maxFound = maxFound;
  this->lostWarning = false;
  wildOffset.SetInt32(0);
  ENGINE = NULL;
  ITEMBUF = NULL;
  deviceName = "GPU #" + std::to_string(gpuId) + " (unusable)";
  kng_engine *h = NULL;
  if (kng_create(gpuId, nbThreadGroup, nbThreadPerGroup, maxFound, &h) != KNG_OK) {
    fprintf(stderr, "GPUEngine: kng_create: %s\n", kng_last_error());
    return;
  }
  ENGINE = h;
  // host landing buffer for one launch's DPs, allocated once (pinned)
  ITEMBUF = (kng_item *)kng_alloc_pinned((size_t)maxFound * sizeof(kng_item));
  if (!ITEMBUF) {
    fprintf(stderr, "GPUEngine: kng_alloc_pinned: %s\n", kng_last_error());
    return;
  }
  char name[256] = "", arch[64] = "";
  int cu = 0;
  kng_device_info(gpuId, name, sizeof name, &cu, NULL, arch, sizeof arch);
  char tmp[512];
  // GPUEngine.cu:176-182 banner; the CUDA "cores per SM" table has no AMD entry (it would print 0)
  snprintf(tmp, sizeof tmp, "GPU #%d %s (%dx%d cores) Grid(%dx%d)", gpuId, name, cu, 64, nbThreadGroup, nbThreadPerGroup);
  deviceName = std::string(tmp);
  initialised = true;
  remember_engine(this, h);
}

GPUEngine::~GPUEngine() {
  forget_engine(this);
  kng_destroy(ENGINE); // waits for an in-flight kernel (Kangaroo.cpp:572-634 deletes mid-flight)
  ENGINE = NULL;
  kng_free_pinned(ITEMBUF);
  ITEMBUF = NULL;
}

void GPUEngine::SetWildOffset(Int *offset) { wildOffset.Set(offset); }

int GPUEngine::GetNbThread() { return nbThread; }
int GPUEngine::GetGroupSize() { return KNG_GRP_SIZE; }

int GPUEngine::GetMemory() {
  if (!ENGINE) return 0;
  // the reference returns int and overflows above 2 GiB (GPUEngine.h:79): saturate
  uint64_t b = kng_memory_bytes(ENGINE);
  return b > 0x7FFFFFFFULL ? 0x7FFFFFFF : (int)b;
}

bool GPUEngine::GetGridSize(int gpuId, int *x, int *y) {
  if (kng_default_grid(gpuId, x, y) != KNG_OK) {
    printf("GPUEngine: %s\n", kng_last_error());
    return false;
  }
  return true;
}

void *GPUEngine::AllocatePinnedMemory(size_t size) {
  void *p = kng_alloc_pinned(size);
  if (!p) printf("GPUEngine: AllocatePinnedMemory: %s\n", kng_last_error());
  return p;
}
void GPUEngine::FreePinnedMemory(void *buff) { kng_free_pinned(buff); }

void GPUEngine::PrintCudaInfo() {
  int n = kng_device_count();
  if (n == 0) {
    printf("GPUEngine: There are no available device(s) that support HIP\n");
    return;
  }
  for (int i = 0; i < n; i++) {
    char name[256] = "", arch[64] = "";
    int cu = 0;
    uint64_t mem = 0;
    if (kng_device_info(i, name, sizeof name, &cu, &mem, arch, sizeof arch) != KNG_OK) continue;
    printf("GPU #%d %s (%dx%d cores) (%s) (%.1f MB) (Multiple host threads)\n", i, name, cu, 64, arch, (double)mem / 1048576.0);
  }
}

void GPUEngine::SetParams(uint64_t dpMask, Int *distance, Int *px, Int *py) {
#ifdef KNG_REFERENCE_HEADER
  this->dpMask = dpMask;
#endif
  uint64_t jd[KNG_NB_JUMP][2], jx[KNG_NB_JUMP][4], jy[KNG_NB_JUMP][4];
  for (int i = 0; i < KNG_NB_JUMP; i++) {
    memcpy(jd[i], distance[i].bits64, 16); // 128-bit jump distances (GPUEngine.cu:563-565)
    memcpy(jx[i], px[i].bits64, 32);
    memcpy(jy[i], py[i].bits64, 32);
  }
  if (!KNG_MUST(kng_set_params(ENGINE, dpMask, &jd[0][0], &jx[0][0], &jy[0][0]), "SetParams")) return;
  // `maxFound` is a floor, not a ceiling.  The program passes a constant sized for a V100 herd (65536*2, Kangaroo.cpp:523;
  // 65536 in Check.cpp:470).  The herd and the mask say what a launch will really yield: herd x NB_RUN jumps, one in
  // 2^(bits of the mask) distinguished.  At the MI355X default grid (2^23 kangaroos) and the DP size the program suggests for
  // eight GPUs (11) that is 262 144 points per launch -- the constant would drop half of them, every launch, with one
  // "items lost" warning (GPUEngine.cu:641-648).  Capacity = max(maxFound, 2 x expected + 4096); the rings of the engine and
  // this object's landing buffer grow accordingly.  Only possible between launches: a SetParams that arrives while a kernel
  // runs (allowed, see kng_set_params) keeps the capacity it has.
  int bits = 0;
  for (uint64_t m = dpMask; m; m &= m - 1) bits++;
  const uint64_t herd = kng_nb_kangaroos(ENGINE);
  const uint64_t expected = bits >= 64 ? 0 : (herd * KNG_NB_RUN) >> bits;
  uint64_t want = 2 * expected + 4096;
  // ... up to 2^22 points per launch (268 MB per pinned ring, 235 MB for this object's landing buffer): enough for every DP size
  // from 8 up at the 2^23 herd.  A mask that asks for more (-d 7 and below at that herd: 8 M points per launch) keeps the
  // reference's behaviour beyond that -- points above the capacity are dropped with the one-time warning.
  if (want > (1ULL << 22)) want = 1ULL << 22;
  if (want > maxFound && !kng_outstanding(ENGINE) && !kng_undrained(ENGINE)) {
    kng_item *bigger = (kng_item *)kng_alloc_pinned((size_t)want * sizeof(kng_item));
    if (bigger && kng_reserve_points(ENGINE, (uint32_t)want) == KNG_OK) {
      kng_free_pinned(ITEMBUF);
      ITEMBUF = bigger;
      maxFound = (uint32_t)want;
    } else {
      fprintf(stderr, "GPUEngine: SetParams: could not raise the DP capacity to %llu points per launch (%s); keeping %u\n",
              (unsigned long long)want, kng_last_error(), maxFound);
      kng_free_pinned(bigger);
    }
  }
}

void GPUEngine::SetKangaroos(Int *px, Int *py, Int *d) {
  if (!usable("SetKangaroos")) return;
  const uint64_t n = kng_nb_kangaroos(ENGINE);
  // device distances: wild (odd index) += wildOffset mod n (GPUEngine.cu:406-411)
  std::vector<uint64_t> dd(2 * n);
  for (uint64_t i = 0; i < n; i++) {
    Int dOff;
    dOff.Set(&d[i]);
    if (i % 2 == WILD) dOff.ModAddK1order(&wildOffset);
    dd[2 * i] = dOff.bits64[0];
    dd[2 * i + 1] = dOff.bits64[1];
  }
  (void)KNG_MUST(kng_set_kangaroos(ENGINE, px[0].bits64, INT_STRIDE, py[0].bits64, INT_STRIDE, dd.data(), 2, n), "SetKangaroos");
}

void GPUEngine::GetKangaroos(Int *px, Int *py, Int *d) {
  if (!usable("GetKangaroos")) return;
  const uint64_t n = kng_nb_kangaroos(ENGINE);
  std::vector<uint64_t> dd(2 * n);
  if (!KNG_MUST(kng_get_kangaroos(ENGINE, px[0].bits64, INT_STRIDE, py[0].bits64, INT_STRIDE, dd.data(), 2, n), "GetKangaroos")) return;
  for (uint64_t i = 0; i < n; i++) {
    px[i].bits64[4] = 0;
    py[i].bits64[4] = 0;
    Int dOff;
    dOff.SetInt32(0);
    dOff.bits64[0] = dd[2 * i];
    dOff.bits64[1] = dd[2 * i + 1];
    if (i % 2 == WILD) dOff.ModSubK1order(&wildOffset); // GPUEngine.cu:477
    d[i].Set(&dOff);
  }
}

void GPUEngine::SetKangaroo(uint64_t kIdx, Int *px, Int *py, Int *d) {
  Int dOff;
  dOff.Set(d);
  if (kIdx % 2 == WILD) dOff.ModAddK1order(&wildOffset); // GPUEngine.cu:526
  (void)KNG_MUST(kng_set_kangaroo(ENGINE, kIdx, px->bits64, py->bits64, dOff.bits64), "SetKangaroo");
}

bool GPUEngine::callKernel() {
  if (!usable("callKernel")) return false;
  if (kng_launch(ENGINE) != KNG_OK) {
    printf("GPUEngine: Kernel: %s\n", kng_last_error());
    return false;
  }
  return true;
}

bool GPUEngine::callKernelAndWait() {
  if (!callKernel()) return false;
  if (kng_wait(ENGINE, 0) != KNG_OK) {
    printf("GPUEngine: callKernelAndWait: %s\n", kng_last_error());
    return false;
  }
  return true;
}

bool GPUEngine::Launch(std::vector<ITEM> &hashFound, bool spinWait) {
  hashFound.clear();
  if (!usable("Launch")) return false;
  // results of the PREVIOUS kernel (GPUEngine.cu:607-676).  Nothing is outstanding on the very
  // first call of Check.cpp:526; the reference then reads an uninitialised counter.
  const bool had = kng_outstanding(ENGINE) == 1;
  if (had && kng_wait(ENGINE, spinWait ? 1 : 0) != KNG_OK) {
    printf("GPUEngine: Launch: %s\n", kng_last_error());
    return false;
  }
  // start the next kernel BEFORE unpacking, so the DP copy and the host work overlap it
  const bool ok = callKernel();
  if (had) {
    kng_item *items = ITEMBUF;
    uint32_t nb = 0, lost = 0;
    if (kng_drain(ENGINE, items, maxFound, &nb, &lost) != KNG_OK) {
      printf("GPUEngine: Launch: %s\n", kng_last_error());
      return false;
    }
    if (lost && !lostWarning) { // GPUEngine.cu:641-648
      printf("\nWarning, %u items lost\nHint: Search with less threads (-g) or increse dp (-d)\n", lost);
      lostWarning = true;
    }
    hashFound.reserve(nb);
    for (uint32_t i = 0; i < nb; i++) {
      ITEM it;
      it.kIdx = items[i].kidx;
      it.x.SetInt32(0);
      memcpy(it.x.bits64, items[i].x, 32);
      it.d.SetInt32(0);
      it.d.bits64[0] = items[i].d[0];
      it.d.bits64[1] = items[i].d[1];
      if (it.kIdx % 2 == WILD) it.d.ModSubK1order(&wildOffset); // GPUEngine.cu:672
      hashFound.push_back(it);
    }
  }
  return ok;
}
