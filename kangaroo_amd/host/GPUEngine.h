// GPUEngine.h -- the reference's `class GPUEngine` boundary (GPU/GPUEngine.h:40-64), re-created
// over the C ABI of the MI355X jump engine (include/kangaroo_hip.h).
//
// Same class name, method names, argument meaning and call protocol as the reference, so that
// Kangaroo::SolveKeyGPU (Kangaroo.cpp:510-644) and Kangaroo::Check (Check.cpp:467-621) work
// unchanged.  This header is for STAND-ALONE use (with our minimal Int.h); when building the
// reference program itself, its own GPU/GPUEngine.h is used and GPUEngine.cpp is compiled with
// -DKNG_REFERENCE_HEADER (see INTEGRATION.md and oracle/Makefile target `ref_hip`).
#ifndef KNG_GPUENGINE_H
#define KNG_GPUENGINE_H

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/kangaroo_hip.h"
#include "Int.h"

#define GPU_GRP_SIZE KNG_GRP_SIZE // Constants.h:32
#define NB_RUN KNG_NB_RUN         // Constants.h:35
#define NB_JUMP KNG_NB_JUMP       // Constants.h:29
#define TAME 0
#define WILD 1

// one distinguished point as handed to the caller (reference: ITEM, GPU/GPUEngine.h:34-38)
struct ITEM {
  Int x;         // canonical x coordinate
  Int d;         // TRUE distance mod n: the wild offset is already removed (GPUEngine.cu:672)
  uint64_t kIdx; // position in the SetKangaroos arrays; kIdx & 1 = TAME / WILD
};

class GPUEngine {
public:
  // ---- lifetime (GPUEngine.cu:144-263): herd = nbThreadGroup * nbThreadPerGroup * 128 kangaroos ----
  GPUEngine(int nbThreadGroup, int nbThreadPerGroup, int gpuId, uint32_t maxFound);
  ~GPUEngine();

  // ---- walk parameters (GPUEngine.cu:140-142, :559-590) ----
  void SetWildOffset(Int *offset); // N/2; must precede SetKangaroos
  void SetParams(uint64_t dpMask, Int *distance, Int *px, Int *py); // 32 jumps: 128-bit distance + point

  // ---- herd state (GPUEngine.cu:381-538): arrays of GetNbThread()*GetGroupSize() Ints ----
  void SetKangaroos(Int *px, Int *py, Int *d);
  void GetKangaroos(Int *px, Int *py, Int *d);                 // waits for the in-flight launch
  void SetKangaroo(uint64_t kIdx, Int *px, Int *py, Int *d);   // lands after the in-flight launch

  // ---- the hot path (GPUEngine.cu:540-557, :592-679): one kernel = NB_RUN jumps of every kangaroo ----
  bool callKernel();                                            // asynchronous start
  bool callKernelAndWait();                                     // debug helper
  bool Launch(std::vector<ITEM> &hashFound, bool spinWait = false); // DPs of the PREVIOUS kernel, then start the next

  // ---- queries (GPUEngine.cu:266-308, :377-379) ----
  int GetNbThread();
  int GetGroupSize(); // 128
  int GetMemory();    // bytes, saturating at INT_MAX
  std::string deviceName;
  static bool GetGridSize(int gpuId, int *x, int *y); // fills x / y when <= 0: 2*CU, 128
  static void PrintCudaInfo();

  // ---- pinned host memory (GPUEngine.cu:311-327) ----
  static void *AllocatePinnedMemory(size_t size);
  static void FreePinnedMemory(void *buff);

private:
  bool initialised;   // as in the reference (GPUEngine.h:67): cleared by a failed constructor or call
  kng_engine *engine; // the whole device side lives behind the C ABI
  kng_item *itemBuf;  // pinned landing buffer for one launch's DPs
  Int wildOffset;
  int nbThread;
  int nbThreadPerGroup;
  uint32_t maxFound;
  bool lostWarning;
};

#endif
