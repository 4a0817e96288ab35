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

typedef struct {
  Int x;
  Int d; // true distance mod n (wild offset already removed, GPUEngine.cu:672)
  uint64_t kIdx;
} ITEM;

class GPUEngine {
public:
  GPUEngine(int nbThreadGroup, int nbThreadPerGroup, int gpuId, uint32_t maxFound);
  ~GPUEngine();
  void SetParams(uint64_t dpMask, Int *distance, Int *px, Int *py);
  void SetKangaroos(Int *px, Int *py, Int *d);
  void GetKangaroos(Int *px, Int *py, Int *d);
  void SetKangaroo(uint64_t kIdx, Int *px, Int *py, Int *d);
  bool Launch(std::vector<ITEM> &hashFound, bool spinWait = false);
  void SetWildOffset(Int *offset);
  int GetNbThread();
  int GetGroupSize();
  int GetMemory();
  bool callKernelAndWait();
  bool callKernel();

  std::string deviceName;

  static void *AllocatePinnedMemory(size_t size);
  static void FreePinnedMemory(void *buff);
  static void PrintCudaInfo();
  static bool GetGridSize(int gpuId, int *x, int *y);

private:
  Int wildOffset;
  int nbThread;
  int nbThreadPerGroup;
  kng_engine *engine; // the whole device side lives behind the C ABI
  kng_item *itemBuf;  // pinned landing buffer for one launch's DPs
  bool lostWarning;
  uint32_t maxFound;
};

#endif
