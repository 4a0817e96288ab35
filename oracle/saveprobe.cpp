/*
 * saveprobe.cpp -- TEST INFRASTRUCTURE: differential probe for the work-file members of the reference program.
 *
 * Linked twice by oracle/Makefile against the reference's objects (compiled -DWITHGPU from /root/reference):
 *     _ref/saveprobe_ref   with the reference's own Backup.o
 *     _ref/saveprobe_kng   with Backup.o's SaveWork / FectchKangaroos weakened and kangaroo_amd/host/Backup_kng.cpp bound
 * Both must write THE SAME BYTES for the same state and read the same kangaroos back (tests/test_backup_class_cpu.py):
 *
 *   saveprobe save <file> <seed> <cpuThreads> <tableAdds> [split]
 *       a Kangaroo with <cpuThreads> parked CPU threads (CPU_GRP_SIZE kangaroos each, pseudo-random 256-bit words: the file
 *       layer moves bytes, it does not look at them) and <tableAdds> table entries -> Kangaroo::SaveWork(count, time, threads, n)
 *       (Backup.cpp:449-563), i.e. header, HashTable::SaveTable, kangaroo section
 *   saveprobe load <file> <cpuThreads>
 *       Kangaroo::LoadWork + Kangaroo::FectchKangaroos (Backup.cpp:149-208, :286-364) -> prints every kangaroo handed to the
 *       threads, nbLoadedWalk, the table's size
 * No GPU is involved: GPU threads' records are covered on the device (tests/test_gpu_snapshot.py, KNG_SAVE_VERIFY=1 in
 * tests/test_gpu_reference_program.py).
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#define private public /* test probe: reach Kangaroo::SaveWork / LoadWork / FectchKangaroos */
#include "Kangaroo.h"
#undef private
#include "SECPK1/SECP256k1.h"
#include "Timer.h"
#include "kng_savework.h" /* declarations only: in saveprobe_ref nothing defines them */

namespace kng_save {
bool take_restore(const void *th_param, Restore &r) __attribute__((weak));
}

static uint64_t g_state;
static uint64_t next64() { /* splitmix64 */
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void fill(Int *v) {
  for (int k = 0; k < 4; k++) v->bits64[k] = next64();
  v->bits64[4] = 0;
}

static Kangaroo *make(Secp256K1 *secp, std::string work, std::string input, bool split) {
  std::string empty;
  return new Kangaroo(secp, 12, false, work, input, 60, true, false, -1.0, 3000, 17403, 3000, empty, empty, split);
}

int main(int argc, char **argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: saveprobe save <file> <seed> <cpuThreads> <tableAdds> [split] | load <file> <cpuThreads>\n");
    return 2;
  }
  Timer::Init();
  Secp256K1 *secp = new Secp256K1();
  secp->Init();
  const std::string mode = argv[1], file = argv[2];

  if (mode == "save") {
    g_state = strtoull(argv[3], NULL, 0);
    const int nth = atoi(argv[4]);
    const uint64_t adds = strtoull(argv[5], NULL, 0);
    const bool split = argc > 6 && !strcmp(argv[6], "split");
    Kangaroo *kg = make(secp, file, "", split);
    kg->rangeStart.SetBase16((char *)"49DCCFD96DC5DF56487436F5A1B18C4F5D34F65DDB48CB5E0000000000000000");
    kg->rangeEnd.SetBase16((char *)"49DCCFD96DC5DF56487436F5A1B18C4F5D34F65DDB48CB5EFFFFFFFFFFFFFFFF");
    Int k1;
    k1.SetBase16((char *)"49DCCFD96DC5DF56487436F5A1B18C4F5D34F65DDB48CB5E7A9B1C2D3E4F5061");
    kg->keysToSearch.clear();
    kg->keysToSearch.push_back(secp->ComputePublicKey(&k1));
    kg->keyIdx = 0;
    kg->SetDP(12);
    kg->nbCPUThread = nth;
    kg->nbGPUThread = 0;
    for (uint64_t i = 0; i < adds; i++) {
      Int x, d;
      fill(&x);
      fill(&d);
      d.bits64[2] = d.bits64[3] = 0;
      d.bits64[1] &= 0x3FFFFFFFFFFFFFFFULL;
      kg->hashTable.Add(&x, &d, (uint32_t)(i & 1));
    }
    std::vector<TH_PARAM> th((size_t)nth);
    memset(th.data(), 0, sizeof(TH_PARAM) * (size_t)nth);
    for (int t = 0; t < nth; t++) {
      th[t].obj = kg;
      th[t].threadId = t;
      th[t].isRunning = true;
      th[t].isWaiting = true; /* parked at the save point, as SolveKeyCPU is (Kangaroo.cpp:485-489) */
      th[t].nbKangaroo = (uint64_t)kg->CPU_GRP_SIZE;
      th[t].px = new Int[kg->CPU_GRP_SIZE];
      th[t].py = new Int[kg->CPU_GRP_SIZE];
      th[t].distance = new Int[kg->CPU_GRP_SIZE];
      for (int g = 0; g < kg->CPU_GRP_SIZE; g++) {
        fill(&th[t].px[g]);
        fill(&th[t].py[g]);
        fill(&th[t].distance[g]);
      }
    }
    kg->SaveWork(0x123456789ABCULL + adds, 4321.5, th.data(), nth);
    printf("\nsaveRequest %d table %" PRIu64 "\n", (int)kg->saveRequest, kg->hashTable.GetNbItem());
    return 0;
  }

  if (mode == "load") {
    const int nth = atoi(argv[3]);
    const uint64_t gpuKang = argc > 4 ? strtoull(argv[4], NULL, 0) : 0; /* one GPU thread with that many kangaroos, after the CPU threads */
    std::string f = file;
    Kangaroo *kg = make(secp, "", file, false);
    if (!kg->LoadWork(f)) {
      printf("LoadWork failed\n");
      return 1;
    }
    kg->nbCPUThread = nth;
    kg->nbGPUThread = gpuKang ? 1 : 0;
    kg->keyIdx = 0;
    kg->InitRange();
    kg->InitSearchKey();
    rseed(0x1234); /* CreateHerd fills what the file does not have */
    kg->totalRW = (uint64_t)nth * (uint64_t)kg->CPU_GRP_SIZE + gpuKang;
    std::vector<TH_PARAM> th((size_t)nth + 1);
    memset(th.data(), 0, sizeof(TH_PARAM) * ((size_t)nth + 1));
    for (int t = 0; t < nth; t++) th[t].nbKangaroo = (uint64_t)kg->CPU_GRP_SIZE;
    th[nth].nbKangaroo = gpuKang;
    kg->FectchKangaroos(th.data());
    if (gpuKang) {
      /* the reference hands the GPU thread 3 x N Int (the tail created when the file is short, Backup.cpp:224-229); ours hands it
       * the place of its records in the file.  Print what the thread would put on the device from the FILE, either way. */
      kng_save::Restore r;
      uint64_t have = 0;
      std::vector<uint64_t> rec;
      if (th[nth].px == NULL && kng_save::take_restore && kng_save::take_restore(&th[nth], r)) {
        FILE *fr = fopen(r.file.c_str(), "rb");
        rec.resize(12 * r.count);
        fseek(fr, (long)r.offset, SEEK_SET);
        have = fread(rec.data(), 96, r.count, fr);
        fclose(fr);
        printf("gpu thread: plan %" PRIu64 " records at byte %" PRIu64 " (%" PRIu64 " read)\n", r.count, r.offset, have);
      } else if (th[nth].px) {
        /* how many of them came from the file: what was left after the CPU threads took theirs */
        printf("gpu thread: arrays\n");
      }
      for (uint64_t g = 0; g < gpuKang; g += 41) {
        if (th[nth].px) {
          printf("G %" PRIu64 " %s %s %s\n", g, th[nth].px[g].GetBase16().c_str(), th[nth].py[g].GetBase16().c_str(), th[nth].distance[g].GetBase16().c_str());
        } else if (g < have) {
          Int x, y, d;
          x.SetInt32(0); y.SetInt32(0); d.SetInt32(0);
          memcpy(x.bits64, &rec[12 * g], 32);
          memcpy(y.bits64, &rec[12 * g + 4], 32);
          memcpy(d.bits64, &rec[12 * g + 8], 32);
          printf("G %" PRIu64 " %s %s %s\n", g, x.GetBase16().c_str(), y.GetBase16().c_str(), d.GetBase16().c_str());
        }
      }
    }
    printf("nbLoadedWalk %" PRId64 " table %" PRIu64 " dp %u count %" PRIu64 " time %.3f\n", (int64_t)kg->nbLoadedWalk, kg->hashTable.GetNbItem(),
           kg->dpSize, kg->offsetCount, kg->offsetTime);
    for (int t = 0; t < nth; t++)
      for (int g = 0; g < kg->CPU_GRP_SIZE; g += 37)
        printf("%d %d %s %s %s\n", t, g, th[t].px ? th[t].px[g].GetBase16().c_str() : "-", th[t].py ? th[t].py[g].GetBase16().c_str() : "-",
               th[t].distance ? th[t].distance[g].GetBase16().c_str() : "-");
    return 0;
  }
  return 2;
}
