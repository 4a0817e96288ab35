// Stand-alone C++ use of our GPUEngine class (kangaroo_amd/host/GPUEngine.h + Int.h) -- no reference
// code involved.  Follows the protocol of Check.cpp:492-612: SetParams / SetWildOffset / SetKangaroos,
// single-kangaroo overwrite, Launch; GetKangaroos; Launch, then every kangaroo is verified through the
// group invariant (x,y) == d*G (tame) / K + d*G (wild) with the host library, and every DP must carry
// the x and true distance of the kangaroo that reported it.   Prints "CPP GPUEngine ok" on success.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "GPUEngine.h"
#include "kng_host.h"

static bool on_track(Int &x, Int &y, Int &d, bool wild, const uint64_t kx[4], const uint64_t ky[4]) {
  uint64_t gx[4], gy[4], rx[4], ry[4];
  if (kngh_pubkey(d.bits64, gx, gy) != 0) return false;
  if (wild) {
    if (kngh_point_add(kx, ky, gx, gy, rx, ry) != 0) return false;
  } else {
    memcpy(rx, gx, 32);
    memcpy(ry, gy, 32);
  }
  return !memcmp(rx, x.bits64, 32) && !memcmp(ry, y.bits64, 32);
}

int main() {
  const int gx = 4, gy = 64, rp = 72, dp = 6;
  int x = 0, y = 0;
  if (!GPUEngine::GetGridSize(0, &x, &y) || x <= 0 || y != 128) { printf("GetGridSize failed\n"); return 1; }
  GPUEngine eng(gx, gy, 0, 65536);
  const uint64_t n = (uint64_t)eng.GetNbThread() * eng.GetGroupSize();
  if (n != (uint64_t)gx * gy * 128 || eng.GetGroupSize() != 128 || eng.GetMemory() <= 0) { printf("geometry\n"); return 1; }

  uint64_t jd[64], jx[128], jy[128];
  kngh_jump_table(rp, jd, jx, jy);
  Int jdist[32], jpx[32], jpy[32];
  for (int i = 0; i < 32; i++) {
    memcpy(jdist[i].bits64, jd + 2 * i, 16);
    memcpy(jpx[i].bits64, jx + 4 * i, 32);
    memcpy(jpy[i].bits64, jy + 4 * i, 32);
  }
  uint64_t key[4] = {0x123456789ABCDEFULL, 0x42, 0, 0}, kx[4], ky[4];
  kngh_pubkey(key, kx, ky);
  Int woff;
  woff.SetBase16("7FFFFFFFFFFFFFFFFF"); // (2^72 - 1) >> 1
  std::vector<uint64_t> hx(4 * n), hy(4 * n), hd(4 * n);
  if (kngh_create_herd(n, rp, woff.bits64, kx, ky, 0, 77, 4, hx.data(), hy.data(), hd.data()) != 0) return 1;
  std::vector<Int> px(n), py(n), pd(n);
  for (uint64_t i = 0; i < n; i++) {
    memcpy(px[i].bits64, &hx[4 * i], 32);
    memcpy(py[i].bits64, &hy[4 * i], 32);
    memcpy(pd[i].bits64, &hd[4 * i], 32);
  }
  eng.SetParams(kngh_dp_mask(dp), jdist, jpx, jpy);
  eng.SetWildOffset(&woff);
  eng.SetKangaroos(px.data(), py.data(), pd.data());
  // single overwrite (Check.cpp:521-524): move kangaroo r to another valid spot of the same type
  const uint64_t r = 4099; // odd: wild
  eng.SetKangaroo(r, &px[r - 2], &py[r - 2], &pd[r - 2]);

  std::vector<ITEM> found;
  eng.Launch(found);
  if (!found.empty()) { printf("first Launch must return nothing\n"); return 1; }
  std::vector<Int> gxs(n), gys(n), gds(n);
  eng.GetKangaroos(gxs.data(), gys.data(), gds.data());
  eng.Launch(found);
  printf("DP found: %zu (expected about %llu)\n", found.size(), (unsigned long long)(n * 64 >> dp));
  if (found.size() < (n * 64 >> dp) * 8 / 10 || found.size() > (n * 64 >> dp) * 12 / 10) return 1;

  uint64_t bad = 0;
  for (uint64_t i = 0; i < n; i += 7) // sample every 7th kangaroo
    if (!on_track(gxs[i], gys[i], gds[i], i & 1, kx, ky)) bad++;
  if (!on_track(gxs[r], gys[r], gds[r], true, kx, ky)) bad++;
  if (gds[r].IsEqual(&gds[r - 2]) == false || !gxs[r].IsEqual(&gxs[r - 2])) { printf("overwritten kangaroo must shadow its source\n"); bad++; }
  for (size_t i = 0; i < found.size(); i += 5) {
    ITEM &it = found[i];
    if (it.kIdx >= n || (it.x.bits64[3] & kngh_dp_mask(dp)) != 0) { bad++; continue; }
    uint64_t qx[4], qy[4], sx[4], sy[4];
    kngh_pubkey(it.d.bits64, qx, qy);
    if (it.kIdx & 1) { kngh_point_add(kx, ky, qx, qy, sx, sy); memcpy(qx, sx, 32); }
    if (memcmp(qx, it.x.bits64, 32)) bad++;
  }
  if (bad) { printf("CPP GPUEngine NOT ok: %llu faults\n", (unsigned long long)bad); return 1; }
  printf("GPU: %s\n", eng.deviceName.c_str());

  // error behaviour at the boundary (GPUEngine.cu:144-253, :540-557): a constructor that cannot allocate its herd
  // (2^33 kangaroos = 960 GB of state) reports it and leaves an object whose calls are refused -- callKernel / Launch
  // return false -- and the process lives on.
  {
    GPUEngine big(1 << 16, 1024, 0, 65536);
    std::vector<ITEM> none;
    if (big.callKernel() || big.callKernelAndWait() || big.Launch(none) || !none.empty() || big.GetMemory() != 0) {
      printf("an unusable engine must refuse to launch\n");
      return 1;
    }
    big.SetParams(kngh_dp_mask(dp), jdist, jpx, jpy); // refused with a message, no crash
    big.SetKangaroo(0, &px[0], &py[0], &pd[0]);
  }
  {
    GPUEngine nodev(2, 64, 9999, 65536); // no such device
    if (nodev.callKernel()) { printf("an engine without a device must refuse to launch\n"); return 1; }
  }
  // calls out of sequence on a healthy engine are refused one by one and do not put it out of service
  {
    GPUEngine fresh(2, 64, 0, 4096);
    if (fresh.callKernel()) { printf("launch without parameters must fail\n"); return 1; }
  }
  printf("CPP GPUEngine ok\n");
  return 0;
}
