// Int.h -- minimal 320-bit integer carrier for the stand-alone build of our GPUEngine class.
//
// The reference's GPUEngine interface (GPU/GPUEngine.h:40-64) exchanges `Int` objects
// (SECPK1/Int.h:190-193: five little-endian uint64 limbs in `bits64`).  When our engine is linked
// into the reference program, the reference's own Int is used and this file
// is not compiled.  For stand-alone C++ users we provide just the members the engine boundary
// needs: limb storage, copy, compare, hex I/O and add/sub modulo the group order.
#ifndef KNG_INT_H
#define KNG_INT_H

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "kng_host.h"

#define NB64BLOCK 5

class Int {
public:
  Int() { SetInt32(0); }
  explicit Int(Int *a) { Set(a); }
  void SetInt32(uint32_t v) {
    memset(bits64, 0, sizeof bits64);
    bits64[0] = v;
  }
  void Set(Int *a) { memcpy(bits64, a->bits64, sizeof bits64); }
  bool IsEqual(Int *a) const { return memcmp(bits64, a->bits64, sizeof bits64) == 0; }
  bool IsZero() const { return (bits64[0] | bits64[1] | bits64[2] | bits64[3] | bits64[4]) == 0; }
  // (this + a) mod n / (this - a) mod n, operands in [0,n)  (SECPK1/IntMod.cpp:1245-1263)
  void ModAddK1order(Int *a) {
    kngh_add_order(bits64, a->bits64, bits64);
    bits64[4] = 0;
  }
  void ModSubK1order(Int *a) {
    kngh_sub_order(bits64, a->bits64, bits64);
    bits64[4] = 0;
  }
  void SetBase16(const char *s) {
    SetInt32(0);
    size_t n = strlen(s);
    for (size_t i = 0; i < n && i < 64; i++) {
      char c = s[n - 1 - i];
      uint64_t v = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : 0;
      bits64[i / 16] |= v << (4 * (i % 16));
    }
  }
  std::string GetBase16() const {
    char buf[65];
    snprintf(buf, sizeof buf, "%016llX%016llX%016llX%016llX", (unsigned long long)bits64[3], (unsigned long long)bits64[2],
             (unsigned long long)bits64[1], (unsigned long long)bits64[0]);
    const char *p = buf;
    while (*p == '0' && p[1]) p++;
    return std::string(p);
  }
  union {
    uint32_t bits[NB64BLOCK * 2];
    uint64_t bits64[NB64BLOCK];
  };
};

#endif
