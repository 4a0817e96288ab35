/*
 * refprobe.cpp -- golden-vector generator linked against the REFERENCE's own objects.
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/refprobe from the
 * reference sources where they lie under /root/reference (never copied into this repo).
 *
 * It drives the reference's SECPK1 arithmetic (Int::ModMulK1, ModSquareK1, ModSub, ModInv,
 * IntGroup::ModInv, ModAddK1order/ModSubK1order, Int::Rand, Secp256K1::ComputePublicKey,
 * AddDirect) and its Kangaroo::CreateJumpTable / CreateHerd / SetDP, and writes the inputs
 * and outputs as JSON.  tools/make_golden.py runs it and commits the result under
 * tests/golden/.  The oracle (oracle/kng_oracle.c) and the HIP engine are both checked
 * against that file.
 *
 * usage: refprobe <out.json> [seed]
 *        refprobe --hashtable <out.json> [seed]
 */
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define private public /* test probe: reach Kangaroo::CreateJumpTable/CreateHerd/SetDP */
#include "Kangaroo.h"
#undef private
#include "SECPK1/IntGroup.h"
#include "SECPK1/SECP256k1.h"
#include "Timer.h"

static FILE *out;

static std::string hex4(Int &a) {
  char buf[80];
  snprintf(buf, sizeof buf, "%016" PRIx64 "%016" PRIx64 "%016" PRIx64 "%016" PRIx64, a.bits64[3],
           a.bits64[2], a.bits64[1], a.bits64[0]);
  return std::string(buf);
}
static std::string hex5(Int &a) {
  char buf[100];
  snprintf(buf, sizeof buf, "%016" PRIx64 "%016" PRIx64 "%016" PRIx64 "%016" PRIx64 "%016" PRIx64,
           a.bits64[4], a.bits64[3], a.bits64[2], a.bits64[1], a.bits64[0]);
  return std::string(buf);
}
static void set4(Int &a, uint64_t l0, uint64_t l1, uint64_t l2, uint64_t l3) {
  a.SetInt32(0);
  a.bits64[0] = l0;
  a.bits64[1] = l1;
  a.bits64[2] = l2;
  a.bits64[3] = l3;
  a.bits64[4] = 0;
}

static std::vector<Int> edge_values() {
  std::vector<Int> v;
  Int t;
  const uint64_t F = ~0ULL;
  set4(t, 0, 0, 0, 0); v.push_back(t);
  set4(t, 1, 0, 0, 0); v.push_back(t);
  set4(t, 2, 0, 0, 0); v.push_back(t);
  set4(t, 0xFFFFFFFEFFFFFC2EULL, F, F, F); v.push_back(t); /* p-1 */
  set4(t, 0xFFFFFFFEFFFFFC2FULL, F, F, F); v.push_back(t); /* p   (non canonical) */
  set4(t, 0xFFFFFFFEFFFFFC30ULL, F, F, F); v.push_back(t); /* p+1 (non canonical) */
  set4(t, F, F, F, F); v.push_back(t);                     /* 2^256-1 (non canonical) */
  set4(t, 0, 0, 0, 0x8000000000000000ULL); v.push_back(t); /* 2^255 */
  set4(t, F, 0, 0, 0); v.push_back(t);
  set4(t, 0, F, 0, 0); v.push_back(t);
  set4(t, 0x1000003D1ULL, 0, 0, 0); v.push_back(t);
  set4(t, F, F, 0, 0); v.push_back(t);
  set4(t, 0, 0, F, F); v.push_back(t);
  /* GPU/GPUEngine.cu:48 GPU_CHECK operand */
  set4(t, 0x0BE3D7593BE1147CULL, 0x4952AAF512875655ULL, 0x08884CCAACCB9B53ULL, 0x9EAE2E2225044292ULL);
  v.push_back(t);
  return v;
}

static void emit_field_kats(int nrand) {
  std::vector<Int> vals = edge_values();
  size_t nedge = vals.size();
  for (int i = 0; i < nrand; i++) {
    Int r;
    r.Rand(256);
    vals.push_back(r);
  }
  /* modmul / modsqr / modsub over pairs */
  fprintf(out, "\"modmul\":[");
  bool first = true;
  for (size_t i = 0; i < vals.size(); i++) {
    for (size_t j = 0; j < vals.size(); j++) {
      if (i >= nedge && j >= nedge && j != ((i * 7 + 3) % vals.size())) continue; /* thin random x random */
      Int r;
      r.ModMulK1(&vals[i], &vals[j]);
      fprintf(out, "%s[\"%s\",\"%s\",\"%s\"]", first ? "" : ",", hex4(vals[i]).c_str(),
              hex4(vals[j]).c_str(), hex5(r).c_str());
      first = false;
    }
  }
  fprintf(out, "],\n\"modsqr\":[");
  for (size_t i = 0; i < vals.size(); i++) {
    Int r;
    r.ModSquareK1(&vals[i]);
    fprintf(out, "%s[\"%s\",\"%s\"]", i ? "," : "", hex4(vals[i]).c_str(), hex5(r).c_str());
  }
  /* ModSub is only defined for operands in [0,p): the reference's Int is 320-bit signed */
  fprintf(out, "],\n\"modsub\":[");
  first = true;
  for (size_t i = 0; i < vals.size(); i++) {
    for (size_t j = 0; j < vals.size(); j++) {
      if (i >= nedge && j >= nedge && j != ((i * 5 + 1) % vals.size())) continue;
      Int r;
      r.ModSub(&vals[i], &vals[j]);
      fprintf(out, "%s[\"%s\",\"%s\",\"%s\"]", first ? "" : ",", hex4(vals[i]).c_str(),
              hex4(vals[j]).c_str(), hex5(r).c_str());
      first = false;
    }
  }
  fprintf(out, "],\n\"modinv\":[");
  for (size_t i = 0; i < vals.size(); i++) {
    Int r(&vals[i]);
    r.ModInv();
    fprintf(out, "%s[\"%s\",\"%s\"]", i ? "," : "", hex4(vals[i]).c_str(), hex5(r).c_str());
  }
  /* grouped inverse, IntGroup.cpp:36-57 */
  fprintf(out, "],\n\"batch_inv\":{\"in\":[");
  const int GN = 37;
  IntGroup grp(GN);
  Int *gv = new Int[GN];
  for (int i = 0; i < GN; i++) {
    gv[i].Rand(256);
    fprintf(out, "%s\"%s\"", i ? "," : "", hex4(gv[i]).c_str());
  }
  grp.Set(gv);
  grp.ModInv();
  fprintf(out, "],\"out\":[");
  for (int i = 0; i < GN; i++) fprintf(out, "%s\"%s\"", i ? "," : "", hex5(gv[i]).c_str());
  fprintf(out, "]},\n");
  delete[] gv;
}

static void emit_order_kats(Secp256K1 *secp, int nrand) {
  std::vector<Int> vals;
  Int t;
  set4(t, 0, 0, 0, 0); vals.push_back(t);
  set4(t, 1, 0, 0, 0); vals.push_back(t);
  t.Set(&secp->order); t.SubOne(); vals.push_back(t);
  set4(t, ~0ULL, ~0ULL, 0, 0); vals.push_back(t);
  for (int i = 0; i < nrand; i++) {
    Int r;
    r.Rand(256);
    r.Mod(&secp->order);
    vals.push_back(r);
    Int s;
    s.Rand(100);
    vals.push_back(s);
  }
  fprintf(out, "\"order\":[");
  bool first = true;
  for (size_t i = 0; i < vals.size(); i++)
    for (size_t j = 0; j < vals.size(); j++) {
      Int a(&vals[i]), s(&vals[i]);
      a.ModAddK1order(&vals[j]);
      s.ModSubK1order(&vals[j]);
      fprintf(out, "%s[\"%s\",\"%s\",\"%s\",\"%s\"]", first ? "" : ",", hex4(vals[i]).c_str(),
              hex4(vals[j]).c_str(), hex5(a).c_str(), hex5(s).c_str());
      first = false;
    }
  fprintf(out, "],\n");
}

static void emit_rand() {
  /* Random.cpp + Int::Rand: fixed seed, assorted widths */
  const int widths[] = {1, 31, 32, 33, 41, 63, 64, 65, 96, 125, 128, 256};
  rseed(0x600DCAFE);
  fprintf(out, "\"rand\":{\"seed\":%u,\"first_rndl\":[", 0x600DCAFEu);
  for (int i = 0; i < 8; i++) fprintf(out, "%s%lu", i ? "," : "", rndl());
  fprintf(out, "],\"int_rand\":[");
  for (size_t i = 0; i < sizeof widths / sizeof *widths; i++) {
    Int r;
    r.Rand(widths[i]);
    fprintf(out, "%s[%d,\"%s\"]", i ? "," : "", widths[i], hex4(r).c_str());
  }
  fprintf(out, "]},\n");
}

static void emit_pubkeys(Secp256K1 *secp) {
  fprintf(out, "\"pubkey\":[");
  std::vector<Int> ks;
  Int k;
  k.SetInt32(1); ks.push_back(k);
  k.SetInt32(2); ks.push_back(k);
  k.SetBase16((char *)"B862A62E"); ks.push_back(k);            /* SURVEY 8d config 1 */
  k.SetBase16((char *)"378ABDEC51BC5D"); ks.push_back(k);      /* in.txt answer, README.md:331-357 */
  k.Set(&secp->order); k.SubOne(); ks.push_back(k);
  for (int i = 0; i < 8; i++) { k.Rand(256); k.Mod(&secp->order); ks.push_back(k); }
  for (size_t i = 0; i < ks.size(); i++) {
    Point p = secp->ComputePublicKey(&ks[i]);
    fprintf(out, "%s[\"%s\",\"%s\",\"%s\"]", i ? "," : "", hex4(ks[i]).c_str(), hex4(p.x).c_str(),
            hex4(p.y).c_str());
  }
  fprintf(out, "],\n");
}

static Kangaroo *new_kangaroo(Secp256K1 *secp) {
  std::string empty;
  return new Kangaroo(secp, 8, false, empty, empty, 0, false, false, -1.0, 3000, 17403, 3000, empty,
                      empty, false);
}

static void emit_jump_tables(Secp256K1 *secp) {
  const int powers[] = {32, 56, 64, 80, 109, 125};
  fprintf(out, "\"jump_tables\":{");
  for (size_t t = 0; t < sizeof powers / sizeof *powers; t++) {
    Kangaroo *kg = new_kangaroo(secp);
    kg->rangePower = powers[t];
    kg->CreateJumpTable(); /* Kangaroo.cpp:742-832 */
    fprintf(out, "%s\"%d\":{\"jd\":[", t ? "," : "", powers[t]);
    for (int i = 0; i < NB_JUMP; i++) fprintf(out, "%s\"%s\"", i ? "," : "", hex4(kg->jumpDistance[i]).c_str());
    fprintf(out, "],\"jx\":[");
    for (int i = 0; i < NB_JUMP; i++) fprintf(out, "%s\"%s\"", i ? "," : "", hex4(kg->jumpPointx[i]).c_str());
    fprintf(out, "],\"jy\":[");
    for (int i = 0; i < NB_JUMP; i++) fprintf(out, "%s\"%s\"", i ? "," : "", hex4(kg->jumpPointy[i]).c_str());
    fprintf(out, "]}");
    delete kg;
  }
  fprintf(out, "},\n");
}

/* The -check scenario (Check.cpp:472-586) at a small herd size: reference range/key constants,
 * reference CreateHerd, 64 jumps with AddDirect + ModAddK1order, DP list with the given dp. */
static void emit_walk(Secp256K1 *secp, const char *name, const char *start_hex, const char *end_hex,
                      const char *key_hex, int nb, int dp, int nsteps, uint32_t seed, bool last) {
  Kangaroo *kg = new_kangaroo(secp);
  kg->SetDP(dp);
  kg->rangeStart.SetBase16((char *)start_hex);
  kg->rangeEnd.SetBase16((char *)end_hex);
  Int k1;
  k1.SetBase16((char *)key_hex);
  Point P = secp->ComputePublicKey(&k1);
  kg->keysToSearch.clear();
  kg->keysToSearch.push_back(P);
  kg->keyIdx = 0;
  kg->InitRange();
  kg->InitSearchKey();
  kg->CreateJumpTable(); /* also reseeds from the clock: reseed below for determinism */
  rseed(seed);

  Int *px = new Int[nb], *py = new Int[nb], *pd = new Int[nb];
  kg->CreateHerd(nb, px, py, pd, TAME);

  fprintf(out, "\"%s\":{\"range_power\":%d,\"dp\":%d,\"dp_mask\":\"%016" PRIx64 "\",\"nsteps\":%d,", name,
          kg->rangePower, dp, kg->dMask, nsteps);
  fprintf(out, "\"range_start\":\"%s\",\"key\":\"%s\",", hex4(kg->rangeStart).c_str(), hex4(k1).c_str());
  fprintf(out, "\"key_to_search\":[\"%s\",\"%s\"],", hex4(kg->keyToSearch.x).c_str(),
          hex4(kg->keyToSearch.y).c_str());
  fprintf(out, "\"wild_offset\":\"%s\",", hex4(kg->rangeWidthDiv2).c_str());
  fprintf(out, "\"start\":[");
  for (int i = 0; i < nb; i++)
    fprintf(out, "%s[\"%s\",\"%s\",\"%s\"]", i ? "," : "", hex4(px[i]).c_str(), hex4(py[i]).c_str(),
            hex4(pd[i]).c_str());
  fprintf(out, "],\"dps\":[");

  Int _1;
  _1.SetInt32(1);
  bool first = true;
  for (int r = 0; r < nsteps; r++) {
    for (int i = 0; i < nb; i++) {
      uint64_t jmp = (px[i].bits64[0] % NB_JUMP);
      Point J(&kg->jumpPointx[jmp], &kg->jumpPointy[jmp], &_1);
      Point Q(&px[i], &py[i], &_1);
      Q = secp->AddDirect(Q, J);
      px[i].Set(&Q.x);
      py[i].Set(&Q.y);
      pd[i].ModAddK1order(&kg->jumpDistance[jmp]);
      if (kg->IsDP(px[i].bits64[3])) {
        fprintf(out, "%s[%d,\"%s\",\"%s\"]", first ? "" : ",", i, hex4(px[i]).c_str(), hex4(pd[i]).c_str());
        first = false;
      }
    }
  }
  fprintf(out, "],\"end\":[");
  for (int i = 0; i < nb; i++)
    fprintf(out, "%s[\"%s\",\"%s\",\"%s\"]", i ? "," : "", hex4(px[i]).c_str(), hex4(py[i]).c_str(),
            hex4(pd[i]).c_str());
  fprintf(out, "]}%s\n", last ? "" : ",");
  delete[] px;
  delete[] py;
  delete[] pd;
  delete kg;
}

/* HashTable KATs: a seeded add sequence through the reference's HashTable::Add
 * (duplicates, collisions, negative distances, buckets grown past 16/20/24 items) with every status,
 * the collision read-back, and the final content of every non-empty bucket incl. the maxItem word. */
static void emit_hashtable(const char *path, uint32_t seed) {
  FILE *f = fopen(path, "w");
  if (!f) exit(1);
  rseed(seed);
  HashTable *ht = new HashTable();
  Secp256K1 *secp = new Secp256K1();
  secp->Init();
  std::vector<Int> xs, ds;
  std::vector<uint32_t> types;
  const uint64_t hot[3] = {0x2A5F1, 0x00000, 0x3FFFF};
  for (int i = 0; i < 900; i++) {
    Int x, d;
    x.Rand(256);
    int mode = (int)(rndl() % 10);
    if (mode < 6) x.bits64[2] = (x.bits64[2] & ~0x3FFFFULL) | hot[rndl() % 3]; /* crowd three buckets */
    if (mode == 6) x.bits64[1] = xs.empty() ? x.bits64[1] : xs[rndl() % xs.size()].bits64[1]; /* equal high word */
    d.Rand(126);
    if (rndl() % 3 == 0) d.ModNegK1order(); /* negative distance: n - d */
    uint32_t type = (uint32_t)(rndl() & 1);
    if (!xs.empty() && mode == 7) { /* exact repeat */
      size_t k = rndl() % xs.size();
      x.Set(&xs[k]); d.Set(&ds[k]); type = types[k];
    } else if (!xs.empty() && mode == 8) { /* same x, other distance/type: collision */
      size_t k = rndl() % xs.size();
      x.Set(&xs[k]);
    }
    xs.push_back(x); ds.push_back(d); types.push_back(type);
  }
  fprintf(f, "{\"generator\":\"oracle/refprobe.cpp --hashtable, reference HashTable::Add\",\"seed\":%u,\n\"adds\":[\n", seed);
  for (size_t i = 0; i < xs.size(); i++) {
    int st = ht->Add(&xs[i], &ds[i], types[i]);
    fprintf(f, "[\"%s\",\"%s\",%u,%d", hex4(xs[i]).c_str(), hex4(ds[i]).c_str(), types[i], st);
    if (st == ADD_COLLISION) fprintf(f, ",\"%s\",%u", hex4(ht->kDist).c_str(), ht->kType);
    fprintf(f, "]%s\n", i + 1 < xs.size() ? "," : "");
  }
  fprintf(f, "],\n\"count\":%" PRIu64 ",\n\"buckets\":[\n", ht->GetNbItem());
  bool first = true;
  for (uint32_t h = 0; h < HASH_SIZE; h++) {
    if (ht->E[h].nbItem == 0 && ht->E[h].maxItem == 0) continue;
    fprintf(f, "%s[%u,%u,%u,[", first ? "" : ",\n", h, ht->E[h].nbItem, ht->E[h].maxItem);
    first = false;
    for (uint32_t i = 0; i < ht->E[h].nbItem; i++) {
      ENTRY *e = ht->E[h].items[i];
      fprintf(f, "%s\"%016" PRIx64 "%016" PRIx64 "%016" PRIx64 "%016" PRIx64 "\"", i ? "," : "", e->x.i64[0], e->x.i64[1],
              e->d.i64[0], e->d.i64[1]);
    }
    fprintf(f, "]]");
  }
  fprintf(f, "\n]}\n");
  fclose(f);
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <out.json> [seed]\n", argv[0]);
    return 2;
  }
  if (argc >= 3 && std::string(argv[1]) == "--hashtable") {
    Timer::Init();
    emit_hashtable(argv[2], argc > 3 ? (uint32_t)strtoul(argv[3], NULL, 0) : 0x7AB1E001u);
    return 0;
  }
  uint32_t seed = argc > 2 ? (uint32_t)strtoul(argv[2], NULL, 0) : 0x5EED1234u;
  out = fopen(argv[1], "w");
  if (!out) return 1;
  Timer::Init();
  rseed(seed);
  Secp256K1 *secp = new Secp256K1();
  secp->Init();

  fprintf(out, "{\"generator\":\"oracle/refprobe.cpp linked to reference SECPK1 objects\",\"seed\":%u,\n", seed);
  emit_field_kats(48);
  emit_order_kats(secp, 6);
  emit_rand();
  rseed(seed + 1);
  emit_pubkeys(secp);
  emit_jump_tables(secp);
  /* Check.cpp:472-476 constants (64-bit range), small herd so the fixture stays small */
  emit_walk(secp, "walk_check64", "5B3F38AF935A3640D158E871CE6E9666DB862636383386EE0000000000000000",
            "5B3F38AF935A3640D158E871CE6E9666DB862636383386EEFFFFFFFFFFFFFFFF",
            "5B3F38AF935A3640D158E871CE6E9666DB862636383386EE0000000000123000", 256, 4, NB_RUN, seed + 2, false);
  /* 80-bit range */
  emit_walk(secp, "walk_80", "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000",
            "B60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF",
            "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000C0FFEE123456789", 128, 3, NB_RUN, seed + 3, false);
  /* 125-bit (max) range: wild distances wrap mod n */
  emit_walk(secp, "walk_125", "0", "1FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF",
            "0000000000000000000000000000000012345678FEDCBA9876543210DEADBEEF", 128, 2,
            2 * NB_RUN, seed + 4, true);
  fprintf(out, "}\n");
  fclose(out);
  return 0;
}
