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
 *        refprobe --hashtable-stress <out.tbl> <adds> [seed] [buckets]
 *        refprobe --hashtable-ingest <out.tbl> <records> [seed] [buckets] [threads]
 *
 * oracle/Makefile links this source twice: _ref/refprobe with the reference's HashTable.o, _ref/refprobe_kng with
 * kangaroo_amd/host/HashTable_kng.o in its place.  The two --hashtable* modes must then write the same bytes.
 */
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define private public /* test probe: reach Kangaroo::CreateJumpTable/CreateHerd/SetDP */
#include "Kangaroo.h"
#undef private
#include "SECPK1/IntGroup.h"
#include "SECPK1/SECP256k1.h"
#include "Timer.h"

static FILE *out;

/* HashTable_kng.cpp keeps a bucket as two ascending runs between Adds and folds them for every reader the reference has
 * (SaveTable, LoadTable); this probe reads E[h].items directly, so it asks for the fold when that object is linked in */
extern "C" void kng_ht_normalize(HashTable *ht) __attribute__((weak));

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
  if (kng_ht_normalize) kng_ht_normalize(ht);
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


/* A long add sequence through `class HashTable` that whichever object is linked in must answer identically: deep buckets
 * (`buckets` distinct bucket indices share `adds` points), equal high words of x, exact repeats, same x with another distance,
 * all three Add overloads, a SaveTable / LoadTable round trip half way, and MergeH of two overlapping tables.  Everything
 * observable goes to stdout (one line) and to the files <path>, <path>.merge. */
static uint64_t fnv(uint64_t h, const void *p, size_t n) {
  const unsigned char *c = (const unsigned char *)p;
  for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 0x100000001B3ULL;
  return h;
}
static uint64_t file_hash(const char *path, uint64_t *size) {
  FILE *f = fopen(path, "rb");
  if (!f) exit(1);
  uint64_t h = 0xCBF29CE484222325ULL, n = 0;
  unsigned char buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) {
    h = fnv(h, buf, got);
    n += got;
  }
  fclose(f);
  *size = n;
  return h;
}
static void hashtable_stress(const char *path, uint64_t adds, uint32_t seed, uint32_t buckets) {
  rseed(seed);
  Secp256K1 *secp = new Secp256K1();
  secp->Init(); /* the group order behind ModNegK1order */
  HashTable *ht = new HashTable();
  HashTable *other = new HashTable();
  std::vector<Int> xs, ds;
  std::vector<uint32_t> types;
  std::vector<uint64_t> hs(buckets);
  for (uint32_t i = 0; i < buckets; i++) hs[i] = rndl() & HASH_MASK;
  hs[0] = 0;
  if (buckets > 1) hs[1] = HASH_MASK;
  uint64_t sh = 0xCBF29CE484222325ULL, counts[3] = {0, 0, 0};
  const size_t keep = 4096;
  for (uint64_t i = 0; i < adds; i++) {
    Int x, d;
    x.Rand(256);
    x.bits64[2] = (x.bits64[2] & ~(uint64_t)HASH_MASK) | hs[rndl() % buckets];
    const int mode = (int)(rndl() % 32);
    if (mode == 0 && !xs.empty()) x.bits64[1] = xs[rndl() % xs.size()].bits64[1]; /* key tie, other low word */
    if (mode == 1) { /* clustered keys: interpolation has to fall back to the search */
      x.bits64[1] &= 0xFFFF;
    }
    d.Rand(126);
    if (rndl() % 3 == 0) d.ModNegK1order();
    uint32_t type = (uint32_t)(rndl() & 1);
    if (!xs.empty() && mode == 2) {
      const size_t k = rndl() % xs.size();
      x.Set(&xs[k]); d.Set(&ds[k]); type = types[k];
    } else if (!xs.empty() && mode == 3) {
      x.Set(&xs[rndl() % xs.size()]);
    }
    if (xs.size() < keep) {
      xs.push_back(x); ds.push_back(d); types.push_back(type);
    } else {
      const size_t k = rndl() % keep;
      xs[k] = x; ds[k] = d; types[k] = type;
    }
    int st;
    const int how = (int)(i % 3);
    if (how == 0) {
      st = ht->Add(&x, &d, type);
    } else {
      uint64_t h;
      int128_t X, D;
      HashTable::Convert(&x, &d, type, &h, &X, &D);
      if (how == 1) {
        st = ht->Add(h, &X, &D);
      } else {
        ENTRY *e = (ENTRY *)malloc(sizeof(ENTRY));
        e->x = X; e->d = D;
        st = ht->Add(h, e);
      }
    }
    counts[st]++;
    sh = fnv(sh, &st, sizeof st);
    if (st == ADD_COLLISION) {
      sh = fnv(sh, ht->kDist.bits64, 32);
      sh = fnv(sh, &ht->kType, sizeof(uint32_t));
    }
    if (i % 5 == 0) (void)other->Add(&x, &d, type ^ (uint32_t)(i % 7 == 0)); /* overlaps ht, sometimes with another type word */
    if (i == adds / 2) { /* what -ws / -i do */
      std::string tmp = std::string(path) + ".half";
      FILE *f = fopen(tmp.c_str(), "wb");
      ht->SaveTable(f, 0, HASH_SIZE, false);
      fclose(f);
      f = fopen(tmp.c_str(), "rb");
      ht->LoadTable(f);
      fclose(f);
      remove(tmp.c_str());
    }
  }
  FILE *f = fopen(path, "wb");
  ht->SaveTable(f, 0, HASH_SIZE, false);
  fclose(f);
  std::string p2 = std::string(path) + ".other", pm = std::string(path) + ".merge";
  f = fopen(p2.c_str(), "wb");
  other->SaveTable(f, 0, HASH_SIZE, false);
  fclose(f);
  FILE *f1 = fopen(path, "rb"), *f2 = fopen(p2.c_str(), "rb"), *fm = fopen(pm.c_str(), "wb");
  uint64_t mh = 0xCBF29CE484222325ULL, merged = 0, dups = 0, colls = 0;
  for (uint32_t h = 0; h < HASH_SIZE; h++) {
    uint32_t nb = 0, dup = 0, k1 = 0, k2 = 0;
    Int d1, d2;
    d1.SetInt32(0); d2.SetInt32(0);
    const int st = HashTable::MergeH(h, f1, f2, fm, &nb, &dup, &d1, &k1, &d2, &k2);
    merged += nb;
    dups += dup;
    if (st == ADD_COLLISION) {
      colls++;
      mh = fnv(mh, d1.bits64, 32); mh = fnv(mh, &k1, 4);
      mh = fnv(mh, d2.bits64, 32); mh = fnv(mh, &k2, 4);
    }
  }
  fclose(f1); fclose(f2); fclose(fm);
  remove(p2.c_str());
  /* the merged file loaded back: the table every later Add would search */
  HashTable *back = new HashTable();
  f = fopen(pm.c_str(), "rb");
  back->LoadTable(f);
  fclose(f);
  uint64_t bh = 0xCBF29CE484222325ULL;
  for (uint32_t h = 0; h < HASH_SIZE; h++)
    for (uint32_t i = 0; i < back->E[h].nbItem; i++) bh = fnv(bh, back->E[h].items[i], 32);
  uint64_t sz = 0, szm = 0;
  const uint64_t th = file_hash(path, &sz), tmh = file_hash(pm.c_str(), &szm);
  printf("adds %" PRIu64 " ok %" PRIu64 " dup %" PRIu64 " coll %" PRIu64 " statuses %016" PRIx64 " items %" PRIu64 " table %016" PRIx64
         " bytes %" PRIu64 " merged %" PRIu64 " mdup %" PRIu64 " mcoll %" PRIu64 " mcollhash %016" PRIx64 " mergefile %016" PRIx64 " bytes %" PRIu64
         " loaded %016" PRIx64 " %" PRIu64 "\n",
         adds, counts[0], counts[1], counts[2], sh, ht->GetNbItem(), th, sz, merged, dups, colls, mh, tmh, szm, bh, back->GetNbItem());
}

/* The batch interface of HashTable_kng.cpp (kng_hashtable_ext.h) against the per-point path it stands in for.  Engine
 * records {x[4], device distance, kidx} go (A) through what GPUEngine::Launch + Kangaroo::AddToTable do per point
 * (GPUEngine.cu:668-674: wild distance - offset mod n; HashTable::Add(Int*, Int*, type)) and, when the replacement object is
 * linked in, (B) through kng_ht_ingest in one thread, in the same order -- statuses, collision read-backs and SaveTable bytes
 * must agree -- and (C) through kng_ht_ingest from `threads` threads at once: same set of x, nothing lost.  Path A alone runs
 * with the reference's HashTable.o too, and prints the same line. */
struct probe_rec { uint64_t x[4], d[2], kidx, reserved; };
struct probe_event { uint32_t index, status; uint64_t stored_d[2]; };
extern "C" int kng_ht_ingest(HashTable *ht, const probe_rec *recs, uint32_t n, const uint64_t wild_off[2], probe_event *ev, uint32_t ev_cap,
                             uint32_t *n_ev) __attribute__((weak));
#include <pthread.h>
struct ingest_job { HashTable *ht; const probe_rec *recs; uint32_t n; const uint64_t *off; uint32_t events; };
static void *ingest_thread(void *p) {
  ingest_job *j = (ingest_job *)p;
  const uint32_t chunk = 4096;
  for (uint32_t at = 0; at < j->n; at += chunk) {
    uint32_t ne = 0;
    kng_ht_ingest(j->ht, j->recs + at, j->n - at < chunk ? j->n - at : chunk, j->off, NULL, 0, &ne);
    j->events += ne;
  }
  return NULL;
}
static void hashtable_ingest(const char *path, uint32_t n, uint32_t seed, uint32_t buckets, int threads) {
  rseed(seed);
  Secp256K1 *secp = new Secp256K1();
  secp->Init();
  Int off;
  off.SetInt32(0);
  off.bits64[1] = 1ULL << 60; /* N/2 of a 125-bit range */
  const uint64_t off2[2] = {off.bits64[0], off.bits64[1]};
  std::vector<probe_rec> recs(n);
  std::vector<uint64_t> hs(buckets);
  for (uint32_t i = 0; i < buckets; i++) hs[i] = rndl() & HASH_MASK;
  for (uint32_t i = 0; i < n; i++) {
    probe_rec &r = recs[i];
    Int x, d;
    x.Rand(256);
    d.Rand(126);
    const int mode = (int)(rndl() % 24);
    if (mode == 0) { d.SetInt32(0); d.bits64[1] = 1ULL << 60; }              /* device distance == offset: true distance 0 */
    if (mode == 1) { d.SetInt32(0); d.bits64[0] = rndl(); }                    /* far below the offset: negative */
    memcpy(r.x, x.bits64, 32);
    r.x[2] = (r.x[2] & ~(uint64_t)HASH_MASK) | hs[rndl() % buckets];
    r.d[0] = d.bits64[0];
    r.d[1] = d.bits64[1];
    r.kidx = rndl();
    r.reserved = 0;
    if (i && mode == 2) r = recs[rndl() % i];                                   /* the same point again */
    if (i && mode == 3) { memcpy(r.x, recs[rndl() % i].x, 32); }                /* same x, other distance */
    if (i && mode == 4) r.x[1] = recs[rndl() % i].x[1];                         /* key tie */
  }
  /* (A) */
  HashTable *a = new HashTable();
  uint64_t sh = 0xCBF29CE484222325ULL, counts[3] = {0, 0, 0};
  std::vector<probe_event> want;
  for (uint32_t i = 0; i < n; i++) {
    Int x, d;
    x.SetInt32(0); d.SetInt32(0);
    memcpy(x.bits64, recs[i].x, 32);
    d.bits64[0] = recs[i].d[0];
    d.bits64[1] = recs[i].d[1];
    const uint32_t type = (uint32_t)(recs[i].kidx % 2);
    if (type == 1) d.ModSubK1order(&off);
    const int st = a->Add(&x, &d, type);
    counts[st]++;
    sh = fnv(sh, &st, sizeof st);
    if (st == ADD_COLLISION) {
      sh = fnv(sh, a->kDist.bits64, 32);
      sh = fnv(sh, &a->kType, sizeof(uint32_t));
    }
    if (st != ADD_OK) {
      probe_event e = {i, (uint32_t)st, {0, 0}};
      if (st == ADD_COLLISION) { /* re-encode what Add decoded, to compare with the raw word path B returns */
        Int kd(&a->kDist);
        uint64_t flags = (uint64_t)a->kType << 62;
        if (kd.bits64[3] > 0x7FFFFFFFFFFFFFFFULL) { kd.ModNegK1order(); flags |= 1ULL << 63; }
        e.stored_d[0] = kd.bits64[0];
        e.stored_d[1] = (kd.bits64[1] & 0x3FFFFFFFFFFFFFFFULL) | flags;
      }
      want.push_back(e);
    }
  }
  FILE *f = fopen(path, "wb");
  a->SaveTable(f, 0, HASH_SIZE, false);
  fclose(f);
  uint64_t sz = 0;
  const uint64_t th = file_hash(path, &sz);
  printf("records %u ok %" PRIu64 " dup %" PRIu64 " coll %" PRIu64 " statuses %016" PRIx64 " table %016" PRIx64 " bytes %" PRIu64 "\n", n, counts[0],
         counts[1], counts[2], sh, th, sz);
  if (!kng_ht_ingest) return;
  /* (B) */
  HashTable *b = new HashTable();
  std::vector<probe_event> got;
  const uint32_t chunk = 3000;
  std::vector<probe_event> ev(chunk);
  for (uint32_t at = 0; at < n; at += chunk) {
    const uint32_t m = n - at < chunk ? n - at : chunk;
    uint32_t ne = 0;
    kng_ht_ingest(b, recs.data() + at, m, off2, ev.data(), chunk, &ne);
    for (uint32_t k = 0; k < ne; k++) {
      ev[k].index += at;
      got.push_back(ev[k]);
    }
  }
  bool same = got.size() == want.size();
  for (size_t k = 0; same && k < got.size(); k++)
    same = got[k].index == want[k].index && got[k].status == want[k].status && got[k].stored_d[0] == want[k].stored_d[0] &&
           got[k].stored_d[1] == want[k].stored_d[1];
  std::string pb = std::string(path) + ".ingest";
  f = fopen(pb.c_str(), "wb");
  b->SaveTable(f, 0, HASH_SIZE, false);
  fclose(f);
  uint64_t szb = 0;
  const uint64_t thb = file_hash(pb.c_str(), &szb);
  printf("ingest: events %zu/%zu %s, table %s\n", got.size(), want.size(), same ? "identical" : "DIFFERENT",
         thb == th && szb == sz ? "identical" : "DIFFERENT");
  /* (C) */
  HashTable *c = new HashTable();
  std::vector<ingest_job> jobs(threads);
  std::vector<pthread_t> tid(threads);
  for (int t = 0; t < threads; t++) {
    const uint32_t lo = (uint32_t)((uint64_t)n * t / threads), hi = (uint32_t)((uint64_t)n * (t + 1) / threads);
    jobs[t] = {c, recs.data() + lo, hi - lo, off2, 0};
    pthread_create(&tid[t], NULL, ingest_thread, &jobs[t]);
  }
  uint64_t events = 0;
  for (int t = 0; t < threads; t++) {
    pthread_join(tid[t], NULL);
    events += jobs[t].events;
  }
  kng_ht_normalize(c);
  bool xs_same = c->GetNbItem() == a->GetNbItem();
  kng_ht_normalize(a);
  for (uint32_t h = 0; xs_same && h < HASH_SIZE; h++) {
    xs_same = c->E[h].nbItem == a->E[h].nbItem;
    for (uint32_t i = 0; xs_same && i < c->E[h].nbItem; i++)
      xs_same = c->E[h].items[i]->x.i64[0] == a->E[h].items[i]->x.i64[0] && c->E[h].items[i]->x.i64[1] == a->E[h].items[i]->x.i64[1];
  }
  printf("ingest x%d threads: %" PRIu64 " entries + %" PRIu64 " events = %" PRIu64 " of %u, x set %s\n", threads, c->GetNbItem(), events,
         c->GetNbItem() + events, n, xs_same && c->GetNbItem() + events == n ? "identical" : "DIFFERENT");
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
  if (argc >= 4 && std::string(argv[1]) == "--hashtable-stress") {
    Timer::Init();
    hashtable_stress(argv[2], strtoull(argv[3], NULL, 0), argc > 4 ? (uint32_t)strtoul(argv[4], NULL, 0) : 0x57E55001u,
                     argc > 5 ? (uint32_t)strtoul(argv[5], NULL, 0) : 48);
    return 0;
  }
  if (argc >= 4 && std::string(argv[1]) == "--hashtable-ingest") {
    Timer::Init();
    hashtable_ingest(argv[2], (uint32_t)strtoul(argv[3], NULL, 0), argc > 4 ? (uint32_t)strtoul(argv[4], NULL, 0) : 0x1465E501u,
                     argc > 5 ? (uint32_t)strtoul(argv[5], NULL, 0) : 4096, argc > 6 ? atoi(argv[6]) : 4);
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
