// kng_workfile.cpp -- see kng_workfile.h.  Product code (host, no GPU needed).
#include "kng_workfile.h"

#include <unistd.h>

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

void *fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return nullptr;
}

constexpr size_t CHUNK = 1u << 15; // kangaroos assembled per fwrite/fread (3 MiB)

} // namespace

struct kngw_file {
    FILE *f = nullptr;
    bool writer = false;
    uint64_t declared = 0, done = 0; // kangaroo section: announced / transferred so far
    std::string path;                // final name
    std::string tmp;                 // writers: the file being written, renamed over `path` by kngw_close
    std::vector<uint64_t> buf;       // CHUNK x 12 limbs
};

extern "C" {

const char *kngw_last_error(void) { return g_err.c_str(); }

kngw_file *kngw_create(const char *path, const kngw_header *h, const kngt_table *table, uint64_t n_kangaroos) {
    if (!path || !h) return (kngw_file *)fail("null argument");
    if (h->magic != KNGW_HEADW && h->magic != KNGW_HEADK) return (kngw_file *)fail("unknown work file type 0x%08X", h->magic);
    if (h->magic == KNGW_HEADW && !table) return (kngw_file *)fail("a HEADW work file needs a hash table");
    // never write over the previous checkpoint in place: a crash, a full disk or a kill during a periodic save would
    // destroy days of distinguished points.  Write a sibling, flush it to the disk, then rename it over the target.
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return (kngw_file *)fail("cannot open %s for writing: %s", tmp.c_str(), std::strerror(errno));
    bool ok = std::fwrite(&h->magic, 4, 1, f) == 1 && std::fwrite(&h->version, 4, 1, f) == 1;
    if (ok && h->magic == KNGW_HEADW) {
        ok = std::fwrite(&h->dp_size, 4, 1, f) == 1 && std::fwrite(h->range_start, 32, 1, f) == 1 &&
             std::fwrite(h->range_end, 32, 1, f) == 1 && std::fwrite(h->key_x, 32, 1, f) == 1 &&
             std::fwrite(h->key_y, 32, 1, f) == 1 && std::fwrite(&h->total_count, 8, 1, f) == 1 &&
             std::fwrite(&h->total_seconds, 8, 1, f) == 1 && kngt_write(table, f) == 0;
    }
    ok = ok && std::fwrite(&n_kangaroos, 8, 1, f) == 1;
    if (!ok) {
        const int e = errno;
        std::fclose(f);
        std::remove(tmp.c_str());
        return (kngw_file *)fail("short write to %s: %s", tmp.c_str(), std::strerror(e));
    }
    kngw_file *w = new kngw_file();
    w->f = f;
    w->writer = true;
    w->declared = n_kangaroos;
    w->path = path;
    w->tmp = tmp;
    return w;
}

int kngw_put_kangaroos(kngw_file *w, const uint64_t *x, const uint64_t *y, const uint64_t *d, uint64_t n) {
    if (!w || !w->writer || !x || !y || !d) return fail("bad argument"), -1;
    if (w->done + n > w->declared) return fail("%s: more kangaroos than the %llu announced", w->path.c_str(), (unsigned long long)w->declared), -1;
    w->buf.resize(CHUNK * 12);
    for (uint64_t c0 = 0; c0 < n; c0 += CHUNK) {
        const size_t m = (size_t)(n - c0 < CHUNK ? n - c0 : CHUNK);
        uint64_t *o = w->buf.data();
        for (size_t i = 0; i < m; i++, o += 12) {
            std::memcpy(o, x + (c0 + i) * 4, 32);
            std::memcpy(o + 4, y + (c0 + i) * 4, 32);
            std::memcpy(o + 8, d + (c0 + i) * 4, 32);
        }
        if (std::fwrite(w->buf.data(), 96, m, w->f) != m) return fail("short write to %s: %s", w->path.c_str(), std::strerror(errno)), -1;
    }
    w->done += n;
    return 0;
}

int kngw_put_records(kngw_file *w, const void *records, uint64_t n) {
    if (!w || !w->writer || (!records && n)) return fail("bad argument"), -1;
    if (w->done + n > w->declared) return fail("%s: more kangaroos than the %llu announced", w->path.c_str(), (unsigned long long)w->declared), -1;
    if (n && std::fwrite(records, 96, n, w->f) != n) return fail("short write to %s: %s", w->path.c_str(), std::strerror(errno)), -1;
    w->done += n;
    return 0;
}

int kngw_get_records(kngw_file *r, void *records, uint64_t n) {
    if (!r || r->writer || (!records && n)) return fail("bad argument"), -1;
    if (r->done + n > r->declared) return fail("%s holds only %llu kangaroos", r->path.c_str(), (unsigned long long)r->declared), -1;
    if (n && std::fread(records, 96, n, r->f) != n) return fail("%s: truncated kangaroo section", r->path.c_str()), -1;
    r->done += n;
    return 0;
}

kngw_file *kngw_open(const char *path, kngw_header *h, kngt_table *table, uint64_t *n_kangaroos) {
    if (!path || !h) return (kngw_file *)fail("null argument");
    FILE *f = std::fopen(path, "rb");
    if (!f) return (kngw_file *)fail("cannot open %s for reading: %s", path, std::strerror(errno));
    std::memset(h, 0, sizeof *h);
    bool ok = std::fread(&h->magic, 4, 1, f) == 1 && std::fread(&h->version, 4, 1, f) == 1;
    if (ok && h->magic != KNGW_HEADW && h->magic != KNGW_HEADK) {
        std::fclose(f);
        return (kngw_file *)fail("%s is not a work file (magic 0x%08X)", path, h->magic);
    }
    if (ok && h->magic == KNGW_HEADW) {
        ok = std::fread(&h->dp_size, 4, 1, f) == 1 && std::fread(h->range_start, 32, 1, f) == 1 &&
             std::fread(h->range_end, 32, 1, f) == 1 && std::fread(h->key_x, 32, 1, f) == 1 &&
             std::fread(h->key_y, 32, 1, f) == 1 && std::fread(&h->total_count, 8, 1, f) == 1 &&
             std::fread(&h->total_seconds, 8, 1, f) == 1;
        if (ok && table) {
            ok = kngt_read(table, f) == 0;
        } else if (ok) { // skip: per bucket u32 nbItem, u32 maxItem, nbItem x 32 B (HashTable.cpp:420-434)
            for (uint32_t b = 0; ok && b < KNGT_BUCKETS; b++) {
                uint32_t head[2];
                ok = std::fread(head, 4, 2, f) == 2 && fseeko(f, (off_t)head[0] * 32, SEEK_CUR) == 0;
            }
        }
    }
    uint64_t n = 0;
    ok = ok && std::fread(&n, 8, 1, f) == 1;
    if (!ok) {
        std::fclose(f);
        return (kngw_file *)fail("%s: truncated work file", path);
    }
    if (n_kangaroos) *n_kangaroos = n;
    kngw_file *r = new kngw_file();
    r->f = f;
    r->declared = n;
    r->path = path;
    return r;
}

int kngw_get_kangaroos(kngw_file *r, uint64_t *x, uint64_t *y, uint64_t *d, uint64_t n) {
    if (!r || r->writer || !x || !y || !d) return fail("bad argument"), -1;
    if (r->done + n > r->declared) return fail("%s holds only %llu kangaroos", r->path.c_str(), (unsigned long long)r->declared), -1;
    r->buf.resize(CHUNK * 12);
    for (uint64_t c0 = 0; c0 < n; c0 += CHUNK) {
        const size_t m = (size_t)(n - c0 < CHUNK ? n - c0 : CHUNK);
        if (std::fread(r->buf.data(), 96, m, r->f) != m) return fail("%s: truncated kangaroo section", r->path.c_str()), -1;
        const uint64_t *o = r->buf.data();
        for (size_t i = 0; i < m; i++, o += 12) {
            std::memcpy(x + (c0 + i) * 4, o, 32);
            std::memcpy(y + (c0 + i) * 4, o + 4, 32);
            std::memcpy(d + (c0 + i) * 4, o + 8, 32);
        }
    }
    r->done += n;
    return 0;
}

int kngw_close(kngw_file *w) {
    if (!w) return 0;
    int rc = 0;
    if (w->writer && w->done != w->declared) {
        fail("%s: %llu kangaroos announced, %llu written", w->path.c_str(), (unsigned long long)w->declared, (unsigned long long)w->done);
        rc = -1;
    }
    if (w->writer && rc == 0 && (std::fflush(w->f) != 0 || fsync(fileno(w->f)) != 0)) {
        fail("flushing %s: %s", w->tmp.c_str(), std::strerror(errno));
        rc = -1;
    }
    if (std::fclose(w->f) != 0 && rc == 0) {
        fail("closing %s: %s", w->writer ? w->tmp.c_str() : w->path.c_str(), std::strerror(errno));
        rc = -1;
    }
    if (w->writer) {
        if (rc == 0 && std::rename(w->tmp.c_str(), w->path.c_str()) != 0) {
            fail("renaming %s to %s: %s", w->tmp.c_str(), w->path.c_str(), std::strerror(errno));
            rc = -1;
        }
        if (rc != 0) std::remove(w->tmp.c_str()); // the previous file at `path`, if any, is untouched
    }
    delete w;
    return rc;
}

} // extern "C"
