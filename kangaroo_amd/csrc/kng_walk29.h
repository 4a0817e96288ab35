// kng_walk29.h -- the walk kernel on the carry-free 9x29-bit field representation (kng_field29.h).
// Same data flow as kng_walk_kernel (kng_engine.hip): one alternating-direction pass per jump,
// software prefetch, LDS jump table, wave-compacted DP output.  Differences:
//   * an element is 36 bytes in HBM: two 16-byte vectors (limbs 0-3, 4-7) + one dword (limb 8);
//     x is stored canonical, y and the running products "almost reduced" (limbs < 2^29);
//   * the LDS table holds the BIASED NEGATIONS 2p - Jx, 2p - Jy (limb-major, 32 distinct banks per
//     ds_read_b32), so dx = x - Jx, dy = y - Jy and rx = s^2 - Jx - x are limb-wise adds.
// Replaces GPUCompute.h:22-117 for walk policy "29".
#pragma once

#include <hip/hip_runtime.h>

#include "kng_field29.h"
#include "kng_modinv.h"

namespace kng {

struct Planes29 { // one field element plane set
    uint4 *a;     // limbs 0..3
    uint4 *b;     // limbs 4..7
    uint32_t *c;  // limb 8
};

#define JT29_NJX 0
#define JT29_NJY (9 * 32)
#define JT29_JD (18 * 32)
#define JT29_WORDS (22 * 32)

struct Walk29Args {
    Planes29 x, y, s;
    ulonglong2 *d;
    const uint32_t *jtab; // limb-major: njx[9][32] njy[9][32] jd[4][32]
    uint32_t m8, m7, m6;  // DP mask spread over limbs 8, 7, 6 (bits 232-255, 203-231, 192-202 of x)
    uint32_t *dp_count;
    void *dp_items; // DpRecord[]
    uint32_t max_found;
    uint32_t lanes, group, nsteps;
    uint64_t n_kang;
};

// distance plane: N low words followed by N high words (layout shared with the policy-32 kernels)
KNG_DEV ulonglong2 ld_d29(const ulonglong2 *d, size_t n, size_t i) {
    const uint64_t *p = reinterpret_cast<const uint64_t *>(d);
    return make_ulonglong2(p[i], p[n + i]);
}
KNG_DEV void st_d29(ulonglong2 *d, size_t n, size_t i, const ulonglong2 &v) {
    uint64_t *p = reinterpret_cast<uint64_t *>(d);
    p[i] = v.x;
    p[n + i] = v.y;
}
KNG_DEV fe29 ld29(const Planes29 &p, size_t i) {
    const uint4 a = p.a[i], b = p.b[i];
    const uint32_t c = p.c[i];
    return fe29{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c}};
}
KNG_DEV void st29(const Planes29 &p, size_t i, const fe29 &v) {
    p.a[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    p.b[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    p.c[i] = v.l[8];
}
KNG_DEV fe29 lds29(const uint32_t *tab, int base, uint32_t j) {
    fe29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = tab[base + 32 * i + j];
    return r;
}

// canonical inverse of an almost-reduced element, as normalised limbs
KNG_DEV fe29 fe29_inv(const fe29 &a) { return fe29_unpack(fe_inv(fe29_pack(fe29_canon(a)))); }

template <typename EmitFn>
KNG_DEV void walk29_body(const Walk29Args &a, const uint32_t *tab, EmitFn emit) {
    const size_t L = a.lanes;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L) return;
    const uint32_t G = (uint32_t)((a.n_kang - t + L - 1) / L);

    // pass 0: running products of dx in ascending order
    fe29 acc;
    for (uint32_t g = 0; g < G; g++) {
        const size_t idx = (size_t)g * L + t;
        const fe29 x = ld29(a.x, idx);
        const uint32_t j = x.l[0] & (KNG_NB_JUMP - 1);
        const fe29 dx = fe29_add(x, lds29(tab, JT29_NJX, j));
        acc = g ? fe29_mul(dx, acc) : fe29_norm(dx);
        st29(a.s, idx, acc);
    }

    for (uint32_t step = 0; step < a.nsteps; step++) {
        fe29 inv = fe29_inv(acc);
        const bool backward = !(step & 1);
        const bool last = (step + 1 == a.nsteps);
        auto slot = [&](uint32_t k) -> size_t { return (size_t)(backward ? (G - 1 - k) : k) * L + t; };

        size_t idx = slot(0);
        fe29 cx = ld29(a.x, idx);
        fe29 cy = ld29(a.y, idx);
        ulonglong2 cd = ld_d29(a.d, a.n_kang, idx);
        fe29 nb = cx;
        if (G > 1) nb = ld29(a.s, slot(1));

        for (uint32_t k = 0; k < G; k++) {
            // prefetch the next kangaroo and the product after it, before any store
            fe29 nx = cx, ny = cy, nnb = nb;
            ulonglong2 nd = cd;
            size_t nidx = idx;
            if (k + 1 < G) {
                nidx = slot(k + 1);
                nx = ld29(a.x, nidx);
                ny = ld29(a.y, nidx);
                nd = ld_d29(a.d, a.n_kang, nidx);
            }
            if (k + 2 < G) nnb = ld29(a.s, slot(k + 2));

            const uint32_t j = cx.l[0] & (KNG_NB_JUMP - 1);
            const fe29 njx = lds29(tab, JT29_NJX, j);
            const fe29 njy = lds29(tab, JT29_NJY, j);
            const fe29 dx = fe29_add(cx, njx); // x - Jx + 2p, lazy
            fe29 invk;
            if (k + 1 < G) {
                invk = fe29_mul(inv, nb); // 1/dx
                inv = fe29_mul(dx, inv);  // 1/(product of the remaining dx)
            } else {
                invk = inv;
            }
            const fe29 dy = fe29_add(cy, njy);
            const fe29 s = fe29_mul(dy, invk);
            const fe29 p2 = fe29_sqr(s);
            // rx = s^2 - Jx - x ; canonical because its bits steer the walk and the DP test
            const fe29 rx = fe29_canon(fe29_add(fe29_add(p2, njx), fe29_neg2p(cx)));
            // ry = (x - rx) * s - y
            const fe29 m = fe29_mul(fe29_sub2p(cx, rx), s);
            const fe29 ry = fe29_norm(fe29_sub2p(m, cy));
            {
                const uint64_t jd0 = (uint64_t)tab[JT29_JD + j] | ((uint64_t)tab[JT29_JD + 32 + j] << 32);
                const uint64_t jd1 = (uint64_t)tab[JT29_JD + 64 + j] | ((uint64_t)tab[JT29_JD + 96 + j] << 32);
                unsigned long long c = 0;
                cd.x = __builtin_addcll(cd.x, jd0, 0, &c);
                cd.y = cd.y + jd1 + c;
            }
            st29(a.x, idx, rx);
            st29(a.y, idx, ry);
            st_d29(a.d, a.n_kang, idx, cd);

            const bool is_dp = (((rx.l[8] & a.m8) | (rx.l[7] & a.m7) | (rx.l[6] & a.m6)) == 0);
            emit(is_dp, rx, cd, (uint64_t)idx);

            if (!last) {
                const uint32_t j2 = rx.l[0] & (KNG_NB_JUMP - 1);
                const fe29 dx2 = fe29_add(rx, lds29(tab, JT29_NJX, j2));
                acc = k ? fe29_mul(dx2, acc) : fe29_norm(dx2);
                st29(a.s, idx, acc);
            }
            cx = nx;
            cy = ny;
            cd = nd;
            nb = nnb;
            idx = nidx;
        }
    }
}

} // namespace kng
