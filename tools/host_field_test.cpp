// host_field_test.cpp -- runs the DEVICE field arithmetic (kng_field.h / kng_modinv.h) on the host
// (clang++ only: uses __builtin_addcll) so that it can be checked without a GPU.
//   reads lines "op a b" (hex, 64 digits) on stdin, prints the result hex per line.
#include <cstdio>
#include <cstring>
#include <string>
#include "../kangaroo_amd/csrc/kng_field.h"
#include "../kangaroo_amd/csrc/kng_modinv.h"
#include "fe_extras.h"
using namespace kng;
static fe parse(const char *s) {
    fe r{{0, 0, 0, 0}};
    size_t n = strlen(s);
    for (size_t i = 0; i < n && i < 64; i++) {
        char c = s[n - 1 - i];
        uint64_t v = (c >= '0' && c <= '9') ? c - '0' : (c | 32) - 'a' + 10;
        r.v[i / 16] |= v << (4 * (i % 16));
    }
    return r;
}
int main() {
    char op[32], a[128], b[128];
    while (scanf("%31s %127s %127s", op, a, b) == 3) {
        fe x = parse(a), y = parse(b), z;
        if (!strcmp(op, "mul")) z = fe_mul(x, y);
        else if (!strcmp(op, "sqr")) z = fe_sqr(x);
        else if (!strcmp(op, "sub")) z = fe_sub(x, y);
        else if (!strcmp(op, "inv")) z = fe_inv(x);
        else if (!strcmp(op, "invf")) z = fe_inv_fermat(x);
        else return 2;
        printf("%016llx%016llx%016llx%016llx\n", (unsigned long long)z.v[3], (unsigned long long)z.v[2], (unsigned long long)z.v[1], (unsigned long long)z.v[0]);
    }
    return 0;
}
