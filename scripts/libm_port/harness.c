// CPU validation of the transliterated glibc functions against the libm they were taken from:  gcc -O2 -ffp-contract=off harness.c -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static int g_unsupported = 0;
static inline double D(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t B(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
#define DB(u) D(u)
#define LIBM_CONST(name, val) static const uint64_t name = val;
#define LIBM_TABLE(name, n) static const uint64_t name[n]
#define LIBM_FN static
#define UNSUPPORTED(msg) do { g_unsupported++; return NAN; } while (0)
#define S64(off) stk[(off) / 8]
#define W64(off, v) (stk[(off) / 8] = (v))
#define S32(off) ((uint32_t)(stk[(off) / 8] >> (((off) & 4) * 8)))
#define W32(off, v) (stk[(off) / 8] = (stk[(off) / 8] & ~(0xffffffffull << (((off) & 4) * 8))) | ((uint64_t)(uint32_t)(v) << (((off) & 4) * 8)))
static inline uint64_t LD64(int64_t a);
#define LD32(a) ((uint32_t)LD64(a))
#include "../../nirrt_star_amd/csrc/glibc235_libm.inc"

static inline uint64_t LD64(int64_t a)
{
    if (a >= LIBM_T_SINCOS_BASE && a < LIBM_T_SINCOS_BASE + 8 * 440) return T_sincos[(a - LIBM_T_SINCOS_BASE) / 8];
    if (a >= LIBM_T_ATAN_BASE && a < LIBM_T_ATAN_BASE + 8 * 241 * 7) return T_atan[(a - LIBM_T_ATAN_BASE) / 8];
    g_unsupported++;
    return 0;
}
static uint64_t rs = 88172645463325252ull;
static double rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 10000000, bad_a = 0, bad_s = 0, bad_c = 0;
    for (long i = 0; i < n; i++) {
        double sc = (i % 7 == 0) ? 1e-3 : (i % 11 == 0 ? 1e-9 : 224.0);
        double dy = (2 * rnd() - 1) * sc, dx = (2 * rnd() - 1) * sc;
        if (i % 1013 == 0) dy = 0; if (i % 1019 == 0) dx = 0; if (i % 10007 == 0) { dy = dx; } if (i % 10009 == 0) dy = -dx;
        if (i % 977 == 0) dx = (double)(long)(dx); if (i % 983 == 0) dy = (double)(long)(dy);
        double th = atan2(dy, dx), th2 = glibc_atan2(dy, dx);
        if (B(th) != B(th2)) { if (bad_a++ < 5) printf("atan2(%a, %a) = %a, port %a\n", dy, dx, th, th2); }
        double s1 = sin(th), s2 = glibc_sin(th, 0), c1 = cos(th), c2 = glibc_cos(th, 0);
        if (B(s1) != B(s2)) { if (bad_s++ < 5) printf("sin(%a) = %a, port %a\n", th, s1, s2); }
        if (B(c1) != B(c2)) { if (bad_c++ < 5) printf("cos(%a) = %a, port %a\n", th, c1, c2); }
    }
    // angles over a wider range than atan2 produces (sin / cos on their own)
    long bad_w = 0;
    for (long i = 0; i < n / 4; i++) {
        double th = (2 * rnd() - 1) * (i % 3 == 0 ? 0.2 : (i % 3 == 1 ? 3.2 : 100.0));
        if (B(sin(th)) != B(glibc_sin(th, 0)) || B(cos(th)) != B(glibc_cos(th, 0))) { if (bad_w++ < 5) printf("wide: %a\n", th); }
    }
    printf("%ld samples: atan2 mismatches %ld, sin %ld, cos %ld, wide-range sin/cos %ld, unsupported paths hit %d\n", n, bad_a, bad_s, bad_c, bad_w, g_unsupported);
    return (bad_a || bad_s || bad_c || bad_w || g_unsupported) ? 1 : 0;
}
