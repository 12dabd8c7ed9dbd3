/*
 * gen_mwc_mults.c — generate the multiplier table of the multiply-with-carry RNG.
 *
 * A multiplier a is kept when p = a*2^32 - 1 is a safe prime, i.e. p and (p-1)/2 are both
 * prime, which gives the lag-1 MWC generator x' = (a*x + c) mod 2^32 its maximal period
 * (p-1)/2.  Candidates run downward from 2^32-1.  Role of helpers/genprimes.c:25-55 and of
 * cuburn/code/primes.bin in the reference (which uses GMP; here: segmented sieve + a
 * deterministic 64-bit Miller-Rabin on unsigned __int128).  Output: little-endian u32.
 *
 *   gcc -O2 -o gen_mwc_mults gen_mwc_mults.c && ./gen_mwc_mults 262144 > mwc_mults.bin
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)((u128)a * b % m); }
static uint64_t powmod(uint64_t b, uint64_t e, uint64_t m)
{
    uint64_t r = 1;
    b %= m;
    while (e) { if (e & 1) r = mulmod(r, b, m); b = mulmod(b, b, m); e >>= 1; }
    return r;
}
/* deterministic for all n < 2^64 with these 12 bases */
static int is_prime64(uint64_t n)
{
    static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (int i = 0; i < 12; ++i) { if (n == bases[i]) return 1; if (n % bases[i] == 0) return 0; }
    uint64_t d = n - 1; int s = 0;
    while (!(d & 1)) { d >>= 1; ++s; }
    for (int i = 0; i < 12; ++i) {
        uint64_t x = powmod(bases[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; ++r) { x = mulmod(x, x, n); if (x == n - 1) { comp = 0; break; } }
        if (comp) return 0;
    }
    return 1;
}

#define SEG (1u << 22)
#define SMALL_LIMIT 65536

int main(int argc, char **argv)
{
    uint32_t want = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 0) : 262144;
    /* small primes */
    static uint8_t comp[SMALL_LIMIT];
    static uint32_t sp[SMALL_LIMIT]; int nsp = 0;
    for (uint32_t i = 2; i < SMALL_LIMIT; ++i) {
        if (!comp[i]) { if (i > 2) sp[nsp++] = i; for (uint32_t j = i * 2; j < SMALL_LIMIT; j += i) comp[j] = 1; }
    }
    uint8_t *mark = malloc(SEG);
    uint32_t found = 0;
    uint64_t hi = 0xFFFFFFFFull;   /* inclusive top of the current segment */
    while (found < want && hi > 0x80000000ull) {
        uint64_t lo = hi + 1 >= SEG ? hi + 1 - SEG : 0;   /* segment = [lo, hi] */
        memset(mark, 0, SEG);
        for (int k = 0; k < nsp; ++k) {
            uint64_t s = sp[k];
            /* p = a*2^32 - 1 == 0 (mod s)  <=>  a == inv(2^32) (mod s);  q = a*2^31 - 1 likewise */
            uint64_t i32 = powmod((uint64_t)(4294967296ull % s), s - 2, s);
            uint64_t i31 = powmod((uint64_t)(2147483648ull % s), s - 2, s);
            uint64_t res[2] = {i32, i31};
            for (int t = 0; t < 2; ++t) {
                uint64_t a0 = lo + ((res[t] + s - lo % s) % s);
                for (uint64_t a = a0; a <= hi; a += s) mark[a - lo] = 1;
            }
        }
        for (uint64_t a = hi; a >= lo && found < want; --a) {
            if (!mark[a - lo]) {
                uint64_t p = (a << 32) - 1, q = (a << 31) - 1;
                if (is_prime64(q) && is_prime64(p)) {
                    uint32_t v = (uint32_t)a;
                    fwrite(&v, 4, 1, stdout);
                    ++found;
                }
            }
            if (a == 0) break;
        }
        if (lo == 0) break;
        hi = lo - 1;
    }
    fprintf(stderr, "found %u multipliers, last segment top 0x%llx\n", found, (unsigned long long)hi);
    return found == want ? 0 : 1;
}
