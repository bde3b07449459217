// exp_table_gen -- the 2^(k/128) table of cafe_amd/csrc/exp_like_host.hpp, computed here in quad precision (libquadmath):
//   gcc -O2 -o /tmp/exp_table_gen tools/exp_table_gen.c -lquadmath -lm && /tmp/exp_table_gen > cafe_amd/csrc/exp_like_host_table.inc
// entry 2k   = the relative tail of 2^(k/128) beyond its double, (v - hi) / hi, as a double (bits)
// entry 2k+1 = bits of hi = double(2^(k/128)) minus (k << 52) / 128, so that adding (k_total << 45) yields the scaled power
// (the layout of the table in the double-precision exp of ARM's optimized routines, which glibc >= 2.28 ships; the VALUES are
// computed, not copied -- tests/test_exp_like_host.py checks the function built on them against this host's exp() bit for bit)
#include <quadmath.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
int main(void)
{
    for (int k = 0; k < 128; ++k) {
        const __float128 v = powq(2.0Q, (__float128)k / 128);
        const double hi = (double)v;
        const double tail = (double)((v - (__float128)hi) / (__float128)hi);
        uint64_t a, b;
        memcpy(&a, &tail, 8);
        memcpy(&b, &hi, 8);
        b -= ((uint64_t)k << 52) / 128;
        printf("0x%016llxull, 0x%016llxull,%s", (unsigned long long)a, (unsigned long long)b, (k & 1) ? "\n" : " ");
    }
    return 0;
}
