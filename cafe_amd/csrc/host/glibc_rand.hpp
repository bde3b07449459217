// glibc_rand.hpp -- a private copy of glibc's rand() stream (the reference's unifrnd), single and bulk draws
// (part of the host driver, cafe_host.cpp; split out in round 4 so that the session file holds the commands only)
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace cafehost_impl {

// The reference draws from glibc's process-global rand() (libcommon/mathfunc.c:91-94).  Other libraries
// in the process (e.g. a collective backend) may call rand() too, so the session keeps a PRIVATE copy of
// the same generator: rand() is random() on the default TYPE_3 state (128-byte table), which random_r
// reproduces draw for draw for the same seed.
struct GlibcRand {
    struct random_data rd;
    char state[128];
    GlibcRand()
    {
        memset(&rd, 0, sizeof rd);
        memset(state, 0, sizeof state);
        initstate_r(1, state, sizeof state, &rd);  // glibc's state before any srand()
    }
    unsigned long long draws = 0;   // values taken from the stream so far (single and bulk)
    void seed(unsigned v) { srandom_r(v, &rd); }
    double unifrnd()
    {
        int32_t r = 0;
        random_r(&rd, &r);
        ++draws;
        return r / (RAND_MAX + 1.0);
    }
    void skip(unsigned long long n)
    {
        int32_t r = 0;
        for (unsigned long long i = 0; i < n; ++i) random_r(&rd, &r);
        draws += n;
    }
    static double to_unit(int32_t r) { return r / (RAND_MAX + 1.0); }
    // The next n values of rand() in one tight loop (the Monte-Carlo null draws 15 million of them): the additive
    // feedback step of glibc's random_r for its default TYPE_3 generator -- *fptr += *rptr, result = *fptr >> 1,
    // both pointers advance and wrap -- on the generator's own state, so single draws before and after continue
    // the same stream.  Any other generator type falls back to random_r.
    void fill_raw(int32_t* out, size_t n)
    {
        draws += n;
        if (rd.rand_type != 3 || !rd.fptr || !rd.rptr || !rd.end_ptr || !rd.state) {
            for (size_t i = 0; i < n; ++i) random_r(&rd, &out[i]);
            return;
        }
        int32_t *f = rd.fptr, *r = rd.rptr, *const end = rd.end_ptr, *const st = rd.state;
        for (size_t i = 0; i < n; ++i) {
            const uint32_t val = (uint32_t)*f + (uint32_t)*r;
            *f = (int32_t)val;
            out[i] = (int32_t)(val >> 1);
            ++f;
            if (f >= end) {
                f = st;
                ++r;
            } else {
                ++r;
                if (r >= end) r = st;
            }
        }
        rd.fptr = f;
        rd.rptr = r;
    }
};

}  // namespace cafehost_impl
