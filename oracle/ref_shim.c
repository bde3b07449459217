/*
 * ref_shim.c -- thin exports over the parts of the reference that compile
 * directly from /root/reference with gcc (no generated config.h needed):
 * libcommon/mathfunc.c, libcommon/memalloc.c, libtree/chooseln_cache.c,
 * libcommon/fminsearch.cpp.  Built into oracle/_ref/libcaferef.so by
 * oracle/Makefile; used only by tests to pin oracle/cafe_oracle.c bit-for-bit.
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference source text; it
 * only calls the reference's public functions through its own headers.
 */
#include <math.h>
#include <mathfunc.h>
#include <chooseln_cache.h>

/* dense dump of the reference's chooseln cache after chooseln_cache_init2(size)
 * (libtree/chooseln_cache.c:59-68); out[n*(size+1)+x] for x <= min(n,size), else NaN */
void ref_chooseln_table(int size, double *out)
{
    struct chooseln_cache c = {0, 0};
    chooseln_cache_init2(&c, size);
    for (int n = 0; n < 2 * size; n++) {
        for (int x = 0; x <= size; x++) {
            out[(long)n * (size + 1) + x] = (x <= n) ? chooseln_get2(&c, n, x) : NAN;
        }
    }
    chooseln_cache_free2(&c);
}

/* fminsearch_min with explicit tolerances (libcommon/fminsearch.cpp:264-302) */
int ref_fminsearch(math_func eq, int N, void *args, double *x0, double tolx, double tolf,
                   double *xmin, double *fmin, int *bymax)
{
    pFMinSearch pfm = fminsearch_new_with_eq(eq, N, args);
    pfm->tolx = tolx;
    pfm->tolf = tolf;
    fminsearch_min(pfm, x0);
    double *re = fminsearch_get_minX(pfm);
    for (int i = 0; i < N; i++) xmin[i] = re[i];
    *fmin = fminsearch_get_minF(pfm);
    *bymax = pfm->bymax;
    int iters = pfm->iters;
    fminsearch_free(pfm);
    return iters;
}
