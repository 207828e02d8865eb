/*
 * sgmx_hamming_compat.h -- the function names and signatures of the reference's core/hamming.h (lines 52-112) on top
 * of libsgmx.so, for code that calls that library today and is to run unchanged on the GPU.
 *
 * Include this INSTEAD of core/hamming.h and link with -lsgmx.  The reference's functions take no context and report
 * no errors; here one process-wide context (device 0, or SGMX_HAMMING_DEVICE) is created on first use and an engine
 * error aborts with its text on stderr, where the reference would have crashed or returned garbage.  Buffers that the
 * reference allocates with malloc are malloc'ed here too: release them with free().
 *
 * Threads: the reference's functions are re-entrant; an sgmx context is not (sgmx.h: one context per host thread), so
 * every call below holds one process-wide mutex while it uses the shared context -- calls from several threads are
 * correct and run one after the other (the GPU would serialise them anyway).
 * hamming() (core/hamming.h:62), the distance of ONE pair of signatures, is a popcount loop on the host: a GPU round
 * trip for one popcount would make code that calls it in a loop orders of magnitude slower than the reference.  Every
 * function that works on sets of signatures runs on the GPU.
 *
 * Define SGMX_HAMMING_COMPAT_EXPORT before including to get external (non-static) definitions, e.g. to build a
 * drop-in libhamming.so from a one-line .c file.
 */
#ifndef SGMX_HAMMING_COMPAT_H
#define SGMX_HAMMING_COMPAT_H

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

#include "sgmx.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifdef SGMX_HAMMING_COMPAT_EXPORT
#define SGMX_HC SGMX_API
#else
#define SGMX_HC static
#endif

typedef unsigned char uint8;
typedef unsigned short uint16;
typedef unsigned int uint32;
typedef unsigned long long uint64;
typedef long long int64;

typedef sgmx_hammatch hammatch_t; /* core/hamming.h:52-56: { int qid; int bid; uint16 score; } */

static sgmx_ctx *sgmx_hc_ctx(void)
{
    static sgmx_ctx *volatile ctx;
    sgmx_ctx *c = __atomic_load_n(&ctx, __ATOMIC_ACQUIRE);
    if (!c) {
        sgmx_params p = {0};
        const char *dev = getenv("SGMX_HAMMING_DEVICE");
        p.disp_range = 2;
        p.device = dev ? atoi(dev) : 0;
        sgmx_ctx *fresh = NULL, *expected = NULL;
        if (sgmx_create(&p, &fresh) != SGMX_OK) {
            fprintf(stderr, "sgmx hamming: %s\n", sgmx_last_error(NULL));
            abort();
        }
        if (__atomic_compare_exchange_n(&ctx, &expected, fresh, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE))
            c = fresh;
        else { /* another thread won */
            sgmx_destroy(fresh);
            c = expected;
        }
    }
    return c;
}

/* The shared context is used by one thread at a time: sgmx_hc_check(CALL) runs CALL under the mutex. */
static pthread_mutex_t sgmx_hc_mutex = PTHREAD_MUTEX_INITIALIZER;

static void sgmx_hc_done(int rc)
{
    if (rc != SGMX_OK) {
        fprintf(stderr, "sgmx hamming: %s\n", sgmx_last_error(sgmx_hc_ctx()));
        abort();
    }
    pthread_mutex_unlock(&sgmx_hc_mutex);
}
#define sgmx_hc_check(call) (pthread_mutex_lock(&sgmx_hc_mutex), sgmx_hc_done(call))

/* core/hamming.h:62 -- one pair: on the host (see the head of this file) */
SGMX_HC uint16 hamming(const uint8 *bs1, const uint8 *bs2, int ncodes)
{
    unsigned d = 0;
    int k;
    for (k = 0; k < ncodes; k++)
        d += (unsigned)__builtin_popcount((unsigned)(bs1[k] ^ bs2[k]));
    return (uint16)d;
}

/* core/hamming.h:67-68 */
SGMX_HC void compute_hamming(uint16 *dis, const uint8 *a, const uint8 *b, int na, int nb, int ncodes)
{
    sgmx_hc_check(sgmx_hamming_matrix_u16(sgmx_hc_ctx(), dis, a, b, na, nb, ncodes));
}

/* core/hamming.h:75-76 */
SGMX_HC void match_hamming_count(const uint8 *bs1, const uint8 *bs2, int n1, int n2, int ht, int ncodes, size_t *nptr)
{
    sgmx_hc_check(sgmx_match_hamming_count(sgmx_hc_ctx(), bs1, bs2, n1, n2, ht, ncodes, nptr));
}

/* core/hamming.h:78-79 */
SGMX_HC void crossmatch_hamming_count(const uint8 *dbs, int n, int ht, int ncodes, size_t *nptr)
{
    sgmx_hc_check(sgmx_crossmatch_hamming_count(sgmx_hc_ctx(), dbs, n, ht, ncodes, nptr));
}

/* core/hamming.h:84-86 */
SGMX_HC void match_hamming_thres(const uint8 *bs1, const uint8 *bs2, int n1, int n2, int ht, int ncodes, size_t bufsize,
                                 hammatch_t **hmptr, size_t *nptr)
{
    sgmx_hc_check(sgmx_match_hamming_thres(sgmx_hc_ctx(), bs1, bs2, n1, n2, ht, ncodes, bufsize, hmptr, nptr));
}

/* core/hamming.h:90-92: the caller's buffers are trusted, as in the reference (sized by match_hamming_count) */
SGMX_HC size_t match_hamming_thres_prealloc(const uint8 *bs1, const uint8 *bs2, int n1, int n2, int ht, int ncodes, int *idx,
                                            uint16 *hams)
{
    size_t n = 0;
    sgmx_hc_check(sgmx_match_hamming_thres_prealloc(sgmx_hc_ctx(), bs1, bs2, n1, n2, ht, ncodes, idx, hams, (size_t)-1, &n));
    return n;
}

/* core/hamming.h:95-96 */
SGMX_HC void crossmatch_hamming(const uint8 *dbs, long n, int ht, int ncodes, long bufsize, hammatch_t **hmptr, size_t *nptr)
{
    sgmx_hc_check(sgmx_crossmatch_hamming(sgmx_hc_ctx(), dbs, n, ht, ncodes, bufsize, hmptr, nptr));
}

/* core/hamming.h:103-104 */
SGMX_HC size_t crossmatch_hamming_prealloc(const uint8 *dbs, long n, int ht, int ncodes, int *idx, uint16 *hams)
{
    size_t m = 0;
    sgmx_hc_check(sgmx_crossmatch_hamming_prealloc(sgmx_hc_ctx(), dbs, n, ht, ncodes, idx, hams, (size_t)-1, &m));
    return m;
}

/* core/hamming.h:108-109 */
SGMX_HC void compute_hamming_thread(uint16 *dis, const uint8 *a, const uint8 *b, int na, int nb, int ncodes)
{
    sgmx_hc_check(sgmx_compute_hamming_thread(sgmx_hc_ctx(), dis, a, b, na, nb, ncodes));
}

/* core/hamming.h:111-112 */
SGMX_HC size_t match_hamming_thres_nt(const uint8 *bs1, const uint8 *bs2, int n1, int n2, int ht, int ncodes, int nt, int **keys,
                                      uint16 **ham)
{
    size_t n = 0;
    sgmx_hc_check(sgmx_match_hamming_thres_nt(sgmx_hc_ctx(), bs1, bs2, n1, n2, ht, ncodes, nt, keys, ham, &n));
    return n;
}

#ifdef __cplusplus
}
#endif
#endif /* SGMX_HAMMING_COMPAT_H */
