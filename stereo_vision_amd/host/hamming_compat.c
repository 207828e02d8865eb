/* libhamming_sgmx.so: the reference's core/hamming.h API (same names, same signatures) served by libsgmx.so --
 * what a maintainer links in place of core/hamming.c.  All code is in include/sgmx_hamming_compat.h. */
#define SGMX_HAMMING_COMPAT_EXPORT
#include "sgmx_hamming_compat.h"
