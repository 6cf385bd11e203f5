/* Shared by the glue files of this directory: turns a non-zero status of liblocus_hip into an R
 * error, as BEGIN_RCPP / END_RCPP do for C++ exceptions in the reference
 * (src/RcppExports.cpp:18,34).
 *
 * NOT COMPILED OR TESTED HERE: the image has no R, Rcpp or RcppEigen, so these
 * files cannot even be parsed against their headers in this repository.  What they forward to --
 * the C ABI of include/locus_hip.h -- is tested from plain C (tests/c/abi_smoke.c) and through
 * ctypes (tests/test_sweep_gpu.py, tests/test_resident_gpu.py) with the same argument order. */
#ifndef LOCUS_HIP_GLUE_H
#define LOCUS_HIP_GLUE_H
#include "utils.h"      /* the reference's own header: RcppEigen + locus_types.h (Map typedefs) */
#include "locus_hip.h"

static inline void locus_hip_check(int rc) {
  if (rc != 0) Rcpp::stop("locus_hip (%d): %s", rc, locus_hip_last_error());
}
#endif
