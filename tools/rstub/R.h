/* tools/rstub/R.h -- NOT R: see Rinternals.h in this directory (syntax check of shim/bfmmm_rcall.cpp only). */
#ifndef BFMMM_RSTUB_R_H
#define BFMMM_RSTUB_R_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
void Rprintf(const char*, ...);
void REprintf(const char*, ...);
#ifdef __cplusplus
}
#endif
#include "R_ext/Random.h"
#endif
