/* tools/rstub/R_ext/Random.h -- NOT R: see ../Rinternals.h (syntax check of shim/bfmmm_rcall.cpp only). */
#ifndef BFMMM_RSTUB_RANDOM_H
#define BFMMM_RSTUB_RANDOM_H
#ifdef __cplusplus
extern "C" {
#endif
void GetRNGstate(void);
void PutRNGstate(void);
double unif_rand(void);
double norm_rand(void);
double exp_rand(void);
#ifdef __cplusplus
}
#endif
#endif
