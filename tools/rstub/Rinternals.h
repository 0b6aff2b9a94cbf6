/* tools/rstub/Rinternals.h -- NOT R.  Declarations (no definitions) of the part of R's public C API that
 * shim/bfmmm_rcall.cpp uses, with the signatures documented in "Writing R Extensions" (R >= 3.5), so that
 * `g++ -fsyntax-only -I tools/rstub shim/bfmmm_rcall.cpp` can type-check the shim in an image without R
 * (tests/test_shim_syntax.py).  Nothing links against this; a real build uses R's own headers. */
#ifndef BFMMM_RSTUB_RINTERNALS_H
#define BFMMM_RSTUB_RINTERNALS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int R_len_t;
typedef unsigned int SEXPTYPE;
typedef enum { FALSE = 0, TRUE } Rboolean;

#define NILSXP 0
#define SYMSXP 1
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19

extern SEXP R_NilValue, R_NamesSymbol, R_DimSymbol, R_DimNamesSymbol, R_NaString;
extern double R_NaReal, R_NaN, R_PosInf, R_NegInf;
extern int R_NaInt;
#define NA_INTEGER R_NaInt
#define NA_REAL R_NaReal
#define NA_LOGICAL R_NaInt

int TYPEOF(SEXP x);
int LENGTH(SEXP x);
R_xlen_t XLENGTH(SEXP x);
double* REAL(SEXP x);
int* INTEGER(SEXP x);
int* LOGICAL(SEXP x);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
const char* CHAR(SEXP x);

SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)

SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_coerceVector(SEXP, SEXPTYPE);
SEXP Rf_mkChar(const char*);
SEXP Rf_mkString(const char*);
SEXP Rf_ScalarReal(double);
SEXP Rf_ScalarInteger(int);
SEXP Rf_ScalarLogical(int);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
int Rf_asLogical(SEXP);
R_len_t Rf_length(SEXP);
R_xlen_t Rf_xlength(SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
Rboolean Rf_isMatrix(SEXP);
Rboolean Rf_isNull(SEXP);
Rboolean Rf_isReal(SEXP);
Rboolean Rf_isInteger(SEXP);
Rboolean Rf_isString(SEXP);
#if defined(__GNUC__)
void Rf_error(const char*, ...) __attribute__((noreturn));
#else
void Rf_error(const char*, ...);
#endif
void Rf_warning(const char*, ...);
char* R_alloc(size_t, int);
Rboolean R_ToplevelExec(void (*fun)(void*), void* data);
void R_CheckUserInterrupt(void);

#ifdef __cplusplus
}
#endif
#endif
