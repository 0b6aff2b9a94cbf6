/* tools/rstub/R_ext/Rdynload.h -- NOT R: see ../Rinternals.h (syntax check of shim/bfmmm_rcall.cpp only). */
#ifndef BFMMM_RSTUB_RDYNLOAD_H
#define BFMMM_RSTUB_RDYNLOAD_H
#include "../Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void* (*DL_FUNC)(void);
typedef struct _DllInfo DllInfo;
typedef struct { const char* name; DL_FUNC fun; int numArgs; void* types; } R_CMethodDef;
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
int R_registerRoutines(DllInfo* info, const R_CMethodDef* const croutines, const R_CallMethodDef* const callRoutines,
                       const R_FortranMethodDef* const fortranRoutines, const R_ExternalMethodDef* const externalRoutines);
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value);
Rboolean R_forceSymbols(DllInfo* info, Rboolean value);
#ifdef __cplusplus
}
#endif
#endif
