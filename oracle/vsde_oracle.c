/*
 * TEST INFRASTRUCTURE ONLY -- see vsde_oracle_impl.h for what this restates and
 * which reference lines each function follows.  Two instantiations:
 *   *_f32 : float arithmetic with libm expf/tanhf (what the reference computes in)
 *   *_f64 : double arithmetic ("truth" used to set the fp32 tolerances)
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SUFFIX _f32
#define EXP expf
#define TANH tanhf
#define LOG logf
#define LOG1P log1pf
#define EXPM1 expm1f
#define POW(a, b) powf((float)(a), (float)(b))
#include "vsde_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef EXP
#undef TANH
#undef LOG
#undef LOG1P
#undef EXPM1
#undef POW

#define REAL double
#define SUFFIX _f64
#define EXP exp
#define TANH tanh
#define LOG log
#define LOG1P log1p
#define EXPM1 expm1
#define POW(a, b) pow((double)(a), (double)(b))
#include "vsde_oracle_impl.h"

int vsde_oracle_abi_version(void) { return 1; }
