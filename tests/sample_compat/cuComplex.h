/* TEST-HARNESS FIXTURE (see cuda_runtime.h in this directory): cutensorMp/cutensorMp_contraction.cu:24 includes
 * <cuComplex.h> for cuComplex / make_cuComplex (:434, :521-522). */
#ifndef SAMPLE_COMPAT_CUCOMPLEX_H_
#define SAMPLE_COMPAT_CUCOMPLEX_H_
#include <hip/hip_complex.h>
typedef hipFloatComplex  cuComplex;
typedef hipDoubleComplex cuDoubleComplex;
#define make_cuComplex       make_hipFloatComplex
#define make_cuDoubleComplex make_hipDoubleComplex
#endif
