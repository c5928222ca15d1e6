// Literal drop-in include path: a program written for mrcdr/lambda-lanczos says
//     #include <lambda_lanczos/lambda_lanczos.hpp>            (the reference's README.md:20)
// Compile it with  -I <this repo>/include/compat -I <this repo>/include  instead of the reference's include directory
// and it gets lambda_lanczos::LambdaLanczos<T> backed by liblanczos_hip.so — no source line changes.
#ifndef LAMBDA_LANCZOS_COMPAT_LAMBDA_LANCZOS_HPP_
#define LAMBDA_LANCZOS_COMPAT_LAMBDA_LANCZOS_HPP_
#include "../../lambda_lanczos_hip/lambda_lanczos.hpp"
#endif
