// Literal drop-in include path for  #include <lambda_lanczos/exponentiator.hpp>  (the reference's README.md:21);
// see lambda_lanczos.hpp in this directory.
#ifndef LAMBDA_LANCZOS_COMPAT_EXPONENTIATOR_HPP_
#define LAMBDA_LANCZOS_COMPAT_EXPONENTIATOR_HPP_
#include "../../lambda_lanczos_hip/exponentiator.hpp"
#endif
