// igemm_pp.h instantiations: f16, conv
#include "igemm_pp.h"

SFAST_PP_UNIT(sfast::f16, 1, f16_conv)
