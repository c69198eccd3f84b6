// igemm_pp.h instantiations: f16, lin
#include "igemm_pp.h"

SFAST_PP_UNIT(sfast::f16, 0, f16_lin)
