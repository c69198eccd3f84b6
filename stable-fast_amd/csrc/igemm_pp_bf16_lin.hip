// igemm_pp.h instantiations: bf16, lin
#include "igemm_pp.h"

SFAST_PP_UNIT(sfast::bf16, 0, bf16_lin)
