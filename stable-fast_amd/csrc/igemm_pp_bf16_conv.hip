// igemm_pp.h instantiations: bf16, conv
#include "igemm_pp.h"

SFAST_PP_UNIT(sfast::bf16, 1, bf16_conv)
