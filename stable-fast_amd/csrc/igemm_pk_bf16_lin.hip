// igemm_pk.h instantiations: bf16, linear activations
#include "igemm_pk.h"

SFAST_PK_UNIT(sfast::bf16, 0, bf16_lin)
