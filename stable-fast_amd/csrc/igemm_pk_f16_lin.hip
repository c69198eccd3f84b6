// igemm_pk.h instantiations: f16, linear activations
#include "igemm_pk.h"

SFAST_PK_UNIT(sfast::f16, 0, f16_lin)
