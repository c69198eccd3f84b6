// igemm_pk.h instantiations: f16, implicit-im2col conv activations
#include "igemm_pk.h"

SFAST_PK_UNIT(sfast::f16, 1, f16_conv)
