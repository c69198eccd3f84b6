// igemm_pk.h instantiations: bf16, implicit-im2col conv activations
#include "igemm_pk.h"

SFAST_PK_UNIT(sfast::bf16, 1, bf16_conv)
