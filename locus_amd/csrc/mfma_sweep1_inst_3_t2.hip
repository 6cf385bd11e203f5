#define CS_ 3
#define T_ 2
#include "mfma_sweep1_inst.inc"
