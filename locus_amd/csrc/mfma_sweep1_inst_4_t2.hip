#define CS_ 4
#define T_ 2
#include "mfma_sweep1_inst.inc"
