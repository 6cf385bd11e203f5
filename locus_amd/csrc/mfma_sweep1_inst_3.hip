#define CS_ 3
#include "mfma_sweep1_inst.inc"
