#define CS_ 1
#include "mfma_sweep1_inst.inc"
