#define T_ 3
#include "mfma_sweep_inst.inc"
