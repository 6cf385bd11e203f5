#define T_ 2
#include "mfma_sweep_inst.inc"
