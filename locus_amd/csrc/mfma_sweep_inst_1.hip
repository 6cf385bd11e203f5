#define T_ 1
#include "mfma_sweep_inst.inc"
