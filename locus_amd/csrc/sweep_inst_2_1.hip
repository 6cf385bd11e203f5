#define NPL_ 2
#define LOGIT_ 1
#include "sweep_inst.inc"
