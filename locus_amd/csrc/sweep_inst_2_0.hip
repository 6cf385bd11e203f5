#define NPL_ 2
#define LOGIT_ 0
#include "sweep_inst.inc"
