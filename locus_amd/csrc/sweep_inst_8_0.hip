#define NPL_ 8
#define LOGIT_ 0
#include "sweep_inst.inc"
