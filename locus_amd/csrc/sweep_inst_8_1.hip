#define NPL_ 8
#define LOGIT_ 1
#include "sweep_inst.inc"
