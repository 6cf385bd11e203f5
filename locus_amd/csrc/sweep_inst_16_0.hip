#define NPL_ 16
#define LOGIT_ 0
#include "sweep_inst.inc"
