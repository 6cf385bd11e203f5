#define NPL_ 16
#define LOGIT_ 1
#include "sweep_inst.inc"
