#define NPL_ 4
#define LOGIT_ 0
#include "sweep_inst.inc"
