#define NPL_ 4
#define LOGIT_ 1
#include "sweep_inst.inc"
