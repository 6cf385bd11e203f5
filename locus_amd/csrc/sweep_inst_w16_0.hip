#define NPL_ 16
#define LOGIT_ 0
#define PCS_ 4
#include "sweep_inst.inc"
