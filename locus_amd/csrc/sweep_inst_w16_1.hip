#define NPL_ 16
#define LOGIT_ 1
#define PCS_ 4
#include "sweep_inst.inc"
