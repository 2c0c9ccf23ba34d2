// eval_kernel builds: later layers <= 128 nodes, 1 candidate(s) per launch, likelihood class categorical/none
#define NPBNN_INST_NAME pick_eval_mti8_cat
#define NPBNN_INST_MTI 8
#define NPBNN_INST_D 1
#define NPBNN_INST_LK 0
#include "npbnn_eval_inst.inc"
