// eval_kernel builds: later layers <= 16 nodes, 2 candidate(s) per launch, likelihood class categorical/none
#define NPBNN_INST_NAME pick_eval_d2_cat
#define NPBNN_INST_MTI 1
#define NPBNN_INST_D 2
#define NPBNN_INST_LK 0
#include "npbnn_eval_inst.inc"
