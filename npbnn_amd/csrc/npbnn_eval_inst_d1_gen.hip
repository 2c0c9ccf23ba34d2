// eval_kernel builds: later layers <= 16 nodes, 1 candidate(s) per launch, likelihood class float64 row-wise
#define NPBNN_INST_NAME pick_eval_d1_gen
#define NPBNN_INST_MTI 1
#define NPBNN_INST_D 1
#define NPBNN_INST_LK 2
#include "npbnn_eval_inst.inc"
