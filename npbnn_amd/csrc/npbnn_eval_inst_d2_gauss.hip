// eval_kernel builds: later layers <= 16 nodes, 2 candidate(s) per launch, likelihood class Gaussian
#define NPBNN_INST_NAME pick_eval_d2_gauss
#define NPBNN_INST_MTI 1
#define NPBNN_INST_D 2
#define NPBNN_INST_LK 1
#include "npbnn_eval_inst.inc"
