// eval_kernel builds: later layers <= 128 nodes, 1 candidate(s) per launch, likelihood class Gaussian
#define NPBNN_INST_NAME pick_eval_mti8_gauss
#define NPBNN_INST_MTI 8
#define NPBNN_INST_D 1
#define NPBNN_INST_LK 1
#include "npbnn_eval_inst.inc"
