// eval_kernel fast builds: 2- or 3-layer networks with later layers <= 16 nodes, 1 candidate(s) per launch, likelihood class categorical
#define NPBNN_INST_NAME pick_eval_d1_cat_fast
#define NPBNN_INST_MTI 1
#define NPBNN_INST_D 1
#define NPBNN_INST_LK 0
#define NPBNN_INST_FAST 1
#include "npbnn_eval_inst.inc"
