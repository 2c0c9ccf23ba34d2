// eval_kernel fast builds for plain evaluations (no chain pass): 2- or 3-layer networks with later layers <= 16 nodes, one weight set per
// launch, likelihood class categorical
#define NPBNN_INST_NAME pick_eval_d1_cat_plain
#define NPBNN_INST_MTI 1
#define NPBNN_INST_D 1
#define NPBNN_INST_LK 0
#define NPBNN_INST_FAST 1
#define NPBNN_INST_CHAIN false
#include "npbnn_eval_inst.inc"
