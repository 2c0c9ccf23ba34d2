// eval_kernel fast builds for NPBNN_SCHED_PERSIST_SERIAL (SPEC: the step workgroup prepares the next pass for every outcome): 2- or 3-layer
// networks with later layers <= 16 nodes, dense first layer, 3 candidate(s) per launch, likelihood class categorical
#define NPBNN_INST_NAME pick_eval_d3_cat_spec
#define NPBNN_INST_MTI 1
#define NPBNN_INST_D 3
#define NPBNN_INST_LK 0
#define NPBNN_INST_FAST 1
#define NPBNN_INST_SPEC true
#include "npbnn_eval_inst.inc"
