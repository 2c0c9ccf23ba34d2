// Explicit instantiations of eval_kernel<MT0, MTI=1, F16, D=1, GEN=true> (one translation unit per group so they build in parallel).
#include "npbnn_kernels.hip.h"

namespace npbnn {

template <bool F16>
static eval_fn_t pick_mt0(int mt0) {
    switch (mt0) {
        case 1: return eval_kernel<1, 1, F16, 1, true>;
        case 2: return eval_kernel<2, 1, F16, 1, true>;
        case 3: return eval_kernel<3, 1, F16, 1, true>;
        case 4: return eval_kernel<4, 1, F16, 1, true>;
        case 5: return eval_kernel<5, 1, F16, 1, true>;
        case 6: return eval_kernel<6, 1, F16, 1, true>;
        case 7: return eval_kernel<7, 1, F16, 1, true>;
        default: return eval_kernel<8, 1, F16, 1, true>;
    }
}

eval_fn_t pick_eval_d1gen(int mt0, int f16) { return f16 ? pick_mt0<true>(mt0) : pick_mt0<false>(mt0); }

}  // namespace npbnn
