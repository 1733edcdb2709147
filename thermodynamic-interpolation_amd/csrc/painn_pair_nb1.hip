// painn_pair_nb1.hip -- pair-major message kernel instantiations for n_features = 32 (painn_pair_kernel.hpp)
#include "painn_pair_kernel.hpp"

namespace ti {
hipError_t configure_pair_nb1() { return configure_pair_nb<1>(); }
hipError_t launch_pair_nb1(bool first, bool last, int prec, const EdgeParams& p, hipStream_t st) { return launch_pair_nb<1>(first, last, prec, p, st); }
}  // namespace ti
