// painn_pair_nb4.hip -- pair-major message kernel instantiations for n_features = 128 (painn_pair_kernel.hpp)
#include "painn_pair_kernel.hpp"

namespace ti {
hipError_t configure_pair_nb4() { return configure_pair_nb<4>(); }
hipError_t launch_pair_nb4(bool first, bool last, int prec, const EdgeParams& p, hipStream_t st) { return launch_pair_nb<4>(first, last, prec, p, st); }
}  // namespace ti
