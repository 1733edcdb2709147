// painn_pair_nb2.hip -- pair-major message kernel instantiations for n_features = 64 (painn_pair_kernel.hpp)
#include "painn_pair_kernel.hpp"

namespace ti {
hipError_t configure_pair_nb2() { return configure_pair_nb<2>(); }
hipError_t launch_pair_nb2(bool first, bool last, int prec, const EdgeParams& p, hipStream_t st) { return launch_pair_nb<2>(first, last, prec, p, st); }
}  // namespace ti
