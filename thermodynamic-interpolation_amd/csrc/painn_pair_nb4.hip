// painn_pair_nb4.hip -- pair-major message kernel instantiations for n_features = 128 (painn_pair_kernel.hpp)
#include "painn_pair_kernel.hpp"

namespace ti {
bool pair_uses_partials() { return pair_writes_partials(); }          // what the kernels are built for: the host reduces (or not) to match
hipError_t configure_pair_nb4() { return configure_pair_nb<4>(); }
hipError_t launch_pair_nb4(bool first, bool last, int prec, const EdgeParams& p, hipStream_t st) { return launch_pair_nb<4>(first, last, prec, p, st); }
}  // namespace ti
