// painn_edge_nb2.hip -- edge-kernel instantiations for n_features = 64 (painn_edge_kernel.hpp)
#include "painn_edge_kernel.hpp"

namespace ti {
hipError_t configure_edge_nb2() { return configure_edge_nb<2>(); }
hipError_t launch_edge_nb2(bool first, bool last, int prec, const EdgeParams& p, hipStream_t st) { return launch_edge_nb<2>(first, last, prec, p, st); }
}  // namespace ti
