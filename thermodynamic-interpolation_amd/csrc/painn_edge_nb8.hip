// painn_edge_nb8.hip -- edge-kernel instantiations for n_features = 256 (painn_edge_kernel.hpp)
#include "painn_edge_kernel.hpp"

namespace ti {
hipError_t configure_edge_nb8() { return configure_edge_nb<8>(); }
hipError_t launch_edge_nb8(bool first, bool last, int prec, const EdgeParams& p, hipStream_t st) { return launch_edge_nb<8>(first, last, prec, p, st); }
}  // namespace ti
