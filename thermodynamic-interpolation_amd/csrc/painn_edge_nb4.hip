// painn_edge_nb4.hip -- edge-kernel instantiations for n_features = 128 (painn_edge_kernel.hpp)
#include "painn_edge_kernel.hpp"

namespace ti {
bool edge_uses_one_chain(int NB, int prec) { return edge_one_chain(NB, prec); }      // what the kernels are built for: the host packs to match
hipError_t configure_edge_nb4() { return configure_edge_nb<4>(); }
hipError_t launch_edge_nb4(bool first, bool last, int prec, const EdgeParams& p, hipStream_t st) { return launch_edge_nb<4>(first, last, prec, p, st); }
}  // namespace ti
