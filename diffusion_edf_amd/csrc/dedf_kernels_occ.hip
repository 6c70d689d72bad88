// explicit instantiations of the occupancy variant (dedf_kernels_occ.h)
#include "dedf_kernels_occ.h"
template __global__ void k_edge_occ<1, 128, false>(EdgeParams);
