// explicit instantiation of the 16-edge tile kernel (dedf_edge16.h)
#include <hip/hip_runtime.h>
#include "dedf_edge16.h"
using namespace dedf;
template __global__ void dedf::k_edge16<2>(Edge16Params);
