// Explicit instantiations of the fused kernels, one group per -DDEDF_KUNIT=n (see dedf_kernels.h).
#include "dedf_kernels.h"
#ifndef DEDF_KUNIT
#error "compile with -DDEDF_KUNIT=<0..kKernelUnits-1>"
#endif
#define DEDF_INST(unit, ...) DEDF_INST_##unit(__VA_ARGS__)
#define DEDF_SEL(...) template __global__ __VA_ARGS__;
#define DEDF_NOP(...)
#if DEDF_KUNIT == 0
#define DEDF_INST_0(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_0(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 1
#define DEDF_INST_1(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_1(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 2
#define DEDF_INST_2(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_2(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 3
#define DEDF_INST_3(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_3(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 4
#define DEDF_INST_4(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_4(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 5
#define DEDF_INST_5(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_5(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 6
#define DEDF_INST_6(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_6(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 7
#define DEDF_INST_7(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_7(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 8
#define DEDF_INST_8(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_8(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 9
#define DEDF_INST_9(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_9(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 10
#define DEDF_INST_10(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_10(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 11
#define DEDF_INST_11(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_11(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 12
#define DEDF_INST_12(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_12(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 13
#define DEDF_INST_13(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_13(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 14
#define DEDF_INST_14(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_14(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 15
#define DEDF_INST_15(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_15(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 16
#define DEDF_INST_16(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_16(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 17
#define DEDF_INST_17(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_17(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 18
#define DEDF_INST_18(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_18(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 19
#define DEDF_INST_19(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_19(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 20
#define DEDF_INST_20(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_20(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 21
#define DEDF_INST_21(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_21(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 22
#define DEDF_INST_22(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_22(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 23
#define DEDF_INST_23(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_23(...) DEDF_NOP(__VA_ARGS__)
#endif
DEDF_KERNEL_LIST(DEDF_INST)
static_assert(kKernelUnits == 24, "one DEDF_INST_n block per unit");
