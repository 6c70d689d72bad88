// Explicit instantiations of the fused kernels, one group per -DDEDF_KUNIT=n (see dedf_kernels.h / dedf_kernel_list.h).
// (The selector blocks below cover units 0 .. 63 so that this file -- part of every unit's build signature -- does not change when a unit is added.)
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
#if DEDF_KUNIT == 24
#define DEDF_INST_24(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_24(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 25
#define DEDF_INST_25(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_25(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 26
#define DEDF_INST_26(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_26(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 27
#define DEDF_INST_27(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_27(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 28
#define DEDF_INST_28(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_28(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 29
#define DEDF_INST_29(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_29(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 30
#define DEDF_INST_30(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_30(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 31
#define DEDF_INST_31(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_31(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 32
#define DEDF_INST_32(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_32(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 33
#define DEDF_INST_33(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_33(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 34
#define DEDF_INST_34(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_34(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 35
#define DEDF_INST_35(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_35(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 36
#define DEDF_INST_36(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_36(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 37
#define DEDF_INST_37(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_37(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 38
#define DEDF_INST_38(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_38(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 39
#define DEDF_INST_39(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_39(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 40
#define DEDF_INST_40(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_40(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 41
#define DEDF_INST_41(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_41(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 42
#define DEDF_INST_42(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_42(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 43
#define DEDF_INST_43(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_43(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 44
#define DEDF_INST_44(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_44(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 45
#define DEDF_INST_45(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_45(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 46
#define DEDF_INST_46(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_46(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 47
#define DEDF_INST_47(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_47(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 48
#define DEDF_INST_48(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_48(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 49
#define DEDF_INST_49(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_49(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 50
#define DEDF_INST_50(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_50(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 51
#define DEDF_INST_51(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_51(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 52
#define DEDF_INST_52(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_52(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 53
#define DEDF_INST_53(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_53(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 54
#define DEDF_INST_54(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_54(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 55
#define DEDF_INST_55(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_55(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 56
#define DEDF_INST_56(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_56(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 57
#define DEDF_INST_57(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_57(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 58
#define DEDF_INST_58(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_58(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 59
#define DEDF_INST_59(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_59(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 60
#define DEDF_INST_60(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_60(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 61
#define DEDF_INST_61(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_61(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 62
#define DEDF_INST_62(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_62(...) DEDF_NOP(__VA_ARGS__)
#endif
#if DEDF_KUNIT == 63
#define DEDF_INST_63(...) DEDF_SEL(__VA_ARGS__)
#else
#define DEDF_INST_63(...) DEDF_NOP(__VA_ARGS__)
#endif
DEDF_KERNEL_LIST(DEDF_INST)
static_assert(kKernelUnits <= 64, "one DEDF_INST_n block per unit");
