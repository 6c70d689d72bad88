// Every instantiation the library launches: X(unit, declaration).  Kept in its own header so that __graft_entry__.build() can tell WHICH units an
// edit of the list touches (a unit is rebuilt when its own lines or one of the kernel headers change, not when another unit's lines do).
#pragma once
// every instantiation the library launches: X(unit, declaration)
// (Round 5: the full-precision kernels run in the edge-aligned frame -- the instantiations with the trailing `true`.  The GENERAL form of a
//  full-precision shape is kept only where it serves the A/B switch DEDF_SO2=0: the headline shapes <2 | 3, 128, false, 128, 64> per edge and
//  table-reading.  Half precision (HP) runs in the edge frame only.  Unit numbers of dropped instantiations are left unused.)
#define DEDF_KERNEL_LIST(X)                                           \
    X(0, void k_edge<2, 128, false>(EdgeParams))                      \
    X(1, void k_edge<2, 128, true, 128, 64, false, 0, false, true>(EdgeParams)) \
    X(4, void k_edge<1, 128, true, 128, 64, false, 0, false, true>(EdgeParams)) \
    X(8, void k_edge<2, 192, true, 128, 64, false, 0, false, true>(EdgeParams)) \
    X(8, void k_edge<2, 64, true, 128, 64, false, 0, false, true>(EdgeParams))  \
    X(9, void k_edge<2, 128, true, 32, 32, false, 0, false, true>(EdgeParams))  \
    X(9, void k_edge<1, 64, true, 128, 64, false, 0, false, true>(EdgeParams))  \
    X(9, void k_edge<1, 128, true, 32, 32, false, 0, false, true>(EdgeParams))  \
    X(9, void k_node<2, true, true>(NodeParams))                      \
    X(9, void k_node<1, true, true>(NodeParams))                      \
    X(5, void k_node<2, false, false>(NodeParams))                    \
    X(5, void k_node<2, false, true>(NodeParams))                     \
    X(6, void k_node<2, true, false>(NodeParams))                     \
    X(6, void k_node<1, false, false>(NodeParams))                    \
    X(6, void k_node<1, false, true>(NodeParams))                     \
    X(6, void k_node<1, true, false>(NodeParams))                     \
    X(10, void k_node<2, false, false, true>(NodeParams))             \
    X(12, void k_edge<2, 128, false, 128, 64, false, 1>(EdgeParams))  \
    X(11, void k_radial_table<2, 128>(EdgeParams))                    \
    X(11, void k_radial_check<2, 128>(EdgeParams))                    \
    X(10, void k_radial_table<2, 192>(EdgeParams))                    \
    X(10, void k_radial_check<2, 192>(EdgeParams))                    \
    X(10, void k_radial_table<2, 128, false, 32, 32>(EdgeParams))     \
    X(10, void k_radial_check<2, 128, false, 32, 32>(EdgeParams))     \
    X(15, void k_edge<2, 64, true, 32, 32, true, 0, false, true>(EdgeParams))   \
    X(15, void k_node<2, false, true, true>(NodeParams))               \
    X(16, void k_edge<3, 128, false>(EdgeParams))                     \
    X(17, void k_node<3, false, false>(NodeParams))                   \
    X(17, void k_node<3, true, false>(NodeParams))                    \
    X(20, void k_node<3, false, false, true>(NodeParams))             \
    X(21, void k_edge<3, 64, true, 32, 32, true, 0, false, true>(EdgeParams))   \
    X(20, void k_node<3, false, true, true>(NodeParams))              \
    X(22, void k_edge<3, 128, false, 128, 64, false, 1>(EdgeParams))  \
    X(23, void k_radial_table<3, 128>(EdgeParams))                    \
    X(23, void k_radial_check<3, 128>(EdgeParams))                    \
    X(24, void k_edge<3, 128, true, 128, 64, false, 0, false, true>(EdgeParams)) \
    X(25, void k_edge<3, 64, true, 128, 64, false, 0, false, true>(EdgeParams))  \
    X(26, void k_edge<3, 64, true, 32, 32, false, 0, false, true>(EdgeParams))   \
    X(25, void k_node<3, false, true>(NodeParams))                    \
    X(26, void k_node<3, true, true>(NodeParams))                     \
    X(27, void k_edge<2, 64, true, 32, 32, false, 0, false, true>(EdgeParams))   \
    X(30, void k_edge<2, 64, true, 32, 32, true, 0, true, true>(EdgeParams))  \
    X(31, void k_edge<3, 64, true, 32, 32, true, 0, true, true>(EdgeParams))  \
    X(32, void k_edge<2, 128, false, 128, 64, false, 0, false, true>(EdgeParams)) \
    X(33, void k_edge<2, 128, false, 128, 64, false, 1, false, true>(EdgeParams)) \
    X(34, void k_edge<2, 128, false, 32, 32, false, 0, false, true>(EdgeParams))  \
    X(34, void k_edge<1, 128, false, 128, 64, false, 0, false, true>(EdgeParams)) \
    X(35, void k_edge<2, 128, false, 32, 32, false, 1, false, true>(EdgeParams))  \
    X(35, void k_edge<1, 64, false, 128, 64, false, 0, false, true>(EdgeParams))  \
    X(36, void k_edge<2, 192, false, 128, 64, false, 0, false, true>(EdgeParams)) \
    X(37, void k_edge<2, 192, false, 128, 64, false, 1, false, true>(EdgeParams)) \
    X(38, void k_edge<2, 64, false, 128, 64, false, 0, false, true>(EdgeParams))  \
    X(39, void k_edge<2, 64, false, 32, 32, false, 0, false, true>(EdgeParams))   \
    X(39, void k_edge<1, 128, false, 32, 32, false, 0, false, true>(EdgeParams))  \
    X(40, void k_edge<3, 128, false, 128, 64, false, 0, false, true>(EdgeParams)) \
    X(41, void k_edge<3, 128, false, 128, 64, false, 1, false, true>(EdgeParams)) \
    X(42, void k_edge<3, 64, false, 128, 64, false, 0, false, true>(EdgeParams))  \
    X(43, void k_edge<3, 64, false, 32, 32, false, 0, false, true>(EdgeParams))   \
    X(44, void k_edge<2, 64, false, 32, 32, true, 0, false, true>(EdgeParams))    \
    X(45, void k_edge<2, 64, false, 32, 32, true, 0, true, true>(EdgeParams))     \
    X(46, void k_edge<3, 64, false, 32, 32, true, 0, false, true>(EdgeParams))    \
    X(47, void k_edge<3, 64, false, 32, 32, true, 0, true, true>(EdgeParams))     \
    X(48, void k_edge<2, 128, false, 128, 64, false, 0, false, true, true>(EdgeParams)) \
    X(49, void k_edge<2, 128, false, 128, 64, false, 1, false, true, true>(EdgeParams)) \
    X(50, void k_edge<3, 128, false, 128, 64, false, 0, false, true, true>(EdgeParams)) \
    X(51, void k_edge<3, 128, false, 128, 64, false, 1, false, true, true>(EdgeParams)) \
    X(52, void k_edge<2, 128, false, 32, 32, false, 0, false, true, true>(EdgeParams))  \
    X(53, void k_edge<2, 128, false, 32, 32, false, 1, false, true, true>(EdgeParams))  \
    X(54, void k_edge<2, 192, false, 128, 64, false, 0, false, true, true>(EdgeParams)) \
    X(55, void k_edge<2, 192, false, 128, 64, false, 1, false, true, true>(EdgeParams)) \
    X(56, void k_edge<2, 128, true, 128, 64, false, 0, false, true, true>(EdgeParams)) \
    X(57, void k_edge<1, 128, false, 128, 64, false, 0, false, true, true>(EdgeParams)) \
    X(58, void k_edge<2, 64, false, 128, 64, false, 0, false, true, true>(EdgeParams))  \
    X(59, void k_edge<3, 64, false, 128, 64, false, 0, false, true, true>(EdgeParams))  \
    X(60, void k_edge<1, 64, false, 128, 64, false, 0, false, true, true>(EdgeParams))  \
    X(61, void k_edge<1, 128, false, 128, 64, false, 1, false, true>(EdgeParams))       \
    X(61, void k_radial_table<1, 128>(EdgeParams))                                      \
    X(61, void k_radial_check<1, 128>(EdgeParams))                                      \
    X(62, void k_edge<1, 128, false, 128, 64, false, 1, false, true, true>(EdgeParams))
// (units 48-55: the score head with query_time_encoding -- trailing `true`: the pose's time row joins the 0e block of the gathered message;
//  52-55: the two other lmax-2 score-head shapes the reference ships, radial MLP [128,32,32] and the 192-wide pre-linear; 56: half precision; 57: lmax 1, round 6;
//  58-60: query_time_encoding WITHOUT edge_time_encoding -- the reference constructor's default -- at lmax 2 / 3 / 1: the 64-wide pre-linear of the EBM shapes;
//  61-62: the sampler's radial table at lmax 1 -- BASELINE config C1 -- plain and with query_time_encoding)
constexpr int kKernelUnits = 63;
