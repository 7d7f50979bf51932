// One translation unit per (NT, BW, variant) instantiation of the propagator kernels (parallel builds, and
// the MFMA register form can be chosen per instantiation -- see Makefile).
//   JQ_VARIANT 0: slab kernels, Neumann   1: slab kernels, Jacobi   2: cooperative (row-split) kernels
//              3: lane kernels (one lane per column; JQ_NT = padded Hilbert dimension NP, JQ_BW unused)
//              6: cooperative kernels of the implicit-midpoint integrator
//              7: quad-layout kernels of the implicit-midpoint integrator (JQ_BW = 7)
//              8: quad-layout slab kernels with 1 and 2 slabs per workgroup (JQ_BW = 7; variant 0 holds the 3-slab ones)
//              5: row-lane kernels of the implicit-midpoint integrator (JQ_NT = NPJ)
//              9: cooperative-quad kernels (one 16-row block per wave; single evaluations / small ensembles; JQ_BW = 7)
//              4: row-lane kernels (one lane per (row, column); JQ_NT = padded row length NPJ, JQ_BW unused)
//             11: kernels with the low-rank full leakage weights compiled in: quad layout with one slab per workgroup (JQ_BW = 7),
//                 slab kernels (other JQ_BW; built for <1, 0> and <6, 5>, which have no cooperative sibling)
//             12: quad-layout backward sweep with the state and the adjoint chain of a column quad on two waves (JQ_BW = 7)
#if !defined(JQ_NT) || !defined(JQ_BW) || !defined(JQ_VARIANT)
#error "compile with -DJQ_NT=<tiles> -DJQ_BW=<band> -DJQ_VARIANT=<0..12>"
#endif
#if JQ_VARIANT == 9     // cooperative-quad (latency) kernels of the JQ_BW_T4 structure (JQ_BW = 7)
#include "jq_cq_split_kernels.h"
template __global__ void k_forward_cq<JQ_NT, false>(PropArgs);
template __global__ void k_backward_cq<JQ_NT, false, false>(PropArgs);
template __global__ void k_backward_cq<JQ_NT, false, true>(PropArgs);      // (control q acts on subsystem q only)
template __global__ void k_forward_cq<JQ_NT, true>(PropArgs);      // (odd number of Neumann terms)
template __global__ void k_forward_cq<JQ_NT, false, 2>(PropArgs);      // (two column quads per workgroup: 257 .. 512 quads)
template __global__ void k_forward_cq<JQ_NT, true, 2>(PropArgs);
template __global__ void k_backward_cq<JQ_NT, true, false>(PropArgs);
template __global__ void k_backward_cq<JQ_NT, true, true>(PropArgs);
template __global__ void k_forward_cq<JQ_NT, false, 1, true>(PropArgs);      // (full leakage weights, real, rank <= 4: CqW)
template __global__ void k_forward_cq<JQ_NT, true, 1, true>(PropArgs);
template __global__ void k_backward_cq<JQ_NT, false, false, true>(PropArgs);
template __global__ void k_backward_cq<JQ_NT, false, true, true>(PropArgs);
template __global__ void k_backward_cq<JQ_NT, true, false, true>(PropArgs);
template __global__ void k_backward_cq<JQ_NT, true, true, true>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, false, false>(PropArgs);     // (backward sweep on three workgroups per column quad)
template __global__ void k_backward_cq3<JQ_NT, false, true>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, true, false>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, true, true>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, false, false, 2>(PropArgs);     // (two workgroups per column quad: state | adjoint + traces)
template __global__ void k_backward_cq3<JQ_NT, false, true, 2>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, true, false, 2>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, true, true, 2>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, false, false, 3, true>(PropArgs);     // (... with full leakage weights, real, rank <= 4)
template __global__ void k_backward_cq3<JQ_NT, false, true, 3, true>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, true, false, 3, true>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, true, true, 3, true>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, false, false, 2, true>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, false, true, 2, true>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, true, false, 2, true>(PropArgs);
template __global__ void k_backward_cq3<JQ_NT, true, true, 2, true>(PropArgs);
#if JQ_NT == 2      // the dense policy: 17 .. 32 levels without the 4 x 4 x n structure (jq_cq_kernels.h CoopQ<2, true>)
template __global__ void k_forward_cq<2, false, 1, false, true>(PropArgs);
template __global__ void k_forward_cq<2, true, 1, false, true>(PropArgs);
template __global__ void k_backward_cq<2, false, false, false, true>(PropArgs);
template __global__ void k_backward_cq<2, true, false, false, true>(PropArgs);
template __global__ void k_backward_cq3<2, false, false, 3, false, true>(PropArgs);     // (... with the backward sweep on three / two workgroups per column quad)
template __global__ void k_backward_cq3<2, true, false, 3, false, true>(PropArgs);
template __global__ void k_backward_cq3<2, false, false, 2, false, true>(PropArgs);
template __global__ void k_backward_cq3<2, true, false, 2, false, true>(PropArgs);
#endif
#elif JQ_VARIANT == 12  // quad layout, backward sweep split over two waves per column quad (mid-size ensembles)
#include "jq_quad_split_kernels.h"
template __global__ void k_backward_qsplit<JQ_NT, false, 4>(PropArgs);      // (four quads = one slab per workgroup: two waves per SIMD)
template __global__ void k_backward_qsplit<JQ_NT, true, 4>(PropArgs);       // (ORD: control q acts on subsystem q only)
template __global__ void k_backward_qsplit<JQ_NT, false, 2>(PropArgs);      // (two quads per workgroup: one wave per SIMD)
template __global__ void k_backward_qsplit<JQ_NT, true, 2>(PropArgs);
template __global__ void k_backward_qsplit<JQ_NT, true, 4, true>(PropArgs);      // (RIDE: three single-subsystem controls, all trace products ride along)
template __global__ void k_backward_qsplit<JQ_NT, true, 2, true>(PropArgs);
#elif JQ_VARIANT == 10  // cooperative-quad kernels of the implicit-midpoint integrator (JQ_BW = 7, N = 4)
#include "jq_cq_imr_kernels.h"
template __global__ void k_forward_cq_imr<JQ_NT>(PropArgs);
template __global__ void k_backward_cq_imr<JQ_NT>(PropArgs);
template __global__ void k_backward_cq_imr3<JQ_NT>(PropArgs);      // (three workgroups per evaluation: state chain, adjoint chain, trace products)
#if JQ_NT <= 6
template __global__ void k_backward_cq_imr2<JQ_NT>(PropArgs);      // (state and adjoint chain on two sets of waves)
#endif
#if JQ_NT == 2      // the dense policy (17 .. 32 levels without the structure)
template __global__ void k_forward_cq_imr<2, true>(PropArgs);
template __global__ void k_backward_cq_imr<2, true>(PropArgs);
template __global__ void k_backward_cq_imr3<2, true>(PropArgs);
#endif
#elif JQ_VARIANT == 7
#include "jq_quad_imr_kernels.h"
template __global__ void k_forward_quad_imr<JQ_NT, 1>(PropArgs);
template __global__ void k_backward_quad_imr<JQ_NT, 1>(PropArgs);
// (SPW = 2 -- two slabs per workgroup, two waves per SIMD, operators re-read from LDS per application -- was measured in round 3:
//  cnot3 x 3 072 samples 0.323 s against 0.225 s for three rounds of SPW = 1: the 36 LDS reads per application saturate the
//  LDS port with eight waves per CU; not instantiated)
#elif JQ_VARIANT == 6
#include "jq_coop_imr_kernels.h"
template __global__ void k_forward_coop_imr<JQ_NT, JQ_BW, (JQ_NT > 6)>(PropArgs);
template __global__ void k_backward_coop_imr<JQ_NT, JQ_BW, (JQ_NT > 6)>(PropArgs);
#if JQ_NT >= 2                    // N > 16 columns per evaluation (needs Ntot > 16): one workgroup per evaluation, its parts in turn
template __global__ void k_forward_coop_imr_parts<JQ_NT, JQ_BW, (JQ_NT > 6)>(PropArgs);
template __global__ void k_backward_coop_imr_parts<JQ_NT, JQ_BW, (JQ_NT > 6)>(PropArgs);
#endif
#if JQ_NT == 6 && JQ_BW == 5      // dense 96 x 96: both images of a step do not fit the LDS -- read from HBM / L2 per product
template __global__ void k_forward_coop_imr<6, 5, true>(PropArgs);
template __global__ void k_backward_coop_imr<6, 5, true>(PropArgs);
template __global__ void k_forward_coop_imr_parts<6, 5, true>(PropArgs);
template __global__ void k_backward_coop_imr_parts<6, 5, true>(PropArgs);
#endif
#elif JQ_VARIANT == 5
#include "jq_rowlane_imr_kernels.h"
template __global__ void k_forward_rowlane_imr<JQ_NT>(PropArgs);
template __global__ void k_backward_rowlane_imr<JQ_NT>(PropArgs);
template __global__ void k_backward_rowlane_imr2<JQ_NT>(PropArgs);      // (state and adjoint chain on two waves)
#elif JQ_VARIANT == 4
#include "jq_rowlane_kernels.h"
template __global__ void k_forward_rowlane<JQ_NT>(PropArgs);
template __global__ void k_forward_rowlane<JQ_NT, false, true>(PropArgs);     // (state history: traceobj_verbose)
template __global__ void k_backward_rowlane<JQ_NT>(PropArgs);
template __global__ void k_backward_rowlane2<JQ_NT>(PropArgs);      // (state and adjoint chain on two waves)
template __global__ void k_backward_rowlane3<JQ_NT>(PropArgs);      // (state chain, adjoint chain and traces on three waves)
template __global__ void k_forward_rowlane<JQ_NT, true, true>(PropArgs);      // (low-rank full leakage weights, jq_update_wmat; history if asked for)
template __global__ void k_backward_rowlane<JQ_NT, true>(PropArgs);
#elif JQ_VARIANT == 3
#include "jq_lane_kernels.h"
template __global__ void k_forward_lane<JQ_NT>(PropArgs);
template __global__ void k_backward_lane<JQ_NT>(PropArgs);
template __global__ void k_init_state_lane<JQ_NT>(double*, long long, const double*, int, long long);
template __global__ void k_terminal_lane<JQ_NT>(double*, long long, const double*, const double*, int, int, double, double*);
#elif JQ_VARIANT == 2
#include "jq_coop_kernels.h"
template __global__ void k_forward_coop<JQ_NT, JQ_BW>(PropArgs);
template __global__ void k_backward_coop<JQ_NT, JQ_BW>(PropArgs);
#else
#include "jq_kernels.h"
#ifndef JQ_MINW_MAXNT
#define JQ_MINW_MAXNT 2      // tile counts up to which two workgroups share a CU (slab kernels)
#endif
#define JQ_MINW ((JQ_NT <= JQ_MINW_MAXNT) ? 2 : 1)
#if JQ_BW == 7 && JQ_VARIANT == 11  // quad layout, one slab per workgroup, full leakage weights (jq_update_wmat)
template __global__ void k_forward<JQ_NT, JQ_BW, 1, false, true>(PropArgs);
template __global__ void k_backward<JQ_NT, JQ_BW, 1, false, true>(PropArgs);
#elif JQ_VARIANT == 11              // slab kernels with the full leakage weights
template __global__ void k_forward<JQ_NT, JQ_BW, JQ_MINW, false, true>(PropArgs);
template __global__ void k_backward<JQ_NT, JQ_BW, JQ_MINW, false, true>(PropArgs);
#elif JQ_VARIANT == 13              // ... with the Jacobi solver too (ABI 5)
template __global__ void k_forward<JQ_NT, JQ_BW, JQ_MINW, true, true>(PropArgs);
template __global__ void k_backward<JQ_NT, JQ_BW, JQ_MINW, true, true>(PropArgs);
#elif JQ_BW == 7 && JQ_VARIANT == 8   // quad layout, workgroups of 4 / 8 waves (1 / 2 slabs): built with the max-ILP scheduler (Makefile)
template __global__ void k_forward<JQ_NT, JQ_BW, 1, false>(PropArgs);
template __global__ void k_backward<JQ_NT, JQ_BW, 1, false>(PropArgs);
template __global__ void k_forward<JQ_NT, JQ_BW, 2, false>(PropArgs);
template __global__ void k_backward<JQ_NT, JQ_BW, 2, false>(PropArgs);
#elif JQ_BW == 7                    // quad layout, workgroups of 12 waves (3 slabs, 168 registers per wave): default scheduler
template __global__ void k_forward<JQ_NT, JQ_BW, 3, false>(PropArgs);
template __global__ void k_backward<JQ_NT, JQ_BW, 3, false>(PropArgs);
template __global__ void k_backward<JQ_NT, JQ_BW, 3, false, false, true>(PropArgs);      // (UNI: one ensemble sample per wave)
template __global__ void k_backward<JQ_NT, JQ_BW, 3, false, false, true, true>(PropArgs);      // (UNI + ORD: control q on subsystem q only)
#else
template __global__ void k_forward<JQ_NT, JQ_BW, JQ_MINW, (JQ_VARIANT == 1)>(PropArgs);
template __global__ void k_backward<JQ_NT, JQ_BW, JQ_MINW, (JQ_VARIANT == 1)>(PropArgs);
#endif
#endif
