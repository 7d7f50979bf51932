// One translation unit per (NT, BW) instantiation of the propagator kernels (parallel builds, and the
// MFMA register form can be chosen per instantiation -- see Makefile).
#include "jq_kernels.h"
#ifndef JQ_NT
#error "compile with -DJQ_NT=<tiles> -DJQ_BW=<band>"
#endif
#define JQ_MINW ((JQ_NT <= 2) ? 2 : 1)
template __global__ void k_forward<JQ_NT, JQ_BW, JQ_MINW>(PropArgs);
template __global__ void k_backward<JQ_NT, JQ_BW, JQ_MINW>(PropArgs);
