// One translation unit per (NT, BW, solver) instantiation of the propagator kernels (parallel builds, and
// the MFMA register form can be chosen per instantiation -- see Makefile).
#include "jq_kernels.h"
#if !defined(JQ_NT) || !defined(JQ_BW) || !defined(JQ_JAC)
#error "compile with -DJQ_NT=<tiles> -DJQ_BW=<band> -DJQ_JAC=<0|1>"
#endif
#define JQ_MINW ((JQ_NT <= 2) ? 2 : 1)
template __global__ void k_forward<JQ_NT, JQ_BW, JQ_MINW, (JQ_JAC != 0)>(PropArgs);
template __global__ void k_backward<JQ_NT, JQ_BW, JQ_MINW, (JQ_JAC != 0)>(PropArgs);
