/* The C ABI of libjuqbox_hip.so used from plain C, without the Python mirror or the Julia shim: the reference's smallest case
 * (test/cases/rabi-setup.jl: a 2-level qubit, X gate over one Rabi period, analytic control vector) built by hand, one
 * objective + gradient evaluation (traceobjgrad, src/evalobjgrad.jl:504) on the GPU.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o c_abi_demo -Ljuqbox.jl_amd -ljuqbox_hip -lm -Wl,-rpath,$PWD/juqbox.jl_amd
 *
 * Output (stdout): "objfv <value>" and one "grad <i> <value>" line per coefficient; exit code 0.  Without a gfx950 device the
 * library refuses to create a handle (there is no CPU fallback): the message goes to stderr, exit code 3.
 * tests/test_abi.py builds it; tests/test_gpu_parity.py compares its numbers with the Python mirror's. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "juqbox_hip.h"

int main(void)
{
    enum { NTOT = 2, N = 2, D1 = 3, NCOEFF = 2 * 1 * 1 * D1 };
    const double pi = 3.14159265358979323846;
    const double T = 2.0 * pi, theta = pi / 2.0, aOmega = pi / T;
    /* column-major 2 x 2 operators: lowering operator a = [0 1; 0 0]; Hsym = a + a', Hanti = a - a'; no drift in the rotating frame */
    const double Hconst[4] = {0.0, 0.0, 0.0, 0.0};
    const double Hsym[4] = {0.0, 1.0, 1.0, 0.0};
    const double Hanti[4] = {0.0, -1.0, 1.0, 0.0};
    const double Uinit[4] = {1.0, 0.0, 0.0, 1.0};
    /* target (test/cases/rabi-setup.jl:60-66): rotation by pi about the axis given by theta */
    double Vr[4], Vi[4];
    const double c = cos(aOmega * T), s = sin(aOmega * T);
    Vr[0] = c, Vi[0] = 0.0;                                   /* (0,0) */
    Vr[1] = -sin(theta) * s, Vi[1] = -cos(theta) * s;         /* (1,0) */
    Vr[2] = sin(theta) * s, Vi[2] = -cos(theta) * s;          /* (0,1) */
    Vr[3] = c, Vi[3] = 0.0;                                   /* (1,1) */
    const double wdiag[2] = {0.0, 0.0};                       /* no guard levels: no leakage weights */
    const double Cfreq[1] = {0.0};
    double pcof[NCOEFF];
    for (int i = 0; i < D1; ++i) pcof[i] = aOmega * cos(theta), pcof[D1 + i] = aOmega * sin(theta);

    jq_problem p;
    memset(&p, 0, sizeof p);
    p.Ntot = NTOT, p.N = N, p.Ncoupled = 1, p.Nfreq = 1, p.nsteps = 57, p.neumann_terms = 10, p.objFuncType = 1, p.Nunc = 0;
    p.T = T;
    p.Hconst = Hconst, p.Hsym_ops = Hsym, p.Hanti_ops = Hanti, p.Uinit = Uinit, p.Utarget_r = Vr, p.Utarget_i = Vi;
    p.wmat_real_diag = wdiag, p.Cfreq = Cfreq;

    if (jq_abi_version() != JQ_ABI_VERSION) {
        fprintf(stderr, "c_abi_demo: library ABI %d, header ABI %d\n", jq_abi_version(), JQ_ABI_VERSION);
        return 3;
    }
    if (jq_device_count() < 1) {
        fprintf(stderr, "c_abi_demo: no HIP device visible (%s has no CPU fallback)\n", jq_version());
        return 3;
    }
    jq_handle* h = NULL;
    int rc = jq_create(&p, &h);
    if (rc != JQ_OK) {
        fprintf(stderr, "c_abi_demo: jq_create failed (%d): %s\n", rc, h ? jq_last_error(h) : "no handle");
        if (h) jq_destroy(h);
        return 3;
    }
    double out4[4], total[NCOEFF], infid[NCOEFF], leak[NCOEFF];
    rc = jq_traceobjgrad(h, pcof, NCOEFF, 1, out4, total, infid, leak);
    if (rc != JQ_OK) {
        fprintf(stderr, "c_abi_demo: jq_traceobjgrad failed (%d): %s\n", rc, jq_last_error(h));
        jq_destroy(h);
        return 4;
    }
    printf("objfv %.17g\n", out4[0]);
    for (int i = 0; i < NCOEFF; ++i) printf("grad %d %.17g\n", i, total[i]);
    /* the reference's error behaviour through the ABI: a control vector with an odd number of elements (src/evalobjgrad.jl:604-606) */
    rc = jq_traceobjgrad(h, pcof, NCOEFF - 1, 1, out4, total, infid, leak);
    printf("wrong_length_rc %d\n", rc);
    jq_destroy(h);
    return 0;
}
