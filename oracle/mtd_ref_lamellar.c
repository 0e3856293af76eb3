/* oracle/mtd_ref_lamellar.c — TEST INFRASTRUCTURE ONLY (see mtd_ref.h).
 * Restatement of LamellarOrderParameter.cc (CPU path, Scalar = double).  Parity unpinned: the
 * reference holds no vectors for this path; pinned by analytic KATs in tests/test_oracle_kat.py
 * (single particle, perfect lamellae, numerical gradients incl. the reference's factor 2).
 */
#include "mtd_ref.h"
#include <math.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* reciprocal lattice vectors, LamellarOrderParameter.cc:94-102 == :150-158
 * (a_i from BoxDim::getLatticeVector, V from BoxDim::getVolume) */
static void reciprocal(const ref_box *box, double b1[3], double b2[3], double b3[3])
    {
    double a1[3] = { box->L[0], 0.0, 0.0 };
    double a2[3] = { box->xy * box->L[1], box->L[1], 0.0 };
    double a3[3] = { box->xz * box->L[2], box->yz * box->L[2], box->L[2] };
    double V_box = box->L[0] * box->L[1] * box->L[2];
    double two_pi = 2.0 * M_PI;

    b1[0] = two_pi * (a2[1] * a3[2] - a2[2] * a3[1]) / V_box;
    b1[1] = two_pi * (a2[2] * a3[0] - a2[0] * a3[2]) / V_box;
    b1[2] = two_pi * (a2[0] * a3[1] - a2[1] * a3[0]) / V_box;

    b2[0] = two_pi * (a3[1] * a1[2] - a3[2] * a1[1]) / V_box;
    b2[1] = two_pi * (a3[2] * a1[0] - a3[0] * a1[2]) / V_box;
    b2[2] = two_pi * (a3[0] * a1[1] - a3[1] * a1[0]) / V_box;

    b3[0] = two_pi * (a1[1] * a2[2] - a1[2] * a2[1]) / V_box;
    b3[1] = two_pi * (a1[2] * a2[0] - a1[0] * a2[2]) / V_box;
    b3[2] = two_pi * (a1[0] * a2[1] - a1[1] * a2[0]) / V_box;
    }

/* calculateFourierModes, .cc:143-179: mode-outer loop, every particle re-streamed per mode */
void ref_lamellar_fourier_modes(unsigned int n_wave, const int *lattice, unsigned int N,
                                const double *postype, const double *mode, const ref_box *global_box,
                                double *modes_out)
    {
    double b1[3], b2[3], b3[3];
    reciprocal(global_box, b1, b2, b3);

    for (unsigned int k = 0; k < n_wave; k++)
        {
        modes_out[2 * k + 0] = 0.0;
        modes_out[2 * k + 1] = 0.0;
        double q[3];
        for (int c = 0; c < 3; c++)                                   /* .cc:165 */
            q[c] = b1[c] * (double)lattice[3 * k + 0] + b2[c] * (double)lattice[3 * k + 1]
                   + b3[c] * (double)lattice[3 * k + 2];

        double re = 0.0, im = 0.0;
#ifdef REF_OMP   /* libmtd_ref_omp.so only: the "idealised multi-rank" CPU baseline of bench.py; the checker is serial */
#pragma omp parallel for reduction(+ : re, im) schedule(static)
#endif
        for (unsigned int idx = 0; idx < N; idx++)
            {
            const double *p = postype + 4 * idx;
            unsigned int type = (unsigned int)p[3];                   /* __scalar_as_int(postype.w) */
            double a = mode[type];
            double dotproduct = q[0] * p[0] + q[1] * p[1] + q[2] * p[2];
            re += a * cos(dotproduct);                                /* .cc:175 */
            im += a * sin(dotproduct);                                /* .cc:176 */
            }
        modes_out[2 * k + 0] = re;
        modes_out[2 * k + 1] = im;
        }
    }

/* computeCV, .cc:58-68: sum of real parts / N_global */
double ref_lamellar_cv(unsigned int n_wave, const double *modes, unsigned int N_global)
    {
    double sum = 0.0;
    for (unsigned int k = 0; k < n_wave; k++)
        sum += modes[2 * k];
    sum /= (double)N_global;
    return sum;
    }

/* computeBiasForces, .cc:77-140 (note the factor 2 at :120 — SURVEY Q1 — reproduced) */
void ref_lamellar_forces(unsigned int n_wave, const int *lattice, unsigned int N,
                         const double *postype, const double *mode, const ref_box *global_box,
                         unsigned int N_global, double bias, double *force_out)
    {
    double b1[3], b2[3], b3[3];
    reciprocal(global_box, b1, b2, b3);
    double denom = (double)N_global;

#ifdef REF_OMP
#pragma omp parallel for schedule(static)
#endif
    for (unsigned int idx = 0; idx < N; idx++)
        {
        const double *p = postype + 4 * idx;
        unsigned int type = (unsigned int)p[3];
        double a = mode[type];
        double fx = 0.0, fy = 0.0, fz = 0.0;

        for (unsigned int k = 0; k < n_wave; k++)
            {
            double q[3];
            for (int c = 0; c < 3; c++)
                q[c] = b1[c] * (double)lattice[3 * k + 0] + b2[c] * (double)lattice[3 * k + 1]
                       + b3[c] * (double)lattice[3 * k + 2];
            double dotproduct = p[0] * q[0] + p[1] * q[1] + p[2] * q[2];
            double f = 2.0 * a * sin(dotproduct);                     /* .cc:120 */
            fx += q[0] * f;
            fy += q[1] * f;
            fz += q[2] * f;
            }

        fx *= bias; fy *= bias; fz *= bias;                           /* .cc:127-129 */
        fx /= denom; fy /= denom; fz /= denom;                        /* .cc:131-133 */

        force_out[4 * idx + 0] = fx;
        force_out[4 * idx + 1] = fy;
        force_out[4 * idx + 2] = fz;
        force_out[4 * idx + 3] = 0.0;
        }
    }
