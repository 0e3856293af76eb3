/* oracle/mtd_ref_steinhardt.c — TEST INFRASTRUCTURE ONLY (see mtd_ref.h).
 * Restatement of SteinhardtQl.cc (CPU path — the reference has no GPU class for this CV) and of the
 * spherical-harmonics evaluator it calls (spherical_harmonics.hpp:32-246, third-party fsph).
 * ref_sph_evaluate is PINNED: tests/golden/sph_lmax6.json holds the output of the reference's own header
 * (compiled in place into oracle/_ref) and tests/test_oracle_steinhardt.py compares bit-level-close (1e-14);
 * computeCV / computeBiasForces are parity-unpinned (analytic KATs: fcc Q_l values, numerical gradient).
 */
#include "mtd_ref.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---------------------------------------------------------------- fsph::PointSPHEvaluator */

static unsigned int index2d(unsigned int w, unsigned int i, unsigned int j) { return w * i + j; }      /* :11-15 */
static unsigned int sph_count(unsigned int lmax) { return (lmax + 1) * (lmax + 2) / 2; }                /* :17-21 */
static unsigned int sph_index(unsigned int l, unsigned int m) { return l > 0 ? sph_count(l - 1) + m : 0; }   /* :23-30 */

typedef struct
    {
    unsigned int lmax;
    double *sin_powers;            /* lmax+1 */
    double complex *theta_harm;    /* lmax+1 */
    double *prefactors;            /* 2 (lmax+1) lmax */
    double *jacobi;                /* (lmax+1)^2 */
    double *legendre;              /* sphCount(lmax) */
    } sph_eval;

static void sph_init(sph_eval *e, unsigned int lmax)                     /* ctor :101-109 + evaluatePrefactors :151-175 */
    {
    e->lmax = lmax;
    e->sin_powers = (double *)calloc(lmax + 1, sizeof(double));
    e->theta_harm = (double complex *)calloc(lmax + 1, sizeof(double complex));
    e->prefactors = (double *)calloc(2 * (lmax + 1) * lmax + 1, sizeof(double));
    e->jacobi = (double *)calloc((lmax + 1) * (lmax + 1), sizeof(double));
    e->legendre = (double *)calloc(sph_count(lmax), sizeof(double));
    const unsigned int f1Count = index2d(lmax, lmax + 1, 0);
    for (unsigned int m = 0; m < lmax + 1; ++m)
        for (unsigned int l = 1; l < lmax + 1; ++l)
            {
            const unsigned int idx = index2d(lmax, m, l - 1);
            e->prefactors[idx] = 2 * sqrt(1 + (m - 0.5) / l) * sqrt(1 - (m - 0.5) / (l + 2 * m));
            }
    for (unsigned int m = 0; m < lmax + 1; ++m)
        {
        e->prefactors[f1Count + index2d(lmax, m, 0)] = 0;
        for (unsigned int l = 2; l < lmax + 1; ++l)
            {
            const unsigned int idx = f1Count + index2d(lmax, m, l - 1);
            e->prefactors[idx] = -sqrt(1.0 + 4.0 / (2 * l + 2 * m - 3)) * sqrt(1 - 1.0 / l) * sqrt(1.0 - 1.0 / (l + 2 * m));
            }
        }
    }

static void sph_free(sph_eval *e)
    {
    free(e->sin_powers); free(e->theta_harm); free(e->prefactors); free(e->jacobi); free(e->legendre);
    }

static void sph_compute(sph_eval *e, double phi, double theta)           /* compute :127-138 */
    {
    const unsigned int lmax = e->lmax;
    const double sphi = sin(phi);
    e->sin_powers[0] = 1;                                                   /* compute_sinpows :177-182 */
    for (unsigned int i = 1; i < lmax + 1; ++i) e->sin_powers[i] = e->sin_powers[i - 1] * sphi;
    for (unsigned int i = 0; i < lmax + 1; ++i) e->theta_harm[i] = cexp(I * (i * theta));   /* :184-189 */
    const double cphi = cos(phi);
    const unsigned int f1Count = index2d(lmax, lmax + 1, 0);              /* compute_jacobis :191-213 */
    for (unsigned int m = 0; m < lmax + 1; ++m)
        {
        if (m > 0)
            e->jacobi[index2d(lmax + 1, m, 0)] = e->jacobi[index2d(lmax + 1, m - 1, 0)] * sqrt(1 + 1.0 / 2 / m);
        else
            e->jacobi[index2d(lmax + 1, 0, 0)] = 1 / sqrt(2);
        if (lmax > 0)
            e->jacobi[index2d(lmax + 1, m, 1)] = cphi * e->prefactors[index2d(lmax, m, 0)] * e->jacobi[index2d(lmax + 1, m, 0)];
        for (unsigned int l = 2; l < lmax + 1; ++l)
            e->jacobi[index2d(lmax + 1, m, l)] =
                (cphi * e->prefactors[index2d(lmax, m, l - 1)] * e->jacobi[index2d(lmax + 1, m, l - 1)]
                 + e->prefactors[f1Count + index2d(lmax, m, l - 1)] * e->jacobi[index2d(lmax + 1, m, l - 2)]);
        }
    for (unsigned int l = 0; l < lmax + 1; ++l)                            /* compute_legendres :215-225 */
        for (unsigned int m = 0; m < l + 1; ++m)
            e->legendre[sph_index(l, m)] = e->sin_powers[m] * e->jacobi[index2d(lmax + 1, m, l - m)];
    }

/* iterator order (:62-93): per l, m = 0..l then (full_m) -1..-l; negative m = conjugate harmonic, no Condon-Shortley */
static void sph_emit(const sph_eval *e, int full_m, double complex *out)
    {
    unsigned int n = 0;
    for (unsigned int l = 0; l <= e->lmax; ++l)
        {
        const unsigned int mcount = full_m ? 2 * l + 1 : l + 1;
        for (unsigned int mm = 0; mm < mcount; ++mm)
            {
            if (mm > l)
                {
                const unsigned int m = mm - l;
                out[n++] = (e->legendre[sph_index(l, m)] / sqrt(2 * M_PI)) * conj(e->theta_harm[m]);
                }
            else
                out[n++] = (e->legendre[sph_index(l, mm)] / sqrt(2 * M_PI)) * e->theta_harm[mm];
            }
        }
    }

/* fsph::evaluate_SPH (:229-246): out (re,im) pairs */
void ref_sph_evaluate(double *out, unsigned int lmax, const double *phi, const double *theta, unsigned int N, int full_m)
    {
    sph_eval e;
    sph_init(&e, lmax);
    const unsigned int per = full_m ? (lmax + 1) * (lmax + 1) : sph_count(lmax);
    double complex *tmp = (double complex *)malloc(sizeof(double complex) * per);
    for (unsigned int i = 0; i < N; ++i)
        {
        sph_compute(&e, phi[i], theta[i]);
        sph_emit(&e, full_m, tmp);
        for (unsigned int k = 0; k < per; ++k)
            {
            out[2 * ((size_t)i * per + k)] = creal(tmp[k]);
            out[2 * ((size_t)i * per + k) + 1] = cimag(tmp[k]);
            }
        }
    free(tmp);
    sph_free(&e);
    }

/* ---------------------------------------------------------------- SteinhardtQl.cc */

static void min_image(const ref_box *b, double w[3])
    {
    double img = rint(w[2] / b->L[2]);
    w[2] -= b->L[2] * img;
    w[1] -= b->L[2] * b->yz * img;
    w[0] -= b->L[2] * b->xz * img;
    img = rint(w[1] / b->L[1]);
    w[1] -= b->L[1] * img;
    w[0] -= b->L[1] * b->xy * img;
    w[0] -= b->L[0] * rint(w[0] / b->L[0]);
    }

static double f_smooth(double r_onsq, double r_cutsq, double rsq)         /* :36-48 */
    {
    if (rsq <= r_onsq) return 1.0;
    if (rsq > r_cutsq) return 0.0;
    double r = sqrt(rsq), r_on = sqrt(r_onsq), r_cut = sqrt(r_cutsq);
    return 0.5 * (cos(M_PI * (r - r_on) / (r_cut - r_on)) + 1);
    }

static double fprime_smooth_divr(double r_onsq, double r_cutsq, double rsq)   /* :50-60 */
    {
    if (rsq <= r_onsq || rsq > r_cutsq) return 0.0;
    double r = sqrt(rsq), r_on = sqrt(r_onsq), r_cut = sqrt(r_cutsq);
    return -(0.5 * M_PI) / r / (r_cut - r_on) * sin(M_PI * (r - r_on) / (r_cut - r_on));
    }

/* computeCV, :62-201.  Qlm_out: (re,im) x (lmax+1)^2 in the iterator order; Ql_out: lmax+1.  Returns the CV. */
double ref_ql_compute_cv(unsigned int N, const double *postype, const ref_box *box, const unsigned int *head_list,
                         const unsigned int *n_neigh, const unsigned int *nlist, int half_nlist, double rcut, double ron,
                         unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int N_global, double *Qlm_out,
                         double *Ql_out)
    {
    const double rcutsq = rcut * rcut, ronsq = ron * ron;                    /* ctor :18 */
    const unsigned int count = (lmax + 1) * (lmax + 1);
    double complex *Ylm_pp = (double complex *)calloc(count, sizeof(double complex));
    double complex *Qlm = (double complex *)calloc(count, sizeof(double complex));
    sph_eval e;
    sph_init(&e, lmax);

    for (unsigned int i = 0; i < N; i++)
        {
        const double *pi = postype + 4 * (size_t)i;
        if ((unsigned int)pi[3] != type) continue;                            /* :105 */
        const unsigned int myHead = head_list[i], size = n_neigh[i];
        for (unsigned int k = 0; k < size; k++)
            {
            const unsigned int j = nlist[myHead + k];
            const double *pj = postype + 4 * (size_t)j;
            double dx[3] = { pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2] };
            if ((unsigned int)pj[3] != type) continue;                        /* :126 */
            min_image(box, dx);
            double rsq = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
            if (rsq <= rcutsq)
                {
                double f = f_smooth(ronsq, rcutsq, rsq);
                double theta = acos(dx[2] / sqrt(rsq));                       /* :138 */
                double phi = atan2(dx[1], dx[0]);
                sph_compute(&e, theta, phi);                                  /* evaluate_SPH(..., &theta, &phi, ...) :143 */
                sph_emit(&e, 1, Ylm_pp);
                int n = 0;
                for (int l = 0; l <= (int)lmax; ++l)
                    for (int p = 0; p < 2 * l + 1; ++p)
                        {
                        int m = (p <= l) ? p : (l - p);
                        int phase = (m > 0 && m % 2) ? -1 : 1;                /* Condon-Shortley :150 */
                        Qlm[n] += (double)phase * Ylm_pp[n] * f;
                        n++;
                        }
                }
            }
        }

    unsigned int n = 0;
    for (int l = 0; l <= (int)lmax; ++l)
        {
        Ql_out[l] = 0.0;
        for (int p = 0; p < 2 * l + 1; ++p)
            {
            if (half_nlist)                                                    /* :173-179 */
                {
                if (l % 2 == 0)
                    Qlm[n] *= 2;
                else
                    Qlm[n] = 0.0;
                }
            double Qlm_sq = creal(conj(Qlm[n]) * Qlm[n]);
            Qlm_sq *= (4.0 * M_PI / (2 * l + 1)) / ((double)N_global * (double)N_global);   /* nc = 1 */
            Ql_out[l] += Qlm_sq;
            n++;
            }
        }
    double value = 0.0;
    for (unsigned int l = 0; l <= lmax; ++l) value += Ql_ref[l] * Ql_out[l];  /* :190-194 */
    for (unsigned int q = 0; q < count; ++q)
        {
        Qlm_out[2 * q] = creal(Qlm[q]);
        Qlm_out[2 * q + 1] = cimag(Qlm[q]);
        }
    free(Ylm_pp); free(Qlm);
    sph_free(&e);
    return value;
    }

/* Q_l and the CV value from a (summed) Q_lm table — the tail of computeCV (:181-194) on its own, for particle-sharded checks:
 * Q_lm is linear in the pair sum, so the table of the whole system is the sum of the shards' tables. */
double ref_ql_from_qlm(unsigned int lmax, const double *Qlm_in, const double *Ql_ref, unsigned int N_global, double *Ql_out)
    {
    unsigned int n = 0;
    double value = 0.0;
    for (int l = 0; l <= (int)lmax; ++l)
        {
        Ql_out[l] = 0.0;
        for (int p = 0; p < 2 * l + 1; ++p, ++n)
            {
            double sq = Qlm_in[2 * n] * Qlm_in[2 * n] + Qlm_in[2 * n + 1] * Qlm_in[2 * n + 1];
            sq *= (4.0 * M_PI / (2 * l + 1)) / ((double)N_global * (double)N_global);
            Ql_out[l] += sq;
            }
        value += Ql_ref[l] * Ql_out[l];
        }
    return value;
    }

/* computeBiasForces, :203-339; Qlm_in as left by computeCV (Q20) */
void ref_ql_compute_forces(unsigned int N, const double *postype, const ref_box *box, const unsigned int *head_list,
                           const unsigned int *n_neigh, const unsigned int *nlist, int half_nlist, double rcut, double ron,
                           unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int N_global,
                           const double *Qlm_in, double bias, double *force_out)
    {
    const double rcutsq = rcut * rcut, ronsq = ron * ron;
    const unsigned int count = (lmax + 1) * (lmax + 1);
    double complex *Ylm_pp = (double complex *)calloc(count, sizeof(double complex));
    double complex *Qlm = (double complex *)calloc(count, sizeof(double complex));
    for (unsigned int q = 0; q < count; ++q) Qlm[q] = Qlm_in[2 * q] + I * Qlm_in[2 * q + 1];
    sph_eval e;
    sph_init(&e, lmax);
    memset(force_out, 0, sizeof(double) * 4 * (size_t)N);                     /* :236 */

    for (unsigned int i = 0; i < N; i++)
        {
        const double *pi = postype + 4 * (size_t)i;
        if ((unsigned int)pi[3] != type) continue;
        const unsigned int myHead = head_list[i], size = n_neigh[i];
        for (unsigned int k = 0; k < size; k++)
            {
            const unsigned int j = nlist[myHead + k];
            const double *pj = postype + 4 * (size_t)j;
            double dx[3] = { pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2] };
            if ((unsigned int)pj[3] != type) continue;
            min_image(box, dx);
            double rsq = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
            double force[3] = { 0.0, 0.0, 0.0 };
            if (rsq <= rcutsq)
                {
                double complex r = sqrt(rsq);
                double theta = acos(dx[2] / creal(r));
                double phi = atan2(dx[1], dx[0]);
                double complex e_theta[3] = { cos(theta) * cos(phi), cos(theta) * sin(phi), -sin(theta) };   /* :288 */
                double complex e_phi[3] = { -sin(phi), cos(phi), 0.0 };
                sph_compute(&e, theta, phi);
                sph_emit(&e, 1, Ylm_pp);
                double complex fprime_divr = fprime_smooth_divr(ronsq, rcutsq, rsq);
                double complex f = f_smooth(ronsq, rcutsq, rsq);
                int n = 0;
                for (int l = 0; l <= (int)lmax; ++l)
                    {
                    double del_Ql_i[3] = { 0.0, 0.0, 0.0 };
                    for (int p = 0; p < 2 * l + 1; ++p)
                        {
                        int m = (p <= l) ? p : (l - p);
                        int phase = (m > 0 && m % 2) ? -1 : 1;
                        double complex Ylm = (double)phase * Ylm_pp[n];
                        double complex dYlm_dtheta = (double complex)(m / tan(theta)) * Ylm;            /* :305 */
                        if (m < l)
                            {
                            unsigned int m_plus_one = (m < 0) ? (m == -1 ? n - p : n - 1) : (n + 1);     /* :308 */
                            int phase_plus_one = (m + 1 > 0 && (m + 1) % 2) ? -1 : 1;
                            dYlm_dtheta += (double complex)(phase_plus_one * sqrt((double)((l - m) * (l + m + 1))))
                                           * cexp(-I * phi) * Ylm_pp[m_plus_one];
                            }
                        double complex dYlm_dphi = (I * (double)m) * Ylm;                                 /* :312 */
                        for (int d = 0; d < 3; ++d)
                            {
                            double complex del_Qlm = dx[d] * fprime_divr * Ylm + f / r * e_theta[d] * dYlm_dtheta
                                                     + f * e_phi[d] / (r * (double complex)sin(theta)) * dYlm_dphi;   /* :314 */
                            del_Qlm *= conj(Qlm[n]);
                            del_Ql_i[d] += 2.0 * creal(del_Qlm);
                            }
                        n++;
                        }
                    for (int d = 0; d < 3; ++d)
                        {
                        del_Ql_i[d] *= (4.0 * M_PI / (2 * l + 1)) / ((double)N_global * (double)N_global);   /* :319 */
                        force[d] -= bias * del_Ql_i[d] * Ql_ref[l];                                          /* :321 */
                        }
                    }
                }
            force_out[4 * (size_t)i + 0] += force[0];
            force_out[4 * (size_t)i + 1] += force[1];
            force_out[4 * (size_t)i + 2] += force[2];
            if (half_nlist && j < N)                                            /* :328-333 */
                {
                force_out[4 * (size_t)j + 0] -= force[0];
                force_out[4 * (size_t)j + 1] -= force[1];
                force_out[4 * (size_t)j + 2] -= force[2];
                }
            }
        }
    free(Ylm_pp); free(Qlm);
    sph_free(&e);
    }
