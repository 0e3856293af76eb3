// steinhardt.hip — Steinhardt Q_l order parameter on gfx950.
//
// Reference: SteinhardtQl.cc:62-201 (computeCV), :203-339 (computeBiasForces), :36-60 (smoothing), with the
// spherical harmonics of spherical_harmonics.hpp:32-246 (fsph).  The reference has NO GPU implementation of this
// CV (SURVEY §2.4 "new"): it runs a host loop that constructs a PointSPHEvaluator (five heap arrays) per pair.
//
// MI355X design (vector-ALU bound, ~7 kflop per pair in the force pass, not HBM bound):
//   k_ql_accumulate   one thread per central particle; per pair the Y_lm(m >= 0) come from the same Jacobi
//                     recurrence in registers (cos/sin of the angles from dx/r — no acos/atan2, no heap);
//                     Q'_lm = sum f Y_lm kept in registers (28 complex at lmax = 6), reduced wave -> block in a
//                     fixed order; negative m are conjugates, the Condon-Shortley phase is applied at the end
//   k_reduce_partials (lamellar.hip) -> Q'_lm ; [multi-GPU: all-reduce of (lmax+1)(lmax+2) doubles here]
//   k_ql_finalize     full Q_lm table in the reference's order, third-law scaling, Q_l, CV value
//   k_ql_forces       one thread per central particle, Q_lm broadcast from LDS, spherical-basis gradient of every
//                     (l, m) term exactly as :287-321
// Double precision throughout.
#include "mtd_device.hpp"

#include <cmath>
#include <cstring>

namespace
{

using namespace mtd;

constexpr int QL_THREADS = 128;
constexpr unsigned int QL_MAX_BLOCKS = 1024;

template<int LMAX> struct QlArgs
    {
    double lo[3], L[3], xy, xz, yz;
    double rcutsq, ronsq, r_on, r_cut;
    unsigned int lmax, type, N, n_global;
    int half_nlist, _pad;
    // Jacobi recurrence prefactors (spherical_harmonics.hpp:151-175), [m][l] for l = 1..LMAX
    double f0[LMAX + 1][LMAX + 1];
    double f1[LMAX + 1][LMAX + 1];
    double jac0[LMAX + 1];              // jacobi[m][0] = 1/sqrt(2) prod sqrt(1 + 1/2m) (:197-201)
    double ql_ref[LMAX + 1];
    };

struct cplx
    {
    double re, im;
    };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cconj(cplx a) { return {a.re, -a.im}; }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return {a.re * s, a.im * s}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }

template<int LMAX>
__device__ __forceinline__ void min_image(const QlArgs<LMAX> &a, double &x, double &y, double &z)
    {
    double img = rint(z / a.L[2]);
    z -= a.L[2] * img;
    y -= a.L[2] * a.yz * img;
    x -= a.L[2] * a.xz * img;
    img = rint(y / a.L[1]);
    y -= a.L[1] * img;
    x -= a.L[1] * a.xy * img;
    x -= a.L[0] * rint(x / a.L[0]);
    }

template<int LMAX> __device__ __forceinline__ double f_smooth(const QlArgs<LMAX> &a, double rsq)          // :36-48
    {
    if (rsq <= a.ronsq) return 1.0;
    if (rsq > a.rcutsq) return 0.0;
    const double r = sqrt(rsq);
    return 0.5 * (cospi((r - a.r_on) / (a.r_cut - a.r_on)) + 1.0);
    }

template<int LMAX> __device__ __forceinline__ double fprime_smooth_divr(const QlArgs<LMAX> &a, double rsq)   // :50-60
    {
    if (rsq <= a.ronsq || rsq > a.rcutsq) return 0.0;
    const double r = sqrt(rsq);
    return -(0.5 * M_PI) / r / (a.r_cut - a.r_on) * sinpi((r - a.r_on) / (a.r_cut - a.r_on));
    }

// Y'_lm (no Condon-Shortley phase) for 0 <= m <= l <= lmax at direction (dx,dy,dz)/r:
// Y[m][l] = sin^m(theta) * jacobi[m][l-m] / sqrt(2 pi) * e^{i m phi}   (spherical_harmonics.hpp:78-93, 177-226)
// SCALE: every Y is multiplied by `scale` and ADDED into Y[m][l] (accumulate = true) or stored (false)
template<int LMAX, bool ACCUMULATE>
__device__ __forceinline__ void ylm_table(const QlArgs<LMAX> &a, const double ct, const double st, const double cp, const double sp,
                                          const double scale, cplx (&Y)[LMAX + 1][LMAX + 1])
    {
    const double inv_sqrt_2pi = 0.3989422804014326779399460599343818684758586311649;
    double sinpow = 1.0;
    cplx harm = {1.0, 0.0};                      // e^{i m phi}
#pragma unroll
    for (int m = 0; m <= LMAX; ++m)
        {
        if (m <= (int)a.lmax)
            {
            // jacobi recurrence in the degree n = l - m (:203-211)
            double jm2 = 0.0, jm1 = a.jac0[m];
#pragma unroll
            for (int n = 0; n + m <= LMAX; ++n)
                {
                if (n + m <= (int)a.lmax)
                    {
                    double j;
                    if (n == 0)
                        j = a.jac0[m];
                    else if (n == 1)
                        j = ct * a.f0[m][1] * jm1;
                    else
                        j = ct * a.f0[m][n] * jm1 + a.f1[m][n] * jm2;
                    const double leg = sinpow * j * inv_sqrt_2pi * scale;
                    if (ACCUMULATE)
                        {
                        Y[m][n + m].re += leg * harm.re;
                        Y[m][n + m].im += leg * harm.im;
                        }
                    else
                        Y[m][n + m] = {leg * harm.re, leg * harm.im};
                    jm2 = jm1;
                    jm1 = j;
                    }
                }
            }
        sinpow *= st;
        harm = cmul(harm, {cp, sp});
        }
    }

// ---- CV accumulation -------------------------------------------------------------------------------
template<typename S4, int LMAX>
__global__ __launch_bounds__(QL_THREADS) void k_ql_accumulate(const QlArgs<LMAX> a, const S4 *__restrict__ postype,
                                                              const unsigned int *__restrict__ head_list,
                                                              const unsigned int *__restrict__ n_neigh,
                                                              const unsigned int *__restrict__ nlist, double *__restrict__ partials)
    {
    constexpr int NLM = (LMAX + 1) * (LMAX + 2) / 2;
    __shared__ double s_wave[QL_THREADS / MTD_WAVE][2 * NLM];
    cplx Q[LMAX + 1][LMAX + 1];
#pragma unroll
    for (int m = 0; m <= LMAX; ++m)
#pragma unroll
        for (int l = 0; l <= LMAX; ++l) Q[m][l] = {0.0, 0.0};

    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.N; i += gridDim.x * blockDim.x)
        {
        const Particle pi = scalar4_traits<S4>::load(postype, i);
        if ((unsigned int)pi.type != a.type) continue;                         // :105
        const unsigned int head = head_list[i], size = n_neigh[i];
        for (unsigned int k = 0; k < size; ++k)
            {
            const unsigned int j = nlist[head + k];
            const Particle pj = scalar4_traits<S4>::load(postype, j);
            if ((unsigned int)pj.type != a.type) continue;                     // :126
            double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            min_image(a, dx, dy, dz);
            const double rsq = dx * dx + dy * dy + dz * dz;
            if (rsq <= a.rcutsq)
                {
                const double f = f_smooth(a, rsq);
                const double r = sqrt(rsq);
                const double rho = sqrt(dx * dx + dy * dy);
                const double ct = dz / r, st = rho / r;                         // theta = acos(dz/r) (:138)
                const double cp = rho > 0.0 ? dx / rho : 1.0, sp = rho > 0.0 ? dy / rho : 0.0;   // phi = atan2(dy,dx)
                ylm_table<LMAX, true>(a, ct, st, cp, sp, f, Q);   // Q'_lm += f * Y_lm, straight from the recurrence
                }
            }
        }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int l = 0; l <= LMAX; ++l)
#pragma unroll
        for (int m = 0; m <= l; ++m)
            {
            const int idx = l * (l + 1) / 2 + m;
            const double re = wave_sum(Q[m][l].re), im = wave_sum(Q[m][l].im);
            if (lane == 0)
                {
                s_wave[wave][2 * idx] = re;
                s_wave[wave][2 * idx + 1] = im;
                }
            }
    __syncthreads();
    const unsigned int n_out = (a.lmax + 1) * (a.lmax + 2);     // 2 * n_lm of the RUNTIME lmax (same (l,m) order)
    for (unsigned int q = threadIdx.x; q < n_out; q += blockDim.x)
        {
        double v = 0.0;
        for (int w = 0; w < QL_THREADS / MTD_WAVE; ++w) v += s_wave[w][q];
        partials[(size_t)blockIdx.x * n_out + q] = v;
        }
    }

// ---- finalize: full Q_lm table (reference order), Q_l, CV --------------------------------------------
template<int LMAX>
__global__ void k_ql_finalize(const QlArgs<LMAX> a, const double *__restrict__ qprime, double *__restrict__ qlm_full,
                              double *__restrict__ ql, double *__restrict__ value)
    {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double ng = (double)a.n_global;
    unsigned int n = 0;
    double val = 0.0;
    for (int l = 0; l <= (int)a.lmax; ++l)
        {
        double Ql = 0.0;
        for (int p = 0; p < 2 * l + 1; ++p)
            {
            const int m = (p <= l) ? p : (l - p);
            const int am = m < 0 ? -m : m;
            const int idx = l * (l + 1) / 2 + am;
            cplx q = {qprime[2 * idx], qprime[2 * idx + 1]};
            if (m < 0) q = cconj(q);                                            // fsph negative m: conjugate, phase +1
            if (m > 0 && (m % 2)) q = cscale(q, -1.0);                          // Condon-Shortley (:150)
            if (a.half_nlist)                                                   // :173-179
                {
                if (l % 2 == 0)
                    q = cscale(q, 2.0);
                else
                    q = {0.0, 0.0};
                }
            qlm_full[2 * n] = q.re;
            qlm_full[2 * n + 1] = q.im;
            double sq = q.re * q.re + q.im * q.im;
            sq *= (4.0 * M_PI / (2 * l + 1)) / (ng * ng);                       // nc = 1 (:182)
            Ql += sq;
            ++n;
            }
        ql[l] = Ql;
        val += a.ql_ref[l] * Ql;                                                // :190-194
        }
    *value = val;
    }

// ---- forces ----------------------------------------------------------------------------------------------
template<typename S4, int LMAX, bool HALF>
__global__ __launch_bounds__(QL_THREADS) void k_ql_forces(const QlArgs<LMAX> a, const S4 *__restrict__ postype,
                                                          const unsigned int *__restrict__ head_list,
                                                          const unsigned int *__restrict__ n_neigh,
                                                          const unsigned int *__restrict__ nlist, const double *__restrict__ qlm_full,
                                                          S4 *__restrict__ force, const double *__restrict__ d_bias, const double bias_host)
    {
    typedef typename scalar4_traits<S4>::scalar scalar;
    constexpr int NFULL = (LMAX + 1) * (LMAX + 1);
    __shared__ double s_q[2 * NFULL];
    for (unsigned int q = threadIdx.x; q < 2 * (a.lmax + 1) * (a.lmax + 1); q += blockDim.x) s_q[q] = qlm_full[q];
    __syncthreads();
    const double bias = d_bias ? *d_bias : bias_host;
    const double ng = (double)a.n_global;

    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.N; i += gridDim.x * blockDim.x)
        {
        const Particle pi = scalar4_traits<S4>::load(postype, i);
        double Fx = 0.0, Fy = 0.0, Fz = 0.0;
        if ((unsigned int)pi.type == a.type)
            {
            const unsigned int head = head_list[i], size = n_neigh[i];
            for (unsigned int k = 0; k < size; ++k)
                {
                const unsigned int j = nlist[head + k];
                const Particle pj = scalar4_traits<S4>::load(postype, j);
                if ((unsigned int)pj.type != a.type) continue;
                double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                min_image(a, dx, dy, dz);
                const double rsq = dx * dx + dy * dy + dz * dz;
                if (!(rsq <= a.rcutsq)) continue;
                const double r = sqrt(rsq);
                const double rho = sqrt(dx * dx + dy * dy);
                const double ct = dz / r, st = rho / r;
                const double cp = rho > 0.0 ? dx / rho : 1.0, sp = rho > 0.0 ? dy / rho : 0.0;
                const double e_theta[3] = {ct * cp, ct * sp, -st};               // :288
                const double e_phi[3] = {-sp, cp, 0.0};
                const double d[3] = {dx, dy, dz};
                const double cot = ct / st;                                      // m / tan(theta) (:305); theta = 0 -> inf like the reference
                const cplx emiphi = {cp, -sp};                                   // exp(-i phi)
                const double fprime_divr = fprime_smooth_divr(a, rsq);
                const double f = f_smooth(a, rsq);
                cplx Y[LMAX + 1][LMAX + 1];
                ylm_table<LMAX, false>(a, ct, st, cp, sp, 1.0, Y);
                double fpx = 0.0, fpy = 0.0, fpz = 0.0;
                int n = 0;
#pragma unroll
                for (int l = 0; l <= LMAX; ++l)
                    {
                    if (l <= (int)a.lmax)
                        {
                        double del[3] = {0.0, 0.0, 0.0};
#pragma unroll
                        for (int p = 0; p < 2 * l + 1; ++p)
                            {
                            const int m = (p <= l) ? p : (l - p);
                            const int am = m < 0 ? -m : m;
                            // Ylm_pp[n]: raw fsph value; Ylm = phase * Ylm_pp[n] (:303-304)
                            cplx raw = Y[am][l];
                            if (m < 0) raw = cconj(raw);
                            const double phase = (m > 0 && (m % 2)) ? -1.0 : 1.0;
                            const cplx Ylm = cscale(raw, phase);
                            cplx dth = cscale(Ylm, (double)m * cot);               // (m / tan theta) * Ylm
                            if (m < l)
                                {
                                // Ylm_pp[m_plus_one] (:308): the raw entry of m+1 (for m = -1: m+1 = 0)
                                const int mp = m + 1;
                                const int amp = mp < 0 ? -mp : mp;
                                cplx rawp = Y[amp][l];
                                if (mp < 0) rawp = cconj(rawp);
                                const double phase_p = (mp > 0 && (mp % 2)) ? -1.0 : 1.0;
                                const double c = phase_p * sqrt((double)((l - m) * (l + m + 1)));
                                dth = cadd(dth, cscale(cmul(emiphi, rawp), c));
                                }
                            const cplx dph = {-(double)m * Ylm.im, (double)m * Ylm.re};   // i m Ylm (:312)
                            const cplx qc = {s_q[2 * n], -s_q[2 * n + 1]};                 // conj(Qlm[n])
#pragma unroll
                            for (int c3 = 0; c3 < 3; ++c3)
                                {
                                cplx t = cscale(Ylm, d[c3] * fprime_divr);
                                t = cadd(t, cscale(dth, f / r * e_theta[c3]));
                                t = cadd(t, cscale(dph, f * e_phi[c3] / (r * st)));
                                const cplx tq = cmul(t, qc);
                                del[c3] += 2.0 * tq.re;                                     // :316
                                }
                            ++n;
                            }
                        const double norm = (4.0 * M_PI / (2 * l + 1)) / (ng * ng);        // :319
                        fpx -= bias * del[0] * norm * a.ql_ref[l];                          // :321
                        fpy -= bias * del[1] * norm * a.ql_ref[l];
                        fpz -= bias * del[2] * norm * a.ql_ref[l];
                        }
                    }
                Fx += fpx; Fy += fpy; Fz += fpz;
                if (HALF && j < a.N)                                                        // :328-333
                    {
                    scalar *fj = (scalar *)&force[j];
                    atomicAdd(fj + 0, (scalar)(-fpx));
                    atomicAdd(fj + 1, (scalar)(-fpy));
                    atomicAdd(fj + 2, (scalar)(-fpz));
                    }
                }
            }
        if (HALF)
            {
            scalar *fi = (scalar *)&force[i];
            atomicAdd(fi + 0, (scalar)Fx);
            atomicAdd(fi + 1, (scalar)Fy);
            atomicAdd(fi + 2, (scalar)Fz);
            }
        else
            force[i] = scalar4_traits<S4>::make((scalar)Fx, (scalar)Fy, (scalar)Fz, (scalar)0);
        }
    }

template<int LMAX>
int fill_args(QlArgs<LMAX> &a, unsigned int N, const mtd_box *box, double rcut, double ron, unsigned int lmax, unsigned int type,
              const double *ql_ref, unsigned int n_global, int half)
    {
    if (!box || !ql_ref || n_global == 0 || lmax > (unsigned int)LMAX || !(rcut > 0.0) || !(ron >= 0.0) || !(ron < rcut))
        return MTD_ERR_INVALID_ARGUMENT;
    std::memset(&a, 0, sizeof(a));
    for (int i = 0; i < 3; ++i)
        {
        a.lo[i] = box->lo[i];
        a.L[i] = box->L[i];
        }
    a.xy = box->xy; a.xz = box->xz; a.yz = box->yz;
    a.rcutsq = rcut * rcut;                      // SteinhardtQl.cc:18
    a.ronsq = ron * ron;
    a.r_on = std::sqrt(a.ronsq);
    a.r_cut = std::sqrt(a.rcutsq);
    a.lmax = lmax; a.type = type; a.N = N; a.n_global = n_global; a.half_nlist = half;
    // evaluatePrefactors (spherical_harmonics.hpp:151-175) for the RUNTIME lmax; jacobi[m][0] (:197-201)
    for (unsigned int m = 0; m <= lmax; ++m)
        {
        for (unsigned int l = 1; l <= lmax; ++l) a.f0[m][l] = 2 * std::sqrt(1 + (m - 0.5) / l) * std::sqrt(1 - (m - 0.5) / (l + 2 * m));
        a.f1[m][1] = 0;
        for (unsigned int l = 2; l <= lmax; ++l)
            a.f1[m][l] = -std::sqrt(1.0 + 4.0 / (2 * l + 2 * m - 3)) * std::sqrt(1 - 1.0 / l) * std::sqrt(1.0 - 1.0 / (l + 2 * m));
        a.jac0[m] = m > 0 ? a.jac0[m - 1] * std::sqrt(1 + 1.0 / 2 / m) : 1 / std::sqrt(2.0);
        }
    for (unsigned int l = 0; l <= lmax; ++l) a.ql_ref[l] = ql_ref[l];
    return MTD_SUCCESS;
    }

unsigned int ql_blocks(unsigned int N)
    {
    unsigned int b = (N + QL_THREADS - 1) / QL_THREADS;
    if (b < 1) b = 1;
    if (b > QL_MAX_BLOCKS) b = QL_MAX_BLOCKS;
    return b;
    }

// the recurrence index in f0/f1 is the DEGREE n = l - m in the reference's tables (index2d(lmax, m, l-1) with l the
// degree counter of compute_jacobis), which is what ylm_table uses (a.f0[m][n]).

template<int LMAX>
int accumulate_impl(unsigned int N, const void *d_postype, int dtype, const mtd_box *box, const unsigned int *d_head,
                    const unsigned int *d_nneigh, const unsigned int *d_nlist, int half, double rcut, double ron, unsigned int lmax,
                    unsigned int type, const double *ql_ref, unsigned int n_global, double *d_partials, unsigned int *n_partials,
                    double *d_qprime, double *d_qlm, double *d_ql, double *d_value, bool accumulate, bool finalize, hipStream_t s)
    {
    QlArgs<LMAX> a;
    int rc = fill_args<LMAX>(a, N, box, rcut, ron, lmax, type, ql_ref, n_global, half);
    if (rc) return rc;
    if (!accumulate)
        {
        k_ql_finalize<LMAX><<<1, 64, 0, s>>>(a, d_qprime, d_qlm, d_ql, d_value);
        MTD_LAUNCH_CHECK();
        return MTD_SUCCESS;
        }
    const unsigned int blocks = ql_blocks(N);
    const unsigned int n_out = (lmax + 1) * (lmax + 2);
    if (dtype == MTD_F32)
        k_ql_accumulate<float4, LMAX><<<blocks, QL_THREADS, 0, s>>>(a, (const float4 *)d_postype, d_head, d_nneigh, d_nlist, d_partials);
    else
        k_ql_accumulate<double4, LMAX><<<blocks, QL_THREADS, 0, s>>>(a, (const double4 *)d_postype, d_head, d_nneigh, d_nlist, d_partials);
    MTD_LAUNCH_CHECK();
    *n_partials = blocks;
    rc = mtd_reduce_partials(d_partials, blocks, n_out, n_out, 1.0, 0.0, d_qprime, (mtd_stream_t)s);
    if (rc) return rc;
    if (!finalize) return MTD_SUCCESS;
    k_ql_finalize<LMAX><<<1, 64, 0, s>>>(a, d_qprime, d_qlm, d_ql, d_value);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

template<int LMAX>
int forces_impl(unsigned int N, const void *d_postype, void *d_force, int dtype, const mtd_box *box, const unsigned int *d_head,
                const unsigned int *d_nneigh, const unsigned int *d_nlist, int half, double rcut, double ron, unsigned int lmax,
                unsigned int type, const double *ql_ref, unsigned int n_global, const double *d_qlm, const double *d_bias,
                double bias_host, hipStream_t s)
    {
    QlArgs<LMAX> a;
    int rc = fill_args<LMAX>(a, N, box, rcut, ron, lmax, type, ql_ref, n_global, half);
    if (rc) return rc;
    const unsigned int blocks = ql_blocks(N);
    const size_t s4 = dtype == MTD_F32 ? sizeof(float4) : sizeof(double4);
    if (half) MTD_HIP_TRY(hipMemsetAsync(d_force, 0, s4 * N, s));            // memset of :236, the pair terms are then added atomically
    if (dtype == MTD_F32)
        {
        if (half)
            k_ql_forces<float4, LMAX, true><<<blocks, QL_THREADS, 0, s>>>(a, (const float4 *)d_postype, d_head, d_nneigh, d_nlist, d_qlm, (float4 *)d_force, d_bias, bias_host);
        else
            k_ql_forces<float4, LMAX, false><<<blocks, QL_THREADS, 0, s>>>(a, (const float4 *)d_postype, d_head, d_nneigh, d_nlist, d_qlm, (float4 *)d_force, d_bias, bias_host);
        }
    else
        {
        if (half)
            k_ql_forces<double4, LMAX, true><<<blocks, QL_THREADS, 0, s>>>(a, (const double4 *)d_postype, d_head, d_nneigh, d_nlist, d_qlm, (double4 *)d_force, d_bias, bias_host);
        else
            k_ql_forces<double4, LMAX, false><<<blocks, QL_THREADS, 0, s>>>(a, (const double4 *)d_postype, d_head, d_nneigh, d_nlist, d_qlm, (double4 *)d_force, d_bias, bias_host);
        }
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

} // namespace

extern "C" {

size_t mtd_ql_scratch_doubles(unsigned int lmax)
    {
    // block partial sums + Q'_lm + full Q_lm table + Q_l + value
    const size_t n_out = (size_t)(lmax + 1) * (lmax + 2);
    return (size_t)QL_MAX_BLOCKS * n_out + n_out + 2 * (size_t)(lmax + 1) * (lmax + 1) + (lmax + 1) + 1;
    }

// layout of the scratch buffer
static void ql_layout(double *scratch, unsigned int lmax, double **partials, double **qprime, double **qlm, double **ql, double **value)
    {
    const size_t n_out = (size_t)(lmax + 1) * (lmax + 2);
    *partials = scratch;
    *qprime = *partials + (size_t)QL_MAX_BLOCKS * n_out;
    *qlm = *qprime + n_out;
    *ql = *qlm + 2 * (size_t)(lmax + 1) * (lmax + 1);
    *value = *ql + (lmax + 1);
    }

static int ql_dispatch(unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box, const unsigned int *d_head_list,
                       const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist, double rcut, double ron,
                       unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int n_global, double *d_scratch,
                       const double **d_value, const double **d_Ql, const double **d_Qlm, bool accumulate, bool finalize,
                       mtd_stream_t stream)
    {
    if (!d_scratch) return MTD_ERR_INVALID_ARGUMENT;
    if (accumulate && n_particles && (!d_postype || !d_head_list || !d_n_neigh || !d_nlist)) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    if (lmax > 12) return MTD_ERR_UNSUPPORTED;
    double *partials, *qprime, *qlm, *ql, *value;
    ql_layout(d_scratch, lmax, &partials, &qprime, &qlm, &ql, &value);
    unsigned int n_partials = 0;
    hipStream_t s = (hipStream_t)stream;
    int rc;
#define MTD_QL_ACC(LM) accumulate_impl<LM>(n_particles, d_postype, dtype, box, d_head_list, d_n_neigh, d_nlist, half_nlist, rcut, ron, lmax, \
                                           type, Ql_ref, n_global, partials, &n_partials, qprime, qlm, ql, value, accumulate, finalize, s)
    if (lmax <= 4)
        rc = MTD_QL_ACC(4);
    else if (lmax <= 6)
        rc = MTD_QL_ACC(6);
    else if (lmax <= 8)
        rc = MTD_QL_ACC(8);
    else
        rc = MTD_QL_ACC(12);
#undef MTD_QL_ACC
    if (rc) return rc;
    if (d_value) *d_value = value;
    if (d_Ql) *d_Ql = ql;
    if (d_Qlm) *d_Qlm = qlm;
    return MTD_SUCCESS;
    }

int mtd_ql_accumulate(unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box, const unsigned int *d_head_list,
                      const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist, double rcut, double ron,
                      unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int n_global, double *d_scratch,
                      const double **d_value, const double **d_Ql, const double **d_Qlm, mtd_stream_t stream)
    {
    return ql_dispatch(n_particles, d_postype, dtype, box, d_head_list, d_n_neigh, d_nlist, half_nlist, rcut, ron, lmax, type, Ql_ref,
                       n_global, d_scratch, d_value, d_Ql, d_Qlm, true, true, stream);
    }

int mtd_ql_accumulate_local(unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box,
                            const unsigned int *d_head_list, const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist,
                            double rcut, double ron, unsigned int lmax, unsigned int type, unsigned int n_global, double *d_scratch,
                            double **d_sums, unsigned int *n_sums, mtd_stream_t stream)
    {
    if (!d_sums || !n_sums) return MTD_ERR_INVALID_ARGUMENT;
    double zeros[13] = {0};
    int rc = ql_dispatch(n_particles, d_postype, dtype, box, d_head_list, d_n_neigh, d_nlist, half_nlist, rcut, ron, lmax, type, zeros,
                         n_global, d_scratch, nullptr, nullptr, nullptr, true, false, stream);
    if (rc) return rc;
    double *partials, *qprime, *qlm, *ql, *value;
    ql_layout(d_scratch, lmax, &partials, &qprime, &qlm, &ql, &value);
    *d_sums = qprime;
    *n_sums = (lmax + 1) * (lmax + 2);
    return MTD_SUCCESS;
    }

int mtd_ql_finalize(int half_nlist, unsigned int lmax, const double *Ql_ref, unsigned int n_global, double *d_scratch,
                    const double **d_value, const double **d_Ql, const double **d_Qlm, mtd_stream_t stream)
    {
    if (!Ql_ref) return MTD_ERR_INVALID_ARGUMENT;
    // geometry is not used by the finalize step: any valid box / cut-offs satisfy the argument checks
    mtd_box box;
    std::memset(&box, 0, sizeof(box));
    box.L[0] = box.L[1] = box.L[2] = 1.0;
    return ql_dispatch(0, nullptr, MTD_F64, &box, nullptr, nullptr, nullptr, half_nlist, 1.0, 0.5, lmax, 0, Ql_ref, n_global, d_scratch,
                       d_value, d_Ql, d_Qlm, false, true, stream);
    }

int mtd_ql_forces(unsigned int n_particles, const void *d_postype, void *d_force, int dtype, const mtd_box *box,
                  const unsigned int *d_head_list, const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist,
                  double rcut, double ron, unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int n_global,
                  const double *d_scratch, const double *d_bias, double bias_host, mtd_stream_t stream)
    {
    if (!d_scratch || (n_particles && (!d_postype || !d_force || !d_head_list || !d_n_neigh || !d_nlist))) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    if (lmax > 12) return MTD_ERR_UNSUPPORTED;
    if (n_particles == 0) return MTD_SUCCESS;
    double *partials, *qprime, *qlm, *ql, *value;
    ql_layout((double *)d_scratch, lmax, &partials, &qprime, &qlm, &ql, &value);
    hipStream_t s = (hipStream_t)stream;
#define MTD_QL_F(LM) forces_impl<LM>(n_particles, d_postype, d_force, dtype, box, d_head_list, d_n_neigh, d_nlist, half_nlist, rcut, ron, lmax, \
                                     type, Ql_ref, n_global, qlm, d_bias, bias_host, s)
    if (lmax <= 4) return MTD_QL_F(4);
    if (lmax <= 6) return MTD_QL_F(6);
    if (lmax <= 8) return MTD_QL_F(8);
    return MTD_QL_F(12);
#undef MTD_QL_F
    }

} // extern "C"
