/* oracle/mtd_ref_mesh.c — TEST INFRASTRUCTURE ONLY (see mtd_ref.h).
 * Restatement of OrderParameterMesh.cc (CPU path, Scalar = double, single rank => no ghost cells,
 * m_n_ghost_cells = 0, m_grid_dim = m_mesh_points).  Parity unpinned: the reference holds no vectors
 * for this path and its FFT (kiss_fftnd, vendored by HOOMD, absent here) is restated as a plain
 * unnormalised separable DFT (radix-2 when the length is a power of two, O(n^2) otherwise); pinned by
 * analytic KATs (tests/test_oracle_mesh.py) and a numpy.fft cross-check.
 *
 * BoxDim::makeFraction / makeCoordinates / minImage are restated from HOOMD-blue v2 semantics
 * (SURVEY.md App. B); the header itself is not in the reference tree.
 */
#include "mtd_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { double r, i; } cpx;   /* kiss_fft_cpx */

struct ref_mesh
    {
    unsigned int nx, ny, nz, n_cells, n_types;
    double *mode;
    int bug_compat;                 /* 1: interpolation function with the reference's unsigned division (Q6) */
    int initialized;                /* m_is_first_step */
    cpx *mesh, *fourier_mesh, *fourier_mesh_G, *inv_fourier_mesh;
    double *inf_f, *interpolation_f, *k;
    double mode_sq, cv;
    /* convolution kernel table (setTable :148-189) */
    double *table, *table_d;
    unsigned int n_table;
    double k_min, k_max, delta_k;
    int use_table;
    };

/* ---------------------------------------------------------------- BoxDim pieces */

static void make_fraction(const ref_box *b, const double v[3], double f[3])
    {
    double d[3] = { v[0] - b->lo[0], v[1] - b->lo[1], v[2] - b->lo[2] };
    d[0] -= (b->xz - b->yz * b->xy) * d[2] + b->xy * d[1];
    d[1] -= b->yz * d[2];
    f[0] = d[0] / b->L[0];
    f[1] = d[1] / b->L[1];
    f[2] = d[2] / b->L[2];
    }

static void make_coordinates(const ref_box *b, const double f[3], double v[3])
    {
    /* lo + f.x a1 + f.y a2 + f.z a3 */
    v[0] = b->lo[0] + f[0] * b->L[0] + f[1] * b->xy * b->L[1] + f[2] * b->xz * b->L[2];
    v[1] = b->lo[1] + f[1] * b->L[1] + f[2] * b->yz * b->L[2];
    v[2] = b->lo[2] + f[2] * b->L[2];
    }

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

/* ---------------------------------------------------------------- TSC (OrderParameterMesh.cc:457-511) */

static double assign_tsc(double x)                     /* :457-468 */
    {
    double xsq = x * x;
    double xabs = sqrt(xsq);
    if (xsq <= 1.0 / 4.0)
        return 3.0 / 4.0 - xsq;
    else if (xsq <= 9.0 / 4.0)
        return 1.0 / 2.0 * (3.0 / 2.0 - xabs) * (3.0 / 2.0 - xabs);
    else
        return 0.0;
    }

static double assign_tsc_deriv(double x)               /* :470-483 — copysignf even in double builds (Q9) */
    {
    double xsq = x * x;
    double xabs = (double)copysignf((float)x, 1.0f);
    double fac = 3.0 / 2.0 - xabs;
    double ret = 0.0;
    if (xsq <= 1.0 / 4.0)
        ret = -2.0 * x;
    else if (xsq <= 9.0 / 4.0)
        ret = -fac * x / xabs;
    return ret;
    }

static double assign_tsc_fourier(double x)             /* :487-511 */
    {
    const double c[] = { 1.0, -1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0, -1.0 / 39916800.0 };
    double sinc = 0;
    if (x * x <= 1.0)
        {
        double term = 1.0;
        for (unsigned int i = 0; i < 6; ++i)
            {
            sinc += c[i] * term;
            term *= x * x;
            }
        }
    else
        sinc = sin(x) / x;
    return sinc * sinc * sinc;
    }

/* ---------------------------------------------------------------- unnormalised 3-D DFT (kiss_fftnd stand-in) */

static void dft_line(cpx *x, unsigned int n, unsigned int stride, int inverse, cpx *tmp)
    {
    const double sgn = inverse ? 1.0 : -1.0;
    unsigned int pow2 = n && !(n & (n - 1));
    for (unsigned int i = 0; i < n; i++) tmp[i] = x[i * stride];
    if (pow2)
        {
        /* iterative radix-2, bit reversal */
        for (unsigned int i = 1, j = 0; i < n; i++)
            {
            unsigned int bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) { cpx t = tmp[i]; tmp[i] = tmp[j]; tmp[j] = t; }
            }
        for (unsigned int len = 2; len <= n; len <<= 1)
            {
            double ang = sgn * 2.0 * M_PI / len;
            for (unsigned int i = 0; i < n; i += len)
                for (unsigned int j = 0; j < len / 2; j++)
                    {
                    double wr = cos(ang * j), wi = sin(ang * j);
                    cpx u = tmp[i + j], v = tmp[i + j + len / 2];
                    cpx t = { v.r * wr - v.i * wi, v.r * wi + v.i * wr };
                    tmp[i + j].r = u.r + t.r; tmp[i + j].i = u.i + t.i;
                    tmp[i + j + len / 2].r = u.r - t.r; tmp[i + j + len / 2].i = u.i - t.i;
                    }
            }
        for (unsigned int i = 0; i < n; i++) x[i * stride] = tmp[i];
        }
    else
        {
        for (unsigned int k = 0; k < n; k++)
            {
            double sr = 0.0, si = 0.0;
            for (unsigned int j = 0; j < n; j++)
                {
                double ang = sgn * 2.0 * M_PI * (double)((unsigned long long)k * j % n) / n;
                double wr = cos(ang), wi = sin(ang);
                sr += tmp[j].r * wr - tmp[j].i * wi;
                si += tmp[j].r * wi + tmp[j].i * wr;
                }
            x[k * stride].r = sr;
            x[k * stride].i = si;
            }
        }
    }

/* index = x + nx*(y + ny*z); dims passed to kiss as (z,y,x) (:319-325): a plain 3-D transform */
static void fft3d(const cpx *in, cpx *out, unsigned int nx, unsigned int ny, unsigned int nz, int inverse)
    {
    unsigned int nmax = nx > ny ? nx : ny;
    if (nz > nmax) nmax = nz;
    cpx *tmp = (cpx *)malloc(sizeof(cpx) * nmax);
    memcpy(out, in, sizeof(cpx) * nx * ny * nz);
    for (unsigned int z = 0; z < nz; z++)
        for (unsigned int y = 0; y < ny; y++) dft_line(out + nx * (y + ny * z), nx, 1, inverse, tmp);
    for (unsigned int z = 0; z < nz; z++)
        for (unsigned int x = 0; x < nx; x++) dft_line(out + x + nx * ny * z, ny, nx, inverse, tmp);
    for (unsigned int y = 0; y < ny; y++)
        for (unsigned int x = 0; x < nx; x++) dft_line(out + x + nx * y, nz, nx * ny, inverse, tmp);
    free(tmp);
    }

/* ---------------------------------------------------------------- class */

ref_mesh *ref_mesh_create(unsigned int nx, unsigned int ny, unsigned int nz, unsigned int n_types, const double *mode)
    {
    ref_mesh *m = (ref_mesh *)calloc(1, sizeof(ref_mesh));
    m->nx = nx; m->ny = ny; m->nz = nz;
    m->n_cells = nx * ny * nz;
    m->n_types = n_types;
    m->mode = (double *)malloc(sizeof(double) * n_types);
    memcpy(m->mode, mode, sizeof(double) * n_types);
    m->bug_compat = 1;
    m->mesh = (cpx *)calloc(m->n_cells, sizeof(cpx));
    m->fourier_mesh = (cpx *)calloc(m->n_cells, sizeof(cpx));
    m->fourier_mesh_G = (cpx *)calloc(m->n_cells, sizeof(cpx));
    m->inv_fourier_mesh = (cpx *)calloc(m->n_cells, sizeof(cpx));
    m->inf_f = (double *)calloc(m->n_cells, sizeof(double));
    m->interpolation_f = (double *)calloc(m->n_cells, sizeof(double));
    m->k = (double *)calloc(3 * (size_t)m->n_cells, sizeof(double));
    return m;
    }

void ref_mesh_destroy(ref_mesh *m)
    {
    if (!m) return;
    free(m->mode); free(m->mesh); free(m->fourier_mesh); free(m->fourier_mesh_G); free(m->inv_fourier_mesh);
    free(m->inf_f); free(m->interpolation_f); free(m->k);
    free(m->table); free(m->table_d);
    free(m);
    }

void ref_mesh_set_bug_compat(ref_mesh *m, int on) { m->bug_compat = on; m->initialized = 0; }
double ref_mesh_mode_sq(const ref_mesh *m) { return m->mode_sq; }

void *ref_mesh_array(ref_mesh *m, int which)
    {
    switch (which)
        {
        case 0: return m->mesh;
        case 1: return m->fourier_mesh;
        case 2: return m->fourier_mesh_G;
        case 3: return m->inv_fourier_mesh;
        case 4: return m->interpolation_f;
        case 5: return m->inf_f;
        case 6: return m->k;
        }
    return NULL;
    }

/* computeInfluenceFunction, :344-453 (no convolution table: m_use_table = false => inf_f = 1, unused anyway, Q7) */
static void compute_influence_function(ref_mesh *m, const ref_box *global_box)
    {
    double a1[3] = { global_box->L[0], 0.0, 0.0 };
    double a2[3] = { global_box->xy * global_box->L[1], global_box->L[1], 0.0 };
    double a3[3] = { global_box->xz * global_box->L[2], global_box->yz * global_box->L[2], global_box->L[2] };
    double V = global_box->L[0] * global_box->L[1] * global_box->L[2];
    double two_pi = 2.0 * M_PI;
    double b1[3] = { two_pi * (a2[1] * a3[2] - a2[2] * a3[1]) / V, two_pi * (a2[2] * a3[0] - a2[0] * a3[2]) / V, two_pi * (a2[0] * a3[1] - a2[1] * a3[0]) / V };
    double b2[3] = { two_pi * (a3[1] * a1[2] - a3[2] * a1[1]) / V, two_pi * (a3[2] * a1[0] - a3[0] * a1[2]) / V, two_pi * (a3[0] * a1[1] - a3[1] * a1[0]) / V };
    double b3[3] = { two_pi * (a1[1] * a2[2] - a1[2] * a2[1]) / V, two_pi * (a1[2] * a2[0] - a1[0] * a2[2]) / V, two_pi * (a1[0] * a2[1] - a1[1] * a2[0]) / V };
    unsigned int gx = m->nx, gy = m->ny, gz = m->nz;

    for (unsigned int cell_idx = 0; cell_idx < m->n_cells; ++cell_idx)
        {
        unsigned int wz = cell_idx / (m->ny * m->nx);
        unsigned int wy = (cell_idx - wz * m->nx * m->ny) / m->nx;
        unsigned int wx = cell_idx % m->nx;
        int n[3] = { (int)wx, (int)wy, (int)wz };
        if (n[0] >= (int)(gx / 2 + gx % 2)) n[0] -= (int)gx;          /* Miller indices :417-422 */
        if (n[1] >= (int)(gy / 2 + gy % 2)) n[1] -= (int)gy;
        if (n[2] >= (int)(gz / 2 + gz % 2)) n[2] -= (int)gz;
        for (int c = 0; c < 3; c++) m->k[3 * (size_t)cell_idx + c] = n[0] * b1[c] + n[1] * b2[c] + n[2] * b3[c];
        double val = 1.0;
        const double *kv = m->k + 3 * (size_t)cell_idx;
        double knorm = sqrt(kv[0] * kv[0] + kv[1] * kv[1] + kv[2] * kv[2]);
        if (m->use_table && knorm >= m->k_min && knorm < m->k_max)    /* :431-443; stored, never applied to the mesh (Q7) */
            {
            double value_f = (knorm - m->k_min) / m->delta_k;
            unsigned int value_i = (unsigned int)value_f;
            double K0 = m->table[value_i], K1 = m->table[value_i + 1];
            val = K0 + (value_f - (double)value_i) * (K1 - K0);
            }
        m->inf_f[cell_idx] = val;
        double kH[3];
        if (m->bug_compat)
            {
            /* :448 — `n.x/global_dim.x` is int / unsigned: the int is converted to unsigned (Q6) */
            kH[0] = (M_PI * 2.0) * (double)((unsigned int)n[0] / gx);
            kH[1] = (M_PI * 2.0) * (double)((unsigned int)n[1] / gy);
            kH[2] = (M_PI * 2.0) * (double)((unsigned int)n[2] / gz);
            }
        else
            {
            kH[0] = (M_PI * 2.0) * ((double)n[0] / gx);
            kH[1] = (M_PI * 2.0) * ((double)n[1] / gy);
            kH[2] = (M_PI * 2.0) * ((double)n[2] / gz);
            }
        m->interpolation_f[cell_idx] = assign_tsc_fourier(kH[0]) * assign_tsc_fourier(kH[1]) * assign_tsc_fourier(kH[2]);
        }
    }

/* cell + shift of one particle, shared by assignParticles (:540-573) and interpolateForces (:784-812) */
static void locate(const ref_mesh *m, const ref_box *box, const double pos[3], int cell[3], double shift[3])
    {
    double f[3];
    make_fraction(box, pos, f);
    double reduced[3] = { f[0] * (double)m->nx, f[1] * (double)m->ny, f[2] * (double)m->nz };
    int ix = (int)reduced[0], iy = (int)reduced[1], iz = (int)reduced[2];
    if (ix == (int)m->nx) ix = 0;                                      /* particles on the boundary :556-561 */
    if (iy == (int)m->ny) iy = 0;
    if (iz == (int)m->nz) iz = 0;
    double center_f[3] = { ((double)ix + 0.5) / m->nx, ((double)iy + 0.5) / m->ny, ((double)iz + 0.5) / m->nz };
    double c_cart[3];
    make_coordinates(box, center_f, c_cart);
    double shift_cart[3] = { pos[0] - c_cart[0], pos[1] - c_cart[1], pos[2] - c_cart[2] };
    min_image(box, shift_cart);
    double tmp[3] = { shift_cart[0] + box->lo[0], shift_cart[1] + box->lo[1], shift_cart[2] + box->lo[2] };
    double shift_f[3];
    make_fraction(box, tmp, shift_f);
    shift[0] = shift_f[0] * m->nx;
    shift[1] = shift_f[1] * m->ny;
    shift[2] = shift_f[2] * m->nz;
    cell[0] = ix; cell[1] = iy; cell[2] = iz;
    }

static int wrap(int i, int n)
    {
    if (i == n) return 0;
    if (i < 0) return i + n;
    return i;
    }

/* assignParticles, :517-640 */
static void assign_particles(ref_mesh *m, unsigned int N, const double *postype, const ref_box *box)
    {
    memset(m->mesh, 0, sizeof(cpx) * m->n_cells);
    m->mode_sq = 0.0;
    for (unsigned int idx = 0; idx < N; ++idx)
        {
        const double *p = postype + 4 * (size_t)idx;
        unsigned int type = (unsigned int)p[3];
        int c[3];
        double shift[3];
        locate(m, box, p, c, shift);
        for (int i = -1; i <= 1; ++i)
            for (int j = -1; j <= 1; ++j)
                for (int k = -1; k <= 1; ++k)
                    {
                    int ni = wrap(c[0] + i, (int)m->nx), nj = wrap(c[1] + j, (int)m->ny), nk = wrap(c[2] + k, (int)m->nz);
                    double density_fraction = assign_tsc(shift[0] - i) * assign_tsc(shift[1] - j) * assign_tsc(shift[2] - k);
                    unsigned int neigh_idx = ni + m->nx * (nj + m->ny * nk);
                    m->mesh[neigh_idx].r += m->mode[type] * density_fraction;
                    }
        m->mode_sq += m->mode[type] * m->mode[type];
        }
    }

/* updateMeshes, :642-747 */
static void update_meshes(ref_mesh *m, unsigned int N_global)
    {
    fft3d(m->mesh, m->fourier_mesh, m->nx, m->ny, m->nz, 0);
    for (unsigned int k = 0; k < m->n_cells; ++k)
        {
        cpx f = m->fourier_mesh[k];
        f.r /= (double)N_global;
        f.i /= (double)N_global;
        double val = f.r * f.r + f.i * f.i;
        m->fourier_mesh_G[k].r = f.r * val;
        m->fourier_mesh_G[k].i = f.i * val;
        double diagonal_term = 0.5 * m->interpolation_f[k] * m->interpolation_f[k] * m->mode_sq / (double)N_global / (double)N_global;
        m->fourier_mesh_G[k].r -= f.r * diagonal_term;
        m->fourier_mesh_G[k].i -= f.i * diagonal_term;
        m->fourier_mesh[k] = f;
        }
    fft3d(m->fourier_mesh_G, m->inv_fourier_mesh, m->nx, m->ny, m->nz, 1);
    }

/* computeCV, :866-923 */
static double compute_cv(ref_mesh *m, unsigned int N_global)
    {
    double sum = 0.0;
    for (unsigned int k = 0; k < m->n_cells; ++k)
        {
        if (k == 0) continue;                                           /* exclude DC bin */
        sum += m->fourier_mesh_G[k].r * m->fourier_mesh[k].r + m->fourier_mesh_G[k].i * m->fourier_mesh[k].i;
        double norm2 = m->fourier_mesh[k].r * m->fourier_mesh[k].r + m->fourier_mesh[k].i * m->fourier_mesh[k].i;
        double diagonal_term = 0.5 * norm2 * m->interpolation_f[k] * m->interpolation_f[k] * m->mode_sq / (double)N_global / (double)N_global;
        sum -= diagonal_term;
        }
    sum *= 1.0 / 2.0;
    return sum;
    }

/* getCurrentValue, :925-968 */
double ref_mesh_cv(ref_mesh *m, unsigned int N, const double *postype, const ref_box *box, unsigned int N_global)
    {
    if (!m->initialized)
        {
        compute_influence_function(m, box);
        m->initialized = 1;
        }
    assign_particles(m, N, postype, box);
    update_meshes(m, N_global);
    m->cv = compute_cv(m, N_global);
    return m->cv;
    }

/* the two halves of getCurrentValue for particle-sharded checks: spread this shard (mesh + mode_sq), [sum over shards],
 * then FFT / updateMeshes / computeCV on the summed mesh (the reference sums mode_sq over ranks too, :626-636) */
void ref_mesh_assign(ref_mesh *m, unsigned int N, const double *postype, const ref_box *box)
    {
    if (!m->initialized)
        {
        compute_influence_function(m, box);
        m->initialized = 1;
        }
    assign_particles(m, N, postype, box);
    }

void ref_mesh_set_mode_sq(ref_mesh *m, double mode_sq) { m->mode_sq = mode_sq; }

double ref_mesh_spectral(ref_mesh *m, unsigned int N_global)
    {
    update_meshes(m, N_global);
    m->cv = compute_cv(m, N_global);
    return m->cv;
    }

/* setTable :148-189 (returns 0, or -1 where the reference throws) and setUseTable */
int ref_mesh_set_table(ref_mesh *m, const double *K, const double *d_K, unsigned int n, double kmin, double kmax)
    {
    if (kmin < 0 || kmax < 0 || kmax <= kmin) return -1;
    free(m->table); free(m->table_d);
    m->table = (double *)malloc(sizeof(double) * n);
    m->table_d = (double *)malloc(sizeof(double) * n);
    memcpy(m->table, K, sizeof(double) * n);
    memcpy(m->table_d, d_K, sizeof(double) * n);
    m->n_table = n;
    m->k_min = kmin;
    m->k_max = kmax;
    m->delta_k = (kmax - kmin) / (double)(n - 1);
    m->initialized = 0;
    return 0;
    }

void ref_mesh_set_use_table(ref_mesh *m, int on) { m->use_table = on; m->initialized = 0; }

/* computeQmax :1108-1179 on the Fourier mesh of the last ref_mesh_cv: out = (qx, qy, qz, sq_max) */
void ref_mesh_qmax(const ref_mesh *m, unsigned int N_global, double *out)
    {
    double max_amplitude = 0.0;
    double q_max[3] = { 0.0, 0.0, 0.0 };
    for (unsigned int kidx = 0; kidx < m->n_cells; ++kidx)
        {
        double a = m->fourier_mesh[kidx].r * m->fourier_mesh[kidx].r + m->fourier_mesh[kidx].i * m->fourier_mesh[kidx].i;
        if (a > max_amplitude)
            {
            for (int c = 0; c < 3; c++) q_max[c] = m->k[3 * (size_t)kidx + c];
            max_amplitude = a;
            }
        }
    out[0] = q_max[0]; out[1] = q_max[1]; out[2] = q_max[2];
    out[3] = max_amplitude * (double)N_global;
    }

/* computeVirial :970-1050: virial[6] = bias * sum over k != 0 */
void ref_mesh_virial(const ref_mesh *m, unsigned int N_global, double bias, double *virial)
    {
    for (int i = 0; i < 6; ++i) virial[i] = 0.0;
    for (unsigned int kidx = 0; kidx < m->n_cells; ++kidx)
        {
        if (kidx == 0) continue;                                      /* exclude DC bin */
        cpx fourier = m->fourier_mesh[kidx];
        const double *k = m->k + 3 * (size_t)kidx;
        double ksq = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
        double knorm = sqrt(ksq);
        double kfac = 1.0 / 2.0 / knorm;
        double val_D = 0.0;
        if (m->use_table && knorm >= m->k_min && knorm < m->k_max)
            {
            double value_f = (knorm - m->k_min) / m->delta_k;
            unsigned int value_i = (unsigned int)value_f;
            double dK0 = m->table_d[value_i], dK1 = m->table_d[value_i + 1];
            val_D = dK0 + (value_f - (double)value_i) * (dK1 - dK0);
            }
        kfac *= val_D;
        double val = (fourier.r * fourier.r + fourier.i * fourier.i) / (double)N_global;
        double rhog = (fourier.r * fourier.r + fourier.i * fourier.i) * val / (double)N_global;
        virial[0] += rhog * kfac * k[0] * k[0];
        virial[1] += rhog * kfac * k[0] * k[1];
        virial[2] += rhog * kfac * k[0] * k[2];
        virial[3] += rhog * kfac * k[1] * k[1];
        virial[4] += rhog * kfac * k[1] * k[2];
        virial[5] += rhog * kfac * k[2] * k[2];
        }
    for (int i = 0; i < 6; ++i) virial[i] = bias * virial[i];
    }

/* interpolateForces, :749-864 (after getCurrentValue of the same snapshot) */
void ref_mesh_forces(ref_mesh *m, unsigned int N, const double *postype, const ref_box *box, unsigned int N_global,
                     double bias, double *force_out)
    {
    double a1[3] = { box->L[0], 0.0, 0.0 };
    double a2[3] = { box->xy * box->L[1], box->L[1], 0.0 };
    double a3[3] = { box->xz * box->L[2], box->yz * box->L[2], box->L[2] };
    double V = box->L[0] * box->L[1] * box->L[2];
    double b1[3] = { (a2[1] * a3[2] - a2[2] * a3[1]) / V, (a2[2] * a3[0] - a2[0] * a3[2]) / V, (a2[0] * a3[1] - a2[1] * a3[0]) / V };
    double b2[3] = { (a3[1] * a1[2] - a3[2] * a1[1]) / V, (a3[2] * a1[0] - a3[0] * a1[2]) / V, (a3[0] * a1[1] - a3[1] * a1[0]) / V };
    double b3[3] = { (a1[1] * a2[2] - a1[2] * a2[1]) / V, (a1[2] * a2[0] - a1[0] * a2[2]) / V, (a1[0] * a2[1] - a1[1] * a2[0]) / V };

    for (unsigned int idx = 0; idx < N; ++idx)
        {
        const double *p = postype + 4 * (size_t)idx;
        unsigned int type = (unsigned int)p[3];
        double mode = m->mode[type];
        int c[3];
        double shift[3];
        locate(m, box, p, c, shift);
        double force[3] = { 0.0, 0.0, 0.0 };
        for (int i = -1; i <= 1; ++i)
            for (int j = -1; j <= 1; ++j)
                for (int k = -1; k <= 1; ++k)
                    {
                    int ni = wrap(c[0] + i, (int)m->nx), nj = wrap(c[1] + j, (int)m->ny), nk = wrap(c[2] + k, (int)m->nz);
                    double dx[3] = { shift[0] - i, shift[1] - j, shift[2] - k };
                    unsigned int neigh_idx = ni + m->nx * (nj + m->ny * nk);
                    double inv_r = m->inv_fourier_mesh[neigh_idx].r;
                    double wx = assign_tsc(dx[0]), wy = assign_tsc(dx[1]), wz = assign_tsc(dx[2]);
                    double dwx = assign_tsc_deriv(dx[0]), dwy = assign_tsc_deriv(dx[1]), dwz = assign_tsc_deriv(dx[2]);
                    for (int d = 0; d < 3; d++)
                        {
                        force[d] += -(double)m->nx * b1[d] * mode * dwx * wy * wz * inv_r;     /* :855 */
                        force[d] += -(double)m->ny * b2[d] * mode * wx * dwy * wz * inv_r;     /* :856 */
                        force[d] += -(double)m->nz * b3[d] * mode * wx * wy * dwz * inv_r;     /* :857 */
                        }
                    }
        for (int d = 0; d < 3; d++) force[d] *= 2.0 / (double)N_global * bias;                 /* :861 */
        force_out[4 * (size_t)idx + 0] = force[0];
        force_out[4 * (size_t)idx + 1] = force[1];
        force_out[4 * (size_t)idx + 2] = force[2];
        force_out[4 * (size_t)idx + 3] = 0.0;
        }
    }
