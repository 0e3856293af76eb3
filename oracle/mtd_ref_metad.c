/* oracle/mtd_ref_metad.c — TEST INFRASTRUCTURE ONLY (see mtd_ref.h).
 * Restatement of IndexGrid.cc and of the grid branch of IntegratorMetaDynamics.cc
 * (CPU path, Scalar = double, single rank).  ref_index_* is pinned against the reference's own
 * IndexGrid.cc compiled into oracle/_ref; everything else is parity-unpinned (analytic KATs only).
 */
#include "mtd_ref.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ IndexGrid.cc */

/* IndexGrid::getIndex, IndexGrid.cc:33-44 with factors from setLengths :20-31 */
unsigned int ref_index_get(unsigned int dim, const unsigned int *lengths, const unsigned int *coords)
    {
    unsigned int idx = 0;
    unsigned int factor = 1;
    for (unsigned int i = 0; i < dim; i++)
        {
        factor = (i == 0) ? 1 : (lengths[i - 1] * factor);
        idx += coords[i] * factor;
        }
    return idx;
    }

/* IndexGrid::getCoordinates, IndexGrid.cc:46-58 */
void ref_index_coords(unsigned int dim, const unsigned int *lengths, unsigned int idx, unsigned int *coords)
    {
    unsigned int factors[16];
    for (unsigned int i = 0; i < dim; i++)
        factors[i] = (i == 0) ? 1 : (lengths[i - 1] * factors[i - 1]);
    unsigned int rest = idx;
    for (int i = (int)dim - 1; i >= 0; i--)
        {
        coords[i] = rest / factors[i];
        rest -= coords[i] * factors[i];
        }
    }

/* ------------------------------------------------------------------ engine state */

#define REF_MAX_CV 16

struct ref_metad
    {
    unsigned int n_cv;
    double sigma[REF_MAX_CV], cv_min[REF_MAX_CV], cv_max[REF_MAX_CV];
    unsigned int num_points[REF_MAX_CV];
    double sigma_inv[REF_MAX_CV * REF_MAX_CV];
    double W, T_shift, temp;
    unsigned int stride;
    int mode, add_bias;
    unsigned int num_gaussians;
    double curr_bias_potential, curr_reweight;
    unsigned int n_oob;
    unsigned int len;
    double *grid, *grid_delta, *grid_reweighted, *grid_weight, *sigma_grid, *sigma_grid_delta;
    unsigned int *hist, *hist_delta, *hist_gauss, *hist_gauss_delta;
    };

ref_metad *ref_metad_create(unsigned int n_cv, const double *sigma, const double *cv_min,
                            const double *cv_max, const unsigned int *num_points,
                            double W, double T_shift, double T, unsigned int stride,
                            int mode, int add_bias)
    {
    if (n_cv == 0 || n_cv > REF_MAX_CV) return NULL;
    ref_metad *m = (ref_metad *)calloc(1, sizeof(ref_metad));
    m->n_cv = n_cv;
    m->W = W; m->T_shift = T_shift; m->temp = T; m->stride = stride;   /* ctor :33-56 */
    m->mode = mode; m->add_bias = add_bias;
    m->curr_reweight = 1.0;
    unsigned int len = 1;
    for (unsigned int i = 0; i < n_cv; i++)
        {
        /* setGrid(true) input checks, :798-812 */
        if (!(cv_min[i] < cv_max[i]) || num_points[i] < 2) { free(m); return NULL; }
        m->sigma[i] = sigma[i]; m->cv_min[i] = cv_min[i]; m->cv_max[i] = cv_max[i];
        m->num_points[i] = num_points[i];
        m->sigma_inv[i * n_cv + i] = 1.0 / sigma[i];                    /* prepRun :171-177 */
        len *= num_points[i];
        }
    m->len = len;
    /* setupGrid :590-661 — GPUArray storage is zero-initialised; weight grid reset to one */
    m->grid = (double *)calloc(len, sizeof(double));
    m->grid_delta = (double *)calloc(len, sizeof(double));
    m->grid_reweighted = (double *)calloc(len, sizeof(double));
    m->grid_weight = (double *)calloc(len, sizeof(double));
    m->sigma_grid = (double *)calloc(len, sizeof(double));
    m->sigma_grid_delta = (double *)calloc(len, sizeof(double));
    m->hist = (unsigned int *)calloc(len, sizeof(unsigned int));
    m->hist_delta = (unsigned int *)calloc(len, sizeof(unsigned int));
    m->hist_gauss = (unsigned int *)calloc(len, sizeof(unsigned int));
    m->hist_gauss_delta = (unsigned int *)calloc(len, sizeof(unsigned int));
    for (unsigned int i = 0; i < len; i++) m->grid_weight[i] = 1.0;    /* :657-658 */
    return m;
    }

void ref_metad_destroy(ref_metad *m)
    {
    if (!m) return;
    free(m->grid); free(m->grid_delta); free(m->grid_reweighted); free(m->grid_weight);
    free(m->sigma_grid); free(m->sigma_grid_delta);
    free(m->hist); free(m->hist_delta); free(m->hist_gauss); free(m->hist_gauss_delta);
    free(m);
    }

unsigned int ref_metad_num_elements(const ref_metad *m) { return m->len; }
void ref_metad_set_stride(ref_metad *m, unsigned int stride) { m->stride = stride; }
void ref_metad_set_add_bias(ref_metad *m, int add_bias) { m->add_bias = add_bias; }
void ref_metad_set_mode(ref_metad *m, int mode) { m->mode = mode; }
void ref_metad_set_sigma_inv(ref_metad *m, const double *s) { memcpy(m->sigma_inv, s, sizeof(double) * m->n_cv * m->n_cv); }
double ref_metad_curr_bias(const ref_metad *m) { return m->curr_bias_potential; }
double ref_metad_curr_weight(const ref_metad *m) { return m->curr_reweight; }
unsigned int ref_metad_num_gaussians(const ref_metad *m) { return m->num_gaussians; }
unsigned int ref_metad_num_oob_warnings(const ref_metad *m) { return m->n_oob; }

void ref_metad_reset_histogram(ref_metad *m)                            /* :1195-1203 */
    {
    memset(m->hist, 0, sizeof(unsigned int) * m->len);
    memset(m->hist_delta, 0, sizeof(unsigned int) * m->len);
    }

void *ref_metad_array(ref_metad *m, int which)
    {
    switch (which)
        {
        case 0: return m->grid;
        case 1: return m->grid_delta;
        case 2: return m->grid_reweighted;
        case 3: return m->grid_weight;
        case 4: return m->sigma_grid;
        case 5: return m->sigma_grid_delta;
        case 6: return m->hist;
        case 7: return m->hist_delta;
        case 8: return m->hist_gauss;
        case 9: return m->hist_gauss_delta;
        }
    return NULL;
    }

/* ------------------------------------------------------------------ interpolation / derivative */

/* interpolateGrid, :663-736 */
double ref_metad_interpolate(ref_metad *m, const double *val, int reweight)
    {
    unsigned int dim = m->n_cv;
    unsigned int lower_idx[REF_MAX_CV], upper_idx[REF_MAX_CV];
    double rel_delta[REF_MAX_CV];

    for (unsigned int cv = 0; cv < dim; cv++)
        {
        double delta = (m->cv_max[cv] - m->cv_min[cv]) / (m->num_points[cv] - 1);   /* :675 */

        if (val[cv] < m->cv_min[cv] || val[cv] >= m->cv_max[cv])                     /* :677-683 */
            {
            m->n_oob++;           /* reference prints a warning and assumes zero bias */
            return 0.0;
            }

        int lower = (int)((val[cv] - m->cv_min[cv]) / delta);                        /* :685 */
        int upper = lower + 1;

        if (upper >= (int)m->num_points[cv])                                         /* :689-693 */
            {
            lower--;
            upper--;
            }

        double lower_bound = m->cv_min[cv] + delta * lower;
        double upper_bound = m->cv_min[cv] + delta * upper;
        lower_idx[cv] = (unsigned int)lower;
        upper_idx[cv] = (unsigned int)upper;
        rel_delta[cv] = (val[cv] - lower_bound) / (upper_bound - lower_bound);       /* :699 */
        }

    unsigned int n_term = 1u << dim;
    double res = 0.0;
    for (unsigned int bits = 0; bits < n_term; ++bits)                               /* :711-733 */
        {
        unsigned int coords[REF_MAX_CV];
        double term = 1.0;
        for (unsigned int i = 0; i < dim; i++)
            {
            if (bits & (1u << i))
                {
                coords[i] = lower_idx[i];
                term *= (1.0 - rel_delta[i]);
                }
            else
                {
                coords[i] = upper_idx[i];
                term *= rel_delta[i];
                }
            }
        unsigned int idx = ref_index_get(dim, m->num_points, coords);
        double v = reweight ? m->grid_weight[idx] : m->grid[idx];
        term *= v;
        res += term;
        }
    return res;
    }

/* biasPotentialDerivative, :738-776 */
double ref_metad_derivative(ref_metad *m, unsigned int cv, const double *val)
    {
    double val1[REF_MAX_CV], val2[REF_MAX_CV];
    double delta = (m->cv_max[cv] - m->cv_min[cv]) / (double)(m->num_points[cv] - 1);
    memcpy(val1, val, sizeof(double) * m->n_cv);
    memcpy(val2, val, sizeof(double) * m->n_cv);
    if (val[cv] - delta < m->cv_min[cv])
        {
        val2[cv] += delta;                         /* forward difference :747-755 */
        double y2 = ref_metad_interpolate(m, val2, 0);
        double y1 = ref_metad_interpolate(m, val, 0);
        return (y2 - y1) / delta;
        }
    else if (val[cv] + delta > m->cv_max[cv])
        {
        val2[cv] -= delta;                         /* backward difference :757-764 */
        double y1 = ref_metad_interpolate(m, val2, 0);
        double y2 = ref_metad_interpolate(m, val, 0);
        return (y2 - y1) / delta;
        }
    else
        {
        val1[cv] -= delta;                         /* central difference :766-775 */
        val2[cv] += delta;
        double y1 = ref_metad_interpolate(m, val1, 0);
        double y2 = ref_metad_interpolate(m, val2, 0);
        return (y2 - y1) / (2.0 * delta);
        }
    }

/* sigmaDeterminant, :1296-1313 (Eigen determinant of the n_cv x n_cv inverse-sigma matrix;
 * restated as Gaussian elimination with partial pivoting — Eigen is not in the tree) */
double ref_metad_sigma_determinant(const ref_metad *m)
    {
    unsigned int n = m->n_cv;
    double a[REF_MAX_CV * REF_MAX_CV];
    memcpy(a, m->sigma_inv, sizeof(double) * n * n);
    if (n == 1) return a[0];
    if (n == 2) return a[0] * a[3] - a[1] * a[2];
    double det = 1.0;
    for (unsigned int c = 0; c < n; c++)
        {
        unsigned int p = c;
        for (unsigned int r = c + 1; r < n; r++)
            if (fabs(a[r * n + c]) > fabs(a[p * n + c])) p = r;
        if (a[p * n + c] == 0.0) return 0.0;
        if (p != c)
            {
            for (unsigned int k = 0; k < n; k++) { double t = a[c * n + k]; a[c * n + k] = a[p * n + k]; a[p * n + k] = t; }
            det = -det;
            }
        det *= a[c * n + c];
        for (unsigned int r = c + 1; r < n; r++)
            {
            double f = a[r * n + c] / a[c * n + c];
            for (unsigned int k = c; k < n; k++) a[r * n + k] -= f * a[c * n + k];
            }
        }
    return det;
    }

/* ------------------------------------------------------------------ per-step pieces */

/* updateGrid, :1002-1047 (CPU: overwrites grid_delta, Gaussian evaluated in double — SURVEY Q11;
 * exponent is 1/2 sum_ij d_i d_j (sigma_inv_ij)^2, element-wise square — SURVEY Q12) */
void ref_update_grid(unsigned int dim, const unsigned int *lengths, const double *cv_min, const double *cv_max,
                     const double *sigma_inv, const double *current_val, double scal, double W,
                     double *grid_delta)
    {
    unsigned int len = 1;
    for (unsigned int i = 0; i < dim; i++) len *= lengths[i];
    unsigned int coords[REF_MAX_CV];

    for (unsigned int grid_idx = 0; grid_idx < len; grid_idx++)
        {
        ref_index_coords(dim, lengths, grid_idx, coords);
        double gauss_exp = 0.0;
        for (unsigned int cv_i = 0; cv_i < dim; ++cv_i)
            {
            double delta_i = (cv_max[cv_i] - cv_min[cv_i]) / (lengths[cv_i] - 1);
            double val_i = cv_min[cv_i] + coords[cv_i] * delta_i;
            double d_i = val_i - current_val[cv_i];
            for (unsigned int cv_j = 0; cv_j < dim; ++cv_j)
                {
                double delta_j = (cv_max[cv_j] - cv_min[cv_j]) / (lengths[cv_j] - 1);
                double val_j = cv_min[cv_j] + coords[cv_j] * delta_j;
                double d_j = val_j - current_val[cv_j];
                double sigma_inv_ij = sigma_inv[cv_i * dim + cv_j];
                gauss_exp += d_i * d_j * (1.0 / 2.0) * (sigma_inv_ij * sigma_inv_ij);   /* :1037 */
                }
            }
        double gauss = exp(-gauss_exp);
        grid_delta[grid_idx] = W * scal * gauss;                                        /* :1043 */
        }
    }

/* shared by updateHistogram (:1092-1119) and updateSigmaGrid (:1122-1155): floor-bin of the CV.
 * Scalar -> unsigned conversion of a negative or huge quotient is UB in the reference; treated
 * as off-grid here. */
static int bin_of(const ref_metad *m, const double *current_val, unsigned int *grid_idx)
    {
    unsigned int grid_coord[REF_MAX_CV];
    int on_grid = 1;
    for (unsigned int cv_i = 0; cv_i < m->n_cv; ++cv_i)
        {
        double delta = (m->cv_max[cv_i] - m->cv_min[cv_i]) / (m->num_points[cv_i] - 1);
        double q = (current_val[cv_i] - m->cv_min[cv_i]) / delta;
        if (!(q > -1.0) || !(q < 4294967296.0))
            {
            on_grid = 0;
            grid_coord[cv_i] = 0;
            continue;
            }
        grid_coord[cv_i] = (unsigned int)q;       /* truncation toward zero: (-1,0) -> 0 like the cast */
        if (grid_coord[cv_i] >= m->num_points[cv_i]) on_grid = 0;
        }
    if (on_grid) *grid_idx = ref_index_get(m->n_cv, m->num_points, grid_coord);
    return on_grid;
    }

static void update_histogram(ref_metad *m, const double *current_val)       /* :1092-1119 */
    {
    unsigned int grid_idx;
    if (bin_of(m, current_val, &grid_idx)) m->hist_delta[grid_idx]++;
    }

static void update_sigma_grid(ref_metad *m, const double *current_val)      /* :1122-1155 */
    {
    unsigned int grid_idx;
    if (bin_of(m, current_val, &grid_idx))
        {
        m->sigma_grid_delta[grid_idx] += ref_metad_sigma_determinant(m);
        m->hist_gauss_delta[grid_idx]++;
        }
    }

static void update_reweighted_estimator(ref_metad *m)                       /* :1053-1090 */
    {
    double avg_delta_V = 0.0, norm = 0.0;
    for (unsigned int g = 0; g < m->len; g++)
        {
        m->grid_reweighted[g] += (double)m->hist_delta[g];
        avg_delta_V += m->grid_reweighted[g] * m->grid_delta[g];
        norm += m->grid_reweighted[g];
        }
    avg_delta_V /= norm;                                                     /* :1077 (norm==0 -> NaN) */
    for (unsigned int g = 0; g < m->len; g++)
        {
        double delta_V = m->grid_delta[g];
        double fac = exp(-(delta_V - avg_delta_V) / m->temp);                /* :1084 — T, not deltaT */
        m->grid_reweighted[g] *= fac;
        m->grid_weight[g] /= fac;
        }
    }

/* updateBiasPotential :363-391: histogram every step; on deposit steps sigma grid, scal, updateGrid */
int ref_metad_update_phase_a(ref_metad *m, unsigned int timestep, const double *current_val)
    {
    update_histogram(m, current_val);                                        /* :366 */
    if (m->add_bias && (timestep % m->stride == 0))                          /* :368 */
        {
        update_sigma_grid(m, current_val);                                   /* :371 */
        double scal = 1.0;
        if (m->mode == REF_MODE_WELL_TEMPERED)                               /* :374-379 */
            {
            double V = ref_metad_interpolate(m, current_val, 0);
            scal = exp(-V / m->T_shift);
            }
        ref_update_grid(m->n_cv, m->num_points, m->cv_min, m->cv_max, m->sigma_inv, current_val,
                        scal, m->W, m->grid_delta);                          /* :387-389 */
        return 1;
        }
    return 0;
    }

/* updateBiasPotential :412-451: reweight, accumulate + clear, derivative, V, weight */
void ref_metad_update_phase_b(ref_metad *m, int deposited, const double *current_val, double *bias_out)
    {
    if (deposited)
        {
        update_reweighted_estimator(m);                                      /* :413 */
        for (unsigned int i = 0; i < m->len; ++i)                            /* :426-437 */
            {
            m->grid[i] += m->grid_delta[i];
            m->sigma_grid[i] += m->sigma_grid_delta[i];
            m->hist[i] += m->hist_delta[i];
            m->hist_gauss[i] += m->hist_gauss_delta[i];
            m->grid_delta[i] = 0.0;
            m->sigma_grid_delta[i] = 0.0;
            m->hist_delta[i] = 0;
            m->hist_gauss_delta[i] = 0;
            }
        m->num_gaussians++;                                                  /* :440 */
        }
    for (unsigned int cv_idx = 0; cv_idx < m->n_cv; ++cv_idx)                /* :444-445 */
        bias_out[cv_idx] = ref_metad_derivative(m, cv_idx, current_val);
    m->curr_bias_potential = ref_metad_interpolate(m, current_val, 0);       /* :448 */
    m->curr_reweight = ref_metad_interpolate(m, current_val, 1);             /* :451 */
    }

void ref_metad_update_bias(ref_metad *m, unsigned int timestep, const double *current_val, double *bias_out)
    {
    int dep = ref_metad_update_phase_a(m, timestep, current_val);
    ref_metad_update_phase_b(m, dep, current_val, bias_out);
    }

/* ------------------------------------------------------------------ dump / restart */

/* writeGrid, :831-926.  iostream setprecision(10) in default float format == printf %.10g */
int ref_metad_write_grid(ref_metad *m, const char *filename, unsigned int timestep, const char *const *names)
    {
    char path[4096];
    snprintf(path, sizeof(path), "%s_%u", filename, timestep);               /* :849 */
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    fprintf(f, "#n_cv: %u\n", m->n_cv);
    fprintf(f, "#dim: ");
    for (unsigned int i = 0; i < m->n_cv; i++) fprintf(f, " %u", m->num_points[i]);
    fprintf(f, "\n");
    fprintf(f, "#num_gaussians: %u\n", m->num_gaussians);
    for (unsigned int i = 0; i < m->n_cv; i++) fprintf(f, "%s\t", names[i]);
    fprintf(f, "grid_value\tdet_sigma\tnum_gaussians\thist\thist_reweight\tweight\n");

    unsigned int coords[REF_MAX_CV];
    for (unsigned int g = 0; g < m->len; g++)
        {
        ref_index_coords(m->n_cv, m->num_points, g, coords);
        for (unsigned int cv = 0; cv < m->n_cv; ++cv)
            {
            double delta = (m->cv_max[cv] - m->cv_min[cv]) / (m->num_points[cv] - 1);
            double val = m->cv_min[cv] + coords[cv] * delta;
            fprintf(f, "%.10g\t", val);
            }
        fprintf(f, "%.10g", m->grid[g]);
        double val;
        if (m->hist_gauss[g] > 0)
            val = m->sigma_grid[g] / (double)m->hist_gauss[g];                /* :909-914 */
        else
            val = 0.0;
        fprintf(f, "\t%.10g", val);
        fprintf(f, "\t%u", m->hist_gauss[g]);
        fprintf(f, "\t%u", m->hist[g]);
        fprintf(f, "\t%.10g", m->grid_reweighted[g]);
        fprintf(f, "\t%.10g", m->grid_weight[g]);
        fprintf(f, "\n");
        }
    fclose(f);
    return 0;
    }

/* readGrid, :928-1000 */
int ref_metad_read_grid(ref_metad *m, const char *filename)
    {
    FILE *f = fopen(filename, "r");
    if (!f) return 1;
    char *line = NULL;
    size_t cap = 0;
    int rc = 0;
    if (getline(&line, &cap, f) < 0) rc = 2;                                  /* skip two header lines */
    if (!rc && getline(&line, &cap, f) < 0) rc = 2;
    if (!rc && getline(&line, &cap, f) < 0) rc = 2;                           /* "#num_gaussians: n" */
    if (!rc)
        {
        char tmp[256];
        unsigned int ng = 0;
        if (sscanf(line, "%255s %u", tmp, &ng) == 2) m->num_gaussians = ng;
        }
    if (!rc && getline(&line, &cap, f) < 0) rc = 2;                           /* column names */
    for (unsigned int g = 0; !rc && g < m->len; g++)
        {
        if (getline(&line, &cap, f) < 0) { rc = 3; break; }                   /* premature end */
        char *p = line;
        char *end;
        for (unsigned int i = 0; i < m->n_cv; i++) { strtod(p, &end); p = end; }   /* skip CV values */
        m->grid[g] = strtod(p, &end); p = end;
        m->sigma_grid[g] = strtod(p, &end); p = end;
        m->hist_gauss[g] = (unsigned int)strtoul(p, &end, 10); p = end;
        m->hist[g] = (unsigned int)strtoul(p, &end, 10); p = end;
        m->sigma_grid[g] *= m->hist_gauss[g];                                 /* :992 */
        m->grid_reweighted[g] = strtod(p, &end); p = end;
        m->grid_weight[g] = strtod(p, &end); p = end;
        }
    free(line);
    fclose(f);
    return rc;
    }

/* ------------------------------------------------------------------ umbrella, box CVs */

/* CollectiveVariable::computeForces, CollectiveVariable.cc:22-60: returns the new bias factor */
double ref_umbrella_bias(int umbrella, double val, double bias_in, double cv0, double kappa,
                         double width_flat, double scale)
    {
    double bias = bias_in;
    if (umbrella != REF_NO_UMBRELLA)
        {
        if ((val < cv0 + width_flat / 2.0) && (val > cv0 - width_flat / 2.0))
            {
            /* leave bias as it is */
            }
        else
            {
            double delta = 0.0;
            if (val > cv0)
                delta = val - cv0 - width_flat / 2.0;
            else
                delta = val - cv0 + width_flat / 2.0;

            if (umbrella == REF_LINEAR)
                bias = bias + scale * 1.0;
            else if (umbrella == REF_HARMONIC)
                bias = bias + kappa * delta;
            else if (umbrella == REF_WALL)
                bias = bias + scale * 12.0 * pow(delta / kappa, 11.0) / kappa;
            else if (umbrella == REF_GAUSSIAN)
                bias = bias - scale * (val - cv0) * exp(-(val - cv0) * (val - cv0) / kappa / kappa / 2.0);
            }
        }
    return bias;
    }

/* CollectiveVariable::getUmbrellaPotential, CollectiveVariable.cc:68-106 */
double ref_umbrella_energy(int umbrella, double val, double cv0, double kappa, double width_flat, double scale)
    {
    if (umbrella != REF_NO_UMBRELLA)
        {
        if ((val < cv0 + width_flat / 2.0) && (val > cv0 - width_flat / 2.0))
            return 0.0;
        double delta = 0.0;
        if (val > cv0)
            delta = val - cv0 - width_flat / 2.0;
        else if (val < cv0)
            delta = val - cv0 + width_flat / 2.0;
        if (umbrella == REF_LINEAR) return scale * delta;
        if (umbrella == REF_HARMONIC) return (1.0 / 2.0) * delta * delta * kappa;
        if (umbrella == REF_WALL) return scale * pow(delta / kappa, 12.0);
        if (umbrella == REF_GAUSSIAN) return scale * exp(-(val - cv0) * (val - cv0) / kappa / kappa / 2.0) - scale;
        }
    return 0.0;
    }

/* AspectRatio::getCurrentValue, AspectRatio.cc:24-57 — including the `length1 = L.x` slip in the
 * dir2 switch (:46), so dir2 == 0 yields length2 == 0 */
double ref_aspect_ratio(const ref_box *box, unsigned int dir1, unsigned int dir2)
    {
    double length1 = 0.0, length2 = 0.0;
    switch (dir1)
        {
        case 0: length1 = box->L[0]; break;
        case 1: length1 = box->L[1]; break;
        case 2: length1 = box->L[2]; break;
        }
    switch (dir2)
        {
        case 0: length1 = box->L[0]; break;
        case 1: length2 = box->L[1]; break;
        case 2: length2 = box->L[2]; break;
        }
    return length1 / length2;
    }

/* Density::getCurrentValue, Density.cc:20-27 (3-d) */
double ref_density(const ref_box *box, unsigned int N_group)
    {
    double V = box->L[0] * box->L[1] * box->L[2];
    return (double)N_group / V;
    }

/* ------------------------------------------------------------------ WellTemperedEnsemble */

/* computeCV, WellTemperedEnsemble.cc:45-56 */
double ref_wte_potential_energy(unsigned int N, const double *net_force, double external_energy)
    {
    double pe = 0.0;
    for (unsigned int i = 0; i < N; ++i) pe += net_force[4 * i + 3];
    pe += external_energy;
    return pe;
    }

/* computeBiasForces, WellTemperedEnsemble.cc:135-188 (CPU: torque.w scaled too — SURVEY Q18) */
void ref_wte_scale(unsigned int N, double *net_force, double *net_torque, double *net_virial,
                   unsigned int pitch, double *external_virial, double bias)
    {
    double fac = 1.0 + bias;
    for (unsigned int i = 0; i < N; ++i)
        {
        net_force[4 * i + 0] *= fac;
        net_force[4 * i + 1] *= fac;
        net_force[4 * i + 2] *= fac;
        net_torque[4 * i + 0] *= fac;
        net_torque[4 * i + 1] *= fac;
        net_torque[4 * i + 2] *= fac;
        net_torque[4 * i + 3] *= fac;
        for (unsigned int r = 0; r < 6; r++) net_virial[i + r * pitch] *= fac;
        }
    for (unsigned int i = 0; i < 6; ++i) external_virial[i] = fac * external_virial[i];
    }

/* ------------------------------------------------------------------------------------------------
 * CollectiveWrapper (CollectiveWrapper.cc), CPU path
 * ---------------------------------------------------------------------------------------------- */

/* CollectiveWrapper.cc:31-72: energy = sum_j force_j.w + getExternalEnergy() of the wrapped compute */
double ref_wrapper_energy(unsigned int N, const double *force, double external_energy)
    {
    double e = 0.0;
    for (unsigned int i = 0; i < N; ++i) e += force[4 * i + 3];
    return e + external_energy;
    }

/* CollectiveWrapper.cc:136-179: fac = m_bias */
void ref_wrapper_scale(unsigned int N, double *force, double *torque, double *virial, unsigned int pitch, double bias)
    {
    double fac = bias;
    for (unsigned int i = 0; i < N; ++i)
        {
        force[4 * i + 0] *= fac;
        force[4 * i + 1] *= fac;
        force[4 * i + 2] *= fac;
        torque[4 * i + 0] *= fac;
        torque[4 * i + 1] *= fac;
        torque[4 * i + 2] *= fac;
        torque[4 * i + 3] *= fac;
        for (unsigned int r = 0; r < 6; r++) virial[i + r * pitch] *= fac;
        }
    }

/* ------------------------------------------------------------------------------------------------
 * Adaptive Gaussians: IntegratorMetaDynamics::computeSigma (IntegratorMetaDynamics.cc:1205-1294)
 * The reference inverts with Eigen (absent here); any exact dense inverse agrees to rounding -> Gauss-Jordan with
 * partial pivoting.  Parity at that boundary is unpinned (no reference test exercises adaptive Gaussians).
 * ---------------------------------------------------------------------------------------------- */
void ref_compute_sigma(unsigned int n_cv, unsigned int N, const double *const *forces, const int *can_derive,
                       const double *sigma, double sigma_g, double *sigmasq, double *sigma_inv)
    {
    for (unsigned int i = 0; i < n_cv; ++i)
        for (unsigned int j = 0; j < n_cv; ++j)
            {
            sigmasq[i * n_cv + j] = 0.0;                                               /* :1233 */
            if (can_derive[i] && can_derive[j])                                        /* :1234 */
                {
                for (unsigned int n = 0; n < N; ++n)                                   /* :1239-1247 */
                    {
                    const double *fi = forces[i] + 4 * n, *fj = forces[j] + 4 * n;
                    sigmasq[i * n_cv + j] += sigma_g * sigma_g * (fi[0] * fj[0] + fi[1] * fj[1] + fi[2] * fj[2]);
                    }
                }
            else if (i == j)
                sigmasq[i * n_cv + j] = sigma[i] * sigma[i];                           /* :1249 */
            }

    /* m(i,j) = sqrt(sigmasq) element-wise (:1277-1279), inverse (:1281) */
    double a[REF_MAX_CV][2 * REF_MAX_CV];
    for (unsigned int i = 0; i < n_cv; ++i)
        for (unsigned int j = 0; j < n_cv; ++j)
            {
            a[i][j] = sqrt(sigmasq[i * n_cv + j]);
            a[i][n_cv + j] = (i == j) ? 1.0 : 0.0;
            }
    for (unsigned int k = 0; k < n_cv; ++k)
        {
        unsigned int piv = k;
        for (unsigned int r = k + 1; r < n_cv; ++r)
            if (fabs(a[r][k]) > fabs(a[piv][k])) piv = r;
        if (piv != k)
            for (unsigned int j = 0; j < 2 * n_cv; ++j)
                {
                double t = a[k][j];
                a[k][j] = a[piv][j];
                a[piv][j] = t;
                }
        double d = a[k][k];
        for (unsigned int j = 0; j < 2 * n_cv; ++j) a[k][j] /= d;
        for (unsigned int r = 0; r < n_cv; ++r)
            if (r != k)
                {
                double f = a[r][k];
                for (unsigned int j = 0; j < 2 * n_cv; ++j) a[r][j] -= f * a[k][j];
                }
        }
    for (unsigned int i = 0; i < n_cv; ++i)
        for (unsigned int j = 0; j < n_cv; ++j) sigma_inv[i * n_cv + j] = a[i][n_cv + j];
    }
