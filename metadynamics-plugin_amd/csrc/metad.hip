// metad.hip — device-resident bias grid of IntegratorMetaDynamics on gfx950.
//
// Reference (the CPU path results must match): IntegratorMetaDynamics.cc:314-588 (updateBiasPotential),
// :663-736 (interpolateGrid), :738-776 (biasPotentialDerivative), :1002-1047 (updateGrid),
// :1053-1090 (updateReweightedEstimator), :1092-1155 (updateHistogram / updateSigmaGrid),
// IndexGrid.cc:20-58.  The reference GPU build only runs the Gaussian deposit on the device
// (IntegratorMetaDynamics.cu:6-93) and does every other pass on the host, with D2H/H2D copies of the
// grid arrays around it; here the ten grid arrays never leave HBM/L2 (G = 65 536 cells -> 2.9 MB of
// traffic per deposit, L2 resident) and the CV values / bias factors are handed from and to the
// particle kernels through device memory, so a step needs no host synchronisation at all.
//
// Kernels per step (all double precision, fixed reduction orders => bitwise reproducible):
//   k_prepare (1 block)   CV values from their partial sums; histogram bin; on deposit steps the
//                         sigma grid and the well-tempered scale exp(-V(s)/dT)
//   k_reweight1 (G/256)   [deposit] Gaussian increment per cell, R += hist_delta, block sums of R*dV, R
//   k_apply (G/256)       [deposit] <dV>, fac = exp(-(dV-<dV>)/T), R *= fac, w /= fac, grids += deltas
//   k_evaluate (1 block)  dV/ds_c by finite differences of the multilinear interpolant, V(s), w(s)
#include "mtd_device.hpp"

#include <cmath>
#include <cstring>
#include <new>

namespace
{

using namespace mtd;

constexpr int MAXCV = MTD_METAD_MAX_CV;
constexpr int GRID_THREADS = 256;
constexpr int MAX_POINTS = 2 * MAXCV + 2;
constexpr int MAX_TERMS = 1 << MAXCV;

struct CvSource
    {
    const double *partials;
    unsigned int n_partials, stride, offset, _pad;
    double scale, shift;
    };

struct MetadState
    {
    double cv[MAXCV];
    double bias[MAXCV];
    double V;        // log quantity "bias"   (IntegratorMetaDynamics.cc:448)
    double w;        // log quantity "weight" (:451)
    double scal;     // well-tempered scale of the current deposit (:374-379)
    double avg_dV;   // <dV> of the last reweighting step (:1077)
    unsigned int num_gaussians;
    unsigned int n_oob;
    unsigned int bin;
    unsigned int on_grid;
    };

struct MetadCfg
    {
    unsigned int n_cv, len;
    unsigned int lengths[MAXCV];
    unsigned int factors[MAXCV];
    double cv_min[MAXCV], cv_max[MAXCV], delta[MAXCV];
    double sigma_inv[MAXCV * MAXCV];
    double W, T_shift, temp, det_sigma;
    int mode, _pad;
    double *grid, *grid_delta, *rew, *weight, *sigma_grid, *sigma_grid_delta;
    unsigned int *hist, *hist_delta, *hist_gauss, *hist_gauss_delta;
    MetadState *st;
    double *gpart;
    unsigned int n_gblocks, _pad2;
    CvSource src[MAXCV];
    };

// IndexGrid::getCoordinates (IndexGrid.cc:46-58)
__device__ __forceinline__ void decode(const MetadCfg &c, unsigned int idx, unsigned int *coords)
    {
    unsigned int rest = idx;
    for (int i = (int)c.n_cv - 1; i >= 0; --i)
        {
        coords[i] = rest / c.factors[i];
        rest -= coords[i] * c.factors[i];
        }
    }

// floor-bin shared by updateHistogram (:1092-1119) and updateSigmaGrid (:1122-1155).  The reference
// converts (s-min)/delta to unsigned: undefined for values <= -1 or >= 2^32, treated as off-grid.
__device__ bool bin_of(const MetadCfg &c, const double *val, unsigned int &idx)
    {
    bool on_grid = true;
    unsigned int r = 0;
    for (unsigned int i = 0; i < c.n_cv; ++i)
        {
        const double q = (val[i] - c.cv_min[i]) / c.delta[i];
        if (!(q > -1.0) || !(q < 4294967296.0))
            {
            on_grid = false;
            continue;
            }
        const unsigned int coord = (unsigned int)q;
        if (coord >= c.lengths[i]) on_grid = false;
        r += coord * c.factors[i];
        }
    idx = r;
    return on_grid;
    }

// One term of interpolateGrid's multilinear sum (:711-733) for evaluation point val[], corner `bits`.
// Returns false when the point is out of bounds (:677-683 => whole interpolation is 0).
__device__ bool interp_term(const MetadCfg &c, const double *val, unsigned int bits, const double *arr, double &term)
    {
    double t = 1.0;
    unsigned int idx = 0;
    for (unsigned int i = 0; i < c.n_cv; ++i)
        {
        const double v = val[i];
        if (v < c.cv_min[i] || v >= c.cv_max[i]) return false;
        int lower = (int)((v - c.cv_min[i]) / c.delta[i]);
        int upper = lower + 1;
        if (upper >= (int)c.lengths[i])
            {
            lower--;
            upper--;
            }
        const double lower_bound = c.cv_min[i] + c.delta[i] * lower;
        const double upper_bound = c.cv_min[i] + c.delta[i] * upper;
        const double rel = (v - lower_bound) / (upper_bound - lower_bound);
        if (bits & (1u << i))
            {
            idx += (unsigned int)lower * c.factors[i];
            t *= (1.0 - rel);
            }
        else
            {
            idx += (unsigned int)upper * c.factors[i];
            t *= rel;
            }
        }
    term = t * arr[idx];
    return true;
    }

// Evaluate n_points interpolations cooperatively: every (point, corner) pair is one lane's work (one
// L2 read each, all in flight together); the corner terms are then summed in the reference's order.
// pts[p][i] evaluation points, which[p] != 0 -> weight grid.  Results in s_res[p]; s_oob[p] flags.
__device__ void interpolate_points(const MetadCfg &c, const double (*pts)[MAXCV], const int *which, int n_points,
                                   double (*s_terms)[MAX_TERMS], int *s_oob, double *s_res)
    {
    const int n_term = 1 << c.n_cv;
    for (int p = threadIdx.x; p < n_points; p += blockDim.x) s_oob[p] = 0;
    __syncthreads();
    for (int idx = threadIdx.x; idx < n_points * n_term; idx += blockDim.x)
        {
        const int p = idx / n_term;
        const unsigned int bits = idx % n_term;
        double term = 0.0;
        const bool ok = interp_term(c, pts[p], bits, which[p] ? c.weight : c.grid, term);
        s_terms[p][bits] = ok ? term : 0.0;
        if (!ok && bits == 0) s_oob[p] = 1;
        }
    __syncthreads();
    for (int p = threadIdx.x; p < n_points; p += blockDim.x)
        {
        double res = 0.0;
        for (int b = 0; b < n_term; ++b) res += s_terms[p][b];
        s_res[p] = res;
        }
    __syncthreads();
    }

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GRID_THREADS) void k_prepare(const MetadCfg c, const int deposit)
    {
    __shared__ double s_cv[MAXCV];
    __shared__ double s_pts[1][MAXCV];
    __shared__ int s_which[1];
    __shared__ double s_terms[1][MAX_TERMS];
    __shared__ int s_oob[1];
    __shared__ double s_res[1];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // CV values from their block partial sums (replaces getCurrentValue's D2H + host sum, .cc:323-327)
    for (unsigned int i = wave; i < c.n_cv; i += GRID_THREADS / MTD_WAVE)
        {
        const CvSource s = c.src[i];
        double v = 0.0;
        if (s.partials)
            for (unsigned int b = lane; b < s.n_partials; b += MTD_WAVE) v += s.partials[(size_t)b * s.stride + s.offset];
        v = wave_sum(v);
        if (lane == 0)
            {
            const double val = s.partials ? s.shift + s.scale * v : s.shift;   // host-provided value
            s_cv[i] = val;
            c.st->cv[i] = val;
            }
        }
    __syncthreads();

    if (threadIdx.x == 0)
        {
        unsigned int bin = 0;
        const bool on_grid = bin_of(c, s_cv, bin);
        c.st->bin = bin;
        c.st->on_grid = on_grid ? 1u : 0u;
        if (on_grid)
            {
            c.hist_delta[bin] += 1;                  // updateHistogram, every step (:366)
            if (deposit)
                {
                c.sigma_grid_delta[bin] += c.det_sigma;  // updateSigmaGrid (:371)
                c.hist_gauss_delta[bin] += 1;
                }
            }
        }

    if (deposit)
        {
        double scal = 1.0;
        if (c.mode == MTD_MODE_WELL_TEMPERED)
            {
            if (threadIdx.x < c.n_cv) s_pts[0][threadIdx.x] = s_cv[threadIdx.x];
            if (threadIdx.x == 0) s_which[0] = 0;
            __syncthreads();
            interpolate_points(c, s_pts, s_which, 1, s_terms, s_oob, s_res);
            scal = exp(-s_res[0] / c.T_shift);       // :377-378
            if (threadIdx.x == 0 && s_oob[0]) c.st->n_oob += 1;
            }
        if (threadIdx.x == 0) c.st->scal = scal;
        }
    }

// ---------------------------------------------------------------------------------------------
// updateGrid (:1002-1047) fused with the first pass of updateReweightedEstimator (:1070-1075)
template<bool COMPUTE_DELTA, bool REWEIGHT>
__global__ __launch_bounds__(GRID_THREADS) void k_reweight1(const MetadCfg c)
    {
    __shared__ double s_red[16];
    const unsigned int g = blockIdx.x * GRID_THREADS + threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    if (g < c.len)
        {
        double dV;
        if (COMPUTE_DELTA)
            {
            unsigned int coords[MAXCV];
            decode(c, g, coords);
            double d[MAXCV];
            for (unsigned int i = 0; i < c.n_cv; ++i)
                {
                const double val_i = c.cv_min[i] + coords[i] * c.delta[i];
                d[i] = val_i - c.st->cv[i];
                }
            double gauss_exp = 0.0;
            for (unsigned int i = 0; i < c.n_cv; ++i)
                for (unsigned int j = 0; j < c.n_cv; ++j)
                    {
                    const double sij = c.sigma_inv[i * c.n_cv + j];
                    gauss_exp += d[i] * d[j] * (1.0 / 2.0) * (sij * sij);   // element-wise square: Q12
                    }
            dV = c.W * c.st->scal * exp(-gauss_exp);
            c.grid_delta[g] = dV;                                            // CPU semantics: overwrite (:1043)
            }
        else
            dV = c.grid_delta[g];
        if (REWEIGHT)
            {
            const double R = c.rew[g] + (double)c.hist_delta[g];             // :1072
            c.rew[g] = R;
            s1 = R * dV;
            s2 = R;
            }
        }
    if (REWEIGHT)
        {
        s1 = block_sum(s1, s_red);
        s2 = block_sum(s2, s_red);
        if (threadIdx.x == 0)
            {
            c.gpart[2 * blockIdx.x] = s1;
            c.gpart[2 * blockIdx.x + 1] = s2;
            }
        }
    }

// second pass of updateReweightedEstimator (:1077-1087) fused with accumulate + clear (:426-437)
__global__ __launch_bounds__(GRID_THREADS) void k_apply(const MetadCfg c)
    {
    __shared__ double s_red[16];
    double s1 = 0.0, s2 = 0.0;
    for (unsigned int b = threadIdx.x; b < c.n_gblocks; b += GRID_THREADS)
        {
        s1 += c.gpart[2 * b];
        s2 += c.gpart[2 * b + 1];
        }
    s1 = block_sum(s1, s_red);
    s2 = block_sum(s2, s_red);
    const double avg_dV = s1 / s2;                                           // norm == 0 -> NaN like the reference (Q15)

    const unsigned int g = blockIdx.x * GRID_THREADS + threadIdx.x;
    if (g < c.len)
        {
        const double dV = c.grid_delta[g];
        const double fac = exp(-(dV - avg_dV) / c.temp);                     // T, not deltaT (:1084)
        c.rew[g] *= fac;
        c.weight[g] /= fac;
        c.grid[g] += dV;
        c.sigma_grid[g] += c.sigma_grid_delta[g];
        c.hist[g] += c.hist_delta[g];
        c.hist_gauss[g] += c.hist_gauss_delta[g];
        c.grid_delta[g] = 0.0;
        c.sigma_grid_delta[g] = 0.0;
        c.hist_delta[g] = 0;
        c.hist_gauss_delta[g] = 0;
        }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        {
        c.st->avg_dV = avg_dV;
        c.st->num_gaussians += 1;                                            // :440
        }
    }

// ---------------------------------------------------------------------------------------------
// biasPotentialDerivative for every CV (:444-445 -> :738-776), V(s) (:448) and w(s) (:451)
__global__ __launch_bounds__(GRID_THREADS) void k_evaluate(const MetadCfg c)
    {
    __shared__ double s_pts[MAX_POINTS][MAXCV];
    __shared__ int s_which[MAX_POINTS];
    __shared__ double s_terms[MAX_POINTS][MAX_TERMS];
    __shared__ int s_oob[MAX_POINTS];
    __shared__ double s_res[MAX_POINTS];

    const unsigned int n = c.n_cv;
    const int n_points = 2 + 2 * (int)n;
    // point 0: s on the bias grid; point 1: s on the weight grid; 2+2i: s - delta_i e_i; 3+2i: s + delta_i e_i
    for (unsigned int idx = threadIdx.x; idx < (unsigned int)n_points * n; idx += blockDim.x)
        {
        const unsigned int p = idx / n, i = idx % n;
        double v = c.st->cv[i];
        if (p >= 2 && (p - 2) / 2 == i) v = ((p - 2) & 1) ? v + c.delta[i] : v - c.delta[i];
        s_pts[p][i] = v;
        }
    if (threadIdx.x < (unsigned int)n_points) s_which[threadIdx.x] = (threadIdx.x == 1) ? 1 : 0;
    __syncthreads();

    interpolate_points(c, s_pts, s_which, n_points, s_terms, s_oob, s_res);

    if (threadIdx.x < n)
        {
        const unsigned int i = threadIdx.x;
        const double s = s_pts[0][i];
        const double delta = c.delta[i];
        double b;
        if (s - delta < c.cv_min[i])
            b = (s_res[3 + 2 * i] - s_res[0]) / delta;                        // forward  (:746-755)
        else if (s + delta > c.cv_max[i])
            b = (s_res[0] - s_res[2 + 2 * i]) / delta;                        // backward (:756-764)
        else
            b = (s_res[3 + 2 * i] - s_res[2 + 2 * i]) / (2.0 * delta);        // central  (:765-775)
        c.st->bias[i] = b;
        }
    if (threadIdx.x == 0)
        {
        c.st->V = s_res[0];
        c.st->w = s_res[1];
        if (s_oob[0]) c.st->n_oob += 1;
        }
    }

// drop-in gpu_update_grid (IntegratorMetaDynamics.cu:6-93): grid_delta += W * scal * gauss
struct UpdateGridArgs
    {
    unsigned int dim, len;
    unsigned int lengths[MAXCV], factors[MAXCV];
    double cv_min[MAXCV], delta[MAXCV], sigma_inv[MAXCV * MAXCV];
    double scal, W;
    };

__global__ __launch_bounds__(GRID_THREADS) void k_update_grid(const UpdateGridArgs a, const double *__restrict__ current_val,
                                                              double *__restrict__ grid_delta)
    {
    const unsigned int g = blockIdx.x * GRID_THREADS + threadIdx.x;
    if (g >= a.len) return;
    unsigned int rest = g;
    double d[MAXCV];
    for (int i = (int)a.dim - 1; i >= 0; --i)
        {
        const unsigned int coord = rest / a.factors[i];
        rest -= coord * a.factors[i];
        d[i] = a.cv_min[i] + coord * a.delta[i] - current_val[i];
        }
    double gauss_exp = 0.0;
    for (unsigned int i = 0; i < a.dim; ++i)
        for (unsigned int j = 0; j < a.dim; ++j)
            {
            const double sij = a.sigma_inv[i * a.dim + j];
            gauss_exp += d[i] * d[j] * (1.0 / 2.0) * (sij * sij);
            }
    grid_delta[g] += a.W * a.scal * exp(-gauss_exp);
    }

__global__ void k_reset_hist(unsigned int *hist, unsigned int *hist_delta, unsigned int len)
    {
    const unsigned int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < len)
        {
        hist[g] = 0;
        hist_delta[g] = 0;
        }
    }

__global__ void k_fill(double *p, double v, unsigned int len)
    {
    const unsigned int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < len) p[g] = v;
    }

double host_determinant(const double *m, unsigned int n)
    {
    double a[MAXCV * MAXCV];
    std::memcpy(a, m, sizeof(double) * n * n);
    if (n == 1) return a[0];
    if (n == 2) return a[0] * a[3] - a[1] * a[2];
    double det = 1.0;
    for (unsigned int c = 0; c < n; c++)
        {
        unsigned int p = c;
        for (unsigned int r = c + 1; r < n; r++)
            if (std::fabs(a[r * n + c]) > std::fabs(a[p * n + c])) p = r;
        if (a[p * n + c] == 0.0) return 0.0;
        if (p != c)
            {
            for (unsigned int k = 0; k < n; k++) std::swap(a[c * n + k], a[p * n + k]);
            det = -det;
            }
        det *= a[c * n + c];
        for (unsigned int r = c + 1; r < n; r++)
            {
            const double f = a[r * n + c] / a[c * n + c];
            for (unsigned int k = c; k < n; k++) a[r * n + k] -= f * a[c * n + k];
            }
        }
    return det;
    }

} // namespace

struct mtd_metad
    {
    MetadCfg cfg;
    unsigned int stride;
    int add_bias;
    void *slab;
    };

extern "C" {

int mtd_update_grid(unsigned int num_elements, const unsigned int *lengths, unsigned int dim,
                    const double *d_current_val, double *d_grid_delta, const double *cv_min,
                    const double *cv_max, const double *sigma_inv, double scal, double W,
                    mtd_stream_t stream)
    {
    if (!lengths || !d_current_val || !d_grid_delta || !cv_min || !cv_max || !sigma_inv) return MTD_ERR_INVALID_ARGUMENT;
    if (dim == 0 || dim > (unsigned int)MAXCV) return MTD_ERR_UNSUPPORTED;
    UpdateGridArgs a;
    std::memset(&a, 0, sizeof(a));
    a.dim = dim;
    unsigned int len = 1;
    for (unsigned int i = 0; i < dim; ++i)
        {
        if (lengths[i] < 2) return MTD_ERR_INVALID_ARGUMENT;
        a.lengths[i] = lengths[i];
        a.factors[i] = (i == 0) ? 1 : a.lengths[i - 1] * a.factors[i - 1];
        a.cv_min[i] = cv_min[i];
        a.delta[i] = (cv_max[i] - cv_min[i]) / (double)(lengths[i] - 1);
        len *= lengths[i];
        }
    if (len != num_elements) return MTD_ERR_INVALID_ARGUMENT;
    for (unsigned int i = 0; i < dim * dim; ++i) a.sigma_inv[i] = sigma_inv[i];
    a.len = len;
    a.scal = scal;
    a.W = W;
    k_update_grid<<<(len + GRID_THREADS - 1) / GRID_THREADS, GRID_THREADS, 0, (hipStream_t)stream>>>(a, d_current_val, d_grid_delta);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

int mtd_metad_create(mtd_metad **out, unsigned int n_cv, const double *sigma, const double *cv_min,
                     const double *cv_max, const unsigned int *num_points, double W, double T_shift,
                     double T, unsigned int stride, int mode, int add_bias)
    {
    if (!out || !sigma || !cv_min || !cv_max || !num_points) return MTD_ERR_INVALID_ARGUMENT;
    if (n_cv == 0) return MTD_ERR_INVALID_ARGUMENT;
    if (n_cv > (unsigned int)MAXCV) return MTD_ERR_UNSUPPORTED;
    if (stride == 0 || !(W > 0.0) || !(T_shift > 0.0)) return MTD_ERR_INVALID_ARGUMENT; // asserts at .cc:58-59
    if (mode != MTD_MODE_STANDARD && mode != MTD_MODE_WELL_TEMPERED) return MTD_ERR_INVALID_ARGUMENT;

    mtd_metad *m = new (std::nothrow) mtd_metad();
    if (!m) return (int)hipErrorOutOfMemory;
    std::memset(&m->cfg, 0, sizeof(m->cfg));
    MetadCfg &c = m->cfg;
    c.n_cv = n_cv;
    unsigned long long len = 1;
    for (unsigned int i = 0; i < n_cv; ++i)
        {
        // setGrid(true) input checks (.cc:798-812)
        if (!(cv_min[i] < cv_max[i]) || num_points[i] < 2 || !(sigma[i] > 0.0))
            {
            delete m;
            return MTD_ERR_INVALID_ARGUMENT;
            }
        c.lengths[i] = num_points[i];
        c.factors[i] = (i == 0) ? 1 : c.lengths[i - 1] * c.factors[i - 1];
        c.cv_min[i] = cv_min[i];
        c.cv_max[i] = cv_max[i];
        c.delta[i] = (cv_max[i] - cv_min[i]) / (double)(num_points[i] - 1);
        c.sigma_inv[i * n_cv + i] = 1.0 / sigma[i];                      // prepRun .cc:177
        len *= num_points[i];
        if (len > 0x7fffffffULL)
            {
            delete m;
            return MTD_ERR_UNSUPPORTED;
            }
        }
    c.len = (unsigned int)len;
    c.W = W;
    c.T_shift = T_shift;
    c.temp = T;
    c.mode = mode;
    c.det_sigma = host_determinant(c.sigma_inv, n_cv);
    c.n_gblocks = (c.len + GRID_THREADS - 1) / GRID_THREADS;
    m->stride = stride;
    m->add_bias = add_bias ? 1 : 0;

    const size_t G = c.len;
    const size_t bytes_d = 6 * G * sizeof(double);
    const size_t bytes_u = 4 * G * sizeof(unsigned int);
    const size_t bytes_state = (sizeof(MetadState) + 255) / 256 * 256;
    const size_t bytes_gpart = 2 * (size_t)c.n_gblocks * sizeof(double);
    const size_t total = bytes_d + bytes_u + bytes_state + bytes_gpart;
    hipError_t e = hipMalloc(&m->slab, total);
    if (e != hipSuccess)
        {
        delete m;
        return (int)e;
        }
    e = hipMemset(m->slab, 0, total); // GPUArray storage is zero-initialised (setupGrid .cc:608-652)
    if (e != hipSuccess)
        {
        (void)hipFree(m->slab);
        delete m;
        return (int)e;
        }
    char *p = (char *)m->slab;
    c.grid = (double *)p;
    c.rew = c.grid + G;
    c.weight = c.rew + G;
    c.sigma_grid = c.weight + G;
    c.grid_delta = c.sigma_grid + G;      // {grid_delta, sigma_grid_delta} contiguous: one all-reduce
    c.sigma_grid_delta = c.grid_delta + G;
    p += bytes_d;
    c.hist = (unsigned int *)p;
    c.hist_gauss = c.hist + G;
    c.hist_delta = c.hist_gauss + G;      // {hist_delta, hist_gauss_delta} contiguous
    c.hist_gauss_delta = c.hist_delta + G;
    p += bytes_u;
    c.st = (MetadState *)p;
    p += bytes_state;
    c.gpart = (double *)p;

    k_fill<<<(c.len + 255) / 256, 256>>>(c.weight, 1.0, c.len);           // weight grid reset to one (.cc:657-658)
    MetadState st;
    std::memset(&st, 0, sizeof(st));
    st.w = 1.0;                                                          // m_curr_reweight(1.0) .cc:56
    st.scal = 1.0;
    e = hipMemcpy(c.st, &st, sizeof(st), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess)
        {
        (void)hipFree(m->slab);
        delete m;
        return (int)e;
        }
    // default CV source: a host-provided value (mtd_metad_set_cv_value), initially 0
    *out = m;
    return MTD_SUCCESS;
    }

int mtd_metad_destroy(mtd_metad *m)
    {
    if (!m) return MTD_SUCCESS;
    hipError_t e = hipFree(m->slab);
    delete m;
    return (int)e;
    }

int mtd_metad_set_stride(mtd_metad *m, unsigned int stride)
    {
    if (!m || stride == 0) return MTD_ERR_INVALID_ARGUMENT;
    m->stride = stride;
    return MTD_SUCCESS;
    }

int mtd_metad_set_add_hills(mtd_metad *m, int add_bias)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    m->add_bias = add_bias ? 1 : 0;
    return MTD_SUCCESS;
    }

int mtd_metad_set_mode(mtd_metad *m, int mode)
    {
    if (!m || (mode != MTD_MODE_STANDARD && mode != MTD_MODE_WELL_TEMPERED)) return MTD_ERR_INVALID_ARGUMENT;
    m->cfg.mode = mode;
    return MTD_SUCCESS;
    }

int mtd_metad_set_sigma_inv(mtd_metad *m, const double *sigma_inv)
    {
    if (!m || !sigma_inv) return MTD_ERR_INVALID_ARGUMENT;
    std::memcpy(m->cfg.sigma_inv, sigma_inv, sizeof(double) * m->cfg.n_cv * m->cfg.n_cv);
    m->cfg.det_sigma = host_determinant(m->cfg.sigma_inv, m->cfg.n_cv);
    return MTD_SUCCESS;
    }

int mtd_metad_reset_histogram(mtd_metad *m, mtd_stream_t stream)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    k_reset_hist<<<(m->cfg.len + 255) / 256, 256, 0, (hipStream_t)stream>>>(m->cfg.hist, m->cfg.hist_delta, m->cfg.len);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

int mtd_metad_set_cv_source(mtd_metad *m, unsigned int cv, const double *d_partials, unsigned int n_partials,
                            unsigned int stride, unsigned int offset, double scale, double shift)
    {
    if (!m || cv >= m->cfg.n_cv) return MTD_ERR_INVALID_ARGUMENT;
    CvSource &s = m->cfg.src[cv];
    if (!d_partials || n_partials == 0 || stride == 0) return MTD_ERR_INVALID_ARGUMENT;
    s.partials = d_partials;
    s.n_partials = n_partials;
    s.stride = stride;
    s.offset = offset;
    s.scale = scale;
    s.shift = shift;
    return MTD_SUCCESS;
    }

int mtd_metad_set_cv_value(mtd_metad *m, unsigned int cv, double value)
    {
    if (!m || cv >= m->cfg.n_cv) return MTD_ERR_INVALID_ARGUMENT;
    CvSource &s = m->cfg.src[cv];
    std::memset(&s, 0, sizeof(s));
    s.shift = value;
    return MTD_SUCCESS;
    }

const double *mtd_metad_bias_device(const mtd_metad *m) { return m ? m->cfg.st->bias : nullptr; }
const double *mtd_metad_cv_device(const mtd_metad *m) { return m ? m->cfg.st->cv : nullptr; }
unsigned int mtd_metad_num_elements(const mtd_metad *m) { return m ? m->cfg.len : 0; }
double mtd_metad_sigma_determinant(const mtd_metad *m) { return m ? m->cfg.det_sigma : 0.0; }

int mtd_metad_update_phase_a(mtd_metad *m, unsigned int timestep, int *deposited, mtd_stream_t stream)
    {
    if (!m || !deposited) return MTD_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const int dep = (m->add_bias && (timestep % m->stride == 0)) ? 1 : 0;   // .cc:368
    k_prepare<<<1, GRID_THREADS, 0, s>>>(m->cfg, dep);
    MTD_LAUNCH_CHECK();
    if (dep)
        {
        k_reweight1<true, false><<<m->cfg.n_gblocks, GRID_THREADS, 0, s>>>(m->cfg);
        MTD_LAUNCH_CHECK();
        }
    *deposited = dep;
    return MTD_SUCCESS;
    }

int mtd_metad_update_phase_b(mtd_metad *m, int deposited, mtd_stream_t stream)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    if (deposited)
        {
        k_reweight1<false, true><<<m->cfg.n_gblocks, GRID_THREADS, 0, s>>>(m->cfg);
        MTD_LAUNCH_CHECK();
        k_apply<<<m->cfg.n_gblocks, GRID_THREADS, 0, s>>>(m->cfg);
        MTD_LAUNCH_CHECK();
        }
    k_evaluate<<<1, GRID_THREADS, 0, s>>>(m->cfg);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

int mtd_metad_update_bias(mtd_metad *m, unsigned int timestep, mtd_stream_t stream)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const int dep = (m->add_bias && (timestep % m->stride == 0)) ? 1 : 0;
    k_prepare<<<1, GRID_THREADS, 0, s>>>(m->cfg, dep);
    MTD_LAUNCH_CHECK();
    if (dep)
        {
        k_reweight1<true, true><<<m->cfg.n_gblocks, GRID_THREADS, 0, s>>>(m->cfg);
        MTD_LAUNCH_CHECK();
        k_apply<<<m->cfg.n_gblocks, GRID_THREADS, 0, s>>>(m->cfg);
        MTD_LAUNCH_CHECK();
        }
    k_evaluate<<<1, GRID_THREADS, 0, s>>>(m->cfg);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

int mtd_metad_delta_buffers(mtd_metad *m, double **d_real, unsigned int **d_count, unsigned int *num_elements)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    if (d_real) *d_real = m->cfg.grid_delta;
    if (d_count) *d_count = m->cfg.hist_delta;
    if (num_elements) *num_elements = m->cfg.len;
    return MTD_SUCCESS;
    }

int mtd_metad_get_state(mtd_metad *m, double *cv, double *bias, double *bias_potential, double *weight,
                        unsigned int *num_gaussians, unsigned int *num_out_of_bounds, mtd_stream_t stream)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    MetadState st;
    MTD_HIP_TRY(hipMemcpyAsync(&st, m->cfg.st, sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream));
    MTD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (unsigned int i = 0; i < m->cfg.n_cv; ++i)
        {
        if (cv) cv[i] = st.cv[i];
        if (bias) bias[i] = st.bias[i];
        }
    if (bias_potential) *bias_potential = st.V;
    if (weight) *weight = st.w;
    if (num_gaussians) *num_gaussians = st.num_gaussians;
    if (num_out_of_bounds) *num_out_of_bounds = st.n_oob;
    return MTD_SUCCESS;
    }

static void *array_ptr(mtd_metad *m, int which, size_t *elem)
    {
    MetadCfg &c = m->cfg;
    *elem = which < 6 ? sizeof(double) : sizeof(unsigned int);
    switch (which)
        {
        case 0: return c.grid;
        case 1: return c.grid_delta;
        case 2: return c.rew;
        case 3: return c.weight;
        case 4: return c.sigma_grid;
        case 5: return c.sigma_grid_delta;
        case 6: return c.hist;
        case 7: return c.hist_delta;
        case 8: return c.hist_gauss;
        case 9: return c.hist_gauss_delta;
        }
    return nullptr;
    }

void *mtd_metad_device_array(mtd_metad *m, int which)
    {
    size_t e;
    return m ? array_ptr(m, which, &e) : nullptr;
    }

int mtd_metad_get_array(mtd_metad *m, int which, void *host_out, mtd_stream_t stream)
    {
    if (!m || !host_out) return MTD_ERR_INVALID_ARGUMENT;
    size_t e;
    void *p = array_ptr(m, which, &e);
    if (!p) return MTD_ERR_INVALID_ARGUMENT;
    MTD_HIP_TRY(hipMemcpyAsync(host_out, p, e * m->cfg.len, hipMemcpyDeviceToHost, (hipStream_t)stream));
    MTD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MTD_SUCCESS;
    }

int mtd_metad_set_array(mtd_metad *m, int which, const void *host_in, mtd_stream_t stream)
    {
    if (!m || !host_in) return MTD_ERR_INVALID_ARGUMENT;
    size_t e;
    void *p = array_ptr(m, which, &e);
    if (!p) return MTD_ERR_INVALID_ARGUMENT;
    MTD_HIP_TRY(hipMemcpyAsync(p, host_in, e * m->cfg.len, hipMemcpyHostToDevice, (hipStream_t)stream));
    MTD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MTD_SUCCESS;
    }

int mtd_metad_set_num_gaussians(mtd_metad *m, unsigned int n, mtd_stream_t stream)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    MTD_HIP_TRY(hipMemcpyAsync(&m->cfg.st->num_gaussians, &n, sizeof(n), hipMemcpyHostToDevice, (hipStream_t)stream));
    MTD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MTD_SUCCESS;
    }

} // extern "C"
