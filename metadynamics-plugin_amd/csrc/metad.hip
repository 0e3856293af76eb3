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
#include "metad_device.hpp"

#include <cmath>
#include <cstring>
#include <mutex>
#include <new>

#include "metad_host.hpp"
#include "comm_host.hpp"

#include <cstdlib>

namespace
{

using namespace mtd;

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GRID_THREADS) void k_prepare(const MetadCfg c, const int deposit)
    {
    __shared__ EvalShared sh;
    __shared__ double s_tmp[16];
    reduce_cv_sources(c, sh.cv, s_tmp);     // replaces getCurrentValue's D2H + host sum (.cc:323-327)
    __syncthreads();
    evaluate_bias(c, sh, deposit != 0, false);   // V_old(s) -> well-tempered scale (:374-379); bin
    if (threadIdx.x < c.n_cv) c.st->cv[threadIdx.x] = sh.cv[threadIdx.x];
    if (threadIdx.x == 0)
        {
        c.st->failed = 0;                                   // a step of its own (a poisoned one is flagged by the fused kernels only)
        c.st->bin = sh.bin;
        c.st->on_grid = (unsigned int)sh.on_grid;
        if (sh.on_grid)
            {
            c.hist_delta[sh.bin] += 1;                      // updateHistogram, every step (:366)
            if (deposit)
                {
                c.sigma_grid_delta[sh.bin] += c.det_sigma;  // updateSigmaGrid (:371)
                c.hist_gauss_delta[sh.bin] += 1;
                }
            }
        if (deposit)
            {
            c.st->scal = sh.scal;
            if (c.mode == MTD_MODE_WELL_TEMPERED && sh.oob[0]) c.st->n_oob += 1;
            }
        }
    }

// ---------------------------------------------------------------------------------------------
// updateGrid (:1002-1047) fused with the first pass of updateReweightedEstimator (:1070-1075)
template<bool COMPUTE_DELTA, bool REWEIGHT>
__global__ __launch_bounds__(GRID_THREADS) void k_reweight1(const MetadCfg c)
    {
    __shared__ double s_red[16];
    __shared__ double s_cv[MAXCV];
    if (COMPUTE_DELTA)
        {
        if (threadIdx.x < c.n_cv) s_cv[threadIdx.x] = c.st->cv[threadIdx.x];
        __syncthreads();
        }
    const unsigned int g = blockIdx.x * GRID_THREADS + threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    if (g < c.len)
        {
        double dV;
        if (COMPUTE_DELTA)
            {
            dV = (c.W * c.st->scal) * exp(-gauss_exponent(c, g, s_cv));
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
    const unsigned int b0 = blockIdx.x * GRID_THREADS;
    apply_cells(c, b0, min(c.len, b0 + GRID_THREADS), blockIdx.x == 0, s_red);
    }

// ---------------------------------------------------------------------------------------------
// biasPotentialDerivative for every CV (:444-445 -> :738-776), V(s) (:448) and w(s) (:451)
__global__ __launch_bounds__(GRID_THREADS) void k_evaluate(const MetadCfg c)
    {
    __shared__ EvalShared sh;
    if (c.st->failed) return;                  // poisoned step (expired mailbox wait): the NaN state stays as it is
    if (threadIdx.x < c.n_cv) sh.cv[threadIdx.x] = c.st->cv[threadIdx.x];
    __syncthreads();
    evaluate_bias(c, sh, false, false);
    if (threadIdx.x < c.n_cv) c.st->bias[threadIdx.x] = sh.bias[threadIdx.x];
    if (threadIdx.x == 0)
        {
        c.st->V = sh.res[0];
        c.st->w = sh.res[1];
        if (sh.oob[0]) c.st->n_oob += 1;
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

namespace mtd
{
int metad_flush(mtd_metad *m, hipStream_t s)
    {
    if (m && m->pending_apply)
        {
        k_apply<<<m->cfg.n_gblocks, GRID_THREADS, 0, s>>>(m->cfg);
        MTD_LAUNCH_CHECK();
        m->pending_apply = 0;
        }
    return MTD_SUCCESS;
    }

// ---- the deferred pass as a passenger ------------------------------------------------------------------------------
// The second reweighting pass + accumulate of a deposit (k_apply) depends on nothing but that deposit's own grid launch and has to
// be complete before the NEXT grid launch: a 5 us kernel of pure latency when it runs on its own.  The engine whose grid launch
// left one pending announces it here together with the stream it ran on; a kernel of this library that is launched on the same
// stream before the next grid launch and has room for passengers (Steinhardt's finalize step) takes it along as extra blocks —
// stream order puts it after the deposit and before the next chain.  Nobody taking it is fine: metad_flush runs it as before.
namespace
{
std::mutex g_deferred_mutex;
mtd_metad *g_deferred_engine = nullptr;
hipStream_t g_deferred_stream = nullptr;
int g_deferred_device = -1;             // the NULL stream of two devices compares equal: the slot is keyed by (device, stream)

int current_device()
    {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess)
        {
        (void)hipGetLastError();
        return -1;
        }
    return dev;
    }
}

void announce_pending_apply(mtd_metad *m, hipStream_t s)
    {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_deferred_mutex);
    g_deferred_engine = m;
    g_deferred_stream = s;
    g_deferred_device = dev;
    }

void withdraw_pending_apply(mtd_metad *m)
    {
    std::lock_guard<std::mutex> lock(g_deferred_mutex);
    if (g_deferred_engine == m) g_deferred_engine = nullptr;
    }

// the engine whose deferred pass a launch on (current device, s) may carry, and its configuration; nullptr when there is none.
// The pass stays pending until the carrier's launch is known to have succeeded: commit_pending_apply.
mtd_metad *take_pending_apply(hipStream_t s, MetadCfg &cfg)
    {
    static const bool off = std::getenv("MTD_NO_APPLY_PASSENGER") != nullptr;      // diagnostic: every deferred pass as its own launch
    if (off) return nullptr;
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_deferred_mutex);
    mtd_metad *m = g_deferred_engine;
    if (!m || g_deferred_stream != s || g_deferred_device != dev || dev < 0) return nullptr;
    g_deferred_engine = nullptr;
    if (!m->pending_apply || m->comm) return nullptr;
    cfg = m->cfg;
    return m;
    }

void commit_pending_apply(mtd_metad *m)
    {
    std::lock_guard<std::mutex> lock(g_deferred_mutex);
    m->pending_apply = 0;
    }
}

namespace
{
// Diagnostic: IndexGrid::getCoordinates / the linear index (IndexGrid.cc:20-58) as the kernels compute them (decode,
// coord * factors) — tests/test_gpu_golden.py holds them against the tables the reference's own IndexGrid.cc produced
__global__ void k_debug_index(const MetadCfg c, const unsigned int n, const unsigned int *__restrict__ idx, unsigned int *__restrict__ coords,
                              unsigned int *__restrict__ back)
    {
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    unsigned int co[MAXCV];
    decode(c, idx[t], co);
    unsigned int r = 0;
    for (unsigned int i = 0; i < c.n_cv; ++i)
        {
        coords[(size_t)t * c.n_cv + i] = co[i];
        r += co[i] * c.factors[i];
        }
    back[t] = r;
    }
} // namespace

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
    // host reciprocals for the chain's quotients (metad_device.hpp::chain_div); every divisor has to qualify
    {
    const ExactDivisor dT = make_exact_divisor(T_shift);
    int ok = dT.fast;
    c.rT_shift = dT.y;
    for (unsigned int i = 0; i < n_cv && i < 3; ++i)
        {
        const ExactDivisor d1 = make_exact_divisor(c.delta[i]), d2 = make_exact_divisor(2.0 * c.delta[i]);
        c.rdelta[i] = d1.y;
        c.rdelta2[i] = d2.y;
        ok = ok && d1.fast && d2.fast;
        }
    c.fastdiv = ok;
    }
    m->stride = stride;
    m->add_bias = add_bias ? 1 : 0;
    m->pending_apply = 0;
    m->w_stale = 0;
    m->comm = nullptr;
    m->d_ll = nullptr;
    m->d_step_err = nullptr;
    m->h_step_err = nullptr;
    m->d_step_err_host = nullptr;
    m->step_seq = 0;
    m->last_launches = 0;
    m->step_mode = -1;
    m->walkers_checked = nullptr;
    m->walkers_stride = 0;
    m->walkers_add_bias = 0;

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

int mtd_debug_index_decode(unsigned int n_cv, const unsigned int *lengths, unsigned int n, const unsigned int *h_indices,
                           unsigned int *h_coords, unsigned int *h_index_back)
    {
    if (!lengths || !h_indices || !h_coords || !h_index_back || n == 0 || n_cv == 0) return MTD_ERR_INVALID_ARGUMENT;
    if (n_cv > (unsigned int)MAXCV) return MTD_ERR_UNSUPPORTED;
    MetadCfg c;
    std::memset(&c, 0, sizeof(c));
    c.n_cv = n_cv;
    for (unsigned int i = 0; i < n_cv; ++i)
        {
        c.lengths[i] = lengths[i];
        c.factors[i] = (i == 0) ? 1 : c.lengths[i - 1] * c.factors[i - 1];       // as mtd_metad_create
        }
    unsigned int *d_idx = nullptr, *d_co = nullptr, *d_back = nullptr;
    hipError_t e = hipMalloc(&d_idx, n * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc(&d_co, (size_t)n * n_cv * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc(&d_back, n * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemcpy(d_idx, h_indices, n * sizeof(unsigned int), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        {
        k_debug_index<<<(n + 63) / 64, 64>>>(c, n, d_idx, d_co, d_back);
        e = hipGetLastError();
        }
    if (e == hipSuccess) e = hipMemcpy(h_coords, d_co, (size_t)n * n_cv * sizeof(unsigned int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_index_back, d_back, n * sizeof(unsigned int), hipMemcpyDeviceToHost);
    (void)hipFree(d_idx);
    (void)hipFree(d_co);
    (void)hipFree(d_back);
    return (int)e;
    }

int mtd_metad_destroy(mtd_metad *m)
    {
    if (!m) return MTD_SUCCESS;
    mtd::withdraw_pending_apply(m);
    mtd::fused_step_release(m);
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
    { int frc = mtd::metad_flush(m, (hipStream_t)stream); if (frc) return frc; }
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
    { int frc = mtd::metad_flush(m, (hipStream_t)stream); if (frc) return frc; }
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

void mtd_debug_walker_check_pack(unsigned int stride, int add_bias, unsigned int timestep, double *out12)
    {
    const unsigned int q[3] = { stride, (unsigned int)(add_bias != 0), timestep };
    for (int i = 0; i < 3; ++i)
        {
        const double lo = (double)(q[i] & 0xffffu), hi = (double)(q[i] >> 16);
        out12[4 * i] = lo;
        out12[4 * i + 1] = lo * lo;
        out12[4 * i + 2] = hi;
        out12[4 * i + 3] = hi * hi;
        }
    }

int mtd_debug_walker_check_verify(const double *sums12, const double *mine12, unsigned int world)
    {
    const double W = (double)world;                                  // every product below is an exact integer < 2^53
    for (int i = 0; i < 12; ++i)
        if (sums12[i] != W * mine12[i]) return 0;
    return 1;
    }

int mtd_metad_update_bias_walkers(mtd_metad *m, mtd_rccl *walkers, unsigned int timestep, mtd_stream_t stream)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    int rc;
    if (walkers && (m->walkers_checked != (const void *)walkers || m->walkers_stride != m->stride || m->walkers_add_bias != m->add_bias))
        {
        // Whether a step deposits — and with it whether this walker enters the two all-reduces below — follows from stride,
        // add_hills and the time step.  Walkers that disagree would wait in different collectives for ever (RCCL has no bound,
        // unlike the mailbox).  Checked once per (communicator, stride, add_hills): sum and sum of squares of each quantity over
        // the walkers equal W x and W x^2 only when all are equal (Cauchy-Schwarz).  The quantities travel as 16-bit halves, so every
        // square stays below 2^32 and every sum is an exact integer in a double whatever order the reduction takes (a time
        // step of 3e9 squared does not fit 53 bits: fl(fl(x^2 + x^2) + x^2) != fl(3 x * x) gave false alarms).  One small
        // all-reduce and a synchronisation, at set-up.
        double *d_chk = nullptr;
        MTD_HIP_TRY(hipMalloc(&d_chk, 12 * sizeof(double)));
        double mine[12], h[12];
        mtd_debug_walker_check_pack(m->stride, m->add_bias, timestep, mine);
        hipError_t e = hipMemcpyAsync(d_chk, mine, sizeof(mine), hipMemcpyHostToDevice, (hipStream_t)stream);
        rc = e == hipSuccess ? mtd_comm_allreduce_large(walkers, d_chk, 12, MTD_ELEM_F64, stream) : (int)e;
        if (!rc)
            {
            e = hipMemcpyAsync(h, d_chk, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream);
            if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
            rc = (int)e;
            }
        (void)hipFree(d_chk);
        if (rc) return rc;
        if (!mtd_debug_walker_check_verify(h, mine, mtd_rccl_world(walkers))) return MTD_ERR_COLLECTIVE;
        m->walkers_checked = (const void *)walkers;
        m->walkers_stride = m->stride;
        m->walkers_add_bias = m->add_bias;
        }
    int deposited = 0;
    rc = mtd_metad_update_phase_a(m, timestep, &deposited, stream);
    if (rc) return rc;
    if (deposited && walkers)                                    // walkers == NULL: a single walker, the sum is its own increments
        {
        // sum up increments (:393-409): {grid_delta, sigma_grid_delta} and {hist_delta, hist_gauss_delta} are contiguous
        rc = mtd_comm_allreduce_large(walkers, m->cfg.grid_delta, 2 * (size_t)m->cfg.len, MTD_ELEM_F64, stream);
        if (rc) return rc;
        rc = mtd_comm_allreduce_large(walkers, m->cfg.hist_delta, 2 * (size_t)m->cfg.len, MTD_ELEM_U32, stream);
        if (rc) return rc;
        }
    return mtd_metad_update_phase_b(m, deposited, stream);
    }

int mtd_metad_update_bias(mtd_metad *m, unsigned int timestep, mtd_stream_t stream)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    static const bool four_launches = std::getenv("MTD_METAD_FOUR_LAUNCHES") != nullptr;    // diagnostic: the plain sequence
    if (m->cfg.n_cv <= 3 && !four_launches) return mtd::fused_grid_step(m, timestep, (hipStream_t)stream);
    { int frc = mtd::metad_flush(m, (hipStream_t)stream); if (frc) return frc; }
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
    if (m->pending_apply || m->w_stale)
        {
        int frc = mtd::metad_flush(m, (hipStream_t)stream);                // (nothing to do when launch A or a passenger ran the pass)
        if (frc) return frc;
        k_evaluate<<<1, GRID_THREADS, 0, (hipStream_t)stream>>>(m->cfg);   // w(s) of the now-final weight grid
        MTD_LAUNCH_CHECK();
        m->w_stale = 0;
        }
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
    // a poisoned step (expired mailbox wait): the NaN state above is what the step left; the status says why
    if (m->comm && mtd::comm_failed(m->comm)) return MTD_ERR_COMM_TIMEOUT;
    // the one-launch step's own hand-off between its blocks expired (fused_step.hip): sticky, whether or not a mailbox is attached —
    // blocks that did not time out may have run their grid pass, the arrays are not to be trusted
    if (m->h_step_err && *m->h_step_err) return MTD_ERR_COMM_TIMEOUT;
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
    if (m && which == 0)
        {
        // the caller may write the grid through the pointer: the patch around the last CV values (MetadState::patch_v) is
        // no longer known to be the grid's until the next deferred pass has rewritten it
        if (hipMemset(&m->cfg.st->patch_valid, 0, sizeof(int)) != hipSuccess)
            {
            (void)hipGetLastError();
            return nullptr;
            }
        }
    return m ? array_ptr(m, which, &e) : nullptr;
    }

int mtd_metad_grid_touched(mtd_metad *m, mtd_stream_t stream)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    MTD_HIP_TRY(hipMemsetAsync(&m->cfg.st->patch_valid, 0, sizeof(int), (hipStream_t)stream));
    return MTD_SUCCESS;
    }

int mtd_metad_get_array(mtd_metad *m, int which, void *host_out, mtd_stream_t stream)
    {
    if (!m || !host_out) return MTD_ERR_INVALID_ARGUMENT;
    if (m->h_step_err && *m->h_step_err) return MTD_ERR_COMM_TIMEOUT;      // (one-launch step: an expired in-kernel hand-off, see get_state)
    { int frc = mtd::metad_flush(m, (hipStream_t)stream); if (frc) return frc; }
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
    { int frc = mtd::metad_flush(m, (hipStream_t)stream); if (frc) return frc; }
    size_t e;
    void *p = array_ptr(m, which, &e);
    if (!p) return MTD_ERR_INVALID_ARGUMENT;
    MTD_HIP_TRY(hipMemcpyAsync(p, host_in, e * m->cfg.len, hipMemcpyHostToDevice, (hipStream_t)stream));
    if (which == 0)                                                  // the grid itself: the patch around the last CV values is stale
        MTD_HIP_TRY(hipMemsetAsync(&m->cfg.st->patch_valid, 0, sizeof(int), (hipStream_t)stream));
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
