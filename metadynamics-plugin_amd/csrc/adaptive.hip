// adaptive.hip — adaptive Gaussians: the derivative-product reduction of IntegratorMetaDynamics::computeSigma on gfx950.
//
// Reference: IntegratorMetaDynamics.cc:1205-1294.  The reference maps every CV's force array to the host and runs
// n_cv^2 serial loops over the particles in Scalar precision.  Here one streaming pass reads each force array once
// (N * n_cv * sizeof(Scalar4) bytes — the compulsory traffic), forms the n_cv(n_cv+1)/2 distinct products per particle in
// registers, and sums them in double in a fixed order (thread -> wave -> block -> block index; no atomics), so the result
// is reproducible run to run.  The n_cv x n_cv sqrt / inverse stays on the host like the reference's Eigen call.
#include "mtd_device.hpp"

#include <cmath>
#include <vector>

namespace
{

using namespace mtd;

constexpr int SIG_THREADS = 256;
constexpr unsigned int SIG_MAX_BLOCKS = 512;
constexpr unsigned int SIG_MAX_PAIRS = MTD_METAD_MAX_CV * (MTD_METAD_MAX_CV + 1) / 2;

struct SigmaArgs
    {
    const void *force[MTD_METAD_MAX_CV];
    };

template<typename S4, int NCV>
__global__ __launch_bounds__(SIG_THREADS) void k_sigma_partials(const SigmaArgs args, const unsigned int N,
                                                                double *__restrict__ partials)
    {
    constexpr int NP = NCV * (NCV + 1) / 2;
    __shared__ double s_red[16];
    double acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = 0.0;
    const unsigned int stride = gridDim.x * blockDim.x;
    for (unsigned int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride)
        {
        double fx[NCV], fy[NCV], fz[NCV];
#pragma unroll
        for (int c = 0; c < NCV; ++c)
            {
            const S4 f = ((const S4 *)args.force[c])[n];
            fx[c] = (double)f.x;
            fy[c] = (double)f.y;
            fz[c] = (double)f.z;
            }
        int p = 0;
#pragma unroll
        for (int i = 0; i < NCV; ++i)
#pragma unroll
            for (int j = i; j < NCV; ++j, ++p) acc[p] += fx[i] * fx[j] + fy[i] * fy[j] + fz[i] * fz[j];   // :1241-1246
        }
#pragma unroll
    for (int p = 0; p < NP; ++p)
        {
        const double r = block_sum(acc[p], s_red);
        if (threadIdx.x == 0) partials[(size_t)blockIdx.x * NP + p] = r;
        }
    }

__global__ void k_sigma_final(const double *__restrict__ partials, const unsigned int n_blocks, const unsigned int n_pairs,
                              double *__restrict__ out)
    {
    const unsigned int p = threadIdx.x;
    if (p >= n_pairs) return;
    double r = 0.0;
    for (unsigned int b = 0; b < n_blocks; ++b) r += partials[(size_t)b * n_pairs + p];
    out[p] = r;
    }

template<typename S4>
void launch_sigma(unsigned int n_cv, const SigmaArgs &a, unsigned int N, double *partials, unsigned int blocks, hipStream_t s)
    {
    switch (n_cv)
        {
        case 1: k_sigma_partials<S4, 1><<<blocks, SIG_THREADS, 0, s>>>(a, N, partials); break;
        case 2: k_sigma_partials<S4, 2><<<blocks, SIG_THREADS, 0, s>>>(a, N, partials); break;
        case 3: k_sigma_partials<S4, 3><<<blocks, SIG_THREADS, 0, s>>>(a, N, partials); break;
        case 4: k_sigma_partials<S4, 4><<<blocks, SIG_THREADS, 0, s>>>(a, N, partials); break;
        case 5: k_sigma_partials<S4, 5><<<blocks, SIG_THREADS, 0, s>>>(a, N, partials); break;
        default: k_sigma_partials<S4, 6><<<blocks, SIG_THREADS, 0, s>>>(a, N, partials); break;
        }
    }

} // namespace

extern "C" {

size_t mtd_sigma_scratch_doubles(void) { return (size_t)(SIG_MAX_BLOCKS + 1) * SIG_MAX_PAIRS; }

int mtd_sigma_products(unsigned int n_cv, const void *const *d_force, unsigned int n_particles, int dtype, double sigma_g,
                       double *d_scratch, double *sigmasq, mtd_stream_t stream)
    {
    if (n_cv == 0 || n_cv > MTD_METAD_MAX_CV || !d_force || !d_scratch || !sigmasq) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    // CVs without derivatives (d_force[c] == NULL) are left out of the device pass; their rows/columns come back 0
    SigmaArgs a;
    unsigned int map[MTD_METAD_MAX_CV], m = 0;
    for (unsigned int c = 0; c < n_cv; ++c)
        if (d_force[c])
            {
            a.force[m] = d_force[c];
            map[m++] = c;
            }
    for (unsigned int i = 0; i < n_cv * n_cv; ++i) sigmasq[i] = 0.0;
    if (m == 0) return MTD_SUCCESS;
    const unsigned int n_pairs = m * (m + 1) / 2;
    unsigned int blocks = (n_particles + SIG_THREADS * 4 - 1) / (SIG_THREADS * 4);
    if (blocks < 1) blocks = 1;
    if (blocks > SIG_MAX_BLOCKS) blocks = SIG_MAX_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    double *d_out = d_scratch + (size_t)SIG_MAX_BLOCKS * SIG_MAX_PAIRS;
    if (dtype == MTD_F32)
        launch_sigma<float4>(m, a, n_particles, d_scratch, blocks, s);
    else
        launch_sigma<double4>(m, a, n_particles, d_scratch, blocks, s);
    MTD_LAUNCH_CHECK();
    k_sigma_final<<<1, 64, 0, s>>>(d_scratch, blocks, n_pairs, d_out);
    MTD_LAUNCH_CHECK();
    double host[SIG_MAX_PAIRS];
    hipError_t e = hipMemcpyAsync(host, d_out, sizeof(double) * n_pairs, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    unsigned int p = 0;
    for (unsigned int i = 0; i < m; ++i)
        for (unsigned int j = i; j < m; ++j, ++p)
            {
            const double v = sigma_g * sigma_g * host[p];
            sigmasq[map[i] * n_cv + map[j]] = v;
            sigmasq[map[j] * n_cv + map[i]] = v;
            }
    return MTD_SUCCESS;
    }

// m_ij = sqrt(sigmasq_ij) element-wise (a negative product gives NaN like the reference), then the dense inverse
// (:1273-1286; Eigen's dynamic-size inverse() is a partial-pivoting LU — restated, not linked)
int mtd_sigma_inverse(unsigned int n_cv, const double *sigmasq, double *sigma_inv)
    {
    if (n_cv == 0 || n_cv > MTD_METAD_MAX_CV || !sigmasq || !sigma_inv) return MTD_ERR_INVALID_ARGUMENT;
    const unsigned int n = n_cv;
    double lu[MTD_METAD_MAX_CV][MTD_METAD_MAX_CV];
    unsigned int perm[MTD_METAD_MAX_CV];
    for (unsigned int i = 0; i < n; ++i)
        {
        perm[i] = i;
        for (unsigned int j = 0; j < n; ++j) lu[i][j] = std::sqrt(sigmasq[i * n + j]);
        }
    for (unsigned int k = 0; k < n; ++k)
        {
        unsigned int piv = k;
        double best = std::fabs(lu[k][k]);
        for (unsigned int r = k + 1; r < n; ++r)
            if (std::fabs(lu[r][k]) > best)
                {
                best = std::fabs(lu[r][k]);
                piv = r;
                }
        if (piv != k)
            {
            for (unsigned int j = 0; j < n; ++j) std::swap(lu[k][j], lu[piv][j]);
            std::swap(perm[k], perm[piv]);
            }
        for (unsigned int r = k + 1; r < n; ++r)
            {
            lu[r][k] /= lu[k][k];
            for (unsigned int j = k + 1; j < n; ++j) lu[r][j] -= lu[r][k] * lu[k][j];
            }
        }
    for (unsigned int col = 0; col < n; ++col)
        {
        double y[MTD_METAD_MAX_CV];
        for (unsigned int i = 0; i < n; ++i)
            {
            double v = (perm[i] == col) ? 1.0 : 0.0;
            for (unsigned int j = 0; j < i; ++j) v -= lu[i][j] * y[j];
            y[i] = v;
            }
        for (int i = (int)n - 1; i >= 0; --i)
            {
            double v = y[i];
            for (unsigned int j = i + 1; j < n; ++j) v -= lu[i][j] * sigma_inv[j * n + col];
            sigma_inv[i * n + col] = v / lu[i][i];
            }
        }
    return MTD_SUCCESS;
    }

} // extern "C"
