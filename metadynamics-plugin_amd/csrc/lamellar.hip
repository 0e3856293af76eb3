// lamellar.hip — lamellar (Fourier-mode) order parameter on gfx950.
//
// What it computes (reference: LamellarOrderParameter.cc:42-74, 77-140, 143-179 — the CPU path the
// results must match; LamellarOrderParameterGPU.cu:8-236 is the CUDA design this replaces):
//     F_k = sum_j a(type_j) exp(i q_k . r_j),  s = sum_k Re F_k / N_global,
//     force_j = bias * (2 / N_global) * a(type_j) * sum_k q_k sin(q_k . r_j)        (factor 2: Q1)
// with q_k = 2*pi*(h b1' + k b2' + l b3'), b_i' the reciprocal rows of the GLOBAL box.
//
// MI355X design (not the reference's): ONE pass over the Scalar4 positions serves every fused CV
// and every mode (the reference re-reads all positions n_wave times per CV); the phase is formed as
// turns  t = h*g1 + k*g2 + l*g3  from the double-precision fractional projections g_i = b_i' . r
// (exact range reduction: v_fract / the period of cos(2*pi*t) is 1), trig runs in fp32, per-thread
// and per-wave sums in fp32 (<= 2^11 terms), everything across waves/blocks in fp64 with a fixed
// order (bitwise reproducible, no atomics).  Both kernels are pure streams: 16 B/lane coalesced
// loads, 16 B/lane coalesced stores, no LDS traffic in the inner loop beyond the coefficient lookup.
#include "lamellar_device.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "lamellar_host.hpp"

namespace
{

using namespace mtd;

constexpr int CV_THREADS = 512;
constexpr int CV_UNROLL = 4;
constexpr int FORCE_THREADS = 256;
constexpr int FORCE_UNROLL = 2;

// default: the hardware sine / cosine on the phase in turns, as the reference's GPU kernels (fast::sin / fast::cos,
// LamellarOrderParameterGPU.cu:36-37, 180); parity at 10^6 particles is tested in both modes
int g_fast_trig = 1;

// ---------------------------------------------------------------------------------------------
// Hot path: per-CV sums  partials[b][c] = sum_{j in block b} a_c(type_j) sum_k cos(q_k . r_j)
// ---------------------------------------------------------------------------------------------
template<typename S4, int NCV, bool FAST>
__global__ __launch_bounds__(CV_THREADS) void k_lamellar_cv_partials(const LamKArgs a, const S4 *__restrict__ postype,
                                                                     const unsigned int N, double *__restrict__ partials)
    {
    __shared__ float s_coeff[MTD_MAX_CV * MTD_MAX_TYPES];
    __shared__ double s_wave[(CV_THREADS / MTD_WAVE) * NCV];
    __shared__ ModeTables s_mt;
    RawGroup<S4, CV_UNROLL> first;
    lam_load_group<S4, CV_UNROLL>(postype, N, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, first);
    load_coeff(a, s_coeff);
    load_modes_cv(a, s_mt);
    __syncthreads();
    float acc[NCV];
#pragma unroll
    for (int c = 0; c < NCV; ++c) acc[c] = 0.0f;
    lam_cv_accumulate<S4, NCV, FAST, CV_UNROLL>(a, postype, N, blockIdx.x * blockDim.x + threadIdx.x,
                                                gridDim.x * blockDim.x, s_coeff, s_mt, first, acc);
    lam_cv_block_reduce<NCV>(acc, s_wave, partials, blockIdx.x);
    }

// ---------------------------------------------------------------------------------------------
// Drop-in gpu_calculate_fourier_modes: per-mode (Re, Im) partial sums, modes in chunks of 8
// ---------------------------------------------------------------------------------------------
template<typename S4, bool FAST>
__global__ __launch_bounds__(CV_THREADS) void k_lamellar_mode_partials(const LamKArgs a, const S4 *__restrict__ postype,
                                                                       const unsigned int N, double *__restrict__ partials)
    {
    __shared__ float s_coeff[MTD_MAX_TYPES];
    __shared__ double s_wave[CV_THREADS / MTD_WAVE][16];

    if (threadIdx.x < MTD_MAX_TYPES) s_coeff[threadIdx.x] = a.coeff[0][threadIdx.x];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned int stride = gridDim.x * blockDim.x;

    for (unsigned int k0 = 0; k0 < a.n_modes; k0 += 8)
        {
        float re[8], im[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) re[m] = im[m] = 0.0f;

        for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride)
            {
            const Particle p = scalar4_traits<S4>::load(postype, i);
            float g0, g1, g2;
            project(a, p, g0, g1, g2);
            const float w = s_coeff[p.type];
#pragma unroll
            for (int m = 0; m < 8; ++m)
                {
                const unsigned int k = k0 + m;
                if (k < a.n_modes)
                    {
                    const float4 h = a.h[k];
                    const float t = h.x * g0 + h.y * g1 + h.z * g2;
                    re[m] += w * cos2pi<FAST>(t);
                    im[m] += w * sin2pi<FAST>(t);
                    }
                }
            }

        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m)
            {
            const float r = wave_sum(re[m]);
            const float s = wave_sum(im[m]);
            if (lane == 0)
                {
                s_wave[wave][2 * m] = (double)r;
                s_wave[wave][2 * m + 1] = (double)s;
                }
            }
        __syncthreads();
        if (threadIdx.x < 16 && k0 + threadIdx.x / 2 < a.n_modes)
            {
            double r = 0.0;
            for (int w = 0; w < CV_THREADS / MTD_WAVE; ++w) r += s_wave[w][threadIdx.x];
            partials[(size_t)blockIdx.x * (2 * a.n_modes) + 2 * k0 + threadIdx.x] = r;
            }
        }
    }

// ---------------------------------------------------------------------------------------------
// out[c] = shift + scale * sum_b partials[b*stride + c]; wave w of block g owns output c = 4 g + w
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_partials(const double *__restrict__ partials, const unsigned int n_partials,
                                                         const unsigned int stride, const unsigned int count,
                                                         const double scale, const double shift, double *__restrict__ out)
    {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned int c = blockIdx.x * 4 + wave;
    if (c >= count) return;
    // eight loads in flight per lane, added in the order of the plain loop (same bits): with one load per trip the kernel
    // ran at eight memory latencies for the 1024 block sums of the Steinhardt pass
    double v = 0.0;
    unsigned int b = lane;
    for (; b + 7 * MTD_WAVE < n_partials; b += 8 * MTD_WAVE)
        {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = partials[(size_t)(b + u * MTD_WAVE) * stride + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
        }
    for (; b < n_partials; b += MTD_WAVE) v += partials[(size_t)b * stride + c];
    v = wave_sum(v);
    if (lane == 0) out[c] = shift + scale * v;
    }

// ---------------------------------------------------------------------------------------------
// Forces of every fused CV in one pass; bias factors come from device memory (or a host scalar
// for the drop-in entry point)
// ---------------------------------------------------------------------------------------------
template<typename S4, bool FAST>
__global__ __launch_bounds__(FORCE_THREADS) void k_lamellar_forces(const LamKArgs a, const S4 *__restrict__ postype,
                                                                   const ForcePtrs out, const unsigned int N,
                                                                   const double *__restrict__ d_bias, const double bias_host,
                                                                   const double two_over_n)
    {
    __shared__ float s_wcoef[MTD_MAX_CV * MTD_MAX_TYPES];
    __shared__ ModeTables s_mt;
    load_modes(a, s_mt, true);
    // fold bias_c * 2 / N_global into the per-type coefficient once per block
    for (unsigned int i = threadIdx.x; i < MTD_MAX_CV * MTD_MAX_TYPES; i += blockDim.x)
        {
        const unsigned int c = i / MTD_MAX_TYPES;
        double b = 0.0;
        if (c < a.n_cv) b = d_bias ? d_bias[c] : bias_host;
        s_wcoef[i] = (float)((double)a.coeff[c][i % MTD_MAX_TYPES] * b * two_over_n);
        }
    __syncthreads();
    lam_force_pass<S4, FAST, FORCE_UNROLL>(a, postype, out, N, blockIdx.x * blockDim.x + threadIdx.x,
                                           gridDim.x * blockDim.x, s_wcoef, s_mt);
    }

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

unsigned int env_uint(const char *name, unsigned int dflt)
    {
    const char *s = std::getenv(name);
    if (!s || !*s) return dflt;
    long v = std::strtol(s, nullptr, 10);
    return v > 0 ? (unsigned int)v : dflt;
    }

unsigned int cv_blocks(unsigned int N)
    {
    static const unsigned int forced = env_uint("MTD_LAM_CV_BLOCKS", 0);
    if (forced) return forced > LAM_MAX_BLOCKS ? LAM_MAX_BLOCKS : forced;
    // one 512-thread block per CU at most: few partial sums for the consumer's prologue to reduce
    unsigned int b = (N + CV_THREADS * CV_UNROLL - 1) / (CV_THREADS * CV_UNROLL);
    if (b < 1) b = 1;
    if (b > 256) b = 256;
    return b;
    }

unsigned int force_blocks(unsigned int N)
    {
    static const unsigned int forced = env_uint("MTD_LAM_FORCE_BLOCKS", 0);
    if (forced) return forced;
    unsigned int b = (N + FORCE_THREADS * FORCE_UNROLL * 2 - 1) / (FORCE_THREADS * FORCE_UNROLL * 2);
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return b;
    }

template<typename S4, bool FAST>
int launch_cv(const LamKArgs &k, unsigned int N, const void *d_postype, double *d_partials, unsigned int blocks, hipStream_t s)
    {
    const S4 *p = (const S4 *)d_postype;
    if (k.n_cv == 1)
        k_lamellar_cv_partials<S4, 1, FAST><<<blocks, CV_THREADS, 0, s>>>(k, p, N, d_partials);
    else if (k.n_cv == 2)
        k_lamellar_cv_partials<S4, 2, FAST><<<blocks, CV_THREADS, 0, s>>>(k, p, N, d_partials);
    else if (k.n_cv <= 4)
        {
        // partial rows are n_cv wide only when NCV == n_cv: pad by instantiating exact widths
        if (k.n_cv == 3)
            k_lamellar_cv_partials<S4, 3, FAST><<<blocks, CV_THREADS, 0, s>>>(k, p, N, d_partials);
        else
            k_lamellar_cv_partials<S4, 4, FAST><<<blocks, CV_THREADS, 0, s>>>(k, p, N, d_partials);
        }
    else if (k.n_cv == 5)
        k_lamellar_cv_partials<S4, 5, FAST><<<blocks, CV_THREADS, 0, s>>>(k, p, N, d_partials);
    else if (k.n_cv == 6)
        k_lamellar_cv_partials<S4, 6, FAST><<<blocks, CV_THREADS, 0, s>>>(k, p, N, d_partials);
    else if (k.n_cv == 7)
        k_lamellar_cv_partials<S4, 7, FAST><<<blocks, CV_THREADS, 0, s>>>(k, p, N, d_partials);
    else
        k_lamellar_cv_partials<S4, 8, FAST><<<blocks, CV_THREADS, 0, s>>>(k, p, N, d_partials);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

} // namespace

namespace mtd
{
int fill_kargs(LamKArgs &k, const mtd_lamellar_set *set, const mtd_box *box)
    {
    if (!set || !box) return MTD_ERR_INVALID_ARGUMENT;
    if (set->n_cv == 0 || set->n_cv > MTD_MAX_CV || set->n_modes == 0 || set->n_modes > MTD_MAX_MODES
        || set->n_types == 0 || set->n_types > MTD_MAX_TYPES)
        return MTD_ERR_INVALID_ARGUMENT;
    if (set->first[0] != 0 || set->first[set->n_cv] != set->n_modes) return MTD_ERR_INVALID_ARGUMENT;
    for (unsigned int c = 0; c < set->n_cv; ++c)
        if (set->first[c + 1] <= set->first[c]) return MTD_ERR_INVALID_ARGUMENT; // cv.py:232-234: empty list is an error
    if (!(box->L[0] > 0.0) || !(box->L[1] > 0.0) || !(box->L[2] > 0.0)) return MTD_ERR_INVALID_ARGUMENT;

    std::memset(&k, 0, sizeof(k));
    reciprocal_rows(*box, k.B);
    k.n_cv = set->n_cv;
    k.n_modes = set->n_modes;
    k.n_types = set->n_types;
    if (set->trig_mode != MTD_TRIG_DEFAULT && set->trig_mode != MTD_TRIG_HARDWARE && set->trig_mode != MTD_TRIG_ACCURATE)
        return MTD_ERR_INVALID_ARGUMENT;
    k.trig = (unsigned int)set->trig_mode;
    for (unsigned int c = 0; c <= set->n_cv; ++c) k.first[c] = set->first[c];
    for (unsigned int c = 0; c < MTD_MAX_CV; ++c) k.slot[c] = (unsigned char)c;
    const double two_pi = 2.0 * M_PI;
    for (unsigned int m = 0; m < set->n_modes; ++m)
        for (int d = 0; d < 3; ++d)
            {
            const float hv = (float)set->hkl[m][d];
            const float qv = (float)(two_pi * (set->hkl[m][0] * k.B[0][d] + set->hkl[m][1] * k.B[1][d] + set->hkl[m][2] * k.B[2][d]));
            if (d == 0) { k.h[m].x = hv; k.q[m].x = qv; }
            if (d == 1) { k.h[m].y = hv; k.q[m].y = qv; }
            if (d == 2) { k.h[m].z = hv; k.q[m].z = qv; }
            }
    for (unsigned int c = 0; c < set->n_cv; ++c)
        for (unsigned int t = 0; t < set->n_types; ++t) k.coeff[c][t] = (float)set->coeff[c][t];
    // CV pass: fold every second harmonic 2(h,k,l) into its fundamental (h,k,l) of the same CV (cos 2x = 2 cos^2 x - 1).
    // One level only: a folded mode does not take a harmonic of its own, and every mode is folded at most once.
    static const bool fold = env_uint("MTD_LAM_FOLD_HARMONICS", 1) != 0;
    for (unsigned int c = 0; c < set->n_cv; ++c)
        {
        bool folded[MTD_MAX_MODES] = {false};
        for (unsigned int m = set->first[c]; fold && m < set->first[c + 1]; ++m)
            {
            if (folded[m] || k.h[m].w != 0.0f) continue;
            const int *hm = set->hkl[m];
            if (hm[0] == 0 && hm[1] == 0 && hm[2] == 0) continue;
            for (unsigned int j = set->first[c]; j < set->first[c + 1]; ++j)
                {
                const int *hj = set->hkl[j];
                if (j != m && !folded[j] && k.h[j].w == 0.0f && hj[0] == 2 * hm[0] && hj[1] == 2 * hm[1] && hj[2] == 2 * hm[2])
                    {
                    k.h[m].w = 1.0f;
                    folded[j] = true;
                    break;
                    }
                }
            }
        unsigned int n = 0;
        for (unsigned int m = set->first[c]; m < set->first[c + 1]; ++m)
            if (!folded[m]) k.corder[set->first[c] + n++] = (unsigned char)m;
        k.nact[c] = (unsigned char)n;
        for (unsigned int m = set->first[c]; m < set->first[c + 1]; ++m)       // the rest of the CV's slots: harmless duplicates
            if (folded[m]) k.corder[set->first[c] + n++] = (unsigned char)m;
        }
    return MTD_SUCCESS;
    }


unsigned int lam_cv_blocks(unsigned int N) { return cv_blocks(N); }
unsigned int lam_force_blocks(unsigned int N) { return force_blocks(N); }
// the hardware sine / cosine for this mode set?  The set's own mode decides (mtd_lamellar_set::trig_mode), the process default
// (mtd_lamellar_set_fast_trig) only where the set leaves it open.  Their argument (in turns) must stay inside [-256, 256]: with
// positions inside the box |b_i' . r| <= 1/2, so |phase| <= (|h| + |k| + |l|) / 2 — sets with larger indices than any lamellar
// study uses take the accurate path whatever the mode says (the bound leaves room for particles five box lengths outside:
// the precondition stated in mtd_abi.h)
int lam_fast_trig(const LamKArgs &k)
    {
    if (k.trig == MTD_TRIG_ACCURATE) return 0;
    if (k.trig == MTD_TRIG_DEFAULT && !g_fast_trig) return 0;
    for (unsigned int m = 0; m < k.n_modes; ++m)
        if (std::fabs(k.h[m].x) + std::fabs(k.h[m].y) + std::fabs(k.h[m].z) > 100.0f) return 0;
    return 1;
    }
} // namespace mtd

extern "C" {

int mtd_lamellar_set_fast_trig(int enable)
    {
    g_fast_trig = enable ? 1 : 0;
    return MTD_SUCCESS;
    }

int mtd_lamellar_get_fast_trig(void) { return g_fast_trig; }

size_t mtd_lamellar_scratch_doubles(unsigned int n_particles)
    {
    (void)n_particles;
    return (size_t)LAM_MAX_BLOCKS * 2 * MTD_MAX_MODES;
    }

int mtd_lamellar_cv_partials(const mtd_lamellar_set *set, unsigned int n_particles, const void *d_postype,
                             int dtype, const mtd_box *global_box, double *d_partials,
                             unsigned int *n_partials, mtd_stream_t stream)
    {
    LamKArgs k;
    int rc = fill_kargs(k, set, global_box);
    if (rc) return rc;
    if (!d_partials || !n_partials || (n_particles && !d_postype)) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const unsigned int blocks = cv_blocks(n_particles);
    *n_partials = blocks;
    if (dtype == MTD_F32)
        return lam_fast_trig(k) ? launch_cv<float4, true>(k, n_particles, d_postype, d_partials, blocks, s)
                           : launch_cv<float4, false>(k, n_particles, d_postype, d_partials, blocks, s);
    return lam_fast_trig(k) ? launch_cv<double4, true>(k, n_particles, d_postype, d_partials, blocks, s)
                       : launch_cv<double4, false>(k, n_particles, d_postype, d_partials, blocks, s);
    }

int mtd_reduce_partials(const double *d_partials, unsigned int n_partials, unsigned int stride,
                        unsigned int count, double scale, double shift, double *d_out, mtd_stream_t stream)
    {
    if (!d_partials || !d_out || count == 0) return MTD_ERR_INVALID_ARGUMENT;
    k_reduce_partials<<<(count + 3) / 4, 256, 0, (hipStream_t)stream>>>(d_partials, n_partials, stride, count, scale, shift, d_out);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

int mtd_calculate_fourier_modes(unsigned int n_wave, const int *lattice_vectors, unsigned int n_particles,
                                const void *d_postype, int dtype, const double *mode, unsigned int n_types,
                                double *d_fourier_modes, double *d_scratch, const mtd_box *global_box,
                                mtd_stream_t stream)
    {
    if (!lattice_vectors || !mode || !d_fourier_modes || !d_scratch) return MTD_ERR_INVALID_ARGUMENT;
    if (n_wave == 0 || n_wave > MTD_MAX_MODES || n_types == 0 || n_types > MTD_MAX_TYPES) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    mtd_lamellar_set set;
    std::memset(&set, 0, sizeof(set));
    set.n_cv = 1;
    set.n_types = n_types;
    set.n_modes = n_wave;
    set.first[0] = 0;
    set.first[1] = n_wave;
    for (unsigned int k = 0; k < n_wave; ++k)
        for (int d = 0; d < 3; ++d) set.hkl[k][d] = lattice_vectors[3 * k + d];
    for (unsigned int t = 0; t < n_types; ++t) set.coeff[0][t] = mode[t];
    LamKArgs k;
    int rc = fill_kargs(k, &set, global_box);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const unsigned int blocks = cv_blocks(n_particles);
    if (dtype == MTD_F32)
        {
        if (lam_fast_trig(k))
            k_lamellar_mode_partials<float4, true><<<blocks, CV_THREADS, 0, s>>>(k, (const float4 *)d_postype, n_particles, d_scratch);
        else
            k_lamellar_mode_partials<float4, false><<<blocks, CV_THREADS, 0, s>>>(k, (const float4 *)d_postype, n_particles, d_scratch);
        }
    else
        {
        if (lam_fast_trig(k))
            k_lamellar_mode_partials<double4, true><<<blocks, CV_THREADS, 0, s>>>(k, (const double4 *)d_postype, n_particles, d_scratch);
        else
            k_lamellar_mode_partials<double4, false><<<blocks, CV_THREADS, 0, s>>>(k, (const double4 *)d_postype, n_particles, d_scratch);
        }
    MTD_LAUNCH_CHECK();
    return mtd_reduce_partials(d_scratch, blocks, 2 * n_wave, 2 * n_wave, 1.0, 0.0, d_fourier_modes, stream);
    }

static int lamellar_forces_impl(const mtd_lamellar_set *set, unsigned int n_particles, const void *d_postype,
                                void *const *d_force, int dtype, unsigned int n_global, const double *d_bias,
                                double bias_host, const mtd_box *global_box, mtd_stream_t stream)
    {
    LamKArgs k;
    int rc = fill_kargs(k, set, global_box);
    if (rc) return rc;
    if (!d_force || n_global == 0 || (n_particles && !d_postype)) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    ForcePtrs out;
    std::memset(&out, 0, sizeof(out));
    for (unsigned int c = 0; c < set->n_cv; ++c)
        {
        if (!d_force[c] && n_particles) return MTD_ERR_INVALID_ARGUMENT;
        out.f[c] = d_force[c];
        }
    if (n_particles == 0) return MTD_SUCCESS;
    hipStream_t s = (hipStream_t)stream;
    const unsigned int blocks = force_blocks(n_particles);
    const double two_over_n = 2.0 / (double)n_global;
    if (dtype == MTD_F32)
        {
        if (lam_fast_trig(k))
            k_lamellar_forces<float4, true><<<blocks, FORCE_THREADS, 0, s>>>(k, (const float4 *)d_postype, out, n_particles, d_bias, bias_host, two_over_n);
        else
            k_lamellar_forces<float4, false><<<blocks, FORCE_THREADS, 0, s>>>(k, (const float4 *)d_postype, out, n_particles, d_bias, bias_host, two_over_n);
        }
    else
        {
        if (lam_fast_trig(k))
            k_lamellar_forces<double4, true><<<blocks, FORCE_THREADS, 0, s>>>(k, (const double4 *)d_postype, out, n_particles, d_bias, bias_host, two_over_n);
        else
            k_lamellar_forces<double4, false><<<blocks, FORCE_THREADS, 0, s>>>(k, (const double4 *)d_postype, out, n_particles, d_bias, bias_host, two_over_n);
        }
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

int mtd_lamellar_forces(const mtd_lamellar_set *set, unsigned int n_particles, const void *d_postype,
                        void *const *d_force, int dtype, unsigned int n_global, const double *d_bias,
                        const mtd_box *global_box, mtd_stream_t stream)
    {
    if (!d_bias) return MTD_ERR_INVALID_ARGUMENT;
    return lamellar_forces_impl(set, n_particles, d_postype, d_force, dtype, n_global, d_bias, 0.0, global_box, stream);
    }

int mtd_compute_sq_forces(unsigned int n_particles, const void *d_postype, void *d_force, int dtype,
                          unsigned int n_wave, const int *lattice_vectors, const double *mode,
                          unsigned int n_types, unsigned int n_global, double bias,
                          const mtd_box *global_box, mtd_stream_t stream)
    {
    if (!lattice_vectors || !mode) return MTD_ERR_INVALID_ARGUMENT;
    if (n_wave == 0 || n_wave > MTD_MAX_MODES || n_types == 0 || n_types > MTD_MAX_TYPES) return MTD_ERR_INVALID_ARGUMENT;
    mtd_lamellar_set set;
    std::memset(&set, 0, sizeof(set));
    set.n_cv = 1;
    set.n_types = n_types;
    set.n_modes = n_wave;
    set.first[0] = 0;
    set.first[1] = n_wave;
    for (unsigned int k = 0; k < n_wave; ++k)
        for (int d = 0; d < 3; ++d) set.hkl[k][d] = lattice_vectors[3 * k + d];
    for (unsigned int t = 0; t < n_types; ++t) set.coeff[0][t] = mode[t];
    void *f[1] = { d_force };
    return lamellar_forces_impl(&set, n_particles, d_postype, f, dtype, n_global, nullptr, bias, global_box, stream);
    }

} // extern "C"
