// wte.hip — WellTemperedEnsemble (potential energy as collective variable) on gfx950.
//
// Reference (CPU path to match): WellTemperedEnsemble.cc:30-68 (PE = sum_j net_force_j.w + external
// energy), :135-188 (net force / torque / virial *= 1 + bias).  CUDA design replaced:
// WellTemperedEnsemble.cu:19-89 (scale), :97-243 (two-launch reduction with a CAS-loop double
// atomicAdd and a managed-memory readback).  Here: one streaming pass writes fixed-order block
// partial sums (no atomics), the final sum is folded into the grid engine's k_prepare, and the
// scale kernel reads the bias factor from device memory.
#include "mtd_device.hpp"

namespace
{

using namespace mtd;

constexpr int WTE_THREADS = 256;
constexpr unsigned int WTE_MAX_BLOCKS = 1024;

template<typename S4>
__global__ __launch_bounds__(WTE_THREADS) void k_wte_energy_partials(const S4 *__restrict__ net_force, const unsigned int N,
                                                                     double *__restrict__ partials)
    {
    __shared__ double s_red[16];
    double acc = 0.0;
    const unsigned int stride = gridDim.x * blockDim.x;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) acc += (double)net_force[i].w;
    acc = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
    }

template<typename S4, typename S>
__global__ __launch_bounds__(WTE_THREADS) void k_wte_scale(S4 *__restrict__ net_force, S4 *__restrict__ net_torque,
                                                           S *__restrict__ net_virial, const unsigned int pitch,
                                                           const unsigned int N, const double *__restrict__ d_bias,
                                                           const double bias_host, const int scale_torque_w,
                                                           const double offset)
    {
    // offset 1: WellTemperedEnsemble.cc:140 (fac = 1 + bias); offset 0: CollectiveWrapper.cc:125, :153 (fac = bias)
    const S fac = (S)(offset + (d_bias ? *d_bias : bias_host));
    const unsigned int stride = gridDim.x * blockDim.x;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride)
        {
        S4 f = net_force[i];
        f.x *= fac; f.y *= fac; f.z *= fac;
        net_force[i] = f;
        if (net_torque)
            {
            S4 t = net_torque[i];
            t.x *= fac; t.y *= fac; t.z *= fac;
            if (scale_torque_w) t.w *= fac;                                  // CPU path only (:169, Q18)
            net_torque[i] = t;
            }
        if (net_virial)
            {
#pragma unroll
            for (int r = 0; r < 6; ++r) net_virial[i + (size_t)r * pitch] *= fac;
            }
        }
    }

unsigned int wte_blocks(unsigned int N)
    {
    unsigned int b = (N + WTE_THREADS * 4 - 1) / (WTE_THREADS * 4);
    if (b < 1) b = 1;
    if (b > WTE_MAX_BLOCKS) b = WTE_MAX_BLOCKS;
    return b;
    }

} // namespace

extern "C" {

size_t mtd_wte_scratch_doubles(unsigned int n_particles)
    {
    (void)n_particles;
    return WTE_MAX_BLOCKS;
    }

int mtd_wte_energy_partials(unsigned int n_particles, const void *d_net_force, int dtype,
                            double *d_partials, unsigned int *n_partials, mtd_stream_t stream)
    {
    if (!d_partials || !n_partials || (n_particles && !d_net_force)) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    const unsigned int blocks = wte_blocks(n_particles);
    *n_partials = blocks;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTD_F32)
        k_wte_energy_partials<float4><<<blocks, WTE_THREADS, 0, s>>>((const float4 *)d_net_force, n_particles, d_partials);
    else
        k_wte_energy_partials<double4><<<blocks, WTE_THREADS, 0, s>>>((const double4 *)d_net_force, n_particles, d_partials);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

static int scale_arrays(unsigned int n_particles, void *d_net_force, void *d_net_torque, void *d_net_virial,
                        unsigned int virial_pitch, int dtype, const double *d_bias, double bias_host, int scale_torque_w,
                        double offset, mtd_stream_t stream)
    {
    if (n_particles && !d_net_force) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    if (d_net_virial && virial_pitch < n_particles) return MTD_ERR_INVALID_ARGUMENT;
    if (n_particles == 0) return MTD_SUCCESS;
    unsigned int blocks = (n_particles + WTE_THREADS - 1) / WTE_THREADS;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTD_F32)
        k_wte_scale<float4, float><<<blocks, WTE_THREADS, 0, s>>>((float4 *)d_net_force, (float4 *)d_net_torque,
                                                                  (float *)d_net_virial, virial_pitch, n_particles,
                                                                  d_bias, bias_host, scale_torque_w, offset);
    else
        k_wte_scale<double4, double><<<blocks, WTE_THREADS, 0, s>>>((double4 *)d_net_force, (double4 *)d_net_torque,
                                                                    (double *)d_net_virial, virial_pitch, n_particles,
                                                                    d_bias, bias_host, scale_torque_w, offset);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

int mtd_wte_scale_netforce(unsigned int n_particles, void *d_net_force, void *d_net_torque,
                           void *d_net_virial, unsigned int virial_pitch, int dtype, const double *d_bias,
                           double bias_host, int scale_torque_w, mtd_stream_t stream)
    {
    return scale_arrays(n_particles, d_net_force, d_net_torque, d_net_virial, virial_pitch, dtype, d_bias, bias_host,
                        scale_torque_w, 1.0, stream);
    }

int mtd_wrapper_scale_forces(unsigned int n_particles, void *d_force, void *d_torque, void *d_virial,
                             unsigned int virial_pitch, int dtype, const double *d_bias, double bias_host,
                             int scale_torque_w, mtd_stream_t stream)
    {
    return scale_arrays(n_particles, d_force, d_torque, d_virial, virial_pitch, dtype, d_bias, bias_host, scale_torque_w,
                        0.0, stream);
    }

} // extern "C"
