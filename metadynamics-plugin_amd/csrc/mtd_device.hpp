// mtd_device.hpp — shared host/device helpers for libmtd_hip (gfx950 / CDNA4 only, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mtd_abi.h"

#define MTD_WAVE 64

#define MTD_HIP_TRY(expr)                         \
    do                                            \
        {                                         \
        hipError_t _e = (expr);                   \
        if (_e != hipSuccess) return (int)_e;     \
        } while (0)

#define MTD_LAUNCH_CHECK()                        \
    do                                            \
        {                                         \
        hipError_t _e = hipGetLastError();        \
        if (_e != hipSuccess) return (int)_e;     \
        } while (0)

namespace mtd
{

// ---- wave64 reductions (fixed butterfly order => bitwise reproducible) -------------------------

__device__ __forceinline__ double wave_sum(double v)
    {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, MTD_WAVE);
    return v;
    }

__device__ __forceinline__ float wave_sum(float v)
    {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, MTD_WAVE);
    return v;
    }

// Block barrier that orders LDS traffic only: __syncthreads() also waits for every global load and STORE of the wave
// (s_waitcnt vmcnt(0)) — a microsecond or more behind a batch of stores, and the end of any prefetch in flight.  Use where the
// barrier publishes LDS data and nothing that went to global memory is read back by the block.
__device__ __forceinline__ void lds_barrier()
    {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

// Block sum of one double per thread; result valid in EVERY thread. blockDim.x multiple of 64, <= 1024.
// s_red must hold >= 16 doubles.
__device__ __forceinline__ double block_sum(double v, double *s_red)
    {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n_waves = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads(); // protect s_red from a previous use
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < n_waves; ++w) r += s_red[w];
    return r;
    }

// the same with LDS-only barriers (same order of the adds: same bits)
__device__ __forceinline__ double block_sum_lds(double v, double *s_red)
    {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n_waves = blockDim.x >> 6;
    v = wave_sum(v);
    lds_barrier();
    if (lane == 0) s_red[wave] = v;
    lds_barrier();
    double r = 0.0;
    for (int w = 0; w < n_waves; ++w) r += s_red[w];
    return r;
    }

// ---- non-temporal stores ---------------------------------------------------------------------------
// An array that a kernel writes once and nothing in the L2s is waiting for (force arrays: not re-read by this library) is
// streamed out while the kernel runs; left dirty in the L2s it is written back when the kernel ends, on its tail (measured:
// -1.9 us per step for the 32 MB of the headline force pass, 17.9 -> 15.7 us for the mesh's fused z pass).  Output that the
// NEXT kernel reads is better left where it is (mesh.hip: the L2s are not emptied between the launches of a stream).
typedef double mtd_v2d __attribute__((ext_vector_type(2)));
typedef float mtd_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(const float4 v, float4 *p)
    {
    const mtd_v4f x = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(x, (mtd_v4f *)p);
    }
__device__ __forceinline__ void nt_store(const double4 v, double4 *p)
    {
    const mtd_v2d x = { v.x, v.y }, y = { v.z, v.w };
    __builtin_nontemporal_store(x, (mtd_v2d *)p);
    __builtin_nontemporal_store(y, (mtd_v2d *)p + 1);
    }
__device__ __forceinline__ void nt_store(const double2 v, double2 *p)
    {
    const mtd_v2d x = { v.x, v.y };
    __builtin_nontemporal_store(x, (mtd_v2d *)p);
    }
__device__ __forceinline__ void nt_store(const double v, double *p) { __builtin_nontemporal_store(v, p); }

// ---- particle loads: Scalar4 with the type id bit-cast into w -----------------------------------

struct Particle
    {
    double x, y, z;
    int type;
    };

template<typename S4> struct scalar4_traits;

template<> struct scalar4_traits<float4>
    {
    typedef float scalar;
    static __device__ __forceinline__ Particle unpack(const float4 v)
        {
        Particle r;
        r.x = v.x; r.y = v.y; r.z = v.z;
        r.type = __float_as_int(v.w);
        return r;
        }
    static __device__ __forceinline__ Particle load(const float4 *p, unsigned int i) { return unpack(p[i]); }
    static __device__ __forceinline__ float4 make(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
    };

template<> struct scalar4_traits<double4>
    {
    typedef double scalar;
    static __device__ __forceinline__ Particle unpack(const double4 v)
        {
        Particle r;
        r.x = v.x; r.y = v.y; r.z = v.z;
        r.type = __double2loint(v.w); // HOOMD __scalar_as_int for double builds: low word
        return r;
        }
    static __device__ __forceinline__ Particle load(const double4 *p, unsigned int i) { return unpack(p[i]); }
    static __device__ __forceinline__ double4 make(double x, double y, double z, double w) { return make_double4(x, y, z, w); }
    };

// reciprocal lattice WITHOUT the 2*pi: rows b_i' with b_i' . a_j = delta_ij
// (LamellarOrderParameter.cc:94-102 divided by 2*pi; BoxDim::getLatticeVector per SURVEY App. B)
inline void reciprocal_rows(const mtd_box &box, double B[3][3])
    {
    const double a1[3] = { box.L[0], 0.0, 0.0 };
    const double a2[3] = { box.xy * box.L[1], box.L[1], 0.0 };
    const double a3[3] = { box.xz * box.L[2], box.yz * box.L[2], box.L[2] };
    const double V = box.L[0] * box.L[1] * box.L[2];
    B[0][0] = (a2[1] * a3[2] - a2[2] * a3[1]) / V;
    B[0][1] = (a2[2] * a3[0] - a2[0] * a3[2]) / V;
    B[0][2] = (a2[0] * a3[1] - a2[1] * a3[0]) / V;
    B[1][0] = (a3[1] * a1[2] - a3[2] * a1[1]) / V;
    B[1][1] = (a3[2] * a1[0] - a3[0] * a1[2]) / V;
    B[1][2] = (a3[0] * a1[1] - a3[1] * a1[0]) / V;
    B[2][0] = (a1[1] * a2[2] - a1[2] * a2[1]) / V;
    B[2][1] = (a1[2] * a2[0] - a1[0] * a2[2]) / V;
    B[2][2] = (a1[0] * a2[1] - a1[1] * a2[0]) / V;
    }

} // namespace mtd
