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
// The sum of the xor butterfly with ASCENDING offsets (1, 2, 4, 8, 16, 32), without its twelve dependent trips through the LDS crossbar
// (__shfl_xor is ds_bpermute_b32: ~100 cycles each, two per double — a chain of wave sums was most of the latency of
// chain_wave): the four stages inside a row of 16 lanes are DPP moves (quad_perm for 1 and 2; after those a quad is uniform,
// so row_half_mirror pairs each quad with the other quad of its half row exactly as xor 4 does, and row_mirror the half rows as
// xor 8), the two stages across rows are v_readlane of the four row sums added as (r0 + r1) + (r2 + r3) — the same pairs in
// the same grouping as the butterfly, addition being commutative: the SAME BITS in every lane.
// Every lane of the wave must take part (EXEC all ones): a lane that is switched off contributes garbage, not zero.
#define MTD_DPP_QUAD_XOR1 0xB1        /* quad_perm:[1,0,3,2] */
#define MTD_DPP_QUAD_XOR2 0x4E        /* quad_perm:[2,3,0,1] */
#define MTD_DPP_ROW_HALF_MIRROR 0x141
#define MTD_DPP_ROW_MIRROR 0x140

template<int CTRL> __device__ __forceinline__ float dpp_move(const float v)
    {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
    }

template<int CTRL> __device__ __forceinline__ double dpp_move(const double v)
    {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
    }

// value of lane `lane` (wave-uniform index) in every lane: v_readlane_b32, no LDS trip
__device__ __forceinline__ float wave_read(const float v, const int lane)
    {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
    }

__device__ __forceinline__ double wave_read(const double v, const int lane)
    {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
    }

__device__ __forceinline__ int wave_read(const int v, const int lane) { return __builtin_amdgcn_readlane(v, lane); }

template<typename T> __device__ __forceinline__ T wave_sum_dpp(T v)
    {
    v += dpp_move<MTD_DPP_QUAD_XOR1>(v);
    v += dpp_move<MTD_DPP_QUAD_XOR2>(v);
    v += dpp_move<MTD_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_move<MTD_DPP_ROW_MIRROR>(v);
    const T r0 = wave_read(v, 0), r1 = wave_read(v, 16), r2 = wave_read(v, 32), r3 = wave_read(v, 48);
    return (r0 + r1) + (r2 + r3);
    }

__device__ __forceinline__ double wave_sum(double v) { return wave_sum_dpp(v); }
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }

// the butterfly itself (reference form of the above; tools/probe_wave_sum.hip compares the two bit for bit).  Rounds 1 and 2 of
// this library ran it with descending offsets (32 ... 1): the same sum in another grouping, so sums of this build differ from
// theirs in the last bits — every rank and every path of one build uses the same one.
template<typename T> __device__ __forceinline__ T wave_sum_butterfly(T v)
    {
#pragma unroll
    for (int off = 1; off < MTD_WAVE; off <<= 1)
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
