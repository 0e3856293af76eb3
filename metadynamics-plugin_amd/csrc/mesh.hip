// mesh.hip — particle-mesh order parameter (OrderParameterMesh) on gfx950.
//
// Reference (CPU path results must match): OrderParameterMesh.cc:517-640 (assignParticles, TSC),
// :642-747 (updateMeshes), :866-923 (computeCV), :749-864 (interpolateForces), :344-453
// (computeInfluenceFunction incl. the unsigned-division interpolation function, Q6).  CUDA design
// replaced: OrderParameterMeshGPU.cu — atomicInc binning with overflow re-runs (:90-152), a 27 x M
// float scratch (226 MB at 128^3) with memset + reduce (:179-364), cuFFT C2C, separate update / CV /
// final-reduce launches (:510-541, :771-885), texture-bound force gather (:566-754).
//
// MI355X design: single rank, no ghost cells (the multi-GPU plan replicates the mesh).
// Assignment and force pass run by TILES (steps 1b-5b, 9b further down: particles grouped by tile of 64x8x8 cells, weights summed
// in LDS as 64-bit fixed point, Re(inv) read from an LDS image); the cell-level steps 1-5 and 9 listed here remain for
// meshes with more than 8192 tiles and behind MTD_MESH_ASSIGN=cells.
//   1 k_mesh_bin        cell of every particle and its arrival slot in that cell (ONE returning atomic per particle),
//                       block sums of mode^2
//   2 scan (2 kernels)  exclusive scan of the counts -> cell starts; the counters are cleared for the next call
//   3 k_mesh_place      (shift, mode) record, id and cell of every particle written to start[cell] + slot
//   4 k_mesh_sortfix    cells holding >= 2 particles: records ordered by particle id (=> bitwise reproducible sums)
//   5 k_mesh_gather     a block owns a 16x8x4 tile of mesh cells: the records of the tile + one halo layer are staged in
//                       LDS, then every thread sums the TSC weights of the particles of its cells' 27 neighbour cells from
//                       LDS (the Poisson-distributed cell counts make this loop divergent — in LDS that costs ALU slots,
//                       from global memory it cost 316 us of load latency): no atomics on the mesh, no scratch, every
//                       mesh cell written exactly once
//   6 k_fft_xy_forward  x and y passes of a mesh plane in one launch (power-of-two planes that fit the LDS: 128^2 does);
//     k_fft_xy_forward_split   the same for 128 x 128 planes straight from the tile images, the x transform split by the parity
//                       of its output between the two blocks of a plane (no line transformed twice; the default at 128^3);
//                       otherwise, and on the slab path:
//     k_fft_x_r2c       unnormalised DFT along x of the real mesh, only k_x = 0 .. nx/2 kept (half spectrum, padded rows);
//     k_fft_lines       along y; lines staged in LDS in [pos][line] layout (adjacent lines per block so strided axes
//                       still move 128-B segments)
//   7 k_fft_z_spectral  z lines: forward transform, f = F/N, Hermitian part of G = f(|f|^2 - I^2 sum mode^2 / 2N^2), block
//                       sums of the CV integrand (stored cell + mirror cell), inverse transform — one staging in LDS
//   8 k_fft_xy_inverse  inverse along y and x of a plane in one launch (k_fft_xy_inverse_split: the y transform split by the
//                       parity of its output rows, the default); otherwise
//     k_fft_lines       inverse along y; k_fft_x_c2r inverse along x (other half of the line by symmetry, real part out)
//   9 k_mesh_forces     per particle: 27 reads of Re(inv) with TSC' x TSC x TSC weights (tile path: k_tile_forces, or
//                       k_tile_forces_chain with the bias-grid engine's launch inside: mtd_mesh_forces_update_bias)
// Everything is double precision: the CV is quartic in the Fourier amplitudes, fp32 meshes cannot hold
// 1e-6 on it.  Mesh sizes: 4 ... 256 per axis (any, direct transform in LDS for lengths that are not powers of two), or a
// power of two up to 1024 (radix-2 stages); other sizes return MTD_ERR_UNSUPPORTED.
#include "mtd_device.hpp"
#include "comm_host.hpp"
#include "exact_div.hpp"
#include "lamellar_host.hpp"
#include "metad_host.hpp"

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

namespace
{

using namespace mtd;

// Non-temporal stores (nt_store, mtd_device.hpp) in this file: they pay for the transforms' outputs (fused z pass 17.9 -> 15.7 us
// with its 19 MB, the reader 16.7 -> 17.1) and NOT where the next kernel finds the data in the L2s: the records of the scatter
// pass (force pass 26.7 -> 35.1 us), the per-tile buffers (combine 10.8 -> 13.7), the count kernel's arrays (place 14.4 ->
// 19.1), and not for the scattered 16-byte force stores (26.7 -> 42.3).

struct MeshGeom
    {
    unsigned int nx, ny, nz, n_cells;
    unsigned int hxp;                          // row pitch of the half-spectrum arrays (k_x = 0 .. nx/2 stored, then padding)
    double lo[3], L[3], xy, xz, yz;            // box (local == global: single rank)
    ExactDivisor dL[3], dn[3];                 // the box lengths and mesh dimensions as divisors of locate (exact_div.hpp)
    double binv[3][3];                    // reciprocal rows without 2 pi (force pass, :761-769)
    };

__device__ __forceinline__ double tsc(double x)                       // :457-468
    {
    const double xsq = x * x;
    const double xabs = sqrt(xsq);
    if (xsq <= 1.0 / 4.0) return 3.0 / 4.0 - xsq;
    if (xsq <= 9.0 / 4.0) return 1.0 / 2.0 * (3.0 / 2.0 - xabs) * (3.0 / 2.0 - xabs);
    return 0.0;
    }

// TSC weight of a particle with in-cell shift s (mesh units, |s| <= 1/2) on the cell at offset i in {-1, 0, 1}:
// tsc(s - i) without the branches: 3/4 - s^2 (i = 0), (1/2 + s)^2 / 2 (i = +1), (1/2 - s)^2 / 2 (i = -1).
// Valid for |s| <= 1/2 only: the gather checks the staged records once per tile and takes the general tsc() for a tile
// that holds a particle outside the box (clamped cell, large shift) — per record visit that check cost a quarter of the loop.
__device__ __forceinline__ double tsc_cell(const double s, const int i)
    {
    if (i == 0) return 0.75 - s * s;
    const double t = 0.5 + (i > 0 ? s : -s);
    return 0.5 * t * t;
    }

__device__ __forceinline__ double tsc_deriv(double x)                 // :470-483 (copysignf: float |x|, Q9)
    {
    const double xsq = x * x;
    const double xabs = (double)fabsf((float)x);
    const double fac = 3.0 / 2.0 - xabs;
    double ret = 0.0;
    if (xsq <= 1.0 / 4.0)
        ret = -2.0 * x;
    else if (xsq <= 9.0 / 4.0)
        ret = -fac * x / xabs;
    return ret;
    }

// The three TSC weights (and derivatives) of one axis for the stencil offsets -1, 0, +1, i.e. tsc(s + 1), tsc(s), tsc(s - 1).
// In-box particles have |s| <= 1/2 and take the closed forms — no sqrt, no branches, no fp64 division (the general forms
// cost ~90 fp64 instructions per weight/derivative pair and made the force pass compute bound); anything else (a particle
// outside the box: clamped cell, large shift) takes the general forms.  The closed derivative keeps Q9: |x| rounded to
// float in fac and in the quotient x / |x|, the latter as 1 + (|x| - float|x|) / float|x| with a float reciprocal
// (the correction is ~6e-8, so its own relative error of 1e-7 is invisible in double).
// exactly |s| <= 1/2: a hair beyond it the closed forms are off by O(|s| - 1/2) in the derivative — harmless by itself, but
// the z (or x, y) force sums differences of neighbouring mesh rows, and on a smooth field that cancellation turned 3e-7
// into 3e-4 of the component for a particle sitting on a cell boundary (found by tools/fuzz_mesh.py)
__device__ __forceinline__ bool tsc_inbox(const double s) { return fabs(s) <= 0.5; }

__device__ __forceinline__ void tsc3(const double s, double (&w)[3])
    {
    if (tsc_inbox(s))
        {
        const double tm = 0.5 - s, tp = 0.5 + s;
        w[0] = 0.5 * tm * tm;
        w[1] = 0.75 - s * s;
        w[2] = 0.5 * tp * tp;
        }
    else
        {
        w[0] = tsc(s + 1.0); w[1] = tsc(s); w[2] = tsc(s - 1.0);
        }
    }

__device__ __forceinline__ double tsc_deriv_outer(const double xabs)      // |x| in [1/2, 3/2]: -fac * |x| / float|x|
    {
    const float xf = (float)xabs;
    const double xa = (double)xf;
    // hardware reciprocal (1 ulp): it multiplies a correction of 6e-8; the correctly rounded one is a ten-instruction sequence
    const double ratio = 1.0 + (xabs - xa) * (double)__builtin_amdgcn_rcpf(xf);
    return -(1.5 - xa) * ratio;
    }

__device__ __forceinline__ void tsc3_deriv(const double s, double (&w)[3], double (&d)[3])
    {
    if (tsc_inbox(s))
        {
        const double tm = 0.5 - s, tp = 0.5 + s;
        w[0] = 0.5 * tm * tm;
        w[1] = 0.75 - s * s;
        w[2] = 0.5 * tp * tp;
        d[0] = tsc_deriv_outer(1.0 + s);           // x = s + 1 > 0
        d[1] = -2.0 * s;
        d[2] = -tsc_deriv_outer(1.0 - s);          // x = s - 1 < 0
        }
    else
        {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            {
            w[i] = tsc(s - (i - 1));
            d[i] = tsc_deriv(s - (i - 1));
            }
        }
    }

// BoxDim::makeFraction / makeCoordinates / minImage (HOOMD-blue v2 semantics).
// Cell and in-cell shift follow the reference's operation order EXACTLY (:540-573 == :784-812: makeFraction, truncation,
// makeCoordinates of the cell centre, minImage, makeFraction again) — true divisions, one rounding per operation, no fused
// multiply-add (this translation unit is compiled with -ffp-contract=on: csrc/Makefile explains why =fast ignored the pragma
// below); the divisions by the loop-invariant box lengths and mesh dimensions are exact_div's correctly rounded quotients.  The shift feeds assignTSCderiv, which rounds |x| to
// FLOAT (Q9): a shift that differs from the reference's in its last bit can fall on the other side of a float rounding
// boundary, the derivative weight jumps by 6e-8 and, through the cancelling row differences of the force, one particle
// moved by up to 1e-3 of max|F| (round-1 verdict).  With the same operations in the same order the shift is the same
// double, bit for bit, and every tie resolves as in the reference.
__device__ __forceinline__ void make_fraction(const MeshGeom &g, double x, double y, double z, double &fx, double &fy, double &fz)
    {
#pragma clang fp contract(off)
    double dx = x - g.lo[0], dy = y - g.lo[1], dz = z - g.lo[2];
    dx -= (g.xz - g.yz * g.xy) * dz + g.xy * dy;
    dy -= g.yz * dz;
    fx = exact_div(dx, g.dL[0]);               // == dx / L, correctly rounded (exact_div.hpp)
    fy = exact_div(dy, g.dL[1]);
    fz = exact_div(dz, g.dL[2]);
    }

// cell (ix,iy,iz) and in-cell shift (mesh units) of a particle — :540-573 == :784-812
__device__ __forceinline__ void locate(const MeshGeom &g, const Particle &p, int &ix, int &iy, int &iz, double &sx, double &sy,
                                       double &sz)
    {
#pragma clang fp contract(off)
    double fx, fy, fz;
    make_fraction(g, p.x, p.y, p.z, fx, fy, fz);
    ix = (int)(fx * (double)g.nx);
    iy = (int)(fy * (double)g.ny);
    iz = (int)(fz * (double)g.nz);
    if (ix == (int)g.nx) ix = 0;
    if (iy == (int)g.ny) iy = 0;
    if (iz == (int)g.nz) iz = 0;
    // keep out-of-box particles from indexing outside the mesh (the reference asserts, :588-608)
    ix = min(max(ix, 0), (int)g.nx - 1);
    iy = min(max(iy, 0), (int)g.ny - 1);
    iz = min(max(iz, 0), (int)g.nz - 1);
    const double cfx = exact_div((double)ix + 0.5, g.dn[0]), cfy = exact_div((double)iy + 0.5, g.dn[1]), cfz = exact_div((double)iz + 0.5, g.dn[2]);
    // makeCoordinates(cell centre)
    const double cx = g.lo[0] + cfx * g.L[0] + cfy * g.xy * g.L[1] + cfz * g.xz * g.L[2];
    const double cy = g.lo[1] + cfy * g.L[1] + cfz * g.yz * g.L[2];
    const double cz = g.lo[2] + cfz * g.L[2];
    double wx = p.x - cx, wy = p.y - cy, wz = p.z - cz;
    // minImage
    double img = rint(exact_div(wz, g.dL[2]));
    wz -= g.L[2] * img;
    wy -= g.L[2] * g.yz * img;
    wx -= g.L[2] * g.xz * img;
    img = rint(exact_div(wy, g.dL[1]));
    wy -= g.L[1] * img;
    wx -= g.L[1] * g.xy * img;
    wx -= g.L[0] * rint(exact_div(wx, g.dL[0]));
    double sfx, sfy, sfz;
    make_fraction(g, wx + g.lo[0], wy + g.lo[1], wz + g.lo[2], sfx, sfy, sfz);
    sx = sfx * (double)g.nx;
    sy = sfy * (double)g.ny;
    sz = sfz * (double)g.nz;
    }

__device__ __forceinline__ int wrap(int i, int n)
    {
    if (i == n) return 0;
    if (i < 0) return i + n;
    return i;
    }

// ---- 1. cell ids, arrival slots, sum of mode^2 ----------------------------------------------------
template<typename S4>
__global__ __launch_bounds__(256) void k_mesh_bin(const MeshGeom g, const S4 *__restrict__ postype, const unsigned int N,
                                                  const double *__restrict__ mode, unsigned int *__restrict__ cell_of,
                                                  unsigned int *__restrict__ slot_of, unsigned int *__restrict__ count,
                                                  double *__restrict__ modesq_partials)
    {
    __shared__ double s_red[16];
    double msq = 0.0;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        {
        const Particle p = scalar4_traits<S4>::load(postype, i);
        int ix, iy, iz;
        double sx, sy, sz;
        locate(g, p, ix, iy, iz, sx, sy, sz);
        const unsigned int c = ix + g.nx * (iy + g.ny * iz);
        cell_of[i] = c;
        slot_of[i] = atomicAdd(&count[c], 1u);
        const double a = mode[p.type];
        msq += a * a;
        }
    msq = block_sum(msq, s_red);
    if (threadIdx.x == 0) modesq_partials[blockIdx.x] = msq;
    }

// ---- 2. exclusive scan of the counts (three tiny kernels; 1024 cells per block) ------------------
constexpr unsigned int SCAN_TILE = 1024;

__global__ __launch_bounds__(256) void k_scan_tiles(const unsigned int *__restrict__ in, unsigned int *__restrict__ out,
                                                    unsigned int *__restrict__ tile_sums, const unsigned int n,
                                                    const double *__restrict__ modesq_partials, const unsigned int n_partials,
                                                    double *__restrict__ mode_sq)
    {
    __shared__ unsigned int s_wave[4];
    if (blockIdx.x * SCAN_TILE >= n)
        {
        // one extra block rides along: m_mode_sq (:622) = the block sums of k_mesh_bin added up in a fixed order — a launch
        // of its own cost ~5 us of pure latency in the step's chain of small kernels
        __shared__ double s_red[16];
        double v = 0.0;
        for (unsigned int b = threadIdx.x; b < n_partials; b += 256) v += modesq_partials[b];
        v = block_sum(v, s_red);
        if (threadIdx.x == 0) *mode_sq = v;
        return;
        }
    const unsigned int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
    unsigned int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = base + j < n ? in[base + j] : 0u;
    const unsigned int t = v[0] + v[1] + v[2] + v[3];
    // inclusive scan across the wave, then across the 4 waves
    unsigned int incl = t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
        const unsigned int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
        }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
    unsigned int excl = wave_off + incl - t;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        {
        if (base + j < n) out[base + j] = excl;
        excl += v[j];
        }
    if (threadIdx.x == 255) tile_sums[blockIdx.x] = wave_off + incl;
    }

// second level: every block sums the tile totals in front of its tile itself (n_tiles <= 2^20 values, L2 resident),
// adds the offset, and clears the counters it covers for the next call
__global__ __launch_bounds__(256) void k_scan_finish(unsigned int *__restrict__ out, const unsigned int *__restrict__ tile_sums,
                                                     unsigned int *__restrict__ count, const unsigned int n, const unsigned int total)
    {
    // one block per tile of SCAN_TILE cells (four per thread): the prefix of the tile totals is summed once per tile
    __shared__ unsigned int s_wave[4];
    const unsigned int tile = blockIdx.x;
    unsigned int v = 0;
    for (unsigned int t = threadIdx.x; t < tile; t += blockDim.x) v += tile_sums[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = v;
    __syncthreads();
    const unsigned int prefix = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        {
        const unsigned int i = tile * SCAN_TILE + j * 256 + threadIdx.x;
        if (i < n)
            {
            out[i] += prefix;
            count[i] = 0;
            }
        }
    if (tile == 0 && threadIdx.x == 0) out[n] = total;
    }

// ---- 3. place: record, id and cell of every particle at start[cell] + slot ----------------------------
template<typename S4>
__global__ __launch_bounds__(256) void k_mesh_place(const MeshGeom g, const S4 *__restrict__ postype, const unsigned int N,
                                                    const double *__restrict__ mode, const unsigned int *__restrict__ cell_of,
                                                    const unsigned int *__restrict__ slot_of, const unsigned int *__restrict__ start,
                                                    uint2 *__restrict__ idcell, double4 *__restrict__ packed)
    {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        {
        const Particle p = scalar4_traits<S4>::load(postype, i);
        int ix, iy, iz;
        double sx, sy, sz;
        locate(g, p, ix, iy, iz, sx, sy, sz);
        const unsigned int c = cell_of[i];
        const unsigned int dst = start[c] + slot_of[i];
        packed[dst] = make_double4(sx, sy, sz, mode[p.type]);
        idcell[dst] = make_uint2(i, c);                 // one 8-byte scattered store instead of two 4-byte ones
        }
    }

// ---- 4. cells with >= 2 particles: order by particle id (the arrival order of step 1 is not reproducible) ----
__global__ __launch_bounds__(256) void k_mesh_sortfix(const unsigned int n_cells, const unsigned int *__restrict__ start,
                                                      uint2 *__restrict__ idcell, double4 *__restrict__ packed)
    {
    const unsigned int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const unsigned int b = start[c], e = start[c + 1];
    for (unsigned int i = b + 1; i < e; ++i)          // insertion sort: cells hold O(1) particles
        {
        const uint2 key = idcell[i];
        const double4 rec = packed[i];
        unsigned int j = i;
        while (j > b && idcell[j - 1].x > key.x)
            {
            idcell[j] = idcell[j - 1];
            packed[j] = packed[j - 1];
            --j;
            }
        idcell[j] = key;
        packed[j] = rec;
        }
    }

// ---- 5. gather: mesh[c] = sum over the particles of the 27 neighbour cells, tile by tile from LDS ----------
constexpr int GT_X = 16, GT_Y = 8, GT_Z = 4;                       // output tile (clipped to the mesh)
constexpr int GT_HCELLS = (GT_X + 2) * (GT_Y + 2) * (GT_Z + 2);     // with one halo layer: 1080
constexpr int GT_CAP = 832;                                         // records staged in LDS (26 KB); denser tiles read global memory
constexpr int GT_CPT = (GT_HCELLS + GT_X * GT_Y * GT_Z - 1) / (GT_X * GT_Y * GT_Z);   // halo'd cells per thread in the staging

struct GatherTiling
    {
    unsigned int tx, ty, tz;          // tile dimensions
    unsigned int ntx, nty, ntz;       // tiles per axis
    };

constexpr int GT_THREADS = GT_X * GT_Y * GT_Z;                      // one thread per output cell: 8 waves per block, four blocks per CU hide the LDS latency
                                                                    // of the divergent per-cell loops

__global__ __launch_bounds__(GT_THREADS) void k_mesh_gather(const MeshGeom g, const GatherTiling tl, const unsigned int *__restrict__ start,
                                                            const double4 *__restrict__ packed, double *__restrict__ rho)
    {
    __shared__ unsigned int s_off[GT_HCELLS + 1];      // first staged record of every halo'd cell
    __shared__ unsigned int s_gb[GT_HCELLS];           // global begin of the cell's records
    __shared__ unsigned int s_wave[GT_THREADS / 64];
    __shared__ double4 s_rec[GT_CAP];
    __shared__ int s_general;                          // the tile holds a record with |shift| > 1/2: general tsc() for this tile

    if (threadIdx.x == 0) s_general = 0;
    const unsigned int t_id = blockIdx.x;
    const unsigned int tix = t_id % tl.ntx, tiy = (t_id / tl.ntx) % tl.nty, tiz = t_id / (tl.ntx * tl.nty);
    const int x0 = tix * tl.tx, y0 = tiy * tl.ty, z0 = tiz * tl.tz;
    const unsigned int HX = tl.tx + 2, HY = tl.ty + 2, HZ = tl.tz + 2;
    const unsigned int HC = HX * HY * HZ;

    // counts of the halo'd cells; thread t owns GT_CPT consecutive halo'd cells
    unsigned int cnt[GT_CPT], tsum = 0;
#pragma unroll
    for (int q = 0; q < GT_CPT; ++q)
        {
        const unsigned int hc = threadIdx.x * GT_CPT + q;
        cnt[q] = 0;
        if (hc < HC)
            {
            const int hx = hc % HX, hy = (hc / HX) % HY, hz = hc / (HX * HY);
            int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
            gx = gx < 0 ? gx + (int)g.nx : (gx >= (int)g.nx ? gx - (int)g.nx : gx);
            gy = gy < 0 ? gy + (int)g.ny : (gy >= (int)g.ny ? gy - (int)g.ny : gy);
            gz = gz < 0 ? gz + (int)g.nz : (gz >= (int)g.nz ? gz - (int)g.nz : gz);
            const unsigned int gc = gx + g.nx * (gy + g.ny * gz);
            const unsigned int b = start[gc];
            cnt[q] = start[gc + 1] - b;
            s_gb[hc] = b;
            }
        tsum += cnt[q];
        }
    // block exclusive scan of the per-thread sums
    unsigned int incl = tsum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
        const unsigned int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
        }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned int excl = incl - tsum, total = 0;
#pragma unroll
    for (int w = 0; w < GT_THREADS / 64; ++w)
        {
        if (w < wave) excl += s_wave[w];
        total += s_wave[w];
        }
    const bool in_lds = total <= (unsigned int)GT_CAP;           // block-uniform
#pragma unroll
    for (int q = 0; q < GT_CPT; ++q)
        {
        const unsigned int hc = threadIdx.x * GT_CPT + q;
        if (hc < HC)
            {
            s_off[hc] = excl;
            excl += cnt[q];
            }
        }
    if (threadIdx.x == 0) s_off[HC] = total;
    __syncthreads();
    if (in_lds)
        {
        // flat copy: staged record q belongs to the halo'd cell found by bisection over s_off
        for (unsigned int q = threadIdx.x; q < total; q += GT_THREADS)
            {
            unsigned int lo = 0, hi = HC;                        // largest lo with s_off[lo] <= q
            while (hi - lo > 1)
                {
                const unsigned int mid = (lo + hi) >> 1;
                if (s_off[mid] <= q)
                    lo = mid;
                else
                    hi = mid;
                }
            const double4 rec = packed[s_gb[lo] + (q - s_off[lo])];
            s_rec[q] = rec;
            if (!(fabs(rec.x) <= 0.5000001 && fabs(rec.y) <= 0.5000001 && fabs(rec.z) <= 0.5000001)) s_general = 1;
            }
        __syncthreads();
        }
    const bool fast = in_lds && !s_general;                      // block-uniform

    const unsigned int lx = threadIdx.x % GT_X, ly = (threadIdx.x / GT_X) % GT_Y, lz = threadIdx.x / (GT_X * GT_Y);
    if (lx >= tl.tx || ly >= tl.ty || lz >= tl.tz) return;
    if (x0 + lx >= g.nx || y0 + ly >= g.ny || z0 + lz >= g.nz) return;       // partial tile at the edge of the mesh
    double acc = 0.0;
    // this cell receives from the particle cell at offset (-i,-j,-k) with dx = shift - (i,j,k); fixed loop order
#pragma unroll
    for (int k = -1; k <= 1; ++k)
#pragma unroll
        for (int j = -1; j <= 1; ++j)
            {
            // halo'd coordinates of the source row: (ly + 1 - j, lz + 1 - k); its x cells lx .. lx+2 (= i = +1, 0, -1) sit next to
            // each other in the staged order
            const unsigned int h0 = HX * ((ly + 1 - j) + HY * (lz + 1 - k)) + lx;
            const unsigned int b0 = s_off[h0], b1 = s_off[h0 + 1], b2 = s_off[h0 + 2], b3 = s_off[h0 + 3];
            if (fast)
                {
                for (unsigned int q = b0; q < b3; ++q)
                    {
                    const double4 pk = s_rec[q];
                    const int i = q < b1 ? 1 : (q < b2 ? 0 : -1);
                    acc += pk.w * (tsc_cell(pk.x, i) * tsc_cell(pk.y, j) * tsc_cell(pk.z, k));
                    }
                }
            else
                {
                // dense tile (records straight from global memory; the three cells need not be adjacent there when the row
                // wraps in x) or a tile with an out-of-box particle: same order, general TSC weights
                for (int i = 1; i >= -1; --i)
                    {
                    const unsigned int h = h0 + (1 - i);
                    const unsigned int n = s_off[h + 1] - s_off[h];
                    for (unsigned int r = 0; r < n; ++r)
                        {
                        const double4 pk = in_lds ? s_rec[s_off[h] + r] : packed[s_gb[h] + r];
                        acc += pk.w * (tsc(pk.x - i) * tsc(pk.y - j) * tsc(pk.z - k));
                        }
                    }
                }
            }
    rho[(x0 + lx) + g.nx * ((y0 + ly) + g.ny * (z0 + lz))] = acc;
    }

// ---- 1b-5b. tile path of the assignment and of the force pass ------------------------------------------------------
// The cell-level pipeline above pays for a sort down to single cells: one returning global atomic, two scattered record
// stores and a per-cell gather loop per particle (bin 42 + scan 15 + place 40 + sortfix 18 + gather 78 us at 10^6 particles
// on 128^3).  Here the particles are only grouped by TILE of 64x8x8 cells (16x16x8 until round 3) and the TSC weights are summed in LDS:
//   1b k_tile_count    a block owns a contiguous chunk of particles: tile of every particle, its arrival slot in the
//                      block's LDS histogram (LDS atomic), the histogram written as one row per block, [block][tile]; sum of mode^2
//   2b scan            exclusive scan over the blocks of every tile's counts = where each block's particles of each tile go
//   3b k_tile_place_sorted  a block sorts its chunk by tile in LDS and stores raw position records + ids into tile order in runs
//                      (k_tile_place: one scattered 4-byte id per particle, the fallback when chunk or tables do not fit the LDS)
//   1c k_tile_bin      STEADY STATE (every assignment of a mesh but its first): 1b-3b in one launch on tile segments with slack
//                      planned from the previous snapshot's exact counts, runs reserved with one returning atomic per (chunk, tile),
//                      overflow list for what does not fit; a mixed set's lamellar sums ride in its wait for the atomics
//   4b k_tile_scatter  a block owns a tile: the tile + one halo layer live in LDS as 64-bit FIXED-POINT sums, every particle
//                      of the tile (tile-ordered position records, read coalesced) adds its 27 weights with LDS atomics; integer
//                      sums do not depend on the order of the adds, so the mesh is bitwise reproducible without any sorting;
//                      the LDS tile is written to a per-tile buffer; an extra block plans the next snapshot's segments
//   5b k_tile_combine  every mesh cell adds the entries that stand for it in its own tile's buffer and in the halo layers
//                      of the neighbouring tiles (integers, then one conversion to double)
//   9b k_tile_forces   a block owns a tile: Re(inv) of the tile + halo staged in LDS, 27 LDS reads per particle
// Fixed point: an entry is sum(a * W W W) * 2^k in int64 with 2^k * N * max|a| < 2^62 (k = 42 at 10^6 particles, |a| = 1:
// resolution 2e-13, against 1e-6 asked of the CV).  Meshes with more than TP_MAX_TILES tiles (> 256^3) keep the cell path.
// Tile shape and block sizes (overridable at build time for experiments, csrc/Makefile EXTRA_HIPFLAGS, tools/exp_mesh.sh).
// Round 3: 512 tiles of 64x8x8 cells at 128^3 instead of 1024 of 16x16x8, 512 threads in the force pass: config 3 161 -> 154 us per
// step (force pass 27.8 -> 22.9 us: rows of 66 doubles straddle fewer 128-byte lines per payload byte than rows of 18, and two
// blocks of eight waves per CU stage and sum as well as four of four; place 14.3 -> 13.3: runs of 8 ids per count block and tile).
// Measured beside it (profiles/r3/tile_pipeline_ab.log): 32x16x8 and 16x16x16 (512 tiles) 154-156, 256 tiles with 1024-thread
// blocks 167, 128x8x4 156, 32x8x8 (1024 tiles) 159.
#ifndef MTD_TP_X
#define MTD_TP_X 64
#endif
#ifndef MTD_TP_Y
#define MTD_TP_Y 8
#endif
#ifndef MTD_TP_Z
#define MTD_TP_Z 8
#endif
#ifndef MTD_TP_THREADS
#define MTD_TP_THREADS 512
#endif
#ifndef MTD_TF_THREADS
#define MTD_TF_THREADS 512
#endif
constexpr int TP_X = MTD_TP_X, TP_Y = MTD_TP_Y, TP_Z = MTD_TP_Z;
constexpr unsigned int TP_MAX_TILES = 8192;                         // LDS histogram of k_tile_count: 32 KB
constexpr int TP_HMAX = (TP_X + 2) * (TP_Y + 2) * (TP_Z + 2);       // 6600 entries, 52.8 KB
constexpr int TP_THREADS = MTD_TP_THREADS;      // three blocks per CU (53 KB of LDS each): every tile of a 128^3 mesh resident at once

struct TileGeom
    {
    unsigned int tx, ty, tz;                    // tile dimensions (clamped to the mesh); edge tiles may be narrower
    unsigned int ntx, nty, ntz, n_tiles;
    unsigned int hx, hy, hz, hcells;            // tile + halo: pitch of the LDS image and of the per-tile buffer
    unsigned int n_blocks, chunk;               // count blocks and particles per count block
    double scale, inv_scale;                    // fixed point 2^k and 2^-k
    };

// Diagnostic build only (-DMTD_STAMPS, tools/stamps_tiles.py): per-block time stamps of the scatter and the force pass
#ifdef MTD_STAMPS
__device__ unsigned long long g_tile_stamps[2][8][1024];
#define TILE_STAMP(k, row) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_tile_stamps[k][row][blockIdx.x] = wall_clock64(); } while (0)
#else
#define TILE_STAMP(k, row) do { } while (0)
#endif
#ifdef MTD_STAMPS
__device__ unsigned long long g_cnt_stamps[8][1024];
#define CNT_STAMP(row) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_cnt_stamps[row][blockIdx.x] = wall_clock64(); } while (0)
#else
#define CNT_STAMP(row) do { } while (0)
#endif

// The mode coefficients of the first TP_MODE_LDS types in LDS: looked up by a type that has just arrived from memory, a global
// table costs every trip of the particle loops another dependent round trip.
constexpr unsigned int TP_MODE_LDS = 32;
__device__ __forceinline__ void stage_modes(double *s_mode, const double *__restrict__ mode, const unsigned int n_types)
    {
    if (threadIdx.x < TP_MODE_LDS) s_mode[threadIdx.x] = threadIdx.x < n_types ? mode[threadIdx.x] : 0.0;
    }
__device__ __forceinline__ double mode_of(const double *s_mode, const double *__restrict__ mode, const unsigned int type)
    {
    return type < TP_MODE_LDS ? s_mode[type] : mode[type];
    }

constexpr int TC_THREADS = 1024;       // one block per CU at 10^6 particles: 16 waves hide the load -> locate -> LDS atomic chain

// RIDER (mtd_mesh_set_lamellar_rider): the block partial sums of a set of lamellar CVs — what the CV blocks of k_fused_cv
// (fused.hip) form — are formed here from the particle the block is binning anyway: one read of the 16 MB position array and
// one launch less per step of a mixed lamellar + mesh set.  The deferred second grid pass of the bias-grid engine's previous
// deposit, which launch A used to carry, rides in the next launch (k_tile_rowscan: latency-bound, nearly empty).  (First form:
// the grid pass as extra 1024-thread blocks of THIS kernel — a second generation behind the counting blocks at one block per
// CU, 163 us per step of config 3 against 159 without riders; with two blocks per CU 158.5 against 158.6: the counting blocks
// keep the vector units busy — locate() is ~110 fp64 instructions per particle — and the grid pass beside them cost what it
// costs as a launch of its own.)
struct CountRider
    {
    mtd::LamKArgs k;
    mtd::MetadCfg cfg;
    double *partials;              // [count block][n_cv]
    unsigned int n_apply;          // blocks of the row-scan launch that run apply_cells (256 cells each)
    };

// one particle's contribution to every CV of the set: the arithmetic of lam_cv_accumulate (lamellar_device.hpp), one particle at
// a time — the same operations on the same operands per particle, only the grouping of the sums differs
template<bool FAST>
__device__ __forceinline__ void lam_cv_particle(const mtd::LamKArgs &a, const mtd::ModeTables &mt, const float *s_coeff, const mtd::Particle &p,
                                                float (&acc)[3])
    {
    float g0, g1, g2;
    mtd::project(a, p, g0, g1, g2);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c < (int)a.n_cv)
            {
            float sum = 0.0f;
            const unsigned int k1 = a.first[c] + a.nact[c];
            for (unsigned int k = a.first[c]; k < k1; ++k)
                {
                const float4 h = mt.h[k];
                const float cs = mtd::cos2pi<FAST>(h.x * g0 + h.y * g1 + h.z * g2);
                sum += cs;
                sum += h.w * ((cs * cs) * 2.0f - (FAST ? 0.99999994f : 1.0f));
                }
            acc[c] += s_coeff[c * MTD_MAX_TYPES + p.type] * sum;
            }
    }

template<typename S4, bool RIDER, bool FAST>
__global__ __launch_bounds__(TC_THREADS) void k_tile_count(const MeshGeom g, const TileGeom tg, const S4 *__restrict__ postype, const unsigned int N,
                                                    const double *__restrict__ mode, unsigned int *__restrict__ tile_of,
                                                    unsigned int *__restrict__ slot_of, unsigned int *__restrict__ hist,
                                                    double *__restrict__ modesq_partials, const unsigned int n_types,
                                                    const CountRider *__restrict__ rider)
    {
    extern __shared__ unsigned int s_hist[];
    __shared__ double s_red[16];
    __shared__ double s_mode[TP_MODE_LDS];
    __shared__ float s_coeff[RIDER ? 3 * MTD_MAX_TYPES : 1];
    __shared__ mtd::ModeTables s_mt;
    __shared__ double s_wave[RIDER ? (TC_THREADS / 64) * 3 : 1];
    const unsigned int i0 = blockIdx.x * tg.chunk;
    const unsigned int i1 = min(N, i0 + tg.chunk);
    // the mode coefficient comes from LDS — as a second, dependent global load it doubled the trip
    // the thread's next particle is in flight (raw, from a clamped index, no branch: see k_tile_scatter) while this one is
    // located.  (All four positions of a thread requested up front measured slower again, 10.1 against 9.1 us.)
    const unsigned int i_last = i1 ? i1 - 1 : 0u;
    CNT_STAMP(0);
    S4 raw = scalar4_traits<S4>::make(0, 0, 0, 0);
    if (i0 < i1) raw = postype[min(i0 + threadIdx.x, i_last)];     // (uniform over the block)
    stage_modes(s_mode, mode, n_types);
    if (RIDER)
        {
        const mtd::LamKArgs &a = rider->k;
        for (unsigned int q = threadIdx.x; q < 3 * MTD_MAX_TYPES; q += blockDim.x) s_coeff[q] = a.coeff[q / MTD_MAX_TYPES][q % MTD_MAX_TYPES];
        for (unsigned int q = threadIdx.x; q < a.n_modes; q += blockDim.x) s_mt.h[q] = a.h[a.corder[q]];
        }
    for (unsigned int t = threadIdx.x; t < tg.n_tiles; t += blockDim.x) s_hist[t] = 0;
    __syncthreads();
    CNT_STAMP(1);
    double msq = 0.0;
    float acc[3] = { 0.0f, 0.0f, 0.0f };
    for (unsigned int i = i0 + threadIdx.x; i < i1; i += blockDim.x)
        {
        const Particle p = scalar4_traits<S4>::unpack(raw);
        raw = postype[min(i + blockDim.x, i_last)];
        int ix, iy, iz;
        double sx, sy, sz;
        locate(g, p, ix, iy, iz, sx, sy, sz);
        const unsigned int t = (unsigned int)ix / tg.tx + tg.ntx * ((unsigned int)iy / tg.ty + tg.nty * ((unsigned int)iz / tg.tz));
        tile_of[i] = t;
        slot_of[i] = atomicAdd(&s_hist[t], 1u);
        const double a = mode_of(s_mode, mode, (unsigned int)p.type);
        msq += a * a;
        if (RIDER) lam_cv_particle<FAST>(rider->k, s_mt, s_coeff, p, acc);
        }
    CNT_STAMP(2);
    lds_barrier();                                                   // (the tile / slot stores of the loop drain behind it)
    // [block][tile]: one contiguous row per block (tile-major it was 1024 scattered 4-byte stores per block)
    CNT_STAMP(3);
    for (unsigned int t = threadIdx.x; t < tg.n_tiles; t += blockDim.x) hist[(size_t)blockIdx.x * tg.n_tiles + t] = s_hist[t];
    msq = block_sum_lds(msq, s_red);
    if (threadIdx.x == 0) modesq_partials[blockIdx.x] = msq;
    if (RIDER)
        {
        // fp32 wave sums -> fp64 across the waves in a fixed order, as lam_cv_block_reduce
        const unsigned int n_cv = rider->k.n_cv;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            {
            const float v = wave_sum(acc[c]);
            if (lane == 0) s_wave[wave * 3 + c] = (double)v;
            }
        lds_barrier();
        if (threadIdx.x < n_cv)
            {
            double r = 0.0;
            for (int w = 0; w < TC_THREADS / 64; ++w) r += s_wave[w * 3 + threadIdx.x];
            rider->partials[(size_t)blockIdx.x * n_cv + threadIdx.x] = r;
            }
        }
    CNT_STAMP(4);
    }

// 2b. where the particles of (tile, count block) go = [particles of the tiles in front] + [particles of this tile counted by
// the blocks in front].  ONE launch forms the second term (a scan along each tile's column of the [block][tile] histogram, a
// block per tile) and the tiles' totals; the first term — a prefix over <= 8192 totals — is formed by whoever needs it (the
// place kernel once per block in LDS, the scatter / force kernels for their own tile): a second scan launch cost ~5 us of
// pure latency in this chain of small kernels.  One extra block adds up sum mode^2 (:622).
__global__ __launch_bounds__(256) void k_tile_rowscan(const unsigned int *__restrict__ hist, unsigned int *__restrict__ rowscan,
                                                      unsigned int *__restrict__ tile_total, const unsigned int n_tiles, const unsigned int nb,
                                                      const double *__restrict__ modesq_partials, const unsigned int n_partials,
                                                      double *__restrict__ mode_sq, const CountRider *__restrict__ rider)
    {
    __shared__ unsigned int s_wave[4];
    if (blockIdx.x > n_tiles)
        {
        // passenger (mtd_mesh_set_lamellar_rider): the bias-grid engine's deferred second pass, 256 cells per block — this
        // launch is latency-bound and nearly empty, the pass overlaps with it (in the counting kernel, which keeps the vector
        // units busy, the same blocks cost what they cost as a launch of their own)
        __shared__ double s_red2[16];
        const unsigned int b = blockIdx.x - n_tiles - 1;
        const unsigned int c0 = b * 256;
        mtd::apply_cells(rider->cfg, c0, min(rider->cfg.len, c0 + 256u), b == 0, s_red2);
        return;
        }
    if (blockIdx.x >= n_tiles)
        {
        __shared__ double s_red[16];
        double v = 0.0;
        for (unsigned int b = threadIdx.x; b < n_partials; b += 256) v += modesq_partials[b];
        v = block_sum(v, s_red);
        if (threadIdx.x == 0) *mode_sq = v;
        return;
        }
    const size_t row = (size_t)blockIdx.x * nb;
    const unsigned int base = threadIdx.x * 4;                      // nb <= 1024 (tile_blocks_max)
    unsigned int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = base + j < nb ? hist[(size_t)(base + j) * n_tiles + blockIdx.x] : 0u;   // column of the [block][tile] table
    const unsigned int t = v[0] + v[1] + v[2] + v[3];
    unsigned int incl = t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
        const unsigned int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
        }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
    unsigned int excl = wave_off + incl - t;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        {
        if (base + j < nb) rowscan[row + base + j] = excl;
        excl += v[j];
        }
    if (threadIdx.x == 255) tile_total[blockIdx.x] = wave_off + incl;
    }

// 3b. place: particle id -> its tile's segment.  A block keeps the prefix over the tile totals in LDS.
// (This form — one scattered 4-byte store per particle, the scatter pass gathering the positions through the ids — is the fallback
// of k_tile_place_sorted below: chunks of more than 4096 particles or tables that do not fit the LDS.)
template<typename S4>
__global__ __launch_bounds__(256) void k_tile_place(const TileGeom tg, const unsigned int N, const unsigned int *__restrict__ tile_of,
                                                    const unsigned int *__restrict__ slot_of, const unsigned int *__restrict__ rowscan,
                                                    const unsigned int *__restrict__ tile_total, unsigned int *__restrict__ ids,
                                                    unsigned int *__restrict__ tile_first)
    {
    extern __shared__ unsigned int s_first[];                    // [n_tiles]: first slot of every tile
    __shared__ unsigned int s_wsum[4];
    // exclusive scan of the tile totals: thread t owns the tiles [t * per, (t + 1) * per)
    const unsigned int per = (tg.n_tiles + 255) / 256;
    unsigned int mine = 0;
    for (unsigned int j = 0; j < per; ++j)
        {
        const unsigned int t = threadIdx.x * per + j;
        if (t < tg.n_tiles) mine += tile_total[t];
        }
    unsigned int incl = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
        const unsigned int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
        }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    unsigned int run = incl - mine;
    for (int w = 0; w < wave; ++w) run += s_wsum[w];
    for (unsigned int j = 0; j < per; ++j)
        {
        const unsigned int t = threadIdx.x * per + j;
        if (t < tg.n_tiles)
            {
            s_first[t] = run;
            if (blockIdx.x == 0) tile_first[t] = run;                // for the scatter and force kernels: one load instead of a prefix
            run += tile_total[t];
            }
        }
    __syncthreads();
    // (bound by its scattered 4-byte stores: four particles per thread with batched loads changed nothing, nor did a software
    // pipeline with tile / slot two trips ahead and the gather one trip ahead, 14.4 us either way; ids along a space-filling
    // curve halve it)
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        {
        const unsigned int t = tile_of[i];
        const unsigned int slot = s_first[t] + rowscan[(size_t)t * tg.n_blocks + i / tg.chunk] + slot_of[i];
        ids[slot] = i;
        }
    }

// 3c. place, SORTED (the default): a block owns the chunk of one counting block, sorts it by tile in LDS — local offset = prefix
// over the chunk's own histogram row + arrival slot — and walks the sorted image, so that consecutive lanes store to consecutive
// slots: the particles of a (chunk, tile) pair leave as one run (8 of them on average at 4096 particles per chunk and 512 tiles).
// What travels is the RAW POSITION RECORD next to the id: the scatter pass then reads ids and positions side by side, coalesced,
// in one memory round trip.  (Gathered by id, the 16-byte records arrived in 128-byte requests — 96 MB fetched for 20 MB of payload,
// profiles/r3/pmc_mesh_ql_summary.txt — behind a dependent id -> position chain; copying them WITHOUT the local sort, one
// scattered 16-byte store per particle, cost the place kernel more than the gather cost the scatter pass: 167 against 160 us.)
// LDS: prefix and destination per tile, (tile, local index) and the raw record per particle of the chunk.
constexpr int TPS_THREADS = 1024;
constexpr unsigned int TPS_CHUNK_MAX = 4096;                       // local index and arrival slot in 16 bits with room to spare
constexpr int TPS_PER = TPS_CHUNK_MAX / TPS_THREADS;
constexpr size_t PS_LDS_MAX = 160 * 1024 - 256;                    // dynamic LDS the kernel may be given (the static part is 128 bytes)

template<typename S4> size_t place_sorted_lds_bytes(const unsigned int n_tiles, const unsigned int chunk)
    {
    return sizeof(unsigned int) * (2 * (size_t)n_tiles + chunk) + sizeof(S4) * (size_t)chunk;
    }

template<typename S4>
__global__ __launch_bounds__(TPS_THREADS) void k_tile_place_sorted(const TileGeom tg, const unsigned int N, const S4 *__restrict__ postype,
                                                                   const unsigned int *__restrict__ tile_of,
                                                                   const unsigned int *__restrict__ slot_of,
                                                                   const unsigned int *__restrict__ hist,
                                                                   const unsigned int *__restrict__ rowscan,
                                                                   const unsigned int *__restrict__ tile_total, unsigned int *__restrict__ ids,
                                                                   S4 *__restrict__ possorted, unsigned int *__restrict__ tile_first)
    {
    extern __shared__ __align__(16) unsigned char s_raw[];
    S4 *s_pos = (S4 *)s_raw;                                                   // [chunk]
    unsigned int *s_meta = (unsigned int *)(s_pos + tg.chunk);                 // [chunk]: tile << 16 | index in the chunk
    unsigned int *s_lpre = s_meta + tg.chunk;                                  // [n_tiles]: prefix over this chunk's histogram row
    unsigned int *s_dest = s_lpre + tg.n_tiles;                                // [n_tiles]: first global slot of the (chunk, tile) run
    __shared__ unsigned int s_w[2][TPS_THREADS / 64];
    const unsigned int b = blockIdx.x;
    const unsigned int i0 = min(N, b * tg.chunk), i1 = min(N, i0 + tg.chunk);
    const unsigned int i_last = i1 ? i1 - 1 : 0u;
    // everything this thread will read of the chunk is requested first (clamped indices, no branch): one memory round trip
    unsigned int tl[TPS_PER], sl[TPS_PER];
    S4 raw[TPS_PER];
#pragma unroll
    for (int k = 0; k < TPS_PER; ++k)
        {
        const unsigned int i = min(i0 + threadIdx.x + k * TPS_THREADS, i_last);
        tl[k] = 0; sl[k] = 0;
        raw[k] = scalar4_traits<S4>::make(0, 0, 0, 0);
        if (i0 < i1)                                                             // (uniform over the block)
            {
            tl[k] = tile_of[i];
            sl[k] = slot_of[i];
            raw[k] = postype[i];
            }
        }
    // the two prefixes over the tiles: tile totals (first slot of every tile) and this chunk's histogram row
    const unsigned int per = (tg.n_tiles + TPS_THREADS - 1) / TPS_THREADS;
    unsigned int tot = 0, loc = 0;
    for (unsigned int j = 0; j < per; ++j)
        {
        const unsigned int t = threadIdx.x * per + j;
        if (t < tg.n_tiles)
            {
            tot += tile_total[t];
            loc += hist[(size_t)b * tg.n_tiles + t];
            }
        }
    unsigned int itot = tot, iloc = loc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
        const unsigned int o0 = __shfl_up(itot, off, 64), o1 = __shfl_up(iloc, off, 64);
        if (lane >= off)
            {
            itot += o0;
            iloc += o1;
            }
        }
    if (lane == 63)
        {
        s_w[0][wave] = itot;
        s_w[1][wave] = iloc;
        }
    __syncthreads();
    unsigned int run = itot - tot, lrun = iloc - loc;
    for (int w = 0; w < wave; ++w)
        {
        run += s_w[0][w];
        lrun += s_w[1][w];
        }
    for (unsigned int j = 0; j < per; ++j)
        {
        const unsigned int t = threadIdx.x * per + j;
        if (t < tg.n_tiles)
            {
            s_lpre[t] = lrun;
            s_dest[t] = run + rowscan[(size_t)t * tg.n_blocks + b];
            if (b == 0) tile_first[t] = run;                         // for the scatter and force kernels: one load instead of a prefix
            run += tile_total[t];
            lrun += hist[(size_t)b * tg.n_tiles + t];
            }
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TPS_PER; ++k)
        {
        const unsigned int li = threadIdx.x + k * TPS_THREADS;
        if (i0 + li < i1)
            {
            const unsigned int lofs = s_lpre[tl[k]] + sl[k];
            s_pos[lofs] = raw[k];
            s_meta[lofs] = (tl[k] << 16) | li;
            }
        }
    __syncthreads();
    const unsigned int n = i1 - i0;
    for (unsigned int j = threadIdx.x; j < n; j += TPS_THREADS)
        {
        const unsigned int meta = s_meta[j], t = meta >> 16;
        const unsigned int dst = s_dest[t] + (j - s_lpre[t]);
        possorted[dst] = s_pos[j];
        ids[dst] = i0 + (meta & 0xffffu);
        }
    }

// ---- 1c-3c in ONE launch: the BIN pipeline (steady state of a run; k_tile_count -> k_tile_rowscan -> k_tile_place_sorted remain
// for the first assignment of a mesh, for riders and for chunks / tables that do not fit) ------------------------------------------
// Counting first and placing afterwards exists only because a tile's segment must be known before the first particle is stored.
// In a simulation the tiles' populations change by a few particles per step: every tile owns a segment with SLACK, planned from the
// exact counts of the previous snapshot (cap = count + max(count / 8, 32), rounded to 8), and a block reserves the run of each of
// its (chunk, tile) pairs with ONE returning atomic add on the tile's cursor.  Whatever does not fit its tile's segment goes to
// an overflow list (tile and slot appended with an atomic counter) that the scatter and force passes scan when it is not empty:
// any snapshot is handled, a badly planned one only slowly, and the cursors — exact counts whatever overflowed — plan the
// next one.  Which slot of its tile a particle gets depends on the order the atomics are served in; nothing that is computed does:
// the mesh is a sum of integers, forces are per particle.  One launch instead of three, the positions read once, no tile / slot /
// histogram / row-scan arrays (config 3: count 8.2 + row scan 4.9 + sorted place 11.6 us -> one launch).
struct TileLists                    // where the scatter and force passes find a tile's particles
    {
    const unsigned int *first;      // first slot of the tile's segment
    const unsigned int *count;      // particles of the tile: entry t * cstride
    const unsigned int *cap;        // bin pipeline: slots of the segment (particles beyond it are in the overflow list); else null
    const unsigned int *ovf_count;  // bin pipeline: entries of the overflow list; else null
    const unsigned int *ovf_tile;   // tile of overflow entry k; its slot is ovf_base + k
    unsigned int cstride, ovf_base;
    };

struct TilePlan                     // what the planning block reads and writes
    {
    const unsigned int *count;      // exact particles per tile of this snapshot (entry t * cstride)
    unsigned int cstride, n_tiles;
    unsigned int *first_next, *cap_next;       // the next snapshot's segments
    unsigned int *cursor_next, *ovf_count_next;   // zeroed for the next snapshot (entry t * cstride_next)
    unsigned int cstride_next;
    const double *modesq_partials;  // sum of mode^2 (:622): block sums of the bin kernel -> mode_sq (null: somebody else adds them up)
    unsigned int n_partials;
    double *mode_sq;
    };

constexpr unsigned int TB_CSTRIDE = 1;                             // cursors side by side: the 64 atomics of a wave travel as four 64-byte requests (one per 128-byte line each: 64 requests, 11 us per launch)
__host__ __device__ __forceinline__ unsigned int tile_capacity(const unsigned int count)
    {
    const unsigned int slack = count >> 3;
    return (count + (slack > 32u ? slack : 32u) + 7u) & ~7u;
    }
// slots all segments together can take (the overflow list lives behind them)
static size_t tile_capacity_total_max(const size_t n_particles, const size_t n_tiles) { return n_particles + n_particles / 8 + 40 * n_tiles + 8; }

// one block (any size that is a multiple of 64, up to 1024 threads)
__device__ __forceinline__ void tile_plan_block(const TilePlan &P, unsigned int *s_w, double *s_red)
    {
    const unsigned int B = blockDim.x;
    const unsigned int per = (P.n_tiles + B - 1) / B;
    unsigned int mine = 0;
    for (unsigned int j = 0; j < per; ++j)
        {
        const unsigned int t = threadIdx.x * per + j;
        if (t < P.n_tiles) mine += tile_capacity(P.count[(size_t)t * P.cstride]);
        }
    unsigned int incl = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
        const unsigned int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
        }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned int run = incl - mine;
    for (int w = 0; w < wave; ++w) run += s_w[w];
    for (unsigned int j = 0; j < per; ++j)
        {
        const unsigned int t = threadIdx.x * per + j;
        if (t < P.n_tiles)
            {
            const unsigned int cap = tile_capacity(P.count[(size_t)t * P.cstride]);
            P.first_next[t] = run;
            P.cap_next[t] = cap;
            P.cursor_next[(size_t)t * P.cstride_next] = 0u;
            run += cap;
            }
        }
    if (threadIdx.x == 0) *P.ovf_count_next = 0u;
    if (P.modesq_partials)
        {
        // (the order of k_tile_rowscan's extra block: 256 strided partial sums, then the waves in sequence — the same bits)
        double v = 0.0;
        if (threadIdx.x < 256)
            for (unsigned int b = threadIdx.x; b < P.n_partials; b += 256) v += P.modesq_partials[b];
        v = block_sum(v, s_red);
        if (threadIdx.x == 0) *P.mode_sq = v;
        }
    }

__global__ __launch_bounds__(1024) void k_tile_plan(const TilePlan P)
    {
    __shared__ unsigned int s_w[16];
    __shared__ double s_red[16];
    tile_plan_block(P, s_w, s_red);
    }

constexpr int TB_THREADS = 1024;
constexpr int TB_PER = TPS_CHUNK_MAX / TB_THREADS;                 // particles per thread
constexpr int TB_TILES_PER_MAX = 8;                                // tiles per thread in the prefix: n_tiles <= 8192
constexpr size_t TB_LDS_MAX = 160 * 1024 - 4096;                   // dynamic LDS (the static part — with a rider's tables — is < 4 KB)

template<typename S4> size_t tile_bin_lds_bytes(const unsigned int n_tiles, const unsigned int chunk)
    {
    return sizeof(unsigned int) * (3 * (size_t)n_tiles + chunk) + sizeof(S4) * (size_t)chunk;
    }

// PER: tiles per thread in the prefix over the chunk's histogram (a compile-time constant: 1, 2, 4 or 8 — every lane runs every trip
// with a clamped tile, so that the atomics and the loads beside them are unconditional and STAY IN FLIGHT: behind a branch the
// compiler waited for each atomic right where it was issued, 6 us per block)
// RIDER (mtd_mesh_set_lamellar_rider): 0 none; 1 / 2: the block partial sums of a set of lamellar CVs (accurate / hardware
// trigonometry) are formed from the four positions every thread holds anyway — in the ~3 us the block WAITS for its atomics (245
// blocks add to every cursor; the memory side serves one word's atomics one after the other): four particles per thread in packed
// fp32, lam_cv_group of the fused step's launch A, ~0.8 us of issue slots per block.  Inside the counting kernel's per-particle
// chain the same terms made the step slower (profiles/r3/mesh_rider_ab.log); here they fill a hole: launch A of a mixed lamellar +
// mesh set (7.9 us at config 3) is gone.
// (the rider's arguments travel in the kernel-argument segment like the fused step's: no device copy per step; nothing for RIDER = 0)
struct BinRiderArgs { mtd::LamKArgs k; double *partials; };
struct BinNoRider { };
template<int RIDER> struct BinRider { typedef BinRiderArgs type; };
template<> struct BinRider<0> { typedef BinNoRider type; };
__device__ __forceinline__ const mtd::LamKArgs &bin_rider_k(const BinRiderArgs &r) { return r.k; }
__device__ __forceinline__ double *bin_rider_partials(const BinRiderArgs &r) { return r.partials; }

template<typename S4, int PER, int RIDER>
__global__ __launch_bounds__(TB_THREADS) void k_tile_bin(const MeshGeom g, const TileGeom tg, const S4 *__restrict__ postype, const unsigned int N,
                                                         const double *__restrict__ mode, const unsigned int n_types,
                                                         const unsigned int *__restrict__ plan_first, const unsigned int *__restrict__ plan_cap,
                                                         unsigned int *__restrict__ cursor, unsigned int *__restrict__ ovf_count,
                                                         unsigned int *__restrict__ ovf_tile, const unsigned int ovf_base,
                                                         unsigned int *__restrict__ ids, S4 *__restrict__ possorted,
                                                         double *__restrict__ modesq_partials, const typename BinRider<RIDER>::type rider)
    {
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ float s_coeff[RIDER ? MTD_MAX_CV * MTD_MAX_TYPES : 1];
    __shared__ mtd::ModeTables s_mt;
    __shared__ double s_wave[RIDER ? (TB_THREADS / 64) * 3 : 1];
    S4 *s_pos = (S4 *)s_raw;                                                   // [chunk]
    unsigned int *s_meta = (unsigned int *)(s_pos + tg.chunk);                 // [chunk]: tile << 16 | index in the chunk
    unsigned int *s_lpre = s_meta + tg.chunk;                                  // [n_tiles]: histogram of the chunk, then its prefix
    unsigned int *s_dest = s_lpre + tg.n_tiles;                                // [n_tiles]: first slot of the (chunk, tile) run
    unsigned int *s_room = s_dest + tg.n_tiles;                                // [n_tiles]: slots of the run inside the tile's segment
    __shared__ double s_red[16];
    __shared__ double s_mode[TP_MODE_LDS];
    __shared__ unsigned int s_w[TB_THREADS / 64];
    const unsigned int b = blockIdx.x;
    const unsigned int i0 = min(N, b * tg.chunk), i1 = min(N, i0 + tg.chunk);
    CNT_STAMP(0);
    if (i0 >= i1)                                                              // (uniform over the block: nothing to bin, nothing to read)
        {
        if (threadIdx.x == 0) modesq_partials[b] = 0.0;
        if constexpr (RIDER != 0)
            if (threadIdx.x < bin_rider_k(rider).n_cv) bin_rider_partials(rider)[(size_t)b * bin_rider_k(rider).n_cv + threadIdx.x] = 0.0;
        return;
        }
    const unsigned int i_last = i1 - 1;
    S4 raw[TB_PER];
    unsigned int tl[TB_PER], sl[TB_PER];
    raw[0] = postype[min(i0 + threadIdx.x, i_last)];
    // the rider's tables: ONE unconditional load per thread from clamped indices, requested right behind the first position (the same
    // memory round trip; the host has put the visited modes into a dense list already: the staging loops of launch A — a loop over
    // blockDim, the list through an index array — cost this kernel four dependent round trips in front of its first particle, +6 us)
    mtd::CvTableRegs tab;
    if constexpr (RIDER != 0) tab = mtd::stage_cv_tables_request(bin_rider_k(rider));
    stage_modes(s_mode, mode, n_types);
    for (unsigned int t = threadIdx.x; t < tg.n_tiles; t += TB_THREADS) s_lpre[t] = 0;
    if constexpr (RIDER != 0) mtd::stage_cv_tables_store(tab, s_coeff, s_mt);
    __syncthreads();
    CNT_STAMP(1);
    // 1. as k_tile_count: tile of every particle, arrival slot from the LDS histogram; the next position in flight meanwhile
    //    (unconditional loads from clamped indices)
    double msq = 0.0;
#pragma unroll
    for (int k = 0; k < TB_PER; ++k)
        {
        const unsigned int i = i0 + threadIdx.x + k * TB_THREADS;
        if (k + 1 < TB_PER) raw[k + 1 < TB_PER ? k + 1 : k] = postype[min(i + TB_THREADS, i_last)];
        tl[k] = 0;
        sl[k] = 0;
        if (i < i1)
            {
            const Particle p = scalar4_traits<S4>::unpack(raw[k]);
            int ix, iy, iz;
            double sx, sy, sz;
            locate(g, p, ix, iy, iz, sx, sy, sz);
            const unsigned int t = (unsigned int)ix / tg.tx + tg.ntx * ((unsigned int)iy / tg.ty + tg.nty * ((unsigned int)iz / tg.tz));
            tl[k] = t;
            sl[k] = atomicAdd(&s_lpre[t], 1u);
            const double a = mode_of(s_mode, mode, (unsigned int)p.type);
            msq += a * a;
            }
        }
    CNT_STAMP(2);
    lds_barrier();
    CNT_STAMP(3);
    // 2. the runs of this chunk: one returning atomic per (chunk, tile) pair — requested first, consumed last — and the prefix over
    //    the chunk's histogram meanwhile.  Lanes beyond the last tile add zero to cursors of their own behind the tiles' (the
    //    cursor arrays have TB_TILES_PER_MAX * TB_THREADS entries; all of them on ONE word would serialise: 1.4 ms per launch).
    unsigned int cnt[PER], base[PER], pfirst[PER], pcap[PER];
    unsigned int loc = 0;
    const unsigned int t_last = tg.n_tiles - 1;
#pragma unroll
    for (int j = 0; j < PER; ++j)
        {
        const unsigned int t = threadIdx.x * PER + j, tc = min(t, t_last);
        const unsigned int c = s_lpre[tc];
        cnt[j] = t <= t_last ? c : 0u;
        pfirst[j] = plan_first[tc];
        pcap[j] = plan_cap[tc];
        loc += cnt[j];
        }
    // (waves whose lanes all lie beyond the last tile request nothing: a branch on a wave-uniform SCALAR condition leaves the
    // atomics of the other waves unconditional)
    if (__builtin_amdgcn_readfirstlane((int)((threadIdx.x & ~63u) * PER)) <= (int)t_last)
        {
#pragma unroll
        for (int j = 0; j < PER; ++j)
            base[j] = atomicAdd(&cursor[(size_t)(threadIdx.x * PER + j) * TB_CSTRIDE], cnt[j]);
        }
    else
        {
#pragma unroll
        for (int j = 0; j < PER; ++j) base[j] = 0;
        }
    unsigned int iloc = loc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
        const unsigned int o = __shfl_up(iloc, off, 64);
        if (lane >= off) iloc += o;
        }
    if (lane == 63) s_w[wave] = iloc;
    lds_barrier();
    unsigned int lrun = iloc - loc;
    for (int w = 0; w < wave; ++w) lrun += s_w[w];
#pragma unroll
    for (int j = 0; j < PER; ++j)
        {
        const unsigned int t = threadIdx.x * PER + j;
        if (t <= t_last) s_lpre[t] = lrun;
        lrun += cnt[j];
        }
    lds_barrier();
    CNT_STAMP(4);
    // 3. the chunk sorted by tile in LDS
#pragma unroll
    for (int k = 0; k < TB_PER; ++k)
        {
        const unsigned int li = threadIdx.x + k * TB_THREADS;
        if (i0 + li < i1)
            {
            const unsigned int lofs = s_lpre[tl[k]] + sl[k];
            s_pos[lofs] = raw[k];
            s_meta[lofs] = (tl[k] << 16) | li;
            }
        }
    float acc[3] = { 0.0f, 0.0f, 0.0f };
    if constexpr (RIDER != 0)
        {
        // (kept between the staging and the first use of the atomics' results)
        __builtin_amdgcn_sched_barrier(0);
        mtd::RawGroup<S4, TB_PER> grp;
#pragma unroll
        for (int k = 0; k < TB_PER; ++k) grp.v[k] = raw[k];
        mtd::lam_cv_group<S4, 3, RIDER == 2, TB_PER>(bin_rider_k(rider), i1, i0 + threadIdx.x, TB_THREADS, s_coeff, s_mt, grp, acc);
        __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
    for (int j = 0; j < PER; ++j)
        {
        const unsigned int t = threadIdx.x * PER + j;
        if (t <= t_last)
            {
            s_dest[t] = pfirst[j] + base[j];
            s_room[t] = pcap[j] > base[j] ? pcap[j] - base[j] : 0u;
            }
        }
    __syncthreads();
    CNT_STAMP(5);
    // 4. out, in runs; what does not fit its tile's segment goes to the overflow list
    const unsigned int n = i1 - i0;
    for (unsigned int j = threadIdx.x; j < n; j += TB_THREADS)
        {
        const unsigned int meta = s_meta[j], t = meta >> 16;
        const unsigned int r = j - s_lpre[t];
        unsigned int dst;
        if (r < s_room[t])
            dst = s_dest[t] + r;
        else
            {
            const unsigned int k = atomicAdd(ovf_count, 1u);
            ovf_tile[k] = t;
            dst = ovf_base + k;
            }
        possorted[dst] = s_pos[j];
        ids[dst] = i0 + (meta & 0xffffu);
        }
    CNT_STAMP(6);
    msq = block_sum_lds(msq, s_red);
    if (threadIdx.x == 0) modesq_partials[b] = msq;
    if constexpr (RIDER != 0)
        {
        // fp32 wave sums -> fp64 across the waves in a fixed order, as lam_cv_block_reduce; partials[chunk][n_cv]
        const unsigned int n_cv = bin_rider_k(rider).n_cv;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            {
            const float v = wave_sum(acc[c]);
            if (lane == 0) s_wave[wave * 3 + c] = (double)v;
            }
        lds_barrier();
        if (threadIdx.x < n_cv)
            {
            double r = 0.0;
            for (int w = 0; w < TB_THREADS / 64; ++w) r += s_wave[w * 3 + threadIdx.x];
            bin_rider_partials(rider)[(size_t)b * n_cv + threadIdx.x] = r;
            }
        }
    CNT_STAMP(7);
    }

template<typename S4>
__global__ __launch_bounds__(TP_THREADS) void k_tile_scatter(const MeshGeom g, const TileGeom tg, const S4 *__restrict__ postype,
                                                             const double *__restrict__ mode, const TileLists L,
                                                             const unsigned int *__restrict__ ids, long long *__restrict__ tilebuf,
                                                             double4 *__restrict__ packed, const unsigned int n_types,
                                                             const S4 *__restrict__ possorted, const TilePlan plan,
                                                             const mtd::MetadCfg apply_cfg)
    {
    __shared__ unsigned long long s_t[TP_HMAX];
    __shared__ double s_mode[TP_MODE_LDS];
    if (blockIdx.x > tg.n_tiles)
        {
        // passenger of a bin step with riders: the bias-grid engine's deferred second pass, one cell per thread (blocks of the size
        // launch A of the fused step gives it: the same sums in the same order)
        __shared__ double s_red2[16];
        const unsigned int ab = blockIdx.x - tg.n_tiles - 1;
        const unsigned int c0 = ab * TP_THREADS;
        mtd::apply_cells(apply_cfg, c0, min(apply_cfg.len, c0 + (unsigned int)TP_THREADS), ab == 0, s_red2);
        return;
        }
    if (blockIdx.x == tg.n_tiles)
        {
        // bin pipeline: the extra block plans the NEXT snapshot's segments from this one's exact counts (and adds up sum mode^2)
        __shared__ unsigned int s_w[16];
        __shared__ double s_red[16];
        tile_plan_block(plan, s_w, s_red);
        return;
        }
    const unsigned int t = blockIdx.x;
    const unsigned int tix = t % tg.ntx, tiy = (t / tg.ntx) % tg.nty, tiz = t / (tg.ntx * tg.nty);
    const int x0 = tix * tg.tx, y0 = tiy * tg.ty, z0 = tiz * tg.tz;
    TILE_STAMP(0, 0);
    const unsigned int n_all = L.count[(size_t)t * L.cstride];       // (k_tile_place*, k_tile_rowscan; bin pipeline: the tile's cursor)
    const unsigned int q0 = L.first[t], q1 = q0 + (L.cap ? min(n_all, L.cap[t]) : n_all);
    const unsigned int n_ovf = L.ovf_count ? *L.ovf_count : 0u;      // (uniform; zero in a well-planned step)
    // Software pipeline over the thread's particles: ids run two particles ahead, positions one, and the position travels RAW
    // (as loaded) to the iteration that uses it.  Carried as a converted Particle it was waited for right behind its load — the
    // conversion to double sat there — and every trip paid the id -> position chain of two memory round trips in full.
    // The loads are unconditional (slots clamped to the tile's last one, which exists: the loop runs only if q < q1): behind a
    // branch the compiler cannot count them and waits for all of them at the first use of any.
    unsigned int q = q0 + threadIdx.x;
    const unsigned int q_last = q1 ? q1 - 1 : 0u;
    unsigned int id = 0, id_next = 0;
    S4 raw = scalar4_traits<S4>::make(0, 0, 0, 0);
    // possorted (k_tile_place_sorted, k_tile_bin): the raw position records in tile order — ids and positions come side by side,
    // coalesced, no id -> position chain; without it (fallback place kernel) the positions are gathered through the ids
    if (q0 < q1)                                                     // (uniform over the block; an empty tile — or no particles at
        {                                                            // all, and then no position array either — loads nothing)
        id = ids[min(q, q_last)];
        id_next = ids[min(q + TP_THREADS, q_last)];
        raw = possorted ? possorted[min(q, q_last)] : postype[id];
        }
    stage_modes(s_mode, mode, n_types);
    for (unsigned int e = threadIdx.x; e < tg.hcells; e += TP_THREADS) s_t[e] = 0ull;
    __syncthreads();
    TILE_STAMP(0, 1);
    // one particle: its record for the force pass at `slot`, its 27 weights into the LDS image
    auto deposit = [&](const S4 &rawv, const unsigned int cur_id, const unsigned int slot)
        {
        const Particle cur = scalar4_traits<S4>::unpack(rawv);
        int ix, iy, iz;
        double sx, sy, sz;
        locate(g, cur, ix, iy, iz, sx, sy, sz);
        const double a0 = mode_of(s_mode, mode, (unsigned int)cur.type);
        const double a = a0 * tg.scale;
        double wx[3], wy[3], wz[3];
        tsc3(sx, wx);
        tsc3(sy, wy);
        tsc3(sz, wz);
#pragma unroll
        for (int i = 0; i < 3; ++i) wz[i] *= a;
        const unsigned int base = (unsigned int)(ix - x0) + tg.hx * ((unsigned int)(iy - y0) + tg.hy * (unsigned int)(iz - z0));
        // the force pass of this snapshot walks the same tile order: in-cell shift, mode, id and the stencil's corner in
        // the tile image, stored in place (coalesced) so that it neither gathers nor locates again
        // (32 bytes: the fourth word carries id, type and corner as integers; 40-byte records cost the two passes ~3 us)
        packed[slot] = make_double4(sx, sy, sz, __hiloint2double((int)(base | ((unsigned int)cur.type << 16)), (int)cur_id));
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                {
                const double wyz = wy[j] * wz[k];
                const unsigned int row = base + tg.hx * (j + tg.hy * k);
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    {
                    // round to nearest integer through the mantissa (|value| < 2^51 by the choice of the scale): the 64-bit
                    // convert instruction sequence is ten times longer
                    const double shifted = wx[i] * wyz + 6755399441055744.0;                 // 1.5 * 2^52
#ifdef MTD_EXP_LDS_NOCONFLICT
                    // DIAGNOSTIC build (wrong sums): every lane of a wave adds to its own 8-byte word — no bank conflicts
                    atomicAdd(&s_t[(threadIdx.x + TP_THREADS * ((k * 3 + j) * 3 + i)) % 6144u], (unsigned long long)(__double_as_longlong(shifted) - 0x4338000000000000ll));
                    (void)row;
#else
                    atomicAdd(&s_t[row + i], (unsigned long long)(__double_as_longlong(shifted) - 0x4338000000000000ll));
#endif
                    }
                }
        };
    while (q < q1)
        {
        const S4 cur_raw = raw;
        const unsigned int cur_id = id;
        const unsigned int qn = q + TP_THREADS;
        raw = possorted ? possorted[min(qn, q_last)] : postype[id_next];
        id = id_next;
        id_next = ids[min(qn + TP_THREADS, q_last)];
        deposit(cur_raw, cur_id, q);
        if (q == q0 + threadIdx.x) TILE_STAMP(0, 2);                   // first particle of thread 0 done
        q = qn;
        }
    // bin pipeline, a snapshot the plan did not fit: this tile's particles in the overflow list (every block scans the list)
    for (unsigned int k = threadIdx.x; k < n_ovf; k += TP_THREADS)
        if (L.ovf_tile[k] == t)
            {
            const unsigned int slot = L.ovf_base + k;
            deposit(possorted[slot], ids[slot], slot);
            }
    TILE_STAMP(0, 3);
    lds_barrier();                                                   // (the record stores of the loop drain behind it)
    TILE_STAMP(0, 4);
    for (unsigned int e = threadIdx.x; e < tg.hcells; e += TP_THREADS) tilebuf[(size_t)t * tg.hcells + e] = (long long)s_t[e];
    TILE_STAMP(0, 5);
    }

// Which entries of the per-tile buffers stand for a mesh cell: along one axis coordinate c is held by its own tile, by the
// right halo of the tile to the left if c is the first cell of its tile and by the left halo of the tile to the right if it
// is the last one.  The offset of an entry is a sum of one term per axis, so a table of nx + ny + nz rows
// {offset 0, offset 1, offset 2, count} (built once per mesh) replaces the divisions: a block owns one mesh row, its y and
// z terms are uniform, most cells have one source per axis.
__global__ __launch_bounds__(256) void k_tile_combine(const MeshGeom g, const TileGeom tg, const long long *__restrict__ tilebuf,
                                                      const uint4 *__restrict__ tsrc, double *__restrict__ rho)
    {
    const unsigned int gy = blockIdx.y, gz = blockIdx.z;
    const uint4 ay = tsrc[g.nx + gy], az = tsrc[g.nx + g.ny + gz];
    const size_t row = (size_t)g.nx * (gy + (size_t)g.ny * gz);
    for (unsigned int gx = blockIdx.x * blockDim.x + threadIdx.x; gx < g.nx; gx += gridDim.x * blockDim.x)
        {
        const uint4 ax = tsrc[gx];
        long long sum = 0;
        for (unsigned int k = 0; k < az.w; ++k)
            {
            const unsigned int oz = k == 0 ? az.x : (k == 1 ? az.y : az.z);
            for (unsigned int j = 0; j < ay.w; ++j)
                {
                const size_t base = (size_t)oz + (j == 0 ? ay.x : (j == 1 ? ay.y : ay.z));
                sum += tilebuf[base + ax.x];
                if (ax.w > 1) sum += tilebuf[base + ax.y];
                if (ax.w > 2) sum += tilebuf[base + ax.z];
                }
            }
        rho[row + gx] = (double)sum * tg.inv_scale;
        }
    }

// The same sum when no coordinate has three sources (every tile at least two cells wide): a thread owns TCB_ROWS cells of
// one column and requests every entry that can stand for them — up to two per axis — before it adds anything.  The loop
// form above keeps one load in flight per thread (a load, its add, the next trip), and with a cell or two per thread the
// kernel ran at the memory latency, not the bandwidth (15 us for 43 MB).
constexpr int TCB_ROWS = 4;

__global__ __launch_bounds__(256) void k_tile_combine_rows(const MeshGeom g, const TileGeom tg, const long long *__restrict__ tilebuf,
                                                           const uint4 *__restrict__ tsrc, double *__restrict__ rho)
    {
    const unsigned int gy0 = blockIdx.y * TCB_ROWS, gz = blockIdx.z;
    const uint4 az = tsrc[g.nx + g.ny + gz];
    uint4 ay[TCB_ROWS];
#pragma unroll
    for (int r = 0; r < TCB_ROWS; ++r) ay[r] = tsrc[g.nx + min(gy0 + r, g.ny - 1)];
    for (unsigned int gx = blockIdx.x * blockDim.x + threadIdx.x; gx < g.nx; gx += gridDim.x * blockDim.x)
        {
        const uint4 ax = tsrc[gx];
        const bool x2 = ax.w > 1;
        long long v[TCB_ROWS][2][2][2];
#pragma unroll
        for (int r = 0; r < TCB_ROWS; ++r)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    {
                    v[r][k][j][0] = 0;
                    v[r][k][j][1] = 0;
                    if ((unsigned int)k < az.w && (unsigned int)j < ay[r].w)           // uniform over the block
                        {
                        const size_t base = (size_t)(k ? az.y : az.x) + (j ? ay[r].y : ay[r].x);
                        v[r][k][j][0] = tilebuf[base + ax.x];
                        if (x2) v[r][k][j][1] = tilebuf[base + ax.y];
                        }
                    }
#pragma unroll
        for (int r = 0; r < TCB_ROWS; ++r)
            {
            const long long sum = ((v[r][0][0][0] + v[r][0][0][1]) + (v[r][0][1][0] + v[r][0][1][1])) +
                                  ((v[r][1][0][0] + v[r][1][0][1]) + (v[r][1][1][0] + v[r][1][1][1]));
            if (gy0 + r < g.ny) rho[(size_t)g.nx * (gy0 + r + (size_t)g.ny * gz) + gx] = (double)sum * tg.inv_scale;
            }
        }
    }

constexpr int TF_THREADS = MTD_TF_THREADS;        // two blocks of eight waves per CU (128 VGPRs): one stages its tile while the other sums

// Re(inv) of a tile + halo into LDS by NT threads (thread `tid` of them): loads into registers first (tile_stage_load: a dozen
// instructions per element, every lane busy, all loads of a thread issued before its first LDS store), stores behind them
template<int NT> struct TileStage
    {
    static constexpr int E = (TP_HMAX + NT - 1) / NT;
    double v[E];
    unsigned int dst[E];                                               // ~0u: nothing to stage
    };

template<int NT>
__device__ __forceinline__ void tile_stage_load(TileStage<NT> &st, const unsigned int tid, const unsigned int n_rows, const unsigned int wxn,
                                                const unsigned int *s_grow, const unsigned short *s_lrow, const unsigned short *s_gx,
                                                const double *__restrict__ inv)
    {
    const unsigned int n_img = n_rows * wxn;
    static_assert(TP_HMAX < 12288 && TP_HMAX * (TP_X + 2) <= (1 << 20), "multiply-shift division of the flat image index");
    const unsigned int inv_w = ((1u << 20) + wxn - 1) / wxn;           // idx / wxn = (idx * inv_w) >> 20: exact while idx * wxn < 2^20 (wxn >= 3: no overflow)
#pragma unroll
    for (int e = 0; e < TileStage<NT>::E; ++e)
        {
        const unsigned int idx = tid + e * NT;
        st.v[e] = 0.0;
        st.dst[e] = ~0u;
        if (idx < n_img)
            {
            const unsigned int row = (idx * inv_w) >> 20, lx = idx - row * wxn;
            st.v[e] = inv[s_grow[row] + s_gx[lx]];
            st.dst[e] = s_lrow[row] + lx;
            }
        }
    }

template<int NT>
__device__ __forceinline__ void tile_stage_store(const TileStage<NT> &st, double *s_inv)
    {
#pragma unroll
    for (int e = 0; e < TileStage<NT>::E; ++e)
        if (st.dst[e] != ~0u) s_inv[st.dst[e]] = st.v[e];
    }

// the force on one particle from its record (k_tile_scatter) and Re(inv) of its tile + halo in LDS; s = 2 / N x bias factor (:861)
template<typename S4>
__device__ __forceinline__ void tile_force_of(const MeshGeom &g, const TileGeom &tg, const double *s_inv, const double *s_mode,
                                              const double *__restrict__ mode, const double4 &cur, const double s, S4 *__restrict__ force)
    {
    typedef typename scalar4_traits<S4>::scalar scalar;
    const unsigned int bt = (unsigned int)__double2hiint(cur.w);          // the record of k_tile_scatter
    const uint2 cib = make_uint2((unsigned int)__double2loint(cur.w), bt & 0xffffu);
    const double a = mode_of(s_mode, mode, bt >> 16), sx = cur.x, sy = cur.y, sz = cur.z;
    double wxv[3], wyv[3], wzv[3], dxv[3], dyv[3], dzv[3];
    tsc3_deriv(sx, wxv, dxv);
    tsc3_deriv(sy, wyv, dyv);
    tsc3_deriv(sz, wzv, dzv);
    // contracted axis by axis: rows with W and W' (x), then planes (y), then the three sums (z) — 90 multiply-adds
    // instead of 108 with the weight products formed per row
    double g1 = 0.0, g2 = 0.0, g3 = 0.0;   // sums multiplying n_x b1, n_y b2, n_z b3
#pragma unroll
    for (int k = 0; k < 3; ++k)
        {
        double pd = 0.0, pdy = 0.0, pw = 0.0;                                // plane k: sum_j of W_y ad, W'_y aw, W_y aw
#pragma unroll
        for (int j = 0; j < 3; ++j)
            {
#ifdef MTD_EXP_LDS_NOCONFLICT
            // DIAGNOSTIC build (wrong forces): consecutive lanes read consecutive words — no bank conflicts
            const unsigned int row = (threadIdx.x + 3 * TF_THREADS * (k * 3 + j)) % 6000u + 0u * cib.y;
            const double r0 = s_inv[row], r1 = s_inv[row + TF_THREADS], r2 = s_inv[(row + 2 * TF_THREADS) % 6144u];
#else
            const unsigned int row = cib.y + tg.hx * (j + tg.hy * k);
            const double r0 = s_inv[row], r1 = s_inv[row + 1], r2 = s_inv[row + 2];
#endif
            const double aw = wxv[0] * r0 + wxv[1] * r1 + wxv[2] * r2;       // row sums with W and with W'
            const double ad = dxv[0] * r0 + dxv[1] * r1 + dxv[2] * r2;
            pd += wyv[j] * ad;
            pdy += dyv[j] * aw;
            pw += wyv[j] * aw;
            }
        g1 += wzv[k] * pd;
        g2 += wzv[k] * pdy;
        g3 += dzv[k] * pw;
        }
    const double c1 = -(double)g.nx * a * g1, c2 = -(double)g.ny * a * g2, c3 = -(double)g.nz * a * g3;
    const double fx = (c1 * g.binv[0][0] + c2 * g.binv[1][0] + c3 * g.binv[2][0]) * s;
    const double fy = (c1 * g.binv[0][1] + c2 * g.binv[1][1] + c3 * g.binv[2][1]) * s;
    const double fz = (c1 * g.binv[0][2] + c2 * g.binv[1][2] + c3 * g.binv[2][2]) * s;
    force[cib.x] = scalar4_traits<S4>::make((scalar)fx, (scalar)fy, (scalar)fz, (scalar)0);
    }

template<typename S4>
__global__ __launch_bounds__(TF_THREADS) void k_tile_forces(const MeshGeom g, const TileGeom tg, const TileLists L,
                                                            const double *__restrict__ mode, const double4 *__restrict__ packed,
                                                            const double *__restrict__ inv, S4 *__restrict__ force,
                                                            const double *__restrict__ d_bias, const double bias_host,
                                                            const double two_over_n, const unsigned int n_types)
    {
    __shared__ double s_inv[TP_HMAX];
    __shared__ double s_mode[TP_MODE_LDS];
    stage_modes(s_mode, mode, n_types);                              // (published by the barrier behind the staging of Re(inv))
    // Blocks go to the eight XCDs round robin (block b -> XCD b mod 8) and every XCD has an L2 of its own: the blocks of one XCD take
    // a CONTIGUOUS range of tiles — two z layers of tiles at 128^3 — so that the halo rows neighbouring tiles share are fetched once
    // per XCD instead of once per tile (the launch has 8 * ceil(n_tiles / 8) blocks; the surplus ones leave): config 3 133.6 -> 132.4 us.
    // (Giving the combine pass and the x/y transforms the same slabs, so that a pass would find the mesh the previous one wrote in
    // its XCD's L2, changed nothing beyond that: 132.8.)
    const unsigned int per_xcd = (tg.n_tiles + 7) / 8;
    const unsigned int t = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || t >= tg.n_tiles) return;
    const unsigned int tix = t % tg.ntx, tiy = (t / tg.ntx) % tg.nty, tiz = t / (tg.ntx * tg.nty);
    const int x0 = tix * tg.tx, y0 = tiy * tg.ty, z0 = tiz * tg.tz;
    TILE_STAMP(1, 0);
    const unsigned int n_all = L.count[(size_t)t * L.cstride];
    const unsigned int q0 = L.first[t], q1 = q0 + (L.cap ? min(n_all, L.cap[t]) : n_all);
    const unsigned int n_ovf = L.ovf_count ? *L.ovf_count : 0u;      // (uniform; zero in a well-planned step)
    if (n_all == 0) return;                                          // (nothing of this tile in the overflow list either)
    // (records are loaded unconditionally from clamped slots: behind a branch the compiler cannot count the load and waits
    // for everything in flight — the previous particle's force store included — at the first use of the record)
    unsigned int q = q0 + threadIdx.x;
    const unsigned int q_last = q1 > q0 ? q1 - 1 : q0;               // (q1 == q0: every particle of the tile overflowed; slot q0 is allocated)
    double4 pk = packed[min(q, q_last)];
    // Re(inv) of the tile + halo.  The counters of round 3 (profiles/r3) put this kernel on the instruction-issue limit with
    // two thirds of its vector instructions in THIS staging (23 rows per thread with wraps and compares, 14 of 32 lanes idle
    // in every row: ~1100 instructions per wave against ~170 per particle and 3.8 particles per thread).  Now a table per
    // block: row r = ly + wyn lz of the image -> its global row offset and its LDS row offset (one thread per row, once), and
    // the image is walked flat, idx = tid + 256 k -> (row, lx) by a multiply-shift: a dozen instructions per element, every
    // lane busy, all loads of a thread issued before its first LDS store.
    const unsigned int wxn = min(tg.tx, g.nx - x0) + 2, wyn = min(tg.ty, g.ny - y0) + 2, wzn = min(tg.tz, g.nz - z0) + 2;
    constexpr int TF_MAXROWS = (TP_Y + 2) * (TP_Z + 2);
    __shared__ unsigned int s_grow[TF_MAXROWS];                        // global offset of the row's x = 0 cell
    __shared__ unsigned short s_lrow[TF_MAXROWS];                      // LDS offset of the row's lx = 0 entry
    __shared__ unsigned short s_gx[TP_X + 2];                          // wrapped global x of image column lx
    const unsigned int n_rows = wyn * wzn;
    for (unsigned int r = threadIdx.x; r < n_rows; r += TF_THREADS)   // (one trip at the default tile shape)
        {
        const unsigned int lz = r / wyn, ly = r - lz * wyn;
        int gy = y0 + (int)ly - 1, gz = z0 + (int)lz - 1;
        gy = gy < 0 ? gy + (int)g.ny : (gy >= (int)g.ny ? gy - (int)g.ny : gy);
        gz = gz < 0 ? gz + (int)g.nz : (gz >= (int)g.nz ? gz - (int)g.nz : gz);
        s_grow[r] = g.nx * ((unsigned int)gy + g.ny * (unsigned int)gz);
        s_lrow[r] = (unsigned short)(tg.hx * (ly + tg.hy * lz));
        }
    if (threadIdx.x < wxn)
        {
        int gx = x0 + (int)threadIdx.x - 1;
        gx = gx < 0 ? gx + (int)g.nx : (gx >= (int)g.nx ? gx - (int)g.nx : gx);
        s_gx[threadIdx.x] = (unsigned short)gx;
        }
    __syncthreads();
    {
    TileStage<TF_THREADS> st;
    tile_stage_load<TF_THREADS>(st, threadIdx.x, n_rows, wxn, s_grow, s_lrow, s_gx, inv);
    tile_stage_store<TF_THREADS>(st, s_inv);
    }
    __syncthreads();
    TILE_STAMP(1, 1);
    const double bias = d_bias ? *d_bias : bias_host;
    const double s = two_over_n * bias;                                // :861
    auto force_of = [&](const double4 &cur) { tile_force_of<S4>(g, tg, s_inv, s_mode, mode, cur, s, force); };
    while (q < q1)
        {
        const double4 cur = pk;
        const unsigned int qn = q + TF_THREADS;
        pk = packed[min(qn, q_last)];
        force_of(cur);
        if (q == q0 + threadIdx.x) TILE_STAMP(1, 2);
        q = qn;
        }
    // bin pipeline: this tile's particles in the overflow list (their records sit at the list's slots)
    for (unsigned int k = threadIdx.x; k < n_ovf; k += TF_THREADS)
        if (L.ovf_tile[k] == t) force_of(packed[L.ovf_base + k]);
    TILE_STAMP(1, 3);
    }

// ---- 9b. the force pass WITH the bias-grid engine's launch inside (mtd_mesh_forces_update_bias) -----------------------------------
// A mixed set of one mesh CV and up to three lamellar CVs ends its step with two launches that need each other's neighbourhood but
// not each other's hardware: the engine's launch (scalar chain on the CV sums -> bias factors; first grid pass; lamellar forces:
// k_fused_force, 11 us, HBM streaming) and this file's force pass (22 us, LDS images, scattered stores), which waits for the
// mesh's bias factor.  Here they are ONE launch of the force pass's shape: wave 0 of every block runs the chain (chain_wave, as in
// k_fused_force: everything it reads requested at entry) while waves 1-7 stage Re(inv) of the block's tile; every block streams
// its 1/n_blocks share of the particles for the lamellar forces (group 0 requested at entry and summed beside the staging loads,
// scaled and stored behind the barrier that publishes the chain), the first n_grid_blocks blocks take 256 cells of the first
// grid pass each (grid_first_pass_256: k_fused_force's sums in its order), block 0 publishes the step's scalars.  The mesh's
// bias factor comes out of LDS.  The engine's state and the lamellar forces are what k_fused_force produces for the same input,
// the mesh forces what k_tile_forces produces (same statements; this file is compiled with -ffp-contract=on, fused.hip with
// =fast: the chain's and the lamellar forces' last bits may differ between the two forms — tests hold them together at 1e-13
// / one fp32 rounding).
constexpr int TFC_STREAM_THREADS = TF_THREADS - MTD_WAVE;
constexpr int TFC_U = 4;           // particles per register group of a streaming thread (lamellar_device.hpp: ForceRegs)

template<typename S4, int NCV, bool FAST>
__global__ __launch_bounds__(TF_THREADS, 4) void k_tile_forces_chain(const MeshGeom g, const TileGeom tg, const TileLists L,
                                                                  const double *__restrict__ mode, const double4 *__restrict__ packed,
                                                                  const double *__restrict__ inv, S4 *__restrict__ force,
                                                                  const double two_over_n, const unsigned int n_types, const unsigned int mesh_slot,
                                                                  const mtd::LamKArgs a, const S4 *__restrict__ postype, const mtd::ForcePtrs out,
                                                                  const unsigned int N, const mtd::MetadCfg c, const int deposit,
                                                                  const unsigned int n_grid_blocks)
    {
    using namespace mtd;
    static_assert(TF_THREADS >= 320, "threads 0 .. 255 of a block take a grid block's cells, wave 0 runs the chain");
    __shared__ double s_inv[TP_HMAX];
    __shared__ double s_mode[TP_MODE_LDS];
    __shared__ ChainResult s_chain;
    __shared__ float s_wcoef[MTD_MAX_CV * MTD_MAX_TYPES];
    __shared__ double s_red[8];
    __shared__ ModeTables s_mt;
    constexpr int TF_MAXROWS = (TP_Y + 2) * (TP_Z + 2);
    __shared__ unsigned int s_grow[TF_MAXROWS];
    __shared__ unsigned short s_lrow[TF_MAXROWS];
    __shared__ unsigned short s_gx[TP_X + 2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    // ---- entry: every memory request that needs nothing but the arguments
    constexpr int NCH = CHAIN_MAX_CV;
    ChainPre<NCH> pre;
    float4 th = make_float4(0.f, 0.f, 0.f, 0.f), tq = th;
    float tc = 0.0f;
    if (wave == 0)
        {
        if (lane < (int)a.n_modes)
            {
            th = a.h[lane];
            tq = a.q[lane];
            }
        if (lane < NCV * MTD_MAX_TYPES) tc = a.coeff[lane / MTD_MAX_TYPES][lane % MTD_MAX_TYPES];
        chain_preload<NCH>(c, pre);
        }
    // this block's share of the particles (lamellar forces): streaming thread sid of n_blocks x TFC_STREAM_THREADS
    const unsigned int sid = threadIdx.x - MTD_WAVE;                  // (waves 1 .. 7)
    const unsigned int stride = gridDim.x * TFC_STREAM_THREADS;
    const unsigned int first = blockIdx.x * TFC_STREAM_THREADS + sid;
    RawGroup<S4, TFC_U> raw0;
    if (wave != 0 && N) lam_force_request<S4, TFC_U>(postype, N, first, stride, raw0);
    // this block's tile (k_tile_forces: contiguous ranges of tiles per XCD)
    const unsigned int per_xcd = (tg.n_tiles + 7) / 8;
    const unsigned int t = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const bool has_tile = (blockIdx.x >> 3) < per_xcd && t < tg.n_tiles;
    const unsigned int tt = has_tile ? t : 0u;
    const unsigned int tix = tt % tg.ntx, tiy = (tt / tg.ntx) % tg.nty, tiz = tt / (tg.ntx * tg.nty);
    const int x0 = tix * tg.tx, y0 = tiy * tg.ty, z0 = tiz * tg.tz;
    const unsigned int n_all = has_tile ? L.count[(size_t)tt * L.cstride] : 0u;
    const unsigned int q0 = L.first[tt], q1 = q0 + (L.cap ? min(n_all, L.cap[tt]) : n_all);
    const unsigned int n_ovf = L.ovf_count ? *L.ovf_count : 0u;
    const bool work = n_all != 0;                                     // (uniform over the block)
    unsigned int q = q0 + threadIdx.x;
    const unsigned int q_last = q1 > q0 ? q1 - 1 : q0;
    double4 pk = make_double4(0.0, 0.0, 0.0, 0.0);
    if (work) pk = packed[min(q, q_last)];
    stage_modes(s_mode, mode, n_types);
    const unsigned int wxn = min(tg.tx, g.nx - x0) + 2, wyn = min(tg.ty, g.ny - y0) + 2, wzn = min(tg.tz, g.nz - z0) + 2;
    const unsigned int n_rows = wyn * wzn;
    if (work)
        {
        for (unsigned int r = threadIdx.x; r < n_rows; r += TF_THREADS)
            {
            const unsigned int lz = r / wyn, ly = r - lz * wyn;
            int gy = y0 + (int)ly - 1, gz = z0 + (int)lz - 1;
            gy = gy < 0 ? gy + (int)g.ny : (gy >= (int)g.ny ? gy - (int)g.ny : gy);
            gz = gz < 0 ? gz + (int)g.nz : (gz >= (int)g.nz ? gz - (int)g.nz : gz);
            s_grow[r] = g.nx * ((unsigned int)gy + g.ny * (unsigned int)gz);
            s_lrow[r] = (unsigned short)(tg.hx * (ly + tg.hy * lz));
            }
        if (threadIdx.x < wxn)
            {
            int gx = x0 + (int)threadIdx.x - 1;
            gx = gx < 0 ? gx + (int)g.nx : (gx >= (int)g.nx ? gx - (int)g.nx : gx);
            s_gx[threadIdx.x] = (unsigned short)gx;
            }
        }
    if (wave == 0)
        {
        if (lane < (int)a.n_modes)
            {
            s_mt.h[lane] = th;
            s_mt.q[lane] = tq;
            }
        if (lane < NCV * MTD_MAX_TYPES) s_wcoef[lane] = tc;            // raw coefficient; scaled in place behind the chain
        }
    lds_barrier();                 // mode tables, row tables (LDS only: the chain's and the particles' loads stay in flight across it)

    // ---- wave 0: the chain.  Waves 1-7: Re(inv) of the tile into LDS, the unscaled lamellar forces of group 0 beside its loads
    ForceRegs<NCV, TFC_U> R;
    if (wave == 0)
        {
        double vi[3];
#pragma unroll
        for (int i = 0; i < NCH; ++i) vi[i] = (pre.x[i][0] + pre.x[i][1]) + (pre.x[i][2] + pre.x[i][3]);
        const ChainResult r = chain_wave(c, deposit != 0, true, nullptr, nullptr, false, &pre.patch, pre.patch_ok != 0, true, vi[0], vi[1], vi[2]);
        if (lane == 0) chain_share(s_chain, r);
        if (lane < NCV * MTD_MAX_TYPES)
            {
            const unsigned int cv = lane / MTD_MAX_TYPES;
            const unsigned int gs = cv < a.n_cv ? a.slot[cv] : 0u;                 // the grid's variable behind CV cv of the set
            const double b = gs == 0 ? r.bias[0] : (gs == 1 ? r.bias[1] : r.bias[2]);
            s_wcoef[lane] = (cv < a.n_cv) ? (float)((double)s_wcoef[lane] * b * two_over_n) : 0.0f;
            }
        }
    else
        {
        TileStage<TFC_STREAM_THREADS> st;
        if (work) tile_stage_load<TFC_STREAM_THREADS>(st, sid, n_rows, wxn, s_grow, s_lrow, s_gx, inv);
        if (N) lam_force_unscaled_from<S4, NCV, FAST, TFC_U>(a, N, first, stride, s_mt, raw0, R);
        if (work) tile_stage_store<TFC_STREAM_THREADS>(st, s_inv);
        }
    __syncthreads();               // publishes s_chain / s_wcoef / s_inv

    // ---- behind the chain
    if (wave != 0 && N)
        {
        lam_force_store<S4, NCV, TFC_U>(a, out, first, stride, s_wcoef, R);
        // (further groups: a launch whose blocks x 448 x 4 particles do not cover N)
        for (unsigned int f = first + TFC_U * stride; f < N; f += TFC_U * stride)
            {
            lam_force_unscaled<S4, NCV, FAST, TFC_U>(a, postype, N, f, stride, s_mt, R);
            lam_force_store<S4, NCV, TFC_U>(a, out, f, stride, s_wcoef, R);
            }
        }
    if (blockIdx.x < n_grid_blocks) grid_first_pass_256(c, s_chain, blockIdx.x, s_red);
    if (blockIdx.x == 0 && wave == 0) publish_step(c, s_chain, deposit);
    if (!work) return;
    const double bias = mesh_slot == 0 ? s_chain.bias[0] : (mesh_slot == 1 ? s_chain.bias[1] : s_chain.bias[2]);
    const double s = two_over_n * bias;                                // :861
    while (q < q1)
        {
        const double4 cur = pk;
        const unsigned int qn = q + TF_THREADS;
        pk = packed[min(qn, q_last)];
        tile_force_of<S4>(g, tg, s_inv, s_mode, mode, cur, s, force);
        q = qn;
        }
    for (unsigned int k = threadIdx.x; k < n_ovf; k += TF_THREADS)
        if (L.ovf_tile[k] == tt) tile_force_of<S4>(g, tg, s_inv, s_mode, mode, packed[L.ovf_base + k], s, force);
    }

// ---- 6/8. DFT of lines staged in LDS ---------------------------------------------------------------
// A block transforms `tile` lines of length n.  Element p of line t sits at data[base + t*line_stride + p*elem_stride].
// LDS layout [p][tile] (consecutive lines in consecutive 16-B slots: conflict-free butterflies).
constexpr int FFT_THREADS = 256;

// Line lengths that are not powers of two (log2n == 0 marks them) take a direct O(n^2) DFT in LDS from buffer `in` to
// buffer `out` (natural order both): n <= 256, so a pass costs n^2 per line — tens of microseconds on this machine, and
// any mesh size the reference accepts (kiss_fft / cuFFT take arbitrary n) works.  twiddle[j] = exp(-2 pi i j / n), j < n.
__device__ __forceinline__ void dft_direct(const double2 *in, double2 *out, const double2 *__restrict__ twiddle, const unsigned int n,
                                           const unsigned int tile, const int inverse)
    {
    for (unsigned int idx = threadIdx.x; idx < n * tile; idx += FFT_THREADS)
        {
        const unsigned int k = idx / tile, t = idx % tile;
        double re = 0.0, im = 0.0;
        unsigned int r = 0;                                      // (p * k) mod n
        for (unsigned int p = 0; p < n; ++p)
            {
            double2 w = twiddle[r];
            if (inverse) w.y = -w.y;
            const double2 v = in[p * tile + t];
            re += v.x * w.x - v.y * w.y;
            im += v.x * w.y + v.y * w.x;
            r += k;
            if (r >= n) r -= n;
            }
        out[k * tile + t] = make_double2(re, im);
        }
    __syncthreads();
    }

// LDS slot of line position p at load time: bit-reversed for the radix-2 path (decimation in time), natural otherwise
__device__ __forceinline__ unsigned int lds_slot(const unsigned int p, const unsigned int log2n)
    {
    return log2n ? (__brev(p) >> (32 - log2n)) : p;
    }

// Power-of-two lines, LDS layout [p][tile] with tile = 2^log2tile lines.  Two radix-2 stages per sweep over LDS (radix-4
// butterflies in registers: half the LDS traffic and half the barriers of a stage-by-stage loop), all index arithmetic
// in shifts and masks (tile, n and the stage lengths are powers of two; as run-time divisors they cost more than the
// butterflies).  twiddle[k] = exp(-2 pi i k / n).
__device__ __forceinline__ double2 cmul(const double2 a, const double2 w) { return make_double2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
__device__ __forceinline__ double2 cadd(const double2 a, const double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(const double2 a, const double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

// decimation in time: bit-reversed order in, natural order out; forward (inverse = 0) or inverse sign
// `stride`: elements between consecutive positions of a line in LDS (the tile, or tile + 1 where a kernel also walks the
// image along the positions — the x passes — and an odd stride keeps those walks off one bank group)
__device__ __forceinline__ void fft_dit_pow2(double2 *s, const double2 *__restrict__ twiddle, const unsigned int log2n,
                                             const unsigned int log2tile, const int inverse, const unsigned int stride)
    {
    const unsigned int n = 1u << log2n, tile = 1u << log2tile, tmask = tile - 1;
    unsigned int log2len = 1;                                    // next stage: len = 2^log2len
    if (log2n & 1)
        {
        // single radix-2 stage len = 2 (twiddle 1)
        for (unsigned int idx = threadIdx.x; idx < (n >> 1) << log2tile; idx += FFT_THREADS)
            {
            const unsigned int t = idx & tmask, bf = idx >> log2tile;
            const unsigned int i0 = bf << 1;
            const double2 u = s[i0 * stride + t], v = s[(i0 + 1) * stride + t];
            s[i0 * stride + t] = cadd(u, v);
            s[(i0 + 1) * stride + t] = csub(u, v);
            }
        __syncthreads();
        log2len = 2;
        }
    for (; log2len < log2n + 1; log2len += 2)
        {
        // stages len = 2^log2len and 2 len in one sweep: elements a, b = a + len/2, c = a + len, d = c + len/2
        const unsigned int log2half = log2len - 1, half = 1u << log2half;
        for (unsigned int idx = threadIdx.x; idx < (n >> 2) << log2tile; idx += FFT_THREADS)
            {
            const unsigned int t = idx & tmask, bf = idx >> log2tile;          // radix-4 butterfly 0 .. n/4 - 1
            const unsigned int grp = bf >> log2half, j = bf & (half - 1);
            const unsigned int ia = (grp << (log2len + 1)) + j;
            double2 w1 = twiddle[j << (log2n - log2len)];                        // exp(-2 pi i j / len)
            double2 w2 = twiddle[j << (log2n - log2len - 1)];                    // exp(-2 pi i j / 2 len)
            if (inverse)
                {
                w1.y = -w1.y;
                w2.y = -w2.y;
                }
            double2 *pa = s + ia * stride + t;
            const unsigned int sh = half * stride;
            const double2 a = pa[0], b = cmul(pa[sh], w1), c = pa[2 * sh], d = cmul(pa[3 * sh], w1);
            const double2 a1 = cadd(a, b), b1 = csub(a, b), c1 = cmul(cadd(c, d), w2), d1 = cmul(csub(c, d), w2);
            // second stage: (a1, c1) with w2 and (b1, d1) with w2 * exp(-+ i pi / 2) = -+ i w2
            const double2 d1r = inverse ? make_double2(-d1.y, d1.x) : make_double2(d1.y, -d1.x);
            pa[0] = cadd(a1, c1);
            pa[2 * sh] = csub(a1, c1);
            pa[sh] = cadd(b1, d1r);
            pa[3 * sh] = csub(b1, d1r);
            }
        __syncthreads();
        }
    }

// decimation in frequency, inverse sign: natural order in, bit-reversed order out
__device__ __forceinline__ void fft_dif_pow2_inverse(double2 *s, const double2 *__restrict__ twiddle, const unsigned int log2n,
                                                     const unsigned int log2tile)
    {
    const unsigned int n = 1u << log2n, tile = 1u << log2tile, tmask = tile - 1;
    int log2len = (int)log2n;
    for (; log2len >= 2; log2len -= 2)
        {
        // stages len = 2^log2len and len / 2 in one sweep: elements a, b = a + len/4, c = a + len/2, d = a + 3 len/4
        const unsigned int log2q = log2len - 2, q = 1u << log2q;
        for (unsigned int idx = threadIdx.x; idx < (n >> 2) << log2tile; idx += FFT_THREADS)
            {
            const unsigned int t = idx & tmask, bf = idx >> log2tile;
            const unsigned int grp = bf >> log2q, j = bf & (q - 1);
            const unsigned int ia = (grp << log2len) + j;
            double2 wa = twiddle[j << (log2n - log2len)];                        // conj -> exp(+2 pi i j / len)
            double2 w2 = twiddle[j << (log2n - log2len + 1)];                    // conj -> exp(+2 pi i j / (len/2))
            wa.y = -wa.y;
            w2.y = -w2.y;
            double2 *pa = s + (ia << log2tile) + t;
            const unsigned int sh = q << log2tile;
            const double2 a = pa[0], b = pa[sh], c = pa[2 * sh], d = pa[3 * sh];
            const double2 a1 = cadd(a, c), c1 = cmul(csub(a, c), wa);
            const double2 bd = csub(b, d);
            const double2 b1 = cadd(b, d), d1 = cmul(make_double2(-bd.y, bd.x), wa);         // (b - d) * i wa
            pa[0] = cadd(a1, b1);
            pa[sh] = cmul(csub(a1, b1), w2);
            pa[2 * sh] = cadd(c1, d1);
            pa[3 * sh] = cmul(csub(c1, d1), w2);
            }
        __syncthreads();
        }
    if (log2len == 1)
        {
        for (unsigned int idx = threadIdx.x; idx < (n >> 1) << log2tile; idx += FFT_THREADS)
            {
            const unsigned int t = idx & tmask, bf = idx >> log2tile;
            const unsigned int i0 = bf << 1;
            const double2 u = s[(i0 << log2tile) + t], v = s[((i0 + 1) << log2tile) + t];
            s[(i0 << log2tile) + t] = cadd(u, v);
            s[((i0 + 1) << log2tile) + t] = csub(u, v);
            }
        __syncthreads();
        }
    }

__device__ __forceinline__ unsigned int ilog2_dev(const unsigned int v) { return 31u - (unsigned int)__clz((int)v); }

template<bool REAL_INPUT, bool REAL_OUTPUT>
__global__ __launch_bounds__(FFT_THREADS) void k_fft_lines(const double *__restrict__ real_in, double2 *__restrict__ data,
                                                           double *__restrict__ real_out,
                                                           const double2 *__restrict__ twiddle, const unsigned int n,
                                                           const unsigned int log2n, const unsigned int tile,
                                                           const unsigned int elem_stride, const unsigned int line_stride,
                                                           const unsigned int tiles_per_row, const unsigned int row_stride,
                                                           const int inverse, const int p_fastest)
    {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *s = (double2 *)smem;
    // block -> (row, tile within row): base offset of the tile's first line
    const unsigned int row = blockIdx.x / tiles_per_row;
    const unsigned int tin = blockIdx.x % tiles_per_row;
    const size_t base = (size_t)row * row_stride + (size_t)tin * tile * line_stride;
    const unsigned int total = n * tile;
    const unsigned int log2tile = ilog2_dev(tile);               // tile is a power of two (fft_tile_for)

    // load with bit-reversed position (decimation in time)
    for (unsigned int idx = threadIdx.x; idx < total; idx += FFT_THREADS)
        {
        unsigned int t, p;
        if (p_fastest)
            {
            t = log2n ? idx >> log2n : idx / n;
            p = log2n ? idx & (n - 1) : idx % n;
            }
        else
            {
            p = idx >> log2tile;
            t = idx & (tile - 1);
            }
        const size_t a = base + (size_t)t * line_stride + (size_t)p * elem_stride;
        double2 v;
        if (REAL_INPUT)
            v = make_double2(real_in[a], 0.0);
        else
            v = data[a];
        s[lds_slot(p, log2n) * tile + t] = v;
        }
    __syncthreads();

    if (!log2n)
        {
        dft_direct(s, s + total, twiddle, n, tile, inverse);
        s += total;                                              // the result buffer
        }
    if (log2n) fft_dit_pow2(s, twiddle, log2n, log2tile, inverse, tile);

    for (unsigned int idx = threadIdx.x; idx < total; idx += FFT_THREADS)
        {
        unsigned int t, p;
        if (p_fastest)
            {
            t = log2n ? idx >> log2n : idx / n;
            p = log2n ? idx & (n - 1) : idx % n;
            }
        else
            {
            p = idx >> log2tile;
            t = idx & (tile - 1);
            }
        const size_t a = base + (size_t)t * line_stride + (size_t)p * elem_stride;
        if (REAL_OUTPUT)
            real_out[a] = s[p * tile + t].x;     // interpolateForces only reads Re(inv) (:851-857)
        else
            data[a] = s[p * tile + t];
        }
    }

// ---- 7. spectral step: updateMeshes :697-712 fused with computeCV :896-905 -------------------------
__device__ __forceinline__ double tsc_fourier(double x)              // :487-511
    {
    const double c[6] = {1.0, -1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0, -1.0 / 39916800.0};
    double sinc = 0.0;
    if (x * x <= 1.0)
        {
        double term = 1.0;
        for (int i = 0; i < 6; ++i)
            {
            sinc += c[i] * term;
            term *= x * x;
            }
        }
    else
        sinc = sin(x) / x;
    return sinc * sinc * sinc;
    }

// ---- 6a / 8c. x lines, real <-> half spectrum ----------------------------------------------------------------
// The mesh is real, so its transform is Hermitian: F(-k) = conj F(k).  Only k_x = 0 .. nx/2 is kept (rows of pitch hxp,
// padded so that the y and z passes still move aligned tiles); everything downstream of the x pass — two line passes, the
// fused z pass, the inverse passes — touches ~56 % of the full-spectrum bytes.  The real part of the inverse transform that
// interpolateForces reads (:851-857) is the inverse of the Hermitian part of G, which the spectral step forms directly.
// LDS layout [p][tile] as in k_fft_lines; `tile` adjacent x lines per block.
// radix-2 stages in place, or the direct transform into the second buffer; returns the buffer that holds the result
__device__ __forceinline__ double2 *lds_transform(double2 *s, const double2 *__restrict__ twiddle, const unsigned int n,
                                                  const unsigned int log2n, const unsigned int tile, const int inverse,
                                                  const unsigned int stride)
    {
    if (log2n)
        {
        fft_dit_pow2(s, twiddle, log2n, ilog2_dev(tile), inverse, stride);
        return s;
        }
    dft_direct(s, s + n * tile, twiddle, n, tile, inverse);
    return s + n * tile;
    }

// Two real lines per complex transform: z = a + i b, Z = FFT z, then A[k] = (Z[k] + conj Z[n-k]) / 2 and
// B[k] = (Z[k] - conj Z[n-k]) / 2i — half the butterflies and half the LDS of one transform per line.  `tile` (even) real
// lines per block = tile / 2 complex lines in LDS.
__global__ __launch_bounds__(FFT_THREADS) void k_fft_x_r2c(const double *__restrict__ real_in, double2 *__restrict__ half_out,
                                                           const double2 *__restrict__ twiddle, const unsigned int n,
                                                           const unsigned int log2n, const unsigned int tile, const unsigned int hxp,
                                                           const unsigned int n_lines)
    {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *s = (double2 *)smem;
    const size_t line0 = (size_t)blockIdx.x * tile;
    const unsigned int pairs = tile / 2;
    // both loops below walk the LDS image along the positions (consecutive lanes = consecutive p or k, as the coalesced
    // global access wants): with a row of `pairs` 16-byte slots = 256 B every lane would hit the same banks, so the
    // power-of-two path pads the row by one slot (fft_x_stride)
    const unsigned int ps = log2n ? pairs + 1 : pairs;
    for (unsigned int idx = threadIdx.x; idx < n * pairs; idx += FFT_THREADS)
        {
        const unsigned int u = log2n ? idx >> log2n : idx / n, p = log2n ? idx & (n - 1) : idx % n;
        const size_t la = line0 + 2 * u, lb = la + 1;                // lines past the end (odd line counts) are zero
        s[lds_slot(p, log2n) * ps + u] = make_double2(la < n_lines ? real_in[la * n + p] : 0.0, lb < n_lines ? real_in[lb * n + p] : 0.0);
        }
    __syncthreads();
    s = lds_transform(s, twiddle, n, log2n, pairs, 0, ps);
    const unsigned int hx = n / 2 + 1;
    for (unsigned int idx = threadIdx.x; idx < hx * pairs; idx += FFT_THREADS)
        {
        const unsigned int u = idx / hx, k = idx % hx;
        const double2 zk = s[k * ps + u], zm = s[((n - k) % n) * ps + u];
        if (line0 + 2 * u < n_lines) half_out[(line0 + 2 * u) * hxp + k] = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
        if (line0 + 2 * u + 1 < n_lines) half_out[(line0 + 2 * u + 1) * hxp + k] = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
        }
    }

// The inverse of the above: Z[k] = A[k] + i B[k] with both half spectra completed by Hermitian symmetry, one inverse
// transform, a = Re z and b = Im z are the two real lines (interpolateForces only reads Re(inv), :851-857).
__global__ __launch_bounds__(FFT_THREADS) void k_fft_x_c2r(const double2 *__restrict__ half_in, double *__restrict__ real_out,
                                                           const double2 *__restrict__ twiddle, const unsigned int n,
                                                           const unsigned int log2n, const unsigned int tile, const unsigned int hxp,
                                                           const unsigned int n_lines)
    {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *s = (double2 *)smem;
    const size_t line0 = (size_t)blockIdx.x * tile;
    const unsigned int pairs = tile / 2;
    const unsigned int ps = log2n ? pairs + 1 : pairs;                // padded rows: see k_fft_x_r2c
    const unsigned int hx = n / 2 + 1;
    for (unsigned int idx = threadIdx.x; idx < hx * pairs; idx += FFT_THREADS)
        {
        const unsigned int u = idx / hx, k = idx % hx;
        const double2 zero = make_double2(0.0, 0.0);
        const double2 A = line0 + 2 * u < n_lines ? half_in[(line0 + 2 * u) * hxp + k] : zero;
        const double2 B = line0 + 2 * u + 1 < n_lines ? half_in[(line0 + 2 * u + 1) * hxp + k] : zero;
        s[lds_slot(k, log2n) * ps + u] = make_double2(A.x - B.y, A.y + B.x);                    // A + i B
        if (k != 0 && 2 * k != n)                                                                 // conj A + i conj B at n - k
            s[lds_slot(n - k, log2n) * ps + u] = make_double2(A.x + B.y, -A.y + B.x);
        }
    __syncthreads();
    s = lds_transform(s, twiddle, n, log2n, pairs, 1, ps);
    for (unsigned int idx = threadIdx.x; idx < n * pairs; idx += FFT_THREADS)
        {
        const unsigned int u = log2n ? idx >> log2n : idx / n, p = log2n ? idx & (n - 1) : idx % n;
        const double2 z = s[p * ps + u];
        if (line0 + 2 * u < n_lines) real_out[(line0 + 2 * u) * n + p] = z.x;
        if (line0 + 2 * u + 1 < n_lines) real_out[(line0 + 2 * u + 1) * n + p] = z.y;
        }
    }

// ---- 6ab / 8bc. x and y passes of one mesh plane in ONE launch -------------------------------------------------------
// A line pass is a load phase, a few sweeps over LDS and a store phase that do not overlap inside a block, and with every
// block of a launch resident at once they do not overlap between blocks either: each pass costs ≈ 10 us at 128^3 whatever
// the transform itself takes (≈ 3 us).  The half spectrum of a plane does not fit the LDS next to the x lines it comes
// from, but half of its k_x columns do: `XY_PARTS` blocks per plane, each transforms ALL x lines of the plane (in batches;
// the second read of the plane comes out of the L2 of the XCD the two blocks share) and keeps only its own k_x columns,
// then transforms those columns along y and writes them: the x-transformed plane never exists in HBM.  The inverse
// direction mirrors it: every block inverts all k_x columns along y (in batches of the same size), keeps its own y rows,
// and inverts those along x.  The arithmetic of every line is that of k_fft_x_r2c / k_fft_lines / k_fft_x_c2r (same
// butterflies in the same order), so the slab path, which has to stop between the passes, gives bitwise the same mesh.
constexpr int XY_THREADS = 512;
constexpr unsigned int XY_PARTS = 2;

// Diagnostic build only (-DMTD_STAMPS, tools/build_stamps.sh, tools/stamps_xy.py): where the blocks of the two kernels spend
// their time (s_memrealtime, 10 ns ticks); the product build has no stamps.
#ifdef MTD_STAMPS
__device__ unsigned long long g_xy_stamps[2][16][256];
#define XY_STAMP(dir, row) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_xy_stamps[dir][row][blockIdx.x] = wall_clock64(); } while (0)
#else
#define XY_STAMP(dir, row) do { } while (0)
#endif

// idx -> (idx / d, idx % d) for a run-time d without a division (d >= 1, idx * d < 2^32)
struct FastDiv
    {
    unsigned int d, magic;
    __device__ __forceinline__ void split(const unsigned int idx, unsigned int &q, unsigned int &r) const
        {
        q = d == 1 ? idx : __umulhi(idx, magic);
        r = idx - q * d;
        }
    };

// The radix-4 butterfly of fft_dit_pow2 on registers (a, b, c, d at ia, ia + half, ia + 2 half, ia + 3 half)
__device__ __forceinline__ void bfly4(double2 &a, double2 &b, double2 &c, double2 &d, const double2 w1, const double2 w2, const int inverse)
    {
    const double2 bw = cmul(b, w1), dw = cmul(d, w1);
    const double2 a1 = cadd(a, bw), b1 = csub(a, bw), c1 = cmul(cadd(c, dw), w2), d1 = cmul(csub(c, dw), w2);
    const double2 d1r = inverse ? make_double2(-d1.y, d1.x) : make_double2(d1.y, -d1.x);
    a = cadd(a1, c1);
    c = csub(a1, c1);
    b = cadd(b1, d1r);
    d = csub(b1, d1r);
    }

__device__ __forceinline__ double2 xy_tw(const double2 *__restrict__ twiddle, const unsigned int i, const int inverse)
    {
    double2 w = twiddle[i];
    if (inverse) w.y = -w.y;
    return w;
    }

// fft_dit_pow2 for any number of lines per tile (`td`) with XY_THREADS threads: the same butterflies in the same order on
// every element (bitwise the same lines), but consecutive stages are chained in registers — radix-2 + radix-4 on 8 elements,
// two radix-4 stages on 16 — so that a 128-point line takes two sweeps over LDS and two barriers instead of four.
__device__ __forceinline__ void fft_dit_xy(double2 *s, const double2 *__restrict__ twiddle, const unsigned int log2n, const FastDiv td,
                                           const int inverse, const unsigned int stride)
    {
    const unsigned int n = 1u << log2n;
    unsigned int log2len = 1;
    if (log2n & 1)
        {
        // stage len = 2 (twiddle 1) and the radix-4 stage len = 4, 8 on positions 8 bf .. 8 bf + 7  (log2n >= 3)
        const double2 w1 = xy_tw(twiddle, 1u << (log2n - 2), inverse), w2 = xy_tw(twiddle, 1u << (log2n - 3), inverse);
        const double2 one1 = xy_tw(twiddle, 0, inverse);
        for (unsigned int idx = threadIdx.x; idx < (n >> 3) * td.d; idx += XY_THREADS)
            {
            unsigned int bf, t;
            td.split(idx, bf, t);
            double2 *pa = s + (bf << 3) * stride + t;
            double2 e[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) e[m] = pa[m * stride];
#pragma unroll
            for (int m = 0; m < 8; m += 2)
                {
                const double2 u = e[m], v = e[m + 1];
                e[m] = cadd(u, v);
                e[m + 1] = csub(u, v);
                }
            bfly4(e[0], e[2], e[4], e[6], one1, one1, inverse);
            bfly4(e[1], e[3], e[5], e[7], w1, w2, inverse);
#pragma unroll
            for (int m = 0; m < 8; ++m) pa[m * stride] = e[m];
            }
        lds_barrier();
        log2len = 4;
        }
    // two radix-4 stages (len = 2^log2len .. 8 len) on positions ib + m half, m < 16
    for (; log2len + 2 < log2n + 1; log2len += 4)
        {
        const unsigned int log2half = log2len - 1, half = 1u << log2half;
        for (unsigned int idx = threadIdx.x; idx < (n >> 4) * td.d; idx += XY_THREADS)
            {
            unsigned int bf, t;
            td.split(idx, bf, t);
            const unsigned int grp = bf >> log2half, j = bf & (half - 1);
            double2 *pa = s + ((grp << (log2len + 3)) + j) * stride + t;
            const unsigned int sh = half * stride;
            double2 e[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) e[m] = pa[m * sh];
            {
            const double2 w1 = xy_tw(twiddle, j << (log2n - log2len), inverse), w2 = xy_tw(twiddle, j << (log2n - log2len - 1), inverse);
#pragma unroll
            for (int q = 0; q < 16; q += 4) bfly4(e[q], e[q + 1], e[q + 2], e[q + 3], w1, w2, inverse);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                {
                const unsigned int j2 = j + r * half;
                const double2 w1 = xy_tw(twiddle, j2 << (log2n - log2len - 2), inverse), w2 = xy_tw(twiddle, j2 << (log2n - log2len - 3), inverse);
                bfly4(e[r], e[r + 4], e[r + 8], e[r + 12], w1, w2, inverse);
                }
#pragma unroll
            for (int m = 0; m < 16; ++m) pa[m * sh] = e[m];
            }
        lds_barrier();
        }
    // a last radix-4 stage on its own
    for (; log2len < log2n + 1; log2len += 2)
        {
        const unsigned int log2half = log2len - 1, half = 1u << log2half;
        for (unsigned int idx = threadIdx.x; idx < (n >> 2) * td.d; idx += XY_THREADS)
            {
            unsigned int bf, t;
            td.split(idx, bf, t);
            const unsigned int grp = bf >> log2half, j = bf & (half - 1);
            double2 *pa = s + ((grp << (log2len + 1)) + j) * stride + t;
            const unsigned int sh = half * stride;
            const double2 w1 = xy_tw(twiddle, j << (log2n - log2len), inverse), w2 = xy_tw(twiddle, j << (log2n - log2len - 1), inverse);
            double2 a = pa[0], b = pa[sh], c = pa[2 * sh], d = pa[3 * sh];
            bfly4(a, b, c, d, w1, w2, inverse);
            pa[0] = a;
            pa[sh] = b;
            pa[2 * sh] = c;
            pa[3 * sh] = d;
            }
        lds_barrier();
        }
    }

struct XYPlan
    {
    unsigned int nx, ny, nz, log2nx, log2ny, hx, hxp;
    unsigned int kc_max;        // k_x columns per part (the last part may hold fewer)
    unsigned int pb;            // forward: line pairs per x batch; inverse: line pairs of a part (ny / 2 / XY_PARTS)
    unsigned int xs, ys;        // row strides of the two LDS images (odd)
    FastDiv d_pb, d_kc[XY_PARTS];
    int xcd_map;                // nz % 8 == 0: the parts of a plane run on one XCD
    };

// d_kc[i] without indexing the kernel argument by a run-time value (that copies the struct to scratch)
__device__ __forceinline__ FastDiv xy_kc(const XYPlan &pl, const unsigned int i)
    {
    FastDiv f = pl.d_kc[0];
#pragma unroll
    for (unsigned int j = 1; j < XY_PARTS; ++j)
        if (i == j) f = pl.d_kc[j];
    return f;
    }

__device__ __forceinline__ void xy_block(const XYPlan &pl, unsigned int &plane, unsigned int &part)
    {
    const unsigned int b = blockIdx.x;
    if (pl.xcd_map)
        {
        // consecutive block ids go to consecutive XCDs: ids b and b + 8 share one
        const unsigned int group = b / (8 * XY_PARTS), r = b % (8 * XY_PARTS);
        plane = group * 8 + (r & 7);
        part = r >> 3;
        }
    else
        {
        plane = b / XY_PARTS;
        part = b % XY_PARTS;
        }
    }

// twiddle tables of both axes at the end of the LDS image (a sweep then waits for LDS, not for the L2)
__device__ __forceinline__ void xy_twiddles(double2 *T, const double2 *__restrict__ tw_x, const double2 *__restrict__ tw_y, const XYPlan &pl)
    {
    for (unsigned int i = threadIdx.x; i < pl.nx / 2 + pl.ny / 2; i += XY_THREADS)
        T[i] = i < pl.nx / 2 ? tw_x[i] : tw_y[i - pl.nx / 2];
    }

constexpr int XY_PREFETCH = 10;     // elements per thread of the next batch held in registers while this one is transformed

// TILES: the real mesh is never written — the forward transform sums, for every cell it loads, the entries of the per-tile
// fixed-point images that stand for it (own tile + the halos of the neighbours that overlap it: k_tile_combine's sum, integers
// added, one conversion) straight out of the scatter pass's buffers.  One launch (10 us) and the 16.8 MB write + 16.8 MB read of
// the mesh less per step; the tile images (29.6 MB) are read either way.  Offsets by arithmetic (no table: the y / z terms are
// wave-uniform), for meshes whose tiles divide the axes and are powers of two, at least two cells wide (xy_tiles_ok).
struct XYTiles
    {
    const long long *buf;
    unsigned int log2t[3], nt[3];               // tile width (log2) and tiles per axis
    unsigned int tile_mul[3], loc_mul[3];       // entry offset = sum over axes of tile * tile_mul + loc * loc_mul
    double inv_scale;
    };

// the (up to two) entries along one axis that stand for coordinate c: its own tile's, and the halo of the neighbour it borders
__device__ __forceinline__ bool xy_tile_src(const XYTiles &tl, const int a, const unsigned int c, unsigned int &o0, unsigned int &o1)
    {
    const unsigned int T = 1u << tl.log2t[a], t = c >> tl.log2t[a], l = c & (T - 1);
    o0 = t * tl.tile_mul[a] + (l + 1) * tl.loc_mul[a];
    o1 = o0;
    if (l == 0)
        {
        o1 = (t == 0 ? tl.nt[a] - 1 : t - 1) * tl.tile_mul[a] + (T + 1) * tl.loc_mul[a];    // the left neighbour's right halo
        return true;
        }
    if (l == T - 1)
        {
        o1 = (t == tl.nt[a] - 1 ? 0 : t + 1) * tl.tile_mul[a];                              // the right neighbour's left halo
        return true;
        }
    return false;
    }

template<bool TILES>
__global__ __launch_bounds__(XY_THREADS) void k_fft_xy_forward(const double *__restrict__ real_in, double2 *__restrict__ half_out,
                                                               const double2 *__restrict__ tw_x, const double2 *__restrict__ tw_y,
                                                               const XYPlan pl, const XYTiles tl)
    {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *X = (double2 *)smem, *Y = X + (size_t)pl.nx * pl.xs, *TX = Y + (size_t)pl.ny * pl.ys, *TY = TX + pl.nx / 2;
    unsigned int plane, part;
    xy_block(pl, plane, part);
    const unsigned int k0 = part * pl.kc_max, k1 = min(pl.hx, k0 + pl.kc_max), kc = k1 - k0;
    const FastDiv dk = xy_kc(pl, part);
    const unsigned int nx = pl.nx, ny = pl.ny, pb = pl.pb, xs = pl.xs, ys = pl.ys;
    const size_t line_base = (size_t)plane * ny;
    const unsigned int n_batches = ny / 2 / pb;
    double2 pre[XY_PREFETCH];
    // TILES: this thread's x position is the same for every element it fetches (XY_THREADS is a multiple of nx): its x entries and the
    // plane's z entries once per block
    unsigned int oz0 = 0, oz1 = 0;
    bool z2 = false;
    if (TILES) z2 = xy_tile_src(tl, 2, plane, oz0, oz1);
    auto fetch = [&](const unsigned int batch)
        {
#pragma unroll
        for (int i = 0; i < XY_PREFETCH; ++i)
            {
            const unsigned int idx = min(threadIdx.x + i * XY_THREADS, nx * pb - 1);   // clamped: see xy_fetch_columns
            const unsigned int u = idx >> pl.log2nx, p = idx & (nx - 1);
            const size_t la = line_base + 2 * (batch * pb + u);
            pre[i] = make_double2(real_in[la * nx + p], real_in[(la + 1) * nx + p]);
            }
        };
    // TILES: the sources of a cell are its own tile's entry plus one per axis on which the cell is the first / last of its tile.
    // A thread takes TWO adjacent cells of a line pair per slot pair (slots 2 i and 2 i + 1: positions 2 lane and 2 lane + 1 of the
    // pair u = wave + 8 i): adjacent cells of a row are adjacent entries of a tile image, so one 16-byte load serves both — half
    // the load instructions of one cell per lane, and it was their number (six eight-byte loads per cell pair where the plain
    // kernel has two), not the bytes, that made the first forms slower (profiles/r4/mesh_ab.log).  Which rows of a thread's pairs
    // are first / last rows of a tile is the same for all of them (their row step, 16, and the batches' are multiples of the tile
    // height: xy_tiles_ok), the plane's z sources are block-uniform: the loads are written with FIXED counts per (z, row class),
    // all of them issued before the first add (a conditional chain of loads and adds made the compiler wait for them one by one).
    // The x halo concerns four lanes of a wave (cells 0, 63, 64, 127): one eight-byte load more per source, which every lane issues
    // (lanes without a halo repeat their own entry's address: same cache line) and masks.
    struct __attribute__((aligned(8))) ll2 { long long a, b; };
    constexpr int TSLOTS = 8;                                          // slots in use: 4 line pairs x 2 cells (nx = 128, 32 pairs per batch)
    const unsigned int t_lane = threadIdx.x & 63u, t_wave = threadIdx.x >> 6;
    unsigned int tx0 = 0, tx1 = 0, txe = 0;                            // entry of cell 2 lane; (unused); the halo entry of this lane, if any
    int edge_j = -1;                                                   // which of the two cells has an x halo (-1: none)
    if (TILES)
        {
        unsigned int o0, o1;
        const bool e0 = xy_tile_src(tl, 0, 2 * t_lane, o0, o1);
        tx0 = o0;
        txe = o0;
        if (e0) { txe = o1; edge_j = 0; }
        unsigned int p0, p1;
        const bool e1 = xy_tile_src(tl, 0, 2 * t_lane + 1, p0, p1);
        tx1 = p0;
        if (e1) { txe = p1; edge_j = 1; }
        (void)tx1;
        }
    auto fetch_tiles = [&](const unsigned int batch, auto z2c, auto yclsc)
        {
        constexpr bool Z2 = decltype(z2c)::value;
        constexpr int YCLS = decltype(yclsc)::value;                 // 0: neither row of the pair borders a tile, 1: the first does, 2: the second
        constexpr int NA = (YCLS == 1 ? 2 : 1) * (Z2 ? 2 : 1), NB = (YCLS == 2 ? 2 : 1) * (Z2 ? 2 : 1), NS = NA + NB;
        constexpr int NP = TSLOTS / 2;                                // line pairs per thread and batch
        constexpr int G = NS <= 3 ? NP : 2;                           // line pairs whose loads are in flight together
        const long long *b = tl.buf;
        for (int i0 = 0; i0 < NP; i0 += G)
            {
            ll2 m[G][NS];
            long long e[G][NS];
#pragma unroll
            for (int g = 0; g < G; ++g)
                {
                const unsigned int u = t_wave + 8u * (unsigned int)(i0 + g);
                const unsigned int gy = 2 * (batch * pb + min(u, pb - 1));
                unsigned int oa0, oa1, ob0, ob1;
                (void)xy_tile_src(tl, 1, gy, oa0, oa1);
                (void)xy_tile_src(tl, 1, gy + 1, ob0, ob1);
                unsigned int off[NS];
                int k = 0;
                off[k++] = oz0 + oa0;
                if (YCLS == 1) off[k++] = oz0 + oa1;
                if (Z2)
                    {
                    off[k++] = oz1 + oa0;
                    if (YCLS == 1) off[k++] = oz1 + oa1;
                    }
                off[k++] = oz0 + ob0;
                if (YCLS == 2) off[k++] = oz0 + ob1;
                if (Z2)
                    {
                    off[k++] = oz1 + ob0;
                    if (YCLS == 2) off[k++] = oz1 + ob1;
                    }
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    {
                    __builtin_memcpy(&m[g][q], b + (off[q] + tx0), sizeof(ll2));      // cells 2 lane and 2 lane + 1: adjacent entries
                    e[g][q] = b[off[q] + txe];
                    }
                }
#pragma unroll
            for (int g = 0; g < G; ++g)
                {
                long long sa0 = 0, sa1 = 0, sb0 = 0, sb1 = 0;
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    {
                    sa0 += m[g][q].a + (edge_j == 0 ? e[g][q] : 0ll);
                    sa1 += m[g][q].b + (edge_j == 1 ? e[g][q] : 0ll);
                    }
#pragma unroll
                for (int q = NA; q < NS; ++q)
                    {
                    sb0 += m[g][q].a + (edge_j == 0 ? e[g][q] : 0ll);
                    sb1 += m[g][q].b + (edge_j == 1 ? e[g][q] : 0ll);
                    }
                pre[2 * (i0 + g)] = make_double2((double)sa0 * tl.inv_scale, (double)sb0 * tl.inv_scale);
                pre[2 * (i0 + g) + 1] = make_double2((double)sa1 * tl.inv_scale, (double)sb1 * tl.inv_scale);
                }
            }
        };
    // the row class of this thread's line pairs (constant over them, see above)
    int ycls = 0;
    if (TILES)
        {
        const unsigned int gy0 = 2 * t_wave, T = 1u << tl.log2t[1];
        ycls = (gy0 & (T - 1)) == 0 ? 1 : (((gy0 + 1) & (T - 1)) == T - 1 ? 2 : 0);
        }
    auto fetch_any = [&](const unsigned int batch)
        {
        if (!TILES)
            fetch(batch);
        else if (z2)
            {
            if (ycls == 0) fetch_tiles(batch, std::true_type(), std::integral_constant<int, 0>());
            else if (ycls == 1) fetch_tiles(batch, std::true_type(), std::integral_constant<int, 1>());
            else fetch_tiles(batch, std::true_type(), std::integral_constant<int, 2>());
            }
        else
            {
            if (ycls == 0) fetch_tiles(batch, std::false_type(), std::integral_constant<int, 0>());
            else if (ycls == 1) fetch_tiles(batch, std::false_type(), std::integral_constant<int, 1>());
            else fetch_tiles(batch, std::false_type(), std::integral_constant<int, 2>());
            }
        };
    XY_STAMP(0, 0);
    fetch_any(0);
    xy_twiddles(TX, tw_x, tw_y, pl);
    for (unsigned int batch = 0; batch < n_batches; ++batch)
        {
        const unsigned int pair0 = batch * pb;
        if (!TILES)
            {
#pragma unroll
            for (int i = 0; i < XY_PREFETCH; ++i)
                {
                const unsigned int idx = threadIdx.x + i * XY_THREADS;
                if (idx < nx * pb) X[(idx & (nx - 1)) * xs + (idx >> pl.log2nx)] = pre[i];
                }
            }
        else
            {
            // (slot 2 i + j: cell 2 lane + j of line pair wave + 8 i)
#pragma unroll
            for (int sl = 0; sl < TSLOTS; ++sl)
                {
                const unsigned int u = t_wave + 8u * (unsigned int)(sl >> 1), p = 2 * t_lane + (unsigned int)(sl & 1);
                if (u < pb) X[p * xs + u] = pre[sl];
                }
            }
        lds_barrier();
        // rows to bit-reversed order, lanes along a row: written straight to their bit-reversed rows, the lanes of a wave
        // (consecutive positions of one line) would queue eight deep on the same banks
        for (unsigned int idx = threadIdx.x; idx < nx * pb; idx += XY_THREADS)
            {
            unsigned int q, u;
            pl.d_pb.split(idx, q, u);
            const unsigned int r = lds_slot(q, pl.log2nx);
            if (q < r)
                {
                const double2 a = X[q * xs + u], b = X[r * xs + u];
                X[q * xs + u] = b;
                X[r * xs + u] = a;
                }
            }
        lds_barrier();
        XY_STAMP(0, 1 + 3 * min(batch, 1u));
        if (batch + 1 < n_batches) fetch_any(batch + 1);         // in flight during the sweeps below
        fft_dit_xy(X, TX, pl.log2nx, pl.d_pb, 0, xs);
        XY_STAMP(0, 2 + 3 * min(batch, 1u));
        // the two real lines of a pair, k_x columns of this part only, to their (bit-reversed) y rows of the column image
        for (unsigned int idx = threadIdx.x; idx < kc * pb; idx += XY_THREADS)
            {
            unsigned int u, kl;
            dk.split(idx, u, kl);
            const unsigned int k = k0 + kl;
            const double2 zk = X[k * xs + u], zm = X[((nx - k) & (nx - 1)) * xs + u];
            const unsigned int ya = 2 * (pair0 + u);
            Y[lds_slot(ya, pl.log2ny) * ys + kl] = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
            Y[lds_slot(ya + 1, pl.log2ny) * ys + kl] = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
            }
        lds_barrier();
        XY_STAMP(0, 3 + 3 * min(batch, 1u));
        }
    fft_dit_xy(Y, TY, pl.log2ny, dk, 0, ys);
    XY_STAMP(0, 7);
    for (unsigned int idx = threadIdx.x; idx < kc * ny; idx += XY_THREADS)
        {
        unsigned int ky, kl;
        dk.split(idx, ky, kl);
        const double2 v = Y[ky * ys + kl];
        double *o = (double *)(half_out + (line_base + ky) * pl.hxp + k0 + kl);
        nt_store(v.x, o);                            // streamed out: nothing left for the write-back at the end of the kernel
        nt_store(v.y, o + 1);
        }
    XY_STAMP(0, 8);
    }

__device__ __forceinline__ void xy_fetch_columns(double2 (&pre)[XY_PREFETCH], const double2 *__restrict__ half_in, const XYPlan &pl,
                                                 const size_t line_base, const unsigned int cpart)
    {
    const unsigned int c0 = cpart * pl.kc_max;
    const FastDiv dc = xy_kc(pl, cpart);
#pragma unroll
    for (int i = 0; i < XY_PREFETCH; ++i)
        {
        // (clamped, not predicated: every load is issued before the first is waited for, and pre[] stays in registers)
        const unsigned int idx = min(threadIdx.x + i * XY_THREADS, dc.d * pl.ny - 1);
        unsigned int ky, cl;
        dc.split(idx, ky, cl);
        const double2 v = half_in[(line_base + ky) * pl.hxp + c0 + cl];
        pre[i].x = v.x;     // (by component: a 16-byte aggregate copy keeps the array in scratch)
        pre[i].y = v.y;
        }
    }

// cvp_*: the CV's partial sums of the fused z pass (one per block of that launch: 1152 at 128^3) folded to one per block of THIS
// launch, which runs between the z pass and whoever reads the CV — the bias-grid engine's chain reads 256 partial sums per variable in
// the memory round trip it starts from and needs another dependent trip for every 512 beyond them, in every block of its launch.
// Wave 0 requests its entries (block + lane * gridDim) in front of everything else, unconditionally (cvp_n = 0: nothing counts, the
// clamped index reads entry 0), and adds them up when the kernel is done: no wait of its own.
__global__ __launch_bounds__(XY_THREADS) void k_fft_xy_inverse(const double2 *__restrict__ half_in, double *__restrict__ real_out,
                                                               const double2 *__restrict__ tw_x, const double2 *__restrict__ tw_y,
                                                               const XYPlan pl, const double *__restrict__ cvp_in, const unsigned int cvp_n,
                                                               double *__restrict__ cvp_out)
    {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *X = (double2 *)smem, *Y = X + (size_t)pl.nx * pl.xs, *TX = Y + (size_t)pl.ny * pl.ys, *TY = TX + pl.nx / 2;
    const unsigned int cvp_k = blockIdx.x + (threadIdx.x & 63u) * gridDim.x;
    double cvp = cvp_in[min(cvp_k, cvp_n ? cvp_n - 1 : 0u)];
    unsigned int plane, part;
    xy_block(pl, plane, part);
    const unsigned int nx = pl.nx, ny = pl.ny, pairs = pl.pb, xs = pl.xs, ys = pl.ys;
    const unsigned int y_base = part * 2 * pairs;                // this block's rows: y_base .. y_base + 2 pairs - 1
    const size_t line_base = (size_t)plane * ny;
    double2 pre[XY_PREFETCH];
    XY_STAMP(1, 0);
    xy_fetch_columns(pre, half_in, pl, line_base, 0);
    xy_twiddles(TX, tw_x, tw_y, pl);
    for (unsigned int cpart = 0; cpart < XY_PARTS; ++cpart)
        {
        const unsigned int c0 = cpart * pl.kc_max;
        const FastDiv dc = xy_kc(pl, cpart);
        const unsigned int cb = dc.d;
#pragma unroll
        for (int i = 0; i < XY_PREFETCH; ++i)
            {
            const unsigned int idx = threadIdx.x + i * XY_THREADS;
            if (idx < cb * ny)
                {
                unsigned int ky, cl;
                dc.split(idx, ky, cl);
                Y[lds_slot(ky, pl.log2ny) * ys + cl] = make_double2(pre[i].x, pre[i].y);
                }
            }
        lds_barrier();
        XY_STAMP(1, 1 + 3 * cpart);
        if (cpart + 1 < XY_PARTS) xy_fetch_columns(pre, half_in, pl, line_base, cpart + 1);   // in flight during the sweeps below
        fft_dit_xy(Y, TY, pl.log2ny, dc, 1, ys);
        XY_STAMP(1, 2 + 3 * cpart);
        // rows of this part: A + i B of a pair of rows at k and its mirror at n - k (k_fft_x_c2r)
        for (unsigned int idx = threadIdx.x; idx < cb * pairs; idx += XY_THREADS)
            {
            unsigned int cl, u;
            pl.d_pb.split(idx, cl, u);
            const unsigned int k = c0 + cl, ya = y_base + 2 * u;
            const double2 A = Y[ya * ys + cl], B = Y[(ya + 1) * ys + cl];
            X[lds_slot(k, pl.log2nx) * xs + u] = make_double2(A.x - B.y, A.y + B.x);
            if (k != 0 && 2 * k != nx) X[lds_slot(nx - k, pl.log2nx) * xs + u] = make_double2(A.x + B.y, -A.y + B.x);
            }
        lds_barrier();
        XY_STAMP(1, 3 + 3 * cpart);
        }
    fft_dit_xy(X, TX, pl.log2nx, pl.d_pb, 1, xs);
    XY_STAMP(1, 7);
    for (unsigned int idx = threadIdx.x; idx < nx * pairs; idx += XY_THREADS)
        {
        const unsigned int u = idx >> pl.log2nx, p = idx & (nx - 1);
        const double2 z = X[p * xs + u];
        const size_t la = line_base + y_base + 2 * u;
        __builtin_nontemporal_store(z.x, real_out + la * nx + p);
        __builtin_nontemporal_store(z.y, real_out + (la + 1) * nx + p);
        }
    if (threadIdx.x < 64)                                        // (wave 0, uniform over the wave)
        {
        cvp = cvp_k < cvp_n ? cvp : 0.0;
        const double tot = wave_sum(cvp);
        if (threadIdx.x == 0 && cvp_n) cvp_out[blockIdx.x] = tot;
        }
    XY_STAMP(1, 8);
    }

// ---- 6ab'. forward x/y passes of a 128 x 128 plane from the tile images WITHOUT the redundant half: the x transform split by the
// parity of its OUTPUT (k_x) ------------------------------------------------------------------------------------------------------
// k_fft_xy_forward lets both blocks of a plane transform all x lines and keep half of the k_x columns.  Decimation in frequency:
// Z[2 m] = DFT_64(z[n] + z[n + 64]), Z[2 m + 1] = DFT_64((z[n] - z[n + 64]) w^n) — block p of a plane folds every line pair while its
// cells arrive (tile images summed as in k_fft_xy_forward<true>) and transforms lines of HALF the length; it keeps the k_x of parity p
// (the two real rows of a pair are untangled from Z[k] and Z[128 - k]: the same parity).  The folded image of ALL 64 line pairs (67 KB)
// fits the LDS next to the column image (68 KB): one batch, 64 lines per sweep.  A lane takes cells 2 j, 2 j + 1 and their partners
// 64 cells on (the same entries of the next x tile) of line pair u = (wave & 3) + 4 half + 8 (g + 4 (wave >> 2)): whether a row of
// the pair borders a tile face then depends on the wave only.
__device__ __forceinline__ void xys_block(const int xcd_map, unsigned int &plane, unsigned int &part);

struct XYSplitF
    {
    unsigned int hxp;
    unsigned int xs, ys;                // row strides of the folded line image (64 pairs, odd) and of the column image (columns of a parity, odd)
    FastDiv d_kc[2];                    // columns of parity 0 / 1 (33 / 32)
    FastDiv d_np;                       // line pairs (64)
    int xcd_map;
    };

__global__ __launch_bounds__(XY_THREADS) void k_fft_xy_forward_split(double2 *__restrict__ half_out, const double2 *__restrict__ tw_x,
                                                                     const double2 *__restrict__ tw_y, const XYSplitF pl, const XYTiles tl)
    {
    constexpr unsigned int NX = 128, NY = 128, NP = NY / 2, NF = NX / 2;     // cells, rows, line pairs, folded positions
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *X = (double2 *)smem, *Y = X + (size_t)NF * pl.xs, *TX = Y + (size_t)NY * pl.ys, *TXh = TX + NX / 2, *TY = TXh + NX / 4;
    unsigned int plane, parity;
    xys_block(pl.xcd_map, plane, parity);
    const unsigned int xs = pl.xs, ys = pl.ys;
    const FastDiv dk = parity ? pl.d_kc[1] : pl.d_kc[0];
    const unsigned int kc = dk.d;
    const size_t line_base = (size_t)plane * NY;
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, j = lane & 31u, half = lane >> 5;
    XY_STAMP(0, 0);
    // this lane's x entries: cells 2 j, 2 j + 1 (left) and 2 j + 64, 2 j + 65 (right), and the halo entry of whichever of them
    // is the first / last cell of its tile
    unsigned int oL, oR, oeL, oeR;
    int edge = -1;                                                    // which of the two cells has an x halo (the same one left and right)
    {
    unsigned int o0, o1;
    bool e = xy_tile_src(tl, 0, 2 * j, o0, o1);
    oL = o0;
    oeL = o0;
    if (e) { oeL = o1; edge = 0; }
    e = xy_tile_src(tl, 0, 2 * j + 1, o0, o1);
    if (e) { oeL = o1; edge = 1; }
    e = xy_tile_src(tl, 0, 2 * j + NF, o0, o1);
    oR = o0;
    oeR = o0;
    if (e) oeR = o1;
    e = xy_tile_src(tl, 0, 2 * j + 1 + NF, o0, o1);
    if (e) oeR = o1;
    }
    unsigned int oz0 = 0, oz1 = 0;
    const bool z2 = xy_tile_src(tl, 2, plane, oz0, oz1);
    // row class of this wave's line pairs: 1 the pair's first row is the first row of a tile, 2 its second row the last one, else 0
    const unsigned int T = 1u << tl.log2t[1], um = wave & 3u;
    const int ycls = ((2 * um) & (T - 1)) == 0 ? 1 : (((2 * um + 1) & (T - 1)) == T - 1 ? 2 : 0);
    for (unsigned int i = threadIdx.x; i < NX / 2 + NX / 4 + NY / 2; i += XY_THREADS)
        TX[i] = i < NX / 2 ? tw_x[i] : (i < NX / 2 + NX / 4 ? tw_x[2 * (i - NX / 2)] : tw_y[i - NX / 2 - NX / 4]);
    struct __attribute__((aligned(8))) ll2 { long long a, b; };
    double2 f0[4], f1[4];                                              // folded values of cells 2 j, 2 j + 1 for the four line pairs of this lane
    auto load_pairs = [&](auto z2c, auto yclsc)
        {
        constexpr bool Z2 = decltype(z2c)::value;
        constexpr int YCLS = decltype(yclsc)::value;
        constexpr int NA = (YCLS == 1 ? 2 : 1) * (Z2 ? 2 : 1), NB = (YCLS == 2 ? 2 : 1) * (Z2 ? 2 : 1), NS = NA + NB;
        constexpr int G = NS <= 3 ? 2 : 1;                            // line pairs whose loads are in flight together
        const long long *b = tl.buf;
        for (int g0 = 0; g0 < 4; g0 += G)
            {
            ll2 mL[G][NS], mR[G][NS];
            long long eL[G][NS], eR[G][NS];
#pragma unroll
            for (int g = 0; g < G; ++g)
                {
                const unsigned int u = um + 4u * half + 8u * ((unsigned int)(g0 + g) + 4u * (wave >> 2));
                unsigned int oa0, oa1, ob0, ob1;
                (void)xy_tile_src(tl, 1, 2 * u, oa0, oa1);
                (void)xy_tile_src(tl, 1, 2 * u + 1, ob0, ob1);
                unsigned int off[NS];
                int k = 0;
                off[k++] = oz0 + oa0;
                if (YCLS == 1) off[k++] = oz0 + oa1;
                if (Z2)
                    {
                    off[k++] = oz1 + oa0;
                    if (YCLS == 1) off[k++] = oz1 + oa1;
                    }
                off[k++] = oz0 + ob0;
                if (YCLS == 2) off[k++] = oz0 + ob1;
                if (Z2)
                    {
                    off[k++] = oz1 + ob0;
                    if (YCLS == 2) off[k++] = oz1 + ob1;
                    }
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    {
                    __builtin_memcpy(&mL[g][q], b + (off[q] + oL), sizeof(ll2));
                    __builtin_memcpy(&mR[g][q], b + (off[q] + oR), sizeof(ll2));
                    eL[g][q] = b[off[q] + oeL];
                    eR[g][q] = b[off[q] + oeR];
                    }
                }
#pragma unroll
            for (int g = 0; g < G; ++g)
                {
                long long aL0 = 0, aL1 = 0, aR0 = 0, aR1 = 0, bL0 = 0, bL1 = 0, bR0 = 0, bR1 = 0;
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    {
                    aL0 += mL[g][q].a + (edge == 0 ? eL[g][q] : 0ll);
                    aL1 += mL[g][q].b + (edge == 1 ? eL[g][q] : 0ll);
                    aR0 += mR[g][q].a + (edge == 0 ? eR[g][q] : 0ll);
                    aR1 += mR[g][q].b + (edge == 1 ? eR[g][q] : 0ll);
                    }
#pragma unroll
                for (int q = NA; q < NS; ++q)
                    {
                    bL0 += mL[g][q].a + (edge == 0 ? eL[g][q] : 0ll);
                    bL1 += mL[g][q].b + (edge == 1 ? eL[g][q] : 0ll);
                    bR0 += mR[g][q].a + (edge == 0 ? eR[g][q] : 0ll);
                    bR1 += mR[g][q].b + (edge == 1 ? eR[g][q] : 0ll);
                    }
                // the pair as one complex line (row a + i row b), folded for this block's parity
                const double2 zL0 = make_double2((double)aL0 * tl.inv_scale, (double)bL0 * tl.inv_scale);
                const double2 zL1 = make_double2((double)aL1 * tl.inv_scale, (double)bL1 * tl.inv_scale);
                const double2 zR0 = make_double2((double)aR0 * tl.inv_scale, (double)bR0 * tl.inv_scale);
                const double2 zR1 = make_double2((double)aR1 * tl.inv_scale, (double)bR1 * tl.inv_scale);
                if (parity)
                    {
                    f0[g0 + g] = cmul(csub(zL0, zR0), TX[2 * j]);          // (published by the barrier in front of the first call)
                    f1[g0 + g] = cmul(csub(zL1, zR1), TX[2 * j + 1]);
                    }
                else
                    {
                    f0[g0 + g] = cadd(zL0, zR0);
                    f1[g0 + g] = cadd(zL1, zR1);
                    }
                }
            }
        };
    lds_barrier();                                                   // the twiddles (the odd block's fold reads them)
    if (z2)
        {
        if (ycls == 0) load_pairs(std::true_type(), std::integral_constant<int, 0>());
        else if (ycls == 1) load_pairs(std::true_type(), std::integral_constant<int, 1>());
        else load_pairs(std::true_type(), std::integral_constant<int, 2>());
        }
    else
        {
        if (ycls == 0) load_pairs(std::false_type(), std::integral_constant<int, 0>());
        else if (ycls == 1) load_pairs(std::false_type(), std::integral_constant<int, 1>());
        else load_pairs(std::false_type(), std::integral_constant<int, 2>());
        }
    // folded positions 2 j, 2 j + 1 of line pair u to their bit-reversed rows
    {
    const unsigned int r0 = lds_slot(2 * j, 6), r1 = lds_slot(2 * j + 1, 6);
#pragma unroll
    for (int g = 0; g < 4; ++g)
        {
        const unsigned int u = um + 4u * half + 8u * ((unsigned int)g + 4u * (wave >> 2));
        X[r0 * xs + u] = f0[g];
        X[r1 * xs + u] = f1[g];
        }
    }
    lds_barrier();
    XY_STAMP(0, 1);
    fft_dit_xy(X, TXh, 6, pl.d_np, 0, xs);
    XY_STAMP(0, 2);
    // the two real rows of a pair at the k_x of this block's parity, to their (bit-reversed) y rows of the column image:
    // k = parity + 2 kl is element kl of the even / odd half transform, its mirror 128 - k element 64 - kl (mod 64) / 63 - kl
    for (unsigned int idx = threadIdx.x; idx < kc * NP; idx += XY_THREADS)
        {
        unsigned int u, kl;
        dk.split(idx, u, kl);
        const unsigned int mm = parity ? NF - 1 - kl : (NF - kl) & (NF - 1);
        const double2 zk = X[kl * xs + u], zm = X[mm * xs + u];
        Y[lds_slot(2 * u, 7) * ys + kl] = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
        Y[lds_slot(2 * u + 1, 7) * ys + kl] = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
        }
    lds_barrier();
    XY_STAMP(0, 3);
    fft_dit_xy(Y, TY, 7, dk, 0, ys);
    XY_STAMP(0, 7);
    for (unsigned int idx = threadIdx.x; idx < kc * NY; idx += XY_THREADS)
        {
        unsigned int ky, kl;
        dk.split(idx, ky, kl);
        const double2 v = Y[ky * ys + kl];
        half_out[(line_base + ky) * pl.hxp + parity + 2 * kl] = v;      // (plain stores: the two blocks of a plane fill the halves of every line, the L2 they share merges them)
        }
    XY_STAMP(0, 8);
    }

// ---- 8bc'. inverse x/y passes of a plane WITHOUT the redundant half: the y transform split by the parity of its OUTPUT ----------
// k_fft_xy_inverse lets both blocks of a plane invert all k_x columns along y and keep half of the rows.  Decimation in frequency
// says which half each block needs: out[2 m] = IDFT_{ny/2}(G[k] + G[k + ny/2]), out[2 m + 1] = IDFT_{ny/2}((G[k] - G[k + ny/2]) w^-k)
// — block p of a plane folds the columns once (while they arrive from memory) and transforms lines of HALF the length; its rows are
// y = p, p + 2, ...  Half the butterflies, and the folded column image of ALL k_x columns (ny/2 x 65 at 128^2: 67 KB) fits the LDS
// next to the row image (68 KB): one batch instead of two, 2 x 65 lines per sweep instead of 33.  The x pass is the old one on the
// block's rows (two of them packed into one complex line: k_fft_x_c2r).  Other rounding than the unsplit kernel (a different
// factorisation of the same transform): the slab path, which runs the separate line passes, agrees to 1e-15, not bitwise.
struct XYSplit
    {
    unsigned int nx, ny, nz, log2nx, log2ny, hx, hxp;
    unsigned int rows;          // ny / 2: folded positions of a column = output rows of a block
    unsigned int pairs;         // rows / 2: packed row pairs of a block
    unsigned int xs, ys;        // row strides of the row image (pairs, odd) and of the column image (hx, odd)
    FastDiv d_hx, d_pairs;
    int xcd_map;                // nz % 8 == 0: the two blocks of a plane run on one XCD
    };
constexpr int XYS_SLOTS = 9;        // folded column elements per thread (65 x 64 / 512 at 128^2)

__device__ __forceinline__ void xys_block(const int xcd_map, unsigned int &plane, unsigned int &part)
    {
    const unsigned int b = blockIdx.x;
    if (xcd_map)
        {
        const unsigned int group = b / 16, r = b % 16;               // consecutive block ids go to consecutive XCDs: ids b and b + 8 share one
        plane = group * 8 + (r & 7);
        part = r >> 3;
        }
    else
        {
        plane = b >> 1;
        part = b & 1;
        }
    }

__global__ __launch_bounds__(XY_THREADS) void k_fft_xy_inverse_split(const double2 *__restrict__ half_in, double *__restrict__ real_out,
                                                                     const double2 *__restrict__ tw_x, const double2 *__restrict__ tw_y,
                                                                     const XYSplit pl, const double *__restrict__ cvp_in, const unsigned int cvp_n,
                                                                     double *__restrict__ cvp_out)
    {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *Yi = (double2 *)smem, *X = Yi + (size_t)pl.rows * pl.ys, *TX = X + (size_t)pl.nx * pl.xs, *TY = TX + pl.nx / 2, *TYh = TY + pl.ny / 2;
    const unsigned int cvp_k = blockIdx.x + (threadIdx.x & 63u) * gridDim.x;          // (cvp_*: see k_fft_xy_inverse)
    double cvp = cvp_in[min(cvp_k, cvp_n ? cvp_n - 1 : 0u)];
    unsigned int plane, parity;
    xys_block(pl.xcd_map, plane, parity);
    const unsigned int nx = pl.nx, rows = pl.rows, pairs = pl.pairs, xs = pl.xs, ys = pl.ys, hx = pl.hx;
    const size_t line_base = (size_t)plane * pl.ny;
    XY_STAMP(1, 0);
    // 1. every column element k < ny/2 and its partner k + ny/2, requested before anything else (clamped, not predicated)
    double2 ra[XYS_SLOTS], rb[XYS_SLOTS];
    const unsigned int n_fold = hx * rows;
#pragma unroll
    for (int i = 0; i < XYS_SLOTS; ++i)
        {
        const unsigned int idx = min(threadIdx.x + i * XY_THREADS, n_fold - 1);
        unsigned int k, c;
        pl.d_hx.split(idx, k, c);
        const double2 a = half_in[(line_base + k) * pl.hxp + c], b = half_in[(line_base + k + rows) * pl.hxp + c];
        ra[i].x = a.x; ra[i].y = a.y;      // (by component: a 16-byte aggregate copy keeps the array in scratch)
        rb[i].x = b.x; rb[i].y = b.y;
        }
    for (unsigned int i = threadIdx.x; i < nx / 2 + pl.ny / 2 + pl.ny / 4; i += XY_THREADS)
        TX[i] = i < nx / 2 ? tw_x[i] : (i < nx / 2 + pl.ny / 2 ? tw_y[i - nx / 2] : tw_y[2 * (i - nx / 2 - pl.ny / 2)]);
    lds_barrier();
    // fold (this block's parity) into the column image, bit-reversed rows for the transform
#pragma unroll
    for (int i = 0; i < XYS_SLOTS; ++i)
        {
        const unsigned int idx = threadIdx.x + i * XY_THREADS;
        if (idx < n_fold)
            {
            unsigned int k, c;
            pl.d_hx.split(idx, k, c);
            double2 f;
            if (parity)
                {
                const double2 w = TY[k];                                   // exp(-2 pi i k / ny): the inverse takes its conjugate
                const double2 d = csub(ra[i], rb[i]);
                f = make_double2(d.x * w.x + d.y * w.y, d.y * w.x - d.x * w.y);
                }
            else
                f = cadd(ra[i], rb[i]);
            Yi[lds_slot(k, pl.log2ny - 1) * ys + c] = f;
            }
        }
    lds_barrier();
    XY_STAMP(1, 1);
    // 2. all columns along y, half length
    fft_dit_xy(Yi, TYh, pl.log2ny - 1, pl.d_hx, 1, ys);
    XY_STAMP(1, 2);
    // 3. this block's rows m = 0 .. ny/2 - 1 (y = parity + 2 m), two of them per complex line: A + i B at k_x and its mirror (k_fft_x_c2r)
    for (unsigned int idx = threadIdx.x; idx < hx * pairs; idx += XY_THREADS)
        {
        unsigned int c, u;
        pl.d_pairs.split(idx, c, u);
        const double2 A = Yi[(2 * u) * ys + c], B = Yi[(2 * u + 1) * ys + c];
        X[lds_slot(c, pl.log2nx) * xs + u] = make_double2(A.x - B.y, A.y + B.x);
        if (c != 0 && 2 * c != nx) X[lds_slot(nx - c, pl.log2nx) * xs + u] = make_double2(A.x + B.y, -A.y + B.x);
        }
    lds_barrier();
    XY_STAMP(1, 3);
    // 4. the rows along x
    fft_dit_xy(X, TX, pl.log2nx, pl.d_pairs, 1, xs);
    XY_STAMP(1, 7);
    // 5. out: line u holds rows y = parity + 4 u (real part) and parity + 4 u + 2 (imaginary part)
    for (unsigned int idx = threadIdx.x; idx < nx * pairs; idx += XY_THREADS)
        {
        const unsigned int u = idx >> pl.log2nx, p = idx & (nx - 1);
        const double2 z = X[p * xs + u];
        const size_t la = line_base + parity + 4 * u;
        __builtin_nontemporal_store(z.x, real_out + la * nx + p);
        __builtin_nontemporal_store(z.y, real_out + (la + 2) * nx + p);
        }
    if (threadIdx.x < 64)                                        // (wave 0, uniform over the wave)
        {
        cvp = cvp_k < cvp_n ? cvp : 0.0;
        const double tot = wave_sum(cvp);
        if (threadIdx.x == 0 && cvp_n) cvp_out[blockIdx.x] = tot;
        }
    XY_STAMP(1, 8);
    }

// ---- 6c+7+8a. z lines: forward transform, spectral step, inverse transform — one pass ----------------------
// The last forward pass, updateMeshes/computeCV and the first inverse pass all work on complete z lines, so they share one
// staging of the lines in LDS: the Fourier mesh is written once (f, normalised: the log quantities and the virial read
// it), G never exists in HBM in k-space, and three of the nine sweeps over the mesh (z-pass write, spectral read + write,
// inverse z-pass read) disappear.  Forward: decimation in time (bit-reversed load, natural order out); inverse: decimation in
// frequency on the natural-order G (bit-reversed order out, undone by the store address) — no second LDS buffer.
// The interpolation function I(k) = T(k_x) T(k_y) T(k_z) (:448-449) depends on the mesh dimensions only: its three
// factors are tabulated once (nx + ny + nz doubles).  In bug-compatible mode the argument of sin() is ~1e9 for every negative
// Miller index (Q6), i.e. the slow argument-reduction path of the library sine — per cell and per step that was a third
// of the spectral pass.
__global__ void k_interp_tables(const unsigned int nx, const unsigned int ny, const unsigned int nz, const int bug_compat,
                                double *__restrict__ itab)
    {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx + ny + nz) return;
    unsigned int dim = nx, w = i;
    if (i >= nx + ny)
        {
        dim = nz;
        w = i - nx - ny;
        }
    else if (i >= nx)
        {
        dim = ny;
        w = i - nx;
        }
    int n = (int)w;
    if (n >= (int)(dim / 2 + dim % 2)) n -= (int)dim;              // Miller index :417-422
    // :448 int / unsigned => unsigned division (Q6): the quotient is 0 for n >= 0 and ~2^32/dim otherwise
    const double kH = bug_compat ? (M_PI * 2.0) * (double)((unsigned int)n / dim) : (M_PI * 2.0) * ((double)n / dim);
    itab[i] = tsc_fourier(kH);
    }

// Slab decomposition (SlabArgs::world > 1): the z lines of this rank's y rows [y0, y0 + ny_loc) are read
// straight out of the peers' exported slabs (element z lives on rank z / nz_loc: the all-to-all of a distributed FFT as
// remote loads of `tile`-wide segments, no transpose buffer), G leaves in pencil layout [z][y_loc][x] for the peers to pull.
struct SlabArgs
    {
    const double2 *f[MTD_COMM_MAX_RANKS];      // exported slabs after the x and y passes, [z_loc][y][hxp]
    unsigned int world, nz_loc, ny_loc, y0;
    };

// TPB tiles per block (non-distributed path; default 1): with TPB > 1 the lines of the block's NEXT tile are requested — into
// registers — before this tile's transforms, so that a block is loading while it is sweeping and its stores drain under the next
// tile's sweeps.  The idea: with one tile per block every block of the launch is resident at once and in the same phase — a burst
// that loads the whole mesh, the sweeps, a burst that stores it (round 2's stamps): 15.9 us for 38 MB, 30 % of the HBM peak.  It
// did not pay (fft_z_tpb).  Needs n * tile <= FFT_ZPRE * FFT_THREADS.
constexpr int FFT_ZPRE = 4;

template<bool DIST, int TPB>
__global__ __launch_bounds__(FFT_THREADS) void k_fft_z_spectral(const MeshGeom g, double2 *__restrict__ fmesh, double2 *__restrict__ gmesh,
                                                                const double2 *__restrict__ twiddle, const unsigned int log2n,
                                                                const unsigned int tile, const unsigned int tiles_per_row,
                                                                const double *__restrict__ mode_sq, const double n_global,
                                                                const double *__restrict__ itab, double *__restrict__ cv_partials,
                                                                const SlabArgs sl, const int keep_f, const unsigned int n_regular,
                                                                const unsigned int edge_col)
    {
    // EDGE blocks (whole-mesh path, blockIdx >= n_regular): the half spectrum has nx/2 + 1 columns in rows of pitch hxp — at
    // nx = 128 that is 8 tiles of 8 columns and ONE more column, for which a ninth tile per row moved, transformed and stored seven
    // columns of padding (10 % of the pass).  The lone column `edge_col` is gathered instead: an edge block takes it from `tile`
    // consecutive ROWS (element t of the tile = row first_row + t), so that the launch has ny * 8 + ny / 8 blocks instead of ny * 9.
    // n_regular = UINT_MAX: no edge blocks, every tile lies in one row (the padding columns of a last tile are zeroed).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double s_red[16];
    const unsigned int n = g.nz;
    const unsigned int plane = g.hxp * g.ny;                             // half-spectrum arrays: rows of pitch hxp
    const unsigned int total = n * tile;
    const unsigned int nxh = g.nx / 2;
    const unsigned int log2tile = ilog2_dev(tile);
    double term = 0.0;
    const double msq = *mode_sq;
    // loop invariants of the spectral step, formed once per thread: sum mode^2 / N^2 and 1 / N.  (Left in the loop, every point
    // paid six double divisions — the expressions of :697-712 / :896-905 divide by N twice per use — plus three integer
    // modulo operations for the mirror cell: more instructions than both transforms of the line.  The factors multiply in a
    // different order than the reference's expressions: a relative 1e-16, against 1e-11 asked of the meshes.)
    const double inv_n = 1.0 / n_global;
    const double msq_nn = msq / n_global / n_global;
    double2 pre[FFT_ZPRE];

    for (int it = 0; it < TPB; ++it)
        {
        double2 *s = (double2 *)smem;
        const unsigned int tile_id = blockIdx.x * TPB + it;
        const bool edge = !DIST && TPB == 1 && tile_id >= n_regular;                // (block-uniform)
        const unsigned int edge_row = edge ? (tile_id - n_regular) * tile : 0u;     // first row of an edge block
        const unsigned int wy = edge ? edge_row : (DIST ? sl.y0 : 0u) + tile_id / tiles_per_row;      // row = y index (global)
        const unsigned int x_first = edge ? edge_col : (tile_id % tiles_per_row) * tile;
        const size_t base = (size_t)wy * g.hxp + x_first;
        const unsigned int t_step = edge ? g.hxp : 1u;                              // element t of the tile: the next column, or the next row
        const unsigned int t_valid = edge ? min(tile, g.ny - edge_row) : tile;      // (an edge block at the end of the rows may run short)

        if (TPB > 1 && it > 0)
            {
            // the lines requested during the previous tile's transforms
#pragma unroll
            for (int k = 0; k < FFT_ZPRE; ++k)
                {
                const unsigned int idx = threadIdx.x + k * FFT_THREADS;
                if (idx < total) s[lds_slot(idx >> log2tile, log2n) * tile + (idx & (tile - 1))] = pre[k];
                }
            }
        else
            for (unsigned int idx = threadIdx.x; idx < total; idx += FFT_THREADS)
                {
                const unsigned int p = idx >> log2tile, t = idx & (tile - 1);
                if (DIST)
                    {
                    const unsigned int q = p / sl.nz_loc, zl = p - q * sl.nz_loc;
                    s[lds_slot(p, log2n) * tile + t] = ld_exported(sl.f[q] + base + t + (size_t)zl * plane);
                    }
                else
                    s[lds_slot(p, log2n) * tile + t] = t < t_valid ? fmesh[base + (size_t)t * t_step + (size_t)p * plane] : make_double2(0.0, 0.0);
                }
        lds_barrier();
        if (TPB > 1 && it + 1 < TPB)
            {
            // the next tile's lines: in flight during the sweeps below (clamped, not predicated: every load is issued before
            // the first is waited for)
            const unsigned int nid = tile_id + 1;
            const size_t nbase = (size_t)(nid / tiles_per_row) * g.hxp + (nid % tiles_per_row) * tile;
#pragma unroll
            for (int k = 0; k < FFT_ZPRE; ++k)
                {
                const unsigned int idx = min(threadIdx.x + k * FFT_THREADS, total - 1);
                const double2 v = fmesh[nbase + (idx & (tile - 1)) + (size_t)(idx >> log2tile) * plane];
                pre[k].x = v.x;
                pre[k].y = v.y;
                }
            }
        double2 *const s_first = s;
        if (!log2n)
            {
            dft_direct(s, s + total, twiddle, n, tile, 0);               // forward, natural order, into the second buffer
            s += total;
            }
        if (log2n) fft_dit_pow2(s, twiddle, log2n, log2tile, 0, tile);    // forward, decimation in time

        // spectral step in place: updateMeshes :697-712 + computeCV :896-905 on the stored half of the spectrum.
        // f(-k) = conj f(k), so the cell -k (not stored for 0 < k_x < nx/2) has the same |f|^2 and its own interpolation factor
        // I(-k) (they differ in bug-compatible mode, Q6).  Stored for the inverse transform: the Hermitian part
        // G_H(k) = (G(k) + conj G(-k)) / 2 = f (|f|^2 - (I(k)^2 + I(-k)^2) / 2 * sum mode^2 / 2 N^2), whose inverse is Re(inv).
        const unsigned int my = wy ? g.ny - wy : 0u;                       // mirror row (-k_y as an array index)
        const double Iy_b = itab[g.nx + wy], Imy_b = itab[g.nx + my];
        for (unsigned int idx = threadIdx.x; idx < total; idx += FFT_THREADS)
            {
            const unsigned int p = idx >> log2tile, t = idx & (tile - 1);  // p = k_z index
            const unsigned int wx = edge ? x_first : x_first + t;
            if (wx > nxh || t >= t_valid)                                  // padding column of the half-spectrum rows (or no such row)
                {
                s[p * tile + t] = make_double2(0.0, 0.0);
                if (!DIST && keep_f && !edge) fmesh[base + t + (size_t)p * plane] = make_double2(0.0, 0.0);
                continue;
                }
            // (an edge block's elements lie in different rows: their y factors are per element)
            const unsigned int wy_t = edge ? wy + t : wy;
            const double Iy = edge ? itab[g.nx + wy_t] : Iy_b, Imy = edge ? itab[g.nx + (wy_t ? g.ny - wy_t : 0u)] : Imy_b;
            const unsigned int mx = wx ? g.nx - wx : 0u, mz = p ? g.nz - p : 0u;
            const double I = itab[wx] * Iy * itab[g.nx + g.ny + p];
            const double Im = itab[mx] * Imy * itab[g.nx + g.ny + mz];
            double2 f = s[p * tile + t];
            f.x *= inv_n;
            f.y *= inv_n;
            const double val = f.x * f.x + f.y * f.y;
            const double diagonal_term = 0.25 * (I * I + Im * Im) * msq_nn;
            double2 G = make_double2(f.x * val, f.y * val);
            G.x -= f.x * diagonal_term;
            G.y -= f.y * diagonal_term;
            // the normalised Fourier mesh is only read by the log quantities (q_max) and the virial: written when asked for
            // (mtd_mesh_set_keep_fourier; 18.9 MB per step at 128^3).  Slab runs keep none.
            if (!DIST && keep_f) fmesh[base + (size_t)t * t_step + (size_t)p * plane] = f;
            s[p * tile + t] = G;
            if (wx != 0 || wy_t != 0 || p != 0)                            // exclude the DC bin (:889-894)
                {
                // Re(G f*) - |f|^2 I^2 sum mode^2 / 2 N^2 (:896-905) = |f|^4 - I^2 |f|^2 sum mode^2 / N^2 for the cell itself ...
                double tk = val * val - val * (I * I) * msq_nn;
                // ... plus the same for its mirror image when that one is not stored (k_x = 0 and, for even nx, nx/2 mirror
                // into their own plane)
                if (wx != 0 && 2 * wx != g.nx) tk += val * val - val * (Im * Im) * msq_nn;
                term += tk;
                }
            }
        lds_barrier();

        if (!log2n)
            {
            dft_direct(s, s_first, twiddle, n, tile, 1);                 // inverse, natural order, back into the first buffer
            s = s_first;
            }
        if (log2n) fft_dif_pow2_inverse(s, twiddle, log2n, log2tile);        // inverse, decimation in frequency
        for (unsigned int idx = threadIdx.x; idx < total; idx += FFT_THREADS)
            {
            const unsigned int q = idx >> log2tile, t = idx & (tile - 1);  // LDS slot q holds position z = bitrev(q) (radix-2 path)
            const unsigned int z = lds_slot(q, log2n);
            if (DIST)
                gmesh[((size_t)z * sl.ny_loc + (wy - sl.y0)) * g.hxp + x_first + t] = s[q * tile + t];
            else if (t < t_valid)
                {
                nt_store(s[q * tile + t], gmesh + base + (size_t)t * t_step + (size_t)z * plane);
                }
            }
        if (TPB > 1 && it + 1 < TPB) lds_barrier();                      // the image is free for the next tile (its stores drain meanwhile)
        }
    term = block_sum(term, s_red);
    if (threadIdx.x == 0) cv_partials[blockIdx.x] = term;
    }

// ---- slab decomposition: the pulls between the local transform passes --------------------------------------------
struct PeerPtrs { const void *p[MTD_COMM_MAX_RANKS]; };

// this rank's slab of the mesh = sum over ranks (rank order) of their local assignments
__global__ __launch_bounds__(256) void k_slab_pull_rho(const PeerPtrs peers, const unsigned int world, const size_t first, const size_t n,
                                                       double *__restrict__ out)
    {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        {
        double v = 0.0;
        for (unsigned int q = 0; q < world; ++q) v += ld_exported((const double *)peers.p[q] + first + i);
        out[i] = v;
        }
    }

// this rank's z slab of G, all y rows: row y comes out of the pencil block of rank y / ny_loc
__global__ __launch_bounds__(256) void k_slab_pull_g(const PeerPtrs peers, const unsigned int z0, const unsigned int nz_loc,
                                                     const unsigned int ny, const unsigned int ny_loc, const unsigned int hxp,
                                                     double2 *__restrict__ out)
    {
    const size_t n = (size_t)nz_loc * ny * hxp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        {
        const unsigned int x = (unsigned int)(i % hxp);
        const size_t row = i / hxp;
        const unsigned int y = (unsigned int)(row % ny), zl = (unsigned int)(row / ny);
        const unsigned int q = y / ny_loc;
        out[i] = ld_exported((const double2 *)peers.p[q] + ((size_t)(z0 + zl) * ny_loc + (y - q * ny_loc)) * hxp + x);
        }
    }

// the whole Re(inv) from the slabs of all ranks
__global__ __launch_bounds__(256) void k_slab_pull_inv(const PeerPtrs peers, const unsigned int world, const size_t slab_cells,
                                                       double *__restrict__ out)
    {
    const size_t n = slab_cells * world;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        {
        const unsigned int q = (unsigned int)(i / slab_cells);
        out[i] = ld_exported((const double *)peers.p[q] + (i - (size_t)q * slab_cells));
        }
    }

__global__ void k_copy_doubles(const double *__restrict__ in, double *__restrict__ out, const size_t n)
    {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
    }

// ---- 9. forces --------------------------------------------------------------------------------------
// One thread per SORTED slot: neighbouring threads read neighbouring stencils of Re(inv) (L1/L2 hits instead of
// 27 uncoalesced gathers), shift and mode come from the packed records (no second locate()), the force is written to
// the particle's original index.
template<typename S4>
__global__ __launch_bounds__(256) void k_mesh_forces(const MeshGeom g, const unsigned int N, const uint2 *__restrict__ idcell,
                                                     const double4 *__restrict__ packed,
                                                     const double *__restrict__ inv, S4 *__restrict__ force,
                                                     const double *__restrict__ d_bias, const double bias_host, const double two_over_n)
    {
    typedef typename scalar4_traits<S4>::scalar scalar;
    const double bias = d_bias ? *d_bias : bias_host;
    for (unsigned int q = blockIdx.x * blockDim.x + threadIdx.x; q < N; q += gridDim.x * blockDim.x)
        {
        const double4 pk = packed[q];
        const uint2 ic = idcell[q];
        const unsigned int c = ic.y;
        const int iz = c / (g.nx * g.ny);
        const int iy = (c - iz * g.nx * g.ny) / g.nx;
        const int ix = c % g.nx;
        const double a = pk.w, sx = pk.x, sy = pk.y, sz = pk.z;
        double wxv[3], wyv[3], wzv[3], dxv[3], dyv[3], dzv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            {
            wxv[i] = tsc(sx - (i - 1)); dxv[i] = tsc_deriv(sx - (i - 1));
            wyv[i] = tsc(sy - (i - 1)); dyv[i] = tsc_deriv(sy - (i - 1));
            wzv[i] = tsc(sz - (i - 1)); dzv[i] = tsc_deriv(sz - (i - 1));
            }
        const unsigned int xs[3] = {(unsigned int)wrap(ix - 1, (int)g.nx), (unsigned int)ix, (unsigned int)wrap(ix + 1, (int)g.nx)};
        double g1 = 0.0, g2 = 0.0, g3 = 0.0;   // sums multiplying n_x b1, n_y b2, n_z b3
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                {
                const unsigned int row = g.nx * (wrap(iy + j - 1, (int)g.ny) + g.ny * wrap(iz + k - 1, (int)g.nz));
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    {
                    const double r = inv[row + xs[i]];
                    g1 += dxv[i] * wyv[j] * wzv[k] * r;
                    g2 += wxv[i] * dyv[j] * wzv[k] * r;
                    g3 += wxv[i] * wyv[j] * dzv[k] * r;
                    }
                }
        const double c1 = -(double)g.nx * a * g1, c2 = -(double)g.ny * a * g2, c3 = -(double)g.nz * a * g3;
        const double s = two_over_n * bias;                            // :861
        const double fx = (c1 * g.binv[0][0] + c2 * g.binv[1][0] + c3 * g.binv[2][0]) * s;
        const double fy = (c1 * g.binv[0][1] + c2 * g.binv[1][1] + c3 * g.binv[2][1]) * s;
        const double fz = (c1 * g.binv[0][2] + c2 * g.binv[1][2] + c3 * g.binv[2][2]) * s;
        force[ic.x] = scalar4_traits<S4>::make((scalar)fx, (scalar)fy, (scalar)fz, (scalar)0);
        }
    }

// ---- 10. log quantities and virial ---------------------------------------------------------
// Miller indices (:417-422) and wave vector k = n_x b1 + n_y b2 + n_z b3 with b_i = 2 pi * reciprocal rows of the box
__device__ __forceinline__ void wave_vector(const MeshGeom &g, const unsigned int cell, double &kx, double &ky, double &kz)
    {
    const unsigned int wz = cell / (g.nx * g.ny);
    const unsigned int wy = (cell - wz * g.nx * g.ny) / g.nx;
    const unsigned int wx = cell % g.nx;
    int n0 = (int)wx, n1 = (int)wy, n2 = (int)wz;
    if (n0 >= (int)(g.nx / 2 + g.nx % 2)) n0 -= (int)g.nx;
    if (n1 >= (int)(g.ny / 2 + g.ny % 2)) n1 -= (int)g.ny;
    if (n2 >= (int)(g.nz / 2 + g.nz % 2)) n2 -= (int)g.nz;
    const double tp = 2.0 * M_PI;
    kx = tp * (n0 * g.binv[0][0] + n1 * g.binv[1][0] + n2 * g.binv[2][0]);
    ky = tp * (n0 * g.binv[0][1] + n1 * g.binv[1][1] + n2 * g.binv[2][1]);
    kz = tp * (n0 * g.binv[0][2] + n1 * g.binv[1][2] + n2 * g.binv[2][2]);
    }

// computeQmax (:1108-1179): the cell with the largest |f|^2 (DC bin included like the reference), first index wins ties.
// On the stored half of the spectrum a cell stands for itself and for its mirror image -k (same |f|^2): the candidate
// index is the smaller of the two FULL-mesh linear indices, which is the cell the reference's strict `>` scan keeps.
__global__ __launch_bounds__(256) void k_mesh_argmax(const MeshGeom g, const double2 *__restrict__ fmesh, double *__restrict__ out_val,
                                                     unsigned int *__restrict__ out_idx)
    {
    __shared__ double s_v[4];
    __shared__ unsigned int s_i[4];
    double best = 0.0;
    unsigned int bi = 0xffffffffu;                    // "no cell yet": the reference keeps q_max = 0 when no amplitude exceeds 0
    const unsigned int hx = g.nx / 2 + 1;
    const unsigned int n_half = hx * g.ny * g.nz;
    for (unsigned int h = blockIdx.x * blockDim.x + threadIdx.x; h < n_half; h += gridDim.x * blockDim.x)
        {
        const unsigned int wx = h % hx, wy = (h / hx) % g.ny, wz = h / (hx * g.ny);
        const double2 f = fmesh[wx + (size_t)g.hxp * (wy + (size_t)g.ny * wz)];
        const double a = f.x * f.x + f.y * f.y;
        const unsigned int mx = (g.nx - wx) % g.nx, my = (g.ny - wy) % g.ny, mz = (g.nz - wz) % g.nz;
        const unsigned int k0 = wx + g.nx * (wy + g.ny * wz), k1 = mx + g.nx * (my + g.ny * mz);
        const unsigned int k = k0 < k1 ? k0 : k1;
        if (a > best || (a == best && a > 0.0 && k < bi))
            {
            best = a;
            bi = k;
            }
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        {
        const double ov = __shfl_xor(best, off, 64);
        const unsigned int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi))
            {
            best = ov;
            bi = oi;
            }
        }
    if ((threadIdx.x & 63) == 0)
        {
        s_v[threadIdx.x >> 6] = best;
        s_i[threadIdx.x >> 6] = bi;
        }
    __syncthreads();
    if (threadIdx.x == 0)
        {
        for (int w = 1; w < 4; ++w)
            if (s_v[w] > best || (s_v[w] == best && s_i[w] < bi))
                {
                best = s_v[w];
                bi = s_i[w];
                }
        out_val[blockIdx.x] = best;
        out_idx[blockIdx.x] = bi;
        }
    }

// computeVirial (:970-1050): sum over k != 0 of |f|^4 / N^2 * K'(|k|) / (2 |k|) * k_a k_b with K' from the derivative table
// (zero outside [k_min, k_max) and without a table); six block partial sums, fixed order
__global__ __launch_bounds__(256) void k_mesh_virial(const MeshGeom g, const double2 *__restrict__ fmesh, const double n_global,
                                                     const double *__restrict__ table_d, const double k_min, const double k_max,
                                                     const double delta_k, const int use_table, double *__restrict__ partials)
    {
    __shared__ double s_red[16];
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const unsigned int hx = g.nx / 2 + 1;
    const unsigned int n_half = hx * g.ny * g.nz;
    for (unsigned int h = blockIdx.x * blockDim.x + threadIdx.x; h < n_half; h += gridDim.x * blockDim.x)
        {
        const unsigned int wx = h % hx, wy = (h / hx) % g.ny, wz = h / (hx * g.ny);
        const unsigned int k = wx + g.nx * (wy + g.ny * wz);            // full-mesh index of the stored cell
        if (k == 0) continue;                                           // exclude DC bin (:1003-1005)
        const double2 f = fmesh[wx + (size_t)g.hxp * (wy + (size_t)g.ny * wz)];
        const double a = f.x * f.x + f.y * f.y;
        const double rhog = a * (a / n_global) / n_global;             // :1034-1035 (f is already F / N: the reference divides again)
        // the stored cell, then its mirror image when that one is not stored: same |f|^2, but its OWN wave vector — on the
        // Nyquist planes of y and z the Miller index of the mirror cell is not the negative one (both are -n/2)
        const unsigned int mx = (g.nx - wx) % g.nx, my = (g.ny - wy) % g.ny, mz = (g.nz - wz) % g.nz;
        const unsigned int cells[2] = {k, mx + g.nx * (my + g.ny * mz)};
        const int n_terms = (wx != 0 && 2 * wx != g.nx) ? 2 : 1;
        for (int term = 0; term < n_terms; ++term)
            {
            double kx, ky, kz;
            wave_vector(g, cells[term], kx, ky, kz);
            const double knorm = sqrt(kx * kx + ky * ky + kz * kz);
            double val_D = 0.0;
            if (use_table && knorm >= k_min && knorm < k_max)           // :1018-1030
                {
                const double value_f = (knorm - k_min) / delta_k;
                const unsigned int value_i = (unsigned int)value_f;
                const double dK0 = table_d[value_i], dK1 = table_d[value_i + 1];
                val_D = dK0 + (value_f - (double)value_i) * (dK1 - dK0);
                }
            const double kfac = 1.0 / 2.0 / knorm * val_D;
            v[0] += rhog * kfac * kx * kx;
            v[1] += rhog * kfac * kx * ky;
            v[2] += rhog * kfac * kx * kz;
            v[3] += rhog * kfac * ky * ky;
            v[4] += rhog * kfac * ky * kz;
            v[5] += rhog * kfac * kz * kz;
            }
        }
#pragma unroll
    for (int c = 0; c < 6; ++c)
        {
        const double r = block_sum(v[c], s_red);
        if (threadIdx.x == 0) partials[blockIdx.x * 6 + c] = r;
        }
    }

bool is_pow2(unsigned int n) { return n && !(n & (n - 1)); }
// log2 of a power of two, 0 otherwise (the kernels take 0 as "direct DFT, natural order")
unsigned int ilog2(unsigned int n)
    {
    if (!is_pow2(n)) return 0;
    unsigned int l = 0;
    while ((1u << l) < n) ++l;
    return l;
    }

// dynamic LDS of a line pass: one buffer of n * tile elements, two for the direct transform
size_t fft_lds_bytes(unsigned int n, unsigned int tile) { return (size_t)n * tile * sizeof(double2) * (is_pow2(n) ? 1 : 2); }
// x passes: rows padded by one slot on the power-of-two path (k_fft_x_r2c)
size_t fft_x_lds_bytes(unsigned int n, unsigned int pairs) { return is_pow2(n) ? (size_t)n * (pairs + 1) * sizeof(double2) : fft_lds_bytes(n, pairs); }

} // namespace

struct mtd_mesh
    {
    unsigned int nx, ny, nz, M, n_types, max_particles;
    unsigned int hxp;          // row pitch of the half-spectrum arrays d_f, d_g (nx/2 + 1 rounded up)
    int bug_compat;
    int keep_fourier;          // the fused z pass also writes the normalised Fourier mesh (log quantities, virial, get_array(1))
    int fourier_valid;         // d_f holds the Fourier mesh of the last spectral step
    hipEvent_t cv_event;       // recorded after the pass that completes the CV partial sums (mtd_mesh_set_cv_event), or null
    void *slab;
    double *d_mode, *d_rho, *d_inv, *d_modesq_partials, *d_mode_sq, *d_cv_partials, *d_cv_folded;
    double2 *d_f, *d_g, *d_tw[3];
    double4 *d_packed;
    unsigned int *d_cell_of, *d_slot_of, *d_count, *d_start, *d_tile_sums;
    unsigned int *d_tile_total;    // tile path: particles per tile (k_tile_rowscan); d_start then holds the row scans
    unsigned int *d_tile_first;    // first slot of every tile in the tile-ordered arrays (k_tile_place)
    uint2 *d_idcell;           // (particle id, cell) of every sorted slot
    unsigned int n_last;   // particle count of the last compute_cv (the sorted list the force pass walks)
    // convolution-kernel table (setTable, :148-189): K is stored and never applied (Q7); K' enters the virial
    double *d_table, *d_table_d, *d_log_scratch, *d_itab;
    unsigned int n_table;
    double k_min, k_max, delta_k;
    int use_table;
    unsigned int n_cv_partials, n_count_blocks;
    // tile path of the assignment / force pass (k_tile_*): tile geometry, per-tile fixed-point buffers, ids grouped by tile
    int tile_path;
    TileGeom tg;               // n_blocks, chunk, scale: as set by the last mtd_mesh_assign
    unsigned int tile_blocks_max;
    long long *d_tilebuf;
    bool combine_two;          // no mesh coordinate has three sources: k_tile_combine_rows
    uint4 *d_tsrc;             // per-axis table of the tile-buffer offsets that stand for a mesh coordinate (k_tile_combine)
    unsigned int *d_ids;
    void *d_possorted;             // raw position records in tile order (k_tile_place_sorted): 32 bytes per particle hold either precision
    // bin pipeline (k_tile_bin): segments with slack planned from the previous snapshot's exact counts, two sets (the force pass of a
    // snapshot reads the set its assignment used while the next one is being planned)
    unsigned int *d_plan_first[2], *d_plan_cap[2], *d_cursor[2], *d_ovf_count[2], *d_ovf_tile;
    unsigned int ovf_base;          // first slot of the overflow list in ids / possorted / packed (= all segments' slots at the most)
    int plan_valid, bin_parity;     // set `bin_parity` is planned (for plan_n particles) and its cursors are zero
    unsigned int plan_n;
    TileLists lists;                // where the last assignment left the tiles' particles (the force pass walks the same lists)
    int last_pipeline;              // of the last assignment: 0 cells, 1 counting, 2 bin (mtd_mesh_assign_info)
    int last_forward;               // of the last forward transform: 0 separate x / y passes, 1 k_fft_xy_forward on the combined mesh, 2 on the tile images
    int rho_valid;                  // the real mesh d_rho holds the last assignment (0: it lives in the tile images only — mtd_mesh_compute_cv
                                    // lets the forward transform read those; anyone who needs d_rho runs the combine pass first: mesh_need_rho)
    double amax;               // max |mode coefficient| (fixed-point scale)
    // slab decomposition over the ranks of a mailbox (mtd_mesh_slab_attach): exported buffers of every rank as mapped here
    struct mtd_comm *slab_comm;
    unsigned int slab_world, slab_rank;
    const void *slab_rho[MTD_COMM_MAX_RANKS], *slab_f[MTD_COMM_MAX_RANKS], *slab_g[MTD_COMM_MAX_RANKS], *slab_inv[MTD_COMM_MAX_RANKS];
    double *d_slab_rho;        // this rank's reduced slab (device memory of its own)
    double *d_slab_sum;        // [0] CV integrand of this rank's pencils -> sum over ranks, [1] barrier token
    // riders of the next assignment's first kernel (mtd_mesh_set_lamellar_rider): device copy of the arguments, host bookkeeping
    CountRider *d_rider;
    CountRider *h_rider;       // what d_rider holds (copied again only when something changed: a box, a grid, a mode set)
    int rider_armed;
    int rider_dirty;                // h_rider is newer than d_rider
    int rider_fast;                 // hardware trigonometry for the rider's mode set (lam_fast_trig)
    unsigned int rider_n_apply;
    struct mtd_metad *rider_engine;
    };

namespace
{

int fill_geom(MeshGeom &g, const mtd_mesh *m, const mtd_box *box)
    {
    if (!box || !(box->L[0] > 0.0) || !(box->L[1] > 0.0) || !(box->L[2] > 0.0)) return MTD_ERR_INVALID_ARGUMENT;
    std::memset(&g, 0, sizeof(g));
    g.nx = m->nx; g.ny = m->ny; g.nz = m->nz; g.n_cells = m->M;
    g.hxp = m->hxp;
    for (int i = 0; i < 3; ++i)
        {
        g.lo[i] = box->lo[i];
        g.L[i] = box->L[i];
        }
    g.xy = box->xy; g.xz = box->xz; g.yz = box->yz;
    for (int i = 0; i < 3; ++i) g.dL[i] = make_exact_divisor(box->L[i]);
    g.dn[0] = make_exact_divisor((double)m->nx); g.dn[1] = make_exact_divisor((double)m->ny); g.dn[2] = make_exact_divisor((double)m->nz);
    reciprocal_rows(*box, g.binv);
    return MTD_SUCCESS;
    }

// axes: bit a set = transform axis a (0 x, 1 y, 2 z); forward order x, y, z; inverse order z, y, x with the real part of the
// last pass written to real_out when given
struct FftPass { unsigned int n, tile, elem_stride, line_stride, tiles_per_row, row_stride, n_blocks; int p_fastest; const double2 *tw; };

unsigned int fft_tile_for(unsigned int n, unsigned int lines)
    {
    unsigned int t = 16;
    while (t > 1 && fft_lds_bytes(n, t) > 64 * 1024) t >>= 1;
    while (t > 1 && lines % t) t >>= 1;
    return t;
    }

// geometry of the line passes over the half-spectrum arrays (rows of pitch hxp along x)
FftPass fft_y_pass(const mtd_mesh *m)
    {
    FftPass py;
    py.n = m->ny; py.tile = fft_tile_for(m->ny, m->hxp); py.elem_stride = m->hxp; py.line_stride = 1; py.tiles_per_row = m->hxp / py.tile;
    py.row_stride = m->hxp * m->ny; py.n_blocks = py.tiles_per_row * m->nz; py.p_fastest = 0; py.tw = m->d_tw[1];
    return py;
    }

// tiles per block of the fused z pass of the whole-mesh path (k_fft_z_spectral<false, TPB>)
unsigned int fft_z_tpb(const FftPass &pz)
    {
    // MEASURED SLOWER, opt-in (MTD_FFT_Z_TPB=2|3): config 3 at 128^3 takes 163.9 / 167.0 us per step with two / three tiles per
    // block against 159.4 with one — the pass wants its 1152 small blocks all resident (4.5 per CU) more than it wants each block
    // to overlap its own phases
    static const int forced = [] { const char *e = std::getenv("MTD_FFT_Z_TPB"); return e && *e ? std::atoi(e) : 1; }();
    const bool fits = pz.n * pz.tile <= (unsigned int)FFT_ZPRE * FFT_THREADS;
    if (forced <= 1 || !fits) return 1;
    if (forced == 3 && pz.n_blocks % 3 == 0) return 3;
    return pz.n_blocks % 2 == 0 ? 2 : 1;
    }

FftPass fft_z_pass(const mtd_mesh *m)
    {
    FftPass pz;
    pz.n = m->nz; pz.tile = fft_tile_for(m->nz, m->hxp); pz.elem_stride = m->hxp * m->ny; pz.line_stride = 1; pz.tiles_per_row = m->hxp / pz.tile;
    pz.row_stride = m->hxp; pz.n_blocks = pz.tiles_per_row * m->ny; pz.p_fastest = 0; pz.tw = m->d_tw[2];
    return pz;
    }

int launch_fft_y(const mtd_mesh *m, double2 *data, int inverse, hipStream_t s)
    {
    const FftPass p = fft_y_pass(m);
    k_fft_lines<false, false><<<p.n_blocks, FFT_THREADS, fft_lds_bytes(p.n, p.tile), s>>>(
        nullptr, data, nullptr, p.tw, p.n, ilog2(p.n), p.tile, p.elem_stride, p.line_stride, p.tiles_per_row, p.row_stride, inverse, p.p_fastest);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

FastDiv fast_div(unsigned int d)
    {
    FastDiv f;
    f.d = d;
    f.magic = d > 1 ? (unsigned int)(((1ull << 32) + d - 1) / d) : 0u;
    return f;
    }

// plan of the fused x/y passes (k_fft_xy_*), or false when the mesh does not qualify (sizes that are not powers of two,
// planes whose images do not fit the 160 KB of LDS) and the separate passes run
constexpr size_t XY_LDS_MAX = 160 * 1024;
// the forward transform can read the tile images itself (k_fft_xy_forward<true>): tiles that divide the axes, powers of two, at
// least two cells wide (a coordinate then has at most two sources per axis), a thread's x position fixed over its elements
bool xy_tiles_ok(const mtd_mesh *m, XYTiles &tl)
    {
    // The default where the shapes allow it (MTD_FFT_FROM_TILES=0 switches it off): config 3 takes 129.3 / 128.9 us per step with it
    // against 135.0 / 135.1 on the same box (alternating processes, profiles/r4/mesh_ab.log), the combine launch (10.1 us) is gone.
    // History: with one cell per lane and eight-byte loads the transform grew by exactly the combine launch's time (29.2 us against
    // 19.3: 6 loads per cell pair where the plain kernel has 2, in the launch whose load phase was already exposed); a form with
    // conditional loads and adds was 30 us SLOWER (the compiler waited for them one by one).  Two cells per lane and 16-byte loads
    // (k_fft_xy_forward<true>) halve the load instructions.  Same bits in every form (tests run both).
    const char *ft_env = std::getenv("MTD_FFT_FROM_TILES");            // (read per call: a test runs both forms in one process)
    const bool off = ft_env && ft_env[0] == '0';
    if (off || !m->tile_path || !m->combine_two) return false;
    const TileGeom &tg = m->tg;
    const unsigned int dims[3] = {m->nx, m->ny, m->nz}, tws[3] = {tg.tx, tg.ty, tg.tz}, nts[3] = {tg.ntx, tg.nty, tg.ntz};
    if ((unsigned int)XY_THREADS % m->nx || tg.ty < 4) return false;    // (ty >= 4: only ONE row of a line pair can border a tile)
    std::memset(&tl, 0, sizeof(tl));
    for (int a = 0; a < 3; ++a)
        {
        if (tws[a] < 2 || !is_pow2(tws[a]) || dims[a] % tws[a] || nts[a] * tws[a] != dims[a]) return false;
        tl.log2t[a] = ilog2(tws[a]);
        tl.nt[a] = nts[a];
        }
    const unsigned long long tile_mul[3] = {1ull * tg.hcells, 1ull * tg.ntx * tg.hcells, 1ull * tg.ntx * tg.nty * tg.hcells};
    const unsigned long long loc_mul[3] = {1ull, tg.hx, 1ull * tg.hx * tg.hy};
    if (tile_mul[2] * tg.ntz >= (1ull << 32)) return false;            // (32-bit offsets, like the combine pass's table)
    for (int a = 0; a < 3; ++a)
        {
        tl.tile_mul[a] = (unsigned int)tile_mul[a];
        tl.loc_mul[a] = (unsigned int)loc_mul[a];
        }
    tl.buf = m->d_tilebuf;
    tl.inv_scale = tg.inv_scale;
    return true;
    }

bool xy_plan(const mtd_mesh *m, int inverse, XYPlan &pl, size_t &lds)
    {
    if (!is_pow2(m->nx) || !is_pow2(m->ny) || m->nx < 4 || m->ny < 2 * XY_PARTS) return false;
    std::memset(&pl, 0, sizeof(pl));
    pl.nx = m->nx; pl.ny = m->ny; pl.nz = m->nz; pl.log2nx = ilog2(m->nx); pl.log2ny = ilog2(m->ny);
    pl.hx = m->nx / 2 + 1; pl.hxp = m->hxp;
    pl.kc_max = (pl.hx + XY_PARTS - 1) / XY_PARTS;
    pl.ys = pl.kc_max | 1u;
    for (unsigned int part = 0; part < XY_PARTS; ++part)
        {
        const unsigned int k0 = part * pl.kc_max, k1 = std::min(pl.hx, k0 + pl.kc_max);
        if (k1 <= k0) return false;
        pl.d_kc[part] = fast_div(k1 - k0);
        }
    const size_t y_bytes = (size_t)pl.ny * pl.ys * sizeof(double2);
    unsigned int pb = m->ny / 2 / (inverse ? XY_PARTS : 1);      // forward: the largest batch of line pairs that fits
    const size_t tw_bytes = (size_t)(pl.nx / 2 + pl.ny / 2) * sizeof(double2);
    while (!inverse && pb > 1 && ((size_t)pl.nx * (pb + 1) * sizeof(double2) + y_bytes + tw_bytes > XY_LDS_MAX || (size_t)pl.nx * pb > (size_t)XY_PREFETCH * XY_THREADS)) pb >>= 1;
    pl.pb = pb;
    pl.xs = pb + 1;
    pl.d_pb = fast_div(pb);
    pl.xcd_map = m->nz % 8 == 0;
    lds = (size_t)pl.nx * pl.xs * sizeof(double2) + y_bytes + (size_t)(pl.nx / 2 + pl.ny / 2) * sizeof(double2);
    // a batch must fit the registers that prefetch it
    const size_t per_batch = inverse ? (size_t)pl.kc_max * pl.ny : (size_t)pl.nx * pb;
    return lds <= XY_LDS_MAX && per_batch <= (size_t)XY_PREFETCH * XY_THREADS;
    }

// plan of the split forward transform from the tile images (k_fft_xy_forward_split): 128 x 128 planes, two x tiles of 64 cells,
// tile height 4 or 8 (the row class of a lane's line pairs must depend on its wave only)
bool xy_split_plan_f(const mtd_mesh *m, XYSplitF &pl, size_t &lds)
    {
    if (m->nx != 128 || m->ny != 128 || m->tg.tx != 64 || (m->tg.ty != 4 && m->tg.ty != 8)) return false;
    std::memset(&pl, 0, sizeof(pl));
    pl.hxp = m->hxp;
    pl.xs = 65;
    pl.ys = 33;
    pl.d_kc[0] = fast_div(33);
    pl.d_kc[1] = fast_div(32);
    pl.d_np = fast_div(64);
    pl.xcd_map = m->nz % 8 == 0;
    lds = ((size_t)64 * pl.xs + (size_t)128 * pl.ys + 64 + 32 + 64) * sizeof(double2);
    return lds <= XY_LDS_MAX;
    }

// plan of the split inverse (k_fft_xy_inverse_split), or false where the shapes do not allow it (the unsplit kernel runs)
bool xy_split_plan(const mtd_mesh *m, XYSplit &pl, size_t &lds)
    {
    if (!is_pow2(m->nx) || !is_pow2(m->ny) || m->nx < 8 || m->ny < 16) return false;
    std::memset(&pl, 0, sizeof(pl));
    pl.nx = m->nx; pl.ny = m->ny; pl.nz = m->nz; pl.log2nx = ilog2(m->nx); pl.log2ny = ilog2(m->ny);
    pl.hx = m->nx / 2 + 1; pl.hxp = m->hxp;
    pl.rows = m->ny / 2;
    pl.pairs = pl.rows / 2;
    pl.xs = pl.pairs | 1u;
    pl.ys = pl.hx | 1u;
    pl.d_hx = fast_div(pl.hx);
    pl.d_pairs = fast_div(pl.pairs);
    pl.xcd_map = m->nz % 8 == 0;
    lds = ((size_t)pl.rows * pl.ys + (size_t)pl.nx * pl.xs + pl.nx / 2 + pl.ny / 2 + pl.ny / 4) * sizeof(double2);
    return lds <= XY_LDS_MAX && (size_t)pl.hx * pl.rows <= (size_t)XYS_SLOTS * XY_THREADS && (size_t)pl.hx * pl.rows * pl.hx < (1ull << 32);
    }

} // namespace

#ifdef MTD_STAMPS
extern "C" int mtd_debug_read_count_stamps(unsigned long long *host)
    {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cnt_stamps), sizeof(unsigned long long) * 8 * 1024);
    }
extern "C" int mtd_debug_read_tile_stamps(unsigned long long *host)
    {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tile_stamps), sizeof(unsigned long long) * 2 * 8 * 1024);
    }
extern "C" int mtd_debug_read_xy_stamps(unsigned long long *host)
    {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_xy_stamps), sizeof(unsigned long long) * 2 * 16 * 256);
    }
#endif

extern "C" {

int mtd_mesh_create(mtd_mesh **out, unsigned int nx, unsigned int ny, unsigned int nz, const double *mode,
                    unsigned int n_types, unsigned int max_particles)
    {
    if (!out || !mode || n_types == 0) return MTD_ERR_INVALID_ARGUMENT;
    if (nx == 0 || ny == 0 || nz == 0) return MTD_ERR_INVALID_ARGUMENT;
    // at least 4 cells per axis (3x3x3 stencils must not alias); powers of two up to 1024 take the radix-2 transforms, any
    // other size up to 256 the direct one
    for (unsigned int n : {nx, ny, nz})
        if (n < 4 || n > 1024 || (!is_pow2(n) && n > 256)) return MTD_ERR_UNSUPPORTED;
    const unsigned long long M64 = (unsigned long long)nx * ny * nz;
    if (M64 > (1ull << 30)) return MTD_ERR_UNSUPPORTED;
    mtd_mesh *m = new (std::nothrow) mtd_mesh();
    if (!m) return (int)hipErrorOutOfMemory;
    std::memset(m, 0, sizeof(*m));
    m->nx = nx; m->ny = ny; m->nz = nz; m->M = (unsigned int)M64;
    m->n_types = n_types;
    m->max_particles = max_particles;
    m->bug_compat = 1;
    m->keep_fourier = 1;
    const size_t M = m->M, N = max_particles;
    m->n_count_blocks = 4096;
    {
    const unsigned int hx = nx / 2 + 1, unit = nx < 8 ? nx : 8;     // (a pitch of 80 with 16-column tiles measured slower: y 10.5 -> 12.3, z 17.1 -> 20.3 us)
    m->hxp = (hx + unit - 1) / unit * unit;                    // 72 at nx = 128
    }
    const size_t MH = (size_t)m->hxp * ny * nz;
    m->n_cv_partials = fft_z_pass(m).n_blocks;                 // one partial sum per block of the fused z pass (an upper bound: the
                                                               // whole-mesh path puts several tiles into a block, fft_z_tpb)
    // tile path (k_tile_*): tiles of 64x8x8 cells clamped to the mesh; MTD_MESH_ASSIGN=cells keeps the cell-level pipeline
    {
    TileGeom &tg = m->tg;
    tg.tx = nx < (unsigned int)TP_X ? nx : TP_X; tg.ty = ny < (unsigned int)TP_Y ? ny : TP_Y; tg.tz = nz < (unsigned int)TP_Z ? nz : TP_Z;
    // a block per tile: a small mesh in 64x8x8 tiles gives the scatter and force passes fewer blocks than there are compute
    // units (64^3: 128); halve the longest tile edge (not below 8) until there are at least two blocks per CU or 8^3 is reached
    auto count_tiles = [&](const TileGeom &t) { return (unsigned long long)((nx + t.tx - 1) / t.tx) * ((ny + t.ty - 1) / t.ty) * ((nz + t.tz - 1) / t.tz); };
    // (more, smaller tiles than that cost more than they bring: 128^3 in 2048 / 4096 tiles 191.7 / 208.6 us per step against 171.9)
    while (count_tiles(tg) < 512 && (tg.tx > 8 || tg.ty > 8 || tg.tz > 8))
        {
        if (tg.tx >= tg.ty && tg.tx >= tg.tz && tg.tx > 8) tg.tx /= 2;
        else if (tg.ty >= tg.tz && tg.ty > 8) tg.ty /= 2;
        else if (tg.tz > 8) tg.tz /= 2;
        else if (tg.tx > 8) tg.tx /= 2;
        else tg.ty /= 2;
        }
    tg.ntx = (nx + tg.tx - 1) / tg.tx; tg.nty = (ny + tg.ty - 1) / tg.ty; tg.ntz = (nz + tg.tz - 1) / tg.tz;
    const unsigned long long nt = (unsigned long long)tg.ntx * tg.nty * tg.ntz;
    tg.hx = tg.tx + 2; tg.hy = tg.ty + 2; tg.hz = tg.tz + 2; tg.hcells = tg.hx * tg.hy * tg.hz;
    const char *env = std::getenv("MTD_MESH_ASSIGN");
    m->tile_path = nt <= TP_MAX_TILES && n_types <= 65536 && !(env && std::strcmp(env, "cells") == 0);   // (16 bits of a record hold the type)
    tg.n_tiles = m->tile_path ? (unsigned int)nt : 0;
    unsigned int nb = (max_particles + 4095) / 4096;
    m->tile_blocks_max = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    m->amax = 0.0;
    for (unsigned int t = 0; t < n_types; ++t) m->amax = std::fmax(m->amax, std::fabs(mode[t]));
    }
    const size_t n_scan = std::max<size_t>(m->M, (size_t)m->tg.n_tiles * m->tile_blocks_max);   // entries the scan kernels may see
    const unsigned int n_tiles = (unsigned int)((n_scan + SCAN_TILE - 1) / SCAN_TILE);
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    // tile-ordered arrays (ids, position records, force records): all segments' slots of the bin pipeline + an overflow list of N
    const size_t n_slots_extra = m->tile_path ? tile_capacity_total_max(N, m->tg.n_tiles) : 0;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += al(b); return o; };
    const size_t o_mode = take(sizeof(double) * n_types), o_rho = take(sizeof(double) * (M + 1)), o_msqp = take(sizeof(double) * m->n_count_blocks), o_cvp = take(sizeof(double) * m->n_cv_partials), o_cvf = take(sizeof(double) * ((size_t)nz * XY_PARTS + 1)), o_f = take(sizeof(double2) * MH),
                 o_g = take(sizeof(double2) * MH), o_tw0 = take(sizeof(double2) * nx), o_tw1 = take(sizeof(double2) * ny),
                 o_tw2 = take(sizeof(double2) * nz), o_packed = take(sizeof(double4) * (N + n_slots_extra)), o_cell = take(sizeof(unsigned int) * N),
                 o_count = take(sizeof(unsigned int) * (n_scan + 1)), o_start = take(sizeof(unsigned int) * (n_scan + 1)),
                 o_ids = take(sizeof(uint2) * N), o_tiles = take(sizeof(unsigned int) * n_tiles),
                 o_inv = take(sizeof(double) * M), o_slot = take(sizeof(unsigned int) * N),
                 o_itab = take(sizeof(double) * (nx + ny + nz)),
                 o_tilebuf = take(sizeof(long long) * (size_t)m->tg.n_tiles * m->tg.hcells), o_ids2 = take(sizeof(unsigned int) * (N + n_slots_extra)),
                 o_tsrc = take(sizeof(uint4) * (nx + ny + nz)), o_ttot = take(sizeof(unsigned int) * 2 * ((size_t)m->tg.n_tiles + 1)),
                 o_psort = take(m->tile_path ? sizeof(double4) * (N + n_slots_extra) : 0),
                 o_plan = take(m->tile_path ? sizeof(unsigned int) * (4 * (size_t)m->tg.n_tiles + 2 * (size_t)TB_TILES_PER_MAX * TB_THREADS * TB_CSTRIDE + 2 * TB_CSTRIDE) : 0),
                 o_ovft = take(m->tile_path ? sizeof(unsigned int) * N : 0);
    hipError_t e = hipMalloc(&m->slab, off);
    if (e != hipSuccess)
        {
        delete m;
        return (int)e;
        }
    if (const char *tr = std::getenv("MTD_TRACE_ALLOC")) if (tr[0] == '1') fprintf(stderr, "[mtd] mesh %ux%ux%u slab %p .. %p (%zu bytes)\n", nx, ny, nz, m->slab, (char *)m->slab + off, off);
    char *p = (char *)m->slab;
    m->d_mode = (double *)(p + o_mode); m->d_rho = (double *)(p + o_rho); m->d_modesq_partials = (double *)(p + o_msqp);
    m->d_mode_sq = m->d_rho + M;   // directly behind the real mesh: one exchange buffer of M + 1 doubles
    m->d_cv_partials = (double *)(p + o_cvp); m->d_cv_folded = (double *)(p + o_cvf); m->d_f = (double2 *)(p + o_f);
    m->d_g = (double2 *)(p + o_g); m->d_tw[0] = (double2 *)(p + o_tw0); m->d_tw[1] = (double2 *)(p + o_tw1);
    m->d_tw[2] = (double2 *)(p + o_tw2); m->d_packed = (double4 *)(p + o_packed); m->d_cell_of = (unsigned int *)(p + o_cell);
    m->d_count = (unsigned int *)(p + o_count); m->d_start = (unsigned int *)(p + o_start); m->d_idcell = (uint2 *)(p + o_ids);
    m->d_tile_sums = (unsigned int *)(p + o_tiles);
    m->d_inv = (double *)(p + o_inv);
    m->d_slot_of = (unsigned int *)(p + o_slot);
    m->d_itab = (double *)(p + o_itab);
    m->d_tilebuf = (long long *)(p + o_tilebuf);
    m->d_ids = (unsigned int *)(p + o_ids2);
    m->d_possorted = m->tile_path ? (void *)(p + o_psort) : nullptr;
    if (m->tile_path)
        {
        const size_t T = m->tg.n_tiles;
        unsigned int *q = (unsigned int *)(p + o_plan);
        for (int i = 0; i < 2; ++i) { m->d_plan_first[i] = q; q += T; m->d_plan_cap[i] = q; q += T; }
        for (int i = 0; i < 2; ++i) { m->d_cursor[i] = q; q += (size_t)TB_TILES_PER_MAX * TB_THREADS * TB_CSTRIDE; }   // (with the idle lanes' cursors)
        for (int i = 0; i < 2; ++i) { m->d_ovf_count[i] = q; q += TB_CSTRIDE; }
        m->d_ovf_tile = (unsigned int *)(p + o_ovft);
        m->ovf_base = (unsigned int)n_slots_extra;
        }
    m->plan_valid = 0; m->bin_parity = 0; m->plan_n = 0; m->last_pipeline = 0; m->rho_valid = 1;
    std::memset(&m->lists, 0, sizeof(m->lists));
    m->d_tsrc = (uint4 *)(p + o_tsrc);
    m->d_tile_total = (unsigned int *)(p + o_ttot);
    m->d_tile_first = m->d_tile_total + m->tg.n_tiles + 1;
    e = hipMemset(m->slab, 0, off);
    if (e == hipSuccess && m->tile_path)
        {
        // entry offset = tile * hcells + lx + hx (ly + hy lz) with tile = tx + ntx (ty + nty tz): one term per axis
        std::vector<unsigned int> tab(4 * (size_t)(nx + ny + nz), 0u);
        m->combine_two = true;
        const TileGeom &tg = m->tg;
        const unsigned int dims[3] = {nx, ny, nz}, tws[3] = {tg.tx, tg.ty, tg.tz}, nts[3] = {tg.ntx, tg.nty, tg.ntz};
        const unsigned long long tile_mul[3] = {1ull * tg.hcells, 1ull * tg.ntx * tg.hcells, 1ull * tg.ntx * tg.nty * tg.hcells};
        const unsigned long long loc_mul[3] = {1ull, tg.hx, 1ull * tg.hx * tg.hy};
        size_t o = 0;
        for (int a = 0; a < 3; ++a)
            for (unsigned int c = 0; c < dims[a]; ++c, ++o)
                {
                const unsigned int n = dims[a], tw = tws[a], nt = nts[a];
                const unsigned int t0 = c / tw, first = t0 * tw, width = std::min(tw, n - first);
                unsigned int k = 0;
                auto put = [&](unsigned int tile, unsigned int loc) { tab[4 * o + k++] = (unsigned int)(tile * tile_mul[a] + loc * loc_mul[a]); };
                put(t0, c - first + 1);
                if (c == first)
                    {
                    const unsigned int tl = t0 == 0 ? nt - 1 : t0 - 1;
                    put(tl, std::min(tw, n - tl * tw) + 1);
                    }
                if (c == first + width - 1) put(t0 == nt - 1 ? 0 : t0 + 1, 0);
                tab[4 * o + 3] = k;
                if (k > 2) m->combine_two = false;
                }
        e = hipMemcpy(m->d_tsrc, tab.data(), sizeof(unsigned int) * tab.size(), hipMemcpyHostToDevice);
        }
    if (e == hipSuccess) e = hipMemcpy(m->d_mode, mode, sizeof(double) * n_types, hipMemcpyHostToDevice);
    // twiddles exp(-2 pi i j / n), j < n (the radix-2 stages use the first half), in double on the host
    const unsigned int dims[3] = {nx, ny, nz};
    for (int a = 0; a < 3 && e == hipSuccess; ++a)
        {
        std::vector<double> tw(2 * (size_t)dims[a], 0.0);
        for (unsigned int j = 0; j < dims[a]; ++j)
            {
            const double ang = -2.0 * M_PI * (double)j / (double)dims[a];
            tw[2 * j] = std::cos(ang);
            tw[2 * j + 1] = std::sin(ang);
            }
        e = hipMemcpy(m->d_tw[a], tw.data(), sizeof(double) * tw.size(), hipMemcpyHostToDevice);
        }
    if (e != hipSuccess)
        {
        (void)hipFree(m->slab);
        delete m;
        return (int)e;
        }
    k_interp_tables<<<(nx + ny + nz + 255) / 256, 256>>>(nx, ny, nz, m->bug_compat, m->d_itab);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess)
        {
        (void)hipFree(m->slab);
        delete m;
        return (int)e;
        }
    *out = m;
    return MTD_SUCCESS;
    }

int mtd_mesh_destroy(mtd_mesh *m)
    {
    if (!m) return MTD_SUCCESS;
    if (m->d_slab_rho) (void)hipFree(m->d_slab_rho);
    if (m->d_slab_sum) (void)hipFree(m->d_slab_sum);
    if (m->d_table) (void)hipFree(m->d_table);
    if (m->d_log_scratch) (void)hipFree(m->d_log_scratch);
    if (m->d_rider) (void)hipFree(m->d_rider);
    delete m->h_rider;
    hipError_t e = hipFree(m->slab);
    delete m;
    return (int)e;
    }

int mtd_mesh_set_lamellar_rider(mtd_mesh *mesh, mtd_metad *engine, const mtd_lamellar_set *set, const mtd_box *global_box,
                                unsigned int n_particles, double *d_partials, unsigned int *n_partials, mtd_stream_t stream)
    {
    if (!mesh || !set || !d_partials || !n_partials) return MTD_ERR_INVALID_ARGUMENT;
    if (!mesh->tile_path) return MTD_ERR_UNSUPPORTED;              // the cell-level pipeline has no such pass: mtd_fused_cv_pass
    if (set->n_cv == 0 || set->n_cv > 3) return MTD_ERR_UNSUPPORTED;
    if (engine && engine->comm) return MTD_ERR_UNSUPPORTED;         // a sharded step sends its sums from launch A of the fused step
    CountRider r;
    std::memset(&r, 0, sizeof(r));
    int rc = mtd::fill_kargs(r.k, set, global_box);
    if (rc) return rc;
    unsigned int nb = (n_particles + 4095) / 4096;                  // as mesh_assign_local
    nb = nb < 1 ? 1 : (nb > mesh->tile_blocks_max ? mesh->tile_blocks_max : nb);
    r.partials = d_partials;
    r.n_apply = 0;
    if (engine)
        {
        r.cfg = engine->cfg;
        std::memset(r.cfg.src, 0, sizeof(r.cfg.src));               // (not read by apply_cells; keeps the comparison below quiet)
        if (engine->pending_apply) r.n_apply = (engine->cfg.len + 255) / 256;
        }
    if (!mesh->d_rider)
        {
        MTD_HIP_TRY(hipMalloc(&mesh->d_rider, sizeof(CountRider)));
        mesh->h_rider = new (std::nothrow) CountRider();
        if (!mesh->h_rider) return (int)hipErrorOutOfMemory;
        std::memset(mesh->h_rider, 0xff, sizeof(CountRider));
        }
    // (the device copy is made by the assignment that needs one — the counting pipeline; the bin pipeline takes the arguments by value)
    (void)stream;
    if (std::memcmp(mesh->h_rider, &r, sizeof(r)) != 0)
        {
        *mesh->h_rider = r;
        mesh->rider_dirty = 1;
        }
    mesh->rider_armed = 1;
    mesh->rider_fast = mtd::lam_fast_trig(r.k);
    mesh->rider_n_apply = r.n_apply;
    mesh->rider_engine = engine;
    *n_partials = nb;
    return MTD_SUCCESS;
    }

int mtd_mesh_clear_rider(mtd_mesh *mesh, int *was_armed)
    {
    if (!mesh) return MTD_ERR_INVALID_ARGUMENT;
    if (was_armed) *was_armed = mesh->rider_armed;
    mesh->rider_armed = 0;
    return MTD_SUCCESS;
    }

int mtd_mesh_set_cv_event(mtd_mesh *m, void *hip_event)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    m->cv_event = (hipEvent_t)hip_event;
    return MTD_SUCCESS;
    }

int mtd_mesh_set_keep_fourier(mtd_mesh *m, int on)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    m->keep_fourier = on ? 1 : 0;
    return MTD_SUCCESS;
    }

int mtd_mesh_set_bug_compat(mtd_mesh *m, int on)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    m->bug_compat = on ? 1 : 0;
    k_interp_tables<<<(m->nx + m->ny + m->nz + 255) / 256, 256>>>(m->nx, m->ny, m->nz, m->bug_compat, m->d_itab);
    MTD_LAUNCH_CHECK();
    MTD_HIP_TRY(hipDeviceSynchronize());
    return MTD_SUCCESS;
    }

// Raise the dynamic-LDS limit of a group of kernels once per DEVICE (a function attribute belongs to the device that is current when
// it is set; a process that drives several GPUs sets it on each).  A runtime that refuses leaves the caller its fallback path.
static bool dyn_lds_ok(const int group, const void *const *fns, const int n, const size_t bytes)
    {
    static std::mutex mu;
    static std::map<std::pair<int, int>, bool> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, group);
    auto it = done.find(key);
    if (it != done.end()) return it->second;
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    done[key] = e == hipSuccess;
    return e == hipSuccess;
    }

static int mesh_assign_local(mtd_mesh *m, unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box, mtd_stream_t stream,
                             bool combine = true);

// per-tile images -> the real mesh (tile path); afterwards d_rho holds the last assignment
static int mesh_combine(mtd_mesh *m, const MeshGeom &g, hipStream_t s)
    {
    const unsigned int cthreads = m->nx >= 256 ? 256 : (m->nx > 64 ? 128 : 64);
    if (m->combine_two)
        k_tile_combine_rows<<<dim3((m->nx + cthreads - 1) / cthreads, (m->ny + TCB_ROWS - 1) / TCB_ROWS, m->nz), cthreads, 0, s>>>(g, m->tg, m->d_tilebuf, m->d_tsrc, m->d_rho);
    else
        k_tile_combine<<<dim3((m->nx + cthreads - 1) / cthreads, m->ny, m->nz), cthreads, 0, s>>>(g, m->tg, m->d_tilebuf, m->d_tsrc, m->d_rho);
    MTD_LAUNCH_CHECK();
    m->rho_valid = 1;
    return MTD_SUCCESS;
    }

// whoever reads d_rho (the separate transform passes, the replicated-mesh exchange, mtd_mesh_get_array(0)) after an assignment that
// left the mesh in the tile images only
static int mesh_need_rho(mtd_mesh *m, hipStream_t s)
    {
    if (m->rho_valid || !m->tile_path) return MTD_SUCCESS;
    MeshGeom g;
    std::memset(&g, 0, sizeof(g));
    g.nx = m->nx; g.ny = m->ny; g.nz = m->nz; g.hxp = m->hxp;           // (the combine pass reads the dimensions only)
    return mesh_combine(m, g, s);
    }

int mtd_mesh_assign(mtd_mesh *m, unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box, mtd_stream_t stream)
    {
    return mesh_assign_local(m, n_particles, d_postype, dtype, box, stream, true);
    }

static int mesh_assign_local(mtd_mesh *m, unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box, mtd_stream_t stream,
                             bool combine)
    {
    if (!m || (n_particles && !d_postype)) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    if (n_particles > m->max_particles) return MTD_ERR_INVALID_ARGUMENT;
    MeshGeom g;
    int rc = fill_geom(g, m, box);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const unsigned int M = m->M, N = n_particles;
    const unsigned int cell_blocks = (M + 255) / 256;
    const unsigned int n_tiles = (M + SCAN_TILE - 1) / SCAN_TILE;

    if (m->tile_path)
        {
        TileGeom &tg = m->tg;
        unsigned int nb = (N + 4095) / 4096;
        nb = nb < 1 ? 1 : (nb > m->tile_blocks_max ? m->tile_blocks_max : nb);
        tg.n_blocks = nb;
        tg.chunk = N ? (N + nb - 1) / nb : 1;
        // fixed point: 2^k * N * max|a| < 2^62 even if every particle sat in one cell
        int k = 40;
        if (m->amax > 0.0)
            {
            k = (int)std::floor(62.0 - std::log2((double)(N ? N : 1) * m->amax)) - 1;
            const int k_one = 50 - (int)std::ceil(std::log2(m->amax));        // one deposit stays below 2^51 (mantissa rounding)
            k = k > k_one ? k_one : k;
            k = k > 60 ? 60 : (k < -900 ? -900 : k);
            }
        tg.scale = std::ldexp(1.0, k);
        tg.inv_scale = std::ldexp(1.0, -k);
        const bool f32 = dtype == MTD_F32;
        // ---- bin pipeline (steady state): one launch bins, sorts and stores the chunk; the scatter launch's extra block plans the next
        // snapshot.  Needs a plan for this particle number (the first assignment of a mesh goes through the counting pipeline and
        // plans from its exact counts), no riders (they travel in the counting / row-scan kernels), a chunk and tables that fit the LDS.
        const char *bin_env = std::getenv("MTD_MESH_BIN");               // (read per call: tests run both pipelines in one process)
        const bool bin_off = bin_env && bin_env[0] == '0';
        const size_t tb_lds = f32 ? tile_bin_lds_bytes<float4>(tg.n_tiles, tg.chunk) : tile_bin_lds_bytes<double4>(tg.n_tiles, tg.chunk);
#define MTD_BIN_FNS(S4, R) (const void *)k_tile_bin<S4, 1, R>, (const void *)k_tile_bin<S4, 2, R>, (const void *)k_tile_bin<S4, 4, R>, (const void *)k_tile_bin<S4, 8, R>
        static const void *const bin_fns[24] = { MTD_BIN_FNS(float4, 0), MTD_BIN_FNS(float4, 1), MTD_BIN_FNS(float4, 2), MTD_BIN_FNS(double4, 0), MTD_BIN_FNS(double4, 1), MTD_BIN_FNS(double4, 2) };
#undef MTD_BIN_FNS
        const bool tb_lds_ok = dyn_lds_ok(0, bin_fns, 24, TB_LDS_MAX);
        const bool bin_fits = !bin_off && tb_lds_ok && m->d_possorted && tg.chunk <= TPS_CHUNK_MAX && tb_lds <= TB_LDS_MAX &&
                              tg.n_tiles <= (unsigned int)(TB_TILES_PER_MAX * TB_THREADS);
        TilePlan plan;
        std::memset(&plan, 0, sizeof(plan));
        mtd::MetadCfg apply_cfg;                                         // (read by the scatter launch's passenger blocks only)
        std::memset(&apply_cfg, 0, sizeof(apply_cfg));
        // (riders of the counting kernel's own loop — MTD_MESH_RIDER=count — keep the counting pipeline)
        const char *rider_env = std::getenv("MTD_MESH_RIDER");
        const bool rider_in_count = rider_env && std::strcmp(rider_env, "count") == 0;
        // a plan made for another particle number still is a plan (a domain-decomposed run's local count changes every step: what does
        // not fit overflows); only a count that differs by a factor of two or more is counted and planned afresh
        const bool plan_usable = m->plan_valid && N >= m->plan_n / 2 && N / 2 <= m->plan_n;
        if (bin_fits && plan_usable && !(m->rider_armed && rider_in_count))
            {
            const int p = m->bin_parity;
            int rider_kind = 0;
            unsigned int n_apply = 0;
            if (m->rider_armed)
                {
                rider_kind = m->rider_fast ? 2 : 1;
                if (m->rider_n_apply)
                    {
                    apply_cfg = m->h_rider->cfg;
                    n_apply = (apply_cfg.len + TP_THREADS - 1) / TP_THREADS;
                    }
                m->rider_armed = 0;
                if (m->rider_engine && m->rider_n_apply) m->rider_engine->pending_apply = 0;
                }
            // no riders: an engine that announced a pending deferred pass on this stream (a mesh variable without lamellar CVs beside
            // it: mtd_mesh_forces_update_bias, mtd_metad_update_bias) still gets it carried by the scatter launch's extra blocks
            mtd_metad *passenger = nullptr;
            if (!n_apply && !rider_kind)
                {
                passenger = mtd::take_pending_apply(s, apply_cfg);
                if (passenger) n_apply = (apply_cfg.len + TP_THREADS - 1) / TP_THREADS;
                }
            const unsigned int tiles_per = (tg.n_tiles + TB_THREADS - 1) / TB_THREADS;
            BinRiderArgs ra;
            BinNoRider no_rider;
            if (rider_kind)
                {
                ra.k = mtd::dense_cv_args(m->h_rider->k);                // (the visited modes as a dense list: flat table loads)
                ra.partials = m->h_rider->partials;
                }
#define MTD_TILE_BIN(S4, PER, R, RA) \
            k_tile_bin<S4, PER, R><<<nb, TB_THREADS, tb_lds, s>>>(g, tg, (const S4 *)d_postype, N, m->d_mode, m->n_types, m->d_plan_first[p], m->d_plan_cap[p], \
                                                                  m->d_cursor[p], m->d_ovf_count[p], m->d_ovf_tile, m->ovf_base, m->d_ids, (S4 *)m->d_possorted, \
                                                                  m->d_modesq_partials, RA)
#define MTD_TILE_BIN_R(S4, PER) \
            do { if (rider_kind == 2) MTD_TILE_BIN(S4, PER, 2, ra); else if (rider_kind == 1) MTD_TILE_BIN(S4, PER, 1, ra); else MTD_TILE_BIN(S4, PER, 0, no_rider); } while (0)
            if (f32)
                {
                if (tiles_per <= 1) MTD_TILE_BIN_R(float4, 1); else if (tiles_per <= 2) MTD_TILE_BIN_R(float4, 2);
                else if (tiles_per <= 4) MTD_TILE_BIN_R(float4, 4); else MTD_TILE_BIN_R(float4, 8);
                }
            else
                {
                if (tiles_per <= 1) MTD_TILE_BIN_R(double4, 1); else if (tiles_per <= 2) MTD_TILE_BIN_R(double4, 2);
                else if (tiles_per <= 4) MTD_TILE_BIN_R(double4, 4); else MTD_TILE_BIN_R(double4, 8);
                }
#undef MTD_TILE_BIN_R
#undef MTD_TILE_BIN
            MTD_LAUNCH_CHECK();
            TileLists L;
            L.first = m->d_plan_first[p]; L.count = m->d_cursor[p]; L.cap = m->d_plan_cap[p]; L.ovf_count = m->d_ovf_count[p];
            L.ovf_tile = m->d_ovf_tile; L.cstride = TB_CSTRIDE; L.ovf_base = m->ovf_base;
            plan.count = m->d_cursor[p]; plan.cstride = TB_CSTRIDE; plan.n_tiles = tg.n_tiles;
            plan.first_next = m->d_plan_first[1 - p]; plan.cap_next = m->d_plan_cap[1 - p];
            plan.cursor_next = m->d_cursor[1 - p]; plan.ovf_count_next = m->d_ovf_count[1 - p]; plan.cstride_next = TB_CSTRIDE;
            plan.modesq_partials = m->d_modesq_partials; plan.n_partials = nb; plan.mode_sq = m->d_mode_sq;
            if (f32)
                k_tile_scatter<float4><<<tg.n_tiles + 1 + n_apply, TP_THREADS, 0, s>>>(g, tg, (const float4 *)d_postype, m->d_mode, L, m->d_ids, m->d_tilebuf, m->d_packed, m->n_types, (const float4 *)m->d_possorted, plan, apply_cfg);
            else
                k_tile_scatter<double4><<<tg.n_tiles + 1 + n_apply, TP_THREADS, 0, s>>>(g, tg, (const double4 *)d_postype, m->d_mode, L, m->d_ids, m->d_tilebuf, m->d_packed, m->n_types, (const double4 *)m->d_possorted, plan, apply_cfg);
            if (passenger && hipPeekAtLastError() == hipSuccess) mtd::commit_pending_apply(passenger);     // (a failed launch leaves the pass pending)
            m->lists = L;
            m->plan_n = N;
            m->last_pipeline = 2;
            m->bin_parity = 1 - p;                                       // (planned and zeroed by the extra block)
            }
        else
            {
        const size_t lds = sizeof(unsigned int) * tg.n_tiles;
        unsigned int n_apply_blocks = 0;
#define MTD_TILE_COUNT(S4, RIDER, FAST, GRID) \
        k_tile_count<S4, RIDER, FAST><<<GRID, TC_THREADS, lds, s>>>(g, tg, (const S4 *)d_postype, N, m->d_mode, m->d_cell_of, m->d_slot_of, \
                                                                    m->d_count, m->d_modesq_partials, m->n_types, m->d_rider)
        if (m->rider_armed)
            {
            if (m->rider_dirty)
                {
                MTD_HIP_TRY(hipMemcpyAsync(m->d_rider, m->h_rider, sizeof(CountRider), hipMemcpyHostToDevice, s));
                m->rider_dirty = 0;
                }
            const unsigned int grid = nb;
            const bool fast = m->rider_fast != 0;
            n_apply_blocks = m->rider_n_apply;
            if (dtype == MTD_F32) { if (fast) MTD_TILE_COUNT(float4, true, true, grid); else MTD_TILE_COUNT(float4, true, false, grid); }
            else { if (fast) MTD_TILE_COUNT(double4, true, true, grid); else MTD_TILE_COUNT(double4, true, false, grid); }
            m->rider_armed = 0;
            if (m->rider_engine && m->rider_n_apply) m->rider_engine->pending_apply = 0;
            }
        else if (dtype == MTD_F32)
            MTD_TILE_COUNT(float4, false, false, nb);
        else
            MTD_TILE_COUNT(double4, false, false, nb);
#undef MTD_TILE_COUNT
        MTD_LAUNCH_CHECK();
        k_tile_rowscan<<<tg.n_tiles + 1 + n_apply_blocks, 256, 0, s>>>(m->d_count, m->d_start, m->d_tile_total, tg.n_tiles, nb, m->d_modesq_partials, nb, m->d_mode_sq, m->d_rider);
        MTD_LAUNCH_CHECK();
        unsigned int pb = (N + 1023) / 1024;                        // >= four particles per thread: the LDS prefix of the tile totals is formed once per block
        pb = pb < 1 ? 1 : (pb > 512 ? 512 : pb);
        // sorted place (the default): raw position records and ids leave in tile order, in runs; MTD_MESH_PLACE=ids keeps the
        // one-store-per-particle form, which is also the fallback when a chunk or the tables do not fit
        const size_t ps_lds = f32 ? place_sorted_lds_bytes<float4>(tg.n_tiles, tg.chunk) : place_sorted_lds_bytes<double4>(tg.n_tiles, tg.chunk);
        const char *ps_env = std::getenv("MTD_MESH_PLACE");           // (read per call: a test runs both forms in one process)
        const bool ps_off = ps_env && std::strcmp(ps_env, "ids") == 0;
        static const void *const ps_fns[2] = { (const void *)k_tile_place_sorted<float4>, (const void *)k_tile_place_sorted<double4> };
        const bool ps_lds_ok = dyn_lds_ok(1, ps_fns, 2, PS_LDS_MAX);
        const bool sorted = !ps_off && ps_lds_ok && m->d_possorted && tg.chunk <= TPS_CHUNK_MAX && ps_lds <= PS_LDS_MAX;
        if (sorted)
            {
            if (f32)
                k_tile_place_sorted<float4><<<nb, TPS_THREADS, ps_lds, s>>>(tg, N, (const float4 *)d_postype, m->d_cell_of, m->d_slot_of, m->d_count, m->d_start, m->d_tile_total, m->d_ids, (float4 *)m->d_possorted, m->d_tile_first);
            else
                k_tile_place_sorted<double4><<<nb, TPS_THREADS, ps_lds, s>>>(tg, N, (const double4 *)d_postype, m->d_cell_of, m->d_slot_of, m->d_count, m->d_start, m->d_tile_total, m->d_ids, (double4 *)m->d_possorted, m->d_tile_first);
            }
        else if (f32)
            k_tile_place<float4><<<pb, 256, sizeof(unsigned int) * tg.n_tiles, s>>>(tg, N, m->d_cell_of, m->d_slot_of, m->d_start, m->d_tile_total, m->d_ids, m->d_tile_first);
        else
            k_tile_place<double4><<<pb, 256, sizeof(unsigned int) * tg.n_tiles, s>>>(tg, N, m->d_cell_of, m->d_slot_of, m->d_start, m->d_tile_total, m->d_ids, m->d_tile_first);
        MTD_LAUNCH_CHECK();
        TileLists L;
        std::memset(&L, 0, sizeof(L));
        L.first = m->d_tile_first; L.count = m->d_tile_total; L.cstride = 1;
        if (f32)
            k_tile_scatter<float4><<<tg.n_tiles, TP_THREADS, 0, s>>>(g, tg, (const float4 *)d_postype, m->d_mode, L, m->d_ids, m->d_tilebuf, m->d_packed, m->n_types, sorted ? (const float4 *)m->d_possorted : nullptr, plan, apply_cfg);
        else
            k_tile_scatter<double4><<<tg.n_tiles, TP_THREADS, 0, s>>>(g, tg, (const double4 *)d_postype, m->d_mode, L, m->d_ids, m->d_tilebuf, m->d_packed, m->n_types, sorted ? (const double4 *)m->d_possorted : nullptr, plan, apply_cfg);
        m->lists = L;
        m->last_pipeline = 1;
        if (bin_fits && !plan_usable)
            {
            // the exact counts of this snapshot plan the segments of the next one
            MTD_LAUNCH_CHECK();
            const int p = m->bin_parity;
            plan.count = m->d_tile_total; plan.cstride = 1; plan.n_tiles = tg.n_tiles;
            plan.first_next = m->d_plan_first[p]; plan.cap_next = m->d_plan_cap[p];
            plan.cursor_next = m->d_cursor[p]; plan.ovf_count_next = m->d_ovf_count[p]; plan.cstride_next = TB_CSTRIDE;
            k_tile_plan<<<1, 1024, 0, s>>>(plan);
            m->plan_valid = 1;
            m->plan_n = N;
            }
            }
        MTD_LAUNCH_CHECK();
        m->n_last = N;
        m->rho_valid = 0;
        if (combine) return mesh_combine(m, g, s);
        return MTD_SUCCESS;
        }

    // the counters are zero on entry: cleared at creation and by k_scan_finish of the previous call
    if (dtype == MTD_F32)
        k_mesh_bin<float4><<<m->n_count_blocks, 256, 0, s>>>(g, (const float4 *)d_postype, N, m->d_mode, m->d_cell_of, m->d_slot_of, m->d_count, m->d_modesq_partials);
    else
        k_mesh_bin<double4><<<m->n_count_blocks, 256, 0, s>>>(g, (const double4 *)d_postype, N, m->d_mode, m->d_cell_of, m->d_slot_of, m->d_count, m->d_modesq_partials);
    MTD_LAUNCH_CHECK();
    k_scan_tiles<<<n_tiles + 1, 256, 0, s>>>(m->d_count, m->d_start, m->d_tile_sums, M, m->d_modesq_partials, m->n_count_blocks, m->d_mode_sq);
    MTD_LAUNCH_CHECK();
    k_scan_finish<<<n_tiles, 256, 0, s>>>(m->d_start, m->d_tile_sums, m->d_count, M, N);
    MTD_LAUNCH_CHECK();
    if (dtype == MTD_F32)
        k_mesh_place<float4><<<m->n_count_blocks, 256, 0, s>>>(g, (const float4 *)d_postype, N, m->d_mode, m->d_cell_of, m->d_slot_of, m->d_start, m->d_idcell, m->d_packed);
    else
        k_mesh_place<double4><<<m->n_count_blocks, 256, 0, s>>>(g, (const double4 *)d_postype, N, m->d_mode, m->d_cell_of, m->d_slot_of, m->d_start, m->d_idcell, m->d_packed);
    MTD_LAUNCH_CHECK();
    k_mesh_sortfix<<<cell_blocks, 256, 0, s>>>(M, m->d_start, m->d_idcell, m->d_packed);
    MTD_LAUNCH_CHECK();
    GatherTiling tl;
    tl.tx = m->nx < (unsigned int)GT_X ? m->nx : GT_X;
    tl.ty = m->ny < (unsigned int)GT_Y ? m->ny : GT_Y;
    tl.tz = m->nz < (unsigned int)GT_Z ? m->nz : GT_Z;
    tl.ntx = (m->nx + tl.tx - 1) / tl.tx; tl.nty = (m->ny + tl.ty - 1) / tl.ty; tl.ntz = (m->nz + tl.tz - 1) / tl.tz;   // edge tiles may be partial
    k_mesh_gather<<<tl.ntx * tl.nty * tl.ntz, GT_THREADS, 0, s>>>(g, tl, m->d_start, m->d_packed, m->d_rho);
    MTD_LAUNCH_CHECK();
    m->n_last = N;
    m->rho_valid = 1;
    return MTD_SUCCESS;
    }

int mtd_mesh_assign_info(mtd_mesh *m, int *pipeline, unsigned int *n_overflow, mtd_stream_t stream)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    const int pl = m->tile_path ? m->last_pipeline : 0;
    if (pipeline) *pipeline = pl;
    if (n_overflow)
        {
        *n_overflow = 0;
        if (pl == 2 && m->lists.ovf_count)
            {
            MTD_HIP_TRY(hipMemcpyAsync(n_overflow, m->lists.ovf_count, sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream));
            MTD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            }
        }
    return MTD_SUCCESS;
    }

int mtd_mesh_transform_info(mtd_mesh *m, int *forward)
    {
    if (!m || !forward) return MTD_ERR_INVALID_ARGUMENT;
    *forward = m->last_forward;
    return MTD_SUCCESS;
    }

int mtd_mesh_exchange_buffer(mtd_mesh *m, double **d_buffer, size_t *count)
    {
    if (!m || !d_buffer || !count) return MTD_ERR_INVALID_ARGUMENT;
    *d_buffer = m->d_rho;
    *count = (size_t)m->M + 1;
    return MTD_SUCCESS;
    }

int mtd_mesh_spectral(mtd_mesh *m, const mtd_box *box, unsigned int n_global, const double **d_partials, unsigned int *n_partials,
                      mtd_stream_t stream)
    {
    if (!m || !d_partials || !n_partials || n_global == 0) return MTD_ERR_INVALID_ARGUMENT;
    MeshGeom g;
    int rc = fill_geom(g, m, box);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // x passes: two real lines per complex transform, `x_tile` real lines per block (the last block may run short)
    const unsigned int n_lines = m->ny * m->nz;
    // 8 pairs = 16 real lines per block: at nx = 128 that is 16.5 KB of LDS and 1024 blocks — four per compute unit, so that
    // one block loads while another transforms and a third stores (16 pairs: 512 blocks, 12.0 + 12.2 us; 8: 10.4 + 10.9)
    unsigned int x_pairs = 8;
    { static const unsigned int forced = [] { const char *e = std::getenv("MTD_FFT_XPAIRS"); return e ? (unsigned int)std::atoi(e) : 0u; }(); if (forced) x_pairs = forced; }
    while (x_pairs > 1 && fft_x_lds_bytes(m->nx, x_pairs) > 64 * 1024) x_pairs >>= 1;
    const unsigned int x_tile = 2 * x_pairs, x_blocks = (n_lines + x_tile - 1) / x_tile;
    const size_t x_lds = fft_x_lds_bytes(m->nx, x_pairs);
    // x and y of a plane in one launch where the plane fits the LDS (k_fft_xy_*); MTD_FFT_XY=0 keeps the separate passes
    XYPlan xy_f, xy_i;
    size_t xy_lds_f = 0, xy_lds_i = 0;
    static const bool xy_off = [] { const char *e = std::getenv("MTD_FFT_XY"); return e && e[0] == '0'; }();
    // (a runtime that refuses the 160 KB of dynamic LDS leaves the separate passes, it does not fail the step)
    static const void *const xy_fns[5] = { (const void *)k_fft_xy_forward<false>, (const void *)k_fft_xy_inverse, (const void *)k_fft_xy_forward<true>,
                                           (const void *)k_fft_xy_inverse_split, (const void *)k_fft_xy_forward_split };
    const bool xy_lds_ok = dyn_lds_ok(2, xy_fns, 5, XY_LDS_MAX);
    const bool xy = !xy_off && xy_lds_ok && xy_plan(m, 0, xy_f, xy_lds_f) && xy_plan(m, 1, xy_i, xy_lds_i);
    XYTiles tiles;
    std::memset(&tiles, 0, sizeof(tiles));
    // (the row class of a thread's elements must not change over its elements nor over the batches: k_fft_xy_forward<true>)
    // (two cells per lane: a wave spans a line of 128 cells; a thread's line pairs are 8 apart, the batches 32: multiples of the tile height)
    const bool tile_rows_ok = xy && m->nx == 128 && XY_THREADS == 512 && xy_f.pb == 32 && m->tg.ty && 16 % m->tg.ty == 0 && (2 * xy_f.pb) % m->tg.ty == 0;
    if (xy && tile_rows_ok && !m->rho_valid && xy_tiles_ok(m, tiles))
        {
        // the assignment left the mesh in the per-tile images (mtd_mesh_compute_cv): the transform sums them itself
        XYSplitF sf;
        size_t sf_lds = 0;
        const char *split_env = std::getenv("MTD_FFT_SPLIT");              // (read per call: a test runs both forms in one process)
        if (!(split_env && split_env[0] == '0') && xy_split_plan_f(m, sf, sf_lds))
            k_fft_xy_forward_split<<<m->nz * 2, XY_THREADS, sf_lds, s>>>(m->d_f, m->d_tw[0], m->d_tw[1], sf, tiles);
        else
            k_fft_xy_forward<true><<<m->nz * XY_PARTS, XY_THREADS, xy_lds_f, s>>>(nullptr, m->d_f, m->d_tw[0], m->d_tw[1], xy_f, tiles);
        MTD_LAUNCH_CHECK();
        m->last_forward = 2;
        }
    else if (xy)
        {
        rc = mesh_need_rho(m, s);
        if (rc) return rc;
        k_fft_xy_forward<false><<<m->nz * XY_PARTS, XY_THREADS, xy_lds_f, s>>>(m->d_rho, m->d_f, m->d_tw[0], m->d_tw[1], xy_f, tiles);
        MTD_LAUNCH_CHECK();
        m->last_forward = 1;
        }
    else
        {
        rc = mesh_need_rho(m, s);
        if (rc) return rc;
        k_fft_x_r2c<<<x_blocks, FFT_THREADS, x_lds, s>>>(m->d_rho, m->d_f, m->d_tw[0], m->nx, ilog2(m->nx), x_tile, m->hxp, n_lines);
        MTD_LAUNCH_CHECK();
        rc = launch_fft_y(m, m->d_f, 0, s);
        if (rc) return rc;
        m->last_forward = 0;
        }
    const FftPass pz = fft_z_pass(m);
    SlabArgs none;
    std::memset(&none, 0, sizeof(none));
    const unsigned int tpb = fft_z_tpb(pz);
    unsigned int z_blocks_whole = pz.n_blocks / tpb;
    // a half spectrum of 8 k + 1 columns: the lone last column goes to edge blocks instead of a ninth tile per row (k_fft_z_spectral)
    // MEASURED SLOWER, opt-in (MTD_FFT_Z_EDGE=1): config 3 at 128^3 takes 133.2 / 133.7 us per step with the edge blocks against 132.1 /
    // 132.2 without (alternating processes on one box, profiles/r4/mesh_ab.log): 112 blocks fewer, but the edge blocks gather 16-byte
    // elements a row pitch apart and look up their y factors per element
    static const bool edge_off = [] { const char *e = std::getenv("MTD_FFT_Z_EDGE"); return !(e && e[0] == '1'); }();
    const unsigned int hx_cols = m->nx / 2 + 1;
    const bool z_edge = !edge_off && tpb == 1 && pz.tile > 1 && hx_cols % pz.tile == 1 && (hx_cols / pz.tile + 1) == pz.tiles_per_row && m->ny >= pz.tile;
    if (z_edge)
        {
        const unsigned int full = hx_cols / pz.tile, n_regular = full * m->ny, n_edge = (m->ny + pz.tile - 1) / pz.tile;
        z_blocks_whole = n_regular + n_edge;
        k_fft_z_spectral<false, 1><<<z_blocks_whole, FFT_THREADS, fft_lds_bytes(pz.n, pz.tile), s>>>(
            g, m->d_f, m->d_g, pz.tw, ilog2(pz.n), pz.tile, full, m->d_mode_sq, (double)n_global, m->d_itab, m->d_cv_partials, none, m->keep_fourier,
            n_regular, full * pz.tile);
        }
    else if (tpb == 3)
        k_fft_z_spectral<false, 3><<<z_blocks_whole, FFT_THREADS, fft_lds_bytes(pz.n, pz.tile), s>>>(
            g, m->d_f, m->d_g, pz.tw, ilog2(pz.n), pz.tile, pz.tiles_per_row, m->d_mode_sq, (double)n_global, m->d_itab, m->d_cv_partials, none, m->keep_fourier, 0xffffffffu, 0u);
    else if (tpb == 2)
        k_fft_z_spectral<false, 2><<<z_blocks_whole, FFT_THREADS, fft_lds_bytes(pz.n, pz.tile), s>>>(
            g, m->d_f, m->d_g, pz.tw, ilog2(pz.n), pz.tile, pz.tiles_per_row, m->d_mode_sq, (double)n_global, m->d_itab, m->d_cv_partials, none, m->keep_fourier, 0xffffffffu, 0u);
    else
        k_fft_z_spectral<false, 1><<<z_blocks_whole, FFT_THREADS, fft_lds_bytes(pz.n, pz.tile), s>>>(
            g, m->d_f, m->d_g, pz.tw, ilog2(pz.n), pz.tile, pz.tiles_per_row, m->d_mode_sq, (double)n_global, m->d_itab, m->d_cv_partials, none, m->keep_fourier, 0xffffffffu, 0u);
    m->fourier_valid = m->keep_fourier;
    bool fold_cv = false;
    unsigned int n_folded = 0;
    if (m->cv_event) MTD_HIP_TRY(hipEventRecord(m->cv_event, s));          // the CV partial sums are complete from here on
    MTD_LAUNCH_CHECK();
    if (xy)
        {
        // (the CV's partial sums folded to one per block of this launch when that is fewer — and nobody was promised them earlier:
        // mtd_mesh_set_cv_event marks the z pass as the point where the sums are complete)
        const unsigned int xy_blocks = m->nz * XY_PARTS;
        fold_cv = !m->cv_event && z_blocks_whole > xy_blocks && z_blocks_whole <= 64 * xy_blocks;
        XYSplit xs_i;
        size_t xs_lds = 0;
        const char *split_env = std::getenv("MTD_FFT_SPLIT");              // (read per call: a test runs both forms in one process)
        const bool split = !(split_env && split_env[0] == '0') && xy_split_plan(m, xs_i, xs_lds);
        if (split)
            k_fft_xy_inverse_split<<<xy_blocks, XY_THREADS, xs_lds, s>>>(m->d_g, m->d_inv, m->d_tw[0], m->d_tw[1], xs_i, m->d_cv_partials,
                                                                         fold_cv ? z_blocks_whole : 0u, m->d_cv_folded);
        else
            k_fft_xy_inverse<<<xy_blocks, XY_THREADS, xy_lds_i, s>>>(m->d_g, m->d_inv, m->d_tw[0], m->d_tw[1], xy_i, m->d_cv_partials,
                                                                     fold_cv ? z_blocks_whole : 0u, m->d_cv_folded);   // Re(inv)
        MTD_LAUNCH_CHECK();
        if (fold_cv) n_folded = xy_blocks;
        }
    else
        {
        rc = launch_fft_y(m, m->d_g, 1, s);
        if (rc) return rc;
        k_fft_x_c2r<<<x_blocks, FFT_THREADS, x_lds, s>>>(m->d_g, m->d_inv, m->d_tw[0], m->nx, ilog2(m->nx), x_tile, m->hxp, n_lines);   // Re(inv)
        MTD_LAUNCH_CHECK();
        }
    *d_partials = fold_cv ? m->d_cv_folded : m->d_cv_partials;
    *n_partials = fold_cv ? n_folded : z_blocks_whole;
    return MTD_SUCCESS;
    }

int mtd_mesh_compute_cv(mtd_mesh *m, unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box,
                        unsigned int n_global, const double **d_partials, unsigned int *n_partials, mtd_stream_t stream)
    {
    if (!m || !d_partials || !n_partials || n_global == 0) return MTD_ERR_INVALID_ARGUMENT;
    // assignment and transforms in one call: the combine pass (tile images -> real mesh, a 10 us launch whose output the forward
    // transform would read back one launch later) is skipped where the transform can sum the tile images itself
    XYPlan xy_probe;
    XYTiles tl_probe;
    size_t lds_probe = 0;
    static const bool xy_off = [] { const char *e = std::getenv("MTD_FFT_XY"); return e && e[0] == '0'; }();
    const bool from_tiles = m->tile_path && !xy_off && xy_plan(m, 0, xy_probe, lds_probe) && xy_tiles_ok(m, tl_probe) && m->tg.ty &&
                            m->nx == 128 && XY_THREADS == 512 && xy_probe.pb == 32 && 16 % m->tg.ty == 0 && (2 * xy_probe.pb) % m->tg.ty == 0;
    int rc = mesh_assign_local(m, n_particles, d_postype, dtype, box, stream, !from_tiles);
    if (rc) return rc;
    return mtd_mesh_spectral(m, box, n_global, d_partials, n_partials, stream);
    }

int mtd_mesh_forces(mtd_mesh *m, unsigned int n_particles, const void *d_postype, void *d_force, int dtype, const mtd_box *box,
                    unsigned int n_global, const double *d_bias, double bias_host, mtd_stream_t stream)
    {
    if (!m || n_global == 0 || (n_particles && (!d_postype || !d_force))) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    if (n_particles == 0) return MTD_SUCCESS;
    MeshGeom g;
    int rc = fill_geom(g, m, box);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    unsigned int blocks = (n_particles + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const double two_over_n = 2.0 / (double)n_global;
    // the force pass walks the cell-sorted list built by the last mtd_mesh_compute_cv of the same snapshot
    if (n_particles != m->n_last) return MTD_ERR_INVALID_ARGUMENT;
    if (m->tile_path)
        {
        if (dtype == MTD_F32)
            k_tile_forces<float4><<<8 * ((m->tg.n_tiles + 7) / 8), TF_THREADS, 0, s>>>(g, m->tg, m->lists, m->d_mode, m->d_packed, m->d_inv, (float4 *)d_force, d_bias, bias_host, two_over_n, m->n_types);
        else
            k_tile_forces<double4><<<8 * ((m->tg.n_tiles + 7) / 8), TF_THREADS, 0, s>>>(g, m->tg, m->lists, m->d_mode, m->d_packed, m->d_inv, (double4 *)d_force, d_bias, bias_host, two_over_n, m->n_types);
        MTD_LAUNCH_CHECK();
        return MTD_SUCCESS;
        }
    if (dtype == MTD_F32)
        k_mesh_forces<float4><<<blocks, 256, 0, s>>>(g, n_particles, m->d_idcell, m->d_packed, m->d_inv, (float4 *)d_force, d_bias, bias_host, two_over_n);
    else
        k_mesh_forces<double4><<<blocks, 256, 0, s>>>(g, n_particles, m->d_idcell, m->d_packed, m->d_inv, (double4 *)d_force, d_bias, bias_host, two_over_n);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

// The force pass of a mixed set with the bias-grid engine's launch inside (k_tile_forces_chain): what mtd_fused_force_pass_slots
// followed by mtd_mesh_forces does, in one launch.  MTD_ERR_UNSUPPORTED where the shapes do not allow it (the caller then makes the
// two calls): cell-level pipeline, a sharded engine, more than three grid variables, more grid blocks than tiles.
int mtd_mesh_forces_update_bias(mtd_mesh *mesh, mtd_metad *m, unsigned int mesh_slot, const mtd_lamellar_set *set, const unsigned int *slots,
                                unsigned int n_particles, const void *d_postype, void *d_force_mesh, void *const *d_force_lamellar,
                                int dtype, unsigned int n_global, const mtd_box *box, unsigned int timestep, mtd_stream_t stream)
    {
    const unsigned int n_lam = set ? set->n_cv : 0u;               // (no set: the mesh variable alone, or beside variables of other kinds)
    if (!mesh || !m || n_global == 0 || !box || (n_lam && (!slots || !d_force_lamellar))) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    if (n_particles == 0 || !d_postype || !d_force_mesh) return MTD_ERR_UNSUPPORTED;
    static const bool off = [] { const char *e = std::getenv("MTD_MESH_FORCE_MERGED"); return e && e[0] == '0'; }();
    if (off || !mesh->tile_path || m->comm) return MTD_ERR_UNSUPPORTED;
    if (m->cfg.n_cv > (unsigned int)mtd::CHAIN_MAX_CV || mesh_slot >= m->cfg.n_cv) return MTD_ERR_UNSUPPORTED;
    if (n_lam > 3 || n_lam >= m->cfg.n_cv) return MTD_ERR_UNSUPPORTED;
    if (n_particles != mesh->n_last) return MTD_ERR_INVALID_ARGUMENT;    // (the force pass walks the last assignment's lists)
    if (m->h_step_err && *m->h_step_err) return MTD_ERR_COMM_TIMEOUT;
    mtd::LamKArgs k;
    int rc = MTD_SUCCESS;
    if (n_lam)
        rc = mtd::fill_kargs(k, set, box);
    else
        std::memset(&k, 0, sizeof(k));                             // no modes, no CVs: the launch's streaming part has nothing to do
    if (rc) return rc;
    for (unsigned int cv = 0; cv < n_lam; ++cv)
        {
        if (slots[cv] >= m->cfg.n_cv || slots[cv] == mesh_slot) return MTD_ERR_INVALID_ARGUMENT;
        if (!d_force_lamellar[cv]) return MTD_ERR_INVALID_ARGUMENT;
        k.slot[cv] = (unsigned char)slots[cv];
        }
    const int dep = (m->add_bias && (timestep % m->stride == 0)) ? 1 : 0;   // .cc:368
    const unsigned int n_grid = dep ? m->cfg.n_gblocks : 0;
    const unsigned int blocks = 8 * ((mesh->tg.n_tiles + 7) / 8);
    if (n_grid > blocks) return MTD_ERR_UNSUPPORTED;
    if ((unsigned long long)blocks * TFC_STREAM_THREADS * TFC_U >= (1ull << 31)) return MTD_ERR_UNSUPPORTED;
    MeshGeom g;
    rc = fill_geom(g, mesh, box);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = mtd::metad_flush(m, s);                                 // a deposit may only be pending across ONE cv pass
    if (rc) return rc;
    mtd::ForcePtrs out;
    for (unsigned int cv = 0; cv < MTD_MAX_CV; ++cv) out.f[cv] = cv < n_lam ? d_force_lamellar[cv] : nullptr;
    const double two_over_n = 2.0 / (double)n_global;
    const bool fast = n_lam ? mtd::lam_fast_trig(k) != 0 : true;
    const unsigned int n_stream = n_lam ? n_particles : 0u;        // particles the blocks stream for the lamellar forces
#define MTD_LAUNCH_TFC(S4, NCV, FASTV) \
    k_tile_forces_chain<S4, NCV, FASTV><<<blocks, TF_THREADS, 0, s>>>(g, mesh->tg, mesh->lists, mesh->d_mode, mesh->d_packed, mesh->d_inv, (S4 *)d_force_mesh, \
        two_over_n, mesh->n_types, mesh_slot, k, (const S4 *)d_postype, out, n_stream, m->cfg, dep, n_grid)
#define MTD_LAUNCH_TFC_NCV(S4, FASTV) \
    switch (n_lam) { case 0: case 1: MTD_LAUNCH_TFC(S4, 1, FASTV); break; case 2: MTD_LAUNCH_TFC(S4, 2, FASTV); break; default: MTD_LAUNCH_TFC(S4, 3, FASTV); break; }
    if (dtype == MTD_F32)
        {
        if (fast) { MTD_LAUNCH_TFC_NCV(float4, true) } else { MTD_LAUNCH_TFC_NCV(float4, false) }
        }
    else
        {
        if (fast) { MTD_LAUNCH_TFC_NCV(double4, true) } else { MTD_LAUNCH_TFC_NCV(double4, false) }
        }
#undef MTD_LAUNCH_TFC_NCV
#undef MTD_LAUNCH_TFC
    MTD_LAUNCH_CHECK();
    m->pending_apply = dep;
    m->w_stale = dep;
    if (dep) mtd::announce_pending_apply(m, s);                  // the next assignment's scatter launch (or a rider) takes the deferred pass along
    return MTD_SUCCESS;
    }

// ---- slab-decomposed mesh (replaces the ghost-cell exchange + dfftlib calls of OrderParameterMesh.cc:263-316, 659-746)
int mtd_mesh_slab_bytes(const mtd_mesh *m, unsigned int world, size_t *bytes)
    {
    if (!m || !bytes || world == 0 || world > MTD_COMM_MAX_RANKS) return MTD_ERR_INVALID_ARGUMENT;
    if (m->nz % world || m->ny % world) return MTD_ERR_UNSUPPORTED;
    bytes[0] = sizeof(double) * (size_t)m->M;                                          // local assignment, whole mesh
    bytes[1] = sizeof(double2) * (size_t)(m->nz / world) * m->ny * m->hxp;             // slab after the x and y passes
    bytes[2] = sizeof(double2) * (size_t)m->nz * (m->ny / world) * m->hxp;             // pencils after the inverse z pass
    bytes[3] = sizeof(double) * (size_t)(m->nz / world) * m->ny * m->nx;               // slab of Re(inv)
    return MTD_SUCCESS;
    }

int mtd_mesh_slab_attach(mtd_mesh *m, mtd_comm *comm, void *const *rho_peers, void *const *f_peers, void *const *g_peers,
                         void *const *inv_peers)
    {
    if (!m || !comm || !rho_peers || !f_peers || !g_peers || !inv_peers) return MTD_ERR_INVALID_ARGUMENT;
    const unsigned int world = mtd_comm_world(comm);
    if (m->nz % world || m->ny % world || !m->tile_path) return MTD_ERR_UNSUPPORTED;
    m->slab_comm = comm;
    m->slab_world = world;
    m->slab_rank = mtd_comm_rank(comm);
    for (unsigned int r = 0; r < world; ++r)
        {
        if (!rho_peers[r] || !f_peers[r] || !g_peers[r] || !inv_peers[r]) return MTD_ERR_INVALID_ARGUMENT;
        m->slab_rho[r] = rho_peers[r]; m->slab_f[r] = f_peers[r]; m->slab_g[r] = g_peers[r]; m->slab_inv[r] = inv_peers[r];
        }
    if (!m->d_slab_rho) MTD_HIP_TRY(hipMalloc((void **)&m->d_slab_rho, sizeof(double) * (size_t)(m->nz / world) * m->ny * m->nx));
    if (!m->d_slab_sum)
        {
        MTD_HIP_TRY(hipMalloc((void **)&m->d_slab_sum, sizeof(double) * 2));
        MTD_HIP_TRY(hipMemset(m->d_slab_sum, 0, sizeof(double) * 2));
        }
    return MTD_SUCCESS;
    }

// One call per step and rank; four exchanges (bounded waits of the mailbox) separate the phases:
//   local assignment -> [all ranks assigned] -> pull + sum this rank's slab, x and y transforms -> [all slabs transformed;
//   carries sum mode^2] -> z lines gathered from all slabs, spectral step, inverse z -> [all pencils done; carries the CV
//   sum] -> pull this rank's slab of G, inverse y and x -> [all slabs of Re(inv) done] -> pull the whole Re(inv)
// *d_cv_sum (device) = sum over ranks of the CV integrand (the CV is half of it, as for mtd_mesh_compute_cv's partial sums).
int mtd_mesh_slab_compute_cv(mtd_mesh *m, unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box,
                             unsigned int n_global, const double **d_cv_sum, mtd_stream_t stream)
    {
    if (!m || !m->slab_comm || !d_cv_sum || n_global == 0) return MTD_ERR_INVALID_ARGUMENT;
    MeshGeom g;
    int rc = fill_geom(g, m, box);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const unsigned int W = m->slab_world, r = m->slab_rank, nzl = m->nz / W, nyl = m->ny / W;
    const size_t slab_cells = (size_t)nzl * m->ny * m->nx;
    double *rho_x = (double *)m->slab_rho[r];
    double2 *f_x = (double2 *)m->slab_f[r], *g_x = (double2 *)m->slab_g[r];
    double *inv_x = (double *)m->slab_inv[r];
    PeerPtrs pp;
    auto peers_of = [&](const void *const *src) { std::memset(&pp, 0, sizeof(pp)); for (unsigned int q = 0; q < W; ++q) pp.p[q] = src[q]; };

    // 1. local assignment of this rank's particles onto the whole mesh, exported
    rc = mesh_assign_local(m, n_particles, d_postype, dtype, box, stream);
    if (rc) return rc;
    k_copy_doubles<<<1024, 256, 0, s>>>(m->d_rho, rho_x, (size_t)m->M);
    MTD_LAUNCH_CHECK();
    rc = mtd_comm_allreduce_small(m->slab_comm, m->d_slab_sum + 1, 1, stream);                 // barrier
    if (rc) return rc;
    // 2. this rank's slab: x and y transforms in memory of its own (nothing is transformed in place in an exported buffer:
    //    plain loads from one may hit stale L2 lines, comm.hip), the result copied out
    peers_of(m->slab_rho);
    k_slab_pull_rho<<<1024, 256, 0, s>>>(pp, W, (size_t)r * slab_cells, slab_cells, m->d_slab_rho);
    MTD_LAUNCH_CHECK();
    const unsigned int n_lines = m->ny * nzl;
    unsigned int x_pairs = 8;
    while (x_pairs > 1 && fft_x_lds_bytes(m->nx, x_pairs) > 64 * 1024) x_pairs >>= 1;
    const unsigned int x_tile = 2 * x_pairs, x_blocks = (n_lines + x_tile - 1) / x_tile;
    const size_t x_lds = fft_x_lds_bytes(m->nx, x_pairs);
    k_fft_x_r2c<<<x_blocks, FFT_THREADS, x_lds, s>>>(m->d_slab_rho, m->d_f, m->d_tw[0], m->nx, ilog2(m->nx), x_tile, m->hxp, n_lines);
    MTD_LAUNCH_CHECK();
    FftPass py = fft_y_pass(m);
    py.n_blocks = py.tiles_per_row * nzl;
    k_fft_lines<false, false><<<py.n_blocks, FFT_THREADS, fft_lds_bytes(py.n, py.tile), s>>>(
        nullptr, m->d_f, nullptr, py.tw, py.n, ilog2(py.n), py.tile, py.elem_stride, py.line_stride, py.tiles_per_row, py.row_stride, 0, py.p_fastest);
    MTD_LAUNCH_CHECK();
    k_copy_doubles<<<1024, 256, 0, s>>>((const double *)m->d_f, (double *)f_x, 2 * (size_t)nzl * m->ny * m->hxp);
    MTD_LAUNCH_CHECK();
    rc = mtd_comm_allreduce_small(m->slab_comm, m->d_mode_sq, 1, stream);                       // barrier + global sum mode^2 (:630)
    if (rc) return rc;
    // 3. z lines of this rank's y rows from all slabs, spectral step, inverse z: pencils exported
    const FftPass pz = fft_z_pass(m);
    SlabArgs sl;
    std::memset(&sl, 0, sizeof(sl));
    for (unsigned int q = 0; q < W; ++q) sl.f[q] = (const double2 *)m->slab_f[q];
    sl.world = W; sl.nz_loc = nzl; sl.ny_loc = nyl; sl.y0 = r * nyl;
    const unsigned int z_blocks = pz.tiles_per_row * nyl;
    k_fft_z_spectral<true, 1><<<z_blocks, FFT_THREADS, fft_lds_bytes(pz.n, pz.tile), s>>>(
        g, nullptr, g_x, pz.tw, ilog2(pz.n), pz.tile, pz.tiles_per_row, m->d_mode_sq, (double)n_global, m->d_itab, m->d_cv_partials, sl, 0, 0xffffffffu, 0u);
    m->fourier_valid = 0;
    MTD_LAUNCH_CHECK();
    rc = mtd_reduce_partials(m->d_cv_partials, z_blocks, 1, 1, 1.0, 0.0, m->d_slab_sum, stream);
    if (rc) return rc;
    rc = mtd_comm_allreduce_small(m->slab_comm, m->d_slab_sum, 1, stream);                      // barrier + CV sum
    if (rc) return rc;
    // 4. this rank's slab of G from the pencils, inverse y and x: slab of Re(inv) exported
    peers_of(m->slab_g);
    k_slab_pull_g<<<1024, 256, 0, s>>>(pp, r * nzl, nzl, m->ny, nyl, m->hxp, m->d_g);
    MTD_LAUNCH_CHECK();
    k_fft_lines<false, false><<<py.n_blocks, FFT_THREADS, fft_lds_bytes(py.n, py.tile), s>>>(
        nullptr, m->d_g, nullptr, py.tw, py.n, ilog2(py.n), py.tile, py.elem_stride, py.line_stride, py.tiles_per_row, py.row_stride, 1, py.p_fastest);
    MTD_LAUNCH_CHECK();
    k_fft_x_c2r<<<x_blocks, FFT_THREADS, x_lds, s>>>(m->d_g, inv_x, m->d_tw[0], m->nx, ilog2(m->nx), x_tile, m->hxp, n_lines);
    MTD_LAUNCH_CHECK();
    rc = mtd_comm_allreduce_small(m->slab_comm, m->d_slab_sum + 1, 1, stream);                 // barrier
    if (rc) return rc;
    // 5. the whole Re(inv) for the local force pass
    peers_of(m->slab_inv);
    k_slab_pull_inv<<<1024, 256, 0, s>>>(pp, W, slab_cells, m->d_inv);
    MTD_LAUNCH_CHECK();
    *d_cv_sum = m->d_slab_sum;
    return MTD_SUCCESS;
    }

int mtd_mesh_set_table(mtd_mesh *m, const double *K, const double *d_K, unsigned int n, double k_min, double k_max)
    {
    if (!m || !K || !d_K || n < 2) return MTD_ERR_INVALID_ARGUMENT;
    if (k_min < 0 || k_max < 0 || k_max <= k_min) return MTD_ERR_INVALID_ARGUMENT;       // :153-158
    if (m->d_table) MTD_HIP_TRY(hipFree(m->d_table));
    m->d_table = nullptr;
    MTD_HIP_TRY(hipMalloc((void **)&m->d_table, sizeof(double) * 2 * n));
    m->d_table_d = m->d_table + n;
    MTD_HIP_TRY(hipMemcpy(m->d_table, K, sizeof(double) * n, hipMemcpyHostToDevice));
    MTD_HIP_TRY(hipMemcpy(m->d_table_d, d_K, sizeof(double) * n, hipMemcpyHostToDevice));
    m->n_table = n;
    m->k_min = k_min;
    m->k_max = k_max;
    m->delta_k = (k_max - k_min) / (double)(n - 1);                                       // :169
    return MTD_SUCCESS;
    }

int mtd_mesh_set_use_table(mtd_mesh *m, int use_table)
    {
    if (!m) return MTD_ERR_INVALID_ARGUMENT;
    if (use_table && !m->d_table) return MTD_ERR_INVALID_ARGUMENT;
    m->use_table = use_table ? 1 : 0;
    return MTD_SUCCESS;
    }

static int log_scratch(mtd_mesh *m)
    {
    if (!m->d_log_scratch) MTD_HIP_TRY(hipMalloc((void **)&m->d_log_scratch, sizeof(double) * 8 * 256));
    return MTD_SUCCESS;
    }

int mtd_mesh_qmax(mtd_mesh *m, const mtd_box *box, unsigned int n_global, double *out, mtd_stream_t stream)
    {
    if (!m || !out || n_global == 0) return MTD_ERR_INVALID_ARGUMENT;
    if (!m->fourier_valid) return MTD_ERR_INVALID_ARGUMENT;          // mtd_mesh_set_keep_fourier(1) before the spectral step
    MeshGeom g;
    int rc = fill_geom(g, m, box);
    if (rc) return rc;
    rc = log_scratch(m);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const unsigned int blocks = 256;
    double *d_val = m->d_log_scratch;
    unsigned int *d_idx = (unsigned int *)(m->d_log_scratch + 256);
    k_mesh_argmax<<<blocks, 256, 0, s>>>(g, m->d_f, d_val, d_idx);
    MTD_LAUNCH_CHECK();
    double val[256];
    unsigned int idx[256];
    MTD_HIP_TRY(hipMemcpyAsync(val, d_val, sizeof(val), hipMemcpyDeviceToHost, s));
    MTD_HIP_TRY(hipMemcpyAsync(idx, d_idx, sizeof(idx), hipMemcpyDeviceToHost, s));
    MTD_HIP_TRY(hipStreamSynchronize(s));
    double best = 0.0;
    unsigned int bi = 0xffffffffu;
    for (unsigned int b = 0; b < blocks; ++b)
        if (val[b] > best || (val[b] == best && idx[b] < bi))
            {
            best = val[b];
            bi = idx[b];
            }
    out[0] = out[1] = out[2] = 0.0;
    if (bi != 0xffffffffu && best > 0.0)
        {
        const unsigned int wz = bi / (m->nx * m->ny), wy = (bi - wz * m->nx * m->ny) / m->nx, wx = bi % m->nx;
        int n0 = (int)wx, n1 = (int)wy, n2 = (int)wz;
        if (n0 >= (int)(m->nx / 2 + m->nx % 2)) n0 -= (int)m->nx;
        if (n1 >= (int)(m->ny / 2 + m->ny % 2)) n1 -= (int)m->ny;
        if (n2 >= (int)(m->nz / 2 + m->nz % 2)) n2 -= (int)m->nz;
        for (int d = 0; d < 3; ++d) out[d] = 2.0 * M_PI * (n0 * g.binv[0][d] + n1 * g.binv[1][d] + n2 * g.binv[2][d]);
        }
    out[3] = best * (double)n_global;                                                      // :1174-1178
    return MTD_SUCCESS;
    }

int mtd_mesh_virial(mtd_mesh *m, const mtd_box *box, unsigned int n_global, double bias, double *virial, mtd_stream_t stream)
    {
    if (!m || !virial || n_global == 0) return MTD_ERR_INVALID_ARGUMENT;
    if (!m->fourier_valid) return MTD_ERR_INVALID_ARGUMENT;
    MeshGeom g;
    int rc = fill_geom(g, m, box);
    if (rc) return rc;
    rc = log_scratch(m);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const unsigned int blocks = 256;
    k_mesh_virial<<<blocks, 256, 0, s>>>(g, m->d_f, (double)n_global, m->d_table_d, m->k_min, m->k_max, m->delta_k,
                                         m->use_table && m->d_table_d, m->d_log_scratch);
    MTD_LAUNCH_CHECK();
    double part[256 * 6];
    MTD_HIP_TRY(hipMemcpyAsync(part, m->d_log_scratch, sizeof(part), hipMemcpyDeviceToHost, s));
    MTD_HIP_TRY(hipStreamSynchronize(s));
    for (int c = 0; c < 6; ++c)
        {
        double v = 0.0;
        for (unsigned int b = 0; b < blocks; ++b) v += part[b * 6 + c];
        virial[c] = bias * v;                                                              // :1046-1047
        }
    return MTD_SUCCESS;
    }

int mtd_mesh_get_array(mtd_mesh *m, int which, void *host_out, mtd_stream_t stream)
    {
    if (!m || !host_out) return MTD_ERR_INVALID_ARGUMENT;
    const void *src = nullptr;
    size_t bytes = 0;
    switch (which)
        {
        case 0:                                                                  // real mesh (assignParticles)
            {
            const int rc0 = mesh_need_rho(m, (hipStream_t)stream);
            if (rc0) return rc0;
            src = m->d_rho; bytes = sizeof(double) * m->M;
            break;
            }
        case 1:                                                                  // fourier_mesh, normalised: full mesh from the stored half
            {
            if (!m->fourier_valid) return MTD_ERR_INVALID_ARGUMENT;
            const size_t MH = (size_t)m->hxp * m->ny * m->nz;
            std::vector<double2> half(MH);
            MTD_HIP_TRY(hipMemcpyAsync(half.data(), m->d_f, sizeof(double2) * MH, hipMemcpyDeviceToHost, (hipStream_t)stream));
            MTD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            double2 *out = (double2 *)host_out;
            for (unsigned int wz = 0; wz < m->nz; ++wz)
                for (unsigned int wy = 0; wy < m->ny; ++wy)
                    for (unsigned int wx = 0; wx < m->nx; ++wx)
                        {
                        double2 v;
                        if (wx <= m->nx / 2)
                            v = half[wx + (size_t)m->hxp * (wy + (size_t)m->ny * wz)];
                        else
                            {
                            const unsigned int mx = m->nx - wx, my = (m->ny - wy) % m->ny, mz = (m->nz - wz) % m->nz;
                            v = half[mx + (size_t)m->hxp * (my + (size_t)m->ny * mz)];
                            v.y = -v.y;
                            }
                        out[wx + (size_t)m->nx * (wy + (size_t)m->ny * wz)] = v;
                        }
            return MTD_SUCCESS;
            }
        case 3: src = m->d_inv; bytes = sizeof(double) * m->M; break;            // Re(inv_fourier_mesh), real double[M]
        case 7: src = m->d_mode_sq; bytes = sizeof(double); break;
        default: return MTD_ERR_INVALID_ARGUMENT;
        }
    MTD_HIP_TRY(hipMemcpyAsync(host_out, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    MTD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MTD_SUCCESS;
    }

unsigned int mtd_mesh_num_cells(const mtd_mesh *m) { return m ? m->M : 0; }

} // extern "C"
