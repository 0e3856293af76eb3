// metad_device.hpp — device structures and functions of the bias-grid engine, shared by the generic
// kernels (metad.hip) and the fused bias-step kernels (fused.hip).
//
// Reference arithmetic: IntegratorMetaDynamics.cc:663-736 (interpolateGrid), :738-776
// (biasPotentialDerivative), :1002-1047 (updateGrid), :1092-1155 (histogram bins), IndexGrid.cc:20-58.
#pragma once

#include "mtd_device.hpp"
#include "comm_device.hpp"
#include "exact_div.hpp"

#ifndef MTD_STAMP
#define MTD_STAMP(slot, cond) do { } while (0)
#endif

namespace mtd
{

constexpr int MAXCV = MTD_METAD_MAX_CV;
constexpr int GRID_THREADS = 256;
constexpr int MAX_POINTS = 2 * MAXCV + 2;
constexpr int MAX_TERMS = 1 << MAXCV;

struct CvSource
    {
    const double *partials;     // nullptr: the value is `shift` (host-provided)
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
    unsigned int failed;   // a mailbox wait expired in the last step (comm_device.hpp): nothing was deposited, the deferred pass is skipped
    unsigned int _pad;
    // The bias grid around the collective variables' last values (6 cells per variable, <= 3 variables), kept current by
    // whoever changes the grid (apply_cells).  The chain of the next step reads its stencil cells from here in the SAME memory
    // round trip as the CV sums — the grid load that depends on the CV values (~0.8 us on the path every block of launch B
    // waits for) becomes a shuffle whenever the variables moved by at most one cell since the origin was set.
    int patch_valid;       // the values below are the grid's (0: the host wrote the grid, or nothing has run the deferred pass yet)
    int patch_org[3];      // grid coordinates of patch slot 0 (may be negative / beyond the grid at its edges)
    int guess_org[3];      // where the next patch should sit: the cell of the values the last grid launch evaluated at, minus 2
    int _pad2;
    double zero;           // always 0.0: where chain_preload's lanes without a partial sum load from (no select behind the load)
    double patch_v[216];   // slot o0 + 6 (o1 + 6 o2)
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
    unsigned int n_gblocks;
    // chain_wave: quotients by the grid spacing, twice the grid spacing and the well-tempered temperature as correctly rounded
    // FMA sequences from host reciprocals (exact_div.hpp: bit-identical to the division, a fifth of its instructions on the one
    // wave whose latency every block of launch B waits for); 0 when any of the divisors does not qualify
    int fastdiv;
    double rdelta[3], rdelta2[3], rT_shift;
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

// exponent of updateGrid's Gaussian at grid cell `idx` for CV values s[] (:1019-1039):
// 1/2 sum_ij d_i d_j (sigma_inv_ij)^2 — element-wise square (Q12)
__device__ __forceinline__ double gauss_exponent(const MetadCfg &c, unsigned int idx, const double *s)
    {
    unsigned int coords[MAXCV];
    decode(c, idx, coords);
    double d[MAXCV];
    for (unsigned int i = 0; i < c.n_cv; ++i)
        {
        const double val_i = c.cv_min[i] + coords[i] * c.delta[i];
        d[i] = val_i - s[i];
        }
    double gauss_exp = 0.0;
    for (unsigned int i = 0; i < c.n_cv; ++i)
        for (unsigned int j = 0; j < c.n_cv; ++j)
            {
            const double sij = c.sigma_inv[i * c.n_cv + j];
            gauss_exp += d[i] * d[j] * (1.0 / 2.0) * (sij * sij);
            }
    return gauss_exp;
    }

// floor-bin shared by updateHistogram (:1092-1119) and updateSigmaGrid (:1122-1155).  The reference
// converts (s-min)/delta to unsigned: undefined for values <= -1 or >= 2^32, treated as off-grid.
__device__ __forceinline__ bool bin_of(const MetadCfg &c, const double *val, unsigned int &idx)
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

// Corner `bits` of interpolateGrid's multilinear sum (:685-733) for evaluation point val[]: grid cell
// and weight.  Returns false when the point is out of bounds (:677-683 => the interpolation is 0).
__device__ __forceinline__ bool interp_corner(const MetadCfg &c, const double *val, unsigned int bits, unsigned int &cell,
                                              double &weight)
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
    cell = idx;
    weight = t;
    return true;
    }

// CV values from their sources, cooperatively by the whole block, in a fixed order (bitwise the same in
// every block).  All loads of a CV are issued before the first add (a serial `v += p[b]` loop costs one
// memory round trip per iteration).  Result in s_cv[] (shared); s_tmp: >= 16 doubles of shared memory.
// The caller must __syncthreads() before reading s_cv.
__device__ __forceinline__ void reduce_cv_sources(const MetadCfg &c, double *s_cv, double *s_tmp)
    {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n_waves = blockDim.x >> 6;
    for (unsigned int i = 0; i < c.n_cv; ++i)
        {
        const CvSource s = c.src[i];
        double v = 0.0;
        if (s.partials)
            {
            for (unsigned int b0 = threadIdx.x; b0 < s.n_partials; b0 += 4 * blockDim.x)
                {
                double x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    {
                    const unsigned int b = b0 + j * blockDim.x;
                    x[j] = b < s.n_partials ? s.partials[(size_t)b * s.stride + s.offset] : 0.0;
                    }
                v += (x[0] + x[1]) + (x[2] + x[3]);
                }
            }
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) s_tmp[wave] = v;
        __syncthreads();
        if (threadIdx.x == 0)
            {
            double r = 0.0;
            for (int w = 0; w < n_waves; ++w) r += s_tmp[w];
            s_cv[i] = s.partials ? s.shift + s.scale * r : s.shift;
            }
        }
    }

// Shared-memory workspace of evaluate_bias()
struct EvalShared
    {
    double cv[MAXCV];
    double pts[MAX_POINTS][MAXCV];
    double val[MAX_POINTS][MAX_TERMS];   // grid value at the corner cell
    double wt[MAX_POINTS][MAX_TERMS];    // multilinear weight of the corner
    unsigned int cell[MAX_POINTS][MAX_TERMS];
    int oob[MAX_POINTS];
    double res[MAX_POINTS];
    double bias[MAXCV];
    double scal;
    double V_old;
    unsigned int bin;
    int on_grid;
    };

// The scalar part of updateBiasPotential for CV values sh.cv[], cooperatively by one block:
//   V_old(s) and the well-tempered scale (:374-379); then dV/ds_c (:444-445 -> :738-776), V(s) (:448)
//   and w(s) (:451) of the grid AFTER the deposit.  With `closed_form` the deposit has not been applied
//   to c.grid yet: the post-deposit node values are formed as grid[cell] + W*scal*exp(-gauss(cell))
//   for the <= (2 n_cv + 1) 2^n_cv cells the finite-difference stencil touches (the values the deferred
//   apply pass will store, up to FMA-contraction rounding of the same dV expression).
// Results: sh.scal, sh.bias[], sh.res[0] = V, sh.res[1] = w (w only meaningful when !closed_form).
__device__ __forceinline__ void evaluate_bias(const MetadCfg &c, EvalShared &sh, const bool deposit, const bool closed_form)
    {
    const unsigned int n = c.n_cv;
    const int n_points = 2 + 2 * (int)n;
    const int n_term = 1 << n;
    // point 0: s on the bias grid; point 1: s on the weight grid; 2+2i: s - delta_i e_i; 3+2i: s + delta_i e_i
    for (unsigned int idx = threadIdx.x; idx < (unsigned int)n_points * n; idx += blockDim.x)
        {
        const unsigned int p = idx / n, i = idx % n;
        double v = sh.cv[i];
        if (p >= 2 && (p - 2) / 2 == i) v = ((p - 2) & 1) ? v + c.delta[i] : v - c.delta[i];
        sh.pts[p][i] = v;
        }
    for (int p = threadIdx.x; p < n_points; p += blockDim.x) sh.oob[p] = 0;
    if (threadIdx.x == 0)
        {
        unsigned int bin = 0;
        sh.on_grid = bin_of(c, sh.cv, bin) ? 1 : 0;
        sh.bin = bin;
        }
    __syncthreads();
    // every (point, corner) pair is one lane's work: one grid read each, all in flight together
    for (int idx = threadIdx.x; idx < n_points * n_term; idx += blockDim.x)
        {
        const int p = idx / n_term;
        const unsigned int bits = idx % n_term;
        unsigned int cell = 0;
        double wt = 0.0;
        const bool ok = interp_corner(c, sh.pts[p], bits, cell, wt);
        sh.cell[p][bits] = cell;
        sh.wt[p][bits] = ok ? wt : 0.0;
        sh.val[p][bits] = ok ? (p == 1 ? c.weight[cell] : c.grid[cell]) : 0.0;
        if (!ok && bits == 0) sh.oob[p] = 1;
        }
    __syncthreads();
    if (threadIdx.x == 0)
        {
        double V = 0.0;
        for (int b = 0; b < n_term; ++b) V += sh.wt[0][b] * sh.val[0][b];      // corner order of :711-733
        sh.V_old = V;
        double scal = 1.0;
        if (deposit && c.mode == MTD_MODE_WELL_TEMPERED) scal = exp(-V / c.T_shift);  // :377-378
        sh.scal = scal;
        }
    __syncthreads();
    if (closed_form && deposit)
        {
        const double amp = c.W * sh.scal;
        for (int idx = threadIdx.x; idx < n_points * n_term; idx += blockDim.x)
            {
            const int p = idx / n_term;
            const unsigned int bits = idx % n_term;
            if (p != 1 && !sh.oob[p])
                {
                const double dV = amp * exp(-gauss_exponent(c, sh.cell[p][bits], sh.cv));
                sh.val[p][bits] += dV;
                }
            }
        __syncthreads();
        }
    for (int p = threadIdx.x; p < n_points; p += blockDim.x)
        {
        double res = 0.0;
        for (int b = 0; b < n_term; ++b) res += sh.wt[p][b] * sh.val[p][b];
        sh.res[p] = res;
        }
    __syncthreads();
    if (threadIdx.x < n)
        {
        const unsigned int i = threadIdx.x;
        const double s = sh.cv[i];
        const double delta = c.delta[i];
        double b;
        if (s - delta < c.cv_min[i])
            b = (sh.res[3 + 2 * i] - sh.res[0]) / delta;                        // forward  (:746-755)
        else if (s + delta > c.cv_max[i])
            b = (sh.res[0] - sh.res[2 + 2 * i]) / delta;                        // backward (:756-764)
        else
            b = (sh.res[3 + 2 * i] - sh.res[2 + 2 * i]) / (2.0 * delta);        // central  (:765-775)
        sh.bias[i] = b;
        }
    __syncthreads();
    }

// second pass of updateReweightedEstimator (:1077-1087) fused with accumulate + clear (:426-437) for grid
// cells [cell_begin, cell_end) (one cell per thread); every block re-derives <dV> from the pass-1 block
// partial sums in the same fixed order, so all blocks use the identical value.  s_red: >= 16 doubles of
// shared memory.  `first` marks the single block that also advances the engine's counters.
__device__ __forceinline__ void apply_cells(const MetadCfg &c, const unsigned int cell_begin, const unsigned int cell_end,
                                            const bool first, double *s_red)
    {
    if (c.st->failed) return;                                                // (block-uniform) the step that would have deposited was poisoned
    double s1 = 0.0, s2 = 0.0;
    for (unsigned int b = threadIdx.x; b < c.n_gblocks; b += blockDim.x)
        {
        s1 += c.gpart[2 * b];
        s2 += c.gpart[2 * b + 1];
        }
    s1 = block_sum(s1, s_red);
    s2 = block_sum(s2, s_red);
    const double avg_dV = s1 / s2;                                           // norm == 0 -> NaN like the reference (Q15)

    const unsigned int g = cell_begin + threadIdx.x;
    if (g < cell_end)
        {
        const double dV = c.grid_delta[g];
        const double fac = exp(-(dV - avg_dV) / c.temp);                     // T, not deltaT (:1084)
        c.rew[g] *= fac;
        c.weight[g] /= fac;
        const double g_new = c.grid[g] + dV;
        c.grid[g] = g_new;
        if (c.n_cv <= 3)
            {
            // the patch around the last CV values follows the grid (MetadState::patch_v)
            unsigned int rest = g;
            int slot = 0, mul = 1;
            bool in = true;
            for (int i = (int)c.n_cv - 1; i >= 0; --i)
                {
                const unsigned int co = rest / c.factors[i];
                rest -= co * c.factors[i];
                const int o = (int)co - c.st->guess_org[i];
                in = in && o >= 0 && o < 6;
                slot += o * (i == 0 ? 1 : (i == 1 ? 6 : 36));
                }
            (void)mul;
            if (in) c.st->patch_v[slot] = g_new;
            }
        // the histogram and width increments are zero everywhere but at the few cells the CV visited since the last pass: only
        // those cells touch the accumulated arrays and clear their increments (x += 0 changes nothing: the same arrays bit for
        // bit, 48 of the 128 bytes per cell this pass moves are not moved)
        const double sgd = c.sigma_grid_delta[g];
        if (sgd != 0.0)                                                      // (NaN too)
            {
            c.sigma_grid[g] += sgd;
            c.sigma_grid_delta[g] = 0.0;
            }
        const unsigned int hd = c.hist_delta[g], hgd = c.hist_gauss_delta[g];
        if (hd)
            {
            c.hist[g] += hd;
            c.hist_delta[g] = 0;
            }
        if (hgd)
            {
            c.hist_gauss[g] += hgd;
            c.hist_gauss_delta[g] = 0;
            }
        c.grid_delta[g] = 0.0;
        }
    if (first && threadIdx.x == 0)
        {
        // (every cell of the region has just been written by the block that owns it: the patch is the grid at guess_org)
        c.st->patch_org[0] = c.st->guess_org[0];
        c.st->patch_org[1] = c.st->guess_org[1];
        c.st->patch_org[2] = c.st->guess_org[2];
        c.st->patch_valid = c.n_cv <= 3 ? 1 : 0;
        c.st->avg_dV = avg_dV;
        c.st->num_gaussians += 1;                                            // :440
        }
    }

// ------------------------------------------------------------------------------------------------
// The same scalar chain as reduce_cv_sources + evaluate_bias, executed by ONE wave with shuffles only
// (no __syncthreads, no LDS), so the other waves of the block can stream particles meanwhile.
// Valid for n_cv <= 3 (then (2 n_cv + 2) 2^n_cv <= 64 (point, corner) pairs fit one wave).
// Must be called by a full wave (all 64 lanes).  Results are returned in every lane.
struct ChainResult
    {
    double cv[3];
    double bias[3];
    double scal, V, w;
    unsigned int bin;
    int on_grid, oob;
    int failed;            // rx / given only: a bounded wait expired — NaN bias factors, nothing may be deposited
    // per LANE (lanes < 2^n_cv, the corners of the point s itself), closed form only: multilinear weight, the weight-grid
    // value before this step's deposit and this deposit's increment at the corner cell — w(s) of the grid AFTER the deferred
    // reweighting pass follows from them in closed form once <dV> is known (k_fused_step)
    double c_wt, c_wold, c_dV;
    };

constexpr int CHAIN_MAX_CV = 3;

// A patch of 6^n_cv bias-grid cells around a GUESS of the collective variables (the previous step's values), loaded by the
// chain's wave before the real values exist (k_fused_step: while the particles are still being summed).  The stencil of the
// finite-difference derivative spans cells l - 1 .. l + 2 per variable (l = the cell of s): the patch l_g - 2 .. l_g + 3
// covers it whenever s moved by at most one cell since the guess — then the chain's one dependent memory round trip (grid
// values at cells that depend on s) is a shuffle; otherwise the cells are loaded as before.  Slot o0 + 6 (o1 + 6 o2) sits in
// lane slot % 64, register slot / 64.
struct GridPatch
    {
    double v[4];
    int org[CHAIN_MAX_CV];
    };

__device__ __forceinline__ GridPatch chain_prefetch(const MetadCfg &c)
    {
    const int lane = threadIdx.x & 63;
    const int n = (int)c.n_cv;
    GridPatch gp;
    int total = 1;
#pragma unroll
    for (int i = 0; i < CHAIN_MAX_CV; ++i)
        {
        gp.org[i] = 0;
        if (i < n)
            {
            const double s = c.st->cv[i];                                 // the previous step's value (any value is a valid guess)
            double q = (s - c.cv_min[i]) / c.delta[i];
            if (!(q > 0.0)) q = 0.0;
            if (q > (double)c.lengths[i]) q = (double)c.lengths[i];
            gp.org[i] = (int)q - 2;
            total *= 6;
            }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        {
        gp.v[r] = 0.0;
        const int slot = lane + 64 * r;
        if (slot < total)
            {
            int rest = slot;
            bool in = true;
            unsigned int cell = 0;
#pragma unroll
            for (int i = 0; i < CHAIN_MAX_CV; ++i)
                if (i < n)
                    {
                    const int coord = gp.org[i] + rest % 6;
                    rest /= 6;
                    in = in && coord >= 0 && coord < (int)c.lengths[i];
                    cell += (unsigned int)coord * c.factors[i];
                    }
            if (in) gp.v[r] = c.grid[cell];
            }
        }
    return gp;
    }

// rx != nullptr: particle-sharded step — the sums over ranks come out of the xGMI mailbox (comm_device.hpp) instead
// of the registered partial sums.  given != nullptr: the (global) sums are handed in (k_fused_step collected them itself);
// a NaN among them marks an expired wait.  want_weight: also read the weight grid at the corners of s (closed form).
// What the chain's wave can ask for before anything else of the launch has happened (k_fused_force: before the barrier that
// publishes the mode tables): the first 512 partial sums of every CV source and the grid patch around the last CV values — one
// memory round trip for everything the chain reads.
// NCH: collective variables of the bias grid the caller is compiled for (the arrays stay in registers: static indices only).
template<int NCH> struct ChainPre
    {
    double x[NCH][4];              // partial sums lane, lane + 64, lane + 128, lane + 192 of every CV source
    GridPatch patch;
    int patch_ok;
    };

// SUMS = false: the grid patch only (sharded step: the sums come out of the mailbox)
template<int NCH, bool SUMS = true>
__device__ __forceinline__ void chain_preload(const MetadCfg &c, ChainPre<NCH> &p)
    {
    const unsigned int lane = threadIdx.x & 63;
    // every scalar the loads need first, in one batch (fetched field by field inside predicated loads they were ~25 dependent
    // scalar-load round trips: 2 us before the first vector load left), then branch-free loads from clamped addresses
    const double *ptr[NCH];
    unsigned int cnt[NCH], str[NCH], off[NCH];
    const double *safe = &c.st->zero;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        {
        const bool on = i < (int)c.n_cv && c.src[i].partials != nullptr;
        ptr[i] = on ? c.src[i].partials : safe;
        cnt[i] = on ? c.src[i].n_partials : 0u;
        str[i] = on ? c.src[i].stride : 0u;
        off[i] = on ? c.src[i].offset : 0u;
        }
    const int ok = c.st->patch_valid;
    int org[CHAIN_MAX_CV];
#pragma unroll
    for (int i = 0; i < CHAIN_MAX_CV; ++i) org[i] = c.st->patch_org[i];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            {
            // (a lane without a partial sum reads the 0.0 of MetadState::zero: nothing behind the load has to wait for it)
            const unsigned int b = lane + j * MTD_WAVE;
            const double *q = (SUMS && b < cnt[i]) ? ptr[i] + ((size_t)b * str[i] + off[i]) : safe;
            p.x[i][j] = SUMS ? *q : 0.0;
            }
    p.patch_ok = ok;
#pragma unroll
    for (int i = 0; i < CHAIN_MAX_CV; ++i) p.patch.org[i] = org[i];
    p.patch.v[0] = c.st->patch_v[lane];                              // 36 slots (two variables) sit in the first 64
#pragma unroll
    for (int r = 1; r < 4; ++r) p.patch.v[r] = NCH == 3 ? c.st->patch_v[min(lane + 64 * r, 215u)] : 0.0;
    }

// a / (the divisor whose reciprocal is y), correctly rounded (exact_div.hpp) when the grid qualifies, else the division
__device__ __forceinline__ double chain_div(const MetadCfg &c, const double a, const double b, const double y)
    {
    if (!c.fastdiv) return a / b;                                    // (uniform)
    const double q0 = a * y;
    const double r0 = __builtin_fma(-b, q0, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-b, q1, a);
    return __builtin_fma(r1, y, q1);
    }

__device__ __forceinline__ ChainResult chain_wave(const MetadCfg &c, const bool deposit, const bool closed_form,
                                                  const CommK *rx = nullptr, const double *given = nullptr,
                                                  const bool want_weight = false, const GridPatch *patch = nullptr,
                                                  const bool patch_ok = true, const bool have_init = false,
                                                  const double v_init0 = 0.0, const double v_init1 = 0.0, const double v_init2 = 0.0)
    {
    const int lane = threadIdx.x & 63;
    const unsigned int n = c.n_cv;
    const int n_term = 1 << n;
    ChainResult r;

    // 1. CV values: the loads of ALL CVs are issued before the first add (one memory round trip, not one
    //    per CV), eight in flight per lane and CV; xor-butterfly sums give every lane the same bits
    double x[CHAIN_MAX_CV][8];
    unsigned int n_part_max = 0;
#pragma unroll
    for (int i = 0; i < CHAIN_MAX_CV; ++i)
        if (i < (int)n && c.src[i].partials) n_part_max = max(n_part_max, c.src[i].n_partials);
    double v[CHAIN_MAX_CV] = { 0.0, 0.0, 0.0 };
    double rx_total[3] = { 0.0, 0.0, 0.0 };
    if (rx)
        {
        comm_recv_sum_wave(*rx, rx_total, n);
        n_part_max = 0;
        }
    else if (given)
        {
#pragma unroll
        for (int i = 0; i < CHAIN_MAX_CV; ++i) rx_total[i] = i < (int)n ? given[i] : 0.0;
        n_part_max = 0;
        }
    const bool handed = rx != nullptr || given != nullptr;
    // v_init0..2: this lane's sum of partial sums lane, lane + 64, lane + 128, lane + 192 of every source, loaded ahead of time
    // (chain_preload); the loop goes on behind them
    if (have_init && n_part_max)
        {
        v[0] = v_init0;
        v[1] = v_init1;
        v[2] = v_init2;
        }
    for (unsigned int b0 = lane + ((have_init && n_part_max) ? 4 * MTD_WAVE : 0); b0 < n_part_max; b0 += 8 * MTD_WAVE)
        {
#pragma unroll
        for (int i = 0; i < CHAIN_MAX_CV; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                {
                const unsigned int b = b0 + j * MTD_WAVE;
                x[i][j] = (i < (int)n && c.src[i].partials && b < c.src[i].n_partials)
                              ? c.src[i].partials[(size_t)b * c.src[i].stride + c.src[i].offset] : 0.0;
                }
#pragma unroll
        for (int i = 0; i < CHAIN_MAX_CV; ++i)
            v[i] += ((x[i][0] + x[i][1]) + (x[i][2] + x[i][3])) + ((x[i][4] + x[i][5]) + (x[i][6] + x[i][7]));
        }
#pragma unroll
    for (int i = 0; i < CHAIN_MAX_CV; ++i)
        {
        r.cv[i] = 0.0;
        r.bias[i] = 0.0;
        if (i < (int)n)
            {
            const double t = handed ? rx_total[i] : wave_sum(v[i]);
            r.cv[i] = (c.src[i].partials || handed) ? c.src[i].shift + c.src[i].scale * t : c.src[i].shift;
            }
        }

    MTD_STAMP(40, blockIdx.x == 0 && threadIdx.x == 0 && closed_form);
    // 2. per-dimension stencil geometry: lane 3 i + v handles CV i at s_i - delta (v=0), s_i (1), s_i + delta (2)
    //    (interpolateGrid :675-699 for the values biasPotentialDerivative :746-775 passes in)
    int g_lower = 0, g_ok = 0;
    double g_rel = 0.0;
    if (lane < 3 * (int)n)
        {
        // per-lane CV index i = lane / 3: select the (scalar, statically indexed) grid geometry instead of
        // indexing the kernel-argument arrays with a vector index (that becomes a global load)
        const int i = lane / 3, v = lane % 3;
        const double si = i == 0 ? r.cv[0] : (i == 1 ? r.cv[1] : r.cv[2]);
        const double delta = i == 0 ? c.delta[0] : (i == 1 ? c.delta[1] : c.delta[2]);
        const double cmin = i == 0 ? c.cv_min[0] : (i == 1 ? c.cv_min[1] : c.cv_min[2]);
        const double cmax = i == 0 ? c.cv_max[0] : (i == 1 ? c.cv_max[1] : c.cv_max[2]);
        const double rdel = i == 0 ? c.rdelta[0] : (i == 1 ? c.rdelta[1] : c.rdelta[2]);
        const int len = (int)(i == 0 ? c.lengths[0] : (i == 1 ? c.lengths[1] : c.lengths[2]));
        double val = si;
        if (v == 0) val = si - delta;
        if (v == 2) val = si + delta;
        g_ok = val >= cmin && val < cmax;             // (:677-683; a NaN value is off the grid too)
        int lower = (int)chain_div(c, val - cmin, delta, rdel);
        int upper = lower + 1;
        if (upper >= len)
            {
            lower--;
            upper--;
            }
        const double lower_bound = cmin + delta * lower;
        const double upper_bound = cmin + delta * upper;
        g_rel = (val - lower_bound) / (upper_bound - lower_bound);
        g_lower = lower;
        }

    MTD_STAMP(41, blockIdx.x == 0 && threadIdx.x == 0 && closed_form);
    // 3. (point, corner) pairs, one per lane.  points: 0 = s on the bias grid, 1+2i = s - delta_i e_i,
    //    2+2i = s + delta_i e_i, and (only when the weight grid is final) 1+2n = s on the weight grid
    const int n_pts = closed_form ? 1 + 2 * (int)n : 2 + 2 * (int)n;
    const int p = lane / n_term;
    const unsigned int bits = lane % n_term;
    const bool is_weight = (!closed_form) && p == 1 + 2 * (int)n;
    bool ok = p < n_pts;
    double wt = 1.0;
    unsigned int cell = 0;
    double d[CHAIN_MAX_CV];
    int pslot = 0, pmul = 1;                 // slot of this lane's cell in the prefetched patch; in_patch: it lies inside
    bool in_patch = patch != nullptr && patch_ok;
#pragma unroll
    for (int i = 0; i < CHAIN_MAX_CV; ++i)
        {
        d[i] = 0.0;
        if (i < (int)n)
            {
            int var = 1;
            if (p == 1 + 2 * i) var = 0;
            if (p == 2 + 2 * i) var = 2;
            const int src_lane = 3 * i + var;
            const int lower = __shfl(g_lower, src_lane, MTD_WAVE);
            const double rel = __shfl(g_rel, src_lane, MTD_WAVE);
            const int oki = __shfl(g_ok, src_lane, MTD_WAVE);
            ok = ok && oki;
            unsigned int coord;
            if (bits & (1u << i))
                {
                coord = (unsigned int)lower;
                wt *= (1.0 - rel);
                }
            else
                {
                coord = (unsigned int)(lower + 1);
                wt *= rel;
                }
            cell += coord * c.factors[i];
            d[i] = (c.cv_min[i] + coord * c.delta[i]) - r.cv[i];     // updateGrid :1023-1026
            if (patch)
                {
                const int o = (int)coord - patch->org[i];
                in_patch = in_patch && o >= 0 && o < 6;
                pslot += o * pmul;
                pmul *= 6;
                }
            }
        }
    double val = 0.0;
    if (patch)
        {
        // every lane takes part in the shuffles; lanes whose cell lies outside the patch (or that read the weight grid) load
        const int src = (in_patch && ok) ? pslot : 0;
        double pv = __shfl(patch->v[0], src & 63, MTD_WAVE);
        if (n == 3)
            {
            const double p1 = __shfl(patch->v[1], src & 63, MTD_WAVE), p2 = __shfl(patch->v[2], src & 63, MTD_WAVE),
                         p3 = __shfl(patch->v[3], src & 63, MTD_WAVE);
            const int reg = src >> 6;
            pv = reg == 0 ? pv : (reg == 1 ? p1 : (reg == 2 ? p2 : p3));
            }
        if (ok) val = (in_patch && !is_weight) ? pv : (is_weight ? c.weight[cell] : c.grid[cell]);
        }
    else if (ok) val = is_weight ? c.weight[cell] : c.grid[cell];
    r.c_wt = wt;
    r.c_wold = (want_weight && ok && p == 0) ? c.weight[cell] : 0.0;
    r.c_dV = 0.0;

    // ---- everything that does not need the grid values runs while they are on their way:
    // the Gaussian of this deposit at the lane's stencil cell (closed form, step 5) ...
    double e_g = 0.0;
    if (closed_form && deposit && ok)
        {
        double gauss_exp = 0.0;
#pragma unroll
        for (int i = 0; i < CHAIN_MAX_CV; ++i)
#pragma unroll
            for (int j = 0; j < CHAIN_MAX_CV; ++j)
                if (i < (int)n && j < (int)n)
                    {
                    const double sij = c.sigma_inv[i * n + j];
                    gauss_exp += d[i] * d[j] * (1.0 / 2.0) * (sij * sij);
                    }
        e_g = exp(-gauss_exp);
        }
    // ... and the histogram bin (updateHistogram :1092-1119), statically unrolled (no private-memory arrays)
    bool on_grid = true;
    unsigned int bin = 0;
#pragma unroll
    for (int i = 0; i < CHAIN_MAX_CV; ++i)
        {
        if (i < (int)n)
            {
            const double q = chain_div(c, r.cv[i] - c.cv_min[i], c.delta[i], c.rdelta[i]);
            if (!(q > -1.0) || !(q < 4294967296.0))
                on_grid = false;
            else
                {
                const unsigned int coord = (unsigned int)q;
                if (coord >= c.lengths[i]) on_grid = false;
                bin += coord * c.factors[i];
                }
            }
        }
    r.on_grid = on_grid ? 1 : 0;
    r.bin = bin;

    MTD_STAMP(42, blockIdx.x == 0 && threadIdx.x == 0 && closed_form);
    // 4. V_old(s): corner terms of point 0 summed in the reference's order (:711-733); the corners sit in lanes 0 .. 2^n - 1:
    //    v_readlane, not a trip through the LDS crossbar
    double term = ok ? wt * val : 0.0;
    double V_old = 0.0;
    for (int b = 0; b < n_term; ++b) V_old += wave_read(term, b);
    r.scal = 1.0;
    if (deposit && c.mode == MTD_MODE_WELL_TEMPERED) r.scal = exp(chain_div(c, -V_old, c.T_shift, c.rT_shift));   // :377-378

    MTD_STAMP(43, blockIdx.x == 0 && threadIdx.x == 0 && closed_form);
    // 5. post-deposit node values in closed form on the stencil
    if (closed_form && deposit && ok)
        {
        r.c_dV = (c.W * r.scal) * e_g;
        val += r.c_dV;
        term = wt * val;
        }
    if (!ok) r.c_wt = 0.0;

    MTD_STAMP(44, blockIdx.x == 0 && threadIdx.x == 0 && closed_form);
    // 6. every lane: the interpolated value of its own point, corners in order.  The 2^n corner lanes of a point are a pair
    //    (n = 1) or a quad (n = 2): DPP broadcasts inside the quad; n = 3 goes through the crossbar, all eight in flight
    double res = 0.0;
    if (n == 2)
        {
        res += dpp_move<0x00>(term);          // quad_perm:[0,0,0,0]
        res += dpp_move<0x55>(term);          // [1,1,1,1]
        res += dpp_move<0xAA>(term);          // [2,2,2,2]
        res += dpp_move<0xFF>(term);          // [3,3,3,3]
        }
    else if (n == 1)
        {
        res += dpp_move<0xA0>(term);          // [0,0,2,2]
        res += dpp_move<0xF5>(term);          // [1,1,3,3]
        }
    else
        {
        const int base = (p < n_pts ? p : 0) * n_term;
        double t8[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) t8[b] = __shfl(term, (base + b) & 63, MTD_WAVE);
#pragma unroll
        for (int b = 0; b < 8; ++b) res += t8[b];
        }

    // 7. finite differences (:738-776); the points sit in wave-uniform lanes: v_readlane
    const double res0 = wave_read(res, 0);
    r.V = res0;
    r.w = closed_form ? 1.0 : wave_read(res, (1 + 2 * (int)n) * n_term);
    r.oob = !wave_read((int)ok, 0);
#pragma unroll
    for (int i = 0; i < CHAIN_MAX_CV; ++i)
        {
        if (i < (int)n)
            {
            const double rm = wave_read(res, (1 + 2 * i) * n_term);
            const double rp = wave_read(res, (2 + 2 * i) * n_term);
            const double s = r.cv[i];
            const double delta = c.delta[i];
            double b;
            if (s - delta < c.cv_min[i])
                b = chain_div(c, rp - res0, delta, c.rdelta[i]);                  // forward  (:746-755)
            else if (s + delta > c.cv_max[i])
                b = chain_div(c, res0 - rm, delta, c.rdelta[i]);                  // backward (:756-764)
            else
                b = chain_div(c, rp - rm, 2.0 * delta, c.rdelta2[i]);             // central  (:765-775)
            r.bias[i] = b;
            }
        }
    MTD_STAMP(45, blockIdx.x == 0 && threadIdx.x == 0 && closed_form);
    r.failed = 0;
    if (handed)
        {
        bool bad = false;
#pragma unroll
        for (int i = 0; i < CHAIN_MAX_CV; ++i) bad = bad || (i < (int)n && is_comm_poison(rx_total[i]));   // (an arithmetic NaN is not a failure)
        if (bad)
            {
            // poisoned step: NaN CV values, bias factors (=> NaN forces), V and w; no hill, no histogram count
            const double nan = __longlong_as_double(0x7ff8000000000000ll);
#pragma unroll
            for (int i = 0; i < CHAIN_MAX_CV; ++i)
                if (i < (int)n) r.cv[i] = r.bias[i] = nan;
            r.V = r.w = nan;
            r.scal = 0.0;
            r.on_grid = 0;
            r.oob = 0;
            r.failed = 1;
            }
        }
    return r;
    }

// gauss_exponent for n_cv <= 3 without private-memory arrays or integer division loops over a runtime
// dimension count (same arithmetic, statically unrolled)
__device__ __forceinline__ double gauss_exponent3(const MetadCfg &c, unsigned int idx, const double s0, const double s1,
                                                  const double s2)
    {
    const unsigned int n = c.n_cv;
    unsigned int rest = idx;
    unsigned int c2 = 0, c1 = 0, c0;
    if (n > 2)
        {
        c2 = rest / c.factors[2];
        rest -= c2 * c.factors[2];
        }
    if (n > 1)
        {
        c1 = rest / c.factors[1];
        rest -= c1 * c.factors[1];
        }
    c0 = rest;
    double d[3];
    d[0] = (c.cv_min[0] + c0 * c.delta[0]) - s0;
    d[1] = n > 1 ? (c.cv_min[1] + c1 * c.delta[1]) - s1 : 0.0;
    d[2] = n > 2 ? (c.cv_min[2] + c2 * c.delta[2]) - s2 : 0.0;
    double gauss_exp = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (i < (int)n && j < (int)n)
                {
                const double sij = c.sigma_inv[i * n + j];
                gauss_exp += d[i] * d[j] * (1.0 / 2.0) * (sij * sij);
                }
    return gauss_exp;
    }

// ---- pieces of the bias-grid engine's launch for kernels of OTHER files that carry it (mesh.hip: k_tile_forces_chain) -----------
// The same statements as k_fused_force's grid blocks and its publishing wave (fused.hip, which keeps its own copy with the diagnostic
// time stamps in it): the same sums in the same order, so the grid arrays come out the same whichever launch carried the pass
// (k_tile_forces_chain in mesh.hip, k_ql_finalize_chain in steinhardt.hip).

// what the chain's wave returns, as the block shares it (lane 0 of the chain's wave writes, a barrier publishes)
__device__ __forceinline__ void chain_share(ChainResult &s_chain, const ChainResult &r)
    {
    s_chain.cv[0] = r.cv[0]; s_chain.cv[1] = r.cv[1]; s_chain.cv[2] = r.cv[2];
    s_chain.bias[0] = r.bias[0]; s_chain.bias[1] = r.bias[1]; s_chain.bias[2] = r.bias[2];
    s_chain.scal = r.scal; s_chain.V = r.V; s_chain.w = r.w;
    s_chain.bin = r.bin; s_chain.on_grid = r.on_grid; s_chain.oob = r.oob; s_chain.failed = r.failed;
    }

// First grid pass of a deposit step (updateGrid :1002-1047, updateHistogram :1092-1119, updateSigmaGrid :1122-1155, first loop of
// updateReweightedEstimator :1070-1075) for the 256 cells of grid block `vb`, by threads 0 .. 255 of the calling block.  EVERY
// thread of the block calls it (one __syncthreads inside); s_red: 8 doubles.
__device__ __forceinline__ void grid_first_pass_256(const MetadCfg &c, const ChainResult &s_chain, const unsigned int vb, double *s_red)
    {
    const unsigned int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mine = threadIdx.x < 256;
    const unsigned int g = vb * 256 + threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    if (mine && g < c.len && !s_chain.failed)
        {
        const double dV = (c.W * s_chain.scal) * exp(-gauss_exponent3(c, g, s_chain.cv[0], s_chain.cv[1], s_chain.cv[2]));
        c.grid_delta[g] = dV;
        unsigned int hd = c.hist_delta[g];
        if (s_chain.on_grid && g == s_chain.bin)
            {
            hd += 1;
            c.hist_delta[g] = hd;
            c.sigma_grid_delta[g] += c.det_sigma;
            c.hist_gauss_delta[g] += 1;
            }
        const double Rw = c.rew[g] + (double)hd;
        c.rew[g] = Rw;
        s1 = Rw * dV;
        s2 = Rw;
        }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (mine && lane == 0)
        {
        s_red[2 * wave] = s1;
        s_red[2 * wave + 1] = s2;
        }
    __syncthreads();
    if (threadIdx.x == 0)
        {
        double t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < 4; ++w)
            {
            t1 += s_red[2 * w];
            t2 += s_red[2 * w + 1];
            }
        c.gpart[2 * vb] = t1;
        c.gpart[2 * vb + 1] = t2;
        }
    }

// One wave of one block publishes the step's scalars for the host (lazy read-back) and, on non-deposit steps, owns the histogram
// increment (:366) and the weight read-out (the weight grid is final then).  Called by a FULL wave (chain_wave inside).
// given: the CV values handed to the chain in registers / LDS (k_ql_finalize_chain) instead of registered partial sums
__device__ __forceinline__ void publish_step(const MetadCfg &c, const ChainResult &s_chain, const int deposit, const double *given = nullptr)
    {
    const int lane = threadIdx.x & 63;
    double w_now = 1.0;
    if (!deposit) w_now = chain_wave(c, false, false, nullptr, given).w;    // w(s) from the (final) weight grid
    if (lane < (int)c.n_cv)
        {
        const double s_l = lane == 0 ? s_chain.cv[0] : (lane == 1 ? s_chain.cv[1] : s_chain.cv[2]);
        c.st->cv[lane] = s_l;
        c.st->bias[lane] = lane == 0 ? s_chain.bias[0] : (lane == 1 ? s_chain.bias[1] : s_chain.bias[2]);
        // where the grid patch of the next step should sit (apply_cells fills it at this origin): the cell of s, minus 2
        const double dl = lane == 0 ? c.delta[0] : (lane == 1 ? c.delta[1] : c.delta[2]);
        const double ml = lane == 0 ? c.cv_min[0] : (lane == 1 ? c.cv_min[1] : c.cv_min[2]);
        const double ll = (double)(lane == 0 ? c.lengths[0] : (lane == 1 ? c.lengths[1] : c.lengths[2]));
        double q = (s_l - ml) / dl;
        if (!(q > 0.0)) q = 0.0;                                   // (NaN too)
        if (q > ll) q = ll;
        c.st->guess_org[lane] = (int)q - 2;
        }
    if (lane == 0)
        {
        c.st->V = s_chain.V;
        c.st->failed = (unsigned int)s_chain.failed;
        c.st->bin = s_chain.bin;
        c.st->on_grid = (unsigned int)s_chain.on_grid;
        if (deposit)
            c.st->scal = s_chain.scal;
        else
            {
            c.st->w = s_chain.failed ? s_chain.V : w_now;            // (V is NaN then)
            if (s_chain.on_grid) c.hist_delta[s_chain.bin] += 1;
            }
        if (s_chain.oob) c.st->n_oob += (deposit && c.mode == MTD_MODE_WELL_TEMPERED) ? 2 : 1;
        }
    }

} // namespace mtd
