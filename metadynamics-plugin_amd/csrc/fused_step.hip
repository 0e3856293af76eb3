// fused_step.hip — the whole metadynamics bias step for lamellar CVs in ONE launch (headline path, mtd_fused_step).
//
// Reference step being replaced: IntegratorMetaDynamics::update -> updateBiasPotential (IntegratorMetaDynamics.cc:219-312,
// 314-588) around LamellarOrderParameterGPU's two kernels per CV and mode (LamellarOrderParameterGPU.cu:8-236).
//
// The two-launch form (fused.hip) reads every position twice and pays two dispatches and two first memory round trips;
// measured 8.3 + 13.5 us at 10^6 particles where a bare read-16 / write-32 stream takes 6.6 us.  Here one PERSISTENT launch —
// every block resident at once, one 1024-thread block per compute unit — keeps its particles IN REGISTERS between the CV
// phase and the force phase (4 per thread: the 16 MB of positions are read once) and hands the two grid-wide sums of a step
// from block to block inside the launch, in the xGMI mailbox's wire format (comm_device.hpp: a value and its exchange number
// travel in one 8-byte write-through store; consumers poll with loads that go past the non-coherent L2s; no atomics, no
// fences, and every wait is bounded):
//
//   phase 0  all waves    request 4 particles per thread; stage the mode tables in LDS
//   phase 1  all waves    per-CV sums of cos(q.r) (packed pairs, folded second harmonics: lamellar_device.hpp), block sum,
//                         posted for all blocks                                              [hand-off 1: CV sums]
//   phase 2  wave 0       collects the sums of ALL blocks (same fixed order in every block => the same bits), then the scalar
//                         chain: V_old(s), well-tempered scale, post-deposit stencil in closed form -> dV/ds_c (metad_device.hpp)
//            waves 1-15   meanwhile the unscaled forces sum_k q_k sin(q_k.r) of their particles from the registers
//                         (wave 0 forms its own after the chain)
//   phase 3  all waves    forces scaled by the bias factors, non-temporal stores (32 MB)
//            deposit step: every block runs the first grid pass on ITS OWN slice of the bias grid (Gaussian increment,
//                         histogram / sigma-grid bin, R += hist_delta) and posts its sums of R dV and R  [hand-off 2]
//   phase 4  deposit step: <dV> from all blocks' sums, second reweighting pass + accumulate on the same slice — the grid is
//                         FINAL when the launch ends (no deferred pass as in the two-launch form); w(s) in closed form from
//                         the weight-grid corners read in phase 2 (before any block's phase 4 can touch them: a block enters
//                         phase 4 only after every block has posted hand-off 2, i.e. has finished its chain)
//
// Sharded step (mailbox attached): block 0 collects the local sums and sends them to every rank; every block's chain polls
// the local mailbox for the ranks' totals exactly as launch B of the two-launch form does.
//
// A launch whose blocks wait for each other is only issued when the whole grid is resident at once (occupancy x compute
// units, checked per launch); otherwise, and for anything outside its envelope (more than 4096 particles per compute unit,
// more than 3 collective variables, a slot map, MTD_FUSED_STEP=0), mtd_fused_step runs the two-launch form.
#include <hip/hip_runtime.h>

// Diagnostic build only (-DMTD_STAMPS, tools/build_stamps.sh): s_memrealtime (100 MHz) stamps of block 0 and of the last
// block, written to a buffer of their own (tools/stamps_step.py); the product build has no stamps.
#ifdef MTD_STAMPS
__device__ unsigned long long g_step_stamps[64];
__device__ unsigned long long g_step_block[6][256];      // per block: entry, posted 1, collected, chain done, posted 2, end
#define MTD_BSTAMP(row) do { g_step_block[row][blockIdx.x] = wall_clock64(); } while (0)
extern "C" int mtd_debug_read_step_blocks(unsigned long long *host)
    {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_step_block), sizeof(unsigned long long) * 6 * 256);
    }
#define MTD_STAMP(slot, cond)                                                   \
    do                                                                          \
        {                                                                       \
        if (cond) g_step_stamps[slot] = wall_clock64();                         \
        } while (0)
extern "C" int mtd_debug_read_step_stamps(unsigned long long *host)
    {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_step_stamps), sizeof(unsigned long long) * 64);
    }
#else
#define MTD_STAMP(slot, cond) do { } while (0)
#define MTD_BSTAMP(row) do { } while (0)
#endif

#include "lamellar_host.hpp"
#include "metad_host.hpp"
#include "comm_host.hpp"

#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <mutex>

namespace
{

using namespace mtd;

constexpr int FS_THREADS = 1024;
constexpr int FS_WAVES = FS_THREADS / MTD_WAVE;
constexpr int FS_U = 4;                                   // particles per thread, held in registers across the phases
constexpr unsigned int FS_MAX_BLOCKS = 256;               // one block per compute unit
// Wave 0 of a block owns NO particles: it prefetches the bias-grid patch, collects the blocks' sums and runs the chain, so that
// nothing of its own stands between the last block's sums and the bias factors.  Waves 1 .. 15 hold FS_U particles per lane
// (slots u * FS_MAIN + (wave - 1) * 64 + lane), waves 1 .. 4 one more (slots FS_U * FS_MAIN + (wave - 1) * 64 + lane): the
// same 4096 per block.
constexpr unsigned int FS_MAIN = (FS_WAVES - 1) * MTD_WAVE;            // 960 particles per register slot
constexpr unsigned int FS_EXTRA_WAVES = 4;
constexpr unsigned int FS_CHUNK = FS_U * FS_MAIN + FS_EXTRA_WAVES * MTD_WAVE;   // particles per block at most: 4096
constexpr unsigned int FS_LL_WORDS = 2 * (CHAIN_MAX_CV + 2);   // per block: <= 3 CV sums + 2 grid sums, two 8-byte words each; stored by
                                                               // columns, ll[word][block] (comm_device.hpp: ll_collect_columns)
// Every block reads every block's sums: 245 waves polling the same few KB would queue on the one or two memory channels
// behind them (measured: the hand-off took 3.3 us instead of one round trip).  So a row of 256 words sits in a 4 KB channel
// stripe of its own and the whole table exists FS_LL_REPLICAS times — a block posts into every replica (a few dozen 8-byte
// stores) and reads replica (block mod FS_LL_REPLICAS): the reads spread over 8 x as many channels.
constexpr int FS_FIRST_PROBE_SLEEP = 12;                       // x 64 clocks ~ 0.35 us
constexpr unsigned int FS_LL_PITCH = 512;                      // words per row: 4 KB
constexpr unsigned int FS_LL_REPLICAS = 8;
constexpr unsigned int FS_LL_REPLICA_WORDS = FS_LL_WORDS * FS_LL_PITCH;

template<int NCV, int U> struct StepRegs
    {
    v2f g0[U / 2], g1[U / 2], g2[U / 2];                  // fractional coordinates (turns), particle pairs
    int type[U];
    bool ok[U];
    float f[U][NCV][3];                                    // unscaled forces
    };

// sum_k cos(2 pi t_k) per CV over this thread's particles, weighted by the type coefficients: lam_cv_accumulate's inner
// part for one register group (same arithmetic, same order)
template<int NCV, bool FAST, int U>
__device__ __forceinline__ void step_cv_sums(const LamKArgs &a, const float *s_coeff, const ModeTables &mt, const StepRegs<NCV, U> &R,
                                             float (&acc)[NCV])
    {
    constexpr int P = U / 2;
#pragma unroll
    for (int c = 0; c < NCV; ++c)
        {
        acc[c] = 0.0f;
        if (c < (int)a.n_cv)
            {
            v2f sum[P];
#pragma unroll
            for (int q = 0; q < P; ++q) sum[q] = (v2f)(0.0f);
            const unsigned int k1 = a.first[c] + a.nact[c];
#pragma unroll 4
            for (unsigned int k = a.first[c]; k < k1; ++k)
                {
                const float4 h = mt.h[k];
#pragma unroll
                for (int q = 0; q < P; ++q)
                    {
                    const v2f t = h.x * R.g0[q] + h.y * R.g1[q] + h.z * R.g2[q];
                    v2f cs;
                    cs.x = cos2pi<FAST>(t.x);
                    cs.y = cos2pi<FAST>(t.y);
                    sum[q] += cs;
                    sum[q] += h.w * ((cs * cs) * 2.0f - (FAST ? 0.99999994f : 1.0f));     // folded second harmonic (lamellar_device.hpp)
                    }
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
                {
                const float w = R.ok[u] ? s_coeff[c * MTD_MAX_TYPES + R.type[u]] : 0.0f;
                acc[c] += w * sum[u / 2][u % 2];
                }
            }
        }
    }

// lam_force_unscaled from the fractional coordinates already in registers
template<int NCV, bool FAST, int U>
__device__ __forceinline__ void step_force_unscaled(const LamKArgs &a, const ModeTables &mt, StepRegs<NCV, U> &R)
    {
#pragma unroll
    for (int c = 0; c < NCV; ++c)
        {
#pragma unroll
        for (int u = 0; u < U; ++u) R.f[u][c][0] = R.f[u][c][1] = R.f[u][c][2] = 0.0f;
        if (c < (int)a.n_cv)
            {
            const unsigned int k1 = a.first[c + 1];
#pragma unroll 4
            for (unsigned int k = a.first[c]; k < k1; ++k)
                {
                const float4 h = mt.h[k];
                const float4 q = mt.q[k];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    {
                    const float s = sin2pi<FAST>(h.x * R.g0[u / 2][u % 2] + h.y * R.g1[u / 2][u % 2] + h.z * R.g2[u / 2][u % 2]);
                    R.f[u][c][0] += q.x * s;
                    R.f[u][c][1] += q.y * s;
                    R.f[u][c][2] += q.z * s;
                    }
                }
            }
        }
    }

// the same two passes for ONE more particle per lane (waves 1 .. 4), scalar
template<int NCV> struct ExtraRegs
    {
    float g0, g1, g2;
    int type;
    bool ok;
    float f[NCV][3];
    };

template<int NCV, bool FAST>
__device__ __forceinline__ void step_cv_sums_one(const LamKArgs &a, const float *s_coeff, const ModeTables &mt, const ExtraRegs<NCV> &X,
                                                 float (&acc)[NCV])
    {
#pragma unroll
    for (int c = 0; c < NCV; ++c)
        if (c < (int)a.n_cv)
            {
            float sum = 0.0f;
            const unsigned int k1 = a.first[c] + a.nact[c];
#pragma unroll 4
            for (unsigned int k = a.first[c]; k < k1; ++k)
                {
                const float4 h = mt.h[k];
                const float cs = cos2pi<FAST>(h.x * X.g0 + h.y * X.g1 + h.z * X.g2);
                sum += cs;
                sum += h.w * ((cs * cs) * 2.0f - (FAST ? 0.99999994f : 1.0f));
                }
            acc[c] += (X.ok ? s_coeff[c * MTD_MAX_TYPES + X.type] : 0.0f) * sum;
            }
    }

template<int NCV, bool FAST>
__device__ __forceinline__ void step_force_unscaled_one(const LamKArgs &a, const ModeTables &mt, ExtraRegs<NCV> &X)
    {
#pragma unroll
    for (int c = 0; c < NCV; ++c)
        {
        X.f[c][0] = X.f[c][1] = X.f[c][2] = 0.0f;
        if (c < (int)a.n_cv)
            {
            const unsigned int k1 = a.first[c + 1];
#pragma unroll 4
            for (unsigned int k = a.first[c]; k < k1; ++k)
                {
                const float4 h = mt.h[k];
                const float4 q = mt.q[k];
                const float sn = sin2pi<FAST>(h.x * X.g0 + h.y * X.g1 + h.z * X.g2);
                X.f[c][0] += q.x * sn;
                X.f[c][1] += q.y * sn;
                X.f[c][2] += q.z * sn;
                }
            }
        }
    }

template<typename S4, int NCV, bool FAST, bool COMM>
__global__ __launch_bounds__(FS_THREADS, 1) void k_fused_step(const LamKArgs a, const S4 *__restrict__ postype, const ForcePtrs out,
                                                           const unsigned int N, const unsigned int chunk, const double two_over_n,
                                                           const MetadCfg c, const int deposit, const unsigned int cells_per_block,
                                                           const CommK lk, const CommK ck)
    {
    typedef typename scalar4_traits<S4>::scalar scalar;
    constexpr int NS = NCV;
    __shared__ ChainResult s_chain;
    __shared__ float s_coeff[MTD_MAX_CV * MTD_MAX_TYPES];
    __shared__ float s_wcoef[MTD_MAX_CV * MTD_MAX_TYPES];
    __shared__ double s_wave[FS_WAVES * NCV];
    __shared__ double s_red[2 * FS_WAVES];
    __shared__ double s_avg[2];

    __shared__ ModeTables s_cvt;                         // CV pass: dense list of the visited modes (fold flag in w)
    __shared__ ModeTables s_mt;                          // force pass: every mode with its wave vector

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const unsigned int b = blockIdx.x, nb = gridDim.x;
    const unsigned int p0 = b * chunk;
    const unsigned int p1 = min(N, p0 + chunk);
    if (threadIdx.x == 0) MTD_BSTAMP(0);

    MTD_STAMP(0, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(16, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0); MTD_STAMP(32, blockIdx.x == 0 && threadIdx.x == 64);
    // ---- phase 0: the particles are requested before the tables are staged; wave 0 requests the bias-grid patch instead
    StepRegs<NCV, FS_U> R;
    ExtraRegs<NCV> X;
    RawGroup<S4, FS_U> raw;
    S4 raw_x;
    const bool has_extra = wave >= 1 && wave <= (int)FS_EXTRA_WAVES;
    const unsigned int slot0 = (unsigned int)(wave - 1) * MTD_WAVE + lane;           // (waves >= 1)
    const unsigned int ix = p0 + FS_U * FS_MAIN + slot0;
    GridPatch patch;
    X.ok = false;
    if (wave == 0)
        {
#pragma unroll
        for (int u = 0; u < FS_U; ++u) R.ok[u] = false;
        patch = chain_prefetch(c);
        }
    else
        {
#pragma unroll
        for (int u = 0; u < FS_U; ++u)
            {
            const unsigned int i = p0 + u * FS_MAIN + slot0;
            R.ok[u] = i < p1;
            if (N) raw.v[u] = postype[R.ok[u] ? i : N - 1];
            }
        X.ok = has_extra && ix < p1;
        if (N && has_extra) raw_x = postype[X.ok ? ix : N - 1];
        }
    load_coeff(a, s_coeff);
    load_modes_cv(a, s_cvt);
    load_modes(a, s_mt, true);
    __syncthreads();

    MTD_STAMP(1, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(17, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0); MTD_STAMP(33, blockIdx.x == 0 && threadIdx.x == 64);
    // ---- phase 1: per-CV sums
    float acc[NCV];
#pragma unroll
    for (int i = 0; i < NCV; ++i) acc[i] = 0.0f;
    if (wave != 0)
        {
#pragma unroll
        for (int u = 0; u < FS_U; ++u)
            {
            float x0 = 0.0f, x1 = 0.0f, x2 = 0.0f;
            R.type[u] = 0;
            if (N)
                {
                const Particle p = scalar4_traits<S4>::unpack(raw.v[u]);
                project(a, p, x0, x1, x2);
                R.type[u] = p.type;
                }
            R.g0[u / 2][u % 2] = x0;
            R.g1[u / 2][u % 2] = x1;
            R.g2[u / 2][u % 2] = x2;
            }
        step_cv_sums<NCV, FAST, FS_U>(a, s_coeff, s_cvt, R, acc);
        if (has_extra)
            {
            X.g0 = X.g1 = X.g2 = 0.0f;
            X.type = 0;
            if (N)
                {
                const Particle p = scalar4_traits<S4>::unpack(raw_x);
                project(a, p, X.g0, X.g1, X.g2);
                X.type = p.type;
                }
            step_cv_sums_one<NCV, FAST>(a, s_coeff, s_cvt, X, acc);
            }
        }
    MTD_STAMP(2, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(18, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0); MTD_STAMP(34, blockIdx.x == 0 && threadIdx.x == 64);
    // block sum: the positions stream in over ~3 us, so the waves finish this phase that far apart; the early ones wait at the
    // barrier (letting them run ahead into their force arithmetic was measured: the late waves then share their SIMDs with
    // it, the block's sums left 1.5 us later and every block waits for the slowest one)
#pragma unroll
    for (int i = 0; i < NCV; ++i)
        {
        const float v = wave_sum(acc[i]);
        if (lane == 0) s_wave[wave * NCV + i] = (double)v;
        }
    __syncthreads();
    if (threadIdx.x < NS * FS_LL_REPLICAS)
        {
        const int i = threadIdx.x % NS, rep = threadIdx.x / NS;
        double r = 0.0;
        for (int w = 0; w < FS_WAVES; ++w) r += s_wave[w * NCV + i];
        ll_store_column(lk.ll + (size_t)rep * FS_LL_REPLICA_WORDS, FS_LL_PITCH, 2 * i, b, lk.seq, r);      // hand-off 1
        if (threadIdx.x == 0) MTD_BSTAMP(1);
        }
    __builtin_amdgcn_s_setprio(0);
    MTD_STAMP(3, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(19, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0); MTD_STAMP(35, blockIdx.x == 0 && threadIdx.x == 64);

    // ---- phase 2: wave 0 runs the chain, the other waves form their unscaled forces meanwhile
    double c_wt = 0.0, c_wold = 0.0, c_dV = 0.0;         // wave 0 of block 0: weight-grid corners for the closed-form w(s)
    // block 0 on a step without deposit also reads w(s) off the (final) weight grid in the same pass
    const bool closed = deposit != 0 || b != 0;
    if (wave == 0)
        {
        __builtin_amdgcn_s_setprio(3);
        double tot[3] = { 0.0, 0.0, 0.0 };
        bool expired = false;
        if (!COMM || b == 0)
            {
            // block sums of ALL blocks: lane l adds blocks l, l + 64, ... in order, then the xor butterfly — every block
            // performs the identical additions, so every block holds the same bits
            double v[3] = { 0.0, 0.0, 0.0 };
            CommK rk = lk;
            rk.ll = lk.ll + (size_t)(b % FS_LL_REPLICAS) * FS_LL_REPLICA_WORDS;
            // the blocks finish their sums within ~1 us of each other: the first probe leaves a little after this block's own
            // post (a probe that misses costs a whole further round trip)
            __builtin_amdgcn_s_sleep(FS_FIRST_PROBE_SLEEP);
            expired = ll_collect_columns<NS>(rk, nb, FS_LL_PITCH, 0, v);
#pragma unroll
            for (int i = 0; i < NS; ++i) tot[i] = wave_sum(v[i]);
            if (expired)
#pragma unroll
                for (int i = 0; i < 3; ++i) tot[i] = comm_poison();
            }
        MTD_STAMP(4, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(20, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0);
        if (lane == 0) MTD_BSTAMP(2);
        if (COMM && b == 0) comm_send_wave(ck, tot, NS);                 // this rank's totals into every rank's mailbox
        const ChainResult r = chain_wave(c, deposit != 0, closed, COMM ? &ck : nullptr, COMM ? nullptr : tot, b == 0 && deposit != 0, &patch);
        c_wt = r.c_wt; c_wold = r.c_wold; c_dV = r.c_dV;
        MTD_STAMP(5, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(21, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0);
        if (lane == 0) MTD_BSTAMP(3);
        if (lane == 0)
            {
            s_chain.cv[0] = r.cv[0]; s_chain.cv[1] = r.cv[1]; s_chain.cv[2] = r.cv[2];
            s_chain.bias[0] = r.bias[0]; s_chain.bias[1] = r.bias[1]; s_chain.bias[2] = r.bias[2];
            s_chain.scal = r.scal; s_chain.V = r.V; s_chain.w = r.w;
            s_chain.bin = r.bin; s_chain.on_grid = r.on_grid; s_chain.oob = r.oob; s_chain.failed = r.failed;
            }
        for (unsigned int i = lane; i < NCV * MTD_MAX_TYPES; i += MTD_WAVE)
            {
            const unsigned int cv = i / MTD_MAX_TYPES;
            const double bf = cv == 0 ? r.bias[0] : (cv == 1 ? r.bias[1] : r.bias[2]);
            s_wcoef[i] = (cv < a.n_cv) ? (float)((double)a.coeff[cv][i % MTD_MAX_TYPES] * bf * two_over_n) : 0.0f;
            }
        __builtin_amdgcn_s_setprio(0);
        }
    else
        {
        step_force_unscaled<NCV, FAST, FS_U>(a, s_mt, R);
        if (has_extra) step_force_unscaled_one<NCV, FAST>(a, s_mt, X);
        }
    MTD_STAMP(6, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(22, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0); MTD_STAMP(38, blockIdx.x == 0 && threadIdx.x == 64);
    lds_barrier();                                                   // (publishes LDS data only: nothing drains behind it)
    MTD_STAMP(7, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(23, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0); MTD_STAMP(39, blockIdx.x == 0 && threadIdx.x == 64);

    // ---- phase 3: scaled forces out.  On a deposit step the waves that own grid cells run the first grid pass BEFORE their
    //      stores (its loads would otherwise queue behind 32 MB of force stores, and its sums are what every block waits
    //      for next); the other waves store at once
    auto store_forces = [&]()
        {
#pragma unroll
        for (int cvi = 0; cvi < NCV; ++cvi)
            {
            if (cvi < (int)a.n_cv)
                {
                S4 *f = (S4 *)out.f[cvi];
#pragma unroll
                for (int u = 0; u < FS_U; ++u)
                    if (R.ok[u])
                        {
                        const float w = s_wcoef[cvi * MTD_MAX_TYPES + R.type[u]];
                        nt_store(scalar4_traits<S4>::make((scalar)(R.f[u][cvi][0] * w), (scalar)(R.f[u][cvi][1] * w), (scalar)(R.f[u][cvi][2] * w), (scalar)0),
                                 &f[p0 + u * FS_MAIN + slot0]);
                        }
                if (X.ok)
                    {
                    const float w = s_wcoef[cvi * MTD_MAX_TYPES + X.type];
                    nt_store(scalar4_traits<S4>::make((scalar)(X.f[cvi][0] * w), (scalar)(X.f[cvi][1] * w), (scalar)(X.f[cvi][2] * w), (scalar)0), &f[ix]);
                    }
                }
            }
        };

    const bool failed = s_chain.failed != 0;
    MTD_STAMP(8, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(24, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0); MTD_STAMP(40, blockIdx.x == 0 && threadIdx.x == 64);
    const bool dep = deposit != 0 && !failed;
    const unsigned int cell0 = min(c.len, b * cells_per_block), cell1 = min(c.len, cell0 + cells_per_block);
    const bool grid_wave = dep && cell0 + (unsigned int)wave * MTD_WAVE < cell1;      // wave-uniform
    if (!grid_wave) store_forces();
    double avg_dV = 0.0;
    bool grid_expired = false;
    if (dep)
        {
        // first grid pass on this block's slice: updateGrid (:1002-1047), updateHistogram (:1092-1119), updateSigmaGrid
        // (:1122-1155), first loop of updateReweightedEstimator (:1070-1075)
        double s1 = 0.0, s2 = 0.0;
        for (unsigned int g = cell0 + threadIdx.x; g < cell1; g += FS_THREADS)
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
            s1 += Rw * dV;
            s2 += Rw;
            }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0)
            {
            s_red[2 * wave] = s1;
            s_red[2 * wave + 1] = s2;
            }
        lds_barrier();                                               // (not __syncthreads: it would wait for the force stores to drain)
        if (threadIdx.x < 2 * FS_LL_REPLICAS)
            {
            const unsigned int i = threadIdx.x & 1, rep = threadIdx.x >> 1;
            double t = 0.0;
            for (int w = 0; w < FS_WAVES; ++w) t += s_red[2 * w + i];
            ll_store_column(lk.ll + (size_t)rep * FS_LL_REPLICA_WORDS, FS_LL_PITCH, 2 * (CHAIN_MAX_CV + i), b, lk.seq, t);   // hand-off 2
            if (threadIdx.x == 0) MTD_BSTAMP(4);
            }
        if (grid_wave) store_forces();
    MTD_STAMP(9, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(25, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0); MTD_STAMP(41, blockIdx.x == 0 && threadIdx.x == 64);
        // ---- phase 4: <dV> from all blocks (same additions in every block), second reweighting pass + accumulate (:1077-1087,
        //      :426-437) on the same slice
        if (wave == 0)
            {
            double v[3] = { 0.0, 0.0, 0.0 };
            CommK rk = lk;
            rk.ll = lk.ll + (size_t)(b % FS_LL_REPLICAS) * FS_LL_REPLICA_WORDS;
            const bool ex = ll_collect_columns<2>(rk, nb, FS_LL_PITCH, 2 * CHAIN_MAX_CV, v);
            const double t1 = wave_sum(v[0]), t2 = wave_sum(v[1]);
            if (lane == 0)
                {
                s_avg[0] = t1 / t2;                                        // norm == 0 -> NaN like the reference (Q15)
                s_avg[1] = ex ? 1.0 : 0.0;
                }
            }
        lds_barrier();
        avg_dV = s_avg[0];
    MTD_STAMP(10, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(26, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0); MTD_STAMP(42, blockIdx.x == 0 && threadIdx.x == 64);
        grid_expired = s_avg[1] != 0.0;
        if (!grid_expired)
            for (unsigned int g = cell0 + threadIdx.x; g < cell1; g += FS_THREADS)
                {
                const double dV = c.grid_delta[g];
                const double fac = exp(-(dV - avg_dV) / c.temp);             // T, not deltaT (:1084)
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
        // (this kernel changes the grid without keeping the two-launch step's grid patch, MetadState::patch_v, current)
        if (blockIdx.x == 0 && threadIdx.x == 0) c.st->patch_valid = 0;
        }

    if (threadIdx.x == 0) MTD_BSTAMP(5);
    // one block publishes the step's scalars for the host (lazy read-back)
    MTD_STAMP(11, blockIdx.x == 0 && threadIdx.x == 0); MTD_STAMP(27, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0); MTD_STAMP(43, blockIdx.x == 0 && threadIdx.x == 64);
    if (b == 0 && wave == 0)
        {
        double w_now = s_chain.w;                                          // step without deposit: read in phase 2
        if (dep && !grid_expired)
            {
            // w(s) of the weight grid AFTER this step's second reweighting pass, from the corner values read in phase 2:
            // weight_new = weight_old / exp(-(dV - <dV>) / T) (:1084-1086), multilinear sum in corner order (:711-733)
            const int n_term = 1 << c.n_cv;
            double term = 0.0;
            if (lane < n_term && c_wt != 0.0) term = c_wt * (c_wold / exp(-(c_dV - avg_dV) / c.temp));
            w_now = 0.0;
            for (int q = 0; q < n_term; ++q) w_now += __shfl(term, q, MTD_WAVE);
            }
        if (lane < (int)c.n_cv)
            {
            c.st->cv[lane] = lane == 0 ? s_chain.cv[0] : (lane == 1 ? s_chain.cv[1] : s_chain.cv[2]);
            c.st->bias[lane] = lane == 0 ? s_chain.bias[0] : (lane == 1 ? s_chain.bias[1] : s_chain.bias[2]);
            }
        if (lane == 0)
            {
            c.st->V = s_chain.V;
            c.st->failed = (failed || grid_expired) ? 1u : 0u;
            c.st->bin = s_chain.bin;
            c.st->on_grid = (unsigned int)s_chain.on_grid;
            if (failed)
                c.st->w = s_chain.V;                                       // NaN
            else if (deposit)
                {
                c.st->scal = s_chain.scal;
                if (!grid_expired)
                    {
                    c.st->w = w_now;
                    c.st->avg_dV = avg_dV;
                    c.st->num_gaussians += 1;                              // :440
                    }
                }
            else
                {
                c.st->w = w_now;
                if (s_chain.on_grid) c.hist_delta[s_chain.bin] += 1;       // updateHistogram on a step without deposit (:366)
                }
            if (s_chain.oob) c.st->n_oob += (deposit && c.mode == MTD_MODE_WELL_TEMPERED) ? 2 : 1;
            }
        }
    }

// blocks of `kernel` the device holds at one time (see fused.hip: resident_capacity)
unsigned int step_capacity(const void *kernel)
    {
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, unsigned int> cache;      // per device
    int per_cu = 0, dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        {
        (void)hipGetLastError();
        return 0;
        }
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(std::make_pair(dev, kernel));
    if (it != cache.end()) return it->second;
    unsigned int cap = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, FS_THREADS, 0) == hipSuccess)
        cap = (unsigned int)per_cu * (unsigned int)n_cu;
    else
        (void)hipGetLastError();
    cache[std::make_pair(dev, kernel)] = cap;
    return cap;
    }

unsigned long long step_timeout_ticks()
    {
    const char *e = std::getenv("MTD_COMM_TIMEOUT_MS");
    double ms = 5000.0;
    if (e && *e) ms = std::atof(e);
    if (!(ms > 0.0)) ms = 5000.0;
    return (unsigned long long)(ms * 1.0e5);            // wall_clock64: 100 MHz
    }

int step_buffers(mtd_metad *m)
    {
    if (m->d_ll) return MTD_SUCCESS;
    void *ll = nullptr, *err = nullptr;
    unsigned int *h = nullptr, *dh = nullptr;
    const size_t ll_bytes = sizeof(unsigned long long) * FS_LL_REPLICAS * FS_LL_REPLICA_WORDS;
    hipError_t e = hipMalloc(&ll, ll_bytes);
    if (e == hipSuccess) e = hipMemset(ll, 0, ll_bytes);
    if (e == hipSuccess) e = hipMalloc(&err, 64);
    if (e == hipSuccess) e = hipMemset(err, 0, 64);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h, sizeof(unsigned int), hipHostMallocMapped);
    if (e == hipSuccess)
        {
        *h = 0;
        e = hipHostGetDevicePointer((void **)&dh, h, 0);
        }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess)
        {
        (void)hipGetLastError();
        if (ll) (void)hipFree(ll);
        if (err) (void)hipFree(err);
        if (h) (void)hipHostFree(h);
        return (int)e;
        }
    m->d_ll = (unsigned long long *)ll;
    m->d_step_err = (unsigned int *)err;
    m->h_step_err = h;
    m->d_step_err_host = dh;
    return MTD_SUCCESS;
    }

template<typename S4, bool FAST>
const void *step_kernel(unsigned int n_cv, bool comm)
    {
    switch (n_cv)
        {
        case 1: return comm ? (const void *)k_fused_step<S4, 1, FAST, true> : (const void *)k_fused_step<S4, 1, FAST, false>;
        case 2: return comm ? (const void *)k_fused_step<S4, 2, FAST, true> : (const void *)k_fused_step<S4, 2, FAST, false>;
        default: return comm ? (const void *)k_fused_step<S4, 3, FAST, true> : (const void *)k_fused_step<S4, 3, FAST, false>;
        }
    }

} // namespace

namespace mtd
{
void fused_step_release(mtd_metad *m)
    {
    if (m->d_ll) (void)hipFree(m->d_ll);
    if (m->d_step_err) (void)hipFree(m->d_step_err);
    if (m->h_step_err) (void)hipHostFree((void *)m->h_step_err);
    m->d_ll = nullptr;
    m->d_step_err = nullptr;
    m->h_step_err = nullptr;
    }
}

extern "C" {

int mtd_fused_step(mtd_metad *m, const mtd_lamellar_set *set, unsigned int n_particles, const void *d_postype,
                   void *const *d_force, int dtype, unsigned int n_global, const mtd_box *global_box, double *d_scratch,
                   unsigned int timestep, mtd_stream_t stream)
    {
    if (!m || !set) return MTD_ERR_INVALID_ARGUMENT;
    LamKArgs k;
    int rc = fill_kargs(k, set, global_box);
    if (rc) return rc;
    if (!d_force || !d_scratch || n_global == 0 || (n_particles && !d_postype)) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    if (set->n_cv != m->cfg.n_cv) return MTD_ERR_UNSUPPORTED;
    for (unsigned int c = 0; c < set->n_cv; ++c)
        if (!d_force[c] && n_particles) return MTD_ERR_INVALID_ARGUMENT;
    if (m->h_step_err && *m->h_step_err) return MTD_ERR_COMM_TIMEOUT;
    hipStream_t s = (hipStream_t)stream;
    // which form runs: the engine's setting (mtd_fused_step_set_mode), else MTD_FUSED_STEP=1 / 0, else the two-launch form —
    // measured faster at the headline size (21.9 against 23.9 us per step)
    static const int env_mode = [] { const char *e = std::getenv("MTD_FUSED_STEP"); return !e ? -1 : (e[0] == '1' ? 1 : 0); }();
    const bool off = m->step_mode >= 0 ? m->step_mode == 0 : env_mode != 1;
    const bool fast = lam_fast_trig(k) != 0;
    const bool comm = m->comm != nullptr;

    unsigned int nb = (n_particles + FS_CHUNK - 1) / FS_CHUNK;
    if (nb < 1) nb = 1;
    bool one_launch = !off && set->n_cv <= (unsigned int)CHAIN_MAX_CV && nb <= FS_MAX_BLOCKS;
    const void *kern = nullptr;
    if (one_launch)
        {
        // enough blocks for the grid passes too (a block's slice of the bias grid: a few cells per thread at most)
        const unsigned int nb_grid = (m->cfg.len + 4 * FS_THREADS - 1) / (4 * FS_THREADS);
        if (nb < nb_grid) nb = nb_grid > FS_MAX_BLOCKS ? FS_MAX_BLOCKS : nb_grid;
        // spread the particles evenly over the blocks (not 4096 each with a short last one)
        if (dtype == MTD_F32)
            kern = fast ? step_kernel<float4, true>(set->n_cv, comm) : step_kernel<float4, false>(set->n_cv, comm);
        else
            kern = fast ? step_kernel<double4, true>(set->n_cv, comm) : step_kernel<double4, false>(set->n_cv, comm);
        if (nb > step_capacity(kern)) one_launch = false;          // blocks wait for each other: the whole grid must be resident
        }
    if (!one_launch)
        {
        // the two-launch form (fused.hip): CV pass + deferred grid pass, then chain + first grid pass + forces
        unsigned int n_partials = 0;
        rc = mtd_fused_cv_pass(m, set, n_particles, d_postype, dtype, global_box, d_scratch, &n_partials, stream);
        if (rc) return rc;
        for (unsigned int c = 0; c < set->n_cv; ++c)
            {
            rc = mtd_metad_set_cv_source(m, c, d_scratch, n_partials, set->n_cv, c, 1.0 / (double)n_global, 0.0);
            if (rc) return rc;
            }
        rc = mtd_fused_force_pass(m, set, n_particles, d_postype, d_force, dtype, n_global, global_box, timestep, stream);
        if (rc) return rc;
        m->last_launches = 2;
        return MTD_SUCCESS;
        }

    rc = metad_flush(m, s);                                     // (a deposit of the two-launch form may still be pending)
    if (rc) return rc;
    rc = step_buffers(m);
    if (rc) return rc;
    CommK lk, ck;
    std::memset(&lk, 0, sizeof(lk));
    std::memset(&ck, 0, sizeof(ck));
    if (comm)
        {
        if (comm_failed(m->comm)) return MTD_ERR_COMM_TIMEOUT;
        rc = comm_next(m->comm, ck);                            // block 0 sends exchange seq, every block's chain receives it
        if (rc) return rc;
        }
    m->step_seq = (m->step_seq == 0xffffffffu) ? 1u : m->step_seq + 1u;
    lk.ll = m->d_ll;
    lk.seq = m->step_seq;
    lk.err = m->d_step_err;
    lk.err_host = m->d_step_err_host;
    lk.timeout_ticks = step_timeout_ticks();
    lk.world = 1;

    MetadCfg cfg = m->cfg;
    for (unsigned int c = 0; c < cfg.n_cv; ++c)
        {
        cfg.src[c].partials = nullptr;
        cfg.src[c].n_partials = 0;
        cfg.src[c].scale = 1.0 / (double)n_global;
        cfg.src[c].shift = 0.0;
        }
    ForcePtrs out;
    for (unsigned int c = 0; c < MTD_MAX_CV; ++c) out.f[c] = c < set->n_cv ? d_force[c] : nullptr;
    const int dep = (m->add_bias && (timestep % m->stride == 0)) ? 1 : 0;   // .cc:368
    const double two_over_n = 2.0 / (double)n_global;
    unsigned int chunk = (n_particles + nb - 1) / nb;
    chunk = (chunk + MTD_WAVE - 1) / MTD_WAVE * MTD_WAVE;                 // whole waves of consecutive particles
    if (chunk > FS_CHUNK) chunk = FS_CHUNK;
    if (chunk == 0) chunk = MTD_WAVE;
    const unsigned int cells_per_block = (cfg.len + nb - 1) / nb;

#define MTD_LAUNCH_FS(S4, NCV, FASTV, COMMV) \
    k_fused_step<S4, NCV, FASTV, COMMV><<<nb, FS_THREADS, 0, s>>>(k, (const S4 *)d_postype, out, n_particles, chunk, two_over_n, cfg, dep, cells_per_block, lk, ck)
#define MTD_LAUNCH_FS_NCV(S4, FASTV, COMMV) \
    switch (set->n_cv) { case 1: MTD_LAUNCH_FS(S4, 1, FASTV, COMMV); break; case 2: MTD_LAUNCH_FS(S4, 2, FASTV, COMMV); break; default: MTD_LAUNCH_FS(S4, 3, FASTV, COMMV); break; }
#define MTD_LAUNCH_FS_ALL(S4) \
    do { if (fast) { if (comm) { MTD_LAUNCH_FS_NCV(S4, true, true) } else { MTD_LAUNCH_FS_NCV(S4, true, false) } } \
         else { if (comm) { MTD_LAUNCH_FS_NCV(S4, false, true) } else { MTD_LAUNCH_FS_NCV(S4, false, false) } } } while (0)
    if (dtype == MTD_F32)
        MTD_LAUNCH_FS_ALL(float4);
    else
        MTD_LAUNCH_FS_ALL(double4);
#undef MTD_LAUNCH_FS_ALL
#undef MTD_LAUNCH_FS_NCV
#undef MTD_LAUNCH_FS
    MTD_LAUNCH_CHECK();
    m->pending_apply = 0;
    m->w_stale = 0;                                             // (w(s) comes out of the launch in closed form)
    m->last_launches = 1;
    return MTD_SUCCESS;
    }

unsigned int mtd_fused_step_launches(const mtd_metad *m) { return m ? m->last_launches : 0; }

int mtd_fused_step_set_mode(mtd_metad *m, int mode)
    {
    if (!m || mode < -1 || mode > 1) return MTD_ERR_INVALID_ARGUMENT;
    m->step_mode = mode;
    return MTD_SUCCESS;
    }

} // extern "C"
