// steinhardt.hip — Steinhardt Q_l order parameter on gfx950.
//
// Reference: SteinhardtQl.cc:62-201 (computeCV), :203-339 (computeBiasForces), :36-60 (smoothing), with the
// spherical harmonics of spherical_harmonics.hpp:32-246 (fsph).  The reference has NO GPU implementation of this
// CV (SURVEY §2.4 "new"): it runs a host loop that constructs a PointSPHEvaluator (five heap arrays) per pair.
//
// MI355X design.  Neither pass is HBM bound (40 B per pair against ~0.3 kflop of fp64); both are bound by instruction ISSUE — a
// SIMD issues one instruction per four cycles of whatever kind, so scalar moves and branches cost what fp64 FMAs cost
// (profiles/r3: 97 % of the issue slots of the force pass in use before it was rewritten for the slot count):
//   units             a block walks chunks of 64 consecutive central particles; the neighbour-list entries of a chunk, flattened,
//                     are cut into units of <= 1024 entries, one per thread and round, so the waves stay full whatever the
//                     per-particle neighbour counts are; every memory trip of a unit — (own particle, list head, count) -> list
//                     entries -> neighbour positions — is requested a stage ahead and stays in flight across the arithmetic
//                     of the units before it (QlUnit / QlFeed, LDS-only barriers)
//   monic amplitudes  Y_lm = nrm(l, m) p_m,l-m(cos theta) h^m with h = sin(theta) e^{i phi} = (dx + i dy) / r and the monic form of
//                     fsph's Jacobi recurrence (one constant and two instructions per entry); no acos / atan2, no heap;
//                     constants from a table in device memory through the scalar cache (QlTab)
//   k_ql_accumulate   Q'_lm = sum f Y_lm: every (l, m >= 0) in the registers of ONE wave up to lmax = 8 (49 doubles at lmax = 6;
//                     28 for half / symmetric lists, whose odd degrees the finalize step zeroes); entries filtered and compacted
//                     before the neighbour is fetched; lane sums through LDS in a fixed order; negative m are conjugates, the
//                     normalisation is applied once per block, the Condon-Shortley phase in the finalize step
//   k_reduce_partials (lamellar.hip) -> Q'_lm ; [multi-GPU: all-reduce of (lmax+1)(lmax+2) doubles here]
//   k_ql_finalize     full Q_lm table in the reference's order, third-law scaling, Q_l, CV value
//   k_ql_finalize_chain  (round 4) the same as the HEAD of the bias-grid engine's launch when cv.steinhardt is the grid's only
//                     variable: value -> scalar chain -> first grid pass in one launch (mtd_ql_finalize_update_bias); the
//                     engine's deferred pass then rides in the force pass of the same step
//   k_ql_forces       the spherical-basis gradient of :287-321 contracted BEFORE it is expanded: with Z_lm = h^m q_lm
//                     (q_lm = nrm w_l conj(Q_lm) in LDS) the pair force is -(alpha U + beta V + gamma W) with three scalars
//                     U = sum P Re Z, V = cot sum m P Re Z + sin sum d_lm p_m+1,l-m-1 Re Z, W = -sum m P Im Z and three real
//                     vectors alpha = d f'/r, beta = (f/r) e_theta, gamma = (f/rho) e_phi; m < 0 terms equal their m > 0
//                     partners (weight 2), degrees with Ql_ref[l] = 0 are skipped as a whole.  Per-particle sums over the
//                     pairs of a unit in a fixed order through LDS (deterministic, no atomics for full lists).
// Double precision throughout.
#include "mtd_device.hpp"
#include "metad_host.hpp"

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

namespace
{

using namespace mtd;

constexpr int QL_THREADS = 256;
constexpr unsigned int QL_MAX_BLOCKS = 1024;

template<int LMAX> struct QlArgs
    {
    double lo[3], L[3], Linv[3], xy, xz, yz;
    double rcutsq, ronsq, r_on, r_cut, inv_width;
    unsigned int lmax, type, N, n_global;
    int half_nlist, _pad;
    double ql_ref[LMAX + 1];
    };

// Jacobi recurrence prefactors (spherical_harmonics.hpp:151-175) and jacobi[m][0] (:197-201).  They depend on (m, n) only,
// so in the fully unrolled loops they fold to literals: no table in the kernel arguments, no scalar registers tied up
// (a [m][n] table in the argument segment cost ~340 v_readlane per pair in spilled scalars)
__host__ __device__ __forceinline__ double jac_f0(const int m, const int n)
    {
    return 2 * sqrt(1 + (m - 0.5) / n) * sqrt(1 - (m - 0.5) / (n + 2 * m));
    }
__host__ __device__ __forceinline__ double jac_f1(const int m, const int n)
    {
    return -sqrt(1.0 + 4.0 / (2 * n + 2 * m - 3)) * sqrt(1 - 1.0 / n) * sqrt(1.0 - 1.0 / (n + 2 * m));
    }
__host__ __device__ __forceinline__ double jac_0(const int m)
    {
    double v = 0.70710678118654752440084436210484903928483593768847;    // 1 / sqrt(2)
    for (int k = 1; k <= m; ++k) v *= sqrt(1 + 1.0 / 2 / k);
    return v;
    }

struct cplx
    {
    double re, im;
    };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cconj(cplx a) { return {a.re, -a.im}; }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return {a.re * s, a.im * s}; }

template<int LMAX>
__device__ __forceinline__ void min_image(const QlArgs<LMAX> &a, double &x, double &y, double &z)
    {
    // HOOMD BoxDim::minImage: nearest image counts from the reciprocal box lengths
    double img = rint(z * a.Linv[2]);
    z -= a.L[2] * img;
    y -= a.L[2] * a.yz * img;
    x -= a.L[2] * a.xz * img;
    img = rint(y * a.Linv[1]);
    y -= a.L[1] * img;
    x -= a.L[1] * a.xy * img;
    x -= a.L[0] * rint(x * a.Linv[0]);
    }

// ---- constants of the pair passes, read through the scalar cache -------------------------------------------------------------
// Counters of round 3 (profiles/r3): the pair kernels are bound by instruction ISSUE — vector + scalar + LDS + branch
// instructions times four cycles add up to the launch time.  A 64-bit literal is two s_mov_b32, and the ~100 literals of a pair
// (recurrence prefactors, derivative factors, the smoothing polynomial) were a quarter of all issue slots (or, where the
// compiler kept them in VGPRs, a v_mov per use and 44 registers).  They now sit in a small table in device memory that the
// kernel reads with s_load_dwordx8/x16 (eight constants per issue slot) right where they are used; the table pointer carries an
// offset the compiler cannot see through (always zero), or it would hoist ~90 loads out of the pair loop and spill them.
//   [SM_SIN, +10)  [SM_COS, +10)   Taylor coefficients of sincospi_unit_tab, highest order first
//   beta(m, n)     monic form of the Jacobi recurrence of spherical_harmonics.hpp:203-211 in the degree n = l - m:
//                  J_m(n) = kappa(m, n) p_mn(x), p_m0 = 1, p_m1 = x, p_mn = x p_m,n-1 - beta(m, n) p_m,n-2
//                  (kappa(m, 0) = jacobi[m][0], kappa(m, n) = f0(m, n) kappa(m, n - 1), beta = -f1(m, n) / (f0(m, n) f0(m, n - 1)))
//   nrm(l, m)      (-1)^m kappa(m, l - m) / sqrt(2 pi): Y_lm = nrm(l, m) p_m,l-m(cos theta) (sin theta e^{i phi})^m
//   d(l, m)        sqrt((l - m)(l + m + 1)) nrm(l, m + 1) / nrm(l, m): the A_l,m+1 term of dY_lm/dtheta (:305-309)
template<int LMAX> struct QlTab
    {
    static constexpr int SM_SIN = 0, SM_COS = 10, BETA = 24;
    static constexpr int N_BETA = (LMAX - 1) * LMAX / 2;
    static constexpr int D = (BETA + N_BETA + 7) / 8 * 8;
    static constexpr int N_D = LMAX * (LMAX + 1) / 2;
    static constexpr int NRM = (D + N_D + 7) / 8 * 8;
    static constexpr int SIZE = NRM + (LMAX + 1) * (LMAX + 2) / 2;
    __host__ __device__ static constexpr int beta(const int m, const int n)         // n >= 2, m + n <= LMAX
        {
        return BETA + m * (LMAX - 1) - m * (m - 1) / 2 + (n - 2);
        }
    __host__ __device__ static constexpr int d(const int l, const int m) { return D + l * (l - 1) / 2 + m; }       // m < l
    __host__ __device__ static constexpr int nrm(const int l, const int m) { return NRM + l * (l + 1) / 2 + m; }
    };

template<int LMAX> void ql_build_table(double *t)
    {
    typedef QlTab<LMAX> T;
    for (int i = 0; i < T::SIZE; ++i) t[i] = 0.0;
    // sin: -1/21!, 1/19!, ..., 1/3! ; cos: 1/20!, -1/18!, ..., -1/2!   (sincospi_unit_tab)
    double fact = 1.0;                                       // k!
    double inv[22];
    inv[0] = 1.0;
    for (int k = 1; k <= 21; ++k)
        {
        fact *= k;
        inv[k] = 1.0 / fact;
        }
    for (int i = 0; i < 10; ++i)
        {
        const int ks = 21 - 2 * i, kc = 20 - 2 * i;
        t[T::SM_SIN + i] = (i % 2 == 0 ? -1.0 : 1.0) * inv[ks];
        t[T::SM_COS + i] = (i % 2 == 0 ? 1.0 : -1.0) * inv[kc];
        }
    double kappa[LMAX + 1][LMAX + 1];
    for (int m = 0; m <= LMAX; ++m)
        {
        kappa[m][0] = jac_0(m);
        for (int n = 1; m + n <= LMAX; ++n) kappa[m][n] = jac_f0(m, n) * kappa[m][n - 1];
        for (int n = 2; m + n <= LMAX; ++n) t[T::beta(m, n)] = -jac_f1(m, n) / (jac_f0(m, n) * jac_f0(m, n - 1));
        }
    for (int l = 0; l <= LMAX; ++l)
        for (int m = 0; m <= l; ++m)
            t[T::nrm(l, m)] = ((m % 2) ? -1.0 : 1.0) * 0.3989422804014326779399460599343818684758586311649 * kappa[m][l - m];
    for (int l = 1; l <= LMAX; ++l)
        for (int m = 0; m < l; ++m)
            t[T::d(l, m)] = std::sqrt((double)((l - m) * (l + m + 1))) * t[T::nrm(l, m + 1)] / t[T::nrm(l, m)];
    }

// the table of this LMAX on the current device (built and uploaded once per device)
template<int LMAX> const double *ql_device_table(hipStream_t s, int &rc)
    {
    static std::mutex mtx;
    static std::map<int, double *> tables;
    rc = MTD_SUCCESS;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        {
        rc = MTD_ERR_INVALID_ARGUMENT;
        return nullptr;
        }
    std::lock_guard<std::mutex> lock(mtx);
    auto it = tables.find(dev);
    if (it != tables.end()) return it->second;
    static double host[QlTab<LMAX>::SIZE];
    ql_build_table<LMAX>(host);
    double *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, sizeof(host));
    if (e == hipSuccess) e = hipMemcpy(d, host, sizeof(host), hipMemcpyHostToDevice);      // once per device: synchronous
    if (e != hipSuccess)
        {
        (void)hipGetLastError();
        if (d) (void)hipFree(d);
        rc = (int)e;
        return nullptr;
        }
    (void)s;
    tables.emplace(dev, d);
    return d;
    }

// a * b + c with c a wave-uniform constant in scalar registers: ONE v_fma_f64.  Left to itself the compiler selects the
// two-operand v_fmac_f64 and first copies the constant into the destination (two v_mov_b32 per Horner step).
__device__ __forceinline__ double fma_uniform_addend(const double a, const double b, const double c)
    {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
    }

// cos(pi x) and sin(pi x) for x in [0, 1] (the smoothing window): with y = x - 1/2, cos(pi x) = -sin(pi y) and
// sin(pi x) = cos(pi y), |pi y| <= pi/2, Taylor series in z^2 to z^21 / z^20 (truncation < 3e-16); ~25 FMAs instead of
// the ~80 instructions of the general-range library routine.  Coefficients from the table (c = table + SM_SIN).
__device__ __forceinline__ void sincospi_unit_tab(const double *__restrict__ c, const double x, double &sn, double &cs)
    {
    const double z = M_PI * (x - 0.5), z2 = z * z;
    double s = c[0];
#pragma unroll
    for (int i = 1; i < 10; ++i) s = fma_uniform_addend(s, z2, c[i]);
    const double sin_z = z - z * z2 * s;
    double k = c[10];
#pragma unroll
    for (int i = 1; i < 10; ++i) k = fma_uniform_addend(k, z2, c[10 + i]);
    const double cos_z = 1.0 + z2 * k;
    cs = -sin_z;
    sn = cos_z;
    }

template<int LMAX>
__device__ __forceinline__ void smoothing_tab(const QlArgs<LMAX> &a, const double *__restrict__ tab, const double rsq, const double inv_r,
                                              double &f, double &fprime_divr)
    {
    f = 1.0;
    fprime_divr = 0.0;
    if (rsq > a.ronsq)
        {
        double sn, cs;
        sincospi_unit_tab(tab + QlTab<LMAX>::SM_SIN, (rsq * inv_r - a.r_on) * a.inv_width, sn, cs);
        f = 0.5 * (cs + 1.0);
        fprime_divr = -(0.5 * M_PI) * inv_r * a.inv_width * sn;
        }
    }

// ---- units of work of the two pair passes ----------------------------------------------------------------------------------
// A block walks chunks of QL_PPB consecutive central particles; the neighbour-list segments of a chunk, flattened, are cut
// into batches of <= QL_CAP entries: one UNIT, spread one entry per thread (QL_K per thread), so the waves stay full whatever
// the per-particle neighbour counts are.  Every memory trip of a unit — (own particle, list head, count) -> list entries ->
// neighbour positions — is requested one stage ahead and left in flight across the arithmetic of the units before it (the
// block barriers are LDS-only: lds_barrier).  The last wave of the block prepares the unit tables in LDS (QlFeed): prefix of
// the counts, own positions, and the owner of every entry slot (a byte per slot, filled by the owner's lane: one LDS read
// per entry instead of a six-step search).  The tables rotate in a ring; what is in flight when is told at the two kernels.
constexpr int QL_PPB = 64, QL_CAP = 1024, QL_K = QL_CAP / QL_THREADS;
static_assert(QL_PPB == MTD_WAVE, "one wave scans the neighbour counts of a chunk");
static_assert(QL_K == 4 && QL_PPB <= 256, "owner bytes of a thread's entries packed in one register");

struct QlUnit
    {
    unsigned int start[QL_PPB];          // head_list of the particle
    unsigned int off[QL_PPB + 1];        // exclusive prefix of the neighbour counts inside the chunk
    double px[QL_PPB], py[QL_PPB], pz[QL_PPB];   // the central particles' own positions
    unsigned char owner[QL_CAP];         // owner[t]: central particle (index in the chunk) of list entry base + t
    unsigned int chunk, base, total, valid;
    };

template<typename S4> __device__ __forceinline__ S4 zero_s4();
template<> __device__ __forceinline__ float4 zero_s4<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template<> __device__ __forceinline__ double4 zero_s4<double4>() { return make_double4(0.0, 0.0, 0.0, 0.0); }

// registers and logic of the wave that prepares the units (all 64 lanes: lane p = central particle p of the chunk)
template<typename S4, int LMAX> struct QlFeed
    {
    const QlArgs<LMAX> &a;
    const S4 *__restrict__ postype;
    const unsigned int *__restrict__ head_list;
    const unsigned int *__restrict__ n_neigh;
    unsigned int n_chunks, lane;
    unsigned int n_work;                  // blocks that walk the chunks (gridDim.x, minus the passenger blocks of a carrying launch)
    S4 pos;                               // requested ahead: the next chunk's particle, list head and count
    unsigned int start, cnt;
    unsigned int pend_kind, pend_chunk;   // the next unit to publish: 0 none, 1 first batch of chunk pend_chunk, 2 next batch of the same chunk

    __device__ __forceinline__ QlFeed(const QlArgs<LMAX> &a_, const S4 *p_, const unsigned int *h_, const unsigned int *n_)
        : a(a_), postype(p_), head_list(h_), n_neigh(n_), n_chunks((a_.N + QL_PPB - 1) / QL_PPB), lane(threadIdx.x & 63u),
          n_work(gridDim.x), pos(zero_s4<S4>()), start(0), cnt(0), pend_kind(0), pend_chunk(0)
        {
        }
    __device__ __forceinline__ void request(const unsigned int chunk)
        {
        const unsigned int i = chunk * QL_PPB + lane;
        pos = zero_s4<S4>();
        start = cnt = 0;
        if (i < a.N)
            {
            pos = postype[i];
            start = head_list[i];
            cnt = n_neigh[i];
            }
        }
    // the owner bytes of the unit's slots that belong to this lane's particle (entries [lo, hi) of the chunk)
    __device__ __forceinline__ void fill_owner(QlUnit &u, const unsigned int base, unsigned int lo, unsigned int hi)
        {
        if (lo < base) lo = base;
        if (hi > base + QL_CAP) hi = base + QL_CAP;
#pragma clang loop vectorize(disable) unroll(disable)
        for (unsigned int e = lo; e < hi; ++e) u.owner[e - base] = (unsigned char)lane;
        }
    // first batch of a chunk from the registers requested earlier; returns the chunk's number of list entries (wave-uniform)
    __device__ __forceinline__ unsigned int publish_chunk(QlUnit &u, const unsigned int chunk)
        {
        const unsigned int i = chunk * QL_PPB + lane;
        const Particle pi = scalar4_traits<S4>::unpack(pos);
        const unsigned int c = (i < a.N && (unsigned int)pi.type == a.type) ? cnt : 0u;             // :105
        u.px[lane] = pi.x;
        u.py[lane] = pi.y;
        u.pz[lane] = pi.z;
        u.start[lane] = start;
        unsigned int incl = c;
#pragma unroll
        for (int d = 1; d < MTD_WAVE; d <<= 1)
            {
            const unsigned int up = __shfl_up(incl, d, MTD_WAVE);
            if (lane >= (unsigned int)d) incl += up;
            }
        u.off[lane + 1] = incl;
        fill_owner(u, 0, incl - c, incl);
        const unsigned int total = (unsigned int)wave_read((int)incl, 63);
        if (lane == 0)
            {
            u.off[0] = 0;
            u.chunk = chunk;
            u.base = 0;
            u.total = total;
            u.valid = 1;
            }
        return total;
        }
    // what follows the unit (chunk, base, total): the next batch of the same chunk, the block's next chunk (its registers are
    // requested here), or nothing
    __device__ __forceinline__ void plan_after(const unsigned int chunk, const unsigned int base, const unsigned int total)
        {
        if (base + QL_CAP < total)
            {
            pend_kind = 2;
            return;
            }
        pend_chunk = chunk + n_work;
        pend_kind = pend_chunk < n_chunks ? 1 : 0;
        if (pend_kind) request(pend_chunk);
        }
    // publish the planned unit into `u` (`prev` is the unit before it) and plan the one after it
    __device__ __forceinline__ void publish_planned(QlUnit &u, const QlUnit &prev)
        {
        if (pend_kind == 1)
            {
            const unsigned int total = publish_chunk(u, pend_chunk);
            plan_after(pend_chunk, 0, total);
            }
        else if (pend_kind == 2)
            {
            const unsigned int lo = prev.off[lane], hi = prev.off[lane + 1];
            u.px[lane] = prev.px[lane];
            u.py[lane] = prev.py[lane];
            u.pz[lane] = prev.pz[lane];
            u.start[lane] = prev.start[lane];
            u.off[lane + 1] = hi;
            const unsigned int chunk = prev.chunk, base = prev.base + QL_CAP, total = prev.total;
            fill_owner(u, base, lo, hi);
            if (lane == 0)
                {
                u.off[0] = 0;
                u.chunk = chunk;
                u.base = base;
                u.total = total;
                u.valid = 1;
                }
            plan_after(chunk, base, total);
            }
        else if (lane == 0)
            u.valid = 0;
        }
    // the block's first units, one after the other (each waits for its own loads); the unit after them is planned
    template<int N_FIRST> __device__ __forceinline__ void begin(QlUnit *ring)
        {
        if (blockIdx.x < n_chunks)
            {
            request(blockIdx.x);
            const unsigned int total = publish_chunk(ring[0], blockIdx.x);
            plan_after(blockIdx.x, 0, total);
            }
        else if (lane == 0)
            ring[0].valid = 0;
#pragma unroll
        for (int n = 1; n < N_FIRST; ++n) publish_planned(ring[n], ring[n - 1]);
        }
    };

// every thread: the list entries t = k * QL_THREADS + tid of a unit are requested (owner bytes of the thread's entries packed)
__device__ __forceinline__ void ql_request_entries(const QlUnit &u, const unsigned int *__restrict__ nlist, unsigned int (&j)[QL_K],
                                                   unsigned int &owners)
    {
    owners = 0;
#pragma unroll
    for (int k = 0; k < QL_K; ++k) j[k] = 0;
    if (!u.valid) return;
    const unsigned int base = u.base, n = min(u.total - base, (unsigned int)QL_CAP);
#pragma unroll
    for (int k = 0; k < QL_K; ++k)
        {
        const unsigned int t = k * QL_THREADS + threadIdx.x;
        if (t < n)
            {
            const unsigned int p = u.owner[t];
            j[k] = nlist[u.start[p] + (base + t - u.off[p])];
            owners |= p << (8 * k);
            }
        }
    }

// ---- CV accumulation -------------------------------------------------------------------------------
// Q'_lm = sum over pairs of f Y_lm, in the monic amplitudes of the table (QlTab): a thread keeps R_lm += f p_m,l-m(cos theta) h^m
// (h = sin(theta) e^{i phi} = (dx + i dy) / r) for every (l, m >= 0) in registers — 7 real + 21 complex sums at lmax = 6 — and
// the normalisation |nrm(l, m)| is applied once per block at the end (the Condon-Shortley phase in the finalize step).  One
// role per wave up to lmax = 8 (every wave visits different pairs); above that the orders m are split between two roles
// (even / odd waves visit the same pairs with half of the sums each), or the sums would not fit the register file.
// The pipeline (ring of four unit tables), iteration n:
//   compaction   the list entries of unit n + 1 (requested two iterations ago) are filtered — SYM: a symmetric full list
//                without ghost particles ((i, j) listed <=> (j, i) listed) is visited from the lower index only:
//                Y_lm(-d) = (-1)^l Y_lm(d), so the two visits of the reference add up to twice the even degrees and cancel
//                in the odd ones, which is exactly the scaling the finalize step applies to half lists (SteinhardtQl.cc:173-179)
//                — and the survivors written densely to LDS, every wave its own segment in entry order (ballot + popcount)
//   arithmetic   of unit n on the dense pairs, one per thread and round: the next pair's neighbour position is requested first;
//                the last round requests the first pair of unit n + 1.  Pairs beyond the cut-off (a list built with a
//                buffer) or of another type are dropped here
//   last wave    publishes unit n + 3 from registers requested an iteration ago, requests unit n + 4's
//   every thread requests the list entries of unit n + 3
__host__ __device__ constexpr bool ql_role_has(const int m, const int role, const int n_roles)
    {
    return n_roles == 1 || ((m % 4 == 0 || m % 4 == 3) ? 0 : 1) == role;
    }

struct QlDense                                     // the pairs of a unit that are visited, densely
    {
    unsigned int j[QL_CAP];
    unsigned char p[QL_CAP];
    unsigned int kept[QL_THREADS / MTD_WAVE];      // per gathering wave: its segment [w * QL_CAP / 4, ...) holds this many
    };

struct QlAccPipe
    {
    QlUnit su[4];
    QlDense dense[2];
    };

constexpr int QL_RED = 16, QL_RED_STRIDE = 65;     // the final sums over the lanes, 16 values at a time, rows padded against bank conflicts
constexpr size_t QL_ACC_LDS = sizeof(QlAccPipe) > (QL_THREADS / MTD_WAVE) * (QL_RED * QL_RED_STRIDE + 64) * sizeof(double)
                                  ? sizeof(QlAccPipe)
                                  : (QL_THREADS / MTD_WAVE) * (QL_RED * QL_RED_STRIDE + 64) * sizeof(double);

__device__ __forceinline__ void ql_compact(const QlUnit &u, const unsigned int (&j)[QL_K], const unsigned int owners, const bool sym, QlDense &d)
    {
    constexpr unsigned int SEG = QL_CAP / (QL_THREADS / MTD_WAVE);
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned int n = u.valid ? min(u.total - u.base, (unsigned int)QL_CAP) : 0u;
    const unsigned int first = u.chunk * QL_PPB;
    unsigned int cnt = 0;                                                        // wave-uniform
#pragma unroll
    for (int k = 0; k < QL_K; ++k)
        {
        const unsigned int t = k * QL_THREADS + threadIdx.x;
        const unsigned int p = (owners >> (8 * k)) & 255u;
        const bool keep = t < n && !(sym && j[k] < first + p);
        const unsigned long long mask = __ballot(keep);
        if (keep)
            {
            const unsigned int slot = wave * SEG + cnt + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
            d.j[slot] = j[k];
            d.p[slot] = (unsigned char)p;
            }
        cnt += (unsigned int)__popcll(mask);
        }
    if (lane == 0) d.kept[wave] = cnt;
    }

// flat index of a dense pair -> its slot (the segments of the four gathering waves one after the other)
struct QlDenseMap
    {
    unsigned int c0, c1, c2, total;
    __device__ __forceinline__ explicit QlDenseMap(const QlDense &d)
        {
        c0 = d.kept[0];
        c1 = c0 + d.kept[1];
        c2 = c1 + d.kept[2];
        total = c2 + d.kept[3];
        }
    __device__ __forceinline__ unsigned int slot(const unsigned int fi) const
        {
        constexpr unsigned int SEG = QL_CAP / (QL_THREADS / MTD_WAVE);
        // segment s starts at flat index c_{s-1} and at slot s * SEG: every segment boundary passed adds its unused tail
        return fi + (fi >= c0 ? SEG - c0 : 0u) + (fi >= c1 ? SEG - (c1 - c0) : 0u) + (fi >= c2 ? SEG - (c2 - c1) : 0u);
        }
    };

// the whole loop of one role with its own sums: for two roles the instantiations sit in the two arms of a wave-uniform branch,
// so their registers overlap (one shared array would keep every sum live in both); both arms run the same sequence of block barriers
template<typename S4, int LMAX, int ROLE, int NROLES, bool SYM, bool EVEN>
__device__ __forceinline__ void ql_accumulate_role(const QlArgs<LMAX> &a, const S4 *__restrict__ postype,
                                                   const unsigned int *__restrict__ head_list, const unsigned int *__restrict__ n_neigh,
                                                   const unsigned int *__restrict__ nlist, const double *__restrict__ tab, QlAccPipe &pipe,
                                                   double *red /* this wave's scratch, aliases the pipe */, double *s_row)
    {
    typedef QlTab<LMAX> T;
    constexpr int NLM = (LMAX + 1) * (LMAX + 2) / 2;
    constexpr unsigned int TPR = QL_THREADS / NROLES;                          // threads of a role
    const unsigned int tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned int r_tid = (wave / NROLES) * MTD_WAVE + lane;              // this thread among those of its role
    const bool setup_wave = wave == QL_THREADS / MTD_WAVE - 1;
    QlFeed<S4, LMAX> feed(a, postype, head_list, n_neigh);
    cplx Q[LMAX + 1][LMAX + 1];                                                // [m][l]
#pragma unroll
    for (int m = 0; m <= LMAX; ++m)
#pragma unroll
        for (int l = 0; l <= LMAX; ++l) Q[m][l] = {0.0, 0.0};

    // prologue: units 0..2 published, unit 3 planned; entries of units 0..2 requested, unit 0 compacted, its first pair requested
    if (setup_wave) feed.template begin<3>(pipe.su);
    lds_barrier();
    unsigned int j_a[QL_K], j_b[QL_K], own_a, own_b;
        {
        unsigned int j_0[QL_K], own_0;
        ql_request_entries(pipe.su[0], nlist, j_0, own_0);
        ql_request_entries(pipe.su[1], nlist, j_a, own_a);
        ql_request_entries(pipe.su[2], nlist, j_b, own_b);
        ql_compact(pipe.su[0], j_0, own_0, SYM, pipe.dense[0]);
        }
    lds_barrier();
    S4 pos_raw = zero_s4<S4>();
    unsigned int p_cur = 0;
        {
        const QlDenseMap map(pipe.dense[0]);
        if (r_tid < map.total)
            {
            const unsigned int s = map.slot(r_tid);
            p_cur = pipe.dense[0].p[s];
            pos_raw = postype[pipe.dense[0].j[s]];
            }
        }

    for (unsigned int n = 0;; ++n)
        {
        const QlUnit &U = pipe.su[n % 4];
        if (!U.valid) break;                                                      // published before the last barrier
        const QlDense &D = pipe.dense[n & 1];
        QlDense &DN = pipe.dense[(n + 1) & 1];
        ql_compact(pipe.su[(n + 1) % 4], j_a, own_a, SYM, DN);                    // (an invalid unit compacts to nothing)
        lds_barrier();
        const QlDenseMap map(D), map_next(DN);
        const unsigned int rounds = (map.total + TPR - 1) / TPR;
        if (rounds == 0 && r_tid < map_next.total)                                // nothing to visit: only the hand-over to the next unit
            {
            const unsigned int s = map_next.slot(r_tid);
            p_cur = DN.p[s];
            pos_raw = postype[DN.j[s]];
            }
#pragma unroll 1
        for (unsigned int it = 0; it < rounds; ++it)
            {
            const unsigned int fi = it * TPR + r_tid;
            unsigned int tab_shift = 0;
            asm volatile("" : "+s"(tab_shift));                 // a zero the compiler cannot see through: the table loads stay in the loop
            const double *__restrict__ tab_k = tab + tab_shift;
            // the next pair's neighbour position is requested before this pair's arithmetic
            const bool last = it + 1 == rounds;
            const unsigned int fn = last ? r_tid : fi + TPR;
            const QlDense &DA = last ? DN : D;
            const bool want = fn < (last ? map_next.total : map.total);
            S4 pos_ahead = zero_s4<S4>();
            unsigned int p_ahead = 0;
            if (want)
                {
                const unsigned int s = last ? map_next.slot(fn) : map.slot(fn);
                p_ahead = DA.p[s];
                pos_ahead = postype[DA.j[s]];
                }
            // a slot beyond the unit's pairs, a neighbour of another type or beyond the cut-off takes part with weight zero (on a
            // harmless separation): no divergent branch around the 49 running sums
                {
                const Particle pj = scalar4_traits<S4>::unpack(pos_raw);
                double dx = U.px[p_cur] - pj.x, dy = U.py[p_cur] - pj.y, dz = U.pz[p_cur] - pj.z;
                min_image(a, dx, dy, dz);
                double rsq = dx * dx + dy * dy + dz * dz;
                const bool visit = fi < map.total && (unsigned int)pj.type == a.type && rsq <= a.rcutsq;     // :126, :141
                if (!visit)
                    {
                    dx = dy = 0.0;
                    dz = rsq = 1.0;
                    }
                const double inv_r = rsqrt(rsq);
                const double ct = dz * inv_r, ex = dx * inv_r, ey = dy * inv_r;
                double f, fprime_divr;
                smoothing_tab<LMAX>(a, tab_k, rsq, inv_r, f, fprime_divr);
                if (!visit) f = 0.0;
                cplx fh = {f, 0.0};                                              // f h^m
#pragma unroll
                for (int m = 0; m <= LMAX; ++m)
                    {
                    if (ql_role_has(m, ROLE, NROLES))
                        {
                        double pm2 = 1.0, pm1 = ct;                              // p_m,n-2 and p_m,n-1
#pragma unroll
                        for (int nn = 0; m + nn <= LMAX; ++nn)
                            {
                            const int l = m + nn;
                            const bool wanted = !(EVEN && (l % 2));              // odd degrees of a half / symmetric list: never read
                            if (nn == 0)
                                {
                                if (wanted) Q[m][l].re += fh.re;
                                if (wanted && m > 0) Q[m][l].im += fh.im;
                                }
                            else
                                {
                                double pn = ct;
                                if (nn >= 2)
                                    {
                                    pn = ct * pm1 - tab_k[T::beta(m, nn)] * pm2;
                                    pm2 = pm1;
                                    pm1 = pn;
                                    }
                                if (wanted) Q[m][l].re += pn * fh.re;
                                if (wanted && m > 0) Q[m][l].im += pn * fh.im;
                                }
                            }
                        }
                    if (m < LMAX) fh = m == 0 ? cplx{f * ex, f * ey} : cmul(fh, {ex, ey});
                    }
                }
            pos_raw = pos_ahead;
            p_cur = p_ahead;
            }
        if (setup_wave) feed.publish_planned(pipe.su[(n + 3) % 4], pipe.su[(n + 2) % 4]);
        lds_barrier();                             // unit n + 3's table is in LDS; dense[n & 1] and unit n's table are free
#pragma unroll
        for (int k = 0; k < QL_K; ++k) j_a[k] = j_b[k];
        own_a = own_b;
        ql_request_entries(pipe.su[(n + 3) % 4], nlist, j_b, own_b);
        }

    // the sums over the lanes of this wave, QL_RED values at a time through LDS (the pipe's memory: every wave has left the
    // loop): lane-major rows, 16 lanes add a row's quarter each in a fixed order, lanes 0..15 the four quarters.  A wave owns
    // the (l, m) of its role; the other entries of its row stay zero.
    lds_barrier();
    double *part = red + QL_RED * QL_RED_STRIDE;
#pragma unroll
    for (int r = 0; r * QL_RED < 2 * NLM; ++r)
        {
#pragma unroll
        for (int v = 0; v < QL_RED; ++v)
            {
            const int q = r * QL_RED + v;                        // slot in the row: 2 * (l (l + 1) / 2 + m) + (0: re, 1: im)
            if (q < 2 * NLM)
                {
                int l = 0;
                while ((l + 1) * (l + 2) / 2 <= q / 2) ++l;
                const int m = q / 2 - l * (l + 1) / 2;
                double val = 0.0;
                if (ql_role_has(m, ROLE, NROLES) && !(EVEN && (l % 2))) val = (q & 1) ? (m > 0 ? Q[m][l].im : 0.0) : Q[m][l].re;
                red[v * QL_RED_STRIDE + lane] = val;
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned int v = lane & (QL_RED - 1), quarter = lane / QL_RED;
        double sum = 0.0;
        if (r * QL_RED + (int)v < 2 * NLM)
            {
#pragma unroll
            for (int i = 0; i < MTD_WAVE / 4; ++i) sum += red[v * QL_RED_STRIDE + quarter * (MTD_WAVE / 4) + i];
            }
        part[quarter * QL_RED + v] = sum;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < QL_RED && r * QL_RED + (int)lane < 2 * NLM)
            s_row[r * QL_RED + lane] = (part[lane] + part[QL_RED + lane]) + (part[2 * QL_RED + lane] + part[3 * QL_RED + lane]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }

template<int LMAX> constexpr int ql_acc_roles() { return LMAX <= 8 ? 1 : 2; }

// EVEN: the list is a half list or a symmetric full one — the finalize step sets the odd degrees to zero (:173-179), so their
// sums are not formed (16 of the 28 (l, m) at lmax = 6: half the registers, and room for a third or fourth wave per SIMD)
#ifdef MTD_QL_ACC_OCC          // (experiment builds: csrc/Makefile EXTRA_HIPFLAGS)
template<int LMAX, bool EVEN> constexpr int ql_acc_blocks_per_cu() { return MTD_QL_ACC_OCC; }
#else
template<int LMAX, bool EVEN> constexpr int ql_acc_blocks_per_cu() { return LMAX <= 6 ? (EVEN ? 3 : 2) : (LMAX <= 8 && EVEN ? 2 : 1); }
#endif

template<typename S4, int LMAX, bool SYM, bool EVEN>
__global__ __launch_bounds__(QL_THREADS, (ql_acc_blocks_per_cu<LMAX, EVEN>())) void k_ql_accumulate(const QlArgs<LMAX> a, const S4 *__restrict__ postype,
                                                                 const unsigned int *__restrict__ head_list,
                                                                 const unsigned int *__restrict__ n_neigh,
                                                                 const unsigned int *__restrict__ nlist, double *__restrict__ partials,
                                                                 const double *__restrict__ tab)
    {
    constexpr int NLM = (LMAX + 1) * (LMAX + 2) / 2;
    constexpr int NROLES = ql_acc_roles<LMAX>();
    __shared__ double s_wave[QL_THREADS / MTD_WAVE][2 * NLM];
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[QL_ACC_LDS];
    QlAccPipe &pipe = *reinterpret_cast<QlAccPipe *>(s_raw);
    const int wave = threadIdx.x >> 6;
    double *red = reinterpret_cast<double *>(s_raw) + (size_t)wave * (QL_RED * QL_RED_STRIDE + 64);
    if (NROLES == 1)
        ql_accumulate_role<S4, LMAX, 0, 1, SYM, EVEN>(a, postype, head_list, n_neigh, nlist, tab, pipe, red, s_wave[wave]);
    else if (wave & 1)
        ql_accumulate_role<S4, LMAX, 1, 2, SYM, EVEN>(a, postype, head_list, n_neigh, nlist, tab, pipe, red, s_wave[wave]);
    else
        ql_accumulate_role<S4, LMAX, 0, 2, SYM, EVEN>(a, postype, head_list, n_neigh, nlist, tab, pipe, red, s_wave[wave]);
    __syncthreads();
    const unsigned int n_out = (a.lmax + 1) * (a.lmax + 2);     // 2 * n_lm of the RUNTIME lmax (same (l,m) order)
    for (unsigned int q = threadIdx.x; q < n_out; q += blockDim.x)
        {
        double v = 0.0;
        for (int w = 0; w < QL_THREADS / MTD_WAVE; ++w) v += s_wave[w][q];
        partials[(size_t)blockIdx.x * n_out + q] = v * fabs(tab[QlTab<LMAX>::NRM + q / 2]);     // the monic amplitudes' normalisation
        }
    }

// ---- finalize: full Q_lm table (reference order), Q_l, CV --------------------------------------------
// the bias-grid engine's deferred pass riding in the finalize launch (metad.hip: take_pending_apply): blocks 1 .. n run
// apply_cells on 256 grid cells each; kept apart from the plain kernel so that a step without a passenger carries no extra
// kernel arguments
template<int LMAX>
__global__ __launch_bounds__(256) void k_ql_finalize(const QlArgs<LMAX> a, const double *__restrict__ qprime,
                                                     double *__restrict__ qlm_full, double *__restrict__ ql, double *__restrict__ value);

template<int LMAX>
__global__ __launch_bounds__(256) void k_ql_finalize_carrier(const QlArgs<LMAX> a, const double *__restrict__ qprime,
                                                             double *__restrict__ qlm_full, double *__restrict__ ql,
                                                             double *__restrict__ value, const mtd::MetadCfg cfg);

template<int LMAX>
__device__ __forceinline__ void ql_finalize_body(const QlArgs<LMAX> &a, const double *__restrict__ qprime,
                                                 double *__restrict__ qlm_full, double *__restrict__ ql, double *__restrict__ value,
                                                 const bool write = true, double *s_value = nullptr)
    {
    // one thread per entry of the full table (<= 169 at lmax = 12), the sums over m and l in the serial order of the reference
    __shared__ double s_sq[(LMAX + 1) * (LMAX + 1)];
    __shared__ double s_ql[LMAX + 1];
    const double ng = (double)a.n_global;
    const unsigned int n = threadIdx.x;
    const unsigned int count = (a.lmax + 1) * (a.lmax + 1);
    if (n < count)
        {
        int l = 0;
        while ((l + 1) * (l + 1) <= (int)n) ++l;
        const int p = (int)n - l * l;
        const int m = (p <= l) ? p : (l - p);
        const int am = m < 0 ? -m : m;
        const int idx = l * (l + 1) / 2 + am;
        cplx q = {qprime[2 * idx], qprime[2 * idx + 1]};
        if (m < 0) q = cconj(q);                                            // fsph negative m: conjugate, phase +1
        if (m > 0 && (m % 2)) q = cscale(q, -1.0);                          // Condon-Shortley (:150)
        if (a.half_nlist)                                                   // :173-179
            {
            if (l % 2 == 0)
                q = cscale(q, 2.0);
            else
                q = {0.0, 0.0};
            }
        if (write)
            {
            qlm_full[2 * n] = q.re;
            qlm_full[2 * n + 1] = q.im;
            }
        double sq = q.re * q.re + q.im * q.im;
        sq *= (4.0 * M_PI / (2 * l + 1)) / (ng * ng);                       // nc = 1 (:182)
        s_sq[n] = sq;
        }
    __syncthreads();
    if (threadIdx.x <= a.lmax)
        {
        const int l = threadIdx.x;
        double Ql = 0.0;
        for (int p = 0; p < 2 * l + 1; ++p) Ql += s_sq[l * l + p];
        if (write) ql[l] = Ql;
        s_ql[l] = Ql;
        }
    __syncthreads();
    if (threadIdx.x == 0)
        {
        double val = 0.0;
        for (int l = 0; l <= (int)a.lmax; ++l) val += a.ql_ref[l] * s_ql[l];   // :190-194
        if (write) *value = val;
        if (s_value) *s_value = val;
        }
    }

template<int LMAX>
__global__ __launch_bounds__(256) void k_ql_finalize(const QlArgs<LMAX> a, const double *__restrict__ qprime,
                                                     double *__restrict__ qlm_full, double *__restrict__ ql, double *__restrict__ value)
    {
    ql_finalize_body<LMAX>(a, qprime, qlm_full, ql, value);
    }

template<int LMAX>
__global__ __launch_bounds__(256) void k_ql_finalize_carrier(const QlArgs<LMAX> a, const double *__restrict__ qprime,
                                                             double *__restrict__ qlm_full, double *__restrict__ ql,
                                                             double *__restrict__ value, const mtd::MetadCfg cfg)
    {
    if (blockIdx.x > 0)
        {
        __shared__ double s_red[16];
        const unsigned int c0 = (blockIdx.x - 1) * 256;
        mtd::apply_cells(cfg, c0, min(cfg.len, c0 + 256u), blockIdx.x == 1, s_red);
        return;
        }
    ql_finalize_body<LMAX>(a, qprime, qlm_full, ql, value);
    }

// ---- finalize + the bias-grid engine's launch in ONE (cv.steinhardt as the only variable of the grid) ---------------
// Between the two pair passes of a step sat three launches that do next to nothing but wait for memory: the reduction of the
// block sums (4.7 us), the finalize step (5.2 us: a few hundred flops) and the grid engine's launch (7.0 us: its scalar chain).
// Here the finalize step IS the head of the grid engine's launch: every block of it (the grid's first-pass blocks, at least one)
// forms Q_lm, Q_l and the value from the reduced sums — the same arithmetic in every block, so all of them hold the same bits —,
// hands the value to the engine's scalar chain in registers (chain_wave's `given`: no trip through memory), and goes on with
// the first grid pass of a deposit (updateGrid :1002-1047, updateHistogram :1092-1119, updateSigmaGrid :1122-1155, first loop of
// updateReweightedEstimator :1070-1075); block 0 writes the tables the force pass reads and publishes the step's scalars.  The
// grid-pass and publishing code (metad_device.hpp: grid_first_pass_256, publish_step) is the twin of k_fused_force's (fused.hip): tests
// hold the two against each other bit for bit
// (tests/test_gpu_steinhardt.py::test_ql_merged_launch_matches_separate_launches).  The engine's DEFERRED pass of the previous
// deposit cannot ride here (this launch's chain reads the grid it writes): it travels in the force pass of its own step
// (k_ql_forces<..., CARRY>), which follows the deposit's launch directly.
template<int LMAX>
__global__ __launch_bounds__(256) void k_ql_finalize_chain(const QlArgs<LMAX> a, const double *__restrict__ qprime,
                                                           double *__restrict__ qlm_full, double *__restrict__ ql,
                                                           double *__restrict__ value, const mtd::MetadCfg c, const int deposit,
                                                           const unsigned int n_grid_blocks)
    {
    using namespace mtd;
    __shared__ double s_given[3];
    __shared__ ChainResult s_chain;
    __shared__ double s_red[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ChainPre<1> pre;
    if (wave == 0) chain_preload<1, false>(c, pre);                  // the grid patch around the last value: in flight during the finalize step
    if (threadIdx.x < 3) s_given[threadIdx.x] = 0.0;
    ql_finalize_body<LMAX>(a, qprime, qlm_full, ql, value, blockIdx.x == 0, &s_given[0]);
    __syncthreads();
    if (wave == 0)
        {
        const ChainResult r = chain_wave(c, deposit != 0, true, nullptr, s_given, false, &pre.patch, pre.patch_ok != 0);
        if (lane == 0) chain_share(s_chain, r);
        }
    __syncthreads();
    if (blockIdx.x < n_grid_blocks) grid_first_pass_256(c, s_chain, blockIdx.x, s_red);
    if (blockIdx.x == 0 && wave == 0) publish_step(c, s_chain, deposit, s_given);
    }

// ---- half lists: order-independent (exact) sums of the pair forces ---------------------------------------------
// With a half list a pair is visited once and its reaction force goes to the OTHER particle (SteinhardtQl.cc:328-333): many
// blocks add to one particle in an order that changes from run to run.  Floating-point atomics made the half-list forces the
// one result of this library that was not bitwise reproducible.  Now every contribution is split exactly into integers —
// QL_BINS 64-bit accumulators per component, bin k holding the bits of weight 2^(QL_BIN_LOW + 48 k) .. 2^(QL_BIN_LOW + 48 (k + 1))
// — and added with integer atomics: the sums are exact, so they cannot depend on the order, and a last pass converts them to
// the caller's force array (one rounding per component).  Range: |x| < 2^72 per contribution, bits below 2^-120 are dropped
// (forces of order 1: 1e-36 relative); 15 bits of headroom per bin = 32768 contributions at the top of a bin.  Non-finite
// contributions (theta = 0, Q_l = 0: the reference produces inf / NaN there too) set flags that the conversion turns back into
// what IEEE addition in any order would have given.
constexpr int QL_BINS = 4;
constexpr int QL_BIN_BITS = 48;
constexpr int QL_BIN_LOW = -120;
constexpr size_t QL_ACC_WORDS = 3 * QL_BINS + 1;                 // per particle: 3 components x QL_BINS bins + one flag word

__device__ __forceinline__ void ql_exact_add(unsigned long long *acc /* this particle's words */, const int comp, double x)
    {
    if (!(fabs(x) < 4722366482869645213696.0))                     // 2^72: inf, NaN or out of range
        {
        const unsigned int bit = x != x ? 4u : (x > 0.0 ? 1u : 2u);  // NaN, +overflow, -overflow
        atomicOr((unsigned int *)(acc + 3 * QL_BINS), bit << (4 * comp));
        return;
        }
#pragma unroll
    for (int k = QL_BINS - 1; k >= 0; --k)
        {
        const int w = QL_BIN_LOW + QL_BIN_BITS * k;
        const double t = trunc(ldexp(x, -w));                       // |t| < 2^48 (k = top: |x| < 2^72), exact
        if (t != 0.0)
            {
            atomicAdd(acc + comp * QL_BINS + k, (unsigned long long)(long long)t);
            x -= ldexp(t, w);                                        // exact
            }
        }
    }

template<typename S4>
__global__ __launch_bounds__(256) void k_ql_exact_to_force(const unsigned long long *__restrict__ acc, const double *__restrict__ own, const unsigned int N,
                                                           S4 *__restrict__ force)
    {
    typedef typename scalar4_traits<S4>::scalar scalar;
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const unsigned long long *a = acc + (size_t)i * QL_ACC_WORDS;
    const unsigned int flags = (unsigned int)a[3 * QL_BINS];
    double f[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < QL_BINS; ++k) v += ldexp((double)(long long)a[c * QL_BINS + k], QL_BIN_LOW + QL_BIN_BITS * k);   // small to large
        v += own[3 * (size_t)i + c];                                   // the particle's own pairs (inf / NaN arrive through it as they are)
        const unsigned int fl = (flags >> (4 * c)) & 7u;
        if (fl & 4u) v = nan("");
        if (fl & 1u) v += INFINITY;                                     // (-inf + inf = NaN, like the additions themselves)
        if (fl & 2u) v -= INFINITY;
        f[c] = v;
        }
    force[i] = scalar4_traits<S4>::make((scalar)f[0], (scalar)f[1], (scalar)f[2], (scalar)0);
    }

// ---- forces ----------------------------------------------------------------------------------------------
// Round 3: a pipeline over the block's batches of pairs instead of load-everything / compute-everything phases.  The counters
// of round 3's first half (profiles/r3: 486 vector instructions per list entry, VALU busy 56 % of the launch, every wave
// waiting half of its cycles) said the three dependent memory trips of a chunk — (own particle, list head, count) -> list
// entries -> neighbour positions — were paid in full per chunk with four waves per SIMD to hide them.  Now every thread keeps
// the SAME list entries in both roles (it fetches the neighbour of entry t and computes the pair of entry t), so the
// separations no longer pass through LDS, and every trip is requested one stage ahead and left in flight across the
// arithmetic (the block barriers are LDS-only: lds_barrier):
//   unit n      = one batch of <= QL_CAP list entries of one chunk of QL_PPB central particles
//   arithmetic  of unit n, entry k of the thread: the neighbour position of entry k + 1 is requested first; the last entry
//               requests the first neighbour of unit n + 1 (its list entries are in registers since the end of unit n - 1)
//   wave 3      (the wave with the fewest entries in a partly filled batch) publishes unit n + 2's prefix table and own
//               positions to LDS from registers loaded during unit n - 1's arithmetic, then requests unit n + 3's
//   after the barrier: per-particle sums of unit n's pair forces (four threads per particle, fixed order), the list entries of
//               unit n + 2 are requested
// Three unit tables rotate in LDS (unit n in use, n + 1 referenced by the requests, n + 2 being written).
// EXACT (half lists): reaction forces into the exact accumulators, the particle's own sum into own[i] — the conversion pass
// adds it; !EXACT: the floating-point atomics of round 1 (mtd_ql_set_half_list_exact(0): 2.5 x faster, sums in arrival order).
// the pair force of one list entry (SteinhardtQl.cc:287-333 contracted as the header describes), in the monic amplitudes:
// with h = sin(theta) e^{i phi} = (dx + i dy) / r, P = p_m,l-m(cos theta) and Z_lm = h^m q_lm (q_lm = nrm(l, m) w_l conj(Q_lm)
// from LDS),
//   U = sum P Re Z,   V = cot(theta) sum m P Re Z + sin(theta) sum_{m < l} d(l, m) p_m+1,l-m-1 Re Z,   W = -sum m P Im Z
// and the force is -(f'/r U d + f/r V e_theta + f/rho W e_phi).  Degrees with Ql_ref[l] = 0 are skipped as a whole (one scalar
// branch per degree); per order m the sums over l are kept apart (U_m, W_m) so that the factor m is applied once.
// On the z axis 1/rho is infinite and the force comes out NaN, as from the reference's 0 * (1 / tan(0)).
template<int LMAX>
__device__ __forceinline__ void ql_pair_force(const QlArgs<LMAX> &a, const double *__restrict__ tab, const double *s_qw, const unsigned int act,
                                              const unsigned int opaque0, const double dx, const double dy, const double dz, const double rsq,
                                              double &fpx, double &fpy, double &fpz)
    {
    typedef QlTab<LMAX> T;
    const double inv_r = rsqrt(rsq);
    const double rho2 = dx * dx + dy * dy;
    const double inv_rho = rsqrt(rho2);
    const double ct = dz * inv_r, ex = dx * inv_r, ey = dy * inv_r;
    double f, fprime_divr;
    smoothing_tab<LMAX>(a, tab, rsq, inv_r, f, fprime_divr);
    // monic amplitudes p[m][n], n = l - m (n = 0: 1, n = 1: cos theta)
    double p[LMAX + 1][LMAX + 1];
#pragma unroll
    for (int m = 0; m <= LMAX; ++m)
        {
        p[m][0] = 1.0;
        if (m + 1 <= LMAX) p[m][1] = ct;
#pragma unroll
        for (int n = 2; m + n <= LMAX; ++n) p[m][n] = ct * p[m][n - 1] - tab[T::beta(m, n)] * p[m][n - 2];
        }
    // h^m
    cplx h[LMAX + 1];
    h[0] = {1.0, 0.0};
    if (LMAX >= 1) h[1] = {ex, ey};
#pragma unroll
    for (int m = 2; m <= LMAX; ++m) h[m] = cmul(h[m - 1], {ex, ey});
    double Um[LMAX + 1], Wm[LMAX + 1], VB = 0.0;
#pragma unroll
    for (int m = 0; m <= LMAX; ++m) Um[m] = Wm[m] = 0.0;
#pragma unroll
    for (int l = 0; l <= LMAX; ++l)
        {
        if (act & (1u << l))                                     // degrees with Ql_ref[l] != 0 (and l <= lmax)
            {
#pragma unroll
            for (int m = 0; m <= l; ++m)
                {
                const int idx = l * (l + 1) / 2 + m;
                // `opaque0` (always 0, but derived from the pair slot) keeps these reads inside the pair loop: hoisted, the
                // loop-invariant table takes ~110 registers
                const cplx q = {s_qw[2 * idx + opaque0], s_qw[2 * idx + 1 + opaque0]};
                const cplx Z = m == 0 ? q : cmul(h[m], q);
                if (m == l)
                    Um[m] += Z.re;
                else
                    Um[m] += p[m][l - m] * Z.re;
                if (m > 0)
                    {
                    if (m == l)
                        Wm[m] += Z.im;
                    else
                        Wm[m] += p[m][l - m] * Z.im;
                    }
                if (m < l)
                    {
                    if (l - m - 1 == 0)
                        VB += tab[T::d(l, m)] * Z.re;
                    else
                        VB += (tab[T::d(l, m)] * p[m + 1][l - m - 1]) * Z.re;
                    }
                }
            }
        }
    double U = Um[0], VA = 0.0, W = 0.0;
#pragma unroll
    for (int m = 1; m <= LMAX; ++m)
        {
        U += Um[m];
        VA += (double)m * Um[m];
        W -= (double)m * Wm[m];
        }
    const double st = rho2 * inv_rho * inv_r, cot = dz * inv_rho;      // sin(theta) = rho / r, 1 / tan(theta)
    const double cp = dx * inv_rho, sp = dy * inv_rho;
    const double V = cot * VA + st * VB;
    const double fa = fprime_divr * U, fb = f * inv_r * V, fc = f * inv_rho * W;   // 1/(r sin theta) = 1/rho
    fpx = -(fa * dx + fb * (ct * cp) - fc * sp);                                  // e_theta = (ct cp, ct sp, -st), e_phi = (-sp, cp, 0)  (:288)
    fpy = -(fa * dy + fb * (ct * sp) + fc * cp);
    fpz = -(fa * dz - fb * st);
    }

// CARRY: the launch carries the bias-grid engine's deferred pass (second reweighting pass + accumulate of the deposit that has
// just been made, metad.hip: take_pending_apply) in its LAST n_apply blocks — the working blocks are fewer than the resident
// capacity by that many, so the passengers start with everybody else and are gone after ~2 us.  NoCarry: an empty struct, the
// plain instantiations keep their argument list and their code.
struct QlNoCarry { };
struct QlCarry { mtd::MetadCfg cfg; unsigned int n_apply; };
template<bool CARRY> struct QlCarryArg { typedef QlNoCarry type; };
template<> struct QlCarryArg<true> { typedef QlCarry type; };

template<typename S4, int LMAX, bool HALF, bool EXACT, bool CARRY = false>
__global__ __launch_bounds__(QL_THREADS, (LMAX <= 6 ? 3 : 2)) void k_ql_forces(const QlArgs<LMAX> a, const S4 *__restrict__ postype,
                                                             const unsigned int *__restrict__ head_list,
                                                             const unsigned int *__restrict__ n_neigh,
                                                             const unsigned int *__restrict__ nlist, const double *__restrict__ qlm_full,
                                                             S4 *__restrict__ force, const double *__restrict__ d_bias, const double bias_host,
                                                             unsigned long long *__restrict__ exact_acc, double *__restrict__ own,
                                                             const double *__restrict__ tab, const typename QlCarryArg<CARRY>::type carry)
    {
    unsigned int n_work = gridDim.x;
    if constexpr (CARRY)
        {
        n_work = gridDim.x - carry.n_apply;
        if (blockIdx.x >= n_work)
            {
            __shared__ double s_red_apply[16];
            const unsigned int c0 = (blockIdx.x - n_work) * 256;
            mtd::apply_cells(carry.cfg, c0, min(carry.cfg.len, c0 + 256u), blockIdx.x == n_work, s_red_apply);
            return;
            }
        }
    typedef typename scalar4_traits<S4>::scalar scalar;
    constexpr int NLM = (LMAX + 1) * (LMAX + 2) / 2;
    constexpr unsigned int PPB = QL_PPB, CAP = QL_CAP;
    constexpr int K = QL_K;                                    // list entries per thread and unit
    static_assert(QL_THREADS == 4 * QL_PPB, "four summing threads per particle");
    __shared__ double s_qw[2 * NLM];                 // w_l (2 or 4) conj(Q_lm), m >= 0, index l(l+1)/2 + m
    __shared__ QlUnit su[3];
    __shared__ double s_fx[CAP], s_fy[CAP], s_fz[CAP];          // pair forces of the unit, by list entry
    const unsigned int tid = threadIdx.x;
    const bool setup_wave = (tid >> 6) == QL_THREADS / MTD_WAVE - 1;
    const double bias = d_bias ? *d_bias : bias_host;
    const double ng = (double)a.n_global;
    unsigned int active_l = 0;
#pragma unroll
    for (int l = 0; l <= LMAX; ++l)
        if (l <= (int)a.lmax && a.ql_ref[l] != 0.0) active_l |= 1u << l;
    active_l = __builtin_amdgcn_readfirstlane(active_l);          // a scalar: one s_bitcmp + branch per degree in the pair loop
    for (unsigned int q = tid; q < (unsigned int)NLM; q += blockDim.x)
        {
        // (l, m) of the packed index
        int l = 0;
        while ((l + 1) * (l + 2) / 2 <= (int)q) ++l;
        const int m = (int)q - l * (l + 1) / 2;
        double re = 0.0, im = 0.0;
        if (l <= (int)a.lmax)
            {
            // :316-321, times the normalisation of the monic amplitude (ql_pair_force)
            const double w = bias * (4.0 * M_PI / (2 * l + 1)) / (ng * ng) * a.ql_ref[l] * (m > 0 ? 4.0 : 2.0) * tab[QlTab<LMAX>::NRM + q];
            const int n = l * l + m;                                             // reference order: m = 0..l, then -1..-l
            re = w * qlm_full[2 * n];
            im = -w * qlm_full[2 * n + 1];
            }
        s_qw[2 * q] = re;
        s_qw[2 * q + 1] = im;
        }
    // ---- prologue: units 0 and 1 published, unit 2 planned; entries of units 0 and 1 and the first neighbour requested ----
    QlFeed<S4, LMAX> feed(a, postype, head_list, n_neigh);
    feed.n_work = n_work;
    if (setup_wave) feed.template begin<2>(su);
    lds_barrier();
    unsigned int j_cur[K], j_next[K], own_cur, own_next;
    ql_request_entries(su[0], nlist, j_cur, own_cur);
    ql_request_entries(su[1], nlist, j_next, own_next);
    S4 pos_raw = zero_s4<S4>();
    if (su[0].valid && tid < min(su[0].total, CAP)) pos_raw = postype[j_cur[0]];
    double Fx = 0.0, Fy = 0.0, Fz = 0.0;             // thread (p = tid / 4, q = tid % 4): every fourth pair force of particle p

    for (unsigned int n = 0;; ++n)
        {
        const QlUnit &U = su[n % 3];
        if (!U.valid) break;                                                      // published before the last barrier
        const QlUnit &NU = su[(n + 1) % 3];
        const unsigned int base = U.base, total = U.total, chunk = U.chunk;
        const unsigned int n_cur = min(total - base, CAP);
        const unsigned int n_next = NU.valid ? min(NU.total - NU.base, CAP) : 0u;
        unsigned int j0 = j_cur[0], j1 = j_cur[1], j2 = j_cur[2], j3 = j_cur[3], owners = own_cur;
#pragma unroll 1
        for (int k = 0; k < K; ++k)
            {
            const unsigned int t = k * QL_THREADS + tid;
            unsigned int tab_shift = 0;
            asm volatile("" : "+s"(tab_shift));                 // a zero the compiler cannot see through: the table loads stay in the loop
            const double *__restrict__ tab_k = tab + tab_shift;
            // the next neighbour position is requested before this entry's arithmetic
            const bool last = k == K - 1;
            const unsigned int jx = last ? j_next[0] : j1;
            const bool want = last ? tid < n_next : t + QL_THREADS < n_cur;
            S4 pos_ahead = zero_s4<S4>();
            if (want) pos_ahead = postype[jx];
            if (t < n_cur)
                {
                const unsigned int p = owners & 255u;
                const Particle pj = scalar4_traits<S4>::unpack(pos_raw);
                double dx = U.px[p] - pj.x, dy = U.py[p] - pj.y, dz = U.pz[p] - pj.z;
                min_image(a, dx, dy, dz);
                const double rsq = dx * dx + dy * dy + dz * dz;
                double fpx = 0.0, fpy = 0.0, fpz = 0.0;
                if ((unsigned int)pj.type == a.type && rsq <= a.rcutsq)        // :126, :141
                    {
                    ql_pair_force<LMAX>(a, tab_k, s_qw, active_l, t >> 31, dx, dy, dz, rsq, fpx, fpy, fpz);
                    if (HALF)                                                    // :328-333
                        {
                        if (j0 < a.N)
                            {
                            if (EXACT)
                                {
                                unsigned long long *aj = exact_acc + (size_t)j0 * QL_ACC_WORDS;
                                ql_exact_add(aj, 0, -fpx);
                                ql_exact_add(aj, 1, -fpy);
                                ql_exact_add(aj, 2, -fpz);
                                }
                            else
                                {
                                scalar *fj = (scalar *)&force[j0];
                                atomicAdd(fj + 0, (scalar)(-fpx));
                                atomicAdd(fj + 1, (scalar)(-fpy));
                                atomicAdd(fj + 2, (scalar)(-fpz));
                                }
                            }
                        }
                    }
                s_fx[t] = fpx;
                s_fy[t] = fpy;
                s_fz[t] = fpz;
                }
            pos_raw = pos_ahead;
            j0 = j1; j1 = j2; j2 = j3;
            owners >>= 8;
            }
        if (setup_wave) feed.publish_planned(su[(n + 2) % 3], NU);
        lds_barrier();                             // the unit's pair forces and unit n + 2's table are in LDS
        // per-particle sums of this unit's pair forces: thread q of particle p takes entries lo + q, lo + q + 4, ...
            {
            const unsigned int p = tid >> 2, q = tid & 3u;
            const unsigned int lo = U.off[p] > base ? U.off[p] - base : 0u;
            unsigned int hi = U.off[p + 1] > base ? U.off[p + 1] - base : 0u;
            if (hi > n_cur) hi = n_cur;
            for (unsigned int t = lo + q; t < hi; t += 4)
                {
                Fx += s_fx[t];
                Fy += s_fy[t];
                Fz += s_fz[t];
                }
            if (base + CAP >= total)               // the chunk's last batch: (q0 + q1) + (q2 + q3), every lane of the quad gets it
                {
                Fx += dpp_move<MTD_DPP_QUAD_XOR1>(Fx);
                Fy += dpp_move<MTD_DPP_QUAD_XOR1>(Fy);
                Fz += dpp_move<MTD_DPP_QUAD_XOR1>(Fz);
                Fx += dpp_move<MTD_DPP_QUAD_XOR2>(Fx);
                Fy += dpp_move<MTD_DPP_QUAD_XOR2>(Fy);
                Fz += dpp_move<MTD_DPP_QUAD_XOR2>(Fz);
                const unsigned int i = chunk * PPB + p;
                if (q == 0 && i < a.N)
                    {
                    if (HALF && EXACT)
                        {
                        own[3 * (size_t)i + 0] = Fx;
                        own[3 * (size_t)i + 1] = Fy;
                        own[3 * (size_t)i + 2] = Fz;
                        }
                    else if (HALF)
                        {
                        scalar *fi = (scalar *)&force[i];
                        atomicAdd(fi + 0, (scalar)Fx);
                        atomicAdd(fi + 1, (scalar)Fy);
                        atomicAdd(fi + 2, (scalar)Fz);
                        }
                    else
                        nt_store(scalar4_traits<S4>::make((scalar)Fx, (scalar)Fy, (scalar)Fz, (scalar)0), force + i);   // written once, not re-read here
                    }
                Fx = Fy = Fz = 0.0;
                }
            }
        // rotate: unit n + 1's entries become the current ones, unit n + 2's are requested
#pragma unroll
        for (int k = 0; k < K; ++k) j_cur[k] = j_next[k];
        own_cur = own_next;
        ql_request_entries(su[(n + 2) % 3], nlist, j_next, own_next);
        lds_barrier();                             // the sums are done with s_f*: the next unit may write its pair forces
        }
    }

int g_half_exact = 1;        // mtd_ql_set_half_list_exact

template<int LMAX>
int fill_args(QlArgs<LMAX> &a, unsigned int N, const mtd_box *box, double rcut, double ron, unsigned int lmax, unsigned int type,
              const double *ql_ref, unsigned int n_global, int half)
    {
    if (!box || !ql_ref || n_global == 0 || lmax > (unsigned int)LMAX || !(rcut > 0.0) || !(ron >= 0.0) || !(ron < rcut))
        return MTD_ERR_INVALID_ARGUMENT;
    std::memset(&a, 0, sizeof(a));
    for (int i = 0; i < 3; ++i)
        {
        a.lo[i] = box->lo[i];
        a.L[i] = box->L[i];
        }
    a.xy = box->xy; a.xz = box->xz; a.yz = box->yz;
    a.rcutsq = rcut * rcut;                      // SteinhardtQl.cc:18
    a.ronsq = ron * ron;
    a.r_on = std::sqrt(a.ronsq);
    a.r_cut = std::sqrt(a.rcutsq);
    a.inv_width = 1.0 / (a.r_cut - a.r_on);
    for (int i = 0; i < 3; ++i) a.Linv[i] = 1.0 / box->L[i];
    a.lmax = lmax; a.type = type; a.N = N; a.n_global = n_global; a.half_nlist = half;
    for (unsigned int l = 0; l <= lmax; ++l) a.ql_ref[l] = ql_ref[l];
    return MTD_SUCCESS;
    }

unsigned int ql_blocks(unsigned int N, int ppb, unsigned int resident)
    {
    unsigned int b = (N + ppb - 1) / ppb;
    if (b < 1) b = 1;
    if (b > resident) b = resident;            // every block resident at once: 4 per CU in the CV pass, 2 in the force pass
    return b;
    }

// the recurrence index in f0/f1 is the DEGREE n = l - m in the reference's tables (index2d(lmax, m, l-1) with l the
// degree counter of compute_jacobis), which is what ylm_table uses (a.f0[m][n]).

// the finalize launch, with the bias-grid engine's deferred pass as a passenger when one is waiting on this stream
template<int LMAX>
void launch_finalize(const QlArgs<LMAX> &a, const double *d_qprime, double *d_qlm, double *d_ql, double *d_value, hipStream_t s)
    {
    mtd::MetadCfg cfg;
    if (mtd_metad *engine = mtd::take_pending_apply(s, cfg))
        {
        k_ql_finalize_carrier<LMAX><<<1 + (cfg.len + 255) / 256, 256, 0, s>>>(a, d_qprime, d_qlm, d_ql, d_value, cfg);
        if (hipPeekAtLastError() == hipSuccess) mtd::commit_pending_apply(engine);      // a failed launch leaves the pass pending
        }
    else
        k_ql_finalize<LMAX><<<1, 256, 0, s>>>(a, d_qprime, d_qlm, d_ql, d_value);
    }

// blocks of `kernel` that are resident at once on the current device (both pair passes launch exactly that many, or fewer if the
// system is small: every block then walks its chunks in a pipeline); cached per (kernel, device)
template<typename Kernel> unsigned int ql_resident_blocks(Kernel kernel)
    {
    static std::mutex mtx;
    static std::map<std::pair<const void *, int>, unsigned int> resident;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mtx);
    const std::pair<const void *, int> key((const void *)kernel, dev);
    auto it = resident.find(key);
    if (it == resident.end())
        {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, QL_THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        (void)hipGetLastError();
        unsigned int n = (unsigned int)(per_cu * cus);
        if (n > QL_MAX_BLOCKS) n = QL_MAX_BLOCKS;                      // the scratch holds this many rows of partial sums
        it = resident.emplace(key, n).first;
        }
    return it->second;
    }

template<typename S4, int LMAX, bool SYM, bool EVEN>
unsigned int launch_accumulate(const QlArgs<LMAX> &a, const S4 *postype, const unsigned int *d_head, const unsigned int *d_nneigh,
                               const unsigned int *d_nlist, double *d_partials, const double *tab, hipStream_t s)
    {
    const unsigned int blocks = ql_blocks(a.N, QL_PPB, ql_resident_blocks(k_ql_accumulate<S4, LMAX, SYM, EVEN>));
    k_ql_accumulate<S4, LMAX, SYM, EVEN><<<blocks, QL_THREADS, 0, s>>>(a, postype, d_head, d_nneigh, d_nlist, d_partials, tab);
    return blocks;
    }

template<int LMAX>
int accumulate_impl(unsigned int N, const void *d_postype, int dtype, const mtd_box *box, const unsigned int *d_head,
                    const unsigned int *d_nneigh, const unsigned int *d_nlist, int half, double rcut, double ron, unsigned int lmax,
                    unsigned int type, const double *ql_ref, unsigned int n_global, double *d_partials, unsigned int *n_partials,
                    double *d_qprime, double *d_qlm, double *d_ql, double *d_value, bool accumulate, bool finalize, hipStream_t s)
    {
    // half: 0 full list, 1 half list, 2 full list that is symmetric and indexes no ghost particle — the CV pass then visits
    // every pair once and the finalize step scales like for a half list (ql_compact)
    QlArgs<LMAX> a;
    int rc = fill_args<LMAX>(a, N, box, rcut, ron, lmax, type, ql_ref, n_global, half != 0);
    if (rc) return rc;
    if (!accumulate)
        {
        launch_finalize<LMAX>(a, d_qprime, d_qlm, d_ql, d_value, s);
        MTD_LAUNCH_CHECK();
        return MTD_SUCCESS;
        }
    const double *tab = ql_device_table<LMAX>(s, rc);
    if (rc) return rc;
    const unsigned int n_out = (lmax + 1) * (lmax + 2);
    unsigned int blocks = 0;
    if (dtype == MTD_F32)
        blocks = half == 2   ? launch_accumulate<float4, LMAX, true, true>(a, (const float4 *)d_postype, d_head, d_nneigh, d_nlist, d_partials, tab, s)
                 : half == 1 ? launch_accumulate<float4, LMAX, false, true>(a, (const float4 *)d_postype, d_head, d_nneigh, d_nlist, d_partials, tab, s)
                             : launch_accumulate<float4, LMAX, false, false>(a, (const float4 *)d_postype, d_head, d_nneigh, d_nlist, d_partials, tab, s);
    else
        blocks = half == 2   ? launch_accumulate<double4, LMAX, true, true>(a, (const double4 *)d_postype, d_head, d_nneigh, d_nlist, d_partials, tab, s)
                 : half == 1 ? launch_accumulate<double4, LMAX, false, true>(a, (const double4 *)d_postype, d_head, d_nneigh, d_nlist, d_partials, tab, s)
                             : launch_accumulate<double4, LMAX, false, false>(a, (const double4 *)d_postype, d_head, d_nneigh, d_nlist, d_partials, tab, s);
    MTD_LAUNCH_CHECK();
    *n_partials = blocks;
    rc = mtd_reduce_partials(d_partials, blocks, n_out, n_out, 1.0, 0.0, d_qprime, (mtd_stream_t)s);
    if (rc) return rc;
    if (!finalize) return MTD_SUCCESS;
    launch_finalize<LMAX>(a, d_qprime, d_qlm, d_ql, d_value, s);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

template<typename S4, int LMAX, bool HALF, bool EXACT>
void launch_forces(const QlArgs<LMAX> &a, const S4 *postype, const unsigned int *d_head, const unsigned int *d_nneigh, const unsigned int *d_nlist,
                   const double *d_qlm, S4 *force, const double *d_bias, const double bias_host, unsigned long long *acc, double *own, const double *tab,
                   hipStream_t s)
    {
    if constexpr (!HALF && !EXACT)
        {
        QlCarry carry;
        if (mtd_metad *engine = mtd::take_pending_apply(s, carry.cfg))
            {
            carry.n_apply = (carry.cfg.len + 255) / 256;
            const unsigned int cap = ql_resident_blocks(k_ql_forces<S4, LMAX, false, false, true>);
            if (carry.n_apply + 1 <= cap && carry.n_apply <= 64)          // (a grid of at most 16 384 cells: the passengers stay a sliver of the launch)
                {
                const unsigned int blocks = ql_blocks(a.N, QL_PPB, cap - carry.n_apply);
                k_ql_forces<S4, LMAX, false, false, true><<<blocks + carry.n_apply, QL_THREADS, 0, s>>>(a, postype, d_head, d_nneigh, d_nlist, d_qlm, force, d_bias,
                                                                                                     bias_host, acc, own, tab, carry);
                if (hipPeekAtLastError() == hipSuccess) mtd::commit_pending_apply(engine);      // a failed launch leaves the pass pending
                return;
                }
            // (not taken: the pass stays pending, metad_flush runs it as a launch of its own)
            }
        }
    const unsigned int cap = ql_resident_blocks(k_ql_forces<S4, LMAX, HALF, EXACT>);
    const unsigned int blocks = ql_blocks(a.N, QL_PPB, cap);
    k_ql_forces<S4, LMAX, HALF, EXACT><<<blocks, QL_THREADS, 0, s>>>(a, postype, d_head, d_nneigh, d_nlist, d_qlm, force, d_bias, bias_host, acc, own, tab, QlNoCarry());
    }

template<int LMAX>
int forces_impl(unsigned int N, const void *d_postype, void *d_force, int dtype, const mtd_box *box, const unsigned int *d_head,
                const unsigned int *d_nneigh, const unsigned int *d_nlist, int half, double rcut, double ron, unsigned int lmax,
                unsigned int type, const double *ql_ref, unsigned int n_global, const double *d_qlm, const double *d_bias,
                double bias_host, hipStream_t s)
    {
    half = half == 1 ? 1 : 0;                                                // a symmetric full list (2) is a full list here
    QlArgs<LMAX> a;
    int rc = fill_args<LMAX>(a, N, box, rcut, ron, lmax, type, ql_ref, n_global, half);
    if (rc) return rc;
    const double *tab = ql_device_table<LMAX>(s, rc);
    if (rc) return rc;
    // half lists: exact integer accumulators (memset of :236 = clearing them), stream-ordered scratch from the device's pool
    unsigned long long *acc = nullptr;
    double *own = nullptr;
    const bool exact = half && g_half_exact;
    const size_t acc_bytes = sizeof(unsigned long long) * QL_ACC_WORDS * (size_t)N, own_bytes = sizeof(double) * 3 * (size_t)N;
    const size_t s4 = dtype == MTD_F32 ? sizeof(float4) : sizeof(double4);
    if (exact && N)
        {
        // a pool of this library's own that keeps what it has handed out (the device's default pool returns everything to the
        // driver at the next synchronise, and the next call pays a fresh allocation)
        // (one pool per device: a process may drive several GPUs)
        static std::mutex pool_mutex;
        static std::map<int, hipMemPool_t> pools;
        hipMemPool_t pool = nullptr;
        {
        int dev = 0;
        MTD_HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(pool_mutex);
        auto it = pools.find(dev);
        if (it == pools.end())
            {
            hipMemPool_t p = nullptr;
            hipMemPoolProps props;
            std::memset(&props, 0, sizeof(props));
            props.allocType = hipMemAllocationTypePinned;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            if (hipMemPoolCreate(&p, &props) != hipSuccess)
                {
                (void)hipGetLastError();
                p = nullptr;
                }
            else
                {
                unsigned long long keep = ~0ull;
                (void)hipMemPoolSetAttribute(p, hipMemPoolAttrReleaseThreshold, &keep);
                }
            it = pools.emplace(dev, p).first;
            }
        pool = it->second;
        }
        if (pool)
            MTD_HIP_TRY(hipMallocFromPoolAsync((void **)&acc, acc_bytes + own_bytes, pool, s));
        else
            MTD_HIP_TRY(hipMallocAsync((void **)&acc, acc_bytes + own_bytes, s));
        own = (double *)((char *)acc + acc_bytes);
        const hipError_t me = hipMemsetAsync(acc, 0, acc_bytes + own_bytes, s);
        if (me != hipSuccess)
            {
            (void)hipFreeAsync(acc, s);                              // nothing was launched on it
            return (int)me;
            }
        }
    else if (half)
        MTD_HIP_TRY(hipMemsetAsync(d_force, 0, s4 * N, s));               // memset of :236, the pair terms are then added atomically
    if (dtype == MTD_F32)
        {
        if (exact)
            launch_forces<float4, LMAX, true, true>(a, (const float4 *)d_postype, d_head, d_nneigh, d_nlist, d_qlm, (float4 *)d_force, d_bias, bias_host, acc, own, tab, s);
        else if (half)
            launch_forces<float4, LMAX, true, false>(a, (const float4 *)d_postype, d_head, d_nneigh, d_nlist, d_qlm, (float4 *)d_force, d_bias, bias_host, nullptr, nullptr, tab, s);
        else
            launch_forces<float4, LMAX, false, false>(a, (const float4 *)d_postype, d_head, d_nneigh, d_nlist, d_qlm, (float4 *)d_force, d_bias, bias_host, nullptr, nullptr, tab, s);
        }
    else
        {
        if (exact)
            launch_forces<double4, LMAX, true, true>(a, (const double4 *)d_postype, d_head, d_nneigh, d_nlist, d_qlm, (double4 *)d_force, d_bias, bias_host, acc, own, tab, s);
        else if (half)
            launch_forces<double4, LMAX, true, false>(a, (const double4 *)d_postype, d_head, d_nneigh, d_nlist, d_qlm, (double4 *)d_force, d_bias, bias_host, nullptr, nullptr, tab, s);
        else
            launch_forces<double4, LMAX, false, false>(a, (const double4 *)d_postype, d_head, d_nneigh, d_nlist, d_qlm, (double4 *)d_force, d_bias, bias_host, nullptr, nullptr, tab, s);
        }
    hipError_t launch_err = hipGetLastError();
    if (exact && N)
        {
        if (launch_err == hipSuccess)
            {
            if (dtype == MTD_F32)
                k_ql_exact_to_force<float4><<<(N + 255) / 256, 256, 0, s>>>(acc, own, N, (float4 *)d_force);
            else
                k_ql_exact_to_force<double4><<<(N + 255) / 256, 256, 0, s>>>(acc, own, N, (double4 *)d_force);
            launch_err = hipGetLastError();
            }
        const hipError_t free_err = hipFreeAsync(acc, s);
        if (launch_err == hipSuccess) launch_err = free_err;
        }
    return launch_err == hipSuccess ? MTD_SUCCESS : (int)launch_err;
    }

} // namespace

namespace
{
// ---- half list -> the symmetric full list it stands for (mtd_ql_symmetrize_half_list) -------------------------------------
// counts[i] = own entries + the times i is listed by somebody else (integer atomics: the count does not depend on their order)
__global__ void k_sym_count(const unsigned int n, const unsigned int *__restrict__ head, const unsigned int *__restrict__ n_neigh,
                            const unsigned int *__restrict__ nlist, unsigned int *__restrict__ counts, unsigned int *__restrict__ flag)
    {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int h = head[i], c = n_neigh[i];
    unsigned int own = 0;
    for (unsigned int k = 0; k < c; ++k)
        {
        const unsigned int j = nlist[h + k];
        if (j >= n || j == i)
            {
            atomicOr(flag, 1u);                       // a ghost particle (or a self pair): the reaction would be lost (SteinhardtQl.cc:328)
            continue;
            }
        atomicAdd(&counts[j], 1u);
        own++;
        }
    atomicAdd(&counts[i], own);
    }

// exclusive scan of counts[0..n) by ONE block (once per neighbour-list update: 250 tiles at 256 000 particles), total behind it
__global__ __launch_bounds__(1024) void k_sym_scan(const unsigned int n, const unsigned int *__restrict__ counts, unsigned int *__restrict__ head,
                                                   unsigned int *__restrict__ total)
    {
    __shared__ unsigned int s_wave[16];
    __shared__ unsigned int s_carry;
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (unsigned int base = 0; base < n; base += 1024)
        {
        const unsigned int i = base + threadIdx.x;
        const unsigned int v = i < n ? counts[i] : 0u;
        unsigned int incl = v;
#pragma unroll
        for (int d = 1; d < MTD_WAVE; d <<= 1)
            {
            const unsigned int up = __shfl_up(incl, d, MTD_WAVE);
            if ((int)lane >= d) incl += up;
            }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned int before = s_carry;
        for (unsigned int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < n) head[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
        }
    if (threadIdx.x == 0) *total = s_carry;
    }

// every pair into both segments (arrival order: sorted afterwards)
__global__ void k_sym_fill(const unsigned int n, const unsigned int *__restrict__ head, const unsigned int *__restrict__ n_neigh,
                           const unsigned int *__restrict__ nlist, const unsigned int *__restrict__ full_head,
                           unsigned int *__restrict__ cursor, unsigned int *__restrict__ full_nlist, const size_t capacity)
    {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int h = head[i], c = n_neigh[i];
    for (unsigned int k = 0; k < c; ++k)
        {
        const unsigned int j = nlist[h + k];
        if (j >= n || j == i) continue;
        const size_t a = (size_t)full_head[i] + atomicAdd(&cursor[i], 1u), b = (size_t)full_head[j] + atomicAdd(&cursor[j], 1u);
        if (a < capacity) full_nlist[a] = j;
        if (b < capacity) full_nlist[b] = i;
        }
    }

// a particle's partners in ascending order: the list — and with it the order of every per-particle sum of the force pass — no
// longer depends on the order the atomics of the fill were served in
__global__ void k_sym_sort(const unsigned int n, const unsigned int *__restrict__ full_head, const unsigned int *__restrict__ counts,
                           unsigned int *__restrict__ full_nlist, unsigned int *__restrict__ full_n_neigh, const size_t capacity)
    {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int c = counts[i];
    full_n_neigh[i] = c;
    if ((size_t)full_head[i] + c > capacity) return;
    unsigned int *seg = full_nlist + full_head[i];
    for (unsigned int a = 1; a < c; ++a)
        {
        const unsigned int v = seg[a];
        unsigned int b = a;
        while (b > 0 && seg[b - 1] > v)
            {
            seg[b] = seg[b - 1];
            --b;
            }
        seg[b] = v;
        }
    }
} // namespace

namespace
{
// Diagnostic: the spherical harmonics exactly as the pair kernels evaluate them — h = (dx + i dy) / r and cos(theta) = dz / r
// straight from the separation, the monic recurrence with its constants from the device table, h^m by repeated multiplication,
// |nrm(l, m)| at the end (ql_accumulate_role with f = 1) — one thread per direction, written in fsph's order
// (spherical_harmonics.hpp:229-246, full_m: per degree l the orders 0..l, then -1..-l as plain conjugates).
// tests/test_gpu_golden.py holds it against the vectors the reference's own header produced.
template<int LMAX>
__global__ void k_debug_sph(const unsigned int n, const unsigned int lmax, const double *__restrict__ sep, double *__restrict__ out,
                            const double *__restrict__ tab)
    {
    typedef QlTab<LMAX> T;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double dx = sep[3 * t], dy = sep[3 * t + 1], dz = sep[3 * t + 2];
    const double inv_r = rsqrt(dx * dx + dy * dy + dz * dz);
    const double ct = dz * inv_r, ex = dx * inv_r, ey = dy * inv_r;
    double *o = out + (size_t)t * 2 * (lmax + 1) * (lmax + 1);
    cplx h = {1.0, 0.0};
#pragma unroll
    for (int m = 0; m <= LMAX; ++m)
        {
        double pm2 = 1.0, pm1 = ct;
#pragma unroll
        for (int nn = 0; m + nn <= LMAX; ++nn)
            {
            const int l = m + nn;
            double pn = nn == 0 ? 1.0 : ct;
            if (nn >= 2)
                {
                pn = ct * pm1 - tab[T::beta(m, nn)] * pm2;
                pm2 = pm1;
                pm1 = pn;
                }
            if (l <= (int)lmax)
                {
                const double nrm = fabs(tab[T::nrm(l, m)]);
                const double re = nrm * (pn * h.re), im = nrm * (pn * h.im);
                o[2 * (l * l + m)] = re;
                o[2 * (l * l + m) + 1] = im;
                if (m > 0)
                    {
                    o[2 * (l * l + l + m)] = re;
                    o[2 * (l * l + l + m) + 1] = -im;
                    }
                }
            }
        h = cmul(h, {ex, ey});
        }
    }
} // namespace

namespace
{
template<int LMAX>
int finalize_chain_impl(mtd_metad *m, int half, unsigned int lmax, const double *ql_ref, unsigned int n_global, const double *d_qprime,
                               double *d_qlm, double *d_ql, double *d_value, int dep, unsigned int n_grid, hipStream_t s)
    {
    mtd_box box;
    std::memset(&box, 0, sizeof(box));
    box.L[0] = box.L[1] = box.L[2] = 1.0;                               // (geometry is not used by the finalize step)
    QlArgs<LMAX> a;
    int rc = fill_args<LMAX>(a, 0, &box, 1.0, 0.5, lmax, 0, ql_ref, n_global, half != 0);
    if (rc) return rc;
    k_ql_finalize_chain<LMAX><<<n_grid ? n_grid : 1, 256, 0, s>>>(a, d_qprime, d_qlm, d_ql, d_value, m->cfg, dep, n_grid);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

} // namespace

extern "C" {

int mtd_ql_set_half_list_exact(int enable)
    {
    g_half_exact = enable ? 1 : 0;
    return MTD_SUCCESS;
    }

size_t mtd_ql_symmetrize_workspace_uints(unsigned int n_particles)
    {
    return 2 * (size_t)n_particles + 2;                               // counts[n] | cursor[n] | total | flag
    }

int mtd_ql_symmetrize_half_list_ws(unsigned int n_particles, const unsigned int *d_head_list, const unsigned int *d_n_neigh,
                                   const unsigned int *d_nlist, unsigned int *d_full_head, unsigned int *d_full_n_neigh,
                                   unsigned int *d_full_nlist, size_t full_capacity, size_t *n_full_entries, unsigned int *d_workspace,
                                   mtd_stream_t stream)
    {
    if (!n_full_entries || (n_particles && (!d_head_list || !d_n_neigh || !d_full_head || !d_full_n_neigh || !d_workspace)))
        return MTD_ERR_INVALID_ARGUMENT;
    *n_full_entries = 0;
    if (n_particles == 0) return MTD_SUCCESS;
    hipStream_t s = (hipStream_t)stream;
    unsigned int *work = d_workspace;
    unsigned int *counts = work, *cursor = work + n_particles, *total = work + 2 * (size_t)n_particles, *flag = total + 1;
    MTD_HIP_TRY(hipMemsetAsync(work, 0, (2 * (size_t)n_particles + 2) * sizeof(unsigned int), s));
    const unsigned int blocks = (n_particles + 255) / 256;
    unsigned int host[2] = {0, 0};
    k_sym_count<<<blocks, 256, 0, s>>>(n_particles, d_head_list, d_n_neigh, d_nlist, counts, flag);
    k_sym_scan<<<1, 1024, 0, s>>>(n_particles, counts, d_full_head, total);
    MTD_LAUNCH_CHECK();
    // the one synchronisation of the call: the number of entries decides whether the caller's arrays hold the result
    MTD_HIP_TRY(hipMemcpyAsync(host, total, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    MTD_HIP_TRY(hipStreamSynchronize(s));
    *n_full_entries = host[0];
    if (host[1]) return MTD_ERR_UNSUPPORTED;                         // ghost particles in a half list: use the third-law pass (or a full list)
    if (host[0] > full_capacity || (host[0] && !d_full_nlist)) return MTD_ERR_INVALID_ARGUMENT;   // *n_full_entries tells the caller what to provide
    if (host[0])
        {
        k_sym_fill<<<blocks, 256, 0, s>>>(n_particles, d_head_list, d_n_neigh, d_nlist, d_full_head, cursor, d_full_nlist, full_capacity);
        k_sym_sort<<<blocks, 256, 0, s>>>(n_particles, d_full_head, counts, d_full_nlist, d_full_n_neigh, full_capacity);
        MTD_LAUNCH_CHECK();
        }
    else
        MTD_HIP_TRY(hipMemsetAsync(d_full_n_neigh, 0, (size_t)n_particles * sizeof(unsigned int), s));
    return MTD_SUCCESS;                                               // (stream order: consumers on `stream` see the finished list)
    }

int mtd_ql_symmetrize_half_list(unsigned int n_particles, const unsigned int *d_head_list, const unsigned int *d_n_neigh,
                                const unsigned int *d_nlist, unsigned int *d_full_head, unsigned int *d_full_n_neigh,
                                unsigned int *d_full_nlist, size_t full_capacity, size_t *n_full_entries, mtd_stream_t stream)
    {
    if (!n_full_entries) return MTD_ERR_INVALID_ARGUMENT;
    *n_full_entries = 0;
    if (n_particles == 0) return MTD_SUCCESS;
    // the form that owns its workspace for the duration of the call: an allocation, and a second synchronisation before it is
    // released.  A caller that rebuilds lists regularly keeps the workspace itself (mtd_ql_symmetrize_half_list_ws).
    unsigned int *work = nullptr;
    MTD_HIP_TRY(hipMalloc(&work, mtd_ql_symmetrize_workspace_uints(n_particles) * sizeof(unsigned int)));
    int rc = mtd_ql_symmetrize_half_list_ws(n_particles, d_head_list, d_n_neigh, d_nlist, d_full_head, d_full_n_neigh, d_full_nlist,
                                            full_capacity, n_full_entries, work, stream);
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(work);
    return rc ? rc : (int)e;
    }

int mtd_debug_sph_harmonics(unsigned int lmax, unsigned int n, const double *h_separations, double *h_out)
    {
    if (!h_separations || !h_out || n == 0) return MTD_ERR_INVALID_ARGUMENT;
    if (lmax > 12) return MTD_ERR_UNSUPPORTED;
    const size_t n_out = (size_t)n * 2 * (lmax + 1) * (lmax + 1);
    double *d_sep = nullptr, *d_out = nullptr;
    MTD_HIP_TRY(hipMalloc(&d_sep, (size_t)n * 3 * sizeof(double)));
    hipError_t e = hipMalloc(&d_out, n_out * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_sep, h_separations, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        {
        const unsigned int blocks = (n + 63) / 64;
        int rc = MTD_SUCCESS;
        if (lmax <= 4)
            {
            const double *tab = ql_device_table<4>(nullptr, rc);
            if (!rc) k_debug_sph<4><<<blocks, 64>>>(n, lmax, d_sep, d_out, tab);
            }
        else if (lmax <= 6)
            {
            const double *tab = ql_device_table<6>(nullptr, rc);
            if (!rc) k_debug_sph<6><<<blocks, 64>>>(n, lmax, d_sep, d_out, tab);
            }
        else if (lmax <= 8)
            {
            const double *tab = ql_device_table<8>(nullptr, rc);
            if (!rc) k_debug_sph<8><<<blocks, 64>>>(n, lmax, d_sep, d_out, tab);
            }
        else
            {
            const double *tab = ql_device_table<12>(nullptr, rc);
            if (!rc) k_debug_sph<12><<<blocks, 64>>>(n, lmax, d_sep, d_out, tab);
            }
        e = rc ? (hipError_t)rc : hipGetLastError();
        }
    if (e == hipSuccess) e = hipMemcpy(h_out, d_out, n_out * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_sep);
    (void)hipFree(d_out);
    return (int)e;
    }

size_t mtd_ql_scratch_doubles(unsigned int lmax)
    {
    // block partial sums + Q'_lm + full Q_lm table + Q_l + value
    const size_t n_out = (size_t)(lmax + 1) * (lmax + 2);
    return (size_t)QL_MAX_BLOCKS * n_out + n_out + 2 * (size_t)(lmax + 1) * (lmax + 1) + (lmax + 1) + 1;
    }

// layout of the scratch buffer
static void ql_layout(double *scratch, unsigned int lmax, double **partials, double **qprime, double **qlm, double **ql, double **value)
    {
    const size_t n_out = (size_t)(lmax + 1) * (lmax + 2);
    *partials = scratch;
    *qprime = *partials + (size_t)QL_MAX_BLOCKS * n_out;
    *qlm = *qprime + n_out;
    *ql = *qlm + 2 * (size_t)(lmax + 1) * (lmax + 1);
    *value = *ql + (lmax + 1);
    }

static int ql_dispatch(unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box, const unsigned int *d_head_list,
                       const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist, double rcut, double ron,
                       unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int n_global, double *d_scratch,
                       const double **d_value, const double **d_Ql, const double **d_Qlm, bool accumulate, bool finalize,
                       mtd_stream_t stream)
    {
    if (!d_scratch) return MTD_ERR_INVALID_ARGUMENT;
    // (an empty neighbour list may come with a null d_nlist: it is only dereferenced for particles with n_neigh > 0)
    if (accumulate && n_particles && (!d_postype || !d_head_list || !d_n_neigh)) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    if (lmax > 12) return MTD_ERR_UNSUPPORTED;
    double *partials, *qprime, *qlm, *ql, *value;
    ql_layout(d_scratch, lmax, &partials, &qprime, &qlm, &ql, &value);
    unsigned int n_partials = 0;
    hipStream_t s = (hipStream_t)stream;
    int rc;
#define MTD_QL_ACC(LM) accumulate_impl<LM>(n_particles, d_postype, dtype, box, d_head_list, d_n_neigh, d_nlist, half_nlist, rcut, ron, lmax, \
                                           type, Ql_ref, n_global, partials, &n_partials, qprime, qlm, ql, value, accumulate, finalize, s)
    if (lmax <= 4)
        rc = MTD_QL_ACC(4);
    else if (lmax <= 6)
        rc = MTD_QL_ACC(6);
    else if (lmax <= 8)
        rc = MTD_QL_ACC(8);
    else
        rc = MTD_QL_ACC(12);
#undef MTD_QL_ACC
    if (rc) return rc;
    if (d_value) *d_value = value;
    if (d_Ql) *d_Ql = ql;
    if (d_Qlm) *d_Qlm = qlm;
    return MTD_SUCCESS;
    }

int mtd_ql_accumulate(unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box, const unsigned int *d_head_list,
                      const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist, double rcut, double ron,
                      unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int n_global, double *d_scratch,
                      const double **d_value, const double **d_Ql, const double **d_Qlm, mtd_stream_t stream)
    {
    return ql_dispatch(n_particles, d_postype, dtype, box, d_head_list, d_n_neigh, d_nlist, half_nlist, rcut, ron, lmax, type, Ql_ref,
                       n_global, d_scratch, d_value, d_Ql, d_Qlm, true, true, stream);
    }

int mtd_ql_accumulate_local(unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box,
                            const unsigned int *d_head_list, const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist,
                            double rcut, double ron, unsigned int lmax, unsigned int type, unsigned int n_global, double *d_scratch,
                            double **d_sums, unsigned int *n_sums, mtd_stream_t stream)
    {
    if (!d_sums || !n_sums) return MTD_ERR_INVALID_ARGUMENT;
    double zeros[13] = {0};
    int rc = ql_dispatch(n_particles, d_postype, dtype, box, d_head_list, d_n_neigh, d_nlist, half_nlist, rcut, ron, lmax, type, zeros,
                         n_global, d_scratch, nullptr, nullptr, nullptr, true, false, stream);
    if (rc) return rc;
    double *partials, *qprime, *qlm, *ql, *value;
    ql_layout(d_scratch, lmax, &partials, &qprime, &qlm, &ql, &value);
    *d_sums = qprime;
    *n_sums = (lmax + 1) * (lmax + 2);
    return MTD_SUCCESS;
    }

int mtd_ql_finalize(int half_nlist, unsigned int lmax, const double *Ql_ref, unsigned int n_global, double *d_scratch,
                    const double **d_value, const double **d_Ql, const double **d_Qlm, mtd_stream_t stream)
    {
    if (!Ql_ref) return MTD_ERR_INVALID_ARGUMENT;
    // geometry is not used by the finalize step: any valid box / cut-offs satisfy the argument checks
    mtd_box box;
    std::memset(&box, 0, sizeof(box));
    box.L[0] = box.L[1] = box.L[2] = 1.0;
    return ql_dispatch(0, nullptr, MTD_F64, &box, nullptr, nullptr, nullptr, half_nlist, 1.0, 0.5, lmax, 0, Ql_ref, n_global, d_scratch,
                       d_value, d_Ql, d_Qlm, false, true, stream);
    }

int mtd_ql_finalize_update_bias(mtd_metad *m, int half_nlist, unsigned int lmax, const double *Ql_ref, unsigned int n_global,
                                double *d_scratch, unsigned int timestep, const double **d_value, const double **d_Ql,
                                const double **d_Qlm, mtd_stream_t stream)
    {
    if (!m || !Ql_ref || !d_scratch || n_global == 0) return MTD_ERR_INVALID_ARGUMENT;
    if (lmax > 12) return MTD_ERR_UNSUPPORTED;
    if (m->cfg.n_cv != 1 || m->comm) return MTD_ERR_UNSUPPORTED;        // the value feeds a one-variable chain directly
    if (m->h_step_err && *m->h_step_err) return MTD_ERR_COMM_TIMEOUT;
    double *partials, *qprime, *qlm, *ql, *value;
    ql_layout(d_scratch, lmax, &partials, &qprime, &qlm, &ql, &value);
    hipStream_t s = (hipStream_t)stream;
    // the value stays registered as the variable's source: what a later mtd_metad_get_state evaluates w(s) at
    int rc = mtd_metad_set_cv_source(m, 0, value, 1, 1, 0, 1.0, 0.0);
    if (rc) return rc;
    rc = mtd::metad_flush(m, s);                                        // (nothing to do when the force pass carried the deferred pass)
    if (rc) return rc;
    const int dep = (m->add_bias && (timestep % m->stride == 0)) ? 1 : 0;      // IntegratorMetaDynamics.cc:368
    const unsigned int n_grid = dep ? m->cfg.n_gblocks : 0;
    if (lmax <= 4)
        rc = finalize_chain_impl<4>(m, half_nlist, lmax, Ql_ref, n_global, qprime, qlm, ql, value, dep, n_grid, s);
    else if (lmax <= 6)
        rc = finalize_chain_impl<6>(m, half_nlist, lmax, Ql_ref, n_global, qprime, qlm, ql, value, dep, n_grid, s);
    else if (lmax <= 8)
        rc = finalize_chain_impl<8>(m, half_nlist, lmax, Ql_ref, n_global, qprime, qlm, ql, value, dep, n_grid, s);
    else
        rc = finalize_chain_impl<12>(m, half_nlist, lmax, Ql_ref, n_global, qprime, qlm, ql, value, dep, n_grid, s);
    if (rc) return rc;
    m->pending_apply = dep;
    m->w_stale = dep;
    if (dep) mtd::announce_pending_apply(m, s);                         // the force pass of this step takes the deferred pass along
    if (d_value) *d_value = value;
    if (d_Ql) *d_Ql = ql;
    if (d_Qlm) *d_Qlm = qlm;
    return MTD_SUCCESS;
    }

int mtd_ql_forces(unsigned int n_particles, const void *d_postype, void *d_force, int dtype, const mtd_box *box,
                  const unsigned int *d_head_list, const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist,
                  double rcut, double ron, unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int n_global,
                  const double *d_scratch, const double *d_bias, double bias_host, mtd_stream_t stream)
    {
    if (!d_scratch || (n_particles && (!d_postype || !d_force || !d_head_list || !d_n_neigh))) return MTD_ERR_INVALID_ARGUMENT;
    if (dtype != MTD_F32 && dtype != MTD_F64) return MTD_ERR_INVALID_ARGUMENT;
    if (lmax > 12) return MTD_ERR_UNSUPPORTED;
    if (n_particles == 0) return MTD_SUCCESS;
    double *partials, *qprime, *qlm, *ql, *value;
    ql_layout((double *)d_scratch, lmax, &partials, &qprime, &qlm, &ql, &value);
    hipStream_t s = (hipStream_t)stream;
#define MTD_QL_F(LM) forces_impl<LM>(n_particles, d_postype, d_force, dtype, box, d_head_list, d_n_neigh, d_nlist, half_nlist, rcut, ron, lmax, \
                                     type, Ql_ref, n_global, qlm, d_bias, bias_host, s)
    if (lmax <= 4) return MTD_QL_F(4);
    if (lmax <= 6) return MTD_QL_F(6);
    if (lmax <= 8) return MTD_QL_F(8);
    return MTD_QL_F(12);
#undef MTD_QL_F
    }

} // extern "C"
