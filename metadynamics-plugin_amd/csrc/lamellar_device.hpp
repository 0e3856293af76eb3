// lamellar_device.hpp — device code of the lamellar order parameter, shared by the stand-alone
// kernels (lamellar.hip) and the fused bias-step kernels (fused.hip).
//
// Reference arithmetic: LamellarOrderParameter.cc:143-179 (Fourier modes), :77-140 (forces).
// Phase in TURNS: t_k = h g1 + k g2 + l g3 with g_i = b_i' . r formed in double (b_i' = reciprocal
// rows without 2 pi), rounded once to fp32; cos/sin(2 pi t) in fp32.  Loops are mode-outer /
// particle-inner over U particles held in registers, so each mode's constants are fetched once
// per U particles (broadcast LDS reads of the staged mode tables) and the trig pipes see U
// independent chains.
#pragma once

#include "mtd_device.hpp"

namespace mtd
{

struct LamKArgs
    {
    double B[3][3];                           // reciprocal rows without 2*pi
    unsigned int n_cv, n_modes, n_types;
    unsigned int trig;                        // mtd_lamellar_set::trig_mode (host side only: selects the instantiation, lam_fast_trig)
    unsigned int first[MTD_MAX_CV + 1];
    unsigned char slot[MTD_MAX_CV];           // CV c of the set is collective variable slot[c] of the bias grid (fused force pass)
    unsigned int _pad2;
    float4 h[MTD_MAX_MODES];                  // Miller indices (h, k, l, fold): fold = 1 when the mode 2(h,k,l) of the same CV is
                                              // folded into this one by the CV pass, else 0
    unsigned char corder[MTD_MAX_MODES];      // CV pass: the modes of CV c it visits are corder[first[c] .. first[c] + nact[c])
    unsigned char nact[MTD_MAX_CV];           // (second harmonics folded into their fundamental are left out)
    float4 q[MTD_MAX_MODES];                  // Cartesian wave vectors with 2*pi (qx, qy, qz, 0)
    float coeff[MTD_MAX_CV][MTD_MAX_TYPES];
    };

struct ForcePtrs
    {
    void *f[MTD_MAX_CV];
    };

// cos / sin of 2*pi*t.  FAST: hardware v_cos_f32 / v_sin_f32 take the angle in turns, domain [-256, 256] (outside it they
// return 1 / 0: the host only selects FAST for mode sets whose phases stay far inside, lam_fast_trig).  No v_fract in front:
// the hardware reduces the range itself and is MORE accurate on the raw phase (tools/probe_fract.hip: max error 1.25e-7
// against 2.65e-7 with an explicit fract) — and the pair kernels sit on the instruction-issue limit (profiles/r3), where one
// instruction per particle and mode is 6 % of the step.
template<bool FAST> __device__ __forceinline__ float cos2pi(float t)
    {
    if (FAST)
        return __builtin_amdgcn_cosf(t);
    else
        return cospif(2.0f * t);
    }

template<bool FAST> __device__ __forceinline__ float sin2pi(float t)
    {
    if (FAST)
        return __builtin_amdgcn_sinf(t);
    else
        return sinpif(2.0f * t);
    }

__device__ __forceinline__ void project(const LamKArgs &a, const Particle &p, float &g0, float &g1, float &g2)
    {
    g0 = (float)(a.B[0][0] * p.x + a.B[0][1] * p.y + a.B[0][2] * p.z);
    g1 = (float)(a.B[1][0] * p.x + a.B[1][1] * p.y + a.B[1][2] * p.z);
    g2 = (float)(a.B[2][0] * p.x + a.B[2][1] * p.y + a.B[2][2] * p.z);
    }

// s_coeff[MTD_MAX_CV * MTD_MAX_TYPES] <- per-type mode coefficients (call from all threads, then sync)
__device__ __forceinline__ void load_coeff(const LamKArgs &a, float *s_coeff)
    {
    for (unsigned int i = threadIdx.x; i < MTD_MAX_CV * MTD_MAX_TYPES; i += blockDim.x)
        s_coeff[i] = a.coeff[i / MTD_MAX_TYPES][i % MTD_MAX_TYPES];
    }

// Mode tables staged in LDS: the inner loops read them with wave-uniform (broadcast) ds_read_b128,
// which the compiler can issue several modes ahead; reading them from the kernel-argument segment
// costs an s_load + s_waitcnt lgkmcnt(0) stall every other mode (seen in the gfx950 ISA).
struct ModeTables
    {
    float4 h[MTD_MAX_MODES];
    float4 q[MTD_MAX_MODES];
    };

__device__ __forceinline__ void load_modes(const LamKArgs &a, ModeTables &t, const bool with_q)
    {
    for (unsigned int k = threadIdx.x; k < a.n_modes; k += blockDim.x)
        {
        t.h[k] = a.h[k];
        if (with_q) t.q[k] = a.q[k];
        }
    }

// CV pass: the visited modes of every CV, gathered into a dense list at [first[c], first[c] + nact[c])
__device__ __forceinline__ void load_modes_cv(const LamKArgs &a, ModeTables &t)
    {
    for (unsigned int k = threadIdx.x; k < a.n_modes; k += blockDim.x) t.h[k] = a.h[a.corder[k]];
    }

// The same two tables for a launch whose LamKArgs the host has made DENSE (dense_cv_args: h[k] is already the k-th visited mode,
// corder the identity): ONE unconditional load per thread from a clamped index, to be requested right behind the first particles —
// stage_cv_tables_request — and stored just in front of the barrier — stage_cv_tables_store.  The loops above compile to three
// dependent memory round trips in front of a kernel's first particle (a batched loop over blockDim for the coefficients, the index
// array, the modes through it): ~1.5 us of launch A's 7, +6 us for a kernel that had a barrier of its own in front (mesh.hip).
struct CvTableRegs
    {
    float4 h;
    float c;
    };
__device__ __forceinline__ CvTableRegs stage_cv_tables_request(const LamKArgs &a)
    {
    CvTableRegs r;
    r.h = a.h[min(threadIdx.x, (unsigned int)MTD_MAX_MODES - 1)];
    r.c = (&a.coeff[0][0])[min(threadIdx.x, (unsigned int)(MTD_MAX_CV * MTD_MAX_TYPES) - 1)];
    return r;
    }
__device__ __forceinline__ void stage_cv_tables_store(const CvTableRegs &r, float *s_coeff, ModeTables &t)   // blockDim >= 128
    {
    if (threadIdx.x < MTD_MAX_MODES) t.h[threadIdx.x] = r.h;
    if (threadIdx.x < MTD_MAX_CV * MTD_MAX_TYPES) s_coeff[threadIdx.x] = r.c;
    }
// host: the argument block with the visited modes of the CV pass as a dense list (what load_modes_cv gathers on the device)
inline LamKArgs dense_cv_args(const LamKArgs &k)
    {
    LamKArgs d = k;
    for (unsigned int q = 0; q < k.n_modes && q < MTD_MAX_MODES; ++q)
        {
        d.h[q] = k.h[k.corder[q]];
        d.corder[q] = (unsigned char)q;
        }
    return d;
    }

// acc[c] += sum over this thread's particles of a_c(type_j) sum_k cos(q_k . r_j)
// thread `tid` of `n_threads` walks particles tid, tid + n_threads, ... in groups of U (even).
// Two things keep this pass off the instruction-issue limit it otherwise sits on (M = 16 modes: 36 issue cycles per
// particle-mode, ~3.7 us per 10^6 particles):
//  * particles are handled two at a time in <2 x float> lanes, so the phase and the sums are v_pk_mul / v_pk_fma / v_pk_add
//    (one instruction per two particles; v_fract / v_cos have no packed form);
//  * a mode whose Miller indices are exactly twice those of another mode of the same CV (second harmonics, half of a typical
//    lamellar mode set) costs one packed fma instead of a phase + fract + cos: cos 2x = 2 cos^2 x - 1.
typedef float v2f __attribute__((ext_vector_type(2)));

// raw particle records of one group (U particles of this thread), loaded without being looked at: the loads can be in
// flight while something else happens (mode tables being staged, the previous group being summed)
template<typename S4, int U> struct RawGroup
    {
    S4 v[U];
    };

template<typename S4, int U>
__device__ __forceinline__ void lam_load_group(const S4 *__restrict__ postype, const unsigned int N, const unsigned int base,
                                               const unsigned int n_threads, RawGroup<S4, U> &g)
    {
    if (N == 0) return;                       // nothing to read (postype may be NULL): the caller's loop does not run either
#pragma unroll
    for (int u = 0; u < U; ++u)
        {
        const unsigned int i = base + u * n_threads;
        g.v[u] = postype[i < N ? i : N - 1];  // out-of-range slots re-read the last particle and are masked by ok[]
        }
    }

// The same without the look at N (N >= 1: the caller knows there is a particle, or hands in any readable address with N = 1):
// the loads sit in the caller's basic block, so the scheduler is free to keep them in flight — behind lam_load_group's
// `if (N == 0) return` the compiler packed the loaded values into pairs INSIDE the branch and waited for every load right there.
template<typename S4, int U>
__device__ __forceinline__ void lam_load_group_nc(const S4 *__restrict__ postype, const unsigned int N, const unsigned int base,
                                                  const unsigned int n_threads, RawGroup<S4, U> &g)
    {
#pragma unroll
    for (int u = 0; u < U; ++u)
        {
        const unsigned int i = base + u * n_threads;
        g.v[u] = postype[i < N ? i : N - 1];
        }
    }

// `first` holds the group at base = tid (lam_load_group, issued by the caller before it staged the tables); every further
// group is requested before the current one is summed, so only the very first memory round trip of the launch is exposed
// (measured: 2.6 us per exposed round trip, more than summing a group)
// one group's terms: acc[c] += sum over the U particles of `cur` (at base, base + n_threads, ...) of a_c(type_j) sum_k cos(q_k . r_j)
template<typename S4, int NCV, bool FAST, int U>
__device__ __forceinline__ void lam_cv_group(const LamKArgs &a, const unsigned int N, const unsigned int base, const unsigned int n_threads,
                                             const float *s_coeff, const ModeTables &mt, const RawGroup<S4, U> &cur, float (&acc)[NCV])
    {
    static_assert(U % 2 == 0, "particles are processed in pairs");
    constexpr int P = U / 2;
    v2f g0[P], g1[P], g2[P];
    int type[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        {
        ok[u] = base + u * n_threads < N;
        const Particle p = scalar4_traits<S4>::unpack(cur.v[u]);
        float x0, x1, x2;
        project(a, p, x0, x1, x2);
        g0[u / 2][u % 2] = x0;
        g1[u / 2][u % 2] = x1;
        g2[u / 2][u % 2] = x2;
        type[u] = p.type;
        }
#pragma unroll
    for (int c = 0; c < NCV; ++c)
        {
        if (c < (int)a.n_cv)
            {
            v2f sum[P];
#pragma unroll
            for (int q = 0; q < P; ++q) sum[q] = (v2f)(0.0f);
            const unsigned int k1 = a.first[c] + a.nact[c];
#pragma unroll 4
            for (unsigned int k = a.first[c]; k < k1; ++k)
                {
                const float4 h = mt.h[k];                         // staged by load_modes_cv: dense, fold flag in w
#pragma unroll
                for (int q = 0; q < P; ++q)
                    {
                    const v2f t = h.x * g0[q] + h.y * g1[q] + h.z * g2[q];
                    v2f cs;
                    cs.x = cos2pi<FAST>(t.x);
                    cs.y = cos2pi<FAST>(t.y);
                    sum[q] += cs;
                    // the second harmonic of this mode when folded (w = 1), branch-free.  The hardware cosine of the FAST
                    // path truncates: |c| is low by 3.2e-8 on average, which cancels in sums of c but leaves 2 c^2 - 1 low by
                    // 6.5e-8 on average — a bias of that size times sum_j a_j / N in the CV, whatever N (tools/probe_cos.hip;
                    // found by tools/fuzz_fused.py on a one-type system).  One float ulp off the constant takes 92 % of it out.
                    sum[q] += h.w * ((cs * cs) * 2.0f - (FAST ? 0.99999994f : 1.0f));
                    }
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
                {
                const float w = ok[u] ? s_coeff[c * MTD_MAX_TYPES + type[u]] : 0.0f;
                acc[c] += w * sum[u / 2][u % 2];
                }
            }
        }
    }

template<typename S4, int NCV, bool FAST, int U>
__device__ __forceinline__ void lam_cv_accumulate(const LamKArgs &a, const S4 *__restrict__ postype, const unsigned int N,
                                                  const unsigned int tid, const unsigned int n_threads,
                                                  const float *s_coeff, const ModeTables &mt, RawGroup<S4, U> cur,
                                                  float (&acc)[NCV])
    {
    for (unsigned int base = tid; base < N; base += U * n_threads)
        {
        const unsigned int next = base + U * n_threads;
        // UNCONDITIONAL (lam_load_group clamps every index to the last particle; the loop runs only if there is one): behind
        // `if (next < N)` the compiler could not count the loads and put an s_waitcnt vmcnt(0) into the projection of the CURRENT
        // group — the next group's round trip was paid in full before the first cosine (gfx950 ISA), not overlapped
        RawGroup<S4, U> nxt;
        lam_load_group_nc<S4, U>(postype, N, next < N ? next : N - 1, n_threads, nxt);
        lam_cv_group<S4, NCV, FAST, U>(a, N, base, n_threads, s_coeff, mt, cur, acc);
        cur = nxt;
        }
    }

// Block-level tail of the CV pass: fp32 wave sums -> fp64 across waves (fixed order) ->
// partials[block_id * NCV + c].  s_wave: [blockDim/64][NCV] doubles.
template<int NCV>
__device__ __forceinline__ void lam_cv_block_reduce(const float (&acc)[NCV], double *s_wave, double *__restrict__ partials,
                                                    const unsigned int block_id)
    {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n_waves = blockDim.x >> 6;
#pragma unroll
    for (int c = 0; c < NCV; ++c)
        {
        const float v = wave_sum(acc[c]);
        if (lane == 0) s_wave[wave * NCV + c] = (double)v;
        }
    __syncthreads();
    if (threadIdx.x < NCV)
        {
        double r = 0.0;
        for (int w = 0; w < n_waves; ++w) r += s_wave[w * NCV + threadIdx.x];
        partials[block_id * NCV + threadIdx.x] = r;
        }
    }

// Forces of every fused CV for this thread's particles.  s_wcoef[c*MTD_MAX_TYPES + type] must hold
// a_c(type) * bias_c * 2 / N_global (LamellarOrderParameter.cc:120-133 folded into one factor).
template<typename S4, bool FAST, int U>
__device__ __forceinline__ void lam_force_pass(const LamKArgs &a, const S4 *__restrict__ postype, const ForcePtrs &out,
                                               const unsigned int N, const unsigned int tid, const unsigned int n_threads,
                                               const float *s_wcoef, const ModeTables &mt)
    {
    typedef typename scalar4_traits<S4>::scalar scalar;
    for (unsigned int base = tid; base < N; base += U * n_threads)
        {
        float g0[U], g1[U], g2[U];
        int type[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            {
            const unsigned int i = base + u * n_threads;
            ok[u] = i < N;
            const Particle p = scalar4_traits<S4>::load(postype, ok[u] ? i : base);
            project(a, p, g0[u], g1[u], g2[u]);
            type[u] = p.type;
            }
        for (unsigned int c = 0; c < a.n_cv; ++c)
            {
            float fx[U], fy[U], fz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) fx[u] = fy[u] = fz[u] = 0.0f;
            const unsigned int k1 = a.first[c + 1];
#pragma unroll 4
            for (unsigned int k = a.first[c]; k < k1; ++k)
                {
                const float4 h = mt.h[k];
                const float4 q = mt.q[k];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    {
                    const float s = sin2pi<FAST>(h.x * g0[u] + h.y * g1[u] + h.z * g2[u]);
                    fx[u] += q.x * s;
                    fy[u] += q.y * s;
                    fz[u] += q.z * s;
                    }
                }
            S4 *f = (S4 *)out.f[c];
#pragma unroll
            for (int u = 0; u < U; ++u)
                {
                if (ok[u])
                    {
                    const float w = s_wcoef[c * MTD_MAX_TYPES + type[u]];
                    nt_store(scalar4_traits<S4>::make((scalar)(fx[u] * w), (scalar)(fy[u] * w), (scalar)(fz[u] * w), (scalar)0),
                             &f[base + u * n_threads]);
                    }
                }
            }
        }
    }

// ------------------------------------------------------------------------------------------------
// Split form of the force pass for the fused kernel: the trig sums do not depend on the bias factor,
// so a thread first forms the UNSCALED forces  a-free sum_k q_k sin(q_k . r_j)  of its U particles in
// registers (lam_force_unscaled) and scales + stores them once the bias is known (lam_force_store).
template<int NCV, int U> struct ForceRegs
    {
    v2f f[U / 2][NCV][3];          // particles 2 p and 2 p + 1 of the group in the two halves
    int type[U];
    bool ok[U];
    };

template<typename S4, int U>
__device__ __forceinline__ void lam_force_request(const S4 *__restrict__ postype, const unsigned int N, const unsigned int first,
                                                  const unsigned int stride, RawGroup<S4, U> &raw);
template<typename S4, int NCV, bool FAST, int U>
__device__ __forceinline__ void lam_force_unscaled_from(const LamKArgs &a, const unsigned int N, const unsigned int first,
                                                        const unsigned int stride, const ModeTables &mt, const RawGroup<S4, U> &raw,
                                                        ForceRegs<NCV, U> &R);

// Two particles per <2 x float> lane like the CV pass: the phase and the three force components are v_pk_mul / v_pk_fma (one
// instruction per two particles; v_fract / v_sin have no packed form) — the compiler packed some of the scalar form on its own
// and paid a v_mov per pair for it (16 modes x 4 particles: 173 issue slots, now 128).
template<typename S4, int NCV, bool FAST, int U>
__device__ __forceinline__ void lam_force_unscaled(const LamKArgs &a, const S4 *__restrict__ postype, const unsigned int N,
                                                   const unsigned int first, const unsigned int stride, const ModeTables &mt,
                                                   ForceRegs<NCV, U> &R)
    {
    RawGroup<S4, U> raw;
    lam_force_request<S4, U>(postype, N, first, stride, raw);
    lam_force_unscaled_from<S4, NCV, FAST, U>(a, N, first, stride, mt, raw, R);
    }

// the particle records of a group requested without being looked at (launch B asks for them before the barrier that publishes
// the mode tables: the loads need no table, and the first memory round trip of a launch is its longest)
template<typename S4, int U>
__device__ __forceinline__ void lam_force_request(const S4 *__restrict__ postype, const unsigned int N, const unsigned int first,
                                                  const unsigned int stride, RawGroup<S4, U> &raw)
    {
    if (N == 0) return;
#pragma unroll
    for (int u = 0; u < U; ++u)
        {
        const unsigned int i = first + u * stride;
        raw.v[u] = postype[i < N ? i : N - 1];
        }
    }

template<typename S4, int NCV, bool FAST, int U>
__device__ __forceinline__ void lam_force_unscaled_from(const LamKArgs &a, const unsigned int N, const unsigned int first,
                                                        const unsigned int stride, const ModeTables &mt, const RawGroup<S4, U> &raw,
                                                        ForceRegs<NCV, U> &R)
    {
    static_assert(U % 2 == 0, "particles are processed in pairs");
    constexpr int P = U / 2;
    v2f g0[P], g1[P], g2[P];
#pragma unroll
    for (int u = 0; u < U; ++u)
        {
        const unsigned int i = first + u * stride;
        R.ok[u] = i < N;
        const Particle p = scalar4_traits<S4>::unpack(raw.v[u]);
        float x0, x1, x2;
        project(a, p, x0, x1, x2);
        g0[u / 2][u % 2] = x0;
        g1[u / 2][u % 2] = x1;
        g2[u / 2][u % 2] = x2;
        R.type[u] = p.type;
        }
#pragma unroll
    for (int c = 0; c < NCV; ++c)
        {
#pragma unroll
        for (int pp = 0; pp < P; ++pp) R.f[pp][c][0] = R.f[pp][c][1] = R.f[pp][c][2] = (v2f)(0.0f);
        if (c < (int)a.n_cv)
            {
            const unsigned int k1 = a.first[c + 1];
#pragma unroll 4
            for (unsigned int k = a.first[c]; k < k1; ++k)
                {
                const float4 h = mt.h[k];
                const float4 q = mt.q[k];
#pragma unroll
                for (int pp = 0; pp < P; ++pp)
                    {
                    const v2f t = h.x * g0[pp] + h.y * g1[pp] + h.z * g2[pp];
                    v2f sn;
                    sn.x = sin2pi<FAST>(t.x);
                    sn.y = sin2pi<FAST>(t.y);
                    R.f[pp][c][0] += q.x * sn;
                    R.f[pp][c][1] += q.y * sn;
                    R.f[pp][c][2] += q.z * sn;
                    }
                }
            }
        }
    }

template<typename S4, int NCV, int U>
__device__ __forceinline__ void lam_force_store(const LamKArgs &a, const ForcePtrs &out, const unsigned int first,
                                                const unsigned int stride, const float *s_wcoef, const ForceRegs<NCV, U> &R)
    {
    typedef typename scalar4_traits<S4>::scalar scalar;
#pragma unroll
    for (int c = 0; c < NCV; ++c)
        {
        if (c < (int)a.n_cv)
            {
            S4 *f = (S4 *)out.f[c];
#pragma unroll
            for (int u = 0; u < U; ++u)
                {
                if (R.ok[u])
                    {
                    const float w = s_wcoef[c * MTD_MAX_TYPES + R.type[u]];
                    const S4 v = scalar4_traits<S4>::make((scalar)(R.f[u / 2][c][0][u % 2] * w), (scalar)(R.f[u / 2][c][1][u % 2] * w),
                                                          (scalar)(R.f[u / 2][c][2][u % 2] * w), (scalar)0);
                    nt_store(v, &f[first + u * stride]);
                    }
                }
            }
        }
    }

} // namespace mtd
