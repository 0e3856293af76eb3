// comm_device.hpp — device side of the xGMI mailbox (comm.hip): all-reduce of a few doubles between the GPUs of one
// node without a collective library call on the critical path.
//
// The reference reduces its Fourier-mode sums with a host-staged MPI_Allreduce (LamellarOrderParameterGPU.cc:69-77:
// D2H copy, MPI, host sum).  On MI355X the exchanged payload of the lamellar step is n_cv doubles; a library
// all-reduce costs a stream hop and a kernel of its own (measured ~8 us with one rank, before any link latency)
// against ~22 us for the whole step.  Here every rank owns a small mailbox in its HBM that all peers have mapped
// (hipIpcOpenMemHandle); a sender stores its values straight into every peer's mailbox over the xGMI links and the
// consumer kernel polls its LOCAL mailbox.
//
// Wire format ("LL": data and flag travel in one 8-byte store, which the fabric delivers atomically, so no fence and
// no second round trip is needed): a double is split in two 32-bit halves, each stored as one 64-bit word
// { half, seq }.  A word is valid for exchange number `seq` when its upper half equals seq.  seq is never 0 (the
// mailbox starts zeroed).  Slots are double-buffered by the parity of seq: a peer can run at most one exchange ahead
// of a rank that has not yet read (its exchange seq+1 completes only after this rank's send seq+1, which is ordered
// after this rank's read of seq on the stream), so two buffers suffice — PROVIDED every send of a rank is followed by its own
// receive of the same exchange before it sends again.  A caller that sends without receiving (bench.py times launch A alone)
// must put a barrier in front of that: its second such send would overwrite what a slower peer is still polling for.
//
// Every poll loop is bounded (wall clock, 100 MHz) and an expired wait is FATAL for the communicator, like a failed
// MPI_Allreduce in the reference (LamellarOrderParameterGPU.cc:69-77 aborts): the waiting kernel counts it, raises the
// sticky flag in host-visible memory (comm_fail) and hands NaN to its consumer — the fused step then writes NaN bias factors
// and forces and deposits nothing, the stand-alone all-reduce returns NaN — and every later call that touches the mailbox
// returns MTD_ERR_COMM_TIMEOUT (comm.hip: comm_next / comm_current).  It never hangs and never goes on with a stale value.
#pragma once

#include "mtd_device.hpp"

namespace mtd
{

// What an expired wait hands on instead of a sum: a quiet NaN with a payload of its own, so that it travels in band — through the
// sums over blocks and ranks, and through the mailbox to ranks whose own waits were fine — and can still be told from an
// ARITHMETIC NaN.  A diverged simulation (NaN positions) produces NaN sums with the default payload: that is not a communication
// failure, and it goes through the step exactly as without a mailbox (the reference deposits the NaN too).  Nothing relies on
// an adder carrying the payload along: every hand-over tests its operands and sets the value again.
constexpr unsigned long long MTD_COMM_POISON_BITS = 0x7ff8c0de00000000ull;
__device__ __forceinline__ double comm_poison() { return __longlong_as_double((long long)MTD_COMM_POISON_BITS); }
__device__ __forceinline__ bool is_comm_poison(const double x)
    {
    return (((unsigned long long)__double_as_longlong(x) >> 32) & 0x7fffffffull) == (MTD_COMM_POISON_BITS >> 32);
    }


constexpr int COMM_MAX_RANKS = MTD_COMM_MAX_RANKS;

struct CommK
    {
    unsigned long long *box[COMM_MAX_RANKS];   // mailbox of every rank as mapped in THIS process (box[rank] is local)
    unsigned int rank, world;
    unsigned int seq;                           // exchange number of this launch (never 0)
    unsigned int words_per_rank;                // 2 * max_doubles
    unsigned long long *ll;                     // local: block sums of the sending launch in the same wire format, [block][2 * n]
    unsigned int *err;                          // timeouts seen (local)
    unsigned int *err_host;                     // sticky failure flag in pinned host memory (read by the host without a sync)
    unsigned long long timeout_ticks;           // wall_clock64 ticks (100 MHz)
    };

__device__ __forceinline__ unsigned long long *comm_slot(const CommK &k, unsigned int owner, unsigned int from)
    {
    return k.box[owner] + ((size_t)(k.seq & 1u) * k.world + from) * k.words_per_rank;
    }

// an expired wait: counted, and the communicator is dead from here on (the host sees the flag at its next call)
__device__ __forceinline__ void comm_fail(const CommK &k)
    {
    atomicAdd(k.err, 1u);
    __hip_atomic_store(k.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }

// one word of this rank's payload into the mailbox of rank `dst`
__device__ __forceinline__ void comm_send_word(const CommK &k, unsigned int dst, unsigned int w, unsigned int half)
    {
    const unsigned long long word = ((unsigned long long)k.seq << 32) | (unsigned long long)half;
    __hip_atomic_store(comm_slot(k, dst, k.rank) + w, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }

// word `w` of rank `src`'s payload from the LOCAL mailbox; spins until it carries this exchange's number; ok = false
// when the wait expired (the returned word is then meaningless)
__device__ __forceinline__ unsigned int comm_recv_word(const CommK &k, unsigned int src, unsigned int w, bool &ok)
    {
    ok = true;
    const unsigned long long *p = comm_slot(k, k.rank, src) + w;
    unsigned long long word = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned int)(word >> 32) != k.seq)
        {
        const unsigned long long t0 = wall_clock64();
        for (;;)
            {
            __builtin_amdgcn_s_sleep(1);
            word = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned int)(word >> 32) == k.seq) break;
            if (wall_clock64() - t0 > k.timeout_ticks)
                {
                comm_fail(k);
                ok = false;
                break;
                }
            }
        }
    return (unsigned int)word;
    }

// The same wire format inside one GPU (block -> collector block of the same launch): agent-scope stores and loads go
// past the non-coherent caches, no fence, no atomic, one memory round trip between the last store and the collector.
__device__ __forceinline__ void ll_store(unsigned long long *p, unsigned int seq, double v)
    {
    __hip_atomic_store(p, ((unsigned long long)seq << 32) | (unsigned int)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, ((unsigned long long)seq << 32) | (unsigned int)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

// Collector side, ONE wave: lane l adds up the NS sums of blocks l, l + 64, ... (fixed order), four blocks' worth of loads
// in flight per round trip; v[] accumulates.  Block b's sums sit at k.ll + b * stride_words + offset_words (two words per
// double).  Spins (bounded) until every word carries this exchange's number.  Returns true (in every lane) when a wait
// expired: v[] then holds NaN.
template<int NS>
__device__ __forceinline__ bool ll_collect_strided(const CommK &k, const unsigned int n_blocks, const unsigned int stride_words,
                                                   const unsigned int offset_words, double (&v)[3])
    {
    const unsigned int lane = threadIdx.x & 63;
    bool any_expired = false;
    for (unsigned int b0 = lane; b0 < n_blocks; b0 += 4 * MTD_WAVE)
        {
        unsigned long long w[4][NS][2];
        unsigned long long t0 = 0;
        bool expired = false;
        for (;;)
            {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                {
                const unsigned int b = b0 + j * MTD_WAVE;
                if (b < n_blocks)
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        {
                        const unsigned long long *p = k.ll + (size_t)b * stride_words + offset_words + 2 * i;
                        w[j][i][0] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w[j][i][1] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (b0 + j * MTD_WAVE < n_blocks)
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        ok = ok && (unsigned int)(w[j][i][0] >> 32) == k.seq && (unsigned int)(w[j][i][1] >> 32) == k.seq;
            if (ok) break;
            if (t0 == 0) t0 = wall_clock64();
            else if (wall_clock64() - t0 > k.timeout_ticks)
                {
                comm_fail(k);
                expired = true;
                break;
                }
            __builtin_amdgcn_s_sleep(1);
            }
        if (expired)
            {
            any_expired = true;
            break;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (b0 + j * MTD_WAVE < n_blocks)
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    v[i] += __hiloint2double((int)(unsigned int)w[j][i][1], (int)(unsigned int)w[j][i][0]);
        }
    any_expired = __any(any_expired);
    if (any_expired)
        {
#pragma unroll
        for (int i = 0; i < NS; ++i) v[i] = comm_poison();                                  // poisons the totals
        }
    return any_expired;
    }

// The same for sums stored by COLUMNS: word w of block b at k.ll[w * pitch + b] (value i = words first_word + 2 i and + 2 i + 1).
// Consecutive lanes read consecutive 8-byte words, so a wave-wide load touches 8 cache lines instead of 64 — this is the
// layout of the all-to-all hand-offs of the one-launch step, where EVERY block reads every block's sums.
__device__ __forceinline__ void ll_store_column(unsigned long long *ll, const unsigned int pitch, const unsigned int word, const unsigned int b,
                                                const unsigned int seq, const double v)
    {
    __hip_atomic_store(ll + (size_t)word * pitch + b, ((unsigned long long)seq << 32) | (unsigned int)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ll + (size_t)(word + 1) * pitch + b, ((unsigned long long)seq << 32) | (unsigned int)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

// Every block of a launch polls here at the same time: a retry waits ~0.3 us (s_sleep 10) so that the readers do not
// keep the few memory channels behind these lines busy while the writers' stores are still on their way.
template<int NS>
__device__ __forceinline__ bool ll_collect_columns(const CommK &k, const unsigned int n_blocks, const unsigned int pitch,
                                                   const unsigned int first_word, double (&v)[3])
    {
    const unsigned int lane = threadIdx.x & 63;
    bool any_expired = false;
    for (unsigned int b0 = lane; b0 < n_blocks; b0 += 4 * MTD_WAVE)
        {
        unsigned long long w[4][NS][2];
        unsigned long long t0 = 0;
        bool expired = false;
        for (;;)
            {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                {
                const unsigned int b = b0 + j * MTD_WAVE;
                if (b < n_blocks)
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        {
                        const unsigned long long *p = k.ll + (size_t)(first_word + 2 * i) * pitch + b;
                        w[j][i][0] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w[j][i][1] = __hip_atomic_load(p + pitch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (b0 + j * MTD_WAVE < n_blocks)
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        ok = ok && (unsigned int)(w[j][i][0] >> 32) == k.seq && (unsigned int)(w[j][i][1] >> 32) == k.seq;
            if (ok) break;
            if (t0 == 0) t0 = wall_clock64();
            else if (wall_clock64() - t0 > k.timeout_ticks)
                {
                comm_fail(k);
                expired = true;
                break;
                }
            __builtin_amdgcn_s_sleep(10);
            }
        if (expired)
            {
            any_expired = true;
            break;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (b0 + j * MTD_WAVE < n_blocks)
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    v[i] += __hiloint2double((int)(unsigned int)w[j][i][1], (int)(unsigned int)w[j][i][0]);
        }
    any_expired = __any(any_expired);
    if (any_expired)
        {
#pragma unroll
        for (int i = 0; i < NS; ++i) v[i] = comm_poison();                                  // poisons the totals
        }
    return any_expired;
    }

// the sums of one launch stored densely, [block][NS] (sharded CV pass of the two-launch step)
template<int NS>
__device__ __forceinline__ bool ll_collect_wave(const CommK &k, const unsigned int n_blocks, double (&v)[3])
    {
    return ll_collect_strided<NS>(k, n_blocks, 2 * NS, 0, v);
    }

// Loads from an EXPORTED bulk buffer (mtd_comm_share: uncached device memory, local or a peer's mapping): system-scope loads
// go past the L2.  A plain load may hit a stale L2 line left by an earlier ordinary life of the buffer's addresses — uncached
// stores neither update nor invalidate it (measured: tools/probe_uncached.hip, comm.hip).
__device__ __forceinline__ double ld_exported(const double *p)
    {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
__device__ __forceinline__ double2 ld_exported(const double2 *p)
    {
    const double *q = (const double *)p;
    return make_double2(ld_exported(q), ld_exported(q + 1));
    }

__device__ __forceinline__ unsigned int dbl_lo(double v) { return (unsigned int)__double2loint(v); }
__device__ __forceinline__ unsigned int dbl_hi(double v) { return (unsigned int)__double2hiint(v); }

// ONE full wave sends n <= 3 doubles (identical in every lane) to every rank: lane = dst * 2n + word
__device__ __forceinline__ void comm_send_wave(const CommK &k, const double (&v)[3], const unsigned int n)
    {
    const unsigned int lane = threadIdx.x & 63;
    const unsigned int nw = 2 * n;
    if (lane < k.world * nw)
        {
        const unsigned int dst = lane / nw, w = lane % nw;
        const double x = (w >> 1) == 0 ? v[0] : ((w >> 1) == 1 ? v[1] : v[2]);
        comm_send_word(k, dst, w, (w & 1) ? dbl_hi(x) : dbl_lo(x));
        }
    }

// ONE full wave receives n <= 3 doubles from every rank and adds them up in rank order (the same bits on every rank);
// result in every lane.  world * 2n <= 48 lanes poll one word each.
__device__ __forceinline__ void comm_recv_sum_wave(const CommK &k, double (&total)[3], const unsigned int n)
    {
    const unsigned int lane = threadIdx.x & 63;
    const unsigned int nw = 2 * n;
    unsigned int half = 0;
    bool ok = true;
    if (lane < k.world * nw) half = comm_recv_word(k, lane / nw, lane % nw, ok);
    const bool failed = __any(!ok);                                  // wave-uniform: any expired wait poisons every total
    bool remote = false;                                             // a rank whose own collection expired sent the poison value
#pragma unroll
    for (int i = 0; i < 3; ++i)
        {
        total[i] = 0.0;
        if (i < (int)n)
            for (unsigned int r = 0; r < k.world; ++r)
                {
                const unsigned int lo = __shfl(half, (int)(r * nw + 2 * i), MTD_WAVE);
                const unsigned int hi = __shfl(half, (int)(r * nw + 2 * i + 1), MTD_WAVE);
                const double x = __hiloint2double((int)hi, (int)lo);
                remote = remote || is_comm_poison(x);
                total[i] += x;
                }
        }
    if (failed || remote)
#pragma unroll
        for (int i = 0; i < 3; ++i) total[i] = comm_poison();
    }

} // namespace mtd
