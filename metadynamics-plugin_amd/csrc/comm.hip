// comm.hip — xGMI mailbox: all-reduce (sum) of a few doubles between the GPUs of one node.
//
// Replaces, for the small per-step messages of the sharded bias step, the reference's host-staged
// MPI_Allreduce (LamellarOrderParameterGPU.cc:69-77; SteinhardtQl.cc:183-191; WellTemperedEnsemble.cc:57-63) and a
// library all-reduce: one process per GPU, every rank allocates a mailbox (uncached device memory), exports it with
// hipIpcGetMemHandle, the handles travel once through the caller's control plane (torch.distributed / MPI), every
// rank maps its peers' mailboxes and from then on kernels store into them directly over xGMI.  Large buffers (the
// replicated mesh, the packed walker deltas) stay on RCCL.
//
// Protocol and bounds: comm_device.hpp.
#include "comm_host.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace
{

using namespace mtd;

constexpr int AR_THREADS = 256;

// Stand-alone all-reduce of n doubles in place (one block): send, poll, sum in rank order.
__global__ __launch_bounds__(AR_THREADS) void k_comm_allreduce(const CommK k, double *__restrict__ values, const unsigned int n)
    {
    extern __shared__ unsigned int s_half[];            // world * 2n halves
    const unsigned int nw = 2 * n;
    for (unsigned int i = threadIdx.x; i < k.world * nw; i += AR_THREADS)
        {
        const unsigned int dst = i / nw, w = i % nw;
        const double x = values[w >> 1];
        comm_send_word(k, dst, w, (w & 1) ? dbl_hi(x) : dbl_lo(x));
        }
    __shared__ int s_failed;
    if (threadIdx.x == 0) s_failed = 0;
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < k.world * nw; i += AR_THREADS)
        {
        bool ok;
        s_half[i] = comm_recv_word(k, i / nw, i % nw, ok);
        if (!ok) s_failed = 1;
        }
    __syncthreads();
    for (unsigned int j = threadIdx.x; j < n; j += AR_THREADS)
        {
        if (s_failed)                                   // an expired wait: NaN, never a stale or partial sum
            {
            values[j] = comm_poison();
            continue;
            }
        double t = 0.0;
        bool remote = false;                            // a rank handed its own failure on
        for (unsigned int r = 0; r < k.world; ++r)
            {
            const double x = __hiloint2double((int)s_half[r * nw + 2 * j + 1], (int)s_half[r * nw + 2 * j]);
            remote = remote || is_comm_poison(x);
            t += x;
            }
        values[j] = remote ? comm_poison() : t;
        }
    }

// ---- large all-reduce by remote loads ("pull"), no collective library ------------------------------------------------
// Every rank stages its contribution in an exported (uncached) buffer; after a mailbox barrier rank r adds up slice r of all
// ranks' staging buffers IN RANK ORDER (remote loads over xGMI, one reduce-scatter) into its exported slice buffer; after a
// second barrier every rank copies all slices home (an all-gather by remote loads).  Every rank ends with the same bits.
// Re-use is safe without further barriers: a staging buffer is rewritten by the NEXT call's stage kernel, which stream
// order puts after this call's second barrier — by then every peer has finished reading it (its reduce kernel precedes its
// send of barrier 2); a slice buffer is rewritten by the next call's reduce kernel, which follows the next call's first
// barrier — which every peer sends only after its gather of this call.
struct PullK
    {
    const double *in[COMM_MAX_RANKS];
    const double *out[COMM_MAX_RANKS];
    unsigned int rank, world;
    size_t count, chunk;                                // chunk: doubles per slice (a multiple of 32)
    const double *token;                                // the last barrier's sum: the comm poison when a wait expired
    };

constexpr int PULL_THREADS = 256;

__global__ __launch_bounds__(PULL_THREADS) void k_pull_stage(const double *__restrict__ src, double *__restrict__ staged, const size_t n, double *token)
    {
    const size_t stride = (size_t)gridDim.x * PULL_THREADS;
    for (size_t i = (size_t)blockIdx.x * PULL_THREADS + threadIdx.x; i < n; i += stride) staged[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) *token = 1.0;
    }

__global__ __launch_bounds__(PULL_THREADS) void k_pull_reduce(const PullK p, double *__restrict__ slice_out, double *token_next)
    {
    const bool failed = is_comm_poison(*p.token);
    const size_t lo = (size_t)p.rank * p.chunk, hi = lo + p.chunk < p.count ? lo + p.chunk : p.count;
    const size_t stride = (size_t)gridDim.x * PULL_THREADS;
    for (size_t i = lo + (size_t)blockIdx.x * PULL_THREADS + threadIdx.x; i < hi; i += stride)
        {
        double t = 0.0;
        for (unsigned int q = 0; q < p.world; ++q) t += ld_exported(p.in[q] + i);
        slice_out[i] = failed ? comm_poison() : t;
        }
    if (blockIdx.x == 0 && threadIdx.x == 0) *token_next = failed ? comm_poison() : 1.0;
    }

__global__ __launch_bounds__(PULL_THREADS) void k_pull_gather(const PullK p, double *__restrict__ dst)
    {
    const bool failed = is_comm_poison(*p.token);
    const size_t stride = (size_t)gridDim.x * PULL_THREADS;
    for (size_t i = (size_t)blockIdx.x * PULL_THREADS + threadIdx.x; i < p.count; i += stride)
        {
        const double v = ld_exported(p.out[i / p.chunk] + i);
        dst[i] = failed ? comm_poison() : v;
        }
    }

// Uncached device buffers are never handed back to the runtime while the process lives.  Measured with a stand-alone
// program that contains no code of this library (tools/diag/probe_uncached.hip, log in profiles/r2/uncached_reuse_probe.log;
// ROCm 7.2, gfx950): once a range of device addresses has lived as a hipDeviceMallocUncached allocation and has been freed,
// an ORDINARY hipMalloc that lands on it is no longer coherent between the L2 and HBM — kernels see what kernels wrote, but a
// copy engine (hipMemcpy to the host) reads stale HBM under it, and host-to-device copies (twiddle tables, mode
// coefficients) can be shadowed by stale L2 lines: 77 of 480 regions came back with wrong contents, none before the first
// uncached life.  That is what round 1's slab fuzzing hit (a later mesh on those addresses read garbage tables and, with
// garbage particle offsets, faulted) — with this library's own kernels in bounds (the replay tools/diag/diag_slab_fault.py maps
// every allocation; same wrong results from the stand-alone probe).  The other direction exists as well: PLAIN loads from an
// uncached buffer can hit stale L2 lines of an earlier ordinary life of its addresses, so every read of an exported buffer
// goes past the L2 (system-scope loads: comm_device.hpp ld_exported) and nothing is transformed in place in one.
// Released buffers wait here for the next request of EXACTLY their size (a larger one would hide an out-of-bounds access
// behind its slack); the pool is bounded by the largest set of buffers alive at one time.
struct PooledBuffer { void *p; size_t bytes; bool in_use; };
std::vector<PooledBuffer> g_uncached_pool;
std::mutex g_uncached_mutex;

// diagnostic switches (tools/diag/diag_slab_fault.py): MTD_COMM_POOL=0 hands released buffers back to the runtime (reproduces the
// hazard above), MTD_TRACE_ALLOC=1 prints every allocation
bool pool_off() { const char *e = std::getenv("MTD_COMM_POOL"); return e && e[0] == '0'; }
bool trace_on() { const char *e = std::getenv("MTD_TRACE_ALLOC"); return e && e[0] == '1'; }

hipError_t uncached_acquire(void **out, size_t bytes)
    {
    std::lock_guard<std::mutex> lock(g_uncached_mutex);
    if (!pool_off())
        for (PooledBuffer &b : g_uncached_pool)
            if (!b.in_use && b.bytes == bytes)
                {
                b.in_use = true;
                *out = b.p;
                return hipSuccess;
                }
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess)
        {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
        }
    if (e != hipSuccess)
        {
        (void)hipGetLastError();
        return e;
        }
    if (trace_on()) fprintf(stderr, "[mtd] uncached alloc %p .. %p (%zu bytes)\n", p, (char *)p + bytes, bytes);
    if (!pool_off()) g_uncached_pool.push_back({p, bytes, true});
    *out = p;
    return hipSuccess;
    }

void uncached_release(void *p)
    {
    std::lock_guard<std::mutex> lock(g_uncached_mutex);
    if (pool_off())
        {
        if (trace_on()) fprintf(stderr, "[mtd] uncached free %p\n", p);
        (void)hipFree(p);
        return;
        }
    for (PooledBuffer &b : g_uncached_pool)
        if (b.p == p) b.in_use = false;
    }

unsigned long long env_timeout_ticks()
    {
    const char *e = std::getenv("MTD_COMM_TIMEOUT_MS");
    double ms = 5000.0;
    if (e && *e) ms = std::atof(e);
    if (!(ms > 0.0)) ms = 5000.0;
    return (unsigned long long)(ms * 1.0e5);            // wall_clock64: 100 MHz
    }

} // namespace

namespace mtd
{

int comm_next(mtd_comm *c, CommK &k)
    {
    if (!c || !c->connected) return MTD_ERR_INVALID_ARGUMENT;
    if (comm_failed(c)) return MTD_ERR_COMM_TIMEOUT;
    // never 0 (the mailbox starts zeroed), and the parity keeps alternating across the wrap (the slot buffers are
    // double-buffered by it): 0xffffffff is odd, so 2 follows, not 1
    c->seq = (c->seq == 0xffffffffu) ? 2u : c->seq + 1u;
    k = c->k;
    k.seq = c->seq;
    return MTD_SUCCESS;
    }

int comm_current(const mtd_comm *c, CommK &k)
    {
    if (!c || !c->connected || c->seq == 0) return MTD_ERR_INVALID_ARGUMENT;
    if (comm_failed(c)) return MTD_ERR_COMM_TIMEOUT;
    k = c->k;
    k.seq = c->seq;
    return MTD_SUCCESS;
    }

} // namespace mtd

extern "C" {

int mtd_comm_create(mtd_comm **out, unsigned int rank, unsigned int world, unsigned int max_doubles)
    {
    if (!out || world == 0 || world > MTD_COMM_MAX_RANKS || rank >= world || max_doubles == 0 || max_doubles > 4096)
        return MTD_ERR_INVALID_ARGUMENT;
    mtd_comm *c = new (std::nothrow) mtd_comm;
    if (!c) return (int)hipErrorOutOfMemory;
    std::memset(c, 0, sizeof(*c));
    c->k.rank = rank;
    c->k.world = world;
    c->k.words_per_rank = 2 * max_doubles;
    c->k.timeout_ticks = env_timeout_ticks();
    c->max_doubles = max_doubles;
    c->bytes = sizeof(unsigned long long) * 2 * world * c->k.words_per_rank;
    // uncached device memory: peers' stores and this GPU's polling loads must not sit in a non-coherent L2
    hipError_t e = uncached_acquire(&c->local, c->bytes);
    if (e != hipSuccess)
        {
        delete c;
        return (int)e;
        }
    void *aux = nullptr;
    const size_t aux_bytes = 64 + sizeof(unsigned long long) * COMM_LL_BLOCKS * 2 * COMM_LL_DOUBLES;
    e = hipMalloc(&aux, aux_bytes);
    if (e == hipSuccess) e = hipMemset(aux, 0, aux_bytes);
    unsigned int *h_err = nullptr, *d_err_host = nullptr;
    if (e == hipSuccess) e = hipHostMalloc((void **)&h_err, sizeof(unsigned int), hipHostMallocMapped);
    if (e == hipSuccess)
        {
        *h_err = 0;
        e = hipHostGetDevicePointer((void **)&d_err_host, h_err, 0);
        }
    if (e == hipSuccess) e = hipMemset(c->local, 0, c->bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess)
        {
        if (aux) (void)hipFree(aux);
        if (h_err) (void)hipHostFree(h_err);
        uncached_release(c->local);
        delete c;
        return (int)e;
        }
    c->h_err = h_err;
    c->k.err_host = d_err_host;
    c->k.err = (unsigned int *)aux;
    c->k.ll = (unsigned long long *)((char *)aux + 64);
    c->pull_token = (double *)((char *)aux + 32);        // (bytes 32 .. 47 of the header: two barrier tokens)
    c->k.box[rank] = (unsigned long long *)c->local;
    if (world == 1) c->connected = 1;
    *out = c;
    return MTD_SUCCESS;
    }

int mtd_comm_handle(mtd_comm *c, void *out_handle)
    {
    if (!c || !out_handle) return MTD_ERR_INVALID_ARGUMENT;
    static_assert(sizeof(hipIpcMemHandle_t) == MTD_COMM_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    MTD_HIP_TRY(hipIpcGetMemHandle(&h, c->local));
    std::memcpy(out_handle, &h, sizeof(h));
    return MTD_SUCCESS;
    }

int mtd_comm_connect(mtd_comm *c, const void *handles)
    {
    if (!c || !handles) return MTD_ERR_INVALID_ARGUMENT;
    if (c->connected) return MTD_SUCCESS;
    for (unsigned int r = 0; r < c->k.world; ++r)
        {
        if (r == c->k.rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char *)handles + (size_t)r * MTD_COMM_HANDLE_BYTES, sizeof(h));
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess)
            {
            (void)hipGetLastError();
            for (unsigned int q = 0; q < r; ++q)
                if (q != c->k.rank && c->k.box[q])
                    {
                    (void)hipIpcCloseMemHandle(c->k.box[q]);
                    c->k.box[q] = nullptr;
                    }
            return (int)e;
            }
        c->k.box[r] = (unsigned long long *)p;
        }
    c->connected = 1;
    return MTD_SUCCESS;
    }

int mtd_comm_share(mtd_comm *c, size_t bytes, void **d_local, unsigned int *slot, void *out_handle)
    {
    if (!c || !d_local || !slot || !out_handle || bytes == 0) return MTD_ERR_INVALID_ARGUMENT;
    if (c->n_shared >= MTD_COMM_MAX_SHARED) return MTD_ERR_UNSUPPORTED;
    void *p = nullptr;
    hipError_t e = uncached_acquire(&p, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof(h));
    if (e == hipSuccess && c->k.world > 1) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess)
        {
        (void)hipGetLastError();
        uncached_release(p);
        return (int)e;
        }
    const unsigned int s = c->n_shared++;
    c->shared_local[s] = p;
    c->shared_peer[s][c->k.rank] = p;
    std::memcpy(out_handle, &h, sizeof(h));
    *d_local = p;
    *slot = s;
    return MTD_SUCCESS;
    }

int mtd_comm_open(mtd_comm *c, unsigned int slot, const void *handles, void **peers)
    {
    if (!c || slot >= c->n_shared || !peers || (c->k.world > 1 && !handles)) return MTD_ERR_INVALID_ARGUMENT;
    for (unsigned int r = 0; r < c->k.world; ++r)
        {
        if (r != c->k.rank && !c->shared_peer[slot][r])
            {
            hipIpcMemHandle_t h;
            std::memcpy(&h, (const char *)handles + (size_t)r * MTD_COMM_HANDLE_BYTES, sizeof(h));
            void *p = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess)
                {
                (void)hipGetLastError();
                return (int)e;
                }
            c->shared_peer[slot][r] = p;
            }
        peers[r] = c->shared_peer[slot][r];
        }
    return MTD_SUCCESS;
    }

int mtd_comm_allreduce_small(mtd_comm *c, double *d_values, unsigned int n, mtd_stream_t stream)
    {
    if (!c || !d_values || n == 0 || n > c->max_doubles) return MTD_ERR_INVALID_ARGUMENT;
    CommK k;
    int rc = comm_next(c, k);
    if (rc) return rc;
    const size_t lds = sizeof(unsigned int) * 2 * n * k.world;
    if (lds > 60000) return MTD_ERR_UNSUPPORTED;
    k_comm_allreduce<<<1, AR_THREADS, lds, (hipStream_t)stream>>>(k, d_values, n);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

size_t mtd_comm_pull_bytes(size_t max_doubles)
    {
    return ((max_doubles * sizeof(double) + 255) / 256) * 256;
    }

int mtd_comm_pull_attach(mtd_comm *c, size_t max_doubles, void *const *in_peers, void *const *out_peers)
    {
    if (!c || !c->connected || max_doubles == 0 || !in_peers || !out_peers) return MTD_ERR_INVALID_ARGUMENT;
    for (unsigned int r = 0; r < c->k.world; ++r)
        {
        if (!in_peers[r] || !out_peers[r]) return MTD_ERR_INVALID_ARGUMENT;
        c->pull_in[r] = (const double *)in_peers[r];
        c->pull_out[r] = (const double *)out_peers[r];
        }
    c->pull_max = max_doubles;
    return MTD_SUCCESS;
    }

int mtd_comm_allreduce_pull(mtd_comm *c, double *d_buffer, size_t count, mtd_stream_t stream)
    {
    if (!c || !d_buffer || count == 0) return MTD_ERR_INVALID_ARGUMENT;
    if (!c->connected) return MTD_ERR_INVALID_ARGUMENT;
    if (comm_failed(c)) return MTD_ERR_COMM_TIMEOUT;
    if (c->k.world == 1) return MTD_SUCCESS;                     // the sum over one rank
    if (count > c->pull_max) return MTD_ERR_INVALID_ARGUMENT;    // (also: never attached)
    hipStream_t s = (hipStream_t)stream;
    PullK p;
    std::memset(&p, 0, sizeof(p));
    for (unsigned int r = 0; r < c->k.world; ++r)
        {
        p.in[r] = c->pull_in[r];
        p.out[r] = c->pull_out[r];
        }
    p.rank = c->k.rank;
    p.world = c->k.world;
    p.count = count;
    p.chunk = (((count + p.world - 1) / p.world + 31) / 32) * 32;
    double *tok = c->pull_token;
    size_t nb = (count + PULL_THREADS - 1) / PULL_THREADS;
    const unsigned int blocks = (unsigned int)(nb > 2048 ? 2048 : nb);
    k_pull_stage<<<blocks, PULL_THREADS, 0, s>>>(d_buffer, (double *)c->pull_in[p.rank], count, tok);
    MTD_LAUNCH_CHECK();
    int rc = mtd_comm_allreduce_small(c, tok, 1, stream);        // barrier 1: every rank's contribution is staged
    if (rc) return rc;
    p.token = tok;
    nb = (p.chunk + PULL_THREADS - 1) / PULL_THREADS;
    k_pull_reduce<<<(unsigned int)(nb > 2048 ? 2048 : nb), PULL_THREADS, 0, s>>>(p, (double *)c->pull_out[p.rank], tok + 1);
    MTD_LAUNCH_CHECK();
    rc = mtd_comm_allreduce_small(c, tok + 1, 1, stream);        // barrier 2: every slice is reduced
    if (rc) return rc;
    p.token = tok + 1;
    k_pull_gather<<<blocks, PULL_THREADS, 0, s>>>(p, d_buffer);
    MTD_LAUNCH_CHECK();
    return MTD_SUCCESS;
    }

int mtd_comm_status(mtd_comm *c, unsigned int *timeouts, mtd_stream_t stream)
    {
    if (!c || !timeouts) return MTD_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    MTD_HIP_TRY(hipMemcpyAsync(timeouts, c->k.err, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    MTD_HIP_TRY(hipStreamSynchronize(s));
    return MTD_SUCCESS;
    }

unsigned int mtd_comm_world(const mtd_comm *c) { return c ? c->k.world : 0; }
unsigned int mtd_comm_rank(const mtd_comm *c) { return c ? c->k.rank : 0; }

int mtd_comm_destroy(mtd_comm *c)
    {
    if (!c) return MTD_SUCCESS;
    (void)hipDeviceSynchronize();
    for (unsigned int r = 0; r < c->k.world; ++r)
        if (r != c->k.rank && c->k.box[r]) (void)hipIpcCloseMemHandle(c->k.box[r]);
    for (unsigned int s = 0; s < c->n_shared; ++s)
        {
        for (unsigned int r = 0; r < c->k.world; ++r)
            if (r != c->k.rank && c->shared_peer[s][r]) (void)hipIpcCloseMemHandle(c->shared_peer[s][r]);
        if (c->shared_local[s]) uncached_release(c->shared_local[s]);
        }
    if (c->k.err) (void)hipFree(c->k.err);
    if (c->h_err) (void)hipHostFree((void *)c->h_err);
    if (c->local) uncached_release(c->local);
    delete c;
    return MTD_SUCCESS;
    }

} // extern "C"
