// comm_host.hpp — host-side handle of the xGMI mailbox (opaque `mtd_comm` of mtd_abi.h)
#pragma once

#include "comm_device.hpp"

namespace mtd
{
constexpr unsigned int COMM_LL_BLOCKS = 1024;   // >= LAM_MAX_BLOCKS: block sums of one launch
constexpr unsigned int COMM_LL_DOUBLES = 3;     // per block (CHAIN_MAX_CV)
}

struct mtd_comm
    {
    mtd::CommK k;               // seq is filled per exchange
    void *local;                // this rank's mailbox (uncached device memory)
    volatile unsigned int *h_err;   // sticky failure flag (pinned host memory, mapped into the device: CommK::err_host)
    size_t bytes;
    unsigned int max_doubles;
    unsigned int seq;           // number of the last exchange started (0: none yet)
    int connected;
    // bulk buffers shared with the peers (mtd_comm_share / mtd_comm_open: the slab-decomposed mesh): released with the comm
    void *shared_local[MTD_COMM_MAX_SHARED];
    void *shared_peer[MTD_COMM_MAX_SHARED][MTD_COMM_MAX_RANKS];
    unsigned int n_shared;
    // large all-reduce by remote loads (mtd_comm_pull_attach / mtd_comm_allreduce_pull): every rank's staging buffer and
    // its buffer of reduced slices as mapped in this process, the capacity in doubles, and a device token for the barriers
    const double *pull_in[MTD_COMM_MAX_RANKS];
    const double *pull_out[MTD_COMM_MAX_RANKS];
    size_t pull_max;
    double *pull_token;
    };

namespace mtd
{
// start exchange seq+1 / describe the exchange started last (for the kernel that receives it)
int comm_next(mtd_comm *c, CommK &k);
int comm_current(const mtd_comm *c, CommK &k);
// a wait of this communicator has expired (sticky): every entry point that would use it returns MTD_ERR_COMM_TIMEOUT
inline bool comm_failed(const mtd_comm *c) { return c && c->h_err && *c->h_err != 0; }
}
