// metad_host.hpp — host-side handle of the bias-grid engine (opaque `mtd_metad` of mtd_abi.h)
#pragma once

#include "metad_device.hpp"

struct mtd_metad
    {
    mtd::MetadCfg cfg;
    unsigned int stride;
    int add_bias;
    void *slab;
    // fused path (fused.hip): the second reweighting pass + accumulate of the last deposit is deferred
    // into the next CV launch; pending_apply != 0 means the grid arrays are one k_apply behind.
    int pending_apply;
    // the last grid launch deposited a hill and left w(s) to whoever reads it: mtd_metad_get_state evaluates it on the final
    // weight grid (whoever ran the deferred pass — a flush, launch A, a passenger — the read-out is still owed)
    int w_stale;
    // particle-sharded fused step: the per-rank CV totals travel through this xGMI mailbox (comm.hip); not owned
    struct mtd_comm *comm;
    // one-launch step (fused_step.hip): block sums of the launch in the mailbox's wire format (hand-off between the blocks of
    // ONE launch), its exchange counter, and the sticky failure flag of its bounded waits (pinned host memory)
    unsigned long long *d_ll;
    unsigned int *d_step_err;
    volatile unsigned int *h_step_err;
    unsigned int *d_step_err_host;
    unsigned int step_seq;
    unsigned int last_launches;     // launches the last mtd_fused_step took (1: the persistent kernel, 2: the two-launch form)
    int step_mode;                  // mtd_fused_step_set_mode: -1 environment / default, 0 two launches, 1 one launch where possible
    // multiple walkers: the (communicator, stride, add_hills) for which all walkers were found to agree (mtd_metad_update_bias_walkers)
    const void *walkers_checked;
    unsigned int walkers_stride;
    int walkers_add_bias;
    };

namespace mtd
{
// run the deferred k_apply if one is pending (called by every entry point that reads or updates the grid)
int metad_flush(mtd_metad *m, hipStream_t s);
// fused.hip: deferred apply + one launch for the whole update (n_cv <= 3), MTD_ERR_UNSUPPORTED otherwise
int fused_grid_step(mtd_metad *m, unsigned int timestep, hipStream_t s);
// the deferred pass as a passenger of another kernel of this library on the same stream (metad.hip): the engine announces a
// pending pass, a kernel with room takes it (cfg copied out, pending flag cleared) and runs apply_cells in extra blocks
void announce_pending_apply(mtd_metad *m, hipStream_t s);
void withdraw_pending_apply(mtd_metad *m);
mtd_metad *take_pending_apply(hipStream_t s, MetadCfg &cfg);
void commit_pending_apply(mtd_metad *m);
// fused_step.hip: release the one-launch step's buffers (mtd_metad_destroy)
void fused_step_release(mtd_metad *m);
}
