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
    // particle-sharded fused step: the per-rank CV totals travel through this xGMI mailbox (comm.hip); not owned
    struct mtd_comm *comm;
    };

namespace mtd
{
// run the deferred k_apply if one is pending (called by every entry point that reads or updates the grid)
int metad_flush(mtd_metad *m, hipStream_t s);
// fused.hip: deferred apply + one launch for the whole update (n_cv <= 3), MTD_ERR_UNSUPPORTED otherwise
int fused_grid_step(mtd_metad *m, unsigned int timestep, hipStream_t s);
}
