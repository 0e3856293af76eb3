// lamellar_host.hpp — host helpers of the lamellar kernels shared between lamellar.hip and fused.hip
#pragma once

#include "lamellar_device.hpp"

namespace mtd
{
constexpr unsigned int LAM_MAX_BLOCKS = 1024;

// validate a mtd_lamellar_set + box and expand it into the kernel-argument block
int fill_kargs(LamKArgs &k, const mtd_lamellar_set *set, const mtd_box *box);
unsigned int lam_cv_blocks(unsigned int N);
unsigned int lam_force_blocks(unsigned int N);
// hardware sine / cosine for this mode set? (the library setting AND phases inside the instructions' domain)
int lam_fast_trig(const LamKArgs &k);
}
