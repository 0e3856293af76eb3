// core.hip — library-level entry points of libmtd_hip.so
#include "mtd_device.hpp"

extern "C" {

int mtd_abi_version(void) { return MTD_ABI_VERSION; }

const char *mtd_status_string(int status)
    {
    switch (status)
        {
        case MTD_SUCCESS: return "success";
        case MTD_ERR_INVALID_ARGUMENT: return "invalid argument";
        case MTD_ERR_UNSUPPORTED: return "unsupported configuration";
        case MTD_ERR_NO_DEVICE: return "no HIP device";
        case MTD_ERR_COMM_TIMEOUT: return "xGMI mailbox: a peer did not answer in time (step poisoned, communicator dead)";
        case MTD_ERR_COLLECTIVE: return "RCCL call failed (mtd_rccl_last_error has the text), or the walkers disagree on stride / add_hills / time step";
        }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown mtd status";
    }

int mtd_device_count(void)
    {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return MTD_ERR_NO_DEVICE;
    return n;
    }

} // extern "C"
