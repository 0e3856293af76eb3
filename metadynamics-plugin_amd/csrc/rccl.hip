// rccl.hip — large all-reduces over RCCL (xGMI ring collectives) behind the C ABI: the multiple-walker exchange of the bias
// grid's delta arrays (IntegratorMetaDynamics.cc:393-409: four MPI_Allreduce over m_partition_comm, host staged) and the
// replicated mesh of a particle-sharded mesh CV (M + 1 doubles; replaces the ghost-cell exchange + distributed FFT of
// OrderParameterMesh.cc:263-316, 659-746).  The few doubles of the per-step CV sums go through the xGMI mailbox (comm.hip).
//
// RCCL is bound at run time (dlopen): the library carries no link-time dependency on it, a single-GPU run never loads it, and
// in a process that already holds an RCCL (PyTorch bundles one) the loader hands back that copy instead of a second one.
// The communicator is built from a caller-supplied unique id: rank 0 calls mtd_rccl_unique_id, the 128 bytes travel through
// the caller's control plane (MPI_Bcast in HOOMD, the launcher's store here), every rank calls mtd_rccl_create.
#include "mtd_device.hpp"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

struct mtd_rccl
    {
    ncclComm_t comm;
    unsigned int rank, world;
    };

namespace
{

struct RcclApi
    {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    };

// the last failure in words (the ABI returns MTD_ERR_COLLECTIVE; the cause would otherwise be lost)
std::mutex g_err_mutex;
char g_last_error[256] = "";

RcclApi &rccl_api()
    {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []
        {
        for (const char *name : {"librccl.so.1", "librccl.so"})
            {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
            }
        if (!api.handle) return;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy;
        });
    return api;
    }

int rccl_failed(const char *call, ncclResult_t r)
    {
    RcclApi &api = rccl_api();
    std::lock_guard<std::mutex> lock(g_err_mutex);
    std::snprintf(g_last_error, sizeof(g_last_error), "%s: %s (ncclResult %d)", call, api.GetErrorString ? api.GetErrorString(r) : "?", (int)r);
    return MTD_ERR_COLLECTIVE;
    }

} // namespace

extern "C" {

int mtd_rccl_unique_id(void *out_id)
    {
    static_assert(sizeof(ncclUniqueId) == MTD_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!out_id) return MTD_ERR_INVALID_ARGUMENT;
    RcclApi &api = rccl_api();
    if (!api.ok) return MTD_ERR_UNSUPPORTED;
    ncclUniqueId id;
    const ncclResult_t res = api.GetUniqueId(&id);
    if (res != ncclSuccess) return rccl_failed("ncclGetUniqueId", res);
    std::memcpy(out_id, &id, sizeof(id));
    return MTD_SUCCESS;
    }

int mtd_rccl_create(mtd_rccl **out, const void *unique_id, unsigned int rank, unsigned int world)
    {
    if (!out || !unique_id || world == 0 || rank >= world) return MTD_ERR_INVALID_ARGUMENT;
    RcclApi &api = rccl_api();
    if (!api.ok) return MTD_ERR_UNSUPPORTED;
    mtd_rccl *r = new (std::nothrow) mtd_rccl;
    if (!r) return (int)hipErrorOutOfMemory;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    r->rank = rank;
    r->world = world;
    const ncclResult_t res = api.CommInitRank(&r->comm, (int)world, id, (int)rank);
    if (res != ncclSuccess)
        {
        delete r;
        return rccl_failed("ncclCommInitRank", res);
        }
    *out = r;
    return MTD_SUCCESS;
    }

int mtd_comm_allreduce_large(mtd_rccl *r, void *d_buffer, size_t count, int elem, mtd_stream_t stream)
    {
    if (!r || (!d_buffer && count) || (elem != MTD_ELEM_F64 && elem != MTD_ELEM_U32)) return MTD_ERR_INVALID_ARGUMENT;
    if (count == 0) return MTD_SUCCESS;
    RcclApi &api = rccl_api();
    if (!api.ok) return MTD_ERR_UNSUPPORTED;
    const ncclDataType_t t = elem == MTD_ELEM_F64 ? ncclFloat64 : ncclUint32;
    const ncclResult_t res = api.AllReduce(d_buffer, d_buffer, count, t, ncclSum, r->comm, (hipStream_t)stream);
    if (res != ncclSuccess) return rccl_failed("ncclAllReduce", res);
    return MTD_SUCCESS;
    }

const char *mtd_rccl_last_error(void)
    {
    static thread_local char copy[256];
    std::lock_guard<std::mutex> lock(g_err_mutex);
    std::memcpy(copy, g_last_error, sizeof(copy));
    return copy;
    }

unsigned int mtd_rccl_world(const mtd_rccl *r) { return r ? r->world : 0; }
unsigned int mtd_rccl_rank(const mtd_rccl *r) { return r ? r->rank : 0; }

int mtd_rccl_destroy(mtd_rccl *r)
    {
    if (!r) return MTD_SUCCESS;
    RcclApi &api = rccl_api();
    if (api.ok) (void)api.CommDestroy(r->comm);
    delete r;
    return MTD_SUCCESS;
    }

} // extern "C"
