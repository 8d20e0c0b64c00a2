// pt_comm.hip -- multi-GPU exchange behind the C ABI (SURVEY 8b "RCCL communicator per context", 8e).
//
// The path shards with no data-path collective: every rank renders its own interleaved row blocks
// (pt_create_tiled).  The ONE exchange is the assembly of the frame: an all-gather of the ranks'
// radiance slabs over RCCL/xGMI (a gather, not an arithmetic reduce: the tiles are disjoint, and an
// all-gather moves 1/N of the frame per link where a reduce of zero-padded frames would move
// 2(N-1)/N) followed by a de-interleave kernel.  Both run on the context's stream.
//
// librccl is bound at run time (dlopen), so that a single-GPU host needs no RCCL and a process that
// already carries an RCCL (e.g. PyTorch's) shares that one instead of loading a second copy.
#include "pt_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

namespace ptamd {

int fail_ctx(pt_context* ctx, int code, const std::string& msg);   // pt_host.cpp

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

void bind_rccl(RcclApi& api) {
    const char* resident[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : resident)
        if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy this process already uses
    const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : fresh)
        if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!api.handle) {
        api.error = std::string("librccl not found: ") + dlerror();
        return;
    }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString)
        api.error = "librccl lacks ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclAllGather/ncclGetErrorString";
}

RcclApi* rccl() {          // bound once, whichever host thread (one per context) asks first
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] { bind_rccl(api); });
    return &api;
}

}  // namespace

// frame[gid] = gathered[owner(gid) * slab_stride + local_index(gid)] -- the inverse of the tiling of
// pt_create_tiled: row y belongs to rank (y / rb) % world and is that rank's local row
// (y / rb / world) * rb + y % rb.  One thread per pixel of the GLOBAL frame; reads are contiguous per
// row segment, writes fully coalesced.
__global__ void __launch_bounds__(256) k_deinterleave(const float4* __restrict__ gathered, float4* __restrict__ frame,
                                                     int W, int H, int world, int rb, long long slab_stride) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)W * H) return;
    const int y = (int)(gid / W), x = (int)(gid - (long long)y * W);
    const int blk = y / rb;
    const int owner = blk % world;
    const int lrow = (blk / world) * rb + (y - blk * rb);
    frame[gid] = gathered[(long long)owner * slab_stride + (long long)lrow * W + x];
}

hipError_t launch_deinterleave(const float4* gathered, float4* frame, int W, int H, int world, int rb, long long slab_stride, hipStream_t stream) {
    const long long n = (long long)W * H;
    hipLaunchKernelGGL(k_deinterleave, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, gathered, frame, W, H, world, rb, slab_stride);
    return hipGetLastError();
}

// the host-side statement of the same map (tests; pt_gather_index)
void gather_source_index(int W, int H, int world, int rb, long long slab_stride, int64_t* out) {
    for (int y = 0; y < H; ++y) {
        const int blk = y / rb, owner = blk % world, lrow = (blk / world) * rb + (y - blk * rb);
        for (int x = 0; x < W; ++x) out[(size_t)y * W + x] = (int64_t)owner * slab_stride + (int64_t)lrow * W + x;
    }
}

int comm_available(std::string* err) {
    RcclApi* r = rccl();
    if (!r->error.empty()) { *err = r->error; return PT_ECOMM; }
    return PT_OK;
}

int comm_unique_id(void* id128, std::string* err) {
    RcclApi* r = rccl();
    if (!r->error.empty()) { *err = r->error; return PT_ECOMM; }
    ncclUniqueId id;
    const ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) { *err = std::string("ncclGetUniqueId: ") + r->GetErrorString(rc); return PT_ECOMM; }
    static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "PT_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    std::memcpy(id128, &id, sizeof id);
    return PT_OK;
}

int comm_init(const void* id128, int rank, int world, void** comm_out, std::string* err) {
    RcclApi* r = rccl();
    if (!r->error.empty()) { *err = r->error; return PT_ECOMM; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r->CommInitRank(&comm, world, id, rank);
    if (rc != ncclSuccess) { *err = std::string("ncclCommInitRank: ") + r->GetErrorString(rc); return PT_ECOMM; }
    *comm_out = comm;
    return PT_OK;
}

void comm_destroy(void* comm) {
    RcclApi* r = rccl();
    if (comm && r->error.empty()) (void)r->CommDestroy((ncclComm_t)comm);
}

int comm_all_gather(void* comm, const void* send, void* recv, size_t floats_per_rank, hipStream_t stream, std::string* err) {
    RcclApi* r = rccl();
    if (!r->error.empty()) { *err = r->error; return PT_ECOMM; }
    const ncclResult_t rc = r->AllGather(send, recv, floats_per_rank, ncclFloat, (ncclComm_t)comm, stream);
    if (rc != ncclSuccess) { *err = std::string("ncclAllGather: ") + r->GetErrorString(rc); return PT_ECOMM; }
    return PT_OK;
}

}  // namespace ptamd
