// Calibration of rocprofv3's FETCH_SIZE (and TCC_MISS / TCC_EA0_RDREQ) on gfx950 for the access pattern of the BVH traversal:
// every lane reads one random 64-byte record with four 16-byte loads (a 4-wide node; a triangle packet is three of them), from
// tables below and above the 256-MiB Infinity Cache.  Known: how many records each kernel reads.  Next to it the pattern the
// guide's rule was measured on (16 B per lane, coalesced stream) as the control.
// build: hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib ; run under rocprofv3 --kernel-trace --pmc <set>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// ID only gives every (pattern, table size) its own kernel name in the trace
template <int ID>
__global__ void __launch_bounds__(256) gather64(const float4* __restrict__ t, unsigned n_rec, int iters, float* out) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    unsigned h = hash(gid * 2654435761u + 12345u);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        h = hash(h + (unsigned)it);
        const size_t r = (size_t)(h % n_rec) * 4;
        const float4 a = t[r], b = t[r + 1], c = t[r + 2], d = t[r + 3];
        acc += a.x + b.y + c.z + d.w;
    }
    out[gid] = acc;
}
template <int ID>
__global__ void __launch_bounds__(256) stream16(const float4* __restrict__ t, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += t[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int ID>
static void run_gather(const float4* d, size_t mib, float* o) {
    const unsigned n_rec = (unsigned)(mib * 1024 * 1024 / 64);
    const int blocks = 256 * 16, iters = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {          // rep 0 warms the caches; the profile is read for the last launch of each name
        hipEventRecord(a);
        hipLaunchKernelGGL(gather64<ID>, dim3(blocks), dim3(256), 0, 0, d, n_rec, iters, o);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    const double recs = (double)blocks * 256 * iters;
    std::printf("gather64<%d> table %5zu MiB: %10.0f records (%8.1f MB) per launch, %7.3f ms, %7.1f GB/s of records\n", ID, mib, recs, recs * 64 / 1e6, ms, recs * 64 / ms / 1e6);
}

int main() {
    const size_t max_mib = 4096;
    float4* d; float* o;
    if (hipMalloc(&d, max_mib << 20) != hipSuccess || hipMalloc(&o, 256 * 16 * 256 * sizeof(float)) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, max_mib << 20);
    hipDeviceSynchronize();
    run_gather<16>(d, 16, o); run_gather<64>(d, 64, o); run_gather<192>(d, 192, o); run_gather<1024>(d, 1024, o); run_gather<4096>(d, 4096, o);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(stream16<1>, dim3(256 * 8), dim3(256), 0, 0, d, (size_t)(2048ull << 20) / 16, o);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    std::printf("stream16<1> 2048 MiB coalesced: %.3f ms, %.1f GB/s\n", ms, 2048.0 * 1.048576 / ms);
    return 0;
}
