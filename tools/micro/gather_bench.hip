// Microbenchmark: throughput of per-lane random 64-B node gathers from a small (L1/L2-resident)
// table on MI355X, (A) as the traversal does it today: every lane issues 4 dwordx4 loads to its own
// node; (B) quad-cooperative: in load i the four lanes of a quad fetch the four 16-B quarters of
// quad-lane i's node (same 64-B line -> one coalesced request), (C) LDS copy of the table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ nodes, int n_nodes, int iters, float* out) {
    extern __shared__ float4 lds[];
    if (MODE == 2) { for (int i = threadIdx.x; i < n_nodes * 4; i += 256) lds[i] = nodes[i]; __syncthreads(); }
    const unsigned lane = threadIdx.x & 63, gid = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    unsigned h = hash(gid * 2654435761u + 1);
    for (int it = 0; it < iters; ++it) {
        h = hash(h + it);
        const int idx = (int)(h % (unsigned)n_nodes);
        if (MODE == 0) {
            const float4 a = nodes[idx * 4 + 0], b = nodes[idx * 4 + 1], c = nodes[idx * 4 + 2], d = nodes[idx * 4 + 3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 1) {
            const int q = lane & 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx_i = __shfl(idx, (lane & ~3u) + i, 64);
                const float4 v = nodes[idx_i * 4 + q];
                acc += v.x + v.w;
            }
        } else {
            const float4 a = lds[idx * 4 + 0], b = lds[idx * 4 + 1], c = lds[idx * 4 + 2], d = lds[idx * 4 + 3];
            acc += a.x + b.y + c.z + d.w;
        }
    }
    out[gid] = acc;
}

int main() {
    const int n_nodes = 941, iters = 2000, blocks = 256 * 8;
    std::vector<float4> h(n_nodes * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_float4((float)i, 1.f, 2.f, 3.f);
    float4* d; float* o;
    hipMalloc(&d, h.size() * sizeof(float4)); hipMalloc(&o, blocks * 256 * sizeof(float));
    hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, iters, o);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, iters, o);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), n_nodes * 64, 0, d, n_nodes, iters, o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double lane_nodes = (double)blocks * 256 * iters;
            if (rep) std::printf("mode %d: %.3f ms  %.2f G lane-nodes/s  = %.3f lane-nodes/clk/CU (2.4 GHz, 256 CUs)\n", mode, ms, lane_nodes / ms / 1e6, lane_nodes / (ms * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
