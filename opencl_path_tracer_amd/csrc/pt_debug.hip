// pt_debug.hip -- test entry point of libptamd.so: closest hit of caller-supplied rays through the
// same traversal code (pt_device.hpp) and the same node placement the render kernels use.
#include "pt_device.hpp"

#include <algorithm>

namespace ptamd {

// persistent blocks, grid-stride over the rays; one ray per lane at a time
template <int MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_debug_closest_hit(RenderParams p, const pt_ray* rays, long long n, float* out_t, int* out_tri) {
    LaneStack<typename StackOf<MODE>::type> stk;
    SceneView sv;
    setup_traversal<MODE, BLOCK>(p, &sv, &stk);
    WorkCount wc;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK) {
        const float4* r = reinterpret_cast<const float4*>(&rays[i]);
        const float4 a = r[0], b = r[1];
        float t;
        const int ti = closest_hit<MODE, false>(sv, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), stk, &t, &wc);
        out_t[i] = ti >= 0 ? t : -1.0f;
        out_tri[i] = ti;
    }
}

template <int MODE, int BLOCK>
static hipError_t launch_debug_t(const RenderParams& p, const pt_ray* rays, int64_t n, float* out_t, int32_t* out_tri, int cu_count, hipStream_t stream) {
    const size_t lds = traversal_lds_bytes(p, BLOCK);
    auto kern = k_debug_closest_hit<MODE, BLOCK>;
    static LdsMark mark;
    const hipError_t e = ensure_dynamic_lds((const void*)kern, mark, lds);
    if (e != hipSuccess) return e;
    const long long need = (n + BLOCK - 1) / BLOCK;
    const int blocks = (int)std::min<long long>(need, (long long)cu_count * (2048 / BLOCK));
    if (p.stack_ovf && (long long)blocks * BLOCK > (long long)p.stack_ovf_lanes) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), lds, stream, p, rays, (long long)n, out_t, out_tri);
    return hipGetLastError();
}

hipError_t launch_debug_closest_hit(const RenderParams& p, const pt_ray* rays, int64_t n, float* out_t, int32_t* out_tri, int cu_count, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    switch (p.node_mode) {
    case kNodesLds: return launch_debug_t<kNodesLds, 512>(p, rays, n, out_t, out_tri, cu_count, stream);
    case kNodesGlobal: return launch_debug_t<kNodesGlobal, 256>(p, rays, n, out_t, out_tri, cu_count, stream);
    case kNodesWide: return launch_debug_t<kNodesWide, 256>(p, rays, n, out_t, out_tri, cu_count, stream);
    case kNodesTreelet: return launch_debug_t<kNodesTreelet, 1024>(p, rays, n, out_t, out_tri, cu_count, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace ptamd
