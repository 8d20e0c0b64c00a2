// Microbenchmark: what does a wave64 VALU instruction cost on gfx950 as a function of the EXEC
// mask?  Every wave runs the same unrolled loop of 8 independent dependent-chains under a lane
// mask; 16 waves per SIMD keep the VALU issue-bound.  Reports shader clocks per wave instruction
// (s_memtime) and the shader clock itself (s_memtime / s_memrealtime, the latter is 100 MHz).
// build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize exec_skip.hip -o exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) k(unsigned long long mask, int iters, float* out, unsigned long long* clk) {
    const unsigned lane = threadIdx.x & 63;
    float a[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; u[i] = threadIdx.x * 7 + i; }
    const float m = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (OP == 0) a[j] = __builtin_fmaf(a[j], m, c);
                    if (OP == 1) u[j] = u[j] * 3u + (unsigned)i;          // v_mad_u32_u24 / v_mul_lo + add
                    if (OP == 2) u[j] = (u[j] ^ (u[j] >> 3)) + 1u;        // v_lshrrev, v_xor, v_add (or v_xad)
                    if (OP == 3) a[j] = a[j] > 3.0f ? a[j] * m : a[j] + c; // cmp + cndmask mix
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

static unsigned long long low(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }

int main() {
    const int blocks = 256 * 16, iters = 2000;   // 16 blocks x 4 waves per CU = 16 waves per SIMD
    float* o; hipMalloc(&o, blocks * 256 * sizeof(float));
    unsigned long long* clk; hipMalloc(&clk, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct Case { const char* name; unsigned long long m; };
    std::vector<Case> cases = {{"all 64", ~0ull}, {"low 48", low(48)}, {"low 33", low(33)}, {"low 32", low(32)}, {"low 31", low(31)},
        {"low 28", low(28)}, {"low 24", low(24)}, {"low 20", low(20)}, {"low 17", low(17)}, {"low 16", low(16)}, {"low 8", low(8)},
        {"lane 0", 1ull}, {"every 2nd lane (32)", 0x5555555555555555ull}, {"every 4th lane (16)", 0x1111111111111111ull},
        {"8 per quarter (32)", 0x00FF00FF00FF00FFull}, {"7 per quarter (28)", 0x007F007F007F007Full},
        {"quarters 0+2 (32)", 0x0000FFFF0000FFFFull}, {"all 64 again", ~0ull}};
    const char* opn[] = {"v_fma_f32", "int mul-add", "shift/xor/add", "cmp+cndmask+mul/add"};
    for (int op = 0; op < 4; ++op) {
        printf("--- %s\n", opn[op]);
        float base = 0;
        for (auto& cs : cases) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, cs.m, iters, o, clk);
                if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, cs.m, iters, o, clk);
                if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, cs.m, iters, o, clk);
                if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, cs.m, iters, o, clk);
                hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            if (base == 0) base = ms;
            printf("%-24s %8.3f ms (%.2fx of full mask)  wave0: %llu shader clks, %llu x10ns => %.0f MHz\n", cs.name, ms, ms / base,
                   h[0], h[1], h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0);
        }
    }
    return 0;
}
