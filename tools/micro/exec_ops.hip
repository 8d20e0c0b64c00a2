// Microbenchmark: shader clocks per wave64 VALU instruction on gfx950, per opcode and per number
// of active lanes (EXEC = low n lanes).  16 waves per SIMD, 8 independent chains per wave.
// build: hipcc -O3 --offload-arch=gfx950 exec_ops.hip -o exec_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(OPSTR)                                                                                         \
    asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)                      \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                 : "v"(m), "v"(c) : "vcc")

#define OP_FMA(i)  "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MUL(i)  "v_mul_f32 %" #i ", %" #i ", %8\n"
#define OP_ADD(i)  "v_add_f32 %" #i ", %" #i ", %9\n"
#define OP_MAX(i)  "v_max_f32 %" #i ", %" #i ", %9\n"
#define OP_CMP(i)  "v_cmp_lt_f32 vcc, %" #i ", %9\n"
#define OP_CND(i)  "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define OP_MOV(i)  "v_mov_b32 %" #i ", %9\n"
#define OP_ADDU(i) "v_add_u32 %" #i ", %" #i ", %9\n"
#define OP_AND(i)  "v_and_b32 %" #i ", %" #i ", %9\n"
#define OP_LSH(i)  "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define OP_RCP(i)  "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %9\n"
#define OP_SUB(i)  "v_sub_f32 %" #i ", %" #i ", %9\n"
#define OP_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"

template <int OP>
__global__ void __launch_bounds__(256) k(unsigned long long mask, int iters, float* out) {
    const unsigned lane = threadIdx.x & 63;
    float a0 = threadIdx.x + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0001f, c = 0.5f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (OP == 0) CHAIN8(OP_FMA);
                if (OP == 1) CHAIN8(OP_MUL);
                if (OP == 2) CHAIN8(OP_ADD);
                if (OP == 3) CHAIN8(OP_MAX);
                if (OP == 4) CHAIN8(OP_CMP);
                if (OP == 5) CHAIN8(OP_CND);
                if (OP == 6) CHAIN8(OP_MOV);
                if (OP == 7) CHAIN8(OP_ADDU);
                if (OP == 8) CHAIN8(OP_AND);
                if (OP == 9) CHAIN8(OP_LSH);
                if (OP == 10) CHAIN8(OP_RCP);
                if (OP == 11) CHAIN8(OP_MULLO);
                if (OP == 12) CHAIN8(OP_SUB);
                if (OP == 13) CHAIN8(OP_MIN3);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

static unsigned long long low(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }

template <int OP>
void run(const char* name, float* o) {
    const int blocks = 256 * 16, iters = 1000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int ns[] = {64, 33, 32, 17, 16, 9, 8, 1};
    printf("%-14s", name);
    for (int n : ns) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, low(n), iters, o);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        const double insts_per_simd = (double)blocks * 4 / 1024 * iters * 64;
        printf(" %6.2f", ms * 1e-3 * 2.4e9 / insts_per_simd);
    }
    printf("\n");
}

int main() {
    float* o; hipMalloc(&o, 256 * 16 * 256 * sizeof(float));
    printf("clocks (at 2.4 GHz) per wave instruction per SIMD; columns = active lanes\n%-14s %6d %6d %6d %6d %6d %6d %6d %6d\n", "op", 64, 33, 32, 17, 16, 9, 8, 1);
    run<0>("v_fma_f32", o); run<1>("v_mul_f32", o); run<2>("v_add_f32", o); run<12>("v_sub_f32", o); run<3>("v_max_f32", o);
    run<13>("v_min3_f32", o); run<4>("v_cmp_lt_f32", o); run<5>("v_cndmask_b32", o); run<6>("v_mov_b32", o); run<7>("v_add_u32", o);
    run<8>("v_and_b32", o); run<9>("v_lshlrev_b32", o); run<10>("v_rcp_f32", o); run<11>("v_mul_lo_u32", o);
    return 0;
}
