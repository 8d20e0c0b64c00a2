// Microbenchmark: shader clocks per wave64 VALU instruction on gfx950, per opcode and per number
// of active lanes (EXEC = low n lanes).  16 waves per SIMD, 8 independent chains per wave.
// build: hipcc -O3 --offload-arch=gfx950 exec_ops.hip -o exec_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(OPSTR)                                                                                         \
    asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)                      \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                 : "v"(m), "v"(c) : "vcc", "s10", "s11", "s12", "v20", "v21", "v22", "v23", "v24", "v25")

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
#define OP_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[10:11]\n"
#define OP_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 6, %9\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP_CMPS(i) "v_cmp_le_f32_e64 s[10:11], %" #i ", %9\n"
#define OP_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_XOR(i)  "v_xor_b32 %" #i ", %" #i ", %9\n"
#define OP_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define OP_FMAS(i) "v_fma_f32 %" #i ", %" #i ", s12, %9\n"
#define OP_MULL(i) "v_mul_f32 %" #i ", 0x3f800004, %" #i "\n"
#define OP_MIN(i)  "v_min_f32 %" #i ", %" #i ", %9\n"
#define OP_SUBU(i) "v_sub_u32 %" #i ", %" #i ", %9\n"
#define OP_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define OP_PKFMA(i) "v_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[24:25]\n"
#define OP_PKMUL(i) "v_pk_mul_f32 v[20:21], v[20:21], v[22:23]\n"
#define OP_FMA64(i) "v_fma_f64 v[20:21], v[20:21], v[22:23], v[24:25]\n"
#define OP_MUL64(i) "v_mul_f64 v[20:21], v[20:21], v[22:23]\n"
#define OP_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define OP_BITOP(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0xcf\n"
#define OP_CVT(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define OP_MAD64(i) "v_mad_u64_u32 v[20:21], s[10:11], %" #i ", %8, v[20:21]\n"
#define OP_CNDV(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define OP_CND64V(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %9, vcc\n"
#define OP_CNDSDWA(i) "v_cndmask_b32_sdwa %" #i ", %" #i ", %9, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
#define OP_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %9\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_CMPGAP(i) "v_cmp_lt_f32 vcc, %" #i ", %9\nv_add_f32 v20, v20, %9\nv_add_f32 v21, v21, %9\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_SMOVCND(i) "s_mov_b64 vcc, s[10:11]\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_CMPCND2(i) "v_cmp_lt_f32 vcc, %" #i ", %9\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\nv_cndmask_b32 v20, v20, %8, vcc\n"
#define OP_CMPCND64(i) "v_cmp_lt_f32_e64 s[10:11], %" #i ", %9\nv_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"

#define OP_CVTUB(i) "v_cvt_f32_ubyte1_e32 %" #i ", %" #i "\n"
#define OP_MIXLO(i) "v_fma_mix_f32 %" #i ", %" #i ", %8, %9 op_sel_hi:[1,0,0]\n"
#define OP_MIXHI(i) "v_fma_mix_f32 %" #i ", %" #i ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define OP_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define OP_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_MINU(i) "v_min_u32 %" #i ", %" #i ", %9\n"
#define OP_MED3U(i) "v_med3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_MIN3U(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", %9\n"
#define OP_LSHSDWA(i) "v_lshlrev_b32_sdwa %" #i ", %9, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define OP_CVTF16(i) "v_cvt_f32_f16 %" #i ", %" #i "\n"
#define OP_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 8, %9\n"
#define OP_CMPU(i) "v_cmp_lt_u32 vcc, %" #i ", %9\n"
#define OP_CMPADDC(i) "v_cmp_lt_f32 vcc, %" #i ", %9\nv_addc_co_u32 v20, vcc, 0, v20, vcc\n"
#define OP_SANDCND(i) "v_cmp_lt_f32 vcc, %" #i ", %9\ns_and_b64 vcc, vcc, s[10:11]\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_SANDCND64(i) "v_cmp_lt_f32 vcc, %" #i ", %9\ns_and_b64 s[10:11], vcc, s[10:11]\nv_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define OP_CVTU32(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define OP_PKFMAF16(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define OP_DSW(i) "ds_write_b32 %8, %" #i "\n"

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
                if (OP == 14) CHAIN8(OP_CNDS);
                if (OP == 15) CHAIN8(OP_LSHLADD);
                if (OP == 16) CHAIN8(OP_ADD3);
                if (OP == 17) CHAIN8(OP_MAD24);
                if (OP == 18) CHAIN8(OP_CMPS);
                if (OP == 19) CHAIN8(OP_MED3);
                if (OP == 20) CHAIN8(OP_XOR);
                if (OP == 21) CHAIN8(OP_FMAC);
                if (OP == 22) CHAIN8(OP_FMAS);
                if (OP == 23) CHAIN8(OP_MULL);
                if (OP == 24) CHAIN8(OP_MIN);
                if (OP == 25) CHAIN8(OP_SUBU);
                if (OP == 26) CHAIN8(OP_LSHR);
                if (OP == 27) CHAIN8(OP_PKFMA);
                if (OP == 28) CHAIN8(OP_PKMUL);
                if (OP == 29) CHAIN8(OP_FMA64);
                if (OP == 30) CHAIN8(OP_MUL64);
                if (OP == 31) CHAIN8(OP_SQRT);
                if (OP == 32) CHAIN8(OP_BITOP);
                if (OP == 33) CHAIN8(OP_CVT);
                if (OP == 34) CHAIN8(OP_MAD64);
                if (OP == 35) CHAIN8(OP_CNDV);
                if (OP == 36) CHAIN8(OP_CND64V);
                if (OP == 37) CHAIN8(OP_CNDSDWA);
                if (OP == 38) CHAIN8(OP_CMPCND);
                if (OP == 39) CHAIN8(OP_CMPCND64);
                if (OP == 40) CHAIN8(OP_CMPGAP);
                if (OP == 41) CHAIN8(OP_SMOVCND);
                if (OP == 42) CHAIN8(OP_CMPCND2);
                if (OP == 43) CHAIN8(OP_CVTUB);
                if (OP == 44) CHAIN8(OP_MIXLO);
                if (OP == 45) CHAIN8(OP_MIXHI);
                if (OP == 46) CHAIN8(OP_ANDOR);
                if (OP == 47) CHAIN8(OP_BFE);
                if (OP == 48) CHAIN8(OP_PERM);
                if (OP == 49) CHAIN8(OP_MINU);
                if (OP == 50) CHAIN8(OP_MED3U);
                if (OP == 51) CHAIN8(OP_MIN3U);
                if (OP == 52) CHAIN8(OP_LDEXP);
                if (OP == 53) CHAIN8(OP_LSHSDWA);
                if (OP == 54) CHAIN8(OP_CVTF16);
                if (OP == 55) CHAIN8(OP_MAX3);
                if (OP == 56) CHAIN8(OP_LSHLOR);
                if (OP == 57) CHAIN8(OP_CMPU);
                if (OP == 58) CHAIN8(OP_CMPADDC);
                if (OP == 59) CHAIN8(OP_SANDCND);
                if (OP == 60) CHAIN8(OP_SANDCND64);
                if (OP == 61) CHAIN8(OP_CVTU32);
                if (OP == 62) CHAIN8(OP_PKFMAF16);
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

template <int OP>
void run_masks(const char* name, float* o) {      // the same instruction stream under differently PLACED sets of active lanes
    const int blocks = 256 * 16, iters = 1000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned long long ms_[] = {0xffull, 0xff00000000000000ull, 0x0101010101010101ull, 0x000f000f000f000full, 0x1ffull, 0x0101010101010103ull,
                                      0xffffull, 0x1111111111111111ull, 0x00ff00ff00000000ull, 0xffffffffull, 0xffffffff00000000ull, 0x5555555555555555ull};
    printf("%-14s", name);
    for (unsigned long long mk : ms_) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, mk, iters, o);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        const double insts_per_simd = (double)blocks * 4 / 1024 * iters * 64;
        printf(" %6.2f", ms * 1e-3 * 2.4e9 / insts_per_simd);
    }
    printf("\n");
}

template <int OP>
void run_occ(const char* name, float* o) {      // waves per SIMD 1 / 2 / 4 / 8, 8 active lanes against 64: a per-wave issue interval or SIMD time?
    const int iters = 1000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("%-14s", name);
    for (unsigned long long mk : {0xffull, ~0ull})
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = 256 * wps;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, mk, iters, o);
                hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            printf(" %6.2f", ms * 1e-3 * 2.4e9 / ((double)iters * 64));      // clocks per instruction of ONE wave's stream
        }
    printf("\n");
}

int main() {
    float* o; hipMalloc(&o, 256 * 16 * 256 * sizeof(float));
    printf("clocks (at 2.4 GHz) per wave instruction per SIMD; columns = active lanes\n%-14s %6d %6d %6d %6d %6d %6d %6d %6d\n", "op", 64, 33, 32, 17, 16, 9, 8, 1);
    run<0>("v_fma_f32", o); run<1>("v_mul_f32", o); run<2>("v_add_f32", o); run<12>("v_sub_f32", o); run<3>("v_max_f32", o);
    run<13>("v_min3_f32", o); run<4>("v_cmp_lt_f32", o); run<5>("v_cndmask_b32", o); run<6>("v_mov_b32", o); run<7>("v_add_u32", o);
    run<8>("v_and_b32", o); run<9>("v_lshlrev_b32", o); run<10>("v_rcp_f32", o); run<11>("v_mul_lo_u32", o);
    run<14>("cndmask sgpr", o); run<15>("v_lshl_add_u32", o); run<16>("v_add3_u32", o); run<17>("v_mad_u32_u24", o); run<18>("v_cmp e64 sgpr", o);
    run<19>("v_med3_f32", o); run<20>("v_xor_b32", o); run<21>("v_fmac_f32", o); run<22>("v_fma sgpr op", o); run<23>("v_mul literal", o);
    run<24>("v_min_f32", o); run<25>("v_sub_u32", o); run<26>("v_lshrrev_b32", o); run<27>("v_pk_fma_f32", o); run<28>("v_pk_mul_f32", o);
    run<29>("v_fma_f64", o); run<30>("v_mul_f64", o); run<31>("v_sqrt_f32", o); run<32>("v_bitop3_b32", o); run<33>("v_cvt_f32_i32", o);
    run<34>("v_mad_u64_u32", o); run<35>("cndmask indep", o);
    run<36>("cndmask e64 vcc", o); run<37>("cndmask sdwa", o); run<38>("cmp+cnd vcc x2", o); run<39>("cmp+cnd sgpr x2", o);
    run<40>("cmp,add,add,cnd", o); run<41>("s_mov vcc+cnd", o); run<42>("cmp,cnd,cnd", o);
    run<43>("v_cvt_f32_ubyte", o); run<44>("fma_mix lo", o); run<45>("fma_mix hi", o); run<46>("v_and_or_b32", o); run<47>("v_bfe_u32", o); run<48>("v_perm_b32", o); run<49>("v_min_u32", o); run<50>("v_med3_u32", o); run<51>("v_min3_u32", o); run<52>("v_ldexp_f32", o); run<53>("lshl sdwa byte", o); run<54>("v_cvt_f32_f16", o); run<55>("v_max3_f32", o); run<56>("v_lshl_or_b32", o); run<57>("v_cmp_lt_u32", o); run<58>("cmp+addc x2", o); run<59>("cmp,s_and,cnd x2", o); run<60>("cmp,s_and,cnd64", o); run<61>("v_cvt_f32_u32", o); run<62>("v_pk_fma_f16", o);
    printf("\nthe same under placed lane sets: 8 low | 8 high | 8 spread (1 per 8) | 4x4 spread | 9 low | 9 spread | 16 low | 16 spread (1 per 4) | 16 in upper half | 32 low | 32 high | 32 alternate\n");
    run_masks<0>("v_fma_f32", o); run_masks<3>("v_max_f32", o); run_masks<4>("v_cmp_lt_f32", o); run_masks<14>("cndmask sgpr", o); run_masks<43>("v_cvt_f32_ubyte", o);
    run_masks<19>("v_med3_f32", o); run_masks<44>("fma_mix lo", o); run_masks<2>("v_add_f32", o); run_masks<10>("v_rcp_f32", o);
    printf("\nclocks per instruction of one wave's stream (elapsed / instructions per wave); columns: 8 active lanes at 1 / 2 / 4 / 8 waves per SIMD, then 64 active lanes at 1 / 2 / 4 / 8\n");
    run_occ<0>("v_fma_f32", o); run_occ<3>("v_max_f32", o); run_occ<4>("v_cmp_lt_f32", o); run_occ<14>("cndmask sgpr", o); run_occ<2>("v_add_f32", o); run_occ<10>("v_rcp_f32", o);
    return 0;
}
