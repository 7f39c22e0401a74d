// Issue cost of the instruction kinds the 8-lane kernel is made of, at one and at two wavefronts per SIMD (the headline's
// occupancy).  G blocks of one wavefront; every kind is eight independent dependent-chains interleaved, 16x unrolled.
// hipcc --offload-arch=gfx950 -O3 profiles/probe_op_rates.hip -o probe_op_rates && ./probe_op_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8F(S) asm volatile(S : "+v"(y0)); asm volatile(S : "+v"(y1)); asm volatile(S : "+v"(y2)); asm volatile(S : "+v"(y3)); \
                 asm volatile(S : "+v"(y4)); asm volatile(S : "+v"(y5)); asm volatile(S : "+v"(y6)); asm volatile(S : "+v"(y7));
#define REP8D(S) asm volatile(S : "+v"(z0)); asm volatile(S : "+v"(z1)); asm volatile(S : "+v"(z2)); asm volatile(S : "+v"(z3)); \
                 asm volatile(S : "+v"(z4)); asm volatile(S : "+v"(z5)); asm volatile(S : "+v"(z6)); asm volatile(S : "+v"(z7));
#define REP8DF(S) asm volatile(S : "+v"(z0) : "v"(y0)); asm volatile(S : "+v"(z1) : "v"(y1)); asm volatile(S : "+v"(z2) : "v"(y2)); asm volatile(S : "+v"(z3) : "v"(y3)); \
                  asm volatile(S : "+v"(z4) : "v"(y4)); asm volatile(S : "+v"(z5) : "v"(y5)); asm volatile(S : "+v"(z6) : "v"(y6)); asm volatile(S : "+v"(z7) : "v"(y7));
#define REP8FD(S) asm volatile(S : "+v"(y0) : "v"(z0)); asm volatile(S : "+v"(y1) : "v"(z1)); asm volatile(S : "+v"(y2) : "v"(z2)); asm volatile(S : "+v"(y3) : "v"(z3)); \
                  asm volatile(S : "+v"(y4) : "v"(z4)); asm volatile(S : "+v"(y5) : "v"(z5)); asm volatile(S : "+v"(y6) : "v"(z6)); asm volatile(S : "+v"(y7) : "v"(z7));
#define REP8FF(S) asm volatile(S : "+v"(y0) : "v"(y1)); asm volatile(S : "+v"(y1) : "v"(y2)); asm volatile(S : "+v"(y2) : "v"(y3)); asm volatile(S : "+v"(y3) : "v"(y4)); \
                  asm volatile(S : "+v"(y4) : "v"(y5)); asm volatile(S : "+v"(y5) : "v"(y6)); asm volatile(S : "+v"(y6) : "v"(y7)); asm volatile(S : "+v"(y7) : "v"(y0));
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int n) {
    unsigned y0 = threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3, y4 = y0 + 4, y5 = y0 + 5, y6 = y0 + 6, y7 = y0 + 7;
    double z0 = y0, z1 = y1, z2 = y2, z3 = y3, z4 = y4, z5 = y5, z6 = y6, z7 = y7;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) { REP8F("v_add_u32 %0, %0, %0") }
            if (KIND == 1) { REP8D("v_add_f64 %0, %0, %0") }
            if (KIND == 2) { REP8F("v_mul_lo_u32 %0, %0, %0") }
            if (KIND == 3) { REP8D("v_lshlrev_b64 %0, 1, %0") }
            if (KIND == 4) { REP8D("v_cmp_lt_f64 vcc, %0, %0") }
            if (KIND == 5) { REP8F("v_cndmask_b32 %0, %0, %0, vcc") }
            if (KIND == 6) { REP8F("v_readlane_b32 s20, %0, 3") }
            if (KIND == 7) { REP8D("v_fma_f64 %0, %0, %0, %0") }
            if (KIND == 8) { REP8D("v_mul_f64 %0, %0, %0") }
            if (KIND == 9) { REP8DF("v_cvt_f64_i32 %0, %1") }
            if (KIND == 10) { REP8DF("v_mad_u64_u32 %0, vcc, %1, %1, %0") }
            if (KIND == 11) { REP8F("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") }
            if (KIND == 12) { REP8F("v_cmp_lt_u32_e64 s[20:21], %0, %0") }
            if (KIND == 13) { REP8D("v_lshl_add_u64 %0, %0, 1, %0") }
            if (KIND == 14) { REP8F("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM,\"ppp00\")") asm volatile("s_waitcnt lgkmcnt(0)"); }
            if (KIND == 15) { REP8F("v_readfirstlane_b32 s20, %0") }
            if (KIND == 16) { REP8D("v_max_f64 %0, %0, %0") }
            if (KIND == 17) { REP8D("v_floor_f64 %0, %0") }
            if (KIND == 18) { REP8FD("v_cvt_i32_f64 %0, %1") }
            if (KIND == 19) { REP8F("v_bfe_u32 %0, %0, 1, 3") }
            if (KIND == 20) { REP8F("s_and_b64 s[20:21], s[20:21], exec") }
            if (KIND == 21) { REP8F("v_cmp_lt_f32 vcc, %0, %0") }
            if (KIND == 22) { REP8F("v_add_f32 %0, %0, %0") }
            if (KIND == 23) { REP8F("v_and_or_b32 %0, %0, %0, %0") }
            if (KIND == 24) { REP8D("v_lshrrev_b64 %0, 1, %0") }
            if (KIND == 25) { REP8F("v_mul_hi_u32 %0, %0, %0") }
            if (KIND == 26) { REP8F("v_mul_u32_u24 %0, %0, %0") }
            if (KIND == 27) { REP8D("v_cndmask_b32 %0, %0, %0, vcc") }   // half of a 64-bit select
            if (KIND == 28) { REP8F("v_cvt_f32_i32 %0, %0") }
            if (KIND == 29) { REP8F("v_mul_f32 %0, %0, %0") }
            if (KIND == 30) { REP8F("v_fma_f32 %0, %0, %0, %0") }
            if (KIND == 31) { REP8F("v_writelane_b32 %0, s20, 3") }
            if (KIND == 32) { REP8F("v_lshlrev_b32 %0, 1, %0") }
            if (KIND == 33) { REP8F("s_bfe_u32 s20, s20, 0x10003") }
            if (KIND == 34) { REP8F("v_mbcnt_lo_u32_b32 %0, %0, %0") }
            if (KIND == 35) { REP8F("v_add_co_u32 %0, vcc, %0, %0") }
            if (KIND == 36) { REP8F("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
            if (KIND == 37) { REP8F("v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf") }
            if (KIND == 38) { REP8F("v_min_u32 %0, %0, %0") }
            if (KIND == 39) { REP8F("v_med3_i32 %0, %0, %0, %0") }
            if (KIND == 40) { REP8F("v_cndmask_b32_e64 %0, %0, %0, s[20:21]") }
            if (KIND == 41) { REP8FF("v_cndmask_b32 %0, %0, %1, vcc") }
            if (KIND == 42) { REP8F("v_mov_b32 %0, %0") }
            if (KIND == 43) { REP8F("v_and_b32 %0, %0, %0") }
            if (KIND == 44) { REP8F("v_or_b32 %0, %0, %0") }
            if (KIND == 45) { REP8F("v_sub_u32 %0, %0, %0") }
            if (KIND == 46) { REP8F("v_xor_b32 %0, %0, %0") }
            if (KIND == 47) { REP8F("v_add3_u32 %0, %0, %0, %0") }
            if (KIND == 48) { REP8F("v_lshl_add_u32 %0, %0, 1, %0") }
            if (KIND == 49) { REP8F("v_lshl_or_b32 %0, %0, 1, %0") }
            if (KIND == 50) { REP8F("v_sub_f32 %0, %0, %0") }
            if (KIND == 51) { REP8F("v_fmac_f32 %0, %0, %0") }
            if (KIND == 52) { REP8F("v_max_f32 %0, %0, %0") }
            if (KIND == 53) { REP8F("v_mad_u32_u24 %0, %0, %0, %0") }
            if (KIND == 54) { REP8F("v_ashrrev_i32 %0, 1, %0") }
            if (KIND == 55) { REP8D("v_pk_add_f32 %0, %0, %0") }
            if (KIND == 56) { REP8F("v_add_u32 %0, s20, %0") }
            if (KIND == 57) { REP8F("v_add_u32 %0, 0x12345, %0") }
            if (KIND == 58) { REP8F("v_mul_f32 %0, 0x40490fdb, %0") }
            if (KIND == 59) { REP8F("v_subrev_u32 %0, %0, %0") }
            if (KIND == 60) { REP8F("v_add_u32_e64 %0, %0, %0") }
            if (KIND == 61) { REP8F("v_cmp_eq_u32 vcc, 3, %0") }
            if (KIND == 62) { REP8F("v_mul_i32_i24 %0, %0, %0") }
            if (KIND == 63) { REP8F("v_cvt_f32_u32 %0, %0") }
            if (KIND == 64) { REP8F("v_add_f32 %0, %0, %0\n v_lshlrev_b32 %0, 1, %0") }
            if (KIND == 65) { REP8F("v_add_f32 %0, %0, %0\n s_and_b64 s[20:21], s[20:21], exec") }
            if (KIND == 66) { REP8F("v_lshlrev_b32 %0, 1, %0\n s_and_b64 s[20:21], s[20:21], exec") }
            if (KIND == 68) { REP8F("v_cmp_lt_u32 vcc, %0, %0\n v_cndmask_b32 %0, %0, %0, vcc") }
            if (KIND == 69) { REP8F("v_cmp_lt_u32_e64 s[20:21], %0, %0\n v_cndmask_b32_e64 %0, %0, %0, s[20:21]") }
            if (KIND == 70) { REP8F("v_cndmask_b32_e64 %0, %0, %0, vcc") }
            if (KIND == 71) { asm volatile("s_mov_b64 vcc, exec"); REP8F("v_cndmask_b32 %0, %0, %0, vcc") }
            if (KIND == 72) { REP8F("v_addc_co_u32 %0, vcc, %0, %0, vcc") }
            if (KIND == 73) { REP8F("v_cmp_lt_u32 vcc, %0, %0\n s_nop 0\n v_cndmask_b32 %0, %0, %0, vcc") }
            if (KIND == 74) { REP8F("v_cmp_lt_u32 vcc, %0, %0\n v_cndmask_b32_e64 %0, %0, %0, vcc") }
            if (KIND == 75) { REP8F("v_cndmask_b32 %0, 0, %0, vcc") }
            if (KIND == 76) { REP8F("v_cndmask_b32_e64 %0, %0, 1, s[20:21]") }
            if (KIND == 77) { REP8F("v_cmp_lt_u32 vcc, %0, %0\n v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %0, %0, %0, vcc") }
            if (KIND == 78) { REP8F("v_cmp_lt_u32 vcc, %0, %0\n v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %0, %0, %0, vcc") }
            if (KIND == 79) { REP8F("v_cndmask_b32 %0, %0, %0, vcc\n v_add_u32 %0, %0, %0") }
            if (KIND == 80) { REP8FF("v_cmp_lt_u32 vcc, %0, %0\n v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %1, %1, %1, vcc") }
            if (KIND == 81) { REP8F("v_cmp_lt_u32_e64 s[20:21], %0, %0\n v_cndmask_b32_e64 %0, %0, %0, s[20:21]\n v_cndmask_b32_e64 %0, %0, %0, s[20:21]") }
            if (KIND == 82) { REP8F("v_cndmask_b32 %0, %0, %0, vcc\n s_nop 0") }
            if (KIND == 83) { REP8F("v_cndmask_b32 %0, %0, %0, vcc\n s_and_b64 s[20:21], s[20:21], exec") }
            if (KIND == 67) { REP8D("v_add_f64 %0, %0, %0\n s_and_b64 s[20:21], s[20:21], exec") }
        }
    }
    if (y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7 + (float)(z0 + z1 + z2 + z3 + z4 + z5 + z6 + z7) == 1234.5f) out[0] = 1;
}
template <int KIND> void run(const char* name, float* d) {
    const int n = 500;
    double r[2]; int gi = 0;
    for (int G : {1024, 2048}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(G), dim3(64), 0, 0, d, n); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(G), dim3(64), 0, 0, d, n); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        r[gi++] = ms * 1e-3 * 2.4e9 / (n * 128.0);
    }
    printf("%-22s 1 wave/SIMD %6.2f clk per instruction    2 waves/SIMD %6.2f clk per instruction per wave\n", name, r[0], r[1]);
}
int main(int argc, char** argv) {
    float* d; hipMalloc(&d, 64);
    if (argc > 1) {   // second sheet: select forms, plain integer / logic kinds, literal and scalar operands, pairs (cost per PAIR)
        run<40>("v_cndmask_e64 sgpr", d); run<5>("v_cndmask_b32 vcc a,a", d); run<41>("v_cndmask_b32 vcc a,b", d);
        run<42>("v_mov_b32", d); run<43>("v_and_b32", d); run<44>("v_or_b32", d); run<46>("v_xor_b32", d); run<45>("v_sub_u32", d); run<59>("v_subrev_u32", d);
        run<0>("v_add_u32", d); run<60>("v_add_u32_e64", d); run<56>("v_add_u32 sgpr", d); run<57>("v_add_u32 literal", d); run<47>("v_add3_u32", d);
        run<48>("v_lshl_add_u32", d); run<49>("v_lshl_or_b32", d); run<54>("v_ashrrev_i32", d); run<53>("v_mad_u32_u24", d); run<62>("v_mul_i32_i24", d);
        run<22>("v_add_f32", d); run<50>("v_sub_f32", d); run<51>("v_fmac_f32", d); run<52>("v_max_f32", d); run<58>("v_mul_f32 literal", d); run<55>("v_pk_add_f32", d); run<63>("v_cvt_f32_u32", d);
        run<61>("v_cmp_eq_u32 vcc const", d);
        run<68>("pair cmp vcc + cndmask vcc", d); run<69>("pair cmp sgpr + cndmask_e64 sgpr", d); run<70>("v_cndmask_e64 vcc", d); run<71>("v_cndmask vcc after s_mov vcc", d);
        run<72>("v_addc_co_u32 vcc", d); run<73>("cmp vcc; s_nop; cndmask vcc", d); run<74>("pair cmp vcc + cndmask_e64 vcc", d); run<75>("v_cndmask vcc const", d); run<76>("v_cndmask_e64 const sgpr", d);
        run<77>("triple cmp+cnd+cnd vcc", d); run<78>("quint cmp+4 cnd vcc", d); run<79>("pair cnd vcc + add_u32", d); run<80>("triple cmp+cnd+cnd(other reg)", d); run<81>("triple cmp+2cnd_e64 sgpr", d);
        run<82>("pair cnd vcc + s_nop", d); run<83>("pair cnd vcc + s_and", d);
        run<64>("pair add_f32+lshl", d); run<65>("pair add_f32+s_and", d); run<66>("pair lshl+s_and", d); run<67>("pair add_f64+s_and", d);
        return 0;
    }
    run<0>("v_add_u32", d); run<22>("v_add_f32", d); run<29>("v_mul_f32", d); run<30>("v_fma_f32", d); run<32>("v_lshlrev_b32", d); run<19>("v_bfe_u32", d); run<23>("v_and_or_b32", d);
    run<38>("v_min_u32", d); run<39>("v_med3_i32", d); run<34>("v_mbcnt_lo", d);
    run<5>("v_cndmask_b32", d); run<21>("v_cmp_lt_f32 vcc", d); run<12>("v_cmp_lt_u32 sgpr", d); run<35>("v_add_co_u32", d);
    run<2>("v_mul_lo_u32", d); run<25>("v_mul_hi_u32", d); run<26>("v_mul_u32_u24", d); run<10>("v_mad_u64_u32", d);
    run<3>("v_lshlrev_b64", d); run<24>("v_lshrrev_b64", d); run<13>("v_lshl_add_u64", d);
    run<1>("v_add_f64", d); run<8>("v_mul_f64", d); run<7>("v_fma_f64", d); run<16>("v_max_f64", d); run<4>("v_cmp_lt_f64", d); run<17>("v_floor_f64", d);
    run<9>("v_cvt_f64_i32", d); run<18>("v_cvt_i32_f64", d); run<28>("v_cvt_f32_i32", d);
    run<11>("v_mov_dpp row_shr", d); run<36>("v_mov_dpp quad_perm", d); run<37>("v_mov_dpp row_bcast15", d);
    run<6>("v_readlane_b32", d); run<15>("v_readfirstlane_b32", d); run<31>("v_writelane_b32", d);
    run<14>("ds_swizzle_b32 (+wait per 8)", d); run<20>("s_and_b64", d); run<33>("s_bfe_u32", d);
    return 0;
}
