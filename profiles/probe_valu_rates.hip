// whole-chip timing: a VALU-only stream per wave, G blocks of one wave each; time vs G tells how waves on a SIMD share issue
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int n) {
    float y0 = threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3, y4 = y0 + 4, y5 = y0 + 5, y6 = y0 + 6, y7 = y0 + 7;
    double z0 = y0, z1 = y1, z2 = y2, z3 = y3, z4 = y4, z5 = y5, z6 = y6, z7 = y7;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) { asm volatile("v_add_f32 %0, %0, %0" : "+v"(y0)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(y1)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(y2)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(y3));
                             asm volatile("v_add_f32 %0, %0, %0" : "+v"(y4)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(y5)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(y6)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(y7)); }
            if (KIND == 1) { asm volatile("v_add_f64 %0, %0, %0" : "+v"(z0)); asm volatile("v_add_f64 %0, %0, %0" : "+v"(z1)); asm volatile("v_add_f64 %0, %0, %0" : "+v"(z2)); asm volatile("v_add_f64 %0, %0, %0" : "+v"(z3));
                             asm volatile("v_add_f64 %0, %0, %0" : "+v"(z4)); asm volatile("v_add_f64 %0, %0, %0" : "+v"(z5)); asm volatile("v_add_f64 %0, %0, %0" : "+v"(z6)); asm volatile("v_add_f64 %0, %0, %0" : "+v"(z7)); }
            if (KIND == 2) { asm volatile("v_add_f32 %0, %0, %0\n s_add_u32 s20, s20, 1" : "+v"(y0)); asm volatile("v_add_f32 %0, %0, %0\n s_add_u32 s20, s20, 1" : "+v"(y1)); asm volatile("v_add_f32 %0, %0, %0\n s_add_u32 s20, s20, 1" : "+v"(y2)); asm volatile("v_add_f32 %0, %0, %0\n s_add_u32 s20, s20, 1" : "+v"(y3));
                             asm volatile("v_add_f32 %0, %0, %0\n s_add_u32 s20, s20, 1" : "+v"(y4)); asm volatile("v_add_f32 %0, %0, %0\n s_add_u32 s20, s20, 1" : "+v"(y5)); asm volatile("v_add_f32 %0, %0, %0\n s_add_u32 s20, s20, 1" : "+v"(y6)); asm volatile("v_add_f32 %0, %0, %0\n s_add_u32 s20, s20, 1" : "+v"(y7)); }
        }
    }
    if (y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7 + (float)(z0 + z1 + z2 + z3 + z4 + z5 + z6 + z7) == 1234.5f) out[0] = 1;
}
template <int KIND> void run(const char* name, float* d) {
    const int n = 2000;  // 2000 * 128 instructions per wave
    for (int G : {256, 1024, 2048, 4096, 8192}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(G), dim3(64), 0, 0, d, n); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(G), dim3(64), 0, 0, d, n); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = 2000.0 * 128.0 * (KIND == 2 ? 2 : 1);
        printf("%-10s blocks %5d (%.1f waves/SIMD): %.3f ms -> %.2f clk@2.4GHz per instruction per wave\n", name, G, G / 1024.0, ms, ms * 1e-3 * 2.4e9 / instr);
    }
}
int main() { float* d; hipMalloc(&d, 64); run<0>("add_f32", d); run<1>("add_f64", d); run<2>("valu+salu", d); return 0; }
