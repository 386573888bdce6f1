// pk_sgpr_rate.hip -- does a packed fp32 instruction with an SGPR-pair operand cost more than one with VGPR operands?  (round 5: the box
// test's matrix products as v_pk_mul_f32 / v_pk_add_f32 with the matrix in scalar registers)   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk pk_sgpr_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float sa, float sb) {
    float2v a[8], x = {sa + threadIdx.x, sb}, y = {sb, sa};
    for (int i = 0; i < 8; ++i) a[i] = float2v{(float)i, (float)threadIdx.x};
    float2v s = {sa, sb};
    asm volatile("" : "+s"(s));
    float s0 = sa, s1 = sb;
    asm volatile("" : "+s"(s0), "+s"(s1));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
                if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == 4) asm volatile("v_mul_f32 %0, %2, %0\n\tv_mul_f32 %1, %3, %1" : "+v"(a[i].x), "+v"(a[i].y) : "s"(s0), "s"(s1));     // two H
                if (KIND == 5) asm volatile("v_mul_f32 %0, %2, %0\n\tv_add_f32 %1, %3, %1" : "+v"(a[i].x), "+v"(a[i].y) : "s"(s0), "v"(y.x));   // H + F
                if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %1, %0\n\tv_pk_add_f32 %0, %2, %0" : "+v"(a[i]) : "s"(s), "v"(y));                // pk H + pk F
                if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "s"(s), "v"(y));
            }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int KIND>
void run(const char *name, int perUnit) {
    float *out; hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(2048), dim3(256), 0, 0, out, 10, 1.0f, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(2048), dim3(256), 0, 0, out, iters, 1.0f, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 2048 workgroups x 4 waves on 1024 SIMDs = 8 waves per SIMD, each iters x 64 units
    const double unitsPerSimd = 8.0 * iters * 64, cycles = ms * 1e-3 * 2.1e9;
    printf("%-44s %.3f ms   %.2f cycles per unit per SIMD (at 2.1 GHz), %d vector instruction(s) per unit -> %.2f each\n", name, ms, cycles / unitsPerSimd, perUnit, cycles / unitsPerSimd / perUnit);
    hipFree(out);
}
int main() {
    run<0>("v_pk_mul_f32 v, v, v", 1);
    run<1>("v_pk_mul_f32 v, s[pair], v", 1);
    run<2>("v_pk_add_f32 v, v, v", 1);
    run<3>("v_pk_add_f32 v, s[pair], v", 1);
    run<4>("v_mul_f32 v,s,v ; v_mul_f32 v,s,v", 2);
    run<5>("v_mul_f32 v,s,v ; v_add_f32 v,v,v", 2);
    run<6>("v_pk_mul_f32 v,s[pair],v ; v_pk_add_f32 v,v,v", 2);
    run<7>("v_pk_fma_f32 v, s[pair], v, v", 1);
    return 0;
}
