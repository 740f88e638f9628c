// Probe (GPU box): fp64 VALU issue rate vs dependent-instruction latency on gfx950, for a lone wave and for 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_latency tools/probe/valu_latency.hip && ./valu_latency
// Question behind it: an under-filled tet launch lasts as long as its slowest wave's dependent chain -- is a lone wave bound by
// instruction issue (4 cycles per wave64 fp64 instruction) or by the dependent-issue latency (then independent work could fill the gaps)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int CHAINS>
__global__ void fma_chain(double *out, int iters, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (CHAINS == 1) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));) REP8(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));) }
        if (CHAINS == 2) { REP8(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));) }
        if (CHAINS == 4) { REP8(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));) REP8(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x2), "+v"(x3) : "v"(a), "v"(b));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (double)(t1 - t0) * 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}
// op 0: v_rcp_f64 chain; 1: v_mul_f64 chain; 2: v_add_f64 chain; 3: v_cndmask pair chain; 4: compiler's division x = a / x; 5: compiler's sqrt
template <int OP>
__global__ void op_chain(double *out, int iters, double a, double b) {
    double x = 1.5 + threadIdx.x * 1e-3;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_rcp_f64 %0, %0" : "+v"(x));) }
        if (OP == 1) { REP8(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));) }
        if (OP == 2) { REP8(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));) }
        if (OP == 3) { REP8(asm volatile("v_mov_b64 %0, %0" : "+v"(x));) }
        if (OP == 4) { REP8(x = a / x; asm volatile("" : "+v"(x));) }
        if (OP == 5) { REP8(x = sqrt(x) + a; asm volatile("" : "+v"(x));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}

int main() {
    double *d; hipMalloc(&d, 8 * 1024 * 512);
    const int iters = 20000;
    struct Cfg { const char *name; int blocks, threads; } cfgs[] = {{"1 wave (lone)", 1, 64}, {"4 waves / CU (1 per SIMD)", 1, 256}, {"8 waves / CU (2 per SIMD)", 1, 512}, {"16 waves / CU (4 per SIMD)", 1, 1024}};
    for (auto &c : cfgs) {
        printf("%s\n", c.name);
        auto run = [&](const char *what, auto kern, int instr_per_iter) {
            double h = 0;
            hipLaunchKernelGGL(kern, dim3(c.blocks), dim3(c.threads), 0, 0, d, iters, 1.0000001, 1e-9);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(kern, dim3(c.blocks), dim3(c.threads), 0, 0, d, iters, 1.0000001, 1e-9);
            hipDeviceSynchronize();
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            const double ns = h * 10.0 / ((double)iters * instr_per_iter);      // 100 MHz counter
            printf("   %-38s %7.2f ns per instruction of one wave (%.1f cycles at 2.4 GHz)\n", what, ns, ns * 2.4);
        };
        run("v_fma_f64, 1 dependent chain", fma_chain<1>, 16);
        run("v_fma_f64, 2 independent chains", fma_chain<2>, 16);
        run("v_fma_f64, 4 independent chains", fma_chain<4>, 32);
        run("v_rcp_f64 dependent", op_chain<0>, 8);
        run("v_mul_f64 dependent", op_chain<1>, 8);
        run("v_add_f64 dependent", op_chain<2>, 8);
        run("v_mov_b64 dependent", op_chain<3>, 8);
        run("x = a / x (compiler's division)", op_chain<4>, 8);
        run("x = sqrt(x) + a (compiler's sqrt)", op_chain<5>, 8);
    }
    hipFree(d);
    return 0;
}
