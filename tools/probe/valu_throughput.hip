// Probe (GPU box): how does fp64 VALU throughput of the WHOLE chip scale with the waves per SIMD, per instruction class?
//   hipcc --offload-arch=gfx950 -O3 -o valu_throughput tools/probe/valu_throughput.hip && ./valu_throughput
// Every wave runs one dependent chain (what the tet kernel's waves do: tools/probe/valu_latency.hip shows a lone wave at one fp64
// instruction per 9.5-11 cycles).  Grid = 256 CUs x W blocks of 256 threads -> W waves per SIMD on every CU.  If the time of a launch
// stays flat while W grows, the SIMDs had idle issue slots; where it starts to grow in proportion, the issue rate is reached.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
// OP: 0 fma dependent | 1 fma, 2 independent chains | 2 compiler division x = a / x | 3 compiler sqrt | 4 v_rcp_f64 | 5 pair of v_cndmask_b32 on a double
//     6 v_mov_b64 | 7 v_cmp_lt_f64 + cndmask pair | 8 v_div_scale_f64 | 9 v_div_fixup_f64 | 10 v_ldexp_f64 | 11 v_max_f64
template <int OP>
__global__ __launch_bounds__(256) void chain(double *out, int iters, double a, double b) {
    double x = 1.5 + threadIdx.x * 1e-3, y = x + 1.0; unsigned xi = threadIdx.x, yi = 7;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));) }
        if (OP == 1) { REP8(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(a), "v"(b));) }
        if (OP == 2) { REP8(x = a / x; asm volatile("" : "+v"(x));) }
        if (OP == 3) { REP8(x = sqrt(x) + a; asm volatile("" : "+v"(x));) }
        if (OP == 4) { REP8(asm volatile("v_rcp_f64 %0, %0" : "+v"(x));) }
        if (OP == 5) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(xi) : "v"(yi) : );) }
        if (OP == 6) { REP8(asm volatile("v_mov_b64 %0, %0" : "+v"(x));) }
        if (OP == 7) { REP8(asm volatile("v_cmp_lt_f64 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(xi) : "v"(x), "v"(a), "v"(yi) : "vcc");) }
        if (OP == 8) { REP8(asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(a) : "vcc");) }
        if (OP == 9) { REP8(asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));) }
        if (OP == 10) { REP8(asm volatile("v_ldexp_f64 %0, %0, 0" : "+v"(x));) }
        if (OP == 11) { REP8(asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));) }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x + y + xi;
}

int main() {
    double *d; hipMalloc(&d, 8ull * 256 * 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char *names[] = {"v_fma_f64 dependent", "v_fma_f64 x2 independent", "a / x (compiler)", "sqrt(x)+a (compiler)", "v_rcp_f64", "v_cndmask_b32", "v_mov_b64", "v_cmp_lt_f64+cndmask", "v_div_scale_f64", "v_div_fixup_f64", "v_ldexp_f64", "v_max_f64"};
    const int per_iter[] = {8, 16, 8, 8, 8, 8, 8, 16, 8, 8, 8, 8};
    printf("%-26s", "ns per chain step, per wave");
    for (int W = 1; W <= 8; ++W) printf("   W=%d  ", W);
    printf("  | SIMD ns per step at W=8\n");
    for (int op = 0; op < 12; ++op) {
        printf("%-26s", names[op]);
        float last = 0;
        for (int W = 1; W <= 8; ++W) {
            auto launch = [&]() {
                dim3 g(256 * W), b(256);
                switch (op) {
                case 0: hipLaunchKernelGGL(chain<0>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break; case 1: hipLaunchKernelGGL(chain<1>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break;
                case 2: hipLaunchKernelGGL(chain<2>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break; case 3: hipLaunchKernelGGL(chain<3>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break;
                case 4: hipLaunchKernelGGL(chain<4>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break; case 5: hipLaunchKernelGGL(chain<5>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break;
                case 6: hipLaunchKernelGGL(chain<6>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break; case 7: hipLaunchKernelGGL(chain<7>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break;
                case 8: hipLaunchKernelGGL(chain<8>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break; case 9: hipLaunchKernelGGL(chain<9>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break;
                case 10: hipLaunchKernelGGL(chain<10>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break; default: hipLaunchKernelGGL(chain<11>, g, b, 0, 0, d, iters, 1.0000001, 1e-9); break;
                }
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            last = ms * 1e6f / ((float)iters * per_iter[op]);
            printf(" %7.2f", last);
        }
        printf("  | %6.2f  (%.1f cycles at 2.4 GHz)\n", last / 8, last / 8 * 2.4);
    }
    return 0;
}
