// Stand-alone probe (GPU box): what bandwidth does the sweep kernels' ACCESS PATTERN reach when nothing else is done?
// A "panel" of F rows x K columns (column-major, ld = F); one block = one 64*(W/8)-row tile; the block's waves split the
// columns; every lane issues D loads of W bytes, two groups in flight, and sums what it reads.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int W, int D, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(const double *__restrict__ P, int F, int K, int tiles, double *out) {
    constexpr int WAVES = THREADS / 64, RPL = W / 8;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int panel = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const double *base = P + (size_t)panel * F * K + (size_t)tile * 64 * RPL + (size_t)lane * RPL;
    const int per = (K + WAVES - 1) / WAVES, jb = wave * per, je = min(jb + per, K);
    double acc = 0.0;
    if (RPL == 1) {
        double cur[D], nxt[D];
#pragma unroll
        for (int q = 0; q < D; ++q) cur[q] = (jb + q < je) ? base[(size_t)F * (jb + q)] : 0.0;
        for (int j = jb; j < je; j += D) {
#pragma unroll
            for (int q = 0; q < D; ++q) nxt[q] = (j + D + q < je) ? base[(size_t)F * (j + D + q)] : 0.0;
#pragma unroll
            for (int q = 0; q < D; ++q) acc += cur[q];
#pragma unroll
            for (int q = 0; q < D; ++q) cur[q] = nxt[q];
        }
    } else {
        double2 cur[D], nxt[D];
        const double2 z = {0.0, 0.0};
#pragma unroll
        for (int q = 0; q < D; ++q) cur[q] = (jb + q < je) ? *reinterpret_cast<const double2 *>(base + (size_t)F * (jb + q)) : z;
        for (int j = jb; j < je; j += D) {
#pragma unroll
            for (int q = 0; q < D; ++q) nxt[q] = (j + D + q < je) ? *reinterpret_cast<const double2 *>(base + (size_t)F * (j + D + q)) : z;
#pragma unroll
            for (int q = 0; q < D; ++q) acc += cur[q].x + cur[q].y;
#pragma unroll
            for (int q = 0; q < D; ++q) cur[q] = nxt[q];
        }
    }
    if (acc == 123.456) out[blockIdx.x] = acc;
}
template <int W, int D, int THREADS> void run(const char *name, const double *P, int F, int K, int panels, double *out) {
    const int tiles = F / (64 * (W / 8));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<W, D, THREADS>), dim3(panels * tiles), dim3(THREADS), 0, 0, P, F, K, tiles, out);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)panels * F * K * 8;
    printf("%-34s F %5d K %5d panels %4d blocks %6d : %7.1f us  %6.2f TB/s\n", name, F, K, panels, panels * tiles, ms * 1e3, bytes / ms / 1e9);
}
int main() {
    const size_t total = (size_t)96 << 20;   // doubles: 768 MB
    double *P, *out; hipMalloc(&P, total * 8); hipMalloc(&out, 1 << 20); hipMemset(P, 0, total * 8);
    struct { int F, K, panels; } cfg[] = {{128, 64, 2068}, {128, 64, 8272}, {2560, 704, 8}, {2560, 704, 32}, {1408, 320, 32}, {1408, 320, 128}, {1024, 1024, 1}, {2176, 1088, 2}};
    for (auto c : cfg) {
        if ((size_t)c.F * c.K * c.panels > total) continue;
        run<8, 8, 1024>("8B/lane D=8 1024thr", P, c.F, c.K, c.panels, out);
        run<8, 16, 1024>("8B/lane D=16 1024thr", P, c.F, c.K, c.panels, out);
        run<16, 4, 1024>("16B/lane D=4 1024thr", P, c.F, c.K, c.panels, out);
        run<16, 8, 1024>("16B/lane D=8 1024thr", P, c.F, c.K, c.panels, out);
        run<8, 8, 256>("8B/lane D=8 256thr", P, c.F, c.K, c.panels, out);
        run<16, 8, 256>("16B/lane D=8 256thr", P, c.F, c.K, c.panels, out);
    }
    return 0;
}
