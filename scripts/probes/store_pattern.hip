// Store-pattern probe (round 5): what does the row-segment shape of a GEMM epilogue's stores cost?  Every workgroup (512
// threads, one per CU and round) writes one 256 x 320 bf16 tile of a row-major [M, N] matrix -- the gemm256 epilogue's 164 KB
// -- with 16-byte stores whose 64 lanes cover, per instruction,
//   mode 0:  1.6 rows x 640 B  (whole tile rows: a cross-wave slab would allow this)
//   mode 1:  6.4 rows x 160 B  (a wave's own 80 columns: the per-wave slab epilogue of gemm256)
//   mode 2:  16 rows x  64 B   (the lane-exchange epilogue that was measured 3 .. 10 % slower)
//   mode 3:  3.2 rows x 320 B  (two waves' columns)
// Usage: store_pattern <mode> <M> <N> <reps>; prints us per pass over the matrix and TB/s.  Nothing is computed: the stored
// values are a function of the lane only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void store_tile(unsigned short* c, int M, int N, int nbn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tm = blockIdx.x / nbn, tn = blockIdx.x % nbn;
    const int m0 = tm * 256, n0 = tn * 320;
    const u32x4 val = {threadIdx.x * 2654435761u, threadIdx.x * 40503u, threadIdx.x, ~threadIdx.x};
    // units of 8 columns; the tile has 256 x 40 of them; a wave stores 1280 units = 20 instructions
    for (int it = 0; it < 20; ++it) {
        int row, col;
        if (MODE == 0) {                 // whole rows: wave w takes rows [32 w, 32 w + 32); unit u = it * 64 + lane
            const int u = it * 64 + lane;
            row = 32 * wave + u / 40; col = (u % 40) * 8;
        } else if (MODE == 1) {          // the wave's own 80 columns of its group's 128 rows
            const int grp = wave >> 2, wc = wave & 3, u = it * 64 + lane;
            row = 128 * grp + u / 10; col = wc * 80 + (u % 10) * 8;
        } else if (MODE == 2) {          // 16 rows x 64 B: lane (g, r) = row r, 8 columns; the fifth 16-column tile pairs two row blocks
            const int grp = wave >> 2, wc = wave & 3, g = lane >> 4, r = lane & 15;
            if (it < 16) { row = 128 * grp + (it & 7) * 16 + r; col = wc * 80 + (it >> 3) * 32 + g * 8; }
            else { row = 128 * grp + (it - 16) * 32 + (g & 1) * 16 + r; col = wc * 80 + 64 + (g >> 1) * 8; }
        } else {                         // two waves' columns: 160 columns, 20 lanes per row
            const int grp = wave >> 2, wp = (wave & 3) >> 1, half = wave & 1, u = it * 64 + lane;
            row = 128 * grp + 64 * half + u / 20; col = wp * 160 + (u % 20) * 8;
        }
        const int m = m0 + row, n = n0 + col;
        if (m < M && n < N) *reinterpret_cast<u32x4*>(c + (size_t)m * N + n) = val;
    }
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1, M = argc > 2 ? atoi(argv[2]) : 8192, N = argc > 3 ? atoi(argv[3]) : 11200;
    const int reps = argc > 4 ? atoi(argv[4]) : 200;
    unsigned short* c;
    if (hipMalloc(&c, (size_t)M * N * 2) != hipSuccess) return 1;
    const int nbm = (M + 255) / 256, nbn = (N + 319) / 320;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        switch (mode) {
            case 0: hipLaunchKernelGGL(store_tile<0>, dim3(nbm * nbn), dim3(512), 0, 0, c, M, N, nbn); break;
            case 1: hipLaunchKernelGGL(store_tile<1>, dim3(nbm * nbn), dim3(512), 0, 0, c, M, N, nbn); break;
            case 2: hipLaunchKernelGGL(store_tile<2>, dim3(nbm * nbn), dim3(512), 0, 0, c, M, N, nbn); break;
            default: hipLaunchKernelGGL(store_tile<3>, dim3(nbm * nbn), dim3(512), 0, 0, c, M, N, nbn); break;
        }
    };
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)M * N * 2;
    printf("mode %d  %d x %d: %8.1f us per pass, %6.2f TB/s, %6.2f us per tile round (%d tiles)\n", mode, M, N, us, bytes / us / 1e6,
           us / ((nbm * nbn + 255) / 256), nbm * nbn);
    return 0;
}
