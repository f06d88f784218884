// micro-benchmark of gemm_f32 shapes used by the COCO caption GRUs (scratch tool)
#include "gemm_f32.h"
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
static float run(const F32Gemm& g, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) gemm_f32(g, 0);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) gemm_f32(g, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}
int main(int argc, char** argv) {
    int R = argc > 1 ? atoi(argv[1]) : 384;
    float *X, *W, *Y;
    hipMalloc(&X, (size_t)R * 102 * 600 * 4); hipMalloc(&W, 600 * 400 * 4); hipMalloc(&Y, (size_t)R * 102 * 600 * 4);
    hipMemset(X, 0, (size_t)R * 102 * 600 * 4); hipMemset(W, 0, 600 * 400 * 4); hipMemset(Y, 0, (size_t)R * 102 * 600 * 4);
    struct Case { const char* name; int M, N, K; long long a_rs, a_cs, b_rs, b_cs, ldc; int acc, ks; };
    Case cs[] = {
        {"fwd gh   x[R][200] W^T  N=600", R, 600, 200, 200, 1, 1, 200, 600, 0, 0},
        {"fwd gi0  x[R][300] W^T  N=600 K=300", R, 600, 300, 300, 1, 1, 400, 600, 0, 0},
        {"fwd out  N=300 K=200 strided C", R, 300, 200, 200, 1, 1, 300, 102 * 300, 0, 0},
        {"bwd dx   dy[R][600] W    N=200 K=600", R, 200, 600, 600, 1, 200, 1, 200, 1, 0},
        {"bwd dx   dy[R][600] W    N=300 K=600 (ldw 400)", R, 300, 600, 600, 1, 400, 1, 102 * 300, 1, 0},
        {"bwd dw   dy^T x  M=600 N=300 K=R", 600, 300, R, 1, 600, 300, 1, 400, 1, 0},
        {"bwd dw   dy^T x  M=300 N=200 K=R", 300, 200, R, 1, 102 * 300, 200, 1, 300, 1, 0},
        {"big dw   M=600 N=200 K=102R ks", 600, 200, 102 * R, 1, 600, 200, 1, 200, 1, 16},
    };
    for (auto& c : cs) {
        F32Gemm g{};
        g.A = X; g.a_rs = c.a_rs; g.a_cs = c.a_cs; g.B = (c.name[4] == 'd' && c.name[5] == 'w') || c.name[0] == 'b' && c.name[5] == 'w' ? Y : W;
        if (c.name[4] == 'd' && c.name[5] == 'w') g.B = Y;
        g.b_rs = c.b_rs; g.b_cs = c.b_cs; g.M = c.M; g.N = c.N; g.K = c.K; g.C = (c.name[4] == 'd' && c.name[5] == 'w') ? W : Y; g.ldc = c.ldc;
        g.accumulate = c.acc; g.ksplit = c.ks;
        float us = run(g, 50);
        printf("%-50s %8.1f us  %7.2f TFLOP/s\n", c.name, us, 2.0 * c.M * c.N * c.K / us * 1e-6);
    }
    return 0;
}
