// How many workgroups of W waves at V VGPRs does a CU of MI355X hold at once?  (placement of a workgroup's waves over the 4 SIMDs)
// Every wave spins for a fixed number of s_sleep periods, so a grid that is resident in one round takes one period and a grid that
// needs two rounds takes two.  hipcc --offload-arch=gfx950 -O3 wg_residency.hip -o wg_residency && ./wg_residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int THREADS, int MINW, int NREG>
__global__ __launch_bounds__(THREADS, MINW) void spin(float *out, int iters) {
    float r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[i]));
        __builtin_amdgcn_s_sleep(100);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) s += r[i];
    if (s == 12345.678f) out[0] = s;
}

template <int THREADS, int MINW, int NREG>
void run(const char *name, float *d) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int grid : {256, 512, 768, 1024, 1280, 2048}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            spin<THREADS, MINW, NREG><<<grid, THREADS>>>(d, 40);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        printf("%s grid %4d: %.1f us\n", name, grid, best * 1e3f);
    }
}

int main() {
    float *d;
    hipMalloc(&d, 4);
    run<512, 8, 48>("512 threads, 56 VGPRs ", d);
    run<320, 5, 80>("320 threads, 88 VGPRs ", d);
    run<512, 4, 100>("512 threads, 104 VGPRs", d);
    run<384, 6, 64>("384 threads, 72 VGPRs ", d);
    run<384, 6, 75>("384 threads, 80 VGPRs ", d);
    run<320, 5, 64>("320 threads, 72 VGPRs ", d);
    run<320, 5, 75>("320 threads, 80 VGPRs ", d);
    run<320, 5, 56>("320 threads, 64 VGPRs ", d);
    run<448, 7, 64>("448 threads, 72 VGPRs ", d);
    return 0;
}
