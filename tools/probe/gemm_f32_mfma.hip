// Probe (stretch item 9): a hand-written fp32 MFMA GEMM for the backward data GEMM of the policy / value MLP,
//     dZ_prev[M, N] = (dZ[M, K] x W[K, N]) * (act[M, N] > 0),   db_prev[N] = column sums of the result
// (learning/nets/fc_3layers_2048units.py backward; today: hipBLASLt GEMM + parc_relu_bwd_bias_grad as a separate pass), against the
// library GEMM + pass on the same shapes.  All row-major fp32; M % 128 == N % 128 == 0, K % 16 == 0.
// Workgroup 256 threads = 2 x 2 waves on a 128 x 128 tile, a wave = 2 x 2 v_mfma_f32_32x32x2f32 tiles (64 accumulator registers),
// k-block 16, A kept transposed in LDS ([k][m]) so that both operands are read as consecutive dwords, global -> register prefetch of the
// next k-block under the MFMAs of the current one, LDS double-buffered: one barrier per k-block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f16v __attribute__((ext_vector_type(16)));
#ifndef ABL
#define ABL 0
#endif
#define TM 128
#define TN 128
#define TK 16
#ifdef MI16
#define LDT (TM + 16)   // 16x16x4: lanes 0-31 read k = 0, 1 x 16 rows -> banks 0-15, 16-31
#else
#define LDT (TM + 4)
#endif

template <bool FUSE>
__global__ __launch_bounds__(256, 2) void gemm_nn_kernel(int M, int N, int K, const float *__restrict__ A, const float *__restrict__ B,
                                                          const float *__restrict__ act, float *__restrict__ C, float *__restrict__ colpart) {
    __shared__ __attribute__((aligned(16))) float As[2][TK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][TK][LDT];
    __shared__ float cs[2][TN];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
        // workgroup i runs on XCD i % 8 (round-robin dispatch): with the n tile fastest an XCD keeps seeing the same N / 128 / 8 panels of B
    // (resident in its 4 MB L2) and streams every panel of A once
    const int tiles_n = N / TN;
    const int bn = blockIdx.x % tiles_n, bm = blockIdx.x / tiles_n;
    const int m0 = bm * TM, n0 = bn * TN;
    // global -> register mapping: A tile 128 x 16 = 512 float4 (row = i / 4, kq = i % 4), B tile 16 x 128 = 512 float4 (krow = i / 32, c4 = i % 32)
    // global -> register mapping: A tile 128 x 16 = 512 float4 (row = i / 4, kq = i % 4), B tile 16 x 128 = 512 float4 (krow = i / 32, c4 = i % 32)
    const int ia0 = t, ia1 = t + 256;
    const float *Ap0 = A + (size_t)(m0 + ia0 / 4) * K + 4 * (ia0 % 4), *Ap1 = A + (size_t)(m0 + ia1 / 4) * K + 4 * (ia1 % 4);
    const float *Bp0 = B + (size_t)(ia0 / 32) * N + n0 + 4 * (ia0 % 32), *Bp1 = B + (size_t)(ia1 / 32) * N + n0 + 4 * (ia1 % 32);
    float4 ra0 = *(const float4 *)Ap0, ra1 = *(const float4 *)Ap1, rb0 = *(const float4 *)Bp0, rb1 = *(const float4 *)Bp1;
#ifdef MI16
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
#else
    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#endif
    auto stage = [&](int buf) {
        As[buf][4 * (ia0 % 4) + 0][ia0 / 4] = ra0.x, As[buf][4 * (ia0 % 4) + 1][ia0 / 4] = ra0.y, As[buf][4 * (ia0 % 4) + 2][ia0 / 4] = ra0.z,
        As[buf][4 * (ia0 % 4) + 3][ia0 / 4] = ra0.w;
        As[buf][4 * (ia1 % 4) + 0][ia1 / 4] = ra1.x, As[buf][4 * (ia1 % 4) + 1][ia1 / 4] = ra1.y, As[buf][4 * (ia1 % 4) + 2][ia1 / 4] = ra1.z,
        As[buf][4 * (ia1 % 4) + 3][ia1 / 4] = ra1.w;
        *(float4 *)&Bs[buf][ia0 / 32][4 * (ia0 % 32)] = rb0;
        *(float4 *)&Bs[buf][ia1 / 32][4 * (ia1 % 32)] = rb1;
    };
    stage(0);
    __syncthreads();
    const int nkb = K / TK;
    const int am = wm * 64 + (lane & 31), bnn = wn * 64 + (lane & 31), kh = lane >> 5;
    // one k-block: all 8 operand pairs of the block are read first (LDS latency once per block, not once per k-step), then 32 MFMAs
#ifdef MI16
    // v_mfma_f32_16x16x4f32: A lane l = (row l % 16, k l / 16), B lane l = (k l / 16, col l % 16); a k-step is 4 deep
    const int l16 = lane & 15, k4 = lane >> 4;
    auto block = [&](int buf) {
#pragma unroll
        for (int h = 0; h < TK / 8; ++h) {                  // two k-steps (8 operand pairs) at a time
            float a[2][4], b[2][4];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 4 * (2 * h + s2) + k4;
                    a[s2][i] = As[buf][k][wm * 64 + 16 * i + l16];
                    b[s2][i] = Bs[buf][k][wn * 64 + 16 * i + l16];
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s2][i], b[s2][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#else
    auto block = [&](int buf) {
#pragma unroll
        for (int h = 0; h < TK / 8; ++h) {                  // groups of 4 k-steps: 16 operand registers live
            float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int k = 2 * (s2 + 4 * h) + kh;
#if ABL == 1                                        // (timing ablation: no operand reads)
                a0[s2] = 1.0f + k, a1[s2] = 2.0f, b0[s2] = 3.0f, b1[s2] = 4.0f + k;
#else
                a0[s2] = As[buf][k][am], a1[s2] = As[buf][k][am + 32];
                b0[s2] = Bs[buf][k][bnn], b1[s2] = Bs[buf][k][bnn + 32];
#endif
            }
            __builtin_amdgcn_sched_barrier(0);              // (the scheduler otherwise sinks each read back in front of its MFMAs)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s2], b0[s2], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s2], b1[s2], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s2], b0[s2], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s2], b1[s2], acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#endif
    auto prefetch = [&](int kb) {
        const size_t ko = (size_t)kb * TK;
        ra0 = *(const float4 *)(Ap0 + ko), ra1 = *(const float4 *)(Ap1 + ko);
        rb0 = *(const float4 *)(Bp0 + ko * N), rb1 = *(const float4 *)(Bp1 + ko * N);
        __builtin_amdgcn_sched_barrier(0);                  // (... and the global loads down to the LDS stores that consume them)
    };
    // two k-blocks per trip so that the LDS buffer is a compile-time constant (nkb even)
    for (int kb = 0; kb < nkb; kb += 2) {
#if ABL == 2                                                // (timing ablation: no global loads, no staging, no barriers)
        block(0);
        block(1);
#else
        prefetch(kb + 1);
        block(0);
        stage(1);
        __syncthreads();
        if (kb + 2 < nkb) prefetch(kb + 2);
        block(1);
        if (kb + 2 < nkb) {
            stage(0);
            __syncthreads();
        }
#endif
    }
#ifdef MI16
    // epilogue: acc[i][j][e] -> row = 16 i + 4 (lane / 16) + e, col = 16 j + lane % 16 of the wave's 64 x 64 tile
    float colsum4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + wm * 64 + 16 * i + 4 * k4 + e, col = n0 + wn * 64 + 16 * j + l16;
                float v = acc[i][j][e];
                if (FUSE) {
                    v = act[(size_t)row * N + col] > 0.f ? v : 0.f;
                    colsum4[j] += v;
                }
                C[(size_t)row * N + col] = v;
            }
    if (FUSE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            colsum4[j] += __shfl_xor(colsum4[j], 16);
            colsum4[j] += __shfl_xor(colsum4[j], 32);
            if (k4 == 0 && wm == 0) cs[0][wn * 64 + 16 * j + l16] = colsum4[j];
        }
        __syncthreads();
        if (wm == 1 && k4 == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = wn * 64 + 16 * j + l16;
                colpart[(size_t)bm * N + n0 + c] = cs[0][c] + colsum4[j];
            }
        }
    }
}
#else
    // epilogue: acc[i][j][e] -> row = 32 i + 8 (e / 4) + 4 (lane / 32) + e % 4, col = 32 j + lane % 32 of the wave's 64 x 64 tile
    float colsum[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + 32 * i + 8 * (e / 4) + 4 * kh + (e % 4), col = n0 + wn * 64 + 32 * j + (lane & 31);
                float v = acc[i][j][e];
                if (FUSE) {
                    v = act[(size_t)row * N + col] > 0.f ? v : 0.f;
                    colsum[j] += v;
                }
                C[(size_t)row * N + col] = v;
            }
    if (FUSE) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            colsum[j] += __shfl_xor(colsum[j], 32);
            if (kh == 0 && wm == 0) cs[0][wn * 64 + 32 * j + (lane & 31)] = colsum[j];
        }
        __syncthreads();
        if (wm == 1 && kh == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = wn * 64 + 32 * j + (lane & 31);
                colpart[(size_t)bm * N + n0 + c] = cs[0][c] + colsum[j];      // fixed order: rows 0..63 + rows 64..127 of the tile
            }
        }
    }
}
#endif

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 1024;
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N), hAct((size_t)M * N), hC((size_t)M * N), hP((size_t)(M / TM) * N);
    srand(1);
    for (auto &v : hA) v = frand();
    for (auto &v : hB) v = frand();
    for (auto &v : hAct) v = frand();
    float *dA, *dB, *dAct, *dC, *dP;
    hipMalloc(&dA, hA.size() * 4), hipMalloc(&dB, hB.size() * 4), hipMalloc(&dAct, hAct.size() * 4), hipMalloc(&dC, hC.size() * 4), hipMalloc(&dP, hP.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice), hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dAct, hAct.data(), hAct.size() * 4, hipMemcpyHostToDevice);
    const dim3 grid((M / TM) * (N / TN)), block(256);
    for (int fuse = 0; fuse < 2; ++fuse) {
        auto launch = [&]() {
            if (fuse) gemm_nn_kernel<true><<<grid, block>>>(M, N, K, dA, dB, dAct, dC, dP);
            else gemm_nn_kernel<false><<<grid, block>>>(M, N, K, dA, dB, dAct, dC, dP);
        };
        launch();
        hipDeviceSynchronize();
        hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hP.data(), dP, hP.size() * 4, hipMemcpyDeviceToHost);
        // spot check against a double-precision dot product
        double worst = 0.0;
        for (int s = 0; s < 2000; ++s) {
            const int r = rand() % M, c = rand() % N;
            double ref = 0.0;
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hB[(size_t)k * N + c];
            if (fuse && !(hAct[(size_t)r * N + c] > 0.f)) ref = 0.0;
            worst = fmax(worst, fabs(ref - hC[(size_t)r * N + c]));
        }
        double worst_cs = 0.0;
        if (fuse)
            for (int s = 0; s < 50; ++s) {
                const int c = rand() % N;
                double ref = 0.0, got = 0.0;
                for (int r = 0; r < M; ++r) ref += hC[(size_t)r * N + c];
                for (int b = 0; b < M / TM; ++b) got += hP[(size_t)b * N + c];
                worst_cs = fmax(worst_cs, fabs(ref - got));
            }
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = fminf(best, ms / 10.f);
        }
        printf("{\"kernel\": \"gemm_nn_f32_mfma%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"us\": %.1f, \"TFLOPs\": %.1f, \"max_abs_err_vs_f64\": %.3g, \"colsum_err\": %.3g}\n",
               fuse ? " + relu mask + column partials" : "", M, N, K, best * 1e3f, 2.0 * M * N * K / (best * 1e-3) / 1e12, worst, worst_cs);
    }
    return 0;
}
