// Terrain-geometry kernels either side of the tracker ("next" rows of the scope table): signed distance of point sets to a
// heightfield seen as a grid of axis-aligned columns.
//
// terrain_util.points_hf_sdf  util/terrain_util.py:1835-1893 of the reference (+ points_boxes_sdf :1774-1804,
// geom_util.sdBox / sdRoundBox  util/geom_util.py:113-143): every heightfield cell (i, j) is a box with centre
// (x_i + cx, y_j + cy) and half extents (dx/2, dy/2); vertically it spans [base_z, hf] - or, "inverted", the AIR column
// [hf, -base_z] above the cell, so that after the final sign flip a point inside the ground gets its (negative) depth to the
// nearest free surface.  The reference materialises [B, N, M, 3] tensors and takes the min over M; here a workgroup keeps a
// tile of cells (centre + vertical half extent, 16 B each) in LDS, every thread owns one point and walks the tile with
// broadcast LDS reads: no temporaries, one pass over the cells per 256 points.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/parc_hip.h"

#define SDF_THREADS 256
#define SDF_TILE 2048        // cells per LDS tile (32 KB)

__global__ __launch_bounds__(SDF_THREADS) void points_hf_sdf_kernel(int n_points, int dim_x, int dim_y, const float *__restrict__ points,
                                                                    const float *__restrict__ hf, const float *__restrict__ min_box_center,
                                                                    const float *__restrict__ x_points, const float *__restrict__ y_points,
                                                                    float half_x, float half_y, float base_z, int inverted, float radius,
                                                                    float *__restrict__ out, int32_t *__restrict__ out_cell) {
    __shared__ float4 cells[SDF_TILE];
    const int bi = blockIdx.y;
    const int p = blockIdx.x * SDF_THREADS + threadIdx.x;
    const int pc = min(p, n_points - 1);                       // tail threads recompute the last point, they only skip the store
    const float *pt = points + ((size_t)bi * n_points + pc) * 3;
    const float px = pt[0], py = pt[1], pz = pt[2];
    const int M = dim_x * dim_y;
    const float *hfb = hf + (size_t)bi * M;
    const float ox = min_box_center[2 * bi], oy = min_box_center[2 * bi + 1];
    const float top_z = -base_z;
    float best = INFINITY;
    int best_cell = 0;
    for (int t0 = 0; t0 < M; t0 += SDF_TILE) {
        const int cnt = min(SDF_TILE, M - t0);
        for (int c = threadIdx.x; c < cnt; c += SDF_THREADS) {
            const int cell = t0 + c, i = cell / dim_y, j = cell - i * dim_y;
            const float h = hfb[cell];
            // box centre / vertical half extent, same fp32 operations as the reference (:1862-1871)
            const float cz = inverted ? (h + top_z) / 2.0f : (h + base_z) / 2.0f;
            const float hz = inverted ? (top_z - h) / 2.0f : (h - base_z) / 2.0f;
            cells[c] = make_float4(x_points[i] + ox, y_points[j] + oy, cz, hz);
        }
        __syncthreads();
#pragma unroll 4
        for (int c = 0; c < cnt; ++c) {
            const float4 cl = cells[c];
            // sdBox: q = |p - centre| - half extents; |max(q, 0)| + min(max(q.x, q.y, q.z), 0)
            const float qx = fabsf(px - cl.x) - half_x, qy = fabsf(py - cl.y) - half_y, qz = fabsf(pz - cl.z) - cl.w;
            const float ax = fmaxf(qx, 0.f), ay = fmaxf(qy, 0.f), az = fmaxf(qz, 0.f);
            const float outside = __fsqrt_rn(ax * ax + ay * ay + az * az);
            const float inside = fminf(fmaxf(qx, fmaxf(qy, qz)), 0.f);
            const float sd = outside + inside;
            if (sd < best) {             // first column that attains the minimum
                best = sd;
                best_cell = t0 + c;
            }
        }
        __syncthreads();
    }
    if (radius > 0.f) best -= radius;           // sdRoundBox: x - r is monotone, so it commutes with the min
    if (inverted) best = -best;
    if (p < n_points) {
        out[(size_t)bi * n_points + p] = best;
        if (out_cell) out_cell[(size_t)bi * n_points + p] = best_cell;
    }
}

extern "C" int parc_points_hf_sdf(void *stream, int batch, int n_points, int dim_x, int dim_y, const float *points, const float *hf,
                                  const float *min_box_center, const float *x_points, const float *y_points, float half_x, float half_y,
                                  float base_z, int inverted, float radius, float *out, int32_t *out_cell) {
    if (batch < 0 || n_points < 0 || dim_x <= 0 || dim_y <= 0 || (int64_t)dim_x * dim_y > (int64_t)1 << 30) return PARC_EINVAL;
    if (batch == 0 || n_points == 0) return PARC_OK;
    if (batch > 65535) return PARC_EUNSUPPORTED;
    if (!points || !hf || !min_box_center || !x_points || !y_points || !out) return PARC_EINVAL;
    hipLaunchKernelGGL(points_hf_sdf_kernel, dim3((n_points + SDF_THREADS - 1) / SDF_THREADS, batch), dim3(SDF_THREADS), 0, (hipStream_t)stream,
                       n_points, dim_x, dim_y, points, hf, min_box_center, x_points, y_points, half_x, half_y, base_z, inverted, radius, out, out_cell);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}
