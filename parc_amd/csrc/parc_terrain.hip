// Terrain-geometry kernels either side of the tracker ("next" rows of the scope table): signed distance of point sets to a
// heightfield seen as a grid of axis-aligned columns.
//
// terrain_util.points_hf_sdf  util/terrain_util.py:1835-1893 of the reference (+ points_boxes_sdf :1774-1804,
// geom_util.sdBox / sdRoundBox  util/geom_util.py:113-143): every heightfield cell (i, j) is a box with centre
// (x_i + cx, y_j + cy) and half extents (dx/2, dy/2); vertically it spans [base_z, hf] - or, "inverted", the AIR column
// [hf, -base_z] above the cell, so that after the final sign flip a point inside the ground gets its (negative) depth to the
// nearest free surface.  The reference materialises [B, N, M, 3] tensors and takes the min over M; here every thread owns one point
// and scans only the window of columns that can hold the minimum (see the kernel): no temporaries, exact result.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/parc_hip.h"

#define SDF_THREADS 64

// sdBox of one column: q = |p - centre| - half extents; |max(q, 0)| + min(max(q.x, q.y, q.z), 0), same fp32 operations as the
// reference (:1862-1871)
__device__ __forceinline__ float column_sd(float px, float py, float pz, float cx, float cy, float h, float half_x, float half_y, float base_z,
                                           float top_z, int inverted) {
    const float cz = inverted ? (h + top_z) / 2.0f : (h + base_z) / 2.0f;
    const float hz = inverted ? (top_z - h) / 2.0f : (h - base_z) / 2.0f;
    const float qx = fabsf(px - cx) - half_x, qy = fabsf(py - cy) - half_y, qz = fabsf(pz - cz) - hz;
    const float ax = fmaxf(qx, 0.f), ay = fmaxf(qy, 0.f), az = fmaxf(qz, 0.f);
    return __fsqrt_rn(ax * ax + ay * ay + az * az) + fminf(fmaxf(qx, fmaxf(qy, qz)), 0.f);
}

// One thread per point.  The minimum over ALL columns is found exactly without visiting all of them: the distance d0 to the column under
// the point bounds the answer, and a column more than R = floor(d0 / cell) + 2 cells away in x or in y is further than d0 in that
// coordinate alone (columns do not overlap in xy), so only the (2R+1)^2 window is scanned - in the same cell order as a full scan, so
// "first column that attains the minimum" is the same column.  Heights come straight from L2 (a heightfield is a few hundred KB).
__global__ __launch_bounds__(SDF_THREADS) void points_hf_sdf_kernel(int n_points, int dim_x, int dim_y, const float *__restrict__ points,
                                                                    const float *__restrict__ hf, const float *__restrict__ min_box_center,
                                                                    const float *__restrict__ x_points, const float *__restrict__ y_points,
                                                                    float half_x, float half_y, float base_z, int inverted, float radius,
                                                                    float *__restrict__ out, int32_t *__restrict__ out_cell) {
    const int bi = blockIdx.y;
    const int p = blockIdx.x * SDF_THREADS + threadIdx.x;
    if (p >= n_points) return;
    const float *pt = points + ((size_t)bi * n_points + p) * 3;
    const float px = pt[0], py = pt[1], pz = pt[2];
    const float *hfb = hf + (size_t)bi * dim_x * dim_y;
    const float ox = min_box_center[2 * bi], oy = min_box_center[2 * bi + 1];
    const float top_z = -base_z;
    const float cell_x = 2.0f * half_x, cell_y = 2.0f * half_y;
    // NaN / inf coordinates: the comparisons below are all false for NaN, so the window degenerates to the whole field
    const float fi = rintf((px - ox) / cell_x), fj = rintf((py - oy) / cell_y);
    const int i0 = (int)fminf(fmaxf(fi, 0.f), (float)(dim_x - 1)), j0 = (int)fminf(fmaxf(fj, 0.f), (float)(dim_y - 1));
    const float d0 = column_sd(px, py, pz, x_points[i0] + ox, y_points[j0] + oy, hfb[i0 * dim_y + j0], half_x, half_y, base_z, top_z, inverted);
    int i_lo = 0, i_hi = dim_x - 1, j_lo = 0, j_hi = dim_y - 1;
    if (d0 < 3.0e8f) {              // also false for NaN
        const float bound = fmaxf(d0, 0.f);
        const int rx = (int)(bound / cell_x) + 2, ry = (int)(bound / cell_y) + 2;
        i_lo = max(i0 - rx, 0);
        i_hi = min(i0 + rx, dim_x - 1);
        j_lo = max(j0 - ry, 0);
        j_hi = min(j0 + ry, dim_y - 1);
    }
    float best = INFINITY;
    int best_cell = 0;
    for (int i = i_lo; i <= i_hi; ++i) {
        const float cx = x_points[i] + ox;
        const float *row = hfb + (size_t)i * dim_y;
        for (int j = j_lo; j <= j_hi; ++j) {
            const float sd = column_sd(px, py, pz, cx, y_points[j] + oy, row[j], half_x, half_y, base_z, top_z, inverted);
            if (sd < best) {             // first column that attains the minimum
                best = sd;
                best_cell = i * dim_y + j;
            }
        }
    }
    if (!(px == px && py == py && pz == pz)) best = __builtin_nanf("");   // a NaN coordinate: torch's abs / clamp / min propagate it, fmaxf / fminf above do not
    if (radius > 0.f) best -= radius;           // sdRoundBox: x - r is monotone, so it commutes with the min
    if (inverted) best = -best;
    out[(size_t)bi * n_points + p] = best;
    if (out_cell) out_cell[(size_t)bi * n_points + p] = best_cell;
}

// Adjoint of the query above with respect to the points: g_points = g_out * d(sd)/dp for the column the forward pass selected
// (out_cell).  The same piecewise expression torch's autograd differentiates in the reference (abs -> sign, clamp(min=0) + norm ->
// unit vector of the positive part, max -> its first arg-max, clamp(max=0) -> passes at <= 0), evaluated once per point.
__global__ __launch_bounds__(SDF_THREADS) void points_hf_sdf_grad_kernel(int n_points, int dim_x, int dim_y, const float *__restrict__ points,
                                                                         const float *__restrict__ hf, const float *__restrict__ min_box_center,
                                                                         const float *__restrict__ x_points, const float *__restrict__ y_points,
                                                                         float half_x, float half_y, float base_z, int inverted,
                                                                         const int32_t *__restrict__ cell, const float *__restrict__ g_out,
                                                                         float *__restrict__ g_points) {
    const int bi = blockIdx.y;
    const int p = blockIdx.x * SDF_THREADS + threadIdx.x;
    if (p >= n_points) return;
    const size_t ip = (size_t)bi * n_points + p;
    const float *pt = points + ip * 3;
    const int ci = cell[ip], i = ci / dim_y, j = ci - i * dim_y;
    const float h = hf[(size_t)bi * dim_x * dim_y + ci];
    const float top_z = -base_z;
    const float cx = x_points[i] + min_box_center[2 * bi], cy = y_points[j] + min_box_center[2 * bi + 1];
    const float cz = inverted ? (h + top_z) / 2.0f : (h + base_z) / 2.0f;
    const float hz = inverted ? (top_z - h) / 2.0f : (h - base_z) / 2.0f;
    const float d[3] = {pt[0] - cx, pt[1] - cy, pt[2] - cz};
    const float q[3] = {fabsf(d[0]) - half_x, fabsf(d[1]) - half_y, fabsf(d[2]) - hz};
    const float a[3] = {fmaxf(q[0], 0.f), fmaxf(q[1], 0.f), fmaxf(q[2], 0.f)};
    const float n = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    float g[3] = {0.f, 0.f, 0.f};
    if (n > 0.f) {
        g[0] = a[0] / n;
        g[1] = a[1] / n;
        g[2] = a[2] / n;
    }
    int km = 0;
    if (q[1] > q[km]) km = 1;
    if (q[2] > q[km]) km = 2;
    if (q[km] <= 0.f) g[km] += 1.0f;
    const float s = (inverted ? -1.0f : 1.0f) * g_out[ip];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float sg = d[k] > 0.f ? 1.0f : (d[k] < 0.f ? -1.0f : 0.0f);
        g_points[ip * 3 + k] = s * g[k] * sg;
    }
}

extern "C" int parc_points_hf_sdf_grad(void *stream, int batch, int n_points, int dim_x, int dim_y, const float *points, const float *hf,
                                       const float *min_box_center, const float *x_points, const float *y_points, float half_x, float half_y,
                                       float base_z, int inverted, const int32_t *cell, const float *g_out, float *g_points) {
    if (batch < 0 || n_points < 0 || dim_x <= 0 || dim_y <= 0) return PARC_EINVAL;
    if (batch == 0 || n_points == 0) return PARC_OK;
    if (batch > 65535) return PARC_EUNSUPPORTED;
    if (!points || !hf || !min_box_center || !x_points || !y_points || !cell || !g_out || !g_points) return PARC_EINVAL;
    hipLaunchKernelGGL(points_hf_sdf_grad_kernel, dim3((n_points + SDF_THREADS - 1) / SDF_THREADS, batch), dim3(SDF_THREADS), 0, (hipStream_t)stream,
                       n_points, dim_x, dim_y, points, hf, min_box_center, x_points, y_points, half_x, half_y, base_z, inverted, cell, g_out, g_points);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
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
