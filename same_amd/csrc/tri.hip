// tri.hip -- per-triangle geometry kernels: classes (a7), weights + source signs (a8),
// signed areas / flips (a12), eager reference signs (a14).  One lane per triangle; vertex XY
// are 16 B gathers.  Every expression keeps the reference's operation order and the file is
// built with -ffp-contract=off; the two places where the reference itself goes through a
// fused multiply-add (OpenBLAS ddot behind 1-D np.linalg.norm / np.dot, see
// oracle/same_oracle.c) use __builtin_fma explicitly.
#include "common.h"
#include "devmath.h"

namespace {

using namespace devmath;   // ld2, norm2, corner_cos, classify_triangle, orient_sign, signed_area: one definition (devmath.h)

__global__ __launch_bounds__(256) void tri_classify_kernel(
    const double *__restrict__ xy, const int32_t *__restrict__ tris, int64_t Tr, double radius,
    int angle_enabled, double cos_thr, const int32_t *__restrict__ type_id, uint8_t *__restrict__ out_class,
    double *__restrict__ out_perim, double *__restrict__ out_maxcos) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tr) return;
    const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    const TriClass r = classify_triangle(ld2(xy, a), ld2(xy, b), ld2(xy, c), radius, angle_enabled, cos_thr,
                                         type_id && type_id[a] == type_id[b] && type_id[b] == type_id[c]);
    out_class[t] = r.cls;
    out_perim[t] = r.perim;
    out_maxcos[t] = r.maxcos;
}

__global__ __launch_bounds__(256) void tri_sign_weight_kernel(
    const double *__restrict__ xy, const double *__restrict__ size, const int32_t *__restrict__ tris, int64_t Tr,
    int8_t *__restrict__ out_sign, double *__restrict__ out_weight) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tr) return;
    const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    out_sign[t] = orient_sign(ld2(xy, a), ld2(xy, b), ld2(xy, c));
    if (out_weight) out_weight[t] = size[a] + size[b] + size[c];  // src/same.py:1131-1133
}

__global__ __launch_bounds__(256) void area_flip_kernel(
    const double *__restrict__ axy, const double *__restrict__ rxy, const int32_t *__restrict__ tris, int64_t Tr,
    const int32_t *__restrict__ match, double *__restrict__ before, double *__restrict__ after,
    uint8_t *__restrict__ matched3, uint8_t *__restrict__ flipped) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tr) return;
    const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    const double bf = signed_area(ld2(axy, a), ld2(axy, b), ld2(axy, c));
    const int32_t ja = match[a], jb = match[b], jc = match[c];
    matched3[3 * t] = ja >= 0;
    matched3[3 * t + 1] = jb >= 0;
    matched3[3 * t + 2] = jc >= 0;
    double af = __builtin_nan("");
    uint8_t fl = 0;
    if (ja >= 0 && jb >= 0 && jc >= 0) {
        af = signed_area(ld2(rxy, ja), ld2(rxy, jb), ld2(rxy, jc));
        fl = bf * af < 0.0;  // src/same.py:1401
    }
    before[t] = bf;
    after[t] = af;
    flipped[t] = fl;
}

// src/helpers.py:425-441: sign(np.round(cross, 3)), np.round(x,3) = rint(x*1000)/1000.
// One lane per (triangle, x, y, z) combination.
__global__ __launch_bounds__(256) void eager_signs_kernel(
    const double *__restrict__ rxy, const int32_t *__restrict__ tris, int64_t Tr, const int32_t *__restrict__ cand,
    int k, int8_t *__restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k3 = (int64_t)k * k * k;
    if (q >= Tr * k3) return;
    const int64_t t = q / k3;
    const int rem = (int)(q - t * k3);
    const int x = rem / (k * k), y = (rem / k) % k, z = rem % k;
    const int32_t r1 = cand[(int64_t)tris[3 * t] * k + x], r2 = cand[(int64_t)tris[3 * t + 1] * k + y],
                  r3 = cand[(int64_t)tris[3 * t + 2] * k + z];
    int8_t s = 2;
    if (r1 >= 0 && r2 >= 0 && r3 >= 0) {
        const double2_t p1 = ld2(rxy, r1), p2 = ld2(rxy, r2), p3 = ld2(rxy, r3);
        double area = (p2.x - p1.x) * (p3.y - p1.y) - (p2.y - p1.y) * (p3.x - p1.x);
        area = __builtin_rint(area * 1000.0) / 1000.0;
        s = (int8_t)((area > 0.0) - (area < 0.0));
    }
    out[q] = s;
}

inline unsigned grid_for(int64_t n) { return (unsigned)ceil_div(n, 256); }

}  // namespace

extern "C" {

int same_tri_classify(same_ctx *ctx, const double *xy, int64_t n_pts, const int32_t *tris, int64_t Tr, double radius,
                      int angle_enabled, double cos_thr, const int32_t *type_id, uint8_t *out_class, double *out_perim,
                      double *out_maxcos) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, n_pts >= 0 && Tr >= 0 && Tr < ((int64_t)1 << 31) * 256);
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, xy && tris && out_class && out_perim && out_maxcos);
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, tris, Tr * 3, 0, n_pts, "triangles"));
    double *dxy, *dperim, *dcos;
    int32_t *dtris, *dtype = nullptr;
    uint8_t *dcls;
    SAME_TRY(up_as(ctx, SL_AXY, xy, (size_t)n_pts * 2, &dxy));
    SAME_TRY(up_as(ctx, SL_TRIS, tris, (size_t)Tr * 3, &dtris));
    if (type_id) SAME_TRY(up_as(ctx, SL_TYPE, type_id, (size_t)n_pts, &dtype));
    SAME_TRY(slot_as(ctx, SL_FLAG0, (size_t)Tr, &dcls));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)Tr, &dperim));
    SAME_TRY(slot_as(ctx, SL_OUT1, (size_t)Tr, &dcos));
    hipLaunchKernelGGL(tri_classify_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, dxy, dtris, Tr, radius,
                       angle_enabled, cos_thr, dtype, dcls, dperim, dcos);
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_down(ctx, out_class, dcls, (size_t)Tr));
    SAME_TRY(same_down(ctx, out_perim, dperim, (size_t)Tr * sizeof(double)));
    SAME_TRY(same_down(ctx, out_maxcos, dcos, (size_t)Tr * sizeof(double)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_tri_sign_weight(same_ctx *ctx, const double *xy, const double *size, int64_t n_pts, const int32_t *tris,
                         int64_t Tr, int8_t *out_sign, double *out_weight) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, n_pts >= 0 && Tr >= 0 && ((size == nullptr) == (out_weight == nullptr)));
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, xy && tris && out_sign);
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, tris, Tr * 3, 0, n_pts, "triangles"));
    double *dxy, *dsize = nullptr, *dw = nullptr;
    int32_t *dtris;
    int8_t *dsign;
    SAME_TRY(up_as(ctx, SL_AXY, xy, (size_t)n_pts * 2, &dxy));
    SAME_TRY(up_as(ctx, SL_TRIS, tris, (size_t)Tr * 3, &dtris));
    if (size) {
        SAME_TRY(up_as(ctx, SL_SIZE, size, (size_t)n_pts, &dsize));
        SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)Tr, &dw));
    }
    SAME_TRY(slot_as(ctx, SL_SIGN, (size_t)Tr, &dsign));
    hipLaunchKernelGGL(tri_sign_weight_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, dxy, dsize, dtris, Tr, dsign, dw);
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_down(ctx, out_sign, dsign, (size_t)Tr));
    if (out_weight) SAME_TRY(same_down(ctx, out_weight, dw, (size_t)Tr * sizeof(double)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_area_flip(same_ctx *ctx, const double *axy, int64_t n_m, const double *rxy, int64_t n_r, const int32_t *tris,
                   int64_t Tr, const int32_t *match, double *out_before, double *out_after, uint8_t *out_matched3,
                   uint8_t *out_flipped) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, n_m >= 0 && n_r >= 0 && Tr >= 0);
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, axy && tris && match && out_before && out_after && out_matched3 && out_flipped && (n_r == 0 || rxy));
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, tris, Tr * 3, 0, n_m, "triangles"));
    SAME_TRY(check_index_range(ctx, match, n_m, -1, n_r, "match"));
    double *dax, *drx, *dbf, *daf;
    int32_t *dtris, *dmatch;
    uint8_t *dm3, *dfl;
    SAME_TRY(up_as(ctx, SL_AXY, axy, (size_t)n_m * 2, &dax));
    SAME_TRY(up_as(ctx, SL_RXY, rxy, (size_t)n_r * 2, &drx));
    SAME_TRY(up_as(ctx, SL_TRIS, tris, (size_t)Tr * 3, &dtris));
    SAME_TRY(up_as(ctx, SL_MATCH, match, (size_t)n_m, &dmatch));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)Tr, &dbf));
    SAME_TRY(slot_as(ctx, SL_OUT1, (size_t)Tr, &daf));
    SAME_TRY(slot_as(ctx, SL_FLAG0, (size_t)Tr * 3, &dm3));
    SAME_TRY(slot_as(ctx, SL_FLAG1, (size_t)Tr, &dfl));
    hipLaunchKernelGGL(area_flip_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, dax, drx, dtris, Tr, dmatch, dbf, daf,
                       dm3, dfl);
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_down(ctx, out_before, dbf, (size_t)Tr * sizeof(double)));
    SAME_TRY(same_down(ctx, out_after, daf, (size_t)Tr * sizeof(double)));
    SAME_TRY(same_down(ctx, out_matched3, dm3, (size_t)Tr * 3));
    SAME_TRY(same_down(ctx, out_flipped, dfl, (size_t)Tr));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

int same_eager_signs(same_ctx *ctx, const double *rxy, int64_t n_r, const int32_t *tris, int64_t Tr,
                     const int32_t *cand, int64_t n_m, int k, int8_t *out) {
    REQUIRE(ctx, ctx != nullptr);
    REQUIRE(ctx, n_r >= 0 && n_m >= 0 && Tr >= 0 && k >= 1 && k <= SAME_MAX_KNN);
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, tris && cand && out && (n_r == 0 || rxy));
    const int64_t total = Tr * (int64_t)k * k * k;
    REQUIRE(ctx, total < ((int64_t)1 << 31) * 256);
    SAME_TRY(same_use(ctx));
    SAME_TRY(check_index_range(ctx, tris, Tr * 3, 0, n_m, "triangles"));
    SAME_TRY(check_index_range(ctx, cand, n_m * k, -1, n_r, "cand"));
    double *drx;
    int32_t *dtris, *dcand;
    int8_t *dout;
    SAME_TRY(up_as(ctx, SL_RXY, rxy, (size_t)n_r * 2, &drx));
    SAME_TRY(up_as(ctx, SL_TRIS, tris, (size_t)Tr * 3, &dtris));
    SAME_TRY(up_as(ctx, SL_PAIRS, cand, (size_t)n_m * k, &dcand));
    SAME_TRY(slot_as(ctx, SL_OUT0, (size_t)total, &dout));
    hipLaunchKernelGGL(eager_signs_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, drx, dtris, Tr, dcand, k, dout);
    HIP_TRY(ctx, hipGetLastError());
    SAME_TRY(same_down(ctx, out, dout, (size_t)total));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SAME_OK;
}

// ---- device-resident forms (operands already in HBM; enqueue only) --------------------------
int same_tri_classify_dev(same_ctx *ctx, const double *dxy, const int32_t *dtris, int64_t Tr, double radius,
                          int angle_enabled, double cos_thr, const int32_t *dtype_id, uint8_t *dout_class,
                          double *dout_perim, double *dout_maxcos) {
    REQUIRE(ctx, ctx && Tr >= 0);
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, dxy && dtris && dout_class && dout_perim && dout_maxcos);
    SAME_TRY(same_use(ctx));
    hipLaunchKernelGGL(tri_classify_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, dxy, dtris, Tr, radius,
                       angle_enabled, cos_thr, dtype_id, dout_class, dout_perim, dout_maxcos);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

int same_tri_sign_weight_dev(same_ctx *ctx, const double *dxy, const double *dsize, const int32_t *dtris, int64_t Tr,
                             int8_t *dout_sign, double *dout_weight) {
    REQUIRE(ctx, ctx && Tr >= 0 && ((dsize == nullptr) == (dout_weight == nullptr)));
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, dxy && dtris && dout_sign);
    SAME_TRY(same_use(ctx));
    hipLaunchKernelGGL(tri_sign_weight_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, dxy, dsize, dtris, Tr, dout_sign,
                       dout_weight);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

int same_area_flip_dev(same_ctx *ctx, const double *daxy, const double *drxy, const int32_t *dtris, int64_t Tr,
                       const int32_t *dmatch, double *dout_before, double *dout_after, uint8_t *dout_matched3,
                       uint8_t *dout_flipped) {
    REQUIRE(ctx, ctx && Tr >= 0);
    if (Tr == 0) return SAME_OK;
    REQUIRE(ctx, daxy && drxy && dtris && dmatch && dout_before && dout_after && dout_matched3 && dout_flipped);
    SAME_TRY(same_use(ctx));
    hipLaunchKernelGGL(area_flip_kernel, dim3(grid_for(Tr)), dim3(256), 0, ctx->stream, daxy, drxy, dtris, Tr, dmatch,
                       dout_before, dout_after, dout_matched3, dout_flipped);
    HIP_TRY(ctx, hipGetLastError());
    return SAME_OK;
}

}  // extern "C"
