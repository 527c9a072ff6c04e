// devmath.h -- the per-triangle arithmetic of the path as device functions, one definition for every kernel that needs it
// (tri.hip, sweep.hip, window.hip).  Each expression keeps the reference's operation order; the library is built with
// -ffp-contract=off, and the two places where the reference itself goes through a fused multiply-add (OpenBLAS ddot behind
// 1-D np.linalg.norm / np.dot, see oracle/same_oracle.c) use __builtin_fma explicitly.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace devmath {

typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double2_t ld2(const double *xy, int64_t i) { return *reinterpret_cast<const double2_t *>(xy + 2 * i); }

// src/helpers.py:305-307 -- np.linalg.norm of a 2-vector = sqrt(ddot(v, v))
__device__ __forceinline__ double norm2(double x, double y) { return __builtin_sqrt(__builtin_fma(y, y, x * x)); }

// src/helpers.py:278-288 -- clipped cosine of the corner at p2; 2.0 encodes the "angle 0" early return
__device__ __forceinline__ double corner_cos(double2_t p1, double2_t p2, double2_t p3) {
    const double v1x = p1.x - p2.x, v1y = p1.y - p2.y, v2x = p3.x - p2.x, v2y = p3.y - p2.y;
    const double n1 = norm2(v1x, v1y), n2 = norm2(v2x, v2y);
    if (n1 == 0.0 || n2 == 0.0) return 2.0;
    double c = __builtin_fma(v1y, v2y, v1x * v2x) / (n1 * n2);
    c = c < -1.0 ? -1.0 : c;
    c = c > 1.0 ? 1.0 : c;
    return c;
}

// class of one Delaunay triangle (src/helpers.py:300-330): 0 kept, 1 a side >= radius, 2 an angle below the threshold,
// 3 all three cell types equal; perimeter (:334) and the largest corner cosine come along
struct TriClass {
    uint8_t cls;
    double perim, maxcos;
};
__device__ __forceinline__ TriClass classify_triangle(double2_t p1, double2_t p2, double2_t p3, double radius, int angle_enabled,
                                                       double cos_thr, bool same_type) {
    const double s1 = norm2(p2.x - p1.x, p2.y - p1.y);
    const double s2 = norm2(p3.x - p2.x, p3.y - p2.y);
    const double s3 = norm2(p1.x - p3.x, p1.y - p3.y);
    double mx = s1 > s2 ? s1 : s2;
    mx = mx > s3 ? mx : s3;
    const double c1 = corner_cos(p2, p1, p3), c2 = corner_cos(p1, p2, p3), c3 = corner_cos(p1, p3, p2);
    double mc = c1 > c2 ? c1 : c2;
    mc = mc > c3 ? mc : c3;
    TriClass out;
    out.cls = 0;
    if (mx >= radius) out.cls = 1;                                    // src/helpers.py:310
    else if (angle_enabled && mc >= cos_thr) out.cls = 2;             // src/helpers.py:319
    else if (same_type) out.cls = 3;                                  // :328-330
    out.perim = s1 + s2 + s3;                                         // src/helpers.py:334
    out.maxcos = mc;
    return out;
}

// src/same.py:1146, :658
__device__ __forceinline__ int8_t orient_sign(double2_t a, double2_t b, double2_t c) {
    const double v = (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
    return (int8_t)((v > 0.0) - (v < 0.0));
}

// The orientation (:57-60) or the signed area (below) of a, b, c is so close to zero that the corners taken in another order could
// give the other sign, or zero: both forms are sums of products of coordinates or coordinate differences, each bounded by
// (|largest coordinate| + longest side) x longest side, with a handful of roundings
__device__ __forceinline__ bool sign_in_doubt(double2_t a, double2_t b, double2_t c) {
    const double v = (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
    const double side = fmax(fmax(fabs(b.x - a.x) + fabs(b.y - a.y), fabs(c.x - a.x) + fabs(c.y - a.y)), fabs(c.x - b.x) + fabs(c.y - b.y));
    const double reach = fmax(fmax(fabs(a.x), fabs(b.x)), fabs(c.x)) + side;
    return fabs(v) <= 3.6e-15 * reach * side;       // 16 eps
}

// src/helpers.py:73-77
__device__ __forceinline__ double signed_area(double2_t p1, double2_t p2, double2_t p3) {
    return 0.5 * (p1.x * (p2.y - p3.y) + p2.x * (p3.y - p1.y) + p3.x * (p1.y - p2.y));
}

// the lazy-constraint body for one triangle (src/same.py:649-669): 0 not checked, 1 checked, 2 checked and flipped
__device__ __forceinline__ uint8_t orient_flag(int8_t src_sign, bool all_matched, double2_t ra, double2_t rb, double2_t rc) {
    if (!all_matched) return 0;                                       // :649-650
    const int8_t rs = orient_sign(ra, rb, rc);
    if (src_sign == 0 || rs == 0) return 0;                           // :663-664
    return src_sign != rs ? 2 : 1;                                    // :665-669
}

// one edge of the XY-order sweep (src/violationhelper.py:68-75): bit 1 = X order broken, bit 2 = Y order broken
__device__ __forceinline__ uint8_t xyorder_edge(double2_t ap, double2_t aq, double2_t rp, double2_t rq) {
    const bool ox = ap.x < aq.x, oy = ap.y < aq.y;                    // :68-69
    const bool mx = rp.x < rq.x, my = rp.y < rq.y;                    // :74-75
    return (uint8_t)((ox != mx ? 2 : 0) | (oy != my ? 4 : 0));
}

// monotone 64-bit key of a double (total order of the values; -0.0 < +0.0, NaNs above +inf)
__device__ __forceinline__ unsigned long long order_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_to_double(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)u);
}

}  // namespace devmath
