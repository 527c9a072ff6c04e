// a6 on the host, without the library call: the Delaunay triangulation of a window's aligned cells (src/same.py:1023 calls
// scipy.spatial.Delaunay = Qhull, ~1.6 us per point and three quarters of a cfg 5 pass).  This file is a sweep-hull triangulator
// (seed triangle, points by distance from its circumcentre, advancing convex hull, edge flips) that answers ONLY when the answer is
// beyond doubt the one Qhull gives:
//   * every orientation / in-circle sign it relies on clears a floating-point error bound by orders of magnitude, else it says
//     SAME_EUNSURE at once (duplicate, collinear and cocircular points end here);
//   * afterwards every interior edge and every hull corner of the finished triangulation is measured the way Qhull sees it -- the
//     distance of the fourth point from the plane of the lifted triangle, in Qhull's 'Qbb'-scaled paraboloid coordinates -- and if
//     any of them is within `guard` x Qhull's own round-off allowance the answer is SAME_EUNSURE again: with points in general
//     position the Delaunay triangulation is unique, so an answer that is given is the same SET of triangles as scipy's.
// SAME_EUNSURE is not an error: the caller asks scipy, exactly as the reference does (same_amd/delaunay.py).  The triangles come in
// this file's order, counter-clockwise; Qhull's order and orientation are its own (same_amd/delaunay.py says what that touches).
// Host code only: no device, no context.
#include "same_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

constexpr double SIGN_MARGIN = 1e-12;   // a sign is trusted when |det| > SIGN_MARGIN x (sum of the |products| it was made of); the
                                        // proven bounds (Shewchuk, 3.4e-16 and 1.2e-15) are three orders below
constexpr double EPS = 2.220446049250313e-16;

struct Unsure {};

struct Tri {
    const double *xy;
    int64_t n;
    std::vector<int32_t> tri, half;      // 3 per triangle: vertex at the start of the half-edge; the twin half-edge or -1
    std::vector<int32_t> hprev, hnext, htri, hash;
    std::vector<int32_t> stack;
    std::vector<uint64_t> order, order_tmp;          // (squared distance from the seed's circumcentre as a float's bits, point), ascending
    std::vector<int32_t> ids;                        // place in the insertion order -> the caller's point
    std::vector<double> pxy;                         // the points in insertion order
    const double *cur = nullptr;                     // the coordinates X() / Y() read: the caller's until the order is known, then pxy
    std::vector<double> judge;                       // per triangle: what turns an in-circle determinant into distance / allowance
    int64_t len = 0;
    int32_t hash_size = 0, hull_start = 0;
    double cx = 0, cy = 0;
    // |det| above these clears SIGN_MARGIN x (the sum of its products) whatever the points: every coordinate difference is at most the
    // set's extent w, so the orientation's products sum to <= 2 w^2 and the in-circle's to <= 12 w^4
    double sure_cross = 0, sure_incircle = 0;

    double X(int32_t i) const { return cur[2 * (int64_t)i]; }
    double Y(int32_t i) const { return cur[2 * (int64_t)i + 1]; }

    // > 0: a, b, c counter-clockwise
    static double cross(double ax, double ay, double bx, double by, double px, double py, double surely = std::numeric_limits<double>::infinity()) {
        const double l = (ax - px) * (by - py), r = (ay - py) * (bx - px);
        const double det = l - r;
        if (std::fabs(det) > surely) return det;                 // clear of the bound for ANY three points of this set
        if (!(std::fabs(det) > SIGN_MARGIN * (std::fabs(l) + std::fabs(r)))) throw Unsure{};
        return det;
    }
    bool ccw(int32_t a, int32_t b, int32_t c) const { return cross(X(a), Y(a), X(b), Y(b), X(c), Y(c), sure_cross) > 0; }
    // > 0: p inside the circle through a, b, c (counter-clockwise)
    static double incircle(double ax, double ay, double bx, double by, double cx, double cy, double px, double py,
                           double surely = std::numeric_limits<double>::infinity()) {
        const double dx = ax - px, dy = ay - py, ex = bx - px, ey = by - py, fx = cx - px, fy = cy - py;
        const double ap = dx * dx + dy * dy, bp = ex * ex + ey * ey, cp = fx * fx + fy * fy;
        const double det = dx * (ey * cp - bp * fy) - dy * (ex * cp - bp * fx) + ap * (ex * fy - ey * fx);
        if (std::fabs(det) > surely) return det;
        const double permanent = (std::fabs(ey * cp) + std::fabs(bp * fy)) * std::fabs(dx) + (std::fabs(ex * cp) + std::fabs(bp * fx)) * std::fabs(dy) +
                                 (std::fabs(ex * fy) + std::fabs(ey * fx)) * ap;
        if (!(std::fabs(det) > SIGN_MARGIN * permanent)) throw Unsure{};
        return det;
    }
    bool inside(int32_t a, int32_t b, int32_t c, int32_t p) const {
        return incircle(X(a), Y(a), X(b), Y(b), X(c), Y(c), X(p), Y(p), sure_incircle) > 0;
    }

    static uint32_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
    // Three counting passes over the 32 distance bits (11 + 11 + 10): non-negative floats order as their bits.  Single precision may
    // swap two points whose distances agree to seven digits; the later one is then still outside the hull of the earlier ones (a hull's
    // edges lie inside the circle through its farthest point by far more than that) -- and if it ever were not, the walk along the hull
    // finds no edge that sees it and the answer is SAME_EUNSURE
    void sort_by_distance() {
        order_tmp.resize(order.size());
        uint32_t count[2048];
        const int shifts[3] = {32, 43, 54}, bits[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            const uint32_t mask = (1u << bits[pass]) - 1;
            std::fill(count, count + 2048, 0u);
            for (uint64_t v : order) ++count[(v >> shifts[pass]) & mask];
            uint32_t sum = 0;
            for (uint32_t q = 0; q <= mask; ++q) { const uint32_t c = count[q]; count[q] = sum; sum += c; }
            for (uint64_t v : order) order_tmp[count[(v >> shifts[pass]) & mask]++] = v;
            order.swap(order_tmp);
        }
    }

    int32_t key(double x, double y) const {
        const double dx = x - cx, dy = y - cy, den = std::fabs(dx) + std::fabs(dy);
        if (!(den > 0)) return 0;                                      // a point AT the seed's circumcentre (it is then refused: no hull edge sees it)
        const double p = dx / den;
        const double a = (dy > 0 ? 3 - p : 1 + p) / 4;                 // [0, 1], grows counter-clockwise
        return (int32_t)((int64_t)std::floor(a * hash_size) % hash_size);
    }
    void link(int32_t a, int32_t b) {
        half[a] = b;
        if (b >= 0) half[b] = a;
    }
    int32_t add(int32_t i0, int32_t i1, int32_t i2, int32_t a, int32_t b, int32_t c) {
        const int32_t t = (int32_t)len;
        tri[t] = i0; tri[t + 1] = i1; tri[t + 2] = i2;
        link(t, a); link(t + 1, b); link(t + 2, c);
        len += 3;
        return t;
    }
    // flips until the edge a (and what the flips disturb) is locally Delaunay; -> the half-edge that leaves the new point along the hull
    int32_t legalize(int32_t a) {
        size_t depth = 0;
        int32_t ar;
        for (;;) {
            const int32_t b = half[a];
            const int32_t a0 = a - a % 3;
            ar = a0 + (a + 2) % 3;
            if (b < 0) {
                if (!depth) break;
                a = stack[--depth];
                continue;
            }
            const int32_t b0 = b - b % 3, al = a0 + (a + 1) % 3, bl = b0 + (b + 2) % 3;
            const int32_t p0 = tri[ar], pr = tri[a], pl = tri[al], p1 = tri[bl];
            if (inside(p0, pr, pl, p1)) {
                tri[a] = p1;
                tri[b] = p0;
                const int32_t hbl = half[bl];
                if (hbl < 0) {          // the flipped edge lay on the hull: the hull's note of it moves along
                    int32_t e = hull_start;
                    do {
                        if (htri[e] == bl) { htri[e] = a; break; }
                        e = hprev[e];
                    } while (e != hull_start);
                }
                link(a, hbl);
                link(b, half[ar]);
                link(ar, bl);
                const int32_t br = b0 + (b + 1) % 3;
                if (depth == stack.size()) stack.resize(stack.size() * 2);
                stack[depth++] = br;
            } else {
                if (!depth) break;
                a = stack[--depth];
            }
        }
        return ar;
    }

    void run() {
        cur = xy;
        double minx = std::numeric_limits<double>::infinity(), miny = minx, maxx = -minx, maxy = -minx;
        for (int64_t i = 0; i < n; ++i) {
            const double x = X((int32_t)i), y = Y((int32_t)i);
            if (!std::isfinite(x) || !std::isfinite(y)) throw Unsure{};
            minx = std::min(minx, x); maxx = std::max(maxx, x);
            miny = std::min(miny, y); maxy = std::max(maxy, y);
        }
        const double mx = (minx + maxx) / 2, my = (miny + maxy) / 2;
        const double w = std::max(maxx - minx, maxy - miny);
        sure_cross = SIGN_MARGIN * 2 * w * w;
        sure_incircle = SIGN_MARGIN * 12 * w * w * w * w;
        auto d2 = [&](int32_t i, double x, double y) { const double dx = X(i) - x, dy = Y(i) - y; return dx * dx + dy * dy; };
        int32_t i0 = 0, i1 = -1, i2 = -1;
        double best = std::numeric_limits<double>::infinity();
        for (int32_t i = 0; i < n; ++i) { const double d = d2(i, mx, my); if (d < best) { best = d; i0 = i; } }
        best = std::numeric_limits<double>::infinity();
        for (int32_t i = 0; i < n; ++i) { if (i == i0) continue; const double d = d2(i, X(i0), Y(i0)); if (d < best && d > 0) { best = d; i1 = i; } }
        if (i1 < 0) throw Unsure{};
        // the third point of the seed: the smallest circumcircle with the first two
        auto circumradius2 = [&](int32_t c) {
            const double dx = X(i1) - X(i0), dy = Y(i1) - Y(i0), ex = X(c) - X(i0), ey = Y(c) - Y(i0);
            const double bl = dx * dx + dy * dy, cl = ex * ex + ey * ey, d = dx * ey - dy * ex;
            if (d == 0) return std::numeric_limits<double>::infinity();
            const double x = (ey * bl - dy * cl) * 0.5 / d, y = (dx * cl - ex * bl) * 0.5 / d;
            const double r = x * x + y * y;
            return r > 0 ? r : std::numeric_limits<double>::infinity();
        };
        best = std::numeric_limits<double>::infinity();
        for (int32_t i = 0; i < n; ++i) { if (i == i0 || i == i1) continue; const double r = circumradius2(i); if (r < best) { best = r; i2 = i; } }
        if (i2 < 0 || !std::isfinite(best)) throw Unsure{};
        if (!ccw(i0, i1, i2)) std::swap(i1, i2);
        {
            const double dx = X(i1) - X(i0), dy = Y(i1) - Y(i0), ex = X(i2) - X(i0), ey = Y(i2) - Y(i0);
            const double bl = dx * dx + dy * dy, cl = ex * ex + ey * ey, d = dx * ey - dy * ex;
            cx = X(i0) + (ey * bl - dy * cl) * 0.5 / d;
            cy = Y(i0) + (dx * cl - ex * bl) * 0.5 / d;
        }
        order.clear();
        order.reserve((size_t)n);
        for (int32_t i = 0; i < n; ++i)
            if (i != i0 && i != i1 && i != i2) order.push_back(((uint64_t)float_bits((float)d2(i, cx, cy)) << 32) | (uint32_t)i);
        sort_by_distance();
        // from here on the points go by their place in the insertion order (seed first): a new point's neighbours on the hull and in the
        // triangle list are recent points -- their coordinates, hull links and triangles sit together in memory
        ids.resize((size_t)n);
        pxy.resize((size_t)n * 2);
        ids[0] = i0; ids[1] = i1; ids[2] = i2;
        for (int64_t q = 0; q < (int64_t)order.size(); ++q) ids[(size_t)q + 3] = (int32_t)(uint32_t)order[(size_t)q];
        for (int64_t q = 0; q < n; ++q) { pxy[2 * q] = xy[2 * (int64_t)ids[(size_t)q]]; pxy[2 * q + 1] = xy[2 * (int64_t)ids[(size_t)q] + 1]; }
        cur = pxy.data();
        i0 = 0; i1 = 1; i2 = 2;

        hash_size = (int32_t)std::ceil(std::sqrt((double)n));
        hash.assign((size_t)hash_size, -1);
        // every entry of the arrays below is written before it is read (a vertex's hull links when it joins the hull, a triangle's
        // corners and twins when it is added): the thread's arrays of its previous call are reused as they are, no fill, no allocation
        if (hprev.size() < (size_t)n) { hprev.resize((size_t)n); hnext.resize((size_t)n); htri.resize((size_t)n); }
        const size_t max_halves = (size_t)std::max<int64_t>(2 * n - 5, 1) * 3;
        if (tri.size() < max_halves) { tri.resize(max_halves); half.resize(max_halves); }
        if (stack.size() < 512) stack.resize(512);
        len = 0;
        hull_start = i0;
        hnext[i0] = hprev[i2] = i1;
        hnext[i1] = hprev[i0] = i2;
        hnext[i2] = hprev[i1] = i0;
        htri[i0] = 0; htri[i1] = 1; htri[i2] = 2;
        hash[key(X(i0), Y(i0))] = i0;
        hash[key(X(i1), Y(i1))] = i1;
        hash[key(X(i2), Y(i2))] = i2;
        add(i0, i1, i2, -1, -1, -1);

        // the hull runs counter-clockwise; the edge e -> next(e) is seen from p when p lies to its right
        auto sees = [&](int32_t p, int32_t e, int32_t q) { return !ccw(e, q, p); };
        for (int32_t i = 3; i < (int32_t)n; ++i) {
            const double x = X(i), y = Y(i);
            int32_t start = 0;
            const int32_t k = key(x, y);
            for (int32_t j = 0; j < hash_size; ++j) {
                start = hash[(k + j) % hash_size];
                if (start != -1 && start != hnext[start]) break;
            }
            start = hprev[start];
            int32_t e = start, q;
            while (q = hnext[e], !sees(i, e, q)) {
                e = q;
                if (e == start) throw Unsure{};      // inside the hull although farther from the seed's centre than all before: round-off
            }
            // the first triangle from the point
            int32_t t = add(e, i, hnext[e], -1, -1, htri[e]);
            htri[i] = legalize(t + 2);
            htri[e] = t;
            // forward along the hull while its edges are seen ...
            int32_t nx = hnext[e];
            while (q = hnext[nx], sees(i, nx, q)) {
                t = add(nx, i, q, htri[i], -1, htri[nx]);
                htri[i] = legalize(t + 2);
                hnext[nx] = nx;                   // no longer on the hull
                nx = q;
            }
            // ... and backward
            if (e == start) {
                while (q = hprev[e], sees(i, q, e)) {
                    t = add(q, i, e, -1, htri[e], htri[q]);
                    legalize(t + 2);
                    htri[q] = t;
                    hnext[e] = e;
                    e = q;
                }
            }
            hull_start = hprev[i] = e;
            hnext[e] = hprev[nx] = i;
            hnext[i] = nx;
            hash[key(x, y)] = i;
            hash[key(X(e), Y(e))] = e;
        }
    }

    // The finished triangulation as Qhull would judge it.  Qhull lifts (x, y) to z = x*x + y*y WITHOUT centring, scales z to the range
    // [0, m], m = the largest |x| or |y| ('Qbb'), and takes a point for coplanar with a facet when its distance from the facet's plane
    // is within a few DISTround = eps * (3 * sqrt(3) * 1.01 + 1) * m; z itself carries eps * z of rounding before the scaling.
    // worst = the smallest (distance / allowance) over interior edges (the neighbour's far corner against the triangle's plane) and
    // hull corners (the corner's distance from the chord of its neighbours, and every hull triangle's height over its hull edge).
    double worst_margin() {
        double m = 0, zmin = std::numeric_limits<double>::infinity(), zmax = 0;
        for (int64_t i = 0; i < n; ++i) {
            const double x = X((int32_t)i), y = Y((int32_t)i), z = x * x + y * y;
            m = std::max(m, std::max(std::fabs(x), std::fabs(y)));
            zmin = std::min(zmin, z); zmax = std::max(zmax, z);
        }
        const double s = zmax > zmin ? m / (zmax - zmin) : 1.0;                      // Qbb's scale of the lifted coordinate
        const double allow = EPS * (6.25 * m + zmax * s);                            // plane distance Qhull cannot tell from zero
        double worst = std::numeric_limits<double>::infinity();
        // per triangle (p0, p1, p2): the plane of the lifted triangle is z = 2 c . x + const, c its circumcentre -- slope 2 |c| before the
        // scaling; a thin triangle's plane is known that much worse (longest side over height); the distance of a point with in-circle
        // determinant det is s * det / (2 area) / sqrt(1 + slope^2)
        const int64_t n_tri = len / 3;
        judge.resize((size_t)n_tri);
        for (int64_t t = 0; t < n_tri; ++t) {
            const int32_t p0 = tri[3 * t], p1 = tri[3 * t + 1], p2 = tri[3 * t + 2];
            const double dx = X(p1) - X(p0), dy = Y(p1) - Y(p0), ex = X(p2) - X(p0), ey = Y(p2) - Y(p0);
            const double bl = dx * dx + dy * dy, cl = ex * ex + ey * ey, d = dx * ey - dy * ex, area2 = std::fabs(d);
            const double ccx = X(p0) + (ey * bl - dy * cl) * 0.5 / d, ccy = Y(p0) + (dx * cl - ex * bl) * 0.5 / d;
            const double slope2 = 4 * s * s * (ccx * ccx + ccy * ccy);
            const double l2 = std::max(bl, std::max(cl, (ex - dx) * (ex - dx) + (ey - dy) * (ey - dy)));
            judge[(size_t)t] = s / (area2 * std::sqrt(1 + slope2) * allow * std::max(1.0, l2 / area2));
        }
        for (int32_t a = 0; a < (int32_t)len; ++a) {
            const int32_t b = half[a];
            const int32_t a0 = a - a % 3;
            const int32_t p0 = tri[a0 + (a + 2) % 3], pr = tri[a], pl = tri[a0 + (a + 1) % 3];
            if (b < 0) {
                // a hull edge pr -> pl with p0 behind it: the height of p0 over the edge, in plain coordinates (the facet next to it
                // is vertical: it holds the point at infinity of 'Qz')
                const double area2 = std::fabs((X(pr) - X(p0)) * (Y(pl) - Y(p0)) - (Y(pr) - Y(p0)) * (X(pl) - X(p0)));
                const double edge = std::hypot(X(pl) - X(pr), Y(pl) - Y(pr));
                worst = std::min(worst, area2 / edge / allow);
                continue;
            }
            if (b < a) continue;
            const int32_t p1 = tri[b - b % 3 + (b + 2) % 3];
            const double det = std::fabs(incircle(X(p0), Y(p0), X(pr), Y(pr), X(pl), Y(pl), X(p1), Y(p1), sure_incircle));
            worst = std::min(worst, det * judge[(size_t)(a0 / 3)]);
        }
        // hull corners: the corner against the chord of its neighbours
        int32_t e = hull_start;
        do {
            const int32_t p = hprev[e], q = hnext[e];
            const double area2 = std::fabs((X(e) - X(p)) * (Y(q) - Y(p)) - (Y(e) - Y(p)) * (X(q) - X(p)));
            worst = std::min(worst, area2 / std::hypot(X(q) - X(p), Y(q) - Y(p)) / allow);
            e = q;
        } while (e != hull_start);
        return worst;
    }
};

}  // namespace

extern "C" int same_delaunay2d(const double *xy, int64_t n, int32_t *tris, int64_t cap, int64_t *n_tris, double guard, double *margin) {
    // half-edge numbers (three per triangle, fewer than 2 n triangles) are 32-bit
    if (!xy || !tris || !n_tris || n < 0 || cap < 0 || n > (int64_t)350000000 || !(guard >= 0.0)) return SAME_EINVAL;
    *n_tris = 0;
    if (margin) *margin = 0.0;
    if (n < 3) return SAME_EUNSURE;
    try {
        static thread_local Tri t;       // a thread's working arrays stay with it from call to call (a window: ~1.5 MB)
        t.xy = xy;
        t.n = n;
        t.run();
        const double worst = t.worst_margin();
        if (margin) *margin = worst;
        if (!(worst > guard)) return SAME_EUNSURE;
        const int64_t count = t.len / 3;
        if (count > cap) return SAME_EINVAL;
        for (int64_t q = 0; q < t.len; ++q) tris[q] = t.ids[(size_t)t.tri[(size_t)q]];
        *n_tris = count;
        return SAME_OK;
    } catch (const Unsure &) {
        return SAME_EUNSURE;
    } catch (const std::bad_alloc &) {
        return SAME_ENOMEM;
    }
}
