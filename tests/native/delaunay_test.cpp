// same_delaunay2d (same_amd/csrc/delaunay.cpp) under AddressSanitizer + UBSan, on the CPU: random sets from three points to thousands,
// degenerate sets (lattice, duplicates, one line, cocircular, too few points), threads calling at once.  Every answer is checked
// for what can be checked without a second triangulator: 2 n - 2 - h triangles, every triangle counter-clockwise, every vertex used,
// no point inside a triangle's circumcircle among the triangle's neighbours' corners (local Delaunay property, which is global).
#include "same_hip.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <thread>
#include <vector>

static uint64_t state = 0x9E3779B97F4A7C15ull;
static double uniform() {
    state ^= state << 13; state ^= state >> 7; state ^= state << 17;
    return (double)(state >> 11) / 9007199254740992.0;
}

static int check(const std::vector<double> &xy, int64_t n, const std::vector<int32_t> &t, int64_t nt) {
    std::map<std::pair<int32_t, int32_t>, int32_t> opposite;     // directed edge -> the corner across it
    std::vector<char> used((size_t)n, 0);
    for (int64_t q = 0; q < nt; ++q) {
        const int32_t v[3] = {t[3 * q], t[3 * q + 1], t[3 * q + 2]};
        for (int c = 0; c < 3; ++c) {
            if (v[c] < 0 || v[c] >= n) return 1;
            used[(size_t)v[c]] = 1;
            opposite[{v[c], v[(c + 1) % 3]}] = v[(c + 2) % 3];
        }
        const double ax = xy[2 * v[0]], ay = xy[2 * v[0] + 1], bx = xy[2 * v[1]], by = xy[2 * v[1] + 1], cx = xy[2 * v[2]], cy = xy[2 * v[2] + 1];
        if (!((bx - ax) * (cy - ay) - (by - ay) * (cx - ax) > 0)) return 2;
    }
    for (int64_t i = 0; i < n; ++i)
        if (!used[(size_t)i]) return 3;
    int64_t hull = 0;
    for (const auto &e : opposite) {
        const auto twin = opposite.find({e.first.second, e.first.first});
        if (twin == opposite.end()) { ++hull; continue; }
        const int32_t a = e.first.first, b = e.first.second, c = e.second, p = twin->second;
        const double dx = xy[2 * a] - xy[2 * p], dy = xy[2 * a + 1] - xy[2 * p + 1], ex = xy[2 * b] - xy[2 * p], ey = xy[2 * b + 1] - xy[2 * p + 1],
                     fx = xy[2 * c] - xy[2 * p], fy = xy[2 * c + 1] - xy[2 * p + 1];
        const double ap = dx * dx + dy * dy, bp = ex * ex + ey * ey, cp = fx * fx + fy * fy;
        if (dx * (ey * cp - bp * fy) - dy * (ex * cp - bp * fx) + ap * (ex * fy - ey * fx) > 0) return 4;      // p inside circle(a, b, c)
    }
    return nt == 2 * n - 2 - hull ? 0 : 5;
}

static int one_set(int64_t n, int kind, int *answered) {
    std::vector<double> xy((size_t)n * 2 + 2);          // never an empty vector: its data() may be null
    for (int64_t i = 0; i < n; ++i) {
        double x = uniform() * 1000, y = uniform() * 1000;
        if (kind == 1) { x = std::floor(x / 50) * 50; y = std::floor(y / 50) * 50; }                       // lattice (and duplicates)
        if (kind == 2) { y = 2 * x + 1; }                                                                  // one line
        if (kind == 3) { const double a = 6.283185307179586 * (double)i / (double)n; x = 500 + 300 * std::cos(a); y = 500 + 300 * std::sin(a); }
        if (kind == 4) { x += 7.0e3; y -= 2.0e4; }                                                         // away from the origin
        xy[2 * i] = x; xy[2 * i + 1] = y;
    }
    if (kind == 5 && n > 4) { xy[8] = xy[0]; xy[9] = xy[1]; }                                              // one duplicate
    const int64_t cap = 2 * n > 5 ? 2 * n - 5 : 1;
    std::vector<int32_t> t((size_t)cap * 3);
    int64_t nt = -1;
    double margin = -1;
    const int rc = same_delaunay2d(xy.data(), n, t.data(), cap, &nt, 16.0, &margin);
    if (rc == SAME_EUNSURE) return nt == 0 ? 0 : 10;
    if (rc != SAME_OK) return 11;
    if (kind == 1 || kind == 2 || kind == 5) return 12;          // these must never be answered
    ++*answered;
    return margin > 16.0 ? check(xy, n, t, nt) : 13;
}

int main() {
    int answered = 0;
    for (int round = 0; round < 60; ++round) {
        const int kind = round % 6;
        const int64_t n = round < 6 ? round : (int64_t)(3 + uniform() * (round % 7 == 0 ? 6000 : 400));
        const int rc = one_set(n, kind, &answered);
        if (rc) { std::printf("set %d (kind %d, n %lld): %d\n", round, kind, (long long)n, rc); return 1; }
    }
    // a point exactly at the seed triangle's circumcentre (a square's four corners and its centre, among others): refused, not crashed
    {
        std::vector<double> xy;
        for (int gy = 0; gy < 9; ++gy)
            for (int gx = 0; gx < 9; ++gx) { xy.push_back(gx * 2.0); xy.push_back(gy * 2.0); }
        xy.push_back(9.0); xy.push_back(9.0);
        const int64_t n = (int64_t)xy.size() / 2;
        std::vector<int32_t> t((size_t)(2 * n) * 3);
        int64_t n_t = 0;
        if (same_delaunay2d(xy.data(), n, t.data(), 2 * n, &n_t, 16.0, nullptr) != SAME_EUNSURE) { std::printf("centre\n"); return 1; }
    }
    // bad arguments
    int64_t nt = 0;
    int32_t t3[3];
    const double three[6] = {0, 0, 1, 0, 0, 1};
    if (same_delaunay2d(nullptr, 3, t3, 1, &nt, 16.0, nullptr) != SAME_EINVAL || same_delaunay2d(three, 3, t3, 0, &nt, 16.0, nullptr) != SAME_EINVAL ||
        same_delaunay2d(three, 3, t3, 1, &nt, 16.0, nullptr) != SAME_OK || nt != 1) { std::printf("arguments\n"); return 1; }
    // eight threads at once: every thread has its own working arrays
    int bad[8] = {0}, got[8] = {0};
    std::vector<std::thread> threads;
    for (int q = 0; q < 8; ++q)
        threads.emplace_back([q, &bad, &got] {
            std::vector<double> xy(2000);
            uint64_t s = 1234567 + 977 * (uint64_t)q;
            for (int rep = 0; rep < 20 && !bad[q]; ++rep) {
                for (double &v : xy) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) / 9007199254740992.0 * 500; }
                std::vector<int32_t> t(1995 * 3);
                int64_t n_t = 0;
                const int rc = same_delaunay2d(xy.data(), 1000, t.data(), 1995, &n_t, 16.0, nullptr);
                if (rc == SAME_OK) { got[q] += 1; bad[q] = check(xy, 1000, t, n_t); } else if (rc != SAME_EUNSURE) bad[q] = 20;
            }
        });
    for (auto &th : threads) th.join();
    for (int q = 0; q < 8; ++q)
        if (bad[q] || got[q] < 15) { std::printf("thread %d: %d (%d answered)\n", q, bad[q], got[q]); return 1; }
    std::printf("%d sets answered and checked; ok\n", answered);
    return answered >= 15 ? 0 : 1;
}
