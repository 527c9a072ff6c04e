// CPU test of same_amd/csrc/spread_plan.h (compiled and run by tests/test_spread_plan.py).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>

#include "spread_plan.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    using namespace spread_plan;
    // the cases seen on the card
    { auto t = water_fill({37, 50, 0, 0}, 75); CHECK(t[0] == 37 && t[1] == 38 && t[2] == 0); CHECK(!lopsided(t, 3, 75, 500)); }
    { auto t = water_fill({1, 39, 0, 0}, 8); CHECK(t[0] == 1 && t[1] == 7); CHECK(lopsided(t, 3, 8, 500)); }
    { auto t = water_fill({30, 30, 30, 0}, 75); CHECK(t[0] == 25 && t[1] == 25 && t[2] == 25); }
    { auto t = water_fill({75, 0, 0, 0}, 75); CHECK(t[0] == 75); CHECK(lopsided(t, 3, 75, 500)); }
    { auto t = water_fill({3, 3, 0, 2}, 8); CHECK(t[0] + t[1] + t[3] == 8 && t[3] == 2); CHECK(!lopsided(t, 3, 8, 500)); }   // 4 of 8 is "half", within the slack
    { auto t = water_fill({2, 1, 0, 0}, 10); CHECK(t[0] == 2 && t[1] == 1); }                                                  // supply short: takes what there is
    // properties on random supplies
    std::mt19937 rng(7);
    for (int it = 0; it < 2000; ++it) {
        const int classes = 4;
        std::vector<size_t> have(classes);
        for (auto &h : have) h = rng() % 60;
        const size_t total = std::accumulate(have.begin(), have.end(), size_t(0)), need = rng() % 120;
        auto take = water_fill(have, need);
        size_t sum = 0;
        for (int c = 0; c < classes; ++c) { CHECK(take[c] <= have[c]); sum += take[c]; }
        CHECK(sum == std::min(need, total));
        // even: a class holds fewer than another only because it ran out
        for (int a = 0; a < classes; ++a)
            for (int b = 0; b < classes; ++b)
                if (take[a] + 1 < take[b]) CHECK(take[a] == have[a]);
        std::vector<std::vector<int>> by(classes);
        int id = 0;
        for (int c = 0; c < classes; ++c) for (size_t i = 0; i < have[c]; ++i) by[c].push_back(id++);
        auto order = interleave(by, take);
        CHECK(order.size() == sum);
        std::vector<int> cls(id, -1);
        for (int c = 0; c < classes; ++c) for (int x : by[c]) cls[x] = c;
        std::vector<size_t> used(classes, 0);
        std::vector<char> seen(id, 0);
        for (size_t i = 0; i < order.size(); ++i) {
            CHECK(order[i] >= 0 && order[i] < id && !seen[order[i]]);
            seen[order[i]] = 1;
            ++used[cls[order[i]]];
            if (i) {   // the same class twice in a row only when nothing else is left
                if (cls[order[i]] == cls[order[i - 1]]) {
                    size_t others = 0;
                    for (int c = 0; c < classes; ++c) if (c != cls[order[i]]) others += take[c] - used[c];
                    CHECK(others == 0);
                }
            }
        }
        for (int c = 0; c < classes; ++c) CHECK(used[c] == take[c]);
        // the first take[c] chunks of each class are the ones used, in their own order
        for (int c = 0; c < classes; ++c) for (size_t i = 0; i < take[c]; ++i) CHECK(seen[by[c][i]]);
    }
    printf("ok\n");
    return 0;
}
