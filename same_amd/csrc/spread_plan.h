// spread_plan.h -- the planning half of same_dev_alloc_spread (spread.hip): which of the labelled 1 GiB chunks to take and in
// which order to lay them.  Host-only C++ with no HIP in it, so tests/test_spread_plan.py compiles it with g++ on the CPU.
#pragma once
#include <cstddef>
#include <vector>

namespace spread_plan {

// have[c] chunks are available in class c (regions first, then the straddlers' class); take `need` of them as evenly as
// the supplies allow (one at a time from the class taken least so far).  Returns take[c]; its sum is min(need, total).
inline std::vector<size_t> water_fill(const std::vector<size_t> &have, size_t need) {
    std::vector<size_t> take(have.size(), 0);
    for (size_t got = 0; got < need; ++got) {
        int best = -1;
        for (int c = 0; c < (int)have.size(); ++c)
            if (take[c] < have[c] && (best < 0 || take[c] < take[best])) best = c;
        if (best < 0) break;
        ++take[best];
    }
    return take;
}

// does one of the first `regions` classes hold more than max_share_permille of the `need` chunks (one chunk of slack)?
inline bool lopsided(const std::vector<size_t> &take, int regions, size_t need, size_t max_share_permille) {
    size_t mx = 0;
    for (int c = 0; c < regions && c < (int)take.size(); ++c)
        if (take[c] > mx) mx = take[c];
    return mx * 1000 > max_share_permille * need + 1000;
}

// Order in which to lay the chosen chunks: always the class with the most chunks left, but not the class just used when
// another one still has chunks -- for counts (a, b, c) this is a, b, c, a, b, c ... until the smaller ones run out.
// by_class[c] lists the chunk ids of class c; the first take[c] of them are used.
inline std::vector<int> interleave(const std::vector<std::vector<int>> &by_class, const std::vector<size_t> &take) {
    std::vector<size_t> left = take, pos(take.size(), 0);
    std::vector<int> order;
    int prev = -1;
    for (;;) {
        int pick = -1;
        for (int c = 0; c < (int)left.size(); ++c)
            if (left[c] && c != prev && (pick < 0 || left[c] > left[pick])) pick = c;
        if (pick < 0 && prev >= 0 && left[prev]) pick = prev;
        if (pick < 0) break;
        order.push_back(by_class[pick][pos[pick]++]);
        --left[pick];
        prev = pick;
    }
    return order;
}

}  // namespace spread_plan
