// ltm_pclsort.h -- the permutation libstdc++'s std::sort leaves behind for (leaf index, point index) pairs compared by leaf index ONLY, computed
// about twice as fast as std::sort computes it.
//
// Why this exists: pcl::VoxelGrid sorts its `cloud_point_index_idx` pairs with std::sort and a comparator that looks at the leaf index alone
// (voxel_grid.hpp: `operator<` of cloud_point_index_idx), then sums the points of a leaf in the order the UNSTABLE sort left them; with three or
// more points in a leaf the float sum depends on that order.  The order is a property of the whole run of the algorithm, so
// following the reference bit for bit means running the same algorithm: introsort as libstdc++ writes it (bits/stl_algo.h: __introsort_loop with
// depth limit 2 floor(log2 n), __move_median_to_first over first+1 / middle / last-1, __unguarded_partition with the pivot left at `first`,
// heap sort below the depth limit, __final_insertion_sort with threshold 16).  This file performs exactly those element moves, but finds them
// without data-dependent branches:
//   * the Hoare partition swaps the i-th element from the left that is not less than the pivot with the i-th from the right that is not
//     greater, while they have not met -- positions that depend on the data alone, collected block-wise (BlockQuicksort, Edelkamp & Weiss 2016);
//   * the final insertion sort never moves an element out of the <= 16-element segment the partitioning left it in (everything left of a segment is
//     <= everything in it) and is stable, so each segment is sorted stably where the recursion ends (a branch-free rank sort);
//   * the heap-sort fallback is the library's own std::partial_sort(first, last, last).
// Checked against std::sort itself: tests/test_abi.py (random sizes and key ranges, duplicates, sorted / reversed / organ-pipe inputs, McIlroy's
// adversary, which drives std::sort into the fallback) through ltm_debug_pcl_sort_order.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>

namespace ltm_pclsort
{
struct Entry { uint32_t idx, cloud_point_index; };
struct Less { bool operator()(const Entry& a, const Entry& b) const { return a.idx < b.idx; } };

inline void insertion(Entry* first, Entry* last)      // stable: what __final_insertion_sort does to a segment the partitioning left unsorted
{
    // n <= 16: stable rank sort without data-dependent branches -- an element's place is the number of smaller keys plus the number of equal keys before it
    const int n = (int)(last - first);
    if (n < 2) return;
    Entry tmp[16];
    uint32_t key[16];
    for (int i = 0; i < n; ++i) { tmp[i] = first[i]; key[i] = first[i].idx; }
    for (int i = 0; i < n; ++i) {
        int r = 0;
        const uint32_t k = key[i];
        for (int j = 0; j < n; ++j) r += (key[j] < k) | ((key[j] == k) & (j < i));
        first[r] = tmp[i];
    }
}

// std::__unguarded_partition(first, last, pivot): the same swaps in the same order.  The scalar loop swaps the i-th element from the left that is
// not less than the pivot with the i-th element from the right that is not greater, for as long as the former lies left of the latter -- positions
// that depend on the data only, so they can be collected a block at a time without data-dependent branches (BlockQuicksort) as long as the two
// blocks do not overlap; the remainder is finished by the scalar loop from exactly the state it would be in.
inline Entry* partition(Entry* first, Entry* last, const Entry* pivot)
{
    constexpr int B = 64;
    const uint32_t p = pivot->idx;
    uint8_t offl[B], offr[B];
    int nl = 0, sl = 0, nr = 0, sr = 0;     // stops of the current left block still to be swapped: offl[sl .. nl); same on the right
    Entry* lb = first;                      // the current (or next) left block starts here; everything left of it is final
    Entry* rb = last;                       // one past the current (or next) right block; everything from here on is final
    while (last - first > 256 || sl < nl || sr < nr) {
        const bool la = sl < nl, ra = sr < nr;
        if (!la) {
            if ((rb - (ra ? B : 0)) - lb < B) break;
            nl = sl = 0;
            for (int i = 0; i < B; ++i) { offl[nl] = (uint8_t)i; nl += !(lb[i].idx < p); }
            if (nl == 0) { lb += B; continue; }
        }
        if (!ra) {
            if (rb - (lb + B) < B) break;          // the left block is active here
            nr = sr = 0;
            for (int i = 0; i < B; ++i) { offr[nr] = (uint8_t)i; nr += !(p < (rb - 1 - i)->idx); }
            if (nr == 0) { rb -= B; continue; }
        }
        const int m = std::min(nl - sl, nr - sr);
        for (int k = 0; k < m; ++k) std::iter_swap(lb + offl[sl + k], rb - 1 - offr[sr + k]);
        sl += m; sr += m;
        if (sl == nl) lb += B;
        if (sr == nr) rb -= B;
    }
    Entry* f = sl < nl ? lb + offl[sl] : lb;
    Entry* l = sr < nr ? rb - offr[sr] : rb;       // one past the pending right stop
    // the rest -- at most 3 B elements -- the same way in one piece: all stops of the window from either side, pairs swapped while they have not met.
    // Right of the window everything is >= pivot and left of it everything is <= pivot (final or swapped-in elements), which is where the scalar
    // scans would end if a side runs out of stops inside the window
    const int n = (int)(l - f);
    uint8_t L[256], R[256];
    int cl = 0, cr = 0;
    for (int i = 0; i < n; ++i) { L[cl] = (uint8_t)i; cl += !(f[i].idx < p); }
    for (int i = 0; i < n; ++i) { R[cr] = (uint8_t)i; cr += !(p < (l - 1 - i)->idx); }
    Entry* prev_r = l;       // the lowest position on the right known to hold an element >= pivot now (the last swapped one, or the window's end)
    for (int i = 0;; ++i) {
        Entry* lp = i < cl ? f + L[i] : l;
        if (!(lp < prev_r)) return prev_r;          // the left scan reaches a swapped-in (or final) element first: it stops there and the scans have met
        if (i >= cr) return lp;                     // the right scan finds nothing above the left stop
        Entry* rp = l - 1 - R[i];
        if (!(lp < rp)) return lp;
        std::iter_swap(lp, rp);
        prev_r = rp;
    }
}

inline void move_median_to_first(Entry* result, Entry* a, Entry* b, Entry* c)
{
    Less comp;
    if (comp(*a, *b)) {
        if (comp(*b, *c)) std::iter_swap(result, b);
        else if (comp(*a, *c)) std::iter_swap(result, c);
        else std::iter_swap(result, a);
    } else if (comp(*a, *c)) std::iter_swap(result, a);
    else if (comp(*b, *c)) std::iter_swap(result, c);
    else std::iter_swap(result, b);
}

// how often the calling thread's sorts have gone into the heap-sort fallback (the tests want to see that their adversarial input got there)
inline unsigned long& heap_sort_fallbacks() { static thread_local unsigned long n = 0; return n; }

inline void loop(Entry* first, Entry* last, long depth_limit)
{
    while (last - first > 16) {
        if (depth_limit == 0) { ++heap_sort_fallbacks(); std::partial_sort(first, last, last, Less()); return; }
        --depth_limit;
        Entry* mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1);
        Entry* cut = partition(first + 1, last, first);
        loop(cut, last, depth_limit);
        last = cut;
    }
    insertion(first, last);
}

inline void sort(Entry* first, Entry* last)
{
    if (first == last) return;
    long lg = 0;
    for (size_t n = (size_t)(last - first); n > 1; n >>= 1) ++lg;
    loop(first, last, 2 * lg);
}
} // namespace ltm_pclsort
