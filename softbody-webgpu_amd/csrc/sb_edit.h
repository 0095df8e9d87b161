// sb_edit.h -- an upload that only REMOVED beams (sb_api.hip rewrite_scene_state): match the records that are left against the
// engine's beams.  The caller's new list is a subsequence of the old one (engineMapping.ts:452-459,500-518: a Map that deletes and
// writes what is left in its old order); ANY strictly increasing match of equal records will do -- equal records are interchangeable,
// their state comes from the upload.  In chunks: where each chunk starts in the old list is found one chunk after the other (the shift
// only grows, by the beams removed in between, and a chunk's records use at least as many old ones), the chunks are then matched side
// by side on host threads.  Header-only and HIP-free: tests/edit_check.cpp runs it under AddressSanitizer / ThreadSanitizer.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "sb_tiling.h" // sbt::parallel_ranges

namespace sbe {

// n_new records against n_old (n_new <= n_old).  same_key(u, o): the cheap test (same endpoints) the chunk starts are searched with;
// take(u, o): the full test -- on success it may take the record's state over (it is called once per accepted pair, and for pairs it
// rejects; never for the same o from two threads at a time).  out[u] = the old index matched to new record u; false: not a subsequence.
template <typename SameKey, typename Take>
inline bool match_subsequence(size_t n_new, size_t n_old, size_t chunk, SameKey same_key, Take take, std::vector<uint32_t> &out)
{
    out.assign(n_new, 0u);
    if (n_new > n_old) return false;
    if (n_new == 0) return true;
    const size_t removed = n_old - n_new, nch = (n_new + chunk - 1) / chunk;
    std::vector<size_t> start(nch + 1, n_old);
    size_t op = 0;
    for (size_t k = 0; k < nch; k++) {
        const size_t u0 = k * chunk;
        op = std::max(op, u0); // (the shift is never negative)
        while (op < n_old && op - u0 <= removed && !same_key(u0, op)) op++;
        if (op >= n_old || op - u0 > removed) return false;
        start[k] = op;
        op += std::min(chunk, n_new - u0); // (a chunk's records use at least as many old ones)
    }
    std::atomic<bool> ok{true};
    sbt::parallel_ranges(nch, 1, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1 && ok.load(std::memory_order_relaxed); k++) {
            size_t o = start[k];
            const size_t u1 = std::min((k + 1) * chunk, n_new);
            for (size_t u = k * chunk; u < u1; u++) {
                while (o < start[k + 1] && !take(u, o)) o++;
                if (o >= start[k + 1]) { // (ran into the next chunk's records, or off the end)
                    ok.store(false, std::memory_order_relaxed);
                    return;
                }
                out[u] = (uint32_t)o++;
            }
        }
    });
    if (ok.load()) return true;
    // A chunk ran into its successor's start: with equal keys in a row the cheap test can place a start too early.  One thread, the
    // greedy earliest match over the whole list -- complete for equal-record matching (scenes with several beams between the same two
    // particles only; a lattice never comes here).
    size_t o = 0;
    for (size_t u = 0; u < n_new; u++) {
        while (o < n_old && !take(u, o)) o++;
        if (o >= n_old) return false;
        out[u] = (uint32_t)o++;
    }
    return true;
}

} // namespace sbe
