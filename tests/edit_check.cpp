// The subsequence matcher behind uploads that only removed beams (csrc/sb_edit.h), on its own: random lists with duplicate keys,
// random removals, small chunks so that many threads work side by side -- every result must be a strictly increasing match of equal
// records, lists that are NOT subsequences must be refused, and the state a match takes over must be the new record's.  Built with
// AddressSanitizer + UBSan and with ThreadSanitizer by `make hostcheck` (tests/test_hostcheck_cpu.py).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "sb_edit.h"

struct Rec { uint32_t a, b; float param, state; };

static int run_case(uint32_t seed, size_t n_old, double cut, size_t chunk, bool spoil)
{
    std::mt19937 rng(seed);
    std::vector<Rec> olds(n_old);
    for (size_t i = 0; i < n_old; i++) { // few distinct keys: long runs of equal endpoints, some with equal parameters too
        olds[i] = Rec{(uint32_t)(rng() % 40u), (uint32_t)(rng() % 3u), (float)(rng() % 2u), -1.0f};
    }
    std::vector<Rec> news;
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (size_t i = 0; i < n_old; i++)
        if (U(rng) >= cut) {
            Rec r = olds[i];
            r.state = (float)news.size();
            news.push_back(r);
        }
    if (spoil && news.size() > 2) { // not a subsequence any more: a record nobody has
        news[news.size() / 2].a = 1000u;
    }
    std::vector<uint32_t> out;
    const bool ok = sbe::match_subsequence(
        news.size(), n_old, chunk, [&](size_t u, size_t o) { return news[u].a == olds[o].a && news[u].b == olds[o].b; },
        [&](size_t u, size_t o) {
            if (!(news[u].a == olds[o].a && news[u].b == olds[o].b && news[u].param == olds[o].param)) return false;
            olds[o].state = news[u].state;
            return true;
        },
        out);
    if (spoil && news.size() > 2) return ok ? 1 : 0;
    if (!ok) return 5; // (a valid list is never refused: the greedy pass behind the chunks is complete)
    for (size_t u = 0; u < news.size(); u++) {
        const Rec &o = olds[out[u]];
        if (u && out[u] <= out[u - 1]) return 2;
        if (o.a != news[u].a || o.b != news[u].b || o.param != news[u].param) return 3;
        if (o.state != news[u].state) return 4;
    }
    return -1; // matched
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 200;
    int matched = 0, refused = 0;
    for (int c = 0; c < cases; c++) {
        const size_t n_old = 200 + (size_t)(c * 37 % 5000), chunk = (size_t)8 << (c % 6);
        const double cut = (c % 7) * 0.02;
        const int r = run_case(1000u + (uint32_t)c, n_old, cut, chunk, false);
        if (r > 0) { fprintf(stderr, "case %d: error %d\n", c, r); return 1; }
        if (r == -1) matched++; else refused++;
        if (run_case(5000u + (uint32_t)c, n_old, cut, chunk, true) != 0) { fprintf(stderr, "case %d: a list that is no subsequence was accepted\n", c); return 1; }
    }
    printf("edit_check: %d lists matched, %d refused, %d spoiled lists refused\n", matched, refused, cases);
    if (refused) return 1;
    return 0;
}
