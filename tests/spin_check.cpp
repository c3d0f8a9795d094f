// Host build of the kernels' bounded wait (gym-exploration-2d_amd/csrc/cagym_spin.h): tests/test_laser_audit.py compiles and runs this.
#include <cstdio>
#include "cagym_spin.h"

int main() {
    int fails = 0;
    // the counter arrives after 5 polls
    {
        int polls = 0, pauses = 0;
        const bool ok = cagym_bounded_wait([&]() { return polls++ >= 5 ? 3 : 0; }, 3, 100u, [&]() { pauses++; });
        if (!ok || pauses != 5 || polls != 6) { printf("arrive: ok %d pauses %d polls %d\n", ok, pauses, polls); fails++; }
    }
    // it never arrives: exactly `limit` pauses, limit + 1 looks, false
    {
        int polls = 0, pauses = 0;
        const bool ok = cagym_bounded_wait([&]() { polls++; return 2; }, 3, 1000u, [&]() { pauses++; });
        if (ok || pauses != 1000 || polls != 1001) { printf("never: ok %d pauses %d polls %d\n", ok, pauses, polls); fails++; }
    }
    // it arrives during the last pause: the final look sees it
    {
        int v = 0, pauses = 0;
        const bool ok = cagym_bounded_wait([&]() { return v; }, 1, 4u, [&]() { if (++pauses == 4) v = 1; });
        if (!ok || pauses != 4) { printf("last: ok %d pauses %d\n", ok, pauses); fails++; }
    }
    // already there: no pause at all; limit 0 still looks once
    {
        int pauses = 0;
        if (!cagym_bounded_wait([&]() { return 7; }, 7, 10u, [&]() { pauses++; }) || pauses) { printf("already\n"); fails++; }
        if (!cagym_bounded_wait([&]() { return 7; }, 7, 0u, [&]() { pauses++; }) || pauses) { printf("limit0\n"); fails++; }
        if (cagym_bounded_wait([&]() { return 6; }, 7, 0u, [&]() { pauses++; })) { printf("limit0 false\n"); fails++; }
    }
    if (CAGYM_SPIN_LIMIT < (1u << 16)) { printf("limit too small for a legitimate wait\n"); fails++; }
    if (!fails) printf("spin_check ok\n");
    return fails;
}
