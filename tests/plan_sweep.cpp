// Host-side sweep of the plan builder (csrc/nfl_plan.cpp) over every field configuration the ABI accepts or rejects,
// forward and backward plans, built with -fsanitize=address,undefined by tests/test_plan_sanitizer_cpu.py: the plan is
// fixed-size tables (NFL_MAX_RT row tiles, NFL_MAX_CHUNKS chunks) filled by loops over the configuration, which is
// exactly where an out-of-bounds write would hide.  Prints "plans ok <n> rejected <m>".
#include <cstdio>
#include <cstring>
#include <initializer_list>

#include "../include/nerf_fl_amd.h"
#include "../nerf_fl_amd/csrc/nfl_plan.h"

int main() {
    int n_ok = 0, n_bad = 0;
    for (int xyz = 0; xyz <= 16; ++xyz)
        for (int dir = 0; dir <= 5; ++dir)
            for (int a = 0; a < 2; ++a)
                for (int t = 0; t < 2; ++t)
                    for (int na : {0, 1, 24, 32, 33, 48, 49})
                        for (int nt : {0, 1, 8, 16, 17}) {
                            nfl_field_desc d;
                            memset(&d, 0, sizeof(d));
                            d.n_emb_xyz = xyz; d.n_emb_dir = dir; d.encode_appearance = a; d.n_a = na;
                            d.encode_transient = t; d.n_tau = nt; d.beta_min = 0.1f;
                            NflPlan* p = new NflPlan;          // heap: redzones around the tables
                            for (int prec = 0; prec < 3; ++prec) (nfl_plan_fill(&d, prec, p) == 0 ? n_ok : n_bad)++;
                            for (int rg = 0; rg < 2; ++rg)
                                for (int bp = 0; bp < 4; ++bp) {
                                    const int rc = nfl_plan_fill_bwd(&d, rg, bp, p);
                                    (rc == 0 ? n_ok : n_bad)++;
                                    if (rc == 0 && (p->n_rt > NFL_MAX_RT || p->n_chunks > NFL_MAX_CHUNKS || p->total_ks <= 0)) {
                                        printf("table overflow not rejected: xyz %d dir %d\n", xyz, dir);
                                        return 1;
                                    }
                                }
                            delete p;
                        }
    printf("plans ok %d rejected %d\n", n_ok, n_bad);
    return 0;
}
