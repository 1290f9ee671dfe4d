// CPU harness of ampligraph_amd/csrc/kge_group_staging.h (tests/test_group_staging.py): reads a problem from stdin, runs the product's
// staging functions for every replica of a row-sharded group IN ONE THREAD PER REPLICA (as rows_rank does on distinct devices) and
// prints what they staged.
//   in : W rows_per N nq  then nq triples (s p o)
//   out: |U| U...  then per replica: lo n_local, x[3 nq], idx[|U|]
#include <stdio.h>

#include <thread>
#include <vector>

#include "../../ampligraph_amd/csrc/kge_group_staging.h"

int main() {
    long long W, rows_per, N, nq;
    if (scanf("%lld %lld %lld %lld", &W, &rows_per, &N, &nq) != 4) return 2;
    std::vector<int32_t> t((size_t)(3 * nq));
    for (auto& v : t) { int x; if (scanf("%d", &x) != 1) return 2; v = x; }
    std::vector<int32_t> U;
    kge::stage_distinct_rows(t.data(), nq, U);
    std::vector<std::vector<int32_t>> xl((size_t)W), idxl((size_t)W);
    std::vector<std::thread> th;
    for (long long d = 0; d < W; ++d)
        th.emplace_back([&, d]() {
            const long long lo = d * rows_per, hi = (lo + rows_per < N) ? lo + rows_per : N;
            kge::stage_replica(t.data(), nq, U, lo, hi > lo ? hi - lo : 0, xl[(size_t)d], idxl[(size_t)d]);
        });
    for (auto& x : th) x.join();
    printf("%zu", U.size());
    for (int32_t v : U) printf(" %d", v);
    printf("\n");
    for (long long d = 0; d < W; ++d) {
        const long long lo = d * rows_per, hi = (lo + rows_per < N) ? lo + rows_per : N;
        printf("%lld %lld", lo, hi > lo ? hi - lo : 0);
        for (int32_t v : xl[(size_t)d]) printf(" %d", v);
        for (int32_t v : idxl[(size_t)d]) printf(" %d", v);
        printf("\n");
    }
    std::vector<int64_t> off = {5, 5, 9, 12, 12, 20}, lo;
    kge::stage_csr_slice(off.data(), 1, 3, lo);
    printf("%lld %lld %lld %lld\n", (long long)lo[0], (long long)lo[1], (long long)lo[2], (long long)lo[3]);
    return 0;
}
