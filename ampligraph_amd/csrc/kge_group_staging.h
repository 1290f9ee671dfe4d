// Host-side staging of amdkge_session_group_rank for a ROW-SHARDED group (kge_session_group.hip rows_rank): which entity rows a chunk of
// queries needs, where they sit behind every shard, and what each replica is asked to gather.  Pure C++ (no HIP types): compiled into the
// library and, by tests/test_group_staging.py, into a CPU harness -- the logic every replica's host thread runs on a multi-GPU node.
//
// The reference's analogue is the partition loop of evaluate() (/root/reference/ampligraph/latent_features/models/ScoringBasedEmbeddingModel.py:1431-1452):
// there one process walks the entity partitions; here partition d lives on replica d, the rows of the queries' own s / o entities are
// gathered from their owners into scratch rows behind every shard, and the queries are re-indexed into that local index space.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace kge {

// the sorted distinct s / o ids of the chunk's queries tq[0 .. 3 nq): slot j of the scratch rows holds entity U[j] on EVERY replica
inline void stage_distinct_rows(const int32_t* tq, int64_t nq, std::vector<int32_t>& U) {
    U.resize((size_t)(2 * nq));
    for (int64_t i = 0; i < nq; ++i) { U[(size_t)(2 * i)] = tq[3 * i]; U[(size_t)(2 * i + 1)] = tq[3 * i + 2]; }
    std::sort(U.begin(), U.end());
    U.erase(std::unique(U.begin(), U.end()), U.end());
}

// One replica's staging (its own vectors: nothing here is shared with another replica's thread, U is read-only):
//   x   [3 nq]  the chunk's triples in the replica's LOCAL index space: s, o -> n_local + slot (the scratch rows behind the shard)
//   idx [|U|]   per slot the shard-local row to gather (the replica owns global rows [lo, lo + n_local)), -1 where another replica does
inline void stage_replica(const int32_t* tq, int64_t nq, const std::vector<int32_t>& U, int64_t lo, int64_t n_local,
                          std::vector<int32_t>& x, std::vector<int32_t>& idx) {
    auto slot_of = [&](int32_t id) { return (int64_t)(std::lower_bound(U.begin(), U.end(), id) - U.begin()); };
    x.resize((size_t)(3 * nq));
    for (int64_t i = 0; i < nq; ++i) {
        x[(size_t)(3 * i)] = (int32_t)(n_local + slot_of(tq[3 * i]));
        x[(size_t)(3 * i + 1)] = tq[3 * i + 1];
        x[(size_t)(3 * i + 2)] = (int32_t)(n_local + slot_of(tq[3 * i + 2]));
    }
    const int64_t nu = (int64_t)U.size();
    idx.resize((size_t)nu);
    for (int64_t j = 0; j < nu; ++j) {
        const int64_t v = (int64_t)U[(size_t)j] - lo;
        idx[(size_t)j] = (v >= 0 && v < n_local) ? (int32_t)v : -1;
    }
}

// the chunk's slice [q0, q0 + nq] of a CSR offset array, zero-based
inline void stage_csr_slice(const int64_t* off, int64_t q0, int64_t nq, std::vector<int64_t>& lo) {
    lo.resize((size_t)(nq + 1));
    for (int64_t i = 0; i <= nq; ++i) lo[(size_t)i] = off[q0 + i] - off[q0];
}

}  // namespace kge
