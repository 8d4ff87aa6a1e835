// exchange_plan.hpp -- the point-to-point schedule of one half-step's exchange between the shards of a fit (SURVEY.md 8e:
// direct placement, every device pushes its updated row block to each peer over the pair's own xGMI link).  A pure function of
// the block boundaries, so that the pairing logic is testable without devices (tests/test_exchange_plan.py through
// cmfrec_hip_exchange_plan); MultiDev::exchange (fit.hip) issues exactly these operations inside one ncclGroup.
#pragma once
#include <cstddef>
#include <vector>

namespace cmfhip {

struct ExchangeOp {
    int dev;            // the device (shard) that issues the operation, on its own communicator and stream
    int peer;           // the other end
    int send;           // 1: ncclSend of the device's own block, 0: ncclRecv of the peer's block
    int first_row;      // first row of the transferred block inside the replica (the sender's rows for a send, the peer's for a receive)
    int rows;           // rows transferred (> 0: empty blocks issue nothing on either end)
};

// bb: D + 1 block boundaries (bb[d] .. bb[d + 1] = the rows device d updates).  Device d visits its peers in the order d + 1,
// d + 2, ... (mod D), so that no two devices start on the same peer; for every peer it sends its own block and receives the
// peer's block into the rows the peer owns.  Every (d -> e) send has exactly one matching receive on e with the same count.
inline std::vector<ExchangeOp> direct_placement_plan(const std::vector<int> &bb)
{
    std::vector<ExchangeOp> ops;
    const int D = (int)bb.size() - 1;
    for (int d = 0; d < D; d++) {
        const int rows_d = bb[d + 1] - bb[d];
        for (int o = 1; o < D; o++) {
            const int e = (d + o) % D;
            const int rows_e = bb[e + 1] - bb[e];
            if (rows_d > 0) ops.push_back(ExchangeOp{d, e, 1, bb[d], rows_d});
            if (rows_e > 0) ops.push_back(ExchangeOp{d, e, 0, bb[e], rows_e});
        }
    }
    return ops;
}

}  // namespace cmfhip
