// C++ client of the epoch exchange (rasr_amd/host/EpochReduce.hh -> amx_comm_* of the C ABI), the way a RASR trainer process would
// link it: no Python, no torch.  Usage: host_epoch_reduce_test <rank> <world> <id-file>.  With world = 1 (the one-GPU box) the sum
// over the ranks is the identity; with one GPU per rank every field must come back as rank-sum.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../rasr_amd/host/EpochReduce.hh"

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s rank world id-file\n", argv[0]);
        return 2;
    }
    const int rank = atoi(argv[1]), world = atoi(argv[2]);
    try {
        amx_ctx* ctx = nullptr;
        AmxHost::epochCheck(amx_init(rank, &ctx), "amx_init");   // one rank per GPU
        if (!amx_comm_available()) {
            printf("RCCL not available\n");
            return 3;
        }
        {
            AmxHost::EpochReduce red(ctx);
            red.addStatistics("acc", 100000);
            red.addCounters("counts", 1000);
            red.addStatistics("score-sum", 1);
            red.allocate();
            std::vector<double>             acc(100000);
            std::vector<unsigned long long> counts(1000);
            for (size_t i = 0; i < acc.size(); ++i)
                acc[i] = (rank + 1) * (0.25 * i - 7.0);
            for (size_t i = 0; i < counts.size(); ++i)
                counts[i] = (unsigned long long)(rank + 1) * (i + (i == 3 ? (1ull << 40) : 0));
            const double sum = -3.5 * (rank + 1);
            AmxHost::epochCheck(amx_copy_to_device(ctx, red.statistics("acc"), acc.data(), acc.size() * 8), "upload");
            AmxHost::epochCheck(amx_copy_to_device(ctx, red.counters("counts"), counts.data(), counts.size() * 8), "upload");
            AmxHost::epochCheck(amx_copy_to_device(ctx, red.statistics("score-sum"), &sum, 8), "upload");
            bool refused = false;   // a zero tag is refused (every job would carry the same one)
            try {
                AmxHost::connect(ctx, rank, world, argv[3], 1, 0);
            } catch (const std::invalid_argument&) {
                refused = true;
            }
            if (!refused)
                throw std::runtime_error("connect accepted jobTag 0");
            amx_comm* comm = AmxHost::connect(ctx, rank, world, argv[3], 120, /*jobTag*/ 0x5eed0001ull);
            if (amx_comm_rank(comm) != rank || amx_comm_world(comm) != world)
                throw std::runtime_error("communicator reports another rank / world size");
            red.allReduce(comm);
            const double f = 0.5 * world * (world + 1);   // sum over ranks of (rank + 1)
            std::vector<double>             acc2(acc.size());
            std::vector<unsigned long long> counts2(counts.size());
            double                          sum2 = 0;
            red.download("acc", acc2.data());
            red.downloadCounters("counts", counts2.data());
            red.download("score-sum", &sum2);
            for (size_t i = 0; i < acc.size(); ++i)
                if (std::fabs(acc2[i] - f * (0.25 * i - 7.0)) > 1e-9 * (1 + std::fabs(acc2[i])))
                    throw std::runtime_error("statistics differ at " + std::to_string(i));
            for (size_t i = 0; i < counts.size(); ++i)
                if (counts2[i] != (unsigned long long)f * (i + (i == 3 ? (1ull << 40) : 0)))
                    throw std::runtime_error("counter differs at " + std::to_string(i));
            if (sum2 != -3.5 * f)
                throw std::runtime_error("score sum differs");
            amx_comm_destroy(comm);
        }
        amx_destroy(ctx);
        printf("rank %d of %d: OK\n", rank, world);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "rank %d: %s\n", rank, e.what());
        return 1;
    }
}
