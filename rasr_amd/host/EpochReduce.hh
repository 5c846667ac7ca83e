// EpochReduce.hh -- host-side C++ mirror of the per-epoch accumulator exchange (header only, plain C ABI underneath).
//
// Reference behaviour being replaced: every `acoustic-model-trainer` process of a partitioned job
// (`partition` / `select-partition`, src/Bliss/CorpusDescription.cc:174-190) writes its own mixture-set estimator file and
// `combine-mixture-set-estimators` (src/Tools/AcousticModelTrainer/AcousticModelTrainer.cc:317-325 ->
// Mm::AbstractMixtureSetEstimator::accumulate(other), src/Mm/AbstractMixtureSetEstimator.cc:173-250) adds the files up.
// With one process per GPU of a node, the sum is ONE RCCL all-reduce over xGMI of ONE flat f64 device buffer:
//
//     AmxHost::EpochReduce red(ctx);                                    // after amx_init
//     red.addStatistics("acc", amx_gmm_accumulator_size(gmm));          // declare the fields ...
//     red.addCounters("counts", nStates);
//     red.addStatistics("score-sum", 1);
//     red.allocate();                                                   // ... then allocate; the pointers are valid from here on
//     double*             acc    = red.statistics("acc");               // kernels accumulate in place
//     unsigned long long* counts = red.counters("counts");              // integer atomics of amx_*_score_stats_dev
//     double*             sum    = red.statistics("score-sum");
//     amx_comm* comm = AmxHost::connect(ctx, rank, world, "/shared/job-1234.amx-id", 600, jobTag);   // id file written by rank 0
//     ... one epoch: amx_gmm_score_stats_dev(..., counts, sum); amx_gmm_accumulate_dev(..., acc); ...
//     red.allReduce(comm);                                              // ONE amx_comm_all_reduce_f64_dev
//     if (rank == 0) { red.download("acc", host); amx_gmm_accumulator_write(gmm, host, "combined.acc"); }
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/amx.h"

namespace AmxHost {

inline void epochCheck(int status, const char* what) {
    if (status != AMX_OK)
        throw std::runtime_error(std::string(what) + ": " + amx_last_error());
}

// Rank 0 creates the communicator id and publishes it as a file (written under a temporary name, then renamed); the other ranks
// wait for the file.  Any other transport of the 128 bytes does as well -- the library only needs every rank to pass the same id.
// A file left behind by an EARLIER job must not be taken for this job's (RCCL has no timeout: ranks holding different ids wait
// for ever): the file carries `jobTag` in front of the id and a rank only accepts a file with its own tag -- pass something unique to
// the run and equal on all ranks (scheduler job id, start time of the launcher).  Rank 0 removes an existing file before it
// publishes and removes its own after amx_comm_init has returned, i.e. after every rank has joined with the id it read.
// The tag is REQUIRED and must not be 0: with a default tag every job carries the same one, and a rank that starts before rank 0 has
// removed last job's file would accept the old id -- the very hang the tag is there to prevent.
inline amx_comm* connect(amx_ctx* ctx, int rank, int world, const std::string& idFile, int timeoutSeconds, uint64_t jobTag) {
    if (jobTag == 0)
        throw std::invalid_argument("AmxHost::connect: jobTag must be a non-zero value unique to this run and equal on all ranks");
    unsigned char id[AMX_COMM_ID_BYTES];
    if (rank == 0) {
        remove(idFile.c_str());
        epochCheck(amx_comm_unique_id(id), "amx_comm_unique_id");
        const std::string tmp = idFile + ".tmp";
        FILE*             f   = fopen(tmp.c_str(), "wb");
        const bool        ok  = f && fwrite(&jobTag, 1, sizeof jobTag, f) == sizeof jobTag && fwrite(id, 1, sizeof id, f) == sizeof id;
        if (f)
            fclose(f);
        if (!ok)
            throw std::runtime_error("cannot write " + tmp);
        if (rename(tmp.c_str(), idFile.c_str()) != 0)
            throw std::runtime_error("cannot publish " + idFile);
    }
    else {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeoutSeconds);
        for (;;) {
            FILE* f = fopen(idFile.c_str(), "rb");
            if (f) {
                uint64_t     tag = 0;
                const size_t nt = fread(&tag, 1, sizeof tag, f), n = fread(id, 1, sizeof id, f);
                fclose(f);
                if (nt == sizeof tag && n == sizeof id && tag == jobTag)
                    break;
            }
            if (std::chrono::steady_clock::now() > deadline)
                throw std::runtime_error("no communicator id of this job in " + idFile);
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
    }
    amx_comm* comm = nullptr;
    epochCheck(amx_comm_init(ctx, rank, world, id, &comm), "amx_comm_init");
    if (rank == 0)
        remove(idFile.c_str());
    return comm;
}

class EpochReduce {
public:
    explicit EpochReduce(amx_ctx* ctx) : ctx_(ctx) {}
    ~EpochReduce() {
        amx_device_free(ctx_, flat_);
        amx_device_free(ctx_, counters_);
    }
    EpochReduce(const EpochReduce&)            = delete;
    EpochReduce& operator=(const EpochReduce&) = delete;

    // declare the fields first, then allocate(); the returned pointers are valid after allocate()
    void addStatistics(const std::string& name, size_t n) { fields_.push_back({name, n, false, flatSize_, 0}); flatSize_ += n; }
    void addCounters(const std::string& name, size_t n) { fields_.push_back({name, n, true, flatSize_, counterSize_}); flatSize_ += n; counterSize_ += n; }
    void allocate() {
        epochCheck(amx_device_malloc(ctx_, flatSize_ * sizeof(double), (void**)&flat_), "amx_device_malloc");
        if (counterSize_)
            epochCheck(amx_device_malloc(ctx_, counterSize_ * sizeof(unsigned long long), (void**)&counters_), "amx_device_malloc");
        clear();
    }
    void clear() {
        std::vector<double> z(flatSize_, 0.0);
        epochCheck(amx_copy_to_device(ctx_, flat_, z.data(), z.size() * sizeof(double)), "amx_copy_to_device");
        if (counterSize_) {
            std::vector<unsigned long long> zc(counterSize_, 0);
            epochCheck(amx_copy_to_device(ctx_, counters_, zc.data(), zc.size() * sizeof(unsigned long long)), "amx_copy_to_device");
        }
        epochCheck(amx_synchronize(ctx_), "amx_synchronize");
    }
    double*             statistics(const std::string& name) { const Field& f = field(name, false); return flat_ + f.offset; }
    unsigned long long* counters(const std::string& name) { const Field& f = field(name, true); return counters_ + f.counterOffset; }
    size_t              size() const { return flatSize_; }   // doubles that travel

    // sum over the ranks: counters into their f64 slots, ONE collective over the whole buffer, counters back
    void allReduce(amx_comm* comm) {
        for (const Field& f : fields_)
            if (f.isCounter)
                epochCheck(amx_counts_to_f64_dev(ctx_, counters_ + f.counterOffset, flat_ + f.offset, f.n), "amx_counts_to_f64_dev");
        epochCheck(amx_comm_all_reduce_f64_dev(comm, flat_, flatSize_), "amx_comm_all_reduce_f64_dev");
        for (const Field& f : fields_)
            if (f.isCounter)
                epochCheck(amx_f64_to_counts_dev(ctx_, flat_ + f.offset, counters_ + f.counterOffset, f.n), "amx_f64_to_counts_dev");
    }
    void download(const std::string& name, double* host) {
        const Field& f = field(name, false);
        epochCheck(amx_copy_to_host(ctx_, host, flat_ + f.offset, f.n * sizeof(double)), "amx_copy_to_host");
    }
    void downloadCounters(const std::string& name, unsigned long long* host) {
        const Field& f = field(name, true);
        epochCheck(amx_copy_to_host(ctx_, host, counters_ + f.counterOffset, f.n * sizeof(unsigned long long)), "amx_copy_to_host");
    }

private:
    struct Field {
        std::string name;
        size_t      n;
        bool        isCounter;
        size_t      offset, counterOffset;
    };
    const Field& field(const std::string& name, bool counter) const {
        for (const Field& f : fields_)
            if (f.name == name && f.isCounter == counter)
                return f;
        throw std::runtime_error("EpochReduce: no such field: " + name);
    }
    amx_ctx*            ctx_;
    std::vector<Field>  fields_;
    size_t              flatSize_ = 0, counterSize_ = 0;
    double*             flat_     = nullptr;
    unsigned long long* counters_ = nullptr;
};

}  // namespace AmxHost
