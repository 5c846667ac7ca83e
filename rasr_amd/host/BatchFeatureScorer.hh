// rasr_amd/host/BatchFeatureScorer.hh -- header-only C++ mirror of the reference's buffered feature-scorer
// protocol on top of the C ABI (include/amx.h).  This is the code a RASR adapter derives from / copies
// (INTEGRATION.md); it has no RASR dependency so it can be compiled and tested on its own.
//
// Mirrors, with the same names and argument meaning:
//   Mm::FeatureScorer buffered protocol            src/Mm/FeatureScorer.hh:89-137
//   Nn::BatchFeatureScorer ring buffer             src/Nn/BatchFeatureScorer.cc:92-171, BatchFeatureScorer.hh:156-171
//   Mm::FeatureScorer::ContextScorer               src/Mm/FeatureScorer.hh:31-46
// Differences: the batch is evaluated on the GPU for ALL emissions the first time any score of an
// up-to-date buffer is requested (the reference does the same with network_.forward(buffer_)), and the
// [bufferSize x nEmissions] score block STAYS IN HBM: a frame's row (40 kB at 10^4 emissions) crosses PCIe only when the
// decoder first asks for one of its scores, after which score(e) is a plain read of the host row cache (thread-safe once
// fetched).  ContextScorer::scores(list) is an extension for decoders that know their active set: one device gather
// and one small copy for the listed emissions only.  bytesToHost() counts what actually crossed.
#ifndef RASR_AMD_HOST_BATCH_FEATURE_SCORER_HH
#define RASR_AMD_HOST_BATCH_FEATURE_SCORER_HH

#include <cstdio>
#include <cstdlib>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/amx.h"

namespace AmxHost {

// The reference aborts on contract violations (Core/Assertions.hh require()).  Tests define
// AMXHOST_REQUIRE_THROWS to observe them as exceptions instead.
#ifdef AMXHOST_REQUIRE_THROWS
#define amxhost_require(expr)                                                                         \
    do {                                                                                              \
        if (!(expr))                                                                                  \
            throw std::logic_error(std::string("precondition ") + #expr + " violated");               \
    } while (0)
#else
#define amxhost_require(expr)                                                                         \
    do {                                                                                              \
        if (!(expr)) {                                                                                \
            std::fprintf(stderr, "PROGRAM DEFECTIVE: precondition %s violated (%s:%d)\n", #expr, __FILE__, __LINE__); \
            std::abort();                                                                             \
        }                                                                                             \
    } while (0)
#endif

/** amx_init + the arithmetic of THIS translation unit's build (round 6).  The adapter is compiled inside RASR's build, with RASR's
 *  flags: AMX_CONTRACT_OF_THIS_BUILD (include/amx.h) is FMA exactly when the compiler that builds RASR's own scorers fuses their
 *  multiply-adds (-march=native on an FMA host: cmake_resources/CompileOptions.cmake:21-48), so the scores, features and statistics the
 *  library hands back are bit-identical to what the replaced RASR build itself would have computed -- nobody has to remember a tuning
 *  string.  Every handle created on the context afterwards inherits it.  `description` (optional) receives the line to log next to the
 *  reference's own initialisation messages.  Returns amx_status. */
inline int initContext(int device, amx_ctx** ctx, const char** description = nullptr) {
    int r = amx_init(device, ctx);
    if (r != AMX_OK)
        return r;
    r = amx_set_contract(*ctx, AMX_CONTRACT_OF_THIS_BUILD);
    if (r != AMX_OK) {
        amx_destroy(*ctx);
        *ctx = nullptr;
        return r;
    }
    if (description)
        *description = amx_contract_description(AMX_CONTRACT_OF_THIS_BUILD);
    return AMX_OK;
}

typedef float              Score;          // Mm::Score
typedef unsigned           EmissionIndex;  // Mm::EmissionIndex
typedef std::vector<float> FeatureVector;  // Mm::FeatureVector

/** Backend: anything that scores a batch of frames for all emissions. */
class BatchBackend {
public:
    virtual ~BatchBackend() {}
    virtual unsigned nEmissions() const = 0;
    virtual unsigned dimension() const  = 0;
    /** feats [T x dim] row-major -> scores [T x nEmissions]; returns amx_status */
    virtual int score(const float* feats, int T, float* scores) = 0;
    /** the same into a device-resident block owned by the backend (nothing comes back to the host) */
    virtual int scoreResident(const float* feats, int T) = 0;
    /** one row of the resident block -> host [nEmissions] */
    virtual int fetchRow(int row, float* dst) = 0;
    /** scores of (row, emission) pairs of the resident block -> host [n] */
    virtual int fetchPairs(int n, const unsigned* rows, const unsigned* emissions, float* dst) = 0;
};

/** device buffers of a backend's resident block: features in, scores out, grown on demand */
class ResidentBlock {
    amx_ctx* ctx_;
    float *  d_feats_, *d_scores_;
    size_t   capFeats_, capScores_;

public:
    explicit ResidentBlock(amx_ctx* ctx)
            : ctx_(ctx), d_feats_(nullptr), d_scores_(nullptr), capFeats_(0), capScores_(0) {}
    ~ResidentBlock() {
        amx_device_free(ctx_, d_feats_);
        amx_device_free(ctx_, d_scores_);
    }
    amx_ctx* ctx() const { return ctx_; }
    float*   feats() const { return d_feats_; }
    float*   scores() const { return d_scores_; }
    int      reserve(size_t nFeats, size_t nScores) {
        if (nFeats > capFeats_) {
            amx_device_free(ctx_, d_feats_);
            d_feats_  = nullptr;
            capFeats_ = 0;
            if (amx_device_malloc(ctx_, nFeats * sizeof(float), (void**)&d_feats_) != AMX_OK)
                return AMX_ERR_DEVICE;
            capFeats_ = nFeats;
        }
        if (nScores > capScores_) {
            amx_device_free(ctx_, d_scores_);
            d_scores_  = nullptr;
            capScores_ = 0;
            if (amx_device_malloc(ctx_, nScores * sizeof(float), (void**)&d_scores_) != AMX_OK)
                return AMX_ERR_DEVICE;
            capScores_ = nScores;
        }
        return AMX_OK;
    }
};

class GmmBackend : public BatchBackend {
    amx_gmm*      h_;
    int           mode_;
    ResidentBlock block_;
    int           rows_ = 0;  // frames of the resident score block (bounds of fetchPairs)

public:
    /** featureScorerType: "diagonal-maximum" (registered in Mm/Module.cc:83-105) or "diagonal-sum" */
    GmmBackend(amx_ctx* ctx, const amx_gmm_model& model, const std::string& featureScorerType = "diagonal-maximum")
            : h_(nullptr), mode_(featureScorerType == "diagonal-sum" ? AMX_GMM_SUM : AMX_GMM_MAX), block_(ctx) {
        if (amx_gmm_create(ctx, &model, &h_) != AMX_OK)
            throw std::runtime_error(amx_last_error());
    }
    ~GmmBackend() { amx_gmm_destroy(h_); }
    unsigned nEmissions() const { return (unsigned)amx_gmm_n_mixtures(h_); }
    unsigned dimension() const { return (unsigned)amx_gmm_dimension(h_); }
    int      score(const float* f, int T, float* s) { return amx_gmm_score(h_, mode_, f, T, s, nullptr); }
    int      scoreResident(const float* f, int T) {
        int r = block_.reserve((size_t)T * dimension(), (size_t)T * nEmissions());
        if (r == AMX_OK)
            r = amx_copy_to_device(block_.ctx(), block_.feats(), f, (size_t)T * dimension() * sizeof(float));
        rows_ = r == AMX_OK ? T : 0;
        return r == AMX_OK ? amx_gmm_score_dev(h_, mode_, block_.feats(), T, block_.scores(), nullptr) : r;
    }
    int fetchRow(int row, float* dst) {
        return amx_copy_to_host(block_.ctx(), dst, block_.scores() + (size_t)row * nEmissions(), (size_t)nEmissions() * sizeof(float));
    }
    int fetchPairs(int n, const unsigned* rows, const unsigned* emissions, float* dst) {
        return amx_gather_scores(block_.ctx(), block_.scores(), rows_, (int)nEmissions(), n, rows, emissions, dst);
    }
};

class FfnnBackend : public BatchBackend {
    amx_ffnn*     h_;
    ResidentBlock block_;
    int           rows_ = 0;  // frames of the resident score block (bounds of fetchPairs)

public:
    FfnnBackend(amx_ctx* ctx, const amx_ffnn_model& model)
            : h_(nullptr), block_(ctx) {
        if (amx_ffnn_create(ctx, &model, &h_) != AMX_OK)
            throw std::runtime_error(amx_last_error());
    }
    ~FfnnBackend() { amx_ffnn_destroy(h_); }
    unsigned nEmissions() const { return (unsigned)amx_ffnn_output_dim(h_); }
    unsigned dimension() const { return (unsigned)amx_ffnn_input_dim(h_); }
    int      score(const float* f, int T, float* s) { return amx_ffnn_score(h_, f, T, s); }
    int      scoreResident(const float* f, int T) {
        int r = block_.reserve((size_t)T * dimension(), (size_t)T * nEmissions());
        if (r == AMX_OK)
            r = amx_copy_to_device(block_.ctx(), block_.feats(), f, (size_t)T * dimension() * sizeof(float));
        rows_ = r == AMX_OK ? T : 0;
        if (r == AMX_OK)
            r = amx_ffnn_score_dev(h_, block_.feats(), (int)dimension(), T, block_.scores());
        // the pass itself must fail if it cannot be trusted (AMX_PREC_F16MX: a value outside the f16 range), not the next one: the
        // reference's scorer hands out scores of a batch it has computed synchronously (Nn/BatchFeatureScorer.cc:148-171)
        if (r == AMX_OK)
            r = amx_ffnn_wait_dev(h_);
        if (r != AMX_OK)
            rows_ = 0;
        return r;
    }
    int fetchRow(int row, float* dst) {
        return amx_copy_to_host(block_.ctx(), dst, block_.scores() + (size_t)row * nEmissions(), (size_t)nEmissions() * sizeof(float));
    }
    int fetchPairs(int n, const unsigned* rows, const unsigned* emissions, float* dst) {
        return amx_gather_scores(block_.ctx(), block_.scores(), rows_, (int)nEmissions(), n, rows, emissions, dst);
    }
    /** the arithmetic the handle COMPUTES in (AMX_PREC_*): AMX_PREC_F16MX requested on heavy-tailed weights runs AMX_PREC_BF16X3
     *  (amx_ffnn_precision; blockRatio: the statistic that decided).  The adapter logs it next to the network's own messages. */
    int effectivePrecision(double* blockRatio = nullptr) const { return amx_ffnn_precision(h_, blockRatio); }
};

class BatchFeatureScorer;

/** Mm::FeatureScorer::ContextScorer: scores of ONE buffered frame. */
class ContextScorer {
    const BatchFeatureScorer* parent_;
    unsigned                  position_;

public:
    ContextScorer(const BatchFeatureScorer* parent, unsigned position)
            : parent_(parent), position_(position) {}
    inline EmissionIndex nEmissions() const;
    inline Score         score(EmissionIndex e) const;
    /** extension: the scores of a list of emissions of this frame in one device gather (no row copy) */
    inline void scores(const EmissionIndex* emissions, unsigned n, Score* out) const;
};
typedef std::shared_ptr<const ContextScorer> Scorer;  // Core::Ref<const ContextScorer>

/** The buffered scorer.  Caller protocol (Speech/Recognizer.cc:271-281,197-205):
 *    if (isBuffered() && !bufferFilled()) addFeature(f); else feed(getScorer(f));
 *    at segment end: while (!bufferEmpty()) feed(flush());
 */
class BatchFeatureScorer {
    friend class ContextScorer;
    std::unique_ptr<BatchBackend> backend_;
    unsigned                      bufferSize_;
    mutable unsigned              nBufferedFeatures_, currentFeature_;
    mutable std::vector<bool>     scoreComputed_;
    mutable std::vector<bool>     rowFetched_;  // row of the resident block copied into the host row cache
    mutable std::vector<float>    buffer_;  // [bufferSize x dim] row-major (frame major)
    mutable std::vector<float>    scores_;  // host row cache [bufferSize x nEmissions]
    mutable size_t                bytesToHost_;
    unsigned                      dim_, nEmissions_;

    [[noreturn]] static void fail(const char* what) {
        // the reference would criticalError() and exit; the message is amx_last_error()
        std::fprintf(stderr, "amx %s failed: %s\n", what, amx_last_error());
        std::abort();
    }

    /** whole buffer in one batch, exactly like network_.forward(buffer_); the block stays on the device */
    void computeScores() const {
        if (backend_->scoreResident(buffer_.data(), (int)bufferSize_) != AMX_OK)
            fail("batch scoring");
        scoreComputed_.assign(bufferSize_, true);
        rowFetched_.assign(bufferSize_, false);
    }

    void setFeature(unsigned position, const FeatureVector& f) const {
        amxhost_require(position < bufferSize_);
        amxhost_require(f.size() == dim_);
        for (unsigned i = 0; i < dim_; ++i)
            buffer_[(size_t)position * dim_ + i] = f[i];
    }

public:
    /** bufferSize: "buffer-size (and also batch size) for the feature scorer"; the reference default of 8
     *  (Nn/BatchFeatureScorer.cc:21-22) is far too small for a GPU -- use 256..1024. */
    BatchFeatureScorer(std::unique_ptr<BatchBackend> backend, unsigned bufferSize)
            : backend_(std::move(backend)),
              bufferSize_(bufferSize),
              nBufferedFeatures_(0),
              currentFeature_(0),
              scoreComputed_(bufferSize, false),
              rowFetched_(bufferSize, false),
              bytesToHost_(0),
              dim_(backend_->dimension()),
              nEmissions_(backend_->nEmissions()) {
        amxhost_require(bufferSize_ >= 1);
        buffer_.assign((size_t)bufferSize_ * dim_, 0.f);
        scores_.assign((size_t)bufferSize_ * nEmissions_, 0.f);
    }

    EmissionIndex nMixtures() const { return nEmissions_; }
    unsigned      dimension() const { return dim_; }

    bool     isBuffered() const { return true; }
    unsigned bufferSize() const { return bufferSize_; }
    bool     bufferFilled() const { return nBufferedFeatures_ + 1 >= bufferSize_; }  // >= bufferSize_ - 1
    bool     bufferEmpty() const { return nBufferedFeatures_ == 0; }
    /** bytes of scores copied device -> host so far (diagnostics) */
    size_t   bytesToHost() const { return bytesToHost_; }

    void reset() const {
        scoreComputed_.assign(bufferSize_, false);
        rowFetched_.assign(bufferSize_, false);
        nBufferedFeatures_ = 0;
        currentFeature_    = 0;
    }

    void addFeature(const FeatureVector& f) const {
        amxhost_require(!bufferFilled());
        setFeature(nBufferedFeatures_, f);
        scoreComputed_[nBufferedFeatures_] = false;
        rowFetched_[nBufferedFeatures_]    = false;
        nBufferedFeatures_++;
    }

    /** stores f in the slot behind the oldest frame and returns the scorer of the oldest frame */
    Scorer getScorer(const FeatureVector& f) const {
        amxhost_require(bufferFilled());
        unsigned position = currentFeature_ ? (currentFeature_ - 1) % bufferSize_ : bufferSize_ - 1;
        setFeature(position, f);
        scoreComputed_[position] = false;
        rowFetched_[position]    = false;
        Scorer scorer(new ContextScorer(this, currentFeature_));
        currentFeature_ = (currentFeature_ + 1) % bufferSize_;
        return scorer;
    }

    Scorer flush() const {
        amxhost_require(!bufferEmpty());
        Scorer scorer(new ContextScorer(this, currentFeature_));
        currentFeature_ = (currentFeature_ + 1) % bufferSize_;
        nBufferedFeatures_--;
        if (bufferEmpty()) {
            currentFeature_ = 0;
            buffer_.assign(buffer_.size(), 0.f);
        }
        return scorer;
    }

    Score getScore(EmissionIndex e, unsigned position) const {
        amxhost_require(position < bufferSize_);
        amxhost_require(e < nEmissions_);
        if (!scoreComputed_[position])
            computeScores();
        if (!rowFetched_[position]) {  // first score of this frame: its row (and only its row) crosses PCIe
            if (backend_->fetchRow((int)position, &scores_[(size_t)position * nEmissions_]) != AMX_OK)
                fail("score row copy");
            rowFetched_[position] = true;
            bytesToHost_ += (size_t)nEmissions_ * sizeof(float);
        }
        return scores_[(size_t)position * nEmissions_ + e];
    }

    void getScores(const EmissionIndex* emissions, unsigned n, unsigned position, Score* out) const {
        amxhost_require(position < bufferSize_);
        if (!scoreComputed_[position])
            computeScores();
        if (rowFetched_[position]) {
            for (unsigned i = 0; i < n; ++i) {
                amxhost_require(emissions[i] < nEmissions_);
                out[i] = scores_[(size_t)position * nEmissions_ + emissions[i]];
            }
            return;
        }
        std::vector<unsigned> rows(n, position);
        for (unsigned i = 0; i < n; ++i)
            amxhost_require(emissions[i] < nEmissions_);
        if (backend_->fetchPairs((int)n, rows.data(), emissions, out) != AMX_OK)
            fail("score gather");
        bytesToHost_ += (size_t)n * sizeof(float);
    }
};

inline EmissionIndex ContextScorer::nEmissions() const {
    return parent_->nMixtures();
}
inline Score ContextScorer::score(EmissionIndex e) const {
    return parent_->getScore(e, position_);
}
inline void ContextScorer::scores(const EmissionIndex* emissions, unsigned n, Score* out) const {
    parent_->getScores(emissions, n, position_, out);
}

}  // namespace AmxHost
#endif
