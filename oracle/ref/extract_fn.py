#!/usr/bin/env python3
"""oracle/ref/extract_fn.py -- FUNCTION-TEXT pins: reference member functions compiled stand-alone.  TEST INFRASTRUCTURE.

Some arithmetic of the hot path lives in translation units that cannot be built here as a whole (anything that reaches
Core/Configuration.hh needs boost, which the image lacks).  For a function whose BODY touches nothing but its arguments, the
text of the definition is taken from where it lies under /root/reference at build time -- by line range, checked against a
SHA-256 so that a moved or edited function fails the build instead of silently pinning something else -- and written into
oracle/_ref/gen/ between a class shell that only declares the member (the real class declaration pulls in the configuration
headers) and a C entry point.  Nothing of the reference's text is stored in this repository: the generated file exists only under
oracle/_ref/ (git-ignored), only where the reference tree is mounted, and only while it is being compiled (the Makefile removes it
once the objects are built: what stays is objects and the two libraries).

What such a pin proves, and what it does not: the function's own operations, in the reference's own words, compiled with the
flag sets of oracle/ref/Makefile (so the compiler's contraction decisions are the real ones); NOT the class around it.

    python3 extract_fn.py <name> <out.cc>
"""
import hashlib
import sys

REF = "/root/reference/src"

DECLS_NORMALIZATION = '''
// ---- shell: the class declarations of Signal/Normalization.hh:35-179 (they need the template above).  The three classes that are
// Core::Components there (for warning() / criticalError() and a configuration constructor) get a base with those two as no-ops.
namespace Signal {
struct ComponentShell {
    void warning(const char*, ...) const {}
    void criticalError(const char*, ...) const {}
};
class Normalization {
public:
    typedef f32                                Value;
    typedef f64                                Sum;
    typedef Flow::DataPtr<Flow::Vector<Value>> Frame;
protected:
    SlidingWindow<Frame> slidingWindow_;
    u32                  length_;
    u32                  right_;
    Sum                  sumWeight_;
    bool                 changed_;
private:
    void normalize(Frame& out);
protected:
    virtual bool init(size_t dimension) { return true; }
    virtual void updateStatistics(const Frame& add, const Frame& remove);
    virtual void finalize();
    virtual void apply(Frame& out) = 0;
public:
    Normalization();
    virtual ~Normalization() {}
    bool         init(size_t length, size_t right, size_t dimension);
    bool         update(const Frame& in, Frame& out);
    bool         flush(Frame& out);
    virtual void reset();
};
class LevelNormalization : public Normalization {
    typedef Normalization Precursor;
protected:
    size_t index_;
    Value  max_;
    virtual void finalize();
    virtual void apply(Frame& out);
public:
    LevelNormalization(size_t index) : index_(index), max_(0) {}
};
class MeanNormalization : public Normalization {
    typedef Normalization Precursor;
protected:
    std::vector<Sum>   sum_;
    std::vector<Value> mean_;
    virtual bool init(size_t dimension);
    virtual void reset();
    virtual void updateStatistics(const Frame& add, const Frame& remove);
    virtual void finalize();
    virtual void apply(Frame& out);
};
class MeanAndVarianceNormalization : public ComponentShell, public MeanNormalization {
    typedef MeanNormalization Precursor;
protected:
    std::vector<Sum>   sumSquare_;
    std::vector<Value> standardDeviation_;
    virtual bool init(size_t dimension);
    virtual void reset();
    virtual void updateStatistics(const Frame& add, const Frame& remove);
    virtual void finalize();
    virtual void apply(Frame& out);
};
class MeanAndVarianceNormalization1D : public ComponentShell, public Normalization {
    typedef Normalization Precursor;
protected:
    Sum   sum_;
    Sum   sumSquare_;
    Value mean_;
    Value standardDeviation_;
    virtual bool init(size_t dimension);
    virtual void reset();
    virtual void updateStatistics(const Frame& add, const Frame& remove);
    virtual void finalize();
    virtual void apply(Frame& out);
};
class DivideByMean : public ComponentShell, public MeanNormalization {
    typedef MeanNormalization Precursor;
protected:
    virtual void finalize();
    virtual void apply(Frame& out);
};
}  // namespace Signal
using namespace Signal;
using namespace Core;
using namespace Flow;
// ---- reference text (Signal/Normalization.cc) ----
'''

# name -> (file, [(first line, last line (inclusive)) or (another file, first, last), ...], sha256 of those lines concatenated,
#          text in front, text behind)
SPECS = {
    # Mm::GaussDiagonalMaximumFeatureScorer::distance (both the __SSE3__ branch and the plain one; the flag sets of the Makefile
    # define __SSE3__, so the first is what gets compiled -- as in the reference's own build, CompileOptions.cmake:21-26)
    "gdm_distance": (
        "Mm/GaussDiagonalMaximumFeatureScorer.cc", [(116, 142), (144, 218), (230, 298)],
        "5c7a6289ede81c709f94063ae8f368a1dc61fe592551827da305b7b239578bf1",
        """#include <Core/Types.hh>
#include <Core/Assertions.hh>
#include <Mm/Types.hh>
#include <algorithm>
#include <cmath>
#include <vector>
#ifdef __SSE3__
#include <pmmintrin.h>
#include <xmmintrin.h>
#endif
namespace Mm {
// shell.  The members whose definitions follow -- calculateScoreAndDensity (the f64 combine, the strict comparison, the 0.5), distance,
// and the log-add scorer's three members -- read four tables through the accessors of the element classes
// (Mm/MixtureFeatureScorerElement.hh:24-38, Mm/GaussDensity.hh, Mm/CovarianceFeatureScorerElement.hh:21-50: all of them sit behind
// Core/Configuration.hh).  Re-declared here with the same accessor names, result types and storage types; nothing of their logic
// (operator=, scale) is involved.
class MixtureFeatureScorerElement {
public:
    std::vector<DensityIndex> densityIndices_;
    std::vector<Score>        minus2LogWeights_;
    size_t                    nDensities() const { return densityIndices_.size(); }
    DensityIndex              densityIndex(size_t dns) const { return densityIndices_[dns]; }
    const std::vector<Score>& minus2LogWeights() const { return minus2LogWeights_; }
};
class GaussDensity {
public:
    MeanIndex       meanIndex_;
    CovarianceIndex covarianceIndex_;
    MeanIndex       meanIndex() const { return meanIndex_; }
    CovarianceIndex covarianceIndex() const { return covarianceIndex_; }
};
class CovarianceFeatureScorerElement {
public:
    std::vector<VarianceType> inverseSquareRootDiagonal_;
    Score                     logNormalizationFactor_;
    const std::vector<VarianceType>& inverseSquareRootDiagonal() const { return inverseSquareRootDiagonal_; }
    Score                            logNormalizationFactor() const { return logNormalizationFactor_; }
};
typedef std::vector<MeanType> Mean;
class AssigningFeatureScorer {
public:
    struct ScoreAndBestDensity {   // Mm/AssigningFeatureScorer.hh:29-32
        Score            score;
        DensityInMixture bestDensity;
    };
    class CachedAssigningContextScorer {
    public:
        virtual ~CachedAssigningContextScorer() {}
    };
};
// Mm/GaussDiagonalMaximumFeatureScorer.hh:28-62
class GaussDiagonalMaximumFeatureScorer : public AssigningFeatureScorer {
public:
    class Context : public CachedAssigningContextScorer {
    public:
        std::vector<FeatureType> featureVector_;
    };
    std::vector<MixtureFeatureScorerElement>    mixtureTable_;
    std::vector<GaussDensity>                   densityTable_;
    std::vector<Mean>                           meanTable_;
    std::vector<CovarianceFeatureScorerElement> covarianceTable_;
    virtual ~GaussDiagonalMaximumFeatureScorer() {}
    virtual ScoreAndBestDensity calculateScoreAndDensity(const CachedAssigningContextScorer* cs, MixtureIndex mixtureIndex) const;
    Score distance(const std::vector<FeatureType>& feature, const std::vector<MeanType>& mean,
                   const std::vector<VarianceType>& inverseSquareRootVar) const;
};
// Mm/GaussDiagonalMaximumFeatureScorer.hh:96-115
class GaussDiagonalSumFeatureScorer : public GaussDiagonalMaximumFeatureScorer {
public:
    typedef GaussDiagonalMaximumFeatureScorer Precursor;
    mutable const CachedAssigningContextScorer* lastContext_      = nullptr;
    mutable MixtureIndex                        lastMixtureIndex_ = 0;
    mutable Score*                              scores_           = nullptr;
    mutable size_t                              nDensities_       = 0;
    void   calculateScoresAndNumberOfDensities(const CachedAssigningContextScorer* cs, MixtureIndex mixtureIndex) const;
    size_t maximumNumberOfDensities() const;
    virtual ScoreAndBestDensity calculateScoreAndDensity(const CachedAssigningContextScorer* cs, MixtureIndex mixtureIndex) const;
    virtual void calculateDensityPosteriorProbabilities(const CachedAssigningContextScorer*, Score denominator, EmissionIndex e,
                                                        std::vector<Mm::Weight>& result) const;
};
}  // namespace Mm
using namespace Mm;
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
// ---- end of reference text ----
extern "C" float ref_gdm_distance(const float* x, const float* mu, const float* isr, int dim) {
    // the reference's containers: std::vector<f32> (glibc's allocator returns 16-byte aligned blocks, which _mm_load_ps needs)
    std::vector<Mm::FeatureType>  f(x, x + dim);
    std::vector<Mm::MeanType>     m(mu, mu + dim);
    std::vector<Mm::VarianceType> v(isr, isr + dim);
    return Mm::GaussDiagonalMaximumFeatureScorer().distance(f, m, v);
}
// ONE mixture of nd densities, every density with its own mean and covariance entry: m2lw [nd] (f32, as MixtureFeatureScorerElement keeps
// them), lognorm [nd] (f32), means / isr [nd x dim].  mode 0: GaussDiagonalMaximumFeatureScorer::calculateScoreAndDensity, mode 1: the
// log-add scorer's; posteriors (nullable, [nd], f64): calculateDensityPosteriorProbabilities with the score as denominator.
extern "C" void ref_gdm_score(int mode, const float* x, int dim, int nd, const float* m2lw, const float* lognorm, const float* means,
                              const float* isr, float* score, unsigned* best, double* posteriors) {
    Mm::GaussDiagonalSumFeatureScorer s;   // (its base part serves mode 0)
    s.mixtureTable_.resize(1);
    for (int k = 0; k < nd; ++k) {
        s.mixtureTable_[0].densityIndices_.push_back((Mm::DensityIndex)k);
        s.mixtureTable_[0].minus2LogWeights_.push_back(m2lw[k]);
        Mm::GaussDensity d;
        d.meanIndex_ = (Mm::MeanIndex)k;
        d.covarianceIndex_ = (Mm::CovarianceIndex)k;
        s.densityTable_.push_back(d);
        s.meanTable_.push_back(Mm::Mean(means + (size_t)k * dim, means + (size_t)(k + 1) * dim));
        Mm::CovarianceFeatureScorerElement c;
        c.inverseSquareRootDiagonal_.assign(isr + (size_t)k * dim, isr + (size_t)(k + 1) * dim);
        c.logNormalizationFactor_ = lognorm[k];
        s.covarianceTable_.push_back(c);
    }
    std::vector<Mm::Score> cache((size_t)(nd > 0 ? nd : 1));
    s.scores_ = cache.data();
    Mm::GaussDiagonalMaximumFeatureScorer::Context ctx;
    ctx.featureVector_.assign(x, x + dim);
    Mm::AssigningFeatureScorer::ScoreAndBestDensity r =
            mode == 0 ? s.Mm::GaussDiagonalMaximumFeatureScorer::calculateScoreAndDensity(&ctx, 0) : s.calculateScoreAndDensity(&ctx, 0);
    *score = r.score;
    *best  = r.bestDensity;
    if (posteriors && mode == 1) {
        std::vector<Mm::Weight> p;
        s.calculateDensityPosteriorProbabilities(&ctx, r.score, 0, p);
        for (int k = 0; k < nd; ++k)
            posteriors[k] = p[k];
    }
}
"""),
    # Signal::Regression::regressFirstOrder / regressSecondOrder (signal-regression): the class declaration
    # (Signal/Regression.hh:75-85) sits behind Flow/Merger.hh -> Flow/Node.hh -> Core/Configuration.hh (boost); its two members work on
    # std::vector<f32> only
    "regression": (
        "Signal/Regression.cc", [(24, 65)],
        "a190f93de931f220e600ac3bedca896a6db5066f6e4c33e9d2266469636a0c7d",
        """#include <Core/Types.hh>
#include <Core/Assertions.hh>
#include <vector>
namespace Signal {
// shell: the two members and the Frame type of Signal/Regression.hh:75-85
class Regression {
protected:
    typedef std::vector<f32> Frame;
public:
    void regressFirstOrder(const std::vector<const Frame*>& in, Frame& out);
    void regressSecondOrder(const std::vector<const Frame*>& in, Frame& out);
};
}  // namespace Signal
using namespace Signal;
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
// ---- end of reference text ----
// in: n_in frames of dim floats (row-major), out: dim floats
extern "C" void ref_regression(int order, const float* in, int n_in, int dim, float* out) {
    std::vector<std::vector<f32>>        frames(n_in);
    std::vector<const std::vector<f32>*> ptr(n_in);
    for (int i = 0; i < n_in; ++i) {
        frames[i].assign(in + (size_t)i * dim, in + (size_t)(i + 1) * dim);
        ptr[i] = &frames[i];
    }
    std::vector<f32> o(dim);
    Signal::Regression r;
    if (order == 1)
        r.regressFirstOrder(ptr, o);
    else
        r.regressSecondOrder(ptr, o);
    for (int c = 0; c < dim; ++c)
        out[c] = o[c];
}
"""),
    # Window functions and the two integration classes of the gammatone front end, in ONE generated file (they share the window classes):
    #  * Signal::WindowFunction::setLength, the rectangular, Hamming and Hanning init() of Signal/WindowFunction.cc:
    #    the window tables, f64 arithmetic stored as f32;
    #  * Signal::TemporalIntegration (f4; whole class as defined in TemporalIntegration.cc: init() = rint of seconds x rate, transform() =
    #    the window-weighted sum of |x| over the frame) on the reference's own Signal::TimeWindowBuffer;
    #  * Signal::SpectralIntegration (f4; whole class as defined in SpectralIntegration.cc: apply()).
    # The class declarations (Signal/WindowFunction.hh:30-135 needs Core/Choice.hh / Core/Parameter.hh, the integration headers need
    # SlidingAlgorithmNode.hh / Flow/Node.hh -> Core/Configuration.hh, boost) are re-declared member for member.
    "windows": (
        "Signal/WindowFunction.cc", [(58, 63), (68, 73), (92, 101), (106, 120), ("Signal/TemporalIntegration.cc", 22, 81),
                                     ("Signal/SpectralIntegration.cc", 25, 75)],
        "c0c6c0d89cc60574d15f9c903529538d405a830985f8f0d5fe805dff2f157bb4",
        """#include <Core/Types.hh>
#include <Core/Assertions.hh>
#include <Flow/Vector.hh>
#include <Signal/TimeWindowBuffer.hh>
#include <cmath>
#include <vector>
namespace Signal {
// Signal/WindowFunction.hh:30-78 without the Core::Choice statics and the factory
class WindowFunction {
public:
    typedef f32 Float;
protected:
    std::vector<Float> window_;
    bool               needInit_;
    virtual bool init() { return !(needInit_ = false); }
public:
    WindowFunction() : needInit_(true) {}
    virtual ~WindowFunction() {}
    void setLength(u32 l);
    u32  length() { return window_.size(); }
    const std::vector<Float>& getWindow() {
        if (needInit_)
            init();
        return window_;
    }
};
class RectangularWindowFunction : public WindowFunction {
protected:
    virtual bool init();
};
class HammingWindowFunction : public WindowFunction {
protected:
    virtual bool init();
};
class HanningWindowFunction : public WindowFunction {
public:
    HanningWindowFunction(bool periodic) : periodic_(periodic) {}
protected:
    virtual bool init();
private:
    bool periodic_;
};
// Signal/TemporalIntegration.hh:33-65
class TemporalIntegration : public TimeWindowBuffer<Flow::Vector<f32>> {
public:
    typedef TimeWindowBuffer<Flow::Vector<f32>>       Precursor;
    typedef TimeWindowBuffer<Flow::Vector<f32>>::Time Time;
    typedef Flow::Vector<f32>                         Sample;
private:
    Time            lengthInS_;
    Time            shiftInS_;
    WindowFunction* windowFunction_;
protected:
    virtual void init();
    virtual void transform(Flow::Vector<Sample>& out);
public:
    TemporalIntegration();
    virtual ~TemporalIntegration();
    void setWindowFunction(WindowFunction* windowFunction);
    void setSampleRate(f64 sampleRate);
    void setLengthInS(Time length);
    Time lengthInS() const { return lengthInS_; }
    void setShiftInS(Time shift);
    Time shiftInS() const { return shiftInS_; }
};
// Signal/SpectralIntegration.hh:34-66
class SpectralIntegration {
public:
    typedef Flow::Time        Time;
    typedef Flow::Vector<f32> Sample;
private:
    u32             length_;
    u32             shift_;
    WindowFunction* windowFunction_;
protected:
    virtual void init();
    void         apply(const Flow::Vector<Sample>& in, Flow::Vector<Sample>& out);
public:
    SpectralIntegration();
    virtual ~SpectralIntegration();
    void setWindowFunction(WindowFunction* windowFunction);
    void setLength(u32 length);
    u32  length() const { return length_; }
    void setShift(u32 shift);
    u32  shift() const { return shift_; }
};
}  // namespace Signal
using namespace Signal;
using Flow::Vector;
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
// ---- end of reference text ----
namespace {
Signal::WindowFunction* make_window(int type) {  // 0 Hanning, 1 periodic Hanning, 2 rectangular, 3 Hamming
    if (type == 2)
        return new Signal::RectangularWindowFunction;
    if (type == 3)
        return new Signal::HammingWindowFunction;
    return new Signal::HanningWindowFunction(type == 1);
}
struct TiProbe : Signal::TemporalIntegration {
    void run(Flow::Vector<Sample>& v) { transform(v); }
    void initialise() { init(); }
};
struct SiProbe : Signal::SpectralIntegration {
    void run(const Flow::Vector<Sample>& in, Flow::Vector<Sample>& out) { apply(in, out); }
};
}  // namespace
extern "C" int ref_window_table(int type, int length, float* out) {
    Signal::WindowFunction* w = make_window(type);
    w->setLength((u32)length);
    const std::vector<f32>& t = w->getWindow();
    int rc = (int)t.size() == length ? 0 : -1;
    for (int i = 0; i < length && i < (int)t.size(); ++i)
        out[i] = t[i];
    delete w;
    return rc;
}
extern "C" int ref_hamming_window(int length, float* out) { return ref_window_table(3, length, out); }
// TemporalIntegration::init: the frame length and shift in samples for a length / shift in seconds
extern "C" void ref_temporal_integration_lengths(double length_s, double shift_s, double sample_rate, unsigned* length, unsigned* shift) {
    TiProbe t;
    t.setWindowFunction(make_window(0));
    t.setSampleRate(sample_rate);
    t.setLengthInS(length_s);
    t.setShiftInS(shift_s);
    t.initialise();
    *length = t.length();
    *shift  = t.shift();
}
// TemporalIntegration::transform on ONE frame [rows x channels] -> out [channels]
extern "C" void ref_temporal_integration(int window, const float* frame, int rows, int channels, float* out) {
    TiProbe t;
    t.setWindowFunction(make_window(window));
    Flow::Vector<Flow::Vector<f32>> v;
    for (int i = 0; i < rows; ++i)
        v.push_back(Flow::Vector<f32>(frame + (size_t)i * channels, frame + (size_t)(i + 1) * channels));
    t.run(v);
    for (int ch = 0; ch < channels; ++ch)
        out[ch] = v[0][ch];
}
// SpectralIntegration::apply on n_frames rows of `channels` values -> out [n_frames x ((channels - length) / shift + 1)]
extern "C" int ref_spectral_integration(int window, int length, int shift, const float* in, int n_frames, int channels, float* out) {
    SiProbe s;
    s.setWindowFunction(make_window(window));
    s.setLength((u32)length);
    s.setShift((u32)shift);
    Flow::Vector<Flow::Vector<f32>> v, o;
    for (int i = 0; i < n_frames; ++i)
        v.push_back(Flow::Vector<f32>(in + (size_t)i * channels, in + (size_t)(i + 1) * channels));
    s.run(v, o);
    const int oc = o.empty() ? 0 : (int)o[0].size();
    for (int i = 0; i < n_frames; ++i)
        for (int c = 0; c < oc; ++c)
            out[(size_t)i * oc + c] = o[i][c];
    return oc;
}
"""),
    # Mm::BatchFloatFeatureScorer::fillScoreCacheTpl ("batch-diagonal-maximum-float"): the SSE loop over the
    # pre-scaled means and features, the horizontal sum, the minimum over the densities and the final 0.5.  The class declaration
    # (Mm/BatchFeatureScorer.hh:105-260) is a FeatureScorer (Core/Component -> Core/Configuration.hh -> boost); the shell declares the
    # members the text reads, with the reference's types (bufferSize_ is an s32 there)
    "batch_float_fill": (
        "Mm/BatchFeatureScorer.cc", [(207, 253)],
        "94a3cac962622929f0eba5d4c32d819df2a4240fd652ebc3f7ca4c61cf63b34e",
        """#include <Core/Types.hh>
#include <Mm/Types.hh>
#include <vector>
#include <xmmintrin.h>
namespace Mm {
// shell: Mm/BatchFeatureScorer.hh:162-199,234-255
class BatchFloatFeatureScorer {
public:
    struct AllDensitySelector {
        bool operator()(size_t, size_t) const { return true; }
    };
    static const size_t         BlockSize;
    std::vector<size_t>         offsets_;
    mutable std::vector<bool>   cached_;
    mutable f32*                scores_;
    u32                         paddedDimension_, dimension_;
    s32                         bufferSize_;
    mutable f32*                features_;
    f32*                        means_;
    f32*                        constants_;
    template<class DensitySelector>
    void fillScoreCacheTpl(EmissionIndex e, u32 featureIndex, u32 length, const DensitySelector& selector) const;
};
const size_t BatchFloatFeatureScorer::BlockSize = 8;   // Mm/BatchFeatureScorer.cc:129
}  // namespace Mm
using namespace Mm;
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
// ---- end of reference text ----
// ONE mixture of n_dens densities: means [n_dens x pdim] and features [T x pdim] already multiplied by 1 / sigma, pdim a multiple of 8
// (init() / setFeature of the reference, restated by the caller), constants [n_dens]; scores [T]
extern "C" void ref_batch_float_fill(const float* means, const float* constants, int n_dens, const float* features, int T, int pdim, float* scores) {
    Mm::BatchFloatFeatureScorer s;
    s.offsets_         = {0, (size_t)n_dens};
    s.bufferSize_      = T;
    s.paddedDimension_ = (u32)pdim;
    s.dimension_       = (u32)pdim;
    s.cached_.assign((size_t)T, false);
    // 16-byte aligned copies (_mm_load_ps)
    float *m = nullptr, *f = nullptr, *c = nullptr, *o = nullptr;
    posix_memalign((void**)&m, 16, sizeof(float) * (size_t)n_dens * pdim);
    posix_memalign((void**)&f, 16, sizeof(float) * (size_t)T * pdim);
    posix_memalign((void**)&c, 16, sizeof(float) * (size_t)(n_dens + 4));
    posix_memalign((void**)&o, 16, sizeof(float) * (size_t)(T + 4));
    for (size_t i = 0; i < (size_t)n_dens * pdim; ++i) m[i] = means[i];
    for (size_t i = 0; i < (size_t)T * pdim; ++i) f[i] = features[i];
    for (int i = 0; i < n_dens; ++i) c[i] = constants[i];
    s.means_ = m; s.features_ = f; s.constants_ = c; s.scores_ = o;
    s.fillScoreCacheTpl<Mm::BatchFloatFeatureScorer::AllDensitySelector>(0, 0, (u32)T, Mm::BatchFloatFeatureScorer::AllDensitySelector());
    for (int t = 0; t < T; ++t) scores[t] = o[t];
    free(m); free(f); free(c); free(o);
}
"""),
    # Signal::Preemphasis: constructor, init, setAlpha, setSampleRate and apply -- the whole class except its
    # declaration, which sits in Signal/Preemphasis.hh behind SleeveNode -> Flow/Node.hh -> Core/Configuration.hh (boost).  The shell
    # re-declares the class member for member (Signal/Preemphasis.hh:29-53); Flow::Vector / Flow::Timestamp are the reference's own
    # (Flow/Vector.hh, Flow/Timestamp.cc is one of the translation units of libref)
    "preemphasis": (
        "Signal/Preemphasis.cc", [(23, 74)],
        "050ca20394e672246a92a01979b299682fd1c70b58cfab6baeb945220264e729",
        """#include <Core/Types.hh>
#include <Core/Assertions.hh>
#include <Flow/Vector.hh>
namespace Signal {
class Preemphasis {
private:
    f32        alpha_;
    f32        previous_;
    Flow::Time previousEndTime_;
    f64        sampleRate_;
    bool       needInit_;
    void       init(f32 initialValue);
public:
    Preemphasis();
    void setAlpha(f32 alpha);
    void setSampleRate(f64 sampleRate);
    void reset(void) { needInit_ = true; }
    void apply(Flow::Vector<f32>& v);
};
}  // namespace Signal
using namespace Signal;
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
// ---- end of reference text ----
// x [n] in blocks of `block` samples with contiguous time stamps, except that block number `gap_at` (if >= 0) starts one second late
// (the node then restarts: previous_ = its first sample); out [n]
extern "C" void ref_preemphasis(float alpha, double sample_rate, const float* x, long n, int block, int gap_at, float* out) {
    Signal::Preemphasis p;
    p.setAlpha(alpha);
    p.setSampleRate(sample_rate);
    double shift = 0;
    int    k     = 0;
    for (long i0 = 0; i0 < n; i0 += block, ++k) {
        const long len = (n - i0 < block) ? n - i0 : block;
        Flow::Vector<f32> v(x + i0, x + i0 + len);
        if (k == gap_at)
            shift = 1.0;
        v.setStartTime((double)i0 / sample_rate + shift);
        v.setEndTime((double)(i0 + len) / sample_rate + shift);
        p.apply(v);
        for (long i = 0; i < len; ++i)
            out[i0 + i] = v[i];
    }
}
"""),
    # Signal::FilterBank::FilterBuilder::{create, setStart, setEnd, setWeights}, the triangular and the trapeze weight() and
    # FilterBank::isAlmostInteger (where a filter starts, where it ends and what its weights are), together with
    # Signal::FilterBank::Filter -- class declaration, constructor and apply() (declared and defined inside Filterbank.cc).  The builder classes are declared inside Filterbank.cc as Core::Components with configuration
    # constructors (boost): the shell re-declares them without that base -- same members, same virtuals, error() a no-op, the two
    # one-line constants normalizedCenterPosition() / normalizedMiddleBorder() retyped from :231-233, :251-253, :263-265 -- and gives
    # Math::AnalyticFunctionFactory::createConstant its one line (Math/AnalyticFunctionFactory.hh:231-233 forwards to
    # Math::createConstant; that header includes Core/Component.hh).  The analytic functions are the reference's own classes.
    "filter_build": (
        "Signal/Filterbank.cc", [(27, 50), (65, 71), (144, 217), (236, 244), (268, 281), (428, 470), (495, 501), (546, 567), (691, 694)],
        "58ba479fc06126d6362de702f6fe1723977e20fa28fa9ac78093253cdaf64ea3",
        """#include <Core/Types.hh>
#include <Core/Assertions.hh>
#include <Core/ReferenceCounting.hh>
#include <Core/Utility.hh>
#include <Core/XmlStream.hh>
#include <Math/AnalyticFunction.hh>
#include <Math/SimpleAnalyticFunctions.hh>
#include <Math/AcousticalAnalyticFunctions.hh>
#include <algorithm>
#include <cmath>
#include <vector>
namespace Math {
struct AnalyticFunctionFactory {  // Math/AnalyticFunctionFactory.hh:230-237
    static UnaryAnalyticFunctionRef createConstant(UnaryAnalyticFunction::Argument c) { return Math::createConstant(c); }
    static UnaryAnalyticFunctionRef createIdentity() { return UnaryAnalyticFunctionRef(new IdentityFunction); }
};
}  // namespace Math
namespace Signal {
class FilterBank {
public:
    typedef f64  Frequency;
    typedef f32  Data;
    typedef Data FilterWeight;
    enum NormalizationType { normalizeNone, normalizeSurface };
    class Filter;
    class FilterBuilder;
    class Boundary;
    static bool isAlmostInteger(Frequency x);
};
// Signal/Filterbank.cc:366-426 without the Core::Component base
class FilterBank::Boundary {
protected:
    typedef FilterBank::Frequency Frequency;
    Frequency                      filterWidth_;
    Frequency                      spacing_;
    Frequency                      normalizedCenterPosition_;
    Frequency                      minimumFrequency_;
    Frequency                      maximumFrequency_;
    Math::UnaryAnalyticFunctionRef warpingFunction_;
    Math::UnaryAnalyticFunctionRef inverseWarpingFunction_;
    void      setSpacing(Frequency Spacing);
    bool      setWarpingFunction(Math::UnaryAnalyticFunctionRef warpingFunction);
    Frequency postprocessNumberOfFilters(Frequency nFilters) const;
    void      error(const char*, ...) const {}
    Boundary() : filterWidth_(0), spacing_(0), normalizedCenterPosition_(0), minimumFrequency_(0), maximumFrequency_(0) {}
public:
    virtual ~Boundary() {}
    virtual void init(Frequency filterWidth, Frequency spacing, Frequency normalizedCenterPosition);
    virtual bool init(Frequency filterWidth, Frequency spacing, Frequency normalizedCenterPosition, Frequency minimumFrequency,
                      Frequency maximumFrequency, Math::UnaryAnalyticFunctionRef warpingFunction);
    Frequency         filterWidth() const { return filterWidth_; }
    Frequency         spacing() const { return spacing_; }   // (probe only)
    virtual Frequency center(size_t filterIndex) const = 0;
    virtual size_t    getNumberOfFilters() const       = 0;
};
// the three boundary types: class bodies retyped from Signal/Filterbank.cc:482-493, 519-544, 576-590 (their one-line center() /
// getNumberOfFilters() are INSIDE the class bodies, next to configuration constructors); IncludeBoundary::getNumberOfFilters and
// StretchToCover::init are reference text below
class IncludeBoundary : public FilterBank::Boundary {
    typedef FilterBank::Boundary Precursor;
public:
    virtual Frequency center(size_t filterIndex) const { return warpingFunction_->value(spacing_ * (filterIndex + 1)); }
    virtual size_t    getNumberOfFilters() const;
};
class StretchToCover : public FilterBank::Boundary {
    typedef FilterBank::Boundary Precursor;
public:
    virtual bool init(Frequency filterWidth, Frequency spacing, Frequency normalizedCenterPosition, Frequency minimumFrequency,
                      Frequency maximumFrequency, Math::UnaryAnalyticFunctionRef warpingFunction);
    virtual Frequency center(size_t filterIndex) const { return minimumFrequency_ + spacing_ * filterIndex + normalizedCenterPosition_ * filterWidth_; }
    virtual size_t    getNumberOfFilters() const {
        return (size_t)Core::floor(postprocessNumberOfFilters((maximumFrequency_ - minimumFrequency_ - filterWidth_) / spacing_ + 1));
    }
};
class EmphasizeBoundary : public FilterBank::Boundary {
    typedef FilterBank::Boundary Precursor;
public:
    virtual Frequency center(size_t filterIndex) const { return warpingFunction_->value(spacing_ * filterIndex); }
    virtual size_t    getNumberOfFilters() const {
        return (size_t)Core::floor(postprocessNumberOfFilters(inverseWarpingFunction_->value(maximumFrequency_) / spacing_ + 1));
    }
};
// Signal/Filterbank.cc:89-133 without the Core::Component base
class FilterBank::FilterBuilder {
protected:
    size_t                         start_;
    size_t                         end_;
    std::vector<FilterWeight>      weights_;
    Frequency                      center_;
    Frequency                      width_;
    Frequency                      maximumFrequency_;
    Frequency                      minimumFrequency_;
    Math::UnaryAnalyticFunctionRef discreteToContinuousFunction_;
    Math::UnaryAnalyticFunctionRef continuousToDiscreteFunction_;
    Math::UnaryAnalyticFunctionRef derivedWarpingFunction_;
    void error(const char*, ...) const {}
private:
    virtual bool setStart();
    virtual bool setEnd();
    bool         setWeights();
protected:
    virtual FilterWeight weight(Frequency) const = 0;
public:
    FilterBuilder() : start_(0), end_(0), center_(0), width_(0) {}
    virtual ~FilterBuilder() {}
    Core::Ref<FilterBank::Filter> create(Frequency center, Frequency width, Frequency minimumFrequency, Frequency maximumFrequency,
                                         Math::UnaryAnalyticFunctionRef discreteToContinuousFunction,
                                         Math::UnaryAnalyticFunctionRef warpingFunction, bool warpDifferentialUnit);
    virtual Frequency normalizedCenterPosition() const = 0;
};
class SymmetricalTriangularFilterBuilder : public FilterBank::FilterBuilder {
protected:
    virtual FilterBank::FilterWeight weight(FilterBank::Frequency) const;
public:
    virtual FilterBank::Frequency normalizedCenterPosition() const { return 0.5; }
};
class TrapezeFilterBuilder : public FilterBank::FilterBuilder {
private:
    FilterBank::Frequency normalizedMiddleBorder() const { return 0.5 / (1.3 - (-2.5)); }
protected:
    virtual FilterBank::FilterWeight weight(FilterBank::Frequency) const;
public:
    virtual FilterBank::Frequency normalizedCenterPosition() const { return 2.5 / (1.3 - (-2.5)); }
};
}  // namespace Signal
using namespace Signal;
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
// ---- end of reference text ----
namespace {
template<class B>
struct Probe : B {
    int run(double center, double width, double fmin, double fmax, Math::UnaryAnalyticFunctionRef d2c, Math::UnaryAnalyticFunctionRef warp,
            bool diff, int* start, int* end, float* weights, int cap) {
        Core::Ref<Signal::FilterBank::Filter> f = this->create(center, width, fmin, fmax, d2c, warp, diff);
        if (!f)
            return -1;
        *start = (int)this->start_;
        *end   = (int)this->end_;
        if ((int)this->weights_.size() > cap)
            return -2;
        for (size_t i = 0; i < this->weights_.size(); ++i)
            weights[i] = this->weights_[i];
        return (int)this->weights_.size();
    }
};
Math::UnaryAnalyticFunctionRef fb_scaling(double a) { return Math::UnaryAnalyticFunctionRef(new Math::ScalingFunction(a)); }
}  // namespace
// Filter::apply on a filter given by its interval and weights (the pin "filter_apply" of earlier in the round: the Filter class, its
// constructor and apply() are the first two ranges of this file)
extern "C" float ref_filter_apply(const float* in, int n_in, int start, int end, const float* weights) {
    std::vector<Signal::FilterBank::FilterWeight> w(weights, weights + (end - start));
    std::vector<Signal::FilterBank::Data>         v(in, in + n_in);
    Signal::FilterBank::Filter* f = new Signal::FilterBank::Filter((size_t)start, (size_t)end, w);
    const float r = f->apply(v);
    delete f;
    return r;
}
// The boundary of a bank (FilterBank::init, Signal/Filterbank.cc:640-650, with warp-center-positions = true, the default: the boundary
// gets no warping function): type 0 stretch-to-cover / 1 include-boundary / 2 emphasize-boundary.  Returns the number of filters
// (-1: init refused), the width and spacing the boundary ends up with, and up to `cap` centres.
extern "C" int ref_filter_boundary(int type, double width, double spacing, double ncp, double fmin, double fmax, double* width_out,
                                   double* spacing_out, double* centers, int cap) {
    Signal::FilterBank::Boundary* b = type == 0 ? (Signal::FilterBank::Boundary*)new Signal::StretchToCover
                                    : type == 1 ? (Signal::FilterBank::Boundary*)new Signal::IncludeBoundary
                                                : (Signal::FilterBank::Boundary*)new Signal::EmphasizeBoundary;
    int n = -1;
    if (b->init(width, spacing, ncp, fmin, fmax, Math::UnaryAnalyticFunctionRef())) {
        n            = (int)b->getNumberOfFilters();
        *width_out   = b->filterWidth();
        *spacing_out = b->spacing();
        for (int i = 0; i < n && i < cap; ++i)
            centers[i] = b->center((size_t)i);
    }
    delete b;
    return n;
}
// ONE filter: type 0 triangular / 1 trapeze, warping 0 mel / 1 bark (the continuous-domain functions of
// Math/AnalyticFunctionFactory.cc:338-341,369-373), d2c = the scaling of the discrete axis (1 / sample rate of the spectrum)
extern "C" int ref_filter_build(int type, int warping, double center, double width, double fmin, double fmax, double d2c, int diff,
                                int* start, int* end, float* weights, int cap) {
    Math::UnaryAnalyticFunctionRef warp =
            warping == 0 ? Math::nest(fb_scaling(2595.0), Math::UnaryAnalyticFunctionRef(new Math::MelWarpingCore))
                         : Math::nest(fb_scaling(6.0), Math::nest(Math::UnaryAnalyticFunctionRef(new Math::Sinh)->invert(), fb_scaling(1.0 / 600.0)));
    if (type == 0)
        return Probe<Signal::SymmetricalTriangularFilterBuilder>().run(center, width, fmin, fmax, fb_scaling(d2c), warp, diff != 0, start, end, weights, cap);
    return Probe<Signal::TrapezeFilterBuilder>().run(center, width, fmin, fmax, fb_scaling(d2c), warp, diff != 0, start, end, weights, cap);
}
"""),
    # Signal::CosineTransform: constructor, init, initNplusOneData, initEvenAboutNminusHalf and apply -- the
    # class the node wraps.  Its declaration (Signal/CosineTransform.hh:26-78) sits in a header that includes Flow/StringExpressionNode.hh
    # (-> Core/Configuration.hh, boost): re-declared member for member, plus one accessor for the table.  Math::Matrix / Math::Vector and
    # the analytic functions are the reference's own headers.
    "cosine_transform": (
        "Signal/CosineTransform.cc", [(20, 83)],
        "9e0dcc07fa42fe003f2fe42cc32da4109d1f528165a84bfa9908b1a70be8f685",
        """#include <Core/Types.hh>
#include <Core/Assertions.hh>
#include <Math/AnalyticFunction.hh>
#include <Math/SimpleAnalyticFunctions.hh>
#include <Math/Matrix.hh>
#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>
namespace Math {
struct AnalyticFunctionFactory {  // Math/AnalyticFunctionFactory.hh:230-237
    static UnaryAnalyticFunctionRef createConstant(UnaryAnalyticFunction::Argument c) { return Math::createConstant(c); }
    static UnaryAnalyticFunctionRef createIdentity() { return UnaryAnalyticFunctionRef(new IdentityFunction); }
};
}  // namespace Math
namespace Signal {
class CosineTransform {
public:
    typedef f32 Value;
    enum InputType { NplusOneData, evenAboutNminusHalf };
private:
    Math::Matrix<Value> transformation_;
    size_t              N_;
    bool                normalize_;
    void initNplusOneData(size_t inputSize, size_t outputSize, Math::UnaryAnalyticFunctionRef warpingFunction,
                          Math::UnaryAnalyticFunctionRef derivedWarpingFunction);
    void initEvenAboutNminusHalf(size_t inputSize, size_t outputSize, Math::UnaryAnalyticFunctionRef warpingFunction,
                                 Math::UnaryAnalyticFunctionRef derivedWarpingFunction);
public:
    CosineTransform();
    void init(InputType inputType, size_t inputSize, size_t outputSize, bool normalize = false) {
        init(inputType, inputSize, outputSize, normalize, Math::AnalyticFunctionFactory::createIdentity(), true);
    }
    void init(InputType inputType, size_t inputSize, size_t outputSize, bool normalize, Math::UnaryAnalyticFunctionRef,
              bool shouldWarpDifferentialUnit);
    void   apply(const std::vector<Value>& in, std::vector<Value>& out) const;
    size_t inputSize() const { return transformation_.nColumns(); }
    const Math::Matrix<Value>& table() const { return transformation_; }  // (probe only)
};
}  // namespace Signal
using namespace Signal;
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
// ---- end of reference text ----
// n_plus_one: input type N-plus-one (else even-about-N-minus-half); identity warping with the differential unit, as the node's default
extern "C" void ref_cosine_transform(int n_plus_one, int n_in, int n_out, int normalize, const float* in, float* out, float* table_out) {
    Signal::CosineTransform t;
    t.init(n_plus_one ? Signal::CosineTransform::NplusOneData : Signal::CosineTransform::evenAboutNminusHalf, (size_t)n_in, (size_t)n_out,
           normalize != 0);
    std::vector<f32> v(in, in + n_in), o;
    t.apply(v, o);
    for (int k = 0; k < n_out; ++k)
        out[k] = o[k];
    if (table_out)
        for (int k = 0; k < n_out; ++k)
            for (int n = 0; n < n_in; ++n)
                table_out[(size_t)k * n_in + n] = t.table()[k][n];
}
"""),
    # Signal::autoregressionToCepstrum (MF-PLP / PLP): a free function on std::vector<f32>; its header pulls in
    # Flow/Node.hh (boost).  Both the C and the C++ math headers are included, as the real include closure has them (Flow/Node.hh ->
    # ... -> <math.h>): the unqualified log(gain) on an f32 then picks whatever that closure picks -- the pin reports it (ref_ar_log_is_f32)
    "ar_to_cepstrum": (
        "Signal/AutoregressionToCepstrum.cc", [(21, 36)],
        "d4362bf2eafbd167d9a3895840bfbe8fe50b136faf78a8afbd0c83289c58f00e",
        """#include <Core/Types.hh>
#include <Core/Assertions.hh>
#include <Math/LevinsonLse.hh>
#include <cmath>
#include <vector>
namespace Signal {
void autoregressionToCepstrum(f32 gain, const std::vector<f32>& a, std::vector<f32>& c);
}
using namespace Signal;
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
// ---- end of reference text ----
extern "C" void ref_ar_to_cepstrum(float gain, const float* a, int na, float* c, int nc) {
    std::vector<f32> av(a, a + na), cv(nc);
    Signal::autoregressionToCepstrum(gain, av, cv);
    for (int i = 0; i < nc; ++i)
        c[i] = cv[i];
}
"""),
    # Signal::WarpingFunction and Signal::GammaTone (the gammatone front end): everything of both classes that is
    # defined in GammaTone.cc -- the design of the filter bank (centre frequencies on the Greenwood / ERB scale through the two-piece
    # warping, bandwidths, the four coefficients of a channel: f32 / f64 mixed arithmetic on unqualified libm calls) and the cascade
    # apply() with its state.  The class declarations (Signal/GammaTone.hh:20-53,79-212, re-declared member for member with two probe
    # accessors) sit behind Flow/StringExpressionNode.hh (boost).  Driven as GammaToneNode drives it (GammaTone.cc:244-252,283-284).
    "gammatone": (
        "Signal/GammaTone.cc", [(20, 223)],
        "c1894cc0529aa5c2246e2dbf276c1a45e6b8cb0c6e24eae1ceafd4a4b2673daf",
        """#include <Core/Types.hh>
#include <Flow/Data.hh>
#include <Flow/Vector.hh>
#include <Math/Complex.hh>
#include <cmath>
#include <complex>
#include <iostream>
#include <sstream>
#include <vector>
namespace Signal {
class WarpingFunction {
private:
    f32  warpingFactor_;
    f32  freqBreak_, maxFreq_;
    f32  beta_, b_;
    f32  warpedFreqBreak_;
    bool needInit_;
    void init();
public:
    WarpingFunction(f32 warpingFactor, f32 freqBreak, f32 maxFreq);
    void setWarpingFactor(f32 warpingFactor) { warpingFactor_ = warpingFactor; reset(); }
    void setFreqBreak(f32 freqBreak) { freqBreak_ = freqBreak; reset(); }
    void setMaxFreq(f32 maxFreq) { maxFreq_ = maxFreq; reset(); }
    bool checkParam();
    void reset() { needInit_ = true; }
    f32  warping(f32 f);
    f32  inverseWarping(f32 f);
};
class GammaTone {
public:
    struct Coefficient { f32 a0, a1, b1, b2; };
    enum CenterFrequencyModeType { Human, Erb };
private:
    WarpingFunction         warp_;
    f32                     minFreq_;
    f32                     maxFreq_;
    f32                     l_, q_;
    CenterFrequencyModeType centerFrequencyMode_;
    u32                     channels_;
    Flow::Time              sampleRate_;
    u32                     cascade_;
    bool                    needInit_;
    std::vector<std::vector<std::vector<f32>>> buffer_;
    std::vector<f32>                           centerFrequencyList_;
    std::vector<Coefficient>                   coefficients_;
    std::vector<f32>                           bandWidthList_;
    void initializeCenterFrequencyList();
    void initBandWidths();
    void initCoefficients();
    f32  invGreenWoodFunction(f32 cf, const std::vector<f32>& parameters);
public:
    void setCascade(const u32& cascade) { if (cascade_ != cascade) { cascade_ = cascade; reset(); } }
    void setCenterFrequencyMode(CenterFrequencyModeType m) { if (centerFrequencyMode_ != m) { centerFrequencyMode_ = m; reset(); } }
    void setMinFreq(const f32& minFreq) { if (minFreq_ != minFreq) { minFreq_ = minFreq; reset(); } }
    void setMaxFreq(const f32& maxFreq) { if (maxFreq_ != maxFreq) { maxFreq_ = maxFreq; reset(); } }
    void setChannels(const u32& channels) { if (channels_ != channels) { channels_ = channels; reset(); } }
    void setSampleRate(const Flow::Time& sampleRate) { if (sampleRate_ != sampleRate) { sampleRate_ = sampleRate; reset(); } }
    void setL(const f32& l) { if (l_ != l) { l_ = l; reset(); } }
    void setQ(const f32& q) { if (q_ != q) { q_ = q; reset(); } }
    void setWarpWarpingFactor(const f32& warpingFactor) { warp_.setWarpingFactor(warpingFactor); reset(); }
    void setWarpFreqBreak(const f32& freqBreak) { warp_.setFreqBreak(freqBreak); reset(); }
    void setWarpMaxFreq(const f32& maxFreq) { warp_.setMaxFreq(maxFreq); reset(); }
    GammaTone();
    virtual ~GammaTone() {}
    virtual void init();
    void reset() { needInit_ = true; }
    bool checkParam() { return warp_.checkParam(); }
    void apply(const Flow::Vector<f32>& in, Flow::Vector<Flow::Vector<f32>>& out);
    const std::vector<f32>&         centerFrequencies() const { return centerFrequencyList_; }  // (probe only)
    const std::vector<Coefficient>& coefficients() const { return coefficients_; }               // (probe only)
};
}  // namespace Signal
using namespace Signal;
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
// ---- end of reference text ----
// pcm [n] in blocks of `block` samples; cf [channels], coef [channels x 4] (a0 a1 b1 b2), filtered [n x channels].  Returns -1 where
// GammaToneNode::init would report an error (checkParam).
extern "C" int ref_gammatone(double sample_rate, int cascade, double minfreq, double maxfreq, double q, int channels, int cfmode,
                             double warp_freqbreak, const char* warping_factor, const float* pcm, long n, int block, float* cf, float* coef,
                             float* filtered) {
    Signal::GammaTone g;
    g.setCenterFrequencyMode(Signal::GammaTone::CenterFrequencyModeType(cfmode));
    g.setMinFreq(minfreq);
    g.setMaxFreq(maxfreq);
    g.setQ(q);
    g.setChannels(channels);
    g.setCascade(cascade);
    g.setWarpFreqBreak(warp_freqbreak);
    g.setSampleRate(sample_rate);
    g.setWarpMaxFreq(sample_rate / 2);
    {
        f32                warpingValue;
        std::istringstream iss(warping_factor);
        iss >> warpingValue;
        g.setWarpWarpingFactor(warpingValue);
    }
    if (!g.checkParam())
        return -1;
    for (long i0 = 0; i0 < n || i0 == 0; i0 += block) {
        const long len = n - i0 < block ? n - i0 : block;
        Flow::Vector<f32>               in(pcm + i0, pcm + i0 + (len > 0 ? len : 0));
        Flow::Vector<Flow::Vector<f32>> out;
        g.apply(in, out);
        for (long i = 0; i < len; ++i)
            for (int ch = 0; ch < channels; ++ch)
                filtered[(size_t)(i0 + i) * channels + ch] = out[i][ch];
        if (n == 0)
            break;
    }
    for (int ch = 0; ch < channels; ++ch) {
        cf[ch]           = g.centerFrequencies()[ch];
        coef[4 * ch]     = g.coefficients()[ch].a0;
        coef[4 * ch + 1] = g.coefficients()[ch].a1;
        coef[4 * ch + 2] = g.coefficients()[ch].b1;
        coef[4 * ch + 3] = g.coefficients()[ch].b2;
    }
    return 0;
}
"""),
    # Signal::Normalization and five of its six algorithms (level, mean, mean-and-variance, mean-and-variance-1D,
    # divide-by-mean; mean-norm needs Flow::NormFunction built from a configuration) on the reference's own sliding window: the template
    # Signal::SlidingWindow is taken whole from its header (which includes Flow/Node.hh, boost, for nothing the template uses), then the
    # class declarations stand between it and the definitions of Normalization.cc.  Pins the arithmetic AND the window's timing: which
    # frame leaves when, with which statistics, and what flush() hands out.
    "normalization": (
        "Signal/Normalization.cc", [("Signal/SlidingWindow.hh", 22, 471), DECLS_NORMALIZATION, (24, 97), (100, 109), (112, 141), (148, 191),
                                    (198, 247), (254, 262)],
        "151c8493588221bf3337b086dea1505de340286f8f1acd204d6d92acfa0c6be8",
        """#include <Core/Types.hh>
#include <Core/Assertions.hh>
#include <Core/Utility.hh>
#include <Flow/Data.hh>
#include <Flow/DataAdaptor.hh>
#include <Flow/Vector.hh>
#include <algorithm>
#include <cmath>
#include <deque>
#include <functional>
#include <vector>
// ---- reference text, %(file)s:%(ranges)s (the first piece: Signal/SlidingWindow.hh) ----
""",
        """
// ---- end of reference text ----
// type 0 level (index `level`), 1 mean, 2 mean-and-variance, 3 mean-and-variance-1D, 4 divide-by-mean; length / right as the node passes
// them (both >= 2^31 - 1: the whole segment).  Frames are fed as NormalizationNode::work feeds them: update() per frame, flush() until it
// fails.  out [n x dim] in emission order; returns the number of frames emitted (-1: init refused).
extern "C" long ref_normalization(int type, int level, unsigned long length, unsigned long right, const float* in, long n, int dim, float* out) {
    Signal::Normalization* a = type == 0 ? (Signal::Normalization*)new Signal::LevelNormalization((size_t)level)
                             : type == 1 ? (Signal::Normalization*)new Signal::MeanNormalization
                             : type == 2 ? (Signal::Normalization*)new Signal::MeanAndVarianceNormalization
                             : type == 3 ? (Signal::Normalization*)new Signal::MeanAndVarianceNormalization1D
                                         : (Signal::Normalization*)new Signal::DivideByMean;
    if (!a->init((size_t)length, (size_t)right, (size_t)dim)) {
        delete a;
        return -1;
    }
    long emitted = 0;
    auto emit = [&](Signal::Normalization::Frame& f) {
        for (int d = 0; d < dim; ++d)
            out[(size_t)emitted * dim + d] = (*f)[d];
        ++emitted;
    };
    Signal::Normalization::Frame o;
    for (long t = 0; t < n; ++t) {
        Signal::Normalization::Frame f(new Flow::Vector<f32>(in + (size_t)t * dim, in + (size_t)(t + 1) * dim));
        if (a->update(f, o))
            emit(o);
    }
    while (emitted < n && a->flush(o))
        emit(o);
    delete a;
    return emitted;
}
"""),
    # Mm::DensityClustering<f32, f32> ("preselection-batch-float"): initializeClusters (srand(1) / rand()),
    # assignDensities, updateClusterMeans, selectClusters (std::sort on the distances) and DensityClusteringBase::init (the cluster count is
    # reduced to the number of densities).  build() itself reads its iteration count from the configuration and talks to a cache archive:
    # its loop -- initialise, then `iterations` x (assign, update) -- is the three lines of the entry point below.  The class declarations
    # (Mm/DensityClustering.hh:29-150, a Core::Component with archive IO) are re-declared without that base and without the IO members.
    "density_clustering": (
        "Mm/DensityClustering.tcc", [(61, 119), (157, 180), ("Mm/DensityClustering.cc", 45, 57)],
        "8101564b64b4c8cebea849e2236ce68b5d81361b82a07e7ca3ea52bc50e7b63d",
        """#include <Core/Types.hh>
#include <Core/Assertions.hh>
#include <Core/Extensions.hh>
#include <Core/Utility.hh>
#include <Mm/Utilities.hh>
#include <algorithm>
#include <functional>
#include <iostream>
#include <set>
#include <vector>
namespace Mm {
class DensityClusteringBase {
public:
    typedef u8                                        ClusterIndex;
    typedef std::vector<ClusterIndex>::const_iterator ClusterIndexIterator;
    DensityClusteringBase(u32 nClusters, u32 nSelected) : nClusters_(nClusters), nSelected_(nSelected), dimension_(0), nDensities_(0), backoffScore_(0) {}
    virtual ~DensityClusteringBase() {}
    void         init(u32 dimension, u32 nDensities);
    ClusterIndex clusterIndexForDensity(size_t density) const { return clusterIndexForDensity_[density]; }
    u32          nClusters() const { return nClusters_; }
protected:
    std::ostream&             log() const { static std::ostream null(nullptr); return null; }
    std::vector<ClusterIndex> clusterIndexForDensity_;
    u32                       nClusters_, nSelected_;
    u32                       dimension_, nDensities_;
    const float               backoffScore_;
};
template<class F, class D>
class DensityClustering : public DensityClusteringBase {
public:
    typedef F FeatureType;
    typedef D DistanceType;
    DensityClustering(u32 nClusters, u32 nSelected) : DensityClusteringBase(nClusters, nSelected), clusterMeans_(0) {}
    ~DensityClustering() { delete[] clusterMeans_; }
    void selectClusters(bool* selection, FeatureType* feature) const;
    // (probe: build() without the configuration and the cache archive)
    void buildLoop(const FeatureType* densities, u32 iterations) {
        clusterMeans_ = new FeatureType[nClusters_ * dimension_];
        initializeClusters(densities);
        for (u32 i = 0; i < iterations; ++i) {
            DensityAssignment densitiesAssignedToClusters(nClusters_);
            assignDensities(densities, densitiesAssignedToClusters);
            updateClusterMeans(densities, densitiesAssignedToClusters);
        }
    }
    const FeatureType* means() const { return clusterMeans_; }
private:
    FeatureType*       meanForCluster(ClusterIndex cluster) { return clusterMeans_ + cluster * dimension_; }
    const FeatureType* meanForCluster(ClusterIndex cluster) const { return clusterMeans_ + cluster * dimension_; }
    const FeatureType* meanForDensity(const FeatureType* means, u32 density) const { return means + density * dimension_; }
    typedef std::vector<std::vector<u32>> DensityAssignment;
    void initializeClusters(const FeatureType* densities);
    void assignDensities(const FeatureType* densities, DensityAssignment& densityAssignment);
    f64  updateClusterMeans(const FeatureType* densities, DensityAssignment& densityAssignment);
    FeatureType* clusterMeans_;
};
}  // namespace Mm
using namespace Mm;
namespace Mm {
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
}  // namespace Mm
// ---- end of reference text ----
// means [n_dens x dim] (already multiplied by 1 / sigma and padded, as BatchFloatFeatureScorer::init leaves them), features [T x dim] likewise.
// cluster_of [n_dens], cluster_means [n_clusters_out x dim], selection [T x n_clusters_out] (1 = selected)
extern "C" int ref_density_clustering(const float* means, int n_dens, int dim, int n_clusters, int n_select, int iterations,
                                      unsigned char* cluster_of, float* cluster_means, const float* feats, int T, unsigned char* selection) {
    Mm::DensityClustering<f32, f32> c((u32)n_clusters, (u32)n_select);
    c.init((u32)dim, (u32)n_dens);
    c.buildLoop(means, (u32)iterations);
    const int nc = (int)c.nClusters();
    for (int k = 0; k < n_dens; ++k)
        cluster_of[k] = c.clusterIndexForDensity((size_t)k);
    for (int i = 0; i < nc * dim; ++i)
        cluster_means[i] = c.means()[i];
    std::vector<char> sel((size_t)nc);
    std::vector<f32>  f((size_t)dim);
    for (int t = 0; t < T; ++t) {
        for (int i = 0; i < dim; ++i)
            f[i] = feats[(size_t)t * dim + i];
        c.selectClusters((bool*)sel.data(), f.data());
        for (int k = 0; k < nc; ++k)
            selection[(size_t)t * nc + k] = sel[k] ? 1 : 0;
    }
    return nc;
}
// the same for Mm::DensityClustering<u8, s32> (preselection-batch-int): means [n_dens x dim] u8 per mixture entry
extern "C" int ref_density_clustering_u8(const unsigned char* means, int n_dens, int dim, int n_clusters, int iterations,
                                         unsigned char* cluster_of, unsigned char* cluster_means) {
    Mm::DensityClustering<u8, s32> c((u32)n_clusters, 1);
    c.init((u32)dim, (u32)n_dens);
    c.buildLoop(means, (u32)iterations);
    const int nc = (int)c.nClusters();
    for (int k = 0; k < n_dens; ++k)
        cluster_of[k] = c.clusterIndexForDensity((size_t)k);
    for (int i = 0; i < nc * dim; ++i)
        cluster_means[i] = c.means()[i];
    return nc;
}
"""),
    # The generic-vector-f32-<function> functors of Flow/SimpleFunction.hh (the MFCC's log10; the PLP chain's
    # power; and the rest of the family: log-plus, ln, exp, sqrt, cos, add, multiply, quantize, abs, minimum, maximum): header-only
    # templates, but the header includes Flow/Node.hh (boost) for the node template behind them.  The templates are taken whole
    # (:32-358); which libm overload an unqualified log10 / pow / rint on an f32 picks depends on the math headers in scope -- <cmath>
    # (the header's own include) here, as there.
    "vector_functions": (
        "Flow/SimpleFunction.hh", [(32, 358)],
        "668385596c1ca9dc1f1fa38e9de7d55ae685431d8adf8244db629a2fe1475ccb",
        """#include <Core/Types.hh>
#include <Core/Utility.hh>
#include <Flow/DataAdaptor.hh>
#include <Flow/Vector.hh>
#include <algorithm>
#include <cmath>
#include <numeric>
#include <string>
namespace Flow {
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
}  // namespace Flow
// ---- end of reference text ----
// kind: 0 log (log10), 1 log-plus, 2 ln, 3 exp, 4 power, 5 sqrt, 6 cos, 7 addition, 8 multiplication, 9 quantize, 10 abs, 11 minimum,
// 12 maximum; in / out [n]
extern "C" void ref_vector_function(int kind, float prm, const float* in, long n, float* out) {
    Flow::Vector<f32> v(in, in + n);
    switch (kind) {
        case 0: Flow::VectorLogFunction<f32>().apply(v, prm); break;
        case 1: Flow::VectorLogPlusFunction<f32>().apply(v, prm); break;
        case 2: Flow::VectorLnFunction<f32>().apply(v, prm); break;
        case 3: Flow::VectorExpFunction<f32>().apply(v, prm); break;
        case 4: Flow::VectorPowerFunction<f32>().apply(v, prm); break;
        case 5: Flow::VectorSqrtFunction<f32>().apply(v, prm); break;
        case 6: Flow::VectorCosFunction<f32>().apply(v, prm); break;
        case 7: Flow::VectorScalarAdditionFunction<f32>().apply(v, prm); break;
        case 8: Flow::VectorScalarMultiplicationFunction<f32>().apply(v, prm); break;
        case 9: Flow::VectorQuantizationFunction<f32>().apply(v, prm); break;
        case 10: Flow::VectorAbsoluteValueFunction<f32>().apply(v, prm); break;
        case 11: Flow::VectorMinimumFunction<f32>().apply(v, prm); break;
        default: Flow::VectorMaximumFunction<f32>().apply(v, prm); break;
    }
    for (long i = 0; i < n; ++i)
        out[i] = v[i];
}
"""),
    # signal-vector-f32-<kind>-normalization: the six functors of Signal/VectorNormalization.hh, header-only
    # templates taken whole (:35-171 without line 98, `hope(!v.empty())`: the assertion's handler lives in Core/Assertions.cc, which needs
    # boost, and no stand-in is written for it); the header includes Flow/Node.hh (boost) for the node template behind them.
    "vector_normalization": (
        "Signal/VectorNormalization.hh", [(35, 97), (99, 171)],
        "14551f62fd97ac7f705f759b503a4246e4d81c152c2b07bacebaeb138f4be731",
        """#include <Core/Types.hh>
#include <Core/Assertions.hh>
#include <Core/Utility.hh>
#include <Flow/Data.hh>
#include <Flow/Vector.hh>
#include <algorithm>
#include <cmath>
#include <functional>
#include <numeric>
#include <string>
#include <vector>
namespace Signal {
// ---- reference text, %(file)s:%(ranges)s ----
""",
        """
}  // namespace Signal
// ---- end of reference text ----
// type 0 amplitude-spectrum-energy, 1 energy, 2 maximum, 3 mean-energy, 4 mean, 5 variance; one vector of dim values
extern "C" void ref_vector_normalize(int type, const float* in, int dim, float* out) {
    std::vector<f32> v(in, in + dim);
    switch (type) {
        case 0: Signal::AmplitudeSpectrumEnergyVectorNormalization<f32>()(v); break;
        case 1: Signal::EnergyVectorNormalization<f32>()(v); break;
        case 2: Signal::MaximumVectorNormalization<f32>()(v); break;
        case 3: Signal::MeanEnergyVectorNormalization<f32>()(v); break;
        case 4: Signal::MeanVectorNormalization<f32>()(v); break;
        default: Signal::VarianceVectorNormalization<f32>()(v); break;
    }
    for (int i = 0; i < dim; ++i)
        out[i] = v[i];
}
"""),
}


def main():
    name, out = sys.argv[1], sys.argv[2]
    file, ranges, sha, head, tail = SPECS[name]
    cache = {}

    def src(name):
        if name not in cache:
            with open("%s/%s" % (REF, name), "r", encoding="utf-8", errors="replace") as f:
                cache[name] = f.readlines()
        return cache[name]
    # a range is (first, last) in `file`, or (other file, first, last); a string is shell text that has to stand BETWEEN two pieces of
    # reference text (declarations that need the text in front of them); it is written out in place and is not part of the hash
    items  = [r if isinstance(r, str) else ((file,) + tuple(r) if len(r) == 2 else tuple(r)) for r in ranges]
    ranges = [r for r in items if not isinstance(r, str)]
    text = "".join("".join(src(fn)[first - 1:last]) for fn, first, last in ranges)
    body = "".join(r if isinstance(r, str) else "".join(src(r[0])[r[1] - 1:r[2]]) for r in items)
    got = hashlib.sha256(text.encode()).hexdigest()
    if len(sys.argv) > 3 and sys.argv[3] == "--print-sha":
        print(got)
        return
    if got != sha:
        sys.exit("extract_fn: %s %s hashes to %s, expected %s -- the reference moved; re-check the line ranges" % (name, ranges, got, sha))
    with open(out, "w") as f:
        f.write("// GENERATED by oracle/ref/extract_fn.py -- do not commit (oracle/_ref/ is git-ignored)\n")
        f.write(head % {"file": file, "ranges": ", ".join("%s:%d-%d" % r if r[0] != file else "%d-%d" % r[1:] for r in ranges)})
        f.write(body)
        f.write(tail)


if __name__ == "__main__":
    main()
