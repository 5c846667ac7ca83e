// oracle/ref/ref_harness.cc -- thin C entry points onto UNMODIFIED reference translation units.
//
// TEST INFRASTRUCTURE.  Built only where /root/reference exists (see Makefile); the resulting
// oracle/_ref/libref.so is used by tests/ to pin oracle/liboracle.so bit-exactly and, via
// tests/golden/make_golden.py, to produce the committed golden vectors.
//
// Everything called here is reference code compiled from where it lies:
//   Math::FastFourierTransform            src/Math/FastFourierTransform.cc
//   Signal::WindowBuffer (put/get/flush)  src/Signal/WindowBuffer.cc (+ Flow/Core closure, see Makefile)
//   Signal::TimeWindowBuffer<Flow::Vector<f32>> (put/get/flush)  src/Signal/TimeWindowBuffer.cc -- the framing under
//       signal-temporalintegration (Signal/TemporalIntegration.hh:33)
//   Math::{ScalingFunction,MelWarpingCore,AnalyticNesting}  header-only, composed exactly like
//       Math::AnalyticFunctionFactory::createMelWarpingFunction (AnalyticFunctionFactory.cc:338-341)
//   Math::{Sinh,ArcSinh,DerivedArcSinh} (SimpleAnalyticFunctions.hh:152-222) composed like createBarkWarpingFunction
//       (AnalyticFunctionFactory.cc:369-373); Math::EqualLoudnessPreemphasis[4Khz]  src/Math/AcousticalAnalyticFunctions.cc
//   Math::mt_vr_exp<f32> (src/Math/FastVectorOperations.hh:57-63) -- what Math::FastMatrix<f32>::exp() runs inside sigmoid()
//       (Math/FastMatrix.hh:785-787,802-808): the unqualified exp() on a float there is ::exp(double), narrowed
//   Mm::gaussLogNormFactor, Mm::inverseSquareRoot           src/Mm/Utilities.hh:53-91
//   Math::Matrix<f32> * Math::Vector<f32>   src/Math/Matrix.hh:485-494, src/Math/Vector.hh:94-101 -- what
//       Signal::CosineTransform::apply runs (f32 products accumulated left to right)
//   Math::transformAlternatingComplex with Math::pointerAbs<f32>   src/Math/Complex.hh:39-46,104-113 (+ Core::abs,
//       Core/Utility.hh:126-129) -- the whole body of Signal::alternatingComplexVectorAmplitude<f32>::operator()
//       (Signal/ComplexVectorFunction.hh:30-47; that header itself pulls in Signal/Node.hh -> Core/Configuration.hh -> boost)
//   Flow::Vector<f32>::{read,write}, Flow::Datatype::{read,write}GatheredData, Core::Binary{In,Out}putStream,
//       Core::XmlWriter          src/Flow/Vector.hh:88-106, src/Flow/Datatype.cc:28-52 (feature-cache payload)
// No reference header, library or tool is replaced by a stand-in; translation units that need
// boost / bison / cblas (anything including Core/Configuration.hh) are simply not built.
#include <Math/AcousticalAnalyticFunctions.hh>
#include <Math/FastFourierTransform.hh>
#include <Math/FastVectorOperations.hh>
#include <Math/LevinsonLse.hh>
#include <Math/SimpleAnalyticFunctions.hh>
#include <Mm/Utilities.hh>
#include <Math/Matrix.hh>
#include <Math/Vector.hh>
#include <Math/Complex.hh>
#include <Signal/TimeWindowBuffer.hh>
#include <Signal/WindowBuffer.hh>
#include <Core/BinaryStream.hh>
#include <Core/Utility.hh>
#include <Core/XmlStream.hh>
#include <Flow/Vector.hh>
#include <functional>
#include <sstream>

#include <algorithm>
#include <cstring>
#include <vector>

namespace {
Math::UnaryAnalyticFunctionRef melWarp() {
    // continuousDomain branch of createMelWarpingFunction
    return Math::nest(Math::UnaryAnalyticFunctionRef(new Math::ScalingFunction(2595.0)),
                      Math::UnaryAnalyticFunctionRef(new Math::MelWarpingCore));
}
Math::UnaryAnalyticFunctionRef scaling(double a) {
    return Math::UnaryAnalyticFunctionRef(new Math::ScalingFunction(a));
}
Math::UnaryAnalyticFunctionRef barkWarp() {
    // continuousDomain branch of createBarkWarpingFunction (AnalyticFunctionFactory.cc:369-373); createSinh() = new Sinh
    return Math::nest(scaling(6.0), Math::nest(Math::UnaryAnalyticFunctionRef(new Math::Sinh)->invert(), scaling(1.0 / 600.0)));
}
}  // namespace

extern "C" {

// Core::isAlmostEqual / isSignificantlyGreater (Core/Utility.hh:322-345): they decide the FFT length, the filter-bank edge rounding
// and which equal-loudness curve plp.flow gets
int ref_is_almost_equal(double a, double b, double tolerance) {
    return Core::isAlmostEqual(a, b, tolerance) ? 1 : 0;
}
int ref_is_significantly_greater(double a, double b, double tolerance) {
    return Core::isSignificantlyGreater(a, b, tolerance) ? 1 : 0;
}

// bark warping, its derivative and inverse, alone and nested with disc-to-cont as FilterBuilder::create composes them
double ref_bark(double f) {
    return barkWarp()->value(f);
}
double ref_bark_derivative(double f) {
    return barkWarp()->derive()->value(f);
}
double ref_bark_inverse(double b) {
    return barkWarp()->invert()->value(b);
}
double ref_bark_bin(double bin, double inputSampleRate) {
    return Math::nest(barkWarp(), scaling(1 / inputSampleRate))->value(bin);
}
double ref_bark_bin_inverse(double warped, double inputSampleRate) {
    return Math::nest(barkWarp(), scaling(1 / inputSampleRate))->invert()->value(warped);
}
double ref_bark_bin_derivative(double bin, double inputSampleRate) {
    return Math::nest(barkWarp()->derive(), scaling(1 / inputSampleRate))->value(bin);
}
// Math::EqualLoudnessPreemphasis / EqualLoudnessPreemphasis4Khz (Math/AcousticalAnalyticFunctions.cc:21-37, compiled unmodified)
double ref_equal_loudness(double f, int fourKhz) {
    Math::UnaryAnalyticFunctionRef e = fourKhz ? Math::UnaryAnalyticFunctionRef(new Math::EqualLoudnessPreemphasis4Khz)
                                               : Math::UnaryAnalyticFunctionRef(new Math::EqualLoudnessPreemphasis);
    return e->value(f);
}
// plp.flow's f = "nest(nest(disc-to-cont, invert(bark)), equal-loudness-preemphasis)": the factory's "nest(g, f)" builds
// Math::nest(f, g) (AnalyticFunctionFactory.cc:161-180), disc-to-cont = scaling(1 / sampleRate) (:322-327)
double ref_plp_equal_loudness(double index, double sampleRate, int fourKhz) {
    Math::UnaryAnalyticFunctionRef g = Math::nest(barkWarp()->invert(), scaling(1 / sampleRate));
    Math::UnaryAnalyticFunctionRef e = fourKhz ? Math::UnaryAnalyticFunctionRef(new Math::EqualLoudnessPreemphasis4Khz)
                                               : Math::UnaryAnalyticFunctionRef(new Math::EqualLoudnessPreemphasis);
    return Math::nest(e, g)->value(index);
}

void ref_fft_real(float* v, int n) {
    std::vector<float> d(v, v + n);
    Math::FastFourierTransform fft;
    fft.transformReal(d, false);
    std::memcpy(v, d.data(), sizeof(float) * n);
}

void ref_fft_complex(float* v, int n_floats) {
    std::vector<float> d(v, v + n_floats);
    Math::FastFourierTransform fft;
    fft.transform(d, false);
    std::memcpy(v, d.data(), sizeof(float) * n_floats);
}

double ref_mel(double f) {
    return melWarp()->value(f);
}
double ref_mel_derivative(double f) {
    return melWarp()->derive()->value(f);
}
double ref_mel_inverse(double m) {
    return melWarp()->invert()->value(m);
}
// nest(warp, disc-to-cont) and its inverse / derivative, as FilterBuilder::create composes them
double ref_warped_bin(double bin, double inputSampleRate) {
    Math::UnaryAnalyticFunctionRef d2c(new Math::ScalingFunction(1 / inputSampleRate));
    return Math::nest(melWarp(), d2c)->value(bin);
}
double ref_warped_bin_inverse(double warped, double inputSampleRate) {
    Math::UnaryAnalyticFunctionRef d2c(new Math::ScalingFunction(1 / inputSampleRate));
    return Math::nest(melWarp(), d2c)->invert()->value(warped);
}
double ref_warped_bin_derivative(double bin, double inputSampleRate) {
    Math::UnaryAnalyticFunctionRef d2c(new Math::ScalingFunction(1 / inputSampleRate));
    return Math::nest(melWarp()->derive(), d2c)->value(bin);
}

// out = M v with the reference's own matrix and vector classes (row-major M [rows x cols])
void ref_matrix_vector(const float* M, int rows, int cols, const float* v, float* out) {
    Math::Matrix<f32> m(rows, cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            m[r][c] = M[(size_t)r * cols + c];
    Math::Vector<f32> x(cols);
    for (int c = 0; c < cols; ++c)
        x[c] = v[c];
    const Math::Vector<f32> y = m * x;
    for (int r = 0; r < rows; ++r)
        out[r] = y[r];
}

// |re + i im| of an alternating complex vector (the signal-vector-alternating-complex-f32-amplitude node's functor)
int ref_complex_amplitude(const float* x, int n_floats, float* out) {
    std::vector<f32> in(x, x + n_floats), res(n_floats / 2);
    Math::transformAlternatingComplex(in.begin(), in.end(), res.begin(), Math::pointerAbs<f32>());
    for (size_t i = 0; i < res.size(); ++i)
        out[i] = res[i];
    return (int)res.size();
}

double ref_gauss_log_norm_factor(const float* var, int n) {
    std::vector<float> v(var, var + n);
    return Mm::gaussLogNormFactor(v.begin(), v.end());
}
float ref_inverse_square_root(float x) {
    return Mm::inverseSquareRoot<float>()(x);
}

// Drives Signal::WindowBuffer the way SlidingAlgorithmNode::work does
// (src/Signal/SlidingAlgorithmNode.hh:60-79): get() until it fails, then put the next input
// block; at end of stream flush() until flushed().  No window function is applied (the
// default WindowBuffer::transform is the identity).  Returns the number of frames;
// frame_len[i] / frame_start_time[i] describe frame i, frames (if not null) receives the
// samples of every frame padded to `length` floats.
long ref_window_frames(const float* pcm, long n, long block, unsigned length, unsigned shift,
                       double sampleRate, long max_frames, int* frame_len,
                       double* frame_start_time, float* frames) {
    Signal::WindowBuffer wb;
    wb.setLength(length);
    wb.setShift(shift);
    wb.setSampleRate(sampleRate);
    long nf  = 0;
    long pos = 0;
    Flow::Vector<float> out;
    auto emit = [&]() {
        if (nf < max_frames) {
            if (frame_len)
                frame_len[nf] = (int)out.size();
            if (frame_start_time)
                frame_start_time[nf] = out.startTime();
            if (frames) {
                std::fill(frames + nf * length, frames + (nf + 1) * length, 0.0f);
                std::copy(out.begin(), out.end(), frames + nf * length);
            }
        }
        ++nf;
    };
    bool eos = false;
    while (true) {
        if (wb.get(out)) {
            emit();
            continue;
        }
        if (!eos) {
            if (pos < n) {
                long                cnt = std::min(block, n - pos);
                Flow::Vector<float> in(pcm + pos, pcm + pos + cnt);
                in.setStartTime(pos / sampleRate);
                in.setEndTime((pos + cnt) / sampleRate);
                wb.put(in);
                pos += cnt;
                continue;
            }
            eos = true;
        }
        // end of stream: flush
        if (wb.flushed() || !wb.flush(out))
            break;
        emit();
    }
    return nf;
}

// The same driver loop over Signal::TimeWindowBuffer<Flow::Vector<f32>> -- the base class of Signal::TemporalIntegration
// (Signal/TemporalIntegration.hh:33; its init() sets length = rint(length-in-s * rate), shift likewise) -- with the identity
// transform.  Sample i of the input carries the value i in every channel, so first_sample[f] is the start index of frame f.
long ref_time_window_frames(long n, long block, int channels, unsigned length, unsigned shift, int flush_all, double sampleRate,
                            long max_frames, int* frame_len, double* frame_start_time, long* first_sample) {
    typedef Flow::Vector<f32> Sample;
    Signal::TimeWindowBuffer<Sample> wb;
    wb.setLength(length);
    wb.setShift(shift);
    wb.setSampleRate(sampleRate);
    wb.setFlushAll(flush_all != 0);
    long                 nf = 0, pos = 0;
    Flow::Vector<Sample> out;
    auto                 emit = [&]() {
        if (nf < max_frames) {
            if (frame_len)
                frame_len[nf] = (int)out.size();
            if (frame_start_time)
                frame_start_time[nf] = out.startTime();
            if (first_sample)
                first_sample[nf] = out.empty() ? -1 : (long)out[0][0];
        }
        ++nf;
    };
    bool eos = false;
    while (true) {
        if (wb.get(out)) {
            emit();
            continue;
        }
        if (!eos) {
            if (pos < n) {
                long                 cnt = std::min(block, n - pos);
                Flow::Vector<Sample> in;
                for (long i = 0; i < cnt; ++i)
                    in.push_back(Sample(channels, (f32)(pos + i)));
                in.setStartTime(pos / sampleRate);
                in.setEndTime((pos + cnt) / sampleRate);
                wb.put(in);
                pos += cnt;
                continue;
            }
            eos = true;
        }
        if (wb.flushed() || !wb.flush(out))
            break;
        emit();
    }
    return nf;
}

// y = exp(x) elementwise exactly as Math::FastMatrix<f32>::exp() computes it (mt_vr_exp, one thread)
void ref_mt_vr_exp(int n, const float* x, float* y) {
    std::vector<float> in(x, x + n);
    Math::mt_vr_exp(n, in.data(), y, 1);
}

// One Flow cache block as Flow::CacheWriter emits it (src/Flow/Cache.cc:88-93: datatype name, then
// Datatype::writeGatheredData) for n vector-f32 packets of `dim` floats; times = [n][2] start/end.
long ref_cache_block_write(const float* feats, const double* times, int n, int dim, unsigned char* out, long cap) {
    std::vector<Flow::DataPtr<Flow::Data>> data;
    for (int i = 0; i < n; ++i) {
        Flow::Vector<f32>* v = new Flow::Vector<f32>(feats + (size_t)i * dim, feats + (size_t)(i + 1) * dim);
        v->setStartTime(times[2 * i]);
        v->setEndTime(times[2 * i + 1]);
        data.push_back(Flow::DataPtr<Flow::Data>(v));
    }
    std::ostringstream       os;
    Core::BinaryOutputStream b(os);
    const Flow::Datatype*    dt = Flow::Vector<f32>::type();
    b << dt->name();
    if (!dt->writeGatheredData(b, data))
        return -1;
    std::string s = os.str();
    if ((long)s.size() > cap)
        return -(long)s.size();
    std::memcpy(out, s.data(), s.size());
    return (long)s.size();
}

// Parse one block back with the reference reader (src/Flow/Cache.cc:47-58 minus the registry lookup).
// Returns n packets (all must have `dim` floats) or <0; *consumed = bytes read.
long ref_cache_block_read(const unsigned char* in, long len, int dim, float* feats, double* times, long cap_frames,
                          long* consumed, char* type_name, int type_cap) {
    std::istringstream      is(std::string((const char*)in, (size_t)len));
    Core::BinaryInputStream b(is);
    std::string             name;
    if (!(b >> name))
        return -1;
    std::snprintf(type_name, type_cap, "%s", name.c_str());
    std::vector<Flow::DataPtr<Flow::Data>> data;
    if (!Flow::Vector<f32>::type()->readGatheredData(b, data))
        return -2;
    if ((long)data.size() > cap_frames)
        return -3;
    for (size_t i = 0; i < data.size(); ++i) {
        const Flow::Vector<f32>* v = static_cast<const Flow::Vector<f32>*>(data[i].get());
        if ((int)v->size() != dim)
            return -4;
        std::memcpy(feats + i * dim, v->data(), sizeof(float) * dim);
        times[2 * i]     = v->startTime();
        times[2 * i + 1] = v->endTime();
    }
    *consumed = (long)is.tellg();
    return (long)data.size();
}

// The ".attribs" side file: the statements of Flow::Attributes' XmlWriter operator (src/Flow/Attributes.hh:67-70,
// 132-138) issued on the reference's Core::XmlWriter (Attributes.hh itself pulls in Core/Configuration.hh -> boost).
long ref_attribs_xml(const char** names, const char** values, int n, char* out, long cap) {
    std::ostringstream os;
    {
        Core::XmlWriter xw(os);
        xw << Core::XmlOpen("flow-attributes");
        for (int i = 0; i < n; ++i)
            xw << Core::XmlEmpty("flow-attribute") + Core::XmlAttribute("name", std::string(names[i])) +
                            Core::XmlAttribute("value", std::string(values[i]));
        xw << Core::XmlClose("flow-attributes");
    }
    std::string s = os.str();
    if ((long)s.size() + 1 > cap)
        return -(long)s.size();
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (long)s.size();
}

// Training statistics and re-estimation arithmetic, as far as it lives in the header-only Mm/Utilities.hh
// (Mm/VectorAccumulator.hh and the estimator classes pull in Core/Configuration.hh -> boost and cannot be built):
// the element-wise update VectorAccumulator::accumulate(v, weight) issues (Mm/VectorAccumulator.hh:60-67) through the
// reference's own unrolledTransform and functors.  kind 0: std::plus (mean, unweighted) 1: plusWeighted 2: plusSquare
// 3: plusSquareWeighted (input f32); kind 4: plusNormalizedSquare (input f64, CovarianceEstimator::estimate's
// WeighedMeanSquareSum, Mm/GaussDensityEstimator.cc:203-214).
void ref_accumulate_vector(double* sum, const void* in, int n, double weight, int kind) {
    std::vector<double> s(sum, sum + n);
    if (kind == 4) {
        const double* y = (const double*)in;
        std::vector<double> v(y, y + n);
        Mm::unrolledTransform(s.begin(), s.end(), v.begin(), s.begin(), Mm::plusNormalizedSquare<double>(weight));
    }
    else {
        const float* y = (const float*)in;
        std::vector<float> v(y, y + n);
        if (kind == 0)
            Mm::unrolledTransform(s.begin(), s.end(), v.begin(), s.begin(), std::plus<double>());
        else if (kind == 1)
            Mm::unrolledTransform(s.begin(), s.end(), v.begin(), s.begin(), Mm::plusWeighted<double>(weight));
        else if (kind == 2)
            Mm::unrolledTransform(s.begin(), s.end(), v.begin(), s.begin(), Mm::plusSquare<double>());
        else
            Mm::unrolledTransform(s.begin(), s.end(), v.begin(), s.begin(), Mm::plusSquareWeighted<double>(weight));
    }
    std::copy(s.begin(), s.end(), sum);
}
// Mixture::normalizeWeights' norm (Mm/Mixture.cc:68-74) and the covariance estimate's (x - y) / weight -> f32
double ref_log_exp_norm(const double* v, int n) {
    std::vector<double> w(v, v + n);
    return Mm::logExpNorm(w.begin(), w.end());
}
void ref_normalized_minus(const double* x, const double* y, int n, double weight, float* out) {
    std::vector<double> a(x, x + n), b(y, y + n);
    std::vector<float>  r(n);
    std::transform(a.begin(), a.end(), b.begin(), r.begin(), Mm::normalizedMinus<double>(weight));
    std::copy(r.begin(), r.end(), out);
}

// quantize<f32, u8> of the SIMD-diagonal-maximum scorer (Mm/Utilities.hh:190-202)
unsigned ref_quantize_u8(float v) {
    return Mm::quantize<float, unsigned char>()(v);
}

// Math::LevinsonLeastSquares (Math/LevinsonLse.cc, compiled unmodified) driven like AutocorrelationToAutoregressionNode::work
// (Signal/ArEstimator.cc:91-101): gain and a1..aN as the f32 values AutoregressiveCoefficients stores.  Returns 0 when work() fails.
int ref_levinson(const float* R, int n, float* gain, float* a) {
    Math::LevinsonLeastSquares lse;
    std::vector<float>         r(R, R + n), av;
    if (!lse.work(r))
        return 0;
    *gain = (float)lse.gain();
    lse.a(av);
    std::copy(av.begin(), av.end(), a);
    return 1;
}

}  // extern "C"
