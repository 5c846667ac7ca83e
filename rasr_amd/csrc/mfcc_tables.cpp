// mfcc_tables.cpp -- host-side table construction for the fused MFCC kernel.
//
// All geometry is computed in f64 and stored as f32 exactly as the reference nodes do at
// configure()/init() time, so that the device kernel consumes bit-identical tables:
//   Hamming window        Signal/WindowFunction.cc:92-101
//   FFT length            Signal/FastFourierTransform.cc:30-41, FastFourierTransform.hh:299-308
//   filter bank           Signal/Filterbank.cc:144-244 (filter builder, triangle), :246-275 (trapeze), :432-470
//                         (include-boundary), :519-567 (stretch-to-cover), :575-595 (emphasize-boundary), :765-819 (node init);
//                         mel: Math/AcousticalAnalyticFunctions.hh:24-60, Math/AnalyticFunctionFactory.cc:338-341;
//                         bark: Math/AnalyticFunctionFactory.cc:369-373, Math/SimpleAnalyticFunctions.hh:152-222
//   equal loudness        Signal/VectorTransform.cc:36-83, Math/AcousticalAnalyticFunctions.cc:21-37 (plp.flow)
//   cosine transform      Signal/CosineTransform.cc:62-74
#include "mfcc_tables.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "common.hpp"

namespace amx {
namespace {

// ---- the analytic functions of the reference, as value-semantics functors (all f64) ----
struct Scaling {  // Math::ScalingFunction: f(x) = a*x, inverse scales by 1/a
    double a;
    double operator()(double x) const { return a * x; }
    Scaling inverse() const { return Scaling{1 / a}; }
};

struct MelWarp {  // nest(Scaling(2595), MelWarpingCore) in the continuous domain
    Scaling outer{2595.0};
    double  operator()(double f) const { return outer(std::log10(1.0 + f / 700.0)); }
    // AnalyticNesting::derive(): (outer' o core)(f) * core'(f), outer' = const(2595)
    double derivative(double f) const { return outer.a * (1.0 / std::log(10) / (700.0 + f)); }
    // AnalyticNesting::invert(): core^-1 o outer^-1
    double inverse(double m) const { return (std::pow(10, outer.inverse()(m)) - 1.0) * 700.0; }
};

struct BarkWarp {  // nest(Scaling(6), nest(ArcSinh, Scaling(1/600))) in the continuous domain
    Scaling outer{6.0}, inner{1.0 / 600.0};
    bool    fma = false;  // contract=fma: DerivedArcSinh::value's u * u + 1 is one vfmadd132sd in the reference's default build
    double  operator()(double f) const { return outer(std::asinh(inner(f))); }
    // derive(): (const(6) o g)(f) * g'(f) with g' = (DerivedArcSinh o inner)(f) * const(1/600)
    double derivative(double f) const {
        const double u = inner(f);
        return outer.a * ((1.0 / std::sqrt(fma ? std::fma(u, u, 1.0) : u * u + 1.0)) * inner.a);
    }
    // invert(): inner^-1 o sinh o outer^-1
    double inverse(double b) const { return inner.inverse()(std::sinh(outer.inverse()(b))); }
};

// the warping-function parameter: one of the two, behind one interface
struct Warp {
    int      kind;  // AMX_WARP_MEL / AMX_WARP_BARK
    MelWarp  mel;
    BarkWarp bark;
    double   operator()(double f) const { return kind == AMX_WARP_BARK ? bark(f) : mel(f); }
    double   derivative(double f) const { return kind == AMX_WARP_BARK ? bark.derivative(f) : mel.derivative(f); }
    double   inverse(double w) const { return kind == AMX_WARP_BARK ? bark.inverse(w) : mel.inverse(w); }
};

// Math::EqualLoudnessPreemphasis / EqualLoudnessPreemphasis4Khz
double equal_loudness(double f) {
    const double omega = 2 * M_PI * f, o2 = omega * omega, o4 = o2 * o2, o6 = o4 * o2;
    return (o4 * (o2 + 56.8e6)) / ((o2 + 6.3e6) * (o2 + 6.3e6) * (o2 + 0.38e9) * (o6 / 9.58e26 + 1));
}
double equal_loudness_4khz(double f) {
    const double omega = 2 * M_PI * f, o2 = omega * omega, t4 = o2 / (o2 + 6.3e6);
    return t4 * t4 * (o2 + 56.8e6) / (o2 + 0.38e9);
}

bool almost_integer(double x) {  // FilterBank::isAlmostInteger, tolerance 1e-10
    return std::fabs(x - std::round(x)) < 1e-10;
}

bool almost_equal(double a, double b, double tolerance = 1.0) {  // Core::isAlmostEqual(f64, f64, tolerance)
    const double eps = 2.2204460492503131e-16, delta = 2.2250738585072014e-308;
    return std::fabs(a - b) < (std::fabs(a) + std::fabs(b) + delta) * eps * tolerance;
}

double postprocess_filter_count(double n) {  // Boundary::postprocessNumberOfFilters
    if (n < 1)
        return 1;
    return almost_integer(n) ? std::round(n) : n;
}

float trapeze_weight(double frequency, double centre, double width) {  // TrapezeFilterBuilder::weight
    const double middle = 0.5 / (1.3 - (-2.5));
    const double rel    = frequency - centre;
    const double left   = -middle * width;
    if (rel < left)
        return (float)std::pow(10, rel - left);
    const double right = middle * width;
    if (rel <= right)
        return 1;
    return (float)std::pow(10, -2.5 * (rel - right));
}

// Flow attributes are strings: "sample-rate" is written with operator<<(f64) (6 significant
// digits, Flow/Attributes.hh:109-113) and parsed back with atof by the next node.
double through_attribute(double v) {
    char text[64];
    std::snprintf(text, sizeof text, "%g", v);
    return std::atof(text);
}

}  // namespace

int MfccTables::build(const amx_mfcc_cfg& c, bool fma) {
    cfg = c;
    AMX_REQUIRE(c.sample_rate > 0, AMX_ERR_INVALID, "mfcc: sample rate (%f) is not positive", c.sample_rate);
    AMX_REQUIRE(c.win_len_s > 0 && c.win_shift_s > 0, AMX_ERR_INVALID, "mfcc: window length/shift must be positive");
    AMX_REQUIRE(c.n_ceps >= 1, AMX_ERR_INVALID, "mfcc: nr-outputs must be >= 1");
    AMX_REQUIRE(c.mel_filter_width > 0, AMX_ERR_INVALID, "mfcc: filter-width must be positive");

    frame_len   = (int)(unsigned)std::rint(c.win_len_s * c.sample_rate);
    frame_shift = (int)(unsigned)std::rint(c.win_shift_s * c.sample_rate);
    AMX_REQUIRE(frame_len >= 2 && frame_shift >= 1, AMX_ERR_INVALID, "mfcc: window of %d samples / shift of %d samples", frame_len, frame_shift);

    // FFT length: smallest power of two >= ceil(maximum-input-size * fs)
    unsigned max_len = (unsigned)std::ceil(c.fft_max_input_s * c.sample_rate);
    AMX_REQUIRE(max_len > 0, AMX_ERR_INVALID, "mfcc: maximum-input-size gives an empty FFT");
    double power = std::log((double)max_len) / std::log((double)2);
    power        = almost_equal(power, std::rint(power)) ? std::rint(power) : std::ceil(power);
    AMX_REQUIRE(power < 32, AMX_ERR_INVALID, "mfcc: FFT length overflow");
    fft_len = 1 << (unsigned)power;
    // FastFourierTransform::transform rejects inputs longer than the FFT (criticalError in the node)
    AMX_REQUIRE(frame_len <= fft_len, AMX_ERR_INVALID, "mfcc: Input data size (%d) is larger then maximal input size (%d).", frame_len, fft_len);
    n_bins                 = fft_len / 2 + 1;
    n_ceps                 = c.n_ceps;
    fft_scale              = 1 / (float)c.sample_rate;
    fft_output_sample_rate = (double)fft_len / c.sample_rate;

    // ---- Hamming window, symmetric fill
    window.assign((size_t)frame_len, 0.f);
    {
        unsigned M = (unsigned)frame_len - 1;
        for (unsigned n = 0; n <= M / 2; ++n)
            window[n] = window[M - n] = (float)(0.54 - 0.46 * std::cos(2.0 * M_PI * n / M));
    }

    // ---- filter bank (warp-center-positions = true: the boundary places the centres on the warped axis)
    AMX_REQUIRE(c.filter_type == AMX_FILTER_TRIANGULAR || c.filter_type == AMX_FILTER_TRAPEZE, AMX_ERR_INVALID, "mfcc: unknown filter type %d", c.filter_type);
    AMX_REQUIRE(c.boundary >= AMX_BOUNDARY_STRETCH_TO_COVER && c.boundary <= AMX_BOUNDARY_EMPHASIZE, AMX_ERR_INVALID, "mfcc: unknown boundary type %d", c.boundary);
    AMX_REQUIRE(c.warping == AMX_WARP_MEL || c.warping == AMX_WARP_BARK, AMX_ERR_INVALID, "mfcc: unknown warping function %d", c.warping);
    AMX_REQUIRE(c.front_end >= AMX_FRONT_END_MFCC && c.front_end <= AMX_FRONT_END_PLP, AMX_ERR_INVALID, "mfcc: unknown front end %d", c.front_end);
    eql.clear();
    {
        const double  bin_rate = through_attribute(fft_output_sample_rate);
        const Scaling disc2cont{1 / bin_rate};
        const Scaling cont2disc = disc2cont.inverse();
        Warp          warp;
        warp.kind           = c.warping;
        warp.bark.fma       = fma;
        // a * b + c at the sites the reference's default build contracts (f64: coverage, the
        // include-boundary count, stretch-to-cover's centres, setStart / setEnd)
        auto mad = [&](double a, double b, double c3) { return fma ? std::fma(a, b, c3) : a * b + c3; };
        const double f_min  = 0.0;
        const double f_max  = warp(disc2cont((double)(n_bins - 1)));
        mel_max             = f_max;

        const double centre_pos = c.filter_type == AMX_FILTER_TRAPEZE ? 2.5 / (1.3 - (-2.5)) : 0.5;  // normalizedCenterPosition
        double       width      = c.mel_filter_width;
        double       spacing    = c.mel_spacing == 0 ? centre_pos * width : c.mel_spacing;
        size_t       nf;
        if (c.boundary == AMX_BOUNDARY_STRETCH_TO_COVER) {
            // StretchToCover::getNumberOfFilters, then init: stretch width and spacing so the last filter ends on f_max
            nf                     = (size_t)std::floor(postprocess_filter_count((f_max - f_min - width) / spacing + 1));
            const double coverage  = mad(spacing, (double)(nf - 1), width) / (f_max - f_min);   // vfmadd132sd
            const bool   single_covers = nf == 1 && coverage > 1 && !almost_equal(coverage, 1);
            if (!single_covers) {
                AMX_REQUIRE(almost_equal(coverage, 1) || coverage < 1, AMX_ERR_INVALID, "mfcc: filter bank coverage %f > 1", coverage);
                width /= coverage;
                spacing /= coverage;
            }
        }
        else if (c.boundary == AMX_BOUNDARY_INCLUDE)  // first centre at `spacing`, last filter reaches beyond f_max
            nf = (size_t)std::ceil(postprocess_filter_count(mad(-(1 - centre_pos), width, f_max) / spacing));   // vfnmadd132sd
        else  // emphasize-boundary: first centre at 0
            nf = (size_t)std::floor(postprocess_filter_count(f_max / spacing + 1));
        n_filters = (int)nf;
        filter_start.assign(nf, 0);
        filter_end.assign(nf, 0);
        filter_offset.assign(nf + 1, 0);
        filter_weights.clear();
        for (size_t i = 0; i < nf; ++i) {
            const double centre = c.boundary == AMX_BOUNDARY_STRETCH_TO_COVER ? mad(centre_pos, width, mad(spacing, (double)i, f_min))   // two vfmadd
                                  : c.boundary == AMX_BOUNDARY_INCLUDE        ? spacing * (double)(i + 1)
                                                                              : spacing * (double)i;
            // FilterBuilder::setStart / setEnd
            double left  = std::max(mad(-centre_pos, width, centre), f_min);   // vfnmadd231sd
            double first = cont2disc(warp.inverse(left));
            first        = almost_integer(first) ? std::round(first) : std::ceil(first);
            AMX_REQUIRE(first >= 0, AMX_ERR_INVALID, "mfcc: Start point of the filter at center %f became negative (%d).", centre, (int)first);
            double right = std::min(mad(1.0 - centre_pos, width, centre), f_max);   // vfmadd132sd
            double last  = cont2disc(warp.inverse(right));
            last         = almost_integer(last) ? std::round(last) + 1 : std::ceil(last);
            const size_t b0 = (size_t)first;
            AMX_REQUIRE(last > 0 && b0 < (size_t)last, AMX_ERR_INVALID,
                        "mfcc: Inconsistent end point of the filter at center %f: start=%zd end=%d.", centre, b0, (int)last);
            const size_t b1 = (size_t)last;
            AMX_REQUIRE(b1 <= (size_t)n_bins, AMX_ERR_INVALID, "mfcc: filter %zu exceeds the spectrum", i);
            filter_start[i]  = (int)b0;
            filter_end[i]    = (int)b1;
            filter_offset[i] = (int)filter_weights.size();
            // FilterBuilder::setWeights: shape (rounded to f32) * d warp / d f (f64) -> f32
            for (unsigned b = (unsigned)b0; b < b1; ++b) {
                const double warped = warp(disc2cont((double)b));
                float        shape;
                if (c.filter_type == AMX_FILTER_TRAPEZE)
                    shape = trapeze_weight(warped, centre, width);
                else {
                    shape = (float)((double)1 - std::fabs(warped - centre) / (width / 2));
                    shape = shape >= 0 ? shape : 0;
                }
                const double slope = c.warp_differential_unit ? warp.derivative(disc2cont((double)b)) : 1.0;
                filter_weights.push_back((float)(shape * slope));
            }
        }
        filter_offset[nf] = (int)filter_weights.size();
        n_inputs          = n_filters;

        if (c.front_end == AMX_FRONT_END_PLP) {
            // plp.flow: the filter-bank vector is extended by copies of its first and last element and multiplied by
            // f(i) = equal-loudness(bark^-1(i / sample-rate)), sample-rate = the filter bank's output attribute 1 / spacing as text
            // ("nest(nest(disc-to-cont, invert(bark)), equal-loudness-preemphasis)", Math/AnalyticFunctionFactory.cc:161-180,322-327);
            // the 4 kHz curve unless the top of the axis is significantly greater than 4000 Hz (:543-556)
            AMX_REQUIRE(c.boundary != AMX_BOUNDARY_STRETCH_TO_COVER, AMX_ERR_INVALID,
                        "plp: stretch-to-cover reports an output sample rate of 1, the equal-loudness transform needs the bark axis");
            n_inputs = n_filters + 2;
            const BarkWarp bark;
            const Scaling  index_to_bark{1 / through_attribute((double)1 / spacing)};
            const double   top  = bark.inverse(index_to_bark((double)(n_inputs - 1)));
            const bool     full = top > 4000.0 && !almost_equal(top, 4000.0, 1e12);
            eql.resize((size_t)n_inputs);
            for (int i = 0; i < n_inputs; ++i) {
                const double f = bark.inverse(index_to_bark((double)i));
                eql[(size_t)i] = full ? equal_loudness(f) : equal_loudness_4khz(f);
            }
        }
    }

    if (c.front_end == AMX_FRONT_END_MFPLP || c.front_end == AMX_FRONT_END_PLP) {
        // ---- cosine transform for N-plus-one input data (Signal/CosineTransform.cc:46-60), identity warping: the inverse DFT of
        // an even spectrum sampled at N + 1 points -> autocorrelation coefficients
        // CosineTransformNode: nr-outputs <= inputs; AutoregressionToCepstrumNode::init: "Incorrect output size"
        AMX_REQUIRE(c.n_autocorrelation >= 2 && c.n_autocorrelation <= n_inputs, AMX_ERR_INVALID,
                    "mfplp: nr-autocorrelation-coefficients (%d) must be in 2..%d (cosine transform inputs)", c.n_autocorrelation, n_inputs);
        AMX_REQUIRE(c.n_ceps >= 2 && c.n_ceps <= c.n_autocorrelation, AMX_ERR_INVALID,
                    "mfplp: Incorrect output size (%d). 2 < nr-outputs <= %d.", c.n_ceps, c.n_autocorrelation);
        AMX_REQUIRE(c.n_autocorrelation <= 64, AMX_ERR_UNSUPPORTED, "mfplp: LPC order %d > 63", c.n_autocorrelation - 1);
        n_transform    = c.n_autocorrelation;
        const size_t C = (size_t)n_inputs, N = C - 1;
        norm_div       = (float)N;
        dct.assign((size_t)n_transform * C, 0.f);
        for (size_t k = 0; k < (size_t)n_transform; ++k) {
            dct[k * C + 0] = (float)0.5;
            dct[k * C + N] = (float)(0.5 * std::pow(-1, (double)k));
            for (size_t n = 1; n < N; ++n) {
                double omega   = M_PI * n / N;
                dct[k * C + n] = (float)(std::cos(omega * k) * 1.0);
            }
        }
    }
    else {
        // ---- DCT-II, even about N-1/2, identity warping
        const size_t N = (size_t)n_filters;
        n_transform    = n_ceps;
        norm_div       = (float)N;
        dct.assign((size_t)n_ceps * N, 0.f);
        for (size_t k = 0; k < (size_t)n_ceps; ++k)
            for (size_t n = 0; n < N; ++n) {
                double omega   = M_PI * (n + 0.5) / N;
                dct[k * N + n] = (float)(std::cos(omega * k) * 1.0);
            }
    }

    // ---- FFT twiddles (device uses direct table values, rounded from f64; the reference's
    // trigonometric recurrence differs from these by O(1e-16), far below f32 resolution)
    {
        const int nc = fft_len / 2;  // complex points
        twiddle.assign((size_t)nc * 2, 0.f);
        for (int k = 0; k < nc; ++k) {
            double a           = 2.0 * M_PI * (double)k / (double)nc;
            twiddle[2 * k]     = (float)std::cos(a);
            twiddle[2 * k + 1] = (float)std::sin(a);
        }
        split_twiddle.assign((size_t)(fft_len / 4) * 2 + 2, 0.f);
        for (int k = 0; k <= fft_len / 4; ++k) {
            double a = M_PI * (double)k / (double)nc;
            if (2 * k + 1 < (int)split_twiddle.size()) {
                split_twiddle[2 * k]     = (float)std::cos(a);
                split_twiddle[2 * k + 1] = (float)std::sin(a);
            }
        }
    }
    return AMX_OK;
}

// Signal/WindowBuffer.cc:84-125: get() while >= 2*max(len,shift) buffered, then flush() every
// `shift` samples until the rest fits into one window; the last frame is short.
long MfccTables::n_frames(long n) const {
    if (n <= 0)
        return 0;
    const long reach = std::max(frame_len, frame_shift);
    if (n <= reach)
        return 1;
    return (n - reach + frame_shift - 1) / frame_shift + 1;
}

double MfccTables::frame_start_time(long frame) const {
    // bufferStartTime_ += (Time)shift_ / (Time)sampleRate_ per emitted frame
    double       t    = 0;
    const double step = (double)frame_shift / cfg.sample_rate;
    for (long k = 0; k < frame; ++k)
        t += step;
    return t;
}

}  // namespace amx
