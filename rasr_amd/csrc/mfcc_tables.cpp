// mfcc_tables.cpp -- host-side table construction for the fused MFCC kernel.
//
// All geometry is computed in f64 and stored as f32 exactly as the reference nodes do at
// configure()/init() time, so that the device kernel consumes bit-identical tables:
//   Hamming window        Signal/WindowFunction.cc:92-101
//   FFT length            Signal/FastFourierTransform.cc:30-41, FastFourierTransform.hh:299-308
//   mel filter bank       Signal/Filterbank.cc:144-244 (filter builder), :519-567 (stretch-to-cover),
//                         :765-819 (node init), Math/AcousticalAnalyticFunctions.hh:24-60,
//                         Math/AnalyticFunctionFactory.cc:338-341
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

bool almost_integer(double x) {  // FilterBank::isAlmostInteger, tolerance 1e-10
    return std::fabs(x - std::round(x)) < 1e-10;
}

bool almost_equal(double a, double b) {  // Core::isAlmostEqual(f64, f64, 1)
    const double eps = 2.2204460492503131e-16, delta = 2.2250738585072014e-308;
    return std::fabs(a - b) < (std::fabs(a) + std::fabs(b) + delta) * eps;
}

// Flow attributes are strings: "sample-rate" is written with operator<<(f64) (6 significant
// digits, Flow/Attributes.hh:109-113) and parsed back with atof by the next node.
double through_attribute(double v) {
    char text[64];
    std::snprintf(text, sizeof text, "%g", v);
    return std::atof(text);
}

}  // namespace

int MfccTables::build(const amx_mfcc_cfg& c) {
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

    // ---- mel filter bank (triangular, stretch-to-cover, warp-center-positions = true)
    {
        const double  bin_rate = through_attribute(fft_output_sample_rate);
        const Scaling disc2cont{1 / bin_rate};
        const Scaling cont2disc = disc2cont.inverse();
        const MelWarp mel;
        const double  f_min = 0.0;
        const double  f_max = mel(disc2cont((double)(n_bins - 1)));
        mel_max             = f_max;

        const double centre_pos = 0.5;  // symmetrical triangle
        double       width      = c.mel_filter_width;
        double       spacing    = c.mel_spacing == 0 ? centre_pos * width : c.mel_spacing;
        // StretchToCover::getNumberOfFilters + postprocessNumberOfFilters
        double count = (f_max - f_min - width) / spacing + 1;
        if (count < 1)
            count = 1;
        else if (almost_integer(count))
            count = std::round(count);
        const size_t nf = (size_t)std::floor(count);
        // StretchToCover::init: stretch width and spacing so the last filter ends on f_max
        const double coverage = (spacing * (double)(nf - 1) + width) / (f_max - f_min);
        const bool   single_covers = nf == 1 && coverage > 1 && !almost_equal(coverage, 1);
        if (!single_covers) {
            AMX_REQUIRE(almost_equal(coverage, 1) || coverage < 1, AMX_ERR_INVALID, "mfcc: filter bank coverage %f > 1", coverage);
            width /= coverage;
            spacing /= coverage;
        }
        n_filters = (int)nf;
        filter_start.assign(nf, 0);
        filter_end.assign(nf, 0);
        filter_offset.assign(nf + 1, 0);
        filter_weights.clear();
        for (size_t i = 0; i < nf; ++i) {
            const double centre = f_min + spacing * (double)i + centre_pos * width;
            // FilterBuilder::setStart / setEnd
            double left  = std::max(centre - centre_pos * width, f_min);
            double first = cont2disc(mel.inverse(left));
            first        = almost_integer(first) ? std::round(first) : std::ceil(first);
            AMX_REQUIRE(first >= 0, AMX_ERR_INVALID, "mfcc: Start point of the filter at center %f became negative (%d).", centre, (int)first);
            double right = std::min(centre + (1.0 - centre_pos) * width, f_max);
            double last  = cont2disc(mel.inverse(right));
            last         = almost_integer(last) ? std::round(last) + 1 : std::ceil(last);
            const size_t b0 = (size_t)first;
            AMX_REQUIRE(last > 0 && b0 < (size_t)last, AMX_ERR_INVALID,
                        "mfcc: Inconsistent end point of the filter at center %f: start=%zd end=%d.", centre, b0, (int)last);
            const size_t b1 = (size_t)last;
            AMX_REQUIRE(b1 <= (size_t)n_bins, AMX_ERR_INVALID, "mfcc: filter %zu exceeds the spectrum", i);
            filter_start[i]  = (int)b0;
            filter_end[i]    = (int)b1;
            filter_offset[i] = (int)filter_weights.size();
            // FilterBuilder::setWeights: triangle (rounded to f32) * d mel / d f (f64) -> f32
            for (unsigned b = (unsigned)b0; b < b1; ++b) {
                const double warped = mel(disc2cont((double)b));
                float        tri    = (float)((double)1 - std::fabs(warped - centre) / (width / 2));
                tri                 = tri >= 0 ? tri : 0;
                const double slope  = c.warp_differential_unit ? mel.derivative(disc2cont((double)b)) : 1.0;
                filter_weights.push_back((float)(tri * slope));
            }
        }
        filter_offset[nf] = (int)filter_weights.size();
    }

    if (c.front_end == AMX_FRONT_END_MFPLP) {
        // ---- cosine transform for N-plus-one input data (Signal/CosineTransform.cc:46-60), identity warping: the inverse DFT of
        // an even spectrum sampled at N + 1 points -> autocorrelation coefficients
        // CosineTransformNode: nr-outputs <= inputs; AutoregressionToCepstrumNode::init: "Incorrect output size"
        AMX_REQUIRE(c.n_autocorrelation >= 2 && c.n_autocorrelation <= n_filters, AMX_ERR_INVALID,
                    "mfplp: nr-autocorrelation-coefficients (%d) must be in 2..%d (filter bank outputs)", c.n_autocorrelation, n_filters);
        AMX_REQUIRE(c.n_ceps >= 2 && c.n_ceps <= c.n_autocorrelation, AMX_ERR_INVALID,
                    "mfplp: Incorrect output size (%d). 2 < nr-outputs <= %d.", c.n_ceps, c.n_autocorrelation);
        AMX_REQUIRE(c.n_autocorrelation <= 64, AMX_ERR_UNSUPPORTED, "mfplp: LPC order %d > 63", c.n_autocorrelation - 1);
        n_transform    = c.n_autocorrelation;
        const size_t C = (size_t)n_filters, N = C - 1;
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
        AMX_REQUIRE(c.front_end == AMX_FRONT_END_MFCC, AMX_ERR_INVALID, "mfcc: unknown front end %d", c.front_end);
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
