// mfcc_tables.hpp -- host-side (f64) construction of the tables the fused MFCC kernel consumes.
#pragma once
#include <vector>

#include "../../include/amx.h"

namespace amx {

// Geometry + tables of one mfcc.flow configuration.  Built once per amx_mfcc handle.
struct MfccTables {
    amx_mfcc_cfg cfg;
    int          frame_len   = 0;  // rint(length * fs)          Signal/Window.cc:69-80
    int          frame_shift = 0;  // rint(shift * fs)
    int          fft_len     = 0;  // next power of two          Signal/FastFourierTransform.cc:30-41
    int          n_bins      = 0;  // fft_len/2 + 1
    int          n_filters   = 0;  // filters of the bank
    int          n_inputs    = 0;  // size of the vector the cosine transform sees: n_filters, or n_filters + 2 (plp.flow copies the first and last output)
    int          n_ceps      = 0;  // output dimension
    int          n_transform = 0;  // rows of the cosine-transform table: n_ceps, or nr-autocorrelation-coefficients (MF-PLP)
    float        norm_div    = 1;  // CosineTransform::apply divides by N_ when normalize is set: inputs (MFCC) or inputs - 1 (N-plus-one)
    float        fft_scale   = 1;  // 1/(f32)fs                  Signal/FastFourierTransform.cc:66-73
    double       fft_output_sample_rate = 0;
    double       mel_max     = 0;

    std::vector<float> window;         // [frame_len]   Hamming
    std::vector<int>   filter_start;   // [n_filters]
    std::vector<int>   filter_end;     // [n_filters]
    std::vector<int>   filter_offset;  // [n_filters+1]
    std::vector<float> filter_weights; // concatenated
    std::vector<float> dct;            // [n_transform][n_inputs]
    std::vector<double> eql;           // [n_inputs] equal-loudness factors (plp.flow), else empty
    std::vector<float> twiddle;        // [fft_len/2][2] cos,sin of +2*pi*k/(fft_len/2)  (complex FFT)
    std::vector<float> split_twiddle;  // [fft_len/4][2] cos,sin of +pi*k/(fft_len/2)    (real split)

    // returns AMX_OK or an error status (message via set_error)
    // fma: the geometry (f64) as the reference's default build contracts it (amx_set_contract / tuning contract=fma)
    int build(const amx_mfcc_cfg& c, bool fma = false);

    long   n_frames(long n_samples) const;
    double frame_start_time(long frame) const;
};

}  // namespace amx
