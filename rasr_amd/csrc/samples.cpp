// samples.cpp -- the sample-stream node in front of the feature chains (samples.flow): signal-dc-detection.
//
// Signal::DcDetection (src/Signal/DcDetection.cc:90-235, DcDetection.hh:75-84) drops runs of "DC" samples -- at least min-dc-length
// seconds that stay within max-dc-increment of the last accepted sample (digital silence, clipped stretches) -- and non-DC segments
// shorter than min-non-dc-segment-length; what it lets through leaves in blocks whose start times carry the gaps, and the window
// buffer behind it flushes at every gap (Signal/WindowBuffer.cc: flush-before-gap).  The node is a sequential scan with three
// counters; it runs on the host over the segment's samples and tells the caller which sample ranges to frame.
#include "common.hpp"

#include <algorithm>
#include <cmath>

int amx_dc_detection(const float* pcm, long long n_samples, double sample_rate, double min_dc_length_s, float max_dc_increment,
                     double min_non_dc_segment_length_s, int maximal_output_size, int merge, long long* starts, long long* lengths,
                     long long capacity, long long* n_blocks) {
    AMX_REQUIRE(n_blocks, AMX_ERR_INVALID, "amx_dc_detection: NULL result");
    *n_blocks = 0;
    AMX_REQUIRE(n_samples >= 0 && (pcm || n_samples == 0), AMX_ERR_INVALID, "amx_dc_detection: bad sample buffer");
    AMX_REQUIRE(sample_rate > 0 && maximal_output_size > 0, AMX_ERR_INVALID, "amx_dc_detection: sample rate and maximal-output-size must be positive");
    AMX_REQUIRE(min_dc_length_s >= 0 && min_non_dc_segment_length_s >= 0 && max_dc_increment >= 0, AMX_ERR_INVALID,
                "amx_dc_detection: negative parameter");
    // DcDetection::init
    const unsigned min_dc  = (unsigned)std::rint(min_dc_length_s * sample_rate);
    const unsigned min_seg = (unsigned)std::rint(min_non_dc_segment_length_s * sample_rate);
    const unsigned long long out_limit = std::max<unsigned long long>(min_seg, (unsigned)maximal_output_size);
    long long          count = 0;
    unsigned long long base = 0;                       // index of buffer_[0] in the segment
    unsigned long long non_dc = 1, dc = 0, seg_len = 0;  // nonDcLength_, dcLength_, nonDcSegmentLength_
    unsigned long long last_start = 0, last_len = 0;
    bool               have_last = false;
    auto emit = [&](unsigned long long start, unsigned long long len) {
        if (merge && have_last && last_start + last_len == start) {  // no gap: the window buffer keeps framing across the block boundary
            last_len += len;
            if (count - 1 < capacity && lengths)
                lengths[count - 1] = (long long)last_len;
            return;
        }
        if (count < capacity && starts && lengths) {
            starts[count]  = (long long)start;
            lengths[count] = (long long)len;
        }
        last_start = start;
        last_len   = len;
        have_last  = true;
        ++count;
    };
    auto flush_block = [&]() {  // copyBlock + eraseBlock
        seg_len += non_dc;
        if (seg_len >= min_seg)
            emit(base, non_dc);
        if (dc > 0)
            seg_len = 0;
        base += non_dc + dc;
        non_dc = 1;
        dc     = 0;
    };
    const unsigned long long n = (unsigned long long)n_samples;
    // get() until the buffer runs out (nextBlock), then flush() once (lastBlock)
    for (;;) {
        bool decided = false;
        while (base + non_dc + dc < n) {
            const float v = pcm[base + non_dc + dc], ref = pcm[base + non_dc - 1];
            if (std::fabs((double)(v - ref)) >= (double)max_dc_increment) {  // isNonDC: f32 difference, fabs(double)
                if (dc >= min_dc) {
                    decided = true;
                    break;
                }
                non_dc += dc;  // include the DC hypothesis
                dc = 0;
                if (non_dc >= out_limit) {
                    decided = true;
                    break;
                }
                ++non_dc;
            }
            else
                ++dc;
        }
        if (!decided)
            break;
        flush_block();
    }
    if (base < n) {  // lastBlock: the buffer is not empty
        if (dc < min_dc) {
            non_dc += dc;
            dc = 0;
        }
        flush_block();
    }
    *n_blocks = count;
    AMX_REQUIRE(!starts || count <= capacity, AMX_ERR_INVALID, "amx_dc_detection: %lld blocks do not fit the capacity %lld", count, capacity);
    return AMX_OK;
}
