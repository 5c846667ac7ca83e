// mfcc.hip -- fused MFCC front-end for gfx950 and the amx_mfcc_* part of the C ABI.
//
// One kernel replaces the per-frame Flow chain of mfcc.flow (Tools/FeatureExtraction/share/
// mfcc.flow:8-34): preemphasis -> Hamming framing -> zero-pad -> real FFT (x 1/fs) -> |X| ->
// mel triangular filter bank -> log10 -> DCT-II.
//
// Mapping to the hardware
//   * a workgroup (4 wavefronts) owns a TILE of up to `frames_per_tile` consecutive frames of one
//     segment.  The PCM span of the tile ((FT-1)*shift + len + 1 samples) is read from HBM once,
//     coalesced, into LDS; the 2.5x frame overlap is served from LDS, not from memory.
//   * each 64-lane wavefront transforms one frame at a time: the fft_len real samples are
//     packed as NC = fft_len/2 complex points, 4 points per lane for NC = 256, and run through
//     an in-LDS Stockham radix-4 (+ one radix-2 stage when log2 NC is odd) with per-lane
//     twiddles held in registers across frames.  Only wavefront-level ordering is needed
//     between stages (DS operations of one wave complete in order), so the 4 waves of a
//     workgroup never wait for each other inside the frame loop.
//   * real split, 1/fs scale and amplitude are a lane-parallel epilogue of the FFT; the mel
//     filters (lane = filter) and the DCT (lane = cepstral coefficient) read LDS-resident tables.
//   * algorithmic HBM traffic per frame: shift*4 B of PCM in + n_ceps*4 B out (800 B at 16 kHz /
//     40 ceps) -- the kernel is bounded by HBM bandwidth once VALU/LDS time is below that.
//
// Numerics: f32 throughout like the reference; window/preemphasis/filter-bank/DCT use separate
// multiply and add (this TU is compiled with -ffp-contract=off) in the reference's summation
// order; the FFT butterflies use explicit fmaf with table twiddles, so spectra differ from the
// reference's f64-recurrence butterflies at the 1e-7 relative level ("parity").
#include "common.hpp"
#include "mfcc_tables.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace amx {

struct MfccTile {
    long long sample_base;  // first sample of the segment in the concatenated PCM buffer
    long long out_frame;    // global index of the tile's first frame in the output
    int       n_samples;    // segment length
    int       frame0;       // first frame of the tile within the segment
    int       n_frames;     // frames in this tile
    int       pad_;
};

struct MfccParams {
    const void*     pcm;    // f32 samples, or s16 (kernel variant bit 2): s16 widened without scaling like Flow/TypeConverter.hh:35-43
    float*          ceps;
    const MfccTile* tiles;
    const float*    window;
    const int*      fstart;
    const int*      fend;
    const int*      foff;
    const float*    fweights;
    const float*    dct_t;  // transposed [n_filters][n_ceps]
    const double*   eql;    // plp.flow: equal-loudness factor per cosine-transform input, else null
    const float2*   tw;     // [NC]  e^{+2 pi i k / NC}
    const float2*   stw;    // [NC/2+1] e^{+pi i k / NC}
    int             frame_len, frame_shift, n_filters, n_ceps, n_weights;  // n_filters = cosine-transform inputs (plp.flow: first / last filter twice)
    int             frames_per_tile, n_tiles;
    float           alpha, fft_scale;
    int             apply_scale, dct_normalize;
    int             front_end;   // 1: power spectrum into the filter bank, ^plp_power instead of log10 (mfplp.flow)
    float           norm_div, plp_power;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    // a * w with fused multiply-adds
    return make_float2(fmaf(a.x, w.x, -(a.y * w.y)), fmaf(a.x, w.y, a.y * w.x));
}

__device__ __forceinline__ void wave_sync() {
    // order this wave's LDS traffic for the compiler; the hardware keeps DS ops of a wave in order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template<int NC>
struct FftPlan {
    static constexpr int log2nc() {
        int l = 0;
        for (int n = NC; n > 1; n >>= 1)
            ++l;
        return l;
    }
    static constexpr int L   = log2nc();
    static constexpr int S4  = L / 2;         // radix-4 stages
    static constexpr bool R2 = (L % 2) != 0;  // one trailing radix-2 stage
    static constexpr int B4  = (NC / 4 + 63) / 64;  // radix-4 butterflies per lane
    static constexpr int B2  = (NC / 2 + 63) / 64;  // radix-2 butterflies per lane
};

constexpr int FT = 16;  // frames per tile = M of the DCT MFMA
// wavefronts per workgroup: 4 (four frames of a tile each).  With 8 the per-wave FFT buffers push the workgroup to 57 KB of LDS and two
// workgroups per CU; 4 give 49 KB and three, whose barrier-separated phases overlap better (A/B on one box: 1.10 -> 1.05 ms)
__host__ __device__ constexpr int mfcc_waves(int nc) { return nc >= 1024 ? 4 : 4; }
// FFT work buffer index swizzle (a bijection inside every 16-point block): makes the stride-4 / stride-16 Stockham
// writes of the first two radix-4 stages bank-conflict free (ds_write_b64, 16-lane groups)
__host__ __device__ constexpr int zpad(int i) { return i ^ (5 * ((i >> 4) & 3)); }

typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- radix-16 form of the 256-point complex transform (VAR & 16; amx_mfcc_cfg.tuning fft=r16).  A wave works on FOUR frames at once:
// lane = 16 q + m holds 16 complex points of frame q in registers, so a 16-point DFT is register arithmetic (two levels of radix-4
// butterflies, constants W16^j) and the whole transform is  DFT-16 over n1 | twiddle W256^(n2 k1) | ONE transposition through LDS |
// DFT-16 over n2  (n = 16 n1 + n2, k = k1 + 16 k2) -- two LDS round trips per frame (transposition, natural order for the real
// split) instead of the Stockham form's five, and a tile's chain of dependent LDS round trips is walked once per wave, not once per
// frame.  Work buffer of a frame: 16 rows of 18 complex (row stride 144 B: sixteen lanes' ds_read_b128 of 16 different rows touch
// 16 different bank groups) + 16 complex of skew between frames (neighbouring frames' rows start 32 banks apart).
// MEASURED (config 2, 993 k frames, profiles/r04/mfcc_r16_ab.log, mfcc_r16_timeline.log): parity-green (tests/test_mfcc_gpu.py) and
// SLOWER, 1.04 ms against 0.76 ms for the Stockham stages.  One 2.4 KB work buffer per frame in flight puts a workgroup at 72 KB of
// LDS: two workgroups per CU, two waves per SIMD -- and a lone wave issues one vector instruction per four cycles, so ~1000 vector
// instructions of a batch are >= 2 us whatever their dependences (batch 5.7 us: samples 1.5, two DFT-16 0.9, transposition 0.75,
// split + next fetch 2.6), while the mel filter bank and the DCT behind the workgroup barriers (7000-8000 ticks of a 17 800-tick
// tile) find one other workgroup to overlap with instead of three.  History of the batch: 1.22 ms (sample loads inside uniform
// branches: sixteen serial memory round trips) -> 1.10 (branch-free loads) -> 1.01 (split operands read before the amplitude
// stores) -> 1.04-1.05 with the next tile's samples fetched ahead (no gain: the wait was not the memory's).  mfcc.flow only.  Kept
// as tuning fft=r16 (round-3 review item 7, second form).
constexpr int kR16Row = 18, kR16Frame = 16 * kR16Row + 16;

__device__ __forceinline__ void r16_dft4(float2& x0, float2& x1, float2& x2, float2& x3) {  // y_d = sum_b x_b (+i)^(b d), in place
    const float2 a = make_float2(x0.x + x2.x, x0.y + x2.y), b = make_float2(x0.x - x2.x, x0.y - x2.y);
    const float2 c = make_float2(x1.x + x3.x, x1.y + x3.y), d = make_float2(x1.x - x3.x, x1.y - x3.y);
    x0 = make_float2(a.x + c.x, a.y + c.y);
    x1 = make_float2(b.x - d.y, b.y + d.x);
    x2 = make_float2(a.x - c.x, a.y - c.y);
    x3 = make_float2(b.x + d.y, b.y - d.x);
}
// position of output index k = c + 4 d of r16_dft16 in its register array
__host__ __device__ constexpr int r16_reg(int k) { return 4 * (k & 3) + (k >> 2); }
// 16-point DFT with W16 = e^{+2 pi i / 16}, in place: input z[n], n = 4 a + b; output index k = c + 4 d at z[r16_reg(k)] = z[4 c + d]
__device__ __forceinline__ void r16_dft16(float2 (&z)[16]) {
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, R = 0.70710678118654752440f;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        r16_dft4(z[b], z[4 + b], z[8 + b], z[12 + b]);  // z[4 c + b] = T[b][c] = sum_a z[4 a + b] W4^(a c)
    // T[b][c] *= W16^(b c)
    auto mul = [](float2& v, float wr, float wi) { v = make_float2(fmaf(v.x, wr, -(v.y * wi)), fmaf(v.x, wi, v.y * wr)); };
    auto mulr = [&](float2& v) { v = make_float2(R * (v.x - v.y), R * (v.x + v.y)); };    // W16^2 = (R, R)
    auto muli = [](float2& v) { v = make_float2(-v.y, v.x); };                            // W16^4 = i
    auto mulm = [&](float2& v) { v = make_float2(-R * (v.x + v.y), R * (v.x - v.y)); };   // W16^6 = (-R, R)
    mul(z[4 * 1 + 1], C1, S1);    // b = 1, c = 1: W^1
    mulr(z[4 * 2 + 1]);           // b = 1, c = 2: W^2
    mul(z[4 * 3 + 1], S1, C1);    // b = 1, c = 3: W^3
    mulr(z[4 * 1 + 2]);           // b = 2, c = 1: W^2
    muli(z[4 * 2 + 2]);           // b = 2, c = 2: W^4
    mulm(z[4 * 3 + 2]);           // b = 2, c = 3: W^6
    mul(z[4 * 1 + 3], S1, C1);    // b = 3, c = 1: W^3
    mulm(z[4 * 2 + 3]);           // b = 3, c = 2: W^6
    mul(z[4 * 3 + 3], -C1, -S1);  // b = 3, c = 3: W^9
#pragma unroll
    for (int c = 0; c < 4; ++c)
        r16_dft4(z[4 * c], z[4 * c + 1], z[4 * c + 2], z[4 * c + 3]);  // z[4 c + d] = sum_b T[b][c] W4^(b d) = D[c + 4 d]
}

// LDS carve-up shared by the kernel and the host-side size computation (all sizes in floats)
struct MfccLds {
    int y, amp, lm, dct, fw, fidx, fft, total;
    int y_len, amp_ld, lm_ld, dct_ld, kpad;
    __host__ __device__ MfccLds(int frame_len, int frame_shift, int fft_len, int n_filters, int n_ceps, int n_weights, bool r16 = false) {
        auto r4 = [](int v) { return (v + 3) & ~3; };
        y_len   = (FT - 1) * frame_shift + (fft_len > frame_len ? fft_len : frame_len);  // zero-pad region of the last frame included
        amp_ld  = fft_len / 2 + 1;                                                       // 2^k + 1: odd, conflict-free across frames
        kpad    = r4(n_filters);                                                         // K of the DCT, multiple of 4
        lm_ld   = kpad + 1;
        dct_ld  = (n_ceps + 15) & ~15;
        y       = 0;  // (no staged PCM span any more, see the kernel)
        amp     = 0;
        lm      = amp + r4(FT * amp_ld);
        dct     = lm + r4(FT * lm_ld);
        fw      = dct + r4(kpad * dct_ld);
        fidx    = fw + r4(n_weights);
        fft     = fidx + r4(3 * n_filters);
        total   = fft + (r16 ? FT * 2 * kR16Frame + 256 + 512                         // radix-16 form: one work buffer per FRAME of the tile + 128 split twiddles + 256 window pairs
                             : mfcc_waves(fft_len / 2) * 2 * (zpad(fft_len / 2) + 4));  // per wave: padded NC float2
    }
};

// generic-vector-f32-power (Flow/SimpleFunction.hh:143-153): the node calls the unqualified pow on two floats, which is
// ::pow(double, double) with the headers that file sees, and narrows the result.  Out of line: inlined, the f64 routine's
// registers spill the MFCC path of the same kernel.
__device__ __noinline__ float power_node(float v, float power) {
    return (float)pow((double)v, (double)power);
}

// VAR bit 1: the 256-point complex FFT (fft length 512) on the f32 matrix cores instead of four Stockham radix-4 stages through LDS;
// bit 2: s16 samples.
//
// FFT on MFMA (NC = 256 = 16 x 16, n = 16 n1 + n2, k = k1 + 16 k2, W_N = e^{+2 pi i / N} -- the reference's sign):
//     Z[k1 + 16 k2] = sum_n2 ( (sum_n1 z[16 n1 + n2] W16^(n1 k1)) W256^(n2 k1) ) W16^(n2 k2)
// i.e. two 16x16x16 complex matrix products with a twiddle in between, all in registers: a lane already holds the complex points
// lane + 64 r (r = 0..3), which IS the A operand of v_mfma_f32_16x16x4_f32 for D^T[n2][k1] = sum_n1 Z^T[n2][n1] F[n1][k1] (K-step r
// covers n1 = 4 r + lane / 16); the result's layout (lane: n2 = 4 (lane / 16) + reg, k1 = lane % 16) is the B operand of the second
// product X^T[k2][k1] = sum_n2 F[k2][n2] B[n2][k1] when the constant A operand is taken in the order n2 = 4 (lane / 16) + K-step.
// A complex product costs three real ones (P1 = Zr Fr, P2 = Zi Fi, P3 = (Zr + Zi)(Fr + Fi); re = P1 - P2, im = P3 - P1 - P2):
// 24 MFMAs per frame on the otherwise idle matrix pipe replace ~160 of the frame's ~310 vector instructions and three of its four LDS
// round trips; the f32 MFMA is an exact fma chain, the error class is that of the f32 butterflies.
// bit 4 (NC = 256; amx_mfcc_cfg.tuning prefetch=1): a wave fetches the samples of its NEXT frame while it transforms the current one
// (12 more registers: 127 instead of 109, still four workgroups per CU).  Measured on config 2, one box, twice: 0.795 ms against
// 0.743 ms -- SLOWER: the samples of a wave's next frame are the neighbours' current ones (60 % overlap, L1 / L2 hits), their
// latency is not what a frame waits for.  Kept as an A/B variant (round-3 review item 7).
#ifdef AMX_LAB
// lab builds: s_memtime stamps of workgroup 0, [wave < 4][tile < 32][7]: tile start, phase B done, barrier passed, phase C done,
// barrier passed, phase D done, last barrier passed; amx_lab_mfcc_stamps (tools/mfcc_timeline.py)
__device__ unsigned long long mfcc_stamps[4 * 32 * 8];
// radix-16 phase B of the same workgroup: samples in registers, first DFT + twiddle, transposed row read, second DFT, split done
__device__ unsigned long long mfcc_r16_stamps[4 * 32 * 8];
#define MFCC_R16_STAMP(k)                                                                                      \
    do {                                                                                                       \
        if (blockIdx.x == 0 && lane == 0 && wave < 4 && lab_tile < 32)                                         \
            mfcc_r16_stamps[(wave * 32 + lab_tile) * 8 + (k)] = __builtin_amdgcn_s_memtime();                  \
    } while (0)
#define MFCC_STAMP(k)                                                                                          \
    do {                                                                                                       \
        if (blockIdx.x == 0 && lane == 0 && wave < 4 && lab_tile < 32)                                         \
            mfcc_stamps[(wave * 32 + lab_tile) * 8 + (k)] = __builtin_amdgcn_s_memtime();                      \
    } while (0)
#else
#define MFCC_STAMP(k) \
    do {              \
    } while (0)
#define MFCC_R16_STAMP(k) \
    do {                  \
    } while (0)
#endif

template<int NC, int VAR>
__global__ __launch_bounds__(mfcc_waves(NC) * 64, (NC >= 1024 ? 1 : ((VAR & 16) != 0 ? 2 : (VAR & 8) != 0 ? 3 : 4))) void mfcc_kernel(MfccParams p) {
    using P = FftPlan<NC>;
    constexpr bool R16 = (VAR & 16) != 0 && NC == 256;  // radix-16 register butterflies, four frames per wave (see r16_dft16)
    constexpr bool MF  = (VAR & 1) != 0 && NC == 256 && !R16;
    constexpr bool S16 = (VAR & 2) != 0;
    constexpr bool PF  = (VAR & 4) != 0 && NC == 256;
    using Sample       = typename std::conditional<S16, short, float>::type;
    constexpr int MW = mfcc_waves(NC), MT = MW * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform for the compiler: frame base, segment tests and LDS rows become scalar work

    const MfccLds  L(p.frame_len, p.frame_shift, 2 * NC, p.n_filters, p.n_ceps, p.n_weights, R16);
    float*  s_amp = smem + L.amp;   // [FT][amp_ld] amplitude spectra
    float*  s_lm  = smem + L.lm;    // [FT][lm_ld]  log10 mel energies (columns >= n_filters are 0)
    float*  s_dct = smem + L.dct;   // [kpad][dct_ld] DCT matrix, transposed and zero padded
    float*  s_fw  = smem + L.fw;    // filter weights
    int*    s_fs  = (int*)(smem + L.fidx);
    int*    s_fe  = s_fs + p.n_filters;
    int*    s_fo  = s_fe + p.n_filters;
    float2* s_z   = (float2*)(smem + L.fft) + wave * (zpad(NC) + 4);  // this wave's FFT work buffer (index via zpad)

    // ---- tables: staged ONCE per workgroup (workgroups are persistent and loop over tiles)
    for (int i = tid; i < p.n_weights; i += MT)
        s_fw[i] = p.fweights[i];
    for (int i = tid; i < p.n_filters; i += MT) {
        s_fs[i] = p.fstart[i];
        s_fe[i] = p.fend[i];
        s_fo[i] = p.foff[i];
    }
    // DCT^T [kpad][dct_ld], zero padded (p.dct_t is [n_filters][n_ceps])
    for (int i = tid; i < L.kpad * L.dct_ld; i += MT) {
        const int n = i / L.dct_ld, c = i - n * L.dct_ld;
        s_dct[i]    = (n < p.n_filters && c < p.n_ceps) ? p.dct_t[n * p.n_ceps + c] : 0.f;
    }
    for (int i = tid; i < FT * L.lm_ld; i += MT)
        s_lm[i] = 0.f;
    if constexpr (R16)
        __syncthreads();  // phase B of the first tile reads the window and split-twiddle tables

    // ---- per-lane constants: window coefficients of this lane's samples and FFT twiddles
    // complex point c = lane + 64*b + r*(NC/4) holds samples 2c, 2c+1 of the zero-padded frame
    float wlo[P::B4][4], whi[P::B4][4];
#pragma unroll
    for (int b = 0; b < P::B4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = lane + 64 * b + r * (NC / 4);
            wlo[b][r]   = (lane + 64 * b < NC / 4 && 2 * c < p.frame_len) ? p.window[2 * c] : 0.f;
            whi[b][r]   = (lane + 64 * b < NC / 4 && 2 * c + 1 < p.frame_len) ? p.window[2 * c + 1] : 0.f;
        }
    float2 tw4[P::S4 > 1 ? P::S4 - 1 : 1][P::B4][3];
#pragma unroll
    for (int s = 1; s < P::S4; ++s) {
        const int Ns = 1 << (2 * s);
#pragma unroll
        for (int b = 0; b < P::B4; ++b) {
            const int j   = lane + 64 * b;
            const int k   = j & (Ns - 1);
            const int idx = k * (NC / (4 * Ns));
#pragma unroll
            for (int r = 0; r < 3; ++r)
                tw4[s - 1][b][r] = p.tw[((r + 1) * idx) & (NC - 1)];
        }
    }
    float2 tw2[P::B2];
    if (P::R2) {
#pragma unroll
        for (int b = 0; b < P::B2; ++b)
            tw2[b] = p.tw[(lane + 64 * b) & (NC / 2 - 1)];
    }
    // split twiddles of the bin pairs this lane owns
    constexpr int NPAIR = NC / 2 - 1;
    constexpr int PB    = (NPAIR + 63) / 64 > 0 ? (NPAIR + 63) / 64 : 1;
    float2        stw[PB];
#pragma unroll
    for (int b = 0; b < PB; ++b) {
        const int i = 1 + lane + 64 * b;
        stw[b]      = p.stw[i <= NPAIR ? i : 0];
    }
    // MFMA FFT: this lane's constants (m = lane % 16, q = lane / 16; W = e^{+2 pi i / 256} from the host table)
    float f1r[4], f1i[4], f1s[4], wr[4], wi[4], gr[4], gi[4], gs[4];
    if (MF) {
        const int m = lane & 15, q = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 f = p.tw[(16 * ((4 * r + q) * m)) & (NC - 1)];  // F[n1 = 4 r + q][k1 = m]
            f1r[r] = f.x, f1i[r] = f.y, f1s[r] = f.x + f.y;
            const float2 w = p.tw[(m * (4 * q + r)) & (NC - 1)];         // W256^(k1 n2), n2 = 4 q + r
            wr[r] = w.x, wi[r] = w.y;
            const float2 g = p.tw[(16 * (m * (4 * q + r))) & (NC - 1)];  // F[k2 = m][n2 = 4 q + r]
            gr[r] = g.x, gi[r] = g.y, gs[r] = g.x + g.y;
        }
    }
    // radix-16 form: lane = 16 q + m; window pairs of its points c = 16 n1 + m, W256^(m k1) in r16_dft16's output order, split
    // twiddles of its bin pairs i = m + 16 j
    [[maybe_unused]] float2 r_tw[16];
    [[maybe_unused]] float2* s_stw  = (float2*)(smem + L.fft + FT * 2 * kR16Frame);  // split twiddles of bins 0..127 (radix-16 form; registers are short there)
    [[maybe_unused]] float2* s_win2 = s_stw + 128;                                   // window values of the samples (2c, 2c + 1) of point c, zero behind the window
    if constexpr (R16) {
        const int m = lane & 15;
        for (int i = tid; i < 128; i += MT)
            s_stw[i] = p.stw[i];
        for (int c = tid; c < 256; c += MT)
            s_win2[c] = make_float2(2 * c < p.frame_len ? p.window[2 * c] : 0.f, 2 * c + 1 < p.frame_len ? p.window[2 * c + 1] : 0.f);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            r_tw[r] = p.tw[(m * ((r >> 2) + 4 * (r & 3))) & 255];
    }
    const float scale   = p.fft_scale;
    const bool  doscale = p.apply_scale != 0;
    const float alpha   = p.alpha;
    const bool  alpha1  = (alpha == 1.0f);

  // radix-16 form: the samples of this wave's four frames of the workgroup's NEXT tile, fetched at the end of phase B so that their
  // memory round trip (1.9 us with two waves per SIMD and nothing else to issue, tools/mfcc_timeline.py) runs under phases C and D
  // Two registers per point: sample 2c - 1 is the neighbour lane's 2c + 1 (lane m - 1 of the same n1, or lane 15 of n1 - 1), taken by
  // two DPP moves when the batch is transformed; pfp = the sample in front of the frame (lane m = 0 of n1 = 0).
  [[maybe_unused]] float pf0[16], pf1[16], pfp = 0.f;
  [[maybe_unused]] bool  pf_have = false;  // wave-uniform
  // all 32 + 1 loads of a batch without a branch between them (a load inside a conditional block is waited for at the block's end:
  // sixteen serial memory round trips, 3.4 us of a batch's 6, tools/mfcc_timeline.py); a point behind the window reads sample 1.
  // Batches whose frames all lie inside their segment with a predecessor sample (wave-uniform test) load without guards; the
  // others clamp every index into the segment and zero what lies behind it (pf_guard: the transform masks those points).
  [[maybe_unused]] int  pf_nvalid = 0;      // samples of the segment from this lane's frame start on (clamped to 2^30)
  [[maybe_unused]] bool pf_guard  = false;  // wave-uniform
  [[maybe_unused]] auto fetch_batch = [&](const MfccTile& nt) -> bool {
      if (wave * 4 >= nt.n_frames)
          return false;
      const int       m  = lane & 15, fq = wave * 4 + (lane >> 4);
      const int       fl = fq < nt.n_frames ? fq : nt.n_frames - 1;
      const long long fb = (long long)(nt.frame0 + fl) * p.frame_shift;
      const long long nv = (long long)nt.n_samples - fb;
      pf_nvalid          = (int)(nv < (1ll << 30) ? nv : (1ll << 30));
      pf_guard           = !__all(fb >= 1 && fb + p.frame_len < (long long)nt.n_samples);
      const Sample* fr   = (const Sample*)p.pcm + nt.sample_base + fb;
      if (!pf_guard) {
#pragma unroll
          for (int n1 = 0; n1 < 16; ++n1) {
              const int   c  = 16 * n1 + m;
              const bool  w  = 2 * c < p.frame_len;
              const int   o  = w ? 2 * c : 1;
              pf0[n1]        = (float)fr[o];  // RAW values: a select here would wait for the load (the window mask is applied by the consumer)
              pf1[n1]        = (float)fr[o + 1];
          }
          pfp = (float)fr[-1];
      }
      else {
          const int last = pf_nvalid - 1;  // >= 0: a frame starts inside its segment
#pragma unroll
          for (int n1 = 0; n1 < 16; ++n1) {
              const int   c  = 16 * n1 + m;
              const bool  w  = 2 * c < p.frame_len;
              const int   o  = w ? 2 * c : 1;
              pf0[n1]        = (float)fr[o < last ? o : last];  // behind the segment: some sample of it, masked out by the consumer
              pf1[n1]        = (float)fr[o + 1 < last ? o + 1 : last];
          }
          pfp = fb >= 1 ? (float)fr[-1] : (float)fr[0];  // segment start: previous_ = x[0] (Signal/Preemphasis.cc)
      }
      return true;
  };
  // PF: samples 2c - 1, 2c, 2c + 1 of this wave's next frame, RAW and as the triple one load instruction delivers (carried from
  // frame to frame in exactly the registers the load writes: any other arrangement costs a copy per value behind a wait)
  struct Raw3 {
      Sample m, x0, x1;
  };
  [[maybe_unused]] Raw3 pn[4];
  [[maybe_unused]] bool have = false;  // wave-uniform: pn holds the frame about to be transformed (carried across tiles too)
  [[maybe_unused]] int lab_tile = -1;
  for (int tile_id = blockIdx.x; tile_id < p.n_tiles; tile_id += gridDim.x) {
    const MfccTile tile = p.tiles[tile_id];
    ++lab_tile;
    MFCC_STAMP(0);
    // ================= (round 1's phase A -- staging the tile's PCM span in LDS behind a workgroup barrier -- is gone.)  A frame's
    // samples are read where they are used: neighbouring frames overlap, so the re-reads come from L2; 11.6 KB of LDS and one
    // barrier per tile less, four workgroups per CU for MFCC-40 instead of three.  A/B on one box (tools/ab_mfcc.sh, ms per
    // 993 k frames, staged -> direct at four workgroups per CU): mfcc.flow 1.02 -> 0.95, mfplp.flow 1.09 -> 0.97, plp.flow 1.32 ->
    // 1.17; five workgroups per CU (the PLP variants' smaller LDS would allow them) are 30 % SLOWER, hence the cap in launch_mfcc.
    const Sample*   seg   = (const Sample*)p.pcm + tile.sample_base;
    const long long nseg  = tile.n_samples;
    // PF: the tile this workgroup walks next (its first frame of this wave is fetched behind the wave's last frame of this tile)
    [[maybe_unused]] const bool     has_next = PF && tile_id + (int)gridDim.x < p.n_tiles;
    [[maybe_unused]] const MfccTile ntile    = p.tiles[has_next ? tile_id + (int)gridDim.x : tile_id];
    // ================= phase B: one frame per wavefront: FFT -> split -> |X| into s_amp[f][*]
    if constexpr (R16) {
      // ================= phase B, radix-16 form: wave w transforms frames 4 w .. 4 w + 3 of the tile together
      if (wave * 4 < tile.n_frames) {  // wave-uniform
        const int  q = lane >> 4, m = lane & 15;
        const int  f = wave * 4 + q;
        const bool live = f < tile.n_frames;
        const int  fl = live ? f : tile.n_frames - 1;  // a lane group behind the tile's last frame repeats it (nothing written)
        const long long fbase = (long long)(tile.frame0 + fl) * p.frame_shift;
        float2 z[16];
        if (!pf_have)  // first tile of the workgroup
            fetch_batch(tile);
        const bool guard  = pf_guard;
        const int  nvalid = pf_nvalid;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            // predecessor sample: lane 0 of a row takes lane 15 of the previous point's odd sample (row_ror:1), the other lanes
            // their left neighbour's odd sample of this point (row_shr:1 keeps `old` in lane 0)
            const bool  w    = 2 * (16 * n1 + m) < p.frame_len;
            const int   wrap = n1 > 0 ? __builtin_amdgcn_update_dpp(0, __float_as_int(pf1[n1 > 0 ? n1 - 1 : 0]), 0x121, 0xf, 0xf, false)
                                      : __float_as_int(pfp);
            const float xm   = w ? __int_as_float(__builtin_amdgcn_update_dpp(wrap, __float_as_int(pf1[n1]), 0x111, 0xf, 0xf, false)) : 0.f;
            const float x0v  = w ? pf0[n1] : 0.f;
            const float x1v  = w ? pf1[n1] : 0.f;
            float       y0, y1;
            if (alpha1) {  // Signal/Preemphasis.cc:69-75
                y0 = x0v - xm;
                y1 = x1v - x0v;
            }
            else {  // :62-67
                const float p0 = alpha * xm, p1 = alpha * x0v;
                y0             = x0v - p0;
                y1             = x1v - p1;
            }
            z[n1] = make_float2(s_win2[16 * n1 + m].x * y0, s_win2[16 * n1 + m].y * y1);  // WindowFunction::work
        }
        if (guard) {  // zero padding behind the segment (short last frames): the pre-emphasised sample itself is zero there (whatever the clamped loads read)
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int n = 2 * (16 * n1 + m);
                z[n1].x     = n < nvalid ? z[n1].x : 0.f;
                z[n1].y     = n + 1 < nvalid ? z[n1].y : 0.f;
            }
        }
        // D[n2 = m][k1] = sum_n1 z[16 n1 + m] W16^(n1 k1), then the twiddle W256^(m k1)
        asm volatile("" ::"v"(z[0].x), "v"(z[15].y));
        MFCC_R16_STAMP(0);
        r16_dft16(z);
#pragma unroll
        for (int r = 1; r < 16; ++r)  // (register 0 holds k1 = 0: unit twiddle)
            z[r] = cmul(z[r], r_tw[r]);
        // transposition: row k1, column n2 = m
        asm volatile("" ::"v"(z[0].x), "v"(z[15].y));
        MFCC_R16_STAMP(1);
        float2* zb = (float2*)(smem + L.fft) + f * kR16Frame;
        wave_sync();  // the previous tile's split readers of this buffer are done (a wave's LDS operations stay in order)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            zb[((r >> 2) + 4 * (r & 3)) * kR16Row + m] = z[r];
        wave_sync();
        // lane (q, k1 = m) reads row k1: E[n2][k1], n2 = 0..15
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 v = *(const f32x4*)(zb + m * kR16Row + 2 * t);
            z[2 * t]      = make_float2(v[0], v[1]);
            z[2 * t + 1]  = make_float2(v[2], v[3]);
        }
        // Z[k1 + 16 k2] = sum_n2 E[n2][k1] W16^(n2 k2)
        asm volatile("" ::"v"(z[0].x), "v"(z[15].y));
        MFCC_R16_STAMP(2);
        r16_dft16(z);
        wave_sync();  // every lane has its row before the buffer takes the natural order
#pragma unroll
        for (int r = 0; r < 16; ++r)
            zb[m + 16 * ((r >> 2) + 4 * (r & 3))] = z[r];
        wave_sync();
        // real split (Math/FastFourierTransform.cc:113-133), 1/fs, amplitude; this lane's bin pairs (i, 256 - i), i = m + 16 j
        MFCC_R16_STAMP(3);
        float* amp = s_amp + f * L.amp_ld;
        float2 sza[8], szb[8];  // all sixteen reads first: the amplitude stores of one pair would hold back the reads of the next
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = m + 16 * j;
            sza[j]      = zb[i];
            szb[j]      = zb[(NC - i) & (NC - 1)];
        }
        const float2 zmid = zb[NC / 2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int    i   = m + 16 * j;
            const float2 za = sza[j], zbb = szb[j];
            const float2 w   = s_stw[i];
            const float  h1r = 0.5f * (za.x + zbb.x);
            const float  h1i = 0.5f * (za.y - zbb.y);
            const float  h2r = 0.5f * (za.y + zbb.y);
            const float  h2i = -0.5f * (za.x - zbb.x);
            float        ar  = fmaf(-w.y, h2i, fmaf(w.x, h2r, h1r));
            float        ai  = fmaf(w.y, h2r, fmaf(w.x, h2i, h1i));
            float        br  = fmaf(w.y, h2i, fmaf(-w.x, h2r, h1r));
            float        bi  = fmaf(w.y, h2r, fmaf(w.x, h2i, -h1i));
            if (j == 0 && m == 0) {  // i = 0: DC and Nyquist bins of the real transform
                ar = za.x + za.y;
                ai = 0.f;
                br = za.x - za.y;
                bi = 0.f;
            }
            if (doscale) {
                ar *= scale;
                ai *= scale;
                br *= scale;
                bi *= scale;
            }
            if (live) {
                if (j == 0 && m == 0) {
                    amp[0]  = fabsf(ar);
                    amp[NC] = fabsf(br);
                }
                else {
                    amp[i]      = __builtin_amdgcn_sqrtf(fmaf(ar, ar, ai * ai));
                    amp[NC - i] = __builtin_amdgcn_sqrtf(fmaf(br, br, bi * bi));
                }
            }
        }
        if (m == 0 && live) {  // bin 128: untouched by the reference's split loop
            float2 zm = zmid;
            if (doscale) {
                zm.x *= scale;
                zm.y *= scale;
            }
            amp[NC / 2] = __builtin_amdgcn_sqrtf(fmaf(zm.x, zm.x, zm.y * zm.y));
        }
      }
      pf_have = false;
      if (tile_id + (int)gridDim.x < p.n_tiles)
          pf_have = fetch_batch(p.tiles[tile_id + gridDim.x]);
    }
    else {
    for (int f = wave; f < FT; f += MW) {
        if (f >= tile.n_frames)
            break;  // wave-uniform
        const long long fbase = (long long)(tile.frame0 + f) * p.frame_shift;  // the frame's first sample within the segment
        wave_sync();  // the previous frame's readers of s_z are done
        // first radix-4 stage (Ns = 1, unit twiddles) straight from the pre-emphasised tile
        {
            float2 y4[P::B4][4];
#pragma unroll
            for (int b = 0; b < P::B4; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 4) {
                    float2 x[4];
                    float  xm[4], x0[4], x1[4];
                    // a frame that lies inside its segment with a predecessor sample (all but the first and the last few of a segment:
                    // wave-uniform) needs none of the per-sample segment guards -- 64-bit compares and selects on twelve loads
                    // (PF: the whole zero-padded span of the transform inside the segment, so that its loads need no index clamp either)
                    const bool inner = fbase >= 1 && fbase + (PF ? 2 * NC : p.frame_len) < nseg;
                    auto       emphasise = [&](int r, float& y0, float& y1) {
                        if (alpha1) {  // Signal/Preemphasis.cc:69-75
                            y0 = x0[r] - xm[r];
                            y1 = x1[r] - x0[r];
                        }
                        else {  // :62-67
                            const float p0 = alpha * xm[r], p1 = alpha * x0[r];
                            y0             = x0[r] - p0;
                            y1             = x1[r] - p1;
                        }
                    };
                    if (inner) {
                        const Sample* fr = seg + fbase;
                        if constexpr (PF) {
                            // Every load of this variant is UNCONDITIONAL and the window mask is applied where the values are used: a load
                            // under a lane mask sits in a conditional block, the compiler cannot count what is in flight behind such a
                            // block and drains the queue (s_waitcnt vmcnt(0)) -- which is what the first form of this variant did right
                            // behind its next-frame loads, in front of the transform they were meant to overlap with (0.795 against 0.743 ms).
                            if (!have) {  // wave-uniform: nothing fetched ahead (first frame of a tile, or behind a guarded frame)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    pn[r] = *(const Raw3*)(fr + 2 * (j + r * (NC / 4)) - 1);
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const bool w = 2 * (j + r * (NC / 4)) < p.frame_len;
                                x0[r]        = w ? (float)pn[r].x0 : 0.f;
                                x1[r]        = w ? (float)pn[r].x1 : 0.f;
                                xm[r]        = w ? (float)pn[r].m : 0.f;
                            }
                            // this wave's next frame of the tile, if it is an inner frame too; otherwise the current frame once more
                            const int     fn = f + MW;
                            const Sample* fq = fr;
                            if (fn < FT && fn < tile.n_frames) {
                                const long long fbn = fbase + (long long)MW * p.frame_shift;
                                have                = fbn + 2 * NC < nseg;
                                fq                  = have ? seg + fbn : fr;
                            }
                            else {  // behind the wave's last frame of this tile: its first frame of the workgroup's next tile
                                const long long fbn = (long long)(ntile.frame0 + wave) * p.frame_shift;
                                have                = has_next && wave < ntile.n_frames && fbn >= 1 && fbn + 2 * NC < (long long)ntile.n_samples;
                                fq                  = have ? (const Sample*)p.pcm + ntile.sample_base + fbn : fr;
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                pn[r] = *(const Raw3*)(fq + 2 * (j + r * (NC / 4)) - 1);
                        }
                        else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int  c = j + r * (NC / 4);
                            const bool w = 2 * c < p.frame_len;
                            x0[r]        = w ? (float)fr[2 * c] : 0.f;
                            x1[r]        = w ? (float)fr[2 * c + 1] : 0.f;   // 2c + 1 <= frame_len: still inside the segment
                            xm[r]        = w ? (float)fr[2 * c - 1] : 0.f;
                        }
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float y0, y1;
                            emphasise(r, y0, y1);
                            x[r] = make_float2(wlo[b][r] * y0, whi[b][r] * y1);  // WindowFunction::work
                        }
                    }
                    else {
                        have = false;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {  // samples 2c - 1, 2c, 2c + 1 of the frame (zero beyond the segment and the window)
                            const int       c = j + r * (NC / 4);
                            const long long n = fbase + 2 * c;
                            const bool      w = 2 * c < p.frame_len;
                            x0[r]             = (w && n < nseg) ? (float)seg[n] : 0.f;
                            x1[r]             = (w && n + 1 < nseg) ? (float)seg[n + 1] : 0.f;
                            xm[r]             = (w && n < nseg) ? (float)seg[n > 0 ? n - 1 : 0] : 0.f;  // segment start: previous_ = x[0]
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int       c = j + r * (NC / 4);
                            const long long n = fbase + 2 * c;
                            float           y0, y1;
                            emphasise(r, y0, y1);
                            y0   = n < nseg ? y0 : 0.f;  // zero padding behind the segment (short last frames)
                            y1   = n + 1 < nseg ? y1 : 0.f;
                            x[r] = make_float2(wlo[b][r] * y0, whi[b][r] * y1);
                        }
                    }
                    if (MF) {
                        // ---- first product: D^T[n2][k1], three real products (A = this lane's points, B = F constants)
                        f32x4 p1 = {0.f, 0.f, 0.f, 0.f}, p2 = p1, p3 = p1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[r].x, f1r[r], p1, 0, 0, 0);
                            p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[r].y, f1i[r], p2, 0, 0, 0);
                            p3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[r].x + x[r].y, f1s[r], p3, 0, 0, 0);
                        }
                        // ---- twiddle W256^(k1 n2) on this lane's four values (n2 = 4 q + r, k1 = m)
                        float br[4], bi[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float dr = p1[r] - p2[r], di = (p3[r] - p1[r]) - p2[r];
                            br[r]          = fmaf(dr, wr[r], -(di * wi[r]));
                            bi[r]          = fmaf(dr, wi[r], di * wr[r]);
                        }
                        // ---- second product: X^T[k2][k1] (A = F constants in the order n2 = 4 q + K-step, B = the twiddled values)
                        f32x4 q1 = {0.f, 0.f, 0.f, 0.f}, q2 = q1, q3 = q1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gr[r], br[r], q1, 0, 0, 0);
                            q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gi[r], bi[r], q2, 0, 0, 0);
                            q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(gs[r], br[r] + bi[r], q3, 0, 0, 0);
                        }
                        // lane (q, m), register r: bin k = k1 + 16 k2 = m + 16 (4 q + r)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            s_z[zpad((lane & 15) + 64 * (lane >> 4) + 16 * r)] = make_float2(q1[r] - q2[r], (q3[r] - q1[r]) - q2[r]);
                        continue;
                    }
                    const float2 a  = make_float2(x[0].x + x[2].x, x[0].y + x[2].y);
                    const float2 bb = make_float2(x[0].x - x[2].x, x[0].y - x[2].y);
                    const float2 c  = make_float2(x[1].x + x[3].x, x[1].y + x[3].y);
                    const float2 d  = make_float2(x[1].x - x[3].x, x[1].y - x[3].y);
                    y4[b][0]        = make_float2(a.x + c.x, a.y + c.y);
                    y4[b][1]        = make_float2(bb.x - d.y, bb.y + d.x);  // (x0-x2) + i(x1-x3)
                    y4[b][2]        = make_float2(a.x - c.x, a.y - c.y);
                    y4[b][3]        = make_float2(bb.x + d.y, bb.y - d.x);  // (x0-x2) - i(x1-x3)
                }
            }
#pragma unroll
            for (int b = 0; b < P::B4; ++b) {
                const int j = lane + 64 * b;
                if (!MF && j < NC / 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        s_z[zpad(4 * j + q)] = y4[b][q];
                }
            }
        }
#pragma unroll
        for (int s = 1; s < (MF ? 0 : P::S4); ++s) {
            const int Ns = 1 << (2 * s);
            wave_sync();
            float2 y4[P::B4][4];
#pragma unroll
            for (int b = 0; b < P::B4; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 4) {
                    const float2 x0 = s_z[zpad(j)];
                    const float2 x1 = cmul(s_z[zpad(j + NC / 4)], tw4[s - 1][b][0]);
                    const float2 x2 = cmul(s_z[zpad(j + 2 * (NC / 4))], tw4[s - 1][b][1]);
                    const float2 x3 = cmul(s_z[zpad(j + 3 * (NC / 4))], tw4[s - 1][b][2]);
                    const float2 a  = make_float2(x0.x + x2.x, x0.y + x2.y);
                    const float2 bb = make_float2(x0.x - x2.x, x0.y - x2.y);
                    const float2 c  = make_float2(x1.x + x3.x, x1.y + x3.y);
                    const float2 d  = make_float2(x1.x - x3.x, x1.y - x3.y);
                    y4[b][0]        = make_float2(a.x + c.x, a.y + c.y);
                    y4[b][1]        = make_float2(bb.x - d.y, bb.y + d.x);
                    y4[b][2]        = make_float2(a.x - c.x, a.y - c.y);
                    y4[b][3]        = make_float2(bb.x + d.y, bb.y - d.x);
                }
            }
            wave_sync();
#pragma unroll
            for (int b = 0; b < P::B4; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 4) {
                    const int k  = j & (Ns - 1);
                    const int j0 = ((j - k) << 2) + k;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        s_z[zpad(j0 + q * Ns)] = y4[b][q];
                }
            }
        }
        if (P::R2 && !MF) {
            constexpr int Ns = NC / 2;
            wave_sync();
            float2 y2[P::B2][2];
#pragma unroll
            for (int b = 0; b < P::B2; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 2) {
                    const float2 x0 = s_z[zpad(j)];
                    const float2 x1 = cmul(s_z[zpad(j + NC / 2)], tw2[b]);
                    y2[b][0]        = make_float2(x0.x + x1.x, x0.y + x1.y);
                    y2[b][1]        = make_float2(x0.x - x1.x, x0.y - x1.y);
                }
            }
            wave_sync();
#pragma unroll
            for (int b = 0; b < P::B2; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 2) {
                    s_z[zpad(j)]      = y2[b][0];
                    s_z[zpad(j + Ns)] = y2[b][1];
                }
            }
        }
        wave_sync();

        // real split (Math/FastFourierTransform.cc:113-133), 1/fs, amplitude; bin pairs (i, NC-i)
        float* amp = s_amp + f * L.amp_ld;
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int i = 1 + lane + 64 * b;
            if (i <= NPAIR) {
                const float2 za  = s_z[zpad(i)];
                const float2 zb  = s_z[zpad(NC - i)];
                const float2 w   = stw[b];
                const float  h1r = 0.5f * (za.x + zb.x);
                const float  h1i = 0.5f * (za.y - zb.y);
                const float  h2r = 0.5f * (za.y + zb.y);
                const float  h2i = -0.5f * (za.x - zb.x);
                float        ar  = fmaf(-w.y, h2i, fmaf(w.x, h2r, h1r));
                float        ai  = fmaf(w.y, h2r, fmaf(w.x, h2i, h1i));
                float        br  = fmaf(w.y, h2i, fmaf(-w.x, h2r, h1r));
                float        bi  = fmaf(w.y, h2r, fmaf(w.x, h2i, -h1i));
                if (doscale) {
                    ar *= scale;
                    ai *= scale;
                    br *= scale;
                    bi *= scale;
                }
                amp[i]      = __builtin_amdgcn_sqrtf(fmaf(ar, ar, ai * ai));
                amp[NC - i] = __builtin_amdgcn_sqrtf(fmaf(br, br, bi * bi));
            }
        }
        if (lane == 0) {
            const float2 z0 = s_z[zpad(0)];
            float        dc = z0.x + z0.y, ny = z0.x - z0.y;
            float2       zm = s_z[zpad(NC / 2)];  // untouched by the reference's split loop
            if (doscale) {
                dc *= scale;
                ny *= scale;
                zm.x *= scale;
                zm.y *= scale;
            }
            amp[0]      = fabsf(dc);
            amp[NC]     = fabsf(ny);
            amp[NC / 2] = __builtin_amdgcn_sqrtf(fmaf(zm.x, zm.x, zm.y * zm.y));
        }
    }
    }  // !R16
    MFCC_STAMP(1);
    __syncthreads();
    MFCC_STAMP(2);

    // ================= phase C: mel filter bank + log10 for the whole tile (Signal/Filterbank.cc:65-71,
    // Flow/SimpleFunction.hh:40-49).  item = filter * FT + frame: the 64 lanes of a wave work on 4
    // neighbouring filters (similar supports, broadcast weights) x 16 frames; f32 sum in ascending bin order.
    // (Dealing the groups of four filters to the waves by load -- widest first to the least loaded wave, so that no wave carries
    // 4 + 12 + 40 trips while another carries 5 + 16 -- was measured too: 0.736-0.750 against 0.730-0.739 ms on one box, not kept; the
    // waves' idle slots at the barrier belong to the CU's other three workgroups already.)
    for (int item = tid; item < p.n_filters * FT; item += MT) {
        const int flt = item / FT, f = item - flt * FT;
        const int b0 = s_fs[flt], b1 = s_fe[flt];
        const float* w   = s_fw + s_fo[flt] - b0;
        const float* amp = s_amp + f * L.amp_ld;
        float        acc = 0.f;
        // The sum is the reference's: f32, ascending bins, one rounding per product and per addition -- a dependent chain.  The PRODUCTS do
        // not depend on it: a trip of four bins fetches the next trip's eight LDS operands before it multiplies and adds the four it fetched a trip earlier, so
        // the LDS latency of a trip hides behind the previous trip's additions (it used to be paid once per four bins, serially: the
        // filter bank was 13-24 % of a tile's time, tools/mfcc_timeline.py).
        // mfplp.flow: generic-vector-f32-power 2 in front of the filter bank ((f32)pow((f64)x, 2.0) = x * x rounded once) -- decided ONCE per
        // item (wave-uniform), not per product: as a run-time select it cost a multiply and a v_cndmask on every bin
        auto bank = [&](auto sq_tag) {
        constexpr bool sq = decltype(sq_tag)::value;
        int            b  = b0;
        auto prod = [&](int i) {
            const float a = amp[i];
            return (sq ? a * a : a) * w[i];
        };
        auto mulw = [&](float a, float wt) { return (sq ? a * a : a) * wt; };
        {
            // two operand sets in turn (trips of four bins): set B is fetched before set A is multiplied and added, and A's next
            // fetch goes out before B is consumed -- written out as a ping-pong, because the compiler renames a rotating single set
            // into "fetch, wait, consume".  A fetch behind the lane's last trip reads LDS it does not use.
            const int n4 = (b1 - b0) >> 2;
            float     A[8], B[8];
            auto fetch = [&](float (&o)[8], int i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e]     = amp[i + e];
                    o[4 + e] = w[i + e];
                }
            };
            auto consume = [&](const float (&o)[8]) {
                const float q0 = mulw(o[0], o[4]), q1 = mulw(o[1], o[5]), q2 = mulw(o[2], o[6]), q3 = mulw(o[3], o[7]);
                acc            = acc + q0;
                acc            = acc + q1;
                acc            = acc + q2;
                acc            = acc + q3;
            };
            int t = 0;
            if (n4 > 0)
                fetch(A, b0);
            for (; t + 2 <= n4; t += 2) {
                fetch(B, b0 + 4 * (t + 1));
                consume(A);
                fetch(A, b0 + 4 * (t + 2));
                consume(B);
            }
            if (t < n4) {
                consume(A);
                ++t;
            }
            b = b0 + 4 * t;
        }
        for (; b < b1; ++b) {
            const float q0 = prod(b);
            acc            = acc + q0;
        }
        };
        if (!R16 && p.front_end)
            bank(std::true_type{});
        else
            bank(std::false_type{});
        if (p.eql)  // plp.flow: in[i] = (f32)((f64)in[i] * f(i)) (Signal/VectorTransform.cc:78-83)
            acc = (float)((double)acc * p.eql[flt]);
        if (f < tile.n_frames)
            s_lm[f * L.lm_ld + flt] = (!R16 && p.front_end) ? power_node(acc, p.plp_power)  // intensity-loudness law (the radix-16 form is built
                                                            : __log10f(acc);  // for mfcc.flow only: the call's registers do not fit beside its batch) // v_log_f32 * log10(2): ~1 ulp of log2
    }
    MFCC_STAMP(3);
    __syncthreads();
    MFCC_STAMP(4);

    // ================= phase D: DCT-II as a [FT x kpad] x [kpad x n_ceps] product on the f32 matrix
    // cores (Signal/CosineTransform.cc:76-83).  v_mfma_f32_16x16x4_f32 is an exact f32 fma chain in
    // ascending n, i.e. the reference's left-to-right accumulation with fused instead of separate rounding.
    {
        const int n_tiles = L.dct_ld / 16;
        const int arow = lane & 15, akq = lane >> 4;
        for (int nt = wave; nt < n_tiles; nt += MW) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // (four K-steps per trip with their eight LDS operands fetched together: measured slower, 0.749 against 0.732 ms on config 2)
            for (int kk = 0; kk < L.kpad / 4; ++kk) {
                const float a = s_lm[arow * L.lm_ld + 4 * kk + akq];
                const float b = s_dct[(4 * kk + akq) * L.dct_ld + nt * 16 + arow];
                acc           = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
            const int cep = nt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = (lane >> 4) * 4 + r;
                if (f < tile.n_frames && cep < p.n_ceps) {
                    float v = acc[r];
                    if (p.dct_normalize)
                        v = v / p.norm_div;
                    p.ceps[(tile.out_frame + f) * (long long)p.n_ceps + cep] = v;
                }
            }
        }
    }
    MFCC_STAMP(5);
    __syncthreads();  // s_lm / s_amp are rewritten by the next tile
    MFCC_STAMP(6);
  }
}

// out[t] = concat(x[clamp(t-left)], ..., x[clamp(t+right)]) per segment
__global__ __launch_bounds__(256) void context_window_kernel(const float* __restrict__ feats, const long long* __restrict__ frame_off,
                                                            int n_seg, int dim, int left, int right,
                                                            float* __restrict__ out, int out_stride, long long total) {
    const long long t = blockIdx.x;
    if (t >= total)
        return;
    // binary search of the segment containing frame t
    int lo = 0, hi = n_seg;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (frame_off[mid] <= t)
            lo = mid;
        else
            hi = mid;
    }
    const long long s0 = frame_off[lo], s1 = frame_off[lo + 1];
    const int       w  = (left + right + 1) * dim;
    for (int i = threadIdx.x; i < out_stride; i += blockDim.x) {
        float v = 0.f;
        if (i < w) {
            int       c  = i / dim, d = i - c * dim;
            long long tt = t - left + c;
            tt           = tt < s0 ? s0 : (tt >= s1 ? s1 - 1 : tt);
            v            = feats[tt * dim + d];
        }
        out[t * (long long)out_stride + i] = v;
    }
}

}  // namespace amx

// ------------------------------------------------------------------------------------ ABI

struct amx_mfcc {
    amx_ctx*        ctx = nullptr;
    bool            tune_fft_mfma = false, tune_lpc_lds = false, tune_prefetch = false;  // amx_mfcc_cfg.tuning
    bool            fft_r16 = false;  // the radix-16 form of the 512-point transform (tuning fft=r16; LDS sized for it)
    int             tune_wgs = 0;
    amx::MfccTables tab;
    int             frames_per_tile = 16;
    // device copies of the tables
    float * d_window = nullptr, *d_fw = nullptr, *d_dct_t = nullptr;
    int *   d_fs = nullptr, *d_fe = nullptr, *d_fo = nullptr;
    float2 *d_tw = nullptr, *d_stw = nullptr;
    double* d_eql = nullptr;
    size_t  lds_bytes = 0;
    float*  d_ac   = nullptr;  // MF-PLP: autocorrelation coefficients of the current call [frames x n_transform]
    size_t  ac_cap = 0;
};

struct amx_mfcc_plan {
    amx_mfcc*              owner = nullptr;
    int                    n_seg = 0;
    std::vector<long>      sample_off, frame_off;
    std::vector<amx::MfccTile> tiles;
    amx::MfccTile*         d_tiles     = nullptr;
    long long*             d_frame_off = nullptr;
};

namespace {

template<class T>
int upload(T** dst, const T* src, size_t n) {
    AMX_HIP(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n)
        AMX_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return AMX_OK;
}

size_t mfcc_lds_bytes(const amx::MfccTables& t, bool r16) {
    amx::MfccLds L(t.frame_len, t.frame_shift, t.fft_len, t.n_inputs, t.n_transform, (int)t.filter_weights.size(), r16);
    return (size_t)L.total * 4;
}

// MF-PLP tail, one lane per frame: autocorrelation -> Levinson recursion in f64 (Math/LevinsonLse.cc:35-70; the first
// reflection coefficient is an f32 division like the reference's expression) -> gain, a1..aN as f32 -> LPC cepstrum
// (Signal/AutoregressionToCepstrum.cc:21-35: 2 log(gain) in f64, the recursion in f32 left to right).  The per-lane arrays
// live in LDS as [index][lane] (every lane touches the same index at the same time: conflict free); a failed recursion
// (zero prediction error) writes NaNs.
__host__ __device__ constexpr size_t lpc_lds_bytes(int n_ac) {
    return (size_t)64 * n_ac * (8 + 8 + 4 + 4 + 4);
}
__global__ __launch_bounds__(64) void lpc_cepstrum_kernel(const float* __restrict__ ac, int n_ac, float* __restrict__ out, int n_out,
                                                         long long n_frames) {
    extern __shared__ __attribute__((aligned(16))) char lpc_lds[];
    const int lane  = threadIdx.x;
    double*   prev  = (double*)lpc_lds + lane;            // [n_ac][64]
    double*   cur   = prev + (size_t)64 * n_ac;
    float*    R     = (float*)(cur - lane + (size_t)64 * n_ac) + lane;
    float*    a     = R + (size_t)64 * n_ac;
    float*    c     = a + (size_t)64 * n_ac;
    const long long t  = (long long)blockIdx.x * 64 + lane;
    const long long tt = t < n_frames ? t : n_frames - 1;
    for (int i = 0; i < n_ac; ++i)
        R[i * 64] = ac[tt * n_ac + i];
    const int N = n_ac - 1;
    auto almost_zero = [](double e) {  // Core::isAlmostEqual(e, 0.0)
        return fabs(e) < (fabs(e) + 0.0 + 2.2250738585072014e-308) * 2.2204460492503131e-16 * 1.0;
    };
    bool   ok = true;
    double E  = (double)R[0];
    if (almost_zero(E))
        ok = false;
    if (ok) {
        const double k1 = (double)(-R[64] / R[0]);
        prev[64]        = k1;
        E               = (double)R[0] + (double)R[64] * k1;
    }
    for (int i = 2; i <= N; ++i) {  // the trip count is uniform; lanes whose recursion failed idle through it
        double k = (double)R[i * 64];
        for (int j = 1; j <= i - 1; ++j)
            k += prev[j * 64] * (double)R[(i - j) * 64];
        if (ok && almost_zero(E))
            ok = false;
        if (ok) {
            k           = -k / E;
            cur[i * 64] = k;
            for (int j = 1; j <= i - 1; ++j)
                cur[j * 64] = prev[j * 64] + k * prev[(i - j) * 64];
            E = (1.0 - k * k) * E;
            for (int j = 1; j <= i; ++j)
                prev[j * 64] = cur[j * 64];
        }
    }
    if (t >= n_frames)
        return;
    float* o = out + t * n_out;
    if (!ok) {
        for (int n = 0; n < n_out; ++n)
            o[n] = __builtin_nanf("");
        return;
    }
    const float gain = (float)sqrt(E);
    for (int j = 1; j <= N; ++j)
        a[(j - 1) * 64] = (float)prev[j * 64];
    c[0]  = (float)(2 * log((double)gain));
    c[64] = -a[0];
    for (int n = 2; n < n_out; ++n) {
        float v = (float)n * a[(n - 1) * 64];
        for (int k = 1; k < n; ++k) {
            float tt2 = (float)(n - k) * c[(n - k) * 64];
            tt2       = tt2 * a[(k - 1) * 64];
            v         = v + tt2;
        }
        c[n * 64] = v / (-(float)n);
    }
    for (int n = 0; n < n_out; ++n)
        o[n] = c[n * 64];
}

// The same recursions with every array in registers (loops unrolled to MAXA coefficients, steps beyond n_ac skipped by uniform
// branches): the LDS version above holds one wave per SIMD -- 36 KB of LDS per wave at 20 coefficients -- and pays an LDS round trip
// per array access; this one runs the O(n_ac^2) f64 work from VGPRs (mfplp.flow: 0.47 -> see DESIGN 4.1).  Identical operations in
// identical order.
template<int MAXA>
__global__ __launch_bounds__(64) void lpc_cepstrum_reg_kernel(const float* __restrict__ ac, int n_ac, float* __restrict__ out, int n_out,
                                                             long long n_frames) {
    const long long t  = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long tt = t < n_frames ? t : n_frames - 1;
    float           R[MAXA];
    double          prev[MAXA], cur[MAXA];
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
        R[i]    = i < n_ac ? ac[tt * n_ac + i] : 0.f;
        prev[i] = 0.0;
        cur[i]  = 0.0;
    }
    const int N = n_ac - 1;
    auto almost_zero = [](double e) {  // Core::isAlmostEqual(e, 0.0)
        return fabs(e) < (fabs(e) + 0.0 + 2.2250738585072014e-308) * 2.2204460492503131e-16 * 1.0;
    };
    bool   ok = true;
    double E  = (double)R[0];
    if (almost_zero(E))
        ok = false;
    if (ok) {
        const double k1 = (double)(-R[1] / R[0]);
        prev[1]         = k1;
        E               = (double)R[0] + (double)R[1] * k1;
    }
#pragma unroll
    for (int i = 2; i < MAXA; ++i) {
        if (i <= N) {  // uniform
            double k = (double)R[i];
#pragma unroll
            for (int j = 1; j <= i - 1; ++j)
                k += prev[j] * (double)R[i - j];
            if (ok && almost_zero(E))
                ok = false;
            if (ok) {
                k      = -k / E;
                cur[i] = k;
#pragma unroll
                for (int j = 1; j <= i - 1; ++j)
                    cur[j] = prev[j] + k * prev[i - j];
                E = (1.0 - k * k) * E;
#pragma unroll
                for (int j = 1; j <= i; ++j)
                    prev[j] = cur[j];
            }
        }
    }
    if (t >= n_frames)
        return;
    float* o = out + t * n_out;
    if (!ok) {
        for (int n = 0; n < n_out; ++n)
            o[n] = __builtin_nanf("");
        return;
    }
    const float gain = (float)sqrt(E);
    float       a[MAXA], c[MAXA];
#pragma unroll
    for (int j = 1; j < MAXA; ++j)
        a[j - 1] = (float)prev[j];
    a[MAXA - 1] = 0.f;
    c[0] = (float)(2 * log((double)gain));
    c[1] = -a[0];
#pragma unroll
    for (int n = 2; n < MAXA; ++n) {
        c[n] = 0.f;
        if (n < n_out) {
            float v = (float)n * a[n - 1];
#pragma unroll
            for (int k = 1; k < n; ++k) {
                float tt2 = (float)(n - k) * c[n - k];
                tt2       = tt2 * a[k - 1];
                v         = v + tt2;
            }
            c[n] = v / (-(float)n);
        }
    }
#pragma unroll
    for (int n = 0; n < MAXA; ++n)
        if (n < n_out)
            o[n] = c[n];
}

template<int NC, int VAR>
int launch_mfcc_var(amx_mfcc* h, const amx::MfccParams& p, int n_tiles) {
    auto kern = amx::mfcc_kernel<NC, VAR>;
    if (h->lds_bytes > 48 * 1024)
        AMX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    // persistent workgroups: as many as are co-resident (LDS bound), each loops over tiles
    // (at most four per CU: five were 30 % slower on the PLP front ends, whose LDS footprint would allow them)
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>((VAR & 16) ? 2 : (VAR & 8) ? 3 : 4, (160 * 1024) / std::max<size_t>(h->lds_bytes, 1)));
    if (h->tune_wgs > 0)  // A/B runs (tuning wgs=N): cap the workgroups per CU
        per_cu = std::max(1, std::min(per_cu, h->tune_wgs));
    int grid   = std::min(n_tiles, per_cu * std::max(h->ctx->n_cu, 1));
    amx::ScopedKernelTimer timer(h->ctx, "mfcc");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(amx::mfcc_waves(NC) * 64), h->lds_bytes, h->ctx->stream, p);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

template<int NC>
int launch_mfcc(amx_mfcc* h, const amx::MfccParams& p, int n_tiles, bool s16) {
    if (n_tiles <= 0)
        return AMX_OK;
    // amx_mfcc_cfg.tuning fft=mfma: the 512-point transform as two 16x16x16 complex products on the f32 matrix cores (see mfcc_kernel).  Measured
    // on config 2 (993 k frames, same box): 0.958 ms against 0.748 ms for the radix-4 LDS stages.  SQ counters (profiles/r03/pmc/
    // mfcc_fft_*): matrix pipe busy 36 % + vector ALUs busy 51 % = 87 % of the dispatch -- v_mfma_f32_16x16x4_f32 runs at the f32
    // vector rate and does not overlap with vector instructions on a SIMD, so 24 of them cost like 192 vector instructions, more
    // than the ~125 they replace.  The butterflies stay the default; the product form is kept for A/B runs.
    const bool mfma = h->tune_fft_mfma;
    if constexpr (NC == 256) {
        if (h->fft_r16)
            return s16 ? launch_mfcc_var<NC, 18>(h, p, n_tiles) : launch_mfcc_var<NC, 16>(h, p, n_tiles);
    }
    if constexpr (NC == 256) {
        if (mfma)
            return s16 ? launch_mfcc_var<NC, 3>(h, p, n_tiles) : launch_mfcc_var<NC, 1>(h, p, n_tiles);
    }
    if constexpr (NC == 256) {
        if (h->tune_prefetch)
            return s16 ? launch_mfcc_var<NC, 6>(h, p, n_tiles) : launch_mfcc_var<NC, 4>(h, p, n_tiles);
    }
    return s16 ? launch_mfcc_var<NC, 2>(h, p, n_tiles) : launch_mfcc_var<NC, 0>(h, p, n_tiles);
}

}  // namespace

extern "C" {

void amx_mfcc_default_cfg(amx_mfcc_cfg* c) {
    if (!c)
        return;
    c->sample_rate            = 16000.0;
    c->win_len_s              = 0.025;
    c->win_shift_s            = 0.01;
    c->preemph_alpha          = 1.0;
    c->fft_max_input_s        = 0.025;
    c->apply_scale            = 1;
    c->mel_filter_width       = 268.258;
    c->mel_spacing            = 0.0;
    c->warp_differential_unit = 1;
    c->n_ceps                 = 16;
    c->dct_normalize          = 0;
    c->front_end              = AMX_FRONT_END_MFCC;
    c->n_autocorrelation      = 0;
    c->plp_power              = 0.33;
    c->filter_type            = AMX_FILTER_TRIANGULAR;
    c->boundary               = AMX_BOUNDARY_STRETCH_TO_COVER;
    c->warping                = AMX_WARP_MEL;
    c->tuning                 = nullptr;
}

void amx_plp_default_cfg(amx_mfcc_cfg* c) {
    if (!c)
        return;
    amx_mfcc_default_cfg(c);
    c->win_len_s         = 0.02;  // plp.flow: window length 0.02, maximum-input-size 0.02, no signal-preemphasis node
    c->fft_max_input_s   = 0.02;
    c->preemph_alpha     = 0.0;
    c->mel_filter_width  = 3.8;
    c->mel_spacing       = 0.93853;
    c->filter_type       = AMX_FILTER_TRAPEZE;
    c->boundary          = AMX_BOUNDARY_INCLUDE;
    c->warping           = AMX_WARP_BARK;
    c->front_end         = AMX_FRONT_END_PLP;
    c->dct_normalize     = 1;
    c->n_autocorrelation = 13;
    c->n_ceps            = 13;
}

void amx_mfplp_default_cfg(amx_mfcc_cfg* c) {
    if (!c)
        return;
    amx_mfcc_default_cfg(c);  // mfplp.flow: same window, FFT and mel filter bank (filter-width 268.258)
    c->front_end         = AMX_FRONT_END_MFPLP;
    c->dct_normalize     = 1;   // normalize="true"
    c->n_autocorrelation = 13;  // LPC order 12
    c->n_ceps            = 13;
}

int amx_mfcc_create(amx_ctx* ctx, const amx_mfcc_cfg* cfg, amx_mfcc** out) {
    // ctx == NULL creates a host-only handle: geometry and tables are available
    // (amx_mfcc_describe / _n_frames / _tables), running it returns AMX_ERR_STATE.
    AMX_REQUIRE(cfg && out, AMX_ERR_INVALID, "amx_mfcc_create: NULL argument");
    *out        = nullptr;
    amx::Tuning tune;
    {
        static const char* const keys[] = {"fft", "wgs", "lpc", "prefetch", "contract", nullptr};
        if (!tune.parse(cfg->tuning, keys, "amx_mfcc_create"))
            return AMX_ERR_INVALID;
    }
    std::string t_fft, t_lpc, t_contract;
    int         t_wgs, t_prefetch;
    {
        static const char* const ffts[] = {"stockham", "mfma", "r16", nullptr};
        static const char* const lpcs[] = {"regs", "lds", nullptr};
        static const char* const cons[] = {"off", "fma", nullptr};
        const char*              who    = "amx_mfcc_create";
        // contract: which build of the reference the TABLES follow bit for bit (filter-bank geometry, f64: amx_set_contract); the
        // kernel's f32 chain is compared at 1e-4 against either build (its log10 / hypot are the device's)
        if (!tune.get_word("fft", "stockham", ffts, &t_fft, who) || !tune.get_word("lpc", "regs", lpcs, &t_lpc, who) ||
            !tune.get_int("wgs", 0, 0, 64, &t_wgs, who) || !tune.get_int("prefetch", 1, 0, 1, &t_prefetch, who) ||
            !tune.get_word("contract", ctx && ctx->contract == AMX_CONTRACT_FMA ? "fma" : "off", cons, &t_contract, who))
            return AMX_ERR_INVALID;
    }
    amx_mfcc* h = new amx_mfcc;
    h->ctx      = ctx;
    h->tune_fft_mfma = t_fft == "mfma";
    h->tune_lpc_lds  = t_lpc == "lds";
    h->tune_wgs      = t_wgs;
    h->tune_prefetch = t_prefetch != 0;  // default since the end of round 4 (0.753 -> 0.73 ms on config 2)
    int r       = h->tab.build(*cfg, t_contract == "fma");
    if (r != AMX_OK) {
        delete h;
        return r;
    }
    const amx::MfccTables& t = h->tab;
    if (t.fft_len < 8 || t.fft_len > 2048) {
        amx::set_error("amx_mfcc_create: FFT length %d not supported by the gfx950 kernel (8..2048)", t.fft_len);
        delete h;
        return AMX_ERR_UNSUPPORTED;
    }
    if (!ctx) {
        *out = h;
        return AMX_OK;
    }
    AMX_HIP(hipSetDevice(ctx->device));
    h->frames_per_tile = amx::FT;
    h->fft_r16         = t_fft == "r16" && t.fft_len == 512 && cfg->front_end == AMX_FRONT_END_MFCC;  // (other lengths and front ends: the Stockham stages)
    h->lds_bytes       = mfcc_lds_bytes(t, h->fft_r16);
    if (h->lds_bytes > 160 * 1024) {
        amx::set_error("amx_mfcc_create: configuration needs %zu bytes of LDS per workgroup (> 160 KiB)", h->lds_bytes);
        delete h;
        return AMX_ERR_UNSUPPORTED;
    }
    std::vector<float> dct_t((size_t)t.n_inputs * t.n_transform);
    for (int k = 0; k < t.n_transform; ++k)
        for (int n = 0; n < t.n_inputs; ++n)
            dct_t[(size_t)n * t.n_transform + k] = t.dct[(size_t)k * t.n_inputs + n];
    // the kernel walks the cosine-transform inputs: plp.flow's copies of the first and last filter output are two more entries
    // that point at the same weights (generic-vector-f32-split port 0 / reversed port 0 + generic-vector-f32-concat)
    std::vector<int> in_start((size_t)t.n_inputs), in_end((size_t)t.n_inputs), in_off((size_t)t.n_inputs);
    for (int i = 0; i < t.n_inputs; ++i) {
        const int src = t.n_inputs == t.n_filters ? i : std::min(std::max(i - 1, 0), t.n_filters - 1);
        in_start[(size_t)i] = t.filter_start[(size_t)src];
        in_end[(size_t)i]   = t.filter_end[(size_t)src];
        in_off[(size_t)i]   = t.filter_offset[(size_t)src];
    }
    if ((r = upload(&h->d_window, t.window.data(), t.window.size())) != AMX_OK ||
        (r = upload(&h->d_fw, t.filter_weights.data(), t.filter_weights.size())) != AMX_OK ||
        (r = upload(&h->d_dct_t, dct_t.data(), dct_t.size())) != AMX_OK ||
        (r = upload(&h->d_fs, in_start.data(), in_start.size())) != AMX_OK ||
        (r = upload(&h->d_fe, in_end.data(), in_end.size())) != AMX_OK ||
        (r = upload(&h->d_fo, in_off.data(), in_off.size())) != AMX_OK ||
        (!t.eql.empty() && (r = upload(&h->d_eql, t.eql.data(), t.eql.size())) != AMX_OK) ||
        (r = upload(&h->d_tw, (const float2*)t.twiddle.data(), t.twiddle.size() / 2)) != AMX_OK ||
        (r = upload(&h->d_stw, (const float2*)t.split_twiddle.data(), t.split_twiddle.size() / 2)) != AMX_OK) {
        amx_mfcc_destroy(h);
        return r;
    }
    *out = h;
    return AMX_OK;
}

void amx_mfcc_destroy(amx_mfcc* h) {
    if (!h)
        return;
    if (!h->ctx) {
        delete h;
        return;
    }
    hipSetDevice(h->ctx->device);
    hipFree(h->d_window);
    hipFree(h->d_fw);
    hipFree(h->d_dct_t);
    hipFree(h->d_fs);
    hipFree(h->d_fe);
    hipFree(h->d_fo);
    hipFree(h->d_tw);
    hipFree(h->d_stw);
    hipFree(h->d_eql);
    hipFree(h->d_ac);
    delete h;
}

int amx_mfcc_describe(const amx_mfcc* h, amx_mfcc_info* info) {
    AMX_REQUIRE(h && info, AMX_ERR_INVALID, "amx_mfcc_describe: NULL argument");
    info->frame_len              = h->tab.frame_len;
    info->frame_shift            = h->tab.frame_shift;
    info->fft_len                = h->tab.fft_len;
    info->n_bins                 = h->tab.n_bins;
    info->n_filters              = h->tab.n_filters;
    info->n_ceps                 = h->tab.n_ceps;
    info->fft_output_sample_rate = h->tab.fft_output_sample_rate;
    info->mel_max                = h->tab.mel_max;
    info->n_transform            = h->tab.n_transform;
    info->n_transform_inputs     = h->tab.n_inputs;
    return AMX_OK;
}

long amx_mfcc_n_frames(const amx_mfcc* h, long n_samples) {
    return h ? h->tab.n_frames(n_samples) : 0;
}

double amx_mfcc_frame_start_time(const amx_mfcc* h, long frame) {
    return h ? h->tab.frame_start_time(frame) : 0.0;
}

int amx_mfcc_tables(const amx_mfcc* h, float* window, int* fs, int* fe, int* fo, float* fw, float* dct) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_mfcc_tables: NULL handle");
    const amx::MfccTables& t = h->tab;
    if (window)
        memcpy(window, t.window.data(), t.window.size() * 4);
    if (fs)
        memcpy(fs, t.filter_start.data(), t.filter_start.size() * 4);
    if (fe)
        memcpy(fe, t.filter_end.data(), t.filter_end.size() * 4);
    if (fo)
        memcpy(fo, t.filter_offset.data(), t.filter_offset.size() * 4);
    if (fw)
        memcpy(fw, t.filter_weights.data(), t.filter_weights.size() * 4);
    if (dct)
        memcpy(dct, t.dct.data(), t.dct.size() * 4);
    return AMX_OK;
}

int amx_mfcc_equal_loudness(const amx_mfcc* h, double* factors) {
    AMX_REQUIRE(h && factors, AMX_ERR_INVALID, "amx_mfcc_equal_loudness: NULL argument");
    AMX_REQUIRE(!h->tab.eql.empty(), AMX_ERR_STATE, "amx_mfcc_equal_loudness: not a plp.flow front end");
    memcpy(factors, h->tab.eql.data(), h->tab.eql.size() * sizeof(double));
    return AMX_OK;
}

int amx_mfcc_plan_create(amx_mfcc* h, int n_seg, const long* sample_offsets, amx_mfcc_plan** out) {
    AMX_REQUIRE(h && out && n_seg >= 0 && (n_seg == 0 || sample_offsets), AMX_ERR_INVALID, "amx_mfcc_plan_create: bad argument");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_mfcc_plan_create: host-only handle (created without a context)");
    *out             = nullptr;
    amx_mfcc_plan* p = new amx_mfcc_plan;
    p->owner         = h;
    p->n_seg         = n_seg;
    p->sample_off.assign(sample_offsets, sample_offsets + n_seg + (n_seg ? 1 : 0));
    if (n_seg == 0)
        p->sample_off.assign(1, 0);
    p->frame_off.assign((size_t)n_seg + 1, 0);
    const int ft = h->frames_per_tile;
    for (int u = 0; u < n_seg; ++u) {
        long len = p->sample_off[u + 1] - p->sample_off[u];
        if (len < 0 || len > 0x7fffffffL) {
            amx::set_error("amx_mfcc_plan_create: segment %d has invalid length %ld", u, len);
            delete p;
            return AMX_ERR_INVALID;
        }
        long T              = h->tab.n_frames(len);
        p->frame_off[u + 1] = p->frame_off[u] + T;
        for (long f0 = 0; f0 < T; f0 += ft) {
            amx::MfccTile t;
            t.sample_base = p->sample_off[u];
            t.out_frame   = p->frame_off[u] + f0;
            t.n_samples   = (int)len;
            t.frame0      = (int)f0;
            t.n_frames    = (int)std::min<long>(ft, T - f0);
            t.pad_        = 0;
            p->tiles.push_back(t);
        }
    }
    hipSetDevice(h->ctx->device);
    int r = upload(&p->d_tiles, p->tiles.data(), p->tiles.size());
    if (r == AMX_OK) {
        std::vector<long long> fo(p->frame_off.begin(), p->frame_off.end());
        r = upload(&p->d_frame_off, fo.data(), fo.size());
    }
    if (r != AMX_OK) {
        amx_mfcc_plan_destroy(p);
        return r;
    }
    *out = p;
    return AMX_OK;
}

void amx_mfcc_plan_destroy(amx_mfcc_plan* p) {
    if (!p)
        return;
    hipFree(p->d_tiles);
    hipFree(p->d_frame_off);
    delete p;
}

long amx_mfcc_plan_total_frames(const amx_mfcc_plan* p) {
    return p ? p->frame_off.back() : 0;
}

int amx_mfcc_plan_frame_offsets(const amx_mfcc_plan* p, long* frame_offsets) {
    AMX_REQUIRE(p && frame_offsets, AMX_ERR_INVALID, "amx_mfcc_plan_frame_offsets: NULL argument");
    std::copy(p->frame_off.begin(), p->frame_off.end(), frame_offsets);
    return AMX_OK;
}

static int mfcc_run_plan(amx_mfcc* h, const amx_mfcc_plan* p, const void* pcm_dev, bool s16, float* ceps_dev);

int amx_mfcc_run_plan_dev(amx_mfcc* h, const amx_mfcc_plan* p, const float* pcm_dev, float* ceps_dev) {
    return mfcc_run_plan(h, p, pcm_dev, false, ceps_dev);
}

int amx_mfcc_run_plan_dev_s16(amx_mfcc* h, const amx_mfcc_plan* p, const int16_t* pcm_dev, float* ceps_dev) {
    return mfcc_run_plan(h, p, pcm_dev, true, ceps_dev);
}

static int mfcc_run_plan(amx_mfcc* h, const amx_mfcc_plan* p, const void* pcm_dev, bool s16, float* ceps_dev) {
    AMX_REQUIRE(h && p, AMX_ERR_INVALID, "amx_mfcc_run_plan_dev: NULL handle");
    AMX_REQUIRE(p->owner == h, AMX_ERR_STATE, "amx_mfcc_run_plan_dev: plan belongs to another front-end handle");
    if (p->tiles.empty())
        return AMX_OK;
    AMX_REQUIRE(pcm_dev && ceps_dev, AMX_ERR_INVALID, "amx_mfcc_run_plan_dev: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    const amx::MfccTables& t = h->tab;
    amx::MfccParams        k;
    k.pcm             = pcm_dev;
    k.ceps            = ceps_dev;
    k.tiles           = p->d_tiles;
    k.window          = h->d_window;
    k.fstart          = h->d_fs;
    k.fend            = h->d_fe;
    k.foff            = h->d_fo;
    k.fweights        = h->d_fw;
    k.dct_t           = h->d_dct_t;
    k.tw              = h->d_tw;
    k.stw             = h->d_stw;
    k.frame_len       = t.frame_len;
    k.frame_shift     = t.frame_shift;
    k.n_filters       = t.n_inputs;
    k.eql             = h->d_eql;
    k.n_ceps          = t.n_transform;
    k.n_weights       = (int)t.filter_weights.size();
    k.front_end       = t.cfg.front_end != AMX_FRONT_END_MFCC ? 1 : 0;
    k.norm_div        = t.norm_div;
    k.plp_power       = (float)t.cfg.plp_power;
    const long long total_frames = p->frame_off.back();
    if (k.front_end) {  // the fused kernel stops at the autocorrelation coefficients; lpc_cepstrum_kernel finishes the chain
        const size_t need = (size_t)total_frames * t.n_transform;
        if (need > h->ac_cap) {
            hipFree(h->d_ac);
            h->d_ac   = nullptr;
            h->ac_cap = 0;
            AMX_HIP(hipMalloc((void**)&h->d_ac, std::max<size_t>(need, 1) * 4));
            h->ac_cap = need;
        }
        k.ceps = h->d_ac;
    }
    k.frames_per_tile = h->frames_per_tile;
    k.alpha           = (float)t.cfg.preemph_alpha;
    k.fft_scale       = t.fft_scale;
    const int n_tiles_total = (int)p->tiles.size();
    k.n_tiles               = n_tiles_total;
    k.apply_scale     = (t.cfg.apply_scale && t.cfg.sample_rate != 1) ? 1 : 0;
    k.dct_normalize   = t.cfg.dct_normalize;
    int r;
    switch (t.fft_len / 2) {
        case 4: r = launch_mfcc<4>(h, k, n_tiles_total, s16); break;
        case 8: r = launch_mfcc<8>(h, k, n_tiles_total, s16); break;
        case 16: r = launch_mfcc<16>(h, k, n_tiles_total, s16); break;
        case 32: r = launch_mfcc<32>(h, k, n_tiles_total, s16); break;
        case 64: r = launch_mfcc<64>(h, k, n_tiles_total, s16); break;
        case 128: r = launch_mfcc<128>(h, k, n_tiles_total, s16); break;
        case 256: r = launch_mfcc<256>(h, k, n_tiles_total, s16); break;
        case 512: r = launch_mfcc<512>(h, k, n_tiles_total, s16); break;
        case 1024: r = launch_mfcc<1024>(h, k, n_tiles_total, s16); break;
        default:
            amx::set_error("amx_mfcc_run_plan_dev: no kernel for FFT length %d", t.fft_len);
            return AMX_ERR_UNSUPPORTED;
    }
    if (r != AMX_OK || !k.front_end || total_frames == 0)
        return r;
    amx::ScopedKernelTimer timer(h->ctx, "lpc_cepstrum");
    const dim3 lgrid((unsigned)((total_frames + 63) / 64));
    const bool in_regs = !h->tune_lpc_lds;  // tuning lpc=lds: the LDS kernel (A/B runs, tests)
    if (in_regs && t.n_transform <= 16)
        hipLaunchKernelGGL(lpc_cepstrum_reg_kernel<16>, lgrid, dim3(64), 0, h->ctx->stream, h->d_ac, t.n_transform, ceps_dev, t.n_ceps, total_frames);
    else if (in_regs && t.n_transform <= 24)
        hipLaunchKernelGGL(lpc_cepstrum_reg_kernel<24>, lgrid, dim3(64), 0, h->ctx->stream, h->d_ac, t.n_transform, ceps_dev, t.n_ceps, total_frames);
    else {
        const size_t lpc_lds = lpc_lds_bytes(t.n_transform);
        hipFuncSetAttribute((const void*)lpc_cepstrum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lpc_lds);
        hipLaunchKernelGGL(lpc_cepstrum_kernel, lgrid, dim3(64), lpc_lds, h->ctx->stream, h->d_ac, t.n_transform, ceps_dev, t.n_ceps,
                           total_frames);
    }
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

static int mfcc_run_batch(amx_mfcc* h, int n_seg, const void* const* pcm_host, bool s16, const long* n_samples, float* const* ceps_host) {
    AMX_REQUIRE(h && n_seg >= 0, AMX_ERR_INVALID, "amx_mfcc_run_batch: bad argument");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_mfcc_run_batch: host-only handle (created without a context)");
    if (n_seg == 0)
        return AMX_OK;
    AMX_REQUIRE(pcm_host && n_samples && ceps_host, AMX_ERR_INVALID, "amx_mfcc_run_batch: NULL argument");
    const size_t ss = s16 ? 2 : 4;  // bytes per sample on the host link and in HBM
    std::vector<long> off((size_t)n_seg + 1, 0);
    for (int u = 0; u < n_seg; ++u) {
        AMX_REQUIRE(n_samples[u] >= 0, AMX_ERR_INVALID, "amx_mfcc_run_batch: negative segment length");
        off[u + 1] = off[u] + n_samples[u];
    }
    amx_mfcc_plan* plan = nullptr;
    int            r    = amx_mfcc_plan_create(h, n_seg, off.data(), &plan);
    if (r != AMX_OK)
        return r;
    const long total_frames = amx_mfcc_plan_total_frames(plan);
    char*      d_pcm  = nullptr;
    float*     d_ceps = nullptr;
    hipStream_t st = h->ctx->stream;
    auto fail = [&](int code) {
        hipFree(d_pcm);
        hipFree(d_ceps);
        amx_mfcc_plan_destroy(plan);
        return code;
    };
    if (hipMalloc((void**)&d_pcm, std::max<long>(off[n_seg], 1) * ss) != hipSuccess ||
        hipMalloc((void**)&d_ceps, std::max<long>(total_frames * h->tab.n_ceps, 1) * 4) != hipSuccess) {
        amx::set_error("amx_mfcc_run_batch: out of device memory");
        return fail(AMX_ERR_DEVICE);
    }
    for (int u = 0; u < n_seg; ++u)
        if (n_samples[u] > 0 &&
            hipMemcpyAsync(d_pcm + off[u] * ss, pcm_host[u], (size_t)n_samples[u] * ss, hipMemcpyHostToDevice, st) != hipSuccess) {
            amx::set_error("amx_mfcc_run_batch: H2D copy failed");
            return fail(AMX_ERR_DEVICE);
        }
    r = mfcc_run_plan(h, plan, d_pcm, s16, d_ceps);
    if (r != AMX_OK)
        return fail(r);
    for (int u = 0; u < n_seg; ++u) {
        long T = plan->frame_off[u + 1] - plan->frame_off[u];
        if (T > 0 && hipMemcpyAsync(ceps_host[u], d_ceps + plan->frame_off[u] * h->tab.n_ceps, (size_t)T * h->tab.n_ceps * 4,
                                    hipMemcpyDeviceToHost, st) != hipSuccess) {
            amx::set_error("amx_mfcc_run_batch: D2H copy failed");
            return fail(AMX_ERR_DEVICE);
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess) {
        amx::set_error("amx_mfcc_run_batch: kernel execution failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(AMX_ERR_DEVICE);
    }
    return fail(AMX_OK);
}

int amx_mfcc_run_batch(amx_mfcc* h, int n_seg, const float* const* pcm_host, const long* n_samples, float* const* ceps_host) {
    return mfcc_run_batch(h, n_seg, (const void* const*)pcm_host, false, n_samples, ceps_host);
}

int amx_mfcc_run_batch_s16(amx_mfcc* h, int n_seg, const int16_t* const* pcm_host, const long* n_samples, float* const* ceps_host) {
    return mfcc_run_batch(h, n_seg, (const void* const*)pcm_host, true, n_samples, ceps_host);
}

int amx_mfcc_run(amx_mfcc* h, const float* pcm_host, long n_samples, float* ceps_host) {
    const float* in[1]  = {pcm_host};
    float*       out[1] = {ceps_host};
    long         n[1]   = {n_samples};
    return amx_mfcc_run_batch(h, 1, in, n, out);
}

int amx_mfcc_run_s16(amx_mfcc* h, const int16_t* pcm_host, long n_samples, float* ceps_host) {
    const int16_t* in[1]  = {pcm_host};
    float*         out[1] = {ceps_host};
    long           n[1]   = {n_samples};
    return amx_mfcc_run_batch_s16(h, 1, in, n, out);
}

// internal (not in amx.h): the segmentation of a plan for the back-end kernels in backend.hip
int amx_internal_plan_view(const amx_mfcc_plan* p, const long long** d_frame_off, int* n_seg, long long* total) {
    AMX_REQUIRE(p && d_frame_off && n_seg && total, AMX_ERR_INVALID, "plan view: NULL argument");
    *d_frame_off = p->d_frame_off;
    *n_seg       = p->n_seg;
    *total       = p->frame_off.back();
    return AMX_OK;
}

int amx_context_window_dev(amx_ctx* ctx, const amx_mfcc_plan* p, const float* feats_dev, int dim, int left, int right,
                           float* out_dev, int out_stride) {
    AMX_REQUIRE(ctx && p && feats_dev && out_dev, AMX_ERR_INVALID, "amx_context_window_dev: NULL argument");
    AMX_REQUIRE(dim > 0 && left >= 0 && right >= 0 && out_stride >= (left + right + 1) * dim, AMX_ERR_INVALID,
                "amx_context_window_dev: out_stride %d < window %d", out_stride, (left + right + 1) * dim);
    const long long total = p->frame_off.back();
    if (total == 0)
        return AMX_OK;
    AMX_HIP(hipSetDevice(ctx->device));
    amx::ScopedKernelTimer timer(ctx, "context_window");
    hipLaunchKernelGGL(amx::context_window_kernel, dim3((unsigned)total), dim3(256), 0, ctx->stream, feats_dev,
                       p->d_frame_off, p->n_seg, dim, left, right, out_dev, out_stride, total);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

}  // extern "C"

#ifdef AMX_LAB
extern "C" int amx_lab_mfcc_r16_stamps(unsigned long long* out /* [4 * 32 * 8] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(amx::mfcc_r16_stamps), sizeof(amx::mfcc_r16_stamps)) == hipSuccess ? AMX_OK : AMX_ERR_DEVICE;
}
extern "C" int amx_lab_mfcc_stamps(unsigned long long* out /* [4 * 32 * 8] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(amx::mfcc_stamps), sizeof(amx::mfcc_stamps)) == hipSuccess ? AMX_OK : AMX_ERR_DEVICE;
}
#endif
