// dmel_kernels.h -- shared between the host API (dmel_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmel {

constexpr int kWave = 64;
constexpr int kThreads = 256;          // prep / dot kernels
constexpr int kMaxChunks = 64;         // partial sums per clip for the DC removal
constexpr int kMaxFastNfft = 4096;     // largest transform of the fused wave-FFT kernel
constexpr int kMaxNfft = 16384;        // largest transform overall: above kMaxFastNfft the one-frame-per-workgroup LDS FFT runs
constexpr int kMinFastNfft = 32;       // below this the direct-DFT kernel runs

// fp32 -> bf16 bits, round to nearest even (v_cvt_pk_bf16_f32 on gfx950); outputs only, the arithmetic stays fp32
__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

enum Mode : int { kTrain = 0, kInfer = 1, kSpec = 2, kSpecTrain = 3 };   // kSpec*: power spectrogram (B,F,T), no mel stage

// Compile-time FFT plan for one wave: N = R * R * C, R points per lane (tools/wavefft_sim.py).
// MINW: waves per SIMD the register allocation must allow (what the LDS footprint admits).
// WAVES waves per workgroup; each wave runs PASSES rounds of FPW = 64/G frames.  In the contraction phase
// every wave works on NLOC = 2 runs of k-steps: with 4 waves, the whole mel tiles w and 7-w; with 8 waves,
// the first half of tile w and the second half of tile 7-w (exchanged through 8 KB of LDS), which
// evens out the HTK bands that widen with frequency.
template <int N> struct FftPlan;
template <> struct FftPlan<32>   { static constexpr int R = 4,  C = 2, PASSES = 1, WAVES = 4, NBPRE = 4, MINW = 4; };
template <> struct FftPlan<64>   { static constexpr int R = 8,  C = 1, PASSES = 1, WAVES = 4, NBPRE = 4, MINW = 4; };
template <> struct FftPlan<128>  { static constexpr int R = 8,  C = 2, PASSES = 1, WAVES = 4, NBPRE = 8, MINW = 4; };
template <> struct FftPlan<256>  { static constexpr int R = 16, C = 1, PASSES = 1, WAVES = 4, NBPRE = 12, MINW = 4; };
template <> struct FftPlan<512>  { static constexpr int R = 16, C = 2, PASSES = 1, WAVES = 4, NBPRE = 20, MINW = 4; };
template <> struct FftPlan<1024> { static constexpr int R = 16, C = 4, PASSES = 1, WAVES = 8, NBPRE = 20, MINW = 4; };
template <> struct FftPlan<2048> { static constexpr int R = 32, C = 2, PASSES = 1, WAVES = 8, NBPRE = 40, MINW = 2; };
template <> struct FftPlan<4096> { static constexpr int R = 64, C = 1, PASSES = 1, WAVES = 4, NBPRE = 8, MINW = 1; };

struct FftGeom {
    int N, R, C, G, FPW, PASSES, WAVES, NLOC, THREADS, SLOTS, MT, EX_STRIDE, SLOT_STRIDE_F2, NBPRE, MINW, LDS_BYTES, AUX_OFF, RED_OFF, WIN_LDS;
};

constexpr int ex_stride(int G, int C) { return G + (G >= 32 ? C : 1); }
constexpr int z_index_host(int k, int R, int C) { return C == 1 ? k : k + (k / (R * R)) * 4; }
constexpr int slot_stride_f2(int N, int R, int C)
{
    const int G = N / R;
    const int a = R * ex_stride(G, C);
    const int b = z_index_host(N - 1, R, C) + 1;
    const int need = (a > b ? a : b) * 8;
    // = 48 bytes (mod 128).  The A operands of phase 2 are read with ds_read_b32 / ds_read2_b32, which bank modulo 32 dwords
    // (128 B) in two 32-lane groups: each group touches 16 B per slot, so the 8 slots must start 16 B apart modulo 128
    // (a stride = 32 mod 256 put slots s and s+4 on the same banks: 2-way conflicts, SQ_LDS_BANK_CONFLICT = 25 % of the
    // LDS-active cycles).  48 rather than 16 keeps the two frames a wave exchanges at n_fft 32 / 64 apart as before.
    return ((need + 127) / 128 * 128 + 48) / 8;
}

template <int N> constexpr FftGeom geom()
{
    using P = FftPlan<N>;
    FftGeom g{};
    g.N = N; g.R = P::R; g.C = P::C; g.G = N / P::R; g.FPW = kWave / g.G; g.PASSES = P::PASSES;
    g.WAVES = P::WAVES; g.NLOC = 2; g.THREADS = kWave * P::WAVES; g.NBPRE = P::NBPRE; g.MINW = P::MINW;
    g.SLOTS = g.WAVES * g.FPW * g.PASSES;
    g.MT = g.SLOTS >= 8 ? g.SLOTS / 8 : 1;
    g.EX_STRIDE = ex_stride(g.G, g.C);
    g.SLOT_STRIDE_F2 = slot_stride_f2(N, P::R, P::C);
    // LDS map: [FFT slots][aux: window table (phase 1) aliased with the half-tile exchange (phase 2)][8 sums]
    g.WIN_LDS = (N <= 2048) ? 1 : 0;                      // n_fft 4096 has no room: its window stays in global memory
    const int xch = (g.WAVES == 8) ? 8 * 64 * 16 : 0;
    const int win = g.WIN_LDS ? N * 8 : 0;
    g.AUX_OFF = g.SLOTS * g.SLOT_STRIDE_F2 * 8;
    g.RED_OFF = g.AUX_OFF + (xch > win ? xch : win);
    // [8 sums (64 B)][tw2 table: R x C complex, the radix-C twiddles: every lane of a wave reads one of C values per p1]
    g.LDS_BYTES = g.RED_OFF + 64 + ((P::C > 1 && N <= 2048) ? P::R * P::C * 8 : 0);
    return g;
}

// One 4(k) x 16(mel) block of the filterbank, pre-arranged as the B operand of
// v_mfma_f32_16x16x4_f32: lane l holds fb[4*ks + (l >> 4)][16*tile + (l & 15)].
struct FwdParams {
    const float* x;            // (B, L)
    float* out;                // (B, 1, M, T) or spec (B, F, T) in kSpec mode
    float* tangent;            // same shape as out or nullptr
    const float* psum;         // (B, nchunks) partial sums of x, or nullptr: the kernel sums the clip itself (short clips)
    const float2* win2;        // [n] = (w[n], dw[n]/d|lambd| * dw_scale)
    const float2* tw1;         // (R, G): w_N^(lg*q)
    const float2* tw2;         // (R, C): w_G^(r*p1)
    const float* ent_b;        // 64 floats per 4x16 block, blocks of one mel tile contiguous in k
    const int4* tile_ranges;   // (groups, WAVES, 2): {first k-step, #k-steps (multiple of 4), offset into ent_b, mel tile or -1}
    int ent_b_floats;          // size of ent_b (buffer bounds)
    const float* ent_pre;      // (WAVES, 2, NBPRE, 64): the first NBPRE k-steps of every run of mel group 0, zero padded
    int pre_groups[16];        // per (wave, run): how many groups of 4 k-steps of ent_pre are real (the rest is padding nobody reads)
    int B, L, T, hop, M, nchunks, groups, tiles_per_clip;
    float inv_L, sign, eps;
    float lambd_abs, dw_scale;  // for the in-kernel window table (time_frequency.py:21-30)
    float dw_k3;                // dw_scale / (|lambd| + 1e-15)^3, computed on the host in fp64
    unsigned flags;
    int remove_dc, normalize;
    int win_half;               // window support = middle half of n_fft (DSPEC: win_length = n_fft / 2)
};

struct PrepParams {
    const float* x; float* psum; float2* win2;
    int B, L, nchunks, chunk, N, normalize;
    float lambd_abs;
    float dw_scale;   // power of two ~ |lambd|: the dw table is stored pre-multiplied by it (see dmel_api.cpp)
    int win_half;
};

hipError_t launch_prep(const PrepParams& p, hipStream_t s);
hipError_t launch_forward(int n_fft, int mode, const FwdParams& p, int grid, hipStream_t s);
int forward_lds_bytes(int n_fft);
int forward_frames_per_tile(int n_fft, int mode);
int forward_waves(int n_fft);              // waves per workgroup of the fused kernel for this n_fft
int forward_nbpre(int n_fft);              // k-steps per run kept in registers (layout of FwdParams::ent_pre)
hipError_t forward_prepare_attributes();   // raises the dynamic-LDS limit of every instantiation once

// direct-DFT kernel for n_fft < 32 (and as an on-device cross-check of the fast path)
struct NaiveParams {
    const float* x; float* out; float* tangent; const float* psum; const float2* win2; const float* fb;
    int B, L, T, hop, M, nchunks, N, F, mode;
    float inv_L, sign, eps; unsigned flags; int remove_dc;
};
hipError_t launch_naive(const NaiveParams& p, hipStream_t s);

// long transforms (4096 < n_fft <= 16384, |lambd| up to 2730 samples): one workgroup per frame (pair), radix-2 FFT in LDS
struct LongParams {
    const float* x; float* out; float* tangent; const float* psum; const float2* win2;
    const float2* tw;        // (N/2): exp(-2 pi i k / N)
    const float* fbT;        // (M, F): filterbank transposed, so that one band is contiguous
    const int2* band;        // (M): [first, last+1) non-zero rows of every filterbank column
    int B, L, T, hop, M, nchunks, N, F, mode, logN;
    float inv_L, sign, eps; unsigned flags; int remove_dc;
};
hipError_t launch_long(const LongParams& p, hipStream_t s);
hipError_t long_prepare_attributes();

// gradient w.r.t. the waveform (dmel_xgrad.hip)
struct XgradParams {
    const float* x; const float* psum; const float2* win2; const float2* tw;
    const float* fb;            // (F, M) dense filterbank
    const int2* rowband;        // (F): [first, last+1) non-zero columns of every filterbank row
    const float* grad_out;      // (B, M, T)
    const float* out;           // (B, M, T) saved log output or nullptr
    float* frames;              // (B, T, N) workspace: windowed gradient of every frame
    float* grad_x;              // (B, L)
    double* csum;               // (B, chunks) fp64 sums of the gather chunks (mean of the clip's gradient)
    int B, L, T, hop, M, nchunks, N, F, logN, remove_dc;
    int tw_in_lds;              // set by launch_xgrad: the twiddle table is copied behind the sequence in LDS
    float inv_L;
};
hipError_t launch_xgrad(const XgradParams& p, hipStream_t s);
hipError_t xgrad_prepare_attributes();
int xgrad_chunks(int L);      // gather chunks per clip (size of XgradParams::csum per clip)

// gradient w.r.t. the filterbank matrix of models.py:53 (adjoint of  mel = spec^T @ fb):
//   grad_fb[f][m] = sum_{b,t} spec[b][f][t] * gm[b][m][t],   gm = grad_out            (linear output)
//                                                            gm = grad_out * exp(-out) (log output: d log(s+eps) = ds / (s+eps))
struct FbGradParams {
    const float* spec;       // (B, F, T) power spectrogram of the same clips (kSpec pass)
    const float* grad_out;   // (B, M, T)
    const float* out;        // (B, M, T) saved log output, or nullptr for the linear layer
    float* gm_ws;            // (B, M, T) workspace for gm when out != nullptr
    const float* gm;         // set by launch_fbgrad: grad_out or gm_ws
    float* partials;         // (splits, F, M)
    float* grad_fb;          // (F, M)
    int B, F, M, T, splits;
};
hipError_t launch_fbgrad(const FbGradParams& p, hipStream_t s);

// g: fp32, or bf16 when g_bf16 != 0 (the gradient of a bf16 output); t (the tangent) is always fp32
hipError_t launch_dot(const void* g, int g_bf16, const float* t, long long count, int accumulate, double* partials,
                      unsigned* counter, int max_partials, float* result, hipStream_t s);

}  // namespace dmel
