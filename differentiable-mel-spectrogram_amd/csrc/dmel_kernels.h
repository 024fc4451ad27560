// dmel_kernels.h -- shared between the host API (dmel_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmel {

constexpr int kWave = 64;
constexpr int kThreads = 256;          // prep / dot kernels
constexpr int kMaxChunks = 64;         // partial sums per clip for the DC removal
constexpr int kMaxFastNfft = 16384;    // largest transform of the fused wave-FFT kernel (8192 and 16384: two / four waves per frame)
constexpr int kMaxNfft = 16384;        // largest power-of-two transform held in LDS; beyond it (and for other lengths) dmel_big.hip
constexpr int kMinFastNfft = 32;       // below this the direct-DFT kernel runs

// fp32 -> bf16 bits, round to nearest even (v_cvt_pk_bf16_f32 on gfx950); outputs only, the arithmetic stays fp32
__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

enum Mode : int { kTrain = 0, kInfer = 1, kSpec = 2, kSpecTrain = 3,      // kSpec*: power spectrogram (B,F,T), no mel stage
                  kTrainH = 4,     // kTrain with the DENSE contraction on the bf16 matrix pipe (DMEL_FLAG_MFMA_BF16X3): the pairing pass
                                   // leaves PD as four bf16 planes (P hi, P lo, D hi, D lo), three v_mfma_f32_16x16x32_bf16 per fp32 product
                  kTrainW = 5,     // kTrain with the WAVE-LOCAL contraction (round 5): every wave contracts the frames it transformed itself with
                                   // v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 blocks: rows = the wave's (frame, P | D) pairs, one block per
                                   // quad of mel bands) -- no workgroup barrier between transform and contraction, no exchange of partial sums
                  kTrainWW = 6 };  // kTrainW with workgroups of 16 waves (n_fft 1024: 32 frames, one workgroup per CU): a clip of up to 32 frames is ONE
                                   // tile -- its mean comes from the samples the frames load anyway (no second pass over the clip), one window and
                                   // twiddle table per CU, whole 128-byte rows in the staged epilogue
constexpr bool mode_wlc(int mode) { return mode == kTrainW || mode == kTrainWW; }
// sizes kTrainW is built for: whole frames inside one wave, compact layout (FftPlan::PAIRING == kPairBperm)
// (n_fft 4096, one frame per wave and one workgroup per CU: 147.6 us against 138.5 for kTrain at the reference's ESC-50 shape -- with two of the
// four MFMA rows in use a wave issues 264 steps for 64 mel bands; n_fft 2048: config 3 46.0 against 48.0 us, config 5 70.4 against 70.2)
// (n_fft 4096 in this scheme -- one frame per wave, quads split over blocks: 192 steps for 64 mel bands instead of 328 -- measured 140.6 us
// against 139.7 for the 16 x 16 x 4 path at the reference's ESC-50 shape: not built by default)
#ifndef DMEL_WLC_4096
#define DMEL_WLC_4096 0
#endif
// (round 6: n_fft 512 -- BASELINE config 1 and the reference's x = 0.035 grid point, search_spaces.py:29 -- in this scheme with a plan of its own
// (FftPlanSel3 below: the 16 x 16 x 2 transform of the other modes in the compact layout, bpermute pairing for G = 32, C = 2): built, parity-green
// (270 tests), and SLOWER -- the reference's ESC-50 shape 24.3 us with 16-frame workgroups, 23.6 with 8-frame ones, against 22.9 for the
// 16 x 16 x 4 tiles with two tiles per workgroup; 256 clips of config 1's layer 23.1 / 24.3 against 22.0.  One phase of 16 quads is as long as
// the widest quad's band (64 mel bands on 257 bins), and the one-tile workgroups pay the prologue twice as often.  Not built by default.)
#ifndef DMEL_WLC_512
#define DMEL_WLC_512 0
#endif
constexpr bool wlc_size(int n_fft) { return (DMEL_WLC_512 && n_fft == 512) || n_fft == 1024 || n_fft == 2048 || (DMEL_WLC_4096 && n_fft == 4096); }
constexpr int kWlMaxPhases = 8;    // phases of 16 mel quads each: up to 512 mel bands (more: the host falls back to kTrain)
constexpr int kHsplitMinNfft = 64, kHsplitMaxNfft = 4096;    // sizes kTrainH is built for (frames inside one wave, N/2 a multiple of 32)
constexpr int hsplit_plane_stride(int n_fft) { return n_fft / 2 + 8; }   // bf16 entries per plane: bins 0 .. N/2, rows stay 16-byte aligned

// ---- where lambd comes from, and what a launch does when it was built for another n_fft ------------------------------
// The reference derives n_fft from the parameter on the host at every forward (time_frequency.py:39: a device->host
// read per sample).  Here lambd may stay on the device: every kernel of the forward reads it with a scalar load, derives
// the window constants itself and checks that next_pow2(int(6|lambd|)) is the n_fft it was launched for.  A forward is a
// GROUP of launches on one stream: the primary (the n_fft the host last saw) and, near a power-of-two boundary, guard
// launches for the neighbouring n_fft that return at once unless the device value says the work is theirs.  If no launch
// of the group matches, the last one fills the outputs with NaN and raises a sticky error word in pinned host memory.
constexpr unsigned kLamFirst = 1u, kLamLast = 2u, kLamQuiet = 4u;   // quiet: read and check only (the window-table kernel)
// Every forward that EXECUTES on a plan draws the next execution number from a device counter (first launch of its group, one
// atomic add by one lane) and reports (number << 32) | bits(lambd) into slot (number % kLamRing) of a pinned ring.  The number
// counts executions, not host calls: a forward replayed from a HIP graph draws a fresh one at every replay, so the host can
// tell a replay's report from an older eager call's, and can look up the report of one particular execution (what a
// multi-rank caller needs to take the same decision on every rank).
constexpr unsigned kLamRing = 256;     // (64 until round 4: nine replays of twenty steps in flight overran it -- dmel_lambd_ring_size)
struct LamArgs {
    const float* dev;                 // device scalar (nullptr: use val)
    float val;                        // lambd by value (dmel_forward: the host read it, as the reference does)
    int n_expected;                   // 0: no check (explicit n_fft: optimized=False branches, dmel_spectrogram_ex)
    unsigned role;                    // kLamFirst | kLamLast inside the group
    unsigned* exec_counter;           // device, plan-owned: execution numbers (nullptr: nothing is reported)
    unsigned* handled;                // 2 device words of the group (caller scratch): [0] set by the launch that does the work,
                                      // [1] the execution number drawn by the first launch (the last one reports errors under it)
    unsigned long long* host_seen;    // pinned ring of kLamRing words, written by the first launch of every group
    unsigned long long* host_error;   // pinned, sticky: (number << 32) | bits(lambd) of a forward no launch covered
    unsigned* dot_counter;            // ticket word of the caller's dot scratch, zeroed by the first launch (or nullptr)
};

// The dot kernel's hand-off (dmel_aux.hip) is a TREE of tickets since round 6: workgroups in groups of kDotGroup draw a ticket on their group's
// counter, the last of a group draws one on the kernel's counter, the last of those combines.  (One counter for all: ~11.7 ns per ticket,
// serialised -- 128 workgroups of 8192 elements were the optimum at BASELINE config 2 and 256 cost +1 us, 512 +4 us.)  Scratch layout
// (dmel_api.cpp: carve): 1024 fp64 partials, then the ticket word; at most kDotMaxBlocks partials are used and the group counters sit in
// the upper half of that array, 64 bytes apart: word 16 g of (partials + kDotMaxBlocks) = 4096 bytes in front of the ticket word.
constexpr int kDotGroup = 16, kDotMaxBlocks = 512;
__host__ __device__ inline unsigned* dot_group_counters(unsigned* ticket_word)
{
    return reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(ticket_word) - 4096);
}

enum LamAction : int { kLamRun = 0, kLamSkip = 1, kLamPoison = 2 };
struct LamState {
    float lam, a, denom;              // lambd, |lambd| (models.py:38), |lambd| + 1e-15 (time_frequency.py:24)
    float s2;                         // 2^(-2e), 2^(e-1) <= |lambd| < 2^e: the tangent window is stored as w d^2 2^(-2e) (same scale as w)
    int e2;                           // 2e
    int action;
};

#if defined(__HIPCC__)
// ---- the clip mean of models.py:38 ------------------------------------------------------------------------------------
// The reference subtracts torch.mean(x[idx]) in fp32.  When a clip's offset is far above its signal, ONE ulp of that mean moves
// the lowest mel bands by up to 3e-2 (tests/golden g13_dc_*: offset 500 x signal), and torch's own fp32 sum is an ulp off the
// correctly rounded mean in a third of such clips: bit parity with it is not a property of the algorithm, staying within an ulp
// or two of it is.  Here: short fp32 accumulator chains per thread, a pairwise fp32 tree over lanes and waves, and the QUOTIENT
// sum / L rounded once (round 5 multiplied by fl32(1 / L): a second and third rounding).  Measured on the g13 fixtures: the
// correctly rounded mean or its neighbour.  An fp64 tree from the accumulators on gives the correctly rounded mean in every clip
// and was built and measured in round 6: +1 % on config 4's batch and config 3 (a chain of ~20 dependent fp64 operations between
// the arrival of the clip and every wave's first window multiply) -- not kept: the reference itself is an ulp off.
// Long clips: the <= 64 chunk sums of dmel_prep_kernel (fp64 inside a chunk, rounded per chunk: a 1 / sqrt(chunks) of the total's ulp).
__device__ __forceinline__ float mean_quotient(float sum, int L, float inv)       // inv = fl32(1 / L), from the host
{
    const float fl = (float)L;                                   // (L < 2^24: exact)
    const float q = sum * inv;
    return fmaf(fmaf(-q, fl, sum), inv, q);                      // one Newton step: the correctly rounded quotient up to a 2^-24 of an ulp
}
__device__ __forceinline__ float clip_mean_psum(const float* psum, int nchunks, int b, int L)
{
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += (double)psum[(size_t)b * nchunks + c];
    return (float)(s / (double)L);
}

// time_frequency.py:39,60-65 on the device, bit for bit the host's dmel_n_fft: fp32 product, int() truncation,
// 1 << bit_length(x-1).  lambd is uniform: after one multiply and one conversion the value is moved to a scalar register and
// the rest is scalar integer arithmetic (the conversion saturates at INT_MAX, which lands in the same "too large" answer
// as the host's 64-bit path; NaN converts to 0).
__device__ __forceinline__ int lam_n_fft(float a)
{
    const float prod = a * 6.0f;
    const int x = __builtin_amdgcn_readfirstlane((int)prod);
    const int v = x - 1;
    const int bits = v < 0 ? 1 : (v == 0 ? 0 : 32 - __builtin_clz((unsigned)v));
    return bits > 30 ? 0x40000000 : (1 << bits);
}

// lambd itself: by value, or one load from the parameter's storage
__device__ __forceinline__ float lam_load(const LamArgs& la)
{
    // The device value through a CONSTANT-address-space pointer: a scalar load (lgkmcnt), not a vector one.  No kernel of a forward writes lambd
    // (the optimizer's kernel, earlier in the stream, did; the scalar cache is invalidated at every kernel start, as it must be for
    // the kernel arguments themselves), so the qualifier tells the truth.  A select between a device pointer and the generic address
    // of `la.val` compiled to a FLAT load followed by `s_waitcnt vmcnt(0)`: every wave of dmel_fwd_kernel waited for all of its sample
    // loads before it could start on the window table (found in round 4 in the assembly; vector loads complete in order).
    // (Only the DEVICE pointer is re-qualified: `la.val` is read as it is -- in a kernel whose argument struct the compiler copies to
    // private memory its address is not a global address, and a scalar load from it faults.)
    typedef const __attribute__((address_space(4))) float cfloat;
    float v = la.val;
    if (la.dev) v = *(cfloat*)la.dev;
    return v;
}

// `leader` = one thread of the whole grid.  n_launch = the n_fft this kernel instance computes.
__device__ __forceinline__ LamState lam_prologue(const LamArgs& la, int n_launch, bool leader, float lam_value)
{
    LamState st;
    st.lam = lam_value;
    st.a = __builtin_fabsf(st.lam);
    st.denom = st.a + 1e-15f;
    int e = __builtin_amdgcn_readfirstlane(st.a > 1e-30f ? __builtin_amdgcn_frexp_expf(st.a) : 1);
    e = e < -30 ? -30 : (e > 30 ? 30 : e);
    st.e2 = 2 * e;
    st.s2 = __builtin_bit_cast(float, (127 - 2 * e) << 23);           // 2^(-2e), built in scalar registers
    const bool match = la.n_expected == 0 || lam_n_fft(st.a) == n_launch;
    st.action = match ? kLamRun : kLamSkip;
    if (la.role & kLamQuiet) return st;
    if (la.role & kLamFirst) {
        if (leader) {
            if (la.exec_counter && la.host_seen) {
                const unsigned num = __hip_atomic_fetch_add(la.exec_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                __hip_atomic_store(la.host_seen + (num % kLamRing), ((unsigned long long)num << 32) | __builtin_bit_cast(unsigned, st.lam),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (la.handled) la.handled[1] = num;
            }
            if (la.dot_counter) {
                *la.dot_counter = 0u;
                // ... and the dot kernel's group counters (kDotGroupCounters below), which live 4 KB in front of the ticket word in the caller's scratch
                unsigned* gc = dot_group_counters(la.dot_counter);
                for (int g2 = 0; g2 < kDotMaxBlocks / kDotGroup; ++g2) gc[16 * g2] = 0u;
            }
            if (la.handled) la.handled[0] = match ? 1u : 0u;
        }
    } else if (match && leader && la.handled) {
        la.handled[0] = 1u;
    }
    if (!match && (la.role & kLamLast)) {
        // earlier launches of the group are complete (stream order): their words are visible to a plain load
        const unsigned prior = ((la.role & kLamFirst) || !la.handled) ? 0u : la.handled[0];
        if (!prior) {
            st.action = kLamPoison;
            if (leader && la.host_error) {
                // the leader of a single-launch group wrote handled[1] itself (program order); otherwise an earlier launch did
                const unsigned num = la.handled ? la.handled[1] : 0u;
                __hip_atomic_store(la.host_error, ((unsigned long long)num << 32) | __builtin_bit_cast(unsigned, st.lam),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    return st;
}

__device__ __forceinline__ LamState lam_prologue(const LamArgs& la, int n_launch, bool leader)
{
    return lam_prologue(la, n_launch, leader, lam_load(la));
}

// sign(lambd) 2^(2e) / (|lambd| + 1e-15)^3: what turns the contraction of the scaled tangent window into d / d lambd
// (fp64, rounded once; lambd == 0 gives 0, an overflow gives 0: w' vanishes wherever w does not)
__device__ __forceinline__ float lam_tangent_scale(const LamState& st)
{
    const double den = (double)st.denom;
    const double sg = st.lam > 0.f ? 1.0 : (st.lam < 0.f ? -1.0 : 0.0);
    const float c = (float)(sg * __builtin_ldexp(1.0, st.e2) / (den * den * den));
    return (c == c && __builtin_fabsf(c) < 3.0e38f) ? c : 0.f;
}
#endif

// Compile-time FFT plan for one wave: N = R * R * C, R points per lane (tools/wavefft_sim.py).
// MINW: waves per SIMD the register allocation must allow (what the LDS footprint admits).
// WAVES waves per workgroup; each wave runs PASSES rounds of FPW = 64/G frames.  In the contraction phase
// every wave works on NLOC = 2 runs of k-steps: with 4 waves, the whole mel tiles w and 7-w; with 8 waves,
// the first half of tile w and the second half of tile 7-w (exchanged through 8 KB of LDS), which
// evens out the HTK bands that widen with frequency.
template <int N> struct FftPlan;
// PAIRING: how the pairing pass gets Z[N-k].  0: the whole spectrum is stored to LDS and read back mirrored.  1 (kPairBperm):
// from the lane that holds it, with ds_bpermute_b32; only PD[0..N/2] is ever stored.  2 (kPairPlane, frames spread over several
// waves): through one plane of N floats, real parts first, imaginary parts second.  SPLIT: the transposition between the two
// radix-R stages moves the real and the imaginary parts one after the other through ONE plane of N floats; together with
// PAIRING != 0 and a half-length (symmetric) window table a frame in flight needs N*4 bytes of LDS instead of N*8 (n_fft 2048:
// two workgroups per CU instead of one).
// G = N / R lanes work on one frame: a part of a wave (G < 64: 64/G frames per wave), one wave, or G/64 waves (n_fft 8192,
// 16384: the exchanges between them are workgroup barriers instead of wave barriers).
constexpr int kPairLds = 0, kPairBperm = 1, kPairPlane = 2;
template <> struct FftPlan<32>   { static constexpr int R = 4,  C = 2, PASSES = 1, WAVES = 4, NBPRE = 4, MINW = 4, PAIRING = 0, SPLIT = 0; };
template <> struct FftPlan<64>   { static constexpr int R = 8,  C = 1, PASSES = 1, WAVES = 4, NBPRE = 4, MINW = 4, PAIRING = 0, SPLIT = 0; };
template <> struct FftPlan<128>  { static constexpr int R = 8,  C = 2, PASSES = 1, WAVES = 4, NBPRE = 8, MINW = 4, PAIRING = 0, SPLIT = 0; };
template <> struct FftPlan<256>  { static constexpr int R = 16, C = 1, PASSES = 1, WAVES = 4, NBPRE = 12, MINW = 4, PAIRING = 0, SPLIT = 0; };
template <> struct FftPlan<512>  { static constexpr int R = 16, C = 2, PASSES = 1, WAVES = 4, NBPRE = 20, MINW = 4, PAIRING = 0, SPLIT = 0; };
// 1024 = 32 x 32: two frames per wave (G = 32 lanes each), no cross-lane stage, compact layout.  16 frames per workgroup of 8 waves at
// 76.6 KB of LDS: two workgroups = 32 frames per CU, which is BASELINE config 2's whole share of a CU in ONE round (21.6 -> 19.9 us).
// (one workgroup of 16 waves per CU -- 32 frames, a whole 16000-sample clip at hop 512, whose mean then comes from the samples the
// frames load anyway, one window table per CU -- measured 20.3-20.6 us against 19.9 at BASELINE config 2: the barriers of 16 waves
// cost more than the second pass over the clip)
template <> struct FftPlan<1024> { static constexpr int R = 32, C = 1, PASSES = 1, WAVES = 8, NBPRE = 20, MINW = 4, PAIRING = 1, SPLIT = 1; };
template <> struct FftPlan<2048> { static constexpr int R = 32, C = 2, PASSES = 1, WAVES = 8, NBPRE = 40, MINW = 4, PAIRING = 1, SPLIT = 1; };
// (4096 as two waves per frame, R = 32, C = 4, two workgroups per CU: 291 us against 184 at the reference's ESC-50 shape --
// the workgroup barriers of a shared frame cost more than the occupancy gives)
template <> struct FftPlan<4096> { static constexpr int R = 64, C = 1, PASSES = 1, WAVES = 8, NBPRE = 40, MINW = 2, PAIRING = 1, SPLIT = 1; };
template <> struct FftPlan<8192> { static constexpr int R = 64, C = 2, PASSES = 1, WAVES = 8, NBPRE = 40, MINW = 2, PAIRING = 2, SPLIT = 1; };
template <> struct FftPlan<16384> { static constexpr int R = 64, C = 4, PASSES = 1, WAVES = 8, NBPRE = 40, MINW = 2, PAIRING = 2, SPLIT = 1; };

// The plan also depends on the MODE: when two neighbouring frames share one complex FFT (inference, spectrogram pass) a wave of
// the 32 x 32 plan would carry four frames and a workgroup 32 -- one workgroup per CU at config 2, half the waves (measured 16.3 us
// against 14.0).  Those modes keep 1024 = 16 x 16 x 4, one frame pair per wave, spectrum in LDS.
template <int N, bool PAIR> struct FftPlanSel : FftPlan<N> {};
template <> struct FftPlanSel<1024, true> { static constexpr int R = 16, C = 4, PASSES = 1, WAVES = 8, NBPRE = 20, MINW = 4, PAIRING = 0, SPLIT = 0; };
constexpr bool mode_pairs(int mode) { return mode == kInfer || mode == kSpec; }
// ... and on the CONTRACTION: kTrainW needs whole frames inside one wave in the compact layout (spectrum kept in registers, pairing by
// ds_bpermute, PD[0 .. N/2] unpadded over one plane).  n_fft 512 = 16 x 16 x 2 has two frames per wave (G = 32 lanes each, as n_fft 1024) and a
// radix-2 stage across adjacent lanes (as n_fft 2048): same radix, same twiddle tables as FftPlan<512>, other LDS layout.
template <int N, bool PAIR, bool WL> struct FftPlanSel3 : FftPlanSel<N, PAIR> {};
#ifndef DMEL_WL512_WAVES
#define DMEL_WL512_WAVES 8
#endif
template <> struct FftPlanSel3<512, false, true> { static constexpr int R = 16, C = 2, PASSES = 1, WAVES = DMEL_WL512_WAVES, NBPRE = 20, MINW = 4, PAIRING = 1, SPLIT = 1; };

// internal bit of FwdParams::flags (the public ones are DMEL_FLAG_* of include/dmel.h, the 0x100.. bits belong to -DDMEL_ABLATE builds):
// the launch fits the chip in ONE round of resident workgroups, see dmel_fwd_kernel's prologue
constexpr unsigned kFwdEdgeFirst = 0x40u;
int forward_resident_workgroups(int n_fft, int mode);      // workgroups of this instantiation the device holds at once (LDS-limited)
constexpr int kWinLdsMaxNfft = 4096;   // (the table in global memory at 4096: 184 us against 171 at the reference's ESC-50 shape)
constexpr int kRedBytes = 160;         // 16 + 16 partial sums (one per wave), then the tangent scale (word 32) computed once per workgroup
constexpr int kRedTan = 32;
struct FftGeom {
    int N, R, C, G, FPW, PASSES, WAVES, NLOC, THREADS, SLOTS, MT, EX_STRIDE, SLOT_STRIDE_F2, NBPRE, MINW, LDS_BYTES, AUX_OFF, RED_OFF, WIN_LDS;
    int PAIRING, SPLIT, WIN_SYM, WPF, PLANE_PAD;
    int TW1_OFF;      // kTrainW, where it fits: byte offset of the first-stage twiddle table in LDS (rows q = 1 .. R - 1 of FwdParams::tw1), else 0
};

constexpr int ex_stride(int G, int C) { return G + (G >= 32 ? C : 1); }
constexpr int z_index_host(int k, int R, int C) { return C == 1 ? k : k + (k / (R * R)) * 4; }
constexpr int slot_stride_f2(int N, int R, int C, int pairing = 0, int split = 0, bool wl = false)
{
    const int G = N / R;
    const int a = R * ex_stride(G, C) * (split ? 4 : 8);                           // transposition: complex entries, or one plane of floats
    const int b = (z_index_host(pairing ? N / 2 : N - 1, R, C) + 1) * 8;           // spectrum Z[0..N-1], or only PD[0..N/2]
    const int need = a > b ? a : b;
    // = 48 bytes (mod 128).  The A operands of phase 2 are read with ds_read_b32 / ds_read2_b32, which bank modulo 32 dwords
    // (128 B) in two 32-lane groups: each group touches 16 B per slot, so the 8 slots must start 16 B apart modulo 128
    // (a stride = 32 mod 256 put slots s and s+4 on the same banks: 2-way conflicts, SQ_LDS_BANK_CONFLICT = 25 % of the
    // LDS-active cycles).  48 rather than 16 keeps the two frames a wave exchanges at n_fft 32 / 64 apart as before.
    // kTrainW reads its A operands with ds_read_b32 too, but lane 4 b + i of a 32-lane group takes row i = (frame, P | D) of
    // block b's current bin: dwords {2 k, 2 k + 1} of the two slots of a wave.  With the slots 64 bytes apart modulo 128 a block
    // covers the bank pairs {k, k + 8} (mod 16): conflict-free as soon as the 8 blocks of a group sit at bins that differ
    // modulo 8, which the host's schedule arranges (build_tables).  (12 dwords apart, as above: 5 500 cycles per wave for 88 MFMAs.)
    if (wl) return ((need + 127) / 128 * 128 + 64) / 8;
    return ((need + 127) / 128 * 128 + 48) / 8;
}

template <int N, bool PAIR = false, bool WL = false, bool WIDE = false> constexpr FftGeom geom()
{
    using P = FftPlanSel3<N, PAIR, WL>;
    FftGeom g{};
    g.N = N; g.R = P::R; g.C = P::C; g.G = N / P::R; g.PASSES = P::PASSES;
    g.WPF = g.G > kWave ? g.G / kWave : 1;                // waves per frame
    g.FPW = g.G > kWave ? 1 : kWave / g.G;                // frames per wave (1 when a frame has several waves: see SLOTS)
    g.WAVES = WIDE ? 16 : P::WAVES; g.NLOC = 2; g.THREADS = kWave * g.WAVES; g.NBPRE = P::NBPRE; g.MINW = P::MINW;
    g.SLOTS = g.WAVES * g.FPW * g.PASSES / g.WPF;
    // the pairing plane is indexed by k + PLANE_PAD * (k / R^2): the C lanes of a quad write to different banks
    g.PLANE_PAD = P::C > 1 ? 64 / P::C : 0;
    g.MT = g.SLOTS >= 8 ? g.SLOTS / 8 : 1;
    g.EX_STRIDE = ex_stride(g.G, g.C);
    g.PAIRING = P::PAIRING; g.SPLIT = P::SPLIT;
    g.SLOT_STRIDE_F2 = slot_stride_f2(N, P::R, P::C, P::PAIRING, P::SPLIT, WL);
    // LDS map: [FFT slots][aux: window table (phase 1) aliased with the half-tile exchange (phase 2)][8 sums]
    // the window table lives in LDS up to n_fft 4096 (half table there: 16 KB next to 8 x 16.6 KB of frames); beyond, in global
    // memory, written by dmel_prep_kernel
    g.WIN_LDS = (N <= kWinLdsMaxNfft) ? 1 : 0;
    const int xch = g.WAVES * 64 * 16;                    // the exchange of run-1 sums passes one 16-row tile at a time
    // the Gaussian window is symmetric about N/2: the compact layout keeps entries 0..N/2 only
    g.WIN_SYM = (g.WIN_LDS && P::SPLIT) ? 1 : 0;
    const int win = g.WIN_LDS ? (g.WIN_SYM ? (N / 2 + 1) * 8 : N * 8) : 0;
    g.AUX_OFF = g.SLOTS * g.SLOT_STRIDE_F2 * 8;
    g.RED_OFF = g.AUX_OFF + (xch > win ? xch : win);
    g.TW1_OFF = 0;
    if (WL) {
        // kTrainW exchanges nothing: behind the window table there is room for the twiddles w_N^(lg q) of the first stage (n_fft 1024:
        // 31 x 32 entries, 7 936 bytes; two workgroups per CU stay resident) -- 31 LDS reads per lane instead of 4 loads and 27 complex products
        const int tw1 = (P::R - 1) * g.G * 8;
        const int win16 = (win + 15) / 16 * 16;               // (the table is copied 16 bytes at a time)
        const int with_tw1 = g.AUX_OFF + win16 + tw1 + kRedBytes + ((P::C > 1 && N <= 2048) ? P::R * P::C * 8 : 0);
        const int per_cu_now = 163840 / (g.RED_OFF + kRedBytes + ((P::C > 1 && N <= 2048) ? P::R * P::C * 8 : 0));
#ifndef DMEL_WL_TW1LDS
#define DMEL_WL_TW1LDS 1
#endif
        if (DMEL_WL_TW1LDS && 163840 / with_tw1 >= per_cu_now) { g.TW1_OFF = g.AUX_OFF + win16; g.RED_OFF = g.TW1_OFF + tw1; }
        else g.RED_OFF = g.AUX_OFF + win16;
    }
    // [16 sums + tangent scale (kRedBytes)][tw2 table: R x C complex, the radix-C twiddles: every lane of a wave reads one of C values per p1]
    g.LDS_BYTES = g.RED_OFF + kRedBytes + ((P::C > 1 && N <= 2048) ? P::R * P::C * 8 : 0);
    return g;
}

// the geometry of a kernel instantiation (kTrainW has its own slot stride)
template <int N, int MODE> constexpr FftGeom geom_mode() { return geom<N, mode_pairs(MODE), mode_wlc(MODE), MODE == kTrainWW>(); }
// sizes kTrainWW is built for: none.  Measured at n_fft 1024 (round 5): config 2 18.3 us against 18.1 for the 8-wave workgroups -- the
// second pass over the clip it saves is not what a one-round launch waits for --, config 4's batch on one GPU 131 us against 110: a
// CU's next workgroup starts only when all 16 waves of the previous one have finished.  (wlc_wide_size(1024) = true builds it; DMEL_WLC=2.)
constexpr bool wlc_wide_size(int n_fft) { return false; }

// One 4(k) x 16(mel) block of the filterbank, pre-arranged as the B operand of
// v_mfma_f32_16x16x4_f32: lane l holds fb[4*ks + (l >> 4)][16*tile + (l & 15)].
struct FwdParams {
    const float* x;            // (B, L)
    const float* const* x_ind; // DMEL_FLAG_X_INDIRECT: the address of x is read from here when the kernel runs (then x is nullptr)
    float* out;                // (B, 1, M, T) or spec (B, F, T) in kSpec mode
    float* tangent;            // same shape as out or nullptr
    const float* psum;         // (B, nchunks) partial sums of x, or nullptr: the kernel sums the clip itself (short clips)
    const float2* win2;        // [n] = (w[n], w[n] d^2 2^(-2e)) from the prep kernel (n_fft 4096 only)
    const float2* tw1;         // (R, G): w_N^(lg*q)      (of the plan of this launch's mode)
    const float2* tw2;         // (R, C): w_G^(r*p1)
    const float* ent_b;        // 64 floats per 4x16 block, blocks of one mel tile contiguous in k
    const int4* tile_ranges;   // (groups, WAVES, 2): {first k-step, #k-steps (multiple of 4), offset into ent_b, mel tile or -1}; run 0 of an
                               // 8-wave plan: bits 16-23 of the tile word = the waves whose run 1 is a piece of this tile
    int ent_b_floats;          // size of ent_b (buffer bounds)
    const float* ent_pre;      // (WAVES, 2, NBPRE, 64): the first NBPRE k-steps of every run of mel group 0, zero padded
    int pre_groups[16];        // per (wave, run): how many groups of 4 k-steps of ent_pre are real (the rest is padding nobody reads)
    int B, L, T, hop, M, nchunks, groups, tiles_per_clip;
    float inv_L, eps;
    LamArgs lam;                // lambd (device scalar or by value) + the n_fft check
    unsigned flags;
    int remove_dc, normalize;
    int win_half;               // window support = middle half of n_fft (DSPEC: win_length = n_fft / 2)
    int wgs_per_clip;           // ceil(tiles_per_clip / tiles per workgroup)
    unsigned xch_groups;        // 4-wave plans: bit g = mel group g has run-1 pieces that go through the exchange (8-wave plans: always)
    const uint4* ent_h;         // kTrainH: the filterbank as B fragments of v_mfma_f32_16x16x32_bf16, [(tile * (N/64) + kstep) * 2 + (hi | lo)][lane]:
                                // lane l holds fb[32 kstep + 8 (l >> 4) + e][16 tile + (l & 15)], e = 0 .. 7, as bf16 (hi) / the bf16 of the rest (lo)
    const float* fb_nyq;        // kTrainH: (n_mels) fp32, the row of bin N/2 (added on the vector pipe: N/2 bins = N/64 steps of 32 exactly)
    float* spec_out;            // training mode only, or nullptr: the power spectrogram (B, F, T) the contraction consumes is ALSO written
                                // out (16.8 MB at BASELINE config 2), so that the filterbank gradient need not recompute it (16.5 us)
    // kTrainW.  The mel quads (4 consecutive bands) are sorted by the width of their band of bins and taken 16 at a time: phase p gives
    // quad (p, b) to block b of the 4x4x1 MFMA; all 16 blocks walk their bands in lock step, wl_len4[p] groups of 4 bins (zero
    // coefficients where a band is shorter).  Lane 4 b + j supplies row j of block b (A: PD of its frame at bin k0(p, b) + step)
    // and column j (B: fb[k0 + step][4 quad + j]) and receives column j of the block's 4 x 4 result.
    const float4* wl_b4;        // [(steps/4 so far + step/4) * 64 + lane]: the B operands of four consecutive steps
    const int2* wl_lane;        // [phase * 64 + lane]: (8 k0(p, b): byte offset of PD[k0] inside a frame slot, mel band 4 quad + j or -1)
    int wl_phases, wl_total4;   // phases; sum of wl_len4 (size of wl_b4 in 1 KB entries)
    int wl_len4[kWlMaxPhases];
    // quads split over 2 or 4 blocks of a phase (one-frame waves): after the phase's loop the pieces' partial sums are added across lanes
    const int* wl_merge;        // [phase * 64 + lane]: partner lane of round 1 | of round 2 << 8 | this lane receives in round 1 << 16 | in round 2 << 17
    int wl_mg[kWlMaxPhases];    // per phase: bit 0 / 1 = round 1 / 2 has anything to do
};

struct PrepParams {
    const float* x; float* psum; float2* win2;
    const float* const* x_ind;      // DMEL_FLAG_X_INDIRECT: the address of x is read from here (then x is nullptr), as in FwdParams
    int B, L, nchunks, chunk, N, normalize;
    LamArgs lam;      // the window block reads lambd itself (role 0: it neither reports nor poisons)
    int win_half;
    float center;     // the window's centre in table coordinates: N/2, or (L/2 as an integer) + L/2 for the whole-clip window of an
                      // odd clip length L placed in an n_fft = 2L frame (time_frequency.py:24 centres at L/2 as a real number)
};

hipError_t launch_prep(const PrepParams& p, hipStream_t s);
hipError_t launch_forward(int n_fft, int mode, int tiles_per_wg, const FwdParams& p, int grid, hipStream_t s);
int forward_tiles_per_wg(int n_fft, int mode, int batch, int tiles_per_clip);          // 1 or 2: what launch_forward should be given
bool forward_two_tiles(int n_fft, int mode);          // the two-tiles-per-workgroup instantiation exists for this size and mode
int forward_lds_bytes(int n_fft, int mode);
int forward_frames_per_tile(int n_fft, int mode);
int forward_waves(int n_fft);              // waves per workgroup of the fused kernel for this n_fft
int forward_nbpre(int n_fft);              // k-steps per run kept in registers (layout of FwdParams::ent_pre)
bool forward_plan_rc(int n_fft, bool pair, int* R, int* C);   // pair: the plan of the modes that pack two frames per FFT
bool forward_has_hsplit(int n_fft);        // kTrainH is built for this size
bool forward_has_wlc(int n_fft);           // kTrainW is built for this size
bool forward_wlc_one_frame(int n_fft);     // ... with one frame per wave (two of the four MFMA rows idle: wide quads are split over blocks)
bool forward_has_wlc_wide(int n_fft);      // ... and kTrainWW
bool forward_window_in_lds(int n_fft);     // the kernel builds its own window table (otherwise dmel_prep_kernel writes FwdParams::win2)   // radix per lane and cross-lane radix of the plan (layout of tw1 / tw2)
hipError_t forward_prepare_attributes();   // raises the dynamic-LDS limit of every instantiation once

// direct-DFT kernel for n_fft < 32 (and as an on-device cross-check of the fast path)
// DMEL_FLAG_X_INDIRECT (every forward kernel since round 6): the batch's address comes from a pointer cell, read with one scalar load
__device__ __forceinline__ const float* resolve_x(const float* x, const float* const* x_ind)
{
#if defined(__HIPCC__)
    if (x_ind) { typedef const float* cfp; return *(const __attribute__((address_space(4))) cfp*)x_ind; }
#endif
    return x;
}

struct NaiveParams {
    const float* const* x_ind;      // DMEL_FLAG_X_INDIRECT: the address of x is read from here (then x is nullptr), as in FwdParams
    const float* x; float* out; float* tangent; const float* psum; const float2* win2; const float* fb;
    int B, L, T, hop, M, nchunks, N, F, mode;
    float inv_L, eps; unsigned flags; int remove_dc;
    LamArgs lam;
};
hipError_t launch_naive(const NaiveParams& p, hipStream_t s);


// transforms beyond the LDS kernels: power-of-two n_fft > 16384 (global-memory FFT) and lengths that are not powers of two
// (Bluestein), dmel_big.hip
constexpr int kMaxBigFft = 1048576;    // largest power-of-two FFT of that path (n_fft itself, or >= 2 n_fft - 1 for Bluestein): BASELINE config 5's
                                       // 220 500-sample clip in the default optimized=False branch is n_fft 441 000 through 2^20-point FFTs
struct BigParams {
    const float* const* x_ind;      // DMEL_FLAG_X_INDIRECT, as in FwdParams
    const float* x; float* out; float* tangent; const float* psum; const float2* win2;
    const float2* tw;        // (Mfft/2): exp(-2 pi i k / Mfft)
    const float2* chirp;     // (N): exp(-i pi n^2 / N), or nullptr when N is a power of two (then Mfft == N)
    const float2* hbr;       // (Mfft): transform of the chirp filter / Mfft at the bit-reversed positions of a DIF transform
    float2* zws;             // workspace, Mfft complex words per workgroup, when the sequence does not fit LDS
    const float* fbT; const int2* band;
    int B, L, T, hop, M, nchunks, N, F, mode, Mfft, logM;
    float inv_L, eps; unsigned flags; int remove_dc;
    LamArgs lam;
    int split;               // 1: N is taken as two transforms of length N/2 (tw, chirp, hbr, Mfft, logM then describe THAT length)
    const float2* wodd;      // split: (N/2) exp(-2 pi i n / N), the twiddle of the odd half
};
hipError_t launch_big(const BigParams& p, hipStream_t s);
bool big_can_split(int m_half, int n_mels);
hipError_t big_prepare_attributes();
bool big_uses_global(int m_fft);
int big_grid(long long units, int m_fft);

// gradient w.r.t. the waveform (dmel_xgrad.hip)
struct XgradParams {
    const float* x; const float* psum; const float2* win2; const float2* tw;
    const float* fb;            // (F, M) dense filterbank
    const int2* rowband;        // (F): [first, last+1) non-zero columns of every filterbank row
    const float4* rowpk;        // (F): row k as (fb[k][b0], fb[k][b0 + 1], bits(b0), bits(columns)); zeros where the row has fewer columns
    int long_rows;              // some row has more than two columns (their tails are read from fb)
    const float* grad_out;      // (B, M, T)
    const float* out;           // (B, M, T) saved log output or nullptr
    float* frames;              // (B, T, N) workspace: windowed gradient of every frame
    float* grad_x;              // (B, L)
    double* csum;               // (B, T) fp64: what every frame contributes to the sum of the clip's gradient (its mean is removed)
    int B, L, T, hop, M, nchunks, N, F, logN, remove_dc;
    int tw_in_lds;              // set by launch_xgrad: the twiddle table is copied behind the sequence in LDS
    int spec_mode;              // 1: grad_out is (B, F, T), the gradient of the power spectrogram itself (SpectrogramLayer): no filterbank
    float inv_L;
    // wave-FFT path (n_fft 32 ... 2048; tiles == 0: the LDS radix-2 kernels): `frames` then holds (B, tiles, span) overlap-added
    // tile segments and csum one fp64 sum per tile
    const float2* tw1; const float2* tw2;     // tables of the plan that packs two frames per FFT (FftPlanSel<N, true>)
    int tiles, span, tile_step;               // tiles per clip, (FPT - 1) hop + N, FPT hop
    int own_prep;                             // the kernel evaluates the window and the clip mean itself (no dmel_prep_kernel launch)
    float win_denom;                          // |lambd| + 1e-15 (time_frequency.py:24), own_prep only
    const float* lam_dev;                     // own_prep only: lambd on the device (then win_denom is derived from it by the kernel), or nullptr
    int check_nfft;                           // DMEL_FLAG_CHECK_NFFT: every kernel returns at once unless lambd (device) asks for THIS n_fft --
                                              // grad_x and the workspace stay as they were (the caller issues one call per candidate)
    int tw2_off;                              // set by launch_xgrad: byte offset of the radix-C twiddles in LDS
    int win_n;                                // window entries kept in LDS: N/2 + 1 (symmetric about N/2) or N
    // dmel_big.hip path (lengths that are not powers of two, powers of two > 16384): `tw` then belongs to the Mfft-point FFT
    const float2* chirp; const float2* hbr; float2* zws; int Mfft, logM;
};
hipError_t launch_xgrad(const XgradParams& p, hipStream_t s);
hipError_t launch_xgrad_big(const XgradParams& p, int grid, hipStream_t s);     // frames kernel on the global-memory FFT / chirp-z transforms
hipError_t launch_xgrad_gather(const XgradParams& p, hipStream_t s);           // ordered overlap-add of the (B, T, N) frame gradients + mean
hipError_t xgrad_prepare_attributes();
bool xgrad_wave_shape(int n_fft, int n_mels, int win_n, int* frames_per_tile);   // the wave-FFT kernel takes this shape
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
    int ntc, blk_base, blk_rem;   // set by launch_fbgrad: K-blocks per clip; slice s takes blk_base (+1 for s < blk_rem) consecutive blocks
    int fold_last_row;       // set by launch_fbgrad: F = 64 k + 1, the last row rides with the last full tile of 64 rows
    int bf16x3;              // DMEL_FLAG_MFMA_BF16X3: three-term split-bf16 products on the bf16 matrix pipe instead of exact fp32 MFMA
    // d lambd = sum(grad_out * tangent) riding in the same launch (dmel_backward_fb_saved_dl): dot_wgs extra workgroups behind the
    // slices (blockIdx.y >= splits) run dmel_dot_kernel's arithmetic -- dot_vblocks virtual blocks of 256 threads, the same partition,
    // the same order of additions, the same bits -- four virtual blocks each.  dot_wgs = 0: no such workgroups.
    const float* dot_g; const float* dot_t; long long dot_count; double* dot_partials; unsigned* dot_counter; float* dot_result;
    int dot_vblocks, dot_wgs, dot_accumulate;
};
int dot_blocks_for(long long count, int max_partials);      // workgroups launch_dot uses for `count` elements
int fbgrad_fuse_dot(int F, int M, int splits, int vblocks, int* dot_wgs);   // slices left when the launch also carries d lambd (0: do not)
hipError_t launch_fbgrad(const FbGradParams& p, hipStream_t s);
int fbgrad_splits(int batch, int F, int M, int T);      // slices of the batch's K-blocks (= F x M partials) of one launch

// device-side refresh of every table that holds filterbank VALUES from an (F, M) fp32 device matrix, for tables built with the
// dense structure (all 4x16 blocks present): what a trainable filterbank needs after each optimizer step
struct RepackParams {
    uint4* ent_h; float* fb_nyq; int ks32;      // kTrainH tables (FwdParams::ent_h / fb_nyq) or nullptr
    float4* wl_b4; const int2* wl_lane; int wl_total4;   // kTrainW tables (FwdParams::wl_b4, its lane table, sum of wl_len4) or nullptr
    int wl_len4[kWlMaxPhases]; int wl_phases;
    int blocks_w;               // set by launch_repack: workgroups of the kTrainW section
    const float* fb;            // (F, M) row-major, device
    float* ent_b; float* ent_pre; const int4* tile_ranges;
    float* fb_dense;            // (F, M) copy for the kernels that read the matrix as it is
    float* fbT;                 // (M, F) transposed copy (long / big transforms) or nullptr
    float4* rowpk;              // (F) packed rows of XgradParams (dense structure: first column 0, M columns)
    int F, M, runs, nbpre, runs_group0;
    int blocks_h, blocks_c;     // set by launch_repack: workgroups of the fragment / copy sections
};
hipError_t launch_repack(const RepackParams& p, hipStream_t s);

// Adam on fp32 parameters of the layer (dmel_adam_step)
struct AdamParams {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq; float* step;
    unsigned* ticket;          // zero between launches; nullptr: one workgroup
    long long n; int maximize;
    double lr, beta1, beta2, eps, weight_decay;
};
int adam_grid(long long n);    // workgroups of one launch (more than one needs the ticket word)
hipError_t launch_adam(const AdamParams& p, hipStream_t s);

// Peer-to-peer all-reduce of the one scalar this path exchanges (d lambd), folded into the tail of the dot kernel: the workgroup
// that draws the last ticket stores (step, local sum) as ONE 8-byte granule into slot `rank` of every rank's inbox -- peer memory
// mapped over xGMI, system-scope stores -- then polls its own inbox until every rank's granule of this step has arrived and adds
// them in rank order (the same fp32 sum on every rank).  Two slots per source, by step parity: a rank cannot be two steps ahead
// of another (it needs the other's granule of step k + 1, written only after that rank has read step k).  Every spin is bounded
// by WALL-CLOCK time (s_memrealtime; default 120 s: a straggler -- a checkpoint, an evaluation pass, a data-loader stall on one
// rank -- is waited for, as RCCL would) and optionally by a poll count: a peer that never arrives yields NaN and raises a sticky
// pinned error word, which the next call on the plan reports (DMEL_ERR_MAILBOX_TIMEOUT), instead of hanging the device.
constexpr int kMailboxMaxWorld = 16;
struct MailboxArgs {
    unsigned long long* peer_inbox[kMailboxMaxWorld];   // inbox of every rank as seen from this device ([2][world] granules each)
    unsigned long long* my_inbox;
    unsigned* step;                   // device word: reduce steps done so far on this rank (identical on all ranks)
    unsigned long long* host_error;   // pinned: (step << 32) | rank that was missing, or nullptr
    int rank, world;
    unsigned spin_limit;              // polls per source before giving up; 0 = no limit on the count (the wall-clock bound below holds)
    unsigned long long timeout_ticks; // s_memrealtime ticks (100 MHz, constant) one exchange may wait in total; 0 = no wall-clock bound
};

// g: fp32, or bf16 when g_bf16 != 0 (the gradient of a bf16 output); t (the tangent) is always fp32
// mb != nullptr: the result is the SUM over all ranks of the mailbox (dmel_comm.cpp)
// fused != nullptr: the last workgroup applies that Adam update (n = 1) with the gradient it has just written (dmel_plan_attach_adam)
hipError_t launch_dot(const void* g, int g_bf16, const float* t, long long count, int accumulate, double* partials,
                      unsigned* counter, int max_partials, float* result, hipStream_t s, const MailboxArgs* mb = nullptr,
                      const AdamParams* fused = nullptr);
// the same exchange for a value that is already in memory (one wave): buf[0] = sum over ranks of buf[0]
hipError_t launch_mailbox_allreduce(float* buf, const MailboxArgs& mb, hipStream_t s);

}  // namespace dmel
