// dmel_big.hip -- the forward for transform lengths the LDS kernels do not reach:
//   * the optimized branch with |lambd| > 2730 samples: n_fft = 32768, 65536, ... (time_frequency.py:39 accepts any lambd);
//   * the constructor's default branch optimized=False on clips whose length is not a power of two (time_frequency.py:41,51:
//     window = whole clip, n_fft = 2 * n_points, e.g. Audio-MNIST's 8000 samples -> n_fft 16000, search_spaces.py:64) and
//     the DSPEC layer on such clips (models.py:171-200).
// One workgroup of 1024 threads transforms one complex sequence (a frame and its lambd-tangent, or two frames), held in LDS
// when it fits and in a per-workgroup slice of a global workspace otherwise (the radix-2 passes of dmel_ldsfft.h work on
// either: a workgroup's own stores are visible to it after __syncthreads()).  A length that is not a power of two goes through
// Bluestein's chirp-z identity  n k = (n^2 + k^2 - (k - n)^2) / 2:
//     X[k] = c[k] * sum_n (x[n] c[n]) conj(c)[k - n],   c[n] = exp(-i pi n^2 / N)
// i.e. one forward FFT of length M >= 2N - 1, a pointwise product with the precomputed transform of the chirp filter, and
// one inverse FFT (run as a forward FFT on the conjugate).  The pairing pass and the band-limited mel stage are those of the
// long-transform kernel (dmel_aux.hip).
// Split mode (BigParams::split): an even length N whose own FFT would have to live in global memory but whose HALF fits LDS is
// taken as one radix-2 decimation-in-frequency step done while loading,
//     X[2m]   = DFT_{N/2}( z[n] + z[n + N/2] )[m],      X[2m+1] = DFT_{N/2}( (z[n] - z[n + N/2]) exp(-2 pi i n / N) )[m],
// i.e. two half-length transforms (each a Bluestein round trip when N/2 is not a power of two) one after the other in the same
// LDS.  The mirror of an even bin is even and of an odd bin odd (N - (2m+1) = 2 (N/2 - 1 - m) + 1), so the pairing pass stays
// inside each half; the mel stage adds the two halves' contributions through a small accumulator behind the sequence.  This
// is what the reference's default branch gives on Audio-MNIST's 8000-sample clips (n_fft 16000 = two 8000-point DFTs through
// 16384-point FFTs in LDS instead of one through 32768-point FFTs in global memory).
#include "dmel_kernels.h"
#include "dmel_ldsfft.h"

namespace dmel {

constexpr int kBigThreads = 1024;

__device__ __forceinline__ float big_wave_sum(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);       // same tree on every lane and every run
    return v;
}

// One M-point FFT of the sequence Z.  In LDS: the radix-2 passes of dmel_ldsfft.h.  In global memory (M > 16384): only the stages
// whose span reaches across blocks of 2^lgb points (an even number of them, so lgb = 14 or 13) run on the global sequence; the
// 13 or 14 stages inside a block run on one block at a time in LDS (a block = an independent DIF / DIT transform of its own length).
// Every pass over the whole sequence in global memory moves M * 16 bytes through L2 -- nine of them per transform of 2^18
// points were what bound the 80 000-point frames of the reference's default branch on ESC-50 clips (340 ms per step); now two
// and one round trip of the blocks.
constexpr int kBigBlockLog = 14;
__device__ __forceinline__ int big_block_log(int logM) { return ((logM - kBigBlockLog) & 1) ? kBigBlockLog - 1 : kBigBlockLog; }

template <bool GLOBAL_Z, bool DIT, class TWG, class TWB>
__device__ __forceinline__ void big_fft(float2* Z, int M, int logM, float2* lds_block, int tid, TWG&& twid, TWB&& twid_block)
{
    if constexpr (!GLOBAL_Z) {
        if constexpr (DIT) lds_fft_dit<kBigThreads>(Z, M, logM, tid, twid); else lds_fft_dif<kBigThreads>(Z, M, logM, tid, twid);
    } else {
        const int lgb = big_block_log(logM), ns = logM - lgb, bl = 1 << lgb;
        if constexpr (!DIT) lds_fft_dif_head<kBigThreads>(Z, M, ns, tid, twid, [&](int n) { return Z[n]; }, M);
        for (int blk = 0; blk < (M >> lgb); ++blk) {
            float2* zb = Z + ((size_t)blk << lgb);
            for (int i = tid; i < bl; i += kBigThreads) lds_block[i] = zb[i];
            __syncthreads();
            if constexpr (DIT) lds_fft_dit<kBigThreads>(lds_block, bl, lgb, tid, twid_block); else lds_fft_dif<kBigThreads>(lds_block, bl, lgb, tid, twid_block);
            for (int i = tid; i < bl; i += kBigThreads) zb[i] = lds_block[i];
            __syncthreads();
        }
        if constexpr (DIT) lds_fft_dit_tail<kBigThreads>(Z, M, ns, tid, twid, [](int, float2 v) { return v; });
    }
}

// The middle of Bluestein's round trip -- forward transform, times the filter's transform (conjugated: the inverse transform runs
// as a forward one), second transform -- for a sequence in global memory: the block stages of the first transform END in LDS and
// those of the second START there on the same block, so the product is formed in LDS and the sequence makes one round trip per
// block instead of three.
// `first(n)` = input element n (chirp applied, zero beyond the data: the sequence is never written before the first pass reads
// it), `last(k, v)` = what output k becomes (conjugate, chirp): the two element-wise passes ride with the transforms' own.
template <class TWG, class TWB, class LD, class ST>
__device__ __forceinline__ void big_convolve_global(float2* Z, int M, int logM, const float2* __restrict__ hbr, float2* lds_block, int tid,
                                                    TWG&& twid, TWB&& twid_block, LD&& first, ST&& last)
{
    const int lgb = big_block_log(logM), ns = logM - lgb, bl = 1 << lgb;
    lds_fft_dif_head<kBigThreads>(Z, M, ns, tid, twid, first, M);
    for (int blk = 0; blk < (M >> lgb); ++blk) {
        float2* zb = Z + ((size_t)blk << lgb);
        const float2* hb = hbr + ((size_t)blk << lgb);
        for (int i = tid; i < bl; i += kBigThreads) lds_block[i] = zb[i];
        __syncthreads();
        lds_fft_dif<kBigThreads>(lds_block, bl, lgb, tid, twid_block);
        for (int i = tid; i < bl; i += kBigThreads) {
            const float2 v = c_mul(lds_block[i], hb[i]);
            lds_block[i] = make_float2(v.x, -v.y);
        }
        __syncthreads();
        lds_fft_dit<kBigThreads>(lds_block, bl, lgb, tid, twid_block);
        for (int i = tid; i < bl; i += kBigThreads) zb[i] = lds_block[i];
        __syncthreads();
    }
    lds_fft_dit_tail<kBigThreads>(Z, M, ns, tid, twid, last);
}

template <bool GLOBAL_Z>
__global__ void __launch_bounds__(kBigThreads) dmel_big_kernel(BigParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* Z = GLOBAL_Z ? p.zws + (size_t)blockIdx.x * p.Mfft : reinterpret_cast<float2*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, M = p.Mfft, F = p.F, sh = 32 - p.logM;
    const int NT = p.split ? N / 2 : N;                    // length of the transform(s) actually run
    float2* macc = reinterpret_cast<float2*>(smem_raw + (size_t)M * sizeof(float2));     // split mode: mel sums of the even half
    // LDS-resident sequences take their twiddles exp(-2 pi i k / M) as a product of two small LDS tables, k = 128 hi + lo (a
    // global-memory table cost an L2 round trip per butterfly group and pass; the full table does not fit next to the sequence)
    // (sequence in global memory: LDS holds one block of 2^lgb points and the factor tables of the BLOCK's twiddles, which are every
    // (M >> lgb)-th entry of the M-point table)
    const int lgb = big_block_log(p.logM), tsh = GLOBAL_Z ? p.logM - lgb : 0, tlen = GLOBAL_Z ? (1 << lgb) : M;
    float2* lds_block = reinterpret_cast<float2*>(smem_raw);
    float2* twa = reinterpret_cast<float2*>(smem_raw + (GLOBAL_Z ? ((size_t)sizeof(float2) << kBigBlockLog) : (size_t)M * sizeof(float2) + (p.split ? (size_t)p.M * sizeof(float2) : 0)));
    float2* twb = twa + 64;
    if (tid < 64) twa[tid] = (128 * tid < (tlen >> 1)) ? p.tw[(128 * tid) << tsh] : make_float2(1.f, 0.f);
    if (tid >= 64 && tid < 192) twb[tid - 64] = (tid - 64 < (tlen >> 1)) ? p.tw[(tid - 64) << tsh] : make_float2(1.f, 0.f);
    auto twid_block = [&](int k) -> float2 { return c_mul(twa[k >> 7], twb[k & 127]); };
    auto twid = [&](int k) -> float2 {
        if (GLOBAL_Z) return p.tw[k];
        return c_mul(twa[k >> 7], twb[k & 127]);
    };
    const bool blue = p.chirp != nullptr;
    const bool pair = (p.mode == kInfer || p.mode == kSpec);
    const bool spec_mode = (p.mode == kSpec || p.mode == kSpecTrain);
    const int tiles = pair ? (p.T + 1) / 2 : p.T;
    const long long units = (long long)p.B * tiles;

    const LamState ls = lam_prologue(p.lam, N, blockIdx.x == 0 && tid == 0);
    if (ls.action != kLamRun) {
        if (ls.action == kLamPoison) {           // no launch of this forward matched the device lambd (dmel_kernels.h)
            const int rows = spec_mode ? F : p.M;
            const long long total = (long long)p.B * rows * p.T;
            for (long long o = (long long)blockIdx.x * kBigThreads + tid; o < total; o += (long long)gridDim.x * kBigThreads) {
                if (p.flags & 4u) reinterpret_cast<unsigned short*>(p.out)[o] = 0x7fc0u; else p.out[o] = __builtin_nanf("");
                if (p.tangent) p.tangent[o] = __builtin_nanf("");
            }
        }
        return;
    }
    const float htan = 0.5f * lam_tangent_scale(ls);
    // position of bin k inside Z: natural order after the Bluestein round trip, bit-reversed after a single DIF transform
    auto pos = [&](int k) -> unsigned { return blue ? (unsigned)k : (M > 1 ? __brev((unsigned)k) >> sh : 0u); };

    for (long long unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const int b = (int)(unit / tiles), tile = (int)(unit % tiles);
        const int tA = pair ? 2 * tile : tile, tB = tA + 1;
        const float* xb = resolve_x(p.x, p.x_ind) + (size_t)b * p.L;
        float mean = 0.f;
        if (p.remove_dc) {
            mean = clip_mean_psum(p.psum, p.nchunks, b, p.L);
        }
        // element n of the windowed complex frame (frame + tangent, or two frames), before any transform
        auto elem = [&](int n) -> float2 {
            const long long ia = (long long)tA * p.hop - N / 2 + n;
            const float va = (ia >= 0 && ia < p.L) ? (xb[ia] - mean) : 0.f;      // zero padding after the DC removal
            const float2 wd = p.win2[n];
            if (pair) {
                const long long ib = ia + p.hop;
                const float vb = (tB < p.T && ib >= 0 && ib < p.L) ? (xb[ib] - mean) : 0.f;
                return make_float2(va * wd.x, vb * wd.x);
            }
            return make_float2(va * wd.x, va * wd.y);
        };
        const int npar = p.split ? 2 : 1;
        for (int par = 0; par < npar; ++par) {
        __syncthreads();                                   // the previous unit's / half's readers are done with Z
        if (GLOBAL_Z && blue) {
            // (no split mode here: its halves fit LDS by definition) the frame is windowed, chirped and zero-extended as the first
            // pass of the first transform reads it, and the last pass of the second writes conj(.) * chirp
            big_convolve_global(Z, M, p.logM, p.hbr, lds_block, tid, twid, twid_block,
                                [&](int n) -> float2 { return n < N ? c_mul(elem(n), p.chirp[n]) : make_float2(0.f, 0.f); },
                                [&](int k, float2 v) -> float2 { return k < N ? c_mul(make_float2(v.x, -v.y), p.chirp[k]) : v; });
        } else {
        for (int n = tid; n < M; n += kBigThreads) {
            float2 z = make_float2(0.f, 0.f);
            if (n < NT) {
                if (!p.split) z = elem(n);
                else {
                    const float2 za = elem(n), zb = elem(n + NT);
                    z = par == 0 ? make_float2(za.x + zb.x, za.y + zb.y) : c_mul(make_float2(za.x - zb.x, za.y - zb.y), p.wodd[n]);
                }
                if (blue) z = c_mul(z, p.chirp[n]);
            }
            Z[n] = z;
        }
        __syncthreads();
        big_fft<GLOBAL_Z, false>(Z, M, p.logM, lds_block, tid, twid, twid_block);
        }
        if (blue && !GLOBAL_Z) {
            {
                // Y H / M at the bit-reversed positions the DIF left, conjugated: a forward DIT of that is the conjugate of the
                // inverse transform, in natural order
                for (int i = tid; i < M; i += kBigThreads) {
                    const float2 v = c_mul(Z[i], p.hbr[i]);
                    Z[i] = make_float2(v.x, -v.y);
                }
                __syncthreads();
                big_fft<GLOBAL_Z, true>(Z, M, p.logM, lds_block, tid, twid, twid_block);
            }
            for (int k = tid; k < NT; k += kBigThreads) {
                const float2 v = Z[k];
                Z[k] = c_mul(make_float2(v.x, -v.y), p.chirp[k]);
            }
            __syncthreads();
        }
        // pairing pass (see dmel_fwd.hip): PD = (|S|^2, Im(conj S * D)) or (|S|^2, |D|^2), in place at pos(m) for the output bin
        // k = m (whole transform) or k = 2 m + par (split); the two addresses a thread touches belong to no other thread
        const int mlim = !p.split ? (N >> 1) : (par == 0 ? NT / 2 : (NT - 1) / 2);
        for (int m = tid; m <= mlim; m += kBigThreads) {
            const int mm = !p.split ? (N - m) % N : (par == 0 ? (NT - m) % NT : NT - 1 - m);
            const unsigned ak = pos(m), an = pos(mm);
            const float2 zk = Z[ak], zn = Z[an];
            const float sx = zk.x + zn.x, sy = zk.y - zn.y, dx = zk.x - zn.x, dy = zk.y + zn.y;
            Z[ak] = pair ? make_float2(fmaf(sx, sx, sy * sy), fmaf(dx, dx, dy * dy))
                         : make_float2(fmaf(sx, sx, sy * sy), fmaf(sx, dy, -(sy * dx)));
        }
        __syncthreads();
        const int kstep = p.split ? 2 : 1;                  // this pass holds the bins k = par, par + kstep, ...
        auto pd_of = [&](int k) -> float2 { return Z[pos(p.split ? (k >> 1) : k)]; };
        if (spec_mode) {
            for (int k = par + kstep * tid; k < F; k += kstep * kBigThreads) {
                const float2 pd = pd_of(k);
                const size_t o = ((size_t)b * F + k) * p.T;
                if (p.mode == kSpec) {
                    p.out[o + tA] = 0.25f * pd.x;
                    if (tB < p.T) p.out[o + tB] = 0.25f * pd.y;
                } else {
                    p.out[o + tA] = 0.25f * pd.x;
                    if (p.tangent) p.tangent[o + tA] = htan * pd.y;
                }
            }
            continue;
        }
        const bool do_log = (p.flags & 1u) != 0;
        for (int m = wave; m < p.M; m += kBigThreads / 64) {
            const int2 bd = p.band[m];
            const float* fr = p.fbT + (size_t)m * F;
            float s0 = 0.f, s1 = 0.f;
            const int k0 = p.split ? bd.x + ((bd.x ^ par) & 1) : bd.x;        // first bin of the band this pass holds
            for (int k = k0 + kstep * lane; k < bd.y; k += kstep * 64) {
                const float c = fr[k];
                const float2 pd = pd_of(k);
                s0 = fmaf(c, pd.x, s0);
                s1 = fmaf(c, pd.y, s1);
            }
            s0 = big_wave_sum(s0);
            s1 = big_wave_sum(s1);
            if (lane != 0) continue;
            if (p.split) {
                // the same wave owns band m in both halves: even bins first, then odd bins on top of them
                if (par == 0) { macc[m] = make_float2(s0, s1); continue; }
                const float2 e = macc[m];
                s0 += e.x; s1 += e.y;
            }
            const size_t o = ((size_t)b * p.M + m) * p.T;
            const bool out_bf16 = (p.flags & 4u) != 0;
            auto put = [&](int t, float v) { if (out_bf16) reinterpret_cast<unsigned short*>(p.out)[o + t] = bf16_bits(v); else p.out[o + t] = v; };
            if (pair) {
                const float ma = 0.25f * s0, mb = 0.25f * s1;
                put(tA, do_log ? logf(ma + p.eps) : ma);
                if (tB < p.T) put(tB, do_log ? logf(mb + p.eps) : mb);
            } else {
                const float mel = 0.25f * s0, dmel = htan * s1;
                if (do_log) {
                    const float me = mel + p.eps;
                    put(tA, logf(me));
                    if (p.tangent) p.tangent[o + tA] = dmel / me;
                } else {
                    put(tA, mel);
                    if (p.tangent) p.tangent[o + tA] = dmel;
                }
            }
        }
        }   // par
    }
}

// ---- gradient w.r.t. the waveform on the same transforms (dmel_backward_x off the power-of-two lengths <= 16384) ---------------
// The adjoint written out in dmel_xgrad.hip, one workgroup per PAIR of frames, with the DFT of this file in both directions:
//   Z = DFT_N(x~_a w + i x~_b w);  per bin  H = c gP X  for both frames, conj(H_a + i H_b) Hermitian-extended in place;
//   R = DFT_N(that) = conj(dv_a + i dv_b);  windowed frame gradients to the (B, T, N) workspace with one fp64 sum per frame
// (dmel_xgrad_gather_kernel overlap-adds them in frame order and removes the mean).  A power-of-two N (> 16384) leaves the first
// transform bit-reversed, which is the input order of the second (DIF then DIT, no permutation); Bluestein's round trip returns
// natural order, so the second transform is simply another round trip.  A correctness path: the reference never asks for this
// gradient, torch autograd would provide it (time_frequency.py:41,51 / models.py:171-200 with x.requires_grad).
template <bool GLOBAL_Z>
__global__ void __launch_bounds__(kBigThreads) dmel_xgrad_big_kernel(XgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[2][kBigThreads / 64];
    float2* Z = GLOBAL_Z ? p.zws + (size_t)blockIdx.x * p.Mfft : reinterpret_cast<float2*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, M = p.Mfft, sh = 32 - p.logM, T = p.T, Mm = p.M;
    const int lgb = big_block_log(p.logM), tsh = GLOBAL_Z ? p.logM - lgb : 0, tlen = GLOBAL_Z ? (1 << lgb) : M;
    float2* lds_block = reinterpret_cast<float2*>(smem_raw);
    float2* twa = reinterpret_cast<float2*>(smem_raw + (GLOBAL_Z ? ((size_t)sizeof(float2) << kBigBlockLog) : (size_t)M * sizeof(float2)));
    float2* twb = twa + 64;
    if (tid < 64) twa[tid] = (128 * tid < (tlen >> 1)) ? p.tw[(128 * tid) << tsh] : make_float2(1.f, 0.f);
    if (tid >= 64 && tid < 192) twb[tid - 64] = (tid - 64 < (tlen >> 1)) ? p.tw[(tid - 64) << tsh] : make_float2(1.f, 0.f);
    auto twid_block = [&](int k) -> float2 { return c_mul(twa[k >> 7], twb[k & 127]); };
    auto twid = [&](int k) -> float2 {
        if (GLOBAL_Z) return p.tw[k];
        return c_mul(twa[k >> 7], twb[k & 127]);
    };
    const bool blue = p.chirp != nullptr;
    auto pos = [&](int k) -> unsigned { return blue ? (unsigned)k : (M > 1 ? __brev((unsigned)k) >> sh : 0u); };
    // DFT_N of the N values in Z[0 .. N) (natural order) by Bluestein's round trip, result in natural order
    auto bluestein = [&]() {
        if constexpr (GLOBAL_Z) {
            // chirp and zero extension ride with the first pass of the first transform, conj(.) * chirp with the last of the second
            big_convolve_global(Z, M, p.logM, p.hbr, lds_block, tid, twid, twid_block,
                                [&](int n) -> float2 { return n < N ? c_mul(Z[n], p.chirp[n]) : make_float2(0.f, 0.f); },
                                [&](int k, float2 v) -> float2 { return k < N ? c_mul(make_float2(v.x, -v.y), p.chirp[k]) : v; });
        } else {
            for (int n = tid; n < M; n += kBigThreads) Z[n] = n < N ? c_mul(Z[n], p.chirp[n]) : make_float2(0.f, 0.f);
            __syncthreads();
            big_fft<GLOBAL_Z, false>(Z, M, p.logM, lds_block, tid, twid, twid_block);
            for (int i = tid; i < M; i += kBigThreads) {
                const float2 v = c_mul(Z[i], p.hbr[i]);
                Z[i] = make_float2(v.x, -v.y);
            }
            __syncthreads();
            big_fft<GLOBAL_Z, true>(Z, M, p.logM, lds_block, tid, twid, twid_block);
            for (int k = tid; k < N; k += kBigThreads) {
                const float2 v = Z[k];
                Z[k] = c_mul(make_float2(v.x, -v.y), p.chirp[k]);
            }
            __syncthreads();
        }
    };
    const int tiles = (T + 1) / 2;
    const long long units = (long long)p.B * tiles;
    for (long long unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const int b = (int)(unit / tiles), tile = (int)(unit % tiles);
        const int tA = 2 * tile, tB = tA + 1;
        const bool hasB = tB < T;
        const float* xb = p.x + (size_t)b * p.L;
        float mean = 0.f;
        {
            mean = clip_mean_psum(p.psum, p.nchunks, b, p.L);
        }
        __syncthreads();                                   // the previous unit's readers are done with Z
        for (int n = tid; n < (blue ? N : M); n += kBigThreads) {
            const long long ia = (long long)tA * p.hop - N / 2 + n, ib = ia + p.hop;
            const float va = (ia >= 0 && ia < p.L) ? (xb[ia] - mean) : 0.f;
            const float vb = (hasB && ib >= 0 && ib < p.L) ? (xb[ib] - mean) : 0.f;
            const float w = p.win2[n].x;
            Z[n] = make_float2(va * w, vb * w);
        }
        __syncthreads();
        if (blue) bluestein(); else big_fft<GLOBAL_Z, false>(Z, M, p.logM, lds_block, tid, twid, twid_block);
        // bin pass: every thread owns bin k and its mirror image (two addresses nobody else touches)
        const float* ga = p.grad_out + (size_t)b * Mm * T + tA;
        const float* ya = p.out ? p.out + (size_t)b * Mm * T + tA : nullptr;
        for (int k = tid; k <= (N >> 1); k += kBigThreads) {
            const unsigned ak = pos(k), an = pos((N - k) % N);
            const float2 zk = Z[ak], zn = Z[an];
            const float xar = 0.5f * (zk.x + zn.x), xai = 0.5f * (zk.y - zn.y);
            const float xbr = 0.5f * (zk.y + zn.y), xbi = -0.5f * (zk.x - zn.x);
            float gpa = 0.f, gpb = 0.f;
            if (p.spec_mode) {
                const float* gs = p.grad_out + ((size_t)b * p.F + k) * T + tA;
                gpa = gs[0];
                gpb = hasB ? gs[1] : 0.f;
            } else {
                const int2 band = p.rowband[k];
                for (int m = band.x; m < band.y; ++m) {
                    const float c = p.fb[(size_t)k * Mm + m];
                    float g0 = ga[(size_t)m * T], g1 = hasB ? ga[(size_t)m * T + 1] : 0.f;
                    if (ya) { g0 *= expf(-ya[(size_t)m * T]); if (hasB) g1 *= expf(-ya[(size_t)m * T + 1]); }
                    gpa = fmaf(c, g0, gpa);
                    gpb = fmaf(c, g1, gpb);
                }
            }
            const bool edge = (k == 0) || (2 * k == N);
            const float sc = edge ? 2.f : 1.f;
            const float har = sc * gpa * xar, hai = edge ? 0.f : gpa * xai;
            const float hbr = sc * gpb * xbr, hbi = edge ? 0.f : gpb * xbi;
            Z[ak] = make_float2(har - hbi, -(hai + hbr));
            if (!edge) Z[an] = make_float2(har + hbi, hai - hbr);
        }
        __syncthreads();
        if (blue) bluestein(); else big_fft<GLOBAL_Z, true>(Z, M, p.logM, lds_block, tid, twid, twid_block);
        float* fa = p.frames + ((size_t)b * T + tA) * N;
        double sa = 0.0, sb = 0.0;
        for (int n = tid; n < N; n += kBigThreads) {
            const float2 r = Z[n];
            const float w = p.win2[n].x;
            const float va = r.x * w, vb = -r.y * w;
            fa[n] = va;
            if (hasB) fa[(size_t)N + n] = vb;
            const long long ia = (long long)tA * p.hop - N / 2 + n, ib = ia + p.hop;
            if (ia >= 0 && ia < p.L) sa += (double)va;
            if (hasB && ib >= 0 && ib < p.L) sb += (double)vb;
        }
        // fixed order: lanes by the shuffle tree, waves ascending
        for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
        if (lane == 0) { red[0][wave] = sa; red[1][wave] = sb; }
        __syncthreads();
        if (tid == 0) {
            double ta = 0.0, tb2 = 0.0;
            for (int w = 0; w < kBigThreads / 64; ++w) { ta += red[0][w]; tb2 += red[1][w]; }
            p.csum[(size_t)b * T + tA] = ta;
            if (hasB) p.csum[(size_t)b * T + tB] = tb2;
        }
    }
}

constexpr int kBigLdsMax = 16384;          // complex entries: 128 KB
constexpr int kBigSplitAccBytes = 16384;   // split mode: (n_mels) float2 sums of the even half behind the sequence
constexpr int kBigTwBytes = (64 + 128) * 8; // the two twiddle factor tables behind that
constexpr int kBigGlobalLds = (int)(sizeof(float2) << kBigBlockLog) + kBigTwBytes;   // sequence in global memory: one block + the block's tables

// split mode is possible when the half-length transform (its own FFT of m_half points) fits LDS together with the mel sums
bool big_can_split(int m_half, int n_mels) { return m_half <= kBigLdsMax && (long long)n_mels * 8 <= kBigSplitAccBytes; }

hipError_t big_prepare_attributes()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmel_big_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kBigLdsMax * (int)sizeof(float2) + kBigSplitAccBytes + kBigTwBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmel_xgrad_big_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kBigLdsMax * (int)sizeof(float2) + kBigTwBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmel_big_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kBigGlobalLds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(dmel_xgrad_big_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kBigGlobalLds);
}

bool big_uses_global(int m_fft) { return m_fft > kBigLdsMax; }

// frames kernel of dmel_backward_x on this file's transforms; launch_xgrad_gather (dmel_xgrad.hip) finishes the job
hipError_t launch_xgrad_big(const XgradParams& p, int grid, hipStream_t s)
{
    if (big_uses_global(p.Mfft)) hipLaunchKernelGGL(dmel_xgrad_big_kernel<true>, dim3((unsigned)grid), dim3(kBigThreads), kBigGlobalLds, s, p);
    else hipLaunchKernelGGL(dmel_xgrad_big_kernel<false>, dim3((unsigned)grid), dim3(kBigThreads), (size_t)p.Mfft * sizeof(float2) + kBigTwBytes, s, p);
    return hipGetLastError();
}

// workgroups of one launch: every unit its own workgroup while the sequence fits LDS, a bounded number of persistent ones
// (each owns m_fft complex words of the workspace) otherwise
int big_grid(long long units, int m_fft)
{
    const long long cap = big_uses_global(m_fft) ? (m_fft > 65536 ? 256 : 1024) : 0x7fffffffLL;
    return (int)(units < cap ? (units > 0 ? units : 1) : cap);
}

hipError_t launch_big(const BigParams& p, hipStream_t s)
{
    const bool pair = (p.mode == kInfer || p.mode == kSpec);
    const long long units = (long long)p.B * (pair ? (p.T + 1) / 2 : p.T);
    const int grid = big_grid(units, p.Mfft);
    if (big_uses_global(p.Mfft)) hipLaunchKernelGGL(dmel_big_kernel<true>, dim3((unsigned)grid), dim3(kBigThreads), kBigGlobalLds, s, p);
    else hipLaunchKernelGGL(dmel_big_kernel<false>, dim3((unsigned)grid), dim3(kBigThreads),
                            (size_t)p.Mfft * sizeof(float2) + (p.split ? (size_t)p.M * sizeof(float2) : 0) + kBigTwBytes, s, p);
    return hipGetLastError();
}

}  // namespace dmel
