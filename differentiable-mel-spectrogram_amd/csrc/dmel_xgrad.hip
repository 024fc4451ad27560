// dmel_xgrad.hip -- gradient w.r.t. the waveform (optional output of the layer's backward; the reference never asks for
// it, torch autograd would return it for x.requires_grad).  Adjoint of models.py:38 (DC removal),
// time_frequency.py:43-53 (zero padding, framing, window, rfft, |.|^2) and models.py:53 (mel contraction):
//
//   gP[k][t]  = sum_m fb[k][m] gm[m][t]                 gm = grad_out, or grad_out * exp(-out) for the log output
//   dv_t[n]   = sum_{k=0}^{N-1} H_k e^{+2 pi i k n / N}   H_k = c_k gP[k] X_t[k]  (c = 2 at k = 0 and N/2, else 1), Hermitian
//   dx~[i]    = sum over the frames t that cover i of dv_t[i - t hop + N/2] w[i - t hop + N/2]
//   dx        = dx~ - mean(dx~)
//
// Kernel 1 (one workgroup per pair of frames, any power-of-two n_fft up to 16384): both frames go through ONE complex
// FFT held in LDS (in-place decimation in frequency, radix-2 stages fused in pairs, spectrum in bit-reversed order), the two spectra are
// separated, scaled by gP and written back -- conjugated, Hermitian-extended, packed as conj(H_a) + i conj(H_b)... -- at
// the same bit-reversed addresses, which is exactly the input order of an in-place decimation-in-time FFT; its output
// is conj(dv_a + i dv_b) in natural order.  No permutation pass, no second buffer.  The windowed frame gradients go to a
// (B, T, N) workspace.
// Kernel 2 (one workgroup per 4096 samples): overlap-add as a gather in increasing frame order (deterministic, no atomics) minus
// the mean of the clip's gradient, which comes from fp64 per-frame sums left by kernel 1 (a third kernel used to subtract it).
// A correctness-first path: about 10x the time of the fused forward at config 2.
#include "dmel_kernels.h"
#include "dmel_ldsfft.h"
#include "dmel_wavefft.h"

namespace dmel {

constexpr int kXgThreads = 256;

// DMEL_FLAG_CHECK_NFFT (the optimized=True layer with x.requires_grad and lambd left on the device): the step's forward ran one
// launch per candidate n_fft and only the one lambd asks for did the work (lam_prologue); the backward does the same
__device__ __forceinline__ bool xgrad_not_this_nfft(const XgradParams& p)
{
    return p.check_nfft && p.lam_dev && lam_n_fft(__builtin_fabsf(*(const __attribute__((address_space(4))) float*)p.lam_dev)) != p.N;
}

template <bool TWLDS>
__global__ void __launch_bounds__(kXgThreads) dmel_xgrad_frames_kernel(XgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* Z = reinterpret_cast<float2*>(smem_raw);
    if (xgrad_not_this_nfft(p)) return;
    // twiddle table in LDS behind the sequence when it fits (n_fft <= 8192): every butterfly stage would otherwise wait for
    // a global (L1) load per twiddle, twenty-odd dependent round trips per workgroup
    float2* twl = Z + p.N;                                     // TWLDS only
    auto twiddle = [&](int k) -> float2 { if constexpr (TWLDS) return twl[k]; else return p.tw[k]; };
    const int tid = threadIdx.x;
    const int N = p.N, M = p.M, T = p.T, sh = 32 - p.logN;
    const int tiles = (T + 1) / 2;
    const int b = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int tA = 2 * tile, tB = tA + 1;
    const bool hasB = tB < T;
    const float* xb = p.x + (size_t)b * p.L;
    float mean = 0.f;
    {
        mean = clip_mean_psum(p.psum, p.nchunks, b, p.L);
    }
    for (int n = tid; n < N; n += kXgThreads) {
        const long long ia = (long long)tA * p.hop - N / 2 + n, ib = ia + p.hop;
        const float va = (ia >= 0 && ia < p.L) ? (xb[ia] - mean) : 0.f;
        const float vb = (hasB && ib >= 0 && ib < p.L) ? (xb[ib] - mean) : 0.f;
        const float w = p.win2[n].x;
        Z[n] = make_float2(va * w, vb * w);
    }
    if constexpr (TWLDS) for (int k = tid; k < (N >> 1); k += kXgThreads) twl[k] = p.tw[k];
    __syncthreads();
    // forward: decimation in frequency, natural order in, bit-reversed order out
    lds_fft_dif<kXgThreads, false>(Z, N, p.logN, tid, twiddle);
    // spectra of the two frames, gradient of the power spectrum, conj(H_a) + i conj(H_b) back in place
    const float* ga = p.grad_out + (size_t)b * M * T + tA;
    const float* ya = p.out ? p.out + (size_t)b * M * T + tA : nullptr;
    for (int k = tid; k <= (N >> 1); k += kXgThreads) {
        const unsigned ak = N > 1 ? __brev((unsigned)k) >> sh : 0u, an = N > 1 ? __brev((unsigned)((N - k) & (N - 1))) >> sh : 0u;
        const float2 zk = Z[ak], zn = Z[an];
        // X_a = (Z_k + conj Z_{N-k}) / 2,  X_b = (Z_k - conj Z_{N-k}) / (2i)
        const float xar = 0.5f * (zk.x + zn.x), xai = 0.5f * (zk.y - zn.y);
        const float xbr = 0.5f * (zk.y + zn.y), xbi = -0.5f * (zk.x - zn.x);
        float gpa = 0.f, gpb = 0.f;
        if (p.spec_mode) {
            // DSPEC (models.py:171-200): the layer's output IS the power spectrogram, its gradient arrives per bin
            const float* gs = p.grad_out + ((size_t)b * p.F + k) * T + tA;
            gpa = gs[0];
            gpb = hasB ? gs[1] : 0.f;
        }
        const int2 band = p.spec_mode ? make_int2(0, 0) : p.rowband[k];
        for (int m = band.x; m < band.y; ++m) {
            const float c = p.fb[(size_t)k * M + m];
            float g0 = ga[(size_t)m * T], g1 = hasB ? ga[(size_t)m * T + 1] : 0.f;
            if (ya) { g0 *= expf(-ya[(size_t)m * T]); if (hasB) g1 *= expf(-ya[(size_t)m * T + 1]); }
            gpa = fmaf(c, g0, gpa);
            gpb = fmaf(c, g1, gpb);
        }
        const bool edge = (k == 0) || (2 * k == N);
        const float sc = edge ? 2.f : 1.f;
        const float har = sc * gpa * xar, hai = edge ? 0.f : gpa * xai;      // X is real at k = 0 and N/2
        const float hbr = sc * gpb * xbr, hbi = edge ? 0.f : gpb * xbi;
        // U_k = conj(H_a,k + i H_b,k) = (har + hbi) + i (-(hai) + ... ): conj(a + i b) with a = har + i hai, b = hbr + i hbi
        //     = conj(har - hbi + i (hai + hbr)) = (har - hbi) - i (hai + hbr)
        Z[ak] = make_float2(har - hbi, -(hai + hbr));
        // k' = N - k carries conj(H_a,k) + i conj(H_b,k) = (har + hbi) + i (hbr - hai); conjugated: (har + hbi) - i (hbr - hai)
        if (!edge) Z[an] = make_float2(har + hbi, hai - hbr);
    }
    __syncthreads();
    // decimation in time, bit-reversed order in, natural order out: R = FFT(conj W) = conj(dv_a + i dv_b)
    lds_fft_dit<kXgThreads, false>(Z, N, p.logN, tid, twiddle);
    float* fa = p.frames + ((size_t)b * T + tA) * N;
    double sa = 0.0, sb = 0.0;                                 // what each frame contributes to the sum of the clip's gradient
    for (int n = tid; n < N; n += kXgThreads) {
        const float2 r = Z[n];
        const float w = p.win2[n].x;
        const float va = r.x * w, vb = -r.y * w;
        fa[n] = va;
        if (hasB) fa[N + n] = vb;
        const long long ia = (long long)tA * p.hop - N / 2 + n, ib = ia + p.hop;
        if (ia >= 0 && ia < p.L) sa += (double)va;
        if (hasB && ib >= 0 && ib < p.L) sb += (double)vb;
    }
    // fixed-order tree over the 256 threads (the sequence is dead: its first 4 KB hold the partials)
    __syncthreads();
    double* red = reinterpret_cast<double*>(smem_raw);
    red[tid] = sa; red[kXgThreads + tid] = sb;
    __syncthreads();
    for (int o = kXgThreads / 2; o > 0; o >>= 1) {
        if (tid < o) { red[tid] += red[tid + o]; red[kXgThreads + tid] += red[kXgThreads + tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { p.csum[(size_t)b * T + tA] = red[0]; if (hasB) p.csum[(size_t)b * T + tB] = red[kXgThreads]; }
}

// grid (chunks, B): every workgroup overlap-adds one chunk of kXgChunk samples of one clip as a gather in increasing frame
// order (deterministic, no atomics) and subtracts the mean of the clip's gradient (models.py:38 removes the clip's DC, so the
// gradient has none either).  The mean comes from the per-frame sums the first kernel left: every workgroup adds them up in the
// same fixed order (strided partial sums, then a tree), so every chunk of a clip subtracts the same bits.
constexpr int kXgChunk = 4096;

__global__ void __launch_bounds__(256) dmel_xgrad_gather_kernel(XgradParams p)
{
    __shared__ double red[256];
    if (xgrad_not_this_nfft(p)) return;
    const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const int N = p.N, T = p.T, hop = p.hop, half = N / 2;
    float mean = 0.f;
    if (p.remove_dc) {
        double acc = 0.0;
        for (int t = tid; t < T; t += 256) acc += p.csum[(size_t)b * T + t];
        red[tid] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        mean = (float)(red[0] / (double)p.L);
    }
    const float* fr = p.frames + (size_t)b * T * N;
    float* gx = p.grad_x + (size_t)b * p.L;
    const int lo = chunk * kXgChunk, hi = min(lo + kXgChunk, p.L);
    for (int i = lo + tid; i < hi; i += 256) {
        // frames with 0 <= i - t hop + N/2 < N, in increasing t
        int t_lo = i + half - N + 1;
        t_lo = t_lo <= 0 ? 0 : (t_lo + hop - 1) / hop;
        int t_hi = (i + half) / hop;
        if (t_hi > T - 1) t_hi = T - 1;
        float s = 0.f;
        for (int t = t_lo; t <= t_hi; ++t) s += fr[(size_t)t * N + (i - t * hop + half)];
        gx[i] = s - mean;
    }
}

// ---- the same gradient on the forward's wave FFT (n_fft 32 ... 2048) ---------------------------------------------------------
// One wave (or G = N / R lanes of it) per PAIR of neighbouring frames, as in the forward's inference mode: the pair rides in one
// complex FFT held in registers (wave_fft, dmel_wavefft.h), the spectrum lands in the pair's LDS slot in natural order, the
// lanes separate the two spectra bin by bin, contract the mel gradient with the filterbank rows (gm of the tile's frames sits
// in LDS; an HTK row has two non-zero columns), write conj(H_a + i H_b) -- Hermitian-extended -- back over the spectrum, and a
// second wave_fft of those N points is conj(dv_a + i dv_b).  Windowed, the two frame gradients stay in the slot: after one
// workgroup barrier the tile's frames are overlap-added as a gather in increasing frame order (deterministic) and only the
// tile's SEGMENT of (FPT - 1) hop + N samples goes to memory -- a quarter of the (B, T, N) frame workspace at BASELINE config 2 --
// together with its fp64 sum over the samples inside the clip.  dmel_xgrad_combine_kernel adds the (at most few) segments that
// cover a sample, in increasing tile order, and subtracts the mean.
#ifdef DMEL_STAMPS
// diagnostic build only (tools/xstamps.py): s_memtime of every wave at the phase boundaries of the wave-FFT kernel
constexpr int kXStampSlots = 16;
__device__ unsigned long long g_xstamps[4096 * 8 * kXStampSlots];
__device__ __forceinline__ void xstamp(int wgid, int wave, int lane, int idx)
{
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0 && wgid < 4096) g_xstamps[((size_t)wgid * 8 + wave) * kXStampSlots + idx] = t;
}
#define XSTAMP(i) xstamp(blockIdx.x, wave, lane, i)
#else
#define XSTAMP(i) do {} while (0)
#endif

template <int N> struct XgPlan {
    using P = FftPlanSel<N, true>;
    static constexpr int R = P::R, C = P::C, G = N / R, FPW = 64 / G;
    // (2048 with 8 waves -- one workgroup per CU, 16-frame tiles: kernel 81 us against 77 at BASELINE config 3, combine pass 8.8 against 10.1)
    static constexpr int WAVES = (N == 1024) ? 8 : 4;
    static constexpr int LDS_MAX = 80 * 1024;                          // two workgroups per CU
    static constexpr int SLOTS = WAVES * FPW, FPT = 2 * SLOTS, THREADS = 64 * WAVES;
    static constexpr int SS = slot_stride_f2(N, R, C, 0, 0);          // float2 entries per slot
    static constexpr int MINW = (N >= 2048) ? 2 : 4;                   // waves per SIMD the LDS footprint admits
    static constexpr int PADP = (R == 16 && C > 1) ? 16 : 0;           // the C groups of 16 lanes write a plane on different banks
};

template <int N>
__global__ void __launch_bounds__((XgPlan<N>::THREADS), (XgPlan<N>::MINW)) dmel_xgrad_wave_kernel(XgradParams p)
{
    using PL = XgPlan<N>;
    constexpr int R = PL::R, C = PL::C, G = PL::G, FPW = PL::FPW, SLOTS = PL::SLOTS, FPT = PL::FPT;
    constexpr int THREADS = PL::THREADS, SS = PL::SS, RR = R * R;
    constexpr int NPAIR = N / (2 * G) + 1;                       // bins lg + G i <= N/2
    constexpr int PADC = (C > 1) ? 4 : 0;                         // spectrum index k + PADC (k / R^2), as the forward (z_index)
    constexpr int PADP = PL::PADP;                                // frame-gradient planes: n + PADP (n / R^2)
    static_assert(G <= 64 && N == R * R * C && RR % G == 0, "one wave (or a part of it) per frame pair");
    static_assert(N + (C - 1) * PADP <= SS, "a plane of frame gradients fits half a slot");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (xgrad_not_this_nfft(p)) return;
    v2f* lds = reinterpret_cast<v2f*>(smem_raw);
    float* win = reinterpret_cast<float*>(smem_raw + SLOTS * SS * 8);
    // the window table: entries 0 .. N/2 when it is symmetric about N/2 (the Gaussian of the optimized=True branch), else all N
    const int WN = p.win_n;
    const bool sym = WN < N;
    float* gm = win + ((WN + 3) & ~3);                            // (M, FPT); later the fp64 partial sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wg = blockIdx.x;
    {   // contiguous tiles per XCD (neighbouring tiles share samples: L2 hits), as the forward
        const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = wg & 7;
        wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (wg >> 3);
    }
    XSTAMP(0);
    const int tiles = p.tiles, M = p.M, T = p.T, hop = p.hop, L = p.L;
    const int b = wg / tiles, tile = wg % tiles;
    const int t0 = tile * FPT;
    const int j = lane / G, lg = lane % G;
    const int slot = wave * FPW + j;
    v2f* sl = lds + slot * SS;
    const int slot_b = slot * (SS * 8);
    const int tA = t0 + 2 * slot;
    const int f0 = tA * hop - N / 2;                              // first sample of frame tA; frame tA + 1 starts hop later
    // samples of both frames: plain offsets when the pair lies inside the clip, else clamped (and zeroed at windowing time)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)b * L, (unsigned)L * 4u);
    const bool inside = __all((f0 >= 0) && (f0 + hop + N <= L));
    float xa[R], xc[R];
    if (inside) {
        static_for<0, R>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            xa[a] = buf_f32(rx, (f0 + lg + G * a) * 4);
            xc[a] = buf_f32(rx, (f0 + hop + lg + G * a) * 4);
        });
    } else {
        static_for<0, R>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            xa[a] = buf_f32(rx, clampi(f0 + lg + G * a, L - 1) * 4);
            xc[a] = buf_f32(rx, clampi(f0 + hop + lg + G * a, L - 1) * 4);
        });
    }
    // everything else the prologue needs is requested before anything is waited for: the first batch of gm (four words per
    // thread: the whole tile at BASELINE config 2), the window entries, the clip's partial sums
    const int total = p.spec_mode ? 0 : M * FPT;
    const float* gb = p.grad_out + (size_t)b * M * T;
    const float* yb = p.out ? p.out + (size_t)b * M * T : nullptr;
    // (requests only: nothing here touches what was loaded -- a select on a loaded value, or a branch around a load, makes the
    // compiler wait for it on the spot, and vector loads return in order: that wait would also sit out the 2 R sample loads above)
    const float* ysrc = yb ? yb : gb;                                             // always a valid address; ignored without the log
    auto gm_fetch = [&](int base, float (&v)[4], float (&y)[4]) {
        static_for<0, 4>([&](auto uu) {
            constexpr int u = decltype(uu)::value;
            const int idx = base + u * THREADS, m = idx / FPT, t = t0 + idx % FPT;
            const bool ok = idx < total && t < T;
            const unsigned o = ok ? (unsigned)(m * T + t) : 0u;
            v[u] = gb[o];
            y[u] = ysrc[o];
        });
    };
    auto gm_store = [&](int base, const float (&v)[4], const float (&y)[4]) {
        static_for<0, 4>([&](auto uu) {
            constexpr int u = decltype(uu)::value;
            const int idx = base + u * THREADS, t = t0 + idx % FPT;
            if (idx < total) {
                const float val = yb ? v[u] * expf(-y[u]) : v[u];
                gm[idx] = t < T ? val : 0.f;
            }
        });
    };
    float gv[4], gy[4];
    if (total > 0) gm_fetch(tid, gv, gy);
    constexpr int WPT = (N + THREADS - 1) / THREADS;
    float mean = 0.f;
    if (p.own_prep) {
        // short clips, plain Gaussian window: no dmel_prep_kernel launch.  The window (time_frequency.py:21-30, the fp32 expression of
        // the forward) is evaluated here, and every workgroup adds up its clip itself in a fixed order (models.py:38; L2 hits
        // after the first toucher): requests in batches of 8 per thread, one round trip per batch.
        const float* xc = p.x + (size_t)b * L;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int i0 = 0;
        constexpr int KB = 8;
        if ((reinterpret_cast<uintptr_t>(xc) & 15) == 0) {
            const float4* x4 = reinterpret_cast<const float4*>(xc);
            const int n4 = L / 4;
            for (int base = 0; base < n4; base += THREADS * KB) {
                float4 v[KB];
                static_for<0, KB>([&](auto jj) {
                    constexpr int jv = decltype(jj)::value;
                    const int q = base + tid + THREADS * jv;
                    v[jv] = x4[q < n4 ? q : n4 - 1];
                });
                static_for<0, KB>([&](auto jj) {
                    constexpr int jv = decltype(jj)::value;
                    const bool ok = base + tid + THREADS * jv < n4;
                    a0 += ok ? v[jv].x : 0.f; a1 += ok ? v[jv].y : 0.f; a2 += ok ? v[jv].z : 0.f; a3 += ok ? v[jv].w : 0.f;
                });
            }
            i0 = n4 * 4;
        }
        for (int i = i0 + tid; i < L; i += THREADS) a0 += xc[i];
        // lambd by value, or read here from the parameter's storage (a uniform scalar load; dmel_backward_x_dev: no host read)
        float win_denom = p.win_denom;
        if (p.lam_dev) win_denom = __builtin_fabsf(*(const __attribute__((address_space(4))) float*)p.lam_dev) + 1e-15f;   // scalar load: see lam_load
        static_for<0, WPT>([&](auto ww) {
            constexpr int wi = decltype(ww)::value;
            const int n = tid + THREADS * wi;
            if (n < WN) {
                const float d = (float)n - (float)N / 2.0f;
                const float tq = d / win_denom;
                win[n] = expf(-0.5f * (tq * tq));
            }
        });
        // the forward's mean (dmel_kernels.h: "the clip mean"): fp32 tree, quotient rounded once
        float ps = wave_sum((a0 + a1) + (a2 + a3));
        float* redm = reinterpret_cast<float*>(smem_raw + p.tw2_off + (C > 1 ? R * C * 8 : 0));
        if (lane == 0) redm[wave] = ps;
        __syncthreads();
        float tot = 0.f;
        for (int q = 0; q < THREADS / 64; ++q) tot += redm[q];
        mean = mean_quotient(tot, L, p.inv_L);
    } else {
        float wv[WPT];
        static_for<0, WPT>([&](auto ww) { constexpr int wi = decltype(ww)::value; const int n = tid + THREADS * wi; wv[wi] = p.win2[n < WN ? n : 0].x; });
        mean = clip_mean_psum(p.psum, p.nchunks, b, p.L);
        static_for<0, WPT>([&](auto ww) { constexpr int wi = decltype(ww)::value; const int n = tid + THREADS * wi; if (n < WN) win[n] = wv[wi]; });
    }
    if (total > 0) {
        gm_store(tid, gv, gy);
        for (int base = tid + 4 * THREADS; base < total; base += 4 * THREADS) { gm_fetch(base, gv, gy); gm_store(base, gv, gy); }
        if (tid < FPT) gm[total + tid] = 0.f;                                     // row M: zeros, read as "the next row" of row M - 1
    }
    // the radix-C twiddles through LDS (R x C entries: a wave-wide global load of them would still move 512 B per p1)
    float2* tw2l = reinterpret_cast<float2*>(smem_raw + p.tw2_off);
    if (C > 1 && tid < R * C) tw2l[tid] = p.tw2[tid];
    XSTAMP(1);   // prologue issued
    __syncthreads();
    XSTAMP(2);   // ... and complete

    // ---- forward transform of the pair: Z = FFT(x~_a w + i x~_b w)
    {
        v2f z[R];
        // window entries n = lg + G a: the first half directly, the second half mirrored when only half the table is kept
        const int wlo = lg, whi = sym ? N - lg : lg;
        const int whs = sym ? -G : G;
        static_for<0, R>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            const float w = (a < R / 2) ? win[wlo + G * a] : win[whi + whs * a];
            if (inside) z[a] = v2f{xa[a] - mean, xc[a] - mean} * splat(w);
            else {
                const int ia = f0 + lg + G * a, ib = ia + hop;
                const float va = (ia >= 0 && ia < L) ? xa[a] - mean : 0.f;
                const float vb = (ib >= 0 && ib < L) ? xc[a] - mean : 0.f;
                z[a] = v2f{va, vb} * splat(w);
            }
        });
        wave_fft<R, C, G>(z, sl, lg, p.tw1, tw2l, [&](auto pp1, int qp, int p2, v2f v) {
            constexpr int p1 = decltype(pp1)::value;
            sl[qp + (RR + PADC) * p2 + R * p1] = v;
        });
    }
    wave_sync();
    XSTAMP(3);   // first transform
    // ---- bin by bin: the two spectra, the gradient of the power spectrum, conj(H_a + i H_b) back in place.
    // Addresses as in the forward's pairing pass: a per-lane byte base plus a compile-time offset --
    //   Z[k],   k = lg + G i:   zb + 8 (G i + pad(G i))
    //   Z[N-k], lg >= 1:        mb + 8 (c_i + pad(c_i)),  mb = slot + 8 (G - lg),  c_i = N - G (i + 1)
    //   lane 0: N - G i itself; one padding step further when it starts an R*R block (mbA), bin 0 for i = 0 (mb0)
    {
        v2f wk[NPAIR], wn[NPAIR];
        const bool hasA = tA < T, hasB = tA + 1 < T;
        // row k of the filterbank as (c0, c1, first column, columns): an HTK row has two non-zero columns, and consecutive lanes
        // read consecutive 16-byte entries (the coefficients themselves lie a whole row of M floats apart: 64 cache lines per
        // load).  Rows with more columns (a trained, dense bank: p.long_rows, uniform) add the rest from the matrix.
        float4 rk[NPAIR];
        if (!p.spec_mode) {
            static_for<0, NPAIR>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                rk[i] = p.rowpk[(i < NPAIR - 1 || lg == 0) ? lg + G * i : 0];
            });
        }
        int zb = slot_b + lg * 8;
        int mb = slot_b + (G - lg) * 8;
        int mbA = mb + ((lg == 0) ? PADC * 8 : 0);
        int mb0 = (lg == 0) ? slot_b : mb + (N - G + PADC * ((N - G) / RR)) * 8;       // full address of Z[N-k] for i = 0
        asm volatile("" : "+v"(zb), "+v"(mb), "+v"(mbA), "+v"(mb0));
        const float* gcol0 = gm + 2 * slot;                                           // this pair's two columns of gm
        static_for<0, NPAIR>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            constexpr int ck = G * i, cm = N - G * (i + 1);
            constexpr bool crossing = PADC != 0 && ((N - G * i) % RR) == 0;
            const int mbase = (i == 0) ? mb0 : (crossing ? mbA : mb);
            const v2f zk = *reinterpret_cast<const v2f*>(smem_raw + zb + (ck + PADC * (ck / RR)) * 8);
            const v2f zn = *reinterpret_cast<const v2f*>(smem_raw + mbase + ((i == 0) ? 0 : (cm + PADC * (cm / RR)) * 8));
            // X_a = (Z_k + conj Z_{N-k}) / 2,  X_b = (Z_k - conj Z_{N-k}) / (2i)
            const float xar = 0.5f * (zk.x + zn.x), xai = 0.5f * (zk.y - zn.y);
            const float xbr = 0.5f * (zk.y + zn.y), xbi = -0.5f * (zk.x - zn.x);
            float gpa = 0.f, gpb = 0.f;
            if (p.spec_mode) {
                const int kc = (i < NPAIR - 1 || lg == 0) ? lg + G * i : 0;
                const float* gs = p.grad_out + ((size_t)b * p.F + kc) * T + tA;
                gpa = hasA ? gs[0] : 0.f;
                gpb = hasB ? gs[1] : 0.f;
            } else {
                const int b0 = __builtin_bit_cast(int, rk[i].z);
                const float* gcol = gcol0 + b0 * FPT;
                const float2 ga = *reinterpret_cast<const float2*>(gcol);
                const float2 gb2 = *reinterpret_cast<const float2*>(gcol + FPT);      // (row M of gm exists and is zero)
                gpa = fmaf(rk[i].y, gb2.x, rk[i].x * ga.x);
                gpb = fmaf(rk[i].y, gb2.y, rk[i].x * ga.y);
                if (p.long_rows) {
                    const int nb = __builtin_bit_cast(int, rk[i].w);
                    const float* fr = p.fb + (size_t)((i < NPAIR - 1 || lg == 0) ? lg + G * i : 0) * M;
                    for (int m = b0 + 2; m < b0 + nb; ++m) {
                        const float c = fr[m];
                        const float2 g2 = *reinterpret_cast<const float2*>(gcol0 + m * FPT);
                        gpa = fmaf(c, g2.x, gpa);
                        gpb = fmaf(c, g2.y, gpb);
                    }
                }
            }
            // k = 0 and k = N/2 (lane 0 of the first / last round): X is real there and the bin is its own mirror image
            if constexpr (i == 0 || i == NPAIR - 1) {
                const bool edge = (lg == 0);
                const float sc = edge ? 2.f : 1.f;
                const float har = sc * gpa * xar, hai = edge ? 0.f : gpa * xai;
                const float hbr = sc * gpb * xbr, hbi = edge ? 0.f : gpb * xbi;
                wk[i] = v2f{har - hbi, -(hai + hbr)};
                wn[i] = v2f{har + hbi, hai - hbr};
            } else {
                const float har = gpa * xar, hai = gpa * xai, hbr = gpb * xbr, hbi = gpb * xbi;
                wk[i] = v2f{har - hbi, -(hai + hbr)};                             // conj(H_a + i H_b) at k
                wn[i] = v2f{har + hbi, hai - hbr};                                // ... at N - k (Hermitian extension)
            }
        });
        wave_sync();                                                              // every read of Z precedes the first write
        static_for<0, NPAIR>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            constexpr int ck = G * i, cm = N - G * (i + 1);
            constexpr bool crossing = PADC != 0 && ((N - G * i) % RR) == 0;
            const int mbase = (i == 0) ? mb0 : (crossing ? mbA : mb);
            const bool edge = (lg == 0) && (i == 0 || i == NPAIR - 1);
            if (i < NPAIR - 1 || lg == 0) {                                       // the last round holds only the Nyquist bin
                *reinterpret_cast<v2f*>(smem_raw + zb + (ck + PADC * (ck / RR)) * 8) = wk[i];
                if (!edge) *reinterpret_cast<v2f*>(smem_raw + mbase + ((i == 0) ? 0 : (cm + PADC * (cm / RR)) * 8)) = wn[i];
            }
        });
    }
    wave_sync();
    XSTAMP(4);   // bin pass
    // ---- second transform: FFT(conj U) = conj(dv_a + i dv_b); windowed, the two frame gradients go to the two halves of the slot
    // (one plane of N floats each: the overlap-add below then reads consecutive words)
    float* pl = reinterpret_cast<float*>(sl);
    {
        v2f z[R];
        static_for<0, R>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            constexpr int ck = G * a;
            z[a] = *reinterpret_cast<const v2f*>(smem_raw + slot_b + lg * 8 + (ck + PADC * (ck / RR)) * 8);
        });
        wave_sync();
        wave_fft<R, C, G>(z, sl, lg, p.tw1, tw2l, [&](auto pp1, int qp, int p2, v2f v) {
            constexpr int p1 = decltype(pp1)::value;
            // window entry of n = qp + R p1 + R^2 p2: mirrored in the upper half when only half the table is kept
            const int n0 = qp + RR * p2;
            float w;
            if constexpr (C > 1) {
                const bool mir = sym && (2 * p2 >= C);
                w = win[(mir ? N - n0 : n0) + (mir ? -R : R) * p1];
            } else {
                w = (p1 < R / 2) ? win[n0 + R * p1] : win[sym ? N - n0 - R * p1 : n0 + R * p1];
            }
            float* dst = pl + qp + (RR + PADP) * p2 + R * p1;
            dst[0] = v.x * w;
            dst[SS] = -(v.y * w);
        });
    }
    XSTAMP(5);   // second transform
    __syncthreads();
    XSTAMP(6);   // barrier
    // ---- overlap-add of the tile's frames: sample i of the segment (clip sample t0 hop - N/2 + i) gathers frames
    // t hop <= i < t hop + N in increasing t; frame t of the tile is the plane at t * SS floats
    const int span = (FPT - 1) * hop + N;
    const long long s0 = (long long)t0 * hop - N / 2;
    float* seg = p.frames + ((size_t)b * tiles + tile) * (size_t)span;
    const float* slf = reinterpret_cast<const float*>(smem_raw);
    // i = t hop + m (0 <= m < hop): frame t is the last one that starts at or before sample i; it and the K - 1 frames before it
    // may cover the sample (at offsets m, m + hop, ...): added in that order.  (t, m) advance by increments, no division per sample.
    const float inv_hop = 1.0f / (float)hop;
    auto div_hop = [&](int v) {                                    // v / hop for 0 <= v < 2^24: float estimate, one correction
        int q = (int)((float)v * inv_hop);
        const int r = v - q * hop;
        q += (r >= hop) ? 1 : 0;
        q -= (r < 0) ? 1 : 0;
        return q;
    };
    const int K = __builtin_amdgcn_readfirstlane(div_hop(N + hop - 1));
    const int qs = __builtin_amdgcn_readfirstlane(div_hop(THREADS)), rs = THREADS - qs * hop;
    const bool all_in = s0 >= 0 && s0 + span <= L;
    int t = div_hop(tid), m = tid - t * hop;
    float fsum = 0.f;
    for (int i = tid; i < span; i += THREADS) {
        float acc = 0.f;
        int tt = t, off = m;
        for (int kk = 0; kk < K; ++kk) {
            const bool ok = (unsigned)tt < (unsigned)FPT && off < N;
            const int a = tt * SS + off + (PADP ? (off / RR) * PADP : 0);
            const float v = slf[ok ? a : 0];
            acc += ok ? v : 0.f;
            tt -= 1; off += hop;
        }
        seg[i] = acc;
        if (all_in) fsum += acc;
        else { const long long ia = s0 + i; fsum += (ia >= 0 && ia < L) ? acc : 0.f; }
        m += rs; t += qs;
        if (m >= hop) { m -= hop; t += 1; }
    }
    XSTAMP(7);   // overlap-add, segment stored
    // what the tile contributes to the sum of the clip's gradient (inside the clip): the lanes in a fixed order, the waves in fp64
    fsum = wave_sum(fsum);
    float* red = gm;                                               // (gm is dead)
    if (lane == 0) red[wave] = fsum;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) tot += (double)red[w];
        p.csum[(size_t)b * tiles + tile] = tot;
    }
    XSTAMP(8);   // sum of the tile
}

// grid (chunks, B): clip sample i is covered by the segments of tiles q with q TS - N/2 <= i < q TS - N/2 + span (TS = FPT hop
// samples between tile starts); they are added in increasing q and the mean of the clip's gradient is subtracted
__global__ void __launch_bounds__(256) dmel_xgrad_combine_kernel(XgradParams p)
{
    if (xgrad_not_this_nfft(p)) return;
    const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const int tiles = p.tiles, span = p.span, ts = p.tile_step, half = p.N / 2;
    const float* sg = p.frames + (size_t)b * tiles * (size_t)span;
    float* gx = p.grad_x + (size_t)b * p.L;
    const int lo = chunk * kXgChunk, hi = min(lo + kXgChunk, p.L);
    // tiles that can cover a sample of this chunk: two divisions per workgroup, a range test per sample and tile
    const int u_lo = lo + half, u_hi = hi - 1 + half;
    const int qa = u_lo < span ? 0 : (u_lo - span) / ts + 1;
    const int qb = min(tiles - 1, u_hi / ts);
    auto clip_mean = [&]() {
        float mean = 0.f;
        if (p.remove_dc) {
            double acc = 0.0;
            for (int q = 0; q < tiles; ++q) acc += p.csum[(size_t)b * tiles + q];  // uniform: every thread adds the same values in the same order
            mean = (float)(acc / (double)p.L);
        }
        return mean;
    };
    // rows of four samples when every row involved starts on a 16-byte boundary (hop, n_fft / 2 and the clip length multiples of 4)
    const bool vec = ((p.hop | half | p.L | span) & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.frames) | reinterpret_cast<uintptr_t>(p.grad_x)) & 15) == 0;
    if (vec) {
        constexpr int PER = kXgChunk / (256 * 4);                  // rows of 4 per thread
        float4 acc[PER];
        static_for<0, PER>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            const int i = lo + (tid + 256 * r) * 4;
            acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < hi) {
                const int u = i + half;
                for (int q = qa; q <= qb; ++q) {
                    const int off = u - q * ts;                   // a multiple of 4: the four samples are inside or outside together
                    if (off >= 0 && off < span) {
                        const float4 v = *reinterpret_cast<const float4*>(sg + (size_t)q * span + off);
                        acc[r].x += v.x; acc[r].y += v.y; acc[r].z += v.z; acc[r].w += v.w;
                    }
                }
            }
        });
        const float mean = clip_mean();                            // (its loads travel with the segment loads above)
        static_for<0, PER>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            const int i = lo + (tid + 256 * r) * 4;
            if (i < hi) *reinterpret_cast<float4*>(gx + i) = make_float4(acc[r].x - mean, acc[r].y - mean, acc[r].z - mean, acc[r].w - mean);
        });
        return;
    }
    const float mean = clip_mean();
    #pragma unroll 4
    for (int i = lo + tid; i < hi; i += 256) {
        const int u = i + half;
        float s = 0.f;
        for (int q = qa; q <= qb; ++q) {
            const int off = u - q * ts;
            if (off >= 0 && off < span) s += sg[(size_t)q * span + off];
        }
        gx[i] = s - mean;
    }
}

template <int N> static size_t xgrad_wave_tw2_off(int M, int win_n)
{
    using PL = XgPlan<N>;
    size_t gm = (size_t)(M + 1) * PL::FPT * sizeof(float);          // one zero row behind the last mel band
    if (gm < PL::THREADS * sizeof(double)) gm = PL::THREADS * sizeof(double);
    return (size_t)PL::SLOTS * PL::SS * 8 + (size_t)((win_n + 3) & ~3) * sizeof(float) + gm;
}

template <int N> static size_t xgrad_wave_lds(int M, int win_n)
{
    using PL = XgPlan<N>;
    return xgrad_wave_tw2_off<N>(M, win_n) + (PL::C > 1 ? (size_t)PL::R * PL::C * 8 : 0) + 64;      // + one partial sum per wave
}

template <class F> static bool xgrad_with_plan(int n, F&& f)
{
    switch (n) {
    case 32: f(IC<32>{}); return true;
    case 64: f(IC<64>{}); return true;
    case 128: f(IC<128>{}); return true;
    case 256: f(IC<256>{}); return true;
    case 512: f(IC<512>{}); return true;
    case 1024: f(IC<1024>{}); return true;
    case 2048: f(IC<2048>{}); return true;
    default: return false;
    }
}

// the wave-FFT path takes this shape (otherwise the LDS radix-2 kernels above run)
bool xgrad_wave_shape(int n_fft, int n_mels, int win_n, int* frames_per_tile)
{
    bool ok = false;
    xgrad_with_plan(n_fft, [&](auto nn) {
        constexpr int N = decltype(nn)::value;
        ok = xgrad_wave_lds<N>(n_mels, win_n) <= (size_t)XgPlan<N>::LDS_MAX;
        if (frames_per_tile) *frames_per_tile = XgPlan<N>::FPT;
    });
    return ok;
}

hipError_t xgrad_prepare_attributes()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmel_xgrad_frames_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kMaxNfft * (int)sizeof(float2));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmel_xgrad_frames_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 12);
    for (int n = 32; n <= 2048 && e == hipSuccess; n *= 2)
        xgrad_with_plan(n, [&](auto nn) {
            constexpr int N = decltype(nn)::value;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmel_xgrad_wave_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, XgPlan<N>::LDS_MAX);
        });
    return e;
}

hipError_t launch_xgrad(const XgradParams& p, hipStream_t s)
{
    if (p.tiles > 0) {
        // wave-FFT path: (B x tiles) workgroups, then the combine pass
        const long long grid = (long long)p.B * p.tiles;
        if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
        hipError_t e = hipErrorInvalidValue;
        xgrad_with_plan(p.N, [&](auto nn) {
            constexpr int N = decltype(nn)::value;
            XgradParams q = p;
            q.tw2_off = (int)xgrad_wave_tw2_off<N>(p.spec_mode ? 0 : p.M, p.win_n);
            hipLaunchKernelGGL(dmel_xgrad_wave_kernel<N>, dim3((unsigned)grid), dim3(XgPlan<N>::THREADS), xgrad_wave_lds<N>(p.spec_mode ? 0 : p.M, p.win_n), s, q);
            e = hipGetLastError();
        });
        if (e != hipSuccess) return e;
        const dim3 g2((unsigned)((p.L + kXgChunk - 1) / kXgChunk), (unsigned)p.B);
        hipLaunchKernelGGL(dmel_xgrad_combine_kernel, g2, dim3(256), 0, s, p);
        return hipGetLastError();
    }
    const long long grid = (long long)p.B * ((p.T + 1) / 2);
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    XgradParams q = p;
    q.tw_in_lds = p.N <= 8192 ? 1 : 0;
    size_t lds = (size_t)p.N * sizeof(float2) + (q.tw_in_lds ? (size_t)(p.N / 2) * sizeof(float2) : 0);
    if (lds < 2 * kXgThreads * sizeof(double)) lds = 2 * kXgThreads * sizeof(double);      // the per-frame sums are reduced where the sequence was
    if (q.tw_in_lds) hipLaunchKernelGGL(dmel_xgrad_frames_kernel<true>, dim3((unsigned)grid), dim3(kXgThreads), lds, s, q);
    else hipLaunchKernelGGL(dmel_xgrad_frames_kernel<false>, dim3((unsigned)grid), dim3(kXgThreads), lds, s, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_xgrad_gather(p, s);
}

hipError_t launch_xgrad_gather(const XgradParams& p, hipStream_t s)
{
    const dim3 g2((unsigned)((p.L + kXgChunk - 1) / kXgChunk), (unsigned)p.B);
    hipLaunchKernelGGL(dmel_xgrad_gather_kernel, g2, dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace dmel

namespace dmel { int xgrad_chunks(int L) { return (L + kXgChunk - 1) / kXgChunk; } }      // (kept for the workspace layout of dmel_api.cpp)

#ifdef DMEL_STAMPS
extern "C" int dmel_debug_read_xstamps(unsigned long long* host, int count)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dmel::g_xstamps), sizeof(unsigned long long) * (size_t)count);
}
#endif
