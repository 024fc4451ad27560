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

namespace dmel {

constexpr int kXgThreads = 256;

template <bool TWLDS>
__global__ void __launch_bounds__(kXgThreads) dmel_xgrad_frames_kernel(XgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* Z = reinterpret_cast<float2*>(smem_raw);
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
        double s = 0.0;
        for (int c = 0; c < p.nchunks; ++c) s += (double)p.psum[(size_t)b * p.nchunks + c];
        mean = (float)(s * (double)p.inv_L);
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
    lds_fft_dif<kXgThreads>(Z, N, p.logN, tid, twiddle);
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
    lds_fft_dit<kXgThreads>(Z, N, p.logN, tid, twiddle);
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

hipError_t xgrad_prepare_attributes()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmel_xgrad_frames_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kMaxNfft * (int)sizeof(float2));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(dmel_xgrad_frames_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               8192 * 12);
}

hipError_t launch_xgrad(const XgradParams& p, hipStream_t s)
{
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
    const dim3 g2((unsigned)((p.L + kXgChunk - 1) / kXgChunk), (unsigned)p.B);
    hipLaunchKernelGGL(dmel_xgrad_gather_kernel, g2, dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace dmel

namespace dmel { int xgrad_chunks(int L) { return (L + kXgChunk - 1) / kXgChunk; } }      // (kept for the workspace layout of dmel_api.cpp)
