// dmel_aux.hip -- small kernels around the fused forward:
//   * dmel_dot_*   : lambd.grad = sum grad_out * tangent  (the whole backward of the layer once the
//                    forward has carried d out / d lambd; train.py:47 reaches exactly this scalar)
//   * dmel_naive_* : direct-DFT forward for n_fft < 32 (lambd < 5.5 samples) and as an on-device
//                    cross-check of the wave-FFT kernel; same semantics (models.py:33-56, :73).
#include "dmel_kernels.h"

namespace dmel {

// ---- deterministic two-stage dot product ------------------------------------------------------
__global__ void __launch_bounds__(kThreads) dmel_dot_partial_kernel(const float* __restrict__ g,
                                                                    const float* __restrict__ t,
                                                                    long long count, double* partials)
{
    __shared__ double red[kThreads];
    const int tid = threadIdx.x;
    const long long stride = (long long)gridDim.x * kThreads;
    double acc = 0.0;
    const long long n4 = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(t)) & 15) == 0 ? count / 4 : 0;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    for (long long i = (long long)blockIdx.x * kThreads + tid; i < n4; i += stride) {
        const float4 a = g4[i], b = t4[i];
        acc += ((double)a.x * (double)b.x + (double)a.y * (double)b.y) + ((double)a.z * (double)b.z + (double)a.w * (double)b.w);
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * kThreads + tid; i < count; i += stride)
        acc += (double)g[i] * (double)t[i];
    red[tid] = acc;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) partials[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(kThreads) dmel_dot_final_kernel(const double* partials, int n, int accumulate, float* result)
{
    __shared__ double red[kThreads];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int i = tid; i < n; i += kThreads) acc += partials[i];
    red[tid] = acc;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) result[0] = accumulate ? (float)((double)result[0] + red[0]) : (float)red[0];
}

hipError_t launch_dot(const float* g, const float* t, long long count, int accumulate, double* partials,
                      int max_partials, float* result, hipStream_t s)
{
    long long want = (count + (long long)kThreads * 8 - 1) / ((long long)kThreads * 8);
    int blocks = (int)(want < 1 ? 1 : (want > max_partials ? max_partials : want));
    hipLaunchKernelGGL(dmel_dot_partial_kernel, dim3(blocks), dim3(kThreads), 0, s, g, t, count, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dmel_dot_final_kernel, dim3(1), dim3(kThreads), 0, s, partials, blocks, accumulate, result);
    return hipGetLastError();
}

// ---- direct-DFT forward ------------------------------------------------------------------------
// one workgroup per frame; dynamic LDS: a[N], b[N], P[F], D[F]
__global__ void __launch_bounds__(128) dmel_naive_kernel(NaiveParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* fa = reinterpret_cast<float*>(smem_raw);
    float* fb_ = fa + p.N;
    float* P = fb_ + p.N;
    float* D = P + p.F;
    const int tid = threadIdx.x;
    const int b = blockIdx.x / p.T, t = blockIdx.x % p.T;
    const float* xb = p.x + (size_t)b * p.L;
    float mean = 0.f;
    if (p.remove_dc) {
        double s = 0.0;
        for (int c = 0; c < p.nchunks; ++c) s += (double)p.psum[(size_t)b * p.nchunks + c];
        mean = (float)(s * (double)p.inv_L);
    }
    for (int n = tid; n < p.N; n += blockDim.x) {
        const long long i = (long long)t * p.hop - p.N / 2 + n;
        const float v = (i >= 0 && i < p.L) ? (xb[i] - mean) : 0.f;
        const float2 wd = p.win2[n];
        fa[n] = v * wd.x;
        fb_[n] = v * wd.y;
    }
    __syncthreads();
    for (int f = tid; f < p.F; f += blockDim.x) {
        float xr = 0.f, xi = 0.f, yr = 0.f, yi = 0.f;
        for (int n = 0; n < p.N; ++n) {
            const int kn = (int)(((long long)f * n) % p.N);
            float sn, cs;
            sincospif(2.0f * (float)kn / (float)p.N, &sn, &cs);      // exp(-i th) = cs - i sn
            xr = fmaf(fa[n], cs, xr); xi = fmaf(-fa[n], sn, xi);
            yr = fmaf(fb_[n], cs, yr); yi = fmaf(-fb_[n], sn, yi);
        }
        P[f] = xr * xr + xi * xi;
        D[f] = 2.f * (xr * yr + xi * yi);
    }
    __syncthreads();
    if (p.mode == kSpec) {
        for (int f = tid; f < p.F; f += blockDim.x) p.out[((size_t)b * p.F + f) * p.T + t] = P[f];
        return;
    }
    const bool do_log = (p.flags & 1u) != 0;
    for (int m = tid; m < p.M; m += blockDim.x) {
        float mel = 0.f, dmel = 0.f;
        for (int f = 0; f < p.F; ++f) {
            const float c = p.fb[(size_t)f * p.M + m];
            mel = fmaf(c, P[f], mel);
            dmel = fmaf(c, D[f], dmel);
        }
        dmel *= p.sign;
        const size_t o = ((size_t)b * p.M + m) * p.T + t;
        if (do_log) {
            const float me = mel + p.eps;
            p.out[o] = logf(me);
            if (p.tangent) p.tangent[o] = dmel / me;
        } else {
            p.out[o] = mel;
            if (p.tangent) p.tangent[o] = dmel;
        }
    }
}

hipError_t launch_naive(const NaiveParams& p, hipStream_t s)
{
    const size_t lds = sizeof(float) * (size_t)(2 * p.N + 2 * p.F);
    hipLaunchKernelGGL(dmel_naive_kernel, dim3((unsigned)(p.B * p.T)), dim3(128), lds, s, p);
    return hipGetLastError();
}

}  // namespace dmel
