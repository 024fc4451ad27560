// dmel_aux.hip -- small kernels around the fused forward:
//   * dmel_dot_*   : lambd.grad = sum grad_out * tangent  (the whole backward of the layer once the
//                    forward has carried d out / d lambd; train.py:47 reaches exactly this scalar)
//   * dmel_naive_* : direct-DFT forward for n_fft < 32 (lambd < 5.5 samples) and as an on-device
//                    cross-check of the wave-FFT kernel; same semantics (models.py:33-56, :73).
#include "dmel_kernels.h"

namespace dmel {

// ---- deterministic dot product, one launch ----------------------------------------------------
// Every workgroup reduces its slice in fp64 and publishes ONE partial with a write-through (sc1) store;
// the workgroup that draws the last ticket adds the partials in index order (fixed order => the same
// bits every run) and writes the scalar.  Hand-off = cdna_hip_programming.md Guideline 16, R1 in its
// counter form: sc1 payload, the storing lane drains vmcnt, relaxed agent-scope ticket; the reader uses
// sc1 loads only, so no acquire fence is needed.  The counter is left at 0 for the next launch.
// fixed-order reduction of one fp64 value per thread: butterfly inside each wave (shuffles, no LDS),
// then the four wave sums in index order.  Result valid in thread 0.
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double block_sum(double v, double* red4)
{
    // inside each row of 16 lanes with DPP (no LDS round trips), then the 4 rows x 4 waves through LDS
    v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);     // row_half_mirror
    v += dpp_f64<0x140>(v);     // row_mirror
    const int tid = threadIdx.x;
    if ((tid & 15) == 0) red4[tid >> 4] = v;
    __syncthreads();
    double s = 0.0;
    for (int q = 0; q < 16; ++q) s += red4[q];
    return s;
}

__global__ void __launch_bounds__(kThreads) dmel_dot_kernel(const float* __restrict__ g, const float* __restrict__ t,
                                                            long long count, double* partials, unsigned* counter,
                                                            int accumulate, float* result)
{
    __shared__ double red4[16], red4b[16];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const long long stride = (long long)gridDim.x * kThreads;
    double acc = 0.0;
    const long long n4 = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(t)) & 15) == 0 ? count / 4 : 0;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    long long i = (long long)blockIdx.x * kThreads + tid;
    for (; i + 3 * stride < n4; i += 4 * stride) {        // four independent 16-byte loads per tensor in flight
        const float4 a0 = g4[i], a1 = g4[i + stride], a2 = g4[i + 2 * stride], a3 = g4[i + 3 * stride];
        const float4 b0 = t4[i], b1 = t4[i + stride], b2 = t4[i + 2 * stride], b3 = t4[i + 3 * stride];
        acc += ((double)a0.x * (double)b0.x + (double)a0.y * (double)b0.y) + ((double)a0.z * (double)b0.z + (double)a0.w * (double)b0.w);
        acc += ((double)a1.x * (double)b1.x + (double)a1.y * (double)b1.y) + ((double)a1.z * (double)b1.z + (double)a1.w * (double)b1.w);
        acc += ((double)a2.x * (double)b2.x + (double)a2.y * (double)b2.y) + ((double)a2.z * (double)b2.z + (double)a2.w * (double)b2.w);
        acc += ((double)a3.x * (double)b3.x + (double)a3.y * (double)b3.y) + ((double)a3.z * (double)b3.z + (double)a3.w * (double)b3.w);
    }
    for (; i < n4; i += stride) {
        const float4 a = g4[i], b = t4[i];
        acc += ((double)a.x * (double)b.x + (double)a.y * (double)b.y) + ((double)a.z * (double)b.z + (double)a.w * (double)b.w);
    }
    for (long long k = n4 * 4 + (long long)blockIdx.x * kThreads + tid; k < count; k += stride)
        acc += (double)g[k] * (double)t[k];
    const double bsum = block_sum(acc, red4);
    if (tid == 0) {
        __hip_atomic_store(&partials[blockIdx.x], bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (ticket == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    double sum = 0.0;
    for (int q = tid; q < (int)gridDim.x; q += kThreads)
        sum += __hip_atomic_load(&partials[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double total = block_sum(sum, red4b);
    if (tid == 0) {
        result[0] = accumulate ? (float)((double)result[0] + total) : (float)total;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

hipError_t launch_dot(const float* g, const float* t, long long count, int accumulate, double* partials,
                      unsigned* counter, int max_partials, float* result, hipStream_t s)
{
    // 32 floats per thread and tensor: few workgroups on purpose -- every workgroup ends with one ticket on a single
    // counter (~11 ns each, serialised), which cost more than the loads at 256+ workgroups
    long long want = (count + (long long)kThreads * 32 - 1) / ((long long)kThreads * 32);
    int blocks = (int)(want < 1 ? 1 : (want > max_partials ? max_partials : want));
    hipLaunchKernelGGL(dmel_dot_kernel, dim3(blocks), dim3(kThreads), 0, s, g, t, count, partials, counter, accumulate, result);
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
    if (p.mode == kSpec || p.mode == kSpecTrain) {
        for (int f = tid; f < p.F; f += blockDim.x) {
            const size_t o = ((size_t)b * p.F + f) * p.T + t;
            p.out[o] = P[f];
            if (p.tangent) p.tangent[o] = p.sign * D[f];
        }
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
