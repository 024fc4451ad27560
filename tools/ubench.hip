// tools/ubench.hip -- micro-measurements behind the round-5 rewrite of dmel_fwd_kernel (NOTEBOOK R5.1).  Standalone:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I differentiable-mel-spectrogram_amd/csrc tools/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
// Reports, per wave and in shader cycles (s_memtime), with 1 / 2 / 4 waves per SIMD resident on every CU:
//   fft32 dif / dit : the register radix-32 transform of dmel_wavefft.h, DIF (228 packed ops) against DIT Linzer-Feig (194)
//   pkfma / fma     : independent chains of v_pk_fma_f32 against v_fma_f32 (same flops)
//   mfma4x4         : v_mfma_f32_4x4x1_16b_f32 on 1 / 2 / 4 accumulators, and its operand / result layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "dmel_wavefft.h"

using namespace dmel;

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

template <int VARIANT> __global__ void k_fft32(const float2* in, float2* out, unsigned long long* cyc, int iters)
{
    extern __shared__ unsigned char smem[];
    v2f z[32];
    for (int i = 0; i < 32; ++i) { const float2 v = in[(threadIdx.x + 64 * i) & 4095]; z[i] = v2f{v.x, v.y}; }
    __syncthreads();
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
        if constexpr (VARIANT == 0) fft_reg<32>(z); else fft_reg_dit<32>(z);
        for (int i = 0; i < 32; ++i) z[i] = z[i] * splat(0.03125f);
    }
    const unsigned long long t1 = now();
    float2 acc = make_float2(0.f, 0.f);
    for (int i = 0; i < 32; ++i) { acc.x += z[i].x; acc.y += z[i].y; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// correctness of the two register transforms against each other and a direct DFT (one thread)
__global__ void k_fft32_check(const float2* in, float2* dif, float2* dit)
{
    v2f a[32], b[32];
    for (int i = 0; i < 32; ++i) { a[i] = v2f{in[i].x, in[i].y}; b[i] = a[i]; }
    fft_reg<32>(a); fft_reg_dit<32>(b);
    static_for<0, 32>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        dif[q] = make_float2(a[bitrev(q, 5)].x, a[bitrev(q, 5)].y);
        dit[q] = make_float2(b[bitrev(q, 5)].x, b[bitrev(q, 5)].y);
    });
}
template <int R> __global__ void k_fft_check(const float2* in, float2* dif, float2* dit)
{
    v2f a[R], b[R];
    for (int i = 0; i < R; ++i) { a[i] = v2f{in[i].x, in[i].y}; b[i] = a[i]; }
    fft_reg<R>(a); fft_reg_dit<R>(b);
    static_for<0, R>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        dif[q] = make_float2(a[bitrev(q, ilog2(R))].x, a[bitrev(q, ilog2(R))].y);
        dit[q] = make_float2(b[bitrev(q, ilog2(R))].x, b[bitrev(q, ilog2(R))].y);
    });
}

template <int PACKED> __global__ void k_valu(const float* in, float* out, unsigned long long* cyc, int iters)
{
    extern __shared__ unsigned char smem[];
    constexpr int NCH = 16;
    v2f a[NCH];
    for (int i = 0; i < NCH; ++i) a[i] = v2f{in[(threadIdx.x + i) & 1023], in[(threadIdx.x + 2 * i + 1) & 1023]};
    const v2f m = v2f{in[5], in[6]}, c = v2f{in[7], in[8]};
    __syncthreads();
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
        if constexpr (PACKED) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
            }
        }
    }
    const unsigned long long t1 = now();
    float acc = 0.f;
    for (int i = 0; i < NCH; ++i) acc += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int NACC> __global__ void k_mfma(const float* in, float* out, unsigned long long* cyc, int iters)
{
    extern __shared__ unsigned char smem[];
    floatx4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float a = in[threadIdx.x & 1023], b = in[(threadIdx.x + 77) & 1023];
    __syncthreads();
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = now();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// MFMAs beside packed VALU work of the same wave: what the contraction costs when it is interleaved with transforms
__global__ void k_mfma_valu(const float* in, float* out, unsigned long long* cyc, int iters, int with_mfma)
{
    extern __shared__ unsigned char smem[];
    floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
    v2f z[8];
    for (int i = 0; i < 8; ++i) z[i] = v2f{in[(threadIdx.x + i) & 1023], in[(threadIdx.x + 3 * i + 1) & 1023]};
    const v2f m = v2f{in[5], in[6]}, c = v2f{in[7], in[8]};
    const float a = in[threadIdx.x & 1023], b = in[(threadIdx.x + 77) & 1023];
    __syncthreads();
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(z[u]) : "v"(m), "v"(c));
            if (with_mfma) acc[u & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u & 1], 0, 0, 0);
        }
    }
    const unsigned long long t1 = now();
    float s = acc[0][0] + acc[1][1];
    for (int i = 0; i < 8; ++i) s += z[i].x + z[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

__global__ void k_mfma_layout(float* out)
{
    // A: lane l supplies 100 + l, B: lane l supplies 1000 + l  ->  D[block][i][j] = A(block, i) * B(block, j)
    const int l = threadIdx.x;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(100 + l), (float)(1000 + l), acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = acc[i];
}

template <class OutT, class K, class... A> static int timed(const char* name, K kern, int threads, int lds, int iters, double per_iter_units, const char* unit, A... args)
{
    const int blocks = 256;
    unsigned long long* cyc; float* outp;
    CK(hipMalloc(&cyc, blocks * (threads / 64) * 8)); CK(hipMalloc(&outp, (size_t)blocks * threads * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, args..., reinterpret_cast<OutT*>(outp), cyc, iters);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * (threads / 64));
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    printf("%-28s waves/SIMD %d  median %9.0f cyc/wave  = %7.2f cyc per %s   (kernel %.1f us)\n", name, threads / 256, med, med / (iters * per_iter_units), unit, ms * 1e3);
    CK(hipFree(cyc)); CK(hipFree(outp));
    return 0;
}

int main()
{
    std::vector<float2> hin(4096);
    for (int i = 0; i < 4096; ++i) hin[i] = make_float2((float)((i * 37) % 101) / 101.f - 0.5f, (float)((i * 53) % 97) / 97.f - 0.5f);
    float2* din; CK(hipMalloc(&din, 4096 * 8)); CK(hipMemcpy(din, hin.data(), 4096 * 8, hipMemcpyHostToDevice));
    // ---- correctness of fft_reg_dit against fft_reg and a direct DFT
    {
        float2 *d1, *d2; CK(hipMalloc(&d1, 64 * 8)); CK(hipMalloc(&d2, 64 * 8));
        auto check = [&](int R, auto kern) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(1), 0, 0, din, d1, d2);
            std::vector<float2> a(R), b(R);
            hipMemcpy(a.data(), d1, R * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, R * 8, hipMemcpyDeviceToHost);
            double e1 = 0, e2 = 0, sc = 0;
            for (int q = 0; q < R; ++q) {
                double re = 0, im = 0;
                for (int n = 0; n < R; ++n) { const double th = -2.0 * M_PI * q * n / R; re += hin[n].x * cos(th) - hin[n].y * sin(th); im += hin[n].x * sin(th) + hin[n].y * cos(th); }
                e1 = std::max(e1, std::max(fabs(a[q].x - re), fabs(a[q].y - im)));
                e2 = std::max(e2, std::max(fabs(b[q].x - re), fabs(b[q].y - im)));
                sc = std::max(sc, std::max(fabs(re), fabs(im)));
            }
            printf("fft_reg<%d>: max |err| dif %.3g  dit %.3g  (largest output %.3g)\n", R, e1, e2, sc);
        };
        check(4, k_fft_check<4>); check(8, k_fft_check<8>); check(16, k_fft_check<16>); check(32, k_fft_check<32>); check(64, k_fft_check<64>);
    }
    // ---- layout of v_mfma_f32_4x4x1_16b_f32
    {
        float* d; CK(hipMalloc(&d, 256 * 4));
        hipLaunchKernelGGL(k_mfma_layout, dim3(1), dim3(64), 0, 0, d);
        std::vector<float> h(256); CK(hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost));
        // hypothesis: lane l = 4 block + j holds D[block][i][j] in register i, A(block, i) from lane 4 block + i, B(block, j) from lane 4 block + j
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
            const int blk = l >> 2, j = l & 3;
            const float want = (float)(100 + 4 * blk + i) * (float)(1000 + 4 * blk + j);
            if (h[l * 4 + i] != want) ++bad;
        }
        printf("mfma 4x4x1 16b layout: D reg i of lane 4b+j = A(lane 4b+i) * B(lane 4b+j): %s (%d mismatches); lane 5: %.0f %.0f %.0f %.0f\n", bad ? "NO" : "yes", bad, h[20], h[21], h[22], h[23]);
    }
    const float* fin = reinterpret_cast<const float*>(din);
    for (int threads : {256, 512, 1024}) {
        const int lds = 100 * 1024;       // one workgroup per CU
        if (timed<float2>("fft32 DIF (fft_reg) + scale", k_fft32<0>, threads, lds, 64, 1.0, "transform", (const float2*)din)) return 1;
        if (timed<float2>("fft32 DIT L-F + scale", k_fft32<1>, threads, lds, 64, 1.0, "transform", (const float2*)din)) return 1;
        if (timed<float>("v_pk_fma_f32 x16 chains", k_valu<1>, threads, lds, 256, 16.0, "pk_fma", fin)) return 1;
        if (timed<float>("v_fma_f32 x32 chains", k_valu<0>, threads, lds, 256, 32.0, "fma", fin)) return 1;
        if (timed<float>("mfma 4x4x1 1 acc", k_mfma<1>, threads, lds, 256, 8.0, "mfma", fin)) return 1;
        if (timed<float>("mfma 4x4x1 2 acc", k_mfma<2>, threads, lds, 256, 16.0, "mfma", fin)) return 1;
        if (timed<float>("mfma 4x4x1 4 acc", k_mfma<4>, threads, lds, 256, 32.0, "mfma", fin)) return 1;
    }
    // MFMA beside VALU
    for (int threads : {256, 1024}) for (int wm : {0, 1}) {
        const int blocks = 256, lds = 100 * 1024, iters = 256;
        unsigned long long* cyc; float* outp;
        CK(hipMalloc(&cyc, blocks * (threads / 64) * 8)); CK(hipMalloc(&outp, (size_t)blocks * threads * 4));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mfma_valu), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_mfma_valu, dim3(blocks), dim3(threads), lds, 0, fin, outp, cyc, iters, wm);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks * (threads / 64));
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("8 pk_fma %s 8 mfma4x4 per iteration: waves/SIMD %d  %.1f cyc per iteration\n", wm ? "+" : "without", threads / 256, (double)h[h.size() / 2] / iters);
        CK(hipFree(cyc)); CK(hipFree(outp));
    }
    return 0;
}
