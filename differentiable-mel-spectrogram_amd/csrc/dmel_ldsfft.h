// dmel_ldsfft.h -- in-place complex FFT of one sequence held in LDS, shared by the long-transform kernel (dmel_aux.hip)
// and the waveform-gradient kernel (dmel_xgrad.hip).  Radix-2 stages are fused in pairs (radix-4 butterflies in registers):
// half the barriers and half the LDS traffic of a plain radix-2 loop, with exactly the radix-2 data flow, so the output
// order stays the bit-reversed one that __brev addresses.
//   lds_fft_dif: natural order in, bit-reversed order out (decimation in frequency)
//   lds_fft_dit: bit-reversed order in, natural order out (decimation in time)
// Both use the forward kernel exp(-2 pi i k n / N); `twiddle(k)` returns exp(-2 pi i k / N) for 0 <= k < N/2 (from LDS or
// from global memory).  Every stage ends with __syncthreads(); the caller synchronises before the first stage.
#pragma once
#include <hip/hip_runtime.h>

namespace dmel {

__device__ __forceinline__ float2 c_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 c_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 c_mul(float2 a, float2 w) { return make_float2(fmaf(a.x, w.x, -(a.y * w.y)), fmaf(a.x, w.y, a.y * w.x)); }
__device__ __forceinline__ float2 c_mul_mi(float2 a) { return make_float2(a.y, -a.x); }          // a * (-i)

// One radix-4 group of two fused DIF stages (spans 2s and s) on four values, twiddles w1 = W^(j ta), w2 = W^(2 j ta): e[1], e[3]
// leave multiplied as the in-place code above does.
__device__ __forceinline__ void dif4(float2& e0, float2& e1, float2& e2, float2& e3, float2 w1, float2 w2)
{
    const float2 a0 = c_add(e0, e2), a2 = c_mul(c_sub(e0, e2), w1);
    const float2 a1 = c_add(e1, e3), a3 = c_mul_mi(c_mul(c_sub(e1, e3), w1));
    e0 = c_add(a0, a1); e1 = c_mul(c_sub(a0, a1), w2);
    e2 = c_add(a2, a3); e3 = c_mul(c_sub(a2, a3), w2);
}
__device__ __forceinline__ void dit4(float2& e0, float2& e1, float2& e2, float2& e3, float2 u1, float2 v)
{
    const float2 c1 = c_mul(e1, u1), c3 = c_mul(e3, u1);
    const float2 a0 = c_add(e0, c1), a1 = c_sub(e0, c1), a2 = c_add(e2, c3), a3 = c_sub(e2, c3);
    const float2 d2 = c_mul(a2, v), d3 = c_mul_mi(c_mul(a3, v));
    e0 = c_add(a0, d2); e2 = c_sub(a0, d2);
    e1 = c_add(a1, d3); e3 = c_sub(a1, d3);
}

template <int THREADS, class TW, class LD>
__device__ __forceinline__ void lds_fft_dif_head(float2* Z, int N, int ns, int tid, TW&& twiddle, LD&& first, int nc0);
template <int THREADS, class TW, class ST>
__device__ __forceinline__ void lds_fft_dit_tail(float2* Z, int N, int ns, int tid, TW&& twiddle, ST&& last);

// R16: four stages per pass where four are left (16 values per thread) -- half the passes and barriers of the two-stage passes:
// -10 % for the 1024-thread workgroups of dmel_big.hip; the 256-thread workgroups of dmel_xgrad_frames_kernel measured
// slower with it (74 -> 94 us at BASELINE config 2) and keep two stages per pass.
template <int THREADS, bool R16 = true, class TW>
__device__ __forceinline__ void lds_fft_dif(float2* Z, int N, int logN, int tid, TW&& twiddle)
{
    int top = N >> 1;                 // span of the next radix-2 stage
    if (logN & 1) {                   // odd number of stages: one plain radix-2 stage first
        for (int i = tid; i < (N >> 1); i += THREADS) {
            const float2 a = Z[i], c = Z[i + top];
            Z[i] = c_add(a, c);
            Z[i + top] = c_mul(c_sub(a, c), twiddle(i));
        }
        __syncthreads();
        top >>= 1;
    }
    if constexpr (R16) {
        // the remaining (even number of) stages four at a time where four are left, then one two-stage pass
        lds_fft_dif_head<THREADS>(Z, N, logN & ~1, tid, twiddle, [&](int n) { return Z[n]; }, top << 1);
    } else {
        // fused pairs (span 2s, span s)
        for (int s = top >> 1; s >= 1; s >>= 2) {
            const int ta = N / (4 * s);
            for (int i = tid; i < (N >> 2); i += THREADS) {
                const int j = i & (s - 1);
                const int lo = ((i - j) << 2) + j;
                float2 e0 = Z[lo], e1 = Z[lo + s], e2 = Z[lo + 2 * s], e3 = Z[lo + 3 * s];
                dif4(e0, e1, e2, e3, twiddle(j * ta), twiddle(2 * j * ta));
                Z[lo] = e0; Z[lo + s] = e1; Z[lo + 2 * s] = e2; Z[lo + 3 * s] = e3;
            }
            __syncthreads();
        }
    }
}

template <int THREADS, bool R16 = true, class TW>
__device__ __forceinline__ void lds_fft_dit(float2* Z, int N, int logN, int tid, TW&& twiddle)
{
    int s = 1;                        // span of the next radix-2 stage
    if (logN & 1) {                   // odd number of stages: one plain radix-2 stage (span 1, twiddle 1) first
        for (int i = tid; i < (N >> 1); i += THREADS) {
            const float2 a = Z[2 * i], c = Z[2 * i + 1];
            Z[2 * i] = c_add(a, c);
            Z[2 * i + 1] = c_sub(a, c);
        }
        __syncthreads();
        s = 2;
    }
    if constexpr (R16) {
        lds_fft_dit_tail<THREADS>(Z, N, logN & ~1, tid, twiddle, [](int, float2 v) { return v; });
    } else {
        // fused pairs (span s, span 2s)
        for (; 4 * s <= N; s <<= 2) {
            const int tb = N / (4 * s);
            for (int i = tid; i < (N >> 2); i += THREADS) {
                const int j = i & (s - 1);
                const int lo = ((i - j) << 2) + j;
                float2 e0 = Z[lo], e1 = Z[lo + s], e2 = Z[lo + 2 * s], e3 = Z[lo + 3 * s];
                dit4(e0, e1, e2, e3, twiddle(2 * j * tb), twiddle(j * tb));
                Z[lo] = e0; Z[lo + s] = e1; Z[lo + 2 * s] = e2; Z[lo + 3 * s] = e3;
            }
            __syncthreads();
        }
    }
}

// ---- transforms longer than LDS: the stages whose span reaches across blocks run on the sequence in global memory, the rest on one
// block at a time in LDS.  `ns` (even) = stages outside the blocks; blocks are contiguous runs of N >> ns points.
// lds_fft_dif_head: the FIRST ns stages of an N-point DIF (spans N/2 ... N >> ns); what remains are 2^ns independent DIF
// transforms of the blocks, each with the twiddles of its own length.
// `first(n)` supplies element n of the input for the FIRST pass (the sequence need not have been written to Z before).
// FOUR stages per pass where four are left (16 values per thread: e[a][b] = Z[base + j + a sA + b sB], sA = 4 sB): a pass over a
// sequence in global memory moves all of it through L2, whatever it computes.
template <int THREADS, class TW, class LD>
__device__ __forceinline__ void lds_fft_dif_head(float2* Z, int N, int ns, int tid, TW&& twiddle, LD&& first, int nc0)
{
    int left = ns;
    int nc = nc0;                                 // length of the independent transforms at this level (N, or less behind earlier stages)
    for (; left >= 4; left -= 4, nc >>= 4) {
        const int sA = nc >> 2, sB = nc >> 4, tA = N / nc, tB = 4 * tA;
        const bool p0 = left == ns;
        for (int i = tid; i < (N >> 4); i += THREADS) {
            const int j = i & (sB - 1);
            const int base = ((i - j) << 4) + j;
            float2 e[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { const int n = base + a * sA + b * sB; e[a][b] = p0 ? first(n) : Z[n]; }
#pragma unroll
            for (int b = 0; b < 4; ++b) {             // spans nc/2, nc/4: position inside the transform = j + b sB
                const int ja = j + b * sB;
                dif4(e[0][b], e[1][b], e[2][b], e[3][b], twiddle(ja * tA), twiddle(2 * ja * tA));
            }
            const float2 w1 = twiddle(j * tB), w2 = twiddle(2 * j * tB);
#pragma unroll
            for (int a = 0; a < 4; ++a) dif4(e[a][0], e[a][1], e[a][2], e[a][3], w1, w2);       // spans nc/8, nc/16
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) Z[base + a * sA + b * sB] = e[a][b];
        }
        __syncthreads();
    }
    for (int s = nc >> 2; left > 0; s >>= 2, left -= 2) {
        const int ta = N / (4 * s);
        const bool p0 = left == ns;
        for (int i = tid; i < (N >> 2); i += THREADS) {
            const int j = i & (s - 1);
            const int lo = ((i - j) << 2) + j;
            float2 e0 = p0 ? first(lo) : Z[lo], e1 = p0 ? first(lo + s) : Z[lo + s];
            float2 e2 = p0 ? first(lo + 2 * s) : Z[lo + 2 * s], e3 = p0 ? first(lo + 3 * s) : Z[lo + 3 * s];
            dif4(e0, e1, e2, e3, twiddle(j * ta), twiddle(2 * j * ta));
            Z[lo] = e0; Z[lo + s] = e1; Z[lo + 2 * s] = e2; Z[lo + 3 * s] = e3;
        }
        __syncthreads();
    }
}

// lds_fft_dit_tail: the LAST ns stages of an N-point DIT (spans N >> ns ... N/2), after the blocks of N >> ns points have been
// transformed (DIT, bit-reversed in, natural out) one by one.
// `last(k, v)` turns output k of the LAST pass into what is stored at Z[k].
template <int THREADS, class TW, class ST>
__device__ __forceinline__ void lds_fft_dit_tail(float2* Z, int N, int ns, int tid, TW&& twiddle, ST&& last)
{
    int left = ns;
    int nc = N >> ns;                             // length of the transforms already done
    if (left & 2) {                               // an odd number of stage pairs: the single pair first, the radix-16 passes behind it
        const int s = nc, tb = N / (4 * s);
        const bool pl = left == 2;
        for (int i = tid; i < (N >> 2); i += THREADS) {
            const int j = i & (s - 1);
            const int lo = ((i - j) << 2) + j;
            float2 e0 = Z[lo], e1 = Z[lo + s], e2 = Z[lo + 2 * s], e3 = Z[lo + 3 * s];
            dit4(e0, e1, e2, e3, twiddle(2 * j * tb), twiddle(j * tb));
            Z[lo] = pl ? last(lo, e0) : e0; Z[lo + s] = pl ? last(lo + s, e1) : e1;
            Z[lo + 2 * s] = pl ? last(lo + 2 * s, e2) : e2; Z[lo + 3 * s] = pl ? last(lo + 3 * s, e3) : e3;
        }
        __syncthreads();
        left -= 2; nc <<= 2;
    }
    for (; left >= 4; left -= 4, nc <<= 4) {
        // e[a][b] = Z[base + j + b sB + a sA], sB = nc (spans nc, 2 nc over b), sA = 4 nc (spans 4 nc, 8 nc over a)
        const int sB = nc, sA = nc << 2, tB = N / (4 * sB), tA = N / (4 * sA);
        const bool pl = left == 4;
        for (int i = tid; i < (N >> 4); i += THREADS) {
            const int j = i & (sB - 1);
            const int base = ((i - j) << 4) + j;
            float2 e[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) e[a][b] = Z[base + a * sA + b * sB];
            const float2 u1 = twiddle(2 * j * tB), v = twiddle(j * tB);
#pragma unroll
            for (int a = 0; a < 4; ++a) dit4(e[a][0], e[a][1], e[a][2], e[a][3], u1, v);
#pragma unroll
            for (int b = 0; b < 4; ++b) {             // position inside the transform of 4 sA points = j + b sB
                const int ja = j + b * sB;
                dit4(e[0][b], e[1][b], e[2][b], e[3][b], twiddle(2 * ja * tA), twiddle(ja * tA));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { const int k = base + a * sA + b * sB; Z[k] = pl ? last(k, e[a][b]) : e[a][b]; }
        }
        __syncthreads();
    }
}

}  // namespace dmel
