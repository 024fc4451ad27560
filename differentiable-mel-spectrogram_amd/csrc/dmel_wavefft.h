// dmel_wavefft.h -- building blocks of the wave-cooperative FFT shared by the fused forward (dmel_fwd.hip) and the
// gradient w.r.t. the waveform (dmel_xgrad.hip): packed complex arithmetic, the register radix-R transform, DPP lane
// exchanges, the fixed-order wave sum.  gfx950 device code only.
#pragma once
#include "dmel_kernels.h"

namespace dmel {

typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---- compile-time helpers -------------------------------------------------------------------
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}
constexpr int ilog2(int v) { int r = 0; while (v > 1) { v >>= 1; ++r; } return r; }
constexpr int bitrev(int i, int bits) { int r = 0; for (int b = 0; b < bits; ++b) { r = (r << 1) | (i & 1); i >>= 1; } return r; }

// cos(2 pi j / 64), j = 0..16
constexpr float kCos64[17] = {
    1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f,
    0.92387953251128675613f, 0.88192126434835502971f, 0.83146961230254523708f, 0.77301045336273696081f,
    0.70710678118654752440f, 0.63439328416364549822f, 0.55557023301960222474f, 0.47139673682599764856f,
    0.38268343236508977173f, 0.29028467725446236764f, 0.19509032201612826785f, 0.09801714032956060199f,
    0.0f};
constexpr float cos64(int j)   // j in [0, 32]
{
    return j <= 16 ? kCos64[j] : -kCos64[32 - j];
}
constexpr float sin64(int j)   // j in [0, 32]
{
    return j <= 16 ? kCos64[16 - j] : kCos64[j - 16];
}

// Complex numbers live in one 64-bit register pair (re, im) through every stage, so that each complex
// add / multiply is one or two packed VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with
// op_sel / neg modifiers for the swaps and sign flips) and no register shuffling is needed between stages.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f rot_mi(v2f a) { return v2f{a.y, -a.x}; }          // a * (-i)
__device__ __forceinline__ v2f splat(float c) { return v2f{c, c}; }

// a * exp(-2 pi i TW / 64), TW in [0, 32)
template <int TW> __device__ __forceinline__ v2f cmul_tw(v2f a)
{
    if constexpr (TW == 0) return a;
    else if constexpr (TW == 16) return rot_mi(a);
    else {
        // (a.x c + a.y s, a.y c - a.x s) = a * (c, c) + a.yx * (s, -s): two packed ops, signs in the constant
        constexpr float c = (TW == 8) ? 0.70710678118654752440f : (TW == 24) ? -0.70710678118654752440f : cos64(TW);
        constexpr float sn = (TW == 8 || TW == 24) ? 0.70710678118654752440f : sin64(TW);
        return __builtin_elementwise_fma(a.yx, v2f{sn, -sn}, a * splat(c));
    }
}

// Radix-2 decimation-in-frequency FFT of R points held in registers; logical output q ends up in
// v[bitrev(q)].  Fully unrolled: every index and twiddle is a compile-time constant.
// The twiddle -i (j == SPAN/2) is never applied where it arises: the element is left unrotated and the
// rotation is folded into its only consumer, the j == 0 butterfly of the odd block one stage later,
// as a +- (b.y, -b.x) packed FMA with a constant -- no swap / sign-flip instructions are issued.
template <int R, int SPAN = R / 2> __device__ __forceinline__ void fft_reg(v2f (&v)[R])
{
    if constexpr (SPAN >= 1) {
        static_for<0, R / (2 * SPAN)>([&](auto blk) {
            constexpr int bi = decltype(blk)::value;
            constexpr int base = bi * 2 * SPAN;
            static_for<0, SPAN>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                const v2f a = v[base + j], b = v[base + j + SPAN];
                if constexpr (j == 0 && (bi & 1) != 0 && 2 * SPAN < R) {
                    // b carries a pending factor -i: a +- rot_mi(b)
                    v[base + j] = __builtin_elementwise_fma(b.yx, v2f{1.f, -1.f}, a);
                    v[base + j + SPAN] = __builtin_elementwise_fma(b.yx, v2f{-1.f, 1.f}, a);
                } else {
                    v[base + j] = a + b;
                    if constexpr (2 * j == SPAN) v[base + j + SPAN] = a - b;          // -i applied by the consumer
                    else v[base + j + SPAN] = cmul_tw<j * 32 / SPAN>(a - b);
                }
            });
        });
        fft_reg<R, SPAN / 2>(v);
    }
}

// cos(2 pi j / 64) in double: the constants of fft_reg_dit are quotients, rounded to fp32 once
constexpr double kCos64d[17] = {
    1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494,
    0.92387953251128675613, 0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081,
    0.70710678118654752440, 0.63439328416364549822, 0.55557023301960222474, 0.47139673682599764856,
    0.38268343236508977173, 0.29028467725446236764, 0.19509032201612826785, 0.09801714032956060199,
    0.0};
constexpr double cos64d(int j) { return j <= 16 ? kCos64d[j] : -kCos64d[32 - j]; }   // j in [0, 32]
constexpr double sin64d(int j) { return j <= 16 ? kCos64d[16 - j] : kCos64d[j - 16]; }

// One decimation-in-time butterfly (a, b) -> (a + w b, a - w b), w = exp(-2 pi i TW / 64), TW in [0, 32), in THREE packed
// FMAs instead of four packed operations (Linzer & Feig): with w = c (1 - i t), t = tan,
//   q = b + t (b.y, -b.x)            one FMA, the rotation rides on the operand selectors
//   a +- c q                          two FMAs
// and for the angles nearer to -i than to 1, w = s (k - i), k = cot:  q = k b + (b.y, -b.x).  |t|, |k| <= 1: the constants are
// as well conditioned as (cos, sin) themselves.  TW = 0 and TW = 16 (w = -i) stay two packed operations.
template <int TW> __device__ __forceinline__ void bfly_dit(v2f& a, v2f& b)
{
    const v2f a0 = a, b0 = b;
    if constexpr (TW == 0) { a = a0 + b0; b = a0 - b0; }
    else if constexpr (TW == 16) {
        a = __builtin_elementwise_fma(b0.yx, v2f{1.f, -1.f}, a0);
        b = __builtin_elementwise_fma(b0.yx, v2f{-1.f, 1.f}, a0);
    } else {
        constexpr double c = cos64d(TW), s = sin64d(TW);
        if constexpr ((c < 0 ? -c : c) >= s) {
            constexpr float t = (float)(s / c), cf = (float)c;
            const v2f q = __builtin_elementwise_fma(b0.yx, v2f{t, -t}, b0);
            a = __builtin_elementwise_fma(q, splat(cf), a0);
            b = __builtin_elementwise_fma(q, splat(-cf), a0);
        } else {
            // k b + rot(b) = rot(b - k rot(b)), rot(v) = (v.y, -v.x): the same shape as the tangent form -- swizzles on the first
            // operand only (hipcc folds those into op_sel; a swizzled, negated THIRD operand cost a v_xor and a v_mov each)
            constexpr float k = (float)(c / s), sf = (float)s;
            const v2f p = __builtin_elementwise_fma(b0.yx, v2f{-k, k}, b0);
            a = __builtin_elementwise_fma(p.yx, v2f{sf, -sf}, a0);
            b = __builtin_elementwise_fma(p.yx, v2f{-sf, sf}, a0);
        }
    }
}

// Radix-2 decimation-in-time FFT of R points held in registers, same interface as fft_reg: natural order in, logical output q
// in v[bitrev(q)] (the stages address the array through the bit reversal, a compile-time renaming).  194 packed instructions for
// R = 32 against fft_reg's 228, 482 against 580 for R = 64.
template <int R, int H = 1> __device__ __forceinline__ void fft_reg_dit(v2f (&v)[R])
{
    if constexpr (H < R) {
        constexpr int LB = ilog2(R);
        static_for<0, R / (2 * H)>([&](auto blk) {
            constexpr int base = decltype(blk)::value * 2 * H;
            static_for<0, H>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                bfly_dit<j * 32 / H>(v[bitrev(base + j, LB)], v[bitrev(base + j + H, LB)]);
            });
        });
        fft_reg_dit<R, 2 * H>(v);
    }
}

// a * w for a table twiddle w = (re, im): a * (re, re) + a.yx * (-im, im)
__device__ __forceinline__ v2f cmul(v2f a, float2 w)
{
    return __builtin_elementwise_fma(a.yx, v2f{-w.y, w.y}, a * splat(w.x));
}

// value of lane (l ^ 1) / (l ^ 2) inside each quad: DPP quad_perm, no LDS traffic
__device__ __forceinline__ float quad_xor1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

// Sum over the 64 lanes in a fixed order, same value in every lane: four DPP steps inside each row of 16
// (xor 1, xor 2, half-mirror, mirror: no LDS round trips, unlike __shfl_xor = ds_bpermute), then the four
// row sums through scalar registers.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);     // row_half_mirror
    v += dpp_f<0x140>(v);     // row_mirror
    const int vi = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return (r0 + r1) + (r2 + r3);
}

template <int R, int C> __device__ __forceinline__ int z_index(int k)
{
    if constexpr (C == 1) return k;
    else return k + (k / (R * R)) * 4;
}

// Raw buffer loads: 32-bit offsets from an SGPR descriptor, and the hardware range check returns 0
// for offsets outside [0, bytes) -- a negative sample index wraps to a huge unsigned offset.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }   // v_med3_i32
// (the 64-bit form __builtin_amdgcn_raw_buffer_load_b64 is mis-lowered to a single dword load by hipcc 7.2:
// twiddle tables are therefore read with ordinary float2 loads)

// exchange among the lanes of one wave through LDS: release, execution barrier, acquire
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One complex FFT of N = R * R * C points by the G = N / R lanes of (a part of) one wave, spectrum handed to
// `store(IC<p1>, qp, p2, Z[qp + R p1 + R^2 p2])` (p1 at compile time: the consumer's addresses are a per-lane base + constants)
// (tools/wavefft_sim.py is the index model; the plain layout of the fused forward: FftPlan::PAIRING = 0, SPLIT = 0).
//   in : lane lg holds points n = lg + G a in z[a]
//   sl : the frame's LDS slot, R * ex_stride(G, C) complex entries, for the one transposition
//   tw1: (R, G) w_N^(lg q), tw2: (R, C) w_G^(r p1)      (the plan's tables, dmel_api.cpp)
//   out: lane (qp, r) = (lg / C, lg % C) produces Z[qp + R p1 + R^2 p2] for p1 = 0 .. R - 1
// The caller fences before reusing `sl` (store may write to it: every read of the transposition precedes the first store).
template <int R, int C, int G, class Store>
__device__ __forceinline__ void wave_fft(v2f (&z)[R], v2f* sl, int lg, const float2* __restrict__ tw1, const float2* tw2, Store&& store)
{
    constexpr int LB = ilog2(R), EXS = ex_stride(G, C);
    const int qp = lg / C, r = lg % C;
    fft_reg<R>(z);
    // first-stage twiddles by powers where the table is large (rows q = 1 and q = 8a loaded, the rest multiplied)
    constexpr bool TW1_POW = (R >= 16);
    v2f wb[TW1_POW ? 8 : 1];
    if constexpr (TW1_POW) {
        const float2 w1 = tw1[G + lg];
        wb[1] = v2f{w1.x, w1.y};
        static_for<2, 8>([&](auto bb) { constexpr int b = decltype(bb)::value; wb[b] = cmul(wb[b - 1], w1); });
    }
    static_for<0, R>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        v2f v = z[bitrev(q, LB)];
        if constexpr (q != 0) {
            if constexpr (!TW1_POW) v = cmul(v, tw1[q * G + lg]);
            else {
                constexpr int a8 = q / 8, b8 = q % 8;
                if constexpr (a8 == 0) v = cmul(v, float2{wb[b8].x, wb[b8].y});
                else {
                    const float2 anchor = tw1[(8 * a8) * G + lg];
                    if constexpr (b8 == 0) v = cmul(v, anchor);
                    else { const v2f t = cmul(wb[b8], anchor); v = cmul(v, float2{t.x, t.y}); }
                }
            }
        }
        sl[q * EXS + lg] = v;
    });
    wave_sync();
    v2f u[R];
    static_for<0, R>([&](auto bb) { constexpr int bi = decltype(bb)::value; u[bi] = sl[qp * EXS + r + C * bi]; });
    wave_sync();
    fft_reg<R>(u);
    const v2f rot_f = splat((C == 4 && r == 3) ? 0.f : 1.f);
    const v2f rot_e = (C == 4 && r == 3) ? v2f{1.f, -1.f} : v2f{0.f, 0.f};
    static_for<0, R>([&](auto pp1) {
        constexpr int p1 = decltype(pp1)::value;
        v2f v = u[bitrev(p1, LB)];
        int p2 = 0;
        if constexpr (C > 1 && p1 != 0) v = cmul(v, tw2[p1 * C + r]);
        if constexpr (C == 2) {
            const v2f o = v2f{quad_xor1(v.x), quad_xor1(v.y)};
            v = __builtin_elementwise_fma(splat((r == 0) ? 1.f : -1.f), v, o);
            p2 = r;
        } else if constexpr (C == 4) {
            v2f o = v2f{quad_xor2(v.x), quad_xor2(v.y)};
            v2f t = __builtin_elementwise_fma(splat((r < 2) ? 1.f : -1.f), v, o);
            t = __builtin_elementwise_fma(t.yx, rot_e, t * rot_f);      // lane r == 3: t * (-i); others: t
            o = v2f{quad_xor1(t.x), quad_xor1(t.y)};
            v = __builtin_elementwise_fma(splat(((r & 1) == 0) ? 1.f : -1.f), t, o);
            p2 = ((r & 1) << 1) | (r >> 1);
        }
        store(pp1, qp, p2, v);                                       // Z[qp + R p1 + R^2 p2]
    });
}

}  // namespace dmel
