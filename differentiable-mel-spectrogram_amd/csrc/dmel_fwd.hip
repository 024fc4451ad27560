// dmel_fwd.hip -- fused forward of the DMEL layer for gfx950 (MI355X).
//
// One workgroup (4 or 8 waves) produces a tile of consecutive STFT frames of one clip, from waveform
// samples to log-mel values, without touching HBM in between:
//
//   phase 0: the window table (w, dw/dlambd) is built in LDS and, for clips up to 32768 samples, the clip
//            is summed for the DC removal (longer clips: partial sums from dmel_prep_kernel).
//   phase 1 (VALU + LDS, per wave): DC-removed, Gaussian-windowed frame -> complex FFT.  A wave
//            holds R points per lane and transforms 64/G frames at a time: radix-R butterflies in
//            registers, one transposition through LDS, radix-R again, and a radix-C stage across
//            adjacent lanes with DPP quad permutes (tools/wavefft_sim.py is the index model).
//            Two real sequences ride in one complex FFT: (x~ w, x~ dw/dlambd) when the tangent is
//            wanted (training), two neighbouring frames otherwise.  The pairing pass separates them
//            once per bin and leaves PD[k] = (|X|^2, d|X|^2/dlambd) (or the two frames' |X|^2) in LDS.
//            n_fft 2048 and 4096 use the compact layout (FftPlan::BPERM / SPLIT): the transposition moves one
//            plane of floats at a time and the pairing pass gets Z[N-k] from the lane that holds it
//            (ds_bpermute_b32), so a frame in flight needs N*4 bytes of LDS instead of N*8.
//   phase 2 (MFMA): the mel contraction of models.py:53.  A operands are plain reads of PD,
//            B fragments are the non-zero 4x16 blocks of the filterbank (prefetched into registers
//            before phase 1), v_mfma_f32_16x16x4_f32 accumulates exact fp32.  With 4 waves each wave
//            owns two mel tiles; with 8 waves each owns one tile and the k-steps of the wide tiles are dealt
//            over the waves with narrow (or no) tiles: their sums reach the owner through LDS.
//   epilogue: scale, log(mel + eps) (models.py:73), tangent d out / d lambd, straight from the
//            accumulators into the (B,1,M,T) layout of models.py:36.
//
// Reference semantics restated here: models.py:33-56 (layer forward), time_frequency.py:21-30
// (window), :32-58 (STFT, |.|^2), models.py:73 (log).
#include "dmel_kernels.h"
#include "dmel_wavefft.h"


namespace dmel {


#ifndef DMEL_TWC
#define DMEL_TWC 8
#endif
#ifndef DMEL_FWD_PART
#define DMEL_FWD_PART 0
#endif
#if DMEL_FWD_PART == 0
// ---- prep kernel: per-clip partial sums (DC removal, models.py:38) + window tables -----------
__global__ void __launch_bounds__(kThreads) dmel_prep_kernel(PrepParams p)
{
    __shared__ double red[kThreads];
    const int tid = threadIdx.x;
    if (blockIdx.y == (unsigned)p.B) {
        // window block: time_frequency.py:21-30 in fp32; second entry = w d^2 2^(-2e), the tangent window up to the
        // factor lam_tangent_scale() that the forward kernels apply in their epilogue (fp64 here: d^2 needs 30 bits)
        if (blockIdx.x != 0) return;
        const LamState ls = lam_prologue(p.lam, p.N, false);
        if (ls.action != kLamRun) return;
        const float denom = ls.denom;
        const double s2 = (double)ls.s2;
        double s_ww = 0.0, s_wd = 0.0;
        for (int n = tid; n < p.N; n += kThreads) {
            const float d = (float)n - p.center;
            const float t = d / denom;
            float w = expf(-0.5f * (t * t));
            if (p.win_half && (n < p.N / 4 || n >= 3 * p.N / 4)) w = 0.f;
            const double dw = (double)w * (double)d * (double)d * s2;
            p.win2[n] = make_float2(w, (float)dw);
            s_ww += (double)w * (double)w;
            s_wd += (double)w * dw;
        }
        if (!p.normalize) return;
        // time_frequency.py:25: w / sqrt(sum w^2); derivative of the quotient
        red[tid] = s_ww; __syncthreads();
        for (int o = kThreads / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        const double ww = red[0]; __syncthreads();
        red[tid] = s_wd; __syncthreads();
        for (int o = kThreads / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        const double wd = red[0];
        const double nrm = sqrt(ww);
        for (int n = tid; n < p.N; n += kThreads) {
            const double w = (double)p.win2[n].x;
            const float d = (float)n - p.center;
            const double dwe = w * (double)d * (double)d * s2;       // recomputed in fp64 (the table holds it rounded)
            p.win2[n] = make_float2((float)(w / nrm), (float)(dwe / nrm - w * wd / (nrm * nrm * nrm)));
        }
        return;
    }
    const int b = blockIdx.y, c = blockIdx.x;
    const long long lo = (long long)c * p.chunk;
    long long hi = lo + p.chunk; if (hi > p.L) hi = p.L;
    const float* xbase = p.x;
    if (p.x_ind) { typedef const float* cfp; xbase = *(const __attribute__((address_space(4))) cfp*)p.x_ind; }      // the batch by address
    const float* xb = xbase + (size_t)b * p.L;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    long long i = lo + tid;
    if (((reinterpret_cast<uintptr_t>(xb + lo)) & 15) == 0) {
        // 16-byte aligned chunk: dwordx4 loads, four per thread in flight before anything is added (round 4: a plain loop waited
        // for every load before it issued the next -- the kernel took 6.9 us for 5 MB at the reference's ESC-50 shape, a third of that
        // step; the sums' order is fixed either way)
        const float4* x4 = reinterpret_cast<const float4*>(xb + lo);
        const long long n4 = (hi - lo) / 4;
        for (long long base = 0; base < n4; base += 4 * kThreads) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long q = base + tid + (long long)u * kThreads; v[u] = x4[q < n4 ? q : n4 - 1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = base + tid + (long long)u * kThreads < n4;
                acc0 += ok ? v[u].x : 0.f; acc1 += ok ? v[u].y : 0.f; acc2 += ok ? v[u].z : 0.f; acc3 += ok ? v[u].w : 0.f;
            }
        }
        i = lo + n4 * 4 + tid;
    } else {
        for (; i + 3 * kThreads < hi; i += 4 * kThreads) {
            acc0 += xb[i]; acc1 += xb[i + kThreads]; acc2 += xb[i + 2 * kThreads]; acc3 += xb[i + 3 * kThreads];
        }
    }
    for (; i < hi; i += kThreads) acc0 += xb[i];
    // fixed order: the four accumulators of a thread, the 64 lanes of a wave (DPP butterfly in fp64, the same value in every lane), the four
    // waves ascending -- one barrier instead of the eight of a 256-entry tree in LDS
    double s = ((double)acc0 + (double)acc1) + ((double)acc2 + (double)acc3);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) p.psum[(size_t)b * p.nchunks + c] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

hipError_t launch_prep(const PrepParams& p, hipStream_t s)
{
    dim3 grid(p.nchunks, p.B + 1);
    hipLaunchKernelGGL(dmel_prep_kernel, grid, dim3(kThreads), 0, s, p);
    return hipGetLastError();
}

#endif   // DMEL_FWD_PART == 0

// ---- fused forward --------------------------------------------------------------------------

// log(me), me = mel + eps, for the fused epilogue (models.py:73).  With the reference's eps (1e-10, any eps >= 1e-30) the argument is a
// normal number: v_log_f32 (log2, 1 ulp) times ln 2 -- 2 instructions against ~12 of logf(), whose extra work is the scaling of
// denormal arguments; absolute error <= 3e-6 at |log| = 23 (the 1e-4 bar of the path is absolute in the log domain).  eps below
// that (or negative): logf().  The choice is uniform over the launch.
__device__ __attribute__((noinline)) float slow_log(float me) { return logf(me); }   // (a call: the compiler does not fold the two paths into a select)
__device__ __forceinline__ float fast_log(float me, float eps)
{
    if (eps >= 1e-30f) return __builtin_amdgcn_logf(me) * 0.69314718055994530942f;
    return slow_log(me);
}

#ifdef DMEL_STAMPS
// Diagnostic build only (tools/stamps.py): s_memtime stamps of every wave at the phase boundaries of the
// fused kernel, kept in a buffer nothing else reads.  Never compiled into libdmel_hip.so.
constexpr int kStampSlots = 32;     // 0-2 prologue, 12-13 placement, 3-11 tile 0, 16 + (2..11) tile 1 (2 = tile start)
__device__ unsigned long long g_stamps[4096 * 8 * kStampSlots];
__device__ __forceinline__ void stamp(int wgid, int wave, int lane, int idx)
{
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0 && wgid < 4096) g_stamps[((size_t)wgid * 8 + wave) * kStampSlots + idx] = t;
}
#define STAMP(i) stamp(blockIdx.x, wave, lane, i)
// where the workgroup runs: HW_ID (wave / SIMD / CU / SH / SE ids) and XCC_ID, slots 12 and 13
__device__ __forceinline__ void stamp_place(int wgid, int wave, int lane)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (lane == 0 && wgid < 4096) {
        g_stamps[((size_t)wgid * 8 + wave) * kStampSlots + 12] = hw;
        g_stamps[((size_t)wgid * 8 + wave) * kStampSlots + 13] = xcc;
    }
}
#define STAMP_PLACE() stamp_place(blockIdx.x, wave, lane)
#else
#define STAMP(i) do {} while (0)
#define STAMP_PLACE() do {} while (0)
#endif

// TPW = tiles one workgroup produces, one after the other.  With TPW = 2 the prologue (lambd, window table, clip sum: a third
// of a tile's lifetime, mostly spent waiting for memory) is paid once for twice the frames, the samples of the second tile
// are requested while the first is transformed, and a launch needs half the workgroups (one round of resident workgroups
// instead of two at BASELINE config 2).
template <int N, int MODE, int TPW>
__global__ void __launch_bounds__((geom_mode<N, MODE>().THREADS), (geom_mode<N, MODE>().MINW)) dmel_fwd_kernel(FwdParams p)
{
    constexpr FftGeom g = geom_mode<N, MODE>();
    constexpr int R = g.R, C = g.C, G = g.G, FPW = g.FPW, PASSES = g.PASSES, SLOTS = g.SLOTS, MT = g.MT;
    constexpr int WAVES = g.WAVES, NLOC = g.NLOC, THREADS = g.THREADS;
    constexpr int LB = ilog2(R);
    constexpr int EXS = g.EX_STRIDE, SS = g.SLOT_STRIDE_F2;
    constexpr bool PAIR = (MODE == kInfer || MODE == kSpec);    // two frames share one complex FFT
    constexpr bool IS_SPEC = (MODE == kSpec || MODE == kSpecTrain);
    constexpr bool HSPLIT = (MODE == kTrainH);                  // dense contraction on the bf16 matrix pipe: PD kept as four bf16 planes
#ifndef DMEL_DIT_MASK
#define DMEL_DIT_MASK (~0)
#endif
    // register radix core: decimation in time with Linzer-Feig butterflies (194 packed operations per 32 points) or the round-4
    // decimation in frequency (228); bit log2(N) of DMEL_DIT_MASK selects (diagnostic builds)
    // (measured, training: n_fft 4096 145.2 -> 138.5 us at the reference's ESC-50 shape, n_fft 1024 -1 % in many-round launches; n_fft 2048
    // with the round-4 contraction 48.0 -> 49.0 us at config 3, 70.2 -> 72.4 at config 5 -- eight spilled registers -- but 46.9 -> 46.0 and
    // 71.9 -> 70.3 with the wave-local contraction: there it stays)
    constexpr bool WLC = mode_wlc(MODE);                        // wave-local contraction: see phase 2
    constexpr bool USE_DIT = ((DMEL_DIT_MASK >> ilog2(N)) & 1) != 0 && (N != 2048 || WLC);
    constexpr bool TRAINLIKE = (MODE == kTrain || MODE == kTrainH || WLC);
    constexpr int NHS = hsplit_plane_stride(N);                 // bf16 entries per plane (bins 0 .. N/2 + padding to 16 bytes)
    static_assert(!HSPLIT || (N >= kHsplitMinNfft && N <= kHsplitMaxNfft), "kTrainH: frames inside one wave, N/2 a multiple of 32");
    constexpr int FPT = PAIR ? 2 * SLOTS : SLOTS;               // frames per tile
    constexpr int F = N / 2 + 1;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* lds = reinterpret_cast<v2f*>(smem_raw);
    float2* wtab = reinterpret_cast<float2*>(smem_raw + g.AUX_OFF);   // window table (phase 1 only)
    float* red = reinterpret_cast<float*>(smem_raw + g.RED_OFF);
    constexpr bool WIN_LDS = g.WIN_LDS != 0;
    // radix-C twiddles through LDS: the table has R*C entries but a wave-wide global load of it still moves 512 B
    constexpr bool TW2_LDS = (C > 1) && (N <= 2048);
    float2* tw2l = reinterpret_cast<float2*>(smem_raw + g.RED_OFF + kRedBytes);
    // the half-tile exchange of phase 2 (8 waves) lives where the window table does: a second tile needs the table back
    constexpr bool WIN_ALIASED = WIN_LDS;
    // compact layouts (dmel_kernels.h): pairing pass through ds_bpermute, transposition one plane at a time, half window table
    constexpr bool BPERM = g.PAIRING == kPairBperm, PLANE = g.PAIRING == kPairPlane, SPLIT = g.SPLIT != 0, WIN_SYM = g.WIN_SYM != 0;
    constexpr bool KEEPZ = BPERM || PLANE;                      // the spectrum stays in registers until the pairing pass
    constexpr int WPF = g.WPF;                                  // waves per frame (n_fft 8192: 2, 16384: 4)
    static_assert(!BPERM || (G <= 64 && PASSES == 1 && (G == 64 || C == 1 || (WLC && C == 2))), "the bpermute pairing pass: whole frames inside one wave");
    static_assert(!SPLIT || KEEPZ, "a one-plane slot cannot hold the whole spectrum");
    static_assert(!WIN_SYM || G <= 64, "half window table: frames inside one wave");
    static_assert(WPF == 1 || (PLANE && SPLIT && PASSES == 1 && !WIN_LDS), "frames spread over several waves exchange through planes");
    static_assert(!PLANE || G >= 64, "plane pairing: whole waves per frame");
    static_assert(!WLC || (BPERM && WPF == 1 && PASSES == 1 && TPW == 1 && wlc_size(N) && FPW <= 2), "kTrainW: whole frames inside one wave, at most four (frame, P | D) rows per wave");
    constexpr int WPT = (N / 2 + THREADS - 1) / THREADS;       // window entries a thread computes (and keeps when TPW > 1)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an L2).  Re-labelling them so
    // that each XCD gets a CONTIGUOUS range of tiles puts the tiles of one clip on one L2: their
    // overlapping sample reads hit, and the 32-byte output pieces that together make up whole lines
    // of out/tangent merge there before going to HBM.  Speed only; any placement is correct.
    int wg = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = wg & 7;
        // (the two workgroups that share a CU -- local indices i and i + 32 of an XCD's 64 at config 2 -- given the two halves of ONE
        // clip, so that the second pass over the clip hits L1: 19.99 against 20.05 us, not kept)
        wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (wg >> 3);
    }
    const int b = wg / p.wgs_per_clip;
    const int tile0 = (wg % p.wgs_per_clip) * TPW;              // first tile of this workgroup inside its clip
    STAMP(0);
    STAMP_PLACE();
#ifdef DMEL_ABLATE
    // timing ablations (tools/ablate.py builds its own library with -DDMEL_ABLATE; never in libdmel_hip.so)
    const bool dbg_skip_fft = (p.flags & 0x200u) != 0;
    const bool dbg_skip_gemm = (p.flags & 0x100u) != 0;
#else
    constexpr bool dbg_skip_fft = false;
#endif

#ifdef DMEL_ABLATE
    // timing experiment: the workgroups of the first round whose CU-local slot is odd start late by (flags >> 24) x 1024 cycles, so that
    // the two workgroups a CU holds stay out of step for the rest of a many-round launch
    if ((p.flags & 0x400000u) && blockIdx.x < 512u) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if ((hw >> 16) & 1u) for (unsigned i = 0; i < (p.flags >> 24) * 16u; ++i) __builtin_amdgcn_s_sleep(1);
    }
#endif
    // the batch: by address, or (DMEL_FLAG_X_INDIRECT) through a pointer cell read with a scalar load -- a captured step is handed a
    // new batch by rewriting 8 bytes
    const float* xbase = p.x;
    if (p.x_ind) { typedef const float* cfp; xbase = *(const __attribute__((address_space(4))) cfp*)p.x_ind; }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xbase + (size_t)b * p.L, (unsigned)p.L * 4u);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.ent_b, (unsigned)p.ent_b_floats * 4u);

    // ---- what phase 2 needs from global memory: (ks0, nks, boff, tile) of this wave's mel runs in group 0 and their
    // first NBPRE B fragments (all of them for the HTK bank at the usual sizes).  Requested after the FFT of a tile, when
    // its registers are free: the pairing pass and the barrier behind it cover the round trip, nothing in phase 2 waits.
    constexpr int NBPRE = g.NBPRE;
    int4 tr0[NLOC];
    float bpre[NLOC][NBPRE];
    auto fetch_bpre = [&]() {
        if constexpr (!IS_SPEC && !WLC) {
            static_for<0, NLOC>([&](auto l) {
                constexpr int loc = decltype(l)::value;
                tr0[loc] = p.tile_ranges[wave * NLOC + loc];
            });
            // one 64-bit base per lane, compile-time offsets from it
            const float* pre_lane = p.ent_pre + ((unsigned)(wave * (NLOC * NBPRE * 64)) + (unsigned)lane);
            static_for<0, NLOC>([&](auto l) {
                constexpr int loc = decltype(l)::value;
                // fixed layout (wave, run, k-step, lane): the address does not wait for tile_ranges; the number of real groups of
                // 4 k-steps comes with the kernel arguments (scalar load), so the padding of short runs is not fetched
                const int ng = p.pre_groups[wave * NLOC + loc];
                static_for<0, NBPRE / 4>([&](auto qq) {
                    constexpr int q4 = decltype(qq)::value;
                    if (q4 < ng) {
                        static_for<0, 4>([&](auto u) {
                            constexpr int u4 = q4 * 4 + decltype(u)::value;
                            bpre[loc][u4] = pre_lane[(loc * NBPRE + u4) * 64];
                        });
                    } else {
                        // defined on every path: the registers are then dead between two tiles instead of carrying the
                        // previous tile's values through the next FFT
                        static_for<0, 4>([&](auto u) { bpre[loc][q4 * 4 + decltype(u)::value] = 0.f; });
                    }
                });
            });
        }
    };

    // kTrainW: what its phase 2 needs from global memory -- the lane table of the first two phases and the first groups of B operands
    // of phase 0 -- is requested where fetch_bpre() is (behind the second radix stage: the pairing pass covers the round trips); the
    // ring is refilled for the next phase in front of each phase's epilogue.  (Asked for where they are used, each phase began
    // with two dependent round trips: 5 400 cycles per wave for 88 MFMAs and two epilogues, tools/stamps.py.)
#ifndef DMEL_WL_DEPTH
#define DMEL_WL_DEPTH 6
#endif
#ifndef DMEL_WL_EARLY
#define DMEL_WL_EARLY 0
#endif
#ifndef DMEL_WL_NEXT
#define DMEL_WL_NEXT 1
#endif
    constexpr int WL_DEPTH = DMEL_WL_DEPTH;                     // groups of four steps (16 bytes per lane) in the ring
    constexpr int WL_EARLY = DMEL_WL_EARLY;                     // 1: lane tables requested before the pairing pass, 2: the ring of phase 0 too
    floatx4 wl_ring[WLC ? WL_DEPTH : 1];
    int2 wl_li[2];
    const __amdgpu_buffer_rsrc_t rbw = make_rsrc(WLC ? (const void*)p.wl_b4 : (const void*)p.x, WLC ? (unsigned)p.wl_total4 * 1024u : 0u);
    // BUFFER loads on purpose: with plain loads InstCombine folds the ring's phi(load, load) into load(phi(address)) at the loop
    // header and every group waits a cache round trip in front of its first use; an intrinsic call is not folded.  Groups past the
    // end of a phase re-read its last group.
#ifndef DMEL_WL_UNCOND
#define DMEL_WL_UNCOND 1
#endif
    // DMEL_WL_UNCOND (round 6): EVERY ring load is issued -- a group past the end of its phase re-reads the phase's last group, which nobody
    // uses.  With conditional loads the compiler cannot know how many loads follow the one a slot waits for (a slot's `s_waitcnt vmcnt(N)`
    // is only safe if N loads were issued behind it), so every turn of the ring opened with `vmcnt(0)`: the refill requested two
    // instructions earlier was waited for in full, ~300 cycles per turn, 15-20 turns per tile at n_fft 2048 (NOTEBOOK R5.20).  Now the ring is
    // a fixed pattern of WL_DEPTH loads per turn and the waits are `vmcnt(WL_DEPTH - 1)`.
    auto wl_bload = [&](floatx4& dst, int grp, int n4, int boff4) {
#ifdef DMEL_ABLATE
        if (p.flags & 0x10000u) return;                          // timing only: no B operand loads
#endif
        if constexpr (DMEL_WL_UNCOND) {
            // the table index stays inside [0, wl_total4 - 1] whatever the phase holds: a phase of quads whose bands are all empty (many mel bands on few
            // bins) has n4 = 0, and the buffer's range check does not cover the scalar offset
            // (`n4` carries the phase's LAST table index here -- wl_last() -- so that a load costs one scalar add and one scalar min)
            const int gi = boff4 + grp < n4 ? boff4 + grp : n4;
            dst = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rbw, lane * 16, __builtin_amdgcn_readfirstlane(gi) * 1024, 0));
        } else
        if (grp < n4) dst = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rbw, lane * 16, (boff4 + __builtin_amdgcn_readfirstlane(grp)) * 1024, 0));
    };
    // last table index a phase of n4 groups at offset boff4 may read: inside [0, wl_total4 - 1] even for an empty phase
    auto wl_last = [&](int n4, int boff4) -> int {
        int l = boff4 + n4; l = l < p.wl_total4 ? l : p.wl_total4; l -= 1;
        return l > 0 ? l : 0;
    };
    auto wl_ring_init = [&](int n4, int boff4) {
        if constexpr (DMEL_WL_UNCOND) n4 = wl_last(n4, boff4);
        // (a slot whose group does not exist in this phase is never used -- the tail groups are guarded -- so it is "defined" by an empty asm
        // statement instead of four zeros: 24 moves less per ring start, 48-72 vector instructions per wave)
        if constexpr (WLC) static_for<0, WL_DEPTH>([&](auto dd) { constexpr int d = decltype(dd)::value; asm volatile("" : "=v"(wl_ring[d])); wl_bload(wl_ring[d], d, n4, boff4); });
    };
    auto wl_prefetch = [&]() {
        if constexpr (WLC && WL_EARLY >= 1) {
            wl_li[0] = p.wl_lane[lane];
            wl_li[1] = p.wl_lane[(p.wl_phases > 1 ? 64 : 0) + lane];
            if constexpr (WL_EARLY >= 2) wl_ring_init(p.wl_len4[0], 0);
        }
    };

    // lane of this thread inside its frame: part of a wave (G < 64), the wave, or one of WPF waves
    const int j = (WPF > 1) ? 0 : lane / G;
    const int lg = (WPF > 1) ? (wave % WPF) * 64 + lane : lane % G;
    const int qp = lg / C, r = lg % C;
    // exchange among the lanes of one frame: a wave-level fence, or a workgroup barrier when the frame has several waves
    // (every wave of the workgroup runs the same sequence of them)
    // Frames of several waves (n_fft 8192: two, 16384: four): their waves meet at a counter in LDS -- one word per frame, words
    // 36 .. 39 of the sums' region, zeroed in the prologue -- instead of at a workgroup barrier: each wave adds one and waits until
    // the word has reached (number of meetings so far) x WPF.  A frame's waves are resident together (one workgroup), so the wait ends;
    // the frames of a workgroup no longer wait for one another between the transposition and pairing steps (nine meetings per tile).
    unsigned* fctr = reinterpret_cast<unsigned*>(red) + 36;
    unsigned fmeet = 0;
    static_assert(WPF == 1 || (WAVES / WPF <= 4 && 36 + 4 <= kRedBytes / 4), "one counter per frame of the workgroup");
    auto sync_frame = [&]() {
        if constexpr (WPF > 1) {
            fmeet += WPF;
            unsigned* c = fctr + wave / WPF;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // this wave's LDS writes are issued (LDS executes a wave's operations in order)
            if (lane == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (;;) {
                const unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - fmeet) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };

    // ---- samples: one register set per tile of the workgroup; tile ti + 1 is requested while tile ti is transformed.
    // Frames that lie wholly inside the clip (all but the first/last few) use plain offsets; the others clamp every index
    // into the clip and are zeroed by a select at windowing time.  (The hardware range check of buffer loads is not relied
    // upon: it covers voffset + immediate but not soffset, and hipcc chooses that split.)
    float xa[TPW][PASSES][R];
    float xb2[TPW][PAIR ? PASSES : 1][PAIR ? R : 1];
    bool inside[TPW][PASSES];
    auto load_tile = [&](auto tt) {
        constexpr int ti = decltype(tt)::value;
        const int t0 = (tile0 + ti) * FPT;
        static_for<0, PASSES>([&](auto pp) {
            constexpr int pass = decltype(pp)::value;
            const int slot = (WPF > 1) ? wave / WPF : pass * (WAVES * FPW) + wave * FPW + j;
            const int tA = PAIR ? (t0 + 2 * slot) : (t0 + slot);
            const int f0 = tA * p.hop - N / 2;                       // first sample of frame tA
            const int f1 = PAIR ? f0 + p.hop : f0;
            inside[ti][pass] = __all((f0 >= 0) && (f1 + N <= p.L));
#ifdef DMEL_ABLATE
            if (p.flags & 0x2000u) inside[ti][pass] = true;      // timing only: no edge path (edge frames come out wrong)
#endif
            const int sA = f0 + lg;
            bool plain_loads = inside[ti][pass];
#ifdef DMEL_ABLATE
            if (p.flags & 0x4000u) plain_loads = true;           // timing only: unclamped loads, selects kept
#endif
            if (plain_loads) {
                static_for<0, R>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    xa[ti][pass][a] = buf_f32(rx, (sA + G * a) * 4);
                    if constexpr (PAIR) xb2[ti][pass][a] = buf_f32(rx, (sA + p.hop + G * a) * 4);
                });
            } else {
                static_for<0, R>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    xa[ti][pass][a] = buf_f32(rx, clampi(sA + G * a, p.L - 1) * 4);
                    if constexpr (PAIR) xb2[ti][pass][a] = buf_f32(rx, clampi(sA + p.hop + G * a, p.L - 1) * 4);
                });
            }
        });
    };

    // ================= prologue, once per workgroup ============================================
    constexpr bool TW1_LDS = g.TW1_OFF != 0;                    // first-stage twiddles from a table in LDS (kTrainW where it fits)
    constexpr bool TW1_POW = (R >= 16) && !TW1_LDS;             // first-stage twiddles by powers: see phase 1
    float mean = 0.f;
    float2 wkeep[WPT];                                          // this thread's window entries (TPW > 1: written back per tile)
    float2 wmid = make_float2(1.f, 0.f);                        // ... and the centre entry
    const float lam_raw = lam_load(p.lam);                       // a scalar load: see lam_load
    if (!dbg_skip_fft) {
        load_tile(IC<0>{});
        // The waves that hold a clip's first / last frames take the slow path (clamped loads, selects at windowing time) and every
        // other wave of their workgroup waits for them at the barrier before phase 2.  In a launch of ONE round of resident
        // workgroups (kFwdEdgeFirst, set by the host) they get instruction-issue priority over the waves they share a SIMD with:
        // 20.1 -> 19.7 us at BASELINE config 2 (training), 14.0 -> 13.7 (inference).  In launches of several rounds the same
        // priority measured 49.0 -> 52.0 us (config 3) and 73.9 -> 76.7 (config 5): there the waves of different rounds overlap
        // and a privileged wave only delays them.  (Measured and not kept, config 2: priority by wave index, one workgroup of
        // each CU first through the transforms, the contraction phase first: 20.1 against 20.1.)
        if ((p.flags & kFwdEdgeFirst) && !inside[0][0]) __builtin_amdgcn_s_setprio(2);
        // long clips: the partial sums of dmel_prep_kernel are requested here, next to the samples -- asked for where they are used
        // (behind the window table's barrier) every workgroup waited a global round trip for 64 floats
        float ps_early = 0.f;
        if (p.remove_dc && p.psum != nullptr) ps_early = (lane < p.nchunks) ? p.psum[(size_t)b * p.nchunks + lane] : 0.f;
        STAMP(1);   // loads issued
        // lambd (device scalar or by value) and the check that this launch is the n_fft the device value asks for
        const LamState ls = lam_prologue(p.lam, N, blockIdx.x == 0 && tid == 0, lam_raw);
        if (ls.action != kLamRun) {
            if (ls.action == kLamPoison) {
                // no launch of this forward matched the device lambd: NaN instead of stale memory (the host raises too)
                const int rows = IS_SPEC ? F : p.M;
                const float qn = __builtin_nanf("");
                for (int idx = tid; idx < rows * FPT * TPW; idx += THREADS) {
                    const int rr = idx / (FPT * TPW), t = tile0 * FPT + idx % (FPT * TPW);
                    if (t >= p.T) continue;
                    const size_t o = ((size_t)b * rows + rr) * p.T + t;
                    if (p.flags & 4u) reinterpret_cast<unsigned short*>(p.out)[o] = 0x7fc0u; else p.out[o] = qn;
                    if (p.tangent) p.tangent[o] = qn;
                }
                // ... and the saved spectrogram (B, F, T): the filterbank gradient contracts it before the next forward raises (ADVICE r04:
                // with a linear output and a finite upstream gradient, uninitialised memory reached the optimizer)
                if constexpr (TRAINLIKE) {
                    if (p.spec_out) {
                        for (int idx = tid; idx < F * FPT * TPW; idx += THREADS) {
                            const int kk = idx / (FPT * TPW), t = tile0 * FPT + idx % (FPT * TPW);
                            if (t < p.T) p.spec_out[((size_t)b * F + kk) * p.T + t] = qn;
                        }
                    }
                }
            }
            return;
        }
        if constexpr (WPF > 1) {
            if (tid < 4) fctr[tid] = 0u;                                  // the frames' meeting counters (sync_frame)
            __syncthreads();
        }
        // d out / d lambd = htan * (contraction of the scaled tangent spectrum): an fp64 division, done by ONE wave of the
        // workgroup and handed to the epilogue through LDS (every barrier below lies between this store and that load)
        if constexpr (TRAINLIKE || MODE == kSpecTrain) {
            if (wave == 0) { const float h = 0.5f * lam_tangent_scale(ls); if (lane == 0) red[kRedTan] = h; }
        }
        if constexpr (TW2_LDS) { if (tid < R * C) tw2l[tid] = p.tw2[tid]; }          // visible after the barrier below
        if constexpr (TW1_LDS) {
            // rows q = 1 .. R - 1 of the first-stage twiddle table, 16 bytes per thread (visible after the window table's barrier)
            constexpr int n16 = (R - 1) * G / 2;
            for (int i = tid; i < n16; i += THREADS)
                reinterpret_cast<float4*>(smem_raw + g.TW1_OFF)[i] = reinterpret_cast<const float4*>(p.tw1 + G)[i];
        }
        // Phase 2 pads every run of k-steps to a multiple of four with zero filterbank blocks and still reads the A operands of the
        // padding: bins past n_fft/2, i.e. floats 2 F .. of the slot.  In the compact layout those are the end of the transposition
        // plane -- finite data of this tile, EXCEPT the padding column of a row (row stride EXS = G + 1: one float nobody writes)
        // -- and the padding of the slot stride.  At n_fft 1024 float 1055 (bin 527, tangent row: read by the HTK bank's last mel
        // tile) and bins 528 .. 531 (a dense bank) are such holes: whatever an earlier kernel left in LDS -- a NaN pattern -- times
        // a zero coefficient poisoned the accumulator (found in round 4: a trainable-filterbank run went NaN after a few hundred
        // steps, and at once on a box whose previous process had left NaNs behind).  Everything from float 2 F to the end of the
        // slot is zeroed once per workgroup, before the barrier in front of the first transform: what the transposition writes
        // there later is finite, what it leaves out stays zero.
        if constexpr (SPLIT && !IS_SPEC) {
            constexpr int first = 2 * F, count = SS * 2 - first;                                // floats
            static_assert(count >= 0, "slot stride covers PD[0 .. N/2]");
            if constexpr (count > 0) {
                constexpr int CP = 1 << (ilog2(count - 1 > 0 ? count - 1 : 1) + 1);               // next power of two: shifts and masks, no division by `count`
                static_assert(count <= CP, "slot padding fits the zeroing loop");
#ifdef DMEL_ABLATE
                if (!(p.flags & 0x100000u))                                                      // timing only: without the zeroing
#endif
                for (int i = tid; i < SLOTS * CP; i += THREADS) {
                    const int slot = i / CP, jz = i % CP;
                    if (jz < count) reinterpret_cast<float*>(smem_raw + slot * (SS * 8))[first + jz] = 0.f;
                }
                // any thread zeroes any slot: the zeroes must land before the slot's own waves write their transposition there.  With
                // the window table in LDS the barrier behind it does that; the sizes without one (n_fft >= 8192) had NO barrier
                // before the first transform -- frames next to a clip's edge came out wrong now and then until this one went in
                // (caught by tests/test_hip_soak.py::test_no_path_depends_on_what_earlier_kernels_left_in_lds)
                if constexpr (!WIN_LDS) __syncthreads();
            }
        }
        // ---- window table into LDS (time_frequency.py:21-30): every workgroup evaluates the same fp32
        // expression, so the table is identical everywhere; this replaces a separate kernel launch
        if constexpr (WIN_LDS) {
            const float denom = ls.denom;
            float s_ww = 0.f, s_wd = 0.f;
            // w[N/2 + d] = w[N/2 - d]: entry n < N/2 is computed once and stored at n and N - n (n = 0 has no mirror);
            // the centre is exp(-0) = 1 with a zero tangent
            static_for<0, WPT>([&](auto ww_) {
                constexpr int wi = decltype(ww_)::value;
                const int n = tid + THREADS * wi;
                wkeep[wi] = make_float2(0.f, 0.f);
                if (n < N / 2) {
                    const float d = (float)n - (float)N / 2.0f;
                    const float t = d / denom;
                    float w = expf(-0.5f * (t * t));
                    if (p.win_half && n < N / 4) w = 0.f;                         // torch.stft pads a win_length = N/2 window
                    // tangent window up to a constant: w d^2 2^(-2e) (d^2 and the scaling are exact in fp32: one rounding); the
                    // factor sign 2^(2e) / (|lambd| + 1e-15)^3 is applied once per output in the epilogue (lam_tangent_scale)
                    const float dw = w * (d * d) * ls.s2;
                    wkeep[wi] = make_float2(w, dw);
                    wtab[n] = wkeep[wi];
                    // the half-length window of torch.stft sits at [N/4, 3N/4): its mirror image stops one entry short
                    const bool has_mirror = n > 0 && !(p.win_half && n <= N / 4);
                    if constexpr (!WIN_SYM) { if (n > 0) wtab[N - n] = has_mirror ? wkeep[wi] : make_float2(0.f, 0.f); }
                    const float mult = has_mirror ? 2.f : 1.f;
                    s_ww += mult * (w * w); s_wd += mult * (w * dw);
                }
            });
            if (tid == 0) { wtab[N / 2] = make_float2(1.f, 0.f); s_ww += 1.f; }
            if (p.normalize) {
                // time_frequency.py:25: w / sqrt(sum w^2) and the derivative of the quotient (fixed-order sums)
                s_ww = wave_sum(s_ww);
                s_wd = wave_sum(s_wd);
                __syncthreads();
                if (lane == 0) { red[wave] = s_ww; red[16 + wave] = s_wd; }
                __syncthreads();
                float ww = 0.f, wd = 0.f;
                for (int q = 0; q < WAVES; ++q) { ww += red[q]; wd += red[16 + q]; }
                __syncthreads();
                const float inv = 1.0f / sqrtf(ww);
                for (int n = tid; n < (WIN_SYM ? N / 2 + 1 : N); n += THREADS) {
                    const float2 e = wtab[n];
                    wtab[n] = make_float2(e.x * inv, e.y * inv - e.x * wd * inv * inv * inv);
                }
                if constexpr (TPW > 1) {
                    __syncthreads();
                    static_for<0, WPT>([&](auto ww_) {
                        constexpr int wi = decltype(ww_)::value;
                        const int n = tid + THREADS * wi;
                        if (n < N / 2) wkeep[wi] = wtab[n];
                    });
                    wmid = wtab[N / 2];
                }
            }
        }
        // ---- clip mean (models.py:38) --------------------------------------------------------------
        // the waves' sums meet in LDS (the barrier is also the window table's) and are added pairwise, in a fixed order; the quotient
        // by L is rounded once (dmel_kernels.h: "the clip mean")
        auto mean_of = [&](float wsum) -> float {
            if (lane == 0) red[wave] = wsum;
            __syncthreads();
            float t[WAVES];
            static_for<0, WAVES>([&](auto qq) { t[decltype(qq)::value] = red[decltype(qq)::value]; });
            static_for<0, ilog2(WAVES)>([&](auto ll) {
                constexpr int st = 1 << decltype(ll)::value;
                static_for<0, WAVES / (2 * st)>([&](auto ii) { constexpr int i = decltype(ii)::value * 2 * st; t[i] += t[i + st]; });
            });
            return mean_quotient(t[0], p.L, p.inv_L);
        };
#ifdef DMEL_ABLATE
        const bool dbg_no_mean = (p.flags & 0x1000u) != 0;      // timing ablation: no clip sum (mean = 0)
#else
        constexpr bool dbg_no_mean = false;
#endif
        if (dbg_no_mean) {
            if constexpr (WIN_LDS) __syncthreads();
        } else if (WPF == 1 && p.remove_dc && p.psum == nullptr && p.tiles_per_clip == 1 && p.hop <= N / 2) {
            // The whole clip is this workgroup's: its frames already hold every sample -- frame t owns the hop segment
            // [t hop, (t + 1) hop) of the clip, which lies in its second half (hop <= N/2) -- so the clip is added up from the
            // registers the transform is about to use and never read a second time.  Fixed order: registers ascending inside a lane,
            // the 64 lanes by wave_sum, the waves ascending.
            float ps = 0.f;
            static_for<0, PASSES>([&](auto pp) {
                constexpr int pass = decltype(pp)::value;
                const int slot = pass * (WAVES * FPW) + wave * FPW + j;
                const int tA = PAIR ? 2 * slot : slot;                       // tile 0 of a one-tile clip: t0 = 0
                static_for<R / 2, R>([&](auto aa) {
                    constexpr int a = decltype(aa)::value;
                    const int n = lg + G * a;                                // >= N/2: inside the frame's second half
                    const bool seg = n < N / 2 + p.hop;
                    const int ia = tA * p.hop - N / 2 + n;
                    ps += (seg && tA < p.T && ia < p.L) ? xa[0][pass][a] : 0.f;
                    if constexpr (PAIR) ps += (seg && tA + 1 < p.T && ia + p.hop < p.L) ? xb2[0][pass][a] : 0.f;
                });
            });
            mean = mean_of(wave_sum(ps));
        } else if (p.remove_dc && p.psum == nullptr) {
            // short clips: every workgroup adds up its clip itself (L2 hits after the first toucher), in a
            // fixed order, instead of a separate pass over x
            // Loads go out in batches of 8 per thread before anything is added: one memory round trip per batch
            // instead of one per load (a plain loop waits for every load before issuing the next).
            const float* xc = xbase + (size_t)b * p.L;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int i = 0;
            constexpr int KB = 8;
            if ((reinterpret_cast<uintptr_t>(xc) & 15) == 0) {
                const float4* x4 = reinterpret_cast<const float4*>(xc);
#ifdef DMEL_ABLATE
                const int n4 = (p.flags & 0x40000u) ? p.L / 8 : ((p.flags & 0x80000u) ? p.L / 16 : p.L / 4);   // timing only: part of the clip
#else
                const int n4 = p.L / 4;
#endif
                for (int base = 0; base < n4; base += THREADS * KB) {
                    float4 v[KB];
                    static_for<0, KB>([&](auto jj) {
                        constexpr int jv = decltype(jj)::value;
                        const int q = base + tid + THREADS * jv;
                        v[jv] = x4[q < n4 ? q : n4 - 1];
                    });
                    static_for<0, KB>([&](auto jj) {
                        constexpr int jv = decltype(jj)::value;
                        // (one select per 16-byte load instead of four: a + 1 v is a + v to the bit, a + 0 v is a -- the clamped load read a
                        // sample of this very clip, so a non-finite one is in the true sum too)
                        const float okf = (base + tid + THREADS * jv < n4) ? 1.f : 0.f;
                        a0 = fmaf(v[jv].x, okf, a0); a1 = fmaf(v[jv].y, okf, a1); a2 = fmaf(v[jv].z, okf, a2); a3 = fmaf(v[jv].w, okf, a3);
                    });
                }
                i = n4 * 4;
            }
            // what is left (the last L % 4 samples, or everything for a clip that is not 16-byte aligned): dword buffer
            // loads, out-of-range offsets return 0
            for (int base = i; base < p.L; base += THREADS * KB) {
                float v[KB];
                static_for<0, KB>([&](auto jj) { v[decltype(jj)::value] = buf_f32(rx, (base + tid + THREADS * decltype(jj)::value) * 4); });
                static_for<0, KB>([&](auto jj) { a0 += v[decltype(jj)::value]; });
            }
            mean = mean_of(wave_sum((a0 + a1) + (a2 + a3)));
        } else {
            if constexpr (WIN_LDS) __syncthreads();              // window table complete
            if (p.remove_dc) {
                // long clips: the <= 64 partial sums of the prep kernel, one per lane, one round trip, added
                // in a fixed butterfly order (deterministic)
                mean = mean_quotient(wave_sum(ps_early), p.L, p.inv_L);
            }
        }
        STAMP(2);   // window table + clip mean done
    }

    // kTrainH: PD[k] of a slot goes out as four bf16 values -- P hi, P lo, D hi, D lo, planes of NHS entries each: hi = bf16(v),
    // lo = bf16(v - hi), v = hi + lo to 2^-17 -- so that phase 2 reads its A operands (8 consecutive bins of one plane) with one
    // ds_read_b128 per lane and the three products hi hi + lo hi + hi lo stand for one fp32 product.
    auto store_h = [&](int slot_bytes, int k, v2f pdv) {
        const unsigned short ph = bf16_bits(pdv.x), dh = bf16_bits(pdv.y);
        const unsigned short pl = bf16_bits(pdv.x - __uint_as_float((unsigned)ph << 16)), dl = bf16_bits(pdv.y - __uint_as_float((unsigned)dh << 16));
        unsigned short* q = reinterpret_cast<unsigned short*>(smem_raw + slot_bytes) + k;
        q[0] = ph; q[NHS] = pl; q[2 * NHS] = dh; q[3 * NHS] = dl;
    };
    auto load_h = [&](int slot_bytes, int plane, int k) -> float {          // hi + lo of one bin of plane pair `plane` (0: P, 1: D)
        const unsigned short* q = reinterpret_cast<const unsigned short*>(smem_raw + slot_bytes) + 2 * plane * NHS + k;
        return __uint_as_float((unsigned)q[0] << 16) + __uint_as_float((unsigned)q[NHS] << 16);
    };

    // ================= the tiles of this workgroup ==============================================
    static_for<0, TPW>([&](auto tt) {
        constexpr int ti = decltype(tt)::value;
        if (ti > 0 && tile0 + ti >= p.tiles_per_clip) return;          // the clip has no such tile (same answer in every wave)
        const int t0 = (tile0 + ti) * FPT;
        if constexpr (ti > 0) STAMP(16 * ti + 2);   // previous tile's epilogue issued, this tile begins
        if constexpr (ti > 0) {
            // the FFT slots (every wave has read the previous tile's spectra before the barriers of its phase 2 or the one
            // here) and the window table / half-tile exchange region are used again
            __syncthreads();
            // this tile's samples: requested here (behind the barrier, where the registers of the previous tile's matrix phase
            // are free; requested earlier they have to live through that phase and spill); the window rewrite covers part
            // of the round trip, and the lines were touched by the neighbouring frames before
            __builtin_amdgcn_sched_barrier(0);
            if (!dbg_skip_fft) load_tile(IC<ti>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WIN_ALIASED) {
                static_for<0, WPT>([&](auto ww_) {
                    constexpr int wi = decltype(ww_)::value;
                    const int n = tid + THREADS * wi;
                    if (n < N / 2) {
                        wtab[n] = wkeep[wi];
                        if constexpr (!WIN_SYM) { if (n > 0) wtab[N - n] = (p.win_half && n <= N / 4) ? make_float2(0.f, 0.f) : wkeep[wi]; }
                    }
                });
                if (tid == 0) wtab[N / 2] = wmid;
                __syncthreads();
            }
        }
        // ================= phase 1: FFT of this wave's frames ====================================
        if (!dbg_skip_fft) {
            static_for<0, PASSES>([&](auto pp) {
                constexpr int pass = decltype(pp)::value;
                const int slot = (WPF > 1) ? wave / WPF : pass * (WAVES * FPW) + wave * FPW + j;
                v2f* sl = lds + slot * SS;
                const int tA = PAIR ? (t0 + 2 * slot) : (t0 + slot);
                const int f0 = tA * p.hop - N / 2;                       // first sample of frame tA
                const int f1 = PAIR ? f0 + p.hop : f0;
                // zero padding of torch.stft(center=True, pad_mode='constant') applies AFTER the DC removal
                bool inside_w = inside[ti][pass];
#ifdef DMEL_ABLATE
                if (p.flags & 0x20000u) inside_w = true;          // timing only: clamped loads kept, no selects
#endif
                v2f z[R];
                // window entries of this lane: one base register + compile-time offsets (ds_read_b64 offset:512a)
                int wbyte = (WIN_LDS ? g.AUX_OFF : 0) + lg * 8;
                int wbyte_m = g.AUX_OFF + (G - lg) * 8;                      // half table: entry N - n of n = lg + G a, a >= R/2
                asm volatile("" : "+v"(wbyte), "+v"(wbyte_m));
                auto wload = [&](auto aa_) -> v2f {
                    constexpr int a = decltype(aa_)::value;
                    float2 wd2;                                              // (w[n], dw[n] / d|lambd| * scale)
                    if constexpr (WIN_SYM && a >= R / 2) {
                        wd2 = *reinterpret_cast<const float2*>(smem_raw + wbyte_m + G * 8 * (R - 1 - a));
                        // the half-length window of torch.stft covers [N/4, 3N/4): entry 3N/4 is zero, its mirror image is not
                        if constexpr (a == 3 * R / 4) { if (p.win_half && lg == 0) wd2 = make_float2(0.f, 0.f); }
                    }
                    else if constexpr (WIN_LDS) wd2 = *reinterpret_cast<const float2*>(smem_raw + wbyte + G * 8 * a);
                    else wd2 = *reinterpret_cast<const float2*>(reinterpret_cast<const unsigned char*>(p.win2) + wbyte + G * 8 * a);
                    return v2f{wd2.x, wd2.y};
                };
#ifdef DMEL_ABLATE
                // timing only (upper bound): the frames lose a mean the wave adds up from its own samples, and what is left of the
                // clip mean is applied in the pairing pass as a frequency-domain correction (a table read + one packed FMA per bin pair)
                float mean_w = mean;
                if (p.flags & 0x200000u) {
                    float sm_loc = 0.f;
                    static_for<0, R>([&](auto aa) { sm_loc += xa[ti][pass][decltype(aa)::value]; });
                    mean_w = wave_sum(sm_loc) * (1.0f / (64.f * R));
                }
                const float dlt_late = mean - mean_w;
#define DMEL_MEAN_W mean_w
#else
#define DMEL_MEAN_W mean
#endif
                // (every mode but the wave-local one, R = 32 or 64: the window multiply in groups of eight entries -- left alone the scheduler requested
                // all R window entries at once, R register pairs on top of the R samples and the growing z, and spilled 6 ... 22 registers at
                // n_fft 1024 ... 4096 (tools/kres.sh); kTrainW fits as it is and is not touched)
#ifndef DMEL_WIN_GROUP
#define DMEL_WIN_GROUP 8
#endif
                constexpr bool WIN_GROUPED = !WLC && R >= 32 && DMEL_WIN_GROUP > 0;
                if (inside_w) {
                    float mean_i = DMEL_MEAN_W;
                    if constexpr (WIN_GROUPED) asm volatile("" : "+v"(mean_i));      // (its own copy: the R subtractions are then not hoisted above the branch as one block)
                    static_for<0, R>([&](auto aa) {
                        constexpr int a = decltype(aa)::value;
                        if constexpr (WIN_GROUPED && a % (DMEL_WIN_GROUP > 0 ? DMEL_WIN_GROUP : 1) == 0 && a > 0) __builtin_amdgcn_sched_barrier(0);
                        const v2f wd = wload(aa);
                        const float va = xa[ti][pass][a] - mean_i;
                        if constexpr (!PAIR) z[a] = splat(va) * wd;
                        else z[a] = v2f{va, xb2[ti][pass][a] - mean_i} * wd.xx;
                    });
                } else {
                    static_for<0, R>([&](auto aa) {
                        constexpr int a = decltype(aa)::value;
                        const int n = lg + G * a;
                        const v2f wd = wload(aa);
                        const int ia = f0 + n;
                        const float va = ((ia >= 0) && (ia < p.L)) ? xa[ti][pass][a] - mean : 0.f;
                        if constexpr (!PAIR) z[a] = splat(va) * wd;
                        else {
                            const int ib = f1 + n;
                            const float vb = ((ib >= 0) && (ib < p.L)) ? xb2[ti][pass][a] - mean : 0.f;
                            z[a] = v2f{va, vb} * wd.xx;
                            // (the frames at a clip's edge, pair modes: four entries at a time.  Interleaved freely, the index arithmetic of all R entries
                            // raised this RARE path's register demand past the budget, and what it spilled -- z[0 .. 4] -- was spilled on the interior path too)
                            if constexpr (WIN_GROUPED && a % 4 == 3) __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                }
                STAMP(16 * ti + 3);   // samples arrived, windowed
                if constexpr (USE_DIT) fft_reg_dit<R>(z); else fft_reg<R>(z);
                STAMP(16 * ti + 4);   // radix-R #1
                // twiddle w_N^(lg*q), transposition through LDS: S[q][lg]
                v2f u[R];
                // The R - 1 twiddles w^q (w = w_N^lg) of a lane are N*8 bytes of table per frame (32 KB at n_fft 4096).  With TW1_POW
                // only rows q = 1 and q = 8a are loaded, w^(8a+b) = w^(8a) w^b with w^2..w^7 by repeated multiplication (<= 7
                // roundings: ~5e-7 relative, the transform's own noise): R/8 loads and about R complex products instead of R - 1 loads.
                // Measured: config 2 21.87 -> 21.52 us, config 3 49.7 -> 47.4, config 5 74.3 -> 71.9, ESC-50 shape 164 -> 157 (n_fft 4096)
                // and 514 -> 485 (8192); errors against the fp64 oracle unchanged (3e-7 of the loudest bin).  The two or four rows are
                // requested HERE: asked for at the top of the kernel (8 more live registers through the prologue) the 32 x 32 plan measured
                // 22.0 us against 20.0 at config 2, copied to LDS once per workgroup 20.2 against 19.7
                v2f wb[TW1_POW ? 8 : 1];
                if constexpr (TW1_POW) {
                    const float2 w1 = p.tw1[G + lg];
                    wb[1] = v2f{w1.x, w1.y};
                    static_for<2, 8>([&](auto bb) { constexpr int b = decltype(bb)::value; wb[b] = cmul(wb[b - 1], w1); });
                }
                int tw1b = g.TW1_OFF + lg * 8;                               // one base register, compile-time offsets (ds_read_b64 offset: 8 G (q - 1))
                asm volatile("" : "+v"(tw1b));
                auto twiddled = [&](auto qq, v2f v) -> v2f {
                    constexpr int q = decltype(qq)::value;
                    if constexpr (q == 0) return v;
                    else if constexpr (TW1_LDS) return cmul(v, *reinterpret_cast<const float2*>(smem_raw + tw1b + (q - 1) * (G * 8)));
                    else if constexpr (!TW1_POW) return cmul(v, p.tw1[q * G + lg]);
                    else {
                        constexpr int a8 = q / 8, b8 = q % 8;
                        if constexpr (a8 == 0) return cmul(v, float2{wb[b8].x, wb[b8].y});
                        else {
                            const float2 anchor = p.tw1[(8 * a8) * G + lg];
                            if constexpr (b8 == 0) return cmul(v, anchor);
                            else { const v2f t = cmul(wb[b8], anchor); return cmul(v, float2{t.x, t.y}); }
                        }
                    }
                };
                if constexpr (!SPLIT) {
                    static_for<0, R>([&](auto qq) {
                        constexpr int q = decltype(qq)::value;
                        sl[q * EXS + lg] = twiddled(qq, z[bitrev(q, LB)]);
                    });
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    static_for<0, R>([&](auto bb) {
                        constexpr int bi = decltype(bb)::value;
                        u[bi] = sl[qp * EXS + r + C * bi];
                    });
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else {
                    // one plane of N floats: real parts through, then imaginary parts (LDS executes a wave's accesses in order, so
                    // the second set of writes cannot overtake the first set of reads)
                    float* slf = reinterpret_cast<float*>(sl);
#ifndef DMEL_TW1_CHUNK
#define DMEL_TW1_CHUNK 4
#endif
                    // first-stage twiddles from the LDS table in chunks, the next chunk requested while one is used (as the radix-C twiddles
                    // below): read one by one, every element waited for its own LDS round trip behind the plane write of the one before
                    // (`ds_read_b64; s_waitcnt lgkmcnt(0)` 31 times per wave in the assembly).  Chunks of 4 or 8: config 4's batch 106.9-107.0 -> 105.9-106.2 us, config 2
                    // within the noise; chunks of 16 spill (+4.5 %)
                    constexpr int TW1C = (TW1_LDS && DMEL_TW1_CHUNK > 0) ? DMEL_TW1_CHUNK : R;
                    float2 tw1r[(TW1_LDS && DMEL_TW1_CHUNK > 0) ? R : 1];
                    auto tw1_fetch = [&](auto cc) {
                        constexpr int c0 = decltype(cc)::value;
                        static_for<(c0 == 0 ? 1 : c0), (c0 + TW1C < R ? c0 + TW1C : R)>([&](auto q1) {
                            constexpr int q = decltype(q1)::value;
                            tw1r[q] = *reinterpret_cast<const float2*>(smem_raw + tw1b + (q - 1) * (G * 8));
                        });
                    };
                    if constexpr (TW1_LDS && DMEL_TW1_CHUNK > 0) {
                        tw1_fetch(IC<0>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    static_for<0, R>([&](auto qq) {
                        constexpr int q = decltype(qq)::value;
                        v2f v;
                        if constexpr (TW1_LDS && DMEL_TW1_CHUNK > 0) {
                            if constexpr (q % TW1C == 0 && q + TW1C < R) {
                                __builtin_amdgcn_sched_barrier(0);
                                tw1_fetch(IC<q + TW1C>{});
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            v = z[bitrev(q, LB)];
                            if constexpr (q != 0) v = cmul(v, tw1r[q]);
                        } else v = twiddled(qq, z[bitrev(q, LB)]);
                        z[bitrev(q, LB)] = v;
                        slf[q * EXS + lg] = v.x;
                    });
                    sync_frame();
                    float ure[R];
                    static_for<0, R>([&](auto bb) {
                        constexpr int bi = decltype(bb)::value;
                        ure[bi] = slf[qp * EXS + r + C * bi];
                    });
                    sync_frame();
                    static_for<0, R>([&](auto qq) {
                        constexpr int q = decltype(qq)::value;
                        slf[q * EXS + lg] = z[bitrev(q, LB)].y;
                    });
                    sync_frame();
                    static_for<0, R>([&](auto bb) {
                        constexpr int bi = decltype(bb)::value;
                        u[bi] = v2f{ure[bi], slf[qp * EXS + r + C * bi]};
                    });
                    sync_frame();
                }
                STAMP(16 * ti + 5);   // twiddle + LDS transposition
                // the radix-C twiddles w_G^(r*p1) are requested before the second radix-R stage, not one by one inside the
                // cross-lane stage (each read there was waited for on the spot)
                // (R = 32: in chunks of 8, the next chunk requested while one is used -- all 31 at once cost 62 registers on top of
                // the 64 of the transform)
                constexpr int TWC = (R > 16) ? DMEL_TWC : R;
                float2 tw2r[R];
                auto tw2_fetch = [&](auto cc) {
                    constexpr int c0 = decltype(cc)::value;
                    static_for<(c0 == 0 ? 1 : c0), (c0 + TWC < R ? c0 + TWC : R)>([&](auto pp1) {
                        constexpr int p1 = decltype(pp1)::value;
                        if constexpr (TW2_LDS) tw2r[p1] = tw2l[p1 * C + r]; else tw2r[p1] = p.tw2[p1 * C + r];
                    });
                };
                if constexpr (C > 1) {
                    tw2_fetch(IC<0>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (USE_DIT) fft_reg_dit<R>(u); else fft_reg<R>(u);
                STAMP(16 * ti + 6);   // radix-R #2
                // twiddle w_G^(r*p1), radix-C across adjacent lanes, spectrum to LDS in natural order
                const v2f rot_f = splat((C == 4 && r == 3) ? 0.f : 1.f);
                const v2f rot_e = (C == 4 && r == 3) ? v2f{1.f, -1.f} : v2f{0.f, 0.f};
                v2f zr[KEEPZ ? R : 1];                                   // Z[qp + R p1 + R R p2] of this lane, by p1
                static_for<0, R>([&](auto pp1) {
                    constexpr int p1 = decltype(pp1)::value;
                    if constexpr (C > 1 && TWC < R && p1 % TWC == 0 && p1 + TWC < R) {
                        __builtin_amdgcn_sched_barrier(0);
                        tw2_fetch(IC<p1 + TWC>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    v2f v = u[bitrev(p1, LB)];
                    int p2 = 0;
                    if constexpr (C > 1) {
                        if constexpr (p1 != 0) v = cmul(v, tw2r[p1]);
                    }
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
                    if constexpr (KEEPZ) zr[p1] = v;
                    else {
                        const int k = qp + R * p1 + R * R * p2;
                        sl[z_index<R, C>(k)] = v;
                    }
                });
                if constexpr (KEEPZ) {
                    // ---- pairing pass without the spectrum in LDS.  Z[N-k] of k = qp + R p1 + R^2 p2 is Z[qp' + R p1' + R^2 p2'] with
                    //   qp > 0:            qp' = R - qp, p1' = R - 1 - p1,  p2' = C - 1 - p2
                    //   qp = 0, p1 > 0:    qp' = 0,      p1' = R - p1,      p2' = C - 1 - p2
                    //   qp = 0, p1 = 0:    qp' = 0,      p1' = 0,           p2' = (C - p2) mod C
                    // i.e. one fixed partner lane (two for the four lanes with qp = 0) and a register index that depends on p1 only:
                    // each lane offers the register its partner wants (a select between the two cases) and ds_bpermute_b32 hands
                    // it over -- LDS crossbar, no LDS storage.  Rounds p1 < R/2 meet every pair {k, N-k} exactly once (the partner
                    // register is >= R/2), one more round covers p1 = R/2 on the qp = 0 lanes.  PD is symmetric in k <-> N-k, so
                    // the lane that holds the upper bin writes PD[N-k]; only PD[0..N/2] exists in LDS, over the transposition plane.
                    constexpr int PADC = (C > 1) ? 4 : 0, RR = R * R;
                    auto lane_of_p2 = [](int v) { return C == 4 ? (((v & 1) << 1) | (v >> 1)) : v; };
                    const int p2 = lane_of_p2(r);                              // (the digit reversal is its own inverse)
                    const bool q0 = (qp == 0);
                    const bool dir_a = (2 * p2 < C);                             // this lane's bins are <= N/2: it writes PD[k], else PD[N-k]
                    const int p2m = C - 1 - p2;
                    const int fl0 = j * G;                                      // first lane of this frame inside the wave (G < 64: FPW frames)
                    const int pull1 = (fl0 + ((R - qp) & (R - 1)) * C + lane_of_p2(p2m)) * 4;
                    const int pull0 = q0 ? (fl0 + lane_of_p2((C - p2) % C)) * 4 : pull1;
                    // PLANE (frames of several waves): the partner lane may sit in another wave, so Z[N-k] goes through one plane
                    // of floats in the frame's slot, indexed k + PLANE_PAD (k / R^2): real parts written, read mirrored, then the
                    // imaginary parts through the same plane.  Mirror index of (qp, p1, p2): the partner's own index, linear in p1.
                    constexpr int PP = g.PLANE_PAD;
                    float* plane = reinterpret_cast<float*>(sl);
                    const int pidx = qp + (RR + PP) * p2;                              // + R p1
                    const int midx = (R - qp) + R * (R - 1) + (RR + PP) * p2m;         // - R p1   (qp = 0: R - qp = R, no wrap)
                    const int midx0 = q0 ? (RR + PP) * ((C - p2) % C) : midx;         // p1 = 0
                    float znx[PLANE ? R / 2 + 1 : 1];
                    if constexpr (PLANE) {
                        static_for<0, R>([&](auto pp1) { constexpr int p1 = decltype(pp1)::value; plane[pidx + R * p1] = zr[p1].x; });
                        sync_frame();
                        static_for<0, R / 2 + 1>([&](auto pp1) {
                            constexpr int p1 = decltype(pp1)::value;
                            znx[p1] = plane[(p1 == 0) ? midx0 : midx - R * p1];
                        });
                        sync_frame();
                        static_for<0, R>([&](auto pp1) { constexpr int p1 = decltype(pp1)::value; plane[pidx + R * p1] = zr[p1].y; });
                        sync_frame();
                    }
                    v2f pdk[PLANE ? R / 2 + 1 : 1];                                    // PLANE: PD overwrites the plane, after everyone has read it
                    const int slot_b = slot * (SS * 8);
                    const int base_a = slot_b + (qp + (RR + PADC) * p2) * 8;
                    const int base_b = slot_b + ((R - qp) + R * (R - 1) + (RR + PADC) * p2m) * 8;
                    const int obase = dir_a ? base_a : base_b;
                    const int ostep = dir_a ? R * 8 : -R * 8;
                    const bool nyq = q0 && (2 * p2 == C);                        // k = N/2 (C > 1: round 0; C = 1: the extra round)
                    // (kTrainW reads PD[k] at 8 k for every k <= N/2: its Nyquist bin sits at the unpadded position)
                    const int addr0 = nyq ? slot_b + (N / 2 + (WLC ? 0 : PADC * (C / 2))) * 8 : obase;
                    const bool w0 = dir_a || !q0 || nyq;
                    static_for<0, R / 2 + 1>([&](auto pp1) {
                        constexpr int p1 = decltype(pp1)::value;
                        constexpr int s_a = (p1 == R / 2) ? R / 2 : (R - 1 - p1);        // what a partner with qp > 0 wants
                        constexpr int s_b = (p1 == R / 2) ? R / 2 : ((R - p1) % R);      // ... with qp = 0
                        // (component by component: a scalar-condition select of two ext vectors lost its second lane here)
                        const float send_x = (s_a == s_b) ? zr[s_a].x : (q0 ? zr[s_b].x : zr[s_a].x);
                        const float send_y = (s_a == s_b) ? zr[s_a].y : (q0 ? zr[s_b].y : zr[s_a].y);
                        v2f zn;
                        if constexpr (PLANE) {
                            zn = v2f{znx[p1], plane[(p1 == 0) ? midx0 : midx - R * p1]};
                        } else {
                            const int pull = (p1 == 0) ? pull0 : pull1;
                            const int got_x = __builtin_amdgcn_ds_bpermute(pull, __builtin_bit_cast(int, send_x));
                            const int got_y = __builtin_amdgcn_ds_bpermute(pull, __builtin_bit_cast(int, send_y));
                            zn = v2f{__builtin_bit_cast(float, got_x), __builtin_bit_cast(float, got_y)};
                        }
                        const v2f zk = zr[p1];
                        float sx = zk.x + zn.x, sy = zk.y - zn.y, dx = zk.x - zn.x, dy = zk.y + zn.y;
#ifdef DMEL_ABLATE
                        if constexpr (WIN_LDS) {
                            if (p.flags & 0x200000u) {          // timing only: stand-in for (2 W[k], 2 W'[k]) of the window's own transform
                                const float2 cw = *reinterpret_cast<const float2*>(smem_raw + wbyte + G * 8 * (p1 % (R / 2)));
                                sx = fmaf(-dlt_late, cw.x, sx); dy = fmaf(-dlt_late, cw.y, dy);
                            }
                        }
#endif
                        v2f pdv;
                        if constexpr (!PAIR) pdv = v2f{fmaf(sx, sx, sy * sy), fmaf(sx, dy, -(sy * dx))};
                        else pdv = v2f{fmaf(sx, sx, sy * sy), fmaf(dx, dx, dy * dy)};
                        if constexpr (PLANE) pdk[p1] = pdv;
                        else if constexpr (HSPLIT) {
                            // the same bins, unpadded, into the bf16 planes of the slot
                            const int kb_a = qp + RR * p2, kb_b = (R - qp) + R * (R - 1) + RR * p2m;
                            const int kb0 = dir_a ? kb_a : kb_b, kbs = dir_a ? R : -R;
                            if constexpr (p1 == 0) { if (w0) store_h(slot_b, nyq ? N / 2 : kb0, pdv); }
                            else if constexpr (p1 < R / 2) store_h(slot_b, kb0 + kbs * p1, pdv);
                            else { if (q0 && dir_a) store_h(slot_b, kb_a + R * (R / 2), pdv); }
                        }
                        else if constexpr (p1 == 0) { if (w0) *reinterpret_cast<v2f*>(smem_raw + addr0) = pdv; }
                        else if constexpr (p1 < R / 2) *reinterpret_cast<v2f*>(smem_raw + obase + ostep * p1) = pdv;
                        else { if (q0 && dir_a) *reinterpret_cast<v2f*>(smem_raw + base_a + R * 8 * (R / 2)) = pdv; }
                    });
                    if constexpr (PLANE) {
                        sync_frame();
                        static_for<0, R / 2 + 1>([&](auto pp1) {
                            constexpr int p1 = decltype(pp1)::value;
                            if constexpr (p1 == 0) { if (w0) *reinterpret_cast<v2f*>(smem_raw + addr0) = pdk[p1]; }
                            else if constexpr (p1 < R / 2) *reinterpret_cast<v2f*>(smem_raw + obase + ostep * p1) = pdk[p1];
                            else { if (q0 && dir_a) *reinterpret_cast<v2f*>(smem_raw + base_a + R * 8 * (R / 2)) = pdk[p1]; }
                        });
                    }
                    // the filterbank fragments of phase 2: requested once the spectrum registers are dead
                    if constexpr (pass == PASSES - 1) { __builtin_amdgcn_sched_barrier(0); fetch_bpre(); wl_prefetch(); }
                } else {
                // the filterbank fragments of phase 2 are requested here: the registers of the FFT are free, and the pairing
                // pass plus the barrier behind it cover the round trip
                if constexpr (pass == PASSES - 1) { fetch_bpre(); __builtin_amdgcn_sched_barrier(0); }
                // ---- pairing pass: the two real spectra packed in Z are separated ONCE per bin here, by the wave
                // that owns the slot, instead of by every A-fragment builder in phase 2:
                //   S = Z[k] + conj Z[N-k], D = Z[k] - conj Z[N-k];  PD[k] = (|S|^2, Im(conj S * D))   (train: 4|X|^2, 2 d|X|^2)
                //                                                    PD[k] = (|S|^2, |D|^2)           (pairs: 4|Xa|^2, 4|Xb|^2)
                // stored in place over Z[0 .. N/2] (all reads of the wave precede its writes: LDS is in order per wave)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                constexpr int NPAIR = N / (2 * G) + 1;           // bins lg + G*i <= N/2
                // Addresses are one of four per-lane byte bases plus a compile-time offset (ds_read_b64 offset:...):
                //   Z[k],   k = lg + G i:            zb + 8 (G i + pad(G i))
                //   Z[N-k], lg >= 1:                 mb + 8 (c_i + pad(c_i)),  mb = slot + 8 (G - lg),  c_i = N - G (i + 1)
                //   lane 0: N - G i itself; it sits one padding step further when it starts an R*R block (mbA), and
                //   wraps to bin 0 for i = 0 (mb0).
                constexpr int PADC = (C > 1) ? 4 : 0, RR = R * R;
                const int slot_b = slot * (SS * 8);
                int zb = slot_b + lg * 8;
                int mb = slot_b + (G - lg) * 8;
                int mbA = mb + ((lg == 0) ? PADC * 8 : 0);
                int mb0 = (lg == 0) ? slot_b : mb + (N - G + PADC * ((N - G) / RR)) * 8;       // full address of Z[N-k] for i = 0
                asm volatile("" : "+v"(zb), "+v"(mb), "+v"(mbA), "+v"(mb0));
                v2f pd[NPAIR];
                static_for<0, NPAIR>([&](auto ii) {
                    constexpr int i = decltype(ii)::value;
                    constexpr int ck = G * i, cm = N - G * (i + 1);
                    constexpr bool crossing = PADC != 0 && ((N - G * i) % RR) == 0;
                    const int mbase = (i == 0) ? mb0 : (crossing ? mbA : mb);
                    const v2f zk = *reinterpret_cast<const v2f*>(smem_raw + zb + (ck + PADC * (ck / RR)) * 8);
                    const v2f zn = *reinterpret_cast<const v2f*>(smem_raw + mbase + ((i == 0) ? 0 : (cm + PADC * (cm / RR)) * 8));
                    const float sx = zk.x + zn.x, sy = zk.y - zn.y, dx = zk.x - zn.x, dy = zk.y + zn.y;
                    if constexpr (!PAIR) pd[i] = v2f{fmaf(sx, sx, sy * sy), fmaf(sx, dy, -(sy * dx))};
                    else pd[i] = v2f{fmaf(sx, sx, sy * sy), fmaf(dx, dx, dy * dy)};
                });
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                static_for<0, NPAIR>([&](auto ii) {
                    constexpr int i = decltype(ii)::value;
                    constexpr int ck = G * i;
                    v2f* dst = reinterpret_cast<v2f*>(smem_raw + zb + (ck + PADC * (ck / RR)) * 8);
                    if constexpr (HSPLIT) { if (i < NPAIR - 1 || lg == 0) store_h(slot_b, lg + ck, pd[i]); }
                    else
                    if (i < NPAIR - 1 || lg == 0) *dst = pd[i];          // the last round holds only the Nyquist bin
                });
                }
            });
        }
        STAMP(16 * ti + 7);   // twiddle + cross-lane radix-C + spectrum to LDS + pairing pass
        if constexpr (WLC) {
            // every wave goes on with the frames it transformed itself: its own LDS writes are all it waits for
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (p.spec_out) __syncthreads();          // (the saved spectrogram is written by all threads from all slots)
        } else __syncthreads();
        STAMP(16 * ti + 8);   // barrier
#ifdef DMEL_ABLATE
        if (dbg_skip_gemm) { if (tid == 0 && lds[0].x == 12345.678f) p.out[0] = 0.f; return; }
#endif
        float htan = 0.f;
        if constexpr (TRAINLIKE || MODE == kSpecTrain) htan = red[kRedTan];       // see the prologue
        if constexpr (TRAINLIKE) {
            // a trainable filterbank's gradient contracts this very spectrogram with the output's gradient (models.py:53): written out
            // here, (B, F, T) as time_frequency.py:53 lays it out, it saves dmel_backward_fb the recompute.  16 consecutive threads
            // write 16 consecutive frames of one bin: 64-byte pieces.
            if (p.spec_out) {
                for (int idx = tid; idx < SLOTS * F; idx += THREADS) {
                    const int k = idx / SLOTS, slot = idx % SLOTS;
                    const int t = t0 + slot;
                    if constexpr (HSPLIT) { if (t < p.T) p.spec_out[((size_t)b * F + k) * p.T + t] = 0.25f * load_h(slot * (SS * 8), 0, k); }
                    else
                    if (t < p.T) p.spec_out[((size_t)b * F + k) * p.T + t] = 0.25f * (lds + slot * SS)[WLC ? k : z_index<R, C>(k)].x;
                }
            }
        }
        if constexpr (IS_SPEC) {
            // power spectrogram (time_frequency.py:53), layout (B, F, T); kSpecTrain also writes d P / d lambd
            for (int idx = tid; idx < SLOTS * F; idx += THREADS) {
                const int k = idx / SLOTS, slot = idx % SLOTS;
                const v2f pdv = (lds + slot * SS)[z_index<R, C>(k)];
                if constexpr (MODE == kSpec) {
                    const int t = t0 + 2 * slot;
                    float* o = p.out + ((size_t)b * F + k) * p.T;
                    if (t < p.T) o[t] = 0.25f * pdv.x;
                    if (t + 1 < p.T) o[t + 1] = 0.25f * pdv.y;
                } else {
                    const int t = t0 + slot;
                    if (t < p.T) {
                        const size_t o = ((size_t)b * F + k) * p.T + t;
                        p.out[o] = 0.25f * pdv.x;
                        if (p.tangent) p.tangent[o] = htan * pdv.y;
                    }
                }
            }
        } else if constexpr (WLC) {
            // ================= phase 2, wave-local: v_mfma_f32_4x4x1_16b_f32 ==========================
            // One instruction multiplies 16 independent (4 x 1) (1 x 4) blocks.  Block b works for one quad of mel bands; its four
            // rows are this wave's (frame, P | D) pairs -- two frames of a 32-lane plan, or one frame (rows 2, 3 repeat 0, 1 and are
            // dropped) -- and a step feeds it ONE bin: lane 4 b + i supplies PD of row i at the block's current bin (one ds_read_b32,
            // consecutive steps are consecutive bins: immediate offsets), lane 4 b + j the filterbank coefficient of column j (four
            // steps per 16-byte load of a table every wave of the grid shares).  The HTK bank has at most two non-zeros per bin: a
            // quad's band is 5 .. 66 bins wide at n_fft 1024 and the 16 blocks of a phase (quads of similar width) walk their bands in
            // lock step -- 88 eight-cycle instructions per wave against 39 of 32 cycles for the banded 16 x 16 x 4 tiles, no workgroup
            // barrier in front (a wave needs nobody else's frames), no exchange of partial sums, the same work in every wave.
            const bool do_log = (p.flags & 1u) != 0;
            const bool out_bf16 = (p.flags & 4u) != 0;
            const int row = lane & 3;
            const int fr = (FPW == 2) ? (row & 1) : 0, typ = (FPW == 2) ? (row >> 1) : (row & 1);
            const int a_lane = (wave * FPW + fr) * (SS * 8) + typ * 4;
            const int tA = t0 + wave * FPW;                                    // first frame of this wave
            int off4 = 0;
            // one 64-bit base per tensor and clip in scalar registers, 32-bit element offsets per lane (M T < 2^31)
            float* const out_clip = p.out + (size_t)b * p.M * p.T;
            unsigned short* const outh_clip = reinterpret_cast<unsigned short*>(p.out) + (size_t)b * p.M * p.T;
            float* const tan_clip = p.tangent ? p.tangent + (size_t)b * p.M * p.T : nullptr;
            // Staged epilogue (two phases at most -- 128 mel bands --, fp32 output, rows of whole 16-byte pieces): a lane's results --
            // 8 bytes per tensor and mel band, 64 different rows per store instruction -- are kept until the wave's last phase, written
            // over the wave's OWN frame slots (its spectra are dead by then), and after ONE workgroup barrier every thread stores
            // 16 bytes of a row: 4 lanes cover the tile's 16 frames of one (tensor, mel band), 64 contiguous bytes.  Measured with the
            // store pattern alone (tools/xtime.py 0x20000): config 2 18.8 -> 17.0 us, config 4's batch 117.7 -> 108.2 -- the scattered
            // 8-byte stores cost the memory pipeline a request per lane.
#ifndef DMEL_WL_STAGE
#define DMEL_WL_STAGE 1
#endif
            constexpr int NST = (FPW == 1) ? 4 : 2;                              // phases whose results a lane can keep (2 FPW values each)
            const bool staged = DMEL_WL_STAGE && p.wl_phases <= NST && p.M <= SS && !out_bf16 && (p.T & 3) == 0 && p.tangent != nullptr && SLOTS % 4 == 0;
            float sv[NST][2 * FPW];
            int sm[NST];
            static_for<0, NST>([&](auto pp) { sm[decltype(pp)::value] = -1; });
            // the lane table (and merge table) of phase ph + 1 is requested at the top of phase ph: asked for where it is used, every phase
            // began with a global round trip (~700 cycles: a phase cost as much as ~36 of its steps, which is what kept config 3's split
            // schedule -- three phases of 100 steps against two of 136 -- from paying)
            int2 li_pref = (WL_EARLY >= 1) ? wl_li[0] : p.wl_lane[lane];
            int mi_pref = (FPW == 1 && p.wl_mg[0] != 0) ? p.wl_merge[lane] : 0;
            for (int ph = 0; ph < p.wl_phases; ++ph) {
                const int n4 = p.wl_len4[ph];
                const int2 li = li_pref;
                const int mi = mi_pref;
                if (ph + 1 < p.wl_phases) {
                    li_pref = p.wl_lane[(ph + 1) * 64 + lane];
                    if constexpr (FPW == 1) mi_pref = (p.wl_mg[ph + 1] != 0) ? p.wl_merge[(ph + 1) * 64 + lane] : 0;
                }
                if ((WL_EARLY < 2 && ph == 0) || (!DMEL_WL_NEXT && ph > 0)) wl_ring_init(n4, off4);      // (off4: still this phase's first group)
                off4 += n4;
                int aaddr = a_lane + li.x;
                floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                // A operands: the next group's four bins are requested before this group's MFMAs are issued (reads past the end of a
                // phase land on finite data of the slot and are not used).  B operands: the ring (see wl_bload).
                float a_cur[4];
                static_for<0, 4>([&](auto uu) { constexpr int u = decltype(uu)::value; a_cur[u] = *reinterpret_cast<const float*>(smem_raw + aaddr + 8 * u); });
                auto group = [&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    float a_nxt[4];
                    static_for<0, 4>([&](auto uu) { constexpr int u = decltype(uu)::value; a_nxt[u] = *reinterpret_cast<const float*>(smem_raw + aaddr + 32 * (d + 1) + 8 * u); });
                    const floatx4 bq = wl_ring[d];
#ifdef DMEL_ABLATE
                    if (p.flags & 0x800u) { acc0[0] += a_cur[0] + a_cur[1] + a_cur[2] + a_cur[3] + bq[0] + bq[1] + bq[2] + bq[3]; static_for<0, 4>([&](auto uu) { constexpr int u = decltype(uu)::value; a_cur[u] = a_nxt[u]; }); return; }   // timing only: no MFMAs
#endif
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur[0], bq[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur[1], bq[1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur[2], bq[2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur[3], bq[3], acc1, 0, 0, 0);
                    static_for<0, 4>([&](auto uu) { constexpr int u = decltype(uu)::value; a_cur[u] = a_nxt[u]; });
                };
                int s4 = 0;
                const int ring_n4 = DMEL_WL_UNCOND ? wl_last(n4, off4 - n4) : n4;
                for (; s4 + WL_DEPTH <= n4; s4 += WL_DEPTH) {
                    static_for<0, WL_DEPTH>([&](auto dd) {
                        constexpr int d = decltype(dd)::value;
                        group(dd);
                        wl_bload(wl_ring[d], s4 + d + WL_DEPTH, ring_n4, off4 - n4);
                        if constexpr (DMEL_WL_UNCOND) {
                            // one group = the next group's A operands (two ds_read2), four MFMAs, the slot's refill -- in this order, group by group
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                    });
                    aaddr += WL_DEPTH * 32;
                }
                static_for<0, WL_DEPTH - 1>([&](auto dd) { if (s4 + decltype(dd)::value < n4) group(dd); });
                // the next phase's first groups: in flight under this phase's epilogue
                if (DMEL_WL_NEXT && ph + 1 < p.wl_phases) wl_ring_init(p.wl_len4[ph + 1], off4);
                floatx4 tot = acc0 + acc1;
                if constexpr (FPW == 1) {
                    // quads split over several blocks of this phase (the host's schedule, dmel_api.cpp): the pieces' partial sums -- rows P
                    // and D of this wave's one frame -- are added across lanes, the piece that carries the mel band receives last
                    const int mg = p.wl_mg[ph];
                    if (mg != 0) {
                        // (the rows are copied out of the vector first: __builtin_bit_cast on the vector's ELEMENTS read element 0 both times --
                        // hipcc 7.2 emitted one ds_bpermute for the two -- and d lambd of g3_c3 came out wrong)
                        float r0 = tot[0], r1 = tot[1];
                        {
                            const int pa = (mi & 63) << 2;
                            const float t0 = __int_as_float(__builtin_amdgcn_ds_bpermute(pa, __float_as_int(r0)));
                            const float t1 = __int_as_float(__builtin_amdgcn_ds_bpermute(pa, __float_as_int(r1)));
                            if (mi & 0x10000) { r0 += t0; r1 += t1; }
                        }
                        if (mg & 2) {
                            const int pb = ((mi >> 8) & 63) << 2;
                            const float t0 = __int_as_float(__builtin_amdgcn_ds_bpermute(pb, __float_as_int(r0)));
                            const float t1 = __int_as_float(__builtin_amdgcn_ds_bpermute(pb, __float_as_int(r1)));
                            if (mi & 0x20000) { r0 += t0; r1 += t1; }
                        }
                        tot[0] = r0; tot[1] = r1;
                    }
                }
                // ---- epilogue: column j of block b = mel band li.y, rows = (frame, P | D) ----------------
                const int m = li.y;
                if (m < 0) continue;
#ifdef DMEL_ABLATE
                if (p.flags & 0x400u) { if (tot[0] == 12345.678f) out_clip[0] = tot[1] + tot[2] + tot[3]; continue; }   // timing only: no epilogue
                if (p.flags & 0x20000u) {
                    // timing only (values are wrong): the store pattern of a staged epilogue -- wave w writes rows 16 w .. 16 w + 15 of one
                    // tensor per phase, 64 contiguous bytes per row (this tile's 16 frames), 16 bytes per lane
                    const int mrow = 16 * wave + (lane >> 2);
                    float* q = (ph == 0 ? out_clip : (tan_clip ? tan_clip : out_clip)) + (unsigned)mrow * (unsigned)p.T + t0 + 4 * (lane & 3);
                    if (ph < 2 && mrow < p.M) *reinterpret_cast<float4*>(q) = make_float4(tot[0], tot[1], tot[2], tot[3]);
                    continue;
                }
#endif
                const unsigned rbase = (unsigned)m * (unsigned)p.T;
                float* orow = out_clip + rbase;
                unsigned short* orow_h = outh_clip + rbase;
                float* trow = tan_clip ? tan_clip + rbase : nullptr;
                float ov[FPW], tv[FPW];
                static_for<0, FPW>([&](auto ff) {
                    constexpr int f = decltype(ff)::value;
                    const float mel = 0.25f * tot[(FPW == 2) ? f : 0];
                    const float dmel = htan * tot[(FPW == 2) ? 2 + f : 1];
                    const float me = mel + p.eps;
                    ov[f] = do_log ? fast_log(me, p.eps) : mel;
                    tv[f] = do_log ? dmel * __builtin_amdgcn_rcpf(me) : dmel;
                });
                if (staged) {
                    static_for<0, NST>([&](auto pp) {
                        if (decltype(pp)::value == ph) {
                            sm[decltype(pp)::value] = m;
                            static_for<0, FPW>([&](auto ff) { constexpr int f = decltype(ff)::value; sv[decltype(pp)::value][f] = ov[f]; sv[decltype(pp)::value][FPW + f] = tv[f]; });
                        }
                    });
                    continue;
                }
                if constexpr (FPW == 2) {
                    if ((p.T & 1) == 0 && tA + 1 < p.T) {
                        // even T (tA is even): both frames of the wave as one aligned 8-byte store per tensor
                        if (out_bf16) *reinterpret_cast<unsigned*>(orow_h + tA) = (unsigned)bf16_bits(ov[0]) | ((unsigned)bf16_bits(ov[1]) << 16);
#ifdef DMEL_WL_NT
                        else __builtin_nontemporal_store(v2f{ov[0], ov[1]}, reinterpret_cast<v2f*>(orow + tA));
                        if (trow) __builtin_nontemporal_store(v2f{tv[0], tv[1]}, reinterpret_cast<v2f*>(trow + tA));
#else
                        else *reinterpret_cast<float2*>(orow + tA) = make_float2(ov[0], ov[1]);
                        if (trow) *reinterpret_cast<float2*>(trow + tA) = make_float2(tv[0], tv[1]);
#endif
                        continue;
                    }
                }
                static_for<0, FPW>([&](auto ff) {
                    constexpr int f = decltype(ff)::value;
                    const int t = tA + f;
                    if (t < p.T) {
                        if (out_bf16) orow_h[t] = bf16_bits(ov[f]); else orow[t] = ov[f];
                        if (trow) trow[t] = tv[f];
                    }
                });
            }
            STAMP(16 * ti + 9);   // contraction and epilogue arithmetic of all phases
            if (staged) {
                float* const stg = reinterpret_cast<float*>(smem_raw + wave * FPW * (SS * 8));          // [tensor][mel band][frame of this wave]
                static_for<0, NST>([&](auto pp) {
                    constexpr int q = decltype(pp)::value;
                    if (sm[q] >= 0) {
                        static_for<0, 2>([&](auto tt2) {
                            constexpr int pl = decltype(tt2)::value;
                            float* d = stg + ((pl * p.M + sm[q]) * FPW);
                            if constexpr (FPW == 2) *reinterpret_cast<float2*>(d) = make_float2(sv[q][pl * FPW], sv[q][pl * FPW + 1]);
                            else d[0] = sv[q][pl * FPW];
                        });
                    }
                });
                __syncthreads();
                STAMP(16 * ti + 10);
                constexpr int QPR = SLOTS / 4;                               // 16-byte pieces per row of the tile
                const int total = 2 * p.M * QPR;
                for (int idx = tid; idx < total; idx += THREADS) {
                    const int rw = idx / QPR, c = idx % QPR;
                    const int pl = rw >= p.M ? 1 : 0, mm = rw - pl * p.M;
                    float4 v;
                    if constexpr (FPW == 2) {
                        const float2 lo = *reinterpret_cast<const float2*>(smem_raw + (2 * c) * FPW * (SS * 8) + ((pl * p.M + mm) * FPW) * 4);
                        const float2 hi = *reinterpret_cast<const float2*>(smem_raw + (2 * c + 1) * FPW * (SS * 8) + ((pl * p.M + mm) * FPW) * 4);
                        v = make_float4(lo.x, lo.y, hi.x, hi.y);
                    } else {
                        float e[4];
                        static_for<0, 4>([&](auto uu) { constexpr int u = decltype(uu)::value; e[u] = *reinterpret_cast<const float*>(smem_raw + (4 * c + u) * (SS * 8) + (pl * p.M + mm) * 4); });
                        v = make_float4(e[0], e[1], e[2], e[3]);
                    }
                    float* dst = (pl ? tan_clip : out_clip) + (unsigned)mm * (unsigned)p.T + t0 + 4 * c;
                    const int t = t0 + 4 * c;
                    if (t + 3 < p.T) *reinterpret_cast<float4*>(dst) = v;
                    else { if (t < p.T) dst[0] = v.x; if (t + 1 < p.T) dst[1] = v.y; if (t + 2 < p.T) dst[2] = v.z; }
                }
            }
            STAMP(16 * ti + 11);
        } else {
            // ================= phase 2: mel contraction on the matrix cores ======================
            const int row16 = lane & 15;
            const int slot8 = 2 * (row16 >> 2) + (row16 & 1);
            const int type = (row16 >> 1) & 1;
            const int kofs = lane >> 4;
            const int cg = lane >> 4;      // accumulator row group of this lane (C/D layout)
            const int col = lane & 15;

            const bool do_log = (p.flags & 1u) != 0;
            const bool out_bf16 = (p.flags & 4u) != 0;
            // ---- epilogue of one 16-mel tile: accumulators -> (B,1,M,T) -------------------------------
            auto write_tile = [&](int nt, const floatx4 (&tt)[MT]) {
                    if (nt < 0) return;
                    const int m = 16 * nt + col;
                    if (m >= p.M) return;
                    const size_t rbase = ((size_t)b * p.M + m) * p.T;
                    float* orow = p.out + rbase;
                    unsigned short* orow_h = reinterpret_cast<unsigned short*>(p.out) + rbase;     // DMEL_FLAG_OUT_BF16: out is bf16
                    float* trow = p.tangent ? p.tangent + rbase : nullptr;
                    auto put = [&](int t, float v) { if (out_bf16) orow_h[t] = bf16_bits(v); else orow[t] = v; };
                    static_for<0, MT>([&](auto mm) {
                        constexpr int mt = decltype(mm)::value;
                        const floatx4 a = tt[mt];
                        if constexpr (TRAINLIKE) {
                            // rows 4cg+i: i=0,1 -> |X|^2 of slots 2cg, 2cg+1; i=2,3 -> d|X|^2 of the same slots
                            const int tp = t0 + mt * 8 + 2 * cg;
                            if (((p.T | t0) & 1) == 0 && mt * 8 + 2 * cg + 1 < SLOTS && tp + 1 < p.T) {
                                // even T: both frames of this lane form one aligned 8-byte store per tensor
                                float2 o2, t2;
                                static_for<0, 2>([&](auto ss) {
                                    constexpr int s = decltype(ss)::value;
                                    const float mel = 0.25f * a[s];
                                    const float dmel = htan * a[2 + s];
                                    const float me = mel + p.eps;
                                    (s == 0 ? o2.x : o2.y) = do_log ? logf(me) : mel;
                                    (s == 0 ? t2.x : t2.y) = do_log ? dmel * __builtin_amdgcn_rcpf(me) : dmel;
                                });
                                if (out_bf16) *reinterpret_cast<unsigned*>(orow_h + tp) = (unsigned)bf16_bits(o2.x) | ((unsigned)bf16_bits(o2.y) << 16);
                                else *reinterpret_cast<float2*>(orow + tp) = o2;
                                if (trow) *reinterpret_cast<float2*>(trow + tp) = t2;
                            } else
                            static_for<0, 2>([&](auto ss) {
                                constexpr int s = decltype(ss)::value;
                                const int slot = mt * 8 + 2 * cg + s;
                                const int t = t0 + slot;
                                if (slot < SLOTS && t < p.T) {
                                    const float mel = 0.25f * a[s];
                                    const float dmel = htan * a[2 + s];
                                    if (do_log) {
                                        const float me = mel + p.eps;
                                        put(t, logf(me));
                                        if (trow) trow[t] = dmel * __builtin_amdgcn_rcpf(me);
                                    } else {
                                        put(t, mel);
                                        if (trow) trow[t] = dmel;
                                    }
                                }
                            });
                        } else {
                            // slot holds frames (2*slot, 2*slot+1) as (type 0, type 1)
                            static_for<0, 4>([&](auto ii) {
                                constexpr int i = decltype(ii)::value;
                                const int slot = mt * 8 + 2 * cg + (i & 1);
                                const int t = t0 + 2 * slot + (i >> 1);
                                if (slot < SLOTS && t < p.T) {
                                    const float mel = 0.25f * a[i];
                                    put(t, do_log ? logf(mel + p.eps) : mel);
                                }
                            });
                        }
                    });
                };

            if constexpr (HSPLIT) {
                // ================= phase 2, dense bank on the bf16 matrix pipe ==============================
                // One v_mfma_f32_16x16x32_bf16 covers 32 bins: lane (row = lane & 15, kg = lane >> 4) supplies bins 32 ks + 8 kg .. + 7 of
                // its row's plane (one ds_read_b128 for hi, one for lo) and of its column of the filterbank (two 16-byte loads of the
                // pre-split table).  Three instructions of 16 cycles per (k-step, row tile) -- lo hi + hi lo + hi hi, smallest first --
                // against eight exact-fp32 ones of 32: the dense contraction's matrix time falls from ~14 us to ~2.5 us at BASELINE
                // config 2.  Wave w owns mel tile w of every group whole (a dense bank has no narrow and wide tiles to balance):
                // no half-tile exchange.  Bin N/2 -- N/2 bins are N/64 steps exactly -- rides on the vector pipe.
                typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                constexpr int KS32 = N / 64;
                const int kgq = lane >> 4;
                int a_base[MT];
                static_for<0, MT>([&](auto mm) {
                    constexpr int mt = decltype(mm)::value;
                    const int slot = mt * 8 + slot8;
                    a_base[mt] = (slot < SLOTS ? slot : 0) * (SS * 8) + type * (2 * NHS * 2) + kgq * 16;
                });
                constexpr int TPWV = 8 / WAVES;                              // mel tiles of a group per wave (8 waves: 1, 4 waves: 2)
                for (int grp = 0; grp < p.groups; ++grp) {
                    static_for<0, TPWV>([&](auto jj) {
                        const int tile = grp * 8 + wave + WAVES * decltype(jj)::value;
                        if (16 * tile >= p.M) return;                       // (uniform per wave)
                        floatx4 acc[MT];
                        static_for<0, MT>([&](auto mm) { acc[decltype(mm)::value] = floatx4{0.f, 0.f, 0.f, 0.f}; });
                        const uint4* bt = p.ent_h + (size_t)tile * (KS32 * 2 * 64) + lane;
                        uint4 bh = bt[0], bl = bt[64];
                        for (int ks = 0; ks < KS32; ++ks) {
                            const uint4 ch = bh, cl = bl;
                            if (ks + 1 < KS32) { bh = bt[(ks + 1) * 128]; bl = bt[(ks + 1) * 128 + 64]; }
                            const bf16x8 vbh = __builtin_bit_cast(bf16x8, ch), vbl = __builtin_bit_cast(bf16x8, cl);
                            static_for<0, MT>([&](auto mm) {
                                constexpr int mt = decltype(mm)::value;
                                const bool valid = mt * 8 + slot8 < SLOTS;
                                uint4 ah = *reinterpret_cast<const uint4*>(smem_raw + a_base[mt] + ks * 64);
                                uint4 al = *reinterpret_cast<const uint4*>(smem_raw + a_base[mt] + NHS * 2 + ks * 64);
                                if constexpr (SLOTS < 8) { if (!valid) { ah = make_uint4(0, 0, 0, 0); al = ah; } }
                                const bf16x8 vah = __builtin_bit_cast(bf16x8, ah), val = __builtin_bit_cast(bf16x8, al);
                                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(val, vbh, acc[mt], 0, 0, 0);
                                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vah, vbl, acc[mt], 0, 0, 0);
                                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vah, vbh, acc[mt], 0, 0, 0);
                            });
                        }
                        // bin N/2: rows 4 cg + i of the accumulator layout -> (slot 2 cg + (i & 1), P | D = i >> 1)
                        const int mcol = 16 * tile + col;
                        const float fbn = mcol < p.M ? p.fb_nyq[mcol] : 0.f;
                        static_for<0, MT>([&](auto mm) {
                            constexpr int mt = decltype(mm)::value;
                            static_for<0, 4>([&](auto ii) {
                                constexpr int i = decltype(ii)::value;
                                const int slot = mt * 8 + 2 * cg + (i & 1);
                                if (slot < SLOTS) acc[mt][i] = fmaf(load_h(slot * (SS * 8), i >> 1, N / 2), fbn, acc[mt][i]);
                            });
                        });
                        write_tile(tile, acc);
                    });
                }
            } else
            for (int grp = 0; grp < p.groups; ++grp) {
                // acc[loc][mt][parity]: two accumulators per tile so that consecutive MFMAs never wait on each other
                floatx4 acc[NLOC][MT][2];
                static_for<0, NLOC>([&](auto l) { static_for<0, MT>([&](auto m) { static_for<0, 2>([&](auto e) {
                    acc[decltype(l)::value][decltype(m)::value][decltype(e)::value] = floatx4{0.f, 0.f, 0.f, 0.f}; }); }); });
                int tile_of[NLOC];
                int helpers = 0;         // bit w set = wave w's run 1 is a piece of THIS wave's tile (bits 16-23 of run 0's tile word)
                bool piece1 = (WAVES == 8);   // run 1 is a piece of another wave's tile (always with 8 waves; bit 30 of its tile word with 4)
                static_for<0, NLOC>([&](auto l) {
                    constexpr int loc = decltype(l)::value;
                    // (ks0, nks, boff, tile): the filterbank is banded, so the non-zero 4x16 blocks of one mel tile
                    // form ONE contiguous run of k-steps; nks is padded to a multiple of 4 with zero blocks
                    int4 tr = tr0[loc];
                    if (grp > 0) tr = p.tile_ranges[(grp * WAVES + wave) * NLOC + loc];
                    const int ks0 = __builtin_amdgcn_readfirstlane(tr.x), nks = __builtin_amdgcn_readfirstlane(tr.y);
                    const int boff = __builtin_amdgcn_readfirstlane(tr.z);
                    {
                        const int tw = __builtin_amdgcn_readfirstlane(tr.w);
                        tile_of[loc] = tw < 0 ? -1 : (tw & 0xffff);
                        if constexpr (loc == 0) helpers = tw < 0 ? 0 : ((tw >> 16) & 0xff);
                        if constexpr (loc == 1 && WAVES != 8) piece1 = tw >= 0 && ((tw >> 30) & 1) != 0;
                    }
#ifdef DMEL_ABLATE
                    if (p.flags & 0x800u) return;                       // timing ablation: skip the MFMA loop
#endif
                    if (nks <= 0) return;
                    const int bbase = (boff + lane) * 4;
                    // One group = 4 consecutive k-steps = 16 consecutive bins starting at a multiple of 16 (the host
                    // aligns every run to 4 k-steps), so the 4 reads of Z[k] share one base address and, except at one
                    // bin per 256, so do the 4 reads of the mirrored Z[N-k].
                    // A operands of one group: 4 LDS reads per 16-row tile
                    auto load_a = [&](int ksg, float (&av)[MT][4]) {
                        const int k0 = 4 * ksg + kofs;                               // bin of k-step 0 for this lane
                        const int zk0 = z_index<R, C>(k0 & (N - 1));
                        static_for<0, MT>([&](auto m) {
                            constexpr int mt = decltype(m)::value;
                            const int slot = mt * 8 + slot8;
                            const bool valid = slot < SLOTS;
                            // rows of type 0 read PD.x (|S|^2), rows of type 1 PD.y: the A operand is a plain 4-byte LDS read
                            const float* slf = reinterpret_cast<const float*>(lds + (valid ? slot : 0) * SS) + type;
                            static_for<0, 4>([&](auto uu) { constexpr int u = decltype(uu)::value; av[mt][u] = slf[2 * (zk0 + 4 * u)]; });
                        });
                    };
                    auto mfma4 = [&](const float (&av)[MT][4], float b0, float b1, float b2, float b3) {
                        const float bq[4] = {b0, b1, b2, b3};
                        static_for<0, MT>([&](auto m) {
                            constexpr int mt = decltype(m)::value;
                            const bool valid = mt * 8 + slot8 < SLOTS;
                            static_for<0, 4>([&](auto uu) {
                                constexpr int u = decltype(uu)::value;
                                float val = av[mt][u];
                                if constexpr (SLOTS < 8) val = valid ? val : 0.f;
                                acc[loc][mt][u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(val, bq[u], acc[loc][mt][u & 1], 0, 0, 0);
                            });
                        });
                    };
                    // With two waves per SIMD (n_fft >= 4096) nobody covers the LDS round trip between the reads of a group and its
                    // MFMAs: a wave with a long run spent ~85 cycles per MFMA instead of 32.  There the A operands of the NEXT group are
                    // requested before the MFMAs of this one (reads past the run land on valid bins and are not used).  With four
                    // waves per SIMD the same prefetch measured slower (23.5 against 23.4 us at config 2) and is not compiled in.
                    constexpr bool APRE = (N >= 4096);       // ESC-50 shape: 157.5 -> 154.5 us at n_fft 4096, 486 -> 476 at 8192
                    float a_cur[MT][4];
                    if constexpr (APRE) load_a(ks0, a_cur);
                    auto group4 = [&](int ksg, float b0, float b1, float b2, float b3) {
                        if constexpr (APRE) {
                            float a_nxt[MT][4];
                            load_a(ksg + 4, a_nxt);
                            mfma4(a_cur, b0, b1, b2, b3);
                            static_for<0, MT>([&](auto m) { static_for<0, 4>([&](auto uu) { a_cur[decltype(m)::value][decltype(uu)::value] = a_nxt[decltype(m)::value][decltype(uu)::value]; }); });
                        } else {
                            float av[MT][4];
                            load_a(ksg, av);
                            mfma4(av, b0, b1, b2, b3);
                        }
                    };
                    if (grp == 0) {
                        // k-steps whose B fragments are already in registers
                        static_for<0, NBPRE / 4>([&](auto qq) {
                            constexpr int i = decltype(qq)::value * 4;
                            // (running the padding groups too, branch-free on zero fragments, was measured: 24.6 against 21.9 us at
                            // config 2 -- the matrix pipe of a SIMD serialises the four waves that reach this phase together)
                            if (i < nks) group4(ks0 + i, bpre[loc][i], bpre[loc][i + 1], bpre[loc][i + 2], bpre[loc][i + 3]);
                        });
                    }
                    // the rest (long runs: dense custom filterbanks, further mel groups) streams with a 4-step prefetch
                    const int istart = (grp == 0) ? NBPRE : 0;
                    if (istart < nks) {
                        // a ring of DEPTH groups of 4 fragments in flight: one group ahead left the loop waiting for an L2 round trip per
                        // 4 MFMAs (dense banks -- a trainable filterbank -- and the long runs of n_fft >= 4096 live in this loop);
                        // offsets past the run are range-checked by the buffer descriptor or read the next run: never used
                        // (4 deep: n_fft 4096 at the reference's ESC-50 shape 171 -> 163 us, 8192 530 -> 512; at 2048 the 8 extra
                        // registers spill and config 3 went 49.8 -> 51.9 us, so the sizes with 128 VGPRs keep two groups)
                        constexpr int DEPTH = (N >= 4096) ? 4 : 2;
                        float br[DEPTH][4];
                        static_for<0, DEPTH>([&](auto dd) {
                            constexpr int d = decltype(dd)::value;
                            static_for<0, 4>([&](auto u) { br[d][decltype(u)::value] = buf_f32(rb, bbase + (istart + 4 * d + decltype(u)::value) * 256); });
                        });
                        for (int i = istart; i < nks; i += 4 * DEPTH) {
                            static_for<0, DEPTH>([&](auto dd) {
                                constexpr int d = decltype(dd)::value;
                                if (i + 4 * d < nks) {
                                    group4(ks0 + i + 4 * d, br[d][0], br[d][1], br[d][2], br[d][3]);
                                    static_for<0, 4>([&](auto u) { br[d][decltype(u)::value] = buf_f32(rb, bbase + (i + 4 * (d + DEPTH) + decltype(u)::value) * 256); });
                                }
                            });
                        }
                    }
                });
                STAMP(16 * ti + 9);   // MFMA loops
                // ---- epilogue: accumulators -> (B,1,M,T) ------------------------------------------
                floatx4 tot[NLOC][MT];
                static_for<0, NLOC>([&](auto l) { static_for<0, MT>([&](auto m) {
                    tot[decltype(l)::value][decltype(m)::value] = acc[decltype(l)::value][decltype(m)::value][0] + acc[decltype(l)::value][decltype(m)::value][1]; }); });
                if (WAVES == 8 || ((p.xch_groups >> (grp & 31)) & 1u)) {
                    // run 1 of a wave is a piece of some OTHER wave's tile (the host deals the k-steps of the wide tiles over the
                    // waves with narrow or no tiles of their own: build_tables): every wave leaves its run-1 sums in its own slot
                    // and an owner adds the slots of its helpers in ascending order (fixed order: deterministic)
                    // (one 16-row tile at a time through the same 8 KB: two tiles at once would put the 16-frame workgroup of
                    // n_fft 1024 past half of the CU's LDS)
                    floatx4* xch = reinterpret_cast<floatx4*>(smem_raw + SLOTS * SS * 8);
                    static_for<0, MT>([&](auto mm) {
                        constexpr int mt = decltype(mm)::value;
                        if constexpr (mt > 0) __syncthreads();
                        xch[wave * 64 + lane] = piece1 ? tot[1][mt] : floatx4{0.f, 0.f, 0.f, 0.f};
                        __syncthreads();
                        static_for<0, WAVES>([&](auto ss) {
                            constexpr int sw = decltype(ss)::value;
                            if (helpers & (1 << sw)) tot[0][mt] += xch[sw * 64 + lane];
                        });
                    });
                    if (p.groups > 1) __syncthreads();
                    if (piece1) tile_of[1] = -1;
                }
                STAMP(16 * ti + 10);  // half-tile exchange
#ifdef DMEL_ABLATE
                if (p.flags & 0x400u) continue;                        // timing ablation: skip the epilogue
#endif
                static_for<0, NLOC>([&](auto l) { write_tile(tile_of[decltype(l)::value], tot[decltype(l)::value]); });
                STAMP(16 * ti + 11);  // epilogue stores issued
            }
        }
    });
}

template <int N, int MODE, int TPW> static hipError_t launch_one(const FwdParams& p, int grid, hipStream_t s)
{
    constexpr FftGeom g = geom_mode<N, MODE>();
    constexpr int lds = g.LDS_BYTES;
    hipLaunchKernelGGL((dmel_fwd_kernel<N, MODE, TPW>), dim3(grid), dim3(g.THREADS), lds, s, p);
    return hipGetLastError();
}

// two tiles per workgroup are built for the sizes whose launches are large enough to use them (forward_tiles_per_wg)
template <int N, bool PAIR> constexpr bool has_tpw2() { return N >= 256 && (N <= 512 || (N == 1024 && geom<N, PAIR>().G == 64)); }   // (32 x 32 plan at 1024: 16-frame tiles, two of them spill)

// kTrainH (dense contraction on the bf16 matrix pipe) exists for the sizes whose frames live inside one wave, one tile per workgroup
template <int N> constexpr bool has_hsplit() { return N >= kHsplitMinNfft && N <= kHsplitMaxNfft; }

template <int N, int MODE> static hipError_t launch_mode(int tpw, const FwdParams& p, int grid, hipStream_t s)
{
    if constexpr (MODE == kTrainH) {
        if constexpr (has_hsplit<N>()) { if (tpw == 1) return launch_one<N, MODE, 1>(p, grid, s); }
        return hipErrorInvalidValue;
    } else if constexpr (MODE == kTrainW) {
        if constexpr (wlc_size(N)) { if (tpw == 1) return launch_one<N, MODE, 1>(p, grid, s); }
        return hipErrorInvalidValue;
    } else if constexpr (MODE == kTrainWW) {
        if constexpr (wlc_wide_size(N)) { if (tpw == 1) return launch_one<N, MODE, 1>(p, grid, s); }
        return hipErrorInvalidValue;
    } else {
        if constexpr (has_tpw2<N, mode_pairs(MODE)>()) { if (tpw == 2) return launch_one<N, MODE, 2>(p, grid, s); }
        if (tpw != 1) return hipErrorInvalidValue;
        return launch_one<N, MODE, 1>(p, grid, s);
    }
}

template <int N> static hipError_t launch_n(int mode, int tpw, const FwdParams& p, int grid, hipStream_t s)
{
    switch (mode) {
        case kTrain: return launch_mode<N, kTrain>(tpw, p, grid, s);
        case kInfer: return launch_mode<N, kInfer>(tpw, p, grid, s);
        case kSpec: return launch_mode<N, kSpec>(tpw, p, grid, s);
        case kSpecTrain: return launch_mode<N, kSpecTrain>(tpw, p, grid, s);
        case kTrainH: return launch_mode<N, kTrainH>(tpw, p, grid, s);
        case kTrainW: return launch_mode<N, kTrainW>(tpw, p, grid, s);
        case kTrainWW: return launch_mode<N, kTrainWW>(tpw, p, grid, s);
    }
    return hipErrorInvalidValue;
}

// -DDMEL_ONLY_NFFT=<n>: development builds that instantiate one transform size only (tools/build_variant.sh).  Never defined for
// libdmel_hip.so.
// -DDMEL_FWD_SPLIT -DDMEL_FWD_PART=<k>, k = 0..3: build.py compiles this file four times for libdmel_hip.so so that the
// instantiations of the large transforms -- minutes of compile time each -- build in parallel: part 0 holds everything that is
// not a template instantiation plus the sizes up to 512, parts 1-3 hold 1024 / 2048 + 16384 / 4096 + 8192 and nothing else.
// Without DMEL_FWD_SPLIT (the tools' one-command builds) everything is in this one translation unit.
#ifndef DMEL_FWD_PART
#define DMEL_FWD_PART 0
#endif
constexpr int fwd_part_of(int n) { return n <= 512 ? 0 : n == 1024 ? 1 : (n == 2048 || n == 16384) ? 2 : 3; }
#if defined(DMEL_ONLY_NFFT)
#define DMEL_FWD_HERE(n) ((n) == DMEL_ONLY_NFFT)
#elif defined(DMEL_FWD_SPLIT)
#define DMEL_FWD_HERE(n) (fwd_part_of(n) == DMEL_FWD_PART)
#else
#define DMEL_FWD_HERE(n) true
#endif
template <int N> static hipError_t launch_size(int mode, int tpw, const FwdParams& p, int grid, hipStream_t s)
{
    if constexpr (DMEL_FWD_HERE(N)) return launch_n<N>(mode, tpw, p, grid, s);
    else return hipErrorInvalidValue;
}

#define DMEL_CAT2(a, b) a##b
#define DMEL_CAT(a, b) DMEL_CAT2(a, b)
hipError_t DMEL_CAT(launch_forward_part, DMEL_FWD_PART)(int n_fft, int mode, int tpw, const FwdParams& p, int grid, hipStream_t s)
{
    switch (n_fft) {
        case 32: return launch_size<32>(mode, tpw, p, grid, s);
        case 64: return launch_size<64>(mode, tpw, p, grid, s);
        case 128: return launch_size<128>(mode, tpw, p, grid, s);
        case 256: return launch_size<256>(mode, tpw, p, grid, s);
        case 512: return launch_size<512>(mode, tpw, p, grid, s);
        case 1024: return launch_size<1024>(mode, tpw, p, grid, s);
        case 2048: return launch_size<2048>(mode, tpw, p, grid, s);
        case 4096: return launch_size<4096>(mode, tpw, p, grid, s);
        case 8192: return launch_size<8192>(mode, tpw, p, grid, s);
        case 16384: return launch_size<16384>(mode, tpw, p, grid, s);
    }
    return hipErrorInvalidValue;
}

#if DMEL_FWD_PART == 0
#if defined(DMEL_FWD_SPLIT) && !defined(DMEL_ONLY_NFFT)
#define DMEL_FWD_PARTS 1
hipError_t launch_forward_part1(int, int, int, const FwdParams&, int, hipStream_t);
hipError_t launch_forward_part2(int, int, int, const FwdParams&, int, hipStream_t);
hipError_t launch_forward_part3(int, int, int, const FwdParams&, int, hipStream_t);
hipError_t forward_prepare_attributes_part1();
hipError_t forward_prepare_attributes_part2();
hipError_t forward_prepare_attributes_part3();
#endif
hipError_t launch_forward(int n_fft, int mode, int tpw, const FwdParams& p, int grid, hipStream_t s)
{
#ifdef DMEL_FWD_PARTS
    switch (fwd_part_of(n_fft)) {
        case 1: return launch_forward_part1(n_fft, mode, tpw, p, grid, s);
        case 2: return launch_forward_part2(n_fft, mode, tpw, p, grid, s);
        case 3: return launch_forward_part3(n_fft, mode, tpw, p, grid, s);
    }
#endif
    return launch_forward_part0(n_fft, mode, tpw, p, grid, s);
}

bool forward_has_hsplit(int n_fft) { return n_fft >= kHsplitMinNfft && n_fft <= kHsplitMaxNfft && (n_fft & (n_fft - 1)) == 0; }
bool forward_has_wlc(int n_fft) { return wlc_size(n_fft); }
bool forward_wlc_one_frame(int n_fft) { return wlc_size(n_fft) && n_fft >= 2048; }      // (G = 64: the frame fills the wave)
bool forward_has_wlc_wide(int n_fft) { return wlc_wide_size(n_fft); }
bool forward_window_in_lds(int n_fft) { return n_fft >= kMinFastNfft && n_fft <= kWinLdsMaxNfft; }

// One place that maps a run-time (n_fft, pair) to the compile-time geometry
template <class F> static bool with_geom(int n_fft, bool pair, F&& f)
{
    switch (n_fft) {
        case 32: f(pair ? geom<32, true>() : geom<32, false>()); return true;
        case 64: f(pair ? geom<64, true>() : geom<64, false>()); return true;
        case 128: f(pair ? geom<128, true>() : geom<128, false>()); return true;
        case 256: f(pair ? geom<256, true>() : geom<256, false>()); return true;
        case 512: f(pair ? geom<512, true>() : geom<512, false>()); return true;
        case 1024: f(pair ? geom<1024, true>() : geom<1024, false>()); return true;
        case 2048: f(pair ? geom<2048, true>() : geom<2048, false>()); return true;
        case 4096: f(pair ? geom<4096, true>() : geom<4096, false>()); return true;
        case 8192: f(pair ? geom<8192, true>() : geom<8192, false>()); return true;
        case 16384: f(pair ? geom<16384, true>() : geom<16384, false>()); return true;
    }
    return false;
}

// (R, C) of the plan for n_fft: the host builds the twiddle tables from these
bool forward_plan_rc(int n_fft, bool pair, int* R, int* C)
{
    return with_geom(n_fft, pair, [&](const FftGeom& g) { *R = g.R; *C = g.C; });
}

int forward_lds_bytes(int n_fft, int mode)
{
    int v = -1;
    with_geom(n_fft, mode_pairs(mode), [&](const FftGeom& g) { v = g.LDS_BYTES; });
    if (mode == kTrainWW && n_fft == 1024 && wlc_wide_size(1024)) return geom<1024, false, true, true>().LDS_BYTES;
    if (mode == kTrainW) {
        switch (n_fft) {                                   // (the sizes kTrainW may be built for)
            case 512: v = geom<512, false, true>().LDS_BYTES; break;
            case 1024: v = geom<1024, false, true>().LDS_BYTES; break;
            case 2048: v = geom<2048, false, true>().LDS_BYTES; break;
            case 4096: v = geom<4096, false, true>().LDS_BYTES; break;
        }
    }
    return v;
}

int forward_frames_per_tile(int n_fft, int mode)
{
    if (mode == kTrainWW && n_fft == 1024 && wlc_wide_size(1024)) return geom<1024, false, true, true>().SLOTS;
    if (mode == kTrainW && n_fft == 512) return geom<512, false, true>().SLOTS;          // (its plan has its own number of waves)
    int slots = -1;
    with_geom(n_fft, mode_pairs(mode), [&](const FftGeom& g) { slots = g.SLOTS; });
    if (slots < 0) return -1;
    return mode_pairs(mode) ? 2 * slots : slots;
}

bool forward_two_tiles(int n_fft, int mode)
{
    if (mode_wlc(mode) || mode == kTrainH) return false;          // one tile per workgroup (launch_mode)
    const bool pr = mode_pairs(mode);
    switch (n_fft) {
        case 256: return pr ? has_tpw2<256, true>() : has_tpw2<256, false>();
        case 512: return pr ? has_tpw2<512, true>() : has_tpw2<512, false>();
        case 1024: return pr ? has_tpw2<1024, true>() : has_tpw2<1024, false>();
    }
    return false;
}

// Tiles per workgroup for a launch over `batch` clips of `tiles_per_clip` tiles.  A launch runs in rounds of the workgroups
// the chip holds at once (160 KB of LDS per CU, 256 CUs); a two-tile workgroup lives about 1.9 times as long as a one-tile
// one (it pays the prologue once: measured 21.97 against 23.1 us at BASELINE config 2 with the 16 x 16 x 4 plan, one round instead of
// two).  Two tiles are used when that model says the launch gets shorter -- e.g. not for 5 rounds becoming 3 double ones.
// Workgroups of the (n_fft, mode) instantiation the device holds at once: LDS-limited (160 KB per CU), at most 32 waves per CU
int forward_resident_workgroups(int n_fft, int mode)
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    const int lds = forward_lds_bytes(n_fft, mode), waves = (mode == kTrainW && n_fft == 512) ? geom<512, false, true>().WAVES : forward_waves(n_fft);
    int per_cu = (lds > 0 && 163840 / lds > 0) ? 163840 / lds : 1;
    if (waves > 0 && per_cu * waves > 32) per_cu = 32 / waves;
    return cus * (per_cu > 0 ? per_cu : 1);
}

int forward_tiles_per_wg(int n_fft, int mode, int batch, int tiles_per_clip)
{
    if (!forward_two_tiles(n_fft, mode) || tiles_per_clip < 2 || batch < 1) return 1;
    const long long resident = forward_resident_workgroups(n_fft, mode);
    const long long wg1 = (long long)batch * tiles_per_clip, wg2 = (long long)batch * ((tiles_per_clip + 1) / 2);
    const long long r1 = (wg1 + resident - 1) / resident, r2 = (wg2 + resident - 1) / resident;
    return 19 * r2 < 10 * r1 ? 2 : 1;
}

// the layout of tile_ranges / ent_pre depends on these two; they are the same for both plans of a size
int forward_waves(int n_fft)
{
    int v = -1;
    with_geom(n_fft, false, [&](const FftGeom& g) { v = g.WAVES; });
    return v;
}

int forward_nbpre(int n_fft)
{
    int v = -1;
    with_geom(n_fft, false, [&](const FftGeom& g) { v = g.NBPRE; });
    return v;
}

#endif   // DMEL_FWD_PART == 0

template <int N, int MODE, int TPW> static hipError_t set_attr()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&dmel_fwd_kernel<N, MODE, TPW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, geom_mode<N, MODE>().LDS_BYTES);
}
template <int N, int MODE> static hipError_t set_attr_mode()
{
    if constexpr (MODE == kTrainH) {
        if constexpr (has_hsplit<N>()) return set_attr<N, MODE, 1>();
        else return hipSuccess;
    } else if constexpr (MODE == kTrainW) {
        if constexpr (wlc_size(N)) return set_attr<N, MODE, 1>();
        else return hipSuccess;
    } else if constexpr (MODE == kTrainWW) {
        if constexpr (wlc_wide_size(N)) return set_attr<N, MODE, 1>();
        else return hipSuccess;
    } else {
        hipError_t e = set_attr<N, MODE, 1>();
        if (e != hipSuccess) return e;
        if constexpr (has_tpw2<N, mode_pairs(MODE)>()) return set_attr<N, MODE, 2>();
        else return hipSuccess;
    }
}
template <int N> static hipError_t set_attr_n()
{
    if constexpr (!DMEL_FWD_HERE(N)) return hipSuccess;
    else {
        static_assert(geom<N, true>().WAVES == geom<N, false>().WAVES && geom<N, true>().NBPRE == geom<N, false>().NBPRE,
                      "both plans of a size share the filterbank fragment layout");
        hipError_t e;
        if ((e = set_attr_mode<N, kTrain>()) != hipSuccess) return e;
        if ((e = set_attr_mode<N, kInfer>()) != hipSuccess) return e;
        if ((e = set_attr_mode<N, kSpec>()) != hipSuccess) return e;
        if ((e = set_attr_mode<N, kTrainH>()) != hipSuccess) return e;
        if ((e = set_attr_mode<N, kTrainW>()) != hipSuccess) return e;
        if ((e = set_attr_mode<N, kTrainWW>()) != hipSuccess) return e;
        return set_attr_mode<N, kSpecTrain>();
    }
}

hipError_t DMEL_CAT(forward_prepare_attributes_part, DMEL_FWD_PART)()
{
    hipError_t e;
    if ((e = set_attr_n<32>()) != hipSuccess) return e;
    if ((e = set_attr_n<64>()) != hipSuccess) return e;
    if ((e = set_attr_n<128>()) != hipSuccess) return e;
    if ((e = set_attr_n<256>()) != hipSuccess) return e;
    if ((e = set_attr_n<512>()) != hipSuccess) return e;
    if ((e = set_attr_n<1024>()) != hipSuccess) return e;
    if ((e = set_attr_n<2048>()) != hipSuccess) return e;
    if ((e = set_attr_n<4096>()) != hipSuccess) return e;
    if ((e = set_attr_n<8192>()) != hipSuccess) return e;
    return set_attr_n<16384>();
}

#if DMEL_FWD_PART == 0
hipError_t forward_prepare_attributes()
{
    hipError_t e;
    if ((e = forward_prepare_attributes_part0()) != hipSuccess) return e;
#ifdef DMEL_FWD_PARTS
    if ((e = forward_prepare_attributes_part1()) != hipSuccess) return e;
    if ((e = forward_prepare_attributes_part2()) != hipSuccess) return e;
    if ((e = forward_prepare_attributes_part3()) != hipSuccess) return e;
#endif
    return hipSuccess;
}
#endif

}  // namespace dmel

#ifdef DMEL_STAMPS
extern "C" int dmel_debug_read_stamps(unsigned long long* host, int count)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dmel::g_stamps), sizeof(unsigned long long) * (size_t)count);
}
#endif
