// dmel_aux.hip -- small kernels around the fused forward:
//   * dmel_dot_*   : lambd.grad = sum grad_out * tangent  (the whole backward of the layer once the
//                    forward has carried d out / d lambd; train.py:47 reaches exactly this scalar)
//   * dmel_naive_* : direct-DFT forward for n_fft < 32 (lambd < 5.5 samples) and as an on-device
//                    cross-check of the wave-FFT kernel; same semantics (models.py:33-56, :73).
#include <algorithm>
#include <cstdlib>
#include "dmel_kernels.h"
#include "dmel_ldsfft.h"

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
#ifndef DMEL_DOT_THREADS
#define DMEL_DOT_THREADS 256
#endif
// 8192 elements per workgroup and tensor whatever the workgroup size (the number of tickets on the one counter stays).  Measured at
// BASELINE config 2, trains of launches: 256 threads (two rounds of four 16-byte loads per tensor) 4.8 us, 512 threads (every load in
// flight at once) 4.9-5.1, 1024 threads 5.3 -- the kernel is bounded by its launch and its hand-off tail, not by the loads.
constexpr int kDotThreads = DMEL_DOT_THREADS;
__device__ __forceinline__ double block_sum(double v, double* red4)
{
    // inside each row of 16 lanes with DPP (no LDS round trips), then the rows of all waves through LDS
    v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);     // row_half_mirror
    v += dpp_f64<0x140>(v);     // row_mirror
    const int tid = threadIdx.x;
    if ((tid & 15) == 0) red4[tid >> 4] = v;
    __syncthreads();
    double s = 0.0;
    for (int q = 0; q < kDotThreads / 16; ++q) s += red4[q];
    return s;
}

// the constants of one update (shared by dmel_adam_kernel and the fused tail of dmel_dot_kernel: the same arithmetic, the same bits)
struct AdamConsts { float step, step_size, bc2_sqrt, omb1, omb2, b2, wd, eps; int maximize; };
__device__ __forceinline__ AdamConsts adam_consts(const AdamParams& p, float old_step)
{
    AdamConsts c;
    c.step = old_step + 1.0f;
    // the hyper-parameters are doubles, as torch passes them to its kernel: 1 - beta is formed in fp64 (in fp32 1 - 0.999f is off by
    // 5e-5 of itself, which the second moment would carry), the state stays fp32
    const float bc1 = (float)(1.0 - pow(p.beta1, (double)c.step)), bc2 = (float)(1.0 - pow(p.beta2, (double)c.step));
    c.step_size = (float)(p.lr / (double)bc1); c.bc2_sqrt = sqrtf(bc2);
    c.omb1 = (float)(1.0 - p.beta1); c.omb2 = (float)(1.0 - p.beta2); c.b2 = (float)p.beta2; c.wd = (float)p.weight_decay; c.eps = (float)p.eps;
    c.maximize = p.maximize;
    return c;
}
__device__ __forceinline__ void adam_update(const AdamParams& p, const AdamConsts& c, long long i, float g)
{
    const float w = p.param[i];
    if (c.maximize) g = -g;
    if (c.wd != 0.f) g = fmaf(w, c.wd, g);
    float m = p.exp_avg[i], v = p.exp_avg_sq[i];
    m = fmaf(c.omb1, g - m, m);                                  // lerp(m, g, 1 - beta1)
    v = fmaf(c.b2, v, c.omb2 * g * g);
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p.param[i] = w - c.step_size * m / denom;
    p.exp_avg[i] = m; p.exp_avg_sq[i] = v;
}

// four consecutive gradient elements as fp32: one 16-byte load (fp32) or one 8-byte load (bf16, exact widening)
template <bool GBF16> __device__ __forceinline__ float4 load_g4(const void* g, long long i)
{
    if constexpr (!GBF16) return reinterpret_cast<const float4*>(g)[i];
    else {
        const uint2 v = reinterpret_cast<const uint2*>(g)[i];
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                           __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    }
}
template <bool GBF16> __device__ __forceinline__ float load_g1(const void* g, long long k)
{
    if constexpr (!GBF16) return reinterpret_cast<const float*>(g)[k];
    else return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(g)[k] << 16);
}

// The exchange step of the mailbox reducer (MailboxArgs in dmel_kernels.h), executed by ONE lane: returns the sum over ranks.
__device__ __forceinline__ float mailbox_exchange(const MailboxArgs& mb, float local)
{
    const unsigned step = *mb.step + 1u;
    *mb.step = step;
    const unsigned long long mine = ((unsigned long long)step << 32) | __builtin_bit_cast(unsigned, local);
    const int par = (int)(step & 1u) * mb.world;
    for (int r = 0; r < mb.world; ++r)                               // own inbox included: one code path, and the sum is in rank order
        __hip_atomic_store(mb.peer_inbox[r] + par + mb.rank, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    float sum = 0.f;
    bool ok = true;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();       // 100 MHz, does not change with the shader clock
    for (int r = 0; r < mb.world; ++r) {
        unsigned long long w = 0;
        unsigned spins = 0;
        for (;;) {
            w = __hip_atomic_load(mb.my_inbox + par + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(w >> 32) == step) break;
            ++spins;
            if (mb.spin_limit && spins >= mb.spin_limit) { ok = false; break; }
            if (mb.timeout_ticks && (spins & 63u) == 0u && __builtin_amdgcn_s_memrealtime() - t_start >= mb.timeout_ticks) { ok = false; break; }
            if (spins < 4096u) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);   // a long wait polls less often
        }
        if (!ok) {
            if (mb.host_error)
                __hip_atomic_store(mb.host_error, ((unsigned long long)step << 32) | (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return __builtin_nanf("");
        }
        sum += __builtin_bit_cast(float, (unsigned)w);
    }
    return sum;
}

__global__ void __launch_bounds__(64) dmel_mailbox_kernel(float* buf, MailboxArgs mb)
{
    if (threadIdx.x == 0) buf[0] = mailbox_exchange(mb, buf[0]);
}

hipError_t launch_mailbox_allreduce(float* buf, const MailboxArgs& mb, hipStream_t s)
{
    hipLaunchKernelGGL(dmel_mailbox_kernel, dim3(1), dim3(64), 0, s, buf, mb);
    return hipGetLastError();
}

// `adam.param != nullptr` (dmel_plan_attach_adam): the workgroup that draws the last ticket also applies Adam's update of the ONE parameter
// this gradient belongs to (lambd) -- dmel_adam_kernel's arithmetic on the value it has just written --, so a step whose only
// trainable parameter of the layer is lambd needs no optimizer launch at all
template <bool GBF16, bool MBOX>
__global__ void __launch_bounds__(kDotThreads) dmel_dot_kernel(const void* __restrict__ g, const float* __restrict__ t,
                                                            long long count, double* partials, unsigned* counter,
                                                            int accumulate, float* result, MailboxArgs mb, AdamParams adam)
{
    __shared__ double red4[kDotThreads / 16], red4b[kDotThreads / 16];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const long long stride = (long long)gridDim.x * kDotThreads;
    double acc = 0.0;
    const uintptr_t galign = GBF16 ? 7 : 15;
    const long long n4 = ((reinterpret_cast<uintptr_t>(g) & galign) | (reinterpret_cast<uintptr_t>(t) & 15)) == 0 ? count / 4 : 0;
    const float4* t4 = reinterpret_cast<const float4*>(t);
    long long i = (long long)blockIdx.x * kDotThreads + tid;
    for (; i + 3 * stride < n4; i += 4 * stride) {        // four independent 16-byte loads per tensor in flight
        const float4 a0 = load_g4<GBF16>(g, i), a1 = load_g4<GBF16>(g, i + stride), a2 = load_g4<GBF16>(g, i + 2 * stride), a3 = load_g4<GBF16>(g, i + 3 * stride);
        const float4 b0 = t4[i], b1 = t4[i + stride], b2 = t4[i + 2 * stride], b3 = t4[i + 3 * stride];
        acc += ((double)a0.x * (double)b0.x + (double)a0.y * (double)b0.y) + ((double)a0.z * (double)b0.z + (double)a0.w * (double)b0.w);
        acc += ((double)a1.x * (double)b1.x + (double)a1.y * (double)b1.y) + ((double)a1.z * (double)b1.z + (double)a1.w * (double)b1.w);
        acc += ((double)a2.x * (double)b2.x + (double)a2.y * (double)b2.y) + ((double)a2.z * (double)b2.z + (double)a2.w * (double)b2.w);
        acc += ((double)a3.x * (double)b3.x + (double)a3.y * (double)b3.y) + ((double)a3.z * (double)b3.z + (double)a3.w * (double)b3.w);
    }
    for (; i < n4; i += stride) {
        const float4 a = load_g4<GBF16>(g, i), b = t4[i];
        acc += ((double)a.x * (double)b.x + (double)a.y * (double)b.y) + ((double)a.z * (double)b.z + (double)a.w * (double)b.w);
    }
    for (long long k = n4 * 4 + (long long)blockIdx.x * kDotThreads + tid; k < count; k += stride)
        acc += (double)load_g1<GBF16>(g, k) * (double)t[k];
    const double bsum = block_sum(acc, red4);
    if (tid == 0) {
        __hip_atomic_store(&partials[blockIdx.x], bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // two levels of tickets (dmel_kernels.h: kDotGroup): a group's counter orders its members' partials before the group's last member,
        // the kernel's counter orders the groups' last members before the one that combines -- every store completed (vmcnt(0)) before its
        // workgroup's first atomic was issued, and an atomic's return orders what follows it
        const unsigned grp = blockIdx.x / kDotGroup, ngroups = (gridDim.x + kDotGroup - 1) / kDotGroup;
        const unsigned gsize = grp + 1 < ngroups ? (unsigned)kDotGroup : gridDim.x - grp * kDotGroup;
        unsigned* gc = dot_group_counters(counter) + 16 * grp;
        bool last = false;
        if (__hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
            __hip_atomic_store(gc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                 // armed for the next launch
            last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1u;
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    double sum = 0.0;
    for (int q = tid; q < (int)gridDim.x; q += kDotThreads)
        sum += __hip_atomic_load(&partials[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double total = block_sum(sum, red4b);
    if (tid == 0) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (MBOX) {
            // data-parallel ranks: what is accumulated is the sum over ranks of this step's local sums
            const float all = mailbox_exchange(mb, (float)total);
            result[0] = accumulate ? result[0] + all : all;
        } else {
            result[0] = accumulate ? (float)((double)result[0] + total) : (float)total;
        }
        if (adam.param) {
            const AdamConsts c = adam_consts(adam, *adam.step);
            adam_update(adam, c, 0, result[0]);
            *adam.step = c.step;
        }
    }
}

int dot_blocks_for(long long count, int max_partials)
{
    // (round 6: 4096 floats per workgroup and tensor -- 4.0-4.1 us at BASELINE config 2 against 4.3 for 2048 or 8192 --, at most kDotMaxBlocks workgroups -- the tickets are a two-level tree now.  Until round 5:
    // 8192 per workgroup, few workgroups on purpose -- every workgroup ended with one ticket on a single counter, ~11.7 ns each, serialised)
    static const long long per_wg = [] { const char* e = std::getenv("DMEL_DOT_PER_WG"); const long long v = e ? std::atoll(e) : 0; return v >= 1024 ? v : 4096LL; }();   // (diagnostics)
    const long long want = (count + per_wg - 1) / per_wg;
    const long long cap = max_partials < kDotMaxBlocks ? max_partials : kDotMaxBlocks;
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

hipError_t launch_dot(const void* g, int g_bf16, const float* t, long long count, int accumulate, double* partials,
                      unsigned* counter, int max_partials, float* result, hipStream_t s, const MailboxArgs* mb, const AdamParams* fused)
{
    const int blocks = dot_blocks_for(count, max_partials);
    const MailboxArgs none{};
    AdamParams adam{};
    if (fused) adam = *fused;
    const dim3 gr(blocks), bl(kDotThreads);
    if (mb) {
        if (g_bf16) hipLaunchKernelGGL((dmel_dot_kernel<true, true>), gr, bl, 0, s, g, t, count, partials, counter, accumulate, result, *mb, adam);
        else hipLaunchKernelGGL((dmel_dot_kernel<false, true>), gr, bl, 0, s, g, t, count, partials, counter, accumulate, result, *mb, adam);
    } else {
        if (g_bf16) hipLaunchKernelGGL((dmel_dot_kernel<true, false>), gr, bl, 0, s, g, t, count, partials, counter, accumulate, result, none, adam);
        else hipLaunchKernelGGL((dmel_dot_kernel<false, false>), gr, bl, 0, s, g, t, count, partials, counter, accumulate, result, none, adam);
    }
    return hipGetLastError();
}

// ---- filterbank tables from a device matrix -----------------------------------------------------------------
// grid.x = runs (one (group, wave, run) entry of tile_ranges each) + the workgroups that copy / transpose the matrix (one
// workgroup doing all of it took 64 us for 513 x 128 entries: the longest kernel of a trainable-filterbank step)
constexpr int kRepackRunSplit = 8;
__global__ void __launch_bounds__(256) dmel_repack_kernel(RepackParams p)
{
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= p.runs * kRepackRunSplit) {
        // one kind of work per workgroup, at most four entries per thread: [split-bf16 fragments][plain copy + transpose][packed rows +
        // Nyquist row] (round 4: 65 workgroups that each walked through all of it in turn took 7 us -- four dependent passes)
        int blk = (int)blockIdx.x - p.runs * kRepackRunSplit;
        const long long n = (long long)p.F * p.M;
        if (blk < p.blocks_h) {
            // the split-bf16 fragments of kTrainH (build_tables is the host's version of this loop): one 16-byte entry per (block, lane)
            const int NT = (p.M + 15) / 16;
            const long long ne = (long long)NT * p.ks32 * 64;
            const long long q = (long long)blk * 256 + tid;
            if (q < ne) {
                const int l = (int)(q & 63);
                const long long fblk = q >> 6;                             // tile * ks32 + kstep
                const int tile = (int)(fblk / p.ks32), ks = (int)(fblk % p.ks32);
                const int m = 16 * tile + (l & 15), f0 = 32 * ks + 8 * (l >> 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = m < p.M ? p.fb[(size_t)(f0 + e) * p.M + m] : 0.f;
                unsigned hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const unsigned short h0 = bf16_bits(v[e]), h1 = bf16_bits(v[e + 1]);
                    hi[e / 2] = (unsigned)h0 | ((unsigned)h1 << 16);
                    lo[e / 2] = (unsigned)bf16_bits(v[e] - __uint_as_float((unsigned)h0 << 16)) | ((unsigned)bf16_bits(v[e + 1] - __uint_as_float((unsigned)h1 << 16)) << 16);
                }
                p.ent_h[fblk * 128 + l] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                p.ent_h[fblk * 128 + 64 + l] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            }
            return;
        }
        blk -= p.blocks_h;
        if (blk < p.blocks_w) {
            // the B operands of kTrainW (build_tables is the host's version): one 16-byte entry = four steps of one lane
            const long long q = (long long)blk * 256 + tid;
            if (q < (long long)p.wl_total4 * 64) {
                const int l = (int)(q & 63);
                int s4 = (int)(q >> 6), ph = 0;
                while (ph + 1 < p.wl_phases && s4 >= p.wl_len4[ph]) { s4 -= p.wl_len4[ph]; ++ph; }
                const int2 li = p.wl_lane[ph * 64 + l];
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (li.y >= 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int f = li.x / 8 + 4 * s4 + u; v[u] = f < p.F ? p.fb[(size_t)f * p.M + li.y] : 0.f; }
                }
                p.wl_b4[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
            return;
        }
        blk -= p.blocks_w;
        if (blk < p.blocks_c) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long i = ((long long)blk * 4 + u) * 256 + tid; v[u] = i < n ? p.fb[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long i = ((long long)blk * 4 + u) * 256 + tid;
                if (i < n) {
                    p.fb_dense[i] = v[u];
                    if (p.fbT) { const int f = (int)(i / p.M), m = (int)(i % p.M); p.fbT[(size_t)m * p.F + f] = v[u]; }
                }
            }
            return;
        }
        blk -= p.blocks_c;
        // packed rows of the waveform gradient (dense structure: every row starts at column 0 and has M columns); the Nyquist row of kTrainH
        const long long f = (long long)blk * 256 + tid;
        if (p.rowpk && f < p.F)
            p.rowpk[f] = make_float4(p.fb[f * p.M], p.M > 1 ? p.fb[f * p.M + 1] : 0.f, __builtin_bit_cast(float, 0), __builtin_bit_cast(float, p.M));
        if (p.ent_h && f < p.M) p.fb_nyq[f] = p.fb[(size_t)(p.F - 1) * p.M + f];
        return;
    }
    const int run = blockIdx.x / kRepackRunSplit, part = blockIdx.x % kRepackRunSplit;       // a run's entries dealt over kRepackRunSplit workgroups
    const int4 tr = p.tile_ranges[run];                 // (first k-step, #k-steps, offset into ent_b, mel tile)
    if (tr.y <= 0 || tr.w < 0) return;
    for (int i = part * 256 + tid; i < tr.y * 64; i += 256 * kRepackRunSplit) {
        const int ks = tr.x + i / 64, l = i % 64;
        const int f = 4 * ks + (l >> 4), m = 16 * (tr.w & 0xffff) + (l & 15);        // (bits 16-23: the helper mask of dmel_fwd_kernel)
        const float v = (f < p.F && m < p.M) ? p.fb[(size_t)f * p.M + m] : 0.f;
        p.ent_b[tr.z + i] = v;
        if (run < p.runs_group0 && i / 64 < p.nbpre) p.ent_pre[((size_t)run * p.nbpre + i / 64) * 64 + l] = v;
    }
}

hipError_t launch_repack(const RepackParams& p0, hipStream_t s)
{
    RepackParams p = p0;
    const long long n = (long long)p.F * p.M;
    const int NT = (p.M + 15) / 16;
    p.blocks_h = p.ent_h ? (int)(((long long)NT * p.ks32 * 64 + 255) / 256) : 0;
    p.blocks_w = p.wl_b4 ? (int)(((long long)p.wl_total4 * 64 + 255) / 256) : 0;
    p.blocks_c = (int)((n + 1023) / 1024);
    const int blocks_r = (std::max(p.F, p.M) + 255) / 256;
    hipLaunchKernelGGL(dmel_repack_kernel, dim3((unsigned)(p.runs * kRepackRunSplit + p.blocks_h + p.blocks_w + p.blocks_c + blocks_r)), dim3(256), 0, s, p);
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
    const float* xb = resolve_x(p.x, p.x_ind) + (size_t)b * p.L;
    const bool spec_mode = (p.mode == kSpec || p.mode == kSpecTrain);
    const LamState ls = lam_prologue(p.lam, p.N, blockIdx.x == 0 && tid == 0);
    if (ls.action != kLamRun) {
        if (ls.action == kLamPoison) {           // no launch of this forward matched the device lambd (dmel_kernels.h)
            const int rows = spec_mode ? p.F : p.M;
            for (int rr = tid; rr < rows; rr += blockDim.x) {
                const size_t o = ((size_t)b * rows + rr) * p.T + t;
                if (p.flags & 4u) reinterpret_cast<unsigned short*>(p.out)[o] = 0x7fc0u; else p.out[o] = __builtin_nanf("");
                if (p.tangent) p.tangent[o] = __builtin_nanf("");
            }
        }
        return;
    }
    const float ctan = lam_tangent_scale(ls);
    float mean = 0.f;
    if (p.remove_dc) {
        mean = clip_mean_psum(p.psum, p.nchunks, b, p.L);
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
    if (spec_mode) {
        for (int f = tid; f < p.F; f += blockDim.x) {
            const size_t o = ((size_t)b * p.F + f) * p.T + t;
            p.out[o] = P[f];
            if (p.tangent) p.tangent[o] = ctan * D[f];
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
        dmel *= ctan;
        const size_t o = ((size_t)b * p.M + m) * p.T + t;
        const bool out_bf16 = (p.flags & 4u) != 0;
        auto put = [&](float v) { if (out_bf16) reinterpret_cast<unsigned short*>(p.out)[o] = bf16_bits(v); else p.out[o] = v; };
        if (do_log) {
            const float me = mel + p.eps;
            put(logf(me));
            if (p.tangent) p.tangent[o] = dmel / me;
        } else {
            put(mel);
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

// ---- Adam on the layer's parameters (lambd; the filterbank matrix) -------------------------------------------
// torch.optim.Adam's update (main.py:52-53; the fused / capturable variant's arithmetic: fp32 state, step count on the device) as
// ONE launch of as many workgroups as the parameter needs: torch launches two kernels for the scalar lambd (a quarter of a
// 33 us step) and puts the 65 664-entry filterbank of config 2 on two workgroups (42 us).  Opt-in.
// Every workgroup reads the step count before it draws a ticket; the one that draws the last ticket therefore knows that all have
// read it, writes the new count and re-arms the ticket word for the next launch on the stream.
constexpr int kAdamPerThread = 4;
__global__ void __launch_bounds__(256) dmel_adam_kernel(AdamParams p)
{
    const AdamConsts c = adam_consts(p, __hip_atomic_load(p.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const float step = c.step;
    const long long stride = 256LL * gridDim.x;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.n; i += stride) adam_update(p, c, i, p.grad[i]);
    __syncthreads();                                               // every thread of this workgroup holds the old count
    if (threadIdx.x == 0) {
        // (relaxed: the ticket orders nothing but itself -- every workgroup's read of the count is complete before its ticket is
        // drawn because the update above consumed it; a release fence here would write back the XCD's L2 once per workgroup)
        if (gridDim.x == 1) *p.step = step;
        else {
            const unsigned t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                __hip_atomic_store(p.step, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int adam_grid(long long n)
{
    const long long wgs = (n + 256 * kAdamPerThread - 1) / (256 * kAdamPerThread);
    return (int)(wgs < 1 ? 1 : (wgs > 1024 ? 1024 : wgs));
}

hipError_t launch_adam(const AdamParams& p, hipStream_t s)
{
    hipLaunchKernelGGL(dmel_adam_kernel, dim3((unsigned)adam_grid(p.n)), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---- filterbank gradient ----------------------------------------------------------------------
// grad_fb = sum over clips of  spec_b (F x T) * gm_b^T (T x M): per clip a small GEMM whose K dimension (time) is
// contiguous in both operands; exact-fp32 MFMA 16x16x4.  (Rounds 1-2: a 32 x 128 tile per 4-wave workgroup with both operands
// read straight from global memory, a pre-pass for gm, 61 batch slices at config 2: 32 us for GEMM + reduction.)
typedef float floatx4_t __attribute__((ext_vector_type(4)));
template <int I> struct IdxC { static constexpr int value = I; };
template <int B, int E, class Fn> __device__ __forceinline__ void unrolled(Fn&& f)
{
    if constexpr (B < E) { f(IdxC<B>{}); unrolled<B + 1, E>(f); }
}
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };       // 16-byte load that only needs 4-byte alignment

// One workgroup of 8 waves owns a 64 (freq) x 128 (mel) tile of grad_fb for one slice of the batch: wave (wf, wm) the 32 x 32
// sub-tile (2 x 2 MFMA tiles).  A K-block is 16 time steps of one clip: spec[b][f0 .. f0+63][t .. t+15] and gm[b][0 .. 127][t .. t+15]
// (12 KB) are fetched ONCE per workgroup as 16-byte pieces along t -- the 32 x 128 kernel above fetched the spectrogram rows
// four times, once per wave -- parked in registers while the previous block is multiplied, and written to one of two LDS images
// whose rows are padded to 24 floats (the 16 lanes of a ds_read_b128 group then touch 16 different 16-byte slots).  For the log
// output gm = grad_out * exp(-out) is formed while staging (no pre-pass, no gm workspace).  Same K order in A and B (lane
// (row, kq) holds t = 16 j + 4 kq + u for k-step u), slices summed in index order by dmel_fbgrad_reduce_kernel: deterministic.
#ifdef DMEL_STAMPS
__device__ unsigned long long g_fstamps[1024 * 8 * 8];
__device__ __forceinline__ void fstamp(int idx)
{
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int wg = blockIdx.x + gridDim.x * blockIdx.y;
    if ((threadIdx.x & 63) == 0 && wg < 1024) g_fstamps[((size_t)wg * 8 + (threadIdx.x >> 6)) * 8 + idx] = t;
}
#define FSTAMP(i) fstamp(i)
#else
#define FSTAMP(i) do {} while (0)
#endif
constexpr int kFbgBF = 64, kFbgBM = 128, kFbgRow = 24, kFbgThreads = 512;     // rows of 24 floats: ds_read_b128 of 16 rows x 4 pieces is conflict-free (20: 2-way)

// BF16X3 (opt-in, DMEL_FLAG_MFMA_BF16X3): the same GEMM on the bf16 matrix pipe, 16 x the fp32 MFMA rate, with both operands split
// into two bf16 halves as they are parked (hi = bf16(v), lo = bf16(v - hi): v = hi + lo to 2^-17) and three products accumulated in
// fp32 -- hi hi + lo hi + hi lo; the dropped lo lo term is 2^-16 of a product (1.7e-5 on the mel contraction, against the
// 1e-4 bar; plain bf16 operands miss it by 60 x).  One v_mfma_f32_32x32x16_bf16 per term covers a wave's 32 x 32 sub-tile and a whole
// K-block of 16 time steps: 3 instructions of 32 cycles where the exact path issues 16 of 32.  LDS images: rows of 16 bf16 padded to
// 24 (48 B: the 16 lanes of a ds_read_b128 group -- 16 different rows mod 16 -- touch 16 different 16-byte slots).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float floatx16_t __attribute__((ext_vector_type(16)));
constexpr int kFbgRowH = 24;                                       // bf16 entries per row of a BF16X3 image
__device__ __forceinline__ void split_bf16x4(float4 v, uint2& hi, uint2& lo)
{
    // round to nearest even both times (v_cvt_pk_bf16_f32); the bf16 bits of hi widened back are exact
    const unsigned short h0 = bf16_bits(v.x), h1 = bf16_bits(v.y), h2 = bf16_bits(v.z), h3 = bf16_bits(v.w);
    hi = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
    const float r0 = v.x - __uint_as_float((unsigned)h0 << 16), r1 = v.y - __uint_as_float((unsigned)h1 << 16);
    const float r2 = v.z - __uint_as_float((unsigned)h2 << 16), r3 = v.w - __uint_as_float((unsigned)h3 << 16);
    lo = make_uint2((unsigned)bf16_bits(r0) | ((unsigned)bf16_bits(r1) << 16), (unsigned)bf16_bits(r2) | ((unsigned)bf16_bits(r3) << 16));
}

// d lambd inside the filterbank gradient's launch (FbGradParams::dot_*): dmel_dot_kernel<false, false> restated for a workgroup of 512
// threads that plays four of its 256-thread blocks, two at a time.  Same elements per (virtual) thread in the same order, same DPP
// rows, same 16-entry sums, the same walk over the partials by the workgroup that draws the last ticket: the result has the bits of
// the stand-alone kernel.  One launch (~4.4 us inside a captured step) less per trainable-filterbank step.
__device__ __forceinline__ double vblock_sum256(double v, double* red16, int vt)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    if ((vt & 15) == 0) red16[vt >> 4] = v;
    __syncthreads();
    double s = 0.0;
    for (int q = 0; q < 16; ++q) s += red16[q];
    return s;
}

__device__ void fbgrad_dot_body(const FbGradParams& p, int wg)
{
    static_assert(kDotThreads == 256, "the fused dot plays dmel_dot_kernel's 256-thread blocks");
    __shared__ double dred[2][2][16];
    __shared__ int is_last;
    const int tid = threadIdx.x, vt = tid & 255, half = tid >> 8;
    const float* g = p.dot_g; const float* t = p.dot_t;
    const long long count = p.dot_count;
    const long long stride = (long long)p.dot_vblocks * 256;
    const long long n4 = ((reinterpret_cast<uintptr_t>(g) & 15) | (reinterpret_cast<uintptr_t>(t) & 15)) == 0 ? count / 4 : 0;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    for (int pass = 0; pass < 2; ++pass) {
        const int vb = wg * 4 + pass * 2 + half;
        double acc = 0.0;
        if (vb < p.dot_vblocks) {
            long long i = (long long)vb * 256 + vt;
            for (; i + 3 * stride < n4; i += 4 * stride) {
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
            for (long long k = n4 * 4 + (long long)vb * 256 + vt; k < count; k += stride)
                acc += (double)g[k] * (double)t[k];
        }
        const double bsum = vblock_sum256(acc, dred[pass][half], vt);
        if (vt == 0 && vb < p.dot_vblocks) {
            __hip_atomic_store(&p.dot_partials[vb], bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();                                   // both storing threads have their acknowledgements
    if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(p.dot_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (ticket == (unsigned)p.dot_wgs - 1u);
    }
    __syncthreads();
    if (!is_last) return;
    double sum = 0.0;
    if (half == 0)
        for (int q = vt; q < p.dot_vblocks; q += 256)
            sum += __hip_atomic_load(&p.dot_partials[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double total = vblock_sum256(sum, dred[0][half], vt);
    if (tid == 0) {
        __hip_atomic_store(p.dot_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p.dot_result[0] = p.dot_accumulate ? (float)((double)p.dot_result[0] + total) : (float)total;
    }
}

template <bool LOG, bool TINY, bool BF16X3 = false>
__global__ void __launch_bounds__(kFbgThreads) dmel_fbgrad_lds_kernel(FbGradParams p)
{
    if (p.dot_wgs > 0 && (int)blockIdx.y >= p.splits) {
        const int wg = ((int)blockIdx.y - p.splits) * (int)gridDim.x + (int)blockIdx.x;
        if (wg < p.dot_wgs) fbgrad_dot_body(p, wg);
        return;
    }
    // exact path: fp32 images, rows of 24 floats.  BF16X3: [hi | lo] images of bf16, rows of 24 entries -- the same bytes either way
    constexpr int kImgA = BF16X3 ? (kFbgBF + 1) * kFbgRowH : (kFbgBF + 1) * kFbgRow;     // floats (+ the folded last row, see `fold`)
    constexpr int kImgB = BF16X3 ? kFbgBM * kFbgRowH : kFbgBM * kFbgRow;
    __shared__ __attribute__((aligned(16))) float lds_a[2][kImgA];
    __shared__ __attribute__((aligned(16))) float lds_b[2][kImgB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    FSTAMP(0);
    const int row = lane & 15, kq = lane >> 4;
    const int wf = wave >> 2, wm = wave & 3;
    const int f0 = blockIdx.x * kFbgBF, m0 = blockIdx.z * kFbgBM;
    const int split = blockIdx.y;
    const int F = p.F, M = p.M, T = p.T;
    // a slice = a run of consecutive K-blocks (16 time steps of one clip) of the batch, cut at block -- not clip -- boundaries:
    // the slices differ by at most one block
    // (the cuts come from the host as quotient and remainder: 64-bit divisions here cost the workgroup microseconds before its
    // first request -- 9 k cycles from the first instruction to the first load issued, a quarter of a wave's life at config 2)
    const int ntc = p.ntc;
    const int blk_lo = split * p.blk_base + min(split, p.blk_rem);
    const int total = p.blk_base + (split < p.blk_rem ? 1 : 0);
    const int b_lo = (int)((unsigned)blk_lo / (unsigned)ntc);
    // staging roles: every thread one 16-byte piece of the B tile (row tid / 4, piece tid % 4), threads 0 .. 255 one of the A tile
    const int srow = tid >> 2, sc = tid & 3;
    const int mrow = min(m0 + srow, M - 1);                       // rows past the edge: valid memory, results never stored
    // n_fft / 2 + 1 rows = a multiple of 64 plus ONE: a ninth tile for the Nyquist row would cost a workgroup slot per slice
    // (56 of 512 at BASELINE config 2) for 1/64 of a tile's work.  Instead the workgroups of the LAST full tile stage that row as a
    // 65th row of their A image (threads 256 .. 259, whose A registers are otherwise unused) and carry its 128 dot products on the
    // vector pipe: four FMAs per thread and K-block, summed over the four threads of a column in a fixed order at the end.
    const bool fold = p.fold_last_row && blockIdx.x == gridDim.x - 1;
    const int frow = tid < 4 * kFbgBF ? min(f0 + srow, F - 1) : F - 1;
    const bool has_a = tid < 4 * kFbgBF || (fold && tid < 4 * kFbgBF + 4);
    // DEPTH blocks of loads in flight per thread (one block ahead left every iteration waiting for a memory round trip).  Blocks are
    // requested strictly in order, so the addresses advance by increments (no division, no 64-bit products per block), and what is
    // loaded is only touched when it is parked in LDS -- exp(-out) included: applied at request time it made every request wait
    // for its own loads (config 3, log output: 143 -> 126 us for the whole gradient).
    constexpr int DEPTH = 3;
    float4 ring_a[DEPTH], ring_g[DEPTH], ring_y[DEPTH];
    int nblk = blk_lo - b_lo * ntc;                                 // 16-step block (inside its clip) of the NEXT request
    int left = total;                                               // requests still to be made
    const float* pg = p.grad_out + ((size_t)b_lo * M + mrow) * T;
    const float* py = LOG ? p.out + ((size_t)b_lo * M + mrow) * T : nullptr;
    const float* pa = p.spec + ((size_t)b_lo * F + frow) * T;
    const size_t step_g = (size_t)M * T, step_a = (size_t)F * T;
    // Requests are branch-free 16-byte loads that land in the ring registers themselves: a piece that would cross the end of its row
    // is read from the row's last four entries instead (always in bounds) and shifted into place when it is PARKED, where the
    // data is touched anyway.  With a branch around the loads -- element-wise loads for such pieces -- the two paths met in a phi,
    // the compiler loaded into temporaries and moved them (s_waitcnt vmcnt right behind every request): the ring never held
    // more than one block in flight, 2 k cycles per request in the prologue alone (tools/fstamps.py).  Every thread requests a
    // piece of A (threads that park none re-read one row: 16 bytes of L1 traffic).  Rows shorter than a piece: TINY, element-wise.
    int ring_t[DEPTH];
    auto fetch = [&](float4& ra, float4& rg, float4& ry, int& rt) {
        if (left <= 0) return;
        --left;
        const int t = nblk * 16 + 4 * sc;
        rt = t;
        if constexpr (!TINY) {
            const int tl = min(t, T - 4);
            const f4u g = *reinterpret_cast<const f4u*>(pg + tl);
            const f4u a = *reinterpret_cast<const f4u*>(pa + tl);
            rg = make_float4(g.v[0], g.v[1], g.v[2], g.v[3]);
            ra = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
            if constexpr (LOG) { const f4u y = *reinterpret_cast<const f4u*>(py + tl); ry = make_float4(y.v[0], y.v[1], y.v[2], y.v[3]); }
        } else {
            float gb[4], yb[4], ga[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = t + u < T;
                const int tt = min(t + u, T - 1);
                const float vg = pg[tt], va = pa[tt];
                gb[u] = ok ? vg : 0.f;
                ga[u] = ok ? va : 0.f;
                if constexpr (LOG) { const float vy = py[tt]; yb[u] = ok ? vy : 0.f; } else yb[u] = 0.f;
            }
            rg = make_float4(gb[0], gb[1], gb[2], gb[3]);
            ry = make_float4(yb[0], yb[1], yb[2], yb[3]);
            ra = make_float4(ga[0], ga[1], ga[2], ga[3]);
        }
        if (++nblk == ntc) { nblk = 0; pg += step_g; pa += step_a; if constexpr (LOG) py += step_g; }
    };
    // entries t .. t + 3 of a row of T, of which the request fetched min(t, T - 4) .. : shift left by sh = t - (T - 4), zeros behind
    auto in_place = [&](float4 v, int t) -> float4 {
        const int sh = t - (T - 4);
        if (TINY || sh <= 0) return v;
        if (sh == 1) return make_float4(v.y, v.z, v.w, 0.f);
        if (sh == 2) return make_float4(v.z, v.w, 0.f, 0.f);
        if (sh == 3) return make_float4(v.w, 0.f, 0.f, 0.f);
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto park = [&](int buf, float4 ra, float4 rg, float4 ry, int t) {
        ra = in_place(ra, t); rg = in_place(rg, t);
        float4 rb = rg;
        if constexpr (LOG) {
            ry = in_place(ry, t);                                   // (zeros behind the row: exp(-0) times a zero gradient)
            rb = make_float4(rg.x * expf(-ry.x), rg.y * expf(-ry.y), rg.z * expf(-ry.z), rg.w * expf(-ry.w));     // gm = grad_out * exp(-out)
        }
        if constexpr (!BF16X3) {
            *reinterpret_cast<float4*>(&lds_b[buf][srow * kFbgRow + 4 * sc]) = rb;
            if (has_a) *reinterpret_cast<float4*>(&lds_a[buf][srow * kFbgRow + 4 * sc]) = ra;
        } else {
            // image = [hi rows | lo rows], a row = 24 bf16 = 12 floats; this thread's four time steps are 8 bytes of each
            unsigned short* ib = reinterpret_cast<unsigned short*>(&lds_b[buf][0]);
            unsigned short* ia = reinterpret_cast<unsigned short*>(&lds_a[buf][0]);
            uint2 hi, lo;
            split_bf16x4(rb, hi, lo);
            *reinterpret_cast<uint2*>(ib + srow * kFbgRowH + 4 * sc) = hi;
            *reinterpret_cast<uint2*>(ib + (kFbgBM + srow) * kFbgRowH + 4 * sc) = lo;
            if (has_a) {
                split_bf16x4(ra, hi, lo);
                *reinterpret_cast<uint2*>(ia + srow * kFbgRowH + 4 * sc) = hi;
                *reinterpret_cast<uint2*>(ia + (kFbgBF + 1 + srow) * kFbgRowH + 4 * sc) = lo;
            }
        }
    };
    floatx4_t acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = floatx4_t{0.f, 0.f, 0.f, 0.f};
    floatx16_t acc32;                                               // BF16X3: the wave's 32 x 32 sub-tile as ONE accumulator
    for (int i = 0; i < 16; ++i) acc32[i] = 0.f;
    const int r32 = lane & 31, kg = lane >> 5;                      // BF16X3 operand lane: row / column r32, time steps 8 kg .. 8 kg + 7
    float nyq = 0.f;                                                // fold: this thread's quarter of column srow of the last row
    // 16-row / 16-column pieces of this wave's sub-tile that exist (n_fft / 2 + 1 rows: the last tile of 64 holds ONE): a wave
    // with none only stages
    const bool act_f0 = f0 + wf * 32 < F, act_f1 = f0 + wf * 32 + 16 < F;
    const bool act_m0 = m0 + wm * 32 < M, act_m1 = m0 + wm * 32 + 16 < M;
    {
        // block k waits in ring slot k % DEPTH: blocks 0 .. DEPTH - 1 requested, block 0 parked, block DEPTH requested in its place
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        unrolled<0, DEPTH>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            ring_a[d] = zero; ring_g[d] = zero; ring_y[d] = zero; ring_t[d] = 0;
            fetch(ring_a[d], ring_g[d], ring_y[d], ring_t[d]);
        });
        FSTAMP(1);   // first requests issued
        park(0, ring_a[0], ring_g[0], ring_y[0], ring_t[0]);
        fetch(ring_a[0], ring_g[0], ring_y[0], ring_t[0]);
    }
    __syncthreads();
    FSTAMP(2);       // first block parked
    for (int c0 = 0; c0 < total; c0 += DEPTH) {
        unrolled<0, DEPTH>([&](auto dd) {
            constexpr int d = decltype(dd)::value;            // compile-time ring slot: the buffers stay in registers
            const int c = c0 + d;
            if (c >= total) return;
            const int buf = c & 1;
            float4 a[2], g[2];
            bf16x8_t ah, al, gh, gl;
            if constexpr (!BF16X3) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *reinterpret_cast<const float4*>(&lds_a[buf][(wf * 32 + 16 * i + row) * kFbgRow + 4 * kq]);
                    g[i] = *reinterpret_cast<const float4*>(&lds_b[buf][(wm * 32 + 16 * i + row) * kFbgRow + 4 * kq]);
                }
                if (fold) {
                    const float4 ar = *reinterpret_cast<const float4*>(&lds_a[buf][kFbgBF * kFbgRow + 4 * sc]);
                    const float4 gr = *reinterpret_cast<const float4*>(&lds_b[buf][srow * kFbgRow + 4 * sc]);
                    nyq = fmaf(ar.x, gr.x, nyq); nyq = fmaf(ar.y, gr.y, nyq); nyq = fmaf(ar.z, gr.z, nyq); nyq = fmaf(ar.w, gr.w, nyq);
                }
            } else {
                const unsigned short* ia = reinterpret_cast<const unsigned short*>(&lds_a[buf][0]);
                const unsigned short* ib = reinterpret_cast<const unsigned short*>(&lds_b[buf][0]);
                ah = *reinterpret_cast<const bf16x8_t*>(ia + (wf * 32 + r32) * kFbgRowH + 8 * kg);
                al = *reinterpret_cast<const bf16x8_t*>(ia + (kFbgBF + 1 + wf * 32 + r32) * kFbgRowH + 8 * kg);
                gh = *reinterpret_cast<const bf16x8_t*>(ib + (wm * 32 + r32) * kFbgRowH + 8 * kg);
                gl = *reinterpret_cast<const bf16x8_t*>(ib + (kFbgBM + wm * 32 + r32) * kFbgRowH + 8 * kg);
                if (fold) {
                    // the folded row on the vector pipe: both operands put together again from their halves (hi + lo: 2^-17 of v)
                    const uint2 arh = *reinterpret_cast<const uint2*>(ia + kFbgBF * kFbgRowH + 4 * sc);
                    const uint2 arl = *reinterpret_cast<const uint2*>(ia + (kFbgBF + 1 + kFbgBF) * kFbgRowH + 4 * sc);
                    const uint2 grh = *reinterpret_cast<const uint2*>(ib + srow * kFbgRowH + 4 * sc);
                    const uint2 grl = *reinterpret_cast<const uint2*>(ib + (kFbgBM + srow) * kFbgRowH + 4 * sc);
                    auto wide = [](unsigned h, unsigned l, int odd) {
                        return __uint_as_float(odd ? (h & 0xffff0000u) : (h << 16)) + __uint_as_float(odd ? (l & 0xffff0000u) : (l << 16));
                    };
                    nyq = fmaf(wide(arh.x, arl.x, 0), wide(grh.x, grl.x, 0), nyq); nyq = fmaf(wide(arh.x, arl.x, 1), wide(grh.x, grl.x, 1), nyq);
                    nyq = fmaf(wide(arh.y, arl.y, 0), wide(grh.y, grl.y, 0), nyq); nyq = fmaf(wide(arh.y, arl.y, 1), wide(grh.y, grl.y, 1), nyq);
                }
            }
            // block c + 1 (requested DEPTH iterations ago) into the other image: nobody reads that image before the barrier below;
            // then block c + 1 + DEPTH is requested into the slot just emptied
            constexpr int dn = (d + 1) % DEPTH;               // slot of block c + 1
            if (c + 1 < total) park(buf ^ 1, ring_a[dn], ring_g[dn], ring_y[dn], ring_t[dn]);
            fetch(ring_a[dn], ring_g[dn], ring_y[dn], ring_t[dn]);
            if constexpr (BF16X3) {
                if (act_f0 && act_m0) {
                    // smallest term first; rows / columns past the edge hold copies of the last valid row and are never stored
                    acc32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh, acc32, 0, 0, 0);
                    acc32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl, acc32, 0, 0, 0);
                    acc32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh, acc32, 0, 0, 0);
                }
                __syncthreads();
                return;
            }
            const float av[2][4] = {{a[0].x, a[0].y, a[0].z, a[0].w}, {a[1].x, a[1].y, a[1].z, a[1].w}};
            const float gv[2][4] = {{g[0].x, g[0].y, g[0].z, g[0].w}, {g[1].x, g[1].y, g[1].z, g[1].w}};
            if (act_f1 && act_m1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][u], gv[0][u], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][u], gv[1][u], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][u], gv[0][u], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][u], gv[1][u], acc[1][1], 0, 0, 0);
                }
            } else if (act_f0 && act_m0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][u], gv[0][u], acc[0][0], 0, 0, 0);
                    if (act_m1) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][u], gv[1][u], acc[0][1], 0, 0, 0);
                    if (act_f1) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][u], gv[0][u], acc[1][0], 0, 0, 0);
                }
            }
            __syncthreads();
        });
    }
    FSTAMP(3);       // K loop
    // D[4 * (lane >> 4) + r][lane & 15]: rows = freq, columns = mel
    float* part = p.partials + (size_t)split * F * M;
    if (fold) {
        // the four quarters of a column sit in one quad: (q0 + q1) + (q2 + q3), the same bits in every lane
        const float o1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, nyq), 0xB1, 0xF, 0xF, true));
        const float pr = (sc & 1) ? o1 + nyq : nyq + o1;
        const float o2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, pr), 0x4E, 0xF, 0xF, true));
        const float tot = (sc & 2) ? o2 + pr : pr + o2;
        if (sc == 0 && m0 + srow < M) part[(size_t)(F - 1) * M + m0 + srow] = tot;
    }
    if constexpr (BF16X3) {
        // D[8 (v / 4) + 4 (lane >> 5) + v % 4][lane & 31] of the 32 x 32 tile: rows = freq, columns = mel
        const int m = m0 + wm * 32 + r32;
        if (m < M) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int f = f0 + wf * 32 + 8 * (v >> 2) + 4 * kg + (v & 3);
                if (f < F) part[(size_t)f * M + m] = acc32[v];
            }
        }
        FSTAMP(4);
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 32 + 16 * j + row;
            if (m >= M) continue;
            for (int r = 0; r < 4; ++r) {
                const int f = f0 + wf * 32 + 16 * i + 4 * kq + r;
                if (f < F) part[(size_t)f * M + m] = acc[i][j][r];
            }
        }
    FSTAMP(4);       // partial tile stored (issued)
}

// Eight waves per 64 elements of grad_fb: wave j adds its eighth of the slices in index order (all of its loads in flight at once
// at the usual slice counts), the eighths meet in LDS and are added as ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)): a fixed
// order, full 256-byte rows per request
constexpr int kFbgRedWaves = 8;
__global__ void __launch_bounds__(64 * kFbgRedWaves) dmel_fbgrad_reduce_kernel(FbGradParams p)
{
    __shared__ float part[kFbgRedWaves][64];
    const size_t n = (size_t)p.F * p.M;
    const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    const size_t ii = i < n ? i : n - 1;
    const int q_lo = (int)((long long)p.splits * j / kFbgRedWaves), q_hi = (int)((long long)p.splits * (j + 1) / kFbgRedWaves);
    float s = 0.f;
    int q = q_lo;
    for (; q + 8 <= q_hi; q += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p.partials[(size_t)(q + u) * n + ii];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (q + u < q_hi) ? p.partials[(size_t)(q + u) * n + ii] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];          // (adding 0.f past the end changes nothing)
    }
    part[j][lane] = s;
    __syncthreads();
    if (j == 0 && i < n)
        p.grad_fb[i] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) + ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
}

// row tiles of 64: a single left-over row (n_fft / 2 + 1 of a power-of-two n_fft) is folded into the last full tile
static bool fbgrad_folds(int F) { return F > kFbgBF && F % kFbgBF == 1; }
static int fbgrad_row_tiles(int F) { return fbgrad_folds(F) ? F / kFbgBF : (F + kFbgBF - 1) / kFbgBF; }

int fbgrad_splits(int batch, int F, int M, int T)
{
    // as many slices as keep EVERY workgroup resident at once -- two workgroups of 8 waves per CU -- and never one more: a 513th
    // workgroup on 256 CUs ran alone after the others (26 us instead of 15 at BASELINE config 2)
    static const int slots = [] {
        if (const char* e = std::getenv("DMEL_FBG_TARGET")) return std::atoi(e);                     // (diagnostics)
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        return 2 * cus;
    }();
    const int tiles = fbgrad_row_tiles(F) * ((M + kFbgBM - 1) / kFbgBM);
    const long long blocks = (long long)batch * ((T + 15) / 16);
    long long s = slots / tiles;
    if (s > blocks) s = blocks;
    return s < 1 ? 1 : (int)s;
}

// d lambd in the same launch: the slice count that leaves room for the dot's workgroups (four virtual blocks each, whole rows of
// the grid), or 0 when the launch should not carry it (several column tiles; the dot would take more than an eighth of the slots)
int fbgrad_fuse_dot(int F, int M, int splits, int vblocks, int* dot_wgs)
{
    static_assert(kFbgThreads == 512, "fbgrad_dot_body: two 256-thread virtual blocks per pass");
    if ((M + kFbgBM - 1) / kFbgBM != 1 || vblocks < 1) return 0;
    const int row_tiles = fbgrad_row_tiles(F);
    const int wgs = (vblocks + 3) / 4, rows = (wgs + row_tiles - 1) / row_tiles;
    if (rows * 8 > splits || splits - rows < 1) return 0;
    *dot_wgs = wgs;
    return splits - rows;
}

hipError_t launch_fbgrad(const FbGradParams& p_in, hipStream_t s)
{
    FbGradParams p = p_in;
    p.gm = p.grad_out;                 // gm = grad_out * exp(-out) is formed by the kernel while it stages the operand (p.out != nullptr)
    p.fold_last_row = fbgrad_folds(p.F) ? 1 : 0;
    p.ntc = (p.T + 15) / 16;
    const long long all_blocks = (long long)p.B * p.ntc;
    if (all_blocks > 0x7fffffffLL || p.splits < 1) return hipErrorInvalidValue;
    p.blk_base = (int)(all_blocks / p.splits); p.blk_rem = (int)(all_blocks % p.splits);
    const int row_tiles = fbgrad_row_tiles(p.F);
    const int dot_rows = p.dot_wgs > 0 ? (p.dot_wgs + row_tiles - 1) / row_tiles : 0;      // (the caller made sure grid.z is 1 then)
    const dim3 grid(row_tiles, p.splits + dot_rows, (p.M + kFbgBM - 1) / kFbgBM);
    if (p.bf16x3) {
        if (p.T < 4) {
            if (p.out) hipLaunchKernelGGL((dmel_fbgrad_lds_kernel<true, true, true>), grid, dim3(kFbgThreads), 0, s, p);
            else hipLaunchKernelGGL((dmel_fbgrad_lds_kernel<false, true, true>), grid, dim3(kFbgThreads), 0, s, p);
        } else {
            if (p.out) hipLaunchKernelGGL((dmel_fbgrad_lds_kernel<true, false, true>), grid, dim3(kFbgThreads), 0, s, p);
            else hipLaunchKernelGGL((dmel_fbgrad_lds_kernel<false, false, true>), grid, dim3(kFbgThreads), 0, s, p);
        }
    } else if (p.T < 4) {                             // rows shorter than a 16-byte piece: element-wise requests
        if (p.out) hipLaunchKernelGGL((dmel_fbgrad_lds_kernel<true, true>), grid, dim3(kFbgThreads), 0, s, p);
        else hipLaunchKernelGGL((dmel_fbgrad_lds_kernel<false, true>), grid, dim3(kFbgThreads), 0, s, p);
    } else {
        if (p.out) hipLaunchKernelGGL((dmel_fbgrad_lds_kernel<true, false>), grid, dim3(kFbgThreads), 0, s, p);
        else hipLaunchKernelGGL((dmel_fbgrad_lds_kernel<false, false>), grid, dim3(kFbgThreads), 0, s, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t n = (size_t)p.F * p.M;
    hipLaunchKernelGGL(dmel_fbgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * kFbgRedWaves), 0, s, p);
    return hipGetLastError();
}

}  // namespace dmel

#ifdef DMEL_STAMPS
extern "C" int dmel_debug_read_fstamps(unsigned long long* host, int count)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dmel::g_fstamps), sizeof(unsigned long long) * (size_t)count);
}
#endif
