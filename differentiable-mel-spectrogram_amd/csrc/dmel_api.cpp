// dmel_api.cpp -- host side of libdmel_hip.so: the C ABI of include/dmel.h.
//
// Owns everything that is not a kernel: n_fft derivation (time_frequency.py:39,60-65), the HTK mel
// filterbank of models.py:42-48 and its packing into non-zero MFMA B-fragments, FFT twiddle tables,
// per-plan workspace, argument checking and kernel dispatch.  No torch types, no oracle code.
#include "../../include/dmel.h"
#include "dmel_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

namespace {

thread_local std::string g_err;

dmel_status fail(dmel_status st, const std::string& msg)
{
    g_err = msg;
    return st;
}
}  // namespace

namespace dmel {
dmel_status set_error(dmel_status st, const std::string& msg) { return fail(st, msg); }   // for dmel_comm.cpp
bool mailbox_args(const dmel_mailbox* mb, MailboxArgs* out);                              // dmel_comm.cpp
int mailbox_device(const dmel_mailbox* mb);
void mailbox_attach_count(dmel_mailbox* mb, int delta);
unsigned long long mailbox_peek_error(const dmel_mailbox* mb);
}

namespace {

#define DMEL_HIP(expr)                                                                             \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(DMEL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

// torch.linspace (fp32, CPU): symmetric evaluation around the midpoint
void linspace_f32(float start, float end, int steps, std::vector<float>& out)
{
    out.resize(steps);
    if (steps == 1) { out[0] = start; return; }
    const float step = (end - start) / (float)(steps - 1);
    const int half = steps / 2;
    for (int i = 0; i < steps; ++i)
        out[i] = i < half ? std::fmaf(step, (float)i, start) : std::fmaf(-step, (float)(steps - i - 1), end);   // (torch's CPU kernel fuses the multiply-add: tests/golden/g8_fbanks.npz)
}

struct NfftTables {
    int n_fft = 0, F = 0, KS = 0, NT = 0, groups = 0;
    int n_entries = 0, n_dense = 0;
    float2* tw1 = nullptr;       // twiddles of the plan of the training modes ...
    float2* tw2 = nullptr;
    float2* tw1p = nullptr;      // ... and of the modes that pack two frames per FFT, where that plan differs (else the same pointers)
    float2* tw2p = nullptr;
    float* ent_b = nullptr;
    float* ent_pre = nullptr;
    int pre_groups[16] = {};     // per (wave, run) of group 0: real groups of 4 k-steps inside ent_pre
    unsigned xch_groups = 0;     // FwdParams::xch_groups
    int ent_b_floats = 0;
    int4* tile_ranges = nullptr;
    float* fb_dense = nullptr;   // (F, M) for the direct-DFT kernel
    float2* tw_long = nullptr;   // (N/2) twiddles exp(-2 pi i k / N): LDS radix-2 kernels (long transforms, dL/dx)
    float* fbT = nullptr;        // long transforms: (M, F) transposed bank, (M) bands
    int2* band = nullptr;
    int2* rowband = nullptr;     // (F): non-zero column range of every filterbank row (dL/dx)
    float4* rowpk = nullptr;     // (F): (first two non-zero coefficients, first column, columns) of every row (dL/dx, wave-FFT kernel)
    bool long_rows = false;      // some row has more than two columns
    uint4* ent_h = nullptr;      // caller-supplied (dense) banks, n_fft 64 .. 4096: B fragments of the split-bf16 contraction (FwdParams::ent_h)
    float* fb_nyq = nullptr;     // ... and the row of bin n_fft/2 (n_mels)
    float4* wl_b4 = nullptr;     // kTrainW (wave-local contraction, FwdParams::wl_*): B operands, lane table, phase lengths
    int2* wl_lane = nullptr;
    int* wl_merge = nullptr;      // [phase * 64 + lane]: partner lane of merge round 1 | round 2 << 8 | receives in round 1 << 16 | in round 2 << 17
    int wl_phases = 0, wl_total4 = 0;
    int wl_len4[dmel::kWlMaxPhases] = {};
    int wl_mg[dmel::kWlMaxPhases] = {};   // per phase: bit 0 / 1 = merge round 1 / 2 is needed
    void release()
    {
        if (tw1p == tw1) tw1p = nullptr;
        if (tw2p == tw2) tw2p = nullptr;
        void* ptrs[] = {tw1, tw2, tw1p, tw2p, ent_b, ent_pre, tile_ranges, fb_dense, tw_long, fbT, band, rowband, rowpk, ent_h, fb_nyq, wl_b4, wl_lane, wl_merge};
        for (void* q : ptrs) (void)hipFree(q);
        *this = NfftTables();
    }
};

}  // namespace

struct dmel_plan {
    dmel_config cfg{};
    int device = 0;
    int T = 0;
    int nchunks = 1, chunk = 0;
    double f_max = 0;
    std::map<int, NfftTables> tables;
    std::map<int, std::vector<float>> custom_fb;
    std::map<int, bool> dense_dev;     // n_fft whose tables have the dense structure and get their values from the device
    // Plan-owned scratch of the entry points that take none from the caller (layout: dmel_scratch_bytes).  Calls that use it
    // are ordered across streams by `xev` (see order_after_last_stream): two streams through one plan serialise, never race.
    unsigned char* own_scratch = nullptr;
    int own_scratch_clips = 0;
    // transforms beyond the LDS kernels (dmel_big.hip): chirp / filter tables per DFT length, twiddles per FFT length, the
    // window table and the per-workgroup sequences in global memory
    struct BigTab { int M = 0, logM = 0; float2* chirp = nullptr; float2* hbr = nullptr; };
    std::map<int, float2*> big_wodd;   // per even N: exp(-2 pi i n / N), n < N/2 (split mode of the big kernel)
    std::map<int, BigTab> big_tabs;
    std::map<int, float2*> big_tw;
    float2* big_win = nullptr; size_t big_win_n = 0;
    float2* big_z = nullptr; size_t big_z_n = 0;
    float* fbw = nullptr;          // workspace of dmel_backward_fb (spectrogram (B, F, T) + slice partials) and dmel_backward_x
    size_t fbw_floats = 0;         // (frame gradients (B, T, N)); grown on demand
    hipStream_t last_stream = nullptr;
    bool last_stream_valid = false;
    hipEvent_t xev = nullptr;
    // device-resident lambd (dmel_forward_dev): what the kernels report back and what the host has concluded from it.
    // Every forward that executes draws an execution number from `exec_counter` (device) and reports under it (dmel_kernels.h).
    unsigned long long* host_words = nullptr;   // pinned, device-visible: [0..kLamRing) reports (number << 32) | bits(lambd), [kLamRing] sticky error
    unsigned* exec_counter = nullptr;           // device: executions so far
    unsigned issued = 0;           // the host's estimate of the execution number of its most recent eagerly issued forward (caught up
                                   // with the reports: replays of captured forwards execute without the host counting them)
    unsigned calls = 0;            // dmel_forward_dev / _fixed calls, captured ones included; never adjusted
    bool lam_known = false;        // lam_seen / seq_seen hold an observation
    float lam_seen = 0.f;
    unsigned seq_seen = 0;         // execution number the observation belongs to
    unsigned seq_floor = 0;        // reports with a smaller number predate the last reset
    int n_obs = 0;                 // observations since the last reset
    float lam_rate = 0.f;          // decayed maximum of |d lambd| per execution
    int max_ahead = 8;             // forwards the host may run ahead of the last observation (0: unbounded)
    int guard_mode = 0;            // 0 auto, 1 always both neighbours, 2 never, 3 auto also under capture
    int forced_n_fft = 0, forced_guards = 0;    // dmel_plan_force_launch: the caller chooses the launches (0: the library does)
    int last_guards = 0;           // bit 0: n_fft/2 launched, bit 1: 2 n_fft launched (most recent call)
    int refs = 1;                  // dmel_plan_retain / dmel_plan_release (guarded by g_plans_mu)
    dmel_mailbox* mailbox = nullptr;    // dmel_plan_attach_mailbox: the backward's result is the sum over the mailbox's ranks
    dmel::AdamParams fused_adam{};      // dmel_plan_attach_adam: param != nullptr -> the backward's dot kernel applies this update
    std::mutex mu;
    dmel_plan_info info{};
    // optional event timing
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Span { size_t a, b; int kind; };
    std::vector<Span> spans;
};

namespace {

constexpr int kMaxPartials = 1024;
constexpr size_t kMaxSpans = 16384;

// returns an event index recorded on s, or (size_t)-1 when profiling is off / full
size_t prof_mark(dmel_plan* pl, hipStream_t s)
{
    if (!pl->profiling || pl->spans.size() >= kMaxSpans) return (size_t)-1;
    if (pl->ev_used == pl->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return (size_t)-1;
        pl->ev_pool.push_back(e);
    }
    const size_t i = pl->ev_used++;
    if (hipEventRecord(pl->ev_pool[i], s) != hipSuccess) return (size_t)-1;
    return i;
}
void prof_span(dmel_plan* pl, size_t a, size_t b, int kind)
{
    if (a != (size_t)-1 && b != (size_t)-1) pl->spans.push_back({a, b, kind});
}

// The contraction phase of an 8-wave plan (and of a 4-wave plan with up to four mel tiles per group): wave tl owns tile tl; the
// units (4 k-steps = 16 bins) of a tile beyond a common limit L go, in contiguous pieces, to the run 1 of other waves (one piece per
// wave).  L = the smallest limit for which "largest remainder first onto largest spare first" fits.  own[w]: units wave w takes of its
// own tile; pieces: (helper wave, tile, first unit, units).  Pure host arithmetic (dmel_contraction_partition_host exposes it to tests).
struct Piece { int wave, tile, a, n; };
void deal_ksteps(const int* units, int ntg, int W, int* own, std::vector<Piece>* pieces)
{
    int total = 0, umax = 0;
    for (int tl = 0; tl < ntg; ++tl) { total += units[tl]; umax = std::max(umax, units[tl]); }
    for (int L = std::max(1, (total + W - 1) / W); ; ++L) {
        pieces->clear();
        int spare[8], rem[8];
        bool used[8] = {false, false, false, false, false, false, false, false};
        for (int w = 0; w < 8; ++w) {
            if (w >= W) { own[w] = 0; rem[w] = 0; spare[w] = 0; used[w] = true; continue; }
            own[w] = w < ntg ? std::min(units[w], L) : 0;
            rem[w] = w < ntg ? units[w] - own[w] : 0;
            spare[w] = L - own[w];
        }
        bool ok = true;
        for (;;) {
            int t = -1;
            for (int tl = 0; tl < ntg; ++tl) if (rem[tl] > 0 && (t < 0 || rem[tl] > rem[t])) t = tl;
            if (t < 0) break;
            int h = -1;
            for (int w = 0; w < 8; ++w) if (!used[w] && w != t && spare[w] > 0 && (h < 0 || spare[w] > spare[h])) h = w;
            if (h < 0) { ok = false; break; }
            const int n = std::min(spare[h], rem[t]);
            pieces->push_back({h, t, units[t] - rem[t], n});
            used[h] = true; rem[t] -= n;
        }
        if (ok) return;
        if (L >= umax) {                      // (not reached: at L = umax nothing is left over)
            pieces->clear();
            for (int w = 0; w < 8; ++w) own[w] = w < ntg ? units[w] : 0;
            return;
        }
    }
}

dmel_status build_tables(dmel_plan* pl, int N, NfftTables** out)
{
    auto it = pl->tables.find(N);
    if (it != pl->tables.end()) { *out = &it->second; return DMEL_OK; }
    NfftTables tb;
    struct Guard {          // a failed allocation half-way must not leak the tables made so far
        NfftTables* t; bool keep = false;
        ~Guard() { if (!keep) t->release(); }
    } guard{&tb};
    tb.n_fft = N;
    tb.F = N / 2 + 1;
    const int M = pl->cfg.n_mels;
    std::vector<float> fb((size_t)tb.F * M);
    auto cit = pl->custom_fb.find(N);
    if (pl->dense_dev.count(N)) {
        std::fill(fb.begin(), fb.end(), 1.0f);        // structure only: every block present; dmel_plan_set_filterbank_dev fills the values
    } else if (cit != pl->custom_fb.end()) {
        fb = cit->second;
    } else {
        dmel_status st = dmel_mel_fbanks_host(tb.F, pl->cfg.f_min, pl->f_max, M, pl->cfg.sample_rate, fb.data());
        if (st != DMEL_OK) return st;
    }
    DMEL_HIP(hipMalloc(&tb.fb_dense, fb.size() * sizeof(float)));
    DMEL_HIP(hipMemcpy(tb.fb_dense, fb.data(), fb.size() * sizeof(float), hipMemcpyHostToDevice));

    {
        std::vector<float2> tw((size_t)std::max(1, N / 2));
        for (int k = 0; k < N / 2; ++k) {
            const double th = -2.0 * M_PI * (double)k / (double)N;
            tw[k] = make_float2((float)std::cos(th), (float)std::sin(th));
        }
        DMEL_HIP(hipMalloc(&tb.tw_long, tw.size() * sizeof(float2)));
        DMEL_HIP(hipMemcpy(tb.tw_long, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
        std::vector<int2> rowband(tb.F);
        for (int f = 0; f < tb.F; ++f) {
            int lo = M, hi = 0;
            for (int m = 0; m < M; ++m)
                if (fb[(size_t)f * M + m] != 0.f) { lo = std::min(lo, m); hi = m + 1; }
            rowband[f] = make_int2(hi > lo ? lo : 0, hi > lo ? hi : 0);
        }
        DMEL_HIP(hipMalloc(&tb.rowband, rowband.size() * sizeof(int2)));
        DMEL_HIP(hipMemcpy(tb.rowband, rowband.data(), rowband.size() * sizeof(int2), hipMemcpyHostToDevice));
        std::vector<float4> rowpk(tb.F);
        for (int f = 0; f < tb.F; ++f) {
            const int b0 = rowband[f].x, nb = rowband[f].y - rowband[f].x;
            float bits0, bitsn;
            std::memcpy(&bits0, &b0, 4); std::memcpy(&bitsn, &nb, 4);
            rowpk[f] = make_float4(nb > 0 ? fb[(size_t)f * M + b0] : 0.f, nb > 1 ? fb[(size_t)f * M + b0 + 1] : 0.f, bits0, bitsn);
            if (nb > 2) tb.long_rows = true;
        }
        DMEL_HIP(hipMalloc(&tb.rowpk, rowpk.size() * sizeof(float4)));
        DMEL_HIP(hipMemcpy(tb.rowpk, rowpk.data(), rowpk.size() * sizeof(float4), hipMemcpyHostToDevice));
    }
    const bool is_pow2 = (N & (N - 1)) == 0;
    if (N > dmel::kMaxFastNfft || !is_pow2) {
        std::vector<float> fbT((size_t)M * tb.F);
        std::vector<int2> band(M);
        for (int m = 0; m < M; ++m) {
            int lo = tb.F, hi = 0;
            for (int f = 0; f < tb.F; ++f) {
                const float c = fb[(size_t)f * M + m];
                fbT[(size_t)m * tb.F + f] = c;
                if (c != 0.f) { lo = std::min(lo, f); hi = f + 1; }
            }
            band[m] = make_int2(hi > lo ? lo : 0, hi > lo ? hi : 0);
        }
        DMEL_HIP(hipMalloc(&tb.fbT, fbT.size() * sizeof(float)));
        DMEL_HIP(hipMemcpy(tb.fbT, fbT.data(), fbT.size() * sizeof(float), hipMemcpyHostToDevice));
        DMEL_HIP(hipMalloc(&tb.band, band.size() * sizeof(int2)));
        DMEL_HIP(hipMemcpy(tb.band, band.data(), band.size() * sizeof(int2), hipMemcpyHostToDevice));
    } else if (N >= dmel::kMinFastNfft) {
        // twiddle tables of a plan (R, C): tw1 (R, G) = w_N^(lg q), tw2 (R, C) = w_G^(r p1)
        auto make_tw = [&](bool pair, float2** d1, float2** d2) -> dmel_status {
            int R = 0, C = 0;
            if (!dmel::forward_plan_rc(N, pair, &R, &C)) return fail(DMEL_ERR_UNSUPPORTED, "n_fft has no FFT plan");
            const int G = N / R;
            std::vector<float2> tw1((size_t)R * G), tw2((size_t)R * C);
            for (int q = 0; q < R; ++q)
                for (int lg = 0; lg < G; ++lg) {
                    const double th = -2.0 * M_PI * (double)((long long)lg * q % N) / (double)N;
                    tw1[(size_t)q * G + lg] = make_float2((float)std::cos(th), (float)std::sin(th));
                }
            for (int p1 = 0; p1 < R; ++p1)
                for (int r = 0; r < C; ++r) {
                    const double th = -2.0 * M_PI * (double)(r * p1 % G) / (double)G;
                    tw2[(size_t)p1 * C + r] = make_float2((float)std::cos(th), (float)std::sin(th));
                }
            DMEL_HIP(hipMalloc(d1, tw1.size() * sizeof(float2)));
            DMEL_HIP(hipMemcpy(*d1, tw1.data(), tw1.size() * sizeof(float2), hipMemcpyHostToDevice));
            DMEL_HIP(hipMalloc(d2, tw2.size() * sizeof(float2)));
            DMEL_HIP(hipMemcpy(*d2, tw2.data(), tw2.size() * sizeof(float2), hipMemcpyHostToDevice));
            return DMEL_OK;
        };
        {
            dmel_status stw = make_tw(false, &tb.tw1, &tb.tw2);
            if (stw != DMEL_OK) return stw;
            int r0 = 0, c0 = 0, r1 = 0, c1 = 0;
            dmel::forward_plan_rc(N, false, &r0, &c0); dmel::forward_plan_rc(N, true, &r1, &c1);
            if (r0 == r1 && c0 == c1) { tb.tw1p = tb.tw1; tb.tw2p = tb.tw2; }
            else if ((stw = make_tw(true, &tb.tw1p, &tb.tw2p)) != DMEL_OK) return stw;
        }

        // Non-zero 4 x 16 blocks of the filterbank as MFMA B-fragments.  Mel tiles are handled in
        // groups of 8 (128 mel bands); inside a group a 4-wave plan gives wave w tiles w and 7-w (the HTK bands
        // widen with frequency, so pairing a narrow and a wide tile balances the four waves), an 8-wave plan
        // deals the k-steps themselves (below).
        tb.KS = (tb.F + 3) / 4;
        tb.NT = (M + 15) / 16;
        tb.groups = (tb.NT + 7) / 8;
        tb.n_dense = tb.KS * tb.NT;
        const int waves = dmel::forward_waves(N);
        std::vector<int4> ranges((size_t)tb.groups * waves * 2, make_int4(0, 0, 0, -1));
        std::vector<float> bfr;
        tb.n_entries = 0;
        // run of k-steps [ks_a, ks_b) of mel tile `tile` appended to the B-fragment stream
        auto emit = [&](int tile, int ks_a, int ks_b) {
            int4 tr = make_int4(ks_a, ks_b - ks_a, (int)bfr.size(), tile);
            for (int ks = ks_a; ks < ks_b; ++ks)
                for (int l = 0; l < 64; ++l) {
                    const int f = 4 * ks + (l >> 4), m = 16 * tile + (l & 15);
                    bfr.push_back((f < tb.F && m < M) ? fb[(size_t)f * M + m] : 0.f);
                }
            return tr;
        };
        for (int g = 0; g < tb.groups; ++g) {
            const int ntg = std::min(8, tb.NT - 8 * g);
            // the band of a mel tile is one contiguous run of k-steps: [first, first + nks) in units of 4 k-steps (16 bins)
            int first_of[8], units[8];
            for (int tl = 0; tl < ntg; ++tl) {
                const int tile = 8 * g + tl;
                int first = tb.KS, last = -1;
                for (int ks = 0; ks < tb.KS; ++ks)
                    for (int l = 0; l < 64; ++l) {
                        const int f = 4 * ks + (l >> 4), m = 16 * tile + (l & 15);
                        if (f < tb.F && m < M && fb[(size_t)f * M + m] != 0.f) { first = std::min(first, ks); last = std::max(last, ks); }
                    }
                if (last < first) { first_of[tl] = 0; units[tl] = 0; continue; }      // an all-zero tile still needs its outputs written
                tb.n_entries += last - first + 1;
                first = first / 4 * 4;                       // groups of 4 k-steps start at multiples of 16 bins
                first_of[tl] = first; units[tl] = (last - first + 1 + 3) / 4;
            }
            if (waves == 8 || (ntg <= waves && g < 32)) {         // (FwdParams::xch_groups is a 32-bit mask: groups beyond it keep whole tiles -- ADVICE r04)
                // (4-wave plans, n_fft <= 512: only when every tile has a wave of its own -- up to 64 mels, the reference's experiments --
                // since run 1 is then free to carry a piece; with more tiles wave w owns tiles w and 7 - w whole, as before)
                // Wave tl owns tile tl (run 0: it writes the tile's outputs).  The HTK bands widen with frequency -- at 128 mels the last
                // tile has five times the k-steps of the first, at 64 mels four tiles meet eight waves -- so the k-steps of the wide
                // tiles beyond a common limit L are dealt, in contiguous pieces, to the run 1 of waves with narrow or no tiles of their
                // own (one piece per wave; their sums reach the owner through LDS, dmel_fwd_kernel's exchange).  L = the smallest
                // limit for which largest-remainder-first onto largest-spare-first fits.  (Round 3 split every tile in two halves,
                // wave tl and 7 - tl: the slowest wave carried 24 k-steps at BASELINE config 2 where 19.5 is the mean, 56 / 36.5 at
                // config 5, 120 / 68 at the reference's ESC-50 shape at n_fft 4096.)
                const int W = waves;
                int own[8];
                std::vector<Piece> pieces;
                deal_ksteps(units, ntg, W, own, &pieces);
                int mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (const Piece& pc : pieces) mask[pc.tile] |= 1 << pc.wave;
                for (int tl = 0; tl < ntg; ++tl) {
                    int4 tr = units[tl] > 0 ? emit(8 * g + tl, first_of[tl], first_of[tl] + 4 * own[tl]) : make_int4(0, 0, 0, 8 * g + tl);
                    tr.w |= mask[tl] << 16;
                    ranges[(size_t)(g * W + tl) * 2 + 0] = tr;
                }
                for (const Piece& pc : pieces) {
                    int4 tr = emit(8 * g + pc.tile, first_of[pc.tile] + 4 * pc.a, first_of[pc.tile] + 4 * (pc.a + pc.n));
                    tr.w |= 1 << 30;                                   // a piece: summed into its owner's tile, not written by this wave
                    ranges[(size_t)(g * W + pc.wave) * 2 + 1] = tr;
                }
                if (W != 8 && !pieces.empty()) tb.xch_groups |= 1u << g;       // g < 32 here
            } else {
                // 4 waves: wave w owns tiles w and 7-w whole
                for (int tl = 0; tl < ntg; ++tl) {
                    const int w = tl < 4 ? tl : 7 - tl, loc = tl < 4 ? 0 : 1;
                    ranges[(size_t)(g * 4 + w) * 2 + loc] = units[tl] > 0 ? emit(8 * g + tl, first_of[tl], first_of[tl] + 4 * units[tl]) : make_int4(0, 0, 0, 8 * g + tl);
                }
            }
        }
        tb.ent_b_floats = (int)bfr.size();
        if ((cit != pl->custom_fb.end() || pl->dense_dev.count(N)) && dmel::forward_has_hsplit(N)) {
            // the same matrix as B fragments of v_mfma_f32_16x16x32_bf16, split into two bf16 halves (DMEL_FLAG_MFMA_BF16X3):
            // [(tile * (N/64) + kstep) * 2 + (hi | lo)][lane][8]: lane l holds rows 32 kstep + 8 (l >> 4) + e of column 16 tile + (l & 15)
            const int ks32 = N / 64;
            auto bf16_rne = [](float v) -> unsigned short {
                unsigned u; std::memcpy(&u, &v, 4);
                if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);      // NaN stays NaN
                return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            };
            std::vector<unsigned short> eh((size_t)tb.NT * ks32 * 2 * 64 * 8);
            for (int tile = 0; tile < tb.NT; ++tile)
                for (int ks = 0; ks < ks32; ++ks)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int f = 32 * ks + 8 * (l >> 4) + e, m = 16 * tile + (l & 15);
                            const float v = (m < M) ? fb[(size_t)f * M + m] : 0.f;
                            const unsigned short hi = bf16_rne(v);
                            unsigned hu = (unsigned)hi << 16; float hf; std::memcpy(&hf, &hu, 4);
                            const unsigned short lo = bf16_rne(v - hf);
                            const size_t base = ((size_t)(tile * ks32 + ks) * 2) * 64 * 8;
                            eh[base + (size_t)l * 8 + e] = hi;
                            eh[base + 64 * 8 + (size_t)l * 8 + e] = lo;
                        }
            DMEL_HIP(hipMalloc(&tb.ent_h, eh.size() * sizeof(unsigned short)));
            DMEL_HIP(hipMemcpy(tb.ent_h, eh.data(), eh.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
            DMEL_HIP(hipMalloc(&tb.fb_nyq, (size_t)M * sizeof(float)));
            DMEL_HIP(hipMemcpy(tb.fb_nyq, fb.data() + (size_t)(tb.F - 1) * M, (size_t)M * sizeof(float), hipMemcpyHostToDevice));
        }
        if (dmel::forward_has_wlc(N) && (M + 3) / 4 <= 16 * dmel::kWlMaxPhases) {
            // kTrainW: the schedule of the wave-local contraction (FwdParams::wl_*).  A quad = 4 consecutive mel bands; its band =
            // the bins where any of them is non-zero.  A block of the 4x4x1 MFMA walks one PIECE of a band, one bin per step; a phase
            // holds 16 pieces and is as long as its longest, rounded up to 4 steps.  Whole quads are one piece.  Where a wave holds ONE
            // frame (n_fft >= 2048: two of the four rows idle, phases 100-330 steps long) a wide quad is split over 2 or 4 blocks of one
            // phase -- pieces of its band, B operands restricted to the piece -- and the partial sums are merged across lanes after the
            // phase's loop (wl_merge: one or two rounds of ds_bpermute + add; the final piece's lanes carry the mel band).
            const int Q = (M + 3) / 4;
            struct Piece { int quad, lo, w, group, idx, s; };
            std::vector<int> first(Q, 0), width(Q, 0);
            for (int q = 0; q < Q; ++q) {
                int lo = tb.F, hi = -1;
                for (int f = 0; f < tb.F; ++f)
                    for (int j = 0; j < 4 && 4 * q + j < M; ++j)
                        if (fb[(size_t)f * M + 4 * q + j] != 0.f) { lo = std::min(lo, f); hi = std::max(hi, f); }
                first[q] = hi >= lo ? lo : 0; width[q] = hi >= lo ? hi - lo + 1 : 0;
            }
            const bool may_split = dmel::forward_wlc_one_frame(N) && !pl->dense_dev.count(N) && std::getenv("DMEL_WLC_NOSPLIT") == nullptr;
            // split factors: pieces no longer than w*, s in {1, 2, 4}; w* by exhaustive search over the step count of the resulting phases
            auto make_groups = [&](int wstar, std::vector<std::vector<Piece>>& phases) -> int {
                struct Grp { int quad, s, len; };
                std::vector<Grp> gs;
                for (int q = 0; q < Q; ++q) {
                    int sp = 1;
                    while (may_split && (width[q] + sp - 1) / sp > wstar && sp < 4) sp *= 2;
                    gs.push_back({q, sp, (width[q] + sp - 1) / sp});
                }
                std::stable_sort(gs.begin(), gs.end(), [](const Grp& a, const Grp& b) { return a.len > b.len; });
                phases.clear();
                std::vector<int> used;
                int gid = 0;
                for (const Grp& g : gs) {
                    size_t ph = 0;
                    while (ph < phases.size() && used[ph] + g.s > 16) ++ph;
                    if (ph == phases.size()) { phases.emplace_back(); used.push_back(0); }
                    for (int i = 0; i < g.s; ++i) {
                        const int lo = first[g.quad] + i * g.len;
                        const int w = std::max(0, std::min(g.len, first[g.quad] + width[g.quad] - lo));
                        phases[ph].push_back({g.quad, lo, w, gid, i, g.s});
                    }
                    used[ph] += g.s; ++gid;
                }
                if ((int)phases.size() > dmel::kWlMaxPhases) return 1 << 30;
                // cost in steps: the phases' lengths plus 40 per phase -- measured, not derived: config 3's three phases of 100 steps in all
                // run as long as its two phases of 136 (45.9-46.2 against 45.5-46.1 us, 40 more vector instructions per wave: kept whole),
                // config 5's three of 104 beat its two of 160 by 4 % (66.1-67.9 against 68.9-71.0 us); and -- decisively -- the staged epilogue
                // is lost beyond four phases (config 3 in five phases: 53.0 us)
                int total = 0;
                for (auto& ph : phases) { int wm = 0; for (auto& pc : ph) wm = std::max(wm, pc.w); total += std::max(4, (wm + 3) / 4 * 4) + 40 + (ph.size() && ph[0].s > 1 ? 2 : 0); }
                if ((int)phases.size() > 4) total += 1000;
                return total;
            };
            std::vector<std::vector<Piece>> phases;
            {
                int wmaxq = 4;
                for (int q = 0; q < Q; ++q) wmaxq = std::max(wmaxq, width[q]);
                int best = 1 << 30, best_w = wmaxq;
                if (may_split)
                    for (int wstar = 4; wstar <= wmaxq; ++wstar) {
                        std::vector<std::vector<Piece>> trial;
                        const int t = make_groups(wstar, trial);
                        if (t < best) { best = t; best_w = wstar; }
                    }
                make_groups(best_w, phases);
                // (narrowest phase first, as before: the lane table and the first B groups of phase 0 are the ones requested early)
                std::reverse(phases.begin(), phases.end());
            }
            tb.wl_phases = (int)phases.size();
            std::vector<int2> lanes((size_t)tb.wl_phases * 64);
            std::vector<int> merge((size_t)tb.wl_phases * 64, 0);
            std::vector<float4> b4;
            tb.wl_total4 = 0;
            for (int ph = 0; ph < tb.wl_phases; ++ph) {
                const std::vector<Piece>& pcs = phases[ph];
                const int nq = (int)pcs.size();
                int wmax = 0;
                for (int i = 0; i < nq; ++i) wmax = std::max(wmax, pcs[i].w);
                // Bank conflicts of the A operand reads (dmel_kernels.h, slot_stride_f2): the 8 blocks of each 32-lane group must sit
                // at bins that differ modulo 8.  A block may start anywhere in [lo, hi] -- the piece must fit into the phase's L steps
                // and the run must not pass bin F - 1 -- so each piece is matched to one of the 16 (group, residue) places it can
                // reach (augmenting paths; 16 x 16); if that fails the phase gets four more steps.
                int L = std::max(4, (wmax + 3) / 4 * 4), slot_of[16], k0_of[16];
                for (;; L += 4) {
                    int lo[16], hi[16], owner[16];
                    for (int i = 0; i < nq; ++i) {
                        hi[i] = std::max(0, std::min(pcs[i].lo, tb.F - L));
                        lo[i] = std::min(hi[i], std::max(0, pcs[i].lo + pcs[i].w - L));
                    }
                    auto can = [&](int i, int r) { for (int k = hi[i]; k >= lo[i] && k > hi[i] - 8; --k) if ((k & 7) == r) return true; return false; };
                    std::fill(owner, owner + 16, -1);
                    bool seen[16];
                    std::function<bool(int)> place = [&](int i) -> bool {
                        for (int sl = 0; sl < 16; ++sl) {
                            if (seen[sl] || !can(i, sl & 7)) continue;
                            seen[sl] = true;
                            if (owner[sl] < 0 || place(owner[sl])) { owner[sl] = i; return true; }
                        }
                        return false;
                    };
                    bool ok = true;
                    for (int i = 0; i < nq && ok; ++i) {               // longest (least freedom) first
                        std::fill(seen, seen + 16, false);
                        ok = place(i);
                    }
                    if (!ok && L < wmax + 64) continue;
                    std::fill(slot_of, slot_of + 16, -1);
                    if (ok) { for (int sl = 0; sl < 16; ++sl) if (owner[sl] >= 0) slot_of[owner[sl]] = sl; }
                    else {                                        // (not reached with 16 places for 16 pieces: any residue has two)
                        bool used[16] = {};
                        for (int i = 0; i < nq; ++i) for (int sl = 0; sl < 16; ++sl) if (!used[sl]) { used[sl] = true; slot_of[i] = sl; break; }
                    }
                    for (int i = 0; i < nq; ++i) {
                        k0_of[i] = hi[i];
                        for (int k = hi[i]; k >= lo[i] && k > hi[i] - 8; --k) if ((k & 7) == (slot_of[i] & 7)) { k0_of[i] = k; break; }
                    }
                    break;
                }
                const int n4 = wmax > 0 ? L / 4 : 0;
                tb.wl_len4[ph] = n4;
                const size_t base = b4.size();
                b4.resize(base + (size_t)n4 * 64, make_float4(0.f, 0.f, 0.f, 0.f));
                for (int sl = 0; sl < 16; ++sl)                  // places nobody took: no mel band, a bin of their residue
                    for (int j = 0; j < 4; ++j) { lanes[(size_t)ph * 64 + 4 * sl + j] = make_int2(8 * (sl & 7), -1); merge[(size_t)ph * 64 + 4 * sl + j] = (4 * sl + j) | ((4 * sl + j) << 8); }
                int mg = 0;
                for (int i = 0; i < nq; ++i) {
                    const Piece& pc = pcs[i];
                    const int b = slot_of[i], k0 = k0_of[i];
                    // the pieces of this piece's group in this phase (they were placed together), in the order of their index
                    int blk[4] = {b, b, b, b};
                    for (int i2 = 0; i2 < nq; ++i2) if (pcs[i2].group == pc.group) blk[pcs[i2].idx] = slot_of[i2];
                    // merge tree: round 1: piece 0 += piece 1, piece 2 += piece 3; round 2: piece 0 += piece 2; the result sits with piece 0
                    int p1 = b, p2 = b, f1 = 0, f2 = 0;
                    if (pc.s >= 2 && pc.idx == 0) { p1 = blk[1]; f1 = 1; }
                    if (pc.s == 4 && pc.idx == 2) { p1 = blk[3]; f1 = 1; }
                    if (pc.s == 4 && pc.idx == 0) { p2 = blk[2]; f2 = 1; }
                    if (pc.s >= 2) mg |= 1;
                    if (pc.s == 4) mg |= 2;
                    const bool fin = pc.idx == 0;
                    for (int j = 0; j < 4; ++j) {
                        const int m = 4 * pc.quad + j;
                        lanes[(size_t)ph * 64 + 4 * b + j] = make_int2(8 * k0, (fin && m < M) ? m : -1);
                        merge[(size_t)ph * 64 + 4 * b + j] = (4 * p1 + j) | ((4 * p2 + j) << 8) | (f1 << 16) | (f2 << 17);
                        if (m >= M) continue;
                        for (int st = 0; st < 4 * n4; ++st) {
                            const int f = k0 + st;
                            const float v = (f < tb.F && f >= pc.lo && f < pc.lo + pc.w) ? fb[(size_t)f * M + m] : 0.f;
                            float4& e = b4[base + (size_t)(st / 4) * 64 + 4 * b + j];
                            (st % 4 == 0 ? e.x : st % 4 == 1 ? e.y : st % 4 == 2 ? e.z : e.w) = v;
                        }
                    }
                }
                tb.wl_mg[ph] = mg;
                tb.wl_total4 += n4;
            }
            DMEL_HIP(hipMalloc(&tb.wl_lane, lanes.size() * sizeof(int2)));
            DMEL_HIP(hipMemcpy(tb.wl_lane, lanes.data(), lanes.size() * sizeof(int2), hipMemcpyHostToDevice));
            DMEL_HIP(hipMalloc(&tb.wl_merge, merge.size() * sizeof(int)));
            DMEL_HIP(hipMemcpy(tb.wl_merge, merge.data(), merge.size() * sizeof(int), hipMemcpyHostToDevice));
            DMEL_HIP(hipMalloc(&tb.wl_b4, std::max<size_t>(b4.size(), 64) * sizeof(float4)));
            if (!b4.empty()) DMEL_HIP(hipMemcpy(tb.wl_b4, b4.data(), b4.size() * sizeof(float4), hipMemcpyHostToDevice));
        }
        {   // register-resident prefix of every run of group 0, at a fixed position per (wave, run)
            const int nbpre = dmel::forward_nbpre(N);
            std::vector<float> pre((size_t)waves * 2 * nbpre * 64, 0.f);
            for (int w = 0; w < waves; ++w)
                for (int loc = 0; loc < 2; ++loc) {
                    const int4 tr = ranges[(size_t)w * 2 + loc];
                    const int n = std::min(tr.y, nbpre);
                    tb.pre_groups[w * 2 + loc] = (n + 3) / 4;
                    if (n > 0)
                        std::copy(bfr.begin() + tr.z, bfr.begin() + tr.z + (size_t)n * 64, pre.begin() + ((size_t)(w * 2 + loc) * nbpre) * 64);
                }
            DMEL_HIP(hipMalloc(&tb.ent_pre, pre.size() * sizeof(float)));
            DMEL_HIP(hipMemcpy(tb.ent_pre, pre.data(), pre.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        const size_t nb = std::max<size_t>(bfr.size(), 64);
        DMEL_HIP(hipMalloc(&tb.ent_b, nb * sizeof(float)));
        if (!bfr.empty())
            DMEL_HIP(hipMemcpy(tb.ent_b, bfr.data(), bfr.size() * sizeof(float), hipMemcpyHostToDevice));
        DMEL_HIP(hipMalloc(&tb.tile_ranges, ranges.size() * sizeof(int4)));
        DMEL_HIP(hipMemcpy(tb.tile_ranges, ranges.data(), ranges.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    auto ins = pl->tables.emplace(N, tb);
    guard.keep = true;
    *out = &ins.first->second;
    return DMEL_OK;
}

// ---- per-call device scratch -----------------------------------------------------------------------------------------
// [0] handled word of the launch group | [64] 1024 fp64 partials of the dot kernel | [8256] its ticket counter |
// [8320] window table (w, w d^2 2^-2e) for the kernels that read it from memory | [kScratchPsum] partial clip sums
constexpr size_t kScratchPartials = 64, kScratchCounter = 64 + 8192, kScratchWin = 8320;
constexpr size_t kScratchPsum = kScratchWin + (size_t)dmel::kMaxNfft * 8;
// the dot kernel's group counters live in the upper half of the partials array, found from the ticket word (dmel_kernels.h: dot_group_counters)
static_assert(kScratchCounter - kScratchPartials == 1024 * sizeof(double) && dmel::kDotMaxBlocks * sizeof(double) == 4096 &&
              (dmel::kDotMaxBlocks / dmel::kDotGroup) * 64 <= 4096, "scratch layout the ticket tree relies on");

struct Scratch {
    unsigned* handled = nullptr; double* partials = nullptr; unsigned* counter = nullptr; float2* win = nullptr; float* psum = nullptr;
};

size_t scratch_bytes(const dmel_plan* pl, int batch)
{
    const size_t ps = ((size_t)std::max(batch, 1) * pl->nchunks * sizeof(float) + 255) / 256 * 256;
    return kScratchPsum + ps;
}

Scratch carve(void* base)
{
    unsigned char* b = static_cast<unsigned char*>(base);
    Scratch sc;
    sc.handled = reinterpret_cast<unsigned*>(b);
    sc.partials = reinterpret_cast<double*>(b + kScratchPartials);
    sc.counter = reinterpret_cast<unsigned*>(b + kScratchCounter);
    sc.win = reinterpret_cast<float2*>(b + kScratchWin);
    sc.psum = reinterpret_cast<float*>(b + kScratchPsum);
    return sc;
}

bool is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

// Plan-owned buffers (own_scratch, fbw) are shared by every call on the plan: a call on another stream than the previous
// one first waits for that stream's work (one event), so concurrent streams serialise on the plan instead of racing.
dmel_status order_after_last_stream(dmel_plan* pl, hipStream_t s)
{
    if (pl->last_stream_valid && pl->last_stream != s && !is_capturing(s)) {
        if (hipEventRecord(pl->xev, pl->last_stream) == hipSuccess) DMEL_HIP(hipStreamWaitEvent(s, pl->xev, 0));
        else { (void)hipGetLastError(); DMEL_HIP(hipDeviceSynchronize()); }      // the old stream is gone: be safe
    }
    pl->last_stream = s;
    pl->last_stream_valid = true;
    return DMEL_OK;
}

dmel_status ensure_own_scratch(dmel_plan* pl, int batch, hipStream_t s, Scratch* sc)
{
    if (!pl->own_scratch || batch > pl->own_scratch_clips) {
        if (is_capturing(s)) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan scratch must grow but the stream is capturing: run one call eagerly first or pass scratch");
        if (pl->own_scratch) { DMEL_HIP(hipDeviceSynchronize()); (void)hipFree(pl->own_scratch); pl->own_scratch = nullptr; pl->own_scratch_clips = 0; }
        const int clips = std::max(batch, pl->cfg.max_batch);
        const size_t bytes = scratch_bytes(pl, clips);
        DMEL_HIP(hipMalloc(&pl->own_scratch, bytes));
        DMEL_HIP(hipMemset(pl->own_scratch, 0, kScratchWin));
        pl->own_scratch_clips = clips;
    }
    *sc = carve(pl->own_scratch);
    return DMEL_OK;
}

// in-place radix-2 FFT in double precision (host: the transform of Bluestein's chirp filter, once per DFT length)
void host_fft(std::vector<double>& re, std::vector<double>& im)
{
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t s0 = 0; s0 < n; s0 += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double wr = std::cos(ang * (double)k), wi = std::sin(ang * (double)k);
                const size_t a = s0 + k, b = a + len / 2;
                const double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] += tr; im[a] += ti;
            }
    }
}

// tables of the big path for DFT length N: twiddles of the power-of-two FFT behind it and, when N is not a power of two,
// Bluestein's chirp c[n] = exp(-i pi n^2 / N) and the transform of its filter
dmel_status big_tables_for(dmel_plan* pl, int N, dmel_plan::BigTab* out, const float2** tw)
{
    const bool pow2 = (N & (N - 1)) == 0;
    int M = N;
    if (!pow2) { M = 1; while (M < 2 * N - 1) M <<= 1; }
    if (M > dmel::kMaxBigFft)
        return fail(DMEL_ERR_UNSUPPORTED, "transform length " + std::to_string(N) + " needs an FFT of " + std::to_string(M) +
                    " points; the HIP path stops at " + std::to_string(dmel::kMaxBigFft));
    auto tit = pl->big_tw.find(M);
    if (tit == pl->big_tw.end()) {
        std::vector<float2> t((size_t)std::max(1, M / 2));
        for (int k = 0; k < M / 2; ++k) {
            const double th = -2.0 * M_PI * (double)k / (double)M;
            t[k] = make_float2((float)std::cos(th), (float)std::sin(th));
        }
        float2* d = nullptr;
        DMEL_HIP(hipMalloc(&d, t.size() * sizeof(float2)));
        DMEL_HIP(hipMemcpy(d, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice));
        tit = pl->big_tw.emplace(M, d).first;
    }
    *tw = tit->second;
    auto bit = pl->big_tabs.find(N);
    if (bit == pl->big_tabs.end()) {
        dmel_plan::BigTab bt;
        bt.M = M; bt.logM = 0; while ((1 << bt.logM) < M) ++bt.logM;
        if (!pow2) {
            std::vector<float2> chirp(N);
            std::vector<double> hr(M, 0.0), hi(M, 0.0);
            for (int k = 0; k < N; ++k) {
                const long long q = ((long long)k * k) % (2LL * N);        // k^2 mod 2N: the phase argument stays small
                const double a = M_PI * (double)q / (double)N;
                chirp[k] = make_float2((float)std::cos(a), (float)(-std::sin(a)));
                hr[k] = std::cos(a); hi[k] = std::sin(a);
                if (k) { hr[M - k] = std::cos(a); hi[M - k] = std::sin(a); }
            }
            host_fft(hr, hi);
            std::vector<float2> hbr(M);
            for (int j = 0; j < M; ++j) {
                unsigned r = 0;
                for (int bbit = 0; bbit < bt.logM; ++bbit) r |= ((unsigned)(j >> bbit) & 1u) << (bt.logM - 1 - bbit);
                hbr[r] = make_float2((float)(hr[j] / M), (float)(hi[j] / M));     // position r of a DIF output holds bin brev(r) = j
            }
            DMEL_HIP(hipMalloc(&bt.chirp, chirp.size() * sizeof(float2)));
            DMEL_HIP(hipMemcpy(bt.chirp, chirp.data(), chirp.size() * sizeof(float2), hipMemcpyHostToDevice));
            DMEL_HIP(hipMalloc(&bt.hbr, hbr.size() * sizeof(float2)));
            DMEL_HIP(hipMemcpy(bt.hbr, hbr.data(), hbr.size() * sizeof(float2), hipMemcpyHostToDevice));
        }
        bit = pl->big_tabs.emplace(N, bt).first;
    }
    *out = bit->second;
    return DMEL_OK;
}

// One launch (plus, when needed, the partial-sum / window-table kernel in front of it) of the forward for a given n_fft.
// `lam` says where lambd comes from and whether the kernels check it against N (dmel_kernels.h); `sc` is the scratch of
// this call.  Shared by every entry point; the plan mutex is held by the caller.
dmel_status launch_forward_n(dmel_plan* pl, const float* x, int batch, int N, dmel::LamArgs lam, unsigned flags, double eps,
                             float* out, float* tangent, int mode, int remove_dc, const Scratch& sc, hipStream_t s, int win_half,
                             bool* sums_done, float* spec_out = nullptr)
{
    if (N < 1) return fail(DMEL_ERR_UNSUPPORTED, "n_fft = " + std::to_string(N));
    const bool pow2 = (N & (N - 1)) == 0;
    const bool big = !pow2 || N > dmel::kMaxNfft;
    if (spec_out && (big || N < dmel::kMinFastNfft || mode != dmel::kTrain))
        return fail(DMEL_ERR_UNSUPPORTED, "the spectrogram is saved by the fused training kernel only (power-of-two n_fft from 32 to 16384, tangent requested)");
    if (big && (N & 1)) return fail(DMEL_ERR_UNSUPPORTED, "n_fft = " + std::to_string(N) + " is odd");
    // DMEL_FLAG_X_INDIRECT: every kernel of the forward reads the batch's address from a pointer cell (round 5: the fused kernel and the
    // partial sums only; a lambd that left n_fft 32 ... 16384 inside a captured loop then ended in NaN + an error -- ADVICE r05)
    const float* const* x_cell = (flags & DMEL_FLAG_X_INDIRECT) ? reinterpret_cast<const float* const*>(x) : nullptr;
    NfftTables* tb = nullptr;
    dmel_status st = build_tables(pl, N, &tb);
    if (st != DMEL_OK) return st;
    // the whole-clip window of the optimized=False branches is centred at L/2 as a real number inside its L = n_fft/2 entries
    // (time_frequency.py:24), which torch.stft places (n_fft - L) / 2 = L/2 (integer) entries into the frame
    const float center = win_half ? (float)((N / 2) / 2) + (float)(N / 2) / 2.0f : (float)N / 2.0f;
    if (big) {
        // ---- lengths the LDS kernels do not reach: global-memory FFT / Bluestein (dmel_big.hip) ------------------------
        dmel_plan::BigTab bt;
        const float2* tw = nullptr;
        if ((st = big_tables_for(pl, N, &bt, &tw)) != DMEL_OK) return st;
        // split mode (dmel_big.hip): the transform itself would live in global memory, its two halves fit LDS
        const float2* wodd = nullptr;
        bool split = false;
        if (dmel::big_uses_global(bt.M)) {
            const int H = N / 2;
            int Mh = H;
            if (H & (H - 1)) { Mh = 1; while (Mh < 2 * H - 1) Mh <<= 1; }
            if (H >= 2 && dmel::big_can_split(Mh, pl->cfg.n_mels)) {
                if ((st = big_tables_for(pl, H, &bt, &tw)) != DMEL_OK) return st;
                auto wit = pl->big_wodd.find(N);
                if (wit == pl->big_wodd.end()) {
                    std::vector<float2> w((size_t)H);
                    for (int n = 0; n < H; ++n) {
                        const double th = -2.0 * M_PI * (double)n / (double)N;
                        w[n] = make_float2((float)std::cos(th), (float)std::sin(th));
                    }
                    float2* d = nullptr;
                    DMEL_HIP(hipMalloc(&d, w.size() * sizeof(float2)));
                    DMEL_HIP(hipMemcpy(d, w.data(), w.size() * sizeof(float2), hipMemcpyHostToDevice));
                    wit = pl->big_wodd.emplace(N, d).first;
                }
                wodd = wit->second;
                split = true;
            }
        }
        if ((st = order_after_last_stream(pl, s)) != DMEL_OK) return st;          // window table and sequences are plan-owned
        const bool pair = (mode == dmel::kInfer || mode == dmel::kSpec);
        const long long units = (long long)batch * (pair ? (pl->T + 1) / 2 : pl->T);
        const int grid = dmel::big_grid(units, bt.M);
        const size_t need_z = dmel::big_uses_global(bt.M) ? (size_t)grid * bt.M : 0;
        if ((size_t)N > pl->big_win_n || need_z > pl->big_z_n) {
            if (is_capturing(s)) return fail(DMEL_ERR_INVALID_ARGUMENT, "workspace must grow but the stream is capturing: run one call eagerly first");
            DMEL_HIP(hipDeviceSynchronize());
            if ((size_t)N > pl->big_win_n) {
                (void)hipFree(pl->big_win); pl->big_win = nullptr; pl->big_win_n = 0;
                DMEL_HIP(hipMalloc(&pl->big_win, (size_t)N * sizeof(float2)));
                pl->big_win_n = (size_t)N;
            }
            if (need_z > pl->big_z_n) {
                (void)hipFree(pl->big_z); pl->big_z = nullptr; pl->big_z_n = 0;
                DMEL_HIP(hipMalloc(&pl->big_z, need_z * sizeof(float2)));
                pl->big_z_n = need_z;
            }
        }
        const bool need_sums = remove_dc && !*sums_done;
        const size_t m0 = prof_mark(pl, s);
        {
            dmel::PrepParams pp{};
            pp.x = x_cell ? nullptr : x; pp.x_ind = x_cell; pp.psum = sc.psum; pp.win2 = pl->big_win;
            pp.B = need_sums ? batch : 0; pp.L = pl->cfg.n_points; pp.nchunks = pl->nchunks; pp.chunk = pl->chunk;
            pp.N = N; pp.normalize = pl->cfg.normalize_window; pp.win_half = win_half; pp.center = center;
            pp.lam = lam; pp.lam.role = dmel::kLamQuiet;
            DMEL_HIP(dmel::launch_prep(pp, s));
            if (need_sums) *sums_done = true;
        }
        const size_t m1 = prof_mark(pl, s);
        prof_span(pl, m0, m1, 0);
        dmel::BigParams bp{};
        bp.x = x_cell ? nullptr : x; bp.x_ind = x_cell; bp.out = out; bp.tangent = tangent; bp.psum = sc.psum; bp.win2 = pl->big_win;
        bp.tw = tw; bp.chirp = bt.chirp; bp.hbr = bt.hbr; bp.zws = pl->big_z; bp.fbT = tb->fbT; bp.band = tb->band;
        bp.B = batch; bp.L = pl->cfg.n_points; bp.T = pl->T; bp.hop = pl->cfg.hop_length; bp.M = pl->cfg.n_mels;
        bp.nchunks = pl->nchunks; bp.N = N; bp.F = tb->F; bp.mode = mode; bp.Mfft = bt.M; bp.logM = bt.logM;
        bp.inv_L = 1.0f / (float)pl->cfg.n_points; bp.eps = (float)eps; bp.flags = flags; bp.remove_dc = remove_dc; bp.lam = lam;
        bp.split = split ? 1 : 0; bp.wodd = wodd;
        DMEL_HIP(dmel::launch_big(bp, s));
        prof_span(pl, m1, prof_mark(pl, s), 1);
        pl->info.n_fft = N; pl->info.n_freqs = tb->F; pl->info.n_time = pl->T;
        pl->info.kernel_path = 3; pl->info.frames_per_tile = pair ? 2 : 1; pl->info.grid_fwd = grid;
        pl->info.fb_blocks = 0; pl->info.fb_blocks_dense = 0; pl->info.lds_bytes = dmel::big_uses_global(bt.M) ? 0 : bt.M * 8;
        pl->info.contraction = -1; pl->info.wl_steps = 0;
        return DMEL_OK;
    }

    // The fused kernel builds its own window table (n_fft <= 4096) and, for clips up to 32768 samples, its
    // own clip mean; the prep kernel only runs for what is left: partial sums of long clips, the window
    // table of n_fft 8192 / 16384, and everything the direct-DFT kernel (n_fft < 32) needs.
    const bool fast = N >= dmel::kMinFastNfft && N <= dmel::kMaxFastNfft;
    const bool kernel_mean = fast && pl->cfg.n_points <= 32768;
    const bool need_sums = remove_dc && !kernel_mean && !*sums_done;                  // the launches of one group share the sums
    const bool need_window = !fast || !dmel::forward_window_in_lds(N);
    const size_t m0 = prof_mark(pl, s);
    if (need_sums || need_window) {
        dmel::PrepParams pp{};
        pp.x = x; pp.psum = sc.psum; pp.win2 = sc.win;
        if (flags & DMEL_FLAG_X_INDIRECT) { pp.x_ind = reinterpret_cast<const float* const*>(x); pp.x = nullptr; }
        pp.B = need_sums ? batch : 0; pp.L = pl->cfg.n_points; pp.nchunks = pl->nchunks; pp.chunk = pl->chunk;
        pp.N = need_window ? N : 0; pp.normalize = pl->cfg.normalize_window; pp.win_half = win_half; pp.center = center;
        pp.lam = lam; pp.lam.role = dmel::kLamQuiet;
        DMEL_HIP(dmel::launch_prep(pp, s));
        if (need_sums) *sums_done = true;
    }
    const size_t m1 = prof_mark(pl, s);
    if (need_sums || need_window) prof_span(pl, m0, m1, 0);

    pl->info.n_fft = N; pl->info.n_freqs = tb->F; pl->info.n_time = pl->T;
    const float inv_L = 1.0f / (float)pl->cfg.n_points;
    if (N < dmel::kMinFastNfft) {
        dmel::NaiveParams np{};
        np.x = x_cell ? nullptr : x; np.x_ind = x_cell; np.out = out; np.tangent = tangent; np.psum = sc.psum; np.win2 = sc.win; np.fb = tb->fb_dense;
        np.B = batch; np.L = pl->cfg.n_points; np.T = pl->T; np.hop = pl->cfg.hop_length; np.M = pl->cfg.n_mels;
        np.nchunks = pl->nchunks; np.N = N; np.F = tb->F; np.mode = mode;
        np.inv_L = inv_L; np.eps = (float)eps; np.flags = flags; np.remove_dc = remove_dc; np.lam = lam;
        DMEL_HIP(dmel::launch_naive(np, s));
        prof_span(pl, m1, prof_mark(pl, s), 1);
        pl->info.kernel_path = 1; pl->info.frames_per_tile = 1; pl->info.grid_fwd = batch * pl->T;
        pl->info.fb_blocks = 0; pl->info.fb_blocks_dense = 0; pl->info.lds_bytes = (2 * N + 2 * tb->F) * 4;
        pl->info.contraction = -1; pl->info.wl_steps = 0;
        return DMEL_OK;
    }
    dmel::FwdParams fp{};
    if (flags & DMEL_FLAG_X_INDIRECT) {
        fp.x_ind = reinterpret_cast<const float* const*>(x);
        x = nullptr;
        flags &= ~DMEL_FLAG_X_INDIRECT;
    }
    fp.x = x; fp.out = out; fp.tangent = tangent; fp.psum = (remove_dc && !kernel_mean) ? sc.psum : nullptr; fp.win2 = sc.win;
    fp.tw1 = dmel::mode_pairs(mode) ? tb->tw1p : tb->tw1; fp.tw2 = dmel::mode_pairs(mode) ? tb->tw2p : tb->tw2; fp.ent_b = tb->ent_b; fp.tile_ranges = tb->tile_ranges; fp.ent_b_floats = tb->ent_b_floats; fp.ent_pre = tb->ent_pre;
    for (int i = 0; i < 16; ++i) fp.pre_groups[i] = tb->pre_groups[i];
    fp.B = batch; fp.L = pl->cfg.n_points; fp.T = pl->T; fp.hop = pl->cfg.hop_length; fp.M = pl->cfg.n_mels;
    fp.nchunks = pl->nchunks; fp.groups = tb->groups; fp.xch_groups = tb->xch_groups;
    // DMEL_FLAG_MFMA_BF16X3: the training forward through a caller-supplied (dense) bank contracts on the bf16 matrix pipe
    const bool hsplit = (flags & DMEL_FLAG_MFMA_BF16X3) && mode == dmel::kTrain && tb->ent_h != nullptr;
    if (hsplit) mode = dmel::kTrainH;
    // the wave-local contraction where it is built (n_fft 1024, up to 512 mel bands): DMEL_WLC=0 keeps the round-4 kernel, DMEL_WLC=1
    // the 8-wave workgroups, DMEL_WLC=2 the 16-wave ones where they are built (diagnostics).
    static const int wlc_env = std::getenv("DMEL_WLC") ? std::atoi(std::getenv("DMEL_WLC")) : -1;
    // (a dense bank -- a caller-supplied or trainable matrix -- makes every quad's band the whole spectrum: 1 032 steps per wave at n_fft 1024
    // against 68 for the HTK bank; the 16 x 16 x 4 tiles of kTrain then win -- trainable-filterbank step 128.9 against 114.4 us, round 5)
    const bool wl_compact = tb->wl_total4 * 4 <= 320;
    if (mode == dmel::kTrain && tb->wl_b4 != nullptr && wlc_env != 0 && (wl_compact || wlc_env > 0)) {
        const int wide_fpt = dmel::forward_has_wlc_wide(N) ? dmel::forward_frames_per_tile(N, dmel::kTrainWW) : 0;
        const bool wide = wide_fpt > 0 && wlc_env == 2;      // (16-wave workgroups: measured slower, dmel_kernels.h; built only on request)
        mode = wide ? dmel::kTrainWW : dmel::kTrainW;
        fp.wl_b4 = tb->wl_b4; fp.wl_lane = tb->wl_lane; fp.wl_phases = tb->wl_phases; fp.wl_total4 = tb->wl_total4;
        for (int i = 0; i < dmel::kWlMaxPhases; ++i) { fp.wl_len4[i] = tb->wl_len4[i]; fp.wl_mg[i] = tb->wl_mg[i]; }
        fp.wl_merge = tb->wl_merge;
    }
    const int fpt = dmel::forward_frames_per_tile(N, mode);
    fp.tiles_per_clip = (pl->T + fpt - 1) / fpt;
    fp.inv_L = inv_L; fp.eps = (float)eps; fp.flags = flags; fp.lam = lam;
    fp.remove_dc = remove_dc; fp.normalize = pl->cfg.normalize_window; fp.win_half = win_half;
    fp.spec_out = spec_out;
    fp.ent_h = tb->ent_h; fp.fb_nyq = tb->fb_nyq;
    static const int force_tpw = std::getenv("DMEL_TILES_PER_WG") ? std::atoi(std::getenv("DMEL_TILES_PER_WG")) : 0;   // diagnostics
    int tpw = dmel::forward_tiles_per_wg(N, mode, batch, fp.tiles_per_clip);
    if (force_tpw == 1 || (force_tpw == 2 && dmel::forward_two_tiles(N, mode))) tpw = force_tpw;
    if (hsplit) tpw = 1;
    fp.wgs_per_clip = (fp.tiles_per_clip + tpw - 1) / tpw;
    const long long grid = (long long)batch * fp.wgs_per_clip;
    if (grid > 0x7fffffffLL) return fail(DMEL_ERR_INVALID_ARGUMENT, "too many tiles for one launch");
    if (grid <= dmel::forward_resident_workgroups(N, mode)) fp.flags |= dmel::kFwdEdgeFirst;      // one round: see the kernel's prologue
    DMEL_HIP(dmel::launch_forward(N, mode, tpw, fp, (int)grid, s));
    prof_span(pl, m1, prof_mark(pl, s), 1);
    pl->info.kernel_path = 0; pl->info.frames_per_tile = fpt; pl->info.grid_fwd = (int)grid;
    pl->info.fb_blocks = tb->n_entries; pl->info.fb_blocks_dense = tb->n_dense;
    pl->info.lds_bytes = dmel::forward_lds_bytes(N, mode);
    const bool spec_only = mode == dmel::kSpec || mode == dmel::kSpecTrain;
    pl->info.contraction = spec_only ? -1 : (dmel::mode_wlc(mode) ? 1 : (mode == dmel::kTrainH ? 2 : 0));
    pl->info.wl_steps = dmel::mode_wlc(mode) ? 4 * fp.wl_total4 : 0;
    return DMEL_OK;
}

// An exchange of the plan's mailbox that gave up on a rank left NaN in lambd.grad on the ranks that waited: every later call on
// the plan says so (one pinned word read, no synchronisation) until dmel_mailbox_error has read and cleared the word.
dmel_status mailbox_status(const dmel_plan* pl)
{
    const unsigned long long w = dmel::mailbox_peek_error(pl->mailbox);
    if (w == 0) return DMEL_OK;
    return fail(DMEL_ERR_MAILBOX_TIMEOUT, "mailbox all-reduce: rank " + std::to_string((unsigned)(w & 0xffffffffu)) + " never arrived at exchange " +
                std::to_string((unsigned)(w >> 32)) + ": the reduced gradient of that step was NaN on this rank (dmel_mailbox_error clears this)");
}

// a plan's tables and scratch live on the device it was created on: launching from a thread whose current device is another
// one would run the kernels there, on pointers that mean nothing there
dmel_status check_device(const dmel_plan* pl)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return fail(DMEL_ERR_NO_DEVICE, "no current HIP device"); }
    if (dev != pl->device)
        return fail(DMEL_ERR_INVALID_ARGUMENT, "the plan belongs to device " + std::to_string(pl->device) + " but the current device is " +
                    std::to_string(dev) + " (hipSetDevice / torch.cuda.device before the call)");
    return DMEL_OK;
}

dmel_status check_forward_args(dmel_plan* pl, const float* x, int batch, const void* out)
{
    if (!pl) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (batch < 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "batch < 0");
    if (batch > 0 && (!x || !out)) return fail(DMEL_ERR_INVALID_ARGUMENT, "x / out is NULL");
    if (batch > 65534) return fail(DMEL_ERR_INVALID_ARGUMENT, "batch > 65534 (split the call)");
    if (pl->mailbox) { dmel_status ms = mailbox_status(pl); if (ms != DMEL_OK) return ms; }
    return check_device(pl);
}

// lambd by value (the host has read it, as time_frequency.py:39 does): one launch, plan-owned scratch unless given.
// The plan mutex is held by the caller.
// lambd_dev != nullptr: the kernels read lambd from the device instead (no host read: capturable); the transform length must then
// be given (n_fft_override: the lengths that do not depend on lambd -- optimized=False branches, the spectrogram layer).
dmel_status run_forward_nolock(dmel_plan* pl, const float* x, int batch, float lambd, unsigned flags, double eps,
                        float* out, float* tangent, int mode, int remove_dc, void* stream,
                        int n_fft_override = 0, int win_half = 0, void* scratch = nullptr, const float* lambd_dev = nullptr)
{
    dmel_status st = check_forward_args(pl, x, batch, out);
    if (st != DMEL_OK) return st;
    if (batch == 0) return DMEL_OK;
    if (lambd_dev && n_fft_override <= 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd on the device needs an explicit n_fft");
    if (!lambd_dev && !std::isfinite(lambd)) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd is not finite");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int N = n_fft_override > 0 ? n_fft_override : dmel_n_fft(lambd);
    Scratch sc;
    if (scratch) sc = carve(scratch);
    else {
        if ((st = order_after_last_stream(pl, s)) != DMEL_OK) return st;
        if ((st = ensure_own_scratch(pl, batch, s, &sc)) != DMEL_OK) return st;
    }
    dmel::LamArgs lam{};
    lam.dev = lambd_dev; lam.val = lambd_dev ? 0.f : lambd; lam.n_expected = 0;      // N was derived from this very value (or given explicitly)
    lam.role = dmel::kLamFirst | dmel::kLamLast;
    lam.dot_counter = scratch ? sc.counter : nullptr;            // plan-owned counters are zeroed at allocation and reset themselves
    bool sums_done = false;
    return launch_forward_n(pl, x, batch, N, lam, flags, eps, out, tangent, mode, remove_dc, sc, s, win_half, &sums_done);
}

dmel_status run_forward(dmel_plan* pl, const float* x, int batch, float lambd, unsigned flags, double eps,
                        float* out, float* tangent, int mode, int remove_dc, void* stream,
                        int n_fft_override = 0, int win_half = 0, void* scratch = nullptr, const float* lambd_dev = nullptr)
{
    if (!pl) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    std::lock_guard<std::mutex> lock(pl->mu);
    return run_forward_nolock(pl, x, batch, lambd, flags, eps, out, tangent, mode, remove_dc, stream, n_fft_override, win_half, scratch, lambd_dev);
}

// ---- device-resident lambd ---------------------------------------------------------------------------------------------
void lam_reset(dmel_plan* pl)
{
    pl->lam_known = false; pl->n_obs = 0; pl->lam_rate = 0.f; pl->seq_seen = pl->issued; pl->seq_floor = pl->issued + 1;
}

// The launch choice as a pure function (also exported: dmel_decide_launch): n_fft from lambd, and the neighbours within reach
// of a boundary when lambd may move by `rate` per forward for `stale` forwards before anybody looks again (factor 2 of margin)
void decide_launch(float lambd, float rate, float stale, int* n_fft, int* guards)
{
    const float a = std::fabs(lambd);
    const int N = dmel_n_fft(lambd);
    int g = 0;
    const float reach = 2.0f * rate * stale + 1e-5f * a;
    if ((a - reach) * 6.0f < (float)(N / 2 + 1)) g |= 1;
    if ((a + reach) * 6.0f >= (float)N + 1.0f) g |= 2;
    if (2 * N > dmel::kMaxBigFft) g &= ~2;
    if (N < 2) g &= ~1;
    *n_fft = N; *guards = g;
}

// which n_fft the next dmel_forward_dev launches for and which neighbours it guards (bit 0: n_fft / 2, bit 1: 2 n_fft), from
// the host's current picture: both neighbours while the drift per call is unknown, on request, and under a graph capture
// nobody manages; otherwise only the ones within reach of a boundary before the host would notice
void lam_decide(const dmel_plan* pl, bool capturing, int* n_fft, int* guards)
{
    if (pl->forced_n_fft > 0) { *n_fft = pl->forced_n_fft; *guards = pl->forced_guards; return; }
    const int N = dmel_n_fft(pl->lam_seen);
    int g = 0;
    if (pl->guard_mode == 1 || (capturing && pl->guard_mode != 3 && pl->guard_mode != 2)) g = 3;
    else if (pl->guard_mode == 0 || pl->guard_mode == 3) {
        if (pl->n_obs < 2) g = 3;
        else {
            const float stale = (float)(pl->issued - pl->seq_seen) + 2.0f + (capturing ? (float)pl->max_ahead : 0.f);
            int n2 = 0;
            decide_launch(pl->lam_seen, pl->lam_rate, stale, &n2, &g);
        }
    }
    if (2 * N > dmel::kMaxBigFft) g &= ~2;
    if (N < 2) g &= ~1;
    *n_fft = N; *guards = g;
}

// fold the kernels' latest report into the host's picture; returns false if nothing new.  The ring is scanned for the
// highest execution number past the last one seen (and past the last reset): whichever forward executed last -- an eager
// call or a replay of a captured one -- is what the picture follows.
bool lam_observe(dmel_plan* pl)
{
    bool any = false;
    unsigned best_seq = 0; unsigned best_bits = 0;
    for (unsigned i = 0; i < dmel::kLamRing; ++i) {
        const unsigned long long w = __atomic_load_n(&pl->host_words[i], __ATOMIC_RELAXED);
        const unsigned seq = (unsigned)(w >> 32);
        if (seq == 0 || (int)(seq - pl->seq_floor) < 0) continue;                       // empty, or older than the last reset
        if (pl->lam_known ? (int)(seq - pl->seq_seen) <= 0 : false) continue;           // not newer than what is known
        if (!any || (int)(seq - best_seq) > 0) { best_seq = seq; best_bits = (unsigned)w; any = true; }
    }
    if (!any) return false;
    float lam; std::memcpy(&lam, &best_bits, 4);
    if (pl->lam_known && pl->n_obs >= 1) {
        const unsigned dseq = best_seq - pl->seq_seen;
        const float r = std::fabs(lam - pl->lam_seen) / (float)(dseq ? dseq : 1);
        pl->lam_rate = std::max(0.98f * pl->lam_rate, r);
    }
    pl->lam_seen = lam; pl->seq_seen = best_seq; pl->lam_known = true; ++pl->n_obs;
    if ((int)(best_seq - pl->issued) > 0) pl->issued = best_seq;                      // replays executed forwards the host never counted
    return true;
}

std::mutex g_plans_mu;             // guards dmel_plan::refs and g_live_plans
std::unordered_set<const dmel_plan*> g_live_plans;     // every plan between dmel_plan_create and its last release (dmel_plan_is_live)

}  // namespace

extern "C" {

int32_t dmel_abi_version(void) { return DMEL_ABI_VERSION; }

const char* dmel_last_error(void) { return g_err.c_str(); }

int32_t dmel_n_fft(float lambd)
{
    const float a = std::fabs(lambd);       // models.py:38
    const float prod = a * 6.0f;            // time_frequency.py:39, fp32 tensor product
    if (!(prod < 9.0e15f)) return 0x40000000;
    long long x = (long long)prod;          // time_frequency.py:61 int(): truncation
    long long v = x - 1;
    int bits = 0;
    if (v < 0) bits = 1;                    // python: (-1).bit_length() == 1
    else while (v > 0) { ++bits; v >>= 1; }
    if (bits > 30) return 0x40000000;
    return (int32_t)(1LL << bits);          // time_frequency.py:62
}

dmel_status dmel_window_host(float lambd, int32_t n_fft, int32_t normalize, float* window, float* dwindow)
{
    if (n_fft < 1 || !window) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_window_host: bad arguments");
    const float denom = std::fabs(lambd) + 1e-15f;
    const double den = (double)denom;
    std::vector<double> dw(n_fft);
    double ww = 0.0, wd = 0.0;
    for (int n = 0; n < n_fft; ++n) {
        const float d = (float)n - (float)n_fft / 2.0f;
        const float t = d / denom;
        window[n] = std::exp(-0.5f * (t * t));
        dw[n] = (double)window[n] * (double)d * (double)d / (den * den * den);
        ww += (double)window[n] * (double)window[n];
        wd += (double)window[n] * dw[n];
    }
    if (normalize) {
        const double nrm = std::sqrt(ww);
        for (int n = 0; n < n_fft; ++n) {
            const double w = (double)window[n];
            dw[n] = dw[n] / nrm - w * wd / (nrm * nrm * nrm);
            window[n] = (float)(w / nrm);
        }
    }
    if (dwindow) for (int n = 0; n < n_fft; ++n) dwindow[n] = (float)dw[n];
    return DMEL_OK;
}

dmel_status dmel_contraction_partition_host(const int32_t* units, int32_t n_tiles, int32_t waves, int32_t* own, int32_t* n_pieces,
                                            int32_t* piece_wave, int32_t* piece_tile, int32_t* piece_first, int32_t* piece_units)
{
    if (!units || !own || !n_pieces || !piece_wave || !piece_tile || !piece_first || !piece_units || (waves != 4 && waves != 8) ||
        n_tiles < 0 || n_tiles > waves)
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_contraction_partition_host: bad arguments");
    int u[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8];
    for (int t = 0; t < n_tiles; ++t) { if (units[t] < 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_contraction_partition_host: negative extent"); u[t] = units[t]; }
    std::vector<Piece> pieces;
    deal_ksteps(u, n_tiles, waves, o, &pieces);
    for (int w = 0; w < 8; ++w) own[w] = o[w];
    *n_pieces = (int32_t)pieces.size();
    for (size_t i = 0; i < pieces.size() && i < 8; ++i) {
        piece_wave[i] = pieces[i].wave; piece_tile[i] = pieces[i].tile; piece_first[i] = pieces[i].a; piece_units[i] = pieces[i].n;
    }
    return DMEL_OK;
}

dmel_status dmel_mel_fbanks_host(int32_t n_freqs, double f_min, double f_max, int32_t n_mels,
                                 int32_t sample_rate, float* fb)
{
    if (n_freqs < 1 || n_mels < 1 || sample_rate < 2 || !fb)
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_mel_fbanks_host: bad arguments");
    std::vector<float> all_freqs, m_pts, f_pts(n_mels + 2);
    linspace_f32(0.0f, (float)(sample_rate / 2), n_freqs, all_freqs);
    const double m_min = 2595.0 * std::log10(1.0 + f_min / 700.0);
    const double m_max = 2595.0 * std::log10(1.0 + f_max / 700.0);
    linspace_f32((float)m_min, (float)m_max, n_mels + 2, m_pts);
    // 10 ** (m / 2595) as torch evaluates it: the CORRECTLY ROUNDED fp32 power (measured against torch 2.10: all 130 points of the 128-band bank;
    // glibc's powf is 1 ulp off at 19 of them, which moved filterbank entries by up to 1.5e-5 -- tests/golden/g8_fbanks.npz)
    for (int i = 0; i < n_mels + 2; ++i) {
        const float e = m_pts[i] / 2595.0f;
        f_pts[i] = 700.0f * ((float)std::pow(10.0, (double)e) - 1.0f);
    }
    for (int f = 0; f < n_freqs; ++f)
        for (int m = 0; m < n_mels; ++m) {
            const float down = (-1.0f * (f_pts[m] - all_freqs[f])) / (f_pts[m + 1] - f_pts[m]);
            const float up = (f_pts[m + 2] - all_freqs[f]) / (f_pts[m + 2] - f_pts[m + 1]);
            fb[(size_t)f * n_mels + m] = std::max(0.0f, std::min(down, up));
        }
    return DMEL_OK;
}

int32_t dmel_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) continue;
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

dmel_status dmel_plan_create(const dmel_config* cfg, dmel_plan** plan)
{
    if (!cfg || !plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "cfg / plan is NULL");
    *plan = nullptr;
    if (cfg->n_points < 1 || cfg->hop_length < 1 || cfg->n_mels < 1 || cfg->sample_rate < 2)
        return fail(DMEL_ERR_INVALID_ARGUMENT, "n_points, hop_length, n_mels must be >= 1 and sample_rate >= 2");
    if (cfg->n_mels > 65535) return fail(DMEL_ERR_INVALID_ARGUMENT, "n_mels too large");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return fail(DMEL_ERR_NO_DEVICE, "no HIP device"); }
    hipDeviceProp_t prop;
    DMEL_HIP(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DMEL_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", libdmel_hip is built for gfx950 only");
    DMEL_HIP(dmel::forward_prepare_attributes());
    DMEL_HIP(dmel::xgrad_prepare_attributes());
    DMEL_HIP(dmel::big_prepare_attributes());
    dmel_plan* pl = new (std::nothrow) dmel_plan();
    if (!pl) return fail(DMEL_ERR_OUT_OF_MEMORY, "host allocation failed");
    pl->cfg = *cfg;
    pl->device = dev;
    pl->T = cfg->n_points / cfg->hop_length + 1;                      // models.py:30
    pl->f_max = cfg->f_max < 0 ? (double)(cfg->sample_rate / 2) : cfg->f_max;   // models.py:25
    int nch = (cfg->n_points + 4095) / 4096;
    nch = std::max(1, std::min(nch, dmel::kMaxChunks));
    pl->nchunks = nch;
    pl->chunk = ((cfg->n_points + nch - 1) / nch + 3) / 4 * 4;   // multiple of 4 samples: chunks keep 16-byte alignment
    hipError_t e = hipEventCreateWithFlags(&pl->xev, hipEventDisableTiming);
    // two words the kernels write and the host reads without synchronising (device-resident lambd, dmel_forward_dev)
    constexpr size_t kHostWordBytes = (dmel::kLamRing + 8) * sizeof(unsigned long long);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&pl->host_words), kHostWordBytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&pl->exec_counter), 64);
    if (e == hipSuccess) e = hipMemset(pl->exec_counter, 0, 64);
    if (e != hipSuccess) { dmel_plan_destroy(pl); return fail(DMEL_ERR_HIP, std::string("plan resources: ") + hipGetErrorString(e)); }
    std::memset(pl->host_words, 0, kHostWordBytes);
    {
        Scratch sc;
        dmel_status st = ensure_own_scratch(pl, std::max(1, cfg->max_batch), nullptr, &sc);
        if (st != DMEL_OK) { dmel_plan_destroy(pl); return st; }
    }
    {
        std::lock_guard<std::mutex> lock(g_plans_mu);
        g_live_plans.insert(pl);
    }
    *plan = pl;
    return DMEL_OK;
}

int32_t dmel_lambd_ring_size(void) { return (int32_t)dmel::kLamRing; }

int32_t dmel_plan_is_live(const dmel_plan* plan)
{
    if (!plan) return 0;
    std::lock_guard<std::mutex> lock(g_plans_mu);
    return g_live_plans.count(plan) ? 1 : 0;
}

dmel_status dmel_plan_retain(dmel_plan* plan)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    std::lock_guard<std::mutex> lock(g_plans_mu);
    if (!g_live_plans.count(plan)) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_plan_retain: not a live plan (destroyed, or never created)");
    if (plan->refs < 1) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_plan_retain: the plan has been destroyed");
    ++plan->refs;
    return DMEL_OK;
}

dmel_status dmel_plan_release(dmel_plan* plan)
{
    if (!plan) return DMEL_OK;
    {
        std::lock_guard<std::mutex> lock(g_plans_mu);
        if (--plan->refs > 0) return DMEL_OK;
        g_live_plans.erase(plan);          // from here on dmel_plan_is_live says no: a stale integer handle is refused, not dereferenced
    }
    // the last holder is gone.  Work that still reads the tables or writes the report words may be queued (a backward of a
    // graph that outlived its layer, the last launches of a loop): let the device finish before the memory goes away
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != plan->device && hipSetDevice(plan->device) == hipSuccess;
    (void)hipDeviceSynchronize();
    dmel::mailbox_attach_count(plan->mailbox, -1);       // after the synchronisation: no queued dot kernel of this plan polls it any more
    plan->mailbox = nullptr;
    for (auto& kv : plan->tables) kv.second.release();
    for (hipEvent_t e : plan->ev_pool) (void)hipEventDestroy(e);
    (void)hipFree(plan->own_scratch); (void)hipFree(plan->fbw);
    for (auto& kv : plan->big_tabs) { (void)hipFree(kv.second.chirp); (void)hipFree(kv.second.hbr); }
    for (auto& kv : plan->big_tw) (void)hipFree(kv.second);
    for (auto& kv : plan->big_wodd) (void)hipFree(kv.second);
    (void)hipFree(plan->big_win); (void)hipFree(plan->big_z);
    (void)hipFree(plan->exec_counter);
    if (plan->host_words) (void)hipHostFree(plan->host_words);
    if (plan->xev) (void)hipEventDestroy(plan->xev);
    if (switched) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    delete plan;
    return DMEL_OK;
}

dmel_status dmel_plan_destroy(dmel_plan* plan) { return dmel_plan_release(plan); }

dmel_status dmel_plan_set_filterbank(dmel_plan* plan, int32_t n_fft, const float* fb)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (n_fft < 1 || n_fft > dmel::kMaxBigFft || (n_fft > 1 && (n_fft & 1)))
        return fail(DMEL_ERR_INVALID_ARGUMENT, "n_fft must be 1 or an even length up to 1048576");
    { dmel_status sd = check_device(plan); if (sd != DMEL_OK) return sd; }      // the synchronisation below must be the plan's device's
    std::lock_guard<std::mutex> lock(plan->mu);
    DMEL_HIP(hipDeviceSynchronize());     // tables of this n_fft may be in use by queued kernels
    auto it = plan->tables.find(n_fft);
    if (it != plan->tables.end()) { it->second.release(); plan->tables.erase(it); }
    plan->dense_dev.erase(n_fft);
    if (fb) {
        const size_t n = (size_t)(n_fft / 2 + 1) * plan->cfg.n_mels;
        plan->custom_fb[n_fft] = std::vector<float>(fb, fb + n);
    } else {
        plan->custom_fb.erase(n_fft);
    }
    return DMEL_OK;
}

dmel_status dmel_plan_set_filterbank_dev(dmel_plan* plan, int32_t n_fft, const float* fb_dev, void* stream)
{
    if (!plan || !fb_dev) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan / fb_dev is NULL");
    if (n_fft < 1 || n_fft > dmel::kMaxBigFft || (n_fft > 1 && (n_fft & 1)))
        return fail(DMEL_ERR_INVALID_ARGUMENT, "n_fft must be 1 or an even length up to 1048576");
    { dmel_status sd = check_device(plan); if (sd != DMEL_OK) return sd; }
    std::lock_guard<std::mutex> lock(plan->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!plan->dense_dev.count(n_fft)) {
        // first use for this n_fft: the tables get the dense structure (every 4x16 block present) once; from then on an update is
        // one small kernel on the caller's stream, no host copy and no synchronisation
        if (is_capturing(s)) return fail(DMEL_ERR_INVALID_ARGUMENT, "first dmel_plan_set_filterbank_dev for an n_fft builds tables: call it once before capturing");
        DMEL_HIP(hipDeviceSynchronize());
        auto it = plan->tables.find(n_fft);
        if (it != plan->tables.end()) { it->second.release(); plan->tables.erase(it); }
        plan->custom_fb.erase(n_fft);
        plan->dense_dev[n_fft] = true;
    }
    NfftTables* tb = nullptr;
    dmel_status st = build_tables(plan, n_fft, &tb);
    if (st != DMEL_OK) return st;
    dmel::RepackParams rp{};
    rp.fb = fb_dev; rp.ent_b = tb->ent_b; rp.ent_pre = tb->ent_pre; rp.tile_ranges = tb->tile_ranges;
    rp.fb_dense = tb->fb_dense; rp.fbT = tb->fbT; rp.rowpk = tb->rowpk; rp.F = tb->F; rp.M = plan->cfg.n_mels;
    rp.ent_h = tb->ent_h; rp.fb_nyq = tb->fb_nyq; rp.ks32 = tb->ent_h ? n_fft / 64 : 0;
    rp.wl_b4 = tb->wl_b4; rp.wl_lane = tb->wl_lane; rp.wl_total4 = tb->wl_total4; rp.wl_phases = tb->wl_phases;
    for (int i = 0; i < dmel::kWlMaxPhases; ++i) rp.wl_len4[i] = tb->wl_len4[i];
    const bool fast = n_fft >= dmel::kMinFastNfft && n_fft <= dmel::kMaxFastNfft;
    const int waves = fast ? dmel::forward_waves(n_fft) : 0;
    rp.runs = fast ? tb->groups * waves * 2 : 0;
    rp.runs_group0 = waves * 2;
    rp.nbpre = fast ? dmel::forward_nbpre(n_fft) : 0;
    DMEL_HIP(dmel::launch_repack(rp, s));
    return DMEL_OK;
}

dmel_status dmel_forward_scratch(dmel_plan* plan, const float* x, int32_t batch, float lambd, uint32_t flags,
                                 double eps, void* out, float* tangent, void* scratch, void* stream)
{
    if (flags & DMEL_FLAG_X_INDIRECT) return fail(DMEL_ERR_INVALID_ARGUMENT, "DMEL_FLAG_X_INDIRECT belongs to dmel_forward_dev");
    if (flags & DMEL_FLAG_FULL_WINDOW) {
        if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
        const int L = plan->cfg.n_points;
        return run_forward(plan, x, batch, lambd, flags & ~DMEL_FLAG_FULL_WINDOW, eps, static_cast<float*>(out), tangent,
                           tangent ? dmel::kTrain : dmel::kInfer, 1, stream, 2 * L, 1, scratch);
    }
    return run_forward(plan, x, batch, lambd, flags, eps, static_cast<float*>(out), tangent,
                       tangent ? dmel::kTrain : dmel::kInfer, /*remove_dc=*/1, stream, 0, 0, scratch);
}

dmel_status dmel_forward(dmel_plan* plan, const float* x, int32_t batch, float lambd, uint32_t flags,
                         double eps, float* out, float* tangent, void* stream)
{
    return dmel_forward_scratch(plan, x, batch, lambd, flags, eps, out, tangent, nullptr, stream);
}

dmel_status dmel_plan_get_config(const dmel_plan* plan, dmel_config* cfg)
{
    if (!plan || !cfg) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan / cfg is NULL");
    *cfg = plan->cfg;
    return DMEL_OK;
}

size_t dmel_scratch_bytes(const dmel_plan* plan, int32_t batch)
{
    if (!plan || batch < 0) return 0;
    return scratch_bytes(plan, batch);
}

dmel_status dmel_forward_dev(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, uint32_t flags,
                             double eps, void* out, float* tangent, void* scratch, void* stream)
{
    dmel_status st = check_forward_args(plan, x, batch, out);
    if (st != DMEL_OK) return st;
    if (!lambd_dev) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd_dev is NULL");
    if (flags & DMEL_FLAG_FULL_WINDOW)
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_forward_dev: DMEL_FLAG_FULL_WINDOW has a fixed n_fft, use dmel_forward");
    std::lock_guard<std::mutex> lock(plan->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool capturing = is_capturing(s);

    // a forward that no launch covered: its outputs are NaN on the device; tell the caller once and start over
    const unsigned long long err = __atomic_load_n(&plan->host_words[dmel::kLamRing], __ATOMIC_RELAXED);
    if (err != 0) {
        float lam; const unsigned bits = (unsigned)err; std::memcpy(&lam, &bits, 4);
        __atomic_store_n(&plan->host_words[dmel::kLamRing], 0ull, __ATOMIC_RELAXED);
        const float seen = plan->lam_seen;
        lam_reset(plan);
        return fail(DMEL_ERR_LAMBD_TRACKING,
                    "lambd moved from " + std::to_string(seen) + " (n_fft " + std::to_string(dmel_n_fft(seen)) + ") to " +
                    std::to_string(lam) + " (n_fft " + std::to_string(dmel_n_fft(lam)) + ") faster than the sync-free forward "
                    "tracks it: call " + std::to_string((unsigned)(err >> 32)) + " produced NaN.  Tracking has been reset; use "
                    "dmel_forward (host read per call), dmel_plan_set_tracking(plan, max_ahead, 1) or a smaller run-ahead");
    }
    if (batch == 0) return DMEL_OK;
    lam_observe(plan);
    if (!plan->lam_known) {
        // cold start: the one blocking read (the reference does one per sample, time_frequency.py:39)
        if (capturing) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_forward_dev: run the layer once eagerly before capturing it into a graph");
        float lam = 0.f;
        DMEL_HIP(hipMemcpyAsync(&lam, lambd_dev, sizeof(float), hipMemcpyDeviceToHost, s));
        DMEL_HIP(hipStreamSynchronize(s));
        if (!std::isfinite(lam)) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd is not finite");
        // the stream is idle: every report of earlier forwards on it is in the ring; whatever executed without the host
        // counting it (replays of captured forwards) is caught up with here
        for (unsigned i = 0; i < dmel::kLamRing; ++i) {
            const unsigned q = (unsigned)(__atomic_load_n(&plan->host_words[i], __ATOMIC_RELAXED) >> 32);
            if (q != 0 && (int)(q - plan->issued) > 0) plan->issued = q;
        }
        plan->lam_seen = lam; plan->seq_seen = plan->issued; plan->lam_known = true; plan->n_obs = 0; plan->lam_rate = 0.f;
    } else if (!capturing && plan->max_ahead > 0) {
        // bounded run-ahead: the picture of lambd the launches are chosen from is at most max_ahead calls old
        for (long spins = 0; plan->issued - plan->seq_seen > (unsigned)plan->max_ahead && spins < 200000000L; ++spins) {
            if (!lam_observe(plan) && (spins & 1023) == 1023 && hipStreamQuery(s) == hipSuccess) { lam_observe(plan); break; }
        }
        (void)hipGetLastError();
    }
    int N = 0, guards = 0;
    lam_decide(plan, capturing, &N, &guards);
    // tables of the neighbouring sizes exist before they are needed: building them allocates and copies (not allowed
    // under capture, and a stall at the moment of a crossing otherwise)
    if (!capturing) {
        NfftTables* tb = nullptr;
        for (int n : {N, 2 * N, N / 2})
            if (n >= 1 && n <= dmel::kMaxBigFft && (st = build_tables(plan, n, &tb)) != DMEL_OK) return st;
    }
    int cand[3], nc = 0;
    cand[nc++] = N;
    if (guards & 2) cand[nc++] = 2 * N;
    if (guards & 1) cand[nc++] = N / 2;
    plan->last_guards = guards;

    Scratch sc;
    if (scratch) sc = carve(scratch);
    else {
        if ((st = order_after_last_stream(plan, s)) != DMEL_OK) return st;
        if ((st = ensure_own_scratch(plan, batch, s, &sc)) != DMEL_OK) return st;
    }
    ++plan->calls;
    if (!capturing) ++plan->issued;          // a captured forward executes when (and as often as) its graph is replayed
    bool sums_done = false;
    dmel_plan_info primary_info = plan->info;
    for (int i = 0; i < nc; ++i) {
        dmel::LamArgs lam{};
        lam.dev = lambd_dev; lam.val = 0.f; lam.n_expected = cand[i];
        lam.role = (i == 0 ? dmel::kLamFirst : 0u) | (i == nc - 1 ? dmel::kLamLast : 0u);
        lam.exec_counter = plan->exec_counter; lam.handled = sc.handled;
        lam.host_seen = &plan->host_words[0]; lam.host_error = &plan->host_words[dmel::kLamRing];
        lam.dot_counter = scratch ? sc.counter : nullptr;
        st = launch_forward_n(plan, x, batch, cand[i], lam, flags, eps, static_cast<float*>(out), tangent,
                              tangent ? dmel::kTrain : dmel::kInfer, /*remove_dc=*/1, sc, s, 0, &sums_done);
        if (st != DMEL_OK) return st;
        if (i == 0) primary_info = plan->info;
    }
    plan->info = primary_info;        // dmel_plan_get_info describes the launch the host expected to do the work
    return DMEL_OK;
}

dmel_status dmel_forward_dev_fixed(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft,
                                   uint32_t flags, double eps, void* out, float* tangent, void* scratch, void* stream)
{
    dmel_status st = check_forward_args(plan, x, batch, out);
    if (st != DMEL_OK) return st;
    if (!lambd_dev) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd_dev is NULL");
    if (flags & DMEL_FLAG_X_INDIRECT) return fail(DMEL_ERR_INVALID_ARGUMENT, "DMEL_FLAG_X_INDIRECT belongs to dmel_forward_dev");
    if (flags & DMEL_FLAG_FULL_WINDOW) {
        // optimized=False (time_frequency.py:41,51): window = the clip, n_fft = 2 n_points whatever lambd is -- nothing to check on
        // the device, nothing for the host to read: lambd only shapes the window, which the kernels build from the device value
        return run_forward(plan, x, batch, 0.f, flags & ~DMEL_FLAG_FULL_WINDOW, eps, static_cast<float*>(out), tangent,
                           tangent ? dmel::kTrain : dmel::kInfer, 1, stream, 2 * plan->cfg.n_points, 1, scratch, lambd_dev);
    }
    if (n_fft < 1 || n_fft > dmel::kMaxNfft || (n_fft & (n_fft - 1)))
        return fail(DMEL_ERR_INVALID_ARGUMENT, "n_fft must be a power of two in [1, 16384]");
    std::lock_guard<std::mutex> lock(plan->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned long long err = __atomic_load_n(&plan->host_words[dmel::kLamRing], __ATOMIC_RELAXED);
    if (err != 0) {
        float lam; const unsigned bits = (unsigned)err; std::memcpy(&lam, &bits, 4);
        __atomic_store_n(&plan->host_words[dmel::kLamRing], 0ull, __ATOMIC_RELAXED);
        lam_reset(plan);
        return fail(DMEL_ERR_LAMBD_TRACKING,
                    "lambd = " + std::to_string(lam) + " asks for n_fft " + std::to_string(dmel_n_fft(lam)) + " but the forward was issued for the "
                    "fixed n_fft " + std::to_string(n_fft) + " (a learnable filterbank is tied to one n_fft): call " +
                    std::to_string((unsigned)(err >> 32)) + " produced NaN");
    }
    if (batch == 0) return DMEL_OK;
    Scratch sc;
    if (scratch) sc = carve(scratch);
    else {
        if ((st = order_after_last_stream(plan, s)) != DMEL_OK) return st;
        if ((st = ensure_own_scratch(plan, batch, s, &sc)) != DMEL_OK) return st;
    }
    dmel::LamArgs lam{};
    lam.dev = lambd_dev; lam.val = 0.f; lam.n_expected = n_fft;
    lam.role = dmel::kLamFirst | dmel::kLamLast;
    ++plan->calls;
    if (!is_capturing(s)) ++plan->issued;
    lam.exec_counter = plan->exec_counter; lam.handled = sc.handled;
    lam.host_seen = &plan->host_words[0]; lam.host_error = &plan->host_words[dmel::kLamRing];
    lam.dot_counter = scratch ? sc.counter : nullptr;
    plan->last_guards = 0;
    bool sums_done = false;
    return launch_forward_n(plan, x, batch, n_fft, lam, flags, eps, static_cast<float*>(out), tangent,
                            tangent ? dmel::kTrain : dmel::kInfer, /*remove_dc=*/1, sc, s, 0, &sums_done);
}

dmel_status dmel_forward_dev_fixed_spec(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft,
                                        uint32_t flags, double eps, void* out, float* tangent, float* spec, void* scratch, void* stream)
{
    if (!spec) return dmel_forward_dev_fixed(plan, x, batch, lambd_dev, n_fft, flags, eps, out, tangent, scratch, stream);
    dmel_status st = check_forward_args(plan, x, batch, out);
    if (st != DMEL_OK) return st;
    if (!lambd_dev || !tangent) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_forward_dev_fixed_spec: lambd_dev / tangent is NULL (the training forward saves the spectrogram)");
    if (flags & DMEL_FLAG_FULL_WINDOW) return fail(DMEL_ERR_UNSUPPORTED, "dmel_forward_dev_fixed_spec: not with DMEL_FLAG_FULL_WINDOW");
    if (n_fft < dmel::kMinFastNfft || n_fft > dmel::kMaxNfft || (n_fft & (n_fft - 1)))
        return fail(DMEL_ERR_UNSUPPORTED, "dmel_forward_dev_fixed_spec: n_fft must be a power of two in [32, 16384]");
    std::lock_guard<std::mutex> lock(plan->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned long long err = __atomic_load_n(&plan->host_words[dmel::kLamRing], __ATOMIC_RELAXED);
    if (err != 0) {
        float lam; const unsigned bits = (unsigned)err; std::memcpy(&lam, &bits, 4);
        __atomic_store_n(&plan->host_words[dmel::kLamRing], 0ull, __ATOMIC_RELAXED);
        lam_reset(plan);
        return fail(DMEL_ERR_LAMBD_TRACKING,
                    "lambd = " + std::to_string(lam) + " asks for n_fft " + std::to_string(dmel_n_fft(lam)) + " but the forward was issued for the "
                    "fixed n_fft " + std::to_string(n_fft) + " (a learnable filterbank is tied to one n_fft): call " +
                    std::to_string((unsigned)(err >> 32)) + " produced NaN");
    }
    if (batch == 0) return DMEL_OK;
    Scratch sc;
    if (scratch) sc = carve(scratch);
    else {
        if ((st = order_after_last_stream(plan, s)) != DMEL_OK) return st;
        if ((st = ensure_own_scratch(plan, batch, s, &sc)) != DMEL_OK) return st;
    }
    dmel::LamArgs lam{};
    lam.dev = lambd_dev; lam.val = 0.f; lam.n_expected = n_fft;
    lam.role = dmel::kLamFirst | dmel::kLamLast;
    ++plan->calls;
    if (!is_capturing(s)) ++plan->issued;
    lam.exec_counter = plan->exec_counter; lam.handled = sc.handled;
    lam.host_seen = &plan->host_words[0]; lam.host_error = &plan->host_words[dmel::kLamRing];
    lam.dot_counter = scratch ? sc.counter : nullptr;
    plan->last_guards = 0;
    bool sums_done = false;
    return launch_forward_n(plan, x, batch, n_fft, lam, flags, eps, static_cast<float*>(out), tangent, dmel::kTrain, /*remove_dc=*/1, sc, s, 0,
                            &sums_done, spec);
}

dmel_status dmel_plan_lambd_status(dmel_plan* plan, dmel_lambd_status* status)
{
    if (!plan || !status) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan / status is NULL");
    std::lock_guard<std::mutex> lock(plan->mu);
    lam_observe(plan);
    dmel_lambd_status r{};
    r.known = plan->lam_known ? 1 : 0;
    r.lambd_seen = plan->lam_seen;
    r.n_fft_seen = plan->lam_known ? dmel_n_fft(plan->lam_seen) : 0;
    r.seq_issued = plan->issued; r.seq_seen = plan->seq_seen; r.calls = plan->calls;
    r.rate = plan->lam_rate; r.guards = plan->last_guards;
    if (plan->lam_known) lam_decide(plan, false, &r.next_n_fft, &r.next_guards);
    const unsigned long long err = __atomic_load_n(&plan->host_words[dmel::kLamRing], __ATOMIC_RELAXED);
    r.error = err != 0 ? 1 : 0;
    r.error_seq = (uint32_t)(err >> 32);
    { const unsigned bits = (unsigned)err; std::memcpy(&r.error_lambd, &bits, 4); }
    *status = r;
    return DMEL_OK;
}

dmel_status dmel_plan_set_tracking(dmel_plan* plan, int32_t max_ahead, int32_t guard_mode)
{
    if (!plan || max_ahead < 0 || guard_mode < 0 || guard_mode > 3) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_plan_set_tracking: bad arguments");
    std::lock_guard<std::mutex> lock(plan->mu);
    plan->max_ahead = max_ahead; plan->guard_mode = guard_mode;
    return DMEL_OK;
}

dmel_status dmel_plan_lambd_reset(dmel_plan* plan)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    std::lock_guard<std::mutex> lock(plan->mu);
    lam_reset(plan);
    return DMEL_OK;
}

dmel_status dmel_plan_lambd_report(dmel_plan* plan, uint32_t number, float* lambd, int32_t* found)
{
    if (!plan || !lambd || !found) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_plan_lambd_report: NULL argument");
    const unsigned long long w = __atomic_load_n(&plan->host_words[number % dmel::kLamRing], __ATOMIC_RELAXED);
    *found = (number != 0 && (unsigned)(w >> 32) == number) ? 1 : 0;
    if (*found) { const unsigned bits = (unsigned)w; std::memcpy(lambd, &bits, 4); }
    return DMEL_OK;
}

dmel_status dmel_plan_force_launch(dmel_plan* plan, int32_t n_fft, int32_t guards)
{
    if (!plan || n_fft < 0 || guards < 0 || guards > 3) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_plan_force_launch: bad arguments");
    if (n_fft > dmel::kMaxBigFft) return fail(DMEL_ERR_UNSUPPORTED, "n_fft beyond the HIP path");
    std::lock_guard<std::mutex> lock(plan->mu);
    plan->forced_n_fft = n_fft; plan->forced_guards = n_fft > 0 ? guards : 0;
    if (n_fft > 0 && 2 * n_fft > dmel::kMaxBigFft) plan->forced_guards &= ~2;
    if (n_fft > 0 && n_fft < 2) plan->forced_guards &= ~1;
    return DMEL_OK;
}

dmel_status dmel_decide_launch(float lambd, float rate, float stale_forwards, int32_t* n_fft, int32_t* guards)
{
    if (!n_fft || !guards || !(rate >= 0.f) || !(stale_forwards >= 0.f)) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_decide_launch: bad arguments");
    int n = 0, g = 0;
    decide_launch(lambd, rate, stale_forwards, &n, &g);
    *n_fft = n; *guards = g;
    return DMEL_OK;
}

dmel_status dmel_spectrogram(dmel_plan* plan, const float* x, int32_t batch, float lambd,
                             int32_t remove_dc, float* spec, void* stream)
{
    return run_forward(plan, x, batch, lambd, 0u, 0.0, spec, nullptr, dmel::kSpec, remove_dc ? 1 : 0, stream);
}

dmel_status dmel_spectrogram_ex(dmel_plan* plan, const float* x, int32_t batch, float lambd, int32_t n_fft,
                                uint32_t flags, float* spec, float* tangent, void* stream)
{
    if (n_fft < 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "n_fft < 0");
    return run_forward(plan, x, batch, lambd, 0u, 0.0, spec, tangent, tangent ? dmel::kSpecTrain : dmel::kSpec,
                       (flags & DMEL_SPEC_REMOVE_DC) ? 1 : 0, stream, n_fft, (flags & DMEL_SPEC_HALF_WINDOW) ? 1 : 0);
}

dmel_status dmel_spectrogram_ex_dev(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft,
                                    uint32_t flags, float* spec, float* tangent, void* stream)
{
    if (!lambd_dev) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd_dev is NULL");
    if (n_fft <= 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_spectrogram_ex_dev: n_fft must be given (a length that depends on lambd needs the host value: dmel_spectrogram_ex)");
    return run_forward(plan, x, batch, 0.f, 0u, 0.0, spec, tangent, tangent ? dmel::kSpecTrain : dmel::kSpec,
                       (flags & DMEL_SPEC_REMOVE_DC) ? 1 : 0, stream, n_fft, (flags & DMEL_SPEC_HALF_WINDOW) ? 1 : 0, nullptr, lambd_dev);
}

dmel_status dmel_backward_ex(dmel_plan* plan, const void* grad_out, int32_t grad_dtype, const float* tangent, int64_t count,
                             int32_t accumulate, float* dlambd, void* stream)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (count < 0 || !dlambd || (count > 0 && (!grad_out || !tangent)))
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward: bad arguments");
    if (grad_dtype != DMEL_DTYPE_F32 && grad_dtype != DMEL_DTYPE_BF16)
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_ex: grad_dtype must be DMEL_DTYPE_F32 or DMEL_DTYPE_BF16");
    return dmel_backward_scratch(plan, grad_out, grad_dtype, tangent, count, accumulate, dlambd, nullptr, stream);
}

dmel_status dmel_backward_scratch(dmel_plan* plan, const void* grad_out, int32_t grad_dtype, const float* tangent, int64_t count,
                                  int32_t accumulate, float* dlambd, void* scratch, void* stream)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (count < 0 || !dlambd || (count > 0 && (!grad_out || !tangent)))
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward: bad arguments");
    if (grad_dtype != DMEL_DTYPE_F32 && grad_dtype != DMEL_DTYPE_BF16)
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward: grad_dtype must be DMEL_DTYPE_F32 or DMEL_DTYPE_BF16");
    { dmel_status sd = check_device(plan); if (sd != DMEL_OK) return sd; }
    std::lock_guard<std::mutex> lock(plan->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Scratch sc;
    if (scratch) sc = carve(scratch);
    else {
        dmel_status st = order_after_last_stream(plan, s);
        if (st != DMEL_OK) return st;
        if ((st = ensure_own_scratch(plan, 1, s, &sc)) != DMEL_OK) return st;
    }
    dmel::MailboxArgs mba;
    const bool use_mb = plan->mailbox != nullptr;
    if (use_mb && !dmel::mailbox_args(plan->mailbox, &mba)) return fail(DMEL_ERR_INVALID_ARGUMENT, "the plan's mailbox is not connected");
    if (use_mb) { dmel_status ms = mailbox_status(plan); if (ms != DMEL_OK) return ms; }
    const size_t m0 = prof_mark(plan, s);
    DMEL_HIP(dmel::launch_dot(grad_out, grad_dtype == DMEL_DTYPE_BF16, tangent, (long long)count, accumulate, sc.partials,
                              sc.counter, kMaxPartials, dlambd, s, use_mb ? &mba : nullptr, plan->fused_adam.param ? &plan->fused_adam : nullptr));
    prof_span(plan, m0, prof_mark(plan, s), 2);
    return DMEL_OK;
}

dmel_status dmel_plan_attach_adam(dmel_plan* plan, float* param, float* exp_avg, float* exp_avg_sq, float* step,
                                  double lr, double beta1, double beta2, double eps, double weight_decay, int32_t maximize)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    std::lock_guard<std::mutex> lock(plan->mu);
    if (!param) { plan->fused_adam = dmel::AdamParams{}; return DMEL_OK; }
    if (!exp_avg || !exp_avg_sq || !step) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_plan_attach_adam: NULL state pointer");
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !std::isfinite(lr) || !(lr >= 0.0) ||
        !std::isfinite(weight_decay) || !(weight_decay >= 0.0))
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_plan_attach_adam: lr / betas / eps / weight_decay out of range");
    dmel::AdamParams ap{};
    ap.param = param; ap.grad = nullptr; ap.exp_avg = exp_avg; ap.exp_avg_sq = exp_avg_sq; ap.step = step; ap.ticket = nullptr;
    ap.n = 1; ap.maximize = maximize ? 1 : 0;
    ap.lr = lr; ap.beta1 = beta1; ap.beta2 = beta2; ap.eps = eps; ap.weight_decay = weight_decay;
    plan->fused_adam = ap;
    return DMEL_OK;
}

dmel_status dmel_plan_attach_mailbox(dmel_plan* plan, dmel_mailbox* mb)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (mb && dmel::mailbox_device(mb) != plan->device) return fail(DMEL_ERR_INVALID_ARGUMENT, "the mailbox and the plan live on different devices");
    std::lock_guard<std::mutex> lock(plan->mu);
    if (plan->mailbox == mb) return DMEL_OK;
    dmel::mailbox_attach_count(plan->mailbox, -1);       // the mailbox counts its plans: dmel_mailbox_destroy refuses while any holds it
    dmel::mailbox_attach_count(mb, +1);
    plan->mailbox = mb;
    return DMEL_OK;
}

dmel_status dmel_backward(dmel_plan* plan, const float* grad_out, const float* tangent, int64_t count,
                          int32_t accumulate, float* dlambd, void* stream)
{
    return dmel_backward_ex(plan, grad_out, DMEL_DTYPE_F32, tangent, count, accumulate, dlambd, stream);
}

}  // extern "C"

namespace {
// lambd by value (lambd_dev == nullptr; n_fft derived from it) or on the device with the n_fft the caller's forward was issued for
// saved_spec != nullptr: the (batch, n_fft_dev/2+1, n_time) power spectrogram the training forward wrote (dmel_forward_dev_fixed_spec):
// no recompute, x and lambd are not touched
// d lambd computed in the launch of the filterbank gradient (dmel_backward_fb_saved_dl)
struct DotRider { const float* tangent; float* dlambd; void* scratch; };

dmel_status backward_fb_impl(dmel_plan* plan, const float* x, int32_t batch, float lambd, const float* lambd_dev, int32_t n_fft_dev,
                             uint32_t flags, const float* grad_out, const float* out, float* grad_fb, void* stream, const float* saved_spec = nullptr,
                             const DotRider* rider = nullptr)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (batch < 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "batch < 0");
    if (!grad_fb || (batch > 0 && ((!x && !saved_spec) || !grad_out))) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_fb: x / grad_out / grad_fb is NULL");
    if ((flags & DMEL_FLAG_LOG) && batch > 0 && !out)
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_fb: DMEL_FLAG_LOG needs the saved log output");
    if (!lambd_dev && !saved_spec && !std::isfinite(lambd)) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd is not finite");
    const bool full = (flags & DMEL_FLAG_FULL_WINDOW) != 0;
    if (lambd_dev && !full && (n_fft_dev < 1 || n_fft_dev > dmel::kMaxNfft || (n_fft_dev & (n_fft_dev - 1))))
        return fail(DMEL_ERR_INVALID_ARGUMENT, "n_fft must be a power of two in [1, 16384]");
    int n_over = 0, win_half = 0;
    if (full) { n_over = 2 * plan->cfg.n_points; win_half = 1; }     // any clip length: the spectrogram pass takes the chirp-z path where it has to
    const int N = n_over ? n_over : ((lambd_dev || saved_spec) ? n_fft_dev : dmel_n_fft(lambd));
    if (N > dmel::kMaxBigFft)
        return fail(DMEL_ERR_UNSUPPORTED, "n_fft = " + std::to_string(N) + " > " + std::to_string(dmel::kMaxBigFft) + " is not supported by the HIP kernels");
    const int F = N / 2 + 1, M = plan->cfg.n_mels, T = plan->T;
    std::lock_guard<std::mutex> lock(plan->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (batch == 0) { DMEL_HIP(hipMemsetAsync(grad_fb, 0, (size_t)F * M * sizeof(float), s)); return DMEL_OK; }
    {
        dmel_status so = order_after_last_stream(plan, s);       // fbw is shared by every call on the plan
        if (so != DMEL_OK) return so;
    }
    int splits = dmel::fbgrad_splits(batch, F, M, T);
    // d lambd rides along unless the plan's reducer lives in the dot kernel (mailbox) or the launch has no room for it
    Scratch rsc;
    int dot_wgs = 0, dot_vblocks = 0;
    const long long dot_count = (long long)batch * M * T;
    if (rider) {
        if (rider->scratch) rsc = carve(rider->scratch);
        else { dmel_status sr = ensure_own_scratch(plan, batch, s, &rsc); if (sr != DMEL_OK) return sr; }
        if (!plan->mailbox) {
            dot_vblocks = dmel::dot_blocks_for(dot_count, kMaxPartials);
            const int left = dmel::fbgrad_fuse_dot(F, M, splits, dot_vblocks, &dot_wgs);
            if (left > 0) splits = left; else dot_wgs = 0;
        }
    }
    const size_t spec_floats = saved_spec ? 0 : ((size_t)batch * F * T + 63) / 64 * 64;
    const size_t part_floats = ((size_t)splits * F * M + 63) / 64 * 64;
    const size_t gm_floats = 0;      // (gm = grad_out * exp(-out) is formed inside the GEMM kernel)
    const size_t need = spec_floats + part_floats + gm_floats;
    if (need > plan->fbw_floats) {
        if (is_capturing(s)) return fail(DMEL_ERR_INVALID_ARGUMENT, "workspace must grow but the stream is capturing: run one call eagerly first");
        DMEL_HIP(hipStreamSynchronize(s));
        (void)hipFree(plan->fbw); plan->fbw = nullptr; plan->fbw_floats = 0;
        DMEL_HIP(hipMalloc(&plan->fbw, need * sizeof(float)));
        plan->fbw_floats = need;
    }
    // the spectrogram the layer contracted at models.py:53 (DC removed, models.py:38): P is recomputed, not saved
    dmel_status st;
    if (saved_spec) {
        st = DMEL_OK;
    } else if (lambd_dev && full) {
        // n_fft = 2 n_points does not depend on lambd: the kernels take the window's width from the device value, nothing to check
        st = run_forward_nolock(plan, x, batch, 0.f, 0u, 0.0, plan->fbw, nullptr, dmel::kSpec, 1, stream, n_over, win_half, nullptr, lambd_dev);
    } else if (lambd_dev) {
        // lambd read on the device and checked against N: a mismatch (the forward of this step reported it) leaves NaN here too
        Scratch sc;
        if ((st = ensure_own_scratch(plan, batch, s, &sc)) != DMEL_OK) return st;
        dmel::LamArgs lam{};
        lam.dev = lambd_dev; lam.n_expected = N; lam.role = dmel::kLamFirst | dmel::kLamLast;
        bool sums_done = false;
        st = launch_forward_n(plan, x, batch, N, lam, 0u, 0.0, plan->fbw, nullptr, dmel::kSpec, 1, sc, s, 0, &sums_done);
    } else {
        st = run_forward_nolock(plan, x, batch, lambd, 0u, 0.0, plan->fbw, nullptr, dmel::kSpec, 1, stream, n_over, win_half);
    }
    if (st != DMEL_OK) return st;
    dmel::FbGradParams fp{};
    fp.spec = saved_spec ? saved_spec : plan->fbw; fp.grad_out = grad_out; fp.out = (flags & DMEL_FLAG_LOG) ? out : nullptr;
    fp.partials = plan->fbw + spec_floats; fp.grad_fb = grad_fb;
    fp.gm_ws = plan->fbw + spec_floats + part_floats;
    fp.B = batch; fp.F = F; fp.M = M; fp.T = T; fp.splits = splits;
    fp.bf16x3 = (flags & DMEL_FLAG_MFMA_BF16X3) ? 1 : 0;
    if (dot_wgs > 0) {
        fp.dot_g = grad_out; fp.dot_t = rider->tangent; fp.dot_count = dot_count; fp.dot_partials = rsc.partials; fp.dot_counter = rsc.counter;
        fp.dot_result = rider->dlambd; fp.dot_vblocks = dot_vblocks; fp.dot_wgs = dot_wgs; fp.dot_accumulate = 0;
    }
    const size_t m0 = prof_mark(plan, s);
    DMEL_HIP(dmel::launch_fbgrad(fp, s));
    if (rider && dot_wgs == 0) {
        // the stand-alone kernel (dmel_backward_scratch): the mailbox exchange lives there
        dmel::MailboxArgs mba;
        const bool use_mb = plan->mailbox != nullptr;
        if (use_mb && !dmel::mailbox_args(plan->mailbox, &mba)) return fail(DMEL_ERR_INVALID_ARGUMENT, "the plan's mailbox is not connected");
        if (use_mb) { dmel_status ms = mailbox_status(plan); if (ms != DMEL_OK) return ms; }
        DMEL_HIP(dmel::launch_dot(grad_out, 0, rider->tangent, dot_count, 0, rsc.partials, rsc.counter, kMaxPartials, rider->dlambd, s,
                                  use_mb ? &mba : nullptr));
    }
    prof_span(plan, m0, prof_mark(plan, s), 2);
    return DMEL_OK;
}
}  // namespace

extern "C" {

dmel_status dmel_backward_fb(dmel_plan* plan, const float* x, int32_t batch, float lambd, uint32_t flags,
                             const float* grad_out, const float* out, float* grad_fb, void* stream)
{
    return backward_fb_impl(plan, x, batch, lambd, nullptr, 0, flags, grad_out, out, grad_fb, stream);
}

dmel_status dmel_backward_fb_dev(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft, uint32_t flags,
                                 const float* grad_out, const float* out, float* grad_fb, void* stream)
{
    if (!lambd_dev) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd_dev is NULL");
    return backward_fb_impl(plan, x, batch, 0.f, lambd_dev, n_fft, flags, grad_out, out, grad_fb, stream);
}

dmel_status dmel_backward_fb_saved(dmel_plan* plan, const float* spec, int32_t batch, int32_t n_fft, uint32_t flags,
                                   const float* grad_out, const float* out, float* grad_fb, void* stream)
{
    if (!spec) return fail(DMEL_ERR_INVALID_ARGUMENT, "spec is NULL");
    if (flags & DMEL_FLAG_FULL_WINDOW) return fail(DMEL_ERR_UNSUPPORTED, "dmel_backward_fb_saved: not with DMEL_FLAG_FULL_WINDOW");
    if (n_fft < 2 || (n_fft & 1)) return fail(DMEL_ERR_INVALID_ARGUMENT, "n_fft must be the (even) transform length the spectrogram was computed with");
    return backward_fb_impl(plan, nullptr, batch, 0.f, nullptr, n_fft, flags, grad_out, out, grad_fb, stream, spec);
}

dmel_status dmel_backward_fb_saved_dl(dmel_plan* plan, const float* spec, int32_t batch, int32_t n_fft, uint32_t flags,
                                      const float* grad_out, const float* out, const float* tangent, float* grad_fb, float* dlambd,
                                      void* scratch, void* stream)
{
    if (!spec) return fail(DMEL_ERR_INVALID_ARGUMENT, "spec is NULL");
    if (!tangent || !dlambd) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_fb_saved_dl: tangent / dlambd is NULL");
    if (batch <= 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_fb_saved_dl: batch must be positive");
    if (flags & DMEL_FLAG_FULL_WINDOW) return fail(DMEL_ERR_UNSUPPORTED, "dmel_backward_fb_saved_dl: not with DMEL_FLAG_FULL_WINDOW");
    if (n_fft < 2 || (n_fft & 1)) return fail(DMEL_ERR_INVALID_ARGUMENT, "n_fft must be the (even) transform length the spectrogram was computed with");
    const DotRider rider{tangent, dlambd, scratch};
    return backward_fb_impl(plan, nullptr, batch, 0.f, nullptr, n_fft, flags, grad_out, out, grad_fb, stream, spec, &rider);
}

}  // extern "C"

namespace {
// gradient w.r.t. the waveform on the power-of-two transforms: the mel layer (spec_mode 0; n_over / win_half describe the
// optimized=False branch) and the spectrogram layer (spec_mode 1: grad_out is the gradient of the power spectrogram)
// lambd_dev != nullptr: lambd is read by the kernels (window tables, the wave kernel's own window); the transform length then comes
// from the caller (n_over), never from a host copy of lambd
dmel_status backward_x_impl(dmel_plan* plan, const float* x, int32_t batch, float lambd, int n_over, int win_half, int spec_mode, bool log,
                            const float* grad_out, const float* out, float* grad_x, void* stream, const float* lambd_dev = nullptr,
                            bool check_nfft = false)
{
    if (lambd_dev && n_over <= 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd on the device needs an explicit n_fft");
    const int N = n_over > 0 ? n_over : dmel_n_fft(lambd);
    const bool big = N > dmel::kMaxNfft || (N & (N - 1));              // dmel_big.hip: chirp-z / global-memory FFT, both directions
    if (N < 1 || N > dmel::kMaxBigFft || (big && (N & 1)))
        return fail(DMEL_ERR_UNSUPPORTED, "gradient w.r.t. the waveform: n_fft = " + std::to_string(N) + " (even lengths up to " +
                    std::to_string(dmel::kMaxBigFft) + ")");
    { dmel_status sd = check_device(plan); if (sd != DMEL_OK) return sd; }
    std::lock_guard<std::mutex> lock(plan->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    NfftTables* tb = nullptr;
    dmel_status st = build_tables(plan, N, &tb);
    if (st != DMEL_OK) return st;
    if ((st = order_after_last_stream(plan, s)) != DMEL_OK) return st;
    Scratch sc;
    if ((st = ensure_own_scratch(plan, batch, s, &sc)) != DMEL_OK) return st;
    dmel_plan::BigTab bt;
    const float2* big_tw = nullptr;
    if (big && (st = big_tables_for(plan, N, &bt, &big_tw)) != DMEL_OK) return st;
    // n_fft 32 ... 2048: the wave-FFT kernel leaves one overlap-added segment per tile of frames; otherwise one row per frame
    int fpt = 0;
    const int win_n = win_half ? N : N / 2 + 1;             // the plain Gaussian is symmetric about N/2 (the whole-clip window need not be)
    const bool wave_path = !big && dmel::xgrad_wave_shape(N, spec_mode ? 0 : plan->cfg.n_mels, win_n, &fpt) && std::getenv("DMEL_XGRAD_LDS") == nullptr;
    const int tiles = wave_path ? (plan->T + fpt - 1) / fpt : 0;
    const long long span = wave_path ? (long long)(fpt - 1) * plan->cfg.hop_length + N : 0;
    if (span > 0x3fffffffLL) return fail(DMEL_ERR_UNSUPPORTED, "gradient w.r.t. the waveform: hop_length too large");
    const size_t rows = wave_path ? (size_t)tiles : (size_t)plan->T, row_len = wave_path ? (size_t)span : (size_t)N;
    const size_t frame_floats = ((size_t)batch * rows * row_len + 63) / 64 * 64;
    size_t need = frame_floats + 2 * (size_t)batch * rows + 16;    // + one fp64 sum per frame / tile
    if (check_nfft && !is_capturing(s)) {
        // One call per candidate n_fft of a tracked forward (DMEL_FLAG_CHECK_NFFT): a step captured later may hold a neighbour -- 2 n_fft
        // or n_fft / 2 -- that no eager call has run yet (GraphedStep captures with the launches lambd can reach, not the ones the warm-up
        // happened to issue), and a workspace cannot grow under capture.  Size it here for the neighbours too (a re-capture
        // near an n_fft boundary raised in the middle of training).
        for (int nn : {N / 2, 2 * N}) {
            if (nn < 2 || nn > dmel::kMaxNfft || (nn & (nn - 1))) continue;
            int fq = 0;
            const bool wp = dmel::xgrad_wave_shape(nn, spec_mode ? 0 : plan->cfg.n_mels, win_half ? nn : nn / 2 + 1, &fq) && std::getenv("DMEL_XGRAD_LDS") == nullptr;
            const size_t r2 = wp ? (size_t)((plan->T + fq - 1) / fq) : (size_t)plan->T;
            const size_t l2 = wp ? (size_t)(fq - 1) * plan->cfg.hop_length + nn : (size_t)nn;
            need = std::max(need, ((size_t)batch * r2 * l2 + 63) / 64 * 64 + 2 * (size_t)batch * r2 + 16);
        }
    }
    if (need > plan->fbw_floats) {
        if (is_capturing(s)) return fail(DMEL_ERR_INVALID_ARGUMENT, "workspace must grow but the stream is capturing: run one call eagerly first");
        DMEL_HIP(hipStreamSynchronize(s));
        (void)hipFree(plan->fbw); plan->fbw = nullptr; plan->fbw_floats = 0;
        DMEL_HIP(hipMalloc(&plan->fbw, need * sizeof(float)));
        plan->fbw_floats = need;
    }
    // the big path's window table and (sequences longer than LDS) per-workgroup workspace are plan-owned, like the forward's
    const long long units = (long long)batch * ((plan->T + 1) / 2);
    const int big_grid = big ? dmel::big_grid(units, bt.M) : 0;
    if (big) {
        const size_t need_z = dmel::big_uses_global(bt.M) ? (size_t)big_grid * bt.M : 0;
        if ((size_t)N > plan->big_win_n || need_z > plan->big_z_n) {
            if (is_capturing(s)) return fail(DMEL_ERR_INVALID_ARGUMENT, "workspace must grow but the stream is capturing: run one call eagerly first");
            DMEL_HIP(hipDeviceSynchronize());
            if ((size_t)N > plan->big_win_n) {
                (void)hipFree(plan->big_win); plan->big_win = nullptr; plan->big_win_n = 0;
                DMEL_HIP(hipMalloc(&plan->big_win, (size_t)N * sizeof(float2)));
                plan->big_win_n = (size_t)N;
            }
            if (need_z > plan->big_z_n) {
                (void)hipFree(plan->big_z); plan->big_z = nullptr; plan->big_z_n = 0;
                DMEL_HIP(hipMalloc(&plan->big_z, need_z * sizeof(float2)));
                plan->big_z_n = need_z;
            }
        }
    }
    // clip sums + window table (the tangent half of the table is not used here)
    dmel::PrepParams pp{};
    pp.x = x; pp.psum = sc.psum; pp.win2 = big ? plan->big_win : sc.win;
    pp.B = batch; pp.L = plan->cfg.n_points; pp.nchunks = plan->nchunks; pp.chunk = plan->chunk;
    pp.N = N; pp.normalize = plan->cfg.normalize_window; pp.win_half = win_half;
    pp.center = win_half ? (float)((N / 2) / 2) + (float)(N / 2) / 2.0f : (float)N / 2.0f;      // as launch_forward_n
    pp.lam.dev = lambd_dev; pp.lam.val = lambd_dev ? 0.f : lambd; pp.lam.role = dmel::kLamQuiet;
    // short clips with the plain Gaussian window: the wave-FFT kernel evaluates the window and adds up its clip itself
    const bool own_prep = wave_path && !win_half && !plan->cfg.normalize_window && plan->cfg.n_points <= 32768 && std::getenv("DMEL_XGRAD_PREP") == nullptr;
    if (!own_prep) DMEL_HIP(dmel::launch_prep(pp, s));
    dmel::XgradParams xp{};
    xp.own_prep = own_prep ? 1 : 0; xp.win_denom = std::fabs(lambd) + 1e-15f; xp.lam_dev = lambd_dev;
    xp.check_nfft = (check_nfft && lambd_dev && !big) ? 1 : 0;
    xp.x = x; xp.psum = sc.psum; xp.win2 = big ? plan->big_win : sc.win; xp.tw = big ? big_tw : tb->tw_long;
    xp.chirp = big ? bt.chirp : nullptr; xp.hbr = big ? bt.hbr : nullptr; xp.zws = plan->big_z; xp.Mfft = big ? bt.M : 0; xp.logM = big ? bt.logM : 0;
    xp.fb = tb->fb_dense; xp.rowband = tb->rowband; xp.rowpk = tb->rowpk; xp.long_rows = tb->long_rows ? 1 : 0; xp.grad_out = grad_out; xp.out = log ? out : nullptr;
    xp.frames = plan->fbw; xp.grad_x = grad_x;
    xp.csum = reinterpret_cast<double*>(plan->fbw + frame_floats);      // 256-byte aligned: frame_floats is a multiple of 64
    xp.B = batch; xp.L = plan->cfg.n_points; xp.T = plan->T; xp.hop = plan->cfg.hop_length; xp.M = plan->cfg.n_mels;
    xp.nchunks = plan->nchunks; xp.N = N; xp.F = tb->F; xp.remove_dc = 1; xp.spec_mode = spec_mode;
    xp.logN = 0; while ((1 << xp.logN) < N) ++xp.logN;
    xp.inv_L = 1.0f / (float)plan->cfg.n_points;
    xp.tw1 = tb->tw1p; xp.tw2 = tb->tw2p;
    xp.win_n = win_n; xp.tiles = tiles; xp.span = (int)span; xp.tile_step = wave_path ? fpt * plan->cfg.hop_length : 0;
    const size_t m0 = prof_mark(plan, s);
    if (big) {
        DMEL_HIP(dmel::launch_xgrad_big(xp, big_grid, s));
        DMEL_HIP(dmel::launch_xgrad_gather(xp, s));
    } else {
        DMEL_HIP(dmel::launch_xgrad(xp, s));
    }
    prof_span(plan, m0, prof_mark(plan, s), 2);
    return DMEL_OK;
}
}  // namespace

extern "C" {

dmel_status dmel_backward_x(dmel_plan* plan, const float* x, int32_t batch, float lambd, uint32_t flags,
                            const float* grad_out, const float* out, float* grad_x, void* stream)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (batch < 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "batch < 0");
    if (batch == 0) return DMEL_OK;
    if (!x || !grad_out || !grad_x) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_x: x / grad_out / grad_x is NULL");
    if ((flags & DMEL_FLAG_LOG) && !out) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_x: DMEL_FLAG_LOG needs the saved log output");
    if (!std::isfinite(lambd)) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd is not finite");
    const bool full = (flags & DMEL_FLAG_FULL_WINDOW) != 0;                      // optimized=False: n_fft = 2 L, window = the clip
    return backward_x_impl(plan, x, batch, lambd, full ? 2 * plan->cfg.n_points : 0, full ? 1 : 0, 0, (flags & DMEL_FLAG_LOG) != 0,
                           grad_out, out, grad_x, stream);
}

dmel_status dmel_backward_x_spec(dmel_plan* plan, const float* x, int32_t batch, float lambd, int32_t n_fft, uint32_t flags,
                                 const float* grad_spec, float* grad_x, void* stream)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (batch < 0 || n_fft < 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "batch / n_fft < 0");
    if (batch == 0) return DMEL_OK;
    if (!x || !grad_spec || !grad_x) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_x_spec: x / grad_spec / grad_x is NULL");
    if (!(flags & DMEL_SPEC_REMOVE_DC)) return fail(DMEL_ERR_UNSUPPORTED, "dmel_backward_x_spec: only the DC-removed spectrogram (models.py:187) is differentiated");
    if (!std::isfinite(lambd)) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd is not finite");
    return backward_x_impl(plan, x, batch, lambd, n_fft, (flags & DMEL_SPEC_HALF_WINDOW) ? 1 : 0, 1, false, grad_spec, nullptr, grad_x, stream);
}

dmel_status dmel_backward_x_dev(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft, uint32_t flags,
                                const float* grad_out, const float* out, float* grad_x, void* stream)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (!lambd_dev) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd_dev is NULL");
    if (batch < 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "batch < 0");
    if (batch == 0) return DMEL_OK;
    if (!x || !grad_out || !grad_x) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_x_dev: x / grad_out / grad_x is NULL");
    if ((flags & DMEL_FLAG_LOG) && !out) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_x_dev: DMEL_FLAG_LOG needs the saved log output");
    const bool full = (flags & DMEL_FLAG_FULL_WINDOW) != 0;
    if (!full && (n_fft < 1 || n_fft > dmel::kMaxNfft || (n_fft & (n_fft - 1))))
        return fail(DMEL_ERR_INVALID_ARGUMENT, "n_fft must be a power of two in [1, 16384] (the n_fft this step's forward was issued for)");
    if ((flags & DMEL_FLAG_CHECK_NFFT) && full)
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_x_dev: DMEL_FLAG_CHECK_NFFT has no meaning with DMEL_FLAG_FULL_WINDOW (n_fft = 2 n_points)");
    return backward_x_impl(plan, x, batch, 0.f, full ? 2 * plan->cfg.n_points : n_fft, full ? 1 : 0, 0, (flags & DMEL_FLAG_LOG) != 0,
                           grad_out, out, grad_x, stream, lambd_dev, (flags & DMEL_FLAG_CHECK_NFFT) != 0);
}

dmel_status dmel_backward_x_spec_dev(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft, uint32_t flags,
                                     const float* grad_spec, float* grad_x, void* stream)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (!lambd_dev) return fail(DMEL_ERR_INVALID_ARGUMENT, "lambd_dev is NULL");
    if (batch < 0 || n_fft <= 0) return fail(DMEL_ERR_INVALID_ARGUMENT, "batch < 0 or n_fft not given");
    if (batch == 0) return DMEL_OK;
    if (!x || !grad_spec || !grad_x) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_backward_x_spec_dev: x / grad_spec / grad_x is NULL");
    if (!(flags & DMEL_SPEC_REMOVE_DC)) return fail(DMEL_ERR_UNSUPPORTED, "dmel_backward_x_spec_dev: only the DC-removed spectrogram (models.py:187) is differentiated");
    return backward_x_impl(plan, x, batch, 0.f, n_fft, (flags & DMEL_SPEC_HALF_WINDOW) ? 1 : 0, 1, false, grad_spec, nullptr, grad_x, stream, lambd_dev);
}

dmel_status dmel_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* step, uint32_t* ticket, int64_t n,
                           double lr, double beta1, double beta2, double eps, double weight_decay, int32_t maximize, void* stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_adam_step: NULL pointer");
    if (n < 0 || n > (1LL << 31)) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_adam_step: n out of range");
    // the same ranges torch.optim.Adam and dmel_amd.LambdAdam accept
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !std::isfinite(lr) || !(lr >= 0.0) ||
        !std::isfinite(weight_decay) || !(weight_decay >= 0.0))
        return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_adam_step: lr / betas / eps / weight_decay out of range");
    if (n == 0) return DMEL_OK;
    if (!ticket && dmel::adam_grid(n) > 1) return fail(DMEL_ERR_INVALID_ARGUMENT, "dmel_adam_step: more than 1024 elements need the ticket word");
    dmel::AdamParams ap{};
    ap.param = param; ap.grad = grad; ap.exp_avg = exp_avg; ap.exp_avg_sq = exp_avg_sq; ap.step = step; ap.ticket = ticket;
    ap.n = n; ap.maximize = maximize ? 1 : 0;
    ap.lr = lr; ap.beta1 = beta1; ap.beta2 = beta2; ap.eps = eps; ap.weight_decay = weight_decay;
    DMEL_HIP(dmel::launch_adam(ap, reinterpret_cast<hipStream_t>(stream)));
    return DMEL_OK;
}

dmel_status dmel_plan_set_profiling(dmel_plan* plan, int32_t enable)
{
    if (!plan) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan is NULL");
    std::lock_guard<std::mutex> lock(plan->mu);
    plan->profiling = enable != 0;
    return DMEL_OK;
}

dmel_status dmel_plan_get_profile(dmel_plan* plan, dmel_profile* profile)
{
    if (!plan || !profile) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan / profile is NULL");
    std::lock_guard<std::mutex> lock(plan->mu);
    dmel_profile pr{};
    for (const auto& sp : plan->spans) {
        DMEL_HIP(hipEventSynchronize(plan->ev_pool[sp.b]));
        float ms = 0.f;
        DMEL_HIP(hipEventElapsedTime(&ms, plan->ev_pool[sp.a], plan->ev_pool[sp.b]));
        if (sp.kind == 0) { pr.prep_ms += ms; ++pr.prep_launches; }
        else if (sp.kind == 1) { pr.fwd_ms += ms; ++pr.fwd_launches; }
        else { pr.bwd_ms += ms; ++pr.bwd_launches; }
    }
    plan->spans.clear();
    plan->ev_used = 0;
    *profile = pr;
    return DMEL_OK;
}

dmel_status dmel_plan_get_info(const dmel_plan* plan, dmel_plan_info* info)
{
    if (!plan || !info) return fail(DMEL_ERR_INVALID_ARGUMENT, "plan / info is NULL");
    *info = plan->info;
    return DMEL_OK;
}

}  // extern "C"
