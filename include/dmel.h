/*
 * dmel.h -- C ABI of libdmel_hip.so: the MI355X (gfx950) implementation of the DMEL hot path.
 *
 * The reference (johnmartinsson/differentiable-mel-spectrogram) is pure Python and has no FFI;
 * its boundary for this path is the nn.Module `MelSpectrogramLayer` (models.py:14-56) plus the
 * `torch.log(s + 1e-10)` line of the nets that wrap it (models.py:73).  Each entry point below
 * names the reference code it replaces.  Signatures carry plain pointers and sizes only (device
 * pointers are HIP device addresses, `stream` is a hipStream_t passed as void*); no torch types.
 *
 * Conventions
 *   - every function returns a dmel_status (0 = DMEL_OK); dmel_last_error() gives the message of
 *     the calling thread's last failure.
 *   - waveforms  x        : (batch, n_points) fp32, row-major, device      models.py:33
 *   - outputs    out      : (batch, 1, n_mels, n_points / hop + 1) fp32    models.py:30,36
 *   - tangent             : same shape as out, d out / d lambd (raw, signed parameter)
 *   - lambd is the trainable window std-dev in samples (models.py:19); the kernels use |lambd|
 *     (models.py:38) and n_fft = next_pow2(int(6*|lambd|)) (time_frequency.py:39,60-65).
 *   - a plan owns its device tables and a default scratch; calls that use the plan's scratch from different streams
 *     are ordered after each other by the library (an event), calls with caller scratch are independent.
 */
#ifndef DMEL_H
#define DMEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMEL_ABI_VERSION 5   /* round 6: dmel_plan_info.contraction / wl_steps; DMEL_FLAG_X_INDIRECT served by every forward kernel (round 4 = 4: *_dev variants of the fixed-length paths, saved spectrogram, DMEL_FLAG_MFMA_BF16X3, mailbox time-out, plan registry) */

typedef enum dmel_status {
    DMEL_OK = 0,
    DMEL_ERR_INVALID_ARGUMENT = 1,  /* bad shape / null pointer / negative size          */
    DMEL_ERR_UNSUPPORTED = 2,       /* a transform that needs an FFT of more than 1048576 points (|lambd| > 174762, or an
                                       optimized=False clip longer than 262144 samples) */
    DMEL_ERR_HIP = 3,               /* a HIP runtime call failed (message has the detail) */
    DMEL_ERR_NO_DEVICE = 4,         /* no gfx950 device visible                           */
    DMEL_ERR_OUT_OF_MEMORY = 5,
    DMEL_ERR_LAMBD_TRACKING = 6,    /* dmel_forward_dev: lambd changed n_fft faster than the sync-free path covers   */
    DMEL_ERR_MAILBOX_TIMEOUT = 7    /* an exchange of the plan's mailbox gave up on a rank (the gradient it returned was NaN): reported by
                                       the next forward / backward on the plan, sticky until dmel_mailbox_error reads it */
} dmel_status;

/* Constructor arguments of the layer: models.py:15-30 (MelSpectrogramLayer.__init__). */
typedef struct dmel_config {
    int32_t n_points;          /* clip length L                                   models.py:30 */
    int32_t hop_length;        /*                                                 models.py:18 */
    int32_t n_mels;            /*                                                 models.py:26 */
    int32_t sample_rate;       /*                                                 models.py:27 */
    double f_min;              /*                                                 models.py:24 */
    double f_max;              /* < 0: sample_rate / 2 (integer division)         models.py:25 */
    int32_t normalize_window;  /* time_frequency.py:25-28 (norm=True branch)                   */
    int32_t max_batch;         /* workspace is sized for this many clips (grown on demand)     */
} dmel_config;

typedef struct dmel_plan dmel_plan;

/* Flags for dmel_forward */
#define DMEL_FLAG_LOG 1u       /* fuse out = log(mel + eps)                        models.py:73 */
#define DMEL_FLAG_FULL_WINDOW 2u /* the layer's optimized=False branch: window = whole clip, n_fft = 2*n_points
                                  (time_frequency.py:41,51); any n_points (powers of two up to 8192 on the FFT kernels,
                                  everything else through the global-memory / chirp-z path)                */
#define DMEL_FLAG_OUT_BF16 4u   /* dmel_forward writes `out` as bf16 (round to nearest even of the fp32 result; BASELINE
                                  config 2 "bf16 activations / fp32 grad"): half the output bytes.  The arithmetic, the
                                  tangent and d lambd stay fp32.  The reference's output is fp32 (models.py:36).          */
#define DMEL_FLAG_MFMA_BF16X3 8u /* opt-in: dense contractions (dmel_backward_fb*: the filterbank gradient's GEMM; dmel_forward* with a
                                   caller-supplied DENSE filterbank) run on the bf16 matrix pipe as three split-bf16 products per fp32
                                   product (hi hi + lo hi + hi lo, fp32 accumulate: ~2e-5 relative, inside the 1e-4 bar; 16 x the fp32 MFMA
                                   rate).  Default: exact fp32 MFMA (v_mfma_f32_16x16x4_f32).  The HTK bank's banded contraction ignores it. */
#define DMEL_FLAG_CHECK_NFFT 16u /* dmel_backward_x_dev only: the kernels do their work only if the device value of lambd asks for exactly the
                                   `n_fft` of this call (time_frequency.py:39,60-65 evaluated on the device) and otherwise leave grad_x
                                   untouched.  For the optimized=True layer whose forward (dmel_forward_dev) issued one launch per candidate
                                   n_fft: the caller fills grad_x with NaN and issues one dmel_backward_x_dev per candidate. */
#define DMEL_FLAG_X_INDIRECT 32u /* dmel_forward_dev only (round 5): `x` is a device pointer to ONE device pointer -- the address of the batch, read by
                                   the kernel when it runs.  A step captured into a HIP graph reads static addresses; with this flag the static
                                   address is that of a pointer cell, and handing the step a new batch (train.py:25-49: one per iteration) is an
                                   8-byte write instead of a copy of the batch (dmel_amd.GraphedStep.feed, zero_copy).  Every kernel of the
                                   forward reads the cell (round 6: the direct-DFT kernel below n_fft 32 and the global-memory / chirp-z path beyond
                                   16384 too, so a lambd that drifts out of the fused kernel's range inside a captured loop is served like any
                                   other).  The batch must stay valid until the forward has executed. */
#define DMEL_DTYPE_F32 0
#define DMEL_DTYPE_BF16 1

/* ---- host-side helpers (no device needed) ------------------------------------------------- */

int32_t dmel_abi_version(void);

/* time_frequency.py:60-65 applied as at :39 -- fp32 product 6*|lambd|, int() truncation,
 * 1 << (x-1).bit_length().  Returns n_fft (>= 1). */
int32_t dmel_n_fft(float lambd);

/* time_frequency.py:21-30: Gaussian window of length n_fft centred at n_fft/2 (fp32 arithmetic as
 * the reference), optional L2 normalisation; dwindow (may be NULL) receives d window / d |lambd|. */
dmel_status dmel_window_host(float lambd, int32_t n_fft, int32_t normalize, float* window, float* dwindow);

/* models.py:42-48: torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate)
 * with norm=None, mel_scale="htk".  fb is (n_freqs, n_mels) row-major fp32, host memory. */
dmel_status dmel_mel_fbanks_host(int32_t n_freqs, double f_min, double f_max, int32_t n_mels,
                                 int32_t sample_rate, float* fb);

/* How the fused forward deals the contraction of models.py:53 over the waves of a workgroup (host arithmetic, for tests and tools):
 * `units[t]` = the non-zero extent of mel tile t (16 mels) of one group of `n_tiles` <= `waves` tiles, in units of 16 bins; wave t owns
 * tile t and takes own[t] units from its start, the rest goes in contiguous pieces to other waves -- at most one piece per wave --:
 * piece i = (piece_wave[i], piece_tile[i], piece_first[i], piece_units[i]).  `waves` is 8 or 4; arrays of 8 entries each. */
dmel_status dmel_contraction_partition_host(const int32_t* units, int32_t n_tiles, int32_t waves, int32_t* own, int32_t* n_pieces,
                                            int32_t* piece_wave, int32_t* piece_tile, int32_t* piece_first, int32_t* piece_units);

const char* dmel_last_error(void);

/* ---- device path --------------------------------------------------------------------------- */

/* Number of visible HIP devices whose architecture is gfx950 (0 when none / no driver). */
int32_t dmel_device_count(void);

/* MelSpectrogramLayer.__init__ (models.py:15-30).  Binds to the current HIP device. */
dmel_status dmel_plan_create(const dmel_config* cfg, dmel_plan** plan);
/* A plan is reference counted: dmel_plan_create returns it with one reference, dmel_plan_destroy (= dmel_plan_release) drops one,
 * and the plan is freed -- after the device has finished what was queued on it -- when the last one goes.  Whoever keeps a
 * dmel_plan* beyond the owner's lifetime takes a reference: the autograd node of torch.ops.dmel.mel_spectrogram does, so
 * `y = layer(x); del layer; y.backward(g)` is safe (a plain torch module, models.py:33-56, has no such hazard either).
 * A HIP graph that captured launches of a plan does NOT hold one by itself: keep the layer, or take a reference for the life of the
 * graph (dmel_amd.GraphedStep does: it retains the plans of its layers at every capture and releases them when the graph goes). */
dmel_status dmel_plan_destroy(dmel_plan* plan);
dmel_status dmel_plan_retain(dmel_plan* plan);
dmel_status dmel_plan_release(dmel_plan* plan);
/* 1 while `plan` is a plan of this process between dmel_plan_create and its last release, 0 for anything else (never dereferences
 * the pointer: a registry lookup).  Bindings that pass plans as integers (torch.ops.dmel.*) check this before every use. */
int32_t dmel_plan_is_live(const dmel_plan* plan);
/* depth of the pinned ring the executed forwards of a plan report into (dmel_plan_lambd_report finds execution `number` while fewer
 * than this many later ones have executed): whoever looks reports up late -- dmel_amd.GraphedStep, (max_ahead + 1) replays of
 * `steps_per_replay` forwards behind -- keeps that distance below it */
int32_t dmel_lambd_ring_size(void);
dmel_status dmel_plan_get_config(const dmel_plan* plan, dmel_config* cfg);

/* Replace the mel filterbank of the plan by a caller-supplied (n_freqs, n_mels) fp32 HOST matrix for
 * the given n_fft (n_freqs = n_fft/2+1; 1 or any even length up to 1048576): the contraction of models.py:53 then uses it instead
 * of the HTK table.  Pass fb = NULL to return to the built-in table. */
dmel_status dmel_plan_set_filterbank(dmel_plan* plan, int32_t n_fft, const float* fb);

/* The same from a DEVICE matrix (a trainable filterbank after an optimizer step): the first call for an n_fft rebuilds that
 * n_fft's tables with the dense structure (every 4x16 block multiplied; one device synchronisation), every later call is one
 * small kernel on `stream` that refreshes the values -- no host copy, no synchronisation, capturable.  Return to the built-in
 * table with dmel_plan_set_filterbank(plan, n_fft, NULL). */
dmel_status dmel_plan_set_filterbank_dev(dmel_plan* plan, int32_t n_fft, const float* fb_dev, void* stream);

/*
 * MelSpectrogramLayer.forward (models.py:33-56) [+ models.py:73 when DMEL_FLAG_LOG]:
 * DC removal, Gaussian-windowed STFT (center=True, zero padding), |.|^2, mel contraction.
 *   x        device, (batch, n_points) fp32
 *   lambd    host value of the parameter (the reference reads it to the host too,
 *            time_frequency.py:39); may be negative or zero
 *   out      device, (batch, 1, n_mels, n_time) fp32 (bf16 elements with DMEL_FLAG_OUT_BF16: pass the pointer cast)
 *   tangent  device, same shape, or NULL for inference: receives d out / d lambd so that
 *            the backward is a single dot product (one trainable scalar -> forward mode)
 *   eps      the 1e-10 of models.py:73 (ignored without DMEL_FLAG_LOG)
 * Asynchronous on `stream`.
 */
dmel_status dmel_forward(dmel_plan* plan, const float* x, int32_t batch, float lambd, uint32_t flags,
                         double eps, float* out, float* tangent, void* stream);

/*
 * The same forward with lambd left ON THE DEVICE: no device->host read, so a training step (forward, backward,
 * optimizer update of lambd) can be queued without the host ever waiting, and captured into a HIP graph.  The
 * reference reads lambd to the host once per sample to derive n_fft (time_frequency.py:39); here every kernel reads
 * the device scalar itself, derives the window from it and checks that next_pow2(int(6|lambd|)) is the n_fft it was
 * launched for.  The host picks that n_fft from what the kernels last reported (a pinned word, no synchronisation)
 * and, while lambd is within reach of a power-of-two boundary (or under graph capture), adds guard launches for the
 * neighbouring n_fft that return at once unless the work is theirs.  A forward that no launch covers (lambd crossed
 * more than the guards allow while the host was running ahead) writes NaN to its outputs and the NEXT call returns
 * DMEL_ERR_LAMBD_TRACKING once; the run-ahead is bounded (dmel_plan_set_tracking) to keep that out of reach.
 *   lambd_dev  device, 1 fp32 (e.g. the nn.Parameter's storage); read by the kernels at execution time
 *   out        device, fp32 (bf16 with DMEL_FLAG_OUT_BF16)
 *   scratch    device, dmel_scratch_bytes(plan, batch) bytes owned by the caller for this call and for the
 *              dmel_backward_scratch that consumes its tangent (no initialisation needed), or NULL to use the plan's
 *              own (then calls on different streams are serialised by the plan)
 * The first call on a plan (and the first after an error or dmel_plan_lambd_reset) reads lambd once, blocking.
 * DMEL_FLAG_FULL_WINDOW is not accepted (its n_fft does not depend on lambd: use dmel_forward_dev_fixed, which needs no tracking).
 */
size_t dmel_scratch_bytes(const dmel_plan* plan, int32_t batch);
/* dmel_forward (lambd by value) with the caller's scratch instead of the plan's */
dmel_status dmel_forward_scratch(dmel_plan* plan, const float* x, int32_t batch, float lambd, uint32_t flags,
                                 double eps, void* out, float* tangent, void* scratch, void* stream);
dmel_status dmel_forward_dev(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, uint32_t flags,
                             double eps, void* out, float* tangent, void* scratch, void* stream);
/* The same forward for callers whose n_fft is fixed by something else than lambd -- a trainable filterbank matrix
 * (dmel_plan_set_filterbank_dev) has n_fft/2+1 rows: exactly one launch for `n_fft`, lambd read and checked on the device, no
 * guard launches, no host picture of lambd (never blocks, capturable from the first call).  If the device value asks for
 * another n_fft the outputs are NaN and the NEXT call returns DMEL_ERR_LAMBD_TRACKING (models.py:42-48 ties the reference's
 * matrix to the n_fft of the forward in the same way: torch.matmul fails on the shape).
 * With DMEL_FLAG_FULL_WINDOW (the layer's optimized=False branch, time_frequency.py:41,51) the transform is n_fft = 2 n_points
 * whatever lambd is: `n_fft` is ignored, nothing is checked, lambd only shapes the window the kernels build from the device value. */
dmel_status dmel_forward_dev_fixed(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft,
                                   uint32_t flags, double eps, void* out, float* tangent, void* scratch, void* stream);
/* dmel_forward_dev_fixed that ALSO writes the power spectrogram the contraction consumed -- spec: device, (batch, n_fft/2+1, n_time) fp32,
 * the layout of time_frequency.py:53 -- for dmel_backward_fb_saved: the filterbank gradient then skips its recompute of the spectrogram
 * (16.5 us of 43 at BASELINE config 2, for 16.8 MB more stored here).  Training forward only (tangent != NULL), power-of-two n_fft
 * from 32 to 16384, not with DMEL_FLAG_FULL_WINDOW; spec = NULL is plain dmel_forward_dev_fixed. */
dmel_status dmel_forward_dev_fixed_spec(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft,
                                        uint32_t flags, double eps, void* out, float* tangent, float* spec, void* scratch, void* stream);

/* Every forward that EXECUTES on a plan (an eager dmel_forward_dev, or one replay of a captured one) draws the next
 * EXECUTION NUMBER from a device counter and reports (number, lambd as it read it) into a pinned ring of 64 entries. */
typedef struct dmel_lambd_status {
    int32_t known;            /* 0 until a value has been seen                                             */
    float lambd_seen;         /* lambd as read by the most recent forward that has EXECUTED                */
    int32_t n_fft_seen;
    uint32_t seq_issued;      /* execution number the host expects its most recent eager call to draw (caught up with the
                                 reports: replays of captured forwards execute without the host counting them)            */
    uint32_t seq_seen;        /* execution number lambd_seen belongs to                                    */
    float rate;               /* decayed maximum of |change of lambd| per call                             */
    int32_t guards;           /* most recent call: bit 0 = n_fft/2 guard launched, bit 1 = 2 n_fft guard   */
    int32_t error;            /* 1: a forward was not covered (reported by the next dmel_forward_dev)      */
    uint32_t error_seq;
    float error_lambd;
    int32_t next_n_fft;       /* what a dmel_forward_dev issued now would launch for, and guard (bits as `guards`)        */
    int32_t next_guards;
    uint32_t calls;           /* dmel_forward_dev / _fixed calls on this plan so far, captured ones included               */
} dmel_lambd_status;
dmel_status dmel_plan_lambd_status(dmel_plan* plan, dmel_lambd_status* status);
/* lambd as read by execution `number`, if its report is still in the ring (found = 1).  Unlike dmel_plan_lambd_status, whose
 * picture is "whatever executed last when the host looked", this answer does not depend on timing: ranks of a data-parallel
 * job that issue the same forwards in the same order read the same value for the same number, and can therefore take the
 * same re-capture decision without a collective (dmel_amd.graph.GraphedStep). */
dmel_status dmel_plan_lambd_report(dmel_plan* plan, uint32_t number, float* lambd, int32_t* found);
/* The launch choice as a pure function: n_fft of `lambd` (time_frequency.py:39,60-65) and the neighbouring n_fft a forward
 * must guard (bit 0: n_fft / 2, bit 1: 2 n_fft) when lambd may move by up to `rate` per forward for `stale_forwards` forwards
 * before anybody looks again. */
dmel_status dmel_decide_launch(float lambd, float rate, float stale_forwards, int32_t* n_fft, int32_t* guards);
/* While n_fft > 0, dmel_forward_dev launches exactly for `n_fft` plus the guards given (bits as above) instead of choosing
 * from its own picture; the kernels still check the device value and poison + report a forward nothing covered.  For callers
 * that decide themselves (a captured step that must hold the same launches on every rank).  n_fft = 0: back to automatic. */
dmel_status dmel_plan_force_launch(dmel_plan* plan, int32_t n_fft, int32_t guards);
/* max_ahead: calls the host may be ahead of the last observation before dmel_forward_dev waits (0 = unbounded, default 8);
 * guard_mode: 0 = near boundaries only, and both neighbours whenever the stream is capturing (default: a captured forward
 * is replayed without the host looking); 1 = always both neighbours; 2 = never; 3 = near boundaries only, also under
 * capture -- for callers that watch dmel_plan_lambd_status between replays and re-capture when next_n_fft / next_guards
 * differ from what the graph holds (dmel_amd.graph.GraphedStep does; max_ahead is then the replays it lets queue up) */
dmel_status dmel_plan_set_tracking(dmel_plan* plan, int32_t max_ahead, int32_t guard_mode);
/* forget what was seen (after lambd was rewritten from outside, e.g. load_state_dict): the next call reads it again */
dmel_status dmel_plan_lambd_reset(dmel_plan* plan);

/*
 * Backward to lambd.grad (what loss.backward() reaches through the layer, train.py:47):
 *   dlambd[0] = sum_i grad_out[i] * tangent[i]        (accumulate != 0: += instead of =)
 * grad_out, tangent: device fp32, `count` elements; dlambd: device fp32 scalar.
 * Deterministic (fixed-order two-stage reduction, fp64 accumulation).  Asynchronous on `stream`.
 */
dmel_status dmel_backward(dmel_plan* plan, const float* grad_out, const float* tangent, int64_t count,
                          int32_t accumulate, float* dlambd, void* stream);

/* dmel_backward for a gradient tensor of another element type: grad_dtype = DMEL_DTYPE_F32 or DMEL_DTYPE_BF16 (the
 * gradient of a DMEL_FLAG_OUT_BF16 output; widened exactly, accumulated in fp64 like the fp32 path). */
dmel_status dmel_backward_ex(dmel_plan* plan, const void* grad_out, int32_t grad_dtype, const float* tangent, int64_t count,
                             int32_t accumulate, float* dlambd, void* stream);
/* dmel_backward_ex with the reduction's partials and ticket in the caller's `scratch` (the block given to the
 * dmel_forward_dev that produced `tangent`, which also zeroed the ticket): nothing plan-owned is touched, so any number of
 * streams may run steps of one plan concurrently.  scratch = NULL: the plan's own. */
dmel_status dmel_backward_scratch(dmel_plan* plan, const void* grad_out, int32_t grad_dtype, const float* tangent, int64_t count,
                                  int32_t accumulate, float* dlambd, void* scratch, void* stream);

/*
 * Backward to the filterbank matrix ("mel params"): the adjoint of the contraction at models.py:53,
 * `torch.matmul(spectrogram, mel_fb)` with mel_fb (n_freqs, n_mels) from models.py:42-48.  The reference keeps
 * mel_fb constant; this is what torch autograd returns when it is made a leaf:
 *   grad_fb[f][m] = sum_{b,t} spec[b][f][t] * gm[b][m][t]
 *   gm = grad_out                 without DMEL_FLAG_LOG
 *   gm = grad_out * exp(-out)     with DMEL_FLAG_LOG (out = the saved log output of dmel_forward; models.py:73)
 * The spectrogram is recomputed from x (same lambd and flags as the forward), not saved.
 *   x         device, (batch, n_points) fp32       grad_out  device, (batch, 1, n_mels, n_time) fp32
 *   out       device, same shape as grad_out, or NULL without DMEL_FLAG_LOG
 *   grad_fb   device, (n_fft/2+1, n_mels) fp32, overwritten
 * flags: DMEL_FLAG_LOG, DMEL_FLAG_FULL_WINDOW as in dmel_forward; any transform length the forward accepts (the spectrogram pass
 * takes the chirp-z path for lengths that are not powers of two).  Deterministic (fixed-order sum over batch
 * slices, exact-fp32 MFMA).  Asynchronous on `stream`; uses a plan-owned workspace, so calls on one plan must be
 * issued on one stream at a time.
 */
dmel_status dmel_backward_fb(dmel_plan* plan, const float* x, int32_t batch, float lambd, uint32_t flags,
                             const float* grad_out, const float* out, float* grad_fb, void* stream);
/* dmel_backward_fb with lambd read on the device (the spectrogram recompute checks it against `n_fft`, the one the forward
 * of this step was issued for: dmel_forward_dev_fixed); no host read, capturable.  With DMEL_FLAG_FULL_WINDOW `n_fft` is ignored
 * (2 n_points) and nothing is checked. */
dmel_status dmel_backward_fb_dev(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft, uint32_t flags,
                                 const float* grad_out, const float* out, float* grad_fb, void* stream);
/* dmel_backward_fb on the spectrogram saved by dmel_forward_dev_fixed_spec (same batch, n_fft, plan): the GEMM and the ordered sum of
 * its slices only.  flags: DMEL_FLAG_LOG (then `out` = the saved log output), DMEL_FLAG_MFMA_BF16X3. */
dmel_status dmel_backward_fb_saved(dmel_plan* plan, const float* spec, int32_t batch, int32_t n_fft, uint32_t flags,
                                   const float* grad_out, const float* out, float* grad_fb, void* stream);
/* dmel_backward_fb_saved and dmel_backward_scratch (d lambd = sum(grad_out * tangent), fp32 grad_out, written to `dlambd`, not
 * accumulated) in ONE launch: a few extra workgroups of the gradient's GEMM kernel play the dot kernel's blocks -- the same partition
 * and order of additions, the same bits -- so a trainable-filterbank step (models.py:53 with learnable_fb) issues one kernel less.
 * `scratch`: the dmel_scratch_bytes area the step's forward was given (NULL: the plan's own).  A plan whose reducer lives in the dot
 * kernel (dmel_plan_attach_mailbox), or a launch without room for the extra workgroups, runs the two kernels one after the other:
 * same results. */
dmel_status dmel_backward_fb_saved_dl(dmel_plan* plan, const float* spec, int32_t batch, int32_t n_fft, uint32_t flags,
                                      const float* grad_out, const float* out, const float* tangent, float* grad_fb, float* dlambd,
                                      void* scratch, void* stream);

/*
 * Backward to the waveform: what torch autograd returns for x.requires_grad through models.py:38 (DC removal),
 * time_frequency.py:43-53 (zero padding, framing, window, rfft, |.|^2), models.py:53 (mel contraction) and, with
 * DMEL_FLAG_LOG, models.py:73.  The reference never differentiates the waveform; provided for completeness
 * (adversarial / saliency uses).  Arguments as dmel_backward_fb; grad_x: device, (batch, n_points) fp32, overwritten.
 * Every transform the forward runs: the wave-FFT kernels up to n_fft 2048, LDS transforms up to 16384, and beyond that -- or for
 * lengths that are not powers of two, i.e. the optimized=False branch (DMEL_FLAG_FULL_WINDOW) on arbitrary clips -- the
 * global-memory FFT / chirp-z transforms in both directions (a correctness path: one (batch, n_time, n_fft) fp32 workspace).
 * Deterministic (overlap-add as an ordered gather).  Asynchronous on `stream`; shares the plan-owned workspace with dmel_backward_fb.
 */
dmel_status dmel_backward_x(dmel_plan* plan, const float* x, int32_t batch, float lambd, uint32_t flags,
                            const float* grad_out, const float* out, float* grad_x, void* stream);
/* The same through SpectrogramLayer.forward (models.py:171-200; dmel_spectrogram_ex with DMEL_SPEC_REMOVE_DC): grad_spec is the
 * gradient of the power spectrogram, device (batch, n_fft/2+1, n_time) fp32; n_fft and flags as in dmel_spectrogram_ex (any
 * even n_fft the forward accepts). */
dmel_status dmel_backward_x_spec(dmel_plan* plan, const float* x, int32_t batch, float lambd, int32_t n_fft, uint32_t flags,
                                 const float* grad_spec, float* grad_x, void* stream);
/* Both with lambd read on the device (no host read, capturable): `n_fft` is the transform length the forward of this step ran --
 * fixed by a trainable filterbank (dmel_forward_dev_fixed), 2 n_points with DMEL_FLAG_FULL_WINDOW (then `n_fft` is ignored), or the
 * spectrogram layer's explicit length.  A lambd that has left `n_fft` made that forward return NaN and raise its error; the
 * gradient computed here for it is not meaningful either.  With DMEL_FLAG_CHECK_NFFT (dmel_backward_x_dev, n_fft <= 16384, not with
 * DMEL_FLAG_FULL_WINDOW) the call is a no-op on the device unless lambd asks for `n_fft`: the sync-free backward of the
 * optimized=True layer, one call per n_fft its forward launched for (dmel_plan_get_info's n_fft and dmel_lambd_status::guards
 * read right after dmel_forward_dev). */
dmel_status dmel_backward_x_dev(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft, uint32_t flags,
                                const float* grad_out, const float* out, float* grad_x, void* stream);
dmel_status dmel_backward_x_spec_dev(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft, uint32_t flags,
                                     const float* grad_spec, float* grad_x, void* stream);

/* Power spectrogram only, (batch, n_fft/2+1, n_time) fp32 = time_frequency.differentiable_spectrogram
 * (time_frequency.py:32-58, optimized branch) applied per clip; remove_dc != 0 adds models.py:38. */
dmel_status dmel_spectrogram(dmel_plan* plan, const float* x, int32_t batch, float lambd,
                             int32_t remove_dc, float* spec, void* stream);

/*
 * SpectrogramLayer.forward (models.py:171-200) = time_frequency.differentiable_spectrogram per clip
 * (time_frequency.py:32-58) with the tangent d spec / d lambd for its backward.
 *   n_fft   0: derive from lambd (optimized branch, :39); otherwise the transform length to use --
 *           the non-optimized branch (:41,:51) is n_fft = 2 * n_points with DMEL_SPEC_HALF_WINDOW
 *           (torch.stft zero-pads the win_length = n_points window to n_fft on both sides).
 *           Any even length (powers of two up to 16384 on the LDS kernels).
 *   spec, tangent   device, (batch, n_fft/2+1, n_time) fp32; tangent may be NULL.
 */
#define DMEL_SPEC_REMOVE_DC 1u      /* models.py:187: x[idx] - mean(x[idx])                          */
#define DMEL_SPEC_HALF_WINDOW 2u    /* window support = middle half of n_fft                          */
dmel_status dmel_spectrogram_ex(dmel_plan* plan, const float* x, int32_t batch, float lambd, int32_t n_fft,
                                uint32_t flags, float* spec, float* tangent, void* stream);
/* the same with lambd read on the device; n_fft must be given (> 0: a length that does not depend on lambd, e.g. the reference's
 * DSPEC configuration optimized=False, search_spaces.py:71-91): no host read, capturable */
dmel_status dmel_spectrogram_ex_dev(dmel_plan* plan, const float* x, int32_t batch, const float* lambd_dev, int32_t n_fft,
                                    uint32_t flags, float* spec, float* tangent, void* stream);

/* ---- the one exchange step: all-reduce of lambd.grad across GPUs (RCCL over xGMI) -------------------
 * The reference has no distributed code; with the batch sharded over one process per GPU the only cross-GPU
 * datum of this path is the scalar gradient.  These calls issue ncclAllReduce natively on the
 * communicator's own stream, ordered against the caller's stream by events (no host blocking, ~5 us of host
 * time), so that the collective of step k overlaps step k+1.  RCCL is dlopen'ed at first use. */
#define DMEL_COMM_ID_BYTES 128
typedef struct dmel_comm dmel_comm;
/* rank 0 creates the id and ships it to the other ranks by any means (dmel_amd.dist uses torch.distributed) */
dmel_status dmel_comm_unique_id(uint8_t id[DMEL_COMM_ID_BYTES]);
/* collective: every rank calls it with the same id; binds to the current HIP device */
dmel_status dmel_comm_create(const uint8_t id[DMEL_COMM_ID_BYTES], int32_t rank, int32_t world, dmel_comm** comm);
dmel_status dmel_comm_destroy(dmel_comm* comm);
/* in-place SUM of `count` fp32 at device address `buf`, after everything already queued on `stream`;
 * `ticket` (0..63, a ring) names the operation for dmel_comm_wait */
dmel_status dmel_comm_allreduce_async(dmel_comm* comm, float* buf, int32_t count, void* stream, int32_t* ticket);
/* the same all-reduce issued IN `stream` (ordered like a kernel launch: what a step needs when the optimizer update that
 * follows consumes the reduced gradient; capturable into a HIP graph together with the step) */
dmel_status dmel_comm_allreduce(dmel_comm* comm, float* buf, int32_t count, void* stream);
/* make `stream` wait for the all-reduce `ticket`; the host does not block */
dmel_status dmel_comm_wait(dmel_comm* comm, int32_t ticket, void* stream);

/* ---- the same exchange without RCCL: a peer-to-peer mailbox folded into the backward's own kernel ---------------------------
 * ncclAllReduce of 4 bytes costs a kernel of its own and the small-message latency of its ring on the critical path of a
 * ~35 us step (the optimizer update needs the reduced gradient).  With a mailbox the workgroup of dmel_backward's dot kernel
 * that finishes last stores (step, local sum) as one 8-byte granule straight into every rank's inbox (peer memory over xGMI,
 * system-scope stores), polls its own inbox for the other ranks' granules of the same step and adds them in rank order: no
 * extra launch, no ring, the same fp32 result on every rank.  Opt-in; RCCL (dmel_comm_*) stays the default.  Every wait is
 * bounded by wall-clock time (default 120 s, dmel_mailbox_set_timeout_ms): an ordinary straggler is waited for; a rank that never
 * arrives makes the result NaN on the ranks that waited and raises a sticky error word, and the NEXT dmel_forward* / dmel_backward*
 * on a plan the mailbox is attached to returns DMEL_ERR_MAILBOX_TIMEOUT (until dmel_mailbox_error has read the word) instead of
 * hanging the device.
 *   1. every rank: dmel_mailbox_create (allocates its inbox on the current device, returns a 64-byte IPC handle)
 *   2. the handles of all ranks, in rank order, travel by any means (dmel_amd.dist uses torch.distributed) to
 *      dmel_mailbox_connect, which maps the peers' inboxes (hipIpcOpenMemHandle; a handle created by THIS process -- several
 *      ranks of one process, one per device -- is looked up in a process-local registry and its pointer used directly, with
 *      peer access enabled between the two devices)
 *   3. dmel_plan_attach_mailbox(plan, mb): from then on dmel_backward / dmel_backward_scratch on that plan return the SUM over
 *      ranks (every rank must call them the same number of times, as with any collective); mb = NULL detaches.  The mailbox
 *      counts the plans it is attached to: dmel_mailbox_destroy refuses while that count is not zero (a plan that is released
 *      detaches itself).
 *      dmel_mailbox_allreduce does the same exchange for a value already in memory (one tiny launch on `stream`). */
#define DMEL_MAILBOX_HANDLE_BYTES 64
#define DMEL_MAILBOX_MAX_WORLD 16
typedef struct dmel_mailbox dmel_mailbox;
dmel_status dmel_mailbox_create(int32_t rank, int32_t world, dmel_mailbox** mb, uint8_t handle[DMEL_MAILBOX_HANDLE_BYTES]);
dmel_status dmel_mailbox_connect(dmel_mailbox* mb, const uint8_t* handles /* world x DMEL_MAILBOX_HANDLE_BYTES, rank order */);
dmel_status dmel_mailbox_destroy(dmel_mailbox* mb);
dmel_status dmel_mailbox_allreduce(dmel_mailbox* mb, float* buf, void* stream);
/* 0 = no exchange has timed out; otherwise *step / *missing_rank name the first one that did (sticky until read) */
dmel_status dmel_mailbox_error(dmel_mailbox* mb, int32_t* failed, uint32_t* step, int32_t* missing_rank);
/* polls per source rank before an exchange gives up; 0 (the default) = no limit on the count, the wall-clock bound holds */
dmel_status dmel_mailbox_set_spin_limit(dmel_mailbox* mb, uint32_t polls);
/* wall-clock bound of one exchange in milliseconds (default 120 000; 0 = wait for ever, as RCCL does) */
dmel_status dmel_mailbox_set_timeout_ms(dmel_mailbox* mb, uint64_t milliseconds);
dmel_status dmel_plan_attach_mailbox(dmel_plan* plan, dmel_mailbox* mb);

/* torch.optim.Adam's update of an fp32 parameter of the layer on the device (main.py:52-53 builds that optimizer; lambd is one
 * scalar, the trainable filterbank a (n_fft/2+1, n_mels) matrix) as ONE launch on `stream`: param, grad, exp_avg, exp_avg_sq are `n`
 * device floats, `step` one device float that counts the updates (starts at 0; torch's capturable Adam keeps it the same way),
 * `ticket` one zero-initialised device word the launch leaves at zero (needed when n > 1024: several workgroups; may be NULL
 * below).  No host synchronisation, capturable.  The hyper-parameters are doubles, as torch hands them to its own kernel
 * (1 - beta is formed in fp64); the state is fp32.  Arithmetic:
 *   step += 1; g = -grad if maximize; g += weight_decay * param; m += (1 - beta1) (g - m); v = beta2 v + (1 - beta2) g^2;
 *   param -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * Opt-in: torch's own optimizer keeps working on the layer's parameters. */
dmel_status dmel_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* step, uint32_t* ticket, int64_t n,
                           double lr, double beta1, double beta2, double eps, double weight_decay, int32_t maximize, void* stream);

/* Adam fused into the backward (round 5; opt-in).  While attached, every dmel_backward / dmel_backward_ex / dmel_backward_scratch on
 * the plan ends with dmel_adam_step's update of ONE fp32 device scalar -- `param`, i.e. lambd -- by the gradient the call has just
 * written to `dlambd` (after the mailbox all-reduce, if one is attached): the workgroup that finishes the dot product applies it, so
 * the step of train.py:47-49 needs no optimizer launch for lambd.  Same arithmetic and state (exp_avg, exp_avg_sq, step: device
 * floats, as dmel_adam_step) bit for bit.  Meant for steps with one backward per update and no other gradient of the layer read
 * after it (no trainable filterbank, no waveform gradient: their kernels read lambd).  param == NULL detaches.  The pointers must
 * stay valid while attached; the hyper-parameters are taken by value at every call (a captured graph holds those of its capture). */
dmel_status dmel_plan_attach_adam(dmel_plan* plan, float* param, float* exp_avg, float* exp_avg_sq, float* step,
                                  double lr, double beta1, double beta2, double eps, double weight_decay, int32_t maximize);

/* Introspection for tests / benchmarks */
typedef struct dmel_plan_info {
    int32_t n_fft;             /* of the most recent forward                                */
    int32_t n_freqs;
    int32_t n_time;
    int32_t frames_per_tile;   /* frames one workgroup of the fused kernel produces         */
    int32_t grid_fwd;          /* workgroups of the fused forward kernel                    */
    int32_t fb_blocks;         /* non-zero 4x16 filterbank blocks fed to the MFMA loop      */
    int32_t fb_blocks_dense;   /* the same count for a dense matrix                         */
    int32_t lds_bytes;         /* dynamic LDS of the fused kernel                           */
    int32_t kernel_path;       /* 0 = wave-FFT + MFMA kernel (32 <= n_fft <= 16384), 1 = direct-DFT kernel (n_fft < 32), 3 = global-memory FFT / chirp-z (dmel_big.hip) */
    /* ABI 5 (round 6): which contraction the most recent fused forward ran, and its size -- what an MFMA-flop count has to be taken from */
    int32_t contraction;       /* 0 = banded / dense 16x16x4 fp32 tiles (fb_blocks of them per 16-row tile), 1 = wave-local 4x4x1 fp32 (kTrainW:
                                  wl_steps instructions per wave, 16 blocks of 4 x 4 each), 2 = dense bf16x3 (kTrainH), -1 = no MFMA stage */
    int32_t wl_steps;          /* contraction 1: v_mfma_f32_4x4x1 instructions every wave issues per tile (all phases)     */
} dmel_plan_info;
dmel_status dmel_plan_get_info(const dmel_plan* plan, dmel_plan_info* info);

/* Per-kernel device timing with HIP events recorded on the caller's stream around each launch
 * (bench.py's roofline line).  Off by default; recording stops silently after 16384 launches. */
typedef struct dmel_profile {
    double prep_ms;            /* sum over launches: partial sums + window tables kernel    */
    double fwd_ms;             /* fused forward kernel                                      */
    double bwd_ms;             /* both kernels of dmel_backward                             */
    int32_t prep_launches, fwd_launches, bwd_launches;
} dmel_profile;
dmel_status dmel_plan_set_profiling(dmel_plan* plan, int32_t enable);
/* Waits for the recorded events, returns the sums since the last call and resets them. */
dmel_status dmel_plan_get_profile(dmel_plan* plan, dmel_profile* profile);

#ifdef __cplusplus
}
#endif
#endif /* DMEL_H */
