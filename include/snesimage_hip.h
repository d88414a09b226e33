/* include/snesimage_hip.h — C ABI of libsnesimage_hip.so, the MI355X (gfx950) implementation of
 * snesimage's palette-optimizer hot path.
 *
 * The reference (aexoden/snesimage, Rust) has no FFI or plugin seam: the hot path is the private
 * method set of `struct OptimizedImage` (/root/reference/src/lib.rs:33-626) called from `run()`
 * (lib.rs:851, 893-910, 988, 1002, 1021).  This header defines that seam as a C ABI; each entry
 * point names the reference method it replaces.  A Rust host binds it with an `extern "C"` block
 * (INTEGRATION.md).  Conventions:
 *   - every function returns 0 on success and a negative code on failure; the message is available
 *     from snesimage_last_error() (thread-local, owned by the library) — the C analogue of the
 *     reference's anyhow::Result + context strings (lib.rs:82, 186, 212 ...; main.rs:16-19);
 *   - snesimage_ctx owns all device memory (the reference struct owns its Vecs, lib.rs:33-43);
 *     the caller owns every pointer it passes; host pointers are copied during the call;
 *   - a ctx is bound to one HIP device and one stream and is not thread-safe (the reference is
 *     single-threaded); independent contexts may be used from different threads;
 *   - there is no CPU fallback: creation fails if no HIP device is usable.
 * Colours cross the boundary as raw 5-bit (r,g,b) byte triples — `SnesColor.data` (lib.rs:629-638)
 * — or as BGR555 words where the reference itself emits them (`as_u16`, lib.rs:679-681).
 */
#ifndef SNESIMAGE_HIP_H
#define SNESIMAGE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct snesimage_ctx snesimage_ctx;

/* config.rs:20-30 (--dither, --perceptual-palettes, --nes) */
enum { SNES_DITHER = 1, SNES_PERCEPTUAL = 2, SNES_NES = 4 };
/* optimizer methods: lib.rs:191 (random), :286 (channel), :242 (nes) */
enum { SNES_METHOD_RANDOM = 0, SNES_METHOD_CHANNEL = 1, SNES_METHOD_NES = 2 };

enum {
    SNES_OK = 0,
    SNES_ERR_ARG = -1,       /* bad argument (sizes, null pointers, slot out of range) */
    SNES_ERR_HIP = -2,       /* a HIP runtime call failed */
    SNES_ERR_STATE = -3,     /* call not valid in the current state */
    SNES_ERR_KMEANS = -4,    /* cogset precondition 2 <= k < n violated (the reference panics) */
    SNES_ERR_UNSUPPORTED = -5
};

/* OptimizedImage::new — lib.rs:46-65.  rgba: w*h*4 bytes, row-major RGBA8.  w must be 256 (tile
 * stride hard-coded to 32, lib.rs:58,565); h a power of two in [8,256].  sub_count*sub_size <= 253.
 * device: HIP device ordinal (>= 0). */
int32_t snesimage_create(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t sub_count,
                         uint32_t sub_size, uint32_t flags, int32_t device, snesimage_ctx **out);
void snesimage_destroy(snesimage_ctx *ctx);

/* Plumbing: run all subsequent work on an existing hipStream_t (e.g. torch's current stream).
 * NULL restores the context's own stream. */
int32_t snesimage_set_stream(snesimage_ctx *ctx, void *hip_stream);
/* Block until all work queued by this context has finished. */
int32_t snesimage_sync(snesimage_ctx *ctx);
/* Most candidates one internal launch group takes (bounds the workspace, which grows on demand);
 * default 4096.  A candidate list is split evenly over the context's launch lanes up to this bound. */
int32_t snesimage_set_chunk(snesimage_ctx *ctx, uint32_t chunk);

int32_t snesimage_initialize_tiles(snesimage_ctx *ctx);     /* lib.rs:79-189  */
int32_t snesimage_recalculate_palettes(snesimage_ctx *ctx); /* lib.rs:407-415 */
int32_t snesimage_optimize(snesimage_ctx *ctx);             /* lib.rs:425-501 */
int32_t snesimage_error(snesimage_ctx *ctx, double *out);   /* lib.rs:503-548 */

/* The loop body of lib.rs:205-220 / 252-262 / 296-306 for an explicit candidate list: for each
 * k < n, entry (palette,index) := rgb5[3k..3k+2], optimize(), error() -> errors[k].  The context's
 * palette and palette_map are left unchanged.  Host pointers; synchronous. */
int32_t snesimage_score_candidates(snesimage_ctx *ctx, uint32_t palette, uint32_t index,
                                   const uint8_t *rgb5, uint32_t n, double *errors);
/* Same with device pointers, asynchronous on the context's stream (no host synchronisation).
 * maps_out (optional, device, n*w*h bytes) receives each candidate's palette_map. */
int32_t snesimage_score_candidates_device(snesimage_ctx *ctx, uint32_t palette, uint32_t index,
                                          const uint8_t *d_rgb5, uint32_t n, double *d_errors,
                                          uint8_t *d_maps_out);
/* The remap alone — the optimize() of lib.rs:210-213 for each candidate, without its error():
 * d_maps_out (device, n*w*h bytes) receives the palette_map each candidate would produce.
 * Device pointers, asynchronous on the context's stream. */
int32_t snesimage_remap_candidates_device(snesimage_ctx *ctx, uint32_t palette, uint32_t index,
                                          const uint8_t *d_rgb5, uint32_t n, uint8_t *d_maps_out);

/* One optimizer call — optimize_palette_entry_{random,channel,nes} (lib.rs:191-328) followed by
 * lib.rs:906-910.  Random candidates come from the counter RNG keyed (seed, step_id);
 * n_random = 0 means the reference's 64 (lib.rs:205).  best_error / best_rgb5 may be NULL. */
int32_t snesimage_step(snesimage_ctx *ctx, uint32_t method, uint32_t palette, uint32_t index,
                       uint32_t channel, uint64_t seed, uint64_t step_id, uint32_t n_random,
                       double *best_error, uint8_t *best_rgb5);
/* The same without any host synchronisation; results via snesimage_last_step(). */
int32_t snesimage_step_async(snesimage_ctx *ctx, uint32_t method, uint32_t palette, uint32_t index,
                             uint32_t channel, uint64_t seed, uint64_t step_id, uint32_t n_random);
int32_t snesimage_last_step(snesimage_ctx *ctx, double *best_error, uint8_t *best_rgb5,
                            int32_t *best_k);

/* Split-phase step for candidate sharding across GPUs.  Phase 1 scores the
 * candidates k with k % shard_count == shard_rank of the n_total candidates of this call and
 * writes d_errors[k] (device, n_total doubles; +inf for candidates owned by other ranks).  The
 * caller min-all-reduces d_errors across ranks (RCCL), then phase 2 applies the reference's
 * acceptance rule (strict <, ascending k, lib.rs:216-219) identically on every rank. */
int32_t snesimage_step_begin(snesimage_ctx *ctx, uint32_t method, uint32_t palette, uint32_t index,
                             uint32_t channel, uint64_t seed, uint64_t step_id, uint32_t n_total,
                             uint32_t shard_rank, uint32_t shard_count, double *d_errors);
int32_t snesimage_step_commit(snesimage_ctx *ctx, const double *d_errors);

/* Speculative multi-slot stepping — the reference's own loop (lib.rs:888-933: one call of optimize_palette_entry_random
 * with 64 candidates, lib.rs:191-240, or _channel with 32, lib.rs:286-328, or _nes with 56, lib.rs:242-284, per scheduler
 * slot), several calls per launch.  Call j+1 of that loop sees the state call j saw whenever call j accepted nothing
 * (strict <, lib.rs:216-219), so a *window* scores the next K calls of the schedule against the current palette in one set
 * of launches, applies their decisions in order and stops behind the first call that changed the state; the calls behind
 * it are void and are scored again by the next window.  Bit-identical to snesimage_schedule_next + snesimage_step per
 * call, for every K.  Windows cover the group-sparse path (every height since round 4; with --dither, subpalettes of two entries and more);
 * what is left — one-entry subpalettes with --dither, a library run with SNES_SPARSE=0 — is stepped call by call inside snesimage_run_slots. */
typedef struct { double error; int32_t best_k; uint8_t rgb5[3]; uint8_t changed; } snesimage_call_result; /* what snesimage_last_step reports after the call */
typedef struct {
    uint32_t calls, accepted, windows, voided;   /* calls that took effect; calls that changed the palette; launch sets collected;
                                                  * launch sets enqueued ahead and voided on the device (they scored nothing) */
    uint64_t scored, useful;                     /* candidates scored in all; candidates of the calls that took effect */
} snesimage_run_stats;
/* n_calls calls from scheduler state (*palette, *index, *channel, *step) — advanced as by snesimage_schedule_next — call j
 * drawing its random candidates from stream (seed, first_step_id + j); n_random = 0: the reference's 64 (lib.rs:205), at most 64
 * in a window (larger calls are stepped one by one).  window: calls per launch set (0 = adaptive: the size that
 * gets most calls through per unit of time at the acceptance rate of the recent windows, at most SNES_WINDOW_MAX, default 64;
 * 1 = call by call).
 * log (optional): n_calls records.  stats (optional). */
int32_t snesimage_run_slots(snesimage_ctx *ctx, uint32_t n_calls, uint64_t seed, uint64_t first_step_id, uint32_t *palette,
                            uint32_t *index, uint32_t *channel, uint32_t *step, uint32_t n_random, uint32_t window,
                            snesimage_call_result *log, snesimage_run_stats *stats);
/* Allocate the storage of windows of up to n_slots calls now instead of on first use (about 0.3 GB of HBM per call; the
 * library never takes more than SNES_WINDOW_MAX, default 64, calls per window). */
int32_t snesimage_slots_reserve(snesimage_ctx *ctx, uint32_t n_slots);
/* The two phases of one window, for sharding its calls over GPUs (the window's calls in runs of six consecutive calls, round robin over the ranks: every call its own base
 * image, its own candidates).  Phase 1 takes at most n_slots calls from the given scheduler state — fewer where the method
 * changes, all calls of a window having the same number of candidates (*stride) — and writes errors[j * stride + k]
 * (device, n_slots * 64 doubles; +inf for calls of other ranks; NULL = the context's own vector, single rank only);
 * n_slots is capped at SNES_WINDOW_MAX * shard_count (and 1024).  The
 * caller min-all-reduces the first *n_taken * *stride doubles, then phase 2 commits identically on every rank: *consumed
 * calls took effect, *accepted = the last of them changed the palette; log (host, optional): *n_taken records. */
int32_t snesimage_slots_begin(snesimage_ctx *ctx, uint32_t n_slots, uint64_t seed, uint64_t first_step_id, uint32_t palette,
                              uint32_t index, uint32_t channel, uint32_t step, uint32_t n_random, uint32_t shard_rank,
                              uint32_t shard_count, double *d_errors, uint32_t *n_taken, uint32_t *stride);
int32_t snesimage_slots_commit(snesimage_ctx *ctx, const double *d_errors, uint32_t *consumed, uint32_t *accepted,
                               snesimage_call_result *log);

/* One process, several GPUs: a group borrows one context per device, all created from the same image and brought to
 * the same state (initialise one, copy tile_palettes and palette to the others, optimize()).  snesimage_group_step is
 * snesimage_step with the candidates sharded over the members (rank r scores k = r mod N): step_begin on every member,
 * one grouped RCCL all-reduce(min) of the error vectors on the members' streams, step_commit on every member — the
 * palettes stay bit-identical on all devices.  n_total as in snesimage_step_begin.  librccl is opened at run time.
 * Lifetime: the group borrows its contexts — destroy the group first.  Destroying a member first retires the group
 * (snesimage_group_step then fails with SNES_ERR_STATE; snesimage_group_destroy is still required).  A context belongs
 * to at most one group.  A failed step leaves no member with a pending split-phase step. */
typedef struct snesimage_group snesimage_group;
int32_t snesimage_group_create(snesimage_ctx **ctxs, uint32_t n, snesimage_group **out);
void snesimage_group_destroy(snesimage_group *group);
int32_t snesimage_group_step(snesimage_group *group, uint32_t method, uint32_t palette, uint32_t index,
                             uint32_t channel, uint64_t seed, uint64_t step_id, uint32_t n_total,
                             double *best_error, uint8_t *best_rgb5 /*3*/);

/* snesimage_run_slots over a group: the calls of every window are dealt to the members (runs of six consecutive calls, round robin), one grouped
 * RCCL all-reduce(min) over the window's error vector, identical in-order commit on every member.  Same arguments, same
 * trajectory (lib.rs:888-933), bit-identical palettes on all devices.  Every member must be in the state
 * snesimage_group_create asks for. */
int32_t snesimage_group_run_slots(snesimage_group *group, uint32_t n_calls, uint64_t seed, uint64_t first_step_id,
                                  uint32_t *palette, uint32_t *index, uint32_t *channel, uint32_t *step, uint32_t n_random, uint32_t window,
                                  snesimage_call_result *log, snesimage_run_stats *stats);

/* Throughput mode — many independent images on one device, one launch per stage of an optimizer call
 * for all of them (the reference runs one image per process: `run()` lib.rs:830-1024 once per file).
 * A batch borrows its contexts (same device, image size, palette geometry, chunk and flags — either distance, with or
 * without --dither): snesimage_batch_step_async is snesimage_step_async for every member — same slot and
 * method, candidate stream of member i keyed (seeds[i], step_id), at most `chunk` candidates — enqueued
 * on the batch's stream.  Any other call on a member context first waits for that stream, and the first batched call
 * after such a call waits for the member's own stream; snesimage_set_chunk is refused on a lent context.  The batch does
 * not own the contexts: destroy it before them (destroying a member first retires the batch: later calls fail with
 * SNES_ERR_STATE). */
typedef struct snesimage_batch snesimage_batch;
int32_t snesimage_batch_create(snesimage_ctx **ctxs, uint32_t n, snesimage_batch **out);
void snesimage_batch_destroy(snesimage_batch *batch);
int32_t snesimage_batch_step_async(snesimage_batch *batch, uint32_t method, uint32_t palette, uint32_t index,
                                   uint32_t channel, const uint64_t *seeds /*n*/, uint64_t step_id,
                                   uint32_t n_random);
int32_t snesimage_batch_sync(snesimage_batch *batch);

/* Dynamic tile -> subpalette reassignment — NOT a reference method: /root/reference/TODO.md:36-37 lists it as missing ("no
 * attempt is made to reassign tiles dynamically if it could improve the overall result").  Every tile with an opaque pixel
 * moves to the subpalette with the strictly smallest cost, cost(p) = sum over the tile's opaque pixels (raster order) of the
 * distance optimize() minimises (lib.rs:1080-1100) to the nearest entry of subpalette p, in binary64; ties keep the current
 * subpalette, then the lower index.  Palettes are kept; optimize() re-runs if a tile moved.  *moved = tiles moved.
 * Refused (SNES_ERR_STATE) between the two phases of a split-phase step.  On the members of a group call it on every
 * member (the result is deterministic: the replicas stay identical), never on one alone. */
int32_t snesimage_reassign_tiles(snesimage_ctx *ctx, uint32_t *moved);

/* State access (the reference mutates these fields directly: lib.rs:1015 and the GUI). */
int32_t snesimage_get_tile_palettes(snesimage_ctx *ctx, uint8_t *out /*1024*/);
int32_t snesimage_set_tile_palettes(snesimage_ctx *ctx, const uint8_t *in /*1024*/);
int32_t snesimage_get_palette_rgb5(snesimage_ctx *ctx, uint8_t *out /*count*size*3*/);
int32_t snesimage_set_palette_rgb5(snesimage_ctx *ctx, const uint8_t *in);
int32_t snesimage_get_palette_u16(snesimage_ctx *ctx, uint16_t *out /*count*size; lib.rs:679-681*/);
int32_t snesimage_get_palette_map(snesimage_ctx *ctx, uint8_t *out /*w*h*/);
int32_t snesimage_set_palette_map(snesimage_ctx *ctx, const uint8_t *in);
int32_t snesimage_as_rgba(snesimage_ctx *ctx, uint8_t *out /*w*h*4; lib.rs:550-577*/);
/* as_json().to_string() — lib.rs:579-625, 1002.  Returns the byte count needed including the
 * terminating NUL (negative on error) and writes at most cap bytes. */
int64_t snesimage_as_json(snesimage_ctx *ctx, char *out, int64_t cap);

/* Candidate generator used by SNES_METHOD_RANDOM (host helper; replaces the reference's unseeded
 * rand::rng() of lib.rs:201-208). */
void snesimage_random_candidates(uint64_t seed, uint64_t step_id, uint32_t n, uint8_t *rgb5);
/* Slot scheduler of lib.rs:881-933 (host helper): reports the method for the current slot and
 * advances (palette, index, channel, step). */
void snesimage_schedule_next(uint32_t sub_count, uint32_t sub_size, int32_t nes, uint32_t *palette,
                             uint32_t *index, uint32_t *channel, uint32_t *step, uint32_t *method);

/* Device-side evaluation of the deterministic math used by the kernels, for bit-parity tests:
 * op 0 sin, 1 cos, 2 exp(x<=0), 3 cbrt, 4 atan2(y,x), 5 CIEDE2000(lab x[3i..], lab y[3i..]),
 * 6 sRGB8->Lab (x holds r,g,b as floats, out 3 per item). Host pointers. */
int32_t snesimage_debug_math(int32_t device, int32_t op, const float *x, const float *y, uint32_t n,
                             float *out);
/* Fault injection for tests: the (n+1)-th workspace allocation made by the library from now on fails as if the
 * device were out of memory (n < 0 switches the hook off).  A failed grow returns SNES_ERR_HIP and leaves the context
 * usable: the next call allocates afresh. */
void snesimage_debug_fail_alloc(int32_t n);
/* Launch timing by the library's own HIP events on the stream each launch group runs on (bench.py's roofline leg).
 * While enabled, every scoring launch group records events; timing_read() returns summed milliseconds:
 *   ms3[0] = the whole launch group (all kernels that score one chunk of candidates);
 *   ms3[1] = the H pass: k_sparse_h2 + k_sparse_h (wide and narrow scales) on the group-sparse path, k_hpass* at scale 0 otherwise;
 *   ms3[2] = the V pass, the dominant kernel: k_sparse_v2 alone (one launch, the scales at least 64 wide; the wait for
 *            B's checkpoints and k_sparse_order come before the opening event) on the group-sparse path — for launch groups
 *            of 1,024 candidates and more, which run on two streams, the launch of scale 0 (three quarters of the V pass)
 *            on its stream, with the other stream's kernels resident beside it — k_vpass* at scale 0 otherwise;
 * plus the number of launch groups and the candidates they scored since timing was enabled.
 * on = 1: all three brackets — six event records per launch group, each a ~4 us bubble in the stream (25 us per group: 1.5 % of a
 * 4,096-candidate group, 9 % of a 64-candidate one; profiles/r4_timing_cost.py).  on = 2: the V pass's bracket only (ms3[0] and
 * ms3[1] stay 0): what bench.py keeps on over its timed region.  on = 0: off. */
int32_t snesimage_timing_enable(snesimage_ctx *ctx, int32_t on);
int32_t snesimage_timing_read(snesimage_ctx *ctx, double *ms3, uint64_t *launches,
                              uint64_t *candidates);

const char *snesimage_last_error(void);
const char *snesimage_version(void);

#ifdef __cplusplus
}
#endif
#endif
