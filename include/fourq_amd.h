/* fourq_amd -- batched FourQ (Curve4Q) scalar multiplication on AMD MI355X (gfx950).
 *
 * C ABI of libfourq_amd.so.  This is the drop-in boundary for the hot path of the reference
 * implementation bifurcation/fourq (pure Python; citations are impl/<file>:<line>).  The
 * reference has no FFI of its own -- its boundary is the Python signatures
 *     MUL_windowed(m, P, table=None)   curve4q.py:188      table_windowed(P)   curve4q.py:179
 *     MUL_endo(m, P, table=None)       curve4q.py:405      table_endo(P)       curve4q.py:385
 *     DH_windowed / DH_endo            curve4q.py:464-468  (DH_core curve4q.py:446)
 * so each entry point below is the batched form of one of those, and the ctypes stub that
 * presents the reference signatures on top of it is fourq_amd/curve4q.py.
 *
 * Data layout (all little-endian 64-bit words; plain pointers, caller-allocated):
 *   GF(p) element      2 words, value in [0, 2^128) accepted, canonical [0, p) returned
 *   GF(p^2) element    4 words  (re, im)                         fields.py:134
 *   scalar             4 words  (256-bit, as the reference's KAT generator curve4q.py:552-559)
 *   affine point       8 words  (x, y)
 *   R1 point          20 words  (X, Y, Z, Ta, Tb)                curve4q.py:100-106
 *   R2 point          16 words  (X+Y, Y-X, 2Z, 2dT)              curve4q.py:109-116
 *   table            128 words  8 R2 points                      curve4q.py:179-185, :385-403
 *
 * Every function returns FOURQ_OK (0) or a negative error code and never throws.  A context is
 * bound to one device and one HIP stream; calls on one context must not overlap, different
 * contexts are independent.  There is no CPU fallback: without a usable gfx950 device
 * fourq_ctx_create fails.
 *
 * A context owns device scratch (per-lane look-up tables, staging) that its launches reuse in stream order:
 * let the current stream's work finish (fourq_ctx_sync or the caller's own synchronisation) before handing the
 * context another stream with fourq_ctx_set_stream (the staged fixed-base tables are re-staged on the new stream).
 *
 * Pointer flavours: functions ending in _dev take DEVICE pointers, enqueue on the context's
 * stream and return without synchronising (use fourq_ctx_sync or the caller's stream).  The same
 * names without _dev take HOST pointers and are synchronous (H2D copy, kernel, D2H copy, pipelined over chunks of
 * the batch on three streams; see fourq_host_alloc for the fast path).
 * Device arrays are read and written as 16-byte vectors: every array pointer handed to a _dev function must be
 * 16-byte aligned (FOURQ_ERR_INVALID otherwise; hipMalloc / fourq_dev_alloc / torch allocations are).  Host
 * pointers need no particular alignment.
 */
#ifndef FOURQ_AMD_H
#define FOURQ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOURQ_OK 0
#define FOURQ_ERR_INVALID (-1)   /* NULL pointer, bad size or bad enum */
#define FOURQ_ERR_NODEVICE (-2)  /* no HIP device / device is not gfx950-compatible */
#define FOURQ_ERR_NOMEM (-3)     /* device or host allocation failed */
#define FOURQ_ERR_HIP (-4)       /* a HIP runtime call or kernel launch failed (see fourq_last_error) */

/* per-element status of the DH entry points; the Python wrapper re-raises 1 and 2 with the
 * reference's exact messages (curve4q.py:448, :460) */
#define FOURQ_DH_OK 0
#define FOURQ_DH_NOT_ON_CURVE 1
#define FOURQ_DH_NEUTRAL 2

/* largest batch any entry point accepts (the kernels' 32-bit element counters round n up to whole 256-lane blocks) */
#define FOURQ_MAX_BATCH 0xffffff00u
#define FOURQ_BYTES_DECODE_BASE 16  /* status of the *_bytes_* calls: this + FOURQ_DECODE_* when an encoded point does not decode */

#define FOURQ_SCALAR_WORDS 4
#define FOURQ_AFFINE_WORDS 8
#define FOURQ_R1_WORDS 20
#define FOURQ_R2_WORDS 16
#define FOURQ_TABLE_WORDS 128

typedef struct fourq_ctx fourq_ctx;

/* ---- library / context ---------------------------------------------------------------------- */
/* The ABI this header describes.  Structs that the library fills (fourq_host_stats) and prototypes may grow between versions: a host
 * compiled against this header must check fourq_version() == FOURQ_ABI_VERSION once at start-up (the Python binding does, fourq_amd/_lib.py)
 * -- or use the size-carrying forms (fourq_ctx_host_stats_sized), which never write past what the caller's header knew. */
#define FOURQ_ABI_VERSION 600
int fourq_version(void);                       /* 10000*major + 100*minor + patch; == FOURQ_ABI_VERSION for a matching library */
const char *fourq_build_id(void);              /* 16 hex digits: hash of the sources and flags the library was built from */
const char *fourq_strerror(int code);
const char *fourq_last_error(const fourq_ctx *ctx);   /* detail of the last FOURQ_ERR_HIP */

/* Number of usable devices (gfx950 only; 0 when there is none -- not an error).  SURVEY.md 8(e): one context per device,
 * contiguous shards, no exchange step; fourq_amd/multi.py (MultiEngine) is the host-side loop over them. */
int fourq_device_count(int *count);
int fourq_ctx_create(int device, fourq_ctx **out);
/* Threads.  Every entry point that takes a context holds that context's lock for its duration, so calls made on ONE context
 * from several threads are safe and take turns (a host-array call for its whole duration, a _dev call for its enqueue); the
 * reference's functions are pure, and a drop-in caller may use them from any thread.  Calls on different contexts run side by
 * side (fourq_amd/multi.py: one context and one host thread per device).  What stays per context and is therefore "of the last
 * call, whoever made it": fourq_last_error, fourq_ctx_host_stats, the stream and mode switches.  fourq_ctx_destroy waits for a
 * call in flight; starting another one after it is the caller's error. */
int fourq_ctx_destroy(fourq_ctx *ctx);
/* Use the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL restores the
 * context's own stream.  The context never synchronises an external stream behind the caller's back. */
int fourq_ctx_set_stream(fourq_ctx *ctx, void *hip_stream);
int fourq_ctx_sync(fourq_ctx *ctx);
/* Constant-time table selection (draft-ladd-cfrg-4q.md:753-758: "memory addresses accessed [must] not depend on secret
 * data").  OFF by default: the ladders then do exactly what the reference does -- a digit of the scalar is a table
 * address (curve4q.py:232, :440) and the SIGN of the digit is applied by masked selects (curve4q.py:193-206), never by
 * an address -- which is NOT constant-time with respect to the scalar's digits.  ON (this call, or
 * FOURQ_CT_SELECT=1 in the environment when the context is created): every ladder step reads the whole table and
 * keeps the wanted entry by masks; per-lane tables live in registers; signs are applied arithmetically.  Results are
 * bit-identical in both modes; the price of ON is in DESIGN.md section 10.
 * Environment: FOURQ_CT_SELECT is the ONE variable the library reads as a product option.  The variables that steer batches onto
 * particular kernels (FOURQ_SPLIT_*, FOURQ_PAIR_MAX, FOURQ_QUAD_MAX, FOURQ_MIXED_QUEUE, FOURQ_NORM_K, FOURQ_BLOCKS_PER_CU,
 * FOURQ_HOST_BOUNCE, FOURQ_HOST_ZERO_COPY, FOURQ_PIPE_SLOTS, FOURQ_PIPE_GENS, FOURQ_PIPE_HOST_WAIT, FOURQ_PIPE_HOST_POLL, FOURQ_PIPE_MEASURE, FOURQ_FUSED_IO) are test hooks: they are ignored unless FOURQ_DEBUG_ROUTES=1 is set (tools/README.md).
 * One variable of the HIP RUNTIME matters to the host-pointer calls: they overlap copy-in, kernels and copy-out on three streams,
 * and the runtime shares GPU_MAX_HW_QUEUES hardware queues (default 4) among all streams the process uses -- in a process with two
 * or more other busy streams the three stages take turns and a large call takes up to twice as long.  Set GPU_MAX_HW_QUEUES=8 in
 * the host's environment before its first HIP call (the library does not touch the environment; the Python package sets it at
 * import when it is unset).  The _dev calls use one stream and are not affected.  INTEGRATION.md section 3. */
int fourq_ctx_set_ct_select(fourq_ctx *ctx, int on);
int fourq_ctx_get_ct_select(const fourq_ctx *ctx, int *on);
/* Resident lanes the ladder kernels are launched with (scratch is sized for this many). */
int fourq_ctx_lanes(const fourq_ctx *ctx, size_t *lanes);
/* Buffer growth.  The context owns the intermediates of the DH and protocol-level calls (deferred-normalisation planes,
 * decoded keys, first-half results); they grow on demand, and a _dev call whose batch is larger than any seen before
 * SYNCHRONISES the context's stream and reallocates before it enqueues (such a call cannot be captured into a HIP graph).
 * fourq_ctx_reserve(ctx, n) sizes them once for batches of up to n elements: afterwards _dev calls of at most n elements
 * only enqueue.  (The host-pointer calls size their pipeline slots themselves and are synchronous anyway.)
 * Capturing _dev calls into a HIP graph: reserve first, and STAGE the caller's tables first -- a fixed-base table or a comb is
 * uploaded from a context-owned host copy when it differs from what is staged, and that upload must not be captured (it would
 * read the mutable copy at replay time).  One un-captured call with the same table (or fourq_comb_stage) stages it; a call that
 * would have to stage while its stream is being captured returns FOURQ_ERR_HIP with fourq_last_error() saying so. */
int fourq_ctx_reserve(fourq_ctx *ctx, size_t n);

/* Pinned (page-locked) host memory.  The host-pointer batch calls cut their arrays into chunks and overlap the
 * H2D copy, the kernels and the D2H copy of consecutive chunks; arrays that live in pinned memory (from here, from
 * hipHostMalloc or from torch's pin_memory) are moved by DMA straight from / to the caller's buffer at the link's
 * rate, pageable arrays take an extra pass through pinned bounce buffers filled by host threads. */
int fourq_host_alloc(fourq_ctx *ctx, size_t bytes, void **out);
int fourq_host_free(fourq_ctx *ctx, void *ptr);
/* transfer statistics of the context's last host-pointer batch call */
typedef struct fourq_host_stats {
    double h2d_ms, d2h_ms;          /* summed durations of the chunk copies (HIP events), measured only while
                                     * fourq_ctx_set_host_timing(ctx, 1) is in force -- 0 otherwise: the six event records per
                                     * chunk are not free, so a production call does not make them (also 0 for a call of at
                                     * most 64 KiB: its kernels read and write a pinned host buffer in place, there are no
                                     * device copies at all) */
    uint64_t h2d_bytes, d2h_bytes;  /* bytes moved by device copies: 0 for such an in-place call */
    uint32_t chunks;
    int pinned_in, pinned_out;      /* 1: every input / output array was pinned (no bounce copy) */
    double kernels_ms;              /* under fourq_ctx_set_host_timing: summed over the chunks, from "the kernel stream has the chunk's
                                     * bytes" to "the chunk's kernels are done" (HIP events on the kernel stream); 0 otherwise */
    double kernels_span_ms;         /* ... and from the first chunk's start to the last chunk's end: span - sum = the kernel stream's idle
                                     * time between chunks (waiting for bytes, launch gaps) */
    /* 0.6.0: what the call's chunks were PLANNED with (fourq_amd/csrc/pipeline_plan.h) -- the context's own measurements of this route in this
     * selection mode when planned_from_measurement is 1 (every multi-chunk call times one middle chunk and leaves the figures for the next
     * call of the same route), the compiled-in first-call guesses when 0 -- and what THIS call measured.  0 for a call of one chunk. */
    double planned_kernel_ns_per_elem, planned_link_in_gbs, planned_link_out_gbs;   /* kernel time per element; link rate each way, GB/s = bytes / ns */
    double measured_kernel_ns_per_elem;
    int planned_from_measurement;
} fourq_host_stats;
int fourq_ctx_host_stats(const fourq_ctx *ctx, fourq_host_stats *out);
/* The same, writing at most `size` bytes (pass sizeof(fourq_host_stats) of the header the caller was compiled against): safe across
 * versions in which the struct grew (0.4.0: 48 bytes, 0.5.0: 64, 0.6.0: 104). */
int fourq_ctx_host_stats_sized(const fourq_ctx *ctx, void *out, size_t size);
/* Diagnostic: time the chunk copies of the following host-pointer calls (h2d_ms / d2h_ms above).  OFF by default; bytes, chunk
 * count and the pinned flags are always reported. */
int fourq_ctx_set_host_timing(fourq_ctx *ctx, int on);
/* Diagnostic: where chunk `chunk` of the context's last host-pointer call sat in time, when that call was made under
 * fourq_ctx_set_host_timing -- six stamps in milliseconds since the call's first event: copy-in start, copy-in end, copy-out start,
 * copy-out end, kernels start (the kernel stream is past its wait for the chunk's bytes), kernels end.  FOURQ_ERR_INVALID past the
 * last timed chunk (tools/pipeline_chunks.py prints the table). */
int fourq_ctx_host_chunk_stamps(const fourq_ctx *ctx, uint32_t chunk, double out_ms[6]);

/* Diagnostic: the shader clock (MHz) the device holds at this moment, measured from inside a kernel -- a 16-wave probe on a stream of
 * the context's own times `window_us` (1 .. 1 000 000) of the constant 100 MHz counter (s_memrealtime) in shader cycles (s_memtime);
 * median, minimum and maximum over the probe's waves (two per XCD).  Called while the context's stream has work queued for longer than
 * the window (the _dev calls only enqueue) it reports the clock under that load: a time measured on one box times this clock is a cycle
 * count comparable with another box's (devices differ by several percent in the clock they hold under the same kernel).  Synchronous
 * for the window; no product kernel carries a stamp.  mhz_min / mhz_max / under_load may be NULL.  *under_load = 1 when the context's
 * stream still had work in flight when the window closed, 0 when it had already drained (or the probe had queued BEHIND it on a shared
 * hardware queue): the reading is then the clock of an idle chip and must not be used to convert the load's time into cycles. */
int fourq_diag_clock(fourq_ctx *ctx, uint32_t window_us, double *mhz_median, double *mhz_min, double *mhz_max, int *under_load);
/* The clock OF a stretch of work, as a bracket: _begin and _stop each enqueue one tiny kernel on the CONTEXT'S stream -- before and behind
 * whatever the caller enqueues in between (_dev calls), so the host never waits and nothing stays resident beside the work -- whose waves
 * record their CU's cycle counter and the global 100 MHz counter; _end (implies _stop) waits for the stream, pairs the two launches' stamps
 * CU by CU (s_memtime is a per-CU counter) and returns the median clock over the CUs, its 5th / 95th percentile as min / max, and the
 * length of the window in microseconds.  bench.py brackets its timed steps with it; no product kernel carries a stamp. */
int fourq_diag_clock_begin(fourq_ctx *ctx);
int fourq_diag_clock_stop(fourq_ctx *ctx);
int fourq_diag_clock_end(fourq_ctx *ctx, double *mhz_median, double *mhz_min, double *mhz_max, double *window_us);

/* Plain device-memory helpers so that a host program without a HIP binding can use the _dev API. */
int fourq_dev_alloc(fourq_ctx *ctx, size_t bytes, void **out);
int fourq_dev_free(fourq_ctx *ctx, void *ptr);
int fourq_dev_upload(fourq_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int fourq_dev_download(fourq_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---- look-up tables: table_windowed(P) curve4q.py:179, table_endo(P) curve4q.py:385 ----------- */
int fourq_table_windowed(fourq_ctx *ctx, const uint64_t *p_r1, uint64_t *table);
int fourq_table_endo(fourq_ctx *ctx, const uint64_t *p_r1, uint64_t *table);

/* ---- variable-base scalar multiplication: MUL_*(m_i, P_i), table=None ------------------------- */
/* out_r1[i] = raw R1 tuple the reference returns (curve4q.py:235, :442), n x 20 words */
int fourq_mul_endo_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_r1, uint64_t *out_r1, size_t n);
int fourq_mul_windowed_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_r1, uint64_t *out_r1, size_t n);
int fourq_mul_endo_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_r1, uint64_t *out_r1, size_t n);
int fourq_mul_windowed_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_r1, uint64_t *out_r1, size_t n);

/* ---- the same with affine / encoded I/O ("affine-only I/O variant") ----------------
 * out_affine[i] = R1toAffine(MUL_<algo>(m_i, AffineToR1(x_i, y_i)))   curve4q.py:100-106; canonical affine, n x 8 words: 160 bytes per
 * operation across the ABI instead of 352.  Like MUL_* itself these check nothing: a point outside the prime-order subgroup gives the
 * reference's (meaningless) answer (draft-ladd-cfrg-4q.md:467-468). */
int fourq_mul_endo_affine_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_affine, uint64_t *out_affine, size_t n);
int fourq_mul_windowed_affine_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_affine, uint64_t *out_affine, size_t n);
int fourq_mul_endo_affine_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_affine, uint64_t *out_affine, size_t n);
int fourq_mul_windowed_affine_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_affine, uint64_t *out_affine, size_t n);
/* out32[i] = encode(R1toAffine(MUL_<algo>(m_i, AffineToR1(decode(points32[i])))))   curve4q.py:41-96: 96 bytes per operation.
 * status[i]: 0 ok; FOURQ_BYTES_DECODE_BASE + FOURQ_DECODE_* when points32[i] does not decode (out32[i] is all zero then). */
int fourq_mul_endo_bytes_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint8_t *points32, uint8_t *out32, uint8_t *status, size_t n);
int fourq_mul_windowed_bytes_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint8_t *points32, uint8_t *out32, uint8_t *status, size_t n);
int fourq_mul_endo_bytes_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint8_t *points32, uint8_t *out32, uint8_t *status, size_t n);
int fourq_mul_windowed_bytes_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint8_t *points32, uint8_t *out32, uint8_t *status, size_t n);

/* ---- fixed-base scalar multiplication: MUL_*(m_i, P, table=T) --------------------------------- */
/* `table` is 128 words (host pointer in both flavours: it is 1 KiB and is staged once per call) */
int fourq_mul_endo_fixed_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *table, uint64_t *out_r1, size_t n);
int fourq_mul_windowed_fixed_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *table, uint64_t *out_r1, size_t n);
int fourq_mul_endo_fixed_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *table, uint64_t *out_r1, size_t n);
int fourq_mul_windowed_fixed_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *table, uint64_t *out_r1, size_t n);

/* ---- mixed batch (BASELINE.json config 5): element i is fixed-base (flags[i] == 0, uses `table`)
 *      or variable-base (flags[i] != 0, uses points_r1[i]); MUL_endo in both cases ---------------- */
int fourq_mul_endo_mixed_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_r1, const uint8_t *flags,
                               const uint64_t *table, uint64_t *out_r1, size_t n);
int fourq_mul_endo_mixed_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_r1, const uint8_t *flags,
                                   const uint64_t *table, uint64_t *out_r1, size_t n);

/* ---- Diffie-Hellman: DH_core(m_i, P_i, mul, table) curve4q.py:446-462 --------------------------
 * table == NULL: variable base (the table is built from [392]P_i); otherwise `table` (host pointer)
 * is used for every element exactly as the reference does (it ignores the point, curve4q.py:209, :426).
 * status[i]: FOURQ_DH_*; out_affine[i] is all zero when status[i] != 0.
 * Large batches share one GFp.inv (fields.py:66-106) among up to eight elements of R1toAffine (curve4q.py:103-106;
 * Montgomery's trick); the affine result is canonical and therefore unchanged. */
int fourq_dh_endo_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_affine, const uint64_t *table,
                        uint64_t *out_affine, uint8_t *status, size_t n);
int fourq_dh_windowed_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_affine, const uint64_t *table,
                            uint64_t *out_affine, uint8_t *status, size_t n);
int fourq_dh_endo_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_affine, const uint64_t *table,
                            uint64_t *out_affine, uint8_t *status, size_t n);
int fourq_dh_windowed_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *points_affine, const uint64_t *table,
                                uint64_t *out_affine, uint8_t *status, size_t n);

/* ---- the protocol step on the device: 32-byte public keys in, 32-byte shared secrets out -----------------------
 * out32[i] = encode(DH_<algo>(m_i, decode(keys32[i]) [, table]))   draft-ladd-cfrg-4q.md:707-723;
 * curve4q.py:49-96 (decode), :446-462 (DH_core), :41-46 (encode).  Decoded points and shared points never leave the
 * GPU.  status[i]: 0 ok; FOURQ_DH_NOT_ON_CURVE / FOURQ_DH_NEUTRAL from the DH stage; 16 + FOURQ_DECODE_* when the
 * key does not decode (takes precedence).  out32[i] is all zero unless status[i] == 0. */
int fourq_dh_endo_bytes_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint8_t *keys32, const uint64_t *table,
                              uint8_t *out32, uint8_t *status, size_t n);
int fourq_dh_windowed_bytes_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint8_t *keys32, const uint64_t *table,
                                  uint8_t *out32, uint8_t *status, size_t n);
int fourq_dh_endo_bytes_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint8_t *keys32, const uint64_t *table,
                                  uint8_t *out32, uint8_t *status, size_t n);
int fourq_dh_windowed_bytes_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint8_t *keys32, const uint64_t *table,
                                      uint8_t *out32, uint8_t *status, size_t n);
/* One exchange per element: out[i] = DH_endo(a_i, DH_endo(b_i, base [, table392])) -- the pattern of curve4q.py:731
 * (BASELINE.json "dh_exchange"); `base_affine` (8 words) and `table392` = table_endo([392]base) or NULL are HOST
 * pointers in both flavours.  The first half's results stay on the device.  status[i]: first failure of either half. */
int fourq_dh_exchange_batch(fourq_ctx *ctx, const uint64_t *a_scalars, const uint64_t *b_scalars, const uint64_t *base_affine,
                            const uint64_t *table392, uint64_t *out_affine, uint8_t *status, size_t n);
int fourq_dh_exchange_batch_dev(fourq_ctx *ctx, const uint64_t *a_scalars, const uint64_t *b_scalars, const uint64_t *base_affine,
                                const uint64_t *table392, uint64_t *out_affine, uint8_t *status, size_t n);

/* ---- point compression: encode(X, Y) curve4q.py:41, decode(B) curve4q.py:49 -----------------------
 * 32 bytes per point: y0 | y1 little-endian, sign(x) in the top bit of the last byte.  decode reports, per
 * element, the exception the reference would raise; out_affine[i] is all zero unless status[i] == 0. */
#define FOURQ_DECODE_OK 0
#define FOURQ_DECODE_RESERVED_BIT 1        /* "Malformed point: reserved bit is not zero" (curve4q.py:53, :62) */
#define FOURQ_DECODE_NOT_ON_CURVE 2        /* "Point not on curve" (curve4q.py:94) */
#define FOURQ_DECODE_REF_ATTRIBUTE_ERROR 3 /* the reference's t == 0 branch (curve4q.py:76-77) raises AttributeError */
int fourq_encode_batch(fourq_ctx *ctx, const uint64_t *points_affine, uint8_t *out32, size_t n);
int fourq_decode_batch(fourq_ctx *ctx, const uint8_t *in32, uint64_t *out_affine, uint8_t *status, size_t n);
int fourq_encode_batch_dev(fourq_ctx *ctx, const uint64_t *points_affine, uint8_t *out32, size_t n);
int fourq_decode_batch_dev(fourq_ctx *ctx, const uint8_t *in32, uint64_t *out_affine, uint8_t *status, size_t n);

/* ---- fixed-base comb (draft-ladd-cfrg-4q.md:725-729: "MAY use any method ... provided that it agrees") ------
 * A comb table (mLSB-set recoding, w = 9, v = 4: 1 024 points, held in one CU's LDS) for a base point B of order N makes [m]B
 * cost 6 doublings + 27 mixed additions instead of 64 + 64; the table object also holds an 80-point comb (w = v = 5) that the
 * constant-time mode scans in 16-entry blocks.  Outputs are canonical AFFINE points, identical to
 * R1toAffine(MUL_endo(m, B)); the un-normalised R1 tuple of the reference is NOT reproduced, so this serves the
 * DH/keygen side (DH_endo(m, G, table) == fourq_comb_mul_batch with the comb of [392]G), not raw MUL_*.
 * comb table: FOURQ_COMB_WORDS words = 1 024 + 80 entries x (x+y, y-x, 2d*x*y), 4 words each (103.5 KiB). */
#define FOURQ_COMB_POINTS 1104
#define FOURQ_COMB_WORDS (1104 * 12)
int fourq_comb_table(fourq_ctx *ctx, const uint64_t *p_r1, uint64_t *comb);
/* The device copy of the comb stays staged between calls.  A batch call with a non-NULL `comb` compares it with the staged
 * copy (103.5 KiB on the host, per call) and uploads it when it differs; fourq_comb_stage does that once, and batch calls
 * with comb == NULL then use the staged table without touching it (FOURQ_ERR_INVALID when no table was ever given; after
 * fourq_ctx_set_stream the context uploads its own copy again by itself on the first use). */
int fourq_comb_stage(fourq_ctx *ctx, const uint64_t *comb);
/* status[i]: FOURQ_DH_OK or FOURQ_DH_NEUTRAL ([m]B is the neutral point); out zeroed in that case */
int fourq_comb_mul_batch(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *comb, uint64_t *out_affine, uint8_t *status, size_t n);
int fourq_comb_mul_batch_dev(fourq_ctx *ctx, const uint64_t *scalars, const uint64_t *comb, uint64_t *out_affine, uint8_t *status, size_t n);
/* One exchange per element with the key-generation half through the comb: out[i] = DH_endo(a_i, K_i) with
 * K_i = [b_i]B affine = DH_endo(b_i, base, table_endo([392]base)) for the comb of B = [392]base (draft :725-729: any method
 * that agrees) -- BASELINE.json "dh_exchange" as bench.py's cfg4 runs it, as ONE call; the public keys K stay on the device.
 * `comb` as in fourq_comb_mul_batch (NULL = the staged table).  status[i]: first failure of either half. */
int fourq_dh_exchange_comb_batch(fourq_ctx *ctx, const uint64_t *a_scalars, const uint64_t *b_scalars, const uint64_t *comb,
                                 uint64_t *out_affine, uint8_t *status, size_t n);
int fourq_dh_exchange_comb_batch_dev(fourq_ctx *ctx, const uint64_t *a_scalars, const uint64_t *b_scalars, const uint64_t *comb,
                                     uint64_t *out_affine, uint8_t *status, size_t n);

/* ---- primitives (one reference function per op, batched) --------------------------------------
 * Used by the Python mirror of the reference's helper API (GFp.*, GFp2.*, DBL, ADD, phi, ...) and by
 * the parity tests to check every layer of the path on the GPU.  in/out are HOST pointers;
 * element strides are fourq_prim_words(). */
enum fourq_prim {
    /* GFp, fields.py:29-106: in = a[2] b[2] */
    FOURQ_FP_ADD = 0, FOURQ_FP_SUB = 1, FOURQ_FP_MUL = 2, FOURQ_FP_SQR = 3, FOURQ_FP_NEG = 4, FOURQ_FP_INV = 5,
    FOURQ_FP_INVSQRT = 6,       /* fields.py:110 */
    /* GFp.select(c, x, y) fields.py:59-64, GFp2.select fields.py:236-238: in = c[2] x y; raw 128-bit words,
     * y ^ ((mask * c) & (x ^ y)) with the reference's mask = 2^512 - 1, no reduction (as the reference) */
    FOURQ_FP_SELECT = 7, FOURQ_FP2_SELECT = 23,
    /* GFp2, fields.py:156-199: in = a[4] b[4] */
    FOURQ_FP2_ADD = 16, FOURQ_FP2_SUB = 17, FOURQ_FP2_MUL = 18, FOURQ_FP2_SQR = 19, FOURQ_FP2_NEG = 20,
    FOURQ_FP2_CONJ = 21, FOURQ_FP2_INV = 22,
    /* curve4q.py:100-175 */
    FOURQ_PT_DBL = 32,          /* R1[20] -> R1[20] */
    FOURQ_PT_ADD = 33,          /* R1[20] R2[16] -> R1[20] */
    FOURQ_PT_ADD_CORE = 34,     /* R3[16] R2[16] -> R1[20] */
    FOURQ_PT_R1TOR2 = 35,       /* R1[20] -> R2[16] */
    FOURQ_PT_R1TOR3 = 36,       /* R1[20] -> R3[16] */
    FOURQ_PT_R2TOR4 = 37,       /* R2[16] -> R4[12] */
    /* curve4q.py:258-322 */
    FOURQ_PT_TAU = 38,          /* (X,Y,Z)[12] -> [12] */
    FOURQ_PT_TAU_DUAL = 39,     /* [12] -> R1[20] */
    FOURQ_PT_UPSILON = 40,      /* [12] -> [12] */
    FOURQ_PT_CHI = 41,          /* [12] -> [12] */
    FOURQ_PT_PHI = 42,          /* R1[20] -> R1[20] */
    FOURQ_PT_PSI = 43,          /* R1[20] -> R1[20] */
    FOURQ_PT_ON_CURVE = 44,     /* affine[8] -> [1] (0/1)            curve4q.py:23 */
    FOURQ_PT_COFACTOR392 = 45,  /* affine[8] -> R1[20]               curve4q.py:450-455 */
    FOURQ_PT_R1TOAFFINE = 46,   /* R1[20] -> affine[8]               curve4q.py:103 */
    /* curve4q.py:216-226, :339-380 */
    FOURQ_SC_DECOMPOSE = 64,    /* m[4] -> a1..a4 [4] */
    FOURQ_SC_RECODE = 65,       /* v[4] = decompose(m) -> [5]: sign bits 0..63, digit bit-planes 0,1,2, digit 64 */
    FOURQ_SC_WINDOWED = 66      /* m[4] -> [8]: 63 bytes, byte i = (sgn[i] << 3) | ind[i] */
};
int fourq_prim_words(int op, size_t *in_words, size_t *out_words);
int fourq_prim_batch(fourq_ctx *ctx, int op, const uint64_t *in, uint64_t *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* FOURQ_AMD_H */
