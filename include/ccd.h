/*
 * ccd.h - C ABI of the MI355X-native Cool-chic decoder (libccd.so).
 *
 * Drop-in boundary for the reference's decode path (paths relative to /root/reference):
 *   cc_decode.py:14-20                                  -> ccd_decode_video()
 *   coolchic/bitstream/decode.py:26   decode_video()    -> ccd_decode_video()
 *   coolchic/bitstream/decode.py:96   decode_frame()    -> ccd_batch_* (one frame = 1-2 cool-chics)
 *   coolchic/bitstream/component/coolchic.py:29
 *        encode_decode_coolchic(mode="decode")          -> ccd_decode_coolchic(), ccd_batch_*
 *   coolchic/bitstream/header/header.py:72 read_header  -> ccd_read_*_header()
 *
 * Conventions: plain pointers and sizes, no torch types; status-code
 * returns (0 = ok, <0 = error, see ccd_strerror); caller-owned output buffers; one HIP stream
 * per call; process-wide state is limited to per-device caches and helper streams (see the note at ccd_pool_trim); inputs are
 * never modified.  Device pointers are ordinary
 * hipMalloc'ed addresses (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as
 * void* (e.g. torch.cuda.current_stream().cuda_stream), NULL = the default stream.
 *
 * The library has NO CPU fallback: every compute entry point fails with CCD_ERR_HIP when no
 * gfx950 device is usable.
 */
#ifndef CCD_H
#define CCD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCD_MAX_GRIDS 40
#define CCD_MAX_SYN_LAYERS 8
#define CCD_MAX_REFS 2

enum {
    CCD_OK = 0,
    CCD_ERR_TRUNCATED = -1,    /* bitstream shorter than its headers claim */
    CCD_ERR_VALUE = -2,        /* header field out of range (reference: ValueError, element.py:289-292) */
    CCD_ERR_INVALID_DATA = -3, /* range decoder met an impossible quantile (constriction: InvalidData) */
    CCD_ERR_UNSUPPORTED = -4,  /* legal stream using a feature this build does not implement yet */
    CCD_ERR_NOMEM = -5,
    CCD_ERR_HIP = -6,          /* HIP runtime error / no usable device */
    CCD_ERR_ARG = -7           /* bad argument (reference: ValueError, coolchic.py:39-51) */
};

const char* ccd_strerror(int code);
/* "ccd <version> gfx950" */
const char* ccd_version(void);

/* ---- headers (host only; reference: bitstream/header/header.py) ------------------------- */
typedef struct {
    int32_t n_frames, n_intras, n_p_frames, n_bytes_header;
    int32_t intra_pos[4096];
    int32_t p_pos[4096];
} ccd_video_header; /* header.py:130-147 */

typedef struct {
    int32_t display_index;
    int32_t frame_type;      /* 0 I, 1 P, 2 B */
    int32_t frame_data_type; /* 0 rgb, 1 yuv420, 2 yuv444, 3 flow */
    int32_t bitdepth;        /* 8..16 */
    int32_t n_bytes_header;
    int32_t n_refs;
    int32_t index_references[CCD_MAX_REFS];
    int32_t global_flow[2 * CCD_MAX_REFS];
    int32_t warp_filter_size;
} ccd_frame_header; /* header.py:172-218 */

typedef struct { int32_t out_ft, k_size, mode /*0 linear 1 residual*/, non_linearity /*0 none 1 relu*/; } ccd_syn_layer;

typedef struct {
    /* transmitted fields, header.py:244-325 */
    int32_t linear_stabiliser_synth, n_layer_synthesis, ups_k_size, ups_preconcat_k_size;
    int32_t output_feature_ifce, spatial_context_arm, linear_stabiliser_arm, n_hidden_layers_arm;
    int32_t img_size[2];
    int32_t latent_resolution[2];
    int32_t n_latent_grids;
    int32_t flag_hyperlatent, flag_common_randomness;
    int32_t final_upsampling_type; /* 0 nearest 1 bilinear 2 bicubic */
    int32_t nn_q_step_log2[8];     /* arm.w arm.b ifce.w ifce.b ups.w ups.b syn.w syn.b */
    int32_t nn_expgol_cnt[8];
    int32_t nn_n_bytes, nn_n_bit_pad, n_bytes_latent, n_bytes_header;
    int32_t has_ifce_resolution;
    int32_t ifce_resolution[2];
    int32_t hyperlatent_resolution[2];
    ccd_syn_layer syn_layer[CCD_MAX_SYN_LAYERS];
    /* derived geometry, component/core/coolchic.py:149-225 */
    int32_t n_grids;
    int32_t grid_h[CCD_MAX_GRIDS], grid_w[CCD_MAX_GRIDS];
    int32_t is_hyperlatent[CCD_MAX_GRIDS];
    int32_t input_features_ifce[CCD_MAX_GRIDS];
    int32_t input_feature_synthesis;
    int32_t total_context_arm;
    int32_t out_channels; /* synthesis output channels */
    int64_t n_symbols;    /* total latent symbols */
} ccd_cc_header;

/* Each returns the number of header bytes consumed (> 0) or an error (< 0). */
int ccd_read_video_header(const uint8_t* p, size_t n, ccd_video_header* h);
int ccd_read_frame_header(const uint8_t* p, size_t n, ccd_frame_header* h);
int ccd_read_cc_header(const uint8_t* p, size_t n, ccd_cc_header* h);

/* VideoHeader.get_coding_structure() (header.py:161 -> utils/codingstructure.py:226-436 CodingStructure.compute_coding_struct):
 * the frames a video header implies, in CODING order.  Each array receives h->n_frames entries (refs: 2 per frame, display
 * orders, -1 where unused; frame_type 0 I / 1 P / 2 B; depth as the reference counts it).  Returns n_frames, or
 * CCD_ERR_VALUE where the reference asserts (first frame not intra, last frame neither intra nor P, a frame both I and P).
 * Positions listed twice are CCD_ERR_VALUE too: the reference only prints a warning, builds a structure that lacks a display
 * index and fails after decoding (decode.py:86 on None).  ccd_decode_video decodes in this order with these references, like
 * decode.py:67-75 (the frame headers' display_index / index_references are not read there).  The frame header's frame_type
 * decides the number of cool-chics and the reconstruction, as in decode.py:119-128, 156-189: an "I" header at a P / B position
 * is decoded as plain intra, a "P" header at a B position predicts from the structure's first reference; a header type that
 * needs MORE references than the structure gives (the reference: IndexError) is CCD_ERR_VALUE. */
int ccd_get_coding_structure(const ccd_video_header* h, int32_t* display_order, int32_t* frame_type, int32_t* refs, int32_t* depth);

/* ---- one cool-chic: encode_decode_coolchic(mode="decode"), coolchic.py:29-207 ------------- */
/* Decodes one cool-chic on `device` and writes the synthesis output [C][H][W] float32 (after the
 * final resize/crop, coolchic.py:187-192) to `out`, a device pointer if out_on_device else host.
 * Synchronous with respect to `stream` on return when out is a host pointer. */
int ccd_decode_coolchic(const uint8_t* cc_header, size_t n_hdr, const uint8_t* bytes_nn, size_t n_nn,
                        const uint8_t* bytes_latent, size_t n_lat, int device, void* stream,
                        float* out, int out_on_device);

/* ---- batches of cool-chics: the throughput API (one frame per slot, many frames in flight) --
 * A batch owns the device-resident inputs (bitstream words, network parameters) and all
 * intermediate buffers of its slots.  Typical use: create, add every cool-chic of a set of
 * frames, then repeatedly run() - all slots decode concurrently (one workgroup per slot walks
 * the serial ARM + range-decoder chain; upsampling/synthesis tiles fill the rest of the chip). */
typedef struct ccd_batch ccd_batch;

int ccd_batch_create(int device, ccd_batch** out);
void ccd_batch_destroy(ccd_batch* b);
/* Adds one cool-chic; parses headers + network on the host and uploads to HBM. Returns the slot
 * index (>= 0) or an error.  bitdepth/frame_data_type describe the frame it belongs to and drive
 * the integer output planes (decode.py:191-206); pass bitdepth 0 to skip integer planes. */
int ccd_batch_add(ccd_batch* b, const uint8_t* cc_header, size_t n_hdr, const uint8_t* bytes_nn, size_t n_nn,
                  const uint8_t* bytes_latent, size_t n_lat, int bitdepth, int frame_data_type);
int ccd_batch_size(const ccd_batch* b);
int ccd_batch_header(const ccd_batch* b, int slot, ccd_cc_header* h);
/* Optional: builds and uploads the launch tables of the slots added so far (entropy descriptors, work lists of the float
 * kernels: one staged copy on `stream`) without launching anything - what the first ccd_batch_run[_stage] after an add does
 * anyway.  A caller that makes `stream` wait for other work (e.g. the decode of the batch before) calls this first, so that
 * the copy is not queued behind that wait. */
int ccd_batch_prepare(ccd_batch* b, void* stream);
/* Enqueues the whole decode of every slot on `stream` (asynchronous).  The entropy launches of a batch (one per kernel
 * instantiation and chain group: slots grouped by the expected length of their serial chains) fork over device-owned side
 * streams; each is followed on its own stream by the float-path launches of ITS frames, and everything joins on `stream`
 * before the call's remaining launches - so whatever is enqueued on `stream` afterwards sees every slot decoded
 * (CCD_OPT_OVERLAP = 0: float stages only behind the join, as three ccd_batch_run_stage calls would do).
 * ccd_batch_prepare and ccd_batch_run[_stage] may use different streams: a run orders itself behind the tables' copy. */
int ccd_batch_run(ccd_batch* b, void* stream);
/* Enqueue only one stage (profiling / tests): 0 entropy, 1 upsampling, 2 synthesis(+integer planes). */
int ccd_batch_run_stage(ccd_batch* b, void* stream, int stage);
/* Waits for `stream` and returns the first per-slot decode error (CCD_ERR_INVALID_DATA ...). */
int ccd_batch_wait(ccd_batch* b, void* stream);
int ccd_batch_slot_status(const ccd_batch* b, int slot);
/* Raw per-slot counters of the entropy kernel after ccd_batch_wait: [0] status, [1] payload words read,
 * [2..3] symbols decoded (lo, hi); [36] latent grids whose body the pipelined kernel decoded as one stream of pixels
 * (batches cut without regard to wavefront steps: DESIGN.md 4.1); [37] batches its decoder took part by part; [39] pixels it
 * redid in int64 (dynamic operand check); [62] symbols that left the decoder's common path (window misses, sentinels), [63] of
 * which took the full 128-way search (pipelined kernel); the other words [4..61] are profiling cycle counters when built with
 * -DCCD_PIPE_PROFILE (which also reuses [36] and [37]).
 * `out64` receives 64 words. */
int ccd_batch_slot_stats(const ccd_batch* b, int slot, int32_t* out64);
/* Host only (tests): the chain groups ccd_batch_run would make of n slots with expected chain lengths est[] (any unit) and kernel
 * instantiations inst[] (0 .. k - 1; -1 = generic kernel) on a device with n_conc concurrent streams and n_cu CUs: cg[i] in 0 .. 2;
 * returns the number of groups in use.  See ccd_batch_run. */
int ccd_debug_chain_groups(const double* est, const int32_t* inst, int n, int n_conc, int n_cu, int32_t* cg);
/* How many of the library's side streams on `device` were MEASURED to run kernels concurrently (1 .. 4; once per process, ~10 ms
 * at the first call or the first ccd_batch_create): HIP multiplexes streams onto a few hardware queues, and launches on streams
 * that share one run one after the other.  The entropy launches of a batch that should overlap are put on these streams only,
 * and a batch is never split into more launches than that (environment: CCD_SIDE_STREAMS=k skips the measurement). */
int ccd_concurrent_streams(int device);
/* With CCD_OPT_TIME_LAUNCHES: duration in ms of every entropy launch of the LAST run (in launch order: longest expected chains
 * first) and the number of streams each holds; waits for those launches.  Returns the number of launches written (<= cap),
 * 0 when the option is off. */
int ccd_batch_launch_ms(ccd_batch* b, float* ms, int* n_streams, int cap);
/* Entropy launches per run of the batch as its launch tables were last built (ccd_batch_prepare / the first run after an add):
 * one per kernel instantiation in use and chain group (see ccd_batch_run); 0 before the tables exist. */
int ccd_batch_entropy_launches(const ccd_batch* b);
/* Which kernels serve this slot: bit 0 = pipelined entropy kernel (else the generic int64 one),
 * bit 1 = fused synthesis kernel (else one launch per layer), bit 2 = the whole float path (upsampling +
 * synthesis + integer samples) in one kernel, ccd_fused.hip, bit 3 = the ARM's layers evaluated on the matrix cores
 * inside the pipelined entropy kernel (exact limb-split int8, ccd_entropy_pipe.hip), bit 4 = the pipelined kernel's
 * instantiation that checks the IFCE features on the device (networks whose worst-case feature does not fit 16 bits),
 * bit 5 = the pipelined kernel's instantiation with a compile-time ARM shape (intra/hop.cfg: 14 + 6 inputs, two hidden layers),
 * bit 6 = the fused float kernel runs behind the batch's pyramid launch (CCD_OPT_FUSED_DEC = 2),
 * bit 7 = the network is OUTSIDE the finite envelope of the float stages (some latents could drive an intermediate value of the
 *         pyramid or the synthesis beyond float32: crafted or corrupt parameter payloads) - such a slot never runs the
 *         matrix-core kernel (bit 2 clear) but the vector-ALU kernels, which stay bit-identical with the reference
 *         arithmetic for inf and propagate NaN like torch.relu. */
int ccd_batch_slot_kernels(const ccd_batch* b, int slot);

/* Batch options, to be set before the slots they concern are added:
 *   CCD_OPT_FUSED_DEC   2 (default): slots whose architecture the fused float kernel covers (every decoder preset of
 *                       the reference, cfg/dec) run it behind ONE pyramid launch per batch that evaluates the latent
 *                       levels >= 1 once per frame (stage 1); the fused kernel's tiles load their level-1 footprint and do
 *                       level 0 + synthesis + integer samples (stage 2).  1: the fused kernel alone, the whole pyramid per
 *                       tile (one launch, nothing but the int8 latents read; ~20 % slower).  0: unfused path (per-level
 *                       upsampling launches + synthesis kernel), which materialises the dense stack ccd_batch_dense()
 *                       returns.  The three produce the same bits.
 *   CCD_OPT_KEEP_FLOAT  1 (default): the f32 synthesis output is always written (ccd_batch_output);
 *                       0: slots that produce integer planes directly (rgb / yuv444 intra frames) write only those.
 *   CCD_OPT_MFMA_ARM    0 (default): the integer ARM on the vector ALU; 1: streams inside the envelope (<= 20 ARM inputs,
 *                       <= 8 IFCE features, |weight| < 2^23, widest grid <= 2 500) evaluate it with
 *                       v_mfma_i32_16x16x64_i8 (exact limb-split int8).  Results are identical bit for bit either way; on
 *                       MI355X the matrix-core variant is the slower one, it is kept as a measured
 *                       alternative.  Values 2..22 lower the activation width above which a task is redone in plain
 *                       int64 - normally 23 bits - so that tests reach that path.
 *   CCD_OPT_RANGE_BITS  0 (default): production limit of the pipelined entropy kernel's dynamic operand check (an IFCE feature
 *                       with |f| >= 2^15 sends its pixel through the int64 redo).  Tests pass 8..14 to lower the limit and
 *                       drive ordinary streams through the redo; results are identical bit for bit.
 *                       ccd_batch_slot_stats word [39] counts the redone pixels; word [37] counts the batches the
 *                       pipelined kernel's decoder took part by part (a producer task at a time, because only the first
 *                       part's tables were there when it looked: DESIGN.md 4.1).
 *   CCD_OPT_OVERLAP     1 (default): ccd_batch_run overlaps the float path of the streams that finish early with the longest
 *                       entropy chains (chain groups, see ccd_batch_run); 0: one entropy launch per kernel instantiation and
 *                       every float launch behind the join.  Results are identical bit for bit (A/B, tests).  Environment:
 *                       CCD_OVERLAP=0.
 *   CCD_OPT_TIME_LAUNCHES  0 (default); 1: timing events around every entropy launch on the stream it runs on; ccd_batch_launch_ms
 *                       returns the durations of the last run (measurement only: bench.py's roofline). */
enum { CCD_OPT_FUSED_DEC = 1, CCD_OPT_KEEP_FLOAT = 2, CCD_OPT_MFMA_ARM = 3, CCD_OPT_RANGE_BITS = 4, CCD_OPT_OVERLAP = 5, CCD_OPT_TIME_LAUNCHES = 6 };
int ccd_batch_set_option(ccd_batch* b, int option, int value);

/* Device pointers of a slot's results (valid until the batch is destroyed / re-run): */
const float* ccd_batch_output(const ccd_batch* b, int slot);    /* [C][H][W] f32, synthesis output */
const float* ccd_batch_dense(const ccd_batch* b, int slot);     /* [L][H0][W0] f32, Upsampling.forward; NULL on the fused path */
const int8_t* ccd_batch_latent(const ccd_batch* b, int slot, int grid); /* [h][w] int8 */
/* Integer planes (value = round(x * (2^bitdepth-1)) after the reference's clamp/round/420 chain):
 * plane p of the frame, uint8 if bitdepth == 8 else uint16; chroma planes are half size for yuv420. */
const void* ccd_batch_plane(const ccd_batch* b, int slot, int plane, int* h, int* w);
/* Copies (device -> host, synchronous on `stream`) for tests and writers. */
int ccd_batch_copy_latent(ccd_batch* b, int slot, int grid, int8_t* host, void* stream);
int ccd_batch_copy_plane(ccd_batch* b, int slot, int plane, void* host, void* stream);
int ccd_batch_copy_output(ccd_batch* b, int slot, float* host, void* stream);
int ccd_batch_copy_dense(ccd_batch* b, int slot, float* host, void* stream);
/* The writer's path (decode.py:84-89 hands every frame to a file writer): a slot's three integer planes sit in ONE block of
 * device memory, plane p at byte offset off3[p] (256-byte aligned) of total_bytes.  ccd_batch_copy_planes_async enqueues one
 * device -> host copy per slot of [first_slot, first_slot + n_slots) into host_blocks[i] (same layout; pinned memory makes
 * them true DMA transfers) and returns without waiting: the caller waits on `stream` (ccd_batch_wait).  A slot whose decode
 * failed is skipped and its error returned. */
int ccd_batch_planes_layout(const ccd_batch* b, int slot, size_t* total_bytes, size_t* off3);
int ccd_batch_copy_planes_async(ccd_batch* b, int first_slot, int n_slots, void* const* host_blocks, void* stream);
/* Device and pinned-host blocks of destroyed batches are cached per device for the next batch (a batch per image set is the
 * normal use); this returns them to the runtime.  The environment variables CCD_POOL_MAX_MB / CCD_PINNED_POOL_MAX_MB cap the
 * caches of EACH device (defaults 16384 / 2048 MB per device = 5.5 % of it: the cache is invisible to other allocators of the process, e.g. PyTorch's; a block beyond
 * the cap is freed at once, and an allocation that fails trims the cache and retries).
 *
 * Threading and global state.  Per device and for the life of the process the library keeps: that block cache, the two
 * Laplace tables, ONE upload stream and eight side streams (entropy launches of one batch that need different kernel
 * instantiations - or hold streams of very different lengths - fork onto those of them that were measured to run concurrently,
 * ccd_concurrent_streams, and join the caller's stream again).  All of it is created under a lock; the fork / join
 * events belong to the batch.  Different batches may be driven from different host threads on one device; ONE batch is not
 * thread-safe.  ccd_batch_destroy drains every stream the caller passed to ccd_batch_run[_stage], ccd_batch_wait and
 * ccd_batch_copy_* before the batch's blocks return to the cache; work the caller enqueued on OTHER streams that reads
 * pointers obtained from ccd_batch_plane / _output / _latent must be finished by the caller before the destroy.  A stream
 * handed to any ccd_batch_* call must stay alive until the batch is destroyed (the destroy synchronises it). */
void ccd_pool_trim(int device);

/* ---- whole file: decode_video(), decode.py:26-91 ----------------------------------------- */
typedef struct {
    int32_t display_index, frame_type, frame_data_type, bitdepth;
    int32_t h, w, ch, cw;
    uint16_t* plane[3]; /* host, malloc'ed by the library, integer samples (u16 for every bitdepth) */
} ccd_frame;

typedef struct {
    int32_t n_frames;
    ccd_frame* frames; /* display order */
} ccd_video;

int ccd_decode_video(const uint8_t* bitstream, size_t n, int device, ccd_video* v);

/* ---- P / B frame reconstruction: decode.py:156-206 ------------------------------------------- */
/* All pointers are DEVICE pointers. residue = synthesis output of the "residue" cool-chic ([4][h][w] for P,
 * [5][h][w] for B: rgb/yuv residue, alpha, beta), motion = output of the "motion" cool-chic ([2] or [4][h][w]:
 * (x, y) flow per reference), refN_planes = the three integer planes of each reference frame (u8 if
 * bitdepth == 8 else u16; half-size chroma for yuv420), global_flow = (x, y) integer translation per reference
 * (frame header), warp_filter_size = 2 (bilinear) / 4 (bicubic grid_sample) / 6..16 even (sinc) as in warp.py:49-56.  Writes the integer planes of the frame. */
int ccd_inter_reconstruct(int device, void* stream, int frame_type, int h, int w, int bitdepth, int frame_data_type,
                          const float* residue, const float* motion, const void* const* ref0_planes,
                          const void* const* ref1_planes, const int32_t* global_flow, int warp_filter_size,
                          void* const* out_planes);
void ccd_video_free(ccd_video* v);

/* ---- bitstream writer + synthetic streams ("next-2" row of SURVEY section 8f) -------------- */
/* Range-encodes `n` symbols with per-symbol (mu_idx, scale_idx) table indices exactly as
 * constriction 0.4.2's RangeEncoder + QuantizedLaplace(-64,63) (rangecoder.py:46-76).
 * Returns the number of bytes written to *out (malloc'ed; free with ccd_free). */
int64_t ccd_range_encode(const int8_t* symbols, const int32_t* mu_idx, const int32_t* scale_idx, int64_t n,
                         uint8_t** out);
/* Bitstream writer for one intra frame (bitstream/encode.py:24-95): frames the video / frame /
 * cool-chic headers (architecture = the transmitted fields of `tmpl`), the NN payload `bytes_nn`
 * verbatim and the range-coded `latents` (latents[g] = int8 [grid_h[g]][grid_w[g]], values in
 * [-64, 63]).  The encoder walks the decoder's integer ARM/IFCE path on the HOST, like the
 * reference (latent.py:168-173).  Used by bench.py and the round-trip tests to manufacture
 * Kodak/CLIC/4K-shaped inputs; never used while decoding. */
int64_t ccd_encode_stream(const ccd_cc_header* tmpl, const uint8_t* bytes_nn, size_t n_nn,
                          const int8_t* const* latents, int bitdepth, int frame_data_type, uint8_t** out);
/* The cool-chic part alone (cool-chic header + NN payload + range-coded latents), for multi-frame /
 * multi-cool-chic streams assembled by the caller (bitstream/encode.py:83-92). */
int64_t ccd_encode_coolchic(const ccd_cc_header* tmpl, const uint8_t* bytes_nn, size_t n_nn,
                            const int8_t* const* latents, uint8_t** out);
/* Number of transmitted integers per (module, weight|bias) group for the architecture in `arch` (only the
 * transmitted fields are read), order arm.w arm.b ifce.w ifce.b ups.w ups.b syn.w syn.b
 * (component/core/types.py:18-19,98-101; neuralnet.py:46-71). */
int ccd_network_layout(const ccd_cc_header* arch, int64_t n_values[8]);
/* Exp-Golomb NN payload writer: encode_network + encode_exp_golomb (neuralnet.py:27-90, expgolomb.py:15-71).
 * `values` = the quantised parameters in stream order, orders taken from arch->nn_expgol_cnt.  Returns the
 * byte count (malloc'ed *out, free with ccd_free); *n_bit_pad = prefix padding bits for the header. */
int64_t ccd_encode_network(const ccd_cc_header* arch, const int32_t* values, int64_t n_values, int32_t* n_bit_pad,
                           uint8_t** out);
/* Header writers, AbstractHeader.to_bytes (header.py:90-105); n_bytes_header is computed. Return bytes written. */
int ccd_write_cc_header(const ccd_cc_header* h, uint8_t* out, size_t cap); /* transmitted fields only */
int ccd_write_frame_header(const ccd_frame_header* f, uint8_t* out, size_t cap);
int ccd_write_video_header(const ccd_video_header* v, uint8_t* out, size_t cap);
void ccd_free(void* p);

/* ---- PNG packing on the device (reference: coolchic/io/format/png.py:44-62 write_png, which hands an HWC uint8
 * array to PIL / zlib on the host).  r, g, b: device pointers to [h][w] uint8 planes (ccd_batch_plane); out: device
 * buffer, 4-byte aligned, of at least ccd_png_bound(h, w) bytes.  ccd_png_pack only enqueues work on `stream`;
 * ccd_png_finish synchronises the stream and returns the size of the file in `out` (or a negative code).  One pack
 * (single or batch) may be in flight per handle.  The file holds the filtered scanlines in literal-only dynamic-Huffman deflate blocks:
 * any PNG reader decodes exactly the input planes; the bytes differ from PIL's (PNG bytes are not normative). */
typedef struct ccd_png ccd_png;
typedef struct {
    const uint8_t *r, *g, *b; /* device planes [h][w] */
    int32_t h, w;
    uint8_t* out;             /* device buffer, 4-byte aligned */
    size_t cap;               /* >= ccd_png_bound(h, w) */
} ccd_png_item;
size_t ccd_png_bound(int h, int w);  /* 0 if a side is outside 1..16383 */
int ccd_png_create(int device, ccd_png** out);
void ccd_png_destroy(ccd_png* p);
int ccd_png_pack(ccd_png* p, const uint8_t* r, const uint8_t* g, const uint8_t* b, int h, int w, uint8_t* out,
                 size_t cap, void* stream);
int64_t ccd_png_finish(ccd_png* p, void* stream);
/* Many pictures in one set of launches (every deflate block of every picture is a workgroup of the same kernels);
 * ccd_png_finish_batch synchronises the stream and fills sizes[n] (n = the count given to the pack). */
int ccd_png_pack_batch(ccd_png* p, const ccd_png_item* items, int n, void* stream);
int ccd_png_finish_batch(ccd_png* p, void* stream, int64_t* sizes, int n);

/* ---- rate model (reference: coolchic/component/core/arm.py:448-485 compute_rate / _laplace_cdf, float32) ------
 * rate[i] = -log2(max(cdf(x+0.5) - cdf(x-0.5), 2^-16)) with the continuous Laplace(mu, scale) of the reference.
 * x, mu, scale, rate (optional), total_bits (optional, one double) are DEVICE pointers; asynchronous on `stream`. */
int ccd_compute_rate(int device, void* stream, const float* x, const float* mu, const float* scale, int64_t n,
                     float* rate, double* total_bits);

/* Leaky-quantised-Laplace boundaries computed ON THE GPU for a list of (mu_idx, scale_idx, s):
 * left[i], right[i] as the entropy kernel sees them (exhaustive parity tests of the f64 CDF). */
int ccd_debug_laplace_bounds(int device, const int32_t* mu_idx, const int32_t* scale_idx, const int32_t* s,
                             int64_t n, uint32_t* left, uint32_t* right);

/* Exhaustive form of the above (tools/cdf_sweep.py): the left cumulative of EVERY symbol s = -63 .. 63 for EVERY mu index and
 * the scale indices [scale_first, scale_first + n_scales), computed on the GPU by the production kernel's table builder
 * (which = 0: window_left, ccd_entropy_pipe.hip) or the generic kernel's (which = 1: laplace_left, ccd_entropy.hip).
 * out (host) receives n_scales * 32768 * 127 words, [scale][mu_idx][s + 63]. */
int ccd_debug_laplace_sweep(int device, int which, int scale_first, int n_scales, uint32_t* out);

/* Host only: 1 when this cool-chic's ARM runs on the pipelined entropy kernel: every ARM / stabiliser weight fits int32, no
 * hidden activation can leave int32 even for worst-case inputs, the worst-case IFCE feature fits the kernel's int32 side
 * plane (< 2^30), <= 32 ARM inputs, <= 8 layers, picture not wider than the symbol ring (5 060).  What depends on the data -
 * IFCE features as 16-bit operands - is checked per task on the device and the pixel redone in plain int64 (never taken on
 * any stream seen so far).  0 when it needs the generic 64-bit kernel (~7x slower; no network the reference encoder produced
 * does), < 0 on a malformed header / payload. */
int ccd_network_fits_fast_path(const uint8_t* cc_header, size_t n_hdr, const uint8_t* bytes_nn, size_t n_nn);
/* Host only: WHICH entropy-kernel instantiation a batch with default options gives this cool-chic (tests and DESIGN.md's
 * instantiation table): bit 0 = pipelined kernel (else the generic one), bit 4 = its instantiation with the device check of the
 * IFCE features (worst-case feature >= 2^15), bit 7 = the network is outside the finite envelope of the float stages (vector-ALU
 * float kernels only), bits 8..11 = NV = ceil(ARM inputs / 4), bits 12..15 = ARM layers (hidden + output).
 * Same bits 0, 4 and 7 as ccd_batch_slot_kernels.  < 0 on a malformed header / payload. */
int ccd_network_kernel_class(const uint8_t* cc_header, size_t n_hdr, const uint8_t* bytes_nn, size_t n_nn);

/* Profile builds only (-DCCD_FD_PROFILE): cycles per phase of the fused float kernel, summed over wave 0 of every
 * workgroup since the last reset; returns 1 with out16 filled, 0 when the library was built without the counters. */
int ccd_debug_fd_profile(uint64_t* out16, int reset);

#ifdef __cplusplus
}
#endif
#endif
