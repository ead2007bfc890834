/*
 * cc_oracle.h - CPU restatement of the Cool-chic 5.0 decode path (TEST INFRASTRUCTURE).
 *
 * This is the parity oracle for the MI355X decoder in cool_chic_amd/. It is plain C,
 * single-threaded, written for clarity and not speed. Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call it; the product path never does.
 *
 * Parity pin: validated against golden vectors produced by importing the reference
 * (/root/reference, this container only) - see tests/golden/gen/dump_reference.py and
 * tests/test_oracle_golden.py. The range coder / leaky Laplace quantiser live in the
 * un-vendored third-party crate constriction==0.4.2 (requirements.txt:10); they are
 * restated from its published algorithm and pinned by the
 * reference's shipped bitstream samples/bitstreams/kodim14.cool, which decodes to the
 * reference's latents bit-exactly and re-encodes to the same payload byte-for-byte.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/coolchic unless noted).
 */
#ifndef CC_ORACLE_H
#define CC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_MAX_GRIDS 40
#define ORA_MAX_SYN_LAYERS 8
#define ORA_MAX_ARM_LAYERS 9 /* n_hidden_layers_arm is 3 bits -> <= 7 hidden + 1 out */

enum { ORA_OK = 0, ORA_ERR_TRUNCATED = -1, ORA_ERR_VALUE = -2, ORA_ERR_INVALID_DATA = -3,
       ORA_ERR_UNSUPPORTED = -4, ORA_ERR_NOMEM = -5 };

/* bitstream/header/header.py:130-147 */
typedef struct {
    int n_frames, n_intras, n_p_frames, n_bytes_header;
    int intra_pos[4096];
    int p_pos[4096];
} ora_video_header;

/* bitstream/header/header.py:172-218 */
typedef struct {
    int display_index;
    int frame_type;      /* 0 I, 1 P, 2 B   (utils/codingstructure.py:20) */
    int frame_data_type; /* 0 rgb, 1 yuv420, 2 yuv444, 3 flow (io/types.py) */
    int bitdepth;        /* 8..16 */
    int n_bytes_header;
    int n_refs;
    int index_references[2];
    int global_flow[4];
    int warp_filter_size;
} ora_frame_header;

typedef struct { int out_ft, k_size, mode /*0 linear 1 residual*/, nl /*0 none 1 relu*/; } ora_syn_layer;

/* bitstream/header/header.py:244-307 */
typedef struct {
    int linear_stabiliser_synth, n_layer_synthesis, ups_k_size, ups_preconcat_k_size;
    int output_feature_ifce, spatial_context_arm, linear_stabiliser_arm, n_hidden_layers_arm;
    int img_size[2];
    int latent_resolution[2];
    int n_latent_grids;
    int flag_hyperlatent, flag_common_randomness;
    int final_upsampling_type; /* 0 nearest 1 bilinear 2 bicubic */
    int nn_q_step_log2[8];     /* arm.w arm.b ifce.w ifce.b ups.w ups.b syn.w syn.b : log2 of the step */
    int nn_expgol_cnt[8];
    int nn_n_bytes, nn_n_bit_pad, n_bytes_latent, n_bytes_header;
    int has_ifce_resolution;
    int ifce_resolution[2];
    int hyperlatent_resolution[2];
    ora_syn_layer syn_layer[ORA_MAX_SYN_LAYERS];
} ora_cc_header;

/* component/core/coolchic.py:149-225 (CoolChicEncoderParameter.__post_init__) */
typedef struct {
    int n_grids;
    int grid_h[ORA_MAX_GRIDS], grid_w[ORA_MAX_GRIDS];
    int is_hyper[ORA_MAX_GRIDS];
    int input_features_ifce[ORA_MAX_GRIDS];
    int flag_ifce;
    int input_feature_synthesis;
    int total_context_arm;
    int n_ups; /* number of upsampling / pre-concat kernels = latent_resolution[1] */
} ora_geometry;

/* Everything one cool-chic decode produces (all buffers malloc'ed, free with ora_cc_result_free). */
typedef struct {
    ora_cc_header hdr;
    ora_geometry geo;
    int n_nn_ints;
    int64_t* nn_ints;                 /* decoded Exp-Golomb integers, stream order */
    /* fixed-point ARM (armint.py:30-170): w[l] is [in][out] row-major */
    int arm_n_layers;
    int arm_dim;
    int64_t* arm_w[ORA_MAX_ARM_LAYERS];
    int64_t* arm_b[ORA_MAX_ARM_LAYERS];
    int64_t* arm_ws; /* [dim][2] */
    int64_t arm_bs[2];
    /* latents, index = grid index (0 finest), int8 [h][w] */
    int8_t* latent[ORA_MAX_GRIDS];
    /* (mu_idx, scale_idx) pairs in decode order, per grid (int32 pairs) */
    int32_t* mu_scale_idx[ORA_MAX_GRIDS];
    /* IFCE context features fed to the ARM for each grid: int32 [C][h][w] or NULL */
    int32_t* ctx_ifce[ORA_MAX_GRIDS];
    uint64_t n_symbols;
    uint64_t words_consumed;
    /* float stages */
    int dense_c, dense_h, dense_w;
    float* dense;  /* [C][h][w] Upsampling.forward output */
    int out_c, out_h, out_w;
    float* syn_out; /* Synthesis.forward output [C][dense_h][dense_w] */
    float* out;     /* after final interpolate+crop: [C][img_h][img_w] */
} ora_cc_result;

/* ---- headers (return bytes consumed >0, or error <0) ------------------------------ */
int ora_read_video_header(const uint8_t* p, size_t n, ora_video_header* h);
int ora_read_frame_header(const uint8_t* p, size_t n, ora_frame_header* h);
int ora_read_cc_header(const uint8_t* p, size_t n, ora_cc_header* h);
int ora_geometry_from_header(const ora_cc_header* h, ora_geometry* g);

/* ---- NN -------------------------------------------------------------------------- */
/* bitstream/neuralnet/expgolomb.py:74-130 */
int ora_decode_exp_golomb(const uint8_t* p, size_t n, int n_pad, const int* count, int n_val, int64_t* out);

/* ---- entropy model ---------------------------------------------------------------- */
/* Leaky quantised Laplace (constriction 0.4.2): boundaries of symbol s. */
void ora_laplace_bounds(int mu_idx, int scale_idx, int s, uint32_t* left, uint32_t* right);
/* dev[mu_idx * 127 + (s + 63)], s = -63 .. 63, against libm for one scale index; mismatches -> bad[4 i ..] = (mu_idx, s, dev, libm) */
int64_t ora_laplace_lefts_check(int scale_idx, const uint32_t* dev, int64_t* bad, int cap);
float ora_scale_table(int idx);

/* ---- one cool-chic: bitstream/component/coolchic.py:29-207 (mode="decode") ---------- */
int ora_decode_coolchic(const uint8_t* cc_header, size_t n_hdr, const uint8_t* bytes_nn, size_t n_nn,
                        const uint8_t* bytes_latent, size_t n_lat, int stop_after_entropy,
                        ora_cc_result* r);
void ora_cc_result_free(ora_cc_result* r);

/* ---- whole stream: bitstream/decode.py:26-212 -------------------------------------- */
typedef struct {
    int display_index, frame_type, frame_data_type, bitdepth;
    int h, w;           /* luma / RGB size */
    int ch, cw;         /* chroma size (== h,w unless yuv420) */
    uint16_t* plane[3]; /* integer planes, value = round(x * (2^bitdepth - 1)) */
} ora_frame;

typedef struct {
    int n_frames;
    ora_frame* frames; /* display order */
} ora_video;

int ora_decode_video(const uint8_t* bitstream, size_t n, ora_video* v);
void ora_video_free(ora_video* v);

/* ---- range encoder (constriction RangeEncoder) for round trips --- */
typedef struct ora_rc_decoder ora_rc_decoder;
ora_rc_decoder* ora_rc_decoder_new(const uint8_t* bytes, size_t n_bytes);
int ora_rc_decode_many(ora_rc_decoder* h, const int* mu_idx, const int* scale_idx, int n, int* out);
void ora_rc_decoder_free(ora_rc_decoder* h);
typedef struct ora_rc_encoder ora_rc_encoder;
ora_rc_encoder* ora_rc_encoder_new(void);
void ora_rc_encode(ora_rc_encoder* e, int s, int mu_idx, int scale_idx);
/* seals a copy; returns number of u32 words written to *words (malloc'ed) */
size_t ora_rc_get_compressed(const ora_rc_encoder* e, uint32_t** words);
void ora_rc_encoder_free(ora_rc_encoder* e);

#ifdef __cplusplus
}
#endif
/* debug hooks for the unit tests of sections 9b / 9c */
void ora_debug_cr_noise(float* out, size_t n);
void ora_debug_resize(const float* in, int c, int h, int w, float* out, int H, int W, int cubic, float scale_y, float scale_x);

#endif
