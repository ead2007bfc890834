/*
 * cc_oracle.c - CPU restatement of the Cool-chic 5.0 decoder (TEST INFRASTRUCTURE, see
 * cc_oracle.h). Plain C99 + libm. Integer stages follow the reference exactly (int64
 * with wrap-around like torch.int64); float stages define the build's canonical
 * accumulation order (fmaf chains, documented at each function) which the HIP kernels
 * reproduce bit-for-bit.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fPIC -shared cc_oracle.c -lm
 * (-ffp-contract=off: every fused multiply-add in the float stages is an explicit fmaf()).
 */
#include "cc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * 1. Bit reader: the reference turns the bytes into a string of '0'/'1' characters, most
 *    significant bit of each byte first (bitstream/header/header.py:73) and slices it.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t* p;
    size_t nbits, pos;
    int err;
} bits_t;

static void bits_init(bits_t* b, const uint8_t* p, size_t nbytes) {
    b->p = p; b->nbits = nbytes * 8; b->pos = 0; b->err = 0;
}

static uint64_t bits_read(bits_t* b, int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) {
        if (b->pos >= b->nbits) { b->err = 1; return 0; }
        int bit = (b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
        v = (v << 1) | (uint64_t)bit;
        b->pos++;
    }
    return v;
}

/* element.py:62-85 from_bits(signed=True): first bit is the sign, the rest the magnitude */
static int bits_read_signmag(bits_t* b, int n) {
    int neg = (int)bits_read(b, 1);
    int v = (int)bits_read(b, n - 1);
    return neg ? -v : v;
}

/* ------------------------------------------------------------------------------------------------
 * 2. Headers
 * ---------------------------------------------------------------------------------------------- */
/* header.py:130-147 VideoHeader: n_frames12 n_intras12 n_p_frames12 n_bytes_header16 | lists */
int ora_read_video_header(const uint8_t* p, size_t n, ora_video_header* h) {
    bits_t b; bits_init(&b, p, n);
    memset(h, 0, sizeof(*h));
    h->n_frames = (int)bits_read(&b, 12);
    h->n_intras = (int)bits_read(&b, 12);
    h->n_p_frames = (int)bits_read(&b, 12);
    h->n_bytes_header = (int)bits_read(&b, 16);
    for (int i = 0; i < h->n_intras; ++i) h->intra_pos[i] = (int)bits_read(&b, 12);
    for (int i = 0; i < h->n_p_frames; ++i) h->p_pos[i] = (int)bits_read(&b, 12);
    if (b.err || (size_t)h->n_bytes_header > n) return ORA_ERR_TRUNCATED;
    return h->n_bytes_header;
}

/* header.py:172-218 FrameHeader */
int ora_read_frame_header(const uint8_t* p, size_t n, ora_frame_header* h) {
    bits_t b; bits_init(&b, p, n);
    memset(h, 0, sizeof(*h));
    h->display_index = (int)bits_read(&b, 12);
    h->frame_type = (int)bits_read(&b, 2);
    h->frame_data_type = (int)bits_read(&b, 2);
    int bd_idx = (int)bits_read(&b, 4);
    h->n_bytes_header = (int)bits_read(&b, 16);
    if (b.err) return ORA_ERR_TRUNCATED;
    if (h->frame_type > 2) return ORA_ERR_VALUE; /* index into ("I","P","B") */
    if (bd_idx > 8) return ORA_ERR_VALUE;        /* index into (8..16), io/types.py */
    h->bitdepth = 8 + bd_idx;
    h->n_refs = h->frame_type == 2 ? 2 : (h->frame_type == 1 ? 1 : 0);
    for (int i = 0; i < h->n_refs; ++i) h->index_references[i] = (int)bits_read(&b, 12);
    for (int i = 0; i < 2 * h->n_refs; ++i) h->global_flow[i] = bits_read_signmag(&b, 14);
    if (h->n_refs) h->warp_filter_size = (int)bits_read(&b, 4);
    if (b.err || (size_t)h->n_bytes_header > n) return ORA_ERR_TRUNCATED;
    return h->n_bytes_header;
}

/* nnquant/quantstep.py:26-43 POSSIBLE_Q_STEP: number of entries and log2 of entry 0,
 * in header order arm.w arm.b ifce.w ifce.b ups.w ups.b syn.w syn.b */
static const int Q_STEP_N[8] = {9, 17, 9, 17, 13, 1, 13, 25};
static const int Q_STEP_LOG2_0[8] = {-8, -16, -8, -16, -12, 0, -12, -24};

/* header.py:244-307 CoolChicHeader */
int ora_read_cc_header(const uint8_t* p, size_t n, ora_cc_header* h) {
    bits_t b; bits_init(&b, p, n);
    memset(h, 0, sizeof(*h));
    h->linear_stabiliser_synth = (int)bits_read(&b, 1);
    h->n_layer_synthesis = (int)bits_read(&b, 3);
    h->ups_k_size = (int)bits_read(&b, 4);
    h->ups_preconcat_k_size = (int)bits_read(&b, 4);
    h->output_feature_ifce = (int)bits_read(&b, 5);
    h->spatial_context_arm = (int)bits_read(&b, 6);
    h->linear_stabiliser_arm = (int)bits_read(&b, 1);
    h->n_hidden_layers_arm = (int)bits_read(&b, 3);
    h->img_size[0] = (int)bits_read(&b, 14);
    h->img_size[1] = (int)bits_read(&b, 14);
    h->latent_resolution[0] = (int)bits_read(&b, 4);
    h->latent_resolution[1] = (int)bits_read(&b, 4);
    h->n_latent_grids = (int)bits_read(&b, 5);
    h->flag_hyperlatent = (int)bits_read(&b, 1);
    h->flag_common_randomness = (int)bits_read(&b, 1);
    h->final_upsampling_type = (int)bits_read(&b, 2);
    int q_idx[8];
    for (int i = 0; i < 8; ++i) q_idx[i] = (int)bits_read(&b, 5);
    for (int i = 0; i < 8; ++i) h->nn_expgol_cnt[i] = (int)bits_read(&b, 4);
    h->nn_n_bytes = (int)bits_read(&b, 14);
    h->nn_n_bit_pad = (int)bits_read(&b, 3);
    h->n_bytes_latent = (int)bits_read(&b, 28);
    h->n_bytes_header = (int)bits_read(&b, 16);
    if (b.err) return ORA_ERR_TRUNCATED;
    if (h->final_upsampling_type > 2) return ORA_ERR_VALUE;
    for (int i = 0; i < 8; ++i) {
        /* element.py:286-292: out-of-range list index -> ValueError */
        if (q_idx[i] >= Q_STEP_N[i]) return ORA_ERR_VALUE;
        h->nn_q_step_log2[i] = Q_STEP_LOG2_0[i] + q_idx[i];
        if (h->nn_expgol_cnt[i] > 12) return ORA_ERR_VALUE; /* nnquant/expgolomb.py:20-37 */
    }
    /* header.py:309-325 variable part */
    if (h->output_feature_ifce > 0) {
        h->has_ifce_resolution = 1;
        h->ifce_resolution[0] = (int)bits_read(&b, 4);
        h->ifce_resolution[1] = (int)bits_read(&b, 4);
    }
    if (h->flag_hyperlatent) {
        h->hyperlatent_resolution[0] = (int)bits_read(&b, 4);
        h->hyperlatent_resolution[1] = (int)bits_read(&b, 4);
    }
    for (int i = 0; i < h->n_layer_synthesis; ++i) { /* element.py:300-373 */
        h->syn_layer[i].out_ft = (int)bits_read(&b, 7);
        h->syn_layer[i].k_size = (int)bits_read(&b, 4);
        h->syn_layer[i].mode = (int)bits_read(&b, 1);
        h->syn_layer[i].nl = (int)bits_read(&b, 1);
    }
    if (b.err || (size_t)h->n_bytes_header > n) return ORA_ERR_TRUNCATED;
    return h->n_bytes_header;
}

static int ceil_div_pow2(int x, int i) { return (int)ceil((double)x / (double)(1 << i)); }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* component/core/coolchic.py:149-225 */
int ora_geometry_from_header(const ora_cc_header* h, ora_geometry* g) {
    memset(g, 0, sizeof(*g));
    int lo = h->latent_resolution[0], hi = h->latent_resolution[1];
    if (lo > hi) return ORA_ERR_VALUE;
    int mn = lo, mx = hi;
    if (h->flag_hyperlatent) { /* :163-168 */
        mn = imin(imin(lo, hi), imin(h->hyperlatent_resolution[0], h->hyperlatent_resolution[1]));
        mx = imax(imax(lo, hi), imax(h->hyperlatent_resolution[0], h->hyperlatent_resolution[1]));
    }
    for (int i = mn; i <= mx; ++i) { /* :170-183 */
        int gh = ceil_div_pow2(h->img_size[0], i), gw = ceil_div_pow2(h->img_size[1], i);
        if (lo <= i && i <= hi) {
            if (g->n_grids >= ORA_MAX_GRIDS) return ORA_ERR_VALUE;
            g->grid_h[g->n_grids] = gh; g->grid_w[g->n_grids] = gw; g->is_hyper[g->n_grids++] = 0;
        }
        if (h->flag_hyperlatent && h->hyperlatent_resolution[0] <= i && i <= h->hyperlatent_resolution[1]) {
            if (g->n_grids >= ORA_MAX_GRIDS) return ORA_ERR_VALUE;
            g->grid_h[g->n_grids] = gh; g->grid_w[g->n_grids] = gw; g->is_hyper[g->n_grids++] = 1;
        }
    }
    g->total_context_arm = h->spatial_context_arm + h->output_feature_ifce; /* :194 */
    g->input_feature_synthesis = (hi - lo + 1) * (h->flag_common_randomness ? 2 : 1); /* :203-207 */
    g->flag_ifce = h->has_ifce_resolution; /* :209 */
    for (int i = 0; i < g->n_grids; ++i) { /* :211-225 */
        if (h->img_size[0] <= 0 || g->grid_h[i] <= 0) return ORA_ERR_VALUE;
        int ratio = (int)ceil(log2((double)h->img_size[0] / (double)g->grid_h[i]));
        if (!g->flag_ifce) g->input_features_ifce[i] = 0;
        else if (h->ifce_resolution[0] <= ratio && ratio <= h->ifce_resolution[1])
            g->input_features_ifce[i] = imax(g->n_grids - 1 - i, 1);
        else g->input_features_ifce[i] = 0;
    }
    g->n_ups = hi; /* component/core/coolchic.py:1077-1086 */
    return ORA_OK;
}

/* ------------------------------------------------------------------------------------------------
 * 3. Exp-Golomb NN parameters: bitstream/neuralnet/expgolomb.py:74-130
 * ---------------------------------------------------------------------------------------------- */
int ora_decode_exp_golomb(const uint8_t* p, size_t n, int n_pad, const int* count, int n_val, int64_t* out) {
    bits_t b; bits_init(&b, p, n);
    bits_read(&b, n_pad); /* padding bits are a PREFIX of the message */
    for (int i = 0; i < n_val; ++i) {
        int n_bits_to_read = 1;
        for (;;) {
            if (b.pos >= b.nbits) return ORA_ERR_TRUNCATED;
            int bit = (b.p[b.pos >> 3] >> (7 - (b.pos & 7))) & 1;
            if (bit) break;
            n_bits_to_read++; b.pos++;
        }
        if (n_bits_to_read > 62) return ORA_ERR_VALUE;
        int64_t quotient = (int64_t)bits_read(&b, n_bits_to_read) - 1;
        int64_t remainder = count[i] ? (int64_t)bits_read(&b, count[i]) : 0;
        if (b.err) return ORA_ERR_TRUNCATED;
        int64_t value = ((int64_t)1 << count[i]) * quotient + remainder;
        if (value & 1) value = (value + 1) / 2; /* odd -> positive */
        else value = -(value / 2);              /* even -> non-positive */
        out[i] = value;
    }
    return ORA_OK;
}

/* ------------------------------------------------------------------------------------------------
 * 4. Network layout (bitstream/neuralnet/neuralnet.py:92-204, component/core/types.py:98-101)
 *    Stream order: modules arm, ifce, upsampling, synthesis; in each all weights then all biases,
 *    tensors in named_parameters() order.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int arm_dim, arm_layers;                      /* layers = n_hidden + 1 */
    int off_arm_w[ORA_MAX_ARM_LAYERS], off_arm_b[ORA_MAX_ARM_LAYERS];
    int off_arm_ws, off_arm_bs;                   /* -1 if no stabiliser */
    int off_ifce_w[ORA_MAX_GRIDS], off_ifce_b[ORA_MAX_GRIDS]; /* per grid, -1 if none */
    int n_ups, ups_wn, pre_wn;
    int off_ups_w, off_pre_w;                     /* [n_ups][ups_wn], [n_ups][pre_wn] */
    int syn_out, syn_in, syn_n_stab;
    int off_syn_ot_w, off_syn_ot_b, off_syn_st_w, off_syn_st_b;
    int off_syn_w[ORA_MAX_SYN_LAYERS], off_syn_b[ORA_MAX_SYN_LAYERS];
    int syn_cin[ORA_MAX_SYN_LAYERS];
    int total;
    int count_by_kind[8]; /* numbers of arm.w arm.b ifce.w ... */
} nn_layout;

static int build_layout(const ora_cc_header* h, const ora_geometry* g, nn_layout* L) {
    memset(L, 0, sizeof(*L));
    int off = 0;
    int dim = g->total_context_arm;
    L->arm_dim = dim; L->arm_layers = h->n_hidden_layers_arm + 1;
    /* arm weights: component/core/arm.py:176-195 */
    for (int l = 0; l < L->arm_layers; ++l) {
        int out = (l == L->arm_layers - 1) ? 2 : dim;
        L->off_arm_w[l] = off; off += out * dim;
    }
    if (h->linear_stabiliser_arm) { L->off_arm_ws = off; off += 2 * dim; } else L->off_arm_ws = -1;
    L->count_by_kind[0] = off;
    int s = off;
    for (int l = 0; l < L->arm_layers; ++l) {
        int out = (l == L->arm_layers - 1) ? 2 : dim;
        L->off_arm_b[l] = off; off += out;
    }
    if (h->linear_stabiliser_arm) { L->off_arm_bs = off; off += 2; } else L->off_arm_bs = -1;
    L->count_by_kind[1] = off - s;
    /* ifce: component/core/arm.py:325-343 (one linear arm per grid with input_ft > 0) */
    s = off;
    for (int i = 0; i < g->n_grids; ++i) {
        L->off_ifce_w[i] = -1; L->off_ifce_b[i] = -1;
        if (g->flag_ifce && g->input_features_ifce[i] > 0) {
            L->off_ifce_w[i] = off; off += h->output_feature_ifce * g->input_features_ifce[i];
        }
    }
    L->count_by_kind[2] = off - s; s = off;
    for (int i = 0; i < g->n_grids; ++i)
        if (g->flag_ifce && g->input_features_ifce[i] > 0) { L->off_ifce_b[i] = off; off += h->output_feature_ifce; }
    L->count_by_kind[3] = off - s; s = off;
    /* upsampling: component/core/upsampling.py:438-452; symmetric params (k+1)//2 each (:66-84) */
    L->n_ups = g->n_ups;
    if (L->n_ups > 0) {
        if (h->ups_k_size < 4 || (h->ups_k_size & 1)) return ORA_ERR_VALUE;      /* :224-226 assert */
        if (!(h->ups_preconcat_k_size & 1)) return ORA_ERR_VALUE;                 /* :112 assert */
    }
    L->ups_wn = (h->ups_k_size + 1) / 2; L->pre_wn = (h->ups_preconcat_k_size + 1) / 2;
    L->off_ups_w = off; off += L->n_ups * L->ups_wn;
    L->off_pre_w = off; off += L->n_ups * L->pre_wn;
    L->count_by_kind[4] = off - s; s = off;
    off += 2 * L->n_ups; /* biases: transmitted, never used (upsampling.py forward ignores them) */
    L->count_by_kind[5] = off - s; s = off;
    /* synthesis: component/core/synthesis.py:169-216 */
    if (h->n_layer_synthesis < 1) return ORA_ERR_VALUE;
    L->syn_in = g->input_feature_synthesis;
    L->syn_out = h->syn_layer[h->n_layer_synthesis - 1].out_ft;
    L->syn_n_stab = h->flag_common_randomness ? L->syn_in / 2 : L->syn_in;
    L->off_syn_ot_w = off; off += L->syn_out * L->syn_out;
    if (h->linear_stabiliser_synth) { L->off_syn_st_w = off; off += L->syn_out * L->syn_n_stab; } else L->off_syn_st_w = -1;
    int cin = L->syn_in;
    for (int l = 0; l < h->n_layer_synthesis; ++l) {
        int k = h->syn_layer[l].k_size;
        L->syn_cin[l] = cin;
        L->off_syn_w[l] = off; off += h->syn_layer[l].out_ft * cin * k * k;
        cin = h->syn_layer[l].out_ft;
    }
    L->count_by_kind[6] = off - s; s = off;
    L->off_syn_ot_b = off; off += L->syn_out;
    if (h->linear_stabiliser_synth) { L->off_syn_st_b = off; off += L->syn_out; } else L->off_syn_st_b = -1;
    for (int l = 0; l < h->n_layer_synthesis; ++l) { L->off_syn_b[l] = off; off += h->syn_layer[l].out_ft; }
    L->count_by_kind[7] = off - s;
    L->total = off;
    return ORA_OK;
}

/* ------------------------------------------------------------------------------------------------
 * 5. Fixed-point ARM parameters: bitstream/component/armint.py:30-170
 *    Output weights are TRANSPOSED: w[l] is [in][out].  Shifts are multiplications by 2^shift on
 *    int64 (torch wraps silently, so do we: use uint64 arithmetic).
 * ---------------------------------------------------------------------------------------------- */
static int64_t shl64(int64_t v, int s) { return (int64_t)((uint64_t)v << s); }

typedef struct {
    int n_layers, dim_in, n_out;
    int64_t* w[ORA_MAX_ARM_LAYERS]; /* [in_l][out_l] */
    int64_t* b[ORA_MAX_ARM_LAYERS];
    int in_l[ORA_MAX_ARM_LAYERS], out_l[ORA_MAX_ARM_LAYERS];
    int64_t* ws; /* [dim_in][n_out] */
    int64_t* bs; /* [n_out] */
} fp_arm;

static void fp_arm_free(fp_arm* a) {
    for (int l = 0; l < a->n_layers; ++l) { free(a->w[l]); free(a->b[l]); }
    free(a->ws); free(a->bs);
    memset(a, 0, sizeof(*a));
}

/* w_int[l]: [out][in] row-major (nn.Linear layout); b_int[l]: [out]; stab may be NULL */
static int arm_to_fixed_point_param(int n_layers, const int* in_l, const int* out_l, int64_t* const* w_int,
                                    int64_t* const* b_int, const int64_t* ws_int, const int64_t* bs_int,
                                    int qs_w_log2, int qs_b_log2, int subtract_last_layer, int n_inter_ft_ctx,
                                    int no_residual_layer, fp_arm* o) {
    memset(o, 0, sizeof(*o));
    o->n_layers = n_layers; o->dim_in = in_l[0]; o->n_out = out_l[n_layers - 1];
    const int WEIGHT_SHIFT = 16, N_FRAC_BIT_INTER_FT_CTX = 8; /* constants.py:17,39 */
    for (int l = 0; l < n_layers; ++l) {
        int in = in_l[l], out = out_l[l];
        o->in_l[l] = in; o->out_l[l] = out;
        o->w[l] = (int64_t*)calloc((size_t)in * out + 1, sizeof(int64_t));
        o->b[l] = (int64_t*)calloc((size_t)out + 1, sizeof(int64_t));
        if (!o->w[l] || !o->b[l]) return ORA_ERR_NOMEM;
        int is_last = (l == n_layers - 1);
        /* weights: armint.py:87-127 */
        int shift_w = WEIGHT_SHIFT + qs_w_log2;
        for (int oc = 0; oc < out; ++oc)
            for (int ic = 0; ic < in; ++ic) {
                int sh = shift_w;
                if (n_inter_ft_ctx > 0 && l == 0 && ic >= in - n_inter_ft_ctx) sh -= N_FRAC_BIT_INTER_FT_CTX;
                if (sh < 0) return ORA_ERR_VALUE; /* torch: negative integer power raises */
                int64_t v = shl64(w_int[l][oc * in + ic], sh);
                if (out == in && !no_residual_layer && oc == ic) { /* :114-124 residual folded into W */
                    int rs = WEIGHT_SHIFT;
                    if (n_inter_ft_ctx > 0 && l == 0 && ic >= in - n_inter_ft_ctx) rs -= N_FRAC_BIT_INTER_FT_CTX;
                    v = (int64_t)((uint64_t)v + ((uint64_t)1 << rs));
                }
                o->w[l][ic * out + oc] = v; /* transposed */
            }
        /* biases: armint.py:84-100,129-130 */
        int shift_b = 2 * WEIGHT_SHIFT + qs_b_log2;
        if (shift_b < 0) return ORA_ERR_VALUE;
        for (int oc = 0; oc < out; ++oc) {
            int64_t v = b_int[l][oc];
            if (is_last && subtract_last_layer && oc == 1)
                v = (int64_t)((uint64_t)v - ((uint64_t)4 << (-qs_b_log2))); /* :98-100 hard-coded -4 on log-scale */
            o->b[l][oc] = shl64(v, shift_b);
        }
    }
    int dim = o->dim_in, no = o->n_out;
    o->ws = (int64_t*)calloc((size_t)dim * no + 1, sizeof(int64_t));
    o->bs = (int64_t*)calloc((size_t)no + 1, sizeof(int64_t));
    if (!o->ws || !o->bs) return ORA_ERR_NOMEM;
    if (ws_int) { /* armint.py:134-157 */
        for (int oc = 0; oc < no; ++oc)
            for (int ic = 0; ic < dim; ++ic) {
                int sh = WEIGHT_SHIFT + qs_w_log2;
                if (n_inter_ft_ctx > 0 && ic >= dim - n_inter_ft_ctx) sh -= N_FRAC_BIT_INTER_FT_CTX;
                if (sh < 0) return ORA_ERR_VALUE;
                o->ws[ic * no + oc] = shl64(ws_int[oc * dim + ic], sh);
            }
        for (int oc = 0; oc < no; ++oc) o->bs[oc] = shl64(bs_int[oc], 2 * WEIGHT_SHIFT + qs_b_log2);
    } /* else zeros: :159-163 */
    return ORA_OK;
}

/* armint.py:180-203 fixed_point_arm for ONE context vector x[dim] -> out[n_out] */
static void fixed_point_arm(const fp_arm* a, const int64_t* x_in, int output_shift, int64_t* out) {
    int64_t x[128], y[128];
    int dim = a->dim_in;
    for (int i = 0; i < dim; ++i) x[i] = shl64(x_in[i], 16); /* x << WEIGHT_SHIFT */
    int64_t stab[128];
    for (int oc = 0; oc < a->n_out; ++oc) {
        uint64_t acc = (uint64_t)a->bs[oc];
        for (int ic = 0; ic < dim; ++ic) acc += (uint64_t)x[ic] * (uint64_t)a->ws[ic * a->n_out + oc];
        stab[oc] = (int64_t)acc;
    }
    for (int l = 0; l < a->n_layers - 1; ++l) {
        int in = a->in_l[l], outn = a->out_l[l];
        for (int oc = 0; oc < outn; ++oc) {
            uint64_t acc = (uint64_t)a->b[l][oc];
            for (int ic = 0; ic < in; ++ic) acc += (uint64_t)x[ic] * (uint64_t)a->w[l][ic * outn + oc];
            int64_t v = (int64_t)acc;
            if (v < 0) v = 0;   /* clamp_min_(0) */
            y[oc] = v >> 16;    /* arithmetic shift */
        }
        for (int oc = 0; oc < outn; ++oc) x[oc] = y[oc];
    }
    int l = a->n_layers - 1, in = a->in_l[l], outn = a->out_l[l];
    for (int oc = 0; oc < outn; ++oc) {
        uint64_t acc = (uint64_t)a->b[l][oc];
        for (int ic = 0; ic < in; ++ic) acc += (uint64_t)x[ic] * (uint64_t)a->w[l][ic * outn + oc];
        acc += (uint64_t)stab[oc];
        out[oc] = ((int64_t)acc) >> output_shift;
    }
}

/* ------------------------------------------------------------------------------------------------
 * 6. Leaky quantised Laplace + range coder (constriction 0.4.2, SURVEY.md appendix A;
 *    call sites bitstream/component/rangecoder.py:30-32,80-94)
 * ---------------------------------------------------------------------------------------------- */
static const uint32_t SCALE_BITS[2561] = {
#include "../include/ccd_scale_table.inc"
};

float ora_scale_table(int idx) {
    float f; memcpy(&f, &SCALE_BITS[idx], 4); return f;
}

#define AC_LO (-64)
#define AC_HI 63
#define RC_PRECISION 24

static double laplace_cdf(double x, double mu, double b) {
    if (x <= mu) return 0.5 * exp((x - mu) / b);
    return 1.0 - 0.5 * exp((mu - x) / b);
}

static int clip_i(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* left/right cumulative of symbol s under (mu_idx, scale_idx), after numpy take(mode="clip") */
void ora_laplace_bounds(int mu_idx, int scale_idx, int s, uint32_t* left, uint32_t* right) {
    mu_idx = clip_i(mu_idx, 0, 32767); scale_idx = clip_i(scale_idx, 0, 2560);
    double mu = (double)(float)(-64.0 + mu_idx / 256.0); /* float32 table value widened */
    double b = (double)ora_scale_table(scale_idx);
    const double free_w = (double)(((1u << RC_PRECISION) - 1u) - (uint32_t)(AC_HI - AC_LO));
    uint32_t slack = (uint32_t)(s - AC_LO);
    if (left) *left = (s == AC_LO) ? 0u : (uint32_t)(free_w * laplace_cdf((double)s - 0.5, mu, b)) + slack;
    if (right) *right = (s == AC_HI) ? (1u << RC_PRECISION)
                                    : (uint32_t)(free_w * laplace_cdf((double)s + 0.5, mu, b)) + slack + 1u;
}

/* Exhaustive check of a device's CDF (tools/cdf_sweep.py): dev[mu_idx * 127 + (s + 63)] must equal the left cumulative of
 * every symbol s = -63 .. 63 under (mu_idx, scale_idx) as computed here with libm.  Returns the number of mismatches and
 * records the first `cap` of them as (mu_idx, s, device value, libm value). */
int64_t ora_laplace_lefts_check(int scale_idx, const uint32_t* dev, int64_t* bad, int cap) {
    int64_t n_bad = 0;
    for (int mu_idx = 0; mu_idx < 32768; ++mu_idx)
        for (int s = AC_LO + 1; s <= AC_HI; ++s) {
            uint32_t l;
            ora_laplace_bounds(mu_idx, scale_idx, s, &l, NULL);
            const uint32_t d = dev[(size_t)mu_idx * 127 + (size_t)(s - AC_LO - 1)];
            if (d != l) {
                if (n_bad < cap) { bad[4 * n_bad] = mu_idx; bad[4 * n_bad + 1] = s; bad[4 * n_bad + 2] = d; bad[4 * n_bad + 3] = l; }
                ++n_bad;
            }
        }
    return n_bad;
}

typedef struct {
    const uint8_t* bytes;
    size_t n_words, pos;
    uint64_t lower, range, point;
} rc_decoder;

static uint32_t rc_next_word(rc_decoder* d) {
    uint32_t w = 0;
    if (d->pos < d->n_words) {
        const uint8_t* p = d->bytes + 4 * d->pos;
        w = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    }
    d->pos++;
    return w;
}

static void rc_decoder_init(rc_decoder* d, const uint8_t* bytes, size_t n_bytes) {
    d->bytes = bytes; d->n_words = n_bytes / 4; d->pos = 0;
    d->lower = 0; d->range = ~(uint64_t)0;
    d->point = (uint64_t)rc_next_word(d) << 32;
    d->point |= rc_next_word(d);
}

static int rc_decode(rc_decoder* d, int mu_idx, int scale_idx, int* sym) {
    uint64_t scale = d->range >> RC_PRECISION;
    uint64_t q = (d->point - d->lower) / scale;
    if (q >= ((uint64_t)1 << RC_PRECISION)) return ORA_ERR_INVALID_DATA;
    /* unique s with left(s) <= q < right(s): bisection on the (strictly increasing) left bound */
    int a = AC_LO, c = AC_HI;
    while (a < c) {
        int m = (a + c + 1) >> 1; /* floor for negatives too: values are > -128 so shift of sum is fine */
        uint32_t l; ora_laplace_bounds(mu_idx, scale_idx, m, &l, NULL);
        if ((uint64_t)l <= q) a = m; else c = m - 1;
    }
    uint32_t l, r; ora_laplace_bounds(mu_idx, scale_idx, a, &l, &r);
    d->lower += scale * (uint64_t)l;
    d->range = scale * (uint64_t)(r - l);
    if (d->range < ((uint64_t)1 << 32)) {
        d->lower <<= 32; d->range <<= 32;
        d->point = (d->point << 32) | rc_next_word(d);
    }
    *sym = a;
    return ORA_OK;
}

/* Incremental decoder handle: lets a caller that owns the symbol schedule (the reference's Python decoder behind an import
 * shim, tools/ref_baseline.py) drive this range decoder one wavefront at a time.  Test / baseline infrastructure only. */
struct ora_rc_decoder { rc_decoder d; uint8_t* copy; };
ora_rc_decoder* ora_rc_decoder_new(const uint8_t* bytes, size_t n_bytes) {
    ora_rc_decoder* h = (ora_rc_decoder*)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->copy = (uint8_t*)malloc(n_bytes ? n_bytes : 1);
    if (!h->copy) { free(h); return NULL; }
    memcpy(h->copy, bytes, n_bytes);
    rc_decoder_init(&h->d, h->copy, n_bytes);
    return h;
}
int ora_rc_decode_many(ora_rc_decoder* h, const int* mu_idx, const int* scale_idx, int n, int* out) {
    for (int i = 0; i < n; ++i) {
        int rc = rc_decode(&h->d, mu_idx[i], scale_idx[i], &out[i]);
        if (rc != ORA_OK) return rc;
    }
    return ORA_OK;
}
void ora_rc_decoder_free(ora_rc_decoder* h) { if (h) { free(h->copy); free(h); } }

struct ora_rc_encoder {
    uint64_t lower, range;
    int inv_active; uint64_t inv_n; uint32_t inv_first;
    uint32_t* out; size_t n_out, cap;
    int any;
};

static void enc_push(uint32_t** out, size_t* n, size_t* cap, uint32_t w) {
    if (*n == *cap) { *cap = *cap ? *cap * 2 : 1024; *out = (uint32_t*)realloc(*out, *cap * sizeof(uint32_t)); }
    (*out)[(*n)++] = w;
}

ora_rc_encoder* ora_rc_encoder_new(void) {
    ora_rc_encoder* e = (ora_rc_encoder*)calloc(1, sizeof(*e));
    e->range = ~(uint64_t)0;
    return e;
}

void ora_rc_encoder_free(ora_rc_encoder* e) { if (e) { free(e->out); free(e); } }

void ora_rc_encode(ora_rc_encoder* e, int s, int mu_idx, int scale_idx) {
    uint32_t l, r; ora_laplace_bounds(mu_idx, scale_idx, s, &l, &r);
    e->any = 1;
    uint64_t scale = e->range >> RC_PRECISION;
    e->range = scale * (uint64_t)(r - l);
    uint64_t nw = e->lower + scale * (uint64_t)l;
    if (e->inv_active && (uint64_t)(nw + e->range) > nw) { /* inverted -> normal */
        int carry = nw < e->lower;
        enc_push(&e->out, &e->n_out, &e->cap, carry ? e->inv_first + 1u : e->inv_first);
        for (uint64_t i = 1; i < e->inv_n; ++i) enc_push(&e->out, &e->n_out, &e->cap, carry ? 0u : 0xFFFFFFFFu);
        e->inv_active = 0;
    }
    e->lower = nw;
    if (e->range < ((uint64_t)1 << 32)) {
        uint32_t word = (uint32_t)(e->lower >> 32);
        e->lower <<= 32; e->range <<= 32;
        if (e->inv_active) e->inv_n++;
        else if ((uint64_t)(e->lower + e->range) > e->lower) enc_push(&e->out, &e->n_out, &e->cap, word);
        else { e->inv_active = 1; e->inv_n = 1; e->inv_first = word; }
    }
}

size_t ora_rc_get_compressed(const ora_rc_encoder* e, uint32_t** words) {
    uint32_t* out = NULL; size_t n = 0, cap = 0;
    for (size_t i = 0; i < e->n_out; ++i) enc_push(&out, &n, &cap, e->out[i]);
    if (e->any) {
        uint64_t point = e->lower + (((uint64_t)1 << 32) - 1);
        if (e->inv_active) {
            int carry = point < e->lower;
            enc_push(&out, &n, &cap, carry ? e->inv_first + 1u : e->inv_first);
            for (uint64_t i = 1; i < e->inv_n; ++i) enc_push(&out, &n, &cap, carry ? 0u : 0xFFFFFFFFu);
        }
        uint32_t pw = (uint32_t)(point >> 32);
        enc_push(&out, &n, &cap, pw);
        if ((uint32_t)((uint64_t)(e->lower + e->range) >> 32) == pw) enc_push(&out, &n, &cap, 0u);
    }
    *words = out;
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * 7. Context tables (component/core/arm.py:493-562) and the wavefront walk
 *    (bitstream/component/latent.py:18-187)
 * ---------------------------------------------------------------------------------------------- */
static const int PRIORITY_ORDER[40] = {
    38, 35, 30, 25, 23, 31, 36, 37, 39,
    33, 28, 21, 20, 6, 15, 22, 29, 34,
    32, 18, 12, 10, 5, 9, 14, 19, 27,
    24, 13, 8, 2, 1, 3, 11, 17, 26,
    16, 7, 4, 0};

/* arm.py:554-562: positions (in the 9x9 mask, row-major) of the n highest-priority neighbours */
static void non_zero_pixel_ctx_index(int n_ctx, int* idx) {
    for (int k = 0; k < n_ctx; ++k)
        for (int j = 0; j < 40; ++j)
            if (PRIORITY_ORDER[j] == k) { idx[k] = j; break; }
}

static int entropy_decode_grid(rc_decoder* rc, const fp_arm* arm, int n_spatial, int n_ifce, const int32_t* ctx_ifce,
                               int h, int w, int8_t* out, int32_t* mu_scale_out, uint64_t* n_sym) {
    const int MASK = 9, PAD = 4; /* arm.py:490 MAX_ARM_MASK_SIZE; latent.py:86 */
    int wp = w + 2 * PAD, hp = h + 2 * PAD;
    int64_t* data = (int64_t*)calloc((size_t)wp * hp, sizeof(int64_t)); /* data_to_fill, zero padded */
    if (!data) return ORA_ERR_NOMEM;
    int nz[40], offset[40];
    non_zero_pixel_ctx_index(n_spatial, nz);
    for (int k = 0; k < n_spatial; ++k) /* latent.py:190-237 compute_offset */
        offset[k] = (PAD - nz[k] % MASK) + (PAD - nz[k] / MASK) * wp;
    int rc_err = ORA_OK;
    int64_t ctx[128], ms[2];
    size_t k_out = 0;
    /* coding order x + 10*y (latent.py:240-265); raster when w <= 9 (:113-122) */
    long n_steps = (w <= MASK) ? (long)h * w : (long)(w + (MASK + 1) * (h - 1));
    for (long c = 0; c < n_steps && rc_err == ORA_OK; ++c) {
        int y0, x0, n;
        if (w <= MASK) { y0 = (int)(c / w); x0 = (int)(c % w); n = 1; }
        else {
            if (c < w) { y0 = 0; x0 = (int)c; }
            else { y0 = (int)((c - w) / (MASK + 1)) + 1; x0 = w - (MASK + 1) + (int)((c - w) % (MASK + 1)); }
            n = 0; /* occurrence count: pixels (y0+i, x0-10 i) inside the grid */
            while (y0 + n < h && x0 - (MASK + 1) * n >= 0) n++;
        }
        /* the reference computes all ARM outputs of the step first, then decodes them in order;
         * contexts of one step never overlap its own pixels, so a fused loop is identical */
        for (int i = 0; i < n; ++i) {
            int y = y0 + i, x = x0 - (MASK + 1) * i;
            int pos = wp * (PAD + y) + PAD + x;
            for (int k = 0; k < n_spatial; ++k) ctx[k] = data[pos - offset[k]];
            for (int k = 0; k < n_ifce; ++k) ctx[n_spatial + k] = ctx_ifce ? ctx_ifce[((size_t)k * h + y) * w + x] : 0;
            fixed_point_arm(arm, ctx, 2 * 16 - 8, ms);
            int mu_idx = (int)clip_i((int)(ms[0] < -1000000 ? -1000000 : (ms[0] > 1000000 ? 1000000 : ms[0])) + 16384, 0, 32767);
            int sc_idx = (int)clip_i((int)(ms[1] < -1000000 ? -1000000 : (ms[1] > 1000000 ? 1000000 : ms[1])) + 1280, 0, 2560);
            int s;
            rc_err = rc_decode(rc, mu_idx, sc_idx, &s);
            if (rc_err != ORA_OK) break;
            data[pos] = s;
            out[(size_t)y * w + x] = (int8_t)s;
            if (mu_scale_out) { mu_scale_out[2 * k_out] = mu_idx; mu_scale_out[2 * k_out + 1] = sc_idx; }
            k_out++;
        }
    }
    *n_sym += k_out;
    free(data);
    return rc_err;
}

/* ------------------------------------------------------------------------------------------------
 * 8. Upsampling (component/core/upsampling.py:463-500, training-mode 2-D branches :189-196,:312-325)
 *
 *    Canonical float order of this build (the reference delegates to oneDNN whose order is not
 *    specified):  K2[ky][kx] = w[ky]*w[kx] rounded to f32 (the kron kernel the reference
 *    materialises), acc = 0, then for ky ascending, kx ascending: acc = fmaf(x, K2, acc).
 * ---------------------------------------------------------------------------------------------- */
static void symmetric_1d(const float* p, int np, int k, float* w) { /* upsampling.py:42-64 */
    for (int i = 0; i < np; ++i) w[i] = p[i];
    /* x_reversed[k % 2:] */
    int j = np;
    for (int i = (k % 2); i < np; ++i) w[j++] = p[np - 1 - i];
}

/* TConv x2: in [h][w] -> out [2h][2w] (caller crops). upsampling.py:287-330 */
static void tconv_up2(const float* in, int h, int w, const float* w1d, int k, float* out) {
    int P0 = k / 2, C = 2 * P0 - 1 + k / 2;
    float K2[16][16];
    for (int a = 0; a < k; ++a) for (int b = 0; b < k; ++b) K2[a][b] = w1d[a] * w1d[b];
    int oh = 2 * h, ow = 2 * w;
    for (int a = 0; a < oh; ++a)
        for (int b = 0; b < ow; ++b) {
            int oy = a + C, ox = b + C;
            float acc = 0.0f;
            for (int ky = oy & 1; ky < k; ky += 2) {
                int iy = (oy - ky) / 2; /* index in the replicate-padded input */
                if (oy - ky < 0 || iy >= h + 2 * P0) continue;
                int sy = clip_i(iy - P0, 0, h - 1);
                for (int kx = ox & 1; kx < k; kx += 2) {
                    int ix = (ox - kx) / 2;
                    if (ox - kx < 0 || ix >= w + 2 * P0) continue;
                    int sx = clip_i(ix - P0, 0, w - 1);
                    acc = fmaf(in[(size_t)sy * w + sx], K2[ky][kx], acc);
                }
            }
            out[(size_t)a * ow + b] = acc;
        }
}

/* Pre-concatenation conv: zero pad, 2-D kron kernel, + x residual. upsampling.py:158-196 */
static void preconv(const float* in, int h, int w, const float* w1d, int k, float* out) {
    int pad = k / 2;
    float K2[16][16];
    for (int a = 0; a < k; ++a) for (int b = 0; b < k; ++b) K2[a][b] = w1d[a] * w1d[b];
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float acc = 0.0f;
            for (int ky = 0; ky < k; ++ky) {
                int sy = y + ky - pad;
                if (sy < 0 || sy >= h) continue; /* zero padding: the tap contributes nothing */
                for (int kx = 0; kx < k; ++kx) {
                    int sx = x + kx - pad;
                    if (sx < 0 || sx >= w) continue;
                    acc = fmaf(in[(size_t)sy * w + sx], K2[ky][kx], acc);
                }
            }
            out[(size_t)y * w + x] = acc + in[(size_t)y * w + x];
        }
}

/* ------------------------------------------------------------------------------------------------
 * 9. Synthesis (component/core/synthesis.py:61-76, 272-294)
 *    Canonical order: acc = bias; for ci, ky, kx ascending: acc = fmaf(w, x, acc); then
 *    (+ x if residual) then ReLU.  Replicate padding.
 * ---------------------------------------------------------------------------------------------- */
static void syn_conv(const float* in, int cin, int h, int w, const float* wt, const float* bias, int cout, int k,
                     int residual, int relu, float* out) {
    int pad = (k - 1) / 2;
    size_t plane = (size_t)h * w;
    for (int co = 0; co < cout; ++co)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float acc = bias[co];
                for (int ci = 0; ci < cin; ++ci)
                    for (int ky = 0; ky < k; ++ky) {
                        int sy = clip_i(y + ky - pad, 0, h - 1);
                        for (int kx = 0; kx < k; ++kx) {
                            int sx = clip_i(x + kx - pad, 0, w - 1);
                            acc = fmaf(wt[(((size_t)co * cin + ci) * k + ky) * k + kx], in[ci * plane + (size_t)sy * w + sx], acc);
                        }
                    }
                if (residual) acc = acc + in[co * plane + (size_t)y * w + x];
                if (relu) acc = acc <= 0.0f ? 0.0f : acc; /* torch.relu: NaN stays NaN (x > 0 ? x : 0 would drop it), -0 -> +0 */
                out[co * plane + (size_t)y * w + x] = acc;
            }
}

/* ------------------------------------------------------------------------------------------------
 * 9b. F.interpolate(mode = "bilinear" | "bicubic", align_corners=False) as the decoder uses it: the final
 *     resize (coolchic.py:187-189, scale = in / out) and the x2 steps of fixed_upsampling (upsampling.py:556-595,
 *     scale_factor=2 -> scale 0.5).  Index / weight rules follow ATen's UpSample.h (area_pixel_compute_source_index,
 *     guard_index_and_lambda, get_cubic_upsample_coefficients with A = -0.75), all in float32.
 *     Canon of this build: plain float ops for the coordinates and coefficients (no contraction), then
 *     row value t_j = w_x0 * v_0, t_j = fmaf(v_i, w_xi, t_j) (taps ascending) and out = w_y0 * t_0,
 *     out = fmaf(t_j, w_yj, out).
 * ---------------------------------------------------------------------------------------------- */
static int interp_taps(int dst, int in_size, float scale, int cubic, int idx[4], float wt[4]) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (!cubic && src < 0.0f) src = 0.0f;
    int i0 = (int)floorf(src);
    if (i0 > in_size - 1) i0 = in_size - 1;
    float t = src - (float)i0;
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    if (!cubic) {
        idx[0] = i0; idx[1] = i0 + (i0 < in_size - 1 ? 1 : 0);
        wt[0] = 1.0f - t; wt[1] = t;
        return 2;
    }
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = (1.0f - t) + 1.0f;
    wt[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    wt[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    wt[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    wt[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
    for (int j = 0; j < 4; ++j) idx[j] = clip_i(i0 - 1 + j, 0, in_size - 1);
    return 4;
}

/* in [c][h][w] -> out [c][H][W] (only the H x W top-left outputs of the virtual out_h x out_w result exist) */
static void resize_interp(const float* in, int c, int h, int w, float* out, int H, int W, int cubic, float scale_y, float scale_x) {
    for (int ch = 0; ch < c; ++ch)
        for (int y = 0; y < H; ++y) {
            int iy[4]; float wy[4];
            const int ny = interp_taps(y, h, scale_y, cubic, iy, wy);
            for (int x = 0; x < W; ++x) {
                int ix[4]; float wx[4];
                const int nx = interp_taps(x, w, scale_x, cubic, ix, wx);
                float acc = 0.0f;
                for (int j = 0; j < ny; ++j) {
                    const float* row = in + ((size_t)ch * h + iy[j]) * w;
                    float t = row[ix[0]] * wx[0];
                    for (int i = 1; i < nx; ++i) t = fmaf(row[ix[i]], wx[i], t);
                    acc = j == 0 ? t * wy[0] : fmaf(t, wy[j], acc);
                }
                out[((size_t)ch * H + y) * W + x] = acc;
            }
        }
}

/* ------------------------------------------------------------------------------------------------
 * 9c. Common randomness (component/core/noise.py:17-54; coolchic.py:1058-1061; coolchic.py(bitstream):179-183)
 *     Park-Miller LCG (seed 18101995, a = 7^5, m = 2^31 - 1), two draws per sample, Box-Muller in Python
 *     floats (f64), rounded to f32; one grid per latent level, finest first, row-major.
 *     Canon of this build: log and cos are restated with fixed fma sequences (< 2 ulp of f64, so the
 *     f32-rounded sample equals CPython's in all but ~1e-8 of the cases); the HIP kernel runs the same sequences.
 * ---------------------------------------------------------------------------------------------- */
static double sin_core(double r);
static double cos_core(double r);

static double cr_log(double x) { /* x normal, positive */
    uint64_t bits; memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1022;
    bits = (bits & 0x000fffffffffffffULL) | 0x3fe0000000000000ULL;
    double f; memcpy(&f, &bits, 8); /* [0.5, 1) */
    if (f < 0.70710678118654752440) { f = f * 2.0; e -= 1; }
    const double s = (f - 1.0) / (f + 1.0), s2 = s * s;
    double p = 1.0 / 23.0;
    p = fma(p, s2, 1.0 / 21.0); p = fma(p, s2, 1.0 / 19.0); p = fma(p, s2, 1.0 / 17.0); p = fma(p, s2, 1.0 / 15.0);
    p = fma(p, s2, 1.0 / 13.0); p = fma(p, s2, 1.0 / 11.0); p = fma(p, s2, 1.0 / 9.0); p = fma(p, s2, 1.0 / 7.0);
    p = fma(p, s2, 1.0 / 5.0); p = fma(p, s2, 1.0 / 3.0);
    const double r = fma(p * s2, 2.0 * s, 2.0 * s); /* 2 atanh(s) */
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, r)); /* ln2 hi, lo (fdlibm) */
}

static double cr_cos(double x) { /* 0 <= x < 8 */
    const double q = rint(x * 6.36619772367581382433e-01);
    double r = fma(-q, 1.57079632679489655800e+00, x);
    r = fma(-q, 6.12323399573676603587e-17, r);
    const int n = (int)q & 3;
    const double v = (n & 1) ? sin_core(r) : cos_core(r);
    return (n == 1 || n == 2) ? -v : v;
}

static void cr_noise(float* out, size_t n);
void ora_debug_cr_noise(float* out, size_t n) { cr_noise(out, n); }
void ora_debug_resize(const float* in, int c, int h, int w, float* out, int H, int W, int cubic, float scale_y, float scale_x) {
    resize_interp(in, c, h, w, out, H, W, cubic, scale_y, scale_x);
}
static void cr_noise(float* out, size_t n) {
    uint64_t seed = 18101995ULL;
    const uint64_t a = 16807ULL, m = 2147483647ULL;
    for (size_t i = 0; i < n; ++i) {
        seed = (a * seed) % m; const double u1 = (double)seed / (double)m;
        seed = (a * seed) % m; const double u2 = (double)seed / (double)m;
        const double g = sqrt(-2.0 * cr_log(u1)) * cr_cos((2.0 * 3.14159265359) * u2);
        out[i] = (float)g;
    }
}

/* ------------------------------------------------------------------------------------------------
 * 10. One cool-chic (bitstream/component/coolchic.py:29-207, decode mode)
 * ---------------------------------------------------------------------------------------------- */
void ora_cc_result_free(ora_cc_result* r) {
    if (!r) return;
    free(r->nn_ints);
    for (int l = 0; l < ORA_MAX_ARM_LAYERS; ++l) { free(r->arm_w[l]); free(r->arm_b[l]); }
    free(r->arm_ws);
    for (int i = 0; i < ORA_MAX_GRIDS; ++i) { free(r->latent[i]); free(r->mu_scale_idx[i]); free(r->ctx_ifce[i]); }
    free(r->dense); free(r->syn_out); free(r->out);
    memset(r, 0, sizeof(*r));
}

/* torch F.interpolate(mode="nearest", size=...) index rule */
static int nearest_src(int dst, int in_size, int out_size) {
    if (in_size == out_size) return dst;
    if (out_size == 2 * in_size) return dst >> 1;
    float scale = (float)in_size / (float)out_size;
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

int ora_decode_coolchic(const uint8_t* cc_header, size_t n_hdr, const uint8_t* bytes_nn, size_t n_nn,
                        const uint8_t* bytes_latent, size_t n_lat, int stop_after_entropy, ora_cc_result* r) {
    memset(r, 0, sizeof(*r));
    int rcod = ora_read_cc_header(cc_header, n_hdr, &r->hdr);
    if (rcod < 0) return rcod;
    const ora_cc_header* h = &r->hdr;
    if ((rcod = ora_geometry_from_header(h, &r->geo)) < 0) return rcod;
    const ora_geometry* g = &r->geo;
    nn_layout L;
    if ((rcod = build_layout(h, g, &L)) < 0) return rcod;
    if (g->total_context_arm > 100 || g->total_context_arm < 1) return ORA_ERR_VALUE;

    /* ---- decode_network: neuralnet.py:92-204 */
    int* count = (int*)malloc(sizeof(int) * (size_t)(L.total + 1));
    r->nn_ints = (int64_t*)calloc((size_t)L.total + 1, sizeof(int64_t));
    if (!count || !r->nn_ints) { free(count); return ORA_ERR_NOMEM; }
    {
        int o = 0;
        for (int kind = 0; kind < 8; ++kind)
            for (int i = 0; i < L.count_by_kind[kind]; ++i) count[o++] = h->nn_expgol_cnt[kind];
    }
    r->n_nn_ints = L.total;
    rcod = ora_decode_exp_golomb(bytes_nn, n_nn, h->nn_n_bit_pad, count, L.total, r->nn_ints);
    free(count);
    if (rcod < 0) return rcod;
    int64_t* P = r->nn_ints;

    /* ---- arm_to_fixed_point_param for the ARM: coolchic.py:72-77 */
    fp_arm arm;
    {
        int in_l[ORA_MAX_ARM_LAYERS], out_l[ORA_MAX_ARM_LAYERS];
        int64_t *wi[ORA_MAX_ARM_LAYERS], *bi[ORA_MAX_ARM_LAYERS];
        for (int l = 0; l < L.arm_layers; ++l) {
            in_l[l] = L.arm_dim; out_l[l] = (l == L.arm_layers - 1) ? 2 : L.arm_dim;
            wi[l] = P + L.off_arm_w[l]; bi[l] = P + L.off_arm_b[l];
        }
        rcod = arm_to_fixed_point_param(L.arm_layers, in_l, out_l, wi, bi,
                                        L.off_arm_ws >= 0 ? P + L.off_arm_ws : NULL,
                                        L.off_arm_bs >= 0 ? P + L.off_arm_bs : NULL,
                                        h->nn_q_step_log2[0], h->nn_q_step_log2[1], 1, h->output_feature_ifce, 0, &arm);
        if (rcod < 0) { fp_arm_free(&arm); return rcod; }
        r->arm_n_layers = arm.n_layers; r->arm_dim = arm.dim_in;
        for (int l = 0; l < arm.n_layers; ++l) {
            size_t nw = (size_t)arm.in_l[l] * arm.out_l[l], nb = (size_t)arm.out_l[l];
            r->arm_w[l] = (int64_t*)malloc(nw * 8); r->arm_b[l] = (int64_t*)malloc(nb * 8);
            memcpy(r->arm_w[l], arm.w[l], nw * 8); memcpy(r->arm_b[l], arm.b[l], nb * 8);
        }
        r->arm_ws = (int64_t*)malloc((size_t)arm.dim_in * 2 * 8);
        memcpy(r->arm_ws, arm.ws, (size_t)arm.dim_in * 2 * 8);
        r->arm_bs[0] = arm.bs[0]; r->arm_bs[1] = arm.bs[1];
    }

    /* ---- range decoder: coolchic.py:81-85, rangecoder.py:80-83 */
    if (n_lat % 4) { fp_arm_free(&arm); return ORA_ERR_VALUE; } /* np.frombuffer(uint32) raises */
    rc_decoder rc; rc_decoder_init(&rc, bytes_latent, n_lat);

    /* ---- grids, coarsest first: coolchic.py:89-169 */
    int n = g->n_grids;
    int n_ifce = h->output_feature_ifce;
    for (int idx = n - 1; idx >= 0 && rcod >= 0; --idx) {
        int hi = g->grid_h[idx], wi = g->grid_w[idx];
        /* fixed_upsampling(coded_latent, "nearest"): upsampling.py:556-595 */
        int uh, uw, uc;
        int64_t* ups = NULL; /* [uc][uh][uw] */
        if (idx == n - 1) {
            uh = hi; uw = wi; uc = 1;
            ups = (int64_t*)calloc((size_t)uh * uw, 8);
        } else {
            /* start from the smallest decoded grid and walk towards grid idx+1 */
            int ch = g->grid_h[n - 1], cw = g->grid_w[n - 1], cc = 1;
            int64_t* cur = (int64_t*)malloc((size_t)ch * cw * 8);
            for (int i = 0; i < ch * cw; ++i) cur[i] = r->latent[n - 1][i];
            for (int t = n - 2; t >= idx + 1; --t) {
                int th = g->grid_h[t], tw = g->grid_w[t];
                int64_t* nxt = (int64_t*)malloc((size_t)(cc + 1) * th * tw * 8);
                for (int i = 0; i < th * tw; ++i) nxt[i] = r->latent[t][i]; /* cat((target, x)) */
                for (int c = 0; c < cc; ++c)
                    for (int y = 0; y < th; ++y)
                        for (int x = 0; x < tw; ++x) {
                            int sy = y, sx = x;
                            if (ch != th || cw != tw) { sy = y >> 1; sx = x >> 1; } /* nearest x2 then crop */
                            nxt[((size_t)(c + 1) * th + y) * tw + x] = cur[((size_t)c * ch + sy) * cw + sx];
                        }
                free(cur); cur = nxt; ch = th; cw = tw; cc++;
            }
            ups = cur; uh = ch; uw = cw; uc = cc;
        }
        /* IFCE: coolchic.py:105-146 */
        int32_t* ctx = NULL;
        if (g->flag_ifce) {
            ctx = (int32_t*)calloc((size_t)n_ifce * hi * wi + 1, sizeof(int32_t));
            if (g->input_features_ifce[idx] != 0) {
                if (g->input_features_ifce[idx] != uc) { free(ups); free(ctx); fp_arm_free(&arm); return ORA_ERR_VALUE; }
                fp_arm ifce;
                int in_l[1] = {uc}, out_l[1] = {n_ifce};
                int64_t* wi_[1] = {P + L.off_ifce_w[idx]}; int64_t* bi_[1] = {P + L.off_ifce_b[idx]};
                rcod = arm_to_fixed_point_param(1, in_l, out_l, wi_, bi_, NULL, NULL, h->nn_q_step_log2[2],
                                                h->nn_q_step_log2[3], 0, 0, 1, &ifce);
                if (rcod >= 0) {
                    int64_t* feat = (int64_t*)malloc((size_t)uh * uw * n_ifce * 8);
                    int64_t xin[128], o[128];
                    for (int y = 0; y < uh; ++y)
                        for (int x = 0; x < uw; ++x) {
                            for (int c = 0; c < uc; ++c) xin[c] = ups[((size_t)c * uh + y) * uw + x];
                            fixed_point_arm(&ifce, xin, 2 * 16 - 8, o);
                            for (int c = 0; c < n_ifce; ++c)
                                feat[((size_t)c * uh + y) * uw + x] = (int64_t)(float)o[c]; /* .to(float) round trip, :142-144 */
                        }
                    /* nearest x2 and crop: :142-146 */
                    for (int c = 0; c < n_ifce; ++c)
                        for (int y = 0; y < hi; ++y)
                            for (int x = 0; x < wi; ++x)
                                ctx[((size_t)c * hi + y) * wi + x] = (int32_t)feat[((size_t)c * uh + (y >> 1)) * uw + (x >> 1)];
                    free(feat);
                }
                fp_arm_free(&ifce);
            }
        }
        free(ups);
        if (rcod < 0) { free(ctx); break; }
        r->ctx_ifce[idx] = ctx;
        r->latent[idx] = (int8_t*)calloc((size_t)hi * wi + 1, 1);
        r->mu_scale_idx[idx] = (int32_t*)calloc((size_t)hi * wi * 2 + 2, sizeof(int32_t));
        rcod = entropy_decode_grid(&rc, &arm, h->spatial_context_arm, g->flag_ifce ? n_ifce : 0, ctx, hi, wi,
                                   r->latent[idx], r->mu_scale_idx[idx], &r->n_symbols);
    }
    fp_arm_free(&arm);
    r->words_consumed = rc.pos;
    if (rcod < 0) return rcod;
    if (stop_after_entropy) return ORA_OK;

    /* ---- Upsampling.forward on the non-hyper grids: coolchic.py:175-177 */
    int lat_idx[ORA_MAX_GRIDS], n_lat_lv = 0;
    for (int i = 0; i < n; ++i) if (!g->is_hyper[i]) lat_idx[n_lat_lv++] = i;
    float wbuf[16], pbuf[16];
    int ch = g->grid_h[lat_idx[n_lat_lv - 1]], cw = g->grid_w[lat_idx[n_lat_lv - 1]], cc = 1;
    float* cur = (float*)malloc((size_t)ch * cw * sizeof(float));
    for (int i = 0; i < ch * cw; ++i) cur[i] = (float)r->latent[lat_idx[n_lat_lv - 1]][i];
    float q_ups = ldexpf(1.0f, h->nn_q_step_log2[4]);
    for (int step = 0; step < n_lat_lv - 1; ++step) {
        int t = lat_idx[n_lat_lv - 2 - step];
        int th = g->grid_h[t], tw = g->grid_w[t];
        if (L.n_ups < 1) { free(cur); return ORA_ERR_VALUE; }
        int kidx = step % L.n_ups;
        for (int i = 0; i < L.ups_wn; ++i) pbuf[i] = (float)P[L.off_ups_w + kidx * L.ups_wn + i] * q_ups;
        symmetric_1d(pbuf, L.ups_wn, h->ups_k_size, wbuf);
        float* nxt = (float*)malloc((size_t)(cc + 1) * th * tw * sizeof(float));
        float* big = (float*)malloc((size_t)4 * ch * cw * sizeof(float));
        for (int c = 0; c < cc; ++c) {
            tconv_up2(cur + (size_t)c * ch * cw, ch, cw, wbuf, h->ups_k_size, big);
            for (int y = 0; y < th; ++y) /* crop to the target size: upsampling.py:495 */
                memcpy(nxt + ((size_t)(c + 1) * th + y) * tw, big + (size_t)y * 2 * cw, (size_t)tw * sizeof(float));
        }
        free(big);
        for (int i = 0; i < L.pre_wn; ++i) pbuf[i] = (float)P[L.off_pre_w + kidx * L.pre_wn + i] * q_ups;
        symmetric_1d(pbuf, L.pre_wn, h->ups_preconcat_k_size, wbuf);
        float* tgt = (float*)malloc((size_t)th * tw * sizeof(float));
        for (int i = 0; i < th * tw; ++i) tgt[i] = (float)r->latent[t][i];
        preconv(tgt, th, tw, wbuf, h->ups_preconcat_k_size, nxt); /* channel 0 = high branch */
        free(tgt); free(cur);
        cur = nxt; ch = th; cw = tw; cc++;
    }
    if (h->flag_common_randomness) { /* coolchic.py:179-183 */
        /* torch.cat of [.., H_lo, W_lo] with the noise resized to img_size fails unless the sizes agree */
        if (ch != h->img_size[0] || cw != h->img_size[1]) { free(cur); return ORA_ERR_VALUE; }
        size_t n_noise = 0;
        for (int i = 0; i < n_lat_lv; ++i) n_noise += (size_t)g->grid_h[lat_idx[i]] * g->grid_w[lat_idx[i]];
        float* noise = (float*)malloc(n_noise * sizeof(float));
        cr_noise(noise, n_noise); /* size_per_latent_cr: finest level first (coolchic.py:187-191) */
        size_t* off = (size_t*)calloc((size_t)n_lat_lv + 1, sizeof(size_t));
        for (int i = 0, o = 0; i < n_lat_lv; ++i) { off[i] = (size_t)o; o += g->grid_h[lat_idx[i]] * g->grid_w[lat_idx[i]]; }
        /* fixed_upsampling(mode="bicubic"): coarsest first, x2 + crop, cat((target, x)) */
        int nh = g->grid_h[lat_idx[n_lat_lv - 1]], nw = g->grid_w[lat_idx[n_lat_lv - 1]], nc = 1;
        float* ncur = (float*)malloc((size_t)nh * nw * sizeof(float));
        memcpy(ncur, noise + off[n_lat_lv - 1], (size_t)nh * nw * sizeof(float));
        for (int lv = n_lat_lv - 2; lv >= 0; --lv) {
            const int th = g->grid_h[lat_idx[lv]], tw = g->grid_w[lat_idx[lv]];
            float* nn = (float*)malloc((size_t)(nc + 1) * th * tw * sizeof(float));
            memcpy(nn, noise + off[lv], (size_t)th * tw * sizeof(float));
            if (th != nh || tw != nw) resize_interp(ncur, nc, nh, nw, nn + (size_t)th * tw, th, tw, 1, 0.5f, 0.5f);
            else memcpy(nn + (size_t)th * tw, ncur, (size_t)nc * th * tw * sizeof(float));
            free(ncur); ncur = nn; nh = th; nw = tw; nc++;
        }
        /* F.interpolate(size=img_size, "bicubic") at equal sizes is the identity (coefficients 0, 1, 0, 0) */
        float* both = (float*)malloc((size_t)(cc + nc) * ch * cw * sizeof(float));
        memcpy(both, cur, (size_t)cc * ch * cw * sizeof(float));
        memcpy(both + (size_t)cc * ch * cw, ncur, (size_t)nc * ch * cw * sizeof(float));
        free(cur); free(ncur); free(noise); free(off);
        cur = both; cc += nc;
    }
    r->dense = cur; r->dense_c = cc; r->dense_h = ch; r->dense_w = cw;
    if (cc != g->input_feature_synthesis) return ORA_ERR_VALUE;

    /* ---- Synthesis.forward: synthesis.py:272-294 */
    float q_sw = ldexpf(1.0f, h->nn_q_step_log2[6]), q_sb = ldexpf(1.0f, h->nn_q_step_log2[7]);
    size_t plane = (size_t)ch * cw;
    float* x = (float*)malloc(plane * cc * sizeof(float));
    memcpy(x, cur, plane * cc * sizeof(float));
    int cin = cc;
    for (int l = 0; l < h->n_layer_synthesis; ++l) {
        int co = h->syn_layer[l].out_ft, k = h->syn_layer[l].k_size;
        if (!(k & 1)) { free(x); return ORA_ERR_UNSUPPORTED; }
        if (h->syn_layer[l].mode == 1 && co != cin) { free(x); return ORA_ERR_VALUE; }
        size_t nw = (size_t)co * cin * k * k;
        float* wt = (float*)malloc(nw * sizeof(float) + 4); float* bs = (float*)malloc((size_t)co * sizeof(float) + 4);
        for (size_t i = 0; i < nw; ++i) wt[i] = (float)P[L.off_syn_w[l] + i] * q_sw;
        for (int i = 0; i < co; ++i) bs[i] = (float)P[L.off_syn_b[l] + i] * q_sb;
        float* y = (float*)malloc(plane * (size_t)co * sizeof(float) + 4);
        syn_conv(x, cin, ch, cw, wt, bs, co, k, h->syn_layer[l].mode == 1, h->syn_layer[l].nl == 1, y);
        free(wt); free(bs); free(x);
        x = y; cin = co;
    }
    int so = L.syn_out;
    if (h->linear_stabiliser_synth) { /* main + stabiliser(x[:, :n_stab]) */
        size_t nw = (size_t)so * L.syn_n_stab;
        float* wt = (float*)malloc(nw * sizeof(float) + 4); float* bs = (float*)malloc((size_t)so * sizeof(float) + 4);
        for (size_t i = 0; i < nw; ++i) wt[i] = (float)P[L.off_syn_st_w + i] * q_sw;
        for (int i = 0; i < so; ++i) bs[i] = (float)P[L.off_syn_st_b + i] * q_sb;
        float* st = (float*)malloc(plane * (size_t)so * sizeof(float) + 4);
        syn_conv(cur, L.syn_n_stab, ch, cw, wt, bs, so, 1, 0, 0, st);
        for (size_t i = 0; i < plane * (size_t)so; ++i) x[i] = x[i] + st[i];
        free(st); free(wt); free(bs);
    }
    { /* output_transform: 1x1 conv */
        size_t nw = (size_t)so * so;
        float* wt = (float*)malloc(nw * sizeof(float) + 4); float* bs = (float*)malloc((size_t)so * sizeof(float) + 4);
        for (size_t i = 0; i < nw; ++i) wt[i] = (float)P[L.off_syn_ot_w + i] * q_sw;
        for (int i = 0; i < so; ++i) bs[i] = (float)P[L.off_syn_ot_b + i] * q_sb;
        float* y = (float*)malloc(plane * (size_t)so * sizeof(float) + 4);
        syn_conv(x, so, ch, cw, wt, bs, so, 1, 0, 0, y);
        free(wt); free(bs); free(x);
        x = y;
    }
    r->syn_out = x; r->out_c = so;

    /* ---- final interpolate + crop: coolchic.py:187-192 */
    int H = h->img_size[0], W = h->img_size[1];
    r->out_h = H; r->out_w = W;
    r->out = (float*)malloc((size_t)so * H * W * sizeof(float) + 4);
    if (ch == H && cw == W) { /* bicubic/bilinear/nearest at scale 1 are exact identities */
        memcpy(r->out, x, (size_t)so * H * W * sizeof(float));
    } else if (h->final_upsampling_type == 0) {
        for (int c = 0; c < so; ++c)
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx)
                    r->out[((size_t)c * H + yy) * W + xx] = x[c * plane + (size_t)nearest_src(yy, ch, H) * cw + nearest_src(xx, cw, W)];
    } else { /* bilinear / bicubic, scale = in / out as float (UpSample.h: area_pixel_compute_scale) */
        resize_interp(x, so, ch, cw, r->out, H, W, h->final_upsampling_type == 2, (float)ch / (float)H, (float)cw / (float)W);
    }
    return ORA_OK;
}

/* ------------------------------------------------------------------------------------------------
 * 11. Inter-frame reconstruction (bitstream/decode.py:156-189): global translation
 *     (component/intercoding/globalmotion.py:151-160), sinc-windowed N-tap warp in TRAINING mode
 *     (component/intercoding/warp.py:226-243, 294-397: no flow quantisation), alpha / beta blending.
 *
 *     Canon of this build: sin / cos are evaluated in double (pi/2 reduction + Taylor cores) and
 *     rounded to float - the correctly rounded float in all but ~1e-7 of the cases, i.e. what glibc's
 *     sinf (torch.sinc) returns; the two separable passes accumulate with fmaf, taps ascending.
 *     The HIP kernel uses the very same formulas (ccd_inter.hip).
 * ---------------------------------------------------------------------------------------------- */
static double sin_core(double r) { /* |r| <= pi/4 */
    const double r2 = r * r;
    double p = -7.6471637318198164759e-13;              /* -1/15! */
    p = fma(p, r2, 1.6059043836821614599e-10);           /*  1/13! */
    p = fma(p, r2, -2.5052108385441718775e-08);          /* -1/11! */
    p = fma(p, r2, 2.7557319223985890653e-06);           /*  1/9!  */
    p = fma(p, r2, -1.9841269841269841270e-04);          /* -1/7!  */
    p = fma(p, r2, 8.3333333333333333333e-03);           /*  1/5!  */
    p = fma(p, r2, -1.6666666666666666667e-01);          /* -1/3!  */
    return fma(p * r2, r, r);
}
static double cos_core(double r) {
    const double r2 = r * r;
    double p = 4.7794773323873852974e-14;                /*  1/16! */
    p = fma(p, r2, -1.1470745597729724714e-11);          /* -1/14! */
    p = fma(p, r2, 2.0876756987868098979e-09);           /*  1/12! */
    p = fma(p, r2, -2.7557319223985890653e-07);          /* -1/10! */
    p = fma(p, r2, 2.4801587301587301587e-05);           /*  1/8!  */
    p = fma(p, r2, -1.3888888888888888889e-03);          /* -1/6!  */
    p = fma(p, r2, 4.1666666666666666667e-02);           /*  1/4!  */
    p = fma(p, r2, -0.5);
    return fma(p, r2, 1.0);
}
static void sincos_f32(float a, float* s_out, float* c_out) {
    const double x = (double)a;
    const double q = rint(x * 6.36619772367581382433e-01); /* 2/pi */
    double r = fma(-q, 1.57079632679489655800e+00, x);     /* pi/2 hi */
    r = fma(-q, 6.12323399573676603587e-17, r);            /* pi/2 lo */
    const int n = (int)q & 3;
    const double sn = sin_core(r), cs = cos_core(r);
    double sv = (n & 1) ? cs : sn, cv = (n & 1) ? sn : cs;
    if (n & 2) sv = -sv;
    if (n == 1 || n == 2) cv = -cv;
    *s_out = (float)sv; *c_out = (float)cv;
}

/* warp.py:238-243: coeff[j] = cos(pi (s - rel_j) / N) * sinc(s - rel_j), rel_j = -N/2+1 .. N/2, all float32 */
static void sinc_coeffs(float s, int n_taps, float* coef) {
    const float pi_f = 3.14159265358979323846f;
    for (int j = 0; j < n_taps; ++j) {
        const float d = s - (float)(j - n_taps / 2 + 1);
        float sn, cs_unused, win_s_unused, win;
        sincos_f32(pi_f * d / (float)n_taps, &win_s_unused, &win);
        float snc = 1.0f;
        if (d != 0.0f) { const float a = pi_f * d; sincos_f32(a, &sn, &cs_unused); snc = sn / a; }
        coef[j] = win * snc;
    }
}

/* One warped sample: ref is [3][H][W] float (already 4:4:4), (gx, gy) the integer global translation. */
static void warp_pixel(const float* ref, int H, int W, int gx, int gy, int n_taps, float fx, float fy, int y, int x, float out[3]) {
    const float rxf = floorf(fx), ryf = floorf(fy);
    const float sx = fx - rxf, sy = fy - ryf;
    const int rx = (int)rxf, ry = (int)ryf;
    float cx[16], cy[16];
    sinc_coeffs(sx, n_taps, cx);
    sinc_coeffs(sy, n_taps, cy);
    const int lo = -(n_taps / 2) + 1;
    const size_t plane = (size_t)H * W;
    for (int c = 0; c < 3; ++c) {
        float acc = 0.0f;
        for (int i = 0; i < n_taps; ++i) {
            /* grid_sample(nearest, border) on the globally shifted reference: two clamps in sequence */
            const int yy = clip_i(clip_i(y + lo + i + ry, 0, H - 1) + gy, 0, H - 1);
            float line = 0.0f;
            for (int j = 0; j < n_taps; ++j) {
                const int xx = clip_i(clip_i(x + lo + j + rx, 0, W - 1) + gx, 0, W - 1);
                line = fmaf(ref[c * plane + (size_t)yy * W + xx], cx[j], line);
            }
            acc = fmaf(line, cy[i], acc);
        }
        out[c] = acc;
    }
}

/* ---- warp_filter_size 2 / 4: Warper's native path (warp.py:92-116, 325-343): F.grid_sample(mode = bilinear | bicubic,
 * padding_mode = "border", align_corners = True) on grid = backward_grid + flow / ((size - 1) / 2), flows unquantised
 * (training mode).  Canon = the float32 operation sequence of the reference run (PyTorch 2.10 CPU kernels as compiled
 * for this container, found by matching grid_sample bit for bit on random inputs, tests/golden/gen/warp_canon.py):
 *   linspace(-1, 1, n)[i]  = fma(step, i, -1) below n / 2, fma(-step, n - 1 - i, 1) from there on, step = 2 / (n - 1)
 *   source index           = (grid + 1) * ((size - 1) / 2)
 *   bilinear               clip to [0, size - 1]; weights s e, s w, n e, n w (plain products); nw * w0 then three fma
 *   bicubic (A = -0.75)    no clip of the index, every tap clipped; outer coefficients ((A x - 5A) x + 8A) x - 4A with
 *                          every step rounded, inner ones fma(fma(A + 2, x, -(A + 3)) * x, x, 1);
 *                          row = fma(p0, c0, p1 c1) + p2 c2 + p3 c3; column = fma chain over fma(r1, d1, r0 d0)
 * The integer global translation acts on the reference first (border replicate): two clamps per tap, as above. */
static float lin_coord(int i, int n) {
    if (n == 1) return -1.0f;
    const float step = 2.0f / (float)(n - 1);
    return i < n / 2 ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(n - 1 - i), 1.0f);
}
static float cubic_inner(float x) { /* |x| <= 1 */
    const float A = -0.75f;
    const float t = fmaf(A + 2.0f, x, -(A + 3.0f)) * x;
    return fmaf(t, x, 1.0f);
}
static float cubic_outer(float x) { /* 1 < |x| < 2 */
    const float A = -0.75f;
    float t = A * x;
    t = t - 5.0f * A;
    t = t * x;
    t = t + 8.0f * A;
    t = t * x;
    return t - 4.0f * A;
}
static void warp_pixel_native(const float* ref, int H, int W, int gx, int gy, int n_taps, float fx, float fy, int y, int x, float out[3]) {
    const float sx = (float)((W - 1.0) / 2.0), sy = (float)((H - 1.0) / 2.0);
    const float gxn = lin_coord(x, W) + fx / sx, gyn = lin_coord(y, H) + fy / sy;
    float ix = (gxn + 1.0f) * sx, iy = (gyn + 1.0f) * sy;
    const size_t plane = (size_t)H * W;
    if (n_taps == 2) {
        ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
        iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float w = ix - x0f, e = 1.0f - w, n = iy - y0f, s = 1.0f - n;
        const float w_nw = s * e, w_ne = s * w, w_sw = n * e, w_se = n * w;
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int xa = clip_i(clip_i(x0, 0, W - 1) + gx, 0, W - 1), xb = clip_i(clip_i(x0 + 1, 0, W - 1) + gx, 0, W - 1);
        const int ya = clip_i(clip_i(y0, 0, H - 1) + gy, 0, H - 1), yb = clip_i(clip_i(y0 + 1, 0, H - 1) + gy, 0, H - 1);
        for (int c = 0; c < 3; ++c) {
            const float* r = ref + c * plane;
            float acc = r[(size_t)ya * W + xa] * w_nw;
            acc = fmaf(r[(size_t)ya * W + xb], w_ne, acc);
            acc = fmaf(r[(size_t)yb * W + xa], w_sw, acc);
            acc = fmaf(r[(size_t)yb * W + xb], w_se, acc);
            out[c] = acc;
        }
        return;
    }
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float tx = ix - x0f, ty = iy - y0f;
    const float cx[4] = {cubic_outer(tx + 1.0f), cubic_inner(tx), cubic_inner(1.0f - tx), cubic_outer(2.0f - tx)};
    const float cy[4] = {cubic_outer(ty + 1.0f), cubic_inner(ty), cubic_inner(1.0f - ty), cubic_outer(2.0f - ty)};
    /* the index is a float in the reference: far-away flows saturate instead of wrapping */
    const int x0 = (int)fminf(fmaxf(x0f, -4.0f), (float)W + 4.0f), y0 = (int)fminf(fmaxf(y0f, -4.0f), (float)H + 4.0f);
    int xs[4];
    for (int j = 0; j < 4; ++j) xs[j] = clip_i(clip_i(x0 - 1 + j, 0, W - 1) + gx, 0, W - 1);
    for (int c = 0; c < 3; ++c) {
        float row[4];
        for (int i = 0; i < 4; ++i) {
            const int yy = clip_i(clip_i(y0 - 1 + i, 0, H - 1) + gy, 0, H - 1);
            const float* r = ref + c * plane + (size_t)yy * W;
            float acc = fmaf(r[xs[0]], cx[0], r[xs[1]] * cx[1]);
            acc = acc + r[xs[2]] * cx[2];
            acc = acc + r[xs[3]] * cx[3];
            row[i] = acc;
        }
        float acc = fmaf(row[1], cy[1], row[0] * cy[0]);
        acc = fmaf(row[2], cy[2], acc);
        acc = fmaf(row[3], cy[3], acc);
        out[c] = acc;
    }
}

/* decode.py:156-189 for one P / B frame. residue [3+1(+1)][H][W], motion [2(+2)][H][W], refs [3][H][W] each. */
static void reconstruct_inter(int frame_type, int H, int W, const float* residue, const float* motion, const float* ref0,
                              const float* ref1, const int* global_flow, int n_taps, float* out /* [3][H][W] */) {
    const size_t plane = (size_t)H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p = (size_t)y * W + x;
            float a = residue[3 * plane + p] + 0.5f;
            a = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);
            float w0[3], pred[3];
            (n_taps < 6 ? warp_pixel_native : warp_pixel)(ref0, H, W, global_flow[0], global_flow[1], n_taps, motion[p], motion[plane + p], y, x, w0);
            if (frame_type == 2) {
                float b = residue[4 * plane + p] + 0.5f;
                b = b < 0.0f ? 0.0f : (b > 1.0f ? 1.0f : b);
                float w1[3];
                (n_taps < 6 ? warp_pixel_native : warp_pixel)(ref1, H, W, global_flow[2], global_flow[3], n_taps, motion[2 * plane + p], motion[3 * plane + p], y, x, w1);
                for (int c = 0; c < 3; ++c) { const float t0 = b * w0[c], t1 = (1.0f - b) * w1[c]; pred[c] = t0 + t1; }
            } else {
                for (int c = 0; c < 3; ++c) pred[c] = w0[c];
            }
            for (int c = 0; c < 3; ++c) { const float m = a * pred[c]; out[c * plane + p] = m + residue[c * plane + p]; }
        }
}

/* ------------------------------------------------------------------------------------------------
 * 12. Whole stream (bitstream/decode.py:26-212).
 * ---------------------------------------------------------------------------------------------- */
void ora_video_free(ora_video* v) {
    if (!v || !v->frames) return;
    for (int i = 0; i < v->n_frames; ++i) for (int p = 0; p < 3; ++p) free(v->frames[i].plane[p]);
    free(v->frames);
    memset(v, 0, sizeof(*v));
}

/* decode.py:191-206 + writers (png.py:57-58, yuv.py:152-160): integer value of one sample */
static uint16_t quantise_sample(float x, float maxv) {
    float q = rintf(maxv * x) / maxv; /* torch.round = half to even */
    q = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
    q = rintf(q * maxv) / maxv;
    return (uint16_t)rintf(q * maxv);
}

/* decode.py:191-206 for a [3][H][W] float frame -> integer planes of `fr` */
static void finish_frame(const float* img, int H, int W, int bitdepth, int frame_data_type, ora_frame* fr) {
    const float maxv = (float)((1 << bitdepth) - 1);
    fr->h = H; fr->w = W;
    if (frame_data_type == 1) { /* yuv420: decode.py:191-206, yuv.py:274-300 */
        fr->ch = H / 2; fr->cw = W / 2;
        fr->plane[0] = (uint16_t*)malloc((size_t)H * W * 2);
        for (size_t i = 0; i < (size_t)H * W; ++i) fr->plane[0][i] = quantise_sample(img[i], maxv);
        for (int p = 1; p < 3; ++p) {
            fr->plane[p] = (uint16_t*)malloc((size_t)fr->ch * fr->cw * 2 + 2);
            const float* src = img + (size_t)p * H * W;
            for (int y = 0; y < fr->ch; ++y)
                for (int x = 0; x < fr->cw; ++x) {
                    /* round to the bit-depth grid, THEN average (decode.py:191 before :196);
                     * F.avg_pool2d: sequential f32 sum over the 2x2 window, divided by 4 */
                    float s = 0.0f;
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx)
                            s += rintf(maxv * src[(size_t)(2 * y + dy) * W + 2 * x + dx]) / maxv;
                    float a = s / 4.0f;
                    a = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);
                    a = rintf(a * maxv) / maxv;
                    fr->plane[p][(size_t)y * fr->cw + x] = (uint16_t)rintf(a * maxv);
                }
        }
    } else {
        fr->ch = H; fr->cw = W;
        for (int p = 0; p < 3; ++p) {
            fr->plane[p] = (uint16_t*)malloc((size_t)H * W * 2);
            const float* src = img + (size_t)p * H * W;
            for (size_t i = 0; i < (size_t)H * W; ++i) fr->plane[p][i] = quantise_sample(src[i], maxv);
        }
    }
}

/* FrameData of a decoded frame as the [3][H][W] float tensor decode_frame feeds to the warper
 * (decode.py:159-162: yuv420 references go through convert_420_to_444 = nearest x2 of u and v) */
static float* frame_as_444(const ora_frame* fr) {
    const int H = fr->h, W = fr->w;
    const float maxv = (float)((1 << fr->bitdepth) - 1);
    float* out = (float*)malloc((size_t)3 * H * W * sizeof(float));
    for (int p = 0; p < 3; ++p)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const int sy = (p && fr->frame_data_type == 1) ? y >> 1 : y, sx = (p && fr->frame_data_type == 1) ? x >> 1 : x;
                const int pw = p ? fr->cw : fr->w;
                out[((size_t)p * H + y) * W + x] = (float)fr->plane[p][(size_t)sy * pw + sx] / maxv;
            }
    return out;
}

int ora_decode_video(const uint8_t* bs, size_t n, ora_video* v) {
    memset(v, 0, sizeof(*v));
    ora_video_header* vh = (ora_video_header*)malloc(sizeof(ora_video_header));
    int used = ora_read_video_header(bs, n, vh);
    if (used < 0) { free(vh); return used; }
    int n_frames = vh->n_frames;
    free(vh);
    size_t pos = (size_t)used;
    v->n_frames = n_frames;
    v->frames = (ora_frame*)calloc((size_t)n_frames, sizeof(ora_frame));
    /* frames arrive in coding order; references are named by display index in each frame header
     * (the reference recomputes them from the coding structure, utils/codingstructure.py:267-436) */
    for (int f = 0; f < n_frames; ++f) {
        ora_frame_header fh;
        used = ora_read_frame_header(bs + pos, n - pos, &fh);
        if (used < 0) return used;
        pos += (size_t)used;
        if (fh.display_index >= n_frames) return ORA_ERR_VALUE;
        const int n_cc = fh.frame_type == 0 ? 1 : 2; /* decode.py:126-128: residue (+ motion) */
        ora_cc_result r[2];
        memset(r, 0, sizeof(r));
        int rc = ORA_OK;
        for (int c = 0; c < n_cc && rc == ORA_OK; ++c) {
            ora_cc_header ch;
            used = ora_read_cc_header(bs + pos, n - pos, &ch);
            if (used < 0) { rc = used; break; }
            const uint8_t* hdr = bs + pos; size_t n_hdr = (size_t)used;
            pos += n_hdr;
            if (pos + (size_t)ch.nn_n_bytes + (size_t)ch.n_bytes_latent > n) { rc = ORA_ERR_TRUNCATED; break; }
            rc = ora_decode_coolchic(hdr, n_hdr, bs + pos, (size_t)ch.nn_n_bytes, bs + pos + ch.nn_n_bytes,
                                     (size_t)ch.n_bytes_latent, 0, &r[c]);
            pos += (size_t)ch.nn_n_bytes + (size_t)ch.n_bytes_latent;
        }
        ora_frame* fr = &v->frames[fh.display_index];
        fr->display_index = fh.display_index; fr->frame_type = fh.frame_type;
        fr->frame_data_type = fh.frame_data_type; fr->bitdepth = fh.bitdepth;
        if (rc == ORA_OK) {
            const int H = r[0].out_h, W = r[0].out_w;
            if (fh.frame_type == 0) {
                if (r[0].out_c < 3) rc = ORA_ERR_VALUE;
                else finish_frame(r[0].out, H, W, fh.bitdepth, fh.frame_data_type, fr);
            } else {
                const int need_res = fh.frame_type == 1 ? 4 : 5, need_mot = fh.frame_type == 1 ? 2 : 4;
                if (r[0].out_c < need_res || r[1].out_c < need_mot || r[1].out_h != H || r[1].out_w != W ||
                    fh.warp_filter_size < 2 || fh.warp_filter_size > 16 || (fh.warp_filter_size & 1)) rc = ORA_ERR_VALUE; /* warp.py:41-47 asserts */
                float* refs[2] = {NULL, NULL};
                for (int k = 0; k < fh.n_refs && rc == ORA_OK; ++k) {
                    const int ri = fh.index_references[k];
                    if (ri >= n_frames || !v->frames[ri].plane[0] || v->frames[ri].h != H || v->frames[ri].w != W) rc = ORA_ERR_VALUE;
                    else refs[k] = frame_as_444(&v->frames[ri]);
                }
                if (rc == ORA_OK) {
                    float* img = (float*)malloc((size_t)3 * H * W * sizeof(float));
                    reconstruct_inter(fh.frame_type, H, W, r[0].out, r[1].out, refs[0], refs[1], fh.global_flow,
                                      fh.warp_filter_size, img);
                    finish_frame(img, H, W, fh.bitdepth, fh.frame_data_type, fr);
                    free(img);
                }
                free(refs[0]); free(refs[1]);
            }
        }
        ora_cc_result_free(&r[0]); ora_cc_result_free(&r[1]);
        if (rc < 0) return rc;
    }
    return ORA_OK;
}
