// ccd_format.cpp - see ccd_format.hpp.
#include "ccd_format.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ccd {

// ---------------------------------------------------------------------------------------------
// Bit cursor
// ---------------------------------------------------------------------------------------------
uint64_t BitReader::get(int n) {
    if (pos_ + static_cast<size_t>(n) > n_bits_) { bad_ = true; pos_ = n_bits_; return 0; }
    uint64_t v = 0;
    size_t pos = pos_;
    int left = n;
    while (left > 0) {  // take up to 8 bits per step
        const int bit_in_byte = static_cast<int>(pos & 7);
        const int take = std::min(left, 8 - bit_in_byte);
        const unsigned chunk = (p_[pos >> 3] >> (8 - bit_in_byte - take)) & ((1u << take) - 1u);
        v = (v << take) | chunk;
        pos += take; left -= take;
    }
    pos_ = pos;
    return v;
}

int BitReader::get_sign_magnitude(int n) {
    const bool neg = get(1) != 0;
    const int mag = static_cast<int>(get(n - 1));
    return neg ? -mag : mag;
}

int BitReader::peek_bit() const {
    if (pos_ >= n_bits_) return -1;
    return (p_[pos_ >> 3] >> (7 - (pos_ & 7))) & 1;
}

// ---------------------------------------------------------------------------------------------
// Headers
// ---------------------------------------------------------------------------------------------
int read_video_header(const uint8_t* p, size_t n, ccd_video_header* h) {
    std::memset(h, 0, sizeof(*h));
    BitReader br(p, n);
    h->n_frames = static_cast<int32_t>(br.get(12));
    h->n_intras = static_cast<int32_t>(br.get(12));
    h->n_p_frames = static_cast<int32_t>(br.get(12));
    h->n_bytes_header = static_cast<int32_t>(br.get(16));
    for (int i = 0; i < h->n_intras; ++i) h->intra_pos[i] = static_cast<int32_t>(br.get(12));
    for (int i = 0; i < h->n_p_frames; ++i) h->p_pos[i] = static_cast<int32_t>(br.get(12));
    if (br.bad() || static_cast<size_t>(h->n_bytes_header) > n) return CCD_ERR_TRUNCATED;
    if (h->n_bytes_header <= 0) return CCD_ERR_VALUE;
    return h->n_bytes_header;
}

int read_frame_header(const uint8_t* p, size_t n, ccd_frame_header* h) {
    std::memset(h, 0, sizeof(*h));
    BitReader br(p, n);
    h->display_index = static_cast<int32_t>(br.get(12));
    h->frame_type = static_cast<int32_t>(br.get(2));
    h->frame_data_type = static_cast<int32_t>(br.get(2));
    const int bitdepth_index = static_cast<int>(br.get(4));
    h->n_bytes_header = static_cast<int32_t>(br.get(16));
    if (br.bad()) return CCD_ERR_TRUNCATED;
    if (h->frame_type > 2 || bitdepth_index > 8) return CCD_ERR_VALUE;
    h->bitdepth = 8 + bitdepth_index;
    h->n_refs = h->frame_type;  // I:0 P:1 B:2
    for (int i = 0; i < h->n_refs; ++i) h->index_references[i] = static_cast<int32_t>(br.get(12));
    for (int i = 0; i < 2 * h->n_refs; ++i) h->global_flow[i] = br.get_sign_magnitude(14);
    if (h->n_refs > 0) h->warp_filter_size = static_cast<int32_t>(br.get(4));
    if (br.bad() || static_cast<size_t>(h->n_bytes_header) > n) return CCD_ERR_TRUNCATED;
    if (h->n_bytes_header <= 0) return CCD_ERR_VALUE;
    return h->n_bytes_header;
}

namespace {
// nnquant/quantstep.py:26-43: table length and exponent of the first entry, per header slot.
struct QStepRange { int n; int log2_first; };
constexpr QStepRange kQStep[8] = {{9, -8}, {17, -16}, {9, -8}, {17, -16}, {13, -12}, {1, 0}, {13, -12}, {25, -24}};

int derive_geometry(ccd_cc_header* h) {
    const int lo = h->latent_resolution[0], hi = h->latent_resolution[1];
    if (lo > hi) return CCD_ERR_VALUE;
    int first = lo, last = hi;
    if (h->flag_hyperlatent) {
        first = std::min({lo, hi, h->hyperlatent_resolution[0], h->hyperlatent_resolution[1]});
        last = std::max({lo, hi, h->hyperlatent_resolution[0], h->hyperlatent_resolution[1]});
    }
    h->n_grids = 0;
    h->n_symbols = 0;
    auto push = [&](int level, int hyper) -> bool {
        if (h->n_grids >= CCD_MAX_GRIDS) return false;
        const int g = h->n_grids++;
        // int(math.ceil(x / 2**i)) on doubles, component/core/coolchic.py:171
        h->grid_h[g] = static_cast<int32_t>(std::ceil(static_cast<double>(h->img_size[0]) / static_cast<double>(1 << level)));
        h->grid_w[g] = static_cast<int32_t>(std::ceil(static_cast<double>(h->img_size[1]) / static_cast<double>(1 << level)));
        h->is_hyperlatent[g] = hyper;
        h->n_symbols += static_cast<int64_t>(h->grid_h[g]) * h->grid_w[g];
        return true;
    };
    for (int level = first; level <= last; ++level) {
        if (lo <= level && level <= hi && !push(level, 0)) return CCD_ERR_VALUE;
        if (h->flag_hyperlatent && h->hyperlatent_resolution[0] <= level && level <= h->hyperlatent_resolution[1] &&
            !push(level, 1))
            return CCD_ERR_VALUE;
    }
    if (h->img_size[0] <= 0 || h->img_size[1] <= 0) return CCD_ERR_VALUE;
    h->total_context_arm = h->spatial_context_arm + h->output_feature_ifce;
    h->input_feature_synthesis = (hi - lo + 1) * (h->flag_common_randomness ? 2 : 1);
    for (int g = 0; g < h->n_grids; ++g) {
        h->input_features_ifce[g] = 0;
        if (!h->has_ifce_resolution) continue;
        const int ratio = static_cast<int>(std::ceil(std::log2(static_cast<double>(h->img_size[0]) / h->grid_h[g])));
        if (h->ifce_resolution[0] <= ratio && ratio <= h->ifce_resolution[1])
            h->input_features_ifce[g] = std::max(h->n_grids - 1 - g, 1);
    }
    h->out_channels = h->n_layer_synthesis > 0 ? h->syn_layer[h->n_layer_synthesis - 1].out_ft : 0;
    return CCD_OK;
}
}  // namespace

int read_cc_header(const uint8_t* p, size_t n, ccd_cc_header* h) {
    std::memset(h, 0, sizeof(*h));
    BitReader br(p, n);
    h->linear_stabiliser_synth = static_cast<int32_t>(br.get(1));
    h->n_layer_synthesis = static_cast<int32_t>(br.get(3));
    h->ups_k_size = static_cast<int32_t>(br.get(4));
    h->ups_preconcat_k_size = static_cast<int32_t>(br.get(4));
    h->output_feature_ifce = static_cast<int32_t>(br.get(5));
    h->spatial_context_arm = static_cast<int32_t>(br.get(6));
    h->linear_stabiliser_arm = static_cast<int32_t>(br.get(1));
    h->n_hidden_layers_arm = static_cast<int32_t>(br.get(3));
    for (int i = 0; i < 2; ++i) h->img_size[i] = static_cast<int32_t>(br.get(14));
    for (int i = 0; i < 2; ++i) h->latent_resolution[i] = static_cast<int32_t>(br.get(4));
    h->n_latent_grids = static_cast<int32_t>(br.get(5));
    h->flag_hyperlatent = static_cast<int32_t>(br.get(1));
    h->flag_common_randomness = static_cast<int32_t>(br.get(1));
    h->final_upsampling_type = static_cast<int32_t>(br.get(2));
    int q_index[8];
    for (int& q : q_index) q = static_cast<int>(br.get(5));
    for (int i = 0; i < 8; ++i) h->nn_expgol_cnt[i] = static_cast<int32_t>(br.get(4));
    h->nn_n_bytes = static_cast<int32_t>(br.get(14));
    h->nn_n_bit_pad = static_cast<int32_t>(br.get(3));
    h->n_bytes_latent = static_cast<int32_t>(br.get(28));
    h->n_bytes_header = static_cast<int32_t>(br.get(16));
    if (br.bad()) return CCD_ERR_TRUNCATED;
    if (h->final_upsampling_type > 2) return CCD_ERR_VALUE;
    for (int i = 0; i < 8; ++i) {
        if (q_index[i] >= kQStep[i].n || h->nn_expgol_cnt[i] > 12) return CCD_ERR_VALUE;
        h->nn_q_step_log2[i] = kQStep[i].log2_first + q_index[i];
    }
    if (h->output_feature_ifce > 0) {
        h->has_ifce_resolution = 1;
        for (int i = 0; i < 2; ++i) h->ifce_resolution[i] = static_cast<int32_t>(br.get(4));
    }
    if (h->flag_hyperlatent)
        for (int i = 0; i < 2; ++i) h->hyperlatent_resolution[i] = static_cast<int32_t>(br.get(4));
    for (int l = 0; l < h->n_layer_synthesis; ++l) {
        h->syn_layer[l].out_ft = static_cast<int32_t>(br.get(7));
        h->syn_layer[l].k_size = static_cast<int32_t>(br.get(4));
        h->syn_layer[l].mode = static_cast<int32_t>(br.get(1));
        h->syn_layer[l].non_linearity = static_cast<int32_t>(br.get(1));
    }
    if (br.bad() || static_cast<size_t>(h->n_bytes_header) > n) return CCD_ERR_TRUNCATED;
    if (h->n_bytes_header <= 0) return CCD_ERR_VALUE;
    const int rc = derive_geometry(h);
    if (rc < 0) return rc;
    return h->n_bytes_header;
}

// ---------------------------------------------------------------------------------------------
// Coding structure (utils/codingstructure.py:226-436): I frames at intra_pos, P frames at p_pos (each predicting
// from the closest already placed frame before it), every remaining display index filled with hierarchical B frames:
// scan for the first missing index, put a B frame in the MIDDLE of the gap it lies in (past + (future - past) / 2),
// start over.  Coding order = order of creation.
// ---------------------------------------------------------------------------------------------
int coding_structure(const ccd_video_header& h, std::vector<CodedFrame>& out) {
    out.clear();
    const int n = h.n_frames;
    if (n < 1 || h.n_intras < 1 || h.n_p_frames < 0) return CCD_ERR_VALUE;
    std::vector<int> intra(h.intra_pos, h.intra_pos + h.n_intras), pp(h.p_pos, h.p_pos + h.n_p_frames);
    std::sort(intra.begin(), intra.end());
    std::sort(pp.begin(), pp.end());
    // codingstructure.py:230-263: the asserts of __post_init__
    if (intra.front() != 0) return CCD_ERR_VALUE;
    if (intra.back() != n - 1 && (pp.empty() || pp.back() != n - 1)) return CCD_ERR_VALUE;
    std::vector<int> slot(n, -1);  // display order -> index in `out`
    auto place = [&](const CodedFrame& f) {
        if (f.display_order < 0 || f.display_order >= n || slot[f.display_order] >= 0) return false;  // outside / twice (I and P)
        slot[f.display_order] = static_cast<int>(out.size());
        out.push_back(f);
        return true;
    };
    auto past = [&](int d) { int r = -1; for (int i = d - 1; i >= 0 && r < 0; --i) if (slot[i] >= 0) r = i; return r; };
    auto future = [&](int d) { int r = -1; for (int i = d + 1; i < n && r < 0; ++i) if (slot[i] >= 0) r = i; return r; };
    for (int d : intra) {
        CodedFrame f; f.display_order = d; f.frame_type = 0;
        if (!place(f)) return CCD_ERR_VALUE;
    }
    for (int d : pp) {
        CodedFrame f; f.display_order = d; f.frame_type = 1; f.n_refs = 1;
        if (d < 0 || d >= n) return CCD_ERR_VALUE;
        const int r = past(d);
        if (r < 0) return CCD_ERR_VALUE;
        f.refs[0] = r; f.depth = out[slot[r]].depth + 1;
        if (!place(f)) return CCD_ERR_VALUE;
    }
    while (static_cast<int>(out.size()) < n) {
        int i = 0;
        while (i < n && slot[i] >= 0) ++i;
        const int a = past(i), b = future(i);  // both exist: frames 0 and n - 1 are placed
        if (a < 0 || b < 0) return CCD_ERR_VALUE;
        CodedFrame f; f.display_order = a + (b - a) / 2; f.frame_type = 2; f.n_refs = 2; f.refs[0] = a; f.refs[1] = b;
        f.depth = std::max(out[slot[a]].depth, out[slot[b]].depth) + 1;
        if (!place(f)) return CCD_ERR_VALUE;
    }
    return CCD_OK;
}

// ---------------------------------------------------------------------------------------------
// Exp-Golomb payload
// ---------------------------------------------------------------------------------------------
int decode_exp_golomb(const uint8_t* p, size_t n, int n_pad_bits, const std::vector<int>& count,
                      std::vector<int64_t>& out) {
    BitReader br(p, n);
    br.skip(static_cast<size_t>(n_pad_bits));
    out.resize(count.size());
    for (size_t i = 0; i < count.size(); ++i) {
        int prefix_zeros = 0;
        for (;;) {
            const int bit = br.peek_bit();
            if (bit < 0) return CCD_ERR_TRUNCATED;
            if (bit) break;
            ++prefix_zeros; br.skip(1);
        }
        if (prefix_zeros > 61) return CCD_ERR_VALUE;
        const int64_t quotient = static_cast<int64_t>(br.get(prefix_zeros + 1)) - 1;
        const int64_t remainder = count[i] ? static_cast<int64_t>(br.get(count[i])) : 0;
        if (br.bad()) return CCD_ERR_TRUNCATED;
        const int64_t folded = (quotient << count[i]) + remainder;  // sign in the LSB: odd = positive
        out[i] = (folded & 1) ? (folded + 1) / 2 : -(folded / 2);
    }
    return CCD_OK;
}

// ---------------------------------------------------------------------------------------------
// Network: stream layout -> typed parameters
// ---------------------------------------------------------------------------------------------
namespace {

using u128 = unsigned __int128;

inline int64_t shl_wrap(int64_t v, int s) { return static_cast<int64_t>(static_cast<uint64_t>(v) << s); }
inline u128 mag(int64_t v) { return v < 0 ? static_cast<u128>(-(v + 1)) + 1 : static_cast<u128>(v); }

// armint.py:30-170 for one MLP. `w_int[l]` is [out][in] (nn.Linear layout).
int to_fixed_point(const std::vector<const int64_t*>& w_int, const std::vector<const int64_t*>& b_int,
                   const std::vector<int>& n_in, const std::vector<int>& n_out, const int64_t* ws_int,
                   const int64_t* bs_int, int q_w_log2, int q_b_log2, bool subtract_last_layer, int n_inter_ft_ctx,
                   bool no_residual_layer, FixedArm& arm) {
    constexpr int kWeightShift = 16, kFracInter = 8;  // constants.py:17,39
    const int n_layers = static_cast<int>(w_int.size());
    arm.dim = n_in[0];
    arm.n_out = n_out[n_layers - 1];
    arm.layers.assign(n_layers, FixedLayer());
    const int shift_b = 2 * kWeightShift + q_b_log2;
    if (shift_b < 0) return CCD_ERR_VALUE;
    for (int l = 0; l < n_layers; ++l) {
        FixedLayer& L = arm.layers[l];
        L.n_in = n_in[l]; L.n_out = n_out[l];
        L.w.assign(static_cast<size_t>(L.n_in) * L.n_out, 0);
        L.b.assign(L.n_out, 0);
        const bool square = (L.n_in == L.n_out) && !no_residual_layer;
        for (int o = 0; o < L.n_out; ++o) {
            for (int i = 0; i < L.n_in; ++i) {
                int shift = kWeightShift + q_w_log2;
                int unit_shift = kWeightShift;  // residual connection folded into W, armint.py:114-124
                if (n_inter_ft_ctx > 0 && l == 0 && i >= L.n_in - n_inter_ft_ctx) { shift -= kFracInter; unit_shift -= kFracInter; }
                if (shift < 0) return CCD_ERR_VALUE;
                int64_t v = shl_wrap(w_int[l][static_cast<size_t>(o) * L.n_in + i], shift);
                if (square && o == i) v = static_cast<int64_t>(static_cast<uint64_t>(v) + (uint64_t{1} << unit_shift));
                L.w[static_cast<size_t>(i) * L.n_out + o] = v;
            }
            int64_t b = b_int[l][o];
            if (l == n_layers - 1 && subtract_last_layer && o == 1)  // armint.py:98-100
                b = static_cast<int64_t>(static_cast<uint64_t>(b) - (uint64_t{4} << (-q_b_log2)));
            L.b[o] = shl_wrap(b, shift_b);
        }
    }
    arm.ws.assign(static_cast<size_t>(arm.dim) * arm.n_out, 0);
    arm.bs.assign(arm.n_out, 0);
    if (ws_int) {
        for (int o = 0; o < arm.n_out; ++o) {
            for (int i = 0; i < arm.dim; ++i) {
                int shift = kWeightShift + q_w_log2;
                if (n_inter_ft_ctx > 0 && i >= arm.dim - n_inter_ft_ctx) shift -= kFracInter;
                if (shift < 0) return CCD_ERR_VALUE;
                arm.ws[static_cast<size_t>(i) * arm.n_out + o] = shl_wrap(ws_int[static_cast<size_t>(o) * arm.dim + i], shift);
            }
            arm.bs[o] = shl_wrap(bs_int[o], shift_b);
        }
    }
    return CCD_OK;
}

// Worst-case magnitudes through the integer MLP for inputs bounded by in_bound[i]; decides whether
// 32-bit operands / non-wrapping 64-bit accumulators are guaranteed. Returns the output bound.
bool analyse_bounds(const FixedArm& arm, const std::vector<u128>& in_bound, int output_shift, u128* out_bound) {
    const u128 lim32 = (u128{1} << 31) - 1, lim62 = u128{1} << 62;
    bool ok = true;
    std::vector<u128> x(in_bound.size());
    for (size_t i = 0; i < x.size(); ++i) { x[i] = in_bound[i] << 16; if (x[i] > lim32) ok = false; }
    std::vector<u128> stab(arm.n_out);
    for (int o = 0; o < arm.n_out; ++o) {
        u128 acc = mag(arm.bs[o]);
        for (int i = 0; i < arm.dim; ++i) {
            const u128 w = mag(arm.ws[static_cast<size_t>(i) * arm.n_out + o]);
            if (w > lim32) ok = false;
            acc += x[i] * w;
        }
        if (acc >= lim62) ok = false;
        stab[o] = acc;
    }
    for (size_t l = 0; l < arm.layers.size(); ++l) {
        const FixedLayer& L = arm.layers[l];
        const bool last = l + 1 == arm.layers.size();
        std::vector<u128> y(L.n_out);
        for (int o = 0; o < L.n_out; ++o) {
            u128 acc = mag(L.b[o]);
            for (int i = 0; i < L.n_in; ++i) {
                const u128 w = mag(L.w[static_cast<size_t>(i) * L.n_out + o]);
                if (w > lim32) ok = false;
                acc += x[i] * w;
            }
            if (last) acc += stab[o];
            if (acc >= lim62) ok = false;
            y[o] = last ? (acc >> output_shift) + 1 : (acc >> 16) + 1;
            if (!last && y[o] > lim32) ok = false;
        }
        x = y;
    }
    if (out_bound) { *out_bound = 0; for (u128 v : x) *out_bound = std::max(*out_bound, v); }
    return ok;
}

void symmetric_kernel(const int64_t* params, int n_params, int k, float q_step, float* out) {
    // upsampling.py:42-64: (a b c d) -> a b c d d c b a (even k) or a b c d c b a (odd k)
    for (int i = 0; i < n_params; ++i) out[i] = static_cast<float>(params[i]) * q_step;
    int j = n_params;
    for (int i = k % 2; i < n_params; ++i) out[j++] = out[n_params - 1 - i];
}

}  // namespace

int network_layout(const ccd_cc_header& h, size_t n_kind[8]) {
    const int dim = h.total_context_arm;
    const int n_arm_layers = h.n_hidden_layers_arm + 1;
    const int n_ifce_out = h.output_feature_ifce;
    if (dim < 1 || h.n_layer_synthesis < 1 || h.spatial_context_arm > 40) return CCD_ERR_VALUE;
    const int n_ups = h.latent_resolution[1];  // component/core/coolchic.py:1077-1086
    if (n_ups > 0 && (h.ups_k_size < 4 || (h.ups_k_size & 1) || !(h.ups_preconcat_k_size & 1))) return CCD_ERR_VALUE;
    const int ups_np = (h.ups_k_size + 1) / 2, pre_np = (h.ups_preconcat_k_size + 1) / 2;
    const int syn_out = h.out_channels, syn_in = h.input_feature_synthesis;
    const int n_stab_in = h.flag_common_randomness ? syn_in / 2 : syn_in;
    for (int k = 0; k < 8; ++k) n_kind[k] = 0;
    for (int l = 0; l < n_arm_layers; ++l) {
        const int out = (l == n_arm_layers - 1) ? 2 : dim;
        n_kind[0] += static_cast<size_t>(out) * dim; n_kind[1] += out;
    }
    if (h.linear_stabiliser_arm) { n_kind[0] += 2 * static_cast<size_t>(dim); n_kind[1] += 2; }
    for (int g = 0; g < h.n_grids; ++g)
        if (h.input_features_ifce[g] > 0) { n_kind[2] += static_cast<size_t>(n_ifce_out) * h.input_features_ifce[g]; n_kind[3] += n_ifce_out; }
    n_kind[4] = static_cast<size_t>(n_ups) * (ups_np + pre_np);
    n_kind[5] = 2 * static_cast<size_t>(n_ups);
    n_kind[6] = static_cast<size_t>(syn_out) * syn_out + (h.linear_stabiliser_synth ? static_cast<size_t>(syn_out) * n_stab_in : 0);
    n_kind[7] = syn_out + (h.linear_stabiliser_synth ? syn_out : 0);
    int c_in = syn_in;
    for (int l = 0; l < h.n_layer_synthesis; ++l) {
        const ccd_syn_layer& s = h.syn_layer[l];
        if (s.k_size < 1 || !(s.k_size & 1)) return CCD_ERR_UNSUPPORTED;
        if (s.mode == 1 && s.out_ft != c_in) return CCD_ERR_VALUE;
        n_kind[6] += static_cast<size_t>(s.out_ft) * c_in * s.k_size * s.k_size;
        n_kind[7] += s.out_ft;
        c_in = s.out_ft;
    }
    return CCD_OK;
}

int decode_network(const ccd_cc_header& h, const uint8_t* bytes_nn, size_t n_nn, Network& net) {
    const int dim = h.total_context_arm;
    const int n_arm_layers = h.n_hidden_layers_arm + 1;
    const int n_ifce_out = h.output_feature_ifce;
    const int n_ups = h.latent_resolution[1];
    const int ups_np = (h.ups_k_size + 1) / 2, pre_np = (h.ups_preconcat_k_size + 1) / 2;
    const int syn_out = h.out_channels, syn_in = h.input_feature_synthesis;
    const int n_stab_in = h.flag_common_randomness ? syn_in / 2 : syn_in;
    // ---- sizes per (module, weight|bias) in stream order --------------------------------------
    size_t n_kind[8];
    {
        const int lrc = network_layout(h, n_kind);
        if (lrc < 0) return lrc;
    }
    std::vector<int> count;
    for (int kind = 0; kind < 8; ++kind) count.insert(count.end(), n_kind[kind], h.nn_expgol_cnt[kind]);
    int rc = decode_exp_golomb(bytes_nn, n_nn, h.nn_n_bit_pad, count, net.ints);
    if (rc < 0) return rc;
    const int64_t* cur = net.ints.data();
    auto take = [&](size_t n) { const int64_t* p = cur; cur += n; return p; };

    // ---- ARM -----------------------------------------------------------------------------------
    {
        std::vector<const int64_t*> w(n_arm_layers), b(n_arm_layers);
        std::vector<int> n_in(n_arm_layers, dim), n_out(n_arm_layers, dim);
        n_out[n_arm_layers - 1] = 2;
        for (int l = 0; l < n_arm_layers; ++l) w[l] = take(static_cast<size_t>(n_out[l]) * dim);
        const int64_t* ws = h.linear_stabiliser_arm ? take(2 * static_cast<size_t>(dim)) : nullptr;
        for (int l = 0; l < n_arm_layers; ++l) b[l] = take(n_out[l]);
        const int64_t* bs = h.linear_stabiliser_arm ? take(2) : nullptr;
        rc = to_fixed_point(w, b, n_in, n_out, ws, bs, h.nn_q_step_log2[0], h.nn_q_step_log2[1], true, n_ifce_out, false, net.arm);
        if (rc < 0) return rc;
    }
    // ---- IFCE (coolchic.py:114-123: no -4, no inter-feature columns, no residual) ----------------
    net.ifce.assign(h.n_grids, FixedArm());
    net.ifce_feat_bound.assign(h.n_grids, 0);
    {
        std::vector<const int64_t*> w(h.n_grids, nullptr), b(h.n_grids, nullptr);
        for (int g = 0; g < h.n_grids; ++g)
            if (h.input_features_ifce[g] > 0) w[g] = take(static_cast<size_t>(n_ifce_out) * h.input_features_ifce[g]);
        for (int g = 0; g < h.n_grids; ++g)
            if (h.input_features_ifce[g] > 0) b[g] = take(n_ifce_out);
        u128 worst_feature = 0;
        for (int g = 0; g < h.n_grids; ++g) {
            if (!w[g]) continue;
            rc = to_fixed_point({w[g]}, {b[g]}, {h.input_features_ifce[g]}, {n_ifce_out}, nullptr, nullptr,
                                h.nn_q_step_log2[2], h.nn_q_step_log2[3], false, 0, true, net.ifce[g]);
            if (rc < 0) return rc;
            u128 fb = 0;
            net.ifce[g].narrow = analyse_bounds(net.ifce[g], std::vector<u128>(h.input_features_ifce[g], 64), 24, &fb);
            net.ifce_feat_bound[g] = fb > (u128{1} << 62) ? (int64_t{1} << 62) : static_cast<int64_t>(fb);
            worst_feature = std::max(worst_feature, fb);
        }
        std::vector<u128> in_bound(dim, 64);
        for (int i = h.spatial_context_arm; i < dim; ++i) in_bound[i] = worst_feature;
        // IFCE features enter the ARM shifted by 16 like the latents; they were produced with 8
        // fractional bits, and a feature above 2^24 would also lose bits in the reference's float
        // round trip (coolchic.py:142-144) - outside the narrow envelope either way.
        net.arm.narrow = analyse_bounds(net.arm, in_bound, 24, nullptr) && worst_feature < (u128{1} << 24);
        // ---- the pipelined kernel's envelope: static on the weights, dynamic on the data -------------
        const u128 lim32 = (u128{1} << 31) - 1;
        net.arm.w32 = true;
        for (const FixedLayer& L : net.arm.layers) for (int64_t w : L.w) if (mag(w) > lim32) net.arm.w32 = false;
        for (int64_t w : net.arm.ws) if (mag(w) > lim32) net.arm.w32 = false;
        net.ifce_w32 = true;
        for (int g = 0; g < h.n_grids; ++g)
            if (w[g]) for (int64_t v : net.ifce[g].layers[0].w) if (mag(v) > lim32) net.ifce_w32 = false;
        net.feat_i32 = worst_feature < (u128{1} << 30);
        net.arm.dyn_feat = worst_feature >= (u128{1} << 15);
        {   // worst hidden activation (after >> 16) with the worst-case features: can it leave int32?
            std::vector<u128> x(dim);
            for (int i = 0; i < dim; ++i) x[i] = std::min(in_bound[i], u128{1} << 40) << 16;
            net.arm.dyn_act = !net.arm.w32;  // (weights beyond int32 never reach the pipelined kernel; keeps the sums below inside u128)
            for (size_t l = 0; net.arm.w32 && l + 1 < net.arm.layers.size(); ++l) {
                const FixedLayer& L = net.arm.layers[l];
                std::vector<u128> y(L.n_out);
                for (int o = 0; o < L.n_out; ++o) {
                    u128 acc = mag(L.b[o]);
                    for (int i = 0; i < L.n_in; ++i) acc += std::min(x[i], u128{1} << 62) * mag(L.w[static_cast<size_t>(i) * L.n_out + o]);
                    y[o] = (acc >> 16) + 1;
                    if (y[o] > lim32) net.arm.dyn_act = true;
                }
                x = y;
            }
        }
    }
    // ---- Upsampling ------------------------------------------------------------------------------
    net.n_ups = n_ups; net.ups_k = h.ups_k_size; net.pre_k = h.ups_preconcat_k_size;
    net.ups_w.assign(static_cast<size_t>(n_ups) * net.ups_k, 0.f);
    net.pre_w.assign(static_cast<size_t>(n_ups) * net.pre_k, 0.f);
    {
        const float q = std::ldexp(1.0f, h.nn_q_step_log2[4]);
        for (int i = 0; i < n_ups; ++i) symmetric_kernel(take(ups_np), ups_np, net.ups_k, q, &net.ups_w[static_cast<size_t>(i) * net.ups_k]);
        for (int i = 0; i < n_ups; ++i) symmetric_kernel(take(pre_np), pre_np, net.pre_k, q, &net.pre_w[static_cast<size_t>(i) * net.pre_k]);
        take(2 * static_cast<size_t>(n_ups));  // biases: transmitted, never used by Upsampling.forward
    }
    // ---- Synthesis (named_parameters order: output_transform, stabiliser_branch, main_branch) -----
    {
        const float qw = std::ldexp(1.0f, h.nn_q_step_log2[6]), qb = std::ldexp(1.0f, h.nn_q_step_log2[7]);
        auto fill = [&](std::vector<float>& dst, size_t n, float q) {
            const int64_t* p = take(n);
            dst.resize(n);
            for (size_t i = 0; i < n; ++i) dst[i] = static_cast<float>(p[i]) * q;
        };
        net.syn_out = SynLayerParams();
        net.syn_out.c_in = net.syn_out.c_out = syn_out;
        fill(net.syn_out.w, static_cast<size_t>(syn_out) * syn_out, qw);
        net.syn_stab = SynLayerParams();
        if (h.linear_stabiliser_synth) {
            net.syn_stab.c_in = n_stab_in; net.syn_stab.c_out = syn_out;
            fill(net.syn_stab.w, static_cast<size_t>(syn_out) * n_stab_in, qw);
        }
        net.syn.assign(h.n_layer_synthesis, SynLayerParams());
        int c_in = syn_in;
        for (int l = 0; l < h.n_layer_synthesis; ++l) {
            SynLayerParams& S = net.syn[l];
            S.c_in = c_in; S.c_out = h.syn_layer[l].out_ft; S.k = h.syn_layer[l].k_size;
            S.residual = h.syn_layer[l].mode == 1; S.relu = h.syn_layer[l].non_linearity == 1;
            fill(S.w, static_cast<size_t>(S.c_out) * c_in * S.k * S.k, qw);
            c_in = S.c_out;
        }
        fill(net.syn_out.b, syn_out, qb);
        if (h.linear_stabiliser_synth) fill(net.syn_stab.b, syn_out, qb);
        for (auto& S : net.syn) fill(S.b, S.c_out, qb);
    }
    return CCD_OK;
}

bool float_path_stays_finite(const Network& net, int n_levels, int noise) {
    const double kLimit = std::ldexp(1.0, 120);  // FLT_MAX ~ 2^128: a partial sum of two bounded terms stays finite too
    auto abs_sum = [](const float* w, size_t n) { double s = 0; for (size_t i = 0; i < n; ++i) s += std::fabs(static_cast<double>(w[i])); return s; };
    // ---- latent pyramid: the stack's bound after each x2 step (coarsest first; filters of index step % n_ups) ----
    double bound = 64.0;
    for (int step = 0; step + 1 < n_levels; ++step) {
        if (net.n_ups < 1) return false;
        const int kidx = step % net.n_ups;
        const double su = abs_sum(&net.ups_w[static_cast<size_t>(kidx) * net.ups_k], net.ups_k);
        const double sp = abs_sum(&net.pre_w[static_cast<size_t>(kidx) * net.pre_k], net.pre_k);
        const double up = bound * su * su;            // every tap of the kron kernel at once: >= any output's taps
        const double pre = 64.0 * (sp * sp + 1.0);    // conv of the level's own latent + the latent (residual)
        bound = std::max(up, pre);
        if (!(bound < kLimit)) return false;
    }
    // ---- synthesis: one bound per layer (max over its outputs) ----
    const double in_bound = std::max(bound, noise ? 1024.0 : 0.0);  // Box-Muller of a 31-bit generator (< 6.6) through <= 8 bicubic x2 steps
    auto layer_bound = [&](const SynLayerParams& L, double x) {
        const size_t per_out = static_cast<size_t>(L.c_in) * L.k * L.k;
        double worst = 0;
        for (int o = 0; o < L.c_out; ++o) {
            const double b = L.b.empty() ? 0.0 : std::fabs(static_cast<double>(L.b[o]));
            worst = std::max(worst, b + abs_sum(&L.w[static_cast<size_t>(o) * per_out], per_out) * x);
        }
        return worst + (L.residual ? x : 0.0);
    };
    double x = in_bound;
    for (const SynLayerParams& L : net.syn) {
        x = layer_bound(L, x);
        if (!(x < kLimit)) return false;
    }
    if (net.syn_stab.c_out) {
        x += layer_bound(net.syn_stab, in_bound);
        if (!(x < kLimit)) return false;
    }
    x = layer_bound(net.syn_out, x);
    return x < kLimit;
}

void context_offsets(int n_spatial, int* dy, int* dx) {
    // arm.py:501-509: priority of each of the 40 causal positions of the 9x9 mask, row-major.
    static const int kPriority[40] = {38, 35, 30, 25, 23, 31, 36, 37, 39, 33, 28, 21, 20, 6,  15, 22, 29, 34, 32, 18,
                                      12, 10, 5,  9,  14, 19, 27, 24, 13, 8,  2,  1,  3,  11, 17, 26, 16, 7,  4,  0};
    for (int pos = 0; pos < 40; ++pos) {
        const int rank = kPriority[pos];
        if (rank >= n_spatial) continue;
        dy[rank] = 4 - pos / 9;   // rows above the current pixel
        dx[rank] = pos % 9 - 4;   // signed column offset
    }
}

void ifce_channel_shifts(const ccd_cc_header& h, int g, std::vector<int>& shifts) {
    // Channel c of the stack is grid g+1+c. Walking from grid g+1 towards coarser grids, every
    // change of size along the way was bridged by one nearest x2 (upsampling.py:582-588).
    shifts.clear();
    int s = 0;
    for (int m = g + 1; m < h.n_grids; ++m) {
        if (m > g + 1 && (h.grid_h[m] != h.grid_h[m - 1] || h.grid_w[m] != h.grid_w[m - 1])) ++s;
        shifts.push_back(s);
    }
}

}  // namespace ccd
