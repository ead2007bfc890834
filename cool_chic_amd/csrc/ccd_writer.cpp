// ccd_writer.cpp - bitstream WRITER side ("next-2" of SURVEY section 8f): the range encoder, header
// serialisation and a host-side ancestral sampler that manufactures synthetic .cool streams with the
// statistics of a real one.  None of this is on the decode path: it only creates inputs for the
// benchmark configurations (only one real bitstream ships with the reference) and for round-trip tests.
//
// Reference behaviour (paths relative to /root/reference/coolchic):
//   bitstream/component/rangecoder.py:46-76   -> constriction 0.4.2 RangeEncoder
//   bitstream/header/header.py:90-105         header to_bytes (MSB-first, zero padded to a byte)
//   bitstream/encode.py:24-95                 framing: video, frame, cool-chic headers, NN, latents
//   bitstream/component/latent.py:142-173     the encoder walks the decoder's integer path
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ccd_device.hpp"
#include "ccd_format.hpp"

namespace ccd {
namespace {

const uint32_t kScaleBitsW[kNumScale] = {
#include "../../include/ccd_scale_table.inc"
};

inline float scale_value(int idx) { float f; std::memcpy(&f, &kScaleBitsW[idx], 4); return f; }

// Leaky quantised Laplace, left cumulative of symbol s.
uint32_t left_cumulative(int mu_idx, int scale_idx, int s) {
    if (s <= kAcLo) return 0;
    const double mu = -64.0 + mu_idx / 256.0;
    const double b = static_cast<double>(scale_value(scale_idx));
    const double x = s - 0.5;
    const double cdf = x <= mu ? 0.5 * std::exp((x - mu) / b) : 1.0 - 0.5 * std::exp((mu - x) / b);
    return static_cast<uint32_t>(16777088.0 * cdf) + static_cast<uint32_t>(s - kAcLo);
}
uint32_t right_cumulative(int mu_idx, int scale_idx, int s) {
    return s >= kAcLo + kAlphabet - 1 ? (1u << kRcPrecision) : left_cumulative(mu_idx, scale_idx, s + 1);
}

class RangeEncoder {
public:
    void put(int s, int mu_idx, int scale_idx) {
        const uint32_t l = left_cumulative(mu_idx, scale_idx, s), r = right_cumulative(mu_idx, scale_idx, s);
        touched_ = true;
        const uint64_t scale = range_ >> kRcPrecision;
        range_ = scale * static_cast<uint64_t>(r - l);
        const uint64_t moved = lower_ + scale * l;
        if (inverted_ && static_cast<uint64_t>(moved + range_) > moved) flush_inverted(moved < lower_, words_);
        lower_ = moved;
        if ((range_ >> 32) == 0) {
            const uint32_t word = static_cast<uint32_t>(lower_ >> 32);
            lower_ <<= 32; range_ <<= 32;
            if (inverted_) ++n_inverted_;
            else if (static_cast<uint64_t>(lower_ + range_) > lower_) words_.push_back(word);
            else { inverted_ = true; n_inverted_ = 1; first_inverted_ = word; }
        }
    }
    std::vector<uint32_t> sealed() const {
        std::vector<uint32_t> out = words_;
        if (!touched_) return out;
        const uint64_t point = lower_ + ((uint64_t{1} << 32) - 1);
        if (inverted_) flush_inverted(point < lower_, out);
        const uint32_t point_word = static_cast<uint32_t>(point >> 32);
        out.push_back(point_word);
        if (static_cast<uint32_t>(static_cast<uint64_t>(lower_ + range_) >> 32) == point_word) out.push_back(0u);
        return out;
    }
private:
    void flush_inverted(bool carry, std::vector<uint32_t>& out) const {
        out.push_back(carry ? first_inverted_ + 1u : first_inverted_);
        for (uint64_t i = 1; i < n_inverted_; ++i) out.push_back(carry ? 0u : 0xFFFFFFFFu);
    }
    void flush_inverted(bool carry, std::vector<uint32_t>& out) {
        static_cast<const RangeEncoder*>(this)->flush_inverted(carry, out);
        inverted_ = false;
    }
    uint64_t lower_ = 0, range_ = ~uint64_t{0};
    bool inverted_ = false, touched_ = false;
    uint64_t n_inverted_ = 0;
    uint32_t first_inverted_ = 0;
    std::vector<uint32_t> words_;
};

uint8_t* words_to_bytes(const std::vector<uint32_t>& w, int64_t* n_bytes) {
    *n_bytes = static_cast<int64_t>(w.size()) * 4;
    uint8_t* p = static_cast<uint8_t*>(std::malloc(w.size() * 4 + 4));
    if (!p) return nullptr;
    for (size_t i = 0; i < w.size(); ++i)
        for (int k = 0; k < 4; ++k) p[4 * i + k] = static_cast<uint8_t>(w[i] >> (8 * k));  // little-endian words
    return p;
}

// MSB-first bit sink (header.py:90-105).
class BitWriter {
public:
    void put(uint64_t v, int n) { for (int i = n - 1; i >= 0; --i) bits_.push_back((v >> i) & 1); }
    void put_sign_magnitude(int v, int n) { put(v < 0 ? 1 : 0, 1); put(static_cast<uint64_t>(v < 0 ? -v : v), n - 1); }
    size_t n_bits() const { return bits_.size(); }
    std::vector<uint8_t> bytes() const {
        std::vector<uint8_t> out((bits_.size() + 7) / 8, 0);
        for (size_t i = 0; i < bits_.size(); ++i) if (bits_[i]) out[i >> 3] |= static_cast<uint8_t>(0x80u >> (i & 7));
        return out;
    }
private:
    std::vector<uint8_t> bits_;
};

constexpr int kQLog2First[8] = {-8, -16, -8, -16, -12, 0, -12, -24};

std::vector<uint8_t> cc_header_bytes(const ccd_cc_header& h, int n_bytes_latent) {
    auto emit = [&](int n_bytes_header) {
        BitWriter w;
        w.put(h.linear_stabiliser_synth, 1); w.put(h.n_layer_synthesis, 3); w.put(h.ups_k_size, 4);
        w.put(h.ups_preconcat_k_size, 4); w.put(h.output_feature_ifce, 5); w.put(h.spatial_context_arm, 6);
        w.put(h.linear_stabiliser_arm, 1); w.put(h.n_hidden_layers_arm, 3);
        w.put(h.img_size[0], 14); w.put(h.img_size[1], 14);
        w.put(h.latent_resolution[0], 4); w.put(h.latent_resolution[1], 4);
        w.put(h.n_latent_grids, 5); w.put(h.flag_hyperlatent, 1); w.put(h.flag_common_randomness, 1);
        w.put(h.final_upsampling_type, 2);
        for (int i = 0; i < 8; ++i) w.put(h.nn_q_step_log2[i] - kQLog2First[i], 5);
        for (int i = 0; i < 8; ++i) w.put(h.nn_expgol_cnt[i], 4);
        w.put(h.nn_n_bytes, 14); w.put(h.nn_n_bit_pad, 3); w.put(n_bytes_latent, 28); w.put(n_bytes_header, 16);
        if (h.output_feature_ifce > 0) { w.put(h.ifce_resolution[0], 4); w.put(h.ifce_resolution[1], 4); }
        if (h.flag_hyperlatent) { w.put(h.hyperlatent_resolution[0], 4); w.put(h.hyperlatent_resolution[1], 4); }
        for (int l = 0; l < h.n_layer_synthesis; ++l) {
            w.put(h.syn_layer[l].out_ft, 7); w.put(h.syn_layer[l].k_size, 4);
            w.put(h.syn_layer[l].mode, 1); w.put(h.syn_layer[l].non_linearity, 1);
        }
        return w;
    };
    const size_t n_bytes = (emit(0).n_bits() + 7) / 8;
    return emit(static_cast<int>(n_bytes)).bytes();
}

std::vector<uint8_t> frame_header_bytes(const ccd_frame_header& f) {
    auto emit = [&](int n_bytes_header) {
        BitWriter w;
        w.put(f.display_index, 12); w.put(f.frame_type, 2); w.put(f.frame_data_type, 2); w.put(f.bitdepth - 8, 4);
        w.put(n_bytes_header, 16);
        const int n_refs = f.frame_type;  // I 0, P 1, B 2 (header.py:189-218)
        if (n_refs > 0) {
            for (int r = 0; r < n_refs; ++r) w.put(f.index_references[r], 12);
            for (int r = 0; r < 2 * n_refs; ++r) w.put_sign_magnitude(f.global_flow[r], 14);
            w.put(f.warp_filter_size, 4);
        }
        return w;
    };
    const size_t n_bytes = (emit(0).n_bits() + 7) / 8;
    return emit(static_cast<int>(n_bytes)).bytes();
}

std::vector<uint8_t> frame_header_bytes(int display_index, int bitdepth, int frame_data_type) {
    ccd_frame_header f{};
    f.display_index = display_index; f.frame_type = 0; f.frame_data_type = frame_data_type; f.bitdepth = bitdepth;
    return frame_header_bytes(f);
}

std::vector<uint8_t> video_header_bytes(const ccd_video_header& v) {
    auto emit = [&](int n_bytes_header) {
        BitWriter w;
        w.put(v.n_frames, 12); w.put(v.n_intras, 12); w.put(v.n_p_frames, 12); w.put(n_bytes_header, 16);
        for (int i = 0; i < v.n_intras; ++i) w.put(v.intra_pos[i], 12);
        for (int i = 0; i < v.n_p_frames; ++i) w.put(v.p_pos[i], 12);
        return w;
    };
    const size_t n_bytes = (emit(0).n_bits() + 7) / 8;
    return emit(static_cast<int>(n_bytes)).bytes();
}

std::vector<uint8_t> video_header_bytes_one_intra() {
    static ccd_video_header v;  // 32 KB: keep it off the stack
    v.n_frames = 1; v.n_intras = 1; v.n_p_frames = 0; v.intra_pos[0] = 0;
    return video_header_bytes(v);
}

// Serialise / parse round trip: fills the derived geometry from the transmitted fields of `tmpl`.
int rederive(const ccd_cc_header& tmpl, ccd_cc_header* h) {
    const std::vector<uint8_t> hb = cc_header_bytes(tmpl, 0);
    return read_cc_header(hb.data(), hb.size(), h);
}

// armint.py:180-203 on the host (writer side only).
void mlp_forward(const FixedArm& a, const int64_t* ctx, int output_shift, int64_t* out) {
    int64_t x[128], y[128], stab[64];
    for (int i = 0; i < a.dim; ++i) x[i] = static_cast<int64_t>(static_cast<uint64_t>(ctx[i]) << 16);
    for (int o = 0; o < a.n_out; ++o) {
        uint64_t acc = static_cast<uint64_t>(a.bs[o]);
        for (int i = 0; i < a.dim; ++i) acc += static_cast<uint64_t>(x[i]) * static_cast<uint64_t>(a.ws[static_cast<size_t>(i) * a.n_out + o]);
        stab[o] = static_cast<int64_t>(acc);
    }
    for (size_t l = 0; l + 1 < a.layers.size(); ++l) {
        const FixedLayer& L = a.layers[l];
        for (int o = 0; o < L.n_out; ++o) {
            uint64_t acc = static_cast<uint64_t>(L.b[o]);
            for (int i = 0; i < L.n_in; ++i) acc += static_cast<uint64_t>(x[i]) * static_cast<uint64_t>(L.w[static_cast<size_t>(i) * L.n_out + o]);
            const int64_t v = static_cast<int64_t>(acc);
            y[o] = (v < 0 ? 0 : v) >> 16;
        }
        for (int o = 0; o < L.n_out; ++o) x[o] = y[o];
    }
    const FixedLayer& L = a.layers.back();
    for (int o = 0; o < L.n_out; ++o) {
        uint64_t acc = static_cast<uint64_t>(L.b[o]);
        for (int i = 0; i < L.n_in; ++i) acc += static_cast<uint64_t>(x[i]) * static_cast<uint64_t>(L.w[static_cast<size_t>(i) * L.n_out + o]);
        acc += static_cast<uint64_t>(stab[o]);
        out[o] = static_cast<int64_t>(acc) >> output_shift;
    }
}

}  // namespace
}  // namespace ccd

using namespace ccd;

extern "C" {

int64_t ccd_range_encode(const int8_t* symbols, const int32_t* mu_idx, const int32_t* scale_idx, int64_t n, uint8_t** out) {
    if (!out || (n > 0 && (!symbols || !mu_idx || !scale_idx))) return CCD_ERR_ARG;
    RangeEncoder enc;
    for (int64_t i = 0; i < n; ++i) {
        const int m = std::min(std::max(mu_idx[i], 0), kNumMu - 1), c = std::min(std::max(scale_idx[i], 0), kNumScale - 1);
        if (symbols[i] < kAcLo || symbols[i] > kAcLo + kAlphabet - 1) return CCD_ERR_VALUE;
        enc.put(symbols[i], m, c);
    }
    int64_t n_bytes = 0;
    *out = words_to_bytes(enc.sealed(), &n_bytes);
    return *out ? n_bytes : CCD_ERR_NOMEM;
}

int ccd_network_layout(const ccd_cc_header* arch, int64_t n_values[8]) {
    if (!arch || !n_values) return CCD_ERR_ARG;
    ccd_cc_header h;
    int rc = rederive(*arch, &h);
    if (rc < 0) return rc;
    size_t n_kind[8];
    rc = network_layout(h, n_kind);
    if (rc < 0) return rc;
    for (int k = 0; k < 8; ++k) n_values[k] = static_cast<int64_t>(n_kind[k]);
    return CCD_OK;
}

int64_t ccd_encode_network(const ccd_cc_header* arch, const int32_t* values, int64_t n_values, int32_t* n_bit_pad,
                           uint8_t** out) {
    if (!arch || !values || !n_bit_pad || !out) return CCD_ERR_ARG;
    int64_t n_kind[8];
    const int rc = ccd_network_layout(arch, n_kind);
    if (rc < 0) return rc;
    int64_t total = 0;
    for (int k = 0; k < 8; ++k) { total += n_kind[k]; if (arch->nn_expgol_cnt[k] < 0 || arch->nn_expgol_cnt[k] > 15) return CCD_ERR_VALUE; }
    if (total != n_values) return CCD_ERR_ARG;
    // expgolomb.py:45-62: sign folded into the LSB, then x + 2^count in binary with (bit length - 1 - count) leading zeros
    BitWriter w;
    const int32_t* v = values;
    for (int k = 0; k < 8; ++k) {
        const int count = arch->nn_expgol_cnt[k];
        for (int64_t i = 0; i < n_kind[k]; ++i, ++v) {
            const uint64_t x = *v <= 0 ? static_cast<uint64_t>(-2 * static_cast<int64_t>(*v)) : static_cast<uint64_t>(2 * static_cast<int64_t>(*v) - 1);
            const uint64_t y = x + (uint64_t{1} << count);
            int len = 0;
            while ((y >> len) > 1) ++len;  // floor(log2 y)
            w.put(0, len - count);
            w.put(y, len + 1);
        }
    }
    const int pad = static_cast<int>((8 - w.n_bits() % 8) % 8);
    BitWriter full;
    full.put(0, pad);
    const std::vector<uint8_t> body = w.bytes();  // re-emit behind the prefix padding (expgolomb.py:64-67)
    for (size_t i = 0; i < w.n_bits(); ++i) full.put((body[i >> 3] >> (7 - (i & 7))) & 1, 1);
    const std::vector<uint8_t> bytes = full.bytes();
    uint8_t* p = static_cast<uint8_t*>(std::malloc(bytes.size() + 1));
    if (!p) return CCD_ERR_NOMEM;
    std::memcpy(p, bytes.data(), bytes.size());
    *n_bit_pad = pad;
    *out = p;
    return static_cast<int64_t>(bytes.size());
}

int ccd_write_cc_header(const ccd_cc_header* h, uint8_t* out, size_t cap) {
    if (!h || !out || h->n_layer_synthesis < 1 || h->n_layer_synthesis > CCD_MAX_SYN_LAYERS) return CCD_ERR_ARG;
    const std::vector<uint8_t> b = cc_header_bytes(*h, h->n_bytes_latent);
    if (b.size() > cap) return CCD_ERR_ARG;
    std::memcpy(out, b.data(), b.size());
    return static_cast<int>(b.size());
}

int ccd_write_frame_header(const ccd_frame_header* f, uint8_t* out, size_t cap) {
    if (!f || !out || f->frame_type < 0 || f->frame_type > 2 || f->bitdepth < 8 || f->bitdepth > 16) return CCD_ERR_ARG;
    const std::vector<uint8_t> b = frame_header_bytes(*f);
    if (b.size() > cap) return CCD_ERR_ARG;
    std::memcpy(out, b.data(), b.size());
    return static_cast<int>(b.size());
}

int ccd_write_video_header(const ccd_video_header* v, uint8_t* out, size_t cap) {
    if (!v || !out || v->n_intras < 0 || v->n_intras > 4096 || v->n_p_frames < 0 || v->n_p_frames > 4096) return CCD_ERR_ARG;
    const std::vector<uint8_t> b = video_header_bytes(*v);
    if (b.size() > cap) return CCD_ERR_ARG;
    std::memcpy(out, b.data(), b.size());
    return static_cast<int>(b.size());
}

int64_t ccd_encode_coolchic(const ccd_cc_header* tmpl, const uint8_t* bytes_nn, size_t n_nn, const int8_t* const* latents,
                            uint8_t** out) {
    if (!tmpl || !bytes_nn || !latents || !out) return CCD_ERR_ARG;
    // Re-derive the geometry from the transmitted fields by a serialise/parse round trip.
    ccd_cc_header h;
    {
        ccd_cc_header t = *tmpl;
        t.nn_n_bytes = static_cast<int32_t>(n_nn);
        const int rc = rederive(t, &h);
        if (rc < 0) return rc;
    }
    Network net;
    int rc = decode_network(h, bytes_nn, n_nn, net);
    if (rc < 0) return rc;
    const int n = h.n_grids, n_sp = h.spatial_context_arm, n_if = h.has_ifce_resolution ? h.output_feature_ifce : 0;
    int dy[kMaxCtx], dx[kMaxCtx];
    context_offsets(n_sp, dy, dx);
    for (int g = 0; g < n; ++g) if (!latents[g]) return CCD_ERR_ARG;
    std::vector<std::vector<int8_t>> grids(n);
    RangeEncoder enc;
    for (int g = n - 1; g >= 0; --g) {
        const int H = h.grid_h[g], W = h.grid_w[g];
        grids[g].assign(static_cast<size_t>(H) * W, 0);  // filled in coding order, like data_to_fill (latent.py:101)
        // IFCE features at the previous grid's size (coolchic.py:94-146)
        const int fin = h.input_features_ifce[g];
        const int fg = (g == n - 1) ? g : g + 1;
        const int fh = h.grid_h[fg], fw = h.grid_w[fg];
        std::vector<int32_t> feat;
        if (fin > 0) {
            std::vector<int> shifts;
            ifce_channel_shifts(h, g, shifts);
            feat.assign(static_cast<size_t>(n_if) * fh * fw, 0);
            int64_t in[64], o[64];
            for (int y = 0; y < fh; ++y)
                for (int x = 0; x < fw; ++x) {
                    if (g == n - 1) in[0] = 0;
                    else for (int c = 0; c < fin; ++c) {
                        const int m = g + 1 + c;
                        in[c] = grids[m][static_cast<size_t>(y >> shifts[c]) * h.grid_w[m] + (x >> shifts[c])];
                    }
                    mlp_forward(net.ifce[g], in, 24, o);
                    for (int k = 0; k < n_if; ++k)
                        feat[(static_cast<size_t>(k) * fh + y) * fw + x] = static_cast<int32_t>(static_cast<int64_t>(static_cast<float>(o[k])));
                }
        }
        const bool raster = W <= 9;
        const long n_steps = raster ? static_cast<long>(H) * W : W + 10L * (H - 1);
        int64_t ctx[128], ms[2];
        for (long c = 0; c < n_steps; ++c) {
            int y0, x0, cnt;
            if (raster) { y0 = static_cast<int>(c / W); x0 = static_cast<int>(c % W); cnt = 1; }
            else {
                if (c < W) { y0 = 0; x0 = static_cast<int>(c); }
                else { y0 = static_cast<int>((c - W) / 10) + 1; x0 = W - 10 + static_cast<int>((c - W) % 10); }
                cnt = std::min(H - y0, x0 / 10 + 1);
            }
            for (int i = 0; i < cnt; ++i) {
                const int y = y0 + i, x = x0 - 10 * i;
                for (int k = 0; k < n_sp; ++k) {
                    const int yy = y - dy[k], xx = x + dx[k];
                    ctx[k] = (yy >= 0 && xx >= 0 && xx < W) ? grids[g][static_cast<size_t>(yy) * W + xx] : 0;
                }
                for (int k = 0; k < n_if; ++k)
                    ctx[n_sp + k] = fin > 0 ? feat[(static_cast<size_t>(k) * fh + (y >> 1)) * fw + (x >> 1)] : 0;
                mlp_forward(net.arm, ctx, 24, ms);
                const int mu_idx = static_cast<int>(std::min<int64_t>(std::max<int64_t>(ms[0] + kMuOffset, 0), kNumMu - 1));
                const int sc_idx = static_cast<int>(std::min<int64_t>(std::max<int64_t>(ms[1] + kScaleOffset, 0), kNumScale - 1));
                // ancestral sample: uniform 24-bit quantile -> symbol through the coder's own CDF
                const int lo = latents[g][static_cast<size_t>(y) * W + x];
                grids[g][static_cast<size_t>(y) * W + x] = static_cast<int8_t>(lo);
                enc.put(lo, mu_idx, sc_idx);
            }
        }
    }
    const std::vector<uint32_t> words = enc.sealed();
    const int n_bytes_latent = static_cast<int>(words.size() * 4);
    const std::vector<uint8_t> ch = cc_header_bytes(h, n_bytes_latent);
    const size_t total = ch.size() + n_nn + static_cast<size_t>(n_bytes_latent);
    uint8_t* p = static_cast<uint8_t*>(std::malloc(total + 4));
    if (!p) return CCD_ERR_NOMEM;
    size_t pos = 0;
    std::memcpy(p + pos, ch.data(), ch.size()); pos += ch.size();
    std::memcpy(p + pos, bytes_nn, n_nn); pos += n_nn;
    for (size_t i = 0; i < words.size(); ++i)
        for (int k = 0; k < 4; ++k) p[pos + 4 * i + k] = static_cast<uint8_t>(words[i] >> (8 * k));
    *out = p;
    return static_cast<int64_t>(total);
}

int64_t ccd_encode_stream(const ccd_cc_header* tmpl, const uint8_t* bytes_nn, size_t n_nn,
                          const int8_t* const* latents, int bitdepth, int frame_data_type, uint8_t** out) {
    if (!out || bitdepth < 8 || bitdepth > 16) return CCD_ERR_ARG;
    uint8_t* cc = nullptr;
    const int64_t n_cc = ccd_encode_coolchic(tmpl, bytes_nn, n_nn, latents, &cc);
    if (n_cc < 0) return n_cc;
    const std::vector<uint8_t> vh = video_header_bytes_one_intra();
    const std::vector<uint8_t> fh = frame_header_bytes(0, bitdepth, frame_data_type);
    const size_t total = vh.size() + fh.size() + static_cast<size_t>(n_cc);
    uint8_t* p = static_cast<uint8_t*>(std::malloc(total + 4));
    if (!p) { std::free(cc); return CCD_ERR_NOMEM; }
    std::memcpy(p, vh.data(), vh.size());
    std::memcpy(p + vh.size(), fh.data(), fh.size());
    std::memcpy(p + vh.size() + fh.size(), cc, static_cast<size_t>(n_cc));
    std::free(cc);
    *out = p;
    return static_cast<int64_t>(total);
}

}  // extern "C"
