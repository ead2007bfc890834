// ccd_format.hpp - host-side parsing of the .cool format: bit-packed headers, derived grid
// geometry, Exp-Golomb network parameters and their fixed-point / float forms.
//
// Reference behaviour (paths relative to /root/reference/coolchic):
//   bitstream/header/header.py:72-88,130-147,172-218,244-325   headers
//   component/core/coolchic.py:149-225                         grid geometry
//   bitstream/neuralnet/{neuralnet.py:92-204,expgolomb.py:74-130}   network payload
//   bitstream/component/armint.py:30-170                       fixed-point ARM / IFCE parameters
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/ccd.h"

namespace ccd {

// MSB-first bit cursor over a byte span; reading past the end latches `bad`.
class BitReader {
public:
    BitReader(const uint8_t* p, size_t n_bytes) : p_(p), n_bits_(n_bytes * 8) {}
    uint64_t get(int n);
    int get_sign_magnitude(int n);  // element.py:62-85 (signed=True)
    int peek_bit() const;           // -1 at end of data
    void skip(size_t n) { pos_ += n; if (pos_ > n_bits_) bad_ = true; }
    bool bad() const { return bad_; }
    size_t pos() const { return pos_; }
private:
    const uint8_t* p_;
    size_t n_bits_, pos_ = 0;
    bool bad_ = false;
};

int read_video_header(const uint8_t* p, size_t n, ccd_video_header* h);

// One frame of the coding structure a video header implies (utils/codingstructure.py:267-436): frames come in CODING order.
struct CodedFrame {
    int display_order = 0, frame_type = 0 /* 0 I, 1 P, 2 B */, n_refs = 0, refs[2] = {0, 0} /* display orders */, depth = 0;
};
// CodingStructure(n_frames, intra_pos, p_pos).frames sorted by coding order; CCD_ERR_VALUE where the reference asserts
// (first frame not intra, last frame neither intra nor P, a frame both I and P) or cannot build the structure.
int coding_structure(const ccd_video_header& h, std::vector<CodedFrame>& out);
int read_frame_header(const uint8_t* p, size_t n, ccd_frame_header* h);
int read_cc_header(const uint8_t* p, size_t n, ccd_cc_header* h);  // also fills the derived geometry

// One linear layer in fixed point, weights stored [in][out] (already transposed like armint.py:127).
struct FixedLayer {
    int n_in = 0, n_out = 0;
    std::vector<int64_t> w, b;
};

struct FixedArm {
    int dim = 0;
    std::vector<FixedLayer> layers;  // hidden layers then the output layer
    std::vector<int64_t> ws, bs;     // stabiliser [dim][n_out], [n_out] (zeros when absent)
    int n_out = 2;
    // true when every operand fits int32 and no accumulator can leave int64 without wrapping, for
    // inputs bounded by |latent| <= 64 and the WORST-CASE IFCE features (every context at +-64 with the
    // sign that hurts): the static envelope of the matrix-core variant of the pipelined kernel.
    bool narrow = false;
    // The pipelined kernel's own, much wider envelope.  It multiplies int32 x int32 -> int64 and sums
    // in wrapping int64, which is the reference's arithmetic (armint.py:180-203, torch int64 wraps)
    // whenever the OPERANDS are exact in 32 bits.  w32: every ARM / stabiliser weight fits int32.
    // dyn_act: a hidden activation could leave int32 for worst-case inputs (no network seen so far;
    // such a slot runs the generic kernel).  dyn_feat: the worst case of an IFCE feature does not
    // fit 16 bits - informational, the device checks every feature it stores anyway and redoes the
    // pixels that meet a wide one in plain int64.
    bool w32 = false, dyn_feat = false, dyn_act = false;
};

struct SynLayerParams {
    int c_in = 0, c_out = 0, k = 1;
    bool residual = false, relu = false;
    std::vector<float> w, b;  // w: [c_out][c_in][k][k]
};

struct Network {
    std::vector<int64_t> ints;           // every transmitted integer, stream order
    FixedArm arm;
    std::vector<FixedArm> ifce;          // one per grid (dim == 0 when the grid has no IFCE)
    std::vector<int64_t> ifce_feat_bound;  // per grid: worst-case |feature| (Q8)
    bool ifce_w32 = false;               // every IFCE weight fits int32 (the register-resident feature pass)
    bool feat_i32 = false;               // the worst-case feature fits the int32 side plane of the pipelined kernel (< 2^30)
    int n_ups = 0, ups_k = 0, pre_k = 0;
    std::vector<float> ups_w;            // [n_ups][ups_k] symmetric 1-D kernels (upsampling.py:42-64)
    std::vector<float> pre_w;            // [n_ups][pre_k]
    std::vector<SynLayerParams> syn;     // main branch
    SynLayerParams syn_stab;             // c_out == 0 when absent; 1x1 on the first c_in channels
    SynLayerParams syn_out;              // output transform 1x1
};

// Static envelope of the float stages (upsampling.py:463-500, synthesis.py:272-294): true when NO intermediate value of the
// latent pyramid or of the synthesis can leave the finite float32 range, whatever the latents (|x| <= 64) - worst-case
// magnitudes propagated through every filter and layer (sum of |weights| x bound + |bias|), with a margin of 2^7.  Networks
// inside it (every trained one: bounds of 2^10 .. 2^20) may run the matrix-core kernel, whose zero-weight padding
// (fma(v, 0, acc) == acc) and NaN-dropping ReLU are exact only for finite v; the others run the vector-ALU kernels, which
// evaluate exactly the oracle's taps and propagate NaN like torch.relu.  `n_levels` latent levels reach the synthesis,
// `noise` common-randomness planes beside them.
bool float_path_stays_finite(const Network& net, int n_levels, int noise);

int decode_exp_golomb(const uint8_t* p, size_t n, int n_pad_bits, const std::vector<int>& count,
                      std::vector<int64_t>& out);
// Number of transmitted integers per (module, weight|bias) group, stream order (types.py:18-19,98-101).
int network_layout(const ccd_cc_header& h, size_t n_kind[8]);
int decode_network(const ccd_cc_header& h, const uint8_t* bytes_nn, size_t n_nn, Network& net);

// Context template: (dy, dx) of the n highest-priority causal neighbours, component/core/arm.py:493-562.
// The neighbour of (y, x) is (y - dy, x + dx) with dy in 0..4, dx in -4..4.
void context_offsets(int n_spatial, int* dy, int* dx);

// For the IFCE input stack seen while decoding grid g (channel c = grid g+1+c, all resampled to the
// size of grid g+1 by repeated nearest x2 + crop, upsampling.py:556-595): right-shift to apply to the
// coordinates for channel c.
void ifce_channel_shifts(const ccd_cc_header& h, int g, std::vector<int>& shifts);

}  // namespace ccd
