"""Crafted cool-chics that drive the FLOAT stages (latent pyramid + synthesis) through the operand classes a corrupt or hostile
network payload can produce - overflow to +-inf, inf - inf = NaN, subnormal values and products that underflow, signed zeros -
and through the extremes of the format's quantisation steps (nnquant/quantstep.py:26-43: upsampling 2^-12 .. 2^0, synthesis
weights 2^-12 .. 2^0, biases 2^-24 .. 2^0).  Built from the reference-encoded `rgb192` stream: ARM / IFCE parameters and the
latents stay as they are (the entropy stage decodes the same symbols), only the upsampling / synthesis integers and their
quantisation steps change.  Used by the CPU test of the finite envelope (ccd::float_path_stays_finite) and by the GPU parity
test of the float stages (tests/test_gpu_parity.py::test_float_stage_operand_classes)."""
import numpy as np


def nan_aware_equal(a: np.ndarray, b: np.ndarray):
    """(all equal, number of differing words): float32 arrays compared bit for bit, except that two NaNs are equal whatever
    their sign and payload (IEEE 754 fixes neither; x86 and gfx950 differ in the default NaN's sign)."""
    ua, ub = np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32)
    both_nan = np.isnan(a) & np.isnan(b)
    bad = (ua != ub) & ~both_nan
    return not bad.any(), int(bad.sum())


def crafted(load_golden, oracle):
    """{name: ((cc_header, bytes_nn, bytes_latent), stream, inside the finite envelope: True / False / None = either)};
    the stream is the one-frame 8-bit RGB .cool file around the cool-chic (oracle.decode_video gives its integer planes)."""
    from cool_chic_amd import writer

    bs, z, _ = load_golden("rgb192")
    hdr, _nn, _lat = oracle.split_stream(bs)[1][0][1][0]
    donor = writer.parse_cc_header(hdr)
    lay = writer.network_layout(donor)
    ints = np.asarray(z["cc0.nn_ints"], dtype=np.int64)
    latents = [z[f"cc0.latent{g}"] for g in range(donor.n_grids)]
    q0 = list(donor.nn_q_step_log2)  # arm.w arm.b ifce.w ifce.b ups.w ups.b syn.w syn.b (log2)

    def make(edit, q=None):
        g = [x.copy() for x in np.split(ints, np.cumsum(lay)[:-1])]
        edit(g)
        qs = list(q0)
        for k, v in (q or {}).items():
            qs[k] = v
        arch = writer.derive_arch(donor, nn_q_step_log2=tuple(qs))
        vals = np.clip(np.concatenate(g), -(2 ** 31) + 1, 2 ** 31 - 1).astype(np.int32)
        nn = writer.encode_network(arch, vals)
        stream = writer.encode_stream(writer.cc_header_bytes(arch), nn, latents)
        return oracle.split_stream(stream)[1][0][1][0], stream

    def shift(k, s):
        def f(g):
            g[k] = g[k] << s
        return f

    def signs(ks):
        def f(g):
            for k in ks:
                g[k] = np.sign(g[k])
        return f

    def zero_syn(g):
        g[6][:] = 0
        g[7][:] = 0

    def big_syn(g):
        g[6] = np.where(g[6] < 0, -(2 ** 31 - 1), 2 ** 31 - 1)

    def tiny_no_bias(g):
        signs((4, 6))(g)
        g[7][:] = 0

    out = {
        # every synthesis weight +-2^31: the second layer is beyond 2^80, the 3x3 layer overflows: inf, then inf - inf = NaN
        "syn_overflow": (*make(big_syn, {6: 0}), False),
        # upsampling filters up to 2^30 (kron products 2^60): the pyramid overflows at its second level
        "ups_overflow": (*make(shift(4, 22), {4: 0}), False),
        # every float weight +-2^-12 (biases +-2^-24): the coarse channels shrink by ~2^-22 per level into the subnormals,
        # the synthesis multiplies them by 2^-12 again: products that underflow
        "tiny": (*make(signs((4, 6, 7)), {4: -12, 6: -12, 7: -24}), True),
        "tiny_no_bias": (*make(tiny_no_bias, {4: -12, 6: -12, 7: -24}), True),   # ... and nothing larger added: subnormal outputs
        # all synthesis parameters zero: every product is a signed zero (negative activations x 0 = -0), accumulators +0
        "zeros": (*make(zero_syn), True),
        # the trained integers at the extremes of POSSIBLE_Q_STEP
        "q_extreme_small_ups": (*make(lambda g: None, {4: -12, 6: 0, 7: -24}), True),
        "q_extreme_large_ups": (*make(lambda g: None, {4: 0, 6: -12, 7: 0}), None),
    }
    # synthesis gains stepping towards FLT_MAX: somewhere in this sweep the largest value passes 2^128
    for s in range(12, 21):
        out[f"syn_gain_{s}"] = (*make(shift(6, s), {6: 0}), None)  # None: whichever side of the envelope it falls on
    return out
