#!/usr/bin/env python3
"""Golden-vector generator (build container only).

Imports the reference decoder from /root/reference (read-only) through the
shims in ./shims, decodes a .cool bitstream and dumps per-stage results as a
small fixture `<name>.npz` + `<name>.json` next to tests/golden/.

    python tests/golden/gen/dump_reference.py /root/reference/samples/bitstreams/kodim14.cool kodim14

Nothing here runs on the GPU box: the fixtures are data (inputs + expected
outputs); the reference source never leaves /root/reference.
"""
import hashlib
import json
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(1, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(int(os.environ.get("REF_THREADS", "1")))

import coolchic.bitstream.component.coolchic as bcc  # noqa: E402
import coolchic.bitstream.decode as bdec  # noqa: E402
from coolchic.bitstream.header.header import CoolChicHeader, FrameHeader, VideoHeader  # noqa: E402
from coolchic.component.core.synthesis import Synthesis  # noqa: E402
from coolchic.component.core.upsampling import Upsampling  # noqa: E402


def sha16(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def header_to_dict(h):
    out = {}
    for el in h._get_all_header_elements():
        v = el.get_value()
        if hasattr(v, "arm"):  # DescriptorCoolChic
            d = {}
            for mod in ["arm", "ifce", "upsampling", "synthesis"]:
                for wb in ["weight", "bias"]:
                    x = v.get_value(mod, wb)
                    d[f"{mod}.{wb}"] = float(x)
            v = d
        out[el.name] = v
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    out_dir = os.path.abspath(os.path.join(HERE, ".."))
    rec = {"headers": [], "cc": []}
    arrays = {}
    cur = {}

    # ---- hooks -------------------------------------------------------------
    orig_cc = bcc.encode_decode_coolchic
    orig_ent = bcc.entropy_coding_latent_arm
    orig_a2f = bcc.arm_to_fixed_point_param
    orig_fpa = bcc.fixed_point_arm
    orig_decnet = bcc.decode_network
    orig_ups_fwd = Upsampling.forward
    orig_syn_fwd = Synthesis.forward

    import coolchic.bitstream.neuralnet.neuralnet as nnmod

    orig_expgol = nnmod.decode_exp_golomb

    def hook_expgol(data_bytes, n_pad, count):
        vals = orig_expgol(data_bytes, n_pad, count)
        cur["nn_ints"] = np.asarray(vals, dtype=np.int64)
        cur["nn_count"] = np.asarray(count, dtype=np.int32)
        return vals

    nnmod.decode_exp_golomb = hook_expgol

    def hook_a2f(arm, q_steps, **kw):
        r = orig_a2f(arm, q_steps, **kw)
        tag = "ifce" if kw.get("no_residual_layer", False) else "arm"
        if tag == "arm":
            key = "arm"
        else:
            key = f"ifce{cur['n_ifce_calls']}"
            cur["n_ifce_calls"] += 1
        w, b, ws, bs = r
        for i, wi in enumerate(w):
            cur["fp"][f"{key}.w{i}"] = wi.numpy().astype(np.int64)
        for i, bi in enumerate(b):
            cur["fp"][f"{key}.b{i}"] = bi.numpy().astype(np.int64)
        cur["fp"][f"{key}.ws"] = ws.numpy().astype(np.int64)
        cur["fp"][f"{key}.bs"] = bs.numpy().astype(np.int64)
        return r

    def hook_fpa(x, *a, **kw):
        r = orig_fpa(x, *a, **kw)
        # only the IFCE call goes through bcc.fixed_point_arm (the ARM one is
        # imported inside latent.py)
        cur["ifce_out"].append(r.numpy().astype(np.int64))
        return r

    def hook_ent(enc, ctx, spatial_dim, *a, **kw):
        import coolchic.bitstream.component.latent as lat

        idxs = []
        rc = a[4]
        orig_decode = rc.decode

        def dec(idx_mu_scale):
            idxs.append(idx_mu_scale.numpy().astype(np.int64).copy())
            return orig_decode(idx_mu_scale)

        rc.decode = dec
        r = orig_ent(enc, ctx, spatial_dim, *a, **kw)
        rc.decode = orig_decode
        g = r.numpy().astype(np.int8).reshape(spatial_dim)
        cur["latents"].append(g)
        cur["mu_scale_idx"].append(np.concatenate(idxs, axis=0) if idxs else np.zeros((0, 2)))
        cur["ctx_ifce"].append(None if ctx is None else ctx.numpy().astype(np.int64)[0])
        return r

    def hook_ups(self, latents):
        r = orig_ups_fwd(self, latents)
        cur["dense"] = r.detach().numpy().astype(np.float32)[0]
        return r

    def hook_syn(self, x):
        r = orig_syn_fwd(self, x)
        cur["dense"] = x.detach().numpy().astype(np.float32)[0]  # synthesis input (latent planes [+ noise planes])
        cur["syn_out"] = r.detach().numpy().astype(np.float32)[0]
        return r

    def hook_cc(header, bytes_nn, mode, **kw):
        cur.clear()
        cur.update(
            dict(fp={}, latents=[], mu_scale_idx=[], ctx_ifce=[], ifce_out=[], n_ifce_calls=0)
        )
        out, b = orig_cc(header, bytes_nn, mode, **kw)
        idx = len(rec["cc"])
        p = header.get_coolchic_parameter()
        info = {
            "header": header_to_dict(header),
            "n_bytes_nn": len(bytes_nn),
            "n_bytes_latent": len(kw["dec_bytes_latent"]),
            "size_per_latent": [list(s[-2:]) for s in p.size_per_latent],
            "flag_is_hyperlatent": list(p.flag_is_hyperlatent),
            "input_features_ifce": list(p.input_features_ifce),
            "nn_ints_sha16": sha16(cur["nn_ints"].astype(np.int32)),
            "n_nn_ints": int(cur["nn_ints"].size),
            "latent_sha16_decode_order": [sha16(g) for g in cur["latents"]],
            "mu_scale_idx_sha16": [sha16(m.astype(np.int32)) for m in cur["mu_scale_idx"]],
            "dense_sha16": sha16(cur["dense"]),
            "syn_out_sha16": sha16(cur["syn_out"]),
            "out_shape": list(out.shape),
            "out_sha16": sha16(out.numpy().astype(np.float32)),
        }
        pre = f"cc{idx}."
        arrays[pre + "bytes_nn"] = np.frombuffer(bytes_nn, dtype=np.uint8)
        arrays[pre + "nn_ints"] = cur["nn_ints"].astype(np.int32)
        for k, v in cur["fp"].items():
            arrays[pre + "fp." + k] = v
        n = len(cur["latents"])
        for j, g in enumerate(cur["latents"]):
            gi = n - 1 - j  # decode order is coarsest (index n-1) first
            arrays[pre + f"latent{gi}"] = g
            ms = cur["mu_scale_idx"][j]
            arrays[pre + f"mu_scale_idx{gi}.head"] = ms[:512].astype(np.int32)
            c = cur["ctx_ifce"][j]
            if c is not None:
                info.setdefault("ctx_ifce_sha16", {})[str(gi)] = sha16(c.astype(np.int32))
                if c.size <= 6 * 64 * 96:
                    arrays[pre + f"ctx_ifce{gi}"] = c.astype(np.int32)
                else:
                    arrays[pre + f"ctx_ifce{gi}.crop"] = c[:, :32, :48].astype(np.int32)
        d = cur["dense"]
        info["dense_shape"] = list(d.shape)
        info["dense_chan_sum_f64"] = [float(x) for x in d.astype(np.float64).sum(axis=(1, 2))]
        hh, ww = d.shape[-2:]
        ch, cw = min(hh, 40), min(ww, 48)
        arrays[pre + "dense.tl"] = d[:, :ch, :cw]
        arrays[pre + "dense.br"] = d[:, hh - ch :, ww - cw :]
        arrays[pre + "dense.mid"] = d[:, hh // 2 : hh // 2 + ch, ww // 2 : ww // 2 + cw]
        s = cur["syn_out"]
        arrays[pre + "syn.tl"] = s[:, :ch, :cw]
        arrays[pre + "syn.br"] = s[:, hh - ch :, ww - cw :]
        arrays[pre + "syn.mid"] = s[:, hh // 2 : hh // 2 + ch, ww // 2 : ww // 2 + cw]
        o = out.numpy().astype(np.float32)[0]
        oh, ow = o.shape[-2:]
        arrays[pre + "out.tl"] = o[:, : min(oh, 40), : min(ow, 48)]
        arrays[pre + "out.br"] = o[:, oh - min(oh, 40) :, ow - min(ow, 48) :]
        rec["cc"].append(info)
        return out, b

    bcc.arm_to_fixed_point_param = hook_a2f
    bcc.fixed_point_arm = hook_fpa
    bcc.entropy_coding_latent_arm = hook_ent
    Upsampling.forward = hook_ups
    Synthesis.forward = hook_syn
    bdec.encode_decode_coolchic = hook_cc

    orig_fh = FrameHeader.read_header

    def hook_fh(self, raw):
        r = orig_fh(self, raw)
        rec["headers"].append({"frame": header_to_dict(self)})
        return r

    FrameHeader.read_header = hook_fh
    orig_vh = VideoHeader.read_header

    def hook_vh(self, raw):
        r = orig_vh(self, raw)
        rec["video_header"] = header_to_dict(self)
        return r

    VideoHeader.read_header = hook_vh

    frames = bdec.decode_video(path, decoded_path=None, verbosity=0)

    rec["bitstream_sha256"] = hashlib.sha256(open(path, "rb").read()).hexdigest()
    rec["bitstream_bytes"] = os.path.getsize(path)
    rec["torch_threads"] = torch.get_num_threads()
    rec["frames"] = {}
    for k, fd in frames.items():
        maxv = 2**fd.bitdepth - 1
        entry = {"bitdepth": fd.bitdepth, "frame_data_type": fd.frame_data_type}
        if fd.frame_data_type == "yuv420":
            for pk, pv in fd.data.items():
                q = np.round(pv.numpy()[0, 0] * maxv).astype(np.uint16)
                arrays[f"frame{k}.{pk}"] = q
                entry[f"{pk}_sha16"] = sha16(q)
        else:
            q = np.round(fd.data.numpy()[0] * maxv).astype(np.uint16)
            if maxv == 255:
                q = q.astype(np.uint8)
            arrays[f"frame{k}.data"] = q
            entry["data_sha16"] = sha16(q)
            entry["sha256"] = hashlib.sha256(q.tobytes()).hexdigest()
            entry["chan_mean"] = [float(x) for x in q.astype(np.float64).mean(axis=(1, 2))]
        rec["frames"][k] = entry

    if len(sys.argv) > 3 and sys.argv[3] == "--block-sums":
        # full-plane coverage of the float stages at fixture size: float64 sums of every 8 x 8 block of the reference's
        # dense planes and synthesis output (the per-stage crops of the main fixture cover three windows only)
        def blocks(a):
            c, h, w = a.shape
            hp, wp = -(-h // 8) * 8, -(-w // 8) * 8
            b = np.zeros((c, hp, wp), np.float64)
            b[:, :h, :w] = a
            return b.reshape(c, hp // 8, 8, wp // 8, 8).sum(axis=(2, 4))

        np.savez_compressed(os.path.join(out_dir, name + "_blocks.npz"), dense=blocks(cur["dense"]), syn_out=blocks(cur["syn_out"]))
        print("wrote", name + "_blocks.npz")
        return
    if len(sys.argv) > 3 and sys.argv[3] == "--planes-only":
        # variants that differ from an existing fixture only after the cool-chics (e.g. the warp filter): keep the decoded planes
        arrays = {k: v for k, v in arrays.items() if k.startswith("frame")}
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **arrays)
    with open(os.path.join(out_dir, name + ".json"), "w") as f:
        json.dump(rec, f, indent=1, default=lambda o: o if not hasattr(o, "tolist") else o.tolist())
    print("wrote", name, {k: v for k, v in rec["frames"].items()})


if __name__ == "__main__":
    main()
