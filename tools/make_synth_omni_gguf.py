#!/usr/bin/env python3
"""tools/make_synth_omni_gguf.py -- synthetic GGUFs of the omni encoder modules, in the layout the reference's loaders read.

  --module vpm : the image module (SigLIP-so400m tower + the 64-query resampler) as convert_vpm.py writes it and vision.cpp:787-1055 loads it.
  --module apm : the audio module (Whisper-medium encoder + audio projector) as tools/omni/convert/convert_apm.py writes it and
                 tools/omni/audition.cpp:790-1135 loads it: arch "whisper", KVs d_model / encoder_attention_heads / encoder_layers / n_mel /
                 n_fft / filters, tensors encoder.conv{1,2}.*, encoder.positional_embedding, encoder.blocks.N.{attn_ln, attn.{query,key,value,out},
                 mlp_ln, mlp.{0,2}}.*, encoder.ln_post.*, audio_projector.linear{1,2}.* -- matrices F16, vectors / positional table F32.

Own implementation of the on-disk format (ggml/src/gguf.cpp); random weights scaled so that 24 layers stay numerically tame."""
import argparse
import struct

import numpy as np

GGUF_MAGIC, GGUF_VERSION, ALIGN = 0x46554747, 3, 32
T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR = 4, 5, 6, 7, 8, 9
F32, F16 = 0, 1


def _s(b):
    b = b.encode() if isinstance(b, str) else b
    return struct.pack("<Q", len(b)) + b


def kv_u32(k, v):
    return _s(k) + struct.pack("<II", T_U32, v)


def kv_i32(k, v):
    return _s(k) + struct.pack("<Ii", T_I32, v)


def kv_f32(k, v):
    return _s(k) + struct.pack("<If", T_F32, v)


def kv_bool(k, v):
    return _s(k) + struct.pack("<IB", T_BOOL, 1 if v else 0)


def kv_str(k, v):
    return _s(k) + struct.pack("<I", T_STR) + _s(v)


def kv_arr_f32(k, a):
    a = np.asarray(a, np.float32)
    return _s(k) + struct.pack("<IIQ", T_ARR, T_F32, a.size) + a.tobytes()


def write_gguf(path, kvs, tensors):
    """tensors: list of (name, numpy array in torch order -- GGUF ne is the reversed shape)"""
    offs, off = [], 0
    for _, a in tensors:
        offs.append(off)
        off = (off + a.nbytes + ALIGN - 1) // ALIGN * ALIGN
    head = struct.pack("<IIQQ", GGUF_MAGIC, GGUF_VERSION, len(tensors), len(kvs)) + b"".join(kvs)
    for (name, a), o in zip(tensors, offs):
        ne = a.shape[::-1]
        head += _s(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", d) for d in ne) + struct.pack("<IQ", F16 if a.dtype == np.float16 else F32, o)
    head += b"\0" * ((-len(head)) % ALIGN)
    with open(path, "wb") as f:
        f.write(head)
        base = f.tell()
        for (_, a), o in zip(tensors, offs):
            f.write(b"\0" * (base + o - f.tell()))
            f.write(np.ascontiguousarray(a).tobytes())
        f.write(b"\0" * ((-f.tell()) % ALIGN))


def apm(path, n_layer, seed, d_model=1024, n_head=16, n_mel=80, n_ctx=1500, d_out=4096):
    rng = np.random.default_rng(seed)
    h = lambda *s: (rng.standard_normal(s) * (1.0 / np.sqrt(s[-1]))).astype(np.float16)          # noqa: E731  (fan-in scaled)
    v = lambda n, m=0.0, sd=0.02: (m + rng.standard_normal(n) * sd).astype(np.float32)           # noqa: E731
    t = [("encoder.positional_embedding", (rng.standard_normal((n_ctx, d_model)) * 0.1).astype(np.float32)),
         ("encoder.conv1.weight", (rng.standard_normal((d_model, n_mel, 3)) / np.sqrt(3 * n_mel)).astype(np.float16)), ("encoder.conv1.bias", v(d_model).reshape(d_model, 1)),
         ("encoder.conv2.weight", (rng.standard_normal((d_model, d_model, 3)) / np.sqrt(3 * d_model)).astype(np.float16)), ("encoder.conv2.bias", v(d_model).reshape(d_model, 1)),
         ("encoder.ln_post.weight", v(d_model, 1.0)), ("encoder.ln_post.bias", v(d_model))]
    for i in range(n_layer):
        p = f"encoder.blocks.{i}."
        t += [(p + "attn_ln.weight", v(d_model, 1.0)), (p + "attn_ln.bias", v(d_model)),
              (p + "attn.query.weight", h(d_model, d_model)), (p + "attn.query.bias", v(d_model)), (p + "attn.key.weight", h(d_model, d_model)),
              (p + "attn.value.weight", h(d_model, d_model)), (p + "attn.value.bias", v(d_model)),
              (p + "attn.out.weight", h(d_model, d_model)), (p + "attn.out.bias", v(d_model)),
              (p + "mlp_ln.weight", v(d_model, 1.0)), (p + "mlp_ln.bias", v(d_model)),
              (p + "mlp.0.weight", h(4 * d_model, d_model)), (p + "mlp.0.bias", v(4 * d_model)), (p + "mlp.2.weight", h(d_model, 4 * d_model)), (p + "mlp.2.bias", v(d_model))]
    t += [("audio_projector.linear1.weight", h(d_out, d_model)), ("audio_projector.linear1.bias", v(d_out)),
          ("audio_projector.linear2.weight", h(d_out, d_out)), ("audio_projector.linear2.bias", v(d_out))]
    n_fft_bins = 201
    kvs = [kv_str("general.architecture", "whisper"), kv_u32("general.file_type", 1), kv_str("general.description", "synthetic audio encoder (Whisper-medium shape) for MiniCPM-o"),
           kv_u32("encoder_attention_heads", n_head), kv_u32("encoder_ffn_dim", 4 * d_model), kv_u32("encoder_layers", n_layer), kv_u32("num_hidden_layers", n_layer),
           kv_u32("d_model", d_model), kv_u32("audio_pool_step", 5), kv_u32("use_f16", 1), kv_u32("n_mel", n_mel), kv_u32("n_fft", n_fft_bins),
           kv_arr_f32("filters", np.abs(rng.standard_normal(n_mel * n_fft_bins)) * 0.01)]
    write_gguf(path, kvs, t)
    print(f"wrote {path}: {len(t)} tensors")


def vpm(path, n_layer, seed, n_embd=1152, n_head=16, n_ff=4304, patch=14, image=448, d_out=4096, n_query=64):
    """SigLIP-so400m tower + MiniCPM-V resampler as tools/omni/convert/convert_vpm.py names them and tools/omni/vision.cpp:787-1055 loads them."""
    rng = np.random.default_rng(seed)
    h = lambda *s: (rng.standard_normal(s) * (1.0 / np.sqrt(s[-1]))).astype(np.float16)          # noqa: E731
    v = lambda n, m=0.0, sd=0.02: (m + rng.standard_normal(n) * sd).astype(np.float32)           # noqa: E731
    t = [("v.position_embd.weight", h(70 * 70, n_embd)),
         ("v.patch_embd.weight", (rng.standard_normal((n_embd, 3, patch, patch)) / np.sqrt(3 * patch * patch)).astype(np.float16)), ("v.patch_embd.bias", v(n_embd)),
         ("v.post_ln.weight", v(n_embd, 1.0)), ("v.post_ln.bias", v(n_embd))]
    for i in range(n_layer):
        p = f"v.blk.{i}."
        for w in ("attn_q", "attn_k", "attn_v", "attn_out"):
            t += [(p + w + ".weight", h(n_embd, n_embd)), (p + w + ".bias", v(n_embd))]
        t += [(p + "ln1.weight", v(n_embd, 1.0)), (p + "ln1.bias", v(n_embd)), (p + "ln2.weight", v(n_embd, 1.0)), (p + "ln2.bias", v(n_embd)),
              (p + "ffn_up.weight", h(n_ff, n_embd)), (p + "ffn_up.bias", v(n_ff)), (p + "ffn_down.weight", h(n_embd, n_ff)), (p + "ffn_down.bias", v(n_embd))]
    t += [("resampler.pos_embed_k", (rng.standard_normal((70 * 70, d_out)) * 0.1).astype(np.float32)), ("resampler.query", (rng.standard_normal((n_query, d_out)) * 0.5).astype(np.float32)),
          ("resampler.proj.weight", h(d_out, d_out)), ("resampler.kv.weight", h(d_out, n_embd))]
    for w in ("q", "k", "v", "out"):
        t += [(f"resampler.attn.{w}.weight", h(d_out, d_out)), (f"resampler.attn.{w}.bias", v(d_out))]
    for w in ("q", "kv", "post"):
        t += [(f"resampler.ln_{w}.weight", v(d_out, 1.0)), (f"resampler.ln_{w}.bias", v(d_out))]
    kvs = [kv_str("general.architecture", "clip"), kv_bool("clip.has_text_encoder", False), kv_bool("clip.has_vision_encoder", True), kv_bool("clip.has_minicpmv_projector", True),
           kv_u32("general.file_type", 1), kv_str("general.description", "synthetic image encoder (SigLIP-so400m shape + resampler) for MiniCPM-o"),
           kv_str("clip.projector_type", "resampler"), kv_i32("clip.minicpmv_version", 100045),
           kv_u32("clip.vision.image_size", image), kv_u32("clip.vision.patch_size", patch), kv_u32("clip.vision.embedding_length", n_embd),
           kv_u32("clip.vision.feed_forward_length", n_ff), kv_u32("clip.vision.projection_dim", 0), kv_u32("clip.vision.attention.head_count", n_head),
           kv_f32("clip.vision.attention.layer_norm_epsilon", 1e-6), kv_u32("clip.vision.block_count", n_layer), kv_u32("clip.minicpmv_query_num", n_query),
           kv_arr_f32("clip.vision.image_mean", [0.5, 0.5, 0.5]), kv_arr_f32("clip.vision.image_std", [0.5, 0.5, 0.5]), kv_bool("clip.use_gelu", True)]
    write_gguf(path, kvs, t)
    print(f"wrote {path}: {len(t)} tensors")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--module", choices=["apm", "vpm"], required=True)
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--layers", type=int, default=0, help="encoder blocks (default: 24 for apm, 27 for vpm)")
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args()
    if a.module == "apm":
        apm(a.out, a.layers or 24, a.seed)
    else:
        vpm(a.out, a.layers or 27, a.seed)
