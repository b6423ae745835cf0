#!/usr/bin/env python3
"""tools/make_synth_omni_gguf.py -- synthetic GGUFs of the omni encoder modules, in the layout the reference's loaders read.

  --module vpm : the image module (SigLIP-so400m tower + the 64-query resampler) as convert_vpm.py writes it and vision.cpp:787-1055 loads it.
  --module t2w : the Token2Wav module set (four GGUFs + a prompt bundle) into the directory given with -o; see t2w() below.
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


def t2w(out_dir, seed, n_prompt_tokens=78):
    """The Token2Wav module set tools/omni/token2wav/token2wav-impl.cpp loads (every tensor F32, names and layouts as its binders request them):
      encoder.gguf        UpsampleConformerEncoderV2 (:2783-2860): embed / up_embed (linear + LayerNorm), pre_lookahead convs [K, Cin, Cout], 6 + 4 rel-pos conformer
                          blocks (512 wide, 8 heads, ffn 2048), up_layer conv (K 5), after_norm
      flow_matching.gguf  DiT estimator (:1840-1882; in 320, hidden 512, 16 blocks, 8 heads x 64, mlp 2048, out 80)
      flow_extra.gguf     input_embedding [6561 x 512], spk_embed_affine_layer 192 -> 80, encoder_proj 512 -> 80 (:6977)
      hifigan2.gguf       HiFT generator (:5503-5565): f0 predictor, source module, conv_pre, 3 transposed-conv stages [K, Cout, Cin] (x8, x5, x3) with 3 + 9 snake
                          resblocks, source_downs, conv_post (18 = n_fft + 2 channels for the 16-point iSTFT)
      prompt/             the prompt bundle of Token2Mel::load_prompt_bundle_dir (:8020-8070): spk_f32.bin [192], prompt_tokens_i32.bin [T], prompt_mel_btc_f32.bin [(T - 3) * 2, 80]"""
    import os
    os.makedirs(os.path.join(out_dir, "prompt"), exist_ok=True)
    rng = np.random.default_rng(seed)
    f = np.float32
    lin = lambda i, o: (rng.standard_normal((o, i)) / np.sqrt(i)).astype(f)                      # noqa: E731  torch [out, in] -> ne [in, out]
    conv = lambda k, ci, co: (rng.standard_normal((co, ci, k)) / np.sqrt(ci * k)).astype(f)       # noqa: E731  torch [Cout, Cin, K] -> ne [K, Cin, Cout]
    vec = lambda n, m=0.0, sd=0.02: (m + rng.standard_normal(n) * sd).astype(f)                   # noqa: E731
    arch = lambda name: [kv_str("general.architecture", name), kv_str("general.description", "synthetic Token2Wav module (random weights)")]     # noqa: E731
    # ---- encoder
    D, H, FF = 512, 8, 2048
    t = []

    def lin_ln(p):
        return [(p + ".out.0.weight", lin(D, D)), (p + ".out.0.bias", vec(D)), (p + ".out.1.weight", vec(D, 1.0)), (p + ".out.1.bias", vec(D))]

    def conformer(p):
        r = [(p + ".norm_ff.weight", vec(D, 1.0)), (p + ".norm_ff.bias", vec(D)), (p + ".norm_mha.weight", vec(D, 1.0)), (p + ".norm_mha.bias", vec(D))]
        for w in ("linear_q", "linear_k", "linear_v", "linear_out"):
            r += [(p + ".self_attn." + w + ".weight", lin(D, D)), (p + ".self_attn." + w + ".bias", vec(D))]
        r += [(p + ".self_attn.linear_pos.weight", lin(D, D)), (p + ".self_attn.pos_bias_u", vec((H, D // H), 0.0, 0.1)), (p + ".self_attn.pos_bias_v", vec((H, D // H), 0.0, 0.1)),
              (p + ".feed_forward.w_1.weight", lin(D, FF)), (p + ".feed_forward.w_1.bias", vec(FF)), (p + ".feed_forward.w_2.weight", lin(FF, D)), (p + ".feed_forward.w_2.bias", vec(D))]
        return r
    t += lin_ln("embed")
    t += [("pre_lookahead_layer.conv1.weight", conv(4, D, D)), ("pre_lookahead_layer.conv1.bias", vec(D)), ("pre_lookahead_layer.conv2.weight", conv(3, D, D)), ("pre_lookahead_layer.conv2.bias", vec(D))]
    for i in range(6):
        t += conformer(f"encoders.{i}")
    t += [("up_layer.conv.weight", conv(5, D, D)), ("up_layer.conv.bias", vec(D))]
    t += lin_ln("up_embed")
    for i in range(4):
        t += conformer(f"up_encoders.{i}")
    t += [("after_norm.weight", vec(D, 1.0)), ("after_norm.bias", vec(D))]
    write_gguf(os.path.join(out_dir, "encoder.gguf"), arch("t2w-encoder"), t)
    # ---- flow matching (DiT)
    E, HD, MLP = 512, 64, 2048
    t = [("estimator.t_embedder.mlp.0.weight", lin(256, E)), ("estimator.t_embedder.mlp.0.bias", vec(E)), ("estimator.t_embedder.mlp.2.weight", lin(E, E)), ("estimator.t_embedder.mlp.2.bias", vec(E)),
         ("estimator.in_proj.weight", lin(320, E)), ("estimator.in_proj.bias", vec(E))]
    for i in range(16):
        p = f"estimator.blocks.{i}."
        t += [(p + "adaLN_modulation.1.weight", (lin(E, 9 * E) * 0.3).astype(f)), (p + "adaLN_modulation.1.bias", vec(9 * E))]
        for w in ("to_q", "to_k", "to_v", "proj"):
            t += [(p + "attn." + w + ".weight", lin(E, E)), (p + "attn." + w + ".bias", vec(E))]
        t += [(p + "attn.q_norm.weight", vec(HD, 1.0)), (p + "attn.q_norm.bias", vec(HD)), (p + "attn.k_norm.weight", vec(HD, 1.0)), (p + "attn.k_norm.bias", vec(HD)),
              (p + "conv.block.1.weight", conv(3, E, E)), (p + "conv.block.1.bias", vec(E)), (p + "conv.block.3.weight", vec(E, 1.0)), (p + "conv.block.3.bias", vec(E)),
              (p + "conv.block.6.weight", conv(3, E, E)), (p + "conv.block.6.bias", vec(E)),
              (p + "mlp.fc1.weight", lin(E, MLP)), (p + "mlp.fc1.bias", vec(MLP)), (p + "mlp.fc2.weight", lin(MLP, E)), (p + "mlp.fc2.bias", vec(E))]
    t += [("estimator.final_layer.adaLN_modulation.1.weight", (lin(E, 2 * E) * 0.3).astype(f)), ("estimator.final_layer.adaLN_modulation.1.bias", vec(2 * E)),
          ("estimator.final_layer.linear.weight", lin(E, 80)), ("estimator.final_layer.linear.bias", vec(80))]
    write_gguf(os.path.join(out_dir, "flow_matching.gguf"), arch("t2w-flow-matching"), t)
    # ---- flow extra
    t = [("input_embedding.weight", (rng.standard_normal((6561, D)) * 0.5).astype(f)), ("spk_embed_affine_layer.weight", lin(192, 80)), ("spk_embed_affine_layer.bias", vec(80)),
         ("encoder_proj.weight", lin(D, 80)), ("encoder_proj.bias", vec(80))]
    write_gguf(os.path.join(out_dir, "flow_extra.gguf"), arch("t2w-flow-extra"), t)
    # ---- vocoder
    def resblock(p, ch, k):
        r = []
        for j in range(3):
            r += [(f"{p}.convs1.{j}.weight", conv(k, ch, ch)), (f"{p}.convs1.{j}.bias", vec(ch)), (f"{p}.convs2.{j}.weight", conv(k, ch, ch)), (f"{p}.convs2.{j}.bias", vec(ch)),
                  (f"{p}.activations1.{j}.alpha", vec(ch, 1.0, 0.1)), (f"{p}.activations2.{j}.alpha", vec(ch, 1.0, 0.1))]
        return r
    t = [("f0_predictor.condnet.0.weight", conv(3, 80, 512)), ("f0_predictor.condnet.0.bias", vec(512))]
    for j in (2, 4, 6, 8):
        t += [(f"f0_predictor.condnet.{j}.weight", conv(3, 512, 512)), (f"f0_predictor.condnet.{j}.bias", vec(512))]
    t += [("f0_predictor.classifier.weight", (lin(512, 1) * 40.0).astype(f)), ("f0_predictor.classifier.bias", vec(1, 120.0, 1.0)),      # f0 around 100..200 Hz: voiced frames
          ("m_source.l_linear.weight", lin(9, 1)), ("m_source.l_linear.bias", vec(1)),
          ("conv_pre.weight", conv(7, 80, 512)), ("conv_pre.bias", vec(512)), ("conv_post.weight", (conv(7, 64, 18) * 0.15).astype(f)),
          ("conv_post.bias", np.concatenate([vec(9, -1.0, 0.1), vec(9)]))]                          # 9 log-magnitude channels (kept below the +-0.99 clamp of the waveform), 9 phase channels
    for i, (k, co, ci) in enumerate(((16, 256, 512), (11, 128, 256), (7, 64, 128))):             # ConvTranspose1d: torch [Cin, Cout, K] -> ne [K, Cout, Cin]
        t += [(f"ups.{i}.weight", (rng.standard_normal((ci, co, k)) / np.sqrt(ci * k / (8, 5, 3)[i])).astype(f)), (f"ups.{i}.bias", vec(co))]
    for i, (k, co) in enumerate(((30, 256), (6, 128), (1, 64))):
        t += [(f"source_downs.{i}.weight", conv(k, 18, co)), (f"source_downs.{i}.bias", vec(co))]
    for i, (ch, k) in enumerate(((256, 7), (128, 7), (64, 11))):
        t += resblock(f"source_resblocks.{i}", ch, k)
    for stage, ch in enumerate((256, 128, 64)):
        for j, k in enumerate((3, 7, 11)):
            t += resblock(f"resblocks.{stage * 3 + j}", ch, k)
    write_gguf(os.path.join(out_dir, "hifigan2.gguf"), arch("t2w-hifigan2"), t)
    # ---- prompt bundle
    spk = rng.standard_normal(192).astype(f)
    (spk / np.linalg.norm(spk)).astype(f).tofile(os.path.join(out_dir, "prompt", "spk_f32.bin"))
    rng.integers(0, 6561, n_prompt_tokens).astype(np.int32).tofile(os.path.join(out_dir, "prompt", "prompt_tokens_i32.bin"))
    (rng.standard_normal(((n_prompt_tokens - 3) * 2, 80)) * 0.5 - 4.0).astype(f).tofile(os.path.join(out_dir, "prompt", "prompt_mel_btc_f32.bin"))
    print(f"wrote {out_dir}/{{encoder,flow_matching,flow_extra,hifigan2}}.gguf + prompt/")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--module", choices=["apm", "vpm", "t2w"], required=True)
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--layers", type=int, default=0, help="encoder blocks (default: 24 for apm, 27 for vpm)")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--prompt-tokens", type=int, default=78, help="t2w: tokens of the prompt bundle (its mel has (n - 3) * 2 frames)")
    a = ap.parse_args()
    if a.module == "t2w":
        t2w(a.out, a.seed, a.prompt_tokens)
    elif a.module == "apm":
        apm(a.out, a.layers or 24, a.seed)
    else:
        vpm(a.out, a.layers or 27, a.seed)
