#!/usr/bin/env python3
"""tools/make_synth_gguf.py -- write a synthetic Qwen3-shaped GGUF v3 file (C2 = Q4_K_M map, C3 = all F16).

Own implementation of the on-disk format the reference reads (ggml/src/gguf.cpp: header, KV section, tensor infos, 32-byte
aligned data; gguf-py/gguf/gguf_writer.py is the reference's writer).  Weights are random *valid* blocks written directly in
their quantised layout (llama_cpp_omni_amd.qwen3.random_blocks), norms are 1.0, the tokenizer is "no_vocab"
(accepted by the reference loader, src/llama-vocab.cpp:1679-1699).  One layer's bytes are generated once and reused for every
layer (the file is for timing and cross-device parity, not for language).

    python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/qwen3-8b-q4km-synth.gguf
    python tools/make_synth_gguf.py --config tiny --types q4_k_m -o /tmp/tiny.gguf --distinct-layers
"""
import argparse
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg  # noqa: E402

GGUF_MAGIC, GGUF_VERSION, ALIGN = 0x46554747, 3, 32
T_U32, T_F32, T_STR = 4, 6, 8


def _s(b):
    b = b.encode() if isinstance(b, str) else b
    return struct.pack("<Q", len(b)) + b


def kv_u32(k, v):
    return _s(k) + struct.pack("<II", T_U32, v)


def kv_f32(k, v):
    return _s(k) + struct.pack("<If", T_F32, v)


def kv_str(k, v):
    return _s(k) + struct.pack("<I", T_STR) + _s(v)


# ---- block formats needed by the separated-logits fixture (ggml-common.h:295-305, :330-335; dequantize_row_q4_K / _q6_K, ggml-quants.c:1352-1374, :1762-1791)
def kv_i32(k, v):
    return _s(k) + struct.pack("<Ii", 5, v)


def kv_bool(k, v):
    return _s(k) + struct.pack("<IB", 7, 1 if v else 0)


def kv_arr_str(k, items):
    return _s(k) + struct.pack("<IIQ", 9, 8, len(items)) + b"".join(_s(x) for x in items)


def kv_arr_i32(k, a):
    a = np.asarray(a, np.int32)
    return _s(k) + struct.pack("<IIQ", 9, 5, a.size) + a.tobytes()


# The special tokens tools/omni/omni.cpp looks up by text (omni_init :3964-3982, the prompt strings of :3518-3537 and stream_prefill :8800-8880) or by
# hard-coded id (g_special_token_ids :4432-4441).  Ids follow Qwen3's tokenizer for the first block and omni.cpp's constants for the rest; tokens the
# reference names without an id sit in free slots of the same range.
OMNI_SPECIALS = {151643: "<|endoftext|>", 151644: "<|im_start|>", 151645: "<|im_end|>", 151667: "<think>", 151668: "</think>",
                 151669: "<image>", 151670: "</image>", 151671: "<slice>", 151672: "</slice>", 151673: "<unit>", 151674: "</unit>",
                 151675: "<|audio_start|>", 151676: "<|audio_end|>", 151677: "<|tts_pad|>", 151703: "<|tts_bos|>", 151704: "<|tts_eos|>",
                 151705: "<|listen|>", 151706: "<|speak|>", 151717: "<|turn_eos|>", 151718: "<|chunk_eos|>", 151721: "<|chunk_tts_eos|>"}


def gpt2_byte_chars():
    """the byte -> printable-character table of byte-level BPE (what src/unicode.cpp unicode_byte_to_utf8 encodes)"""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


def omni_vocab_kvs(V):
    """a byte-level BPE tokenizer (tokenizer.ggml.model gpt2, pre-tokenizer qwen2) with NO learned merges but one: text tokenises to its bytes (ids 0-255),
    ids 256.. are unique ASCII filler words a random-weight model may sample, the specials above are CONTROL tokens parsed from text"""
    b2c = gpt2_byte_chars()
    toks = [b2c[b] for b in range(256)] + ["ab"] + [f"w{i}" for i in range(257, V)]
    types = np.ones(V, np.int32)                               # LLAMA_TOKEN_TYPE_NORMAL
    for i, t in OMNI_SPECIALS.items():
        assert i < V
        toks[i] = t; types[i] = 3                               # LLAMA_TOKEN_TYPE_CONTROL
    return [kv_str("tokenizer.ggml.model", "gpt2"), kv_str("tokenizer.ggml.pre", "qwen2"), kv_arr_str("tokenizer.ggml.tokens", toks),
            kv_arr_i32("tokenizer.ggml.token_type", types), kv_arr_str("tokenizer.ggml.merges", ["a b"]),
            kv_u32("tokenizer.ggml.eos_token_id", 151645), kv_u32("tokenizer.ggml.padding_token_id", 151643), kv_u32("tokenizer.ggml.bos_token_id", 151643),
            kv_bool("tokenizer.ggml.add_bos_token", False)]


def dequant_q4_K(rows, K):
    n, nb = rows.shape[0], K // 256
    b = rows.reshape(n, nb, 144)
    d = b[..., 0:2].copy().view(np.float16).astype(np.float32)[..., 0]; dmin = b[..., 2:4].copy().view(np.float16).astype(np.float32)[..., 0]
    sc8 = b[..., 4:16].astype(np.int32); qs = b[..., 16:144]
    out = np.empty((n, nb, 256), np.float32)
    for j in range(8):
        if j < 4:
            sc, m = sc8[..., j] & 63, sc8[..., j + 4] & 63
        else:
            sc = (sc8[..., j + 4] & 0xF) | ((sc8[..., j - 4] >> 6) << 4); m = (sc8[..., j + 4] >> 4) | ((sc8[..., j] >> 6) << 4)
        q = qs[..., 32 * (j // 2): 32 * (j // 2) + 32]
        q = (q & 0xF) if j % 2 == 0 else (q >> 4)
        out[..., 32 * j: 32 * j + 32] = (d * sc)[..., None] * q - (dmin * m)[..., None]
    return out.reshape(n, K)


def quant_q6_K(x):
    """a plain (not the reference's search) Q6_K encoder: per 16 weights scale = max|x| / 31, super-block d = max|scale| / 127"""
    n, K = x.shape
    nb = K // 256
    xb = x.reshape(n, nb, 16, 16)
    s = np.abs(xb).max(axis=3) / 31.0
    d = np.maximum(s.max(axis=2) / 127.0, 1e-30).astype(np.float16).astype(np.float32)
    sc = np.clip(np.rint(s / d[..., None]), 1, 127).astype(np.int32)
    q = np.clip(np.rint(xb / (d[..., None, None] * sc[..., None])) + 32, 0, 63).astype(np.uint8).reshape(n, nb, 256)
    blk = np.zeros((n, nb, 210), np.uint8)
    for h in range(2):                                          # each half: 128 weights = 4 groups of 32 (l, l + 32, l + 64, l + 96)
        q1, q2, q3, q4 = (q[..., 128 * h + 32 * g: 128 * h + 32 * g + 32] for g in range(4))
        blk[..., 64 * h: 64 * h + 32] = (q1 & 0xF) | ((q3 & 0xF) << 4)
        blk[..., 64 * h + 32: 64 * h + 64] = (q2 & 0xF) | ((q4 & 0xF) << 4)
        blk[..., 128 + 32 * h: 128 + 32 * h + 32] = (q1 >> 4) | ((q2 >> 4) << 2) | ((q3 >> 4) << 4) | ((q4 >> 4) << 6)
    blk[..., 192:208] = sc.astype(np.int8).view(np.uint8)
    blk[..., 208:210] = d.astype(np.float16)[..., None].view(np.uint8).reshape(n, nb, 2)
    return blk.reshape(n, nb * 210)


def dequant_q6_K(rows, K):
    n, nb = rows.shape[0], K // 256
    b = rows.reshape(n, nb, 210)
    ql, qh, sc = b[..., 0:128], b[..., 128:192], b[..., 192:208].view(np.int8).astype(np.float32)
    d = b[..., 208:210].copy().view(np.float16).astype(np.float32)[..., 0]
    out = np.empty((n, nb, 256), np.float32)
    for h in range(2):
        l, hh = ql[..., 64 * h: 64 * h + 64].astype(np.int32), qh[..., 32 * h: 32 * h + 32].astype(np.int32)
        qs = [(l[..., :32] & 0xF) | ((hh & 3) << 4), (l[..., 32:] & 0xF) | (((hh >> 2) & 3) << 4), (l[..., :32] >> 4) | (((hh >> 4) & 3) << 4), (l[..., 32:] >> 4) | (((hh >> 6) & 3) << 4)]
        for g in range(4):
            scale = np.repeat(sc[..., 8 * h + 2 * g: 8 * h + 2 * g + 2], 16, axis=-1)
            out[..., 128 * h + 32 * g: 128 * h + 32 * g + 32] = d[..., None] * scale * (qs[g] - 32)
    return out.reshape(n, K)


def dequant_q8_0(rows, K):
    """block_q8_0 (ggml-common.h:219-224): x = d * q, 32 per block.  rows: uint8 [n, K / 32 * 34]"""
    n = rows.shape[0]; b = rows.reshape(n, K // 32, 34)
    d = b[..., 0:2].copy().view(np.float16).astype(np.float32)                           # [n, nb, 1]
    return (d * b[..., 2:].view(np.int8).astype(np.float32)).reshape(n, K)


def quant_q8_0(x):
    """quantize_row_q8_0_ref (ggml-quants.c:199-222): d = amax / 127, q = round(x / d)"""
    n, K = x.shape
    xb = x.reshape(n, K // 32, 32).astype(np.float32)
    d = np.abs(xb).max(axis=2, keepdims=True) / 127.0
    q = np.where(d > 0, np.round(xb / np.where(d > 0, d, 1)), 0).astype(np.int8)
    out = np.empty((n, K // 32, 34), np.uint8)
    out[..., 0:2] = d.astype(np.float16).view(np.uint8).reshape(n, K // 32, 2)
    out[..., 2:] = q.view(np.uint8)
    return out.reshape(n, -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["8b", "tiny", "tts", "tts-tiny"], default="8b",
                    help="8b / tiny: qwen3 arch; tts / tts-tiny: the omni TTS decoder's shape (arch llama, RoPE NORM, no q/k-norm; SURVEY.md 8(f) rank 2)")
    ap.add_argument("--types", choices=["q4_k_m", "f16", "q8_0", "q4_0", "q5_k"], default="q4_k_m")
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--distinct-layers", action="store_true", help="fresh random bytes per layer (small configs)")
    ap.add_argument("--n-ctx", type=int, default=40960)
    ap.add_argument("--separated", type=int, default=0, metavar="S",
                    help="greedy-decoding fixture with separated logits: S special tokens whose embedding dominates the residual stream, and whose "
                         "successor's lm-head row points along it (token s_i -> s_(i+1)): the winning logit leads by a margin far above any "
                         "summation-order noise, so two correct backends produce IDENTICAL greedy ids.  The ids are printed.")
    ap.add_argument("--vocab", choices=["none", "omni"], default="none",
                    help="omni: a byte-level BPE tokenizer carrying the special tokens tools/omni/omni.cpp looks up (SURVEY.md 8 row g1)")
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (CPU-sized runs of the omni harness)")
    ap.add_argument("--omni-tts-extra", action="store_true",
                    help="tts config: add the tensors omni.cpp's load_tts_weights_from_gguf reads beside the decoder (emb_code.0 / emb_text / projector_semantic.* / head_code.0; "
                         "src/llama-model.cpp:2458-2472 skips them in the model loader) and leave output.weight out (the decoder's head is head_code)")
    args = ap.parse_args()

    load_pkg()
    from llama_cpp_omni_amd import qwen3
    from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, GGML_TYPE_Q4_K, GGML_TYPE_Q8_0, row_size
    TTS = dict(n_embd=768, n_layer=20, n_head=12, n_head_kv=12, head_dim=64, n_ff=3072, n_vocab=32000, rms_eps=1e-6, rope_base=1e4, n_ctx_orig=4096)
    TTS_TINY = dict(n_embd=256, n_layer=2, n_head=4, n_head_kv=4, head_dim=64, n_ff=512, n_vocab=512, rms_eps=1e-6, rope_base=1e4, n_ctx_orig=4096)
    is_llama = args.config.startswith("tts")
    cfg = {"8b": qwen3.QWEN3_8B, "tiny": qwen3.TINY, "tts": TTS, "tts-tiny": TTS_TINY}[args.config]
    if args.layers:
        cfg = dict(cfg, n_layer=args.layers)
    if args.types == "q4_k_m":
        types, embd_ty, ftype = qwen3.q4_k_m_types(cfg), GGML_TYPE_Q4_K, 15        # LLAMA_FTYPE_MOSTLY_Q4_K_M
    elif args.types == "f16":
        types, embd_ty, ftype = qwen3.uniform_types(cfg, GGML_TYPE_F16), GGML_TYPE_F16, 1
    elif args.types == "q4_0":
        types, embd_ty, ftype = qwen3.uniform_types(cfg, 2), 2, 2                   # GGML_TYPE_Q4_0, LLAMA_FTYPE_MOSTLY_Q4_0
    elif args.types == "q5_k":
        types, embd_ty, ftype = qwen3.uniform_types(cfg, 13), 13, 16                # GGML_TYPE_Q5_K, LLAMA_FTYPE_MOSTLY_Q5_K_S
    else:
        types, embd_ty, ftype = qwen3.uniform_types(cfg, GGML_TYPE_Q8_0), GGML_TYPE_Q8_0, 7
    E, H, HK, D, F, V, L = cfg["n_embd"], cfg["n_head"], cfg["n_head_kv"], cfg["head_dim"], cfg["n_ff"], cfg["n_vocab"], cfg["n_layer"]

    # ---- tensor list in file order: (name, type, ne) with ne[0] the contiguous dimension
    tensors = [("token_embd.weight", embd_ty, (E, V)), ("output_norm.weight", GGML_TYPE_F32, (E,))]
    if args.omni_tts_extra:
        assert is_llama and E == 768
        tensors += [("emb_code.0.weight", GGML_TYPE_F16, (768, 6562)), ("emb_text.weight", GGML_TYPE_F16, (768, 152064)),
                    ("projector_semantic.linear1.weight", GGML_TYPE_F16, (4096, 768)), ("projector_semantic.linear1.bias", GGML_TYPE_F32, (768,)),
                    ("projector_semantic.linear2.weight", GGML_TYPE_F16, (768, 768)), ("projector_semantic.linear2.bias", GGML_TYPE_F32, (768,)),
                    ("head_code.0.weight", GGML_TYPE_F16, (768, 6562))]
    else:
        tensors.append(("output.weight", types["output"], (E, V)))
    for il in range(L):
        t = types[il]
        tensors += [(f"blk.{il}.attn_norm.weight", GGML_TYPE_F32, (E,)), (f"blk.{il}.attn_q.weight", t["attn_q"], (E, H * D)),
                    (f"blk.{il}.attn_k.weight", t["attn_k"], (E, HK * D)), (f"blk.{il}.attn_v.weight", t["attn_v"], (E, HK * D)),
                    (f"blk.{il}.attn_output.weight", t["attn_output"], (H * D, E))]
        if not is_llama:
            tensors += [(f"blk.{il}.attn_q_norm.weight", GGML_TYPE_F32, (D,)), (f"blk.{il}.attn_k_norm.weight", GGML_TYPE_F32, (D,))]
        tensors += [(f"blk.{il}.ffn_norm.weight", GGML_TYPE_F32, (E,)),
                    (f"blk.{il}.ffn_gate.weight", t["ffn_gate"], (E, F)), (f"blk.{il}.ffn_up.weight", t["ffn_up"], (E, F)),
                    (f"blk.{il}.ffn_down.weight", t["ffn_down"], (F, E))]

    def nbytes(ty, ne):
        rows = int(np.prod(ne[1:])) if len(ne) > 1 else 1
        return row_size(ty, ne[0]) * rows

    arch = "llama" if is_llama else "qwen3"
    kvs = [kv_str("general.architecture", arch), kv_str("general.name", f"qwen3-{args.config}-{args.types}-synthetic"), kv_u32("general.file_type", ftype),
           kv_u32("general.quantization_version", 2), kv_u32("general.alignment", ALIGN),
           kv_u32(f"{arch}.block_count", L), kv_u32(f"{arch}.context_length", args.n_ctx), kv_u32(f"{arch}.embedding_length", E),
           kv_u32(f"{arch}.feed_forward_length", F), kv_u32(f"{arch}.attention.head_count", H), kv_u32(f"{arch}.attention.head_count_kv", HK),
           kv_u32(f"{arch}.attention.key_length", D), kv_u32(f"{arch}.attention.value_length", D),
           kv_f32(f"{arch}.attention.layer_norm_rms_epsilon", cfg["rms_eps"]), kv_f32(f"{arch}.rope.freq_base", cfg["rope_base"]),
           kv_u32(f"{arch}.vocab_size", V)]
    kvs += omni_vocab_kvs(V) if args.vocab == "omni" else [kv_str("tokenizer.ggml.model", "no_vocab")]
    if is_llama:
        kvs.append(kv_u32(f"{arch}.rope.dimension_count", D))

    offs, off = [], 0
    for _, ty, ne in tensors:
        offs.append(off)
        off = (off + nbytes(ty, ne) + ALIGN - 1) // ALIGN * ALIGN
    head = struct.pack("<IIQQ", GGUF_MAGIC, GGUF_VERSION, len(tensors), len(kvs)) + b"".join(kvs)
    for (name, ty, ne), o in zip(tensors, offs):
        head += _s(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", d) for d in ne) + struct.pack("<IQ", ty, o)
    head += b"\0" * ((-len(head)) % ALIGN)

    rng = np.random.default_rng(args.seed)
    cache = {}

    def data_for(name, ty, ne):
        if ty == GGML_TYPE_F32 and len(ne) == 1:
            return np.ones(ne[0], np.float32).view(np.uint8)
        base = name.split(".", 2)[2] if name.startswith("blk.") else name
        key = name if args.distinct_layers else (base, ty, ne)
        if key not in cache:
            if not args.distinct_layers and len(cache) > 16:
                cache.clear()
            cache[key] = qwen3.random_blocks(rng, ty, ne[1], ne[0]).reshape(-1)
        return cache[key]

    # ---- separated-logits fixture (see --separated)
    special, patch = [], {}
    if args.separated:
        S = args.separated
        q80 = embd_ty == GGML_TYPE_Q8_0 and types["output"] == GGML_TYPE_Q8_0
        assert ((embd_ty == GGML_TYPE_Q4_K and types["output"] == 14) or q80) and E % 256 == 0, "--separated: Q4_K token_embd + Q6_K output, or all Q8_0"
        special = [int(V // 16 + (V - V // 8) * i // S) for i in range(S)]
        r2 = np.random.default_rng(args.seed + 77)
        big = qwen3.random_blocks(r2, embd_ty, S, E, std=300.0)                        # embedding rows ~ 20x the size of what 36 random layers add
        ehat = dequant_q8_0(big.reshape(S, -1), E) if q80 else dequant_q4_K(big.reshape(S, -1), E)
        ehat /= np.sqrt((ehat ** 2).mean(axis=1, keepdims=True))
        out_rows = quant_q8_0(ehat) if q80 else quant_q6_K(ehat)                         # successor rows: unit-rms copies of the embedding directions
        deq = dequant_q8_0(out_rows, E) if q80 else dequant_q6_K(out_rows, E)
        logit = deq @ ehat.T                                                             # [row j, token i]: the lm head applied to the pure embedding direction
        for i in range(S):
            col = logit[:, i].copy(); top = col[i]; col[i] = -np.inf
            assert top > 0.9 * E and col.max() < 0.7 * top, (i, top, col.max())          # the margin the fixture is built for (8B: runner-up ~ 0.06 top)
        patch["token_embd.weight"] = {special[i]: big[i] for i in range(S)}
        patch["output.weight"] = {special[(i + 1) % S]: out_rows[i] for i in range(S)}
        print("separated-logits fixture: start token", special[0], "cycle", special[:4], "...")

    with open(args.out, "wb") as f:
        f.write(head)
        base = f.tell()
        for (name, ty, ne), o in zip(tensors, offs):
            pad = base + o - f.tell()
            assert pad >= 0
            f.write(b"\0" * pad)
            d = data_for(name, ty, ne)
            assert d.nbytes == nbytes(ty, ne), (name, d.nbytes, nbytes(ty, ne))
            if name in patch:
                d = d.reshape(ne[1], -1).copy()
                if name == "output.weight":
                    d = qwen3.random_blocks(np.random.default_rng(args.seed + 78), ty, ne[1], ne[0], std=1e-3).reshape(ne[1], -1)    # every other row: tiny
                for row, bytes_ in patch[name].items():
                    d[row] = bytes_
                d = d.reshape(-1)
            f.write(d.tobytes() if d.nbytes < (1 << 26) else memoryview(np.ascontiguousarray(d)))
        f.write(b"\0" * ((-f.tell()) % ALIGN))
    print(f"wrote {args.out}: {len(tensors)} tensors, {os.path.getsize(args.out) / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
