"""Golden vectors for SURVEY.md row f2 (OpenCLIP ViT-H/14 towers + CLIP BPE tokeniser) from an INDEPENDENT third-party
implementation: HuggingFace `transformers` (installed in this image; `open_clip_torch`, which the reference imports at
lvdm/modules/encoders/condition.py:188,307, is not).

    python tests/golden/make_openclip_golden.py        # -> tests/golden/openclip_hf.npz, tests/golden/clip_bpe_hf.json

What is pinned
  * vision tokens, condition.py:340-372 (`encode_with_vision_transformer`: conv1 patchify, class token, positional
    embedding, ln_pre, ALL 32 resblocks, no ln_post / proj) == `CLIPVisionModel(...).last_hidden_state`
    (transformers' CLIPVisionTransformer returns the encoder output BEFORE `post_layernorm`);
  * text tokens, condition.py:215-231 with layer="penultimate" (23 of 24 causal resblocks, then ln_final) ==
    `final_layer_norm(CLIPTextModel(..., output_hidden_states=True).hidden_states[-2])` -- the way Stable Diffusion 2 drives the
    same ViT-H text tower through transformers;
  * at the real ViT-H/14 geometry (vision 1280 x 32 layers x 16 heads of 80, patch 14, 224 px, mlp 5120; text 1024 x 24 x 16
    heads of 64, 77 tokens, vocabulary 49408, mlp 4096), `hidden_act="gelu"` (open_clip's nn.GELU, not OpenAI's quick_gelu).
  The weights are this repo's synthetic state dict under OPEN_CLIP'S parameter names (`synth.synth_tensor`, a pure function of
  name, shape and seed -- the test regenerates them, only inputs and outputs are stored); `to_hf_*` below renames them to
  transformers' names (in_proj_weight split into q / k / v, `ln_pre` -> `pre_layrnorm`, `mlp.c_fc` -> `mlp.fc1`, ...): the
  published open_clip -> transformers checkpoint conversion.  So the golden states: "an independent CLIP implementation, fed the
  same tensors under that renaming, produces these tokens" -- which is what the reference's open_clip call sites compute.
  * tokeniser: open_clip's `SimpleTokenizer` / `tokenize` (restated in lvdm/clip_tokenizer.py) against transformers'
    `CLIPTokenizer` (the Rust `tokenizers` BPE) on a merge table built here (no CLIP vocabulary file exists on this image): ids
    of a prompt list covering case, whitespace, contractions, digits, punctuation runs, non-ASCII letters and truncation.
Not pinned (no independent implementation in the image): kornia's antialiased bicubic resize (condition.py:323-331).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tooncrafter_amd import synth                                   # noqa: E402
from tooncrafter_amd.lvdm.openclip import ARCH                      # noqa: E402

SEED = 1234
VIS_PREFIX, TXT_PREFIX = "embedder.model.visual.", "cond_stage_model.model."


def open_clip_shapes(tower: str, arch="ViT-H-14"):
    """{open_clip parameter name: shape} of one tower (names as open_clip/transformer.py, model.py @ v2.22.0 register them)."""
    a = ARCH[arch]
    c = a[tower]
    d, mlp = c["width"], c["mlp"]
    s = {}
    for i in range(c["layers"]):
        p = f"transformer.resblocks.{i}."
        s.update({p + "ln_1.weight": (d,), p + "ln_1.bias": (d,), p + "attn.in_proj_weight": (3 * d, d),
                  p + "attn.in_proj_bias": (3 * d,), p + "attn.out_proj.weight": (d, d), p + "attn.out_proj.bias": (d,),
                  p + "ln_2.weight": (d,), p + "ln_2.bias": (d,), p + "mlp.c_fc.weight": (mlp, d), p + "mlp.c_fc.bias": (mlp,),
                  p + "mlp.c_proj.weight": (d, mlp), p + "mlp.c_proj.bias": (d,)})
    if tower == "vision":
        g = c["image"] // c["patch"]
        s.update({"conv1.weight": (d, 3, c["patch"], c["patch"]), "class_embedding": (d,),
                  "positional_embedding": (g * g + 1, d), "ln_pre.weight": (d,), "ln_pre.bias": (d,),
                  "ln_post.weight": (d,), "ln_post.bias": (d,), "proj": (d, a["embed_dim"])})
    else:
        s.update({"token_embedding.weight": (c["vocab"], d), "positional_embedding": (c["context"], d),
                  "ln_final.weight": (d,), "ln_final.bias": (d,), "text_projection": (d, a["embed_dim"])})
    return s


def synth_tower(tower: str, arch="ViT-H-14"):
    pre = VIS_PREFIX if tower == "vision" else TXT_PREFIX
    return {k: synth.synth_tensor(pre + k, shp, SEED) for k, shp in open_clip_shapes(tower, arch).items()}


def _layers_to_hf(sd, n_layers, d, pre):
    out = {}
    for i in range(n_layers):
        s, t = f"transformer.resblocks.{i}.", f"{pre}encoder.layers.{i}."
        w, b = sd[s + "attn.in_proj_weight"], sd[s + "attn.in_proj_bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            out[t + f"self_attn.{n}.weight"], out[t + f"self_attn.{n}.bias"] = w[j * d:(j + 1) * d], b[j * d:(j + 1) * d]
        for a, b_ in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                      ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            out[t + b_ + ".weight"], out[t + b_ + ".bias"] = sd[s + a + ".weight"], sd[s + a + ".bias"]
    return out


def to_hf_vision(sd, cfg, p=""):
    """`p`: transformers <= 4 nests the tower under "vision_model." / "text_model."; 5.x registers it at the top level."""
    out = _layers_to_hf(sd, cfg["layers"], cfg["width"], p)
    out.update({p + "embeddings.class_embedding": sd["class_embedding"], p + "embeddings.patch_embedding.weight": sd["conv1.weight"],
                p + "embeddings.position_embedding.weight": sd["positional_embedding"],
                p + "pre_layrnorm.weight": sd["ln_pre.weight"], p + "pre_layrnorm.bias": sd["ln_pre.bias"],
                p + "post_layernorm.weight": sd["ln_post.weight"], p + "post_layernorm.bias": sd["ln_post.bias"]})
    return out


def to_hf_text(sd, cfg, p=""):
    out = _layers_to_hf(sd, cfg["layers"], cfg["width"], p)
    out.update({p + "embeddings.token_embedding.weight": sd["token_embedding.weight"],
                p + "embeddings.position_embedding.weight": sd["positional_embedding"],
                p + "final_layer_norm.weight": sd["ln_final.weight"], p + "final_layer_norm.bias": sd["ln_final.bias"]})
    return out


def _prefix(model, inner):
    return inner + "." if any(k.startswith(inner + ".") for k in model.state_dict()) else ""


def _load(model, hf_sd):
    own = model.state_dict()
    extra = {k for k in own if k not in hf_sd and not k.endswith("position_ids")}
    assert not extra, f"transformers parameters without a source tensor: {sorted(extra)[:5]}"
    missing, unexpected = model.load_state_dict(hf_sd, strict=False)
    assert not unexpected, unexpected
    return model.eval()


def hf_vision_tokens(sd, image, cfg):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    c = CLIPVisionConfig(hidden_size=cfg["width"], intermediate_size=cfg["mlp"], num_hidden_layers=cfg["layers"],
                         num_attention_heads=cfg["heads"], image_size=cfg["image"], patch_size=cfg["patch"], hidden_act="gelu",
                         layer_norm_eps=1e-5, attention_dropout=0.0, projection_dim=1024)
    c._attn_implementation = "eager"
    m = CLIPVisionModel(c)
    m = _load(m, to_hf_vision(sd, cfg, _prefix(m, "vision_model")))
    with torch.no_grad():
        return m(pixel_values=image).last_hidden_state


def hf_text_tokens(sd, tokens, cfg):
    from transformers import CLIPTextConfig, CLIPTextModel
    c = CLIPTextConfig(vocab_size=cfg["vocab"], hidden_size=cfg["width"], intermediate_size=cfg["mlp"],
                       num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"], max_position_embeddings=cfg["context"],
                       hidden_act="gelu", layer_norm_eps=1e-5, attention_dropout=0.0, projection_dim=1024,
                       bos_token_id=cfg["vocab"] - 2, eos_token_id=cfg["vocab"] - 1, pad_token_id=0)
    c._attn_implementation = "eager"
    m = CLIPTextModel(c)
    m = _load(m, to_hf_text(sd, cfg, _prefix(m, "text_model")))
    ln = (m.text_model if hasattr(m, "text_model") else m).final_layer_norm
    with torch.no_grad():
        hs = m(input_ids=tokens, output_hidden_states=True).hidden_states
        return ln(hs[-2]), ln(hs[-1])


# ---------------------------------------------------------------- tokeniser

CORPUS = ("the quick brown fox jumps over the lazy dog . an anime girl walking in the rain , detailed line art , "
          "a man and a woman are dancing in the street at night ; two frames of a cartoon , the same scene , smooth motion ! "
          "it's a dog's life , they're here , we've won , i'm fine , she'll go , he'd know , don't stop . "
          "1 2 3 4 5 6 7 8 9 0 10 2024 3.14 ; café naïve über señor ; hello hello hello world world the the the and and").split()

PROMPTS = ["", "an anime scene", "The  Quick\tBROWN fox\n jumps!", "it's a dog's life; they're here, we've won -- I'm fine",
           "walking in the rain, detailed line-art (2024), 3.14 px", "café naïve über señor", "smooth motion!!! ... ???",
           "a man and a woman are dancing in the street at night " * 12, "x", "hello   world .", "don't  stop &amp; go &lt;b&gt;",
           "two frames of a cartoon: the same scene @ 24fps #anime"]


def learn_merges(words, n_merges=220):
    """A small, deterministic BPE merge table over the byte alphabet (plain pair counting; ties by first occurrence): the table is
    DATA for both tokenisers, how it was made does not matter."""
    from tooncrafter_amd.lvdm.clip_tokenizer import bytes_to_unicode
    b2u = bytes_to_unicode()
    seqs = []
    for w in words:
        sym = [b2u[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        seqs.append(sym)
    merges = []
    for _ in range(n_merges):
        counts = {}
        for s in seqs:
            for a, b in zip(s, s[1:]):
                counts[(a, b)] = counts.get((a, b), 0) + 1
        if not counts:
            break
        best = max(counts.items(), key=lambda kv: kv[1])[0]
        merges.append(best)
        for s in seqs:
            i = 0
            while i < len(s) - 1:
                if s[i] == best[0] and s[i + 1] == best[1]:
                    s[i:i + 2] = [s[i] + s[i + 1]]
                else:
                    i += 1
    return [f"{a} {b}" for a, b in merges]


def open_clip_vocab(merges):
    """The id assignment of open_clip's SimpleTokenizer (tokenizer.py @ v2.22.0): 256 byte symbols, the same with '</w>', the
    merged symbols in rule order, then the two specials."""
    from tooncrafter_amd.lvdm.clip_tokenizer import bytes_to_unicode
    v = list(bytes_to_unicode().values())
    v = v + [s + "</w>" for s in v] + ["".join(m.split()) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
    return {s: i for i, s in enumerate(v)}


def hf_token_ids(merges, prompts, context=77):
    from transformers import CLIPTokenizer
    vocab = open_clip_vocab(merges)
    tk = CLIPTokenizer(vocab=vocab, merges=[tuple(m.split()) for m in merges])
    out = []
    import html
    for p in prompts:
        # open_clip's basic_clean un-escapes HTML entities (twice) before anything else; transformers has no such step, so it
        # is given the un-escaped text: the pin covers case folding, whitespace, the split regex and the BPE merges
        out.append(tk(html.unescape(html.unescape(p)).strip(), add_special_tokens=True)["input_ids"])
    return out, vocab


def tokenizer_golden():
    merges = learn_merges(CORPUS)
    ids, vocab = hf_token_ids(merges, PROMPTS)
    with open(os.path.join(HERE, "clip_bpe_hf.json"), "w") as f:
        json.dump({"merges": merges, "prompts": PROMPTS, "hf_input_ids": ids, "n_vocab": len(vocab),
                   "made_by": "transformers.CLIPTokenizer (tokenizers BPE) " + __import__("transformers").__version__}, f,
                  ensure_ascii=True, indent=0)
    print("tokeniser:", len(merges), "merges,", len(PROMPTS), "prompts; ids[1] =", ids[1])


def main():
    torch.manual_seed(0)
    a = ARCH["ViT-H-14"]
    g = torch.Generator().manual_seed(20260922)
    image = torch.randn(1, 3, 224, 224, generator=g).half().float()          # stored as fp16: what is stored is what was run
    tokens = torch.randint(0, a["text"]["vocab"] - 2, (2, 77), generator=g)
    tokens[:, 0] = a["text"]["vocab"] - 2
    tokens[0, 20], tokens[0, 21:] = a["text"]["vocab"] - 1, 0                    # a short prompt: eot, then open_clip's zero padding
    tokens[1, 76] = a["text"]["vocab"] - 1
    yv = hf_vision_tokens(synth_tower("vision"), image, a["vision"])
    print("vision tokens", tuple(yv.shape), "std", float(yv.std()))
    yt, yt_last = hf_text_tokens(synth_tower("text"), tokens, a["text"])
    print("text tokens", tuple(yt.shape), "std", float(yt.std()), "penultimate vs last rel", float((yt - yt_last).norm() / yt.norm()))
    # tiny towers as well (seconds on the CPU; the `-m "not gpu"` suite replays them without building 1 B parameters)
    tiny = dict(embed_dim=64, vision=dict(width=160, layers=2, heads=2, patch=14, image=42, mlp=320),
                text=dict(width=128, layers=3, heads=2, context=7, vocab=50, mlp=256))
    ARCH["tiny-golden"] = tiny
    img_t = torch.randn(2, 3, 42, 42, generator=g)
    tok_t = torch.randint(0, 48, (2, 7), generator=g)
    yv_t = hf_vision_tokens(synth_tower("vision", "tiny-golden"), img_t, tiny["vision"])
    yt_t, _ = hf_text_tokens(synth_tower("text", "tiny-golden"), tok_t, tiny["text"])
    np.savez_compressed(os.path.join(HERE, "openclip_hf.npz"), image=image.numpy().astype(np.float16), tokens=tokens.numpy(),
                        vision_tokens=yv.numpy(), text_tokens=yt.numpy(), tiny_image=img_t.numpy(), tiny_tokens=tok_t.numpy(),
                        tiny_vision_tokens=yv_t.numpy(), tiny_text_tokens=yt_t.numpy(), seed=np.int64(SEED))
    tokenizer_golden()


if __name__ == "__main__":
    tokenizer_golden() if sys.argv[1:] == ["tokenizer"] else main()
