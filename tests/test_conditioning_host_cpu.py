"""Host-side pieces of the conditioning stack restated from third-party packages that are absent here
(open_clip's tokeniser, kornia's resize): the mechanics are tested here; the tokeniser's parity is pinned in
test_openclip_golden_cpu.py (transformers.CLIPTokenizer), the resize's is unpinned and the
modules say so."""
import gzip

import pytest
import torch
import torch.nn.functional as F

from tooncrafter_amd.lvdm.clip_tokenizer import CLIPTokenizer, bytes_to_unicode
from tooncrafter_amd.lvdm.condition import kornia_resize


def test_byte_alphabet_is_reversible():
    b2u = bytes_to_unicode()
    assert len(b2u) == 256 and len(set(b2u.values())) == 256
    assert b2u[ord("a")] == "a" and b2u[ord(" ")] == chr(256 + 32)      # printable bytes map to themselves, others move up


def _tiny():
    # rank-ordered merges over the byte alphabet: "th", "the</w>" ..., as in a real table only much shorter
    return CLIPTokenizer(["t h", "th e</w>", "c a", "ca t</w>", "a n", "an d</w>"], context_length=12)


def test_bpe_merges_in_rank_order_and_specials():
    tk = _tiny()
    n_base = 512                                                        # 256 bytes + 256 end-of-word variants
    assert tk.sot_token == n_base + 6 and tk.eot_token == n_base + 7
    assert tk.bpe("the") == "the</w>" and tk.bpe("cat") == "cat</w>"
    assert tk.bpe("then") == "th e n</w>"                               # "e</w>" does not occur: the second rule cannot fire
    ids = tk.encode("The  CAT &amp; the hat")                           # case folding, whitespace and HTML clean-up
    the, cat = tk.encoder["the</w>"], tk.encoder["cat</w>"]
    assert ids[0] == the and ids[1] == cat and ids[2] == tk.encoder["&</w>"] and ids[3] == the
    assert ids[4:] == [tk.encoder["h"], tk.encoder["a"], tk.encoder["t</w>"]]


def test_tokenize_layout_truncation_and_empty_prompt():
    tk = _tiny()
    out = tk(["the cat", "", "the " * 40])
    assert out.shape == (3, 12) and out.dtype == torch.long
    assert out[0].tolist()[:4] == [tk.sot_token, tk.encoder["the</w>"], tk.encoder["cat</w>"], tk.eot_token] and out[0, 4:].sum() == 0
    assert out[1].tolist()[:3] == [tk.sot_token, tk.eot_token, 0]
    assert out[2, 0] == tk.sot_token and out[2, -1] == tk.eot_token and (out[2, 1:-1] == tk.encoder["the</w>"]).all()


def test_vocab_file_layout(tmp_path):
    """open_clip's table: one header line, then the rules; a full-size table gives <start_of_text> = 49406."""
    rules = [f"x{i} y{i}" for i in range(49152 - 256 - 2)]
    path = tmp_path / "bpe.txt.gz"
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: test\n" + "\n".join(rules) + "\n" + "ignored tail\n")
    tk = CLIPTokenizer(str(path))
    assert tk.sot_token == 49406 and tk.eot_token == 49407 and len(tk.bpe_ranks) == 48894
    assert tk("")[0, :3].tolist() == [49406, 49407, 0]


def test_embedder_tokenises_with_a_supplied_table(tmp_path, monkeypatch):
    from tooncrafter_amd.lvdm.condition import FrozenOpenCLIPEmbedder
    emb = FrozenOpenCLIPEmbedder.__new__(FrozenOpenCLIPEmbedder)
    torch.nn.Module.__init__(emb)
    emb.max_length = 77
    assert emb.tokenize(["", ""])[:, :2].tolist() == [[49406, 49407]] * 2
    monkeypatch.delenv("TC_CLIP_BPE_VOCAB", raising=False)
    with pytest.raises(RuntimeError):
        emb.tokenize("a cat")
    path = tmp_path / "bpe.txt"
    path.write_text("#version: test\nc a\nca t</w>\n")
    monkeypatch.setenv("TC_CLIP_BPE_VOCAB", str(path))
    tok = emb.tokenize("a cat")
    assert tok.shape == (1, 77) and tok[0, 0] == emb._bpe.sot_token and tok[0, 3] == emb._bpe.eot_token


def test_kornia_resize_properties():
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 320, 512, generator=g) * 2 - 1
    y = kornia_resize(x, (224, 224))
    assert y.shape == (2, 3, 224, 224) and torch.isfinite(y).all()
    c = torch.full((1, 3, 320, 512), 0.37)
    assert torch.allclose(kornia_resize(c, (224, 224)), torch.full((1, 3, 224, 224), 0.37), atol=1e-6)   # blur + bicubic keep constants
    up = torch.rand(1, 3, 100, 100, generator=g)
    assert torch.equal(kornia_resize(up, (224, 224)), F.interpolate(up, size=(224, 224), mode="bicubic", align_corners=True))  # no blur when upscaling
    assert torch.equal(kornia_resize(x, (224, 224), antialias=False), F.interpolate(x, size=(224, 224), mode="bicubic", align_corners=True))
    # the blur is what removes aliasing: a Nyquist-rate checkerboard survives plain bicubic, not the antialiased path
    cb = ((torch.arange(320).view(-1, 1) + torch.arange(512).view(1, -1)) % 2).float().view(1, 1, 320, 512).expand(1, 3, -1, -1) * 2 - 1
    assert kornia_resize(cb, (224, 224)).std() < 0.3 * kornia_resize(cb, (224, 224), antialias=False).std()
    # kernel geometry of the 320 x 512 -> 224 x 224 case: sigma = (f - 1) / 2, size = odd(int(max(4 sigma, 3)))
    assert [int(max(4 * max((f - 1) / 2, 0.001), 3)) | 1 for f in (320 / 224, 512 / 224)] == [3, 3]
