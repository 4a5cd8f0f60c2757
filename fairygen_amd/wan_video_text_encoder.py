"""umT5-XXL text encoder + tokenizer plumbing for the Wan pipeline, on the HIP kernels.

Mirror of ``diffsynth/models/wan_video_text_encoder.py`` (``WanTextEncoder`` :212-257, ``T5SelfAttention`` :119-150,
``T5Attention`` :41-94, ``T5FeedForward`` :97-116, ``T5RelativeEmbedding`` :153-198, ``HuggingfaceTokenizer`` :285-329):
same constructor kwargs and parameter names (the state dict of the default config hashes to the reference's
``9c8818c2…``), two passes per clip before the denoise loop.  GEMMs run on hipBLASLt; T5LayerNorm, the biased/masked
softmax and the gated GELU are HIP kernels with the reference's bf16 rounding points.  head_dim is 64 here, so the
(512 x 512) attention uses two batched GEMMs around ``hip.softmax_bias`` rather than the head_dim-128 MFMA kernel.
"""
import html
import math
import re

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip


class T5LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))


class T5Attention(nn.Module):
    def __init__(self, dim, dim_attn, num_heads, dropout=0.1):
        assert dim_attn % num_heads == 0
        super().__init__()
        self.dim, self.dim_attn, self.num_heads, self.head_dim = dim, dim_attn, num_heads, dim_attn // num_heads
        self.q, self.k, self.v = (nn.Linear(dim, dim_attn, bias=False) for _ in range(3))
        self.o = nn.Linear(dim_attn, dim, bias=False)
        self.dropout = nn.Dropout(dropout)


class T5FeedForward(nn.Module):
    def __init__(self, dim, dim_ffn, dropout=0.1):
        super().__init__()
        self.dim, self.dim_ffn = dim, dim_ffn
        self.gate = nn.Sequential(nn.Linear(dim, dim_ffn, bias=False), nn.Identity())      # [1] stands for the GELU module
        self.fc1 = nn.Linear(dim, dim_ffn, bias=False)
        self.fc2 = nn.Linear(dim_ffn, dim, bias=False)
        self.dropout = nn.Dropout(dropout)


class T5RelativeEmbedding(nn.Module):
    def __init__(self, num_buckets, num_heads, bidirectional, max_dist=128):
        super().__init__()
        assert bidirectional
        self.num_buckets, self.num_heads, self.max_dist = num_buckets, num_heads, max_dist
        self.embedding = nn.Embedding(num_buckets, num_heads)

    def bucket_table(self, lq, lk):
        """(lq, lk) bucket indices, host integer math of :176-198."""
        rel = torch.arange(lk, device="cpu").unsqueeze(0) - torch.arange(lq, device="cpu").unsqueeze(1)
        nb = self.num_buckets // 2
        buckets = (rel > 0).long() * nb
        rel = rel.abs()
        max_exact = nb // 2
        large = max_exact + (torch.log(rel.float() / max_exact) / math.log(self.max_dist / max_exact) * (nb - max_exact)).long()
        large = torch.min(large, torch.full_like(large, nb - 1))
        return buckets + torch.where(rel < max_exact, rel, large)


class T5SelfAttention(nn.Module):
    def __init__(self, dim, dim_attn, dim_ffn, num_heads, num_buckets, shared_pos=True, dropout=0.1):
        super().__init__()
        self.norm1 = T5LayerNorm(dim)
        self.attn = T5Attention(dim, dim_attn, num_heads, dropout)
        self.norm2 = T5LayerNorm(dim)
        self.ffn = T5FeedForward(dim, dim_ffn, dropout)
        self.pos_embedding = None if shared_pos else T5RelativeEmbedding(num_buckets, num_heads, bidirectional=True)


class WanTextEncoder(nn.Module):
    def __init__(self, vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32,
                 shared_pos=False, dropout=0.1):
        super().__init__()
        if shared_pos:
            raise NotImplementedError("shared_pos=True is not used by the Wan text encoder")
        self.dim, self.dim_attn, self.dim_ffn, self.num_heads = dim, dim_attn, dim_ffn, num_heads
        self.num_layers, self.num_buckets, self.shared_pos = num_layers, num_buckets, shared_pos
        self.token_embedding = nn.Embedding(vocab, dim)
        self.pos_embedding = None
        self.dropout = nn.Dropout(dropout)
        self.blocks = nn.ModuleList([T5SelfAttention(dim, dim_attn, dim_ffn, num_heads, num_buckets, shared_pos, dropout)
                                     for _ in range(num_layers)])
        self.norm = T5LayerNorm(dim)
        self._buckets = {}

    def _norm(self, x, norm):
        return hip.rmsnorm_rope(x, norm.weight, 1, norm.eps)

    def forward(self, ids, mask=None):
        """ids (1, L) int64, mask (1, L) {0,1} -> (1, L, dim) (eval mode: the dropouts are identity)."""
        assert ids.dim() == 2 and ids.shape[0] == 1, "one prompt at a time (as WanVideoUnit_PromptEmbedder calls it)"
        l = ids.shape[1]
        nh, hd = self.num_heads, self.dim_attn // self.num_heads
        x = F.embedding(ids, self.token_embedding.weight).contiguous()
        key_mask = None if mask is None else mask[0].to(device=x.device, dtype=torch.int32).contiguous()
        if l not in self._buckets:
            self._buckets = {l: self.blocks[0].pos_embedding.bucket_table(l, l).to(x.device)}
        buckets = self._buckets[l]
        for blk in self.blocks:
            a = blk.attn
            bias = F.embedding(buckets, blk.pos_embedding.embedding.weight).permute(2, 0, 1).contiguous()      # (heads, L, L)
            h = self._norm(x, blk.norm1)
            q = F.linear(h, a.q.weight).view(l, nh, hd).transpose(0, 1)
            k = F.linear(h, a.k.weight).view(l, nh, hd).transpose(0, 1)
            v = F.linear(h, a.v.weight).view(l, nh, hd).transpose(0, 1)
            scores = torch.bmm(q, k.transpose(1, 2)).contiguous()                                            # (heads, L, L) bf16
            probs = hip.softmax_bias(scores.view(nh * l, l), bias.view(nh * l, l), key_mask).view(nh, l, l)
            y = torch.bmm(probs, v).transpose(0, 1).reshape(1, l, nh * hd).contiguous()
            x = hip.gate_residual(x, F.linear(y, a.o.weight))
            h = self._norm(x, blk.norm2)
            g = hip.gated_gelu(F.linear(h, blk.ffn.fc1.weight), F.linear(h, blk.ffn.gate[0].weight))
            x = hip.gate_residual(x, F.linear(g, blk.ffn.fc2.weight))
        return self._norm(x, self.norm)


# ----------------------------------------------------------------------------------------------- tokenizer
def basic_clean(text):
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ModuleNotFoundError:      # ftfy is optional in this image; it only repairs mojibake
        pass
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text):
    return re.sub(r"\s+", " ", text).strip()


class HuggingfaceTokenizer:
    """``HuggingfaceTokenizer(name=<local dir>, seq_len=512, clean='whitespace')`` as the pipeline builds it
    (pipelines/wan_video.py:157-159); ``name`` must be a local directory (no downloads)."""

    def __init__(self, name, seq_len=None, clean=None, **kwargs):
        assert clean in (None, "whitespace", "lower")
        from transformers import AutoTokenizer
        self.name, self.seq_len, self.clean = name, seq_len, clean
        self.tokenizer = AutoTokenizer.from_pretrained(name, **kwargs)
        self.vocab_size = self.tokenizer.vocab_size

    def __call__(self, sequence, **kwargs):
        return_mask = kwargs.pop("return_mask", False)
        opts = {"return_tensors": "pt"}
        if self.seq_len is not None:
            opts.update(padding="max_length", truncation=True, max_length=self.seq_len)
        opts.update(**kwargs)
        if isinstance(sequence, str):
            sequence = [sequence]
        if self.clean:
            sequence = [self._clean(u) for u in sequence]
        ids = self.tokenizer(sequence, **opts)
        return (ids.input_ids, ids.attention_mask) if return_mask else ids.input_ids

    def _clean(self, text):
        text = whitespace_clean(basic_clean(text))
        return text.lower() if self.clean == "lower" else text
