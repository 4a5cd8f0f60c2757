"""Oracle: umT5-XXL text encoder (WanTextEncoder), functional CPU restatement (test infrastructure only).

Follows ``diffsynth/models/wan_video_text_encoder.py``: ``T5LayerNorm`` :25-38, ``T5Attention`` :41-94,
``T5FeedForward`` (gated tanh-GELU written out op by op) :19-22,97-116, ``T5SelfAttention`` :119-150,
``T5RelativeEmbedding`` :153-198, ``WanTextEncoder.forward`` :246-257, and the zeroing of padded rows in
``WanVideoUnit_PromptEmbedder.encode_prompt`` (``pipelines/wan_video.py:404-412``).
"""
import math

import torch
import torch.nn.functional as F


def t5_layer_norm(x, weight, eps=1e-6):
    y = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    if weight.dtype in (torch.float16, torch.bfloat16):
        y = y.type_as(weight)
    return weight * y


def relative_position_bucket(rel_pos, num_buckets, max_dist=128):
    """Bidirectional bucketing, :176-198."""
    nb = num_buckets // 2
    buckets = (rel_pos > 0).long() * nb
    rel_pos = torch.abs(rel_pos)
    max_exact = nb // 2
    large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(rel_pos < max_exact, rel_pos, large)


def position_bias(embedding, lq, lk):
    """T5RelativeEmbedding.forward :164-174 -> (1, heads, lq, lk)."""
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    emb = F.embedding(relative_position_bucket(rel, embedding.shape[0]), embedding)
    return emb.permute(2, 0, 1).unsqueeze(0).contiguous()


def gelu_tanh_explicit(x):
    """The module-level GELU of the reference (:19-22): every elementwise op rounds in x.dtype."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def t5_attention(sd, p, x, num_heads, mask, pos_bias):
    b, l, _ = x.shape
    q = F.linear(x, sd[p + ".q.weight"]).view(b, -1, num_heads, sd[p + ".q.weight"].shape[0] // num_heads)
    k = F.linear(x, sd[p + ".k.weight"]).view(b, -1, num_heads, q.shape[-1])
    v = F.linear(x, sd[p + ".v.weight"]).view(b, -1, num_heads, q.shape[-1])
    bias = x.new_zeros(b, num_heads, q.size(1), k.size(1))
    if pos_bias is not None:
        bias += pos_bias
    if mask is not None:
        bias.masked_fill_(mask.view(b, 1, 1, -1) == 0, torch.finfo(x.dtype).min)
    attn = torch.einsum("binc,bjnc->bnij", q, k) + bias          # T5: no 1/sqrt(d) scaling
    attn = F.softmax(attn.float(), dim=-1).type_as(attn)
    y = torch.einsum("bnij,bjnc->binc", attn, v).reshape(b, -1, q.shape[2] * q.shape[3])
    return F.linear(y, sd[p + ".o.weight"])


def t5_block(sd, p, x, num_heads, mask):
    e = position_bias(sd[p + ".pos_embedding.embedding.weight"], x.size(1), x.size(1)).to(x.dtype)
    x = x + t5_attention(sd, p + ".attn", t5_layer_norm(x, sd[p + ".norm1.weight"]), num_heads, mask, e)
    h = t5_layer_norm(x, sd[p + ".norm2.weight"])
    h = F.linear(h, sd[p + ".ffn.fc1.weight"]) * gelu_tanh_explicit(F.linear(h, sd[p + ".ffn.gate.0.weight"]))
    return x + F.linear(h, sd[p + ".ffn.fc2.weight"])


def text_encoder(sd, ids, mask, num_heads):
    """WanTextEncoder.forward (shared_pos=False, eval) :246-257."""
    x = F.embedding(ids, sd["token_embedding.weight"])
    i = 0
    while f"blocks.{i}.norm1.weight" in sd:
        x = t5_block(sd, f"blocks.{i}", x, num_heads, mask)
        i += 1
    return t5_layer_norm(x, sd["norm.weight"])


def encode_prompt(sd, ids, mask, num_heads):
    """pipelines/wan_video.py:404-412 — encoder output with rows >= seq_len zeroed."""
    emb = text_encoder(sd, ids, mask, num_heads)
    for v in mask.gt(0).sum(dim=1).long():
        emb[:, v:] = 0
    return emb
