"""CPU oracle for the text side -- TEST INFRASTRUCTURE ONLY.

The reference builds its text encoder with `BertModel.from_pretrained(args.ck_bert)` and drops the pooler
(lib/_utils.py:38-40, :50-52; train.py:595-602).  `BertModel` comes from `./bert`, a copy of HF transformers 3.0.2
`modeling_bert.py` that is ABSENT from /root/reference: a third-party dependency, pinned by the
reference's README / requirements to transformers==3.0.2.  This file restates that model's published forward:

  embeddings   word[ids] + token_type[0] + position[0..N-1]  -> LayerNorm(eps = layer_norm_eps, 1e-12) -> dropout
  layer (x12)  q,k,v = Linear(x); scores = q k^T / sqrt(head_dim) + (1 - mask)[:, None, None, :] * -10000
               probs = softmax(scores) -> dropout; context = probs v (heads merged)
               x1 = LayerNorm(dropout(Linear(context)) + x)
               x2 = LayerNorm(dropout(Linear(gelu_erf(Linear(x1)))) + x1)
  output       last_hidden_state [B, N, H]   (the call sites take `[0]` and permute to (B, H, N))

and the host-side token pipeline of data/dataset_refer_bert.py:58-81 (`pad_ids`).

Parity pin: tests/golden/make_golden.py (`bert_cases`) runs the `transformers.BertModel` installed in the build image (5.15.0; the
arithmetic of the encoder did not change since 3.0.2 apart from the value of the additive mask constant, which underflows to the same
probabilities) on deterministic weights and writes tests/golden/bert_micro.npz; tests/test_oracle_golden.py checks this file against it.
No vector from the reference itself exists for this component (it ships no tests and no weights): parity for the text side is pinned to
the third-party implementation, not to the reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F


def bert_forward(sd, ids, attention_mask, num_heads, eps=1e-12, prefix="", token_type_ids=None):
    """sd: HF BertModel state dict (key names of transformers' `BertModel`, e.g. `encoder.layer.0.attention.self.query.weight`).
    ids int64 [B, N]; attention_mask [B, N] of 0/1.  Eval-mode forward (dropout off) -> last_hidden_state [B, N, H]."""
    g = lambda k: sd[prefix + k]
    B, N = ids.shape
    x = g("embeddings.word_embeddings.weight")[ids]
    tt = token_type_ids if token_type_ids is not None else torch.zeros_like(ids)
    x = x + g("embeddings.token_type_embeddings.weight")[tt]
    x = x + g("embeddings.position_embeddings.weight")[:N][None]
    H = x.shape[-1]
    x = F.layer_norm(x, (H,), g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
    ext = (1.0 - attention_mask.to(x.dtype))[:, None, None, :] * -10000.0
    hd = H // num_heads
    li = 0
    while (prefix + f"encoder.layer.{li}.attention.self.query.weight") in sd:
        p = f"encoder.layer.{li}."
        lin = lambda t, name: F.linear(t, g(p + name + ".weight"), g(p + name + ".bias"))
        split = lambda t: t.view(B, N, num_heads, hd).permute(0, 2, 1, 3)
        q, k, v = split(lin(x, "attention.self.query")), split(lin(x, "attention.self.key")), split(lin(x, "attention.self.value"))
        probs = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd) + ext, dim=-1)
        ctx = (probs @ v).permute(0, 2, 1, 3).reshape(B, N, H)
        x1 = F.layer_norm(lin(ctx, "attention.output.dense") + x, (H,), g(p + "attention.output.LayerNorm.weight"), g(p + "attention.output.LayerNorm.bias"), eps)
        h = F.gelu(lin(x1, "intermediate.dense"))          # exact erf form ("gelu" of BertConfig.hidden_act)
        x = F.layer_norm(lin(h, "output.dense") + x1, (H,), g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), eps)
        li += 1
    return x


def pad_ids(token_ids, max_tokens):
    """data/dataset_refer_bert.py:62-75: truncate the encoded sentence to max_tokens ids, zero-pad, mask = 1 on real tokens."""
    token_ids = list(token_ids)[:max_tokens]
    ids = [0] * max_tokens
    mask = [0] * max_tokens
    ids[:len(token_ids)] = token_ids
    mask[:len(token_ids)] = [1] * len(token_ids)
    return ids, mask
