"""`BertModel` of the reference's `bert.modeling_bert` (HF transformers 3.0.2, absent from the reference tree) on liblavt_hip.

Same constructor / `from_pretrained` / `forward(input_ids, attention_mask=None, token_type_ids=None) -> (last_hidden_state,)` surface the
reference's call sites use (lib/_utils.py:38-52, train.py:216-218, 595-602: `BertModel.from_pretrained(args.ck_bert)`, `.pooler = None`,
`model(sentences, attention_mask=attentions)[0]`), and the same module tree, so a HF `pytorch_model.bin` (or the `bert_model` entry of a
LAVT checkpoint, train.py:612) loads key for key.  All arithmetic runs through lavt_hip.ops (GEMMs with fused bias / GELU / residual,
LayerNorm, the batched masked attention, the embedding-sum and dropout kernels of csrc/text.hip); nothing falls back to torch math.
"""
import json
import os

import torch
import torch.nn as nn

from lavt_hip import ops
from lavt_hip._capi import ACT_GELU
from lavt_hip.runtime import compute_dtype


class BertConfig:
    """bert-base-uncased defaults (transformers' `BertConfig`); `from_json_file` reads a HF config.json."""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, hidden_act="gelu",
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 pad_token_id=0, **unused):
        if hidden_act != "gelu":
            raise NotImplementedError(f"liblavt_hip BertModel: hidden_act={hidden_act!r} (only the erf GELU of bert-base is built)")
        if hidden_size % num_attention_heads:
            raise ValueError("hidden_size must be a multiple of num_attention_heads")
        self.vocab_size, self.hidden_size, self.num_hidden_layers = vocab_size, hidden_size, num_hidden_layers
        self.num_attention_heads, self.intermediate_size, self.hidden_act = num_attention_heads, intermediate_size, hidden_act
        self.hidden_dropout_prob, self.attention_probs_dropout_prob = hidden_dropout_prob, attention_probs_dropout_prob
        self.max_position_embeddings, self.type_vocab_size, self.layer_norm_eps = max_position_embeddings, type_vocab_size, layer_norm_eps
        self.pad_token_id = pad_token_id

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))


class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None):
        B, N = input_ids.shape
        x = ops.bert_embed(input_ids, token_type_ids, self.word_embeddings.weight, self.position_embeddings.weight,
                           self.token_type_embeddings.weight, N, compute_dtype())
        x = ops.layer_norm(x, self.LayerNorm.weight, self.LayerNorm.bias, eps=self.LayerNorm.eps)
        return ops.dropout(x, self.dropout.p, self.training)


class BertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.num_attention_heads = config.num_attention_heads
        self.query = nn.Linear(config.hidden_size, config.hidden_size)
        self.key = nn.Linear(config.hidden_size, config.hidden_size)
        self.value = nn.Linear(config.hidden_size, config.hidden_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def forward(self, x, keybias, B, N):
        qkv = ops.linear_cat(x, (self.query, self.key, self.value))          # one GEMM over the stacked [3H, H] weight
        return ops.masked_self_attention(qkv, keybias, B, N, self.num_attention_heads, self.dropout.p if self.training else 0.0)


class _DenseDropAddNorm(nn.Module):
    """BertSelfOutput / BertOutput: LayerNorm(dropout(dense(h)) + input)"""

    def __init__(self, in_features, config):
        super().__init__()
        self.dense = nn.Linear(in_features, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, h, inp):
        if self.training and self.dropout.p > 0.0:
            y = ops.dropout(ops.linear(h, self.dense.weight, self.dense.bias), self.dropout.p, True, residual=inp)
        else:
            y = ops.linear(h, self.dense.weight, self.dense.bias, residual=inp)
        return ops.layer_norm(y, self.LayerNorm.weight, self.LayerNorm.bias, eps=self.LayerNorm.eps)


class BertSelfOutput(_DenseDropAddNorm):
    def __init__(self, config):
        super().__init__(config.hidden_size, config)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, x, keybias, B, N):
        return self.output(self.self(x, keybias, B, N), x)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)

    def forward(self, x):
        return ops.linear(x, self.dense.weight, self.dense.bias, act=ACT_GELU)


class BertOutput(_DenseDropAddNorm):
    def __init__(self, config):
        super().__init__(config.intermediate_size, config)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, x, keybias, B, N):
        a = self.attention(x, keybias, B, N)
        return self.output(self.intermediate(a), a)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])


class BertPooler(nn.Module):
    """Present for state-dict compatibility; the reference sets `.pooler = None` and never evaluates it."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)


class BertModel(nn.Module):
    def __init__(self, config=None, add_pooling_layer=True):
        super().__init__()
        self.config = config if config is not None else BertConfig()
        self.embeddings = BertEmbeddings(self.config)
        self.encoder = BertEncoder(self.config)
        self.pooler = BertPooler(self.config) if add_pooling_layer else None

    @classmethod
    def from_pretrained(cls, path, *unused, **unused_kw):
        """`path`: a directory with config.json and pytorch_model.bin (what `args.ck_bert` points at).  There is no hub download."""
        cfg = os.path.join(str(path), "config.json")
        if not os.path.isfile(cfg):
            raise OSError(f"BertModel.from_pretrained: {path!r} is not a directory with config.json / pytorch_model.bin (no network download here)")
        model = cls(BertConfig.from_json_file(cfg))
        wfile = os.path.join(str(path), "pytorch_model.bin")
        sfile = os.path.join(str(path), "model.safetensors")
        if os.path.isfile(wfile):
            sd = torch.load(wfile, map_location="cpu", weights_only=True)
        elif os.path.isfile(sfile):
            from safetensors.torch import load_file
            sd = load_file(sfile)
        else:      # a config without weights would silently give an untrained text encoder
            raise OSError(f"BertModel.from_pretrained: {path!r} holds config.json but neither pytorch_model.bin nor model.safetensors")
        model.load_hf_state_dict(sd)
        return model

    def load_hf_state_dict(self, sd):
        """HF checkpoints prefix the encoder with `bert.` (BertForPreTraining), name LayerNorm parameters gamma / beta in old TF conversions and
        carry `embeddings.position_ids` (3.0.2) plus task heads (`cls.*`): normalise, then load strictly on what is left."""
        own = self.state_dict()
        clean = {}
        for k, v in sd.items():
            if k.startswith("bert."):
                k = k[5:]
            k = k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")
            if k in own:
                clean[k] = v
        missing = [k for k in own if k not in clean and not k.startswith("pooler.")]
        if missing:
            raise KeyError(f"BertModel: checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
        self.load_state_dict(clean, strict=False)

    def forward(self, input_ids, attention_mask=None, token_type_ids=None):
        B, N = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        keybias = (1.0 - attention_mask.reshape(B, N).to(torch.float32)) * -10000.0          # get_extended_attention_mask of 3.0.2
        x = self.embeddings(input_ids, token_type_ids)
        for layer in self.encoder.layer:
            x = layer(x, keybias, B, N)
        out = ops.cast_ad(x, torch.float32).view(B, N, self.config.hidden_size)
        return (out,)
