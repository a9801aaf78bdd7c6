"""BertForPreTraining parameter container with the reference's module tree / state_dict names.

Mirrors LanguageModeling/BERT/modeling.py: BertEmbeddings :263-301, BertSelfAttention :304-384, BertSelfOutput
:387-398, BertIntermediate/LinearActivation :130-165,408-415, BertOutput :418-434, BertPooler :512-524,
BertPreTrainingHeads :545-595, BertForPreTraining :890-958 (decoder weight tied to the word embeddings).
The modules only own parameters (checkpoint compatible); the compute lives in engine.py.
"""
import json

import torch
from torch import nn

LARGE = dict(hidden=1024, heads=16, layers=24, intermediate=4096, vocab=30528, real_vocab=30522, max_pos=512,
             type_vocab=2, seq=128)
"""bert_configs/large.json with the vocabulary padded to a multiple of 8 (run_pretraining.py:380-384)."""


def config_from_json(path):
    c = json.load(open(path))
    v = c["vocab_size"]
    return dict(hidden=c["hidden_size"], heads=c["num_attention_heads"], layers=c["num_hidden_layers"],
                intermediate=c["intermediate_size"], vocab=(v + 7) // 8 * 8, real_vocab=v,
                max_pos=c["max_position_embeddings"], type_vocab=c["type_vocab_size"], seq=128,
                hidden_dropout=c.get("hidden_dropout_prob", 0.1), attention_dropout=c.get("attention_probs_dropout_prob", 0.1))


class _LinearAct(nn.Module):          # LinearActivation: weight [out, in] + bias
    def __init__(self, i, o, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i, device=device))
        self.bias = nn.Parameter(torch.zeros(o, device=device))


class _SelfAttention(nn.Module):
    def __init__(self, h, device):
        super().__init__()
        self.query, self.key, self.value = (nn.Linear(h, h, device=device) for _ in range(3))


class _SelfOutput(nn.Module):
    def __init__(self, i, h, device):
        super().__init__()
        self.dense = nn.Linear(i, h, device=device)
        self.LayerNorm = nn.LayerNorm(h, eps=1e-12, device=device)


class _Attention(nn.Module):
    def __init__(self, h, device):
        super().__init__()
        self.self = _SelfAttention(h, device)
        self.output = _SelfOutput(h, h, device)


class _Intermediate(nn.Module):
    def __init__(self, h, i, device):
        super().__init__()
        self.dense_act = _LinearAct(h, i, device)


class _Layer(nn.Module):
    def __init__(self, h, i, device):
        super().__init__()
        self.attention = _Attention(h, device)
        self.intermediate = _Intermediate(h, i, device)
        self.output = _SelfOutput(i, h, device)


class _Encoder(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg["hidden"], cfg["intermediate"], device) for _ in range(cfg["layers"])])


class _Embeddings(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        h = cfg["hidden"]
        self.word_embeddings = nn.Embedding(cfg["vocab"], h, device=device)
        self.position_embeddings = nn.Embedding(cfg["max_pos"], h, device=device)
        self.token_type_embeddings = nn.Embedding(cfg["type_vocab"], h, device=device)
        self.LayerNorm = nn.LayerNorm(h, eps=1e-12, device=device)


class _Pooler(nn.Module):
    def __init__(self, h, device):
        super().__init__()
        self.dense_act = _LinearAct(h, h, device)


class _BertModel(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        self.embeddings = _Embeddings(cfg, device)
        self.encoder = _Encoder(cfg, device)
        self.pooler = _Pooler(cfg["hidden"], device)


class _Transform(nn.Module):
    def __init__(self, h, device):
        super().__init__()
        self.dense_act = _LinearAct(h, h, device)
        self.LayerNorm = nn.LayerNorm(h, eps=1e-12, device=device)


class _Predictions(nn.Module):
    def __init__(self, cfg, word_weight, device):
        super().__init__()
        self.transform = _Transform(cfg["hidden"], device)
        self.decoder = nn.Linear(cfg["hidden"], cfg["vocab"], bias=False, device=device)
        self.decoder.weight = word_weight                       # tied (modeling.py:545-549)
        self.bias = nn.Parameter(torch.zeros(cfg["vocab"], device=device))


class _Heads(nn.Module):
    def __init__(self, cfg, word_weight, device):
        super().__init__()
        self.predictions = _Predictions(cfg, word_weight, device)
        self.seq_relationship = nn.Linear(cfg["hidden"], 2, device=device)


class BertForPreTraining(nn.Module):
    def __init__(self, cfg, device="cuda", initializer_range=0.02):
        super().__init__()
        self.config = dict(cfg)
        self.bert = _BertModel(cfg, device)
        self.cls = _Heads(cfg, self.bert.embeddings.word_embeddings.weight, device)
        for m in self.modules():                                 # init_bert_weights (modeling.py:705-720)
            if isinstance(m, (nn.Linear, nn.Embedding, _LinearAct)):
                m.weight.data.normal_(mean=0.0, std=initializer_range)
            if isinstance(m, nn.LayerNorm):
                m.bias.data.zero_(); m.weight.data.fill_(1.0)
            if isinstance(m, (nn.Linear, _LinearAct)) and m.bias is not None:
                m.bias.data.zero_()

    def fuse_qkv_storage(self):
        """Re-home query/key/value weights and biases of every layer in ONE [3H, H] / [3H] fp32 tensor (the
        parameters become views), so the fused QKV GEMM, its gradient and LAMB all see contiguous memory while the
        state_dict keeps the reference's three separate entries."""
        h = self.config["hidden"]
        for layer in self.bert.encoder.layer:
            sa = layer.attention.self
            if getattr(layer, "qkv_weight", None) is not None:
                continue
            w = torch.cat([sa.query.weight.data, sa.key.weight.data, sa.value.weight.data], 0).contiguous()
            b = torch.cat([sa.query.bias.data, sa.key.bias.data, sa.value.bias.data], 0).contiguous()
            for i, lin in enumerate((sa.query, sa.key, sa.value)):
                lin.weight.data = w[i * h:(i + 1) * h]
                lin.bias.data = b[i * h:(i + 1) * h]
            layer.qkv_weight, layer.qkv_bias = w, b
