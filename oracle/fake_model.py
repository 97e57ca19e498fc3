"""Duck-typed attention-only model implementing the contract the reference's ``generate``
needs from ``self`` (easykv/easykv.py:211-216, :232-234, :264-277).

TEST INFRASTRUCTURE ONLY (see oracle/easykv_oracle.py header).

Per layer the q/k/v of the token at TRUE position ``t`` are rows ``t`` of fixed streams, so
the inputs do not depend on what was evicted.  Logits are one-hot, which makes the
reference's multinomial sampler deterministic.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import easykv_oracle as O

VOCAB = 16


def make_streams(n_layers, hq, h, d, max_pos, seed):
    """fp16-representable N(0,1) streams, generator seed ``seed+layer``."""
    qs, ks, vs = [], [], []
    for l in range(n_layers):
        g = torch.Generator().manual_seed(seed + l)
        qs.append(torch.randn(hq, max_pos, d, generator=g).half())
        ks.append(torch.randn(h, max_pos, d, generator=g).half())
        vs.append(torch.randn(h, max_pos, d, generator=g).half())
    return torch.stack(qs), torch.stack(ks), torch.stack(vs)


class FakeTokenizer:
    eos_token_id = -1

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(i) for i in ids)

    def convert_ids_to_tokens(self, ids):
        return [str(i) for i in ids]


def one_hot_logits(positions, vocab=VOCAB):
    """The token predicted after TRUE position ``p`` is ``(p + 1) % vocab``: with a known prompt length the step at which a
    given token id (e.g. an EOS id) is sampled is known in advance."""
    n = len(positions)
    logits = torch.full((1, n, vocab), -1e4)
    for i, p in enumerate(positions):
        logits[0, i, (int(p) + 1) % vocab] = 0.0
    return logits


class FakeAttnModel:
    """``core(q, k_all, v_all, mask, layer) -> (o, p)`` is pluggable so the golden generator can
    route the attention through the reference's own ``llama_forward``."""

    def __init__(self, qs, ks, vs, arch="LlamaForCausalLM", streaming=False, dtype=torch.float32, core=None, vocab=VOCAB):
        self.qs, self.ks, self.vs = qs.to(dtype), ks.to(dtype), vs.to(dtype)
        self.vocab = vocab
        n_layers, hq, _, d = qs.shape
        h = ks.shape[1]
        self.config = SimpleNamespace(num_hidden_layers=n_layers, num_attention_heads=hq,
                                      num_key_value_heads=h, architectures=[arch])
        self.device = torch.device("cpu")
        self.tokenizer = FakeTokenizer()
        self.streaming = streaming
        self.dtype = dtype
        if streaming:
            self.cos, self.sin = O.rope_tables(qs.shape[2] + 8, d, dtype=dtype)
        self.core = core
        self.outputs_log = []          # per forward: [L,Hq,n,D]

    def _core(self, q, k, v, mask, layer):
        if self.core is not None:
            return self.core(self, q, k, v, mask, layer)
        if self.streaming:
            return O.attention_core_stream(q, k, v, self.cos, self.sin, mask)
        return O.attention_core(q, k, v, mask)

    def __call__(self, input_ids, past_key_values=None, attention_mask=None, position_ids=None,
                 use_cache=True, output_attentions=False):
        n = input_ids.shape[1]
        t_prev = 0 if past_key_values is None else past_key_values[0][0].shape[2]
        pos = torch.arange(t_prev, t_prev + n) if position_ids is None else position_ids[0].cpu()
        new_past, attns, outs = [], [], []
        for l in range(self.config.num_hidden_layers):
            q = self.qs[l][:, pos].unsqueeze(0)
            kn = self.ks[l][:, pos].unsqueeze(0)
            vn = self.vs[l][:, pos].unsqueeze(0)
            if past_key_values is None:
                k, v = kn, vn
            else:
                k = torch.cat((past_key_values[l][0], kn), dim=2)
                v = torch.cat((past_key_values[l][1], vn), dim=2)
            mask = O.causal_chunk_mask(n, k.shape[2], self.dtype)
            o, p = self._core(q, k, v, mask, l)
            new_past.append((k, v))
            outs.append(o[0])
            attns.append(p if output_attentions else None)
        self.outputs_log.append(torch.stack(outs))
        return SimpleNamespace(past_key_values=tuple(new_past) if use_cache else None,
                               logits=one_hot_logits(pos, self.vocab),
                               attentions=tuple(attns) if output_attentions else None)
