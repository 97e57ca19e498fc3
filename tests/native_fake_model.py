"""Attention-only model implementing the PRODUCT's model contract (easykv_amd/api.py): every layer calls
``past_key_values.attend(layer, q, k, v)``.  Same streams / one-hot logits as oracle/fake_model.py, so a run
through easykv_amd.generate is directly comparable with the reference's golden vectors."""
from types import SimpleNamespace

import torch

from oracle.fake_model import FakeTokenizer, one_hot_logits


class NativeFakeModel:
    def __init__(self, qs, ks, vs, device="cuda", arch="LlamaForCausalLM"):
        self.qs, self.ks, self.vs = qs.to(device).half(), ks.to(device).half(), vs.to(device).half()
        n_layers, hq, _, d = qs.shape
        self.config = SimpleNamespace(num_hidden_layers=n_layers, num_attention_heads=hq, num_key_value_heads=ks.shape[1],
                                      head_dim=d, architectures=[arch])
        self.device = torch.device(device)
        self.tokenizer = FakeTokenizer()
        self.outputs_log = []

    def __call__(self, input_ids, past_key_values=None, position_ids=None, use_cache=True, **kw):
        pos = position_ids[0]
        outs = []
        for l in range(self.config.num_hidden_layers):
            o = past_key_values.attend(l, self.qs[l][:, pos].unsqueeze(0), self.ks[l][:, pos].unsqueeze(0),
                                       self.vs[l][:, pos].unsqueeze(0))
            outs.append(o[0])
        self.outputs_log.append(torch.stack(outs).float().cpu())
        return SimpleNamespace(logits=one_hot_logits(pos.cpu()).to(self.device))
