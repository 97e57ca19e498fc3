"""Attention-only model implementing the PRODUCT's model contract (easykv_amd/api.py): every layer calls
``past_key_values.attend(layer, q, k, v)``.  Same streams / one-hot logits as oracle/fake_model.py, so a run
through easykv_amd.generate is directly comparable with the reference's golden vectors.

With ``shard`` (easykv_amd.dist.LayerShard) the model is one stage of a layer-sharded pipeline: it runs only its own
block of layers and moves a real stage output — the running sum over layers of the attention outputs, ``[1, n, Hq*D]``
fp32 — to the next rank (easykv_amd.dist.PipelineStage), the way a decoder stack moves its hidden state."""
from types import SimpleNamespace

import torch

from oracle.fake_model import FakeTokenizer, one_hot_logits


class NativeFakeModel:
    def __init__(self, qs, ks, vs, device="cuda", arch="LlamaForCausalLM", shard=None, vocab=16):
        self.qs, self.ks, self.vs = qs.to(device).half(), ks.to(device).half(), vs.to(device).half()
        n_layers, hq, _, d = qs.shape
        self.config = SimpleNamespace(num_hidden_layers=n_layers, num_attention_heads=hq, num_key_value_heads=ks.shape[1],
                                      head_dim=d, architectures=[arch])
        self.device = torch.device(device)
        self.tokenizer = FakeTokenizer()
        self.outputs_log = []      # per forward: attention outputs of the layers THIS process ran [layers, Hq, n, D]
        self.hidden_log = []       # per forward: the stage output this process produced [1, n, Hq*D]
        self.layer_shard = shard
        self.vocab = vocab
        if shard is not None:
            from easykv_amd.dist import PipelineStage
            self.stage = PipelineStage(shard)

    def __call__(self, input_ids, past_key_values=None, position_ids=None, use_cache=True, **kw):
        pos = position_ids[0]
        n, (_, hq, _, d) = pos.numel(), self.qs.shape
        shard = self.layer_shard
        layers = range(self.config.num_hidden_layers) if shard is None else range(shard.begin, shard.end)
        hidden = torch.zeros(1, n, hq * d, dtype=torch.float32, device=self.device)
        if shard is not None:
            hidden = self.stage.recv_hidden(hidden)
        outs = []
        for l in layers:
            o = past_key_values.attend(l, self.qs[l][:, pos].unsqueeze(0), self.ks[l][:, pos].unsqueeze(0),
                                       self.vs[l][:, pos].unsqueeze(0))
            outs.append(o[0])
            hidden = hidden + o[0].transpose(0, 1).reshape(1, n, hq * d).float()
        if shard is not None:
            self.stage.send_hidden(hidden)
        self.outputs_log.append(torch.stack(outs).float().cpu())
        self.hidden_log.append(hidden.cpu())
        return SimpleNamespace(logits=one_hot_logits(pos.cpu(), self.vocab).to(self.device))
