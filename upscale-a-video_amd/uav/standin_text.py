"""Deterministic stand-in for the CLIP tokenizer + text encoder.

The pipeline cannot be called with `prompt_embeds` (the reference crashes in check_inputs,
pipeline_upscale_a_video.py:401-405), and no CLIP weights exist offline, so benchmarks and
parity tests register these two objects on the pipeline: the prompt STRING determines the
(1,77,dim) embedding through a CRC32-seeded generator.  The text encoder is outside the hot path.
"""
import types
import zlib

import torch


def prompt_embedding(prompt: str, dim: int, seq: int = 77, seed: int = 77) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(("prompt:" + prompt).encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return torch.randn((1, seq, dim), generator=g).half().float()


class StandInTokenizer:
    model_max_length = 77

    def __init__(self):
        self.prompts = []

    def __call__(self, prompt, padding=None, max_length=None, truncation=None, return_tensors=None):
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        ids = []
        for p in prompts:
            if p not in self.prompts:
                self.prompts.append(p)
            ids.append(torch.full((self.model_max_length,), self.prompts.index(p), dtype=torch.long))
        return types.SimpleNamespace(input_ids=torch.stack(ids), attention_mask=None)


class StandInTextEncoder(torch.nn.Module):
    def __init__(self, tokenizer: StandInTokenizer, dim: int, dtype=torch.float16):
        super().__init__()
        self.tok, self.dim, self._dtype = tokenizer, dim, dtype
        self.config = types.SimpleNamespace()
        self._dummy = torch.nn.Parameter(torch.zeros(1), requires_grad=False)

    @property
    def dtype(self):
        return self._dtype

    def forward(self, input_ids, attention_mask=None):
        rows = [prompt_embedding(self.tok.prompts[int(r[0])], self.dim) for r in input_ids.cpu()]
        return (torch.cat(rows).to(device=self._dummy.device, dtype=self._dtype),)
