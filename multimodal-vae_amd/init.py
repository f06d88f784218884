"""Default-PyTorch-style parameter initialisation (what ``MultimodalVAE(n_latents)`` of the reference gets from
nn.Linear / nn.Conv2d / nn.ConvTranspose2d / nn.GRU / nn.Embedding / nn.BatchNorm defaults), drawn on the host
from a seeded generator and copied into the flat device buffer.  Host-side setup, not on the hot path."""
import math

import torch


def default_init_(state, seed: int = 0) -> None:
    g = torch.Generator().manual_seed(seed)
    flat = torch.empty(state.nparams, dtype=torch.float32)
    bn_prefixes = {p for p, _, _ in state.bn_table}
    for name, shape, off in state.table:
        n = 1
        for s in shape:
            n *= s
        prefix = name.rsplit(".", 1)[0]
        if prefix in bn_prefixes:
            v = torch.ones(n) if name.endswith(".weight") else torch.zeros(n)
        elif name.endswith("embed.weight") or (name == "text_encoder.net.0.weight" and tuple(shape) == (10, 50)):
            v = torch.randn(n, generator=g)                                   # nn.Embedding: N(0,1)
        elif ".gru." in name:
            hidden = shape[0] // 3                                            # weight_* / bias_* rows = 3 * hidden
            v = (torch.rand(n, generator=g) * 2 - 1) / math.sqrt(float(hidden))   # nn.GRU: U(+-1/sqrt(hidden))
        else:
            if len(shape) == 4 and "hallucinate" in name:                     # ConvTranspose2d: fan_in = Cout*kh*kw
                fan_in = shape[1] * shape[2] * shape[3]
            elif len(shape) >= 2:
                fan_in = n // shape[0]
            else:                                                             # bias: bound from the matching weight
                wshape = next(s for nm, s, _ in state.table if nm == prefix + ".weight")
                fan_in = 1
                for s in wshape[1:]:
                    fan_in *= s
            v = (torch.rand(n, generator=g) * 2 - 1) / math.sqrt(fan_in)      # kaiming_uniform(a=sqrt(5))
        flat[off:off + n] = v
    state.params.copy_(flat.to(state.device))
    for _, c, o in state.bn_table:
        state.bn_stats[o:o + c] = 0.0
        state.bn_stats[o + c:o + 2 * c] = 1.0
    state.bn_nbt.zero_()
