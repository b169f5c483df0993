"""Test helper: build the HIP-backed modules from a synthetic case and run them on cuda:0."""
import numpy as np
import torch

from oracle import cases


def to_dev(x, dev="cuda:0"):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def load_numpy_state(module, sd_np, dev="cuda:0"):
    sd = module.state_dict()
    sd_np = dict(sd_np)
    if "caption_encoder.embed.embedding.weight" in sd and "caption_encoder.embed.embedding.weight" not in sd_np:
        sd_np["caption_encoder.embed.embedding.weight"] = sd_np["embed.embedding.weight"]
    assert set(sd) == set(sd_np), (set(sd) ^ set(sd_np))
    module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    return module.to(dev).eval()


def editnet_modules(name, dev="cuda:0"):
    from show_edit_tell_amd import editnet, editnet_rl
    d = cases.build_editnet(name)
    c = d["case"]
    args = (d["wm"], c["D"], c["D"], c["D"], c["A"], c["F"])
    xe = load_numpy_state(editnet.DecoderC(*args), d["sd"], dev)
    rl = load_numpy_state(editnet_rl.DecoderC(*args), d["sd"], dev)
    return d, xe, rl


def dcnet_modules(name, dev="cuda:0"):
    from show_edit_tell_amd import dcnet, dcnet_rl
    d = cases.build_dcnet(name)
    c = d["case"]
    args = (d["wm"], None, c["D"], c["A"], c["C"], c["E"])
    xe = load_numpy_state(dcnet.DAE(*args), d["sd"], dev)
    rl = load_numpy_state(dcnet_rl.DAE(*args), d["sd"], dev)
    return d, xe, rl


def adaptive_module(name, dev="cuda:0"):
    from show_edit_tell_amd import editnet_adaptive
    d = cases.build_editnet(name)
    c = d["case"]
    args = (d["wm"], c["D"], c["D"], c["D"], c["A"], c["F"])
    return d, load_numpy_state(editnet_adaptive.DecoderC(*args), d["sd"], dev)
