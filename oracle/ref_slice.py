"""ORACLE tooling (authoring container only): load the reference's model classes.

The reference's training scripts run their epoch loop at import (`editnet.py:743-848`), so they
cannot be imported.  This helper parses a reference file with `ast`, keeps only the `ClassDef`
nodes, and executes them in a namespace that provides the names those classes use.  Nothing is
copied into the repo: the source is read from /root/reference at run time, which exists only in
the authoring container.  Used by oracle/make_golden.py and by tests that are skipped when the
reference is absent.
"""
from __future__ import annotations

import ast
import math
import os

REF_ROOT = os.environ.get("SET_REFERENCE_ROOT", "/root/reference")


def have_reference() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "editnet.py"))


def load_classes(relpath: str, only=None):
    """Return {class_name: class} for the ClassDefs in REF_ROOT/relpath."""
    import numpy as np
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from torch.nn.utils.rnn import PackedSequence, pack_padded_sequence, pad_packed_sequence

    path = os.path.join(REF_ROOT, relpath)
    with open(path, "r") as f:
        tree = ast.parse(f.read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and (only is None or n.name in only)]
    mod = ast.Module(body=body, type_ignores=[])
    ns = dict(torch=torch, nn=nn, F=F, math=math, np=np, device=torch.device("cpu"),
              pack_padded_sequence=pack_padded_sequence, pad_packed_sequence=pad_packed_sequence,
              PackedSequence=PackedSequence, Dataset=object)
    exec(compile(mod, path, "exec"), ns)
    return {n.name: ns[n.name] for n in body}


MODEL_CLASSES = ("LSTMCellC", "CopyLSTMCellC", "EmbeddingC", "CaptionEncoderC", "CaptionAttentionC",
                 "SelectC", "VisualAttentionC", "DecoderC", "RewardCriterion")
DCNET_CLASSES = ("Embedding", "CaptionEncoder", "CaptionAttention", "DAE", "RewardCriterion")


def editnet_xe():
    return load_classes("editnet.py", MODEL_CLASSES)


def editnet_rl():
    return load_classes("editnet_rl.py", MODEL_CLASSES)


def editnet_adaptive():
    return load_classes("adaptive_features/editnet_adaptive.py", MODEL_CLASSES)


def dcnet_xe():
    return load_classes("dcnet.py", DCNET_CLASSES)


def dcnet_rl():
    return load_classes("dcnet_rl.py", DCNET_CLASSES)


def load_state(module, sd_np):
    """Copy a numpy state dict (our synth layout == reference state_dict names) into a module."""
    import torch
    sd = module.state_dict()
    sd_np = dict(sd_np)
    # the reference registers the shared embedding twice (DecoderC.embed and caption_encoder.embed)
    if "caption_encoder.embed.embedding.weight" in sd and "caption_encoder.embed.embedding.weight" not in sd_np:
        sd_np["caption_encoder.embed.embedding.weight"] = sd_np["embed.embedding.weight"]
    missing = set(sd) - set(sd_np)
    extra = set(sd_np) - set(sd)
    assert not missing and not extra, (missing, extra)
    module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    return module
